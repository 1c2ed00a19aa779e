# round 5: run length / reduction levels for SHORT chains (config 3's 1 000 frames, a window rank's 1 442): is the automatic plan (runs of 4) the best?
O=gpurun_out/exp62; mkdir -p $O
for n in 1000 1442 2692; do
  echo "== $n frames"
  timeout 300 python scripts/solver_sweep.py $n "0,0,0" "4,4,3" "4,3,3" "6,3,3" "8,3,3" "8,2,3" "10,2,3" "14,2,3" "14,1,3" 2>&1 | grep "us/step" | grep -v "by kernel" | cut -c1-200
done | tee $O/plans.log
