# round 5: per-step stamps of the new back-substitution (is a step waiting for its stage or for its vector?)
O=gpurun_out/exp46; mkdir -p $O
for wg in 100 0 255 7; do timeout 120 python scripts/backsub_stamps.py $wg 2>&1 | grep -v amdgpu.ids > $O/stamps_$wg.log; done
cat $O/stamps_100.log; tail -8 $O/stamps_0.log; tail -4 $O/stamps_255.log; tail -4 $O/stamps_7.log
