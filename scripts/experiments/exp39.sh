# round 5: two-team sweep, next node from the 325 unordered state pairs
O=gpurun_out/exp39; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chunk.py -m gpu -x -q > $O/chunk_tests.log 2>&1; echo "rc=$?" >> $O/chunk_tests.log; tail -4 $O/chunk_tests.log
for v in 1 3; do
  ACINO_SWEEP=$v timeout 300 python scripts/solver_sweep.py 10000 "0,2,3" "0,2,3" > $O/solver_$v.log 2>&1
  echo "== variant $v: $(grep -o "'chunk_sweep': [0-9.]*" $O/solver_$v.log | tr '\n' ' ') $(grep -o "[0-9.]* us/step" $O/solver_$v.log | tr '\n' ' ') $(grep -o "cost23=[-0-9.]*" $O/solver_$v.log | tail -1)"
done
ACINO_SWEEP=3 timeout 120 python scripts/sweep_stamps.py 100 3 > $O/stamps_v3.log 2>&1; grep -v amdgpu.ids $O/stamps_v3.log
