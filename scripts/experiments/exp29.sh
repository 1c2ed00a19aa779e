# round 5: the two-team sweep (k_chunk_sweep3) - correctness first, then time with / without yielding, stamps
O=gpurun_out/exp29; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chunk.py -m gpu -x -q > $O/chunk_tests.log 2>&1; echo "rc=$?" >> $O/chunk_tests.log; tail -5 $O/chunk_tests.log
for v in "2 1" "3 1" "3 0"; do
  set -- $v
  ACINO_SWEEP=$1 ACINO_SWEEP_YIELD=$2 timeout 300 python scripts/solver_sweep.py 10000 "0,2,3" "0,2,3" > $O/solver_v$1_y$2.log 2>&1
  echo "== variant $1 yield $2: $(grep -o "'chunk_sweep': [0-9.]*" $O/solver_v$1_y$2.log | tr '\n' ' ') $(grep -o "[0-9.]* us/step" $O/solver_v$1_y$2.log | tr '\n' ' ') $(grep -o "cost23=[-0-9.]*" $O/solver_v$1_y$2.log | tail -1)"
done
ACINO_SWEEP=3 timeout 120 python scripts/sweep_stamps.py 100 3 > $O/stamps_v3.log 2>&1; grep -v amdgpu.ids $O/stamps_v3.log
ACINO_SWEEP=3 ACINO_SWEEP_YIELD=0 timeout 120 python scripts/sweep_stamps.py 100 3 > $O/stamps_v3_noyield.log 2>&1; grep -v amdgpu.ids $O/stamps_v3_noyield.log
