# round 5: the T = G F form of the sweep (k_chunk_sweep2) against the round-4 kernel - correctness, then time and stamps
O=gpurun_out/exp26; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chunk.py -m gpu -x -q > $O/chunk_tests.log 2>&1; echo "rc=$?" >> $O/chunk_tests.log; tail -5 $O/chunk_tests.log
for v in 1 2; do
  ACINO_SWEEP=$v timeout 300 python scripts/solver_sweep.py 10000 "0,2,3" "0,2,3" > $O/solver_v$v.log 2>&1
  echo "== variant $v: $(grep -o "'chunk_sweep': [0-9.]*" $O/solver_v$v.log | tr '\n' ' ') $(grep -o "[0-9.]* us/step" $O/solver_v$v.log | tr '\n' ' ') $(grep -o "cost23=[-0-9.]*" $O/solver_v$v.log | tail -1)"
  ACINO_SWEEP=$v timeout 120 python scripts/sweep_stamps.py 100 3 > $O/stamps_v$v.log 2>&1
done
cat $O/stamps_v2.log
