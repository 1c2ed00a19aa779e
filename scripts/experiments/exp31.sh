# round 5: two-team sweep with the relaxed factor-trio protocol (chol80_trio2)
O=gpurun_out/exp31; mkdir -p $O
ACINO_SWEEP_TRIO=2 timeout 900 python -m pytest tests/test_gpu_chunk.py -m gpu -x -q > $O/chunk_tests.log 2>&1; echo "rc=$?" >> $O/chunk_tests.log; tail -4 $O/chunk_tests.log
for v in "2 0 1" "3 0 1" "3 0 2"; do
  set -- $v
  ACINO_SWEEP=$1 ACINO_SWEEP_YIELD=$2 ACINO_SWEEP_TRIO=$3 timeout 300 python scripts/solver_sweep.py 10000 "0,2,3" "0,2,3" > $O/solver_$1_$2_$3.log 2>&1
  echo "== variant $1 yield $2 trio $3: $(grep -o "'chunk_sweep': [0-9.]*" $O/solver_$1_$2_$3.log | tr '\n' ' ') $(grep -o "[0-9.]* us/step" $O/solver_$1_$2_$3.log | tr '\n' ' ') $(grep -o "cost23=[-0-9.]*" $O/solver_$1_$2_$3.log | tail -1)"
done
ACINO_SWEEP=3 ACINO_SWEEP_TRIO=2 timeout 120 python scripts/sweep_stamps.py 100 3 > $O/stamps_v3.log 2>&1; grep -v amdgpu.ids $O/stamps_v3.log
