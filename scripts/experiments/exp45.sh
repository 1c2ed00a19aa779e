# round 5: back-substitution as loader waves + product waves (LDS counters, no barrier in the step loop)
O=gpurun_out/exp45; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chunk.py -m gpu -x -q > $O/chunk_tests.log 2>&1; echo "rc=$?" >> $O/chunk_tests.log; tail -5 $O/chunk_tests.log
timeout 300 python scripts/solver_sweep.py 10000 "0,2,3" "0,2,3" > $O/solver.log 2>&1
echo "== $(grep -o "'chunk_backsub': [0-9.]*" $O/solver.log | tr '\n' ' ') $(grep -o "[0-9.]* us/step" $O/solver.log | tr '\n' ' ') $(grep -o "cost23=[-0-9.]*" $O/solver.log | tail -1)"
grep "us/step by kernel" $O/solver.log | tail -1
