# round 5: the cleaned-up sweep (one kernel): chunk tests, solver timing, stamps, then the whole GPU suite
O=gpurun_out/exp43; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chunk.py -m gpu -x -q > $O/chunk_tests.log 2>&1; echo "rc=$?" >> $O/chunk_tests.log; tail -3 $O/chunk_tests.log
timeout 300 python scripts/solver_sweep.py 10000 "0,2,3" "0,2,3" > $O/solver.log 2>&1
echo "== $(grep -o "'chunk_sweep': [0-9.]*" $O/solver.log | tr '\n' ' ') $(grep -o "[0-9.]* us/step" $O/solver.log | tr '\n' ' ') $(grep -o "cost23=[-0-9.]*" $O/solver.log | tail -1)"
grep "us/step by kernel" $O/solver.log | tail -1
timeout 120 python scripts/sweep_stamps.py 100 3 > $O/stamps.log 2>&1; grep -v amdgpu.ids $O/stamps.log | tail -50
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log
