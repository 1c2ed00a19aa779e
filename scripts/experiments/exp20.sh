mkdir -p gpurun_out/exp20
O=gpurun_out/exp20
timeout 1500 python -m pytest tests/test_gpu_chunk.py -x -q -m gpu > $O/pytest_chunk.log 2>&1; tail -3 $O/pytest_chunk.log
timeout 600 python scripts/solver_sweep.py 10000 "0,2,3" > $O/solver.log 2>&1; tail -4 $O/solver.log | cut -c1-330
