# round 5: sweep: the builders barrier for the coupling tables only where tables were filled: 245.0 us, flat (the waves wait for the last tile of G anyway) - not kept
O=gpurun_out/exp59; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chunk.py -m gpu -x -q > $O/chunk_tests.log 2>&1; echo "rc=$?" >> $O/chunk_tests.log; tail -3 $O/chunk_tests.log
timeout 300 python scripts/solver_sweep.py 10000 "0,2,3" "0,2,3" > $O/solver.log 2>&1
grep "us/step" $O/solver.log | cut -c1-250
