# round 5: the separator chain s narrow-level eliminations factored with the sweep s three-wave trio instead of chol80: 68.5 us for the three launches either way (four waves on four SIMDs were never the problem) - not kept
O=gpurun_out/exp56; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chunk.py -m gpu -x -q > $O/chunk_tests.log 2>&1; echo "rc=$?" >> $O/chunk_tests.log; tail -3 $O/chunk_tests.log
timeout 300 python scripts/solver_sweep.py 10000 "0,2,3" "0,2,3" > $O/solver.log 2>&1
grep "us/step" $O/solver.log | cut -c1-250
