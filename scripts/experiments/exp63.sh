# round 5: separator-chain kernels (elim_deep, update_deep, sep_tail): the status word tested behind the loads of the schedule entry: flat (68.9 / 35.0 / 35.4 us) - not kept
O=gpurun_out/exp63; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chunk.py -m gpu -x -q > $O/chunk_tests.log 2>&1; echo "rc=$?" >> $O/chunk_tests.log; tail -3 $O/chunk_tests.log
timeout 300 python scripts/solver_sweep.py 10000 "0,2,3" "0,2,3" > $O/solver.log 2>&1
grep "us/step" $O/solver.log | cut -c1-250
