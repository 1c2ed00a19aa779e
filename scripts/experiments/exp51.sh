# round 5: back-substitution: loader signals before it requests the next job, row operands requested a job ahead; product waves do a step s stores and trial row under the reads of step s + 1
O=gpurun_out/exp51; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chunk.py -m gpu -x -q > $O/chunk_tests.log 2>&1; echo "rc=$?" >> $O/chunk_tests.log; tail -5 $O/chunk_tests.log
timeout 300 python scripts/solver_sweep.py 10000 "0,2,3" "0,2,3" > $O/solver.log 2>&1
grep "us/step by kernel" $O/solver.log | tail -1; grep -o "cost23=[-0-9.]*" $O/solver.log | tail -1
timeout 120 python scripts/backsub_stamps.py 100 2>&1 | grep -v amdgpu.ids > $O/stamps_100.log; cat $O/stamps_100.log
