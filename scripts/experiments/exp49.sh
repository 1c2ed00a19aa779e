# round 5: back-substitution: four loaders round-robin over the steps again; uniform wave index, prologue in the product waves, vector signalled before the trial row, rows 75..79 not fetched
O=gpurun_out/exp49; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chunk.py -m gpu -x -q > $O/chunk_tests.log 2>&1; echo "rc=$?" >> $O/chunk_tests.log; tail -5 $O/chunk_tests.log
timeout 300 python scripts/solver_sweep.py 10000 "0,2,3" "0,2,3" > $O/solver.log 2>&1
grep "us/step by kernel" $O/solver.log | tail -1; grep -o "cost23=[-0-9.]*" $O/solver.log | tail -1
for wg in 100 0 238; do timeout 120 python scripts/backsub_stamps.py $wg 2>&1 | grep -v amdgpu.ids > $O/stamps_$wg.log; done
head -4 $O/stamps_100.log; tail -3 $O/stamps_100.log; head -3 $O/stamps_0.log;  tail -3 $O/stamps_0.log; head -3 $O/stamps_238.log; tail -3 $O/stamps_238.log
