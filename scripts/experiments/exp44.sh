# round 5: back-substitution with one barrier per product, register ring of tiles inside every wave (66 us: the compiler waits vmcnt(0..1))
O=gpurun_out/exp44; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chunk.py -m gpu -x -q > $O/chunk_tests.log 2>&1; echo "rc=$?" >> $O/chunk_tests.log; tail -5 $O/chunk_tests.log
timeout 300 python scripts/solver_sweep.py 10000 "0,2,3" "0,2,3" > $O/solver.log 2>&1
grep "us/step by kernel" $O/solver.log | tail -1
