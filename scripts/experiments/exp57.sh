# round 5: sweep: helper U s products of the factor s block column 0 done by two strip waves after their T = G F (helper U starts at block column 1): the chain s last pivots 15.4 -> 14.4 us, but the strip waves take > 1 us for the eight products beside their streaming SIMD mates, helper L still starts its second trailing step at 10.0 us and helper U now ends at 17.1: sweep 246 -> 261 us. Not kept.
O=gpurun_out/exp57; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chunk.py -m gpu -x -q > $O/chunk_tests.log 2>&1; echo "rc=$?" >> $O/chunk_tests.log; tail -3 $O/chunk_tests.log
timeout 300 python scripts/solver_sweep.py 10000 "0,2,3" "0,2,3" > $O/solver.log 2>&1
grep "us/step" $O/solver.log | cut -c1-250
timeout 120 python scripts/sweep_stamps.py 100 3 > $O/stamps.log 2>&1; grep -v amdgpu.ids $O/stamps.log | tail -52
