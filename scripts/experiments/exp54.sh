# round 5: H stored as the 325 unordered state pairs per frame (the order the sweep builds a node in): whole GPU suite, solver timing
O=gpurun_out/exp54; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 300 python scripts/solver_sweep.py 10000 "0,2,3" "0,2,3" > $O/solver.log 2>&1
grep "us/step" $O/solver.log | cut -c1-250
