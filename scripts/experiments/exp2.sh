cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp2
O=gpurun_out/exp2
(timeout 300 python scripts/solver_sweep.py 10000 "0,2,3" "0,2,3") > $O/new.log 2>&1
(ACINO_NO_SEP_TAIL=1 timeout 300 python scripts/solver_sweep.py 10000 "0,2,3") > $O/no_tail.log 2>&1
(ACINO_SEP_COMBINE=1 timeout 300 python scripts/solver_sweep.py 10000 "0,2,3") > $O/combine.log 2>&1
(timeout 300 python scripts/solver_sweep.py 10000 "0,1,8" "0,1,10" "0,3,2" "0,3,3") > $O/variants.log 2>&1
(timeout 1500 python -m pytest tests/test_gpu_chunk.py -m gpu -x -q -s) > $O/pytest_chunk.log 2>&1
(timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "incomplete or escalation or window or concurrent or full_size or clips or size_sweep or config3 or randomised") > $O/pytest_par.log 2>&1
tail -n 6 $O/new.log $O/no_tail.log $O/combine.log $O/variants.log; tail -n 15 $O/pytest_chunk.log; tail -n 15 $O/pytest_par.log
