cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp1
(python scripts/solver_sweep.py 10000 "0,2,3" "0,2,3") > gpurun_out/exp1/base.log 2>&1
(ACINO_SWEEP_REGACC=1 python scripts/solver_sweep.py 10000 "0,2,3" "0,2,3") > gpurun_out/exp1/regacc.log 2>&1
(timeout 900 python -m pytest tests/test_gpu_chunk.py tests/test_gpu_parity.py -m gpu -x -q -k "cost_gradient or chunk or escalation or matches_oracle_traj or bf16_rows_assembly") > gpurun_out/exp1/pytest_a.log 2>&1
(ACINO_SWEEP_REGACC=1 timeout 900 python -m pytest tests/test_gpu_chunk.py -m gpu -q) > gpurun_out/exp1/pytest_regacc.log 2>&1
tail -5 gpurun_out/exp1/*.log
