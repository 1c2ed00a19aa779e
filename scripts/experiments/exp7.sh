cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp7
O=$GRAFT_REPO_ROOT/gpurun_out/exp7
(timeout 1200 python -m pytest tests/test_gpu_sba.py -m gpu -x -q) > $O/pytest_sba.log 2>&1
(timeout 300 python scripts/sba_config5.py f64 10) > $O/sba_f64.log 2>&1
(timeout 300 python scripts/sba_config5.py bf16 10) > $O/sba_bf16.log 2>&1
(timeout 300 python scripts/e2e_phases.py) > $O/e2e.log 2>&1
(timeout 300 python scripts/solver_sweep.py 10000 "0,2,3" "0,2,3") > $O/sweep.log 2>&1
(timeout 1500 python -m pytest tests/test_gpu_chunk.py -m gpu -x -q) > $O/pytest_chunk.log 2>&1
export TMPDIR=/tmp; cd /tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_sba -o sba -- python $GRAFT_REPO_ROOT/scripts/sba_config5.py f64 10) > $O/sba_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/rocprof_sba -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/sba_kernel_stats.csv
rm -rf $O/rocprof_sba
tail -n 3 $O/pytest_sba.log; grep -v amdgpu.ids $O/sba_f64.log $O/sba_bf16.log | cut -c1-200; grep -v amdgpu $O/e2e.log; grep -v amdgpu $O/sweep.log; tail -n 3 $O/pytest_chunk.log; cut -c1-150 $O/sba_kernel_stats.csv | head -8
