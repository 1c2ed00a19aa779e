cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp10
O=$GRAFT_REPO_ROOT/gpurun_out/exp10
(timeout 300 python scripts/sba_config5.py f64 10) > $O/sba_f64.log 2>&1
(timeout 300 python scripts/e2e_phases.py) > $O/e2e.log 2>&1
(timeout 2400 python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.log 2>&1
export TMPDIR=/tmp; cd /tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_sba -o sba -- python $GRAFT_REPO_ROOT/scripts/sba_config5.py f64 10) > $O/sba_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/rocprof_sba -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/sba_kernel_stats.csv
rm -rf $O/rocprof_sba
grep -v amdgpu.ids $O/sba_f64.log | cut -c1-120; grep -v amdgpu $O/e2e.log; tail -n 12 $O/pytest_gpu.log; cut -c1-150 $O/sba_kernel_stats.csv | head -7
