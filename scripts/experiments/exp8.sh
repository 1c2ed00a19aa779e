cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp8
O=$GRAFT_REPO_ROOT/gpurun_out/exp8
(timeout 600 python -m pytest tests/test_gpu_sba.py -m gpu -x -q) > $O/pytest_sba.log 2>&1
(timeout 300 python scripts/sba_config5.py f64 10) > $O/sba_f64.log 2>&1
(timeout 300 python scripts/e2e_phases.py) > $O/e2e.log 2>&1
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "device_side or escalation or stationary or config3 or matches_oracle") > $O/pytest_par.log 2>&1
for wg in 100 17; do for k in 3 9; do (timeout 120 python scripts/sweep_stamps.py $wg $k) >> $O/stamps.log 2>&1; echo ---- >> $O/stamps.log; done; done
export TMPDIR=/tmp; cd /tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_sba -o sba -- python $GRAFT_REPO_ROOT/scripts/sba_config5.py f64 10) > $O/sba_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/rocprof_sba -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/sba_kernel_stats.csv
rm -rf $O/rocprof_sba
tail -n 3 $O/pytest_sba.log; grep -v amdgpu.ids $O/sba_f64.log | cut -c1-120; grep -v amdgpu $O/e2e.log; tail -n 5 $O/pytest_par.log; cut -c1-150 $O/sba_kernel_stats.csv | head -6; grep -v amdgpu $O/stamps.log | head -70
