cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp6
O=$GRAFT_REPO_ROOT/gpurun_out/exp6
(timeout 1200 python -m pytest tests/test_gpu_sba.py -m gpu -x -q) > $O/pytest_sba.log 2>&1
(timeout 300 python scripts/sba_config5.py f64 10) > $O/sba_f64.log 2>&1
(timeout 300 python scripts/sba_config5.py bf16 10) > $O/sba_bf16.log 2>&1
(ACINO_SBA_SCHUR_ATOMICS=1 timeout 300 python scripts/sba_config5.py f64 10) > $O/sba_f64_atomics.log 2>&1
export TMPDIR=/tmp; cd /tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_sba -o sba -- python $GRAFT_REPO_ROOT/scripts/sba_config5.py f64 10) > $O/sba_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/rocprof_sba -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/sba_kernel_stats.csv
rm -rf $O/rocprof_sba
tail -n 8 $O/pytest_sba.log; grep -v amdgpu.ids $O/sba_f64.log $O/sba_bf16.log $O/sba_f64_atomics.log; cut -c1-150 $O/sba_kernel_stats.csv | head -9
