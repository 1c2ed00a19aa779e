mkdir -p gpurun_out/exp16
O=gpurun_out/exp16
timeout 900 python -m pytest tests/test_gpu_sba.py -x -q -m gpu > $O/pytest_sba.log 2>&1; tail -15 $O/pytest_sba.log
(timeout 300 python scripts/sba_config5.py f64 10) > $O/sba_f64.log 2>&1; tail -4 $O/sba_f64.log
(timeout 300 python scripts/sba_config5.py bf16 10) > $O/sba_bf16.log 2>&1; tail -4 $O/sba_bf16.log
(ACINO_SBA_UNFUSED=1 timeout 300 python scripts/sba_config5.py f64 10) > $O/sba_f64_unfused.log 2>&1; tail -4 $O/sba_f64_unfused.log
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o sba -- python $GRAFT_REPO_ROOT/scripts/sba_config5.py f64 10 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200; cp "$f" $O/sba_kernel_stats.csv; rm -rf $O/prof
