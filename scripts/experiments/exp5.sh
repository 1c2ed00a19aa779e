mkdir -p $GRAFT_REPO_ROOT/gpurun_out/exp5
O=$GRAFT_REPO_ROOT/gpurun_out/exp5
export TMPDIR=/tmp; cd /tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_sba -o sba -- python $GRAFT_REPO_ROOT/scripts/sba_config5.py f64 10) > $O/sba_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/rocprof_sba -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/sba_kernel_stats.csv
find $O/rocprof_sba -type f ! -name "*kernel_stats.csv" -delete
grep -v "^W2026" $O/sba_prof.log | tail -5; cut -c1-170 $O/sba_kernel_stats.csv | head -14
