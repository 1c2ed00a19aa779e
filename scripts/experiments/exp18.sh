mkdir -p gpurun_out/exp18
O=gpurun_out/exp18
(timeout 300 python scripts/sba_config5.py f64 10) > $O/occ2.log 2>&1; tail -1 $O/occ2.log | cut -c1-60
sed -i 's/#define FU_OCC 2/#define FU_OCC 1/' acinoset_amd/csrc/sba.hip
python -c "from acinoset_amd import _lib; _lib.build(force=True, verbose=True)" > $O/build.log 2>&1
(timeout 300 python scripts/sba_config5.py f64 10) > $O/occ1.log 2>&1; tail -1 $O/occ1.log | cut -c1-60
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o sba -- python $GRAFT_REPO_ROOT/scripts/sba_config5.py f64 10 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -4 "$f" | cut -c1-160; rm -rf $O/prof
