cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp13
O=$GRAFT_REPO_ROOT/gpurun_out/exp13
for wg in 100 17; do (timeout 120 python scripts/sweep_stamps.py $wg 5) 2>&1 | grep -v amdgpu | tail -6 >> $O/clock.log; done
(timeout 600 python scripts/extras_report.py $O) > $O/extras.log 2>&1
cat $O/clock.log; tail -4 $O/extras.log | cut -c1-400
