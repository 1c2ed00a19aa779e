cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp4
O=$GRAFT_REPO_ROOT/gpurun_out/exp4
(timeout 900 python -m pytest tests/test_skel_fte.py -m gpu -x -q) > $O/pytest_skel.log 2>&1
(timeout 300 python - <<'P'
import numpy as np, json, os, time
from acinoset_amd import build
g=np.load('tests/golden/skel_fte_model.npz'); sk=json.loads(str(g['skeleton_json']))
det=np.load('tests/golden/human_dlc_slice.npz')['det'].astype(np.float64)
tabs=[(list(g['parts']),det[:,c]) for c in range(2)]
for n in (100,400):
    model,_=build.build_model(sk,scene=(g['K'],g['D'],g['R'],g['t']),dlc_tables=tabs,n_frames=n,start_frame=60,pairing='name')
    for rep in range(2):
        t=time.perf_counter(); res,info=build.solve_model(model,max_iter=300); dt=time.perf_counter()-t
        print(n,'frames',info,'%.1f ms, %.3f ms/iter'%(1e3*dt,1e3*dt/max(info['iterations'],1)),flush=True)
P
) > $O/skel_time.log 2>&1
export TMPDIR=/tmp; cd /tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_sba -o sba -- python $GRAFT_REPO_ROOT/scripts/sba_config5.py f64 10) > $O/sba_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/rocprof_sba -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/sba_kernel_stats.csv
find $O/rocprof_sba -type f ! -name "*kernel_stats.csv" -delete
tail -n 12 $O/pytest_skel.log; cat $O/skel_time.log; tail -3 $O/sba_prof.log; cut -c1-150 $O/sba_kernel_stats.csv | head -14
