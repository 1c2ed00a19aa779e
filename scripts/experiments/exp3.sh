cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp3
O=gpurun_out/exp3
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
tail -n 30 $O/pytest_skel.log; cat $O/skel_time.log
