"""The widened rows as one stand-alone workload for rocprofv3 (kernel stats and counters of k_ekf_*, k_skel_*, k_triangulate_*):
the EKF + RTS smoother on 64 clips x 1 000 frames, config 2's fused pairwise triangulation + reprojection at 10 000 frames, the
generic-skeleton FTE on 16 windows of the shipped detections (20 iterations).  python scripts/extras_workload.py"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acinoset_amd import build, calib, ekf, synth
seq = synth.make_sequence(1000, "walk")
rig = (seq["K"], seq["D"], seq["R"], seq["t"])
dets = [torch.as_tensor(seq["det"], device="cuda")] * 64
for _ in range(2):
    ekf.ekf_batch(dets, *rig, 1.0 / seq["Ts"], 0.5, (2704, 1520), with_positions=False)
seq = synth.make_sequence(10000, "loop")
rig = (seq["K"], seq["D"], seq["R"], seq["t"])
d = torch.as_tensor(seq["det"], device="cuda")
for _ in range(3):
    calib.triangulate_pairs_dense(d, 0.5, *rig, return_masks=False)
    calib.triangulate_reproject_dense(d, 0.5, *rig)
gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
gsk = np.load(os.path.join(gd, "skel_fte_model.npz"))
sk = json.loads(str(gsk["skeleton_json"]))
full = np.load(os.path.join(gd, "human_dlc_full.npz"))
tabs = [(list(full["parts"]), full[f"det{c}"].astype(np.float64)) for c in range(2)]
scene = (gsk["K"], gsk["D"], gsk["R"], gsk["t"])
models = [build.build_model(sk, scene=scene, dlc_tables=tabs, n_frames=100, start_frame=60 + 80 * i, pairing="name")[0] for i in range(16)]
out = build.solve_models(models, max_iter=20, ftol=0.0, xtol=0.0, gtol=0.0)
torch.cuda.synchronize()
print("extras workload done:", [i["iterations"] for _r, i in out][:4])
