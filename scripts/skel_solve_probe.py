"""ms per LM iteration of the skeleton solve on one clip (400 / 100 frames of the shipped detections)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acinoset_amd import build
gd = os.path.join(ROOT, "tests", "golden")
g = np.load(os.path.join(gd, "skel_fte_model.npz")); sk = json.loads(str(g["skeleton_json"]))
det = np.load(os.path.join(gd, "human_dlc_slice.npz"))["det"].astype(np.float64)
tabs = [(list(g["parts"]), det[:, c]) for c in range(det.shape[1])]
for n in (100, 400):
    model, _ = build.build_model(sk, scene=(g["K"], g["D"], g["R"], g["t"]), dlc_tables=tabs, n_frames=n, start_frame=60, pairing="name")
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res, info = build.solve_model(model, max_iter=300)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{n} frames: {info['iterations']} iterations, {info['status_name']}, cost {info['cost_final']:.6f}, {1e3 * dt:.1f} ms = {1e3 * dt / info['iterations']:.3f} ms per iteration "
          f"= {1e6 * dt / info['iterations'] / n:.1f} us per frame and iteration")
