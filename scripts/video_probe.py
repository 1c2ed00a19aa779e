"""The whole shipped video through build.solve_video: mean |residual| per window with and without the warm-start passes."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, ".")
from acinoset_amd import build
gd = "tests/golden"
g = np.load(os.path.join(gd, "skel_fte_model.npz"))
sk = json.loads(str(g["skeleton_json"]))
full = np.load(os.path.join(gd, "human_dlc_full.npz"))
tabs = [(list(full["parts"]), full[f"det{c}"].astype(np.float64)) for c in range(2)]
scene = (g["K"], g["D"], g["R"], g["t"])
for passes in (0, 3):
    t0 = time.perf_counter()
    res, infos, starts = build.solve_video(sk, scene=scene, dlc_tables=tabs, first_frame=0, last_frame=6239, window=100, overlap=20,
                                           pairing="name", max_iter=1500, warm_passes=passes)
    dt = time.perf_counter() - t0
    px = [i["mean_abs_residual_px"] for i in infos]
    print(f"warm_passes={passes}: {dt:.2f} s; windows above 15 px: {[(k, round(p, 1)) for k, p in enumerate(px) if p > 15]}")
    print("   px:", " ".join(f"{p:.1f}" for p in px))
    print("   warm from:", [(k, i["warm_started_from"]) for k, i in enumerate(infos) if i["warm_started_from"] is not None])
    print("   status:", sorted(set(i["status_name"] for i in infos)), "iterations", sum(i["iterations"] for i in infos))
    # detections per window (weighted rows), to see which windows are thin
    if passes == 3:
        for k, st in enumerate(starts):
            if px[k] > 10:
                m, _ = build.build_model(sk, scene=scene, dlc_tables=tabs, n_frames=100, start_frame=st, pairing="name", initial_line=False)
                print(f"   window {k} (frames {st}..{st + 99}): {px[k]:.1f} px, weighted rows {2 * int((m.weights > 0).sum())}, "
                      f"frames with no detection at all {int(((m.weights > 0).sum(axis=(1, 2)) == 0).sum())}")
