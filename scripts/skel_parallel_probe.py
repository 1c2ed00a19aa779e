import os, sys, time, json
sys.path.insert(0, ".")
import numpy as np, torch
from acinoset_amd import build
gd = "tests/golden"
gsk = np.load(os.path.join(gd, "skel_fte_model.npz")); sk = json.loads(str(gsk["skeleton_json"]))
scene = (gsk["K"], gsk["D"], gsk["R"], gsk["t"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
if n <= 400:
    det = np.load(os.path.join(gd, "human_dlc_slice.npz"))["det"].astype(np.float64)
    tabs = [(list(gsk["parts"]), det[:, c]) for c in range(det.shape[1])]
    sf = 60
else:
    full = np.load(os.path.join(gd, "human_dlc_full.npz"))
    tabs = [(list(full["parts"]), full[f"det{c}"].astype(np.float64)) for c in range(2)]
    sf = 0
model, _ = build.build_model(sk, scene=scene, dlc_tables=tabs, n_frames=n, start_frame=sf, pairing="name")
x0 = None
if n > 1200:
    t0 = time.perf_counter()
    vres, vinfos, vstarts = build.solve_video(sk, scene=scene, dlc_tables=tabs, first_frame=0, last_frame=n - 1, window=100, overlap=20, pairing="name", max_iter=1500)
    x0 = vres["x"]
    r0, i0 = build.solve_model(model, x0=x0, max_iter=0)
    print(f"free windows first: {time.perf_counter() - t0:.2f} s, whole-clip cost of the stitched start {i0['cost_final']:.1f}, gnorm {i0['gnorm_inf']:.3e}")
torch.cuda.synchronize(); t0 = time.perf_counter()
kw = dict(first_max_iter=int(sys.argv[2]), later_max_iter=int(sys.argv[3])) if len(sys.argv) > 3 else {}
res_p, info_p = build.solve_model_parallel(model, x0=x0, outer_max=40, **kw)
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"parallel: {t1 - t0:.2f} s, outer {info_p['outer_iterations']}, status {info_p['status_name']}")
for h in info_p["history"]:
    print("   ", {k: (round(v, 6) if isinstance(v, float) else v) for k, v in h.items()})
if n <= 1200:
    t0 = time.perf_counter()
    res_s, info_s = build.solve_model(model, max_iter=600)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"single workgroup: {t1 - t0:.2f} s, {info_s['iterations']} it, {info_s['status_name']}, cost {info_s['cost_final']:.6f}, gnorm {info_s['gnorm_inf']:.3e}")
    print("cost parallel / single:", info_p["cost_final"], info_s["cost_final"], "max |dx|", float(np.abs(res_p["x"] - res_s["x"]).max()),
          "max |dpos|", float(np.abs(res_p["positions"] - res_s["positions"]).max()))
