import json, os, sys, numpy as np
sys.path.insert(0, ".")
from acinoset_amd import build
gd = "tests/golden"
g = np.load(os.path.join(gd, "skel_fte_model.npz")); sk = json.loads(str(g["skeleton_json"]))
det = np.load(os.path.join(gd, "human_dlc_slice.npz"))["det"].astype(np.float64)
tabs = [(list(g["parts"]), det[:, c]) for c in range(det.shape[1])]
model, _ = build.build_model(sk, scene=(g["K"], g["D"], g["R"], g["t"]), dlc_tables=tabs, n_frames=400, start_frame=60, pairing="name")
res_s, info_s = build.solve_model(model, max_iter=300)
for xt in (1e-7, 1e-9):
    res_p, info_p = build.solve_model_parallel(model, xtol_outer=xt, outer_max=80)
    print("xtol_outer", xt, "outer", info_p["outer_iterations"], info_p["status_name"], "cost", info_p["cost_final"], "gnorm", info_p["gnorm_inf"], "single", info_s["cost_final"], info_s["gnorm_inf"], info_s["iterations"])
    for mi in (1, 2, 5, 50):
        r, i = build.solve_model(model, x0=res_p["x"], max_iter=mi)
        print("   exact solver from the Schwarz end point, max_iter", mi, ":", i["iterations"], i["status_name"], "cost", i["cost_final"], "rel change", (info_p["cost_final"] - i["cost_final"]) / info_p["cost_final"], "max |dx|", float(np.abs(r["x"] - res_p["x"]).max()), "gnorm", i["gnorm_inf"])
r, i = build.solve_model(model, x0=res_s["x"], max_iter=50)
print("exact solver restarted from its own end point:", i["iterations"], i["status_name"], (info_s["cost_final"] - i["cost_final"]) / info_s["cost_final"], float(np.abs(r["x"] - res_s["x"]).max()))
