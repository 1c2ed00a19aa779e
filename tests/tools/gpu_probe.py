"""Ad-hoc GPU diagnostics (not a test): prints parity numbers of every kernel against the oracle."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import ctypes as C
from acinoset_amd import _lib, calib, fte, synth
from acinoset_amd._lib import lib, ptr, stream_ptr, check
from oracle import camera as ocam, fk as ofk, fte as ofte, index_path as oidx, loss as oloss

dev = torch.device("cuda")
rng = np.random.default_rng(0)
# 1. MFMA
for K in (4, 16, 80):
    a = rng.normal(size=(16, K)); b = rng.normal(size=(K, 16))
    da, db = torch.tensor(a, device=dev), torch.tensor(b, device=dev); dc = torch.zeros(16, 16, dtype=torch.float64, device=dev)
    check(lib().acino_selftest_mfma(ptr(da), ptr(db), K, ptr(dc), stream_ptr()))
    print("mfma K", K, "err", np.abs(dc.cpu().numpy() - a @ b).max())
# 2. camera
K_, D_, R_, t_ = synth.make_rig()
X = np.array([2.0, 6.5, 0.7]) + rng.normal(0, 2.0, (1000, 3))
for c in range(6):
    uv = calib.project_points_fisheye(X, K_[c], D_[c], R_[c], t_[c])
    uo = ocam.project_points_fisheye(X, K_[c], D_[c], R_[c], t_[c])
    print("project fisheye cam", c, "err", np.nanmax(np.abs(uv - uo)))
dpin = np.array([0.1, -0.05, 0.001, -0.002, 0.01, 0.02, -0.01, 0.003])
uv = calib.project_points(X, K_[1], dpin, R_[1], t_[1]); uo = ocam.project_points(X, K_[1], dpin, R_[1], t_[1])
print("project pinhole err", np.nanmax(np.abs(uv - uo)))
p1 = ocam.project_points_fisheye(X, K_[0], D_[0], R_[0], t_[0]) + rng.normal(0, 1, (1000, 2))
p2 = ocam.project_points_fisheye(X, K_[1], D_[1], R_[1], t_[1]) + rng.normal(0, 1, (1000, 2))
und = calib.undistort_points_fisheye(p1, K_[0], D_[0]); uno = ocam.undistort_points_fisheye(p1, K_[0], D_[0])
print("undistort err", np.nanmax(np.abs(und - uno)))
tg = calib.triangulate_points_fisheye(p1, p2, K_[0], D_[0], R_[0], t_[0], K_[1], D_[1], R_[1], t_[1])
to = ocam.triangulate_points_fisheye(p1, p2, K_[0], D_[0], R_[0], t_[0], K_[1], D_[1], R_[1], t_[1])
print("triangulate fisheye err", np.nanmax(np.abs(tg - to)), "vs truth", np.nanmedian(np.abs(tg - X)))
q1 = ocam.project_points(X, K_[0], dpin, R_[0], t_[0]); q2 = ocam.project_points(X, K_[1], dpin, R_[1], t_[1])
tg = calib.triangulate_points(q1, q2, K_[0], dpin, R_[0], t_[0], K_[1], dpin, R_[1], t_[1])
to = ocam.triangulate_points(q1, q2, K_[0], dpin, R_[0], t_[0], K_[1], dpin, R_[1], t_[1])
print("triangulate pinhole err", np.nanmax(np.abs(tg - to)))
# 3. FK
g = np.load(os.path.join(os.path.dirname(__file__), "..", "golden", "cheetah_fk.npz"))
pos = fte.cheetah_fk(g["q"])
print("fk err vs golden", np.abs(pos - g["positions"]).max())
# 4. sequence + pairs
seq = synth.make_sequence(60, "sprint")
det = seq["det"]
tri, cnt, mask = calib.triangulate_pairs_dense(det, 0.5, seq["K"], seq["D"], seq["R"], seq["t"])
tro, cno, mko = oidx.pairwise_dense(det, 0.5, seq["K"], seq["D"], seq["R"], seq["t"], ocam.triangulate_points_fisheye)
print("pairs: cnt equal", (cnt == cno).all(), "mask equal", (mask == mko).all(), "tri err", np.nanmax(np.abs(tri - tro)), "nan equal", (np.isnan(tri) == np.isnan(tro)).all())
res, sums = calib.reproject_residuals(tri, det, 0.5, seq["K"], seq["D"], seq["R"], seq["t"])
print("reproj sums", sums)
# 5. FTE evaluate
x0 = fte.nose_line_init(det, seq["K"], seq["D"], seq["R"], seq["t"], 0.5)
xa = seq["q_true"][:, ofk.ACTIVE] + rng.normal(0, 0.02, (60, 25))
lo, hi = ofk.bounds45(); xa = np.clip(xa, lo[ofk.ACTIVE], hi[ofk.ACTIVE])
prob = ofte.FTEProblem(det[..., :2], det[..., 2], seq["K"], seq["D"], seq["R"], seq["t"], seq["Ts"])
Fo, go, Ho, nb = prob.evaluate(xa)
ctx = fte.FTEContext(det, seq["K"], seq["D"], seq["R"], seq["t"], seq["Ts"])
ctx.set_x(xa)
st = ctx.state(); print("state after set_x", st)
print("cost: gpu", st["cost"], "oracle", Fo, "rel", abs(st["cost"] - Fo) / abs(Fo))
gg, hh = ctx.grad_hess(); gg = gg.cpu().numpy(); hh = hh.cpu().numpy()
band = prob.s_band()
Ho2 = Ho.copy(); idx = np.arange(25); Ho2[:, idx, idx] += 2 * prob.q_w[None, :] * band[0][:, None]
print("grad rel err", np.abs(gg - go).max() / np.abs(go).max(), "H rel err", np.abs(hh - Ho2).max() / np.abs(Ho2).max(), "H asym", np.abs(hh - hh.transpose(0, 2, 1)).max())
print("cost-only", ctx.cost(xa), Fo)
# 6. one LM step compare
delta_o, diag = prob.solve_banded(Ho, go, 1e-3, prob.active_set(xa, go, Ho))
ctx.step()
st = ctx.state(); print("after step", st)
xg = ctx.result()[0].cpu().numpy()
xt_o = np.clip(xa + delta_o, prob.lo, prob.hi)
Ft_o = prob.evaluate(xt_o, need_jac=False)[0]
print("trial cost oracle", Ft_o, "gpu", st["cost_trial"], "x diff", np.abs(xg - xt_o).max(), "delta max", np.abs(delta_o).max())
# 7. full solve
for N, kind in ((60, "sprint"), (100, "sprint")):
    seq = synth.make_sequence(N, kind); det = seq["det"]
    x0 = fte.nose_line_init(det, seq["K"], seq["D"], seq["R"], seq["t"], 0.5)
    t0 = time.time()
    res, info = fte.fte_solve(det[..., :2], det[..., 2], seq["K"], seq["D"], seq["R"], seq["t"], seq["Ts"], x0=x0, max_iter=60)
    tg = time.time() - t0
    prob = ofte.FTEProblem(det[..., :2], det[..., 2], seq["K"], seq["D"], seq["R"], seq["t"], seq["Ts"])
    t0 = time.time(); xo, oinfo = ofte.lm_solve(prob, x0[:, ofk.ACTIVE], max_iter=60); to_ = time.time() - t0
    po = ofte.fte_outputs(prob, xo, x0)["positions"]
    print(f"N={N}: gpu {info['iter']} it cost {info['cost']:.9f} {info['status_name']} ({tg:.2f}s) | oracle {oinfo['iterations']} it cost {oinfo['cost']:.9f} {oinfo['status']} ({to_:.2f}s) | max dpos {np.abs(res['positions']-po).max():.3e} vs truth {np.abs(res['positions']-seq['pos_true']).max():.3e}")
print("PROBE DONE")
