#!/usr/bin/env python3
"""SBA report on the GPU box: KAT-2 end states + a large synthetic rig, with timings.
Writes gpurun_out/sba/report.json.  (The scipy oracle is timed beside it on the KAT-2 problems only.)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from acinoset_amd import sba, synth  # noqa: E402
from oracle import camera as ocam  # noqa: E402
from oracle import sba as osba  # noqa: E402
from test_oracle_sba import kat2_problem  # noqa: E402

out = {}
g = np.load(os.path.join(ROOT, "tests", "golden", "kat1_sunday_amelia.npz"))
for tag, ca, cb, row in (("rotating", 1, 2, 0), ("static", 3, 4, 1)):
    img, names, shape, K, D, R, t = kat2_problem(g, tag, ca, cb)
    data = osba.prepare_calib_board_data(img, names, shape, K, D, R, t, ocam.triangulate_points_fisheye)
    sba.bundle_adjust_points_and_extrinsics(*data, K, D, R, t)            # warm
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pts, rm, tt, res = sba.bundle_adjust_points_and_extrinsics(*data, K, D, R, t)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    _p, _r, _t, ores, oopt = osba.bundle_adjust_points_and_extrinsics(*data, K, D, R, t)
    dto = time.perf_counter() - t0
    out[f"kat2_{tag}"] = dict(gpu=sba.last_info, gpu_seconds=dt, after_mean=float(res["after"].mean()),
                              after_std=float(res["after"].std()), recorded=list(map(float, g["recorded_sba"][row])),
                              scipy_cost=float(oopt.cost), scipy_nfev=int(oopt.nfev), scipy_seconds=dto,
                              scipy_optimality=float(oopt.optimality))

rng = np.random.default_rng(7)
K, D, R, t = synth.make_rig()
for n_pts in (20000, 500000):
    X = np.array([2.0, 6.5, 0.7]) + rng.normal(0, 1.0, (n_pts, 3))
    seen = rng.random((n_pts, 6)) < 0.7
    seen[:, 0] = True
    seen[:, 1] |= ~seen[:, 1:].any(axis=1)
    pi, ci = np.nonzero(seen)
    uv = np.zeros((len(pi), 2))
    for c in range(6):
        m = ci == c
        uv[m] = ocam.project_points_fisheye(X[pi[m]], K[c], D[c], R[c], t[c])
    uv += rng.normal(0, 0.3, uv.shape)
    Rp = np.array([ocam.rodrigues(rng.normal(0, 0.01, 3)) @ R[c] for c in range(6)])
    tp = t.reshape(6, 3, 1) + rng.normal(0, 0.01, (6, 3, 1))
    X0 = X + rng.normal(0, 0.03, X.shape)
    sba.bundle_adjust_points_and_extrinsics(uv, X0, pi, ci, K, D, Rp, tp, max_iter=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pts, rm, tt, res = sba.bundle_adjust_points_and_extrinsics(uv, X0, pi, ci, K, D, Rp, tp)
    dt = time.perf_counter() - t0
    info = dict(sba.last_info)
    out[f"synthetic_{n_pts}"] = dict(points=n_pts, observations=int(len(pi)), gpu=info, gpu_seconds_incl_transfers=dt,
                                     rms_after_px=float(np.sqrt(np.mean(res["after"] ** 2))),
                                     obs_iterations_per_s=len(pi) * info["iterations"] / dt)
os.makedirs(os.path.join(ROOT, "gpurun_out", "sba"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "sba", "report.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
