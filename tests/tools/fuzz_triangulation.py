#!/usr/bin/env python3
"""Two-view triangulation on random geometries (baseline 5 cm .. 5 m, depth 1 .. 60 m, noise 0 .. 5 px): the GPU path
(inverse iteration on A^T A, Jacobi SVD fallback) against the numpy-SVD oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from acinoset_amd import calib, synth  # noqa: E402
from oracle import camera as ocam  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
K, D, _R, _t = synth.make_rig()
worst = 0.0
for case in range(60):
    base = 10 ** rng.uniform(np.log10(0.05), np.log10(5.0))
    depth = 10 ** rng.uniform(0, np.log10(60.0))
    noise = rng.choice([0.0, 0.2, 1.0, 5.0])
    R0 = ocam.rodrigues(rng.normal(0, 0.3, 3))
    R1 = ocam.rodrigues(rng.normal(0, 0.2, 3)) @ R0
    c0 = rng.normal(0, 1.0, 3)
    c1 = c0 + base * R0.T @ np.array([1.0, 0.1 * rng.normal(), 0.1 * rng.normal()])
    t0, t1 = (-R0 @ c0).reshape(3, 1), (-R1 @ c1).reshape(3, 1)
    Xc = np.stack([rng.uniform(-0.4, 0.4, 500) * depth, rng.uniform(-0.3, 0.3, 500) * depth, depth * rng.uniform(0.7, 1.3, 500)], 1)
    X = (Xc - t0.ravel()) @ R0                 # world points in front of camera 0
    p0 = ocam.project_points_fisheye(X, K[0], D[0], R0, t0) + rng.normal(0, noise, (500, 2))
    p1 = ocam.project_points_fisheye(X, K[1], D[1], R1, t1) + rng.normal(0, noise, (500, 2))
    ok = np.isfinite(p0).all(1) & np.isfinite(p1).all(1)
    g = calib.triangulate_points_fisheye(p0[ok], p1[ok], K[0], D[0], R0, t0, K[1], D[1], R1, t1)
    o = ocam.triangulate_points_fisheye(p0[ok], p1[ok], K[0], D[0], R0, t0, K[1], D[1], R1, t1)
    rel = np.abs(g - o).max(1) / np.maximum(1.0, np.abs(o).max(1))
    worst = max(worst, float(np.nanmax(rel)))
    print(f"case {case:2d} baseline {base:5.2f} m depth {depth:5.1f} m noise {noise:3.1f} px: median {np.nanmedian(rel):.1e} max {np.nanmax(rel):.1e}"
          f" nan {int(np.isnan(g).any(1).sum())}/{int(np.isnan(o).any(1).sum())}", flush=True)
print("worst relative difference:", worst)
