#!/usr/bin/env python3
"""config3_solution.npz: BASELINE config 3 at its exact size - 6 cameras x 20 markers x 1 000 frames ("trot"
sequence of oracle/synth.py: a straight 1.2 m/s run through the rig, seed 20210313), nose-line initialisation (all_optimizations.py:268-277, 333-337 =
oracle.fte.nose_line_init on the oracle's adjacent-pair triangulation), solved by the oracle's projected LM to its
default tolerances.  Needs only this repo (no reference tree); ~2 minutes on one core.  The detections are NOT stored
(2.9 MB): the GPU test regenerates them with the same seeded oracle generator.

Usage:  python tests/golden/make_config3.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import camera, fk, index_path, synth  # noqa: E402
from oracle import fte as ofte  # noqa: E402

N = 1000
seq = synth.make_sequence(N, "trot")
det, rig = seq["det"], (seq["K"], seq["D"], seq["R"], seq["t"])
tri, _cnt, _mask = index_path.pairwise_dense(det, 0.5, *rig, camera.triangulate_points_fisheye)
nose = tri[:, 2]
ok = np.isfinite(nose).all(1)
x0 = ofte.nose_line_init(np.arange(N, dtype=np.float64)[ok], nose[ok], N, start_frame=0)
prob = ofte.FTEProblem(det[..., :2], det[..., 2], *rig, seq["Ts"])
t0 = time.time()
hist = []
xa, info = ofte.lm_solve(prob, x0[:, fk.ACTIVE], max_iter=200, history=hist)
out = ofte.fte_outputs(prob, xa, x0)
print(f"{info['iterations']} iterations ({info['accepted']} accepted), status {info['status']}, cost {info['cost']:.9f}, "
      f"|g| {info['gnorm']:.3e}, {time.time() - t0:.1f} s")
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "config3_solution.npz"),
                    n_frames=N, kind="trot", seed=20210313, det_checksum=float(det.sum()),
                    x0_line=np.array([x0[0, :3], x0[-1, :3]]), psi0=float(x0[0, 31]),
                    x=xa, cost=info["cost"], iterations=info["iterations"], accepted=info["accepted"],
                    status=info["status"], gnorm=info["gnorm"],
                    cost_history=np.array([h["F"] for h in hist]),
                    positions_probe=out["positions"][::50])
