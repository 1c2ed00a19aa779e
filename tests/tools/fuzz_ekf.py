#!/usr/bin/env python3
"""Random short clips through the GPU EKF + smoother and the numpy oracle (outlier counts equal, states to the float32
ulp the reference's own rounding leaves).  usage: fuzz_ekf.py first_seed n_seeds"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from acinoset_amd import ekf, synth  # noqa: E402
from oracle import ekf as oekf  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(3, 28))
    cams = np.sort(rng.choice(6, size=int(rng.integers(1, 7)), replace=False))
    kind = "sprint" if rng.random() < 0.6 else "walk"
    seq = synth.make_sequence(n, kind, seed=seed)
    s0 = ekf.initial_state(seq["det"], seq["K"], seq["D"], seq["R"], seq["t"], 120.0, 0.5)
    det = seq["det"][:, cams].copy()
    det[rng.random(det.shape[:3]) < rng.uniform(0, 0.3), 2] = 0.0
    rig = tuple(a[cams] for a in (seq["K"], seq["D"], seq["R"], seq["t"]))
    want = oekf.ekf(det, *rig, 120.0, 0.5, 2704, s0)
    got = ekf.ekf(det, *rig, 120.0, 0.5, (2704, 1520), states0=s0, with_positions=False)
    errs = {k: np.abs(got[k] - want[k]).max() / max(1.0, np.abs(want[k]).max()) for k in ("x", "dx", "ddx", "smoothed_x", "smoothed_dx", "smoothed_ddx")}
    ok = got["outliers_ignored"] == want["outliers_ignored"] and errs["x"] < 5e-6 and errs["smoothed_x"] < 5e-6 and \
        errs["dx"] < 5e-5 and errs["smoothed_dx"] < 5e-5 and errs["ddx"] < 1e-3 and errs["smoothed_ddx"] < 1e-3
    bad += not ok
    print(seed, n, kind, [int(c) for c in cams], got["outliers_ignored"], want["outliers_ignored"],
          " ".join(f"{k}={v:.1e}" for k, v in errs.items()), "ok" if ok else "MISMATCH", flush=True)
print("mismatches:", bad)
