#!/usr/bin/env python3
"""Extended version of tests/test_gpu_parity.py::test_randomised_solves_match_oracle: many seeds, GPU solve vs the
oracle LM (cost to 1e-5 relative, marker positions to 1e-3 m).  usage: fuzz_solves.py first_seed n_seeds"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from acinoset_amd import fte, synth  # noqa: E402
from oracle import fk as ofk  # noqa: E402
from oracle import fte as ofte  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(8, 90))
    cams = np.sort(rng.choice(6, size=int(rng.integers(2, 7)), replace=False))
    kind = "sprint" if rng.random() < 0.7 else "loop"
    seq = synth.make_sequence(n, kind, seed=seed)
    det = seq["det"][:, cams].copy()
    det[rng.random(det.shape[:3]) < rng.uniform(0, 0.3), 2] = 0.0
    rig = tuple(a[cams] for a in (seq["K"], seq["D"], seq["R"], seq["t"]))
    x0 = np.zeros((n, 45))
    x0[:, fte.ACTIVE] = seq["q_true"][:, fte.ACTIVE] + rng.normal(0, 0.03, (n, 25))
    lo, hi = fte.bounds45()
    x0 = np.clip(x0, lo, hi)
    res, info = fte.fte_solve(det[..., :2], det[..., 2], *rig, seq["Ts"], x0=x0, max_iter=150, ftol=1e-13)
    prob = ofte.FTEProblem(det[..., :2], det[..., 2], *rig, seq["Ts"])
    xo, oinfo = ofte.lm_solve(prob, x0[:, ofk.ACTIVE], max_iter=150, ftol=1e-13)
    out = ofte.fte_outputs(prob, xo, x0)
    dc = abs(info["cost"] - oinfo["cost"]) / abs(oinfo["cost"])
    dp = np.abs(res["positions"] - out["positions"]).max()
    ok = dc < 1e-5 and dp < 1e-3
    bad += not ok
    print(seed, n, kind, list(cams), info["status_name"], info.get("iter"), oinfo.get("iter", oinfo.get("iterations")), f"{dc:.1e} {dp:.1e}", "ok" if ok else "MISMATCH",
          flush=True)
print("mismatches:", bad)
