#!/usr/bin/env python3
"""fuzz_lm_path.py for clip_len: B clips of one length laid end to end as ONE chain (fte_solve_clips / config 5's batched
form) against the oracle LM with one shared controller over B independent problems (tests/test_gpu_parity.py::
_oracle_lm_clips) - same accept / reject decisions, same trial costs.  usage: fuzz_clips_path.py first_seed n_seeds [iterations]"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from acinoset_amd import fte, synth  # noqa: E402
from oracle import fk as ofk  # noqa: E402
from oracle import fte as ofte  # noqa: E402

spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
tgp = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tgp)

first, count = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 12
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    B, clip = int(rng.integers(2, 7)), int(rng.integers(4, 41))
    cams = np.sort(rng.choice(6, size=int(rng.integers(2, 7)), replace=False))
    seqs = [synth.make_sequence(clip, ("sprint", "trot", "loop")[int(rng.integers(0, 3))], seed=seed * 10 + b) for b in range(B)]
    rig = tuple(a[cams] for a in (seqs[0]["K"], seqs[0]["D"], seqs[0]["R"], seqs[0]["t"]))
    lo, hi = fte.bounds45()
    mode = int(rng.integers(0, 3))
    dets, x0s = [], []
    for sq in seqs:
        det = sq["det"][:, cams].copy()
        det[rng.random(det.shape[:3]) < rng.uniform(0, 0.4), 2] = 0.0
        out = rng.random(det.shape[:3]) < rng.uniform(0, 0.2)
        det[..., :2] += out[..., None] * rng.uniform(-80, 80, det[..., :2].shape)
        x0 = np.zeros((clip, 45))
        if mode == 0:
            x0[:, :3] = sq["q_true"][:, :3] + rng.normal(0, 0.05, (clip, 3))
            x0[:, 31] = sq["q_true"][:, 31].mean()
        else:
            x0[:, fte.ACTIVE] = sq["q_true"][:, fte.ACTIVE] + rng.normal(0, (0.05, 0.05, 0.6)[mode], (clip, 25))
        dets.append(det)
        x0s.append(np.clip(x0, lo, hi))
    probs = [ofte.FTEProblem(d[..., :2], d[..., 2], *rig, seqs[0]["Ts"]) for d in dets]
    hist = tgp._oracle_lm_clips(probs, [x[:, ofk.ACTIVE] for x in x0s], iters)
    ctx = fte.FTEContext(np.concatenate(dets), *rig, seqs[0]["Ts"], clip_len=clip, ftol=0.0, xtol=0.0, gtol=0.0)
    ctx.set_x(np.concatenate(x0s)[:, fte.ACTIVE])
    worst, where, acc = 0.0, -1, 0
    F_prev = None
    for it, (Ft, accepted) in enumerate(hist):
        ctx.step()
        st = ctx.state()
        if st["status"] != 0 or (st["accepted"] > acc) != accepted:
            worst, where = float("inf"), it + 1
            break
        acc = st["accepted"]
        d = abs(st["cost_trial"] - Ft) / abs(Ft) * (1.0 if accepted else 1e-3)
        if d > worst:
            worst, where = d, it + 1
    ctx.close()
    ok = worst < 1e-5
    bad += not ok
    print(seed, f"{B} clips x {clip} frames, cameras {[int(c) for c in cams]}, start {('line', 'near', 'far')[mode]}: worst {worst:.1e} at it {where}",
          "ok" if ok else "MISMATCH", flush=True)
print("mismatches:", bad)
