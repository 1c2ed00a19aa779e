#!/usr/bin/env python3
"""Path identity, not end-point closeness: the HIP solve and the oracle LM must produce the SAME sequence of trial costs
(accepted and rejected) and the same accept / reject decisions, iteration by iteration, from the same start - a much sharper probe of the controller, the
active-set rule and the block solve than comparing converged solutions (it is what exposed the 1e-21 drift off a 0.0
bound in round 2).  Nasty cases on purpose: 3 ... 40 frames, 2 ... 6 cameras, gross outliers, dropped detections, starts
ON the bounds (nose-line style: all angles 0) or random.  usage: [FUZZ_MAX_FRAMES=40] [FUZZ_RANDOM_TS=1] fuzz_lm_path.py first_seed n_seeds [iterations] [v]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from acinoset_amd import fte, synth  # noqa: E402
from oracle import fk as ofk  # noqa: E402
from oracle import fte as ofte  # noqa: E402



def make_case(seed):
    """-> det[n,C,20,3], rig (K, D, R, t of the chosen cameras), Ts, x0[n,45], description fields"""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(3, int(os.environ.get("FUZZ_MAX_FRAMES", "40")) + 1))
    cams = np.sort(rng.choice(6, size=int(rng.integers(2, 7)), replace=False))
    kind = ("sprint", "loop", "trot")[int(rng.integers(0, 3))]
    seq = synth.make_sequence(n, kind, seed=seed)
    det = seq["det"][:, cams].copy()
    det[rng.random(det.shape[:3]) < rng.uniform(0, 0.4), 2] = 0.0
    out = rng.random(det.shape[:3]) < rng.uniform(0, 0.25)
    det[..., :2] += out[..., None] * rng.uniform(-80, 80, det[..., :2].shape)
    rig = tuple(a[cams] for a in (seq["K"], seq["D"], seq["R"], seq["t"]))
    lo, hi = fte.bounds45()
    x0 = np.zeros((n, 45))
    mode = int(rng.integers(0, 3))
    if mode == 0:      # nose-line style: position from the truth, heading, every angle 0 (several ON their 0.0 bound)
        x0[:, :3] = seq["q_true"][:, :3] + rng.normal(0, 0.05, (n, 3))
        x0[:, 31] = seq["q_true"][:, 31].mean()
    elif mode == 1:    # near the truth
        x0[:, fte.ACTIVE] = seq["q_true"][:, fte.ACTIVE] + rng.normal(0, 0.05, (n, 25))
    else:              # far: random angles, many clipped to their bounds
        x0[:, fte.ACTIVE] = seq["q_true"][:, fte.ACTIVE] + rng.normal(0, 0.8, (n, 25))
        x0[:, :3] = seq["q_true"][:, :3] + rng.normal(0, 0.2, (n, 3))
    x0 = np.clip(x0, lo, hi)
    Ts = seq["Ts"]
    if os.environ.get("FUZZ_RANDOM_TS"):                # another frame rate: the smoothness weights 1 / (Q Ts^4) move by 256x either way
        Ts = 1.0 / float((30, 60, 120, 240, 480)[int(rng.integers(0, 5))])
    return det, rig, Ts, x0, (n, kind, cams, mode)


def run_case(seed, iters=10, verbose=False):
    """-> (worst relative trial-cost difference over the iterations, iteration where, description)"""
    det, rig, Ts, x0, (n, kind, cams, mode) = make_case(seed)
    seq = {"Ts": Ts}
    prob = ofte.FTEProblem(det[..., :2], det[..., 2], *rig, seq["Ts"])
    hist = []
    singular = False
    try:
        ofte.lm_solve(prob, x0[:, ofk.ACTIVE], max_iter=iters, ftol=0.0, xtol=0.0, gtol=0.0, history=hist)
    except np.linalg.LinAlgError:
        # a state nobody observes and (fewer than 4 frames) no smoothness row either: zero diagonal, Marquardt scaling
        # cannot lift it - scipy's banded Cholesky raises; the HIP solve must report the same thing (status 5)
        singular = True
    ctx = fte.FTEContext(det, *rig, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0)
    ctx.set_x(x0[:, fte.ACTIVE])
    worst, where = 0.0, -1
    acc_prev = 0
    for h in hist:
        ctx.step()
        st = ctx.state()
        if st["status"] != 0:
            worst, where = float("inf"), h["it"]
            break
        d = abs(st["cost_trial"] - h["Ft"]) / max(abs(h["Ft"]), 1e-300)
        if verbose:
            print(f"   it {h['it']:2d} lam {h['lam']:.2e} oracle Ft {h['Ft']:.12e} gpu {st['cost_trial']:.12e} rel {d:.1e} step {h['step']:.2e} gain {h['gain']:+.2f}")
        accepted_gpu, accepted_or = st["accepted"] > acc_prev, h["Ft"] < h["F"]
        acc_prev = st["accepted"]
        if accepted_gpu != accepted_or:
            if abs(h["Ft"] - h["F"]) < 1e-9 * abs(h["F"]):
                break                             # converged: Ft < F is decided by the last bits of the sums - stop comparing
            worst, where = float("inf"), h["it"]  # a different DECISION away from that: never acceptable
            break
        # a rejected trial point is a far-flung overshoot (gain -10 ... -1e4) of an ill-conditioned step: its cost is 1e3 x
        # more sensitive to the last bits of the step than an accepted point's - scaled accordingly
        d = d if accepted_or else d * 1e-3
        if d > worst:
            worst, where = d, h["it"]
    if singular:
        ctx.step()
        st = ctx.state()
        worst, where = (0.0, len(hist) + 1) if st["status"] == 5 else (float("inf"), len(hist) + 1)
    ctx.close()
    return worst, where, f"{n} frames, {kind}, cameras {[int(c) for c in cams]}, start {('line', 'near', 'far')[mode]}, {len(hist)} iterations{' then SINGULAR in both' if singular else ''}"


if __name__ == "__main__":
    first, count = int(sys.argv[1]), int(sys.argv[2])
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    bad = 0
    for seed in range(first, first + count):
        worst, where, what = run_case(seed, iters, verbose=len(sys.argv) > 4)
        ok = worst < 1e-5     # (4-frame problems at lam ~ 1e-6 reach 1e-5 after 20 iterations with identical decisions: conditioning)
        bad += not ok
        print(seed, what, f"worst rel trial-cost difference {worst:.1e} at it {where}", "ok" if ok else "MISMATCH", flush=True)
    print("mismatches:", bad)
