"""Probe (test tooling, uses the oracle): does a forward-tracking line search along the LM step cut the iteration count?
usage: python tests/tools/lm_relax_probe.py [frames]   (on the GPU box: synth needs the device FK)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from oracle import fte as ofte, fk as ofk
from acinoset_amd import synth
import acinoset_amd.fte as afte
n = int(sys.argv[1]) if len(sys.argv) > 1 else 240
seq = synth.make_sequence(n, "loop")
rig = (seq["K"], seq["D"], seq["R"], seq["t"])
det = seq["det"]
prob = ofte.FTEProblem(det[..., :2], det[..., 2], *rig, seq["Ts"])
q = seq["q_true"]
rng = np.random.default_rng(0)
x0 = np.zeros((n, 45))
x0[:, 0:3] = q[:, 0:3] + rng.normal(0, 0.01, (n, 3))
x0[:, afte.PSI] = q[:, afte.PSI] + rng.normal(0, 0.02, n)
xa0 = x0[:, ofk.ACTIVE]

def solve(alphas, max_iter=60, lam0=1e-3, ftol=1e-10, shrink=1.0 / 3.0, verbose=False):
    lo, hi = prob.lo, prob.hi
    x = np.clip(xa0, lo, hi)
    F, g, H, nb = prob.evaluate(x)
    lam, nu = lam0, 2.0
    nev = 1
    for it in range(1, max_iter + 1):
        fixed = prob.active_set(x, g, H)
        pg = np.where(fixed, 0.0, g)
        delta, diag = prob.solve_banded(H, g, lam, fixed)
        xt = np.clip(x + delta, lo, hi)
        Ft, gt, Ht, nbt = prob.evaluate(xt); nev += 1
        pred = 0.5 * float((delta * (lam * diag * delta - pg)).sum())
        gain = (F - Ft) / pred if pred > 0 else -1.0
        if Ft < F:
            best = (Ft, xt, gt, Ht, nbt, 1.0)
            for a in alphas:
                if gain < 0.25: break
                xa = np.clip(x + a * delta, lo, hi)
                Fa, ga, Ha, nba = prob.evaluate(xa); nev += 1
                if Fa < best[0]:
                    best = (Fa, xa, ga, Ha, nba, a)
                else:
                    break
            dF = F - best[0]
            if verbose: print(f"it {it:2d} F {F:.6e} -> {best[0]:.6e} gain {gain:.2f} alpha {best[5]} lam {lam:.1e}")
            F, x, g, H, nb = best[0], best[1], best[2], best[3], best[4]
            lam = lam * max(shrink, 1.0 - (2.0 * gain - 1.0) ** 3); nu = 2.0
            if dF <= ftol * abs(F): break
        else:
            rej = True
            if verbose: print(f"it {it:2d} F {F:.6e} rejected {Ft:.6e} lam {lam:.1e}")
            lam *= nu; nu *= 2.0
    return x, F, it, nev

from oracle import loss as oloss
_orig = oloss.redescending_dloss
def patched(theta, floor):
    def f(err, a, b, c):
        rho, drho, h = _orig(err, a, b, c)
        e = np.abs(np.asarray(err, dtype=np.float64))
        d = 1e-5
        _r1, d1, _h1 = _orig(e + d, a, b, c)
        _r0, d0, _h0 = _orig(np.maximum(e - d, 0.0), a, b, c)
        h2 = (d1 - d0) / (e + d - np.maximum(e - d, 0.0))
        hn = np.clip(theta * h2 + (1 - theta) * h, floor * h, 1.0)
        return rho, drho, hn
    return f
xref = None
for theta, floor in ((0.0, 1.0), (0.5, 0.1), (1.0, 0.1), (1.0, 0.3), (1.0, 0.01)):
    oloss.redescending_dloss = patched(theta, floor)
    t0 = time.time()
    x, F, it, nev = solve([], verbose=(theta == 1.0 and floor == 0.1))
    if xref is None:
        xref = x
    print(f"theta {theta} floor {floor}: {it} iterations, {nev} evaluations, F = {F:.12e}, max |x - x_ref| = {np.abs(x - xref).max():.2e}, {time.time()-t0:.1f} s", flush=True)
