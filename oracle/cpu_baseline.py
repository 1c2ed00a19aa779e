"""CPU-baseline worker of bench.py (test infrastructure, see oracle/__init__.py): `iters` Levenberg-Marquardt
iterations of the numpy/scipy oracle (oracle/fte.py) on a block of frames, timed.  Run as a module by bench.py -
`python -m oracle.cpu_baseline <npz> <first> <last> <iters>` prints the seconds of the loop - so that the all-cores
figure is N plain numpy processes that never import torch or touch the GPU."""
import sys
import time

import numpy as np


def lm_iterations(det, rig, Ts, x0_active, iters):
    """`iters` LM iterations (+ the initial evaluation) on one block of frames; returns seconds."""
    from . import fte as ofte
    K, D, R, t = rig
    prob = ofte.FTEProblem(det[..., :2], det[..., 2], K, D, R, t, Ts)
    x = np.clip(x0_active, prob.lo, prob.hi)
    t0 = time.perf_counter()
    F, g, H, _ = prob.evaluate(x)
    lam = 1e-3
    for _ in range(iters):
        fixed = ((x <= prob.lo) & (g > 0)) | ((x >= prob.hi) & (g < 0))
        delta, _diag = prob.solve_banded(H, g, lam, fixed)
        xt = np.clip(x + delta, prob.lo, prob.hi)
        Ft, gt, Ht, _ = prob.evaluate(xt)
        if Ft < F:
            x, F, g, H = xt, Ft, gt, Ht
            lam /= 3
        else:
            lam *= 2
    return time.perf_counter() - t0


def main(argv):
    path, a, b, iters = argv[0], int(argv[1]), int(argv[2]), int(argv[3])
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:                                  # pragma: no cover
        pass
    z = np.load(path)
    dt = lm_iterations(z["det"][a:b], (z["K"], z["D"], z["R"], z["t"]), float(z["Ts"]), z["xa"][a:b], iters)
    print(f"{dt:.6f}")


if __name__ == "__main__":
    main(sys.argv[1:])
