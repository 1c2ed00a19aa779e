"""CPU-baseline worker of bench.py (test infrastructure, see oracle/__init__.py): `iters` Levenberg-Marquardt
iterations of the numpy/scipy oracle (oracle/fte.py) on a block of frames, timed.  Run as a module by bench.py -
`python -m oracle.cpu_baseline <npz> <first> <last> <iters>` prints the seconds of the loop - so that the all-cores
figure is N plain numpy processes that never import torch or touch the GPU."""
import sys
import time

import numpy as np


def lm_iterations(det, rig, Ts, x0_active, iters, history=None):
    """`iters` LM iterations (+ the initial evaluation) of oracle.fte.lm_solve on one block of frames, stopping tests
    off; returns seconds.  `history` (a list) receives the per-iteration records (trial cost, lambda, gain, ...) -
    bench.py holds the GPU's first iterations against them at full size."""
    from . import fte as ofte
    K, D, R, t = rig
    prob = ofte.FTEProblem(det[..., :2], det[..., 2], K, D, R, t, Ts)
    t0 = time.perf_counter()
    ofte.lm_solve(prob, x0_active, max_iter=iters, ftol=0.0, xtol=0.0, gtol=0.0, history=history)
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
