#!/usr/bin/env python3
"""Path comparison for the overlapping-window driver: dist.WindowedFTE over the HIP window backend and over the numpy
OracleWindowBackend (tests/oracle_backend.py) in lock step on random (frames, world, halo, cameras, start) - same decisions of
the replicated controller, same global cost after every iteration.  usage: fuzz_window_path.py first_seed n_seeds [iterations]"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from acinoset_amd import dist as adist, fte, synth  # noqa: E402
from oracle_backend import OracleWindowBackend  # noqa: E402

spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
tgp = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tgp)

first, count = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    world = int(rng.integers(2, 5))
    halo = 3 * int(rng.integers(2, 25))                        # 6 ... 72 frames
    n = world * int(rng.integers(halo + 3, halo + 80)) + int(rng.integers(0, world))
    cams = np.sort(rng.choice(6, size=int(rng.integers(2, 7)), replace=False))
    kind = ("sprint", "trot", "loop")[int(rng.integers(0, 3))]
    seq = synth.make_sequence(n, kind, seed=seed)
    det = seq["det"][:, cams].copy()
    det[rng.random(det.shape[:3]) < rng.uniform(0, 0.3), 2] = 0.0
    rig = tuple(a[cams] for a in (seq["K"], seq["D"], seq["R"], seq["t"]))
    lo, hi = fte.bounds45()
    mode = int(rng.integers(0, 2))
    if mode == 0:
        x0 = np.zeros((n, 45))
        x0[:, :3] = seq["q_true"][:, :3] + rng.normal(0, 0.05, (n, 3))
        x0[:, 31] = seq["q_true"][:, 31]
        x0 = x0[:, fte.ACTIVE]
    else:
        x0 = seq["q_true"][:, fte.ACTIVE] + rng.normal(0, 0.05, (n, 25))
    x0 = np.clip(x0, lo[fte.ACTIVE], hi[fte.ACTIVE])
    dett = torch.as_tensor(det)
    box_h, box_o = tgp._LockStepComm(world), tgp._LockStepComm(world)
    hip, ora = [], []
    for r in range(world):
        d, (w0, w1, n0, n1) = adist.make_windowed(dett, *rig, seq["Ts"], r, world, halo=halo, comm=box_h.rank(r), shared_gpu=True,
                                                  ftol=0.0, xtol=0.0, gtol=0.0)
        hip.append((d, w0, w1))
        be = OracleWindowBackend(det[w0:w1], *rig, seq["Ts"], n, w0, n0 - w0, n1 - n0, ftol=0.0, xtol=0.0, gtol=0.0)
        ora.append((adist.WindowedFTE(be, r, world, (n0 - w0, n1 - n0), halo, comm=box_o.rank(r)), w0, w1))
    box_h.run([lambda d=d, a=a, b=b: d.set_x(x0[a:b]) for d, a, b in hip])
    box_o.run([lambda d=d, a=a, b=b: d.set_x(torch.as_tensor(x0[a:b])) for d, a, b in ora])
    worst, where = abs(hip[0][0].state()["cost"] - ora[0][0].b.state()["cost"]) / abs(ora[0][0].b.state()["cost"]), 0
    for it in range(steps):
        box_h.run([d.step for d, *_ in hip])
        box_o.run([d.step for d, *_ in ora])
        sh, so = hip[0][0].state(), ora[0][0].b.state()
        if sh["accepted"] != so["accepted"] or sh["status"] != so["status"]:
            worst, where = float("inf"), it + 1
            break
        d = abs(sh["cost"] - so["cost"]) / abs(so["cost"])
        if d > worst:
            worst, where = d, it + 1
        if so["status"] != 0:
            break
    for d, *_ in hip:
        d.ctx.close()
    ok = worst < 1e-7
    bad += not ok
    print(seed, f"{n} frames x{world}, halo {halo}, {kind}, cameras {[int(c) for c in cams]}, start {('line', 'near')[mode]}: worst rel cost difference "
          f"{worst:.1e} at it {where}", "ok" if ok else "MISMATCH", flush=True)
print("mismatches:", bad)
