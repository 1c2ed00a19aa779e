#!/usr/bin/env python3
"""Overlapping windows on ONE GPU, shards driven in lock step (debug / convergence probe):
   python scripts/window_probe.py <frames> <world> <halo> <iters> [init]"""
import os
import sys
import threading

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from acinoset_amd import dist as adist, fte, synth  # noqa: E402
from test_gpu_parity import _LockStepComm  # noqa: E402

n, world, halo, iters = (int(v) for v in sys.argv[1:5])
init = sys.argv[5] if len(sys.argv) > 5 else "triangulation"
seq = synth.make_sequence(n, "loop")
rig = (seq["K"], seq["D"], seq["R"], seq["t"])
det = torch.as_tensor(seq["det"])
x0 = (fte.triangulation_init(seq["det"], *rig, 0.5) if init == "triangulation" else
      fte.nose_line_init(seq["det"], *rig, 0.5))[:, fte.ACTIVE]
one = fte.FTEContext(seq["det"], *rig, seq["Ts"])
one.set_x(x0)
print("single: initial cost", one.state()["cost"])
hist = []
for it in range(iters):
    one.step()
    hist.append(one.state())
box = _LockStepComm(world)
drv = []
for r in range(world):
    d, (w0, w1, n0, n1) = adist.make_windowed(det, *rig, seq["Ts"], r, world, halo=halo, comm=box.rank(r), shared_gpu=True)
    drv.append((d, w0, w1))
    print("rank", r, "window", (w0, w1), "owned", (n0, n1))
box.run([lambda d=d, w0=w0, w1=w1: d.set_x(x0[w0:w1]) for d, w0, w1 in drv])
print("windows: initial cost", [d.state()["cost"] for d, *_ in drv])
for it in range(iters):
    box.run([d.step for d, *_ in drv])
    st = drv[0][0].state()
    print(it + 1, f"windows cost {st['cost']:.6f} trial {st['cost_trial']:.6f} lam {st['lam']:.2e} acc {st['accepted']} | single cost "
          f"{hist[it]['cost']:.6f} lam {hist[it]['lam']:.2e} acc {hist[it]['accepted']} {st['status_name']}/{hist[it]['status_name']}")
x = np.concatenate([d.result_x().cpu().numpy() for d, *_ in drv])
print("max |dx| vs single", np.abs(x - one.result()[0].cpu().numpy()).max())
