#!/usr/bin/env python3
"""Stress of the fused back-substitution tail (shared_gpu = False: workgroups spin on a progress counter) under
concurrency it is NOT meant for: four contexts on four streams plus LDS-heavy filler kernels.  The wait is bounded, so the
worst outcome must be status 6 (sync_timeout), never a hang; prints what happens."""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acinoset_amd import fte, synth
from acinoset_amd._lib import lib, check
seq = synth.make_sequence(10000, "loop")
rig = (seq["K"], seq["D"], seq["R"], seq["t"])
x0 = fte.triangulation_init(seq["det"], *rig, 0.5)[:, fte.ACTIVE]
wins = [(0, 2691), (2307, 5193), (4809, 7692), (7308, 10000)]
streams = [torch.cuda.Stream() for _ in wins]
filler = torch.cuda.Stream()
ctxs = []
for s, (w0, w1) in zip(streams, wins):
    with torch.cuda.stream(s):
        c = fte.FTEContext(seq["det"][w0:w1], *rig, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True, shared_gpu=False)
        c.set_x(x0[w0:w1])
        ctxs.append(c)
t0 = time.time()
for it in range(30):
    for _k in range(4):
        check(lib().acino_debug_poison_lds(512, 300, C.c_void_p(filler.cuda_stream)))
    for s, c in zip(streams, ctxs):
        with torch.cuda.stream(s):
            c.step()
torch.cuda.synchronize()
print(f"30 rounds in {time.time() - t0:.2f} s")
for c, w in zip(ctxs, wins):
    st = c.state()
    print(w, st["status_name"], "accepted", st["accepted"], "cost", st["cost"])
