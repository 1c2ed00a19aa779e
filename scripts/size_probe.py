#!/usr/bin/env python3
"""Debug: LM steps on sub-windows of the benchmark sequence as stand-alone contexts (sizes that failed in the
multi-process window runs), alone and four at a time on separate streams."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acinoset_amd import fte, synth
seq = synth.make_sequence(10000, "loop")
rig = (seq["K"], seq["D"], seq["R"], seq["t"])
x0 = fte.triangulation_init(seq["det"], *rig, 0.5)[:, fte.ACTIVE]
wins = [(0, 2691, 0, 2499), (2307, 5193, 192, 2502), (4809, 7692, 192, 2499), (7308, 10000, 192, 2500)]
for shared in (True, False):
    for (w0, w1, of, oc) in wins:
        c = fte.FTEContext(seq["det"][w0:w1], *rig, seq["Ts"], n_global=10000, n_offset=w0, own_first=of, own_count=oc, shared_gpu=shared,
                           ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True)
        c.set_x(x0[w0:w1])
        for _ in range(4):
            c.step()
        st = c.state()
        print("alone shared", shared, (w0, w1), "status", st["status_name"], "acc", st["accepted"], "cost", st["cost"])
        c.close()
streams = [torch.cuda.Stream() for _ in wins]
ctxs = []
for s, (w0, w1, of, oc) in zip(streams, wins):
    with torch.cuda.stream(s):
        c = fte.FTEContext(seq["det"][w0:w1], *rig, seq["Ts"], n_global=10000, n_offset=w0, own_first=of, own_count=oc, shared_gpu=True,
                           ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True)
        c.set_x(x0[w0:w1])
        ctxs.append(c)
for it in range(6):
    for s, c in zip(streams, ctxs):
        with torch.cuda.stream(s):
            c.step()
torch.cuda.synchronize()
for c, w in zip(ctxs, wins):
    st = c.state()
    print("concurrent", w[:2], "status", st["status_name"], "acc", st["accepted"], "cost", st["cost"])
