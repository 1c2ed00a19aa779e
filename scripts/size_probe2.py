#!/usr/bin/env python3
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acinoset_amd import fte, synth
N = 10000
seq = synth.make_sequence(N, "loop")
rig = (seq["K"], seq["D"], seq["R"], seq["t"])
x0 = fte.triangulation_init(seq["det"], *rig, 0.5)[:, fte.ACTIVE]
for off, n, own in ((999, 5192, True), (999, 5193, True), (999, 5190, True), (0, 5192, True), (999, 5192, False), (1000, 5192, False), (2307, 2886, True), (999, 2886, True), (999, 2692, True)):
    kw = dict(own_first=96, own_count=n - 192) if own else {}
    c = fte.FTEContext(seq["det"][off:off + n], *rig, seq["Ts"], n_global=N, n_offset=off, ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True, **kw)
    c.set_x(x0[off:off + n])
    for _ in range(3):
        c.step()
    st = c.state()
    print("offset", off, "frames", n, "own", own, "->", st["status_name"], st["accepted"], st["cost"], flush=True)
    c.close()
