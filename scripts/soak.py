#!/usr/bin/env python3
"""Soak run: thousands of LM steps over several sequence lengths (graph replay, spin-waiting tail kernel included);
prints steps/s per size.  Run under `timeout` on the GPU box."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acinoset_amd import fte, synth
for N, steps in ((10000, 3000), (3331, 3000), (999, 4000), (190, 4000)):
    seq = synth.make_sequence(N, "loop"); rig = (seq["K"], seq["D"], seq["R"], seq["t"])
    x0 = fte.triangulation_init(seq["det"], *rig, 0.5)[:, fte.ACTIVE]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ctx = fte.FTEContext(seq["det"], *rig, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True)
        ctx.enable_graph(True); ctx.set_x(x0)
        t0 = time.perf_counter()
        for k in range(steps):
            ctx.step()
            if k % 500 == 499:
                st = ctx.state()
                assert st["status"] == 0 and np.isfinite(st["cost"]), st
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(N, "frames:", steps, "steps in %.2f s" % dt, "-> %.3f ms/step" % (1e3 * dt / steps), "cost", ctx.state()["cost"], flush=True)
        ctx.close()
