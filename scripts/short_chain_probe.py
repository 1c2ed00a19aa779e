#!/usr/bin/env python3
"""Short chains: ms per LM iteration under graph replay and the per-kernel event times, for several sequence lengths and
plans.  usage: short_chain_probe.py [frames ...]   (env CHUNK=<m> forces the run length)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acinoset_amd import fte, synth
sizes = [int(a) for a in sys.argv[1:]] or [190, 400, 999, 1500, 3331, 10000]
kw = {}
if os.environ.get("CHUNK"):
    kw["chunk_nodes"] = int(os.environ["CHUNK"])
if os.environ.get("LEVELS"):
    kw["bcr_levels"] = int(os.environ["LEVELS"])
for N in sizes:
    seq = synth.make_sequence(N, "loop"); rig = (seq["K"], seq["D"], seq["R"], seq["t"])
    x0 = fte.triangulation_init(seq["det"], *rig, 0.5)[:, fte.ACTIVE]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ctx = fte.FTEContext(seq["det"], *rig, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True, **kw)
        ctx.enable_graph(True); ctx.set_x(x0)
        for _ in range(5):
            ctx.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200):
            ctx.step()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        st = ctx.state()
        ctx.enable_graph(False)
        ctx.profile_begin()
        for _ in range(20):
            ctx.step()
        torch.cuda.synchronize()
        prof = ctx.profile_end()
        print(f"{N} frames: {1e3 * dt / 200:.4f} ms/step  plan {fte.solver_plan(ctx.params)} bcr_levels {ctx.params.bcr_levels} status {st['status_name']} "
              f"trunc_eps {st['trunc_eps']:.1e}\n     kernels (us/step, launches/step): " +
              ", ".join(f"{k} {1e3 * v['ms'] / 20:.1f} ({v['launches'] / 20:g})" for k, v in prof.items() if v["launches"]), flush=True)
        ctx.close()
