#!/usr/bin/env python3
"""B independent sequences solved concurrently on ONE GPU, one HIP stream per sequence (the LM controller lives
on the device, so a step never syncs with the host and the streams interleave freely).  Prints aggregate frames/s."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acinoset_amd import fte, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
steps = 20
out = {}
seq = synth.make_sequence(N, "loop")
rig = (seq["K"], seq["D"], seq["R"], seq["t"])
x0 = fte.triangulation_init(seq["det"], *rig, 0.5)[:, fte.ACTIVE]
for B in (1, 2, 4, 8, 16):
    streams = [torch.cuda.Stream() for _ in range(B)]
    ctxs = []
    for s in streams:
        with torch.cuda.stream(s):
            c = fte.FTEContext(seq["det"], *rig, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True, shared_gpu=B > 1)
            c.enable_graph(True)
            c.set_x(x0)
            for _ in range(3):
                c.step()
            ctxs.append(c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for c, s in zip(ctxs, streams):
            with torch.cuda.stream(s):
                c.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out[B] = dict(ms_per_round=1e3 * dt / steps, frames_per_s=B * N * steps / dt)
    print(B, out[B], flush=True)
    for c in ctxs:
        c.close()
os.makedirs(os.path.join(ROOT, "gpurun_out", "batch"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "batch", f"probe_{N}.json"), "w"), indent=1)
