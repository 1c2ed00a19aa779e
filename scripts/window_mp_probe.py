#!/usr/bin/env python3
"""Overlapping windows, one PROCESS per shard over torch.distributed (gloo, all ranks on GPU 0): debug probe.
   python -m torch.distributed.run --nproc-per-node W scripts/window_mp_probe.py <frames> <halo> <iters>"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n, halo, iters = (int(v) for v in sys.argv[1:4])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
from acinoset_amd import dist as adist, fte, synth  # noqa: E402
seq = synth.make_sequence(n, "loop")
rig = (seq["K"], seq["D"], seq["R"], seq["t"])
x0 = fte.triangulation_init(seq["det"], *rig, 0.5)[:, fte.ACTIVE]
use_side = len(sys.argv) > 4 and sys.argv[4].startswith("side")
quiet = len(sys.argv) > 4 and sys.argv[4].endswith("quiet")
stream = torch.cuda.Stream() if use_side else torch.cuda.current_stream()
with torch.cuda.stream(stream):
    d, (w0, w1, n0, n1) = adist.make_windowed(torch.as_tensor(seq["det"]), *rig, seq["Ts"], rank, world, halo=halo, shared_gpu=True,
                                              ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True)
    d.set_x(torch.as_tensor(x0[w0:w1]))
    st = d.state()
    print(f"rank {rank} window {(w0, w1)} owned {(n0, n1)} initial cost {st['cost']:.6f}", flush=True)
    from acinoset_amd._lib import lib, ptr, stream_ptr, check
    nwin = w1 - w0
    buf = torch.zeros((nwin + 6, 25), dtype=torch.float64, device="cuda")
    for it in range(iters):
        d.step()
        if quiet:
            if it == iters - 1:
                st = d.state()
                print(f"rank {rank} after {iters} unsynchronised steps: cost {st['cost']:.6f} acc {st['accepted']} lam {st['lam']:.2e}", flush=True)
            continue
        st = d.state()
        for which in (0, 1):
            check(lib().acino_fte_copy_frames(d.ctx._h, which, 0, -3, nwin + 6, ptr(buf), stream_ptr()))
            bad = torch.isnan(buf).any(1).nonzero().flatten().cpu().numpy()
            if bad.size:
                print(f"rank {rank} it {it + 1} which {which}: NaN frames (window-local, halo rows = -3..-1) {bad[:6] - 3} ... {bad[-3:] - 3} n={bad.size}", flush=True)
        print(f"rank {rank} it {it + 1} partial {d._partial.cpu().numpy()[:4]} edges nan {int(torch.isnan(d._all_edges).sum())} own {int(torch.isnan(d._edges).sum())}", flush=True)
        if rank == 0 or it == 0:
            print(f"rank {rank} it {it + 1} cost {st['cost']:.6f} trial {st['cost_trial']:.6f} lam {st['lam']:.2e} acc {st['accepted']} pred {st['pred']:.3e} step {st['step_inf']:.3e}", flush=True)
dist.barrier()
dist.destroy_process_group()
