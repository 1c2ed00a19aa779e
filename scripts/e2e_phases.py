"""Where the end-to-end time of fte_solve goes at 10 000 frames: python scripts/e2e_phases.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acinoset_amd import fte, synth
seq = synth.make_sequence(10000, "loop"); rig = (seq["K"], seq["D"], seq["R"], seq["t"])
d = torch.as_tensor(seq["det"], device="cuda")
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for reuse in (False, True, True):
    t0 = T(); xa = fte.triangulation_init_active(d, *rig, 0.5); t1 = T()
    ctx, cached = fte._context_for(d, *rig, seq["Ts"], reuse, dict(dlc_thresh=0.5)); t2 = T()
    ctx.set_x(xa); t3 = T()
    info = ctx.solve(200); t4 = T()
    out = ctx.result(); t5 = T()
    if not cached: ctx.close()
    print(f"reuse={reuse}: init {1e3*(t1-t0):.2f} | context {1e3*(t2-t1):.2f} | set_x {1e3*(t3-t2):.2f} | solve {1e3*(t4-t3):.2f} ({info['iter']} it, {info['status_name']}) | result {1e3*(t5-t4):.2f} | total {1e3*(t5-t0):.2f} ms", flush=True)
for rep in range(3):
    t0 = T(); res, info = fte.fte_solve(d[..., :2], d[..., 2], *rig, Ts=seq["Ts"], max_iter=200, init="triangulation", return_numpy=False, reuse_context=True); t1 = T()
    print(f"fte_solve(reuse_context=True) rep {rep}: {1e3*(t1-t0):.2f} ms, {info['iter']} it, cost {info['cost']:.9f}", flush=True)
