"""Where config 3's end-to-end time goes (1 000 frames, nose-line init, fresh context): python scripts/config3_phases.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acinoset_amd import fte, synth
seq = synth.make_sequence(1000, "trot"); rig = (seq["K"], seq["D"], seq["R"], seq["t"])
d = torch.as_tensor(seq["det"], device="cuda")
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for rep in range(4):
    t0 = T(); x0 = fte.nose_line_init(d, *rig, 0.5); t1 = T()
    xa = x0[:, fte.ACTIVE] if not isinstance(x0, torch.Tensor) else x0[:, torch.as_tensor(fte.ACTIVE, device=x0.device)]
    ctx, cached = fte._context_for(d, *rig, seq["Ts"], False, dict(dlc_thresh=0.5)); t2 = T()
    ctx.set_x(xa); t3 = T()
    info = ctx.solve(200); t4 = T()
    out = ctx.result(); t5 = T()
    ctx.close(); t6 = T()
    print(f"rep {rep}: init {1e3*(t1-t0):.2f} | context {1e3*(t2-t1):.2f} | set_x {1e3*(t3-t2):.2f} | solve {1e3*(t4-t3):.2f} ({info['iter']} it, {info['status_name']}) | result {1e3*(t5-t4):.2f} | close {1e3*(t6-t5):.2f} | total {1e3*(t6-t0):.2f} ms", flush=True)
for rep in range(3):
    t0 = T(); res, info = fte.fte_solve(d[..., :2], d[..., 2], *rig, Ts=seq["Ts"], max_iter=200, return_numpy=False); t1 = T()
    print(f"fte_solve rep {rep}: {1e3*(t1-t0):.2f} ms, {info['iter']} it", flush=True)
