"""Largest verified truncation bound (state.trunc_eps) over whole solves, by refinement-sweep count.  usage: refine_probe.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from acinoset_amd import fte, synth
cases = [("loop", 10000, 20210313), ("loop", 3331, 5), ("sprint", 4000, 1), ("trot", 1000, 2), ("walk", 6000, 3), ("loop", 999, 4), ("trot", 20000, 6)]
for kind, n, seed in cases:
    seq = synth.make_sequence(n, kind, seed=seed)
    rig = (seq["K"], seq["D"], seq["R"], seq["t"])
    det = torch.as_tensor(seq["det"], device="cuda")
    x0 = fte.triangulation_init_active(det, *rig, 0.5)
    for r in (3, 2):
        c = fte.FTEContext(det, *rig, seq["Ts"], refine_sweeps=r)
        c.set_x(x0)
        worst, its, lam_at = 0.0, 0, 0.0
        for it in range(60):
            c.step()
            st = c.state()
            if st["trunc_eps"] > worst:
                worst, lam_at = st["trunc_eps"], st["lam"]
            its = st["iter"]
            if st["status"] != 0:
                break
        print(f"{kind:7s} N={n:6d} r={r}: worst trunc_eps {worst:.2e} (lam {lam_at:.1e}) over {its} iterations, status {st['status_name']}, plan {fte.solver_plan(c.params)}", flush=True)
        c.close()
