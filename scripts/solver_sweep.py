"""Timing of the linear-solver variants on the benchmark sequence: chunk length, separator-chain levels, refinement sweeps.
usage: python scripts/solver_sweep.py [frames] "m,K,r" "m,K,r" ...   (m = chunk_nodes, K = bcr_levels, r = refine_sweeps)"""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
from acinoset_amd import fte, synth

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
cfgs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]] or [(-1, 0, 0), (0, 0, 0)]
seq = synth.make_sequence(frames, "loop")
rig = (seq["K"], seq["D"], seq["R"], seq["t"])
dev = torch.device("cuda")
det = torch.as_tensor(seq["det"], device=dev)
x0 = fte.triangulation_init(det, *rig, 0.5)[:, fte.ACTIVE]
side = torch.cuda.Stream()
for (m, K, r) in cfgs:
    c = fte.FTEContext(det, *rig, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True, chunk_nodes=m, bcr_levels=K,
                       refine_sweeps=r, trunc_tol=1e-12)
    with torch.cuda.stream(side):
        c.enable_graph(True)
        c.set_x(x0)
        for _ in range(3):
            c.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            c.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = c.state()
        c.enable_graph(False)
        c.profile_begin()
        for _ in range(5):
            c.step()
        prof = c.profile_end()
    plan = fte.solver_plan(c.params)
    c.close()
    ks = {k: round(1e3 * v["ms"] / 5, 1) for k, v in prof.items() if v["launches"]}
    print(f"m={m} K={K} r={r} plan={plan}: {1e6 * dt / 20:.1f} us/step  cost23={st['cost']:.10f} acc={st['accepted']} "
          f"status={st['status_name']} eps={st['trunc_eps']:.2e}\n    us/step by kernel: {ks}", flush=True)
