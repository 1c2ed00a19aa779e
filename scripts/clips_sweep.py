"""Timing of the solver variants on config 5's chain (64 clips x 1000 frames laid end to end)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from acinoset_amd import fte, synth
seq = synth.make_sequence(1000, "trot")
rig = (seq["K"], seq["D"], seq["R"], seq["t"])
det3 = torch.as_tensor(seq["det"], device="cuda")
det64 = det3.repeat(64, 1, 1, 1)
x3 = fte.nose_line_init(det3, *rig, 0.5)[:, fte.ACTIVE]
x64 = np.tile(x3, (64, 1))
side = torch.cuda.Stream()
for arg in sys.argv[1:]:
    m, K, r = (int(v) for v in arg.split(","))
    kw = dict(chunk_nodes=m)
    if K >= 0:
        kw.update(bcr_levels=K, refine_sweeps=r)
    c = fte.FTEContext(det64, *rig, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True, clip_len=1000, **kw)
    with torch.cuda.stream(side):
        c.enable_graph(True)
        c.set_x(x64)
        for _ in range(3):
            c.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            c.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = c.state()
        c.enable_graph(False)
        c.profile_begin()
        for _ in range(3):
            c.step()
        prof = c.profile_end()
    print(arg, fte.solver_plan(c.params), f"levels={c.params.bcr_levels} r={c.params.refine_sweeps}: {1e3*dt/10:.3f} ms/step cost={st['cost']:.6f} status={st['status_name']} eps={st['trunc_eps']:.1e}",
          {k: round(1e3 * v["ms"] / 3) for k, v in prof.items() if v["launches"]}, flush=True)
    c.close()
