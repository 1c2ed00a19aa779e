"""Reduction levels of the separator chain against refinement sweeps: time per LM iteration (HIP-graph replay) and the largest
verified truncation bound (state.trunc_eps) over a whole solve, per (bcr_levels, refine_sweeps).
usage: levels_probe.py [frames] [kind] [seed]"""
import sys, time
import torch
sys.path.insert(0, ".")
from acinoset_amd import fte, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
kind = sys.argv[2] if len(sys.argv) > 2 else "loop"
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 20210313
base = fte.auto_bcr_levels(fte.make_params(n, 6, 1 / 120), fte.FTEContext.TRUNC_DISTANCE)
seq = synth.make_sequence(n, kind, seed=seed)
rig = (seq["K"], seq["D"], seq["R"], seq["t"])
det = torch.as_tensor(seq["det"], device="cuda")
x0 = fte.triangulation_init_active(det, *rig, 0.5)
arg = sys.argv[4] if len(sys.argv) > 4 else ''
combos = [tuple(int(v) for v in c.split(':')) for c in arg.split(',')] if arg else [(2, 3), (1, 3), (1, 5), (1, 6), (1, 7), (1, 8), (1, 10)]
for K, r in combos:
    # the whole solve, state read every iteration: the worst bound and where it happened
    c = fte.FTEContext(det, *rig, seq["Ts"], bcr_levels=K, refine_sweeps=r, trunc_tol=1e-6)
    c.set_x(x0)
    worst, lam_at, it_at = 0.0, 0.0, 0
    for it in range(80):
        c.step()
        st = c.state()
        if st["trunc_eps"] > worst:
            worst, lam_at, it_at = st["trunc_eps"], st["lam"], st["iter"]
        if st["status"] != 0:
            break
    its, cost, status = st["iter"], st["cost"], st["status_name"]
    c.close()
    # time per iteration, no convergence tests, graph replay
    c = fte.FTEContext(det, *rig, seq["Ts"], bcr_levels=K, refine_sweeps=r, trunc_tol=1e-6, ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True)
    c.enable_graph(True)
    c.set_x(x0)
    for _ in range(20):
        c.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        c.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 200 * 1e3
    c.close()
    print(f"N={n} {kind}: levels {K} sweeps {r}: {ms:.4f} ms/step | worst trunc_eps {worst:.2e} (iteration {it_at}, lam {lam_at:.1e}), "
          f"{its} iterations, {status}, cost {cost!r}", flush=True)
