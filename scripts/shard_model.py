#!/usr/bin/env python3
"""Single-GPU stand-in for one rank's work in the sharded solve: an unsharded context of N/world frames (local reduction
+ back-substitution + trial + assembly) and the separator-chain solve for world-1 separators, timed separately.  The
three collectives are NOT included.  Prints a table for world = 1, 2, 4, 8 at 10 000 frames."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acinoset_amd import fte, synth
from acinoset_amd._lib import lib, ptr, check, stream_ptr, SEP_DOUBLES, BS

N = 10000
seq = synth.make_sequence(N, "loop"); rig = (seq["K"], seq["D"], seq["R"], seq["t"])
x0 = fte.triangulation_init(seq["det"], *rig, 0.5)[:, fte.ACTIVE]
rows = []
for world in (1, 2, 4, 8):
    n = N // world
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ctx = fte.FTEContext(seq["det"][:n], *rig, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True)
        ctx.enable_graph(True); ctx.set_x(x0[:n])
        for _ in range(5): ctx.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): ctx.step()
        torch.cuda.synchronize(); t_local = (time.perf_counter() - t0) / 50
        ctx.close()
        t_sep = 0.0
        if world > 1:
            ns = world - 1
            rng = np.random.default_rng(0)
            sep = np.zeros((ns, SEP_DOUBLES))
            for k in range(ns):
                A = rng.normal(size=(BS, BS)); sep[k, :BS * BS] = (A @ A.T + 80 * np.eye(BS)).ravel()
                sep[k, BS * BS:2 * BS * BS] = 0.01 * rng.normal(size=BS * BS)
            d_sep = torch.as_tensor(sep, device="cuda"); d_x = torch.zeros(ns, BS, dtype=torch.float64, device="cuda")
            nb = lib().acino_sep_scratch_bytes(ns)
            scr = torch.empty(nb + 256, dtype=torch.uint8, device="cuda"); sp = (scr.data_ptr() + 255) // 256 * 256
            for _ in range(3): check(lib().acino_solve_separators(ptr(d_sep), ns, ptr(d_x), C.c_void_p(sp), nb, stream_ptr()))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50): check(lib().acino_solve_separators(ptr(d_sep), ns, ptr(d_x), C.c_void_p(sp), nb, stream_ptr()))
            torch.cuda.synchronize(); t_sep = (time.perf_counter() - t0) / 50
    rows.append((world, n, 1e3 * t_local, 1e3 * t_sep))
t1 = rows[0][2]
for world, n, tl, ts in rows:
    print(f"world {world}: {n} frames/rank, local step {tl:.3f} ms, separator solve {ts:.3f} ms (eager launches), "
          f"sum {tl + ts:.3f} ms -> speed-up over 1 GPU without collectives {t1 / (tl + ts):.2f}x")

# ---- overlapping windows (dist.WindowedFTE): an interior rank's work is the complete step on N/world + 2*halo frames;
#      no separator solve; two small all-gathers (not included).  Also with the incomplete reduction on top.
print()
for halo in (96, 192):
    for world in (2, 4, 8):
        n = min(N, N // world + 2 * halo)
        for tag, kw in (("complete reduction", dict(bcr_levels=0)), ("default (truncated + refined, verified)", {})):
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                ctx = fte.FTEContext(seq["det"][1000:1000 + n], *rig, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True,
                                     n_global=N, n_offset=1000, own_first=halo, own_count=n - 2 * halo, **kw)
                ctx.enable_graph(True); ctx.set_x(x0[1000:1000 + n])
                for _ in range(5): ctx.step()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(50): ctx.step()
                torch.cuda.synchronize(); tw = 1e3 * (time.perf_counter() - t0) / 50
                stw = ctx.state()
                assert stw["status"] == 0 and stw["accepted"] >= 15, stw
                plan, K, r = fte.solver_plan(ctx.params), int(ctx.params.bcr_levels), int(ctx.params.refine_sweeps)
                ctx.close()
            print(f"windows: world {world}, halo {halo}: {n} frames/rank, {tag} (runs of {plan['m']} nodes, {plan['n_sep']} separators, "
                  f"levels {K}, sweeps {r}, bound {stw['trunc_eps']:.1e}): step {tw:.3f} ms -> speed-up over 1 GPU without collectives "
                  f"{t1 / tw:.2f}x, with 2 x 25 us of collectives {t1 / (tw + 0.05):.2f}x")
