#!/usr/bin/env python3
"""One rank's work in the sharded solves, measured on ONE GPU for world = 2, 4, 8 at 10 000 frames (the north star's size: a
latency-bound iteration) and at 40 000 / 80 000 frames (where one GPU is throughput-bound and sharding pays).
usage: shard_model.py [frames ...]   (default 10000 40000 80000; every line is prefixed "frames F | ").  Separator-system driver
(ShardedFTE): all ranks as real pinned contexts stepped in lock step, collectives emulated on the device and not timed, the
four phases of an interior rank timed with HIP events - with the whole-chain reduction (what pinned ranks ran until round 3)
and with the chunked sweep (round 4).  Overlapping windows (WindowedFTE): an interior rank's window as a single context."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acinoset_amd import fte, synth
from acinoset_amd._lib import lib, ptr, check, stream_ptr, SEP_DOUBLES, BS

SIZES = [int(a) for a in sys.argv[1:]] or [10000, 40000, 80000]
N = seq = rig = x0 = None
def load(n):
    global N, seq, rig, x0
    N = n
    seq = synth.make_sequence(N, "loop"); rig = (seq["K"], seq["D"], seq["R"], seq["t"])
    x0 = fte.triangulation_init(seq["det"], *rig, 0.5)[:, fte.ACTIVE]
def single_gpu_step():
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ctx = fte.FTEContext(seq["det"], *rig, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True)
        ctx.enable_graph(True); ctx.set_x(x0)
        for _ in range(5): ctx.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): ctx.step()
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 50
        ctx.close()
    return 1e3 * t


def sharded_rank_step(world, chunk_nodes, steps=30):
    """All `world` ranks of the separator-system driver as REAL pinned contexts on this one GPU, stepped in lock step in one
    thread; the collectives are emulated on the device (sum / concatenation of the ranks' buffers) and NOT timed.  Returns
    the HIP-event time of the four graph-replayed phases of an interior rank (reduce | separator solve + back-substitution +
    trial | halo + assembly + sums | control) per iteration."""
    plan = adist.shard_plan(N, world)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        bes = []
        for r, (n0, n1) in enumerate(plan):
            be = adist.HipBackend(torch.as_tensor(seq["det"][n0:n1]), *rig, seq["Ts"], N, n0, r, world, ftol=0.0, xtol=0.0, gtol=0.0,
                                  clamp_lambda=True, chunk_nodes=chunk_nodes)
            be.enable_graph(True)
            bes.append(be)
        new = bes[0].new
        sep = [new(world - 1, SEP_DOUBLES) for _ in bes]
        sep_x = [new(world - 1, BS) for _ in bes]
        edges = [new(6, 25) for _ in bes]
        all_edges = [new(world, 6, 25) for _ in bes]
        part = [new(8) for _ in bes]
        all_part = [new(world, 8) for _ in bes]
        def halo_and_eval(which):
            for r, be in enumerate(bes): be.export_edges(which, edges[r])
            cat = torch.stack(edges)
            for r, be in enumerate(bes):
                all_edges[r].copy_(cat)
        for r, (be, (n0, n1)) in enumerate(zip(bes, plan)):
            be.load_x(torch.as_tensor(x0[n0:n1]))
        halo_and_eval(0)
        for r, be in enumerate(bes):
            be.phase_eval(0, all_edges[r], part[r])
        cat = torch.stack(part)
        for r, be in enumerate(bes):
            all_part[r].copy_(cat); be.phase_control(all_part[r], True)
        mid = world // 2 if world > 2 else 1 if world == 2 else 0
        ev = []
        for it in range(steps + 5):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
            for r, be in enumerate(bes):
                if r == mid: e[0].record()
                be.phase_reduce(sep[r])
                if r == mid: e[1].record()
            tot = torch.stack(sep).sum(0)
            for r in range(world): sep[r].copy_(tot)
            for r, be in enumerate(bes):
                if r == mid: e[2].record()
                be.phase_solve(sep[r], sep_x[r], edges[r])
                if r == mid: e[3].record()
            cat = torch.stack(edges)
            for r in range(world): all_edges[r].copy_(cat)
            for r, be in enumerate(bes):
                if r == mid: e[4].record()
                be.phase_eval(1, all_edges[r], part[r])
                if r == mid: e[5].record()
            cat = torch.stack(part)
            for r in range(world): all_part[r].copy_(cat)
            for r, be in enumerate(bes):
                if r == mid: e[6].record()
                be.phase_control(all_part[r], False)
                if r == mid: e[7].record()
            if it >= 5: ev.append(e)
        torch.cuda.synchronize()
        ph = [sum(e[2 * k].elapsed_time(e[2 * k + 1]) for e in ev) / len(ev) for k in range(4)]
        st = bes[mid].state()
        assert st["status"] == 0 and st["accepted"] >= 10, st
        n_mid = plan[mid][1] - plan[mid][0]
        pl = fte.solver_plan(bes[mid].ctx.params)
        for be in bes: be.ctx.close()
    return n_mid, ph, pl, st["cost"]


from acinoset_amd import dist as adist
for size in SIZES:
    load(size)
    P = f"frames {N} | "
    t1 = single_gpu_step()
    print(f"{P}world 1: {N} frames, single-GPU step {t1:.3f} ms (hipGraph replay) = {N / t1 / 1e3:.1f} M frames/s", flush=True)
    for world in (2, 4, 8):
        variants = (("whole-chain reduction (rounds 1-3)", -1), ("chunked sweep, pins in the separator chain (round 4)", 0))
        for tag, cn in (variants if N <= 10000 else variants[1:]):
            n_mid, ph, pl, cost = sharded_rank_step(world, cn)
            tot = sum(ph)
            print(f"{P}world {world}: interior rank with {n_mid} frames, {tag} [runs of {pl['m']} nodes, {pl['n_sep']} separators]: "
                  f"reduce {ph[0]:.3f} + separator solve / back-substitution / trial {ph[1]:.3f} + halo / assembly / sums {ph[2]:.3f} + control "
                  f"{ph[3]:.3f} = {tot:.3f} ms per iteration (cost of rank {cost:.6f}) -> speed-up over 1 GPU without collectives "
                  f"{t1 / tot:.2f}x, with 3 x 25 us of collectives {t1 / (tot + 0.075):.2f}x", flush=True)

    # ---- overlapping windows (dist.WindowedFTE): an interior rank's work is the complete step on N/world + 2*halo frames;
    #      no separator solve; two small all-gathers (not included).  Also with the incomplete reduction on top.
    print()
    for halo in (96, 192):
        for world in (2, 4, 8):
            n = min(N, N // world + 2 * halo)
            o = min(1000, N - n)
            for tag, kw in (("complete reduction", dict(bcr_levels=0)), ("default (truncated + refined, verified)", {})):
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    ctx = fte.FTEContext(seq["det"][o:o + n], *rig, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True,
                                         n_global=N, n_offset=o, own_first=halo, own_count=n - 2 * halo, **kw)
                    ctx.enable_graph(True); ctx.set_x(x0[o:o + n])
                    for _ in range(5): ctx.step()
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for _ in range(50): ctx.step()
                    torch.cuda.synchronize(); tw = 1e3 * (time.perf_counter() - t0) / 50
                    stw = ctx.state()
                    assert stw["status"] == 0 and stw["accepted"] >= 15, stw
                    plan, K, r = fte.solver_plan(ctx.params), int(ctx.params.bcr_levels), int(ctx.params.refine_sweeps)
                    ctx.close()
                print(f"{P}windows: world {world}, halo {halo}: {n} frames/rank, {tag} (runs of {plan['m']} nodes, {plan['n_sep']} separators, "
                      f"levels {K}, sweeps {r}, bound {stw['trunc_eps']:.1e}): step {tw:.3f} ms -> speed-up over 1 GPU without collectives "
                      f"{t1 / tw:.2f}x, with 2 x 25 us of collectives {t1 / (tw + 0.05):.2f}x", flush=True)
    print()
