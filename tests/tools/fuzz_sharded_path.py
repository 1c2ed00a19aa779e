#!/usr/bin/env python3
"""The exact multi-GPU driver (dist.ShardedFTE: pinned separators, separator all-reduce + redundant solve, halos, global
control) run by `world` threads on ONE GPU against the single-shard HIP solve on random (frames, world, cameras, start): same
accepted count, cost and iterate after k steps.  usage: fuzz_sharded_path.py first_seed n_seeds [steps]"""
import importlib.util
import os
import sys
import threading

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from acinoset_amd import dist as adist, fte, synth  # noqa: E402

spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
tgp = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tgp)

first, count = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    world = int(rng.integers(2, 6))
    n = world * int(rng.integers(6, 60)) + int(rng.integers(0, world))
    cams = np.sort(rng.choice(6, size=int(rng.integers(2, 7)), replace=False))
    kind = ("sprint", "trot", "loop")[int(rng.integers(0, 3))]
    seq = synth.make_sequence(n, kind, seed=seed)
    det = seq["det"][:, cams].copy()
    det[rng.random(det.shape[:3]) < rng.uniform(0, 0.3), 2] = 0.0
    rig = tuple(a[cams] for a in (seq["K"], seq["D"], seq["R"], seq["t"]))
    lo, hi = fte.bounds45()
    mode = int(rng.integers(0, 3))
    x0 = np.zeros((n, 45))
    if mode == 0:
        x0[:, :3] = seq["q_true"][:, :3] + rng.normal(0, 0.05, (n, 3))
        x0[:, 31] = seq["q_true"][:, 31]
    else:
        x0[:, fte.ACTIVE] = seq["q_true"][:, fte.ACTIVE] + rng.normal(0, (0.0, 0.05, 0.5)[mode], (n, 25))
    x0 = np.clip(x0, lo, hi)[:, fte.ACTIVE]
    ref = fte.FTEContext(det, *rig, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0)
    ref.set_x(x0)
    for _ in range(steps):
        ref.step()
    x_ref, st_ref = ref.result()[0].cpu().numpy(), ref.state()
    ref.close()
    comm = tgp.ThreadComm(world)
    results, errors = [None] * world, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            comm.bind(rank)
            with torch.cuda.stream(torch.cuda.Stream()):
                drv, (n0, n1) = adist.make_sharded(torch.as_tensor(det), *rig, seq["Ts"], rank, world, comm=comm, ftol=0.0, xtol=0.0,
                                                   gtol=0.0, shared_gpu=True)
                drv.set_x(torch.as_tensor(x0[n0:n1]))
                for _ in range(steps):
                    drv.step()
                results[rank] = (drv.b.result_x().cpu().numpy(), drv.b.state())
        except Exception as exc:                                   # pragma: no cover
            errors.append(exc)
            comm.barrier.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    if errors:
        print(seed, f"{n} frames x{world}: ERROR {errors[0]!r}", "MISMATCH", flush=True)
        bad += 1
        continue
    x = np.concatenate([r[0] for r in results])
    same = all(r[1]["accepted"] == st_ref["accepted"] for r in results)
    dc = abs(results[0][1]["cost"] - st_ref["cost"]) / abs(st_ref["cost"])
    dx = np.abs(x - x_ref).max()
    ok = same and dc < 1e-8 and dx < 1e-6
    bad += not ok
    print(seed, f"{n} frames x{world}, {kind}, cameras {[int(c) for c in cams]}, start {('line', 'near', 'far')[mode]}: accepted {st_ref['accepted']} "
          f"{'=' if same else '!='} shards; rel cost diff {dc:.1e}; max |dx| {dx:.1e}", "ok" if ok else "MISMATCH", flush=True)
print("mismatches:", bad)
