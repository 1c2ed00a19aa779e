#!/usr/bin/env python3
"""In-kernel phase timing of k_bcr_elim: wall-clock stamps (100 MHz ticks) of ONE workgroup of the launch at a given
level, under the full load of a real step.   python scripts/gpu_stamps.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from acinoset_amd import fte, synth
from acinoset_amd._lib import lib, ptr, check
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
seq = synth.make_sequence(n, "loop"); det = seq["det"]
x0 = fte.triangulation_init(det, seq["K"], seq["D"], seq["R"], seq["t"], 0.5)
ctx = fte.FTEContext(det, seq["K"], seq["D"], seq["R"], seq["t"], seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0, chunk_nodes=-1)   # (k_bcr_elim levels: the whole-chain solver)
ctx.set_x(x0[:, fte.ACTIVE])
dbg = torch.zeros(72, dtype=torch.int64, device="cuda")
check(lib().acino_fte_debug_stamps(ctx._h, ptr(dbg)))
names = ["load / build", "chol80", "wave 0: two W_l strips (coalesced B operand)", "wave 0: one W_r strip (transposed B operand) [level 0: z and G]", "store"]
for level in (0, 1, 2, 3):
    for wg in (0, 100, 700):
        dbg.zero_(); dbg[64] = wg; dbg[65] = level
        for _ in range(2): ctx.step()
        torch.cuda.synchronize()
        d = dbg.cpu().numpy()
        if d[5] == 0:
            continue
        print(f"level {level} workgroup {wg}: " + ", ".join(f"{names[k]} {10 * int(d[k + 1] - d[k])} ns" for k in range(5)) +
              f"; total {10 * int(d[5] - d[0])} ns; one 16x16 tile chain {10 * int(d[17] - d[16])} ns")
