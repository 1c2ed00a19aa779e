#!/usr/bin/env python3
"""Debug: which piece of an LM step reads LDS it did not write?  LDS is poisoned with NaN on the SAME stream before
every piece; after each piece the whole workspace is scanned for NaN."""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acinoset_amd import fte, synth
from acinoset_amd._lib import lib, check, stream_ptr
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2691
seq = synth.make_sequence(n, "loop")
rig = (seq["K"], seq["D"], seq["R"], seq["t"])
x0 = fte.triangulation_init(seq["det"], *rig, 0.5)[:, fte.ACTIVE]
L = lib()
c = fte.FTEContext(seq["det"], *rig, seq["Ts"], shared_gpu=True, ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True)
c.workspace.zero_()
c.close()
c = fte.FTEContext(seq["det"], *rig, seq["Ts"], shared_gpu=True, ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True)
ws = c.workspace[(c._ws_ptr - c.workspace.data_ptr()):]
ws = ws[: ws.numel() // 8 * 8].view(torch.float64)
def poison():
    for _ in range(3):
        check(L.acino_debug_poison_lds(1024, 4, stream_ptr()))
def nans(tag):
    torch.cuda.synchronize()
    bad = torch.isnan(ws).nonzero().flatten()
    print(f"{tag}: NaN doubles in workspace {bad.numel()}" + (f" first at double offset {int(bad[0])}, last {int(bad[-1])}" if bad.numel() else ""), flush=True)
poison(); c.set_x(x0); nans("set_x (assemble + totals)")
for it in range(2):
    poison(); check(L.acino_fte_reduce_local(c._h, stream_ptr())); nans(f"it {it} reduce_local")
    poison(); check(L.acino_fte_backsub_local(c._h, C.c_void_p(0), 0, 1, stream_ptr())); nans(f"it {it} backsub_local")
    poison(); check(L.acino_fte_trial(c._h, stream_ptr())); nans(f"it {it} trial")
    poison(); check(L.acino_fte_eval(c._h, 1, stream_ptr())); nans(f"it {it} eval(1)")
    check(L.acino_fte_control(c._h, C.c_void_p(0), 0, stream_ptr())); nans(f"it {it} control")
    print(c.state())
