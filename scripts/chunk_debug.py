"""Debug: one linear solve by the chunked solver and by the whole-chain reduction, step per node compared."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from acinoset_amd import fte, synth
from acinoset_amd._lib import check, lib, ptr, stream_ptr

n, m = int(sys.argv[1]), int(sys.argv[2])
seq = synth.make_sequence(n, "sprint")
rig = (seq["K"], seq["D"], seq["R"], seq["t"])
x0 = np.zeros((n, 45))
x0[:, fte.ACTIVE] = seq["q_true"][:, fte.ACTIVE] + np.random.default_rng(n).normal(0, 0.02, (n, 25))
lo, hi = fte.bounds45()
xa = np.clip(x0, lo, hi)[:, fte.ACTIVE]
out = {}
for tag, cn in (("bcr", -1), ("chunk", m)):
    ctx = fte.FTEContext(seq["det"], *rig, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0, chunk_nodes=cn)
    ctx.set_x(xa)
    check(lib().acino_fte_reduce_local(ctx._h, stream_ptr()))
    check(lib().acino_fte_backsub_local(ctx._h, C.c_void_p(0), 0, 1, stream_ptr()))
    T = (n + 2) // 3
    buf = torch.zeros(T * 80, dtype=torch.float64, device="cuda")
    check(lib().acino_fte_debug_read(ctx._h, 0, ptr(buf), T * 80, stream_ptr()))
    torch.cuda.synchronize()
    out[tag] = buf.cpu().numpy().reshape(T, 80)
    ctx.close()
d = np.abs(out["bcr"] - out["chunk"])[:, :75].max(1)
s = np.abs(out["bcr"])[:, :75].max(1)
for t in range(len(d)):
    print(t, f"{d[t]:.3e}  (|delta| {s[t]:.3e})")
