import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from acinoset_amd import fte, synth
from acinoset_amd._lib import check, lib, ptr, stream_ptr
n, m, what = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
seq = synth.make_sequence(n, "sprint")
rig = (seq["K"], seq["D"], seq["R"], seq["t"])
xa = seq["q_true"][:, fte.ACTIVE]
ctx = fte.FTEContext(seq["det"], *rig, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0, chunk_nodes=m)
ctx.set_x(xa)
torch.cuda.synchronize(); print("set_x ok", flush=True)
check(lib().acino_fte_reduce_local(ctx._h, stream_ptr()))
torch.cuda.synchronize(); print("reduce ok", flush=True)
if what == "all":
    check(lib().acino_fte_backsub_local(ctx._h, C.c_void_p(0), 0, 1, stream_ptr()))
    torch.cuda.synchronize(); print("backsub ok", flush=True)
