import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from acinoset_amd import fte, synth
from acinoset_amd._lib import lib, ptr, check
seq = synth.make_sequence(3000, "loop"); det = seq["det"]
x0 = fte.triangulation_init(det, seq["K"], seq["D"], seq["R"], seq["t"], 0.5)
ctx = fte.FTEContext(det, seq["K"], seq["D"], seq["R"], seq["t"], seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0)
ctx.set_x(x0[:, fte.ACTIVE])
dbg = torch.zeros(32, dtype=torch.int64, device="cuda")
check(lib().acino_fte_debug_stamps(ctx._h, ptr(dbg)))
for _ in range(3): ctx.step()
torch.cuda.synchronize()
d = dbg.cpu().numpy()
names = ["load/gen", "chol80", "trsm", "y", "store"]
print("ticks (100MHz => 10ns):", [(names[k], int(d[k+1]-d[k])) for k in range(5)], "total", int(d[5]-d[0]))
print("chol16_inv (kb=1) ticks", int(d[17]-d[16]))
