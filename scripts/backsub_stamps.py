"""Stamps (100 MHz wall clock) of one workgroup of the chunk back-substitution (csrc/chunk.hip: k_chunk_backsub, loader waves +
product waves): steps 8 .. 15 of product wave 0 - stage ready / vector ready / operands read / sums / next vector signalled /
step done - and jobs 10 .. 13 of the loaders - job starts (tiles requested a job earlier) / stage free and tiles arrived / staged.
usage: backsub_stamps.py [workgroup] [xmode]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from acinoset_amd import fte, synth
from acinoset_amd._lib import check, lib, ptr
wg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
xmode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
seq = synth.make_sequence(10000, "loop")
rig = (seq["K"], seq["D"], seq["R"], seq["t"])
det = torch.as_tensor(seq["det"], device="cuda")
x0 = fte.triangulation_init(det, *rig, 0.5)[:, fte.ACTIVE]
c = fte.FTEContext(det, *rig, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True)
c.set_x(x0)
for _ in range(3):
    c.step()
dbg = torch.zeros(72, dtype=torch.int64, device="cuda")
dbg[64] = wg
dbg[66] = 1
dbg[69] = xmode
check(lib().acino_fte_debug_stamps(c._h, ptr(dbg)))
c.step()
torch.cuda.synchronize()
d = dbg.cpu().numpy()
check(lib().acino_fte_debug_stamps(c._h, None))
c.close()
t0 = d[61]
print(f"kernel start 0.00, step loop starts {(d[62] - t0) / 100.0:.2f} us, ends {(d[63] - t0) / 100.0:.2f} us")
names = ["stage ready", "vector ready", "operands read", "sums", "vector signalled", "step done"]
for s in range(8):
    e = d[6 * s:6 * s + 6]
    if e[0] == 0:
        break
    print(f"step {s + 8:2d}: " + "  ".join(f"{n} {(v - t0) / 100.0:6.2f}" for n, v in zip(names, e)))
for j in range(4):
    e = d[48 + 4 * j:48 + 4 * j + 3]
    if e[0]:
        print(f"loader job {10 + j}: starts {(e[0] - t0) / 100.0:6.2f}  stage free + tiles arrived {(e[1] - t0) / 100.0:6.2f}  staged {(e[2] - t0) / 100.0:6.2f}")
