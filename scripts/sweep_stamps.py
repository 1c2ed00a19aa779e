"""Phase stamps (100 MHz wall clock) of one workgroup / one node of the chunk sweep (csrc/chunk.hip: k_chunk_sweep, two teams of
waves).  usage: sweep_stamps.py [workgroup] [node of the run]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from acinoset_amd import fte, synth
from acinoset_amd._lib import check, lib, ptr, stream_ptr
wg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
kk = int(sys.argv[2]) if len(sys.argv) > 2 else 3
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
dbg[65] = kk
check(lib().acino_fte_debug_stamps(c._h, ptr(dbg)))
c.step()
torch.cuda.synchronize()
d = dbg.cpu().numpy()
check(lib().acino_fte_debug_stamps(c._h, None))
c.close()
t0 = d[0]
names = {0: "D: iteration starts (loads of the next node requested)", 1: "D: U_k complete, strip waves through with Xg: queue of G_k opens",
         2: "D: G_k complete", 4: "D: next node built"}
for w in (0, 1, 5, 4):
    names[8 + w] = f"D wave {w}: node done"
    names[32 + w] = f"D wave {w}: no tile of G_k left to draw"
who = ["wave 2 (strip 1)", "wave 6 (strips 2, 0)", "wave 3 (strip 3)", "wave 7 (strip 4)"]
for r in range(4):
    names[16 + 4 * r] = f"S {who[r]}: G_k complete"; names[17 + 4 * r] = f"S {who[r]}: T = G F done"
    names[18 + 4 * r] = f"S {who[r]}: F^T T tiles done"; names[19 + 4 * r] = f"S {who[r]}: stencil done"
for kb in range(4):
    names[40 + 2 * kb] = f"chain: panel {kb} posted"; names[41 + 2 * kb] = f"chain: pivots {kb + 1} done"
    names[48 + kb] = f"helper L: trailing {kb}"; names[52 + kb] = f"helper U: trailing {kb}"; names[56 + kb] = f"helper L: starts trailing {kb}"
for i in sorted(names, key=lambda i: d[i]):
    if d[i]:
        print(f"{(d[i] - t0) / 100.0:8.2f} us  {names[i]}")
if d[70]:
    print(f"run-level (us from the workgroup's start): first node built {(d[60] - d[70]) / 100.0:.2f}, factored {(d[61] - d[70]) / 100.0:.2f}, "
          f"node loop done {(d[62] - d[70]) / 100.0:.2f}, workgroup done {(d[71] - d[70]) / 100.0:.2f}")
print("SIMD of waves 0..7:", [(int(d[67]) >> (4 * w)) & 3 for w in range(8)], " HW_ID of wave 0: %#x" % int(d[68]))
