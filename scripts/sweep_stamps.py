"""Phase stamps (100 MHz wall clock) of one workgroup / one node of the chunk sweep.  usage: sweep_stamps.py [wg] [k]"""
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
dbg[66] = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
check(lib().acino_fte_debug_stamps(c._h, ptr(dbg)))
c.step()
torch.cuda.synchronize()
d = dbg.cpu().numpy()
check(lib().acino_fte_debug_stamps(c._h, None))
c.close()
t0 = d[0]
names = {0: "iter start", 1: "G in LDS", 2: "G stored, E^T G E in registers (barrier)", 3: "next node written", 4: "gradient norm published: parallel part starts",
         7: "end of node (barrier)"}
for w in range(8):
    names[8 + w] = f"wave {w}: parallel role done"; names[32 + w] = f"wave {w}: its tiles of G in registers"
names[2] = "loads of the next node's H, g, x have arrived"
for sw in range(5):
    names[16 + 4 * sw] = f"spike {sw}: W strip"; names[17 + 4 * sw] = f"spike {sw}: barrier 1"
    names[18 + 4 * sw] = f"spike {sw}: syrk"; names[19 + 4 * sw] = f"spike {sw}: T stored"
    if sw < 4: names[59 + sw] = f"spike {sw}: starts T = G F"
for kb in range(4):
    names[40 + 2 * kb] = f"chain: panel {kb} posted"; names[41 + 2 * kb] = f"chain: pivots {kb + 1} done"
    names[48 + kb] = f"helper 1: trailing {kb}"; names[52 + kb] = f"helper 2: trailing {kb}"; names[56 + kb] = f"helper 1: starts trailing {kb}"
import os
if os.environ.get("ACINO_SWEEP", "3") == "3":          # the two-team kernel: other phases behind the same slots
    names.update({0: "D: iteration starts (loads of the next node requested)", 1: "D: U_k complete, strip waves through with Xg", 2: "D: G_k published",
                  4: "D: next node built", 7: "", 3: "D wave 0: the LDS reads of its state pairs have arrived"})
    who = ["wave 2 (strip 1)", "wave 6 (strips 2, 0)", "wave 3 (strip 3)", "wave 7 (strip 4)"]
    for r in range(4):
        names[16 + 4 * r] = f"S {who[r]}: sees G_k"; names[17 + 4 * r] = f"S {who[r]}: T = G F done"
        names[18 + 4 * r] = f"S {who[r]}: F^T T tiles done"; names[19 + 4 * r] = f"S {who[r]}: stencil done"
    for w in range(8):
        names[8 + w] = f"D wave {w}: node done" if w in (0, 1, 5, 4) else ""
    for q in range(59, 63):
        names.pop(q, None)
    for w in (0, 1, 5, 4):
        names[32 + w] = f"D wave {w}: its tiles of G in registers"; names[24 + w] = f"D wave {w}: loads of the next node have arrived"
    for i, w in enumerate((0, 1, 5, 4)):
        names[60 + i] = f"D wave {w}: its state pairs of the next node written"
    for w in (2, 3, 6, 7):
        names.pop(32 + w, None)
for i in sorted(names, key=lambda i: d[i]):
    if d[i] and names[i]:
        print(f"{(d[i] - t0) / 100.0:8.2f} us  {names[i]}")

print("SIMD of waves 0..7:", [(int(d[67]) >> (4 * w)) & 3 for w in range(8)], " HW_ID of wave 0: %#x" % int(d[68]), " skip mask", int(d[66]))
