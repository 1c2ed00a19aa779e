"""Phase stamps (100 MHz wall clock) inside the separator chain's kernels of one LM iteration at 10 000 frames:
k_sep_level (fused narrow level: workgroup bx of the level with T workgroups per node) and k_sep_tail (isolated node bx).
usage: sep_stamps.py [frames]"""
import sys
import torch
sys.path.insert(0, ".")
from acinoset_amd import fte, synth
from acinoset_amd._lib import check, lib, ptr
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
seq = synth.make_sequence(n, "loop")
rig = (seq["K"], seq["D"], seq["R"], seq["t"])
det = torch.as_tensor(seq["det"], device="cuda")
x0 = fte.triangulation_init(det, *rig, 0.5)[:, fte.ACTIVE]
c = fte.FTEContext(det, *rig, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True)
c.set_x(x0)
for _ in range(3):
    c.step()
print("plan", fte.solver_plan(c.params), "bcr_levels", c.params.bcr_levels)
dbg = torch.zeros(72, dtype=torch.int64, device="cuda")
check(lib().acino_fte_debug_stamps(c._h, ptr(dbg)))
lv_names = ["start", "operands staged", "chol80 done", "strips done", "products stored", "y / U stored (g = 0)"]
tl_names = ["start", "operands staged", "chol80 done", "x0 done", "sweeps done, x stored", "level node done"]
for T in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16):
    for wg in (0, 1, 2, 3, 57, 118, 200):
        dbg.zero_(); dbg[64] = wg; dbg[65] = T
        c.step(); torch.cuda.synchronize()
        d = dbg.cpu().numpy()
        if d[5] == 0:
            continue
        print(f"k_sep_level T={T} wg {wg}: " + ", ".join(f"{lv_names[k]} {(d[k] - d[0]) / 100.0:.2f}" for k in range(1, 6) if d[k]))
for wg in (0, 1, 30, 58, 59, 60, 100, 118, 119, 150, 200, 237):
    dbg.zero_(); dbg[64] = wg; dbg[65] = 100
    c.step(); torch.cuda.synchronize()
    d = dbg.cpu().numpy()
    if d[6] == 0:
        continue
    base = d[7]
    if d[0]:
        if d[8]:
            print(f"   sweep 2 of wg {wg}: starts {(d[8] - base) / 100.0:.2f}, neighbours' iterates in LDS {(d[11] - base) / 100.0:.2f}, "
                  f"new iterate stored {(d[12] - base) / 100.0:.2f}")
        print(f"k_sep_tail wg {wg} (isolated): enters {(d[6] - base) / 100.0:.2f}, " + ", ".join(f"{tl_names[k]} {(d[k] - base) / 100.0:.2f}" for k in range(1, 5) if d[k]))
    else:
        print(f"k_sep_tail wg {wg} (level node): enters {(d[6] - base) / 100.0:.2f}, neighbours solved {(d[1] - base) / 100.0:.2f}, done {(d[5] - base) / 100.0:.2f}")
c.close()
