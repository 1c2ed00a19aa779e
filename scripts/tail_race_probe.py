"""Run-to-run reproducibility of thousands of LM steps under foreign load (the soak test's body), many repetitions:
usage: tail_race_probe.py frames steps reps"""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, ".")
from acinoset_amd import fte, synth
from acinoset_amd._lib import check, lib
n, steps, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
seq = synth.make_sequence(n, "loop")
rig = (seq["K"], seq["D"], seq["R"], seq["t"])
det = torch.as_tensor(seq["det"], device="cuda")
x0 = fte.triangulation_init(det, *rig, 0.5)[:, fte.ACTIVE]
main, foreign = torch.cuda.Stream(), torch.cuda.Stream()
ref, bad = None, 0
for rep in range(reps):
    with torch.cuda.stream(main):
        ctx = fte.FTEContext(det, *rig, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0, clamp_lambda=True)
        ctx.enable_graph(True)
        ctx.set_x(x0)
    for k in range(steps):
        with torch.cuda.stream(main):
            ctx.step()
        if k % 16 == 0:
            check(lib().acino_debug_poison_lds(48, 4, ctypes.c_void_p(foreign.cuda_stream)))
    with torch.cuda.stream(main):
        st = ctx.state()
        x = ctx.result()[0].cpu().numpy()
        ctx.close()
    torch.cuda.synchronize()
    key = (st["cost"], st["lam"], st["accepted"], st["status"])
    if ref is None:
        ref = (key, x)
    elif key != ref[0] or not np.array_equal(x, ref[1]):
        bad += 1
        print("rep", rep, "differs:", key, "vs", ref[0], "max |dx|", float(np.abs(x - ref[1]).max()), flush=True)
print(f"n={n}: {reps} repetitions of {steps} steps, {bad} differ from the first", flush=True)
