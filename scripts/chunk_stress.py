"""Stress: many chain lengths / run lengths through the chunked solver, step per node against the whole-chain reduction."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from acinoset_amd import fte, synth
from acinoset_amd._lib import check, lib, ptr, stream_ptr
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
worst = 0.0
for it in range(reps):
    n = int(rng.integers(3, 400)); m = int(rng.choice([0, 2, 3, 4, 5, 7, 9, 14]))
    seq = synth.make_sequence(n, "sprint" if n < 200 else "loop")
    rig = (seq["K"], seq["D"], seq["R"], seq["t"])
    xa = np.clip(seq["q_true"] + rng.normal(0, 0.02, seq["q_true"].shape), *fte.bounds45())[:, fte.ACTIVE]
    out = {}
    for tag, cn in (("bcr", -1), ("chunk", m)):
        ctx = fte.FTEContext(seq["det"], *rig, seq["Ts"], ftol=0.0, xtol=0.0, gtol=0.0, chunk_nodes=cn)
        ctx.set_x(xa)
        for _ in range(3 if tag == "chunk" else 1):      # (repeat: races show up as run-to-run differences too)
            check(lib().acino_fte_reduce_local(ctx._h, stream_ptr()))
            check(lib().acino_fte_backsub_local(ctx._h, C.c_void_p(0), 0, 1, stream_ptr()))
        T = (n + 2) // 3
        buf = torch.zeros(T * 80, dtype=torch.float64, device="cuda")
        check(lib().acino_fte_debug_read(ctx._h, 0, ptr(buf), T * 80, stream_ptr()))
        torch.cuda.synchronize()
        out[tag] = buf.cpu().numpy().reshape(T, 80)[:, :75]
        ctx.close()
    d = np.abs(out["bcr"] - out["chunk"]).max() / np.abs(out["bcr"]).max()
    worst = max(worst, d)
    print(f"n={n} m={m} rel diff {d:.2e}", flush=True)
    assert d < 1e-8, (n, m, d)
print("worst", worst)
