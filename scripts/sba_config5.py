"""Config 5's SBA half (64 x 1 000 frames: 1.28 M points, 6.5 M observations, six shared extrinsics) as a stand-alone
workload for rocprofv3: python scripts/sba_config5.py [f64|bf16] [outer iterations]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acinoset_amd import sba, synth
prec = sys.argv[1] if len(sys.argv) > 1 else "f64"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
def rod(v):
    th = np.linalg.norm(v); k = v / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
seq = synth.make_sequence(1000, "trot")
r3 = (seq["K"], seq["D"], seq["R"], seq["t"])
dev = torch.device("cuda")
det64 = torch.as_tensor(seq["det"], device=dev).repeat(64, 1, 1, 1)
rng = np.random.default_rng(7)
pos64 = torch.as_tensor(seq["pos_true"], device=dev).repeat(64, 1, 1)
pos64 = pos64 + 0.005 * torch.randn(pos64.shape, dtype=torch.float64, device=dev)
Rp = np.array([rod(rng.normal(0, 1, 3) / np.sqrt(3) * np.radians(0.5)) @ r3[2][c] for c in range(len(r3[2]))])
tp = np.asarray(r3[3], dtype=np.float64).reshape(-1, 3, 1) + rng.normal(0, 1, (len(r3[2]), 3, 1)) / np.sqrt(3) * 1e-2
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _pts, _r, _t, info = sba.bundle_adjust_dense_points_and_extrinsics(det64, pos64, r3[0], r3[1], Rp, tp, 0.5, precision=prec, max_iter=iters, ftol=0.0, gtol=0.0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(prec, "%.2f ms total, %.3f ms / outer iteration" % (1e3 * dt, 1e3 * dt / max(info["iterations"], 1)),
          {k: info[k] for k in ("iterations", "accepted", "n_points", "n_obs", "rms_before", "rms_after", "cost_initial", "cost_final")}, flush=True)
