#!/usr/bin/env python3
"""Widened rows on the GPU box -> profiles/<round>/sba_report.json, ekf_report.json:
timings and end states of the SBA on the KAT-2 fixtures' shape and of the EKF + RTS smoother on synthetic clips.
  python scripts/extras_report.py profiles/round4"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acinoset_amd import ekf, sba, synth
dst = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/extras"
os.makedirs(dst, exist_ok=True)

def T():
    torch.cuda.synchronize(); return time.perf_counter()

# ---- EKF + smoother: one long clip, a batch of clips
rep = {}
for tag, n, b in (("one_clip_10000_frames", 10000, 1), ("64_clips_x_1000_frames", 1000, 64)):
    seq = synth.make_sequence(n, "walk")        # the 2 m/s circle: within reach of the filter's constant-acceleration model
    rig = (seq["K"], seq["D"], seq["R"], seq["t"])
    dets = [torch.as_tensor(seq["det"], device="cuda")] * b
    for _ in range(2):
        t0 = T()
        out = ekf.ekf_batch(dets, *rig, 1.0 / seq["Ts"], 0.5, (2704, 1520), with_positions=False)
        dt = T() - t0
    rep[tag] = dict(seconds_incl_transfers=dt, frames=n * b, frames_per_s=n * b / dt, us_per_frame_per_clip=1e6 * dt / n)
json.dump(rep, open(os.path.join(dst, "ekf_report.json"), "w"), indent=1)
print("ekf:", rep)

# ---- SBA: points + extrinsics on a synthetic board scene of the KAT-2 size, and config 5's dense form
srep = {}
seq = synth.make_sequence(1000, "trot")
r3 = (seq["K"], seq["D"], seq["R"], seq["t"])
det64 = torch.as_tensor(seq["det"], device="cuda").repeat(64, 1, 1, 1)
pos64 = torch.as_tensor(seq["pos_true"], device="cuda").repeat(64, 1, 1)
pos64 = pos64 + 0.005 * torch.randn(pos64.shape, dtype=torch.float64, device="cuda")
rng = np.random.default_rng(7)
def rod(v):
    th = np.linalg.norm(v); k = v / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
Rp = np.array([rod(rng.normal(0, 1, 3) / np.sqrt(3) * np.radians(0.5)) @ r3[2][c] for c in range(len(r3[2]))])
tp = np.asarray(r3[3], dtype=np.float64).reshape(-1, 3, 1) + rng.normal(0, 1, (len(r3[2]), 3, 1)) / np.sqrt(3) * 1e-2
for prec in ("f64", "bf16"):
    for _ in range(2):
        t0 = T()
        _p, _r, _t, info = sba.bundle_adjust_dense_points_and_extrinsics(det64, pos64, r3[0], r3[1], Rp, tp, 0.5, precision=prec, max_iter=10, ftol=0.0, gtol=0.0)
        dt = T() - t0
    it = max(info["iterations"], 1)
    srep["config5_64x1000_" + prec] = dict(seconds=dt, ms_per_outer_iteration=1e3 * dt / it, n_points=info["n_points"], n_obs=info["n_obs"],
                                           rms_before_px=info["rms_before"], rms_after_px=info["rms_after"],
                                           bytes_moved_per_observation=112, compulsory_bytes_per_observation=40,
                                           hbm_fraction_on_bytes_moved=112.0 * info["n_obs"] * it / dt / 8e12,
                                           fp64_flop_fraction=2.6e3 * info["n_obs"] * it / dt / 78.6e12)
json.dump(srep, open(os.path.join(dst, "sba_report.json"), "w"), indent=1)
print("sba:", srep)
