#!/usr/bin/env python3
"""EKF + RTS smoother timing on the GPU box: one 10 000-frame clip, 64 clips x 1 000
frames in one launch, and the numpy oracle on a short clip.  Writes gpurun_out/ekf/report.json."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from acinoset_amd import ekf, synth  # noqa: E402
from oracle import ekf as oekf  # noqa: E402

out = {}
seq = synth.make_sequence(10000, "walk")
rig = (seq["K"], seq["D"], seq["R"], seq["t"])
det = torch.as_tensor(seq["det"], device="cuda")
s0 = ekf.initial_state(det, *rig, 120.0, 0.5)
ekf.ekf(det[:100], *rig, 120.0, 0.5, (2704, 1520), states0=s0, with_positions=False)     # warm
for name, dets in (("one_clip_10000_frames", [det]), ("64_clips_x_1000_frames", [det[:1000]] * 64)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = ekf.ekf_batch(dets, *rig, 120.0, 0.5, (2704, 1520), with_positions=False)
    dt = time.perf_counter() - t0
    frames = sum(int(d.shape[0]) for d in dets)
    out[name] = dict(seconds_incl_transfers=dt, frames=frames, frames_per_s=frames / dt,
                     us_per_frame_per_clip=1e6 * dt / int(dets[0].shape[0]))
    if len(dets) == 1:
        qw = seq["q_true"][:, ekf.EKF_ORDER]
        out[name].update(max_head_err_filtered_m=float(np.abs(res[0]["x"][200:, :3] - qw[200:, :3]).max()),
                         max_head_err_smoothed_m=float(np.abs(res[0]["smoothed_x"][200:, :3] - qw[200:, :3]).max()),
                         outliers=res[0]["outliers_ignored"],
                         note="the 2 m/s circle; the 10 m/s loop of the FTE benchmark has 40 m/s^2 of centripetal acceleration, "
                              "which the reference's constant-acceleration model with 5 m/s^2 process noise cannot follow "
                              "(oracle and GPU alike)")
sp = synth.make_sequence(150, "sprint")
rs = ekf.ekf(sp["det"], *rig, 120.0, 0.5, (2704, 1520))
qs = sp["q_true"][:, ekf.EKF_ORDER]
out["sprint_150_frames_accuracy"] = dict(max_head_err_filtered_m=float(np.abs(rs["x"][10:, :3] - qs[10:, :3]).max()),
                                         max_head_err_smoothed_m=float(np.abs(rs["smoothed_x"][10:, :3] - qs[10:, :3]).max()),
                                         rms_angle_err_smoothed_rad=float(np.sqrt(np.mean((rs["smoothed_x"][10:, 3:] - qs[10:, 3:]) ** 2))),
                                         outliers=rs["outliers_ignored"])
t0 = time.perf_counter()
oekf.ekf(seq["det"][:40], *rig, 120.0, 0.5, 2704, s0)
dt = time.perf_counter() - t0
out["oracle_numpy_40_frames"] = dict(seconds=dt, frames_per_s=40 / dt)
os.makedirs(os.path.join(ROOT, "gpurun_out", "ekf"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ekf", "report.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
