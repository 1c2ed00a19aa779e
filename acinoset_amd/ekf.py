"""Extended Kalman filter + RTS smoother on the GPU - the array-level drop-in for ``ekf`` of
src/all_optimizations.py:569-865.

The reference function reads the videos' fps / resolution, the scene file and the DLC tables, then runs the filter
frame by frame (finite-difference Jacobian: 26 FK + 156 projection calls per frame, a 240 x 240 inverse) and pickles
``dict(x, dx, ddx, smoothed_x, smoothed_dx, smoothed_ddx)`` (:848-856).  ``ekf`` here takes the same information as
arrays and returns that dictionary (25 columns, pose parameters in the order of ``qb_list`` :734-746 = ``POSE_PARAMS``),
plus the gated-outlier count and the marker positions of both state sequences.  All model constants are the
reference's literals; the arithmetic runs in csrc/ekf.hip (one workgroup per sequence, covariance resident in LDS).
Several clips are filtered by one launch with ``ekf_batch`` (the filter is sequential in frames, parallel in clips).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, calib, fte
from ._lib import EkfParams, check, lib, ptr, stream_ptr

N_POSE, N_EKF_STATES = 25, 75
POSE_PARAMS = ["x_0", "y_0", "z_0", "phi_0", "theta_0", "psi_0", "phi_1", "theta_1", "psi_1", "theta_2",
               "phi_3", "theta_3", "psi_3", "theta_4", "psi_4", "theta_5", "psi_5", "theta_6", "theta_7",
               "theta_8", "theta_9", "theta_10", "theta_11", "theta_12", "theta_13"]
_BASE = dict(phi=fte.PHI, theta=fte.THETA, psi=fte.PSI)
# index of every pose parameter in the 45-state vector [x y z | phi_0..13 | theta_0..13 | psi_0..13]
EKF_ORDER = np.array([0, 1, 2] + [_BASE[n.split("_")[0]] + int(n.split("_")[1]) for n in POSE_PARAMS[3:]])


def get_pose_params():
    """``misc.get_pose_params()`` (:581): state name -> index."""
    return {name: i for i, name in enumerate(POSE_PARAMS)}


def get_3d_marker_coords(x):
    """``misc.get_3d_marker_coords`` (:617): pose parameters [..., 25] -> marker positions [..., 20, 3]."""
    x = np.asarray(x, dtype=np.float64)
    q = np.zeros(x.shape[:-1] + (fte.N_STATES,))
    q[..., EKF_ORDER] = x
    return fte.cheetah_fk(q.reshape(-1, fte.N_STATES)).reshape(x.shape[:-1] + (20, 3))


def initial_state(det, k_arr, d_arr, r_arr, t_arr, fps, dlc_thresh, start_frame=0):
    """:700-711 - nose position and heading from two regressions of the triangulated nose on the frame number."""
    tri = calib.triangulate_pairs_dense(det, dlc_thresh, k_arr, d_arr, r_arr, t_arr, return_masks=False)
    nose = tri[:, 2] if isinstance(tri, np.ndarray) else tri[:, 2].cpu().numpy()
    ok = np.isfinite(nose).all(1)
    if ok.sum() < 2:
        raise ValueError("fewer than two triangulated nose points: cannot initialise the filter")
    f = np.arange(nose.shape[0], dtype=np.float64)[ok] + start_frame
    A = np.stack([f, np.ones_like(f)], 1)
    (xs, xi), (ys, yi) = (np.linalg.lstsq(A, nose[ok, j], rcond=None)[0] for j in (0, 1))
    sT = 1.0 / fps
    s = np.zeros(N_EKF_STATES)
    s[[0, 1, 5]] = [start_frame * xs + xi, start_frame * ys + yi, np.arctan2(ys, xs)]
    s[[N_POSE + 0, N_POSE + 1]] = [xs / sT, ys / sT]
    return s


def ekf_batch(dets, k_arr, d_arr, r_arr, t_arr, fps, dlc_thresh, camera_resolution, start_frames=None, states0=None,
              with_positions=True, smoother_pivoting=False):
    """Filter + smooth several clips of the same rig.  ``dets``: list of det[N_b, C, 20, 3] (x, y, likelihood);
    clips of equal length share one launch.  Returns one result dictionary per clip."""
    _lib.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())
    B = len(dets)
    start_frames = list(start_frames) if start_frames is not None else [0] * B
    dets_d = [calib._to_dev(d, dev) for d in dets]
    for d in dets_d:
        if d.dim() != 4 or d.shape[2] != 20 or d.shape[3] != 3:
            raise ValueError("det must be [N, C, 20, 3] = (x, y, likelihood)")
    cams = torch.as_tensor(calib.fisheye_records(k_arr, d_arr, r_arr, t_arr), device=dev)
    n_cams = int(cams.shape[0])
    if any(int(d.shape[1]) != n_cams for d in dets_d):
        raise ValueError("camera count mismatch between det and the rig")
    if n_cams > 6:
        raise NotImplementedError("the EKF kernel holds the measurement Jacobian of at most 6 cameras in LDS")
    s0 = []
    for b in range(B):
        if states0 is not None and states0[b] is not None:
            s = np.asarray(states0[b], dtype=np.float64).reshape(-1)
            if s.size != N_EKF_STATES:
                raise ValueError("states0 must have 75 entries (pose, velocity, acceleration)")
        else:
            s = initial_state(dets_d[b], k_arr, d_arr, r_arr, t_arr, fps, dlc_thresh, start_frames[b])
        s0.append(s)
    out = [None] * B
    groups = {}
    for b, d in enumerate(dets_d):
        groups.setdefault(int(d.shape[0]), []).append(b)
    for n_frames, members in groups.items():
        if n_frames < 1:
            raise ValueError("empty sequence")
        det = torch.stack([dets_d[b] for b in members]).contiguous()
        st0 = torch.as_tensor(np.stack([s0[b] for b in members]), device=dev)
        prm = EkfParams(n_frames=n_frames, n_seq=len(members), n_cams=n_cams, fps=float(fps), dlc_thresh=float(dlc_thresh),
                        cam_width=float(camera_resolution[0]), smoother_pivoting=int(bool(smoother_pivoting)))
        nbytes = lib().acino_ekf_workspace_bytes(n_frames, len(members))
        ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
        ws_ptr = (ws.data_ptr() + 255) // 256 * 256
        est = torch.empty((len(members), n_frames, N_EKF_STATES), dtype=torch.float64, device=dev)
        smo = torch.empty_like(est)
        outl = torch.zeros(len(members), dtype=torch.int32, device=dev)
        check(lib().acino_ekf_run(C.byref(prm), ptr(det), ptr(cams), ptr(st0), C.c_void_p(ws_ptr), nbytes, ptr(est),
                                  ptr(smo), C.c_void_p(outl.data_ptr()), stream_ptr()))
        est_h, smo_h, outl_h = est.cpu().numpy(), smo.cpu().numpy(), outl.cpu().numpy()
        for j, b in enumerate(members):
            r = dict(x=est_h[j, :, :N_POSE], dx=est_h[j, :, N_POSE:2 * N_POSE], ddx=est_h[j, :, 2 * N_POSE:],
                     smoothed_x=smo_h[j, :, :N_POSE], smoothed_dx=smo_h[j, :, N_POSE:2 * N_POSE],
                     smoothed_ddx=smo_h[j, :, 2 * N_POSE:], outliers_ignored=int(outl_h[j]), start_frame=start_frames[b])
            if with_positions:
                r["positions"] = get_3d_marker_coords(r["x"])
                r["smoothed_positions"] = get_3d_marker_coords(r["smoothed_x"])
            out[b] = r
    return out


def ekf(det, k_arr, d_arr, r_arr, t_arr, fps, dlc_thresh, camera_resolution, start_frame=0, states0=None,
        with_positions=True, smoother_pivoting=False):
    """One clip: det[N, C, 20, 3] for the frames start_frame .. start_frame + N - 1."""
    return ekf_batch([det], k_arr, d_arr, r_arr, t_arr, fps, dlc_thresh, camera_resolution, [start_frame],
                     None if states0 is None else [states0], with_positions, smoother_pivoting)[0]
