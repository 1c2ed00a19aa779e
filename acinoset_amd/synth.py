"""Synthetic 6-camera x 20-marker sequences generated THROUGH the HIP path
(cheetah FK and fisheye projection kernels).  Used by bench.py and the GPU tests; the CPU twin for the
oracle-side tests lives in oracle/synth.py.  RNG: numpy default_rng(20210313) as the survey prescribes."""
import numpy as np

from . import calib, fte

FPS = 120.0
IMG_W, IMG_H = 2704, 1520
K_DUMMY = np.array([[1239.734301643185, 0.0, 1345.6656873020534],
                    [0.0, 1238.6767415364864, 772.857231264526],
                    [0.0, 0.0, 1.0]])
D_DUMMY = np.array([0.03743898455870094, 0.04784376818211986, -0.03758546926131943, 0.010781487471488252])
CAM_CENTRES = np.array([[0.0, 0.0, 0.0], [4.10, 2.63, 0.25], [8.61, 4.90, 0.52],
                        [3.90, 12.99, 0.52], [-0.59, 10.52, 0.36], [-4.57, 7.30, 0.12]])
LOOK_AT = np.array([2.0, 6.5, 0.7])


def make_rig(n_cams=6):
    """Ring rig with the geometry of the reference's configs/dummy_scene.json (identical intrinsics)."""
    K = np.tile(K_DUMMY, (n_cams, 1, 1))
    D = np.tile(D_DUMMY, (n_cams, 1))
    R = np.zeros((n_cams, 3, 3))
    t = np.zeros((n_cams, 3, 1))
    for i in range(n_cams):
        c = CAM_CENTRES[i % len(CAM_CENTRES)]
        zc = LOOK_AT - c
        zc = zc / np.linalg.norm(zc)
        xc = np.cross(zc, np.array([0.0, 0.0, 1.0]))
        xc = xc / np.linalg.norm(xc)
        yc = np.cross(zc, xc)
        R[i] = np.stack([xc, yc, zc])
        t[i, :, 0] = -R[i] @ c
    return K, D, R, t


def trajectory(n_frames, kind="loop", seed_phase=0.0):
    """Ground-truth 45-state trajectory inside all 21 angle boxes."""
    PHI, THETA, PSI = fte.PHI, fte.THETA, fte.PSI
    tt = np.arange(n_frames) / FPS
    q = np.zeros((n_frames, 45))
    if kind in ("loop", "walk"):
        # "walk": the same circle at 2 m/s - slow enough for the EKF's constant-acceleration model (5 m/s^2 process
        # noise on x, y) to follow the centripetal acceleration, which it cannot at 10 m/s
        radius, speed = 2.5, (10.0 if kind == "loop" else 2.0)
        ang = speed / radius * tt + seed_phase
        q[:, 0] = LOOK_AT[0] + radius * np.cos(ang)
        q[:, 1] = LOOK_AT[1] + radius * np.sin(ang)
        q[:, 2] = LOOK_AT[2]
        q[:, PSI + 0] = ang + np.pi / 2
    elif kind in ("sprint", "trot"):
        # "trot": the same straight run at 1.2 m/s - 1 000 frames span 10 m, i.e. stay inside the rig's field of view
        # (a 10 m/s sprint of 1 000 frames is 83 m long: most of it is seen by no camera and constrained by the
        # smoothness prior alone).  The straight nose-line initialisation of the reference fits both.
        speed = 10.0 if kind == "sprint" else 1.2
        span = speed * (n_frames - 1) / FPS
        heading = 0.35 + seed_phase
        dirv = np.array([np.cos(heading), np.sin(heading)])
        start = LOOK_AT[:2] - 0.5 * span * dirv
        q[:, 0] = start[0] + speed * tt * dirv[0]
        q[:, 1] = start[1] + speed * tt * dirv[1]
        q[:, 2] = LOOK_AT[2] + 0.03 * np.sin(2 * np.pi * 3.0 * tt)
        q[:, PSI + 0] = heading + 0.05 * np.sin(2 * np.pi * 1.0 * tt)
    else:
        raise ValueError(kind)
    offs = np.array([0, -np.pi / 2, 0, -np.pi / 2, 0, np.pi / 2, 0, np.pi / 2])
    for k in range(8):
        q[:, THETA + 6 + k] = offs[k] + 0.5 * np.sin(2 * np.pi * 3.0 * tt + 0.7 * k)
    small = [PHI + 0, PHI + 1, PHI + 3, THETA + 0, THETA + 1, THETA + 2, THETA + 3, THETA + 4, THETA + 5,
             PSI + 1, PSI + 3, PSI + 4, PSI + 5]
    for j, idx in enumerate(small):
        q[:, idx] = 0.1 * np.sin(2 * np.pi * 1.5 * tt + 0.5 * j)
    return q


def detections_from_positions(pos, K, D, R, t, seed=20210313, noise_px=2.0, outlier_frac=0.15):
    rng = np.random.default_rng(seed)
    N, L, _ = pos.shape
    Cn = K.shape[0]
    det = np.zeros((N, Cn, L, 3))
    flat = pos.reshape(-1, 3)
    for c in range(Cn):
        uv = calib.project_points_fisheye(flat, K[c], D[c], R[c], t[c]).reshape(N, L, 2)
        zc = pos @ R[c][2] + t[c].reshape(3)[2]
        uv = uv + rng.normal(0.0, noise_px, uv.shape)
        lik = rng.uniform(0.55, 1.0, (N, L))
        out = rng.uniform(size=(N, L)) < outlier_frac
        lik = np.where(out, rng.uniform(0.0, 0.4, (N, L)), lik)
        uv = np.where(out[..., None], uv + rng.uniform(-100, 100, uv.shape), uv)
        bad = (zc < 0.5) | (uv[..., 0] < 0) | (uv[..., 0] >= IMG_W) | (uv[..., 1] < 0) | (uv[..., 1] >= IMG_H) \
            | ~np.isfinite(uv).all(-1)
        lik = np.where(bad, 0.05, lik)
        uv = np.where(np.isfinite(uv), uv, 0.0)
        det[:, c, :, :2] = uv
        det[:, c, :, 2] = lik
    return det


def make_sequence(n_frames, kind="loop", seed=20210313, rig=None):
    K, D, R, t = make_rig() if rig is None else rig
    q = trajectory(n_frames, kind)
    pos = fte.cheetah_fk(q)
    det = detections_from_positions(pos, K, D, R, t, seed=seed)
    return dict(K=K, D=D, R=R, t=t, q_true=q, pos_true=pos, det=det, Ts=1.0 / FPS)
