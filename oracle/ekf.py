"""Oracle extended Kalman filter + Rauch-Tung-Striebel smoother (numpy).  Test infrastructure - see oracle/__init__.py.

Restates ``ekf`` of src/all_optimizations.py:569-865 on arrays (the reference reads videos / DLC .h5 files and
writes pickles around the same arithmetic): constant-acceleration model over the 25 pose parameters (:602-607,
:761-766), measurement = fisheye projection of the 20 cheetah markers in every camera (:613-620), Jacobian by
FORWARD differences with eps = 1e-3 (:631-646, kept: the filter's output depends on it), measurement covariance
(5 px)^4 / (camera width)^2 on the diagonal by likelihood (:805-809 squares the already-squared 5**2), 3-sigma
gating per pixel pair (:815-819), gain through the explicit inverse of S (:822), covariance update (I - K H) P (:829),
RTS smoother over frames N-2 .. 1 (:838-841, frame 0 is left unsmoothed), float32 rounding of the predicted state
(:628).

PINNED (tests/test_ekf.py::test_oracle_matches_reference_ekf_text) to ``tests/golden/ekf_ref.npz``: the reference's
OWN filter + smoother text (:582-593, :603-611, :615-649, :668-679, :684, :699-845) slice-executed on two synthetic
clips by tests/golden/make_golden.py::gen_ekf.  Frame by frame from the reference's own previous state and
covariance the oracle reproduces it to 1e-9; over whole clips to float32-ulp level (the reference rounds every
predicted state to float32, :628, so a 1e-13 difference flips an ulp now and then and the filter carries it on).
ONE ASSUMPTION LEFT: ``lib.misc`` (``get_pose_params``, ``get_markers``, ``get_3d_marker_coords``) is not part of the
reference tree; the parameter order is taken from the ``qb_list`` comments (:734-746) and the marker function is the
cheetah FK of the same file (:66-190, oracle/fk.py) evaluated in float64.
"""
import numpy as np

from . import camera, fk

# pose parameters in the order of qb_list (:734-746), as indices into the 45-state [x y z | phi | theta | psi]
PHI, THETA, PSI = 3, 17, 31
EKF_ORDER = np.array([0, 1, 2,
                      PHI + 0, THETA + 0, PSI + 0,
                      PHI + 1, THETA + 1, PSI + 1,
                      THETA + 2,
                      PHI + 3, THETA + 3, PSI + 3,
                      THETA + 4, PSI + 4,
                      THETA + 5, PSI + 5,
                      THETA + 6, THETA + 7, THETA + 8, THETA + 9,
                      THETA + 10, THETA + 11, THETA + 12, THETA + 13])
QB_LIST = np.array([5.0, 5.0, 5.0, 10.0, 10.0, 10.0, 5.0, 25.0, 5.0, 50.0, 5.0, 50.0, 25.0, 100.0, 30.0, 140.0, 40.0,
                    350.0, 200.0, 350.0, 200.0, 450.0, 400.0, 450.0, 400.0])
N_POSE = 25
N_STATES = 75
IDX_X0, IDX_Y0, IDX_PSI0 = 0, 1, 5          # positions of x_0, y_0, psi_0 in EKF_ORDER


def marker_coords(pose25):
    """get_3d_marker_coords: 25 pose parameters -> (20, 3) marker positions (cheetah FK, :66-190), float64 arithmetic
    whatever the input dtype."""
    q = np.zeros(fk.N_STATES)
    q[EKF_ORDER] = np.asarray(pose25, dtype=np.float64)
    return fk.cheetah_fk(q[None])[0]


def h_function(pose25, k, d, r, t):
    return camera.project_points_fisheye(marker_coords(pose25), k, d, r, t)


def numerical_jacobian(func, x, *args):
    """:631-646 - forward differences, eps = 1e-3, perturbation restored exactly.  In the reference ``x`` is the
    float32 array ``predict_next_state`` returned (:628), so ``xpeturb[i] + eps`` is a FLOAT32 sum (the Python float
    is cast to float32 first) while the difference quotient still divides by the float64 1e-3: every column is off
    by up to ~1e-4 relative, and the filter's output carries that.  Reproduced explicitly for float32 input."""
    n = len(x)
    eps = 1e-3
    fx = func(x, *args).flatten()
    xp = x.copy()
    jac = np.empty((len(fx), n))
    for i in range(n):
        if xp.dtype == np.float32:
            xp[i] = np.float32(xp[i]) + np.float32(eps)
        else:
            xp[i] = xp[i] + eps
        jac[:, i] = (func(xp, *args).flatten() - fx) / eps
        xp[i] = x[i]
    return jac


def model_matrices(sT):
    """P0 (:713-730), Q (:733-757), F (:763-766)."""
    n_ang = N_POSE - 3
    p_ang_acc = np.ones(n_ang) * 3 ** 2
    p_ang_acc[10:] = 5 ** 2
    P0 = np.diag(np.concatenate([np.ones(3) * 3 ** 2, np.ones(n_ang) * (np.pi / 4) ** 2,
                                 np.ones(3) * 5 ** 2, np.ones(n_ang) * 3 ** 2,
                                 np.ones(3) * 3 ** 2, p_ang_acc]))
    qb = (np.diag(QB_LIST) / 2) ** 2
    Q = np.block([[sT ** 4 / 4 * qb, sT ** 3 / 2 * qb, sT ** 2 / 2 * qb],
                  [sT ** 3 / 2 * qb, sT ** 2 * qb, sT * qb],
                  [sT ** 2 / 2 * qb, sT * qb, qb]])
    F = np.eye(N_STATES)
    rng = np.arange(N_STATES - N_POSE)
    rng_acc = np.arange(N_STATES - 2 * N_POSE)
    F[rng, rng + N_POSE] = sT
    F[rng_acc, rng_acc + 2 * N_POSE] = sT ** 2 / 2
    return P0, Q, F


def initial_state(nose_frames, nose_xyz, start_frame, sT):
    """:700-711 - two separate regressions of x and y on the frame number (z_0 stays 0)."""
    f = np.asarray(nose_frames, dtype=np.float64)
    A = np.stack([f, np.ones_like(f)], 1)
    (xs, xi), *_ = np.linalg.lstsq(A, np.asarray(nose_xyz, dtype=np.float64)[:, 0], rcond=None)
    (ys, yi), *_ = np.linalg.lstsq(A, np.asarray(nose_xyz, dtype=np.float64)[:, 1], rcond=None)
    states = np.zeros(N_STATES)
    states[[IDX_X0, IDX_Y0, IDX_PSI0]] = [start_frame * xs + xi, start_frame * ys + yi, np.arctan2(ys, xs)]
    states[[N_POSE + IDX_X0, N_POSE + IDX_Y0]] = [xs / sT, ys / sT]
    return states


def ekf(det, k_arr, d_arr, r_arr, t_arr, fps, dlc_thresh, cam_width, states0, keep_cov=False, P_init=None):
    """det[N, C, 20, 3] (x, y, likelihood) for the frames to filter -> dict of x, dx, ddx, smoothed_x, smoothed_dx,
    smoothed_ddx [N, 25] in EKF_ORDER, plus ``outliers_ignored``.  ``P_init`` replaces P0 (used by the one-step
    checks against the reference's stored per-frame covariances)."""
    det = np.asarray(det, dtype=np.float64)
    n_frames, n_cams, n_markers, _ = det.shape
    sT = 1.0 / fps
    sigma_bound = 3
    max_pixel_err = cam_width
    dlc_cov = 5 ** 2
    P, Q, F = model_matrices(sT)
    if P_init is not None:
        P = np.asarray(P_init, dtype=np.float64).copy()
    states = np.asarray(states0, dtype=np.float64).copy()
    pixels_arr = det[..., :2].reshape(n_frames, -1)
    likelihood_arr = det[..., 2].reshape(n_frames, -1)
    cams = [(k_arr[j], d_arr[j], r_arr[j], t_arr[j]) for j in range(n_cams)]
    m = n_markers * 2
    states_est_hist = np.zeros((n_frames, N_STATES))
    states_pred_hist = states_est_hist.copy()
    P_est_hist = np.zeros((n_frames, N_STATES, N_STATES))
    P_pred_hist = P_est_hist.copy()
    outliers_ignored = 0
    for i in range(n_frames):
        acc = states[2 * N_POSE:]
        vel = states[N_POSE:2 * N_POSE] + sT * acc
        pos = states[:N_POSE] + sT * vel + (0.5 * sT ** 2) * acc
        states32 = np.concatenate([pos, vel, acc]).astype(np.float32)                       # :628
        states = states32.astype(np.float64)
        states_pred_hist[i] = states
        P = F @ P @ F.T + Q
        P_pred_hist[i] = P
        z_k = pixels_arr[i]
        H = np.zeros((n_cams * m, N_STATES))
        h = np.zeros(n_cams * m)
        for j in range(n_cams):
            h[j * m:(j + 1) * m] = h_function(states[:N_POSE], *cams[j]).flatten()
            H[j * m:(j + 1) * m, 0:N_POSE] = numerical_jacobian(h_function, states32[:N_POSE], *cams[j])
        bad = np.repeat(likelihood_arr[i] < dlc_thresh, 2)
        dlc_cov_arr = dlc_cov * np.ones(n_cams * m)
        dlc_cov_arr[bad] = max_pixel_err
        R = np.diag(dlc_cov_arr ** 2)
        residual = z_k - h
        S = (H @ P @ H.T) + R
        temp = sigma_bound * np.sqrt(np.diag(S))
        for j in range(0, len(residual), 2):
            if np.abs(residual[j]) > temp[j] or np.abs(residual[j + 1]) > temp[j + 1]:
                residual[j:j + 2] = 0
                outliers_ignored += 1
        K = P @ H.T @ np.linalg.inv(S)
        states = states + K @ residual
        states_est_hist[i] = states
        P = (np.eye(K.shape[0]) - K @ H) @ P
        P_est_hist[i] = P
    smooth = states_est_hist.copy()
    for i in range(n_frames - 2, 0, -1):
        A = P_est_hist[i] @ F.T @ np.linalg.inv(P_pred_hist[i + 1])
        smooth[i] = states_est_hist[i] + A @ (smooth[i + 1] - states_pred_hist[i + 1])
    out = dict(x=states_est_hist[:, :N_POSE], dx=states_est_hist[:, N_POSE:2 * N_POSE], ddx=states_est_hist[:, 2 * N_POSE:],
               smoothed_x=smooth[:, :N_POSE], smoothed_dx=smooth[:, N_POSE:2 * N_POSE], smoothed_ddx=smooth[:, 2 * N_POSE:],
               outliers_ignored=outliers_ignored)
    if keep_cov:
        out.update(P_est=P_est_hist, P_pred=P_pred_hist, x_pred=states_pred_hist)
    return out
