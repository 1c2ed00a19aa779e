"""GPU (-m gpu): the HIP bundle adjustment (csrc/sba.hip through the C ABI) against the scipy oracle and KAT-2.

Parity statement for an optimiser: (1) the residual / cost FUNCTION is the reference's (compared point-wise with
the oracle, 1e-8 px); (2) where the problem is well-posed (points only) the minimiser agrees with scipy's
to 1e-6 m; (3) on the gauge-free points+extrinsics problem, where the reference's TRF run stops on xtol far from
stationarity (recorded first-order optimality 3.6e+03 / 7.1e+03), the GPU solve must end at or below the recorded
KAT-2 cost, with the returned parameters reproducing that cost through the ORACLE's residual function."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import camera as ocam
from oracle import sba as osba
from test_oracle_sba import kat2_problem


@pytest.fixture(scope="module")
def gsba(gpu_lib):
    from acinoset_amd import calib, sba
    return sba, calib


def _kat(golden_dir, tag, ca, cb):
    g = np.load(os.path.join(golden_dir, "kat1_sunday_amelia.npz"))
    img, names, shape, K, D, R, t = kat2_problem(g, tag, ca, cb)
    data = osba.prepare_calib_board_data(img, names, shape, K, D, R, t, ocam.triangulate_points_fisheye)
    return g, (img, names, shape), data, (K, D, R, t)


def test_prepare_board_data_matches_oracle(gsba, golden_dir):
    sba, calib = gsba
    g, (img, names, shape), data, (K, D, R, t) = _kat(golden_dir, "static", 3, 4)
    names = [names[0], names[1][:-2] + ["only_b_1", "only_b_2"]]
    want = osba.prepare_calib_board_data(img, names, shape, K, D, R, t, ocam.triangulate_points_fisheye)
    got = sba.prepare_calib_board_data_for_bundle_adjustment(img, names, shape, K, D, R, t,
                                                             calib.triangulate_points_fisheye)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3])
    assert got[1].dtype == np.float32 and np.abs(got[1].astype(np.float64) - want[1]).max() < 1e-6   # float32 storage
    pts = np.full((5, 3, 2), np.nan)
    X = np.array([[0.1, 0.2, 3.0], [0.4, -0.2, 2.5], [-0.3, 0.1, 2.8], [0.0, 0.0, 3.3], [0.2, 0.2, 2.2]])
    K3, D3 = np.stack([K[0], K[1], K[0]]), np.stack([D[0], D[1], D[0]])
    R3 = np.stack([np.eye(3), ocam.rodrigues(np.array([0.0, 0.1, 0.0])), np.eye(3)])
    t3 = np.array([[[0.0], [0], [0]], [[-0.3], [0], [0.05]], [[0.2], [0], [0]]])
    for c in range(3):
        pts[:, c] = ocam.project_points_fisheye(X, K3[c], D3[c], R3[c], t3[c])
    pts[1, 0] = np.nan
    pts[3, 1:] = np.nan                                   # seen by one camera only: dropped
    p2, p3, pi, ci = sba.prepare_manual_points_for_bundle_adjustment(pts, K3, D3, R3, t3)
    assert p3.shape == (4, 1, 3) and list(np.bincount(pi)) == [3, 2, 3, 3] and list(ci[:5]) == [0, 1, 2, 1, 2]
    assert np.abs(p3[:, 0] - X[[0, 1, 2, 4]]).max() < 1e-5


def test_cost_function_is_the_reference_one(gsba, golden_dir):
    sba, _ = gsba
    for tag, ca, cb, row in (("rotating", 1, 2, 0), ("static", 3, 4, 1)):
        g, _, data, (K, D, R, t) = _kat(golden_dir, tag, ca, cb)
        pts, rm, tt, res = sba.bundle_adjust_points_and_extrinsics(*data, K, D, R, t, max_iter=0)
        Rq = np.array([ocam.rodrigues(osba.rodrigues_to_vec(r)) for r in R])     # x0 of calib.py:373-375
        f0 = osba.residuals(data[1].astype(np.float64), Rq, t, K, D, data[2], data[3], data[0])
        assert np.abs(res["before"] - f0).max() < 1e-8 and np.abs(res["after"] - f0).max() < 1e-8      # px
        assert abs(sba.last_info["cost_initial"] - osba.cauchy_cost(f0)) < 1e-9
        assert abs(sba.last_info["cost_initial"] - g["recorded"][row][2]) / g["recorded"][row][2] < 5e-5   # KAT-1 cost
        assert np.array_equal(pts, data[1].astype(np.float64)) and np.abs(rm - Rq).max() < 1e-14
        _p, res50 = sba.bundle_adjust_points_only(*data, K, D, R, t, max_iter=0)
        f50 = osba.residuals(data[1].astype(np.float64), R, t, K, D, data[2], data[3], data[0])   # rotations as given
        assert np.abs(res50["before"] - f50).max() < 1e-8
        assert abs(sba.last_info["cost_initial"] - osba.cauchy_cost(f50, 50)) < 1e-8


def test_kat2_points_and_extrinsics(gsba, golden_dir):
    sba, calib = gsba
    for tag, ca, cb, row in (("rotating", 1, 2, 0), ("static", 3, 4, 1)):
        g, _, data, (K, D, R, t) = _kat(golden_dir, tag, ca, cb)
        pts, rm, tt, res = sba.bundle_adjust_points_and_extrinsics(*data, K, D, R, t, calib.project_points_fisheye)
        info = dict(sba.last_info)
        cost_rec = g["recorded_sba"][row][0]
        assert info["status_name"] in ("ftol", "gtol"), info
        assert info["cost_final"] <= cost_rec * (1 + 5e-5), (tag, info)       # at or below the reference's end state
        assert info["cost_final"] < info["cost_initial"]
        # the returned parameters carry that cost through the oracle's residual function
        fa = osba.residuals(pts, rm, tt, K, D, data[2], data[3], data[0])
        assert np.abs(fa - res["after"]).max() < 1e-8 and abs(osba.cauchy_cost(fa) - info["cost_final"]) < 1e-8
        assert np.abs(rm @ rm.transpose(0, 2, 1) - np.eye(3)).max() < 1e-12 and tt.shape == (2, 3, 1)
        # ... and it is a stationary point of the robust cost, which the recorded runs were not
        assert info["gnorm_inf"] < 1e-2 * (3.63e3 if row == 0 else 7.05e3), info


def test_kat2_distance_from_the_reference_end_state(gsba, golden_dir, capsys):
    """How far the returned extrinsics are from the reference's (= the scipy oracle's) end state, in gauge-invariant
    terms (rotation of camera 2 relative to camera 1 in degrees, baseline direction in degrees, baseline length in
    mm) - and that the difference is "the reference stopped early", not "a different problem": the same scipy call
    with a Jacobian mask that matches the parameter layout (oracle.sba.sparsity_by_layout; the reference's mask
    calib.py:196-207 does not, see tests/test_oracle_sba.py) converges to the GPU's stationary point."""
    sba, calib = gsba
    g, _, data, (K, D, R, t) = _kat(golden_dir, "static", 3, 4)
    _pts, rm, tt, _res = sba.bundle_adjust_points_and_extrinsics(*data, K, D, R, t, calib.project_points_fisheye)
    cost_gpu = sba.last_info["cost_final"]
    _p, rm_ref, tt_ref, _r, res_ref = osba.bundle_adjust_points_and_extrinsics(*data, K, D, R, t)
    _p, rm_ok, tt_ok, _r, res_ok = osba.bundle_adjust_points_and_extrinsics(*data, K, D, R, t, consistent_mask=True)
    d_ref = osba.pose_distance(rm, tt, rm_ref, tt_ref)
    d_ok = osba.pose_distance(rm, tt, rm_ok, tt_ok)
    d_move = osba.pose_distance(R, t, rm_ref, tt_ref)
    with capsys.disabled():
        print(f"\nKAT-2 static: GPU cost {cost_gpu:.4f}; scipy/reference mask {res_ref.cost:.4f}; scipy/consistent mask "
              f"{res_ok.cost:.4f}\n  GPU vs reference end state: {d_ref[0]:.4f} deg rel. rotation, {d_ref[1]:.4f} deg "
              f"baseline direction, baseline {d_ref[2]:.2f} vs {d_ref[3]:.2f} mm\n  GPU vs consistent-mask scipy: "
              f"{d_ok[0]:.2e} deg, {d_ok[1]:.2e} deg, {d_ok[2]:.3f} vs {d_ok[3]:.3f} mm\n  reference end state vs its own "
              f"start: {d_move[0]:.4f} deg, {d_move[1]:.4f} deg, {d_move[2]:.2f} -> {d_move[3]:.2f} mm")
    assert abs(cost_gpu - res_ok.cost) < 1e-4 * res_ok.cost                       # same minimum of the same cost
    assert d_ok[0] < 5e-3 and d_ok[1] < 5e-3 and abs(d_ok[2] - d_ok[3]) < 0.2      # deg, deg, mm
    assert d_ref[0] < 1.0 and d_ref[1] < 1.0 and abs(d_ref[2] - d_ref[3]) < 20.0   # bounded distance from the reference's
    assert d_move[0] < 0.05                                                        # ... which barely left its start
    # rotating pair: the consistent-mask run is still creeping after 300 evaluations; the GPU end state lies below it
    g, _, data, (K, D, R, t) = _kat(golden_dir, "rotating", 1, 2)
    _pts, rm, tt, _res = sba.bundle_adjust_points_and_extrinsics(*data, K, D, R, t, calib.project_points_fisheye)
    cost_gpu = sba.last_info["cost_final"]
    _p, rm_ref, tt_ref, _r, res_ref = osba.bundle_adjust_points_and_extrinsics(*data, K, D, R, t)
    _p, rm_ok, tt_ok, _r, res_ok = osba.bundle_adjust_points_and_extrinsics(*data, K, D, R, t, consistent_mask=True,
                                                                            max_nfev=300)
    d_ref = osba.pose_distance(rm, tt, rm_ref, tt_ref)
    d_ok = osba.pose_distance(rm, tt, rm_ok, tt_ok)
    with capsys.disabled():
        print(f"KAT-2 rotating: GPU cost {cost_gpu:.4f}; scipy/reference mask {res_ref.cost:.4f}; scipy/consistent mask "
              f"(300 nfev) {res_ok.cost:.4f}\n  GPU vs reference end state: {d_ref[0]:.4f} deg, {d_ref[1]:.4f} deg, "
              f"baseline {d_ref[2]:.2f} vs {d_ref[3]:.2f} mm\n  GPU vs consistent-mask scipy: {d_ok[0]:.4f} deg, "
              f"{d_ok[1]:.4f} deg, {d_ok[2]:.2f} vs {d_ok[3]:.2f} mm")
    assert cost_gpu <= res_ok.cost * (1 + 1e-6) < res_ref.cost
    assert d_ref[0] < 2.0 and d_ref[1] < 5.0 and d_ok[0] <= d_ref[0] + 1e-3


def test_points_only_matches_scipy_minimiser(gsba, golden_dir):
    sba, calib = gsba
    g, _, data, (K, D, R, t) = _kat(golden_dir, "static", 3, 4)
    want, wres, wopt = osba.bundle_adjust_points_only(*data, K, D, R, t)
    got, gres = sba.bundle_adjust_points_only(*data, K, D, R, t, calib.project_points_fisheye)
    assert abs(sba.last_info["cost_final"] - wopt.cost) / wopt.cost < 1e-9
    assert np.abs(got - want).max() < 1e-6                                        # metres
    assert np.abs(gres["after"] - wres["after"]).max() < 1e-4                     # px
    with pytest.raises(NotImplementedError):
        sba.bundle_adjust_points_only(*data, K, D, R, t, lambda *a: None)          # neither of the reference's two models


def test_six_camera_rig_recovers_from_perturbation(gsba):
    """Synthetic 6-camera rig, 400 points seen by 2..6 cameras, 0.3 px noise, extrinsics perturbed by ~1 degree /
    2 cm: SBA must bring the cost down to the noise floor and agree with scipy on the final cost."""
    sba, calib = gsba
    from acinoset_amd import synth
    rng = np.random.default_rng(5)
    K, D, R, t = synth.make_rig()
    X = np.array([2.0, 6.5, 0.7]) + rng.normal(0, 1.0, (400, 3))
    p2, pi, ci = [], [], []
    for p in range(400):
        cams = np.sort(rng.choice(6, size=rng.integers(2, 7), replace=False))
        for c in cams:
            p2.append(ocam.project_points_fisheye(X[p:p + 1], K[c], D[c], R[c], t[c])[0] + rng.normal(0, 0.3, 2))
            pi.append(p)
            ci.append(c)
    p2, pi, ci = np.array(p2), np.array(pi), np.array(ci)
    Rp = np.array([ocam.rodrigues(rng.normal(0, 0.015, 3)) @ R[c] for c in range(6)])
    tp = t.reshape(6, 3, 1) + rng.normal(0, 0.02, (6, 3, 1))
    X0 = X + rng.normal(0, 0.05, X.shape)
    pts, rm, tt, res = sba.bundle_adjust_points_and_extrinsics(p2, X0, pi, ci, K, D, Rp, tp)
    info = dict(sba.last_info)
    assert info["status_name"] in ("ftol", "gtol") and info["cost_final"] < 0.02 * info["cost_initial"], info
    assert np.sqrt(np.mean(res["after"] ** 2)) < 0.35                              # px: the injected noise level
    _p, _r, _t, ores, oopt = osba.bundle_adjust_points_and_extrinsics(p2, X0, pi, ci, K, D, Rp, tp, max_nfev=200)
    assert info["cost_final"] <= oopt.cost * (1 + 1e-6), (info, oopt.cost)


@pytest.mark.parametrize("model", ["fisheye", "pinhole"])
def test_first_lm_step_equals_a_dense_numpy_step(gsba, model):
    """ONE Levenberg-Marquardt iteration of the HIP solver (analytic Jacobians of points and of the left-perturbed poses,
    Cauchy IRLS weights, (1 + lam) diagonal damping, Schur complement onto the cameras, back-substitution, manifold update)
    against the same step formed densely in numpy from CENTRAL DIFFERENCES of the oracle's residual function: same trial
    cost and same updated parameters.  A wrong Jacobian entry, weight or Schur term still converges - slowly - so the
    converged-state tests cannot see it; this one does."""
    sba, calib = gsba
    from acinoset_amd import synth
    rng = np.random.default_rng(17)
    K6, D6, R6, t6 = synth.make_rig()
    C, P, fs, lam = 3, 30, 1.0, 1e-3
    K, R, t = K6[:C], R6[:C], t6[:C].reshape(C, 3, 1)
    if model == "fisheye":
        D, proj, ofun = D6[:C], calib.project_points_fisheye, ocam.project_points_fisheye
    else:
        D = np.tile(np.array([0.05, -0.02, 1e-3, -5e-4, 0.01, 0.02, -0.01, 0.005]), (C, 1))
        proj, ofun = calib.project_points, ocam.project_points
    X = np.array([2.0, 6.5, 0.7]) + rng.normal(0, 0.6, (P, 3))
    pi, ci = np.repeat(np.arange(P), C), np.tile(np.arange(C), P)
    uv = np.concatenate([np.stack([ofun(X[p:p + 1], K[c], D[c], R[c], t[c])[0] for c in range(C)]) for p in range(P)])
    uv += rng.normal(0, 1.0, uv.shape)
    uv[rng.choice(len(uv), 6, replace=False)] += rng.uniform(-15, 15, (6, 2))      # outliers: the weights matter
    X0 = X + rng.normal(0, 0.03, X.shape)
    R0 = np.array([ocam.rodrigues(rng.normal(0, 0.01, 3)) @ R[c] for c in range(C)])
    t0 = t + rng.normal(0, 0.01, t.shape)

    def resid(Xp, Rm, tv):
        return osba.residuals(Xp, Rm, tv, K, D, pi, ci, uv, project_func=ofun)

    def cost(r):
        return 0.5 * fs * fs * np.log1p((r / fs) ** 2).sum()

    def apply(dc, dp):
        Rn = np.array([ocam.rodrigues(dc[6 * c:6 * c + 3]) @ R0[c] for c in range(C)])
        tn = t0 + dc.reshape(C, 6)[:, 3:].reshape(C, 3, 1)
        return X0 + dp.reshape(P, 3), Rn, tn

    r0 = resid(X0, R0, t0)
    n = 6 * C + 3 * P
    J, h = np.zeros((r0.size, n)), 1e-6
    for k in range(n):
        e = np.zeros(n)
        e[k] = h
        J[:, k] = (resid(*apply(e[:6 * C], e[6 * C:])) - resid(*apply(-e[:6 * C], -e[6 * C:]))) / (2 * h)
    w = 1.0 / (1.0 + (r0 / fs) ** 2)
    A = J.T @ (w[:, None] * J)
    g = J.T @ (w * r0)
    delta = -np.linalg.solve(A + lam * np.diag(np.diag(A)), g)
    Xn, Rn, tn = apply(delta[:6 * C], delta[6 * C:])
    c0, c1 = cost(r0), cost(resid(Xn, Rn, tn))
    assert c1 < c0
    pts, rm, tt, _res = sba.bundle_adjust_points_and_extrinsics(uv, X0, pi, ci, K, D, R0, t0, proj, max_iter=1)
    info = dict(sba.last_info)
    assert info["iterations"] == 1 and info["accepted"] == 1
    assert abs(info["cost_initial"] - c0) < 1e-9 * c0 and abs(info["cost_final"] - c1) < 1e-6 * c1, (info, c0, c1)
    assert np.abs(pts - Xn).max() < 1e-6 and np.abs(rm - Rn).max() < 1e-6 and np.abs(tt - tn).max() < 1e-6
    # points only (calib.py:327-341, Cauchy scale 50 px): the 3 x 3 point blocks alone
    fs = 50.0
    w = 1.0 / (1.0 + (r0 / fs) ** 2)
    Jp = J[:, 6 * C:]
    Ap = Jp.T @ (w[:, None] * Jp)
    dp = -np.linalg.solve(Ap + lam * np.diag(np.diag(Ap)), Jp.T @ (w * r0))
    c0, c1 = cost(r0), cost(resid(X0 + dp.reshape(P, 3), R0, t0))
    pts, _r = sba.bundle_adjust_points_only(uv, X0, pi, ci, K, D, R0, t0, proj, f_scale=50, max_iter=1)
    info = dict(sba.last_info)
    assert info["iterations"] == 1 and info["accepted"] == 1 and c1 < c0
    assert abs(info["cost_initial"] - c0) < 1e-9 * c0 and abs(info["cost_final"] - c1) < 1e-6 * c1, (info, c0, c1)
    assert np.abs(pts - (X0 + dp.reshape(P, 3))).max() < 1e-6


@pytest.mark.parametrize("n_cams", [2, 4, 5, 6, 7])
def test_first_lm_step_with_ragged_visibility_for_every_camera_count(gsba, n_cams):
    """The fused kernels deal 64 / C points to a wave and feed the matrix cores in chunks of five points: every camera count
    of the fused path (two tile rows up to five cameras, three from six on; a short last chunk for 4, 5 and 7 cameras), a
    point count that fills neither the last batch nor the last wave, and points that some cameras do not see - one LM step
    against the dense numpy step built from central differences, as above."""
    sba, calib = gsba
    from acinoset_amd import synth
    rng = np.random.default_rng(100 + n_cams)
    K6, D6, R6, t6 = synth.make_rig()
    C, P, fs, lam = n_cams, 37, 1.0, 1e-3
    sel = np.arange(C) % 6                                  # (a seventh camera: a copy of the first one, moved)
    K, D, R, t = K6[sel], D6[sel], R6[sel].copy(), t6[sel].reshape(C, 3, 1).copy()
    if C == 7:
        R[6] = ocam.rodrigues(np.array([0.02, -0.05, 0.03])) @ R[6]
        t[6] = t[6] + np.array([[0.3], [-0.2], [0.1]])
    ofun, proj = ocam.project_points_fisheye, calib.project_points_fisheye
    X = np.array([2.0, 6.5, 0.7]) + rng.normal(0, 0.6, (P, 3))
    pi, ci = [], []
    for p in range(P):
        cams = np.sort(rng.choice(C, size=rng.integers(2, C + 1), replace=False))
        pi += [p] * len(cams)
        ci += list(cams)
    pi, ci = np.array(pi), np.array(ci)
    uv = np.stack([ofun(X[p:p + 1], K[c], D[c], R[c], t[c])[0] for p, c in zip(pi, ci)]) + rng.normal(0, 1.0, (len(pi), 2))
    uv[rng.choice(len(uv), 5, replace=False)] += rng.uniform(-15, 15, (5, 2))
    X0 = X + rng.normal(0, 0.03, X.shape)
    R0 = np.array([ocam.rodrigues(rng.normal(0, 0.01, 3)) @ R[c] for c in range(C)])
    t0 = t + rng.normal(0, 0.01, t.shape)

    def resid(Xp, Rm, tv):
        return osba.residuals(Xp, Rm, tv, K, D, pi, ci, uv, project_func=ofun)

    def apply(dc, dp):
        Rn = np.array([ocam.rodrigues(dc[6 * c:6 * c + 3]) @ R0[c] for c in range(C)])
        return X0 + dp.reshape(P, 3), Rn, t0 + dc.reshape(C, 6)[:, 3:].reshape(C, 3, 1)

    r0 = resid(X0, R0, t0)
    n = 6 * C + 3 * P
    J, h = np.zeros((r0.size, n)), 1e-6
    for k in range(n):
        e = np.zeros(n)
        e[k] = h
        J[:, k] = (resid(*apply(e[:6 * C], e[6 * C:])) - resid(*apply(-e[:6 * C], -e[6 * C:]))) / (2 * h)
    w = 1.0 / (1.0 + (r0 / fs) ** 2)
    A = J.T @ (w[:, None] * J)
    delta = -np.linalg.solve(A + lam * np.diag(np.diag(A)), J.T @ (w * r0))
    Xn, Rn, tn = apply(delta[:6 * C], delta[6 * C:])
    c0 = 0.5 * np.log1p(r0 ** 2).sum()
    c1 = 0.5 * np.log1p(resid(Xn, Rn, tn) ** 2).sum()
    pts, rm, tt, _res = sba.bundle_adjust_points_and_extrinsics(uv, X0, pi, ci, K, D, R0, t0, proj, max_iter=1)
    info = dict(sba.last_info)
    assert info["iterations"] == 1 and info["accepted"] == (1 if c1 < c0 else 0)
    assert abs(info["cost_initial"] - c0) < 1e-9 * c0
    if c1 < c0:
        assert abs(info["cost_final"] - c1) < 1e-6 * c1, (info, c0, c1)
        assert np.abs(pts - Xn).max() < 1e-6 and np.abs(rm - Rn).max() < 1e-6 and np.abs(tt - tn).max() < 1e-6


def test_pinhole_model_bundle_adjustment(gsba):
    """The reference's second SBA call site (sba_board_points, app.py:215-218) injects the cv2.projectPoints pinhole
    model (rational + tangential distortion, calibrated with CALIB_RATIONAL_MODEL, calib.py:18)."""
    sba, calib = gsba
    from acinoset_amd import synth
    rng = np.random.default_rng(9)
    K, _D, R, t = synth.make_rig()
    D = np.tile(np.array([0.08, -0.05, 0.001, -0.002, 0.01, 0.02, -0.01, 0.003]), (6, 1))
    X = np.array([2.0, 6.5, 0.7]) + rng.normal(0, 0.6, (300, 3))
    p2, pi, ci = [], [], []
    for p in range(300):
        for c in np.sort(rng.choice(6, size=rng.integers(2, 5), replace=False)):
            Y = R[c] @ X[p] + t[c].reshape(3)
            if Y[2] < 1.0 or (Y[0] / Y[2]) ** 2 + (Y[1] / Y[2]) ** 2 > 0.5:
                continue                                   # keep to the field of view where the rational model is sane
            p2.append(ocam.project_points(X[p:p + 1], K[c], D[c], R[c], t[c])[0] + rng.normal(0, 0.3, 2))
            pi.append(p)
            ci.append(c)
    p2, pi, ci = np.array(p2), np.array(pi), np.array(ci)
    keep = np.isin(pi, np.nonzero(np.bincount(pi, minlength=300) >= 2)[0])
    p2, ci = p2[keep], ci[keep]
    _u, pi = np.unique(pi[keep], return_inverse=True)
    X = X[_u]
    Rp = np.array([ocam.rodrigues(rng.normal(0, 0.01, 3)) @ R[c] for c in range(6)])
    tp = t.reshape(6, 3, 1) + rng.normal(0, 0.01, (6, 3, 1))
    X0 = X + rng.normal(0, 0.03, X.shape)
    # the residual function is the oracle's pinhole one
    pts, rm, tt, res = sba.bundle_adjust_points_and_extrinsics(p2, X0, pi, ci, K, D, Rp, tp, calib.project_points, max_iter=0)
    Rq = np.array([ocam.rodrigues(osba.rodrigues_to_vec(r)) for r in Rp])
    f0 = osba.residuals(X0, Rq, tp, K, D, pi, ci, p2, ocam.project_points)
    assert np.abs(res["before"] - f0).max() < 1e-8
    pts, rm, tt, res = sba.bundle_adjust_points_and_extrinsics(p2, X0, pi, ci, K, D, Rp, tp, calib.project_points)
    info = dict(sba.last_info)
    assert info["status_name"] in ("ftol", "gtol") and info["cost_final"] < 0.05 * info["cost_initial"], info
    fa = osba.residuals(pts, rm, tt, K, D, pi, ci, p2, ocam.project_points)
    assert np.abs(fa - res["after"]).max() < 1e-8 and abs(osba.cauchy_cost(fa) - info["cost_final"]) < 1e-8
    assert np.sqrt(np.mean(res["after"] ** 2)) < 0.4                               # px: the injected noise level
    _p, _r, _t, _res, oopt = osba.bundle_adjust_points_and_extrinsics(p2, X0, pi, ci, K, D, Rp, tp, max_nfev=150,
                                                                      project_func=ocam.project_points)
    assert info["cost_final"] <= oopt.cost * (1 + 1e-6), (info, oopt.cost)


def _rig_problem(seed=5, n_pts=400):
    from acinoset_amd import synth
    rng = np.random.default_rng(seed)
    K, D, R, t = synth.make_rig()
    X = np.array([2.0, 6.5, 0.7]) + rng.normal(0, 1.0, (n_pts, 3))
    p2, pi, ci = [], [], []
    for p in range(n_pts):
        for c in np.sort(rng.choice(6, size=rng.integers(2, 7), replace=False)):
            p2.append(ocam.project_points_fisheye(X[p:p + 1], K[c], D[c], R[c], t[c])[0] + rng.normal(0, 0.3, 2))
            pi.append(p)
            ci.append(c)
    Rp = np.array([ocam.rodrigues(rng.normal(0, 0.015, 3)) @ R[c] for c in range(6)])
    tp = t.reshape(6, 3, 1) + rng.normal(0, 0.02, (6, 3, 1))
    return np.array(p2), X + rng.normal(0, 0.05, X.shape), np.array(pi), np.array(ci), K, D, Rp, tp


def _mp_sba_worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from acinoset_amd import sba
        torch.cuda.set_device(0)
        p2, X0, pi, ci, K, D, Rp, tp = _rig_problem()
        lo, hi = rank * len(X0) // world, (rank + 1) * len(X0) // world        # this rank's points
        m = (pi >= lo) & (pi < hi)
        pts, rm, tt, res = sba.bundle_adjust_points_and_extrinsics_sharded(p2[m], X0[lo:hi], pi[m] - lo, ci[m], K, D, Rp, tp)
        np.savez(out_path + f".{rank}.npz", pts=pts, r=rm, t=tt, after=res["after"], cost=sba.last_info["cost_final"],
                 cost0=sba.last_info["cost_initial"], it=sba.last_info["iterations"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_points_sharded_over_processes_equal_single_solve(gsba, world, tmp_path):
    """BASELINE config 5's extrinsic refinement: points sharded over processes (gloo here, every rank on this GPU; RCCL
    on a node), cameras shared through the all-reduced camera system.  Must reproduce the single-process solve."""
    import torch.multiprocessing as mp
    sba, _ = gsba
    p2, X0, pi, ci, K, D, Rp, tp = _rig_problem()
    pts1, r1, t1, res1 = sba.bundle_adjust_points_and_extrinsics(p2, X0, pi, ci, K, D, Rp, tp)
    one = dict(sba.last_info)
    out = str(tmp_path / "sba")
    mp.spawn(_mp_sba_worker, args=(world, 29760 + world, out), nprocs=world, join=True)
    parts = [np.load(out + f".{r}.npz") for r in range(world)]
    for p in parts:                                   # every rank: the same global cost, iteration count and poses
        assert abs(float(p["cost0"]) - one["cost_initial"]) < 1e-9 * one["cost_initial"]
        assert abs(float(p["cost"]) - one["cost_final"]) < 1e-6 * one["cost_final"], (float(p["cost"]), one)
        assert np.array_equal(p["r"], parts[0]["r"]) and np.array_equal(p["t"], parts[0]["t"])
        assert np.abs(p["r"] - r1).max() < 1e-6 and np.abs(p["t"] - t1).max() < 1e-6
    assert np.abs(np.concatenate([p["pts"] for p in parts]) - pts1).max() < 1e-5


def test_sharded_solve_reports_a_failing_reduction(gsba):
    """No process group: the reduction callback fails inside the C solve, which must stop with ACINO_ERR_CALLBACK and
    hand the Python exception back (no exception may unwind through the C frames)."""
    sba, _ = gsba
    import torch.distributed as dist
    assert not dist.is_initialized()
    p2, X0, pi, ci, K, D, Rp, tp = _rig_problem(n_pts=40)
    with pytest.raises((RuntimeError, ValueError)):
        sba.bundle_adjust_points_and_extrinsics_sharded(p2, X0, pi, ci, K, D, Rp, tp)
    pts, _r, _t, _res = sba.bundle_adjust_points_and_extrinsics(p2, X0, pi, ci, K, D, Rp, tp)     # the library is still usable
    assert np.isfinite(pts).all()


# ---------------------------------------------------------------------------------------------------------------
# BASELINE config 5: batched FTE -> extrinsic refinement over every sequence (src/calib/calib.py:345-390, app.py:201-223)
# ---------------------------------------------------------------------------------------------------------------
def _pair_distances(r1, t1, r2, t2):
    """Gauge-invariant comparison of two 6-camera rigs: over the adjacent camera pairs the largest relative-rotation
    difference (deg) and baseline-direction difference (deg), and the largest difference of baseline RATIOS (the overall
    scale is free in points + extrinsics)."""
    rot, dire, lens1, lens2 = [], [], [], []
    n = len(r1)
    for a in range(n - 1):
        Ra, ba = osba.relative_pose(r1, t1, a, a + 1)
        Rb, bb = osba.relative_pose(r2, t2, a, a + 1)
        rot.append(np.degrees(np.arccos(np.clip((np.trace(Ra @ Rb.T) - 1) / 2, -1, 1))))
        dire.append(np.degrees(np.arccos(np.clip(ba @ bb / (np.linalg.norm(ba) * np.linalg.norm(bb)), -1, 1))))
        lens1.append(np.linalg.norm(ba))
        lens2.append(np.linalg.norm(bb))
    lens1, lens2 = np.array(lens1), np.array(lens2)
    return max(rot), max(dire), float(np.abs(lens1 / lens1.sum() - lens2 / lens2.sum()).max())


def _clips(synth, n_clips, n_frames, kind="trot"):
    seqs = [synth.make_sequence(n_frames, kind, seed=20210313 + i) for i in range(n_clips)]
    return seqs, (seqs[0]["K"], seqs[0]["D"], seqs[0]["R"], seqs[0]["t"])


def _perturb(R, t, rng, deg=0.5, cm=1.0):
    Rp = np.array([ocam.rodrigues(rng.normal(0, 1, 3) / np.sqrt(3) * np.radians(deg)) @ R[c] for c in range(len(R))])
    tp = np.asarray(t, dtype=np.float64).reshape(-1, 3, 1) + rng.normal(0, 1, (len(R), 3, 1)) / np.sqrt(3) * cm * 1e-2
    return Rp, tp


def test_dense_extrinsic_refinement_against_the_scipy_oracle(gsba):
    """The dense entry (device-side observation lists) on 2 clips x 40 frames of FTE-like points: (1) the observation
    lists are the ones the reference's loops would build (points with > 1 view, calib.py:281); (2) the cost the solver
    reports at its end state is the oracle's cost function of that state; (3) it ends at or below the scipy oracle
    (calib.py:369-390 settings, consistent Jacobian mask) started from the same point; (4) bf16 / fp32 mode ends within a
    stated distance of the fp64 end state."""
    sba, calib = gsba
    import torch
    from acinoset_amd import fte, synth
    seqs, (K, D, R, t) = _clips(synth, 2, 40)
    det = np.concatenate([s["det"] for s in seqs], 0)
    rng = np.random.default_rng(3)
    pos_true = np.concatenate([np.asarray(fte.cheetah_fk(s["q_true"])) for s in seqs], 0)
    X0 = pos_true + rng.normal(0, 0.01, pos_true.shape)
    Rp, tp = _perturb(R, t, rng)
    # (1) observation lists
    keep, uv, cam_idx, pt_start, pt_obs = sba.dense_observations(torch.as_tensor(det, device="cuda"), 0.5)
    seen = det[..., 2].transpose(0, 2, 1) > 0.5
    assert np.array_equal(keep.cpu().numpy(), seen.sum(-1) >= 2)
    pi, ci, p2 = [], [], []
    for pid, (n, l) in enumerate(zip(*np.nonzero(seen.sum(-1) >= 2))):
        for c in np.nonzero(seen[n, l])[0]:
            pi.append(pid); ci.append(c); p2.append(det[n, c, l, :2])
    pi, ci, p2 = np.array(pi), np.array(ci), np.array(p2)
    assert np.array_equal(cam_idx.cpu().numpy(), ci) and np.array_equal(uv.cpu().numpy(), p2)
    assert np.array_equal(np.diff(pt_start.cpu().numpy()), np.bincount(pi))
    # (2) + (3)
    pts, rm, tt, info = sba.bundle_adjust_dense_points_and_extrinsics(det, X0, K, D, Rp, tp, 0.5, max_iter=100)
    kp = keep.cpu().numpy()
    end = osba.residuals(pts.cpu().numpy()[kp], rm, tt, K, D, pi, ci, p2)
    assert abs(osba.cauchy_cost(end) - info["cost_final"]) < 1e-9 * info["cost_final"]
    # (the gauge-free points + extrinsics problem creeps along its flat directions: the iteration limit is a normal end)
    assert info["status_name"] in ("ftol", "gtol", "max_iter") and info["n_points"] == int(kp.sum()) and info["n_obs"] == len(pi)
    _p, _r, _t, _res, oopt = osba.bundle_adjust_points_and_extrinsics(p2, X0[kp], pi, ci, K, D, Rp, tp, max_nfev=60,
                                                                      consistent_mask=True)
    assert info["cost_final"] <= oopt.cost * (1 + 1e-6), (info["cost_final"], oopt.cost)
    # (4) mixed precision
    _pb, rb, tb, ib = sba.bundle_adjust_dense_points_and_extrinsics(det, X0, K, D, Rp, tp, 0.5, max_iter=100, precision="bf16")
    rot, dire, ratio = _pair_distances(rm, tt, rb, tb)
    print(f"dense SBA 2 x 40: fp64 cost {info['cost_final']:.4f} ({info['iterations']} it), bf16 rows {ib['cost_final']:.4f} "
          f"({ib['iterations']} it); bf16 vs fp64 end state: {rot:.2e} deg, {dire:.2e} deg, baseline ratio {ratio:.2e}")
    # (observed 2.3e-2 deg / 2.1e-2 deg / 3.7e-5 with both runs at the iteration limit, still creeping along the gauge directions)
    assert ib["cost_final"] < 1.001 * info["cost_final"] and rot < 0.06 and dire < 0.06 and ratio < 2e-4
    # ... and the bf16 mode against the ORACLE directly: its end state priced by the oracle's fp64 cost function (not by the
    # mode's own sums) - what the bf16 rows cost the reported sum, and against the scipy end state from the same start (60
    # function evaluations of the reference's settings: scipy is nowhere near converged there, which is the KAT-2 story again,
    # so only the cost is compared - the rig is compared with the fp64 mode's above)
    end_b = osba.residuals(_pb.cpu().numpy()[kp], rb, tb, K, D, pi, ci, p2)
    cost_b = osba.cauchy_cost(end_b)
    print(f"dense SBA 2 x 40, bf16 rows vs the oracle: oracle-priced cost {cost_b:.4f} (the mode reports {ib['cost_final']:.4f}), "
          f"fp64 mode {info['cost_final']:.4f}, scipy after 60 evaluations {oopt.cost:.4f}")
    assert abs(cost_b - ib["cost_final"]) < 2e-3 * cost_b
    assert cost_b <= oopt.cost * (1 + 1e-6) and cost_b < 1.001 * osba.cauchy_cost(end)


@pytest.mark.parametrize("precision", ["f64", "bf16"])
def test_config5_chain_recovers_a_perturbed_rig(gsba, precision):
    """BASELINE config 5 end to end on one GPU at test size (8 clips x 120 frames): the rig is perturbed by 0.5 deg /
    1 cm per camera, the clips are solved as one FTE chain with that rig, and the marker positions of every clip with
    their above-threshold detections refine the six shared extrinsics.  The refined rig must be closer to the true one
    than the perturbed rig in every gauge-invariant measure, and the reprojection rms must come down to the detection
    noise (2 px injected)."""
    sba, calib = gsba
    from acinoset_amd import synth
    seqs, (K, D, R, t) = _clips(synth, 8, 120)
    rng = np.random.default_rng(11)
    Rp, tp = _perturb(R, t, rng)
    r_new, t_new, info = sba.refine_extrinsics_from_clips([s["det"] for s in seqs], K, D, Rp, tp, seqs[0]["Ts"],
                                                          precision=precision, fte_iter=40, sba_iter=60)
    before = _pair_distances(R, t, Rp, tp)
    after = _pair_distances(R, t, r_new, t_new)
    print(f"config 5 chain ({precision}): FTE {info['fte']['iter']} it ({info['fte']['status_name']}); SBA {info['sba']['iterations']} "
          f"it, {info['sba']['n_points']} points / {info['sba']['n_obs']} observations, rms {info['sba']['rms_before']:.2f} -> "
          f"{info['sba']['rms_after']:.2f} px; rig error (rot deg, dir deg, baseline ratio) {before} -> {after}")
    assert info["sba"]["status_name"] in ("ftol", "gtol", "max_iter")
    assert after[0] < 0.5 * before[0] and after[1] < 0.5 * before[1] and after[2] < 0.5 * before[2], (before, after)
    assert info["sba"]["rms_after"] < 1.2 * info["sba"]["rms_before"] and info["sba"]["rms_after"] < 6.0


def _mp_dense_worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from acinoset_amd import fte, sba, synth
        torch.cuda.set_device(0)
        seqs, (K, D, R, t) = _clips(synth, 4, 30)
        rng = np.random.default_rng(5)
        Rp, tp = _perturb(R, t, rng)
        mine = seqs[rank::world]                                # this rank's sequences
        det = np.concatenate([s["det"] for s in mine], 0)
        X0 = np.concatenate([np.asarray(fte.cheetah_fk(s["q_true"])) for s in mine], 0)
        _p, rm, tt, info = sba.bundle_adjust_dense_points_and_extrinsics(det, X0, K, D, Rp, tp, 0.5, max_iter=30,
                                                                         group=dist.group.WORLD)
        np.savez(out_path + f".{rank}.npz", r=rm, t=tt, cost=info["cost_final"], sizes=np.array(info["reduce_sizes"]),
                 calls=info["reduce_calls"], it=info["iterations"])
    finally:
        dist.destroy_process_group()


def test_config5_extrinsics_sharded_by_sequence_all_reduce_payload(gsba, tmp_path):
    """Config 5's placement: the sequences are spread over the ranks, the six extrinsics are shared.  Per LM iteration the
    ranks exchange the reduced camera system - SURVEY 8(e): a 36 x 36 block + 36-vector = (6C)^2 + 6C = 1 332 doubles -,
    the camera blocks [U | g_c] (21 C + 6 C = 162) and three scalars; nothing that grows with the number of points.
    Two processes (gloo, both on this GPU; RCCL on a node) must end with identical poses equal to the one-process solve."""
    import torch.multiprocessing as mp
    sba, _ = gsba
    from acinoset_amd import fte, synth
    seqs, (K, D, R, t) = _clips(synth, 4, 30)
    rng = np.random.default_rng(5)
    Rp, tp = _perturb(R, t, rng)
    order = seqs[0::2] + seqs[1::2]
    det = np.concatenate([s["det"] for s in order], 0)
    X0 = np.concatenate([np.asarray(fte.cheetah_fk(s["q_true"])) for s in order], 0)
    _p, r1, t1, one = sba.bundle_adjust_dense_points_and_extrinsics(det, X0, K, D, Rp, tp, 0.5, max_iter=30)
    out = str(tmp_path / "dense")
    mp.spawn(_mp_dense_worker, args=(2, 29811, out), nprocs=2, join=True)
    parts = [np.load(out + f".{r}.npz") for r in range(2)]
    C = 6
    for p in parts:
        assert set(p["sizes"].tolist()) == {1, 21 * C + 6 * C, (6 * C) ** 2 + 6 * C}, p["sizes"]
        assert abs(float(p["cost"]) - one["cost_final"]) < 1e-3 * one["cost_final"]      # (30 iterations, not converged)
        assert np.array_equal(p["r"], parts[0]["r"]) and np.array_equal(p["t"], parts[0]["t"])
        # (another summation order: the iterates drift apart along the gauge directions at the 1e-5 level within the 30
        #  iterations; the gauge-invariant part agrees)
        rot, dire, ratio = _pair_distances(p["r"], p["t"], r1, t1)
        assert rot < 2e-3 and dire < 2e-3 and ratio < 1e-5 and np.abs(p["r"] - r1).max() < 1e-3, (rot, dire, ratio)


@pytest.mark.parametrize("precision", ["f64", "bf16"])
def test_config5_extrinsic_refinement_at_full_size(gsba, precision):
    """BASELINE config 5's SBA half at its FULL size (64 sequences x 1 000 frames: 1.28 M marker positions, 6.5 M
    observations, six shared extrinsics), through size-independent properties: the observation lists are the detections
    above the threshold of points with > 1 view; the cost never increases; the reprojection rms falls from the perturbed
    rig's level to the detection-noise floor (2 px injected + 5 mm of position noise); every gauge-invariant rig error is at
    least halved; both precisions end at the same rig."""
    sba, calib = gsba
    import torch
    from acinoset_amd import synth
    seq = synth.make_sequence(1000, "trot")
    K, D, R, t = seq["K"], seq["D"], seq["R"], seq["t"]
    det64 = torch.as_tensor(seq["det"], device="cuda").repeat(64, 1, 1, 1)
    gen = torch.Generator(device="cuda").manual_seed(3)
    pos64 = torch.as_tensor(seq["pos_true"], device="cuda").repeat(64, 1, 1)
    pos64 = pos64 + 0.005 * torch.randn(pos64.shape, dtype=torch.float64, device="cuda", generator=gen)
    Rp, tp = _perturb(R, t, np.random.default_rng(7))
    _pts, r_new, t_new, info = sba.bundle_adjust_dense_points_and_extrinsics(det64, pos64, K, D, Rp, tp, 0.5, precision=precision,
                                                                             max_iter=25)
    lik = det64[..., 2] > 0.5
    views = lik.sum(1)                                   # [N, L]
    assert info["n_points"] == int((views > 1).sum()) and info["n_obs"] == int((lik & (views > 1).unsqueeze(1)).sum())
    assert info["n_points"] > 1.2e6 and info["n_obs"] > 6e6
    before, after = _pair_distances(R, t, Rp, tp), _pair_distances(R, t, r_new, t_new)
    print(f"config 5 SBA at full size ({precision}): {info['iterations']} it, cost {info['cost_initial']:.6g} -> {info['cost_final']:.6g}, "
          f"rms {info['rms_before']:.2f} -> {info['rms_after']:.2f} px, rig error {before} -> {after}")
    assert info["cost_final"] < info["cost_initial"] and info["accepted"] >= 5
    assert info["rms_before"] > 4.0 and info["rms_after"] < 2.2
    assert after[0] < 0.5 * before[0] and after[1] < 0.5 * before[1] and after[2] < 0.5 * before[2], (before, after)


_TABLE_PATH_SCRIPT = r'''
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from acinoset_amd import fte, sba, synth
seq = synth.make_sequence(40, "trot")
K, D, R, t = seq["K"], seq["D"], seq["R"], seq["t"]
rng = np.random.default_rng(11)
dw = np.deg2rad(0.4) * rng.standard_normal((6, 3))
from oracle import camera as ocam
Rp = np.stack([ocam.rodrigues(dw[c]) @ R[c] for c in range(6)])
tp = np.asarray(t, dtype=np.float64).reshape(-1, 3, 1) + 0.008 * rng.standard_normal((6, 3, 1))
X0 = np.asarray(fte.cheetah_fk(seq["q_true"])) + 0.004 * rng.standard_normal((40, 20, 3))
out = {}
def A(x):
    return (x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x))
for prec in ("f64", "bf16"):
    p, rm, tt, info = sba.bundle_adjust_dense_points_and_extrinsics(seq["det"], X0, K, D, Rp, tp, 0.5, max_iter=12, precision=prec)
    out[prec] = dict(cost0=info["cost_initial"], cost=info["cost_final"], it=info["iterations"], acc=info["accepted"],
                     r=A(rm).tolist(), t=A(tt).tolist(), p=A(p)[:50].tolist())
print("RESULT" + json.dumps(out))
'''


def test_fused_path_equals_the_table_path(gsba):
    """The fused kernels (one lane per (point, camera) slot, Schur complement on the matrix cores, no coupling table) against
    the table path (one thread per point, dense W table, LDS atomics; ACINO_SBA_UNFUSED=1) on the same problem: the same
    LM trajectory in fp64 - equal iteration / acceptance counts, costs to 1e-10, poses and points to 1e-8; with bf16 rows
    (fp32 sums in another order: accept / reject decisions at the noise floor differ, the free rig drifts along its gauge)
    the same end cost to 1e-2 (12 iterations, not converged)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = {}
    for tag, extra in (("fused", {}), ("table", {"ACINO_SBA_UNFUSED": "1"})):
        env = dict(os.environ, **extra)
        if not extra:
            env.pop("ACINO_SBA_UNFUSED", None)
        r = subprocess.run([sys.executable, "-c", _TABLE_PATH_SCRIPT, root], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        runs[tag] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1][6:])
    for prec in ("f64", "bf16"):
        a, b = runs["fused"][prec], runs["table"][prec]
        if prec == "f64":
            assert a["it"] == b["it"] and a["acc"] == b["acc"], (prec, a["it"], b["it"], a["acc"], b["acc"])
        tol = 1e-10 if prec == "f64" else 1e-2
        assert abs(a["cost0"] - b["cost0"]) <= 1e-12 * abs(b["cost0"])
        assert abs(a["cost"] - b["cost"]) <= tol * abs(b["cost"]), (prec, a["cost"], b["cost"])
        if prec == "f64":                                  # (bf16 rows: the iterates drift along the 7 gauge directions of a free rig)
            assert np.abs(np.array(a["r"]) - np.array(b["r"])).max() < 1e-8 and np.abs(np.array(a["t"]) - np.array(b["t"])).max() < 1e-8
            assert np.abs(np.array(a["p"]) - np.array(b["p"])).max() < 1e-8


def test_fused_path_edge_cases(gsba):
    """max_iter = 0 (evaluation only: the cost of the start, its gradient norm, nothing moved), three points (less than one
    batch of a wave), a point seen by ONE camera (its 3 x 3 block is singular without the damping), a point the lists never
    mention, duplicates and bad camera indices (refused before anything runs)."""
    sba, calib = gsba
    from acinoset_amd import synth
    rng = np.random.default_rng(23)
    K, D, R, t = synth.make_rig()
    t = t.reshape(6, 3, 1)
    X = np.array([2.0, 6.5, 0.7]) + rng.normal(0, 0.4, (4, 3))
    pi = np.array([0, 0, 0, 1, 1, 2, 2, 2, 2])            # point 3 has no observation at all; point 1 two; ...
    ci = np.array([0, 1, 2, 1, 4, 0, 2, 3, 5])
    uv = np.stack([ocam.project_points_fisheye(X[p:p + 1], K[c], D[c], R[c], t[c])[0] for p, c in zip(pi, ci)]) + rng.normal(0, 0.5, (9, 2))
    X0 = X + rng.normal(0, 0.02, X.shape)
    r0 = osba.residuals(X0, R, t, K, D, pi, ci, uv)
    c0 = 0.5 * np.log1p(r0 ** 2).sum()
    pts, rm, tt, res = sba.bundle_adjust_points_and_extrinsics(uv, X0, pi, ci, K, D, R, t, max_iter=0)
    info = dict(sba.last_info)
    assert info["iterations"] == 0 and abs(info["cost_initial"] - c0) < 1e-9 * c0 and info["cost_final"] == info["cost_initial"]
    assert info["gnorm_inf"] > 0 and np.array_equal(pts, X0) and np.abs(rm - R).max() < 1e-12
    assert np.abs(res["before"] - r0).max() < 1e-9
    # a few iterations: the cost falls, the unobserved point stays where it was
    pts, rm, tt, res = sba.bundle_adjust_points_and_extrinsics(uv, X0, pi, ci, K, D, R, t, max_iter=15)
    info = dict(sba.last_info)
    assert info["cost_final"] < info["cost_initial"] and np.array_equal(pts[3], X0[3]) and np.isfinite(pts).all()
    # one camera only for a point: points-only solve, the damping carries the rank-2 block
    pi1, ci1 = np.array([0, 0, 1]), np.array([0, 1, 3])
    uv1 = np.stack([ocam.project_points_fisheye(X[p:p + 1], K[c], D[c], R[c], t[c])[0] for p, c in zip(pi1, ci1)])
    pts1, _res = sba.bundle_adjust_points_only(uv1, X0[:2], pi1, ci1, K, D, R, t, max_iter=20)
    assert np.isfinite(pts1).all() and dict(sba.last_info)["cost_final"] <= dict(sba.last_info)["cost_initial"]
    with pytest.raises(ValueError, match="observed twice"):
        sba.bundle_adjust_points_and_extrinsics(np.vstack([uv, uv[:1]]), X0, np.append(pi, 0), np.append(ci, 0), K, D, R, t, max_iter=2)
    with pytest.raises(ValueError, match="out of range"):
        sba.bundle_adjust_points_and_extrinsics(uv, X0, pi, np.where(ci == 5, 6, ci), K, D, R, t, max_iter=2)


@pytest.mark.parametrize("n_cams", [6, 8])
def test_library_refuses_duplicate_observations_on_both_paths(gsba, n_cams):
    """The C ABI's own input check (k_sba_check), without the Python wrapper's: two observations of one (point, camera) pair
    and a camera index out of range answer ACINO_ERR_INVALID_ARG on the fused path (6 cameras) AND on the table path (8
    cameras: one coupling block per pair, a duplicate would silently overwrite it); a clean list still solves."""
    sba, calib = gsba
    from acinoset_amd import synth
    rng = np.random.default_rng(5)
    K6, D6, R6, t6 = synth.make_rig()
    idx = np.arange(n_cams) % 6
    K, D, R = K6[idx], D6[idx], R6[idx]
    t = t6.reshape(6, 3, 1)[idx] + rng.normal(0, 0.05, (n_cams, 3, 1)) * (np.arange(n_cams) >= 6)[:, None, None]
    X = np.array([2.0, 6.5, 0.7]) + rng.normal(0, 0.4, (12, 3))
    pi = np.repeat(np.arange(12), n_cams)
    ci = np.tile(np.arange(n_cams), 12)
    uv = np.stack([ocam.project_points_fisheye(X[p:p + 1], K[c], D[c], R[c], t[c])[0] for p, c in zip(pi, ci)])
    X0 = X + rng.normal(0, 0.02, X.shape)
    args = dict(optimize_cameras=True, f_scale=1.0, max_iter=3, ftol=1e-10, gtol=1e-12, host_checks=False)
    out = sba._solve(uv, X0, pi, ci, K, D, R, t, **args)
    assert np.isfinite(out[0]).all()
    with pytest.raises(ValueError, match="two observations"):
        sba._solve(np.vstack([uv, uv[:1]]), X0, np.append(pi, 0), np.append(ci, 0), K, D, R, t, **args)
    with pytest.raises(ValueError, match="two observations|out of range"):
        sba._solve(uv, X0, pi, np.where(ci == n_cams - 1, n_cams, ci), K, D, R, t, **args)
    out = sba._solve(uv, X0, pi, ci, K, D, R, t, **args)                 # the library is still usable
    assert np.isfinite(out[0]).all()


def test_fused_solve_repeats_bit_for_bit(gsba):
    """Every sum that decides accept / reject (cost, predicted reduction, trial cost) and every block of the normal equations
    is reduced in a fixed order: the same problem solved three times gives the same bits - points, poses, costs, counts."""
    sba, calib = gsba
    import torch
    from acinoset_amd import fte, synth
    seqs, (K, D, R, t) = _clips(synth, 3, 200)
    Rp, tp = _perturb(R, t, np.random.default_rng(9))
    det = np.concatenate([s["det"] for s in seqs], 0)
    X0 = np.concatenate([np.asarray(fte.cheetah_fk(s["q_true"])) for s in seqs], 0)
    outs = []
    for rep in range(3):
        if rep == 2:                                        # (foreign work on the device between the runs)
            junk = torch.randn(4096, 4096, device="cuda") @ torch.randn(4096, 4096, device="cuda")
            del junk
        p, rm, tt, info = sba.bundle_adjust_dense_points_and_extrinsics(det, X0, K, D, Rp, tp, 0.5, max_iter=15)
        A = lambda x: (x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x))
        outs.append((A(p).copy(), A(rm).copy(), A(tt).copy(), info["cost_initial"], info["cost_final"], info["iterations"], info["accepted"]))
    assert outs[0][5] >= 5 and outs[0][4] < outs[0][3]
    for o in outs[1:]:
        assert o[3:] == outs[0][3:]
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1]) and np.array_equal(o[2], outs[0][2])
