"""CPU: the oracle bundle adjustment (oracle/sba.py) against KAT-2, the SBA end state the reference recorded in
src/calib_with_gui.ipynb cell 29 on the shipped sunday_amelia checkerboard points."""
import os

import numpy as np
import pytest

from oracle import camera, sba


def kat2_problem(g, tag, ca, cb):
    K, D, R, t = g[f"{tag}_K"], g[f"{tag}_D"], g[f"{tag}_R"], g[f"{tag}_t"]
    img = [g[f"cam{ca}_points"], g[f"cam{cb}_points"]]
    names = [list(g[f"cam{ca}_fnames"]), list(g[f"cam{cb}_fnames"])]
    return img, names, tuple(int(v) for v in g["board_shape"]), K, D, R, t


def test_prepare_board_data_and_sparsity(golden_dir):
    g = np.load(os.path.join(golden_dir, "kat1_sunday_amelia.npz"))
    img, names, shape, K, D, R, t = kat2_problem(g, "static", 3, 4)
    names[1] = names[1][:-2] + ["only_b_1", "only_b_2"]          # two boards seen by one camera only
    p2, p3, pi, ci = sba.prepare_calib_board_data(img, names, shape, K, D, R, t, camera.triangulate_points_fisheye)
    assert p2.dtype == np.float32 and p3.dtype == np.float32     # calib.py:259-260
    assert p3.shape == (14 * 54, 3) and p2.shape == (2 * 14 * 54, 2)
    assert np.array_equal(np.bincount(pi), np.full(14 * 54, 2)) and set(ci) == {0, 1}
    A = sba.sparsity(2, 6, ci, len(p3), pi).tocsr()
    assert A.shape == (2 * len(p2), 12 + 3 * len(p3)) and (A.sum(axis=1) == 9).all()
    r0 = sba.residuals(p3.astype(np.float64), R, t, K, D, pi, ci, p2)
    assert np.abs(r0).max() < 2.0                                 # a calibrated rig reprojects to sub-pixel / pixel level


def test_kat2_static_pair(golden_dir):
    """Static pair: scipy TRF with the reference's settings reproduces the recorded run exactly (cost to the 5
    printed digits, the 50 function evaluations, after-statistics to 1e-6)."""
    g = np.load(os.path.join(golden_dir, "kat1_sunday_amelia.npz"))
    img, names, shape, K, D, R, t = kat2_problem(g, "static", 3, 4)
    data = sba.prepare_calib_board_data(img, names, shape, K, D, R, t, camera.triangulate_points_fisheye)
    pts, rm, tt, residuals, res = sba.bundle_adjust_points_and_extrinsics(*data, K, D, R, t)
    cost_rec, nfev_rec, mean_rec, std_rec = g["recorded_sba"][1]
    assert abs(sba.cauchy_cost(residuals["before"]) - g["recorded"][1][2]) / g["recorded"][1][2] < 5e-5
    assert abs(res.cost - cost_rec) / cost_rec < 5e-5 and abs(sba.cauchy_cost(residuals["after"]) - res.cost) < 1e-9
    assert res.nfev == int(nfev_rec)
    assert abs(residuals["after"].mean() - mean_rec) < 1e-6 and abs(residuals["after"].std() - std_rec) < 1e-6
    # the solution is a camera rig: rotations stay orthonormal, residual function consistent with the outputs
    assert np.abs(rm @ rm.transpose(0, 2, 1) - np.eye(3)).max() < 1e-12
    assert np.abs(sba.residuals(pts, rm, tt, K, D, data[2], data[3], data[0]) - residuals["after"]).max() < 1e-9


@pytest.mark.timeout(600)
def test_kat2_rotating_pair(golden_dir):
    """Rotating pair: the recorded run crept for 690 evaluations without converging (first-order optimality
    3.6e+03) and its stopping point is not reproducible - TRF's xtol test fires anywhere on that plateau (a 1e-8
    change of the start rotations moves it from 296 to 51 evaluations here).  Pinned: the start cost (KAT-1), and
    an end state within 0.2 % of the recorded cost / after-std."""
    g = np.load(os.path.join(golden_dir, "kat1_sunday_amelia.npz"))
    img, names, shape, K, D, R, t = kat2_problem(g, "rotating", 1, 2)
    data = sba.prepare_calib_board_data(img, names, shape, K, D, R, t, camera.triangulate_points_fisheye)
    pts, rm, tt, residuals, res = sba.bundle_adjust_points_and_extrinsics(*data, K, D, R, t)
    cost_rec, _nfev, mean_rec, std_rec = g["recorded_sba"][0]
    assert abs(sba.cauchy_cost(residuals["before"]) - g["recorded"][0][2]) / g["recorded"][0][2] < 5e-5
    assert res.cost < sba.cauchy_cost(residuals["before"]) and abs(res.cost - cost_rec) / cost_rec < 2e-3
    assert abs(residuals["after"].mean() - mean_rec) < 1e-3 and abs(residuals["after"].std() - std_rec) / std_rec < 2e-3


def test_points_only_lowers_cost(golden_dir):
    g = np.load(os.path.join(golden_dir, "kat1_sunday_amelia.npz"))
    img, names, shape, K, D, R, t = kat2_problem(g, "static", 3, 4)
    data = sba.prepare_calib_board_data(img, names, shape, K, D, R, t, camera.triangulate_points_fisheye)
    pts, residuals, res = sba.bundle_adjust_points_only(*data, K, D, R, t)
    assert res.cost < sba.cauchy_cost(residuals["before"], 50) and res.optimality < 1e-2    # points-only converges (gradient from 1e+3 to 1e-3)


def test_kat2_end_state_is_an_artefact_of_the_jacobian_mask(golden_dir):
    """Why the recorded KAT-2 runs stop far from stationarity: calib.py:196-207 marks six CONTIGUOUS columns per
    camera, while calib.py:373-375 lays the parameters out as [all rvecs | all tvecs | points].  The SAME scipy call
    with a mask that matches the layout converges (static pair: cost 18.8205, first-order optimality < 1) - the
    stationary point the GPU solve reaches (tests/test_gpu_sba.py).  The reference's end state differs from it by the
    relative pose printed in DESIGN.md section 7."""
    g = np.load(os.path.join(golden_dir, "kat1_sunday_amelia.npz"))
    img, names, shape, K, D, R, t = kat2_problem(g, "static", 3, 4)
    data = sba.prepare_calib_board_data(img, names, shape, K, D, R, t, camera.triangulate_points_fisheye)
    A_ref = sba.sparsity(2, 6, data[3], len(data[1]), data[2]).tocsr()
    A_ok = sba.sparsity_by_layout(2, data[3], len(data[1]), data[2]).tocsr()
    row0 = int(np.where(data[3] == 0)[0][0]) * 2                   # an observation of camera 0
    assert set(A_ref[row0].indices[:6]) == {0, 1, 2, 3, 4, 5}      # the reference's pattern ...
    assert set(A_ok[row0].indices[:6]) == {0, 1, 2, 6, 7, 8}       # ... and where camera 0's parameters really are
    _p, rm_ref, tt_ref, _r, res_ref = sba.bundle_adjust_points_and_extrinsics(*data, K, D, R, t)
    _p, rm_ok, tt_ok, _r, res_ok = sba.bundle_adjust_points_and_extrinsics(*data, K, D, R, t, consistent_mask=True)
    assert res_ref.optimality > 1e3 and res_ok.optimality < 1.0
    assert abs(res_ok.cost - 18.8205) < 2e-3 and res_ok.cost < res_ref.cost - 3.5
    ang, dire, len_ref, len_ok = sba.pose_distance(rm_ref, tt_ref, rm_ok, tt_ok)
    assert 0.0 < ang < 1.0 and abs(len_ref - len_ok) < 20.0        # degrees, mm: same rig, measurably different state
