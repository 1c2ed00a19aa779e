"""EKF + RTS smoother.  CPU: the oracle (oracle/ekf.py, a restatement of
src/all_optimizations.py:569-865) against the reference's OWN filter text slice-executed on two synthetic clips
(tests/golden/ekf_ref.npz, made by make_golden.py::gen_ekf).  GPU: the HIP filter against the oracle on seeded
synthetic clips and against the same reference vectors."""
import os

import numpy as np
import pytest

from oracle import ekf as oekf
from oracle import fk as ofk
from oracle import synth as osynth


def test_model_matrices_and_order():
    P0, Q, F = oekf.model_matrices(1 / 120)
    assert P0.shape == (75, 75) and np.allclose(np.diag(P0)[:3], 9) and np.allclose(np.diag(P0)[3:25], (np.pi / 4) ** 2)
    assert np.allclose(np.diag(P0)[25:28], 25) and np.allclose(np.diag(P0)[50 + 3 + 10:], 25) and np.diag(P0)[50 + 12] == 9
    assert np.allclose(Q, Q.T) and Q[0, 0] == (1 / 120) ** 4 / 4 * 6.25 and Q[74, 74] == 200.0 ** 2
    x = np.arange(75.0)
    assert np.allclose((F @ x)[:25], x[:25] + x[25:50] / 120 + x[50:] / 120 ** 2 / 2)
    assert sorted(oekf.EKF_ORDER) == sorted(ofk.ACTIVE) and len(set(oekf.EKF_ORDER)) == 25
    # forward-difference Jacobian against the analytic FK Jacobian: first order in eps = 1e-3
    rng = np.random.default_rng(0)
    pose = rng.uniform(-0.3, 0.3, 25)
    J = oekf.numerical_jacobian(lambda p: oekf.marker_coords(p), pose)
    q = np.zeros(45)
    q[oekf.EKF_ORDER] = pose
    _pos, Ja = ofk.cheetah_fk(q[None], with_jac=True)
    Ja = Ja[0].reshape(60, 45)[:, oekf.EKF_ORDER]
    assert np.abs(J - Ja).max() < 2e-3


def _ref(golden_dir):
    return np.load(os.path.join(golden_dir, "ekf_ref.npz"), allow_pickle=False)


def test_oracle_matches_reference_ekf_text(golden_dir):
    g = _ref(golden_dir)
    assert np.array_equal(g["order45"], oekf.EKF_ORDER)                 # qb_list order -> the FK's sym_list slots
    P0, Q, F = oekf.model_matrices(1 / 120)
    for tag in "ab":
        det, rig = g[f"{tag}_det"], (g[f"{tag}_K"], g[f"{tag}_D"], g[f"{tag}_R"], g[f"{tag}_t"])
        sf = int(g[f"{tag}_start_frame"])
        assert np.array_equal(Q, g[f"{tag}_Q"]) and np.array_equal(F, g[f"{tag}_F"])            # :733-766
        assert np.array_equal(F @ P0 @ F.T + Q, g[f"{tag}_P0"])                                  # :713-730 through :787
        s0 = oekf.initial_state(g[f"{tag}_nose_frames"], g[f"{tag}_nose_xyz"], sf, 1 / 120)     # :699-711
        out = oekf.ekf(det, *rig, 120.0, 0.5, 2704, s0, keep_cov=True)
        est = np.hstack([out["x"], out["dx"], out["ddx"]])
        sm = np.hstack([out["smoothed_x"], out["smoothed_dx"], out["smoothed_ddx"]])
        assert out["outliers_ignored"] == int(g[f"{tag}_outliers"]) > 20                        # the gate is exercised
        # frame 0: nothing has been rounded differently yet
        assert np.array_equal(out["x_pred"][0], g[f"{tag}_pred"][0])
        assert np.abs(est[0] - g[f"{tag}_est"][0]).max() < 1e-9
        # whole clip: float32-ulp level (:628 rounds every prediction to float32; one flipped ulp is carried on)
        for blk, tol in ((slice(0, 25), 5e-6), (slice(25, 50), 5e-5), (slice(50, 75), 1e-3)):
            for got, want in ((est, g[f"{tag}_est"]), (sm, g[f"{tag}_smooth"])):
                scale = max(1.0, np.abs(want[:, blk]).max())
                assert np.abs(got[:, blk] - want[:, blk]).max() < tol * scale
        assert np.array_equal(sm[0], est[0]) and np.array_equal(sm[-1], est[-1])                # :842 leaves both ends
        # one step at a time from the reference's own previous state and covariance: no accumulated flips
        Ph = g[f"{tag}_P_est_head"]
        for i in range(1, len(Ph)):
            one = oekf.ekf(det[i:i + 1], *rig, 120.0, 0.5, 2704, g[f"{tag}_est"][i - 1], keep_cov=True, P_init=Ph[i - 1])
            assert np.array_equal(one["x_pred"][0], g[f"{tag}_pred"][i])
            got = np.hstack([one["x"], one["dx"], one["ddx"]])[0]
            assert np.abs(got - g[f"{tag}_est"][i]).max() < 1e-9 * max(1.0, np.abs(g[f"{tag}_est"][i]).max())
            assert np.abs(one["P_est"][0] - Ph[i]).max() < 1e-9 * np.abs(Ph[i]).max()


def test_oracle_tracks_a_synthetic_sprint():
    seq = osynth.make_sequence(24, "sprint")
    s0 = oekf.initial_state(np.arange(24.0), seq["pos_true"][:, 2], 0, 1 / 120)
    out = oekf.ekf(seq["det"], seq["K"], seq["D"], seq["R"], seq["t"], 120.0, 0.5, 2704, s0)
    q = seq["q_true"][:, oekf.EKF_ORDER]
    assert np.abs(out["smoothed_x"][:, :3] - q[:, :3]).max() < 0.08                 # metres
    assert np.array_equal(out["smoothed_x"][0], out["x"][0]) and np.array_equal(out["smoothed_x"][-1], out["x"][-1])
    assert out["x"].shape == (24, 25) and out["ddx"].shape == (24, 25)


@pytest.mark.gpu
def test_hip_ekf_matches_oracle(gpu_lib):
    from acinoset_amd import ekf, synth
    for n, kind, seed, cams in ((30, "sprint", 1, slice(None)), (20, "loop", 2, slice(None)), (16, "sprint", 3, [0, 2, 5]),
                                (12, "sprint", 4, [1])):
        seq = synth.make_sequence(n, kind, seed=20210313 + seed)
        seq["det"] = seq["det"][:, cams]
        rig = (seq["K"][cams], seq["D"][cams], seq["R"][cams], seq["t"][cams])       # 6, 3 and 1 cameras
        s0 = ekf.initial_state(synth.make_sequence(n, kind, seed=20210313 + seed)["det"], seq["K"], seq["D"], seq["R"],
                               seq["t"], 120.0, 0.5)
        want = oekf.ekf(seq["det"], *rig, 120.0, 0.5, 2704, s0)
        got = ekf.ekf(seq["det"], *rig, 120.0, 0.5, (2704, 1520), states0=s0)
        assert got["outliers_ignored"] == want["outliers_ignored"]
        # tolerance: the reference rounds every predicted state to float32 (:628), so a 1e-16 difference in the
        # arithmetic occasionally flips one float32 ulp (6e-8 relative) and is carried on by the filter
        for k, tol in (("x", 5e-6), ("dx", 5e-5), ("ddx", 1e-3), ("smoothed_x", 5e-6), ("smoothed_dx", 5e-5),
                       ("smoothed_ddx", 1e-3)):
            scale = max(1.0, np.abs(want[k]).max())
            assert np.abs(got[k] - want[k]).max() < tol * scale, (n, kind, k, np.abs(got[k] - want[k]).max())
        assert np.abs(got["smoothed_positions"] - ekf.get_3d_marker_coords(want["smoothed_x"])).max() < 1e-5   # metres


@pytest.mark.gpu
def test_hip_ekf_batch_and_edges(gpu_lib):
    from acinoset_amd import ekf, synth
    seqs = [synth.make_sequence(n, "sprint", seed=7 + i) for i, n in enumerate((12, 12, 9, 2, 1))]
    rig = (seqs[0]["K"], seqs[0]["D"], seqs[0]["R"], seqs[0]["t"])
    s0 = [ekf.initial_state(seqs[0]["det"], *rig, 120.0, 0.5)] * 5
    batch = ekf.ekf_batch([s["det"] for s in seqs], *rig, 120.0, 0.5, (2704, 1520), states0=s0)
    for s, r in zip(seqs, batch):
        one = ekf.ekf(s["det"], *rig, 120.0, 0.5, (2704, 1520), states0=s0[0])
        assert r["x"].shape == (s["det"].shape[0], 25)
        for k in ("x", "dx", "ddx", "smoothed_x", "smoothed_dx", "smoothed_ddx"):
            assert np.array_equal(r[k], one[k])                  # same kernel, same data: bit-identical
    assert np.array_equal(batch[3]["smoothed_x"], batch[3]["x"])  # N < 3: nothing to smooth
    want = oekf.ekf(seqs[2]["det"], *rig, 120.0, 0.5, 2704, s0[0])
    assert np.abs(batch[2]["smoothed_x"] - want["smoothed_x"]).max() < 5e-6
    assert ekf.get_pose_params()["psi_0"] == 5 and len(ekf.POSE_PARAMS) == 25
    with pytest.raises(ValueError):
        ekf.ekf(np.zeros((4, 6, 19, 3)), *rig, 120.0, 0.5, (2704, 1520), states0=s0[0])


@pytest.mark.gpu
def test_hip_smoother_solvers_agree_and_long_clip_tracks(gpu_lib):
    """The smoother's Cholesky path and its pivoting fallback (always taken with smoother_pivoting) give the same
    gains; a 2 000-frame clip of the 2 m/s circle stays on target (covariances stay positive definite)."""
    from acinoset_amd import ekf, synth
    seq = synth.make_sequence(2000, "walk")
    rig = (seq["K"], seq["D"], seq["R"], seq["t"])
    a = ekf.ekf(seq["det"], *rig, 120.0, 0.5, (2704, 1520), with_positions=False)
    b = ekf.ekf(seq["det"], *rig, 120.0, 0.5, (2704, 1520), with_positions=False, smoother_pivoting=True)
    assert np.array_equal(a["x"], b["x"])
    # the first ~20 predicted covariances are ill-conditioned (P0 is far from the steady state): two backward-stable
    # solvers differ there by cond(P_pred) x 1e-16 (measured 4e-9 relative; the median difference is 1e-13)
    for k, tol in (("smoothed_x", 1e-7), ("smoothed_dx", 1e-6), ("smoothed_ddx", 1e-5)):
        assert np.abs(a[k] - b[k]).max() < tol * max(1.0, np.abs(b[k]).max()), k
        assert np.median(np.abs(a[k] - b[k])) < 1e-9
    truth = seq["q_true"][:, ekf.EKF_ORDER]
    # (metres; looser than an exact-Jacobian filter would allow: the reference's float32 forward-difference
    #  perturbation, reproduced since round 2, scales every Jacobian column by up to 1e-4 - measured 0.044 / 0.03)
    assert np.abs(a["x"][200:, :3] - truth[200:, :3]).max() < 0.07
    assert np.abs(a["smoothed_x"][200:, :3] - truth[200:, :3]).max() < 0.05


@pytest.mark.gpu
def test_hip_ekf_matches_reference_ekf_text(gpu_lib, golden_dir):
    """The HIP filter against vectors produced by the reference's own filter + smoother text (ekf_ref.npz)."""
    from acinoset_amd import ekf
    g = _ref(golden_dir)
    for tag in "ab":
        det, rig = g[f"{tag}_det"], (g[f"{tag}_K"], g[f"{tag}_D"], g[f"{tag}_R"], g[f"{tag}_t"])
        s0 = oekf.initial_state(g[f"{tag}_nose_frames"], g[f"{tag}_nose_xyz"], int(g[f"{tag}_start_frame"]), 1 / 120)
        got = ekf.ekf(det, *rig, 120.0, 0.5, (2704, 1520), states0=s0)
        assert got["outliers_ignored"] == int(g[f"{tag}_outliers"])
        est = np.hstack([got["x"], got["dx"], got["ddx"]])
        sm = np.hstack([got["smoothed_x"], got["smoothed_dx"], got["smoothed_ddx"]])
        assert np.abs(est[0] - g[f"{tag}_est"][0]).max() < 1e-8          # before any float32 ulp can have flipped
        for blk, tol in ((slice(0, 25), 5e-6), (slice(25, 50), 5e-5), (slice(50, 75), 1e-3)):
            for have, want in ((est, g[f"{tag}_est"]), (sm, g[f"{tag}_smooth"])):
                scale = max(1.0, np.abs(want[:, blk]).max())
                assert np.abs(have[:, blk] - want[:, blk]).max() < tol * scale, (tag, blk)
