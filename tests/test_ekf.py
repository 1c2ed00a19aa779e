"""EKF + RTS smoother.  CPU: structure of the oracle (oracle/ekf.py, a line-by-line
restatement of src/all_optimizations.py:569-865; parity unpinned - no EKF output ships with the reference).
GPU: the HIP filter against the oracle on seeded synthetic clips."""
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
    assert np.abs(a["x"][200:, :3] - truth[200:, :3]).max() < 0.03            # metres
    assert np.abs(a["smoothed_x"][200:, :3] - truth[200:, :3]).max() < 0.02
