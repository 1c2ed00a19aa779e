"""CPU: the oracle FTE (cost / gradient / Gauss-Newton blocks / LM) is self-consistent and converges."""
import numpy as np

from oracle import camera, fk, index_path, synth
from oracle import fte as ofte


def _problem(n=24, kind="sprint"):
    s = synth.make_sequence(n, kind)
    prob = ofte.FTEProblem(s["det"][..., :2], s["det"][..., 2], s["K"], s["D"], s["R"], s["t"], s["Ts"])
    return s, prob


def test_gradient_matches_finite_differences():
    s, prob = _problem(12)
    rng = np.random.default_rng(0)
    x = s["q_true"][:, fk.ACTIVE] + rng.normal(0, 0.01, (12, 25))
    F, g, H, _ = prob.evaluate(x)
    for _ in range(4):
        d = rng.normal(size=x.shape)
        eps = 1e-6
        num = (prob.evaluate(x + eps * d, need_jac=False)[0] - prob.evaluate(x - eps * d, need_jac=False)[0]) / (2 * eps)
        assert abs(num - (g * d).sum()) < 1e-6 * abs(num)
    assert np.abs(H - H.transpose(0, 2, 1)).max() < 1e-9 * np.abs(H).max()
    assert np.linalg.eigvalsh(H).min() > -1e-6 * np.abs(H).max()          # Gauss-Newton blocks are PSD


def test_banded_solve_against_dense():
    s, prob = _problem(9)
    rng = np.random.default_rng(1)
    x = np.clip(s["q_true"][:, fk.ACTIVE] + rng.normal(0, 0.05, (9, 25)), prob.lo, prob.hi)
    F, g, H, _ = prob.evaluate(x)
    fixed = ((x <= prob.lo) & (g > 0)) | ((x >= prob.hi) & (g < 0))
    lam = 1e-2
    delta, diag = prob.solve_banded(H, g, lam, fixed)
    N, P = g.shape
    A = np.zeros((N * P, N * P))
    band = prob.s_band()
    for n in range(N):
        A[n * P:(n + 1) * P, n * P:(n + 1) * P] = H[n]
        for k in range(4):
            if n + k < N:
                for p in range(P):
                    v = 2 * prob.q_w[p] * band[k, n]
                    A[n * P + p, (n + k) * P + p] += v
                    if k:
                        A[(n + k) * P + p, n * P + p] += v
    dg = np.diag(A).copy()
    assert np.allclose(dg.reshape(N, P), diag)
    A[np.arange(N * P), np.arange(N * P)] += lam * dg
    free = ~fixed.reshape(-1)
    sol = np.zeros(N * P)
    sol[free] = np.linalg.solve(A[np.ix_(free, free)], -g.reshape(-1)[free])
    assert np.abs(sol - delta.reshape(-1)).max() < 1e-8 * max(1.0, np.abs(sol).max())


def test_lm_converges_from_nose_line_init():
    s, prob = _problem(30)
    det = s["det"]
    tri, cnt, _ = index_path.pairwise_dense(det, 0.5, s["K"], s["D"], s["R"], s["t"], camera.triangulate_points_fisheye)
    nose = tri[:, 2]
    ok = np.isfinite(nose[:, 0])
    x0 = ofte.nose_line_init(np.arange(30)[ok], nose[ok], 30)
    hist = []
    x, info = ofte.lm_solve(prob, x0[:, fk.ACTIVE], max_iter=40, history=hist)
    assert info["status"] in ("ftol", "xtol", "gtol")
    costs = [h["F"] for h in hist]
    assert all(b <= a + 1e-9 for a, b in zip(costs, costs[1:]))            # monotone in accepted cost
    out = ofte.fte_outputs(prob, x, x0)
    err = np.linalg.norm(out["positions"] - s["pos_true"], axis=-1)
    assert np.median(err) < 0.01 and err.max() < 0.06
    assert (x >= prob.lo - 1e-12).all() and (x <= prob.hi + 1e-12).all()
    # reference output conventions (all_optimizations.py:530-559)
    assert out["x"].shape == (30, 25) and out["positions"].shape == (30, 20, 3)
    Ts = s["Ts"]
    assert np.allclose(out["x"][1:], out["x"][:-1] + Ts * out["dx"][1:])
    assert np.allclose(out["dx"][1:], out["dx"][:-1] + Ts * out["ddx"][1:], atol=1e-6 * np.abs(out["dx"]).max())


def test_active_set_ignores_gradient_noise_and_zero_diagonals_are_damped():
    """The two rules round 2 added to the LM (GPU and oracle alike): (1) a variable on its bound is pinned only if its
    gradient entry pushes outward by more than 1e-14 x H_ii - an analytically vanishing entry must not be decided by
    rounding noise; (2) a diagonal entry that is exactly 0 (three frames, unobserved joint) is damped by lam x 1e-30: the
    banded Cholesky goes through and the variable does not move."""
    s, prob = _problem(12)
    x = np.clip(s["q_true"][:, fk.ACTIVE], prob.lo, prob.hi)
    F, g, H, _ = prob.evaluate(x)
    p = int(np.where(np.isfinite(prob.hi))[0][0])
    x[5, p] = prob.hi[p]
    diag = H[5, p, p] + 2 * prob.q_w[p] * prob.s_band()[0][5]
    for gval, want in ((-1.0, True), (-1e-20, False), (0.0, False), (-0.5e-14 * diag, False), (-2e-14 * diag, True), (+1.0, False)):
        g2 = g.copy()
        g2[5, p] = gval
        assert bool(prob.active_set(x, g2, H)[5, p]) == want, gval
    # (2) three frames, the tail markers never detected
    s3 = synth.make_sequence(3, "trot")
    d3 = s3["det"].copy()
    d3[:, :, 6:8, 2] = 0.0
    p3 = ofte.FTEProblem(d3[..., :2], d3[..., 2], s3["K"], s3["D"], s3["R"], s3["t"], s3["Ts"])
    x3 = np.clip(s3["q_true"][:, fk.ACTIVE] + np.random.default_rng(3).normal(0, 0.03, (3, 25)), p3.lo, p3.hi)
    H3 = p3.evaluate(x3)[2]
    dead = H3[:, np.arange(25), np.arange(25)] == 0.0
    assert dead.any()
    xo, info = ofte.lm_solve(p3, x3, max_iter=40)
    assert info["status"] in ("ftol", "xtol", "gtol") and np.array_equal(xo[dead], x3[dead])
