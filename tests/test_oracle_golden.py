"""CPU: the oracle against the golden vectors generated from the reference (tests/golden/make_golden.py)
and the reference's own recorded numbers (KAT-1, KAT-4)."""
import json
import os

import numpy as np
import pytest

from oracle import camera, fk, index_path, loss
from oracle import fte as ofte


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_redescending_loss_table(golden_dir):
    g = _g(golden_dir, "ref_helpers.npz")
    assert np.abs(loss.redescending_loss(g["e"], 3, 10, 20) - g["rho_3_10_20"]).max() < 1e-12
    assert np.abs(loss.redescending_loss(g["e"], 3, 5, 15) - g["rho_3_5_15"]).max() < 1e-12
    # SURVEY 8c spot values of build.redescending_loss(e, 3, 10, 20)
    for e, v in ((2, 1.862311), (5, 10.713318), (15, 36.800639), (40, 40.5), (0, -0.214097)):
        assert abs(loss.redescending_loss(e, 3, 10, 20) - v) < 1e-6


def test_redescending_derivative_and_weight():
    e = np.linspace(0.01, 45, 3000)
    rho, d, h = loss.redescending_dloss(e, 3, 10, 20)
    num = (loss.redescending_loss(e + 1e-6, 3, 10, 20) - loss.redescending_loss(e - 1e-6, 3, 10, 20)) / 2e-6
    assert np.abs(num - d).max() < 1e-6
    assert np.abs(rho - loss.redescending_loss(e, 3, 10, 20)).max() < 1e-13
    assert h.min() >= 0 and h.max() <= 1
    assert abs(loss.drho_at_zero(3, 10, 20) - loss.redescending_dloss(np.array([0.0]), 3, 10, 20)[1][0]) < 1e-15


def _scene(golden_dir):
    sc = json.load(open(os.path.join(golden_dir, "dummy_scene.json")))
    K = np.array([c["k"] for c in sc["cameras"]])
    D = np.array([c["d"] for c in sc["cameras"]]).reshape(-1, 4)
    R = np.array([c["r"] for c in sc["cameras"]])
    t = np.array([c["t"] for c in sc["cameras"]])
    return K, D, R, t


def test_pt3d_to_2d_matches_reference(golden_dir):
    g = _g(golden_dir, "ref_helpers.npz")
    K, D, R, t = _scene(golden_dir)
    uv = np.array([camera.pt3d_to_2d(X, K[c], D[c], R[c], t[c]) for X, c in zip(g["p2d_X"], g["p2d_cam"])])
    assert np.abs(uv - g["p2d_uv"]).max() < 1e-9
    # analytic Jacobian against central differences
    X = g["p2d_X"][3]
    _, J, _ = camera.pt3d_to_2d(X, K[1], D[1], R[1], t[1], with_jac=True)
    Jn = np.zeros((2, 3))
    for j in range(3):
        d = np.zeros(3)
        d[j] = 1e-6
        Jn[:, j] = (camera.pt3d_to_2d(X + d, K[1], D[1], R[1], t[1]) - camera.pt3d_to_2d(X - d, K[1], D[1], R[1], t[1])) / 2e-6
    assert np.abs(J - Jn).max() < 1e-5
    # the closed form equals cv2-style fisheye projection to <= 1e-6 px
    uv2 = np.array([camera.project_points_fisheye(X, K[c], D[c], R[c], t[c])[0] for X, c in zip(g["p2d_X"], g["p2d_cam"])])
    assert np.abs(uv2 - g["p2d_uv"]).max() < 1e-5


def test_rotation_convention(golden_dir):
    g = _g(golden_dir, "ref_helpers.npz")
    for ax, key in (("x", "rot_x"), ("y", "rot_y"), ("z", "rot_z")):
        R, _ = fk._rot(ax, g["rot_ang"])
        assert np.abs(R - g[key]).max() < 1e-15


def test_cheetah_fk_positions_and_jacobian(golden_dir):
    g = _g(golden_dir, "cheetah_fk.npz")
    P, J = fk.cheetah_fk(g["q"], with_jac=True)
    assert np.abs(P - g["positions"]).max() < 1e-13
    assert np.abs(J.reshape(-1, 60, 45) - g["jac"]).max() < 1e-13
    assert (fk.ACTIVE == g["active"]).all()
    dep = (np.abs(J).max(0).reshape(60, 45) > 0)
    assert (dep <= g["dep"].astype(bool)).all()          # never depends on more than the symbolic pattern
    # the 20 states with Q == 0 are exactly the inactive ones and never move a marker
    inactive = np.setdiff1d(np.arange(45), fk.ACTIVE)
    assert set(inactive.tolist()) == set(np.where(fk.Q_SIGMA == 0)[0].tolist())
    assert np.abs(J[..., inactive]).max() == 0
    assert np.isfinite(fk.bounds45()[0]).sum() == 21


def test_index_path_matches_reference(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "index_path.json")))["cases"]

    def fake_tri(a, b, k1, d1, r1, t1, k2, d2, r2, t2):
        a = np.asarray(a, dtype=np.float64).reshape(-1, 2)
        b = np.asarray(b, dtype=np.float64).reshape(-1, 2)
        return np.stack([a[:, 0] + 2.0 * b[:, 0] + k1[0, 0], a[:, 1] - b[:, 1] + 3.0 * k2[0, 0],
                         a[:, 0] * 0.5 + b[:, 1] * 0.25 + k1[0, 0] * k2[0, 0]], axis=1)

    import pandas as pd
    for case in cases:
        det = np.array(case["det"])
        N, C, L = case["N"], case["C"], case["L"]
        k_arr = np.array([np.diag([kd[0], kd[1], 1.0]) for kd in case["kdiag"]])
        z = np.zeros((C, 4))
        r_arr = np.tile(np.eye(3), (C, 1, 1))
        t_arr = np.zeros((C, 3, 1))
        rows = [dict(frame=n, camera=c, marker=case["markers"][l], x=det[n, c, l, 0], y=det[n, c, l, 1],
                     likelihood=det[n, c, l, 2]) for c in range(C) for n in range(N) for l in range(L)]
        df = pd.DataFrame(rows)
        df = df[df["likelihood"] > case["thresh"]]
        if case.get("raises"):
            with pytest.raises(KeyError):
                index_path.get_pairwise_3d_points_from_df(df, k_arr, z, r_arr, t_arr, fake_tri)
            continue
        out = index_path.get_pairwise_3d_points_from_df(df, k_arr, z, r_arr, t_arr, fake_tri)
        assert list(out["frame"]) == case["out_frame"]                    # index path: exact
        assert list(out["marker"]) == case["out_marker"]
        assert out["frame"].dtype == np.float64
        ref = np.array(case["out_xyz"])
        assert np.array_equal(out[["x", "y", "z"]].to_numpy(), ref)       # Kahan mean in pair order: bit-exact


def test_kat1_triangulate_project_statistics(golden_dir):
    """KAT-1: residual statistics recorded in src/calib_with_gui.ipynb cell 29 (before SBA)."""
    g = _g(golden_dir, "kat1_sunday_amelia.npz")
    for row, (tag, ca, cb) in enumerate((("rotating", 1, 2), ("static", 3, 4))):
        K, D, R, t = g[f"{tag}_K"], g[f"{tag}_D"], g[f"{tag}_R"], g[f"{tag}_t"]
        pa, pb = g[f"cam{ca}_points"], g[f"cam{cb}_points"]
        assert (g[f"cam{ca}_fnames"] == g[f"cam{cb}_fnames"]).all()
        res = []
        for f in range(pa.shape[0]):
            p3 = camera.triangulate_points_fisheye(pa[f], pb[f], K[0], D[0], R[0], t[0], K[1], D[1], R[1], t[1])
            p3 = p3.astype(np.float32).astype(np.float64)                 # calib.py:259-260 stores float32
            for ci, pp in ((0, pa), (1, pb)):
                pr = camera.project_points_fisheye(p3, K[ci], D[ci], R[ci], t[ci])
                res.append((pr - pp[f].reshape(-1, 2)).ravel())
        r = np.concatenate(res)
        assert r.size == 3456
        mean, std, cost = g["recorded"][row]
        assert abs(r.std() - std) / std < 1e-6
        assert abs(0.5 * np.sum(np.log1p(r ** 2)) - cost) / cost < 5e-5     # recorded to 5 significant digits
        assert abs(r.mean() - mean) < 1e-5


def test_kat4_integration_identities(golden_dir):
    """KAT-4: the stored IPOPT runs satisfy the backward-Euler relations, hence the third-difference form."""
    g = _g(golden_dir, "kat34_build_runs.npz")
    h = 1.0 / 120
    for tag in ("traj", "run1"):
        x, dx, ddx = g[f"{tag}_x"], g[f"{tag}_dx"], g[f"{tag}_ddx"]
        assert np.abs(x[1:] - x[:-1] - h * dx[1:]).max() < 1e-12
        assert np.abs(dx[1:] - dx[:-1] - h * ddx[1:]).max() < 1e-11
        slack = ddx[1:] - ddx[:-1]                                            # slack_n, n = 2..N (1-based)
        third = (x[3:] - 3 * x[2:-1] + 3 * x[1:-2] - x[:-3]) / h ** 2
        assert np.abs(slack[2:] - third).max() < 1e-6 * max(1.0, np.abs(third).max())
        assert np.abs(slack[:2]).max() < 1e-4 * np.median(np.abs(slack[2:]) + 1e-12) + 1e-4


def test_smoothness_term_equals_third_difference():
    rng = np.random.default_rng(3)
    N = 12
    det = np.zeros((N, 2, 20, 3))
    K = np.tile(np.eye(3), (2, 1, 1))
    prob = ofte.FTEProblem(det[..., :2], det[..., 2], K, np.zeros((2, 4)), K, np.zeros((2, 3)), 1 / 120)
    x = rng.normal(size=(N, 25))
    cost, grad = prob.smooth_terms(x)
    d = x[3:] - 3 * x[2:-1] + 3 * x[1:-2] - x[:-3]
    assert abs(cost - (prob.q_w * d * d).sum()) < 1e-6 * abs(cost)
    band = prob.s_band()
    gb = np.zeros_like(x)
    for n in range(N):
        for k in range(-3, 4):
            if 0 <= n + k < N:
                gb[n] += 2 * prob.q_w * band[abs(k), min(n, n + k)] * x[n + k]
    assert np.abs(gb - grad).max() < 1e-9 * np.abs(grad).max()
    # shard consistency: two halves with halos reproduce the full gradient and cost
    pl = ofte.FTEProblem(det[:6, ..., :2], det[:6, ..., 2], K, np.zeros((2, 4)), K, np.zeros((2, 3)), 1 / 120, n_global=N, n_offset=0)
    pr = ofte.FTEProblem(det[6:, ..., :2], det[6:, ..., 2], K, np.zeros((2, 4)), K, np.zeros((2, 3)), 1 / 120, n_global=N, n_offset=6)
    cl, gl = pl.smooth_terms(x[:6], None, x[6:9])
    cr, gr = pr.smooth_terms(x[6:], x[3:6], None)
    assert abs(cl + cr - cost) < 1e-9 * abs(cost)
    assert np.abs(np.vstack([gl, gr]) - grad).max() < 1e-9 * np.abs(grad).max()
    assert np.allclose(np.hstack([pl.s_band(), pr.s_band()]), band)


# ---------------------------------------------------------------------------------------------------------------
# fte_model.npz: the reference's OWN model text (all_optimizations.py:25-27, 243-252, 268-277, 283-500) executed on
# floats by tests/golden/make_golden.py::gen_fte_model - constants, weights, boxes, initialisation, objective.
def _fte_model_problem(g, **kw):
    s, e = int(g["start_frame"]), int(g["end_frame"])
    det = g["det"][s:e]
    return ofte.FTEProblem(det[..., :2], det[..., 2], g["K"], g["D"], g["R"], g["t"], 1.0 / float(g["fps"]),
                           dlc_thresh=float(g["dlc_thresh"]), **kw)


def test_constants_match_reference_model_text(golden_dir):
    g = _g(golden_dir, "fte_model.npz")
    assert np.array_equal(fk.Q_SIGMA ** 2, g["Q"])                       # :245-252
    assert float(g["R_meas"]) == 5.0 and g["redesc"].tolist() == [3.0, 10.0, 20.0]
    wq = np.where(g["Q"] != 0, 1.0 / np.where(g["Q"] != 0, g["Q"], 1.0), 0.0)
    assert np.array_equal(wq, g["model_err_weight"])                    # :310-315
    lo, hi = fk.bounds45()
    assert np.array_equal(np.isfinite(lo), np.isfinite(g["bounds_lo"])) and len(g["bound_rules"]) == 21
    fin = np.isfinite(lo)
    assert np.abs(lo[fin] - g["bounds_lo"][fin]).max() < 1e-15 and np.abs(hi[fin] - g["bounds_hi"][fin]).max() < 1e-15
    prob = _fte_model_problem(g)
    assert np.array_equal(prob.w, g["meas_err_weight"])                 # :302-308, binary 1/R
    assert np.array_equal(prob.meas, g["meas"])
    assert np.array_equal(prob.q_w, (g["model_err_weight"] / prob.Ts ** 4)[fk.ACTIVE])
    # the inactive states are exactly those the reference gives weight 0
    assert set(np.where(g["model_err_weight"] == 0)[0]) == set(np.setdiff1d(np.arange(45), fk.ACTIVE))


def test_nose_line_init_matches_reference_text(golden_dir):
    g = _g(golden_dir, "fte_model.npz")
    s, e = int(g["start_frame"]), int(g["end_frame"])
    tab = g["nose_table"]                                               # regression over ALL triangulated frames
    x0 = ofte.nose_line_init(tab[:, 0], tab[:, 1:4], e - s, start_frame=s)
    assert np.abs(x0 - g["init_x"]).max() < 1e-12
    assert np.abs(fk.cheetah_fk(x0) - g["init_poses"]).max() < 1e-13   # :351-355


def test_objective_matches_reference_model_text(golden_dir):
    """obj (:486-500) with every equality of :359-399 satisfied == the oracle's reduced objective."""
    g = _g(golden_dir, "fte_model.npz")
    prob = _fte_model_problem(g)
    assert g["case_max_eq_residual"].max() < 1e-9
    for X, want, slack in zip(g["case_x"], g["case_obj"], g["case_slack_model"]):
        cost, _g_, _H, _nb = prob.evaluate(X[:, fk.ACTIVE], need_jac=False)
        assert abs(cost - want) < 1e-10 * abs(want), (cost, want)
        # the eliminated slacks are the scaled third differences; slack_2 = slack_3 = 0 (free dx_1, ddx_1)
        third = (X[3:] - 3 * X[2:-1] + 3 * X[1:-2] - X[:-3]) / prob.Ts ** 2
        assert np.abs(slack[3:] - third).max() < 1e-6 * max(1.0, np.abs(third).max())
        assert np.abs(slack[:3]).max() < 1e-9 * max(1.0, np.abs(third).max())
    # case 4 has the animal behind camera 0: the reference applies no z cut, neither does the oracle
    prob.measurement_terms(g["case_x"][4][:, fk.ACTIVE], need_jac=False)
    assert prob.measurement_terms(g["case_x"][4][:, fk.ACTIVE], need_jac=False)[3] > 0


def _complete_equalities(fp, m, X):
    """Give the model's dependent variables the values its own equalities define at the states X [N, 45] (each equality
    is affine with slope +-1 or +-Ts in the variable it defines), the free dx_1, ddx_1 at their optimum; returns the
    worst equality residual afterwards."""
    N, C, L, P = m._shape

    def solve(comp, idx, var):
        v0 = var.value = 0.0 if var.value is None else var.value
        r0 = comp.rule(m, *idx).r
        var.value = v0 + 1.0
        slope = comp.rule(m, *idx).r - r0
        var.value = v0 - r0 / slope

    for n in range(1, N + 1):
        for p in range(1, P + 1):
            m.x[n, p].value = float(X[n - 1, p - 1])
            m.dx[n, p].value = m.ddx[n, p].value = m.slack_model[n, p].value = 0.0
    for p in range(1, P + 1):
        for n in range(2, N + 1):
            solve(m.integrate_p, (n, p), m.dx[n, p])
        acc = (m.dx[3, p].value - m.dx[2, p].value) / m.Ts
        m.ddx[1, p].value = m.ddx[2, p].value = acc
        m.dx[1, p].value = m.dx[2, p].value - m.Ts * acc
        for n in range(3, N + 1):
            solve(m.integrate_v, (n, p), m.ddx[n, p])
        for n in range(2, N + 1):
            solve(m.constant_acc, (n, p), m.slack_model[n, p])
    for n in range(1, N + 1):
        for l in range(1, L + 1):
            for d in (1, 2, 3):
                solve(m.pose_constraint, (n, l, d), m.poses[n, l, d])
            for c in range(1, C + 1):
                for d in (1, 2):
                    m.slack_meas[n, c, l, d].value = 0.0
                    solve(m.measurement, (n, c, l, d), m.slack_meas[n, c, l, d])
    worst = 0.0
    for name in ("pose_constraint", "integrate_p", "integrate_v", "constant_acc", "measurement"):
        for v in getattr(m, name).evaluate(m).values():
            if v is not fp.Constraint.Skip:
                worst = max(worst, abs(v.r))
    return worst


def test_own_pyomo_formulation_reproduces_the_reference_model_text(golden_dir):
    """oracle/pyomo_model.py - the formulation bench.py times through IPOPT where Pyomo exists - built against the float
    stand-ins: same variable blocks and equality counts as the reference's model, and at the five recorded iterates its
    objective, slacks and derivatives equal what the reference's own model text gave (fte_model.npz)."""
    import sys
    sys.path.insert(0, golden_dir)
    import _float_pyomo as fp
    from oracle import pyomo_model
    g = _g(golden_dir, "fte_model.npz")
    m = pyomo_model.build_fte_model(fp, g["meas"], g["det"][int(g["start_frame"]):int(g["end_frame"]), ..., 2], g["K"], g["D"],
                                    g["R"], g["t"], 1.0 / float(g["fps"]), g["init_x"], dlc_thresh=float(g["dlc_thresh"]),
                                    r_meas=float(g["R_meas"]), Q=g["Q"], redesc=tuple(g["redesc"]))
    N, C, L, P = m._shape
    assert (N, C, L, P) == (6, 3, 20, 45)
    n_eq = {k: sum(v is not fp.Constraint.Skip for v in getattr(m, k).data.values())
            for k in ("pose_constraint", "integrate_p", "integrate_v", "constant_acc", "measurement")}
    assert n_eq == dict(pose_constraint=N * L * 3, integrate_p=(N - 1) * P, integrate_v=(N - 1) * P,
                        constant_acc=(N - 1) * P, measurement=N * C * L * 2)
    # start values (:333-355): x = init_x, poses = FK(init_x), everything else 0
    assert max(abs(m.poses[n, l, d].value - g["init_poses"][n - 1, l - 1, d - 1])
               for n in m.N for l in m.L for d in m.D3) < 1e-13
    for X, want, slack, dx, ddx in zip(g["case_x"], g["case_obj"], g["case_slack_model"], g["case_dx"], g["case_ddx"]):
        assert _complete_equalities(fp, m, X) < 1e-9
        got = m.obj.value(m)
        assert abs(got - want) < 1e-10 * abs(want), (got, want)
        val = lambda v: np.array([[v[n, p].value for p in m.P] for n in m.N])
        assert np.abs(val(m.slack_model) - slack).max() < 1e-9 * max(1.0, np.abs(slack).max())
        assert np.abs(val(m.dx) - dx).max() < 1e-9 * max(1.0, np.abs(dx).max())
        assert np.abs(val(m.ddx) - ddx).max() < 1e-9 * max(1.0, np.abs(ddx).max())


def test_lm_fixed_point_is_stationary_for_the_reference_objective(golden_dir):
    """Row a-10 (the IPOPT call): IPOPT itself cannot run here, so its end state stays unpinned - but the point the
    projected LM converges to is pinned as a first-order stationary point of the REFERENCE's own objective.
    fte_stationary.npz (make_golden.py::gen_fte_stationary) holds central differences of the reference's model text
    (all_optimizations.py:283-500 on floats, every equality constraint satisfied) in all N x 25 active states, at the
    reference's initial point and at the LM end point x*."""
    g, st = _g(golden_dir, "fte_model.npz"), _g(golden_dir, "fte_stationary.npz")
    prob = _fte_model_problem(g)
    act = fk.ACTIVE
    x0 = g["init_x"][:, act]
    xs, info = ofte.lm_solve(prob, x0, max_iter=300, ftol=1e-15, xtol=1e-13, gtol=1e-9)
    assert np.abs(xs - st["x_star"][:, act]).max() < 1e-9, info
    cost, grad, _H, _nb = prob.evaluate(xs)
    assert abs(cost - float(st["obj_ref_star"])) < 1e-10 * abs(cost)
    assert float(st["obj_ref_star"]) < float(st["obj_ref_init"]) and float(st["max_eq_residual"]) < 1e-9
    # the oracle's analytic gradient IS the gradient of the reference objective (finite differences of the model text)
    _c0, g0, _H0, _nb0 = prob.evaluate(x0)
    scale = np.abs(st["grad_ref_init"]).max()
    assert np.abs(g0 - st["grad_ref_init"]).max() < 1e-6 * scale
    assert np.abs(grad - st["grad_ref_star"]).max() < 1e-6 * scale
    # first-order optimality of the reference objective over the 21 boxes, at x*
    gr = st["grad_ref_star"]
    active = ((xs <= prob.lo) & (gr > 0)) | ((xs >= prob.hi) & (gr < 0))
    assert np.abs(np.where(active, 0.0, gr)).max() < 1e-6 * scale


def test_third_party_quasi_newton_ends_beside_the_lm_fixed_point(golden_dir):
    """Row a-10, independent evidence: the reference solves its NLP with IPOPT's limited-memory quasi-Newton
    (all_optimizations.py:503-522); here scipy's L-BFGS-B - a third-party quasi-Newton code with box bounds - minimises the
    same objective over the 21 boxes from the reference's init_x (make_golden.py::gen_fte_lbfgs; its end point's objective
    is evaluated by the reference's own model text).  What it shows, stated as it came out: after 50 000 iterations in the
    flat valley it sits 4e-6 (relative) BELOW the LM fixed point x*, 2.3 mm away in the worst marker (median 0.08 mm) and
    with a projected gradient still 2e-4 of the start's - two nearby points of one valley of a non-convex objective
    (20 % gross outliers, redescending loss), neither of them the IPOPT end state (tol = 1e-1 stops far earlier)."""
    g, st, lb = _g(golden_dir, "fte_model.npz"), _g(golden_dir, "fte_stationary.npz"), _g(golden_dir, "fte_lbfgs.npz")
    prob = _fte_model_problem(g)
    act = fk.ACTIVE
    xl = lb["x_lbfgs"][:, act]
    assert np.all(xl >= prob.lo - 1e-15) and np.all(xl <= prob.hi + 1e-15)
    cost_l, _g2, _H, _nb = prob.evaluate(xl)
    # the oracle's objective at the L-BFGS-B end point IS the reference text's objective there
    assert abs(cost_l - float(lb["obj_ref_lbfgs"])) < 1e-10 * abs(cost_l) and float(lb["max_eq_residual"]) < 1e-9
    cost_s = float(st["obj_ref_star"])
    assert abs(cost_s - cost_l) < 1e-5 * abs(cost_l), (cost_s, cost_l)
    d = np.linalg.norm(fk.cheetah_fk(lb["x_lbfgs"]) - fk.cheetah_fk(st["x_star"]), axis=-1)
    assert d.max() < 3e-3 and np.median(d) < 2e-4, (d.max(), np.median(d))
    assert float(lb["proj_grad_inf"]) < 1e-3 * np.abs(st["grad_ref_init"]).max()
