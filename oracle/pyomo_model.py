"""This repo's own Pyomo formulation of the reference's FTE NLP (test / baseline infrastructure, see oracle/__init__.py).

Specification: src/all_optimizations.py:283-522 as summarised in SURVEY.md section 8 row a-7 - variables x, dx, ddx,
slack_model [N, 45], poses [N, 20, 3], slack_meas [N, C, 20, 2]; equalities poses = FK(x_n), backward-Euler integration,
constant acceleration up to slack_model, pi_c(poses) - meas - slack_meas = 0; the 21 boxes; objective
sum w_p slack_model^2 + sum rho(w_ncl slack_meas); solver options of :503-522 (tol 1e-1, limited-memory Hessian).  Written
from that description, NOT copied: the model is built by ``build_fte_model(env, ...)`` against any namespace ``env`` that
offers ConcreteModel / RangeSet / Param / Var / Constraint / Objective / sin / cos / atan / exp, i.e.

* ``pyomo.environ`` where Pyomo is installed - ``solve_with_ipopt`` then times the reference CPU path (bench.py's probe
  calls it when ``pyomo`` and an ``ipopt`` binary are found; neither exists in the build container or on the GPU boxes);
* ``tests/golden/_float_pyomo.py`` - float-valued stand-ins - which is how the formulation is VALIDATED here: with every
  equality solved for the variable it defines, its objective reproduces the recorded objective of the reference's own
  model text (tests/golden/fte_model.npz) at the five recorded iterates (tests/test_oracle_golden.py).
"""
import math

import numpy as np

from . import fk

N_STATES, N_MARKERS = fk.N_STATES, 20


def _rot_entries(axis, c, s):
    """The reference's rot_x / rot_y / rot_z (:66-91: transposes of the usual active rotations) as nested lists."""
    if axis == "x":
        return [[1.0, 0.0, 0.0], [0.0, c, s], [0.0, -s, c]]
    if axis == "y":
        return [[c, 0.0, -s], [0.0, 1.0, 0.0], [s, 0.0, c]]
    return [[c, s, 0.0], [-s, c, 0.0], [0.0, 0.0, 1.0]]


def _matmul(A, B):
    return [[sum(A[i][k] * B[k][j] for k in range(3)) for j in range(3)] for i in range(3)]


def fk_expressions(q, sin, cos):
    """Marker positions [20][3] as expressions of the 45 state entries q (any scalar type with + - *), through the
    kinematic chain tables of oracle/fk.py (:101-165 of the reference): RI_k = (product of elementary rotations) RI_parent,
    p_child = p_parent + RI_k^T offset."""
    RI = {}
    for k, parent, rots in fk._CHAIN:
        E = None
        for axis, idx in rots:
            R = _rot_entries(axis, cos(q[idx]), sin(q[idx]))
            E = R if E is None else _matmul(E, R)
        RI[k] = E if parent is None else _matmul(E, RI[parent])
    pos, out = {}, []
    for name, parent, k, off in fk._LINKS:
        base = [q[0], q[1], q[2]] if parent is None else pos[parent]
        R = RI[k]
        pos[name] = [base[i] + sum(R[j][i] * off[j] for j in range(3)) for i in range(3)]      # RI_k^T offset
        out.append(pos[name])
    return out


def project_expr(X, K, D, R, t, atan):
    """pt3d_to_2d (:193-209): the reference's closed-form fisheye projection, +1e-12 under the root, no z cut."""
    xc = [sum(R[i][j] * X[j] for j in range(3)) + t[i] for i in range(3)]
    a, b = xc[0] / xc[2], xc[1] / xc[2]
    r = (a * a + b * b + 1e-12) ** 0.5
    th = atan(r)
    thd = th * (1 + D[0] * th ** 2 + D[1] * th ** 4 + D[2] * th ** 6 + D[3] * th ** 8)
    return K[0][0] * a * thd / r + K[0][2], K[1][1] * b * thd / r + K[1][2]


def redescending_expr(err, a, b, c, exp):
    """build.py:382-395: logistic blend of quadratic / linear / redescending / constant pieces (unit-slope steps)."""
    e = abs(err)
    step = lambda s: 1.0 / (1.0 + exp(-(e - s)))
    sa, sb, sc = step(a), step(b), step(c)
    return ((1 - sa) / 2 * e ** 2 + (sa - sb) * (a * e - a ** 2 / 2) +
            (sb - sc) * (a * b - a ** 2 / 2 + (a * (c - b) / 2) * (1 - ((c - e) / (c - b)) ** 2)) +
            sc * (a * b - a ** 2 / 2 + a * (c - b) / 2))


def build_fte_model(env, meas, likelihood, K_arr, D_arr, R_arr, t_arr, Ts, x0, dlc_thresh=0.5, r_meas=5.0, Q=None,
                    redesc=(3.0, 10.0, 20.0), use_bounds=True):
    """The FTE NLP as a ConcreteModel of ``env``.  meas[N, C, 20, 2], likelihood[N, C, 20], x0[N, 45] initial states.
    Index sets are 1-based RangeSets like the reference's (n, p, l, c, d)."""
    meas = np.asarray(meas, dtype=np.float64)
    lik = np.asarray(likelihood, dtype=np.float64)
    N, C, L = meas.shape[0], meas.shape[1], meas.shape[2]
    P = N_STATES
    Qs = (fk.Q_SIGMA ** 2 if Q is None else np.asarray(Q, dtype=np.float64))
    w_model = np.where(Qs > 0, 1.0 / np.where(Qs > 0, Qs, 1.0), 0.0)           # (:310-315) 1/Q_p, 0 where Q_p = 0
    w_meas = np.where(lik > dlc_thresh, 1.0 / r_meas, 0.0)                      # (:302-308) binary in the likelihood
    lo, hi = fk.bounds45()
    exp = getattr(env, "exp", math.exp)
    m = env.ConcreteModel(name="acinoset_fte")
    m.N, m.P, m.L, m.C = env.RangeSet(N), env.RangeSet(P), env.RangeSet(L), env.RangeSet(C)
    m.D2, m.D3 = env.RangeSet(2), env.RangeSet(3)
    m.Ts = float(Ts)
    m.x = env.Var(m.N, m.P)
    m.dx = env.Var(m.N, m.P)
    m.ddx = env.Var(m.N, m.P)
    m.slack_model = env.Var(m.N, m.P)
    m.poses = env.Var(m.N, m.L, m.D3)
    m.slack_meas = env.Var(m.N, m.C, m.L, m.D2, initialize=0.0)
    x0 = np.asarray(x0, dtype=np.float64)
    for n in range(1, N + 1):                                                  # (:333-355) start values
        q0 = list(x0[n - 1])
        pos0 = fk_expressions(q0, math.sin, math.cos)
        for p in range(1, P + 1):
            m.x[n, p].value = float(x0[n - 1, p - 1])
            m.dx[n, p].value = 0.0
            m.ddx[n, p].value = 0.0
            m.slack_model[n, p].value = 0.0
            if use_bounds and hasattr(m.x[n, p], "setlb"):                      # the 21 boxes (:403-483) as variable bounds
                if np.isfinite(lo[p - 1]):
                    m.x[n, p].setlb(float(lo[p - 1]))
                if np.isfinite(hi[p - 1]):
                    m.x[n, p].setub(float(hi[p - 1]))
        for l in range(1, L + 1):
            for d in range(1, 4):
                m.poses[n, l, d].value = float(pos0[l - 1][d - 1])

    fk_of = {}                                                                 # FK(x_n) built once per frame, not per (l, d)

    def pose_rule(mm, n, l, d):                                                # (:359-365) poses = FK(x_n)
        if fk_of.get("n") != n or (l, d) == (1, 1):
            fk_of["n"], fk_of["pos"] = n, fk_expressions([mm.x[n, p] for p in range(1, P + 1)], env.sin, env.cos)
        return fk_of["pos"][l - 1][d - 1] - mm.poses[n, l, d] == 0.0

    def integrate_p(mm, n, p):                                                 # (:369-376) x_n = x_{n-1} + Ts dx_n
        return (mm.x[n, p] - mm.x[n - 1, p] - mm.Ts * mm.dx[n, p] == 0.0) if n > 1 else env.Constraint.Skip

    def integrate_v(mm, n, p):                                                 # (:377-383) dx_n = dx_{n-1} + Ts ddx_n
        return (mm.dx[n, p] - mm.dx[n - 1, p] - mm.Ts * mm.ddx[n, p] == 0.0) if n > 1 else env.Constraint.Skip

    def constant_acc(mm, n, p):                                                # (:386-391) ddx_n = ddx_{n-1} + slack_n
        return (mm.ddx[n, p] - mm.ddx[n - 1, p] - mm.slack_model[n, p] == 0.0) if n > 1 else env.Constraint.Skip

    cams = [(np.asarray(K_arr[c]).tolist(), np.asarray(D_arr[c]).reshape(-1).tolist(), np.asarray(R_arr[c]).tolist(),
             np.asarray(t_arr[c]).reshape(-1).tolist()) for c in range(C)]

    def measurement(mm, n, c, l, d):                                           # (:394-399) pi_c(poses) - meas - slack = 0
        X = [mm.poses[n, l, 1], mm.poses[n, l, 2], mm.poses[n, l, 3]]
        uv = project_expr(X, *cams[c - 1], env.atan)
        return uv[d - 1] - float(meas[n - 1, c - 1, l - 1, d - 1]) - mm.slack_meas[n, c, l, d] == 0.0

    m.pose_constraint = env.Constraint(m.N, m.L, m.D3, rule=pose_rule)
    m.integrate_p = env.Constraint(m.N, m.P, rule=integrate_p)
    m.integrate_v = env.Constraint(m.N, m.P, rule=integrate_v)
    m.constant_acc = env.Constraint(m.N, m.P, rule=constant_acc)
    m.measurement = env.Constraint(m.N, m.C, m.L, m.D2, rule=measurement)

    def objective(mm):                                                         # (:486-500)
        total = 0.0
        for n in range(1, N + 1):
            for p in range(1, P + 1):
                if w_model[p - 1] != 0.0:
                    total = total + float(w_model[p - 1]) * mm.slack_model[n, p] ** 2
            for c in range(1, C + 1):
                for l in range(1, L + 1):
                    w = float(w_meas[n - 1, c - 1, l - 1])
                    for d in (1, 2):
                        total = total + redescending_expr(w * mm.slack_meas[n, c, l, d], *redesc, exp)
        return total

    m.obj = env.Objective(rule=objective)
    m._shape = (N, C, L, P)
    return m


def solve_with_ipopt(m, time_limit=3600):
    """opt.solve with the reference's options (:503-522).  Needs pyomo and an ipopt binary; returns (results, seconds)."""
    import time
    from pyomo.opt import SolverFactory
    opt = SolverFactory("ipopt")
    opt.options.update({"print_level": 5, "max_iter": 10000, "max_cpu_time": time_limit, "tol": 1e-1,
                        "hessian_approximation": "limited-memory"})
    t0 = time.perf_counter()
    res = opt.solve(m, tee=False)
    return res, time.perf_counter() - t0


def time_reference_cpu_path(det, rig, Ts, x0_full, n_frames=100):
    """bench.py's conditional baseline: build + IPOPT solve of the first n_frames frames through this formulation.
    Only callable where ``import pyomo.environ`` works and ``ipopt`` is on the PATH."""
    import time
    import pyomo.environ as pyo
    n = min(n_frames, det.shape[0])
    t0 = time.perf_counter()
    m = build_fte_model(pyo, det[:n, ..., :2], det[:n, ..., 2], *rig, Ts, x0_full[:n])
    t_build = time.perf_counter() - t0
    res, t_solve = solve_with_ipopt(m)
    return dict(frames=n, build_seconds=t_build, solve_seconds=t_solve, frames_per_s=n / (t_build + t_solve),
                termination=str(res.solver.termination_condition))
