#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference tree.

Runs ONLY in the build container (needs /root/reference, read-only). Nothing
here travels to the GPU box except the small .npz/.json DATA files it writes.
The reference's Python cannot be imported as-is (cv2, nptyping, pyomo, PyQt5
and its own `lib` package are absent), so this script uses the two routes of
SURVEY.md section 8c:

  * stub-import : inject empty stand-in modules for the missing third-party
    imports, then import the reference module and call its pure-Python
    helpers (redescending_loss, pt3d_to_2d, rot_*, get_pairwise_3d_points_from_df).
  * slice-exec  : exec lines 64-190 of src/all_optimizations.py (the sympy
    cheetah FK) in a namespace {sp, np, sin, cos} to obtain pose_to_3d.

Outputs (all DATA: inputs + expected outputs):
  ref_helpers.npz     redescending_loss table, pt3d_to_2d vectors, rot_x/y/z
  cheetah_fk.npz      q[64,45] -> positions[64,20,3], dpos/dq[64,60,45] (sympy)
  index_path.json     synthetic long DataFrames -> get_pairwise_3d_points_from_df rows
  kat1_sunday_amelia.npz   scene (K,D,R,t) + checkerboard points of cams 1-4 (KAT-1 inputs)
  kat34_build_runs.npz     stored IPOPT runs x,dx,ddx,positions + skeletons (KAT-3/4)
  dummy_scene.json    the 6-camera rig (configs/dummy_scene.json is a data file)
  dlc_tables.json     synthetic wide DLC tables -> utils.create_dlc_points_2d_file long table (read_hdf patched)
  skel_fte_model.npz  build.py's skeleton-driven FTE model text on floats, on the shipped human skeleton / scene / DLC tables

Usage:  python tests/golden/make_golden.py
"""
import json
import os
import pickle
import sys
import textwrap
import types
import warnings

import numpy as np

REF = os.environ.get("ACINO_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    import math

    class _Anything:
        def __getitem__(self, k):
            return self

        def __call__(self, *a, **k):
            return self

    _stub("cv2")
    _stub("nptyping", Array=_Anything())
    _stub("PyQt5")
    _stub("PyQt5.QtWidgets", QApplication=object)
    _stub("PyQt5.QtGui")
    _stub("PyQt5.QtCore")
    _stub("pyqtgraph", mkColor=lambda *a, **k: None, setConfigOption=lambda *a, **k: None,
          setConfigOptions=lambda *a, **k: None)
    _stub("pyqtgraph.opengl", GLViewWidget=object, GLGridItem=object,
          GLLinePlotItem=object, GLScatterPlotItem=object, GLMeshItem=object, MeshData=object)
    _stub("pyomo")
    _stub("pyomo.core")
    _stub("pyomo.core.base")
    _stub("pyomo.core.base.constraint", Constraint=object, ConstraintList=object)
    _stub("pyomo.core.base.PyomoModel", ConcreteModel=object)
    _stub("pyomo.opt", SolverFactory=object, SolverStatus=object, TerminationCondition=object)
    _stub("pyomo.environ", sin=math.sin, cos=math.cos, atan=math.atan)
    # numpy >= 1.24 dropped these aliases; calib.py:261-262,300-301,409-410 use them
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "int"):
        np.int = int


def gen_ref_helpers():
    sys.path.insert(0, os.path.join(REF, "src"))
    import build as ref_build  # noqa: E402  (reference src/build.py, stubbed)

    e = np.concatenate([np.linspace(-45, 45, 361), [0.0, 1e-9, -1e-9, 2, 5, 15, 40, 3, 10, 20]])
    rho = np.array([ref_build.redescending_loss(float(v), 3, 10, 20) for v in e])
    rho2 = np.array([ref_build.redescending_loss(float(v), 3, 5, 15) for v in e])

    rng = np.random.default_rng(7)
    scene = json.load(open(os.path.join(REF, "configs", "dummy_scene.json")))
    cams = scene["cameras"]
    pts, uv, cam_idx = [], [], []
    for i in range(200):
        c = int(rng.integers(0, len(cams)))
        K = np.array(cams[c]["k"]); D = np.array(cams[c]["d"]).reshape(-1)
        R = np.array(cams[c]["r"]); t = np.array(cams[c]["t"])
        X = np.array([2.0, 6.5, 0.7]) + rng.normal(0, 2.0, 3)
        u, v = ref_build.pt3d_to_2d(X[0], X[1], X[2], K, D, R, t)
        pts.append(X); uv.append([u, v]); cam_idx.append(c)
    ang = rng.uniform(-np.pi, np.pi, 16)
    rx = np.array([ref_build.np_rot_x(a) for a in ang])
    ry = np.array([ref_build.np_rot_y(a) for a in ang])
    rz = np.array([ref_build.np_rot_z(a) for a in ang])
    np.savez(os.path.join(OUT, "ref_helpers.npz"),
             e=e, rho_3_10_20=rho, rho_3_5_15=rho2,
             p2d_X=np.array(pts), p2d_uv=np.array(uv), p2d_cam=np.array(cam_idx),
             rot_ang=ang, rot_x=rx, rot_y=ry, rot_z=rz)
    print("ref_helpers.npz: rho(2,5,15,40,0) =", rho[-7:-2] if False else
          [ref_build.redescending_loss(v, 3, 10, 20) for v in (2, 5, 15, 40, 0)])


def gen_cheetah_fk():
    import sympy as sp
    src = open(os.path.join(REF, "src", "all_optimizations.py")).read().splitlines()
    block = textwrap.dedent("\n".join(src[63:190]))  # lines 64..190 (1-based)
    ns = {"sp": sp, "np": np, "sin": np.sin, "cos": np.cos}
    exec(compile(block, "all_optimizations.py[64:190]", "exec"), ns)
    pose_to_3d, positions, sym_list = ns["pose_to_3d"], ns["positions"], ns["sym_list"]
    assert len(sym_list) == 45 and positions.shape == (20, 3)
    flat = sp.Matrix([positions[i, j] for i in range(20) for j in range(3)])
    jac = flat.jacobian(sp.Matrix(sym_list))
    jac_f = sp.lambdify(sym_list, jac, modules="numpy")
    dep = np.array([[1 if jac[r, c] != 0 else 0 for c in range(45)] for r in range(60)], dtype=np.uint8)

    rng = np.random.default_rng(11)
    active = [0, 1, 2, 3, 4, 6] + list(range(17, 31)) + [31, 32, 34, 35, 36]
    Q = np.zeros((64, 45))
    Q[:, 0:3] = rng.uniform(-5, 10, (64, 3))
    for a in active[3:]:
        Q[:, a] = rng.uniform(-np.pi, np.pi, 64)
    Q[0, 3:] = 0.0  # zero pose
    Q[1, :] = 0.0
    P = np.array([np.asarray(pose_to_3d(*q), dtype=np.float64) for q in Q])
    J = np.array([np.asarray(jac_f(*q), dtype=np.float64) for q in Q])
    np.savez_compressed(os.path.join(OUT, "cheetah_fk.npz"), q=Q, positions=P, jac=J, dep=dep,
                        active=np.array(active))
    print("cheetah_fk.npz:", P.shape, J.shape, "deps/marker", dep.reshape(20, 3, 45).max(1).sum(1).tolist())


def gen_index_path():
    import pandas as pd
    sys.path.insert(0, os.path.join(REF, "src"))
    from calib import calib as ref_calib  # noqa: E402

    def fake_tri(a, b, k1, d1, r1, t1, k2, d2, r2, t2):
        # deterministic, pair-identifying stand-in (cv2 is absent): exercises only the index path
        a = np.asarray(a, dtype=np.float64).reshape(-1, 2)
        b = np.asarray(b, dtype=np.float64).reshape(-1, 2)
        return np.stack([a[:, 0] + 2.0 * b[:, 0] + k1[0, 0],
                         a[:, 1] - b[:, 1] + 3.0 * k2[0, 0],
                         a[:, 0] * 0.5 + b[:, 1] * 0.25 + k1[0, 0] * k2[0, 0]], axis=1)

    markers = ["l_eye", "r_eye", "nose", "neck_base", "spine", "tail_base", "tail1", "tail2",
               "l_shoulder", "l_front_knee", "l_front_ankle", "r_shoulder", "r_front_knee",
               "r_front_ankle", "l_hip", "l_back_knee", "l_back_ankle", "r_hip", "r_back_knee",
               "r_back_ankle"]
    cases = []
    for seed, (N, C, L, thr, p_out) in enumerate([(4, 6, 20, 0.5, 0.4), (7, 6, 20, 0.5, 0.7),
                                                  (3, 4, 5, 0.8, 0.5), (5, 2, 20, 0.5, 0.3),
                                                  (6, 6, 20, 0.5, 0.85), (3, 6, 20, 0.5, 0.999)]):
        rng = np.random.default_rng(100 + seed)
        det = np.zeros((N, C, L, 3))
        det[..., 0] = rng.uniform(0, 2704, (N, C, L))
        det[..., 1] = rng.uniform(0, 1520, (N, C, L))
        lik = rng.uniform(thr + 0.01, 1.0, (N, C, L))
        lo = rng.uniform(0, thr - 0.01, (N, C, L))
        det[..., 2] = np.where(rng.uniform(size=(N, C, L)) < p_out, lo, lik)
        det[0, 0, 0, 2] = 0.99  # keep DataFrame row label 0 (calib.py:398 reads ['frame'][0])
        rows = [dict(frame=n, camera=c, marker=markers[l], x=det[n, c, l, 0], y=det[n, c, l, 1],
                     likelihood=det[n, c, l, 2])
                for c in range(C) for n in range(N) for l in range(L)]
        df = pd.DataFrame(rows, columns=["frame", "camera", "marker", "x", "y", "likelihood"])
        k_arr = np.array([np.diag([1000.0 + 10 * c, 1001.0 + 7 * c, 1.0]) for c in range(C)])
        z = np.zeros((C, 4)); r_arr = np.tile(np.eye(3), (C, 1, 1)); t_arr = np.zeros((C, 3, 1))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            import io, contextlib
            with contextlib.redirect_stdout(io.StringIO()):
                try:
                    out = ref_calib.get_pairwise_3d_points_from_df(df[df["likelihood"] > thr].copy(),
                                                                   k_arr, z, r_arr, t_arr, fake_tri)
                except KeyError as exc:  # reference behaviour when NO adjacent pair exists (calib.py:422)
                    cases.append(dict(N=N, C=C, L=L, thresh=thr, markers=markers[:L], det=det.tolist(),
                                      kdiag=[[float(k[0, 0]), float(k[1, 1])] for k in k_arr],
                                      raises="KeyError"))
                    print("index_path case", seed, "raises KeyError", exc)
                    continue
        cases.append(dict(N=N, C=C, L=L, thresh=thr, markers=markers[:L], det=det.tolist(),
                          kdiag=[[float(k[0, 0]), float(k[1, 1])] for k in k_arr],
                          out_frame=[float(v) for v in out["frame"]],
                          out_marker=list(out["marker"]),
                          out_xyz=out[["x", "y", "z"]].astype(float).values.tolist()))
        print("index_path case", seed, "rows", len(out))
    json.dump(dict(cases=cases), open(os.path.join(OUT, "index_path.json"), "w"))


def _ref_lines(lo, hi):
    """Lines lo..hi (1-based, inclusive) of the reference's src/all_optimizations.py, dedented."""
    src = open(os.path.join(REF, "src", "all_optimizations.py")).read().splitlines()
    return textwrap.dedent("\n".join(src[lo - 1:hi]))


def _fte_model_setup():
    """Shared by gen_fte_model / gen_fte_stationary: inputs, slice-exec of the model text, constants, boxes.

    The reference's OWN FTE model text on floats (fte_model.npz).

    Slice-exec of src/all_optimizations.py: :25-27 (redescending a, b, c), :64-217 (sympy FK, pt3d_to_2d),
    :226-241 (DataFrame accessors, proj_funcs), :243-252 (R, Q), :268-277 (nose-line estimate), :283-500 (sets, weights,
    parameters, variables, initialisation, every constraint, the objective) against tests/golden/_float_pyomo.py,
    whose Var/Param/Constraint/Objective hold floats: constraints evaluate to residuals, the objective to a number.
    Recorded: Q, R, both weight tables, the 21 boxes (recovered by probing each inequality rule), init_x, and for
    random iterates x the objective value with all equality constraints satisfied (dx, ddx, slack_model, poses and
    slack_meas are SOLVED from the reference's own constraint residuals, one variable per constraint)."""
    import pandas as pd
    import sympy as sp
    from scipy.stats import linregress
    sys.path.insert(0, OUT)
    import _float_pyomo as fp
    sys.path.insert(0, os.path.join(REF, "src"))
    import build as ref_build  # noqa: E402  (stubbed import; supplies redescending_loss = lib.misc's)

    rng = np.random.default_rng(2024)
    scene = json.load(open(os.path.join(REF, "configs", "dummy_scene.json")))
    C = 3
    K_arr = np.array([c["k"] for c in scene["cameras"]][:C], dtype=np.float64)
    D_arr = np.array([c["d"] for c in scene["cameras"]][:C], dtype=np.float64).reshape((-1, 4))
    R_arr = np.array([c["r"] for c in scene["cameras"]][:C], dtype=np.float64)
    t_arr = np.array([c["t"] for c in scene["cameras"]][:C], dtype=np.float64)
    markers = ["l_eye", "r_eye", "nose", "neck_base", "spine", "tail_base", "tail1", "tail2",
               "l_shoulder", "l_front_knee", "l_front_ankle", "r_shoulder", "r_front_knee",
               "r_front_ankle", "l_hip", "l_back_knee", "l_back_ankle", "r_hip", "r_back_knee", "r_back_ankle"]
    tot_frames, start_frame, end_frame, fps, dlc_thresh = 9, 2, 8, 120.0, 0.5      # 0-based start (after :56)
    N, L = end_frame - start_frame, 20
    # INPUT data only: detections near the projections of a smooth trajectory (so residuals fall in every zone of
    # the redescending loss), built with the repo's oracle - the expected outputs below come from the reference text
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import camera as ocam, fk as ofk
    act = [0, 1, 2, 3, 4, 6] + list(range(17, 31)) + [31, 32, 34, 35, 36]
    tt = np.arange(tot_frames)[:, None] / 120.0
    X_true = np.zeros((tot_frames, 45))
    X_true[:, act] = 0.15 * np.sin(2 * np.pi * 2.0 * tt + rng.uniform(0, 6, len(act))[None, :])
    X_true[:, 0:3] = np.array([2.0, 6.0, 0.7]) + tt * np.array([8.0, -3.0, 0.1])
    pos_true = ofk.cheetah_fk(X_true)
    det = np.zeros((tot_frames, C, L, 3))
    for c in range(C):
        det[:, c, :, :2] = ocam.pt3d_to_2d(pos_true, K_arr[c], D_arr[c], R_arr[c], t_arr[c].reshape(3))
    det[..., :2] += rng.normal(0, 3.0, det[..., :2].shape)
    outl = rng.uniform(size=(tot_frames, C, L)) < 0.2
    det[..., :2] += outl[..., None] * rng.uniform(-80, 80, det[..., :2].shape)
    det[..., 2] = np.where(rng.uniform(size=(tot_frames, C, L)) < 0.3, rng.uniform(0, 0.49, (tot_frames, C, L)),
                           rng.uniform(0.51, 1, (tot_frames, C, L)))
    rows = [dict(frame=n, camera=c, marker=markers[l], x=det[n, c, l, 0], y=det[n, c, l, 1], likelihood=det[n, c, l, 2])
            for c in range(C) for n in range(tot_frames) for l in range(L)]
    points_2d_df = pd.DataFrame(rows, columns=["frame", "camera", "marker", "x", "y", "likelihood"])
    nose_tab = np.stack([np.arange(tot_frames, dtype=np.float64),
                         1.0 + 0.08 * np.arange(tot_frames) + rng.normal(0, 0.01, tot_frames),
                         5.0 - 0.03 * np.arange(tot_frames) + rng.normal(0, 0.01, tot_frames),
                         0.6 + rng.normal(0, 0.01, tot_frames)], 1)
    nose_tab = np.delete(nose_tab, 4, axis=0)                                      # a frame without a nose
    points_3d_df = pd.DataFrame(dict(frame=nose_tab[:, 0], marker="nose", x=nose_tab[:, 1], y=nose_tab[:, 2],
                                     z=nose_tab[:, 3]))

    class _Misc:
        redescending_loss = staticmethod(ref_build.redescending_loss)
    ns = dict(sp=sp, np=np, sin=fp.sin, cos=fp.cos, atan=fp.atan, linregress=linregress, misc=_Misc,
              ConcreteModel=fp.ConcreteModel, RangeSet=fp.RangeSet, Param=fp.Param, Var=fp.Var,
              Constraint=fp.Constraint, Objective=fp.Objective,
              K_arr=K_arr, D_arr=D_arr, R_arr=R_arr, t_arr=t_arr, markers=markers, points_2d_df=points_2d_df,
              points_3d_df=points_3d_df, start_frame=start_frame, end_frame=end_frame, dlc_thresh=dlc_thresh,
              Ts=1.0 / fps, print=lambda *a, **k: None)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        for lo, hi in ((25, 27), (64, 217), (226, 241), (243, 252), (268, 277), (283, 500)):
            exec(compile(_ref_lines(lo, hi), f"all_optimizations.py[{lo}:{hi}]", "exec"), ns)
    m, P = ns["m"], ns["P"]
    assert P == 45 and ns["N"] == N and ns["L"] == L and ns["C"] == C
    out = dict(det=det, nose_table=nose_tab, start_frame=start_frame, end_frame=end_frame, fps=fps,
               dlc_thresh=dlc_thresh, K=K_arr, D=D_arr, R=R_arr, t=t_arr,
               redesc=np.array([ns["redesc_a"], ns["redesc_b"], ns["redesc_c"]], dtype=np.float64),
               R_meas=float(ns["R"]), Q=np.asarray(ns["Q"], dtype=np.float64),
               model_err_weight=np.array([m.model_err_weight[p] for p in range(1, P + 1)], dtype=np.float64),
               meas_err_weight=np.array([[[float(fp._val(m.meas_err_weight[n, c, l])) for l in range(1, L + 1)]
                                          for c in range(1, C + 1)] for n in range(1, N + 1)]),
               meas=np.array([[[[m.meas[n, c, l, d] for d in (1, 2)] for l in range(1, L + 1)]
                               for c in range(1, C + 1)] for n in range(1, N + 1)]),
               init_x=np.array([[m.x[n, p].value for p in range(1, P + 1)] for n in range(1, N + 1)]),
               init_poses=np.array([[[m.poses[n, l, d].value for d in (1, 2, 3)] for l in range(1, L + 1)]
                                    for n in range(1, N + 1)]),
               x_est=np.stack([ns["x_est"], ns["y_est"], ns["z_est"]], 1), psi_est=float(ns["psi_est"]))

    # ---- the 21 boxes: every component whose rule returned an inequality touches ONE state p as |x_p + s| <= c
    lo_b, hi_b = np.full(P, -np.inf), np.full(P, np.inf)
    names = []
    for cname, comp in m._components.items():
        if not isinstance(comp, fp.Constraint) or not isinstance(next(iter(comp.data.values())), fp.Ineq):
            continue
        for p in range(1, P + 1):
            m.x[1, p].reads = 0
        comp.rule(m, 1)
        touched = [p for p in range(1, P + 1) if m.x[1, p].reads]
        assert len(touched) == 1, (cname, touched)
        p = touched[0]
        keep = m.x[1, p].value
        f = []
        for v in (0.0, 1.0):
            m.x[1, p].value = v
            q = comp.rule(m, 1)
            f.append(q.lhs)
            c_lim = q.rhs
        m.x[1, p].value = keep
        s = (f[1] ** 2 - f[0] ** 2 - 1.0) / 2.0
        assert abs(abs(s) - f[0]) < 1e-12
        lo_b[p - 1], hi_b[p - 1] = -c_lim - s, c_lim - s
        names.append(f"{cname}:{p}")
    out["bounds_lo"], out["bounds_hi"], out["bound_rules"] = lo_b, hi_b, np.array(names)

    return dict(m=m, P=P, N=N, L=L, C=C, fp=fp, out=out, rng=rng, X_true=X_true, act=act, start_frame=start_frame,
                end_frame=end_frame, lo_b=lo_b, hi_b=hi_b, names=names)


def _feasible_objective(ctx, X):
    """obj (:486-500) at the iterate X [N,45] with every equality of :359-399 satisfied through the reference's own
    residuals (each is affine with unit slope in the variable it defines).  Returns (obj, worst equality residual)."""
    m, P, N, L, C, fp = (ctx[k] for k in ("m", "P", "N", "L", "C", "fp"))
    Ts = getattr(m, ctx.get("ts_attr", "Ts"))

    def set_from_residual(comp, idx, var):
        v0 = 0.0 if var.value is None else var.value
        var.value = v0
        first = comp.rule(m, *idx)
        if first is fp.Constraint.Skip:                # (build.py skips the measurement rows of the marker "neck")
            return
        r0 = first.r
        var.value = v0 + 1.0
        slope = comp.rule(m, *idx).r - r0              # +-1 or +-Ts: the constraint is affine in `var`
        var.value = v0 - r0 / slope

    for n in range(1, N + 1):
        for p in range(1, P + 1):
            m.x[n, p].value = float(X[n - 1, p - 1])
            m.dx[n, p].value = 0.0
            m.ddx[n, p].value = 0.0
            m.slack_model[n, p].value = 0.0
    for p in range(1, P + 1):                      # n = 2..N: dx_n from integrate_p, then dx_1, ddx free
        for n in range(2, N + 1):
            set_from_residual(m.integrate_p, (n, p), m.dx[n, p])
        # free variables dx_1, ddx_1 chosen so that slack_2 = slack_3 = 0 (the optimum of the free variables)
        ddx3 = (m.dx[3, p].value - m.dx[2, p].value) / Ts
        m.ddx[1, p].value = m.ddx[2, p].value = ddx3
        m.dx[1, p].value = m.dx[2, p].value - Ts * ddx3
        for n in range(3, N + 1):
            set_from_residual(m.integrate_v, (n, p), m.ddx[n, p])
        for n in range(2, N + 1):
            set_from_residual(m.constant_acc, (n, p), m.slack_model[n, p])
    for n in range(1, N + 1):
        for l in range(1, L + 1):
            for d in (1, 2, 3):
                set_from_residual(m.pose_constraint, (n, l, d), m.poses[n, l, d])
            for c in range(1, C + 1):
                for d in (1, 2):
                    m.slack_meas[n, c, l, d].value = 0.0
                    set_from_residual(m.measurement, (n, c, l, d), m.slack_meas[n, c, l, d])
    worst = 0.0
    for cname in ("pose_constraint", "integrate_p", "integrate_v", "constant_acc", "measurement"):
        for v in getattr(m, cname).evaluate(m).values():
            if v is not fp.Constraint.Skip:
                worst = max(worst, abs(v.r))
    return m.obj.value(m), worst


def gen_fte_model():
    """fte_model.npz: see _fte_model_setup; plus, for random iterates x, the objective value with all equality
    constraints satisfied (dx, ddx, slack_model, poses and slack_meas are SOLVED from the reference's own constraint
    residuals, one variable per constraint)."""
    ctx = _fte_model_setup()
    m, P, N, fp, out, rng, X_true, act = (ctx[k] for k in ("m", "P", "N", "fp", "out", "rng", "X_true", "act"))
    start_frame, end_frame, lo_b, names = ctx["start_frame"], ctx["end_frame"], ctx["lo_b"], ctx["names"]
    cases_x, cases_obj, cases_slack, cases_dx, cases_ddx, max_res = [], [], [], [], [], []
    for case in range(5):
        X = X_true[start_frame:end_frame].copy()
        X[:, act] += rng.normal(0, (1e-5, 3e-4, 3e-3, 3e-4, 3e-4)[case], (N, len(act)))
        if case == 3:
            X[:, 5] = rng.normal(0, 0.01, N)          # a state with Q = 0: weight 0, must not change the objective
        if case == 4:
            X[:, 0:2] -= np.array([4.5, 8.0])         # the animal BEHIND camera 0: pt3d_to_2d has no z cut (:193-209)
        obj, worst = _feasible_objective(ctx, X)
        max_res.append(worst)
        cases_x.append(X)
        cases_obj.append(obj)
        cases_slack.append(np.array([[m.slack_model[n, p].value for p in range(1, P + 1)] for n in range(1, N + 1)]))
        cases_dx.append(np.array([[m.dx[n, p].value for p in range(1, P + 1)] for n in range(1, N + 1)]))
        cases_ddx.append(np.array([[m.ddx[n, p].value for p in range(1, P + 1)] for n in range(1, N + 1)]))
    out.update(x_true=X_true, case_x=np.array(cases_x), case_obj=np.array(cases_obj), case_slack_model=np.array(cases_slack),
               case_dx=np.array(cases_dx), case_ddx=np.array(cases_ddx), case_max_eq_residual=np.array(max_res))
    np.savez_compressed(os.path.join(OUT, "fte_model.npz"), **out)
    print("fte_model.npz: obj", out["case_obj"], "max equality residual", out["case_max_eq_residual"],
          "boxes", int(np.isfinite(lo_b).sum()), names[:3])


def _build_lines(lo, hi):
    """Lines lo..hi (1-based, inclusive) of the reference's src/build.py, dedented."""
    src = open(os.path.join(REF, "src", "build.py")).read().splitlines()
    return textwrap.dedent("\n".join(src[lo - 1:hi]))


def _skel_model_setup(n_frames=8, start_frame=60):
    """The reference's OWN skeleton-driven FTE model text (build_model, src/build.py:28-304) on floats, on the SHIPPED
    inputs: skeletons/new_human.pickle (the one its __main__ loads, :491), data/4_cam_scene_static_sba.json (2 cameras),
    the two DeepLabCut tables data/*.h5 (read with oracle/skel_fte.read_dlc_h5: pytables is absent).

    Slice-exec of build.py :32-95 (sympy poses from the skeleton dictionary, pose_to_3d / pos_funcs), :100-102, :113-129
    (DataFrame accessors), :131 (h), :134-142 (P, L, C, proj_funcs, R), :151-165 (forehead line estimate), :168-302 (sets,
    weights, parameters, variables, initialisation, every constraint, the ConstraintList of angle limits :263-266, the objective)
    against tests/golden/_float_pyomo.py.  NOT executed: :97-99 / :104-109 (file loading: the arrays are supplied), :132-133
    (start_frame = 60 kept, N = 100 replaced by n_frames to keep the fixture small - the text is O(N^2)), :144-147 (the
    triangulation, which needs cv2: the 3-D table comes from the reference's get_pairwise_3d_points_from_df with the
    repo's oracle as the injected triangulate_func)."""
    import glob
    import pandas as pd
    import sympy as sp
    from scipy import stats
    sys.path.insert(0, OUT)
    import _float_pyomo as fp
    sys.path.insert(0, os.path.join(REF, "src"))
    import build as ref_build  # noqa: E402
    from calib import calib as ref_calib  # noqa: E402
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import camera as ocam, skel_fte as osf

    skel = pickle.load(open(os.path.join(REF, "skeletons", "new_human.pickle"), "rb"))
    scene = json.load(open(os.path.join(REF, "data", "4_cam_scene_static_sba.json")))
    K_arr = np.array([c["k"] for c in scene["cameras"]], dtype=np.float64)
    D_arr = np.array([c["d"] for c in scene["cameras"]], dtype=np.float64)
    R_arr = np.array([c["r"] for c in scene["cameras"]], dtype=np.float64)
    t_arr = np.array([c["t"] for c in scene["cameras"]], dtype=np.float64)
    # body-part order of the shipped tables (the column index itself needs pytables)
    parts = ["ankle1", "knee1", "hip1", "hip2", "knee2", "ankle2", "wrist1", "elbow1", "shoulder1", "shoulder2", "elbow2",
             "wrist2", "chin", "forehead"]
    paths = sorted(glob.glob(os.path.join(REF, "data", "*.h5")))
    tabs = [osf.read_dlc_h5(f)[1] for f in paths]
    assert len(tabs) == len(K_arr) == 2 and all(t.shape[1] == len(parts) for t in tabs)
    f0, f1 = 0, start_frame + n_frames + 40            # rows of the long table (the line estimate regresses over all of them)
    rows = [dict(frame=n, camera=c, marker=parts[k], x=tabs[c][n, k, 0], y=tabs[c][n, k, 1], likelihood=tabs[c][n, k, 2])
            for c in range(len(tabs)) for n in range(f0, f1) for k in range(len(parts))]
    points_2d_df = pd.DataFrame(rows, columns=["frame", "camera", "marker", "x", "y", "likelihood"])

    def tri(a, b, k1, d1, r1, t1, k2, d2, r2, t2):
        return ocam.triangulate_points_fisheye(np.asarray(a, float).reshape(-1, 2), np.asarray(b, float).reshape(-1, 2),
                                               k1, np.asarray(d1).reshape(4), r1, np.asarray(t1).reshape(3),
                                               k2, np.asarray(d2).reshape(4), r2, np.asarray(t2).reshape(3))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        points_3d_df = ref_calib.get_pairwise_3d_points_from_df(points_2d_df[points_2d_df["likelihood"] > 0.4], K_arr,
                                                                D_arr.reshape((-1, 4)), R_arr, t_arr, tri)
    ns = dict(sp=sp, np=np, sin=fp.sin, cos=fp.cos, stats=stats, skel_dict=skel, points_2d_df=points_2d_df,
              points_3d_df=points_3d_df, K_arr=K_arr, D_arr=D_arr, R_arr=R_arr, t_arr=t_arr,
              rot_x=ref_build.rot_x, rot_y=ref_build.rot_y, rot_z=ref_build.rot_z,
              pt3d_to_x2d=ref_build.pt3d_to_x2d, pt3d_to_y2d=ref_build.pt3d_to_y2d,
              ConcreteModel=fp.ConcreteModel, RangeSet=fp.RangeSet, Param=fp.Param, Var=fp.Var, Constraint=fp.Constraint,
              ConstraintList=fp.ConstraintList, Objective=fp.Objective, Any=fp.Any, print=lambda *a, **k: None)
    import io, contextlib

    def run(lo, hi):
        exec(compile(_build_lines(lo, hi), f"build.py[{lo}:{hi}]", "exec"), ns)
    with contextlib.redirect_stdout(io.StringIO()):
        run(32, 95)
        run(100, 102)
        run(113, 129)
        run(131, 131)
        ns["start_frame"], ns["N"] = start_frame, n_frames
        run(134, 142)
        run(151, 165)
        run(168, 262)
        fp.READ_LOG = []
        run(263, 266)
        fp.READ_LOG = None
        run(267, 302)
    m, P, N, L, C = ns["m"], ns["P"], ns["N"], ns["L"], ns["C"]
    names = osf.pose_names(skel)
    assert P == 48 and L == len(names) == 15 and C == 2 and N == n_frames
    # the angle limits: every ConstraintList entry is |x[n, i]| <= c on ONE variable
    where = {id(v): k for k, v in m.x.data.items()}
    lo_b, hi_b = np.full((N, P), -np.inf), np.full((N, P), np.inf)
    for ineq, touched in m.angs.entries:
        assert len(touched) == 1 and isinstance(ineq, fp.Ineq)
        n, i = where[id(touched[0])]
        v = touched[0].value
        assert abs(ineq.lhs - abs(0.0 if v is None else v)) < 1e-15
        lo_b[n - 1, i - 1], hi_b[n - 1, i - 1] = -ineq.rhs, ineq.rhs
    meas = np.full((N, C, L, 2), np.nan)
    for n in range(1, N + 1):
        for c in range(1, C + 1):
            for l in range(1, L + 1):
                for d in (1, 2):
                    v = m.meas[n, c, l, d]
                    if v is not fp.Constraint.Skip:
                        meas[n - 1, c - 1, l - 1, d - 1] = float(v)
    out = dict(skeleton_json=np.array(json.dumps(skel)), parts=np.array(parts), pose_names=np.array(names),
               K=K_arr, D=D_arr.reshape(-1, 4), R=R_arr, t=t_arr, h=float(ns["h"]), R_meas=float(ns["R"]),
               start_frame=start_frame, n_frames=N,
               det=np.stack([t[f0:f1] for t in tabs], 1),                       # [frames, C, 14, 3] rows of the shipped tables
               forehead_table=points_3d_df[points_3d_df["marker"] == "forehead"][["frame", "x", "y", "z"]].values.astype(float),
               model_err_weight=np.array([m.model_err_weight[p] for p in range(1, P + 1)], dtype=np.float64),
               meas_err_weight=np.array([[[float(fp._val(m.meas_err_weight[n, c, l])) for l in range(1, L + 1)]
                                          for c in range(1, C + 1)] for n in range(1, N + 1)]),
               meas=meas, bounds_lo=lo_b, bounds_hi=hi_b,
               init_x=np.array([[m.x[n, p].value for p in range(1, P + 1)] for n in range(1, N + 1)]),
               init_poses=np.array([[[m.poses[n, l, d].value for d in (1, 2, 3)] for l in range(1, L + 1)]
                                    for n in range(1, N + 1)]))
    return dict(m=m, P=P, N=N, L=L, C=C, fp=fp, out=out, skel=skel, ts_attr="h")


def gen_skel_fte():
    """skel_fte_model.npz: constants, weight tables, the box table, the initialisation and - at iterates around the
    initial point - the OBJECTIVE of the reference's skeleton-driven model with every equality satisfied, plus its
    central-difference gradient in the 36 states the poses depend on at one of them."""
    ctx = _skel_model_setup()
    m, P, N, out = ctx["m"], ctx["P"], ctx["N"], ctx["out"]
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import skel_fte as osf
    act = osf.active_states(ctx["skel"])
    rng = np.random.default_rng(77)
    X0 = out["init_x"].copy()
    cases_x, cases_obj, worst = [], [], []
    for case in range(4):
        X = X0.copy()
        X[:, act] += rng.normal(0, (0.0, 1e-3, 2e-2, 0.2)[case], (N, len(act)))
        if case == 3:
            inact = np.setdiff1d(np.arange(P), act)
            X[:, inact] = rng.normal(0, 0.3, (N, len(inact))) * np.linspace(0, 1, N)[:, None] ** 3   # moves no pose: only the model term sees it
        obj, w = _feasible_objective(ctx, X)
        cases_x.append(X)
        cases_obj.append(obj)
        worst.append(w)
    Xg = cases_x[2]
    g = np.zeros((N, len(act)))
    hstep = 1e-6
    for n in range(N):
        for k, p in enumerate(act):
            Xp, Xm = Xg.copy(), Xg.copy()
            Xp[n, p] += hstep
            Xm[n, p] -= hstep
            g[n, k] = (_feasible_objective(ctx, Xp)[0] - _feasible_objective(ctx, Xm)[0]) / (2 * hstep)
    out.update(case_x=np.array(cases_x), case_obj=np.array(cases_obj), case_max_eq_residual=np.array(worst),
               grad_case=2, grad_fd=g, grad_fd_step=hstep, active=act)
    np.savez_compressed(os.path.join(OUT, "skel_fte_model.npz"), **out)
    # a longer slice of the SHIPPED detections (data rows only) for the real-data workload of the GPU tests / bench.py
    import glob
    tabs = [osf.read_dlc_h5(f)[1] for f in sorted(glob.glob(os.path.join(REF, "data", "*.h5")))]
    np.savez_compressed(os.path.join(OUT, "human_dlc_slice.npz"), det=np.stack([t[:460] for t in tabs], 1).astype(np.float32),
                        parts=out["parts"], note=np.array("rows 0..459 of data/Ex1Cam{3,4}...h5 (x, y, likelihood), float32 as stored by DeepLabCut"))
    # ... and the whole shipped video (every row of both tables): the full-length real-data workload (build.solve_video)
    np.savez_compressed(os.path.join(OUT, "human_dlc_full.npz"), parts=out["parts"],
                        note=np.array("every row of data/Ex1Cam{3,4}...h5 (x, y, likelihood), float32 as stored by DeepLabCut; one array "
                                      "per camera (the tables differ in length), row i = frame i"),
                        **{f"det{c}": t.astype(np.float32) for c, t in enumerate(tabs)})
    print("skel_fte_model.npz: obj", out["case_obj"], "max equality residual", out["case_max_eq_residual"],
          "bounded entries", int(np.isfinite(out["bounds_lo"]).sum()), "weights > 0:", int((out["meas_err_weight"] > 0).sum()))



def gen_fte_stationary():
    """fte_stationary.npz: the fixed point of the projected LM is a first-order stationary point of the REFERENCE's
    objective, evaluated by the reference's own model text (row a-10: IPOPT itself cannot run here).

    On the inputs of fte_model.npz the oracle's LM (oracle/fte.py, the control flow of the HIP path) is run from the
    reference's own initialisation to a tight tolerance; at its end point x* the reference objective (feasible
    completion as in gen_fte_model) is differentiated by central differences in all N x 25 active states.  Recorded:
    x*, obj_ref(x*), the finite-difference gradient at x* and at the initial point.  (About 15 minutes: 600 evaluations
    of the model text, each with every equality re-solved.)"""
    ctx = _fte_model_setup()
    out, N = ctx["out"], ctx["N"]
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import fk as ofk, fte as ofte
    s, e = ctx["start_frame"], ctx["end_frame"]
    det = out["det"][s:e]
    prob = ofte.FTEProblem(det[..., :2], det[..., 2], out["K"], out["D"], out["R"], out["t"], 1.0 / float(out["fps"]),
                           dlc_thresh=float(out["dlc_thresh"]))
    X0 = out["init_x"].copy()
    act = np.asarray(ofk.ACTIVE)
    xs, info = ofte.lm_solve(prob, X0[:, act], max_iter=300, ftol=1e-15, xtol=1e-13, gtol=1e-9)
    Xs = X0.copy()
    Xs[:, act] = xs

    def fd_grad(X, h=1e-6):
        g = np.zeros((N, len(act)))
        for n in range(N):
            for k, p in enumerate(act):
                Xp, Xm = X.copy(), X.copy()
                Xp[n, p] += h
                Xm[n, p] -= h
                g[n, k] = (_feasible_objective(ctx, Xp)[0] - _feasible_objective(ctx, Xm)[0]) / (2 * h)
        return g

    obj_star, worst = _feasible_objective(ctx, Xs)
    obj_init, _ = _feasible_objective(ctx, X0)
    g_star, g_init = fd_grad(Xs), fd_grad(X0)
    np.savez_compressed(os.path.join(OUT, "fte_stationary.npz"), x_star=Xs, obj_ref_star=obj_star, obj_ref_init=obj_init,
                        grad_ref_star=g_star, grad_ref_init=g_init, fd_step=1e-6, max_eq_residual=worst,
                        oracle_cost=info["cost"], oracle_iterations=info["iterations"], oracle_status=info["status"],
                        oracle_gnorm=info["gnorm"])
    lo, hi = prob.lo, prob.hi
    fixed = ((xs <= lo) & (g_star > 0)) | ((xs >= hi) & (g_star < 0))
    print("fte_stationary.npz: oracle", info["status"], info["iterations"], "it, cost", info["cost"], "obj_ref(x*)", obj_star,
          "| grad_ref(init)|_inf", np.abs(g_init).max(), "| projected grad_ref(x*)|_inf", np.abs(np.where(fixed, 0, g_star)).max(),
          "bound-active", int(fixed.sum()))


def gen_fte_lbfgs():
    """fte_lbfgs.npz: a THIRD-PARTY quasi-Newton optimiser on the reference's NLP (row a-10: the reference hands its model
    to IPOPT with hessian_approximation = limited-memory, all_optimizations.py:503-522; IPOPT cannot run here).

    scipy.optimize.minimize(method="L-BFGS-B") minimises the reduced objective over the N x 25 active states inside the
    21 boxes recovered from the reference's inequality rules, from the reference's own init_x.  Function and gradient come
    from oracle.fte.FTEProblem.evaluate - pinned to the reference's model text at five iterates (fte_model.npz, 1e-10) and
    its gradient to central differences of that text (fte_stationary.npz) -; the END POINT is then evaluated by the
    reference's own objective text with every equality re-solved (_feasible_objective), as x* of fte_stationary.npz was."""
    from scipy.optimize import minimize
    ctx = _fte_model_setup()
    out, N = ctx["out"], ctx["N"]
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import fk as ofk, fte as ofte
    s, e = ctx["start_frame"], ctx["end_frame"]
    det = out["det"][s:e]
    prob = ofte.FTEProblem(det[..., :2], det[..., 2], out["K"], out["D"], out["R"], out["t"], 1.0 / float(out["fps"]),
                           dlc_thresh=float(out["dlc_thresh"]))
    X0 = out["init_x"].copy()
    act = np.asarray(ofk.ACTIVE)
    x0 = np.clip(X0[:, act], prob.lo, prob.hi)
    nfev = [0]

    def fun(v):
        nfev[0] += 1
        F, g, _H, _nb = prob.evaluate(v.reshape(N, len(act)))
        return F, g.reshape(-1)

    bounds = [(None if not np.isfinite(lo) else float(lo), None if not np.isfinite(hi) else float(hi))
              for _ in range(N) for lo, hi in zip(prob.lo, prob.hi)]
    res = minimize(fun, x0.reshape(-1), jac=True, method="L-BFGS-B", bounds=bounds,
                   options=dict(maxiter=50000, maxfun=500000, ftol=1e-16, gtol=1e-9, maxcor=30))
    xl = res.x.reshape(N, len(act))
    Xl = X0.copy()
    Xl[:, act] = xl
    obj_ref, worst = _feasible_objective(ctx, Xl)
    F, g, _H, _nb = prob.evaluate(xl)
    fixed = ((xl <= prob.lo) & (g > 0)) | ((xl >= prob.hi) & (g < 0))
    np.savez_compressed(os.path.join(OUT, "fte_lbfgs.npz"), x_lbfgs=Xl, cost_oracle=F, obj_ref_lbfgs=obj_ref,
                        max_eq_residual=worst, nit=res.nit, nfev=res.nfev, message=str(res.message),
                        proj_grad_inf=np.abs(np.where(fixed, 0.0, g)).max(), bound_active=int(fixed.sum()))
    print("fte_lbfgs.npz: L-BFGS-B", res.message, "nit", res.nit, "nfev", res.nfev, "cost", F, "obj_ref", obj_ref,
          "|proj grad|_inf", np.abs(np.where(fixed, 0.0, g)).max(), "bound-active", int(fixed.sum()))


def gen_ekf():
    """The reference's OWN EKF + RTS-smoother text on two short synthetic clips (ekf_ref.npz).

    Slice-exec of src/all_optimizations.py ``ekf``: :582-593 (state indices), :603, :606-611, :615-649 (h_function,
    predict_next_state, numerical_jacobian), :668-679 (DLC table -> pixels / likelihood arrays), :684, :699-845
    (initial state, P0, Q, F, the filter loop, the smoother).  Supplied from outside, because ``lib.misc`` and cv2 are
    not in the reference tree: ``misc.get_pose_params`` = the 25 parameter names in the order of the ``qb_list``
    comments (:734-746) - THE ONE ASSUMPTION LEFT -, ``misc.get_3d_marker_coords`` = the reference's own sympy
    ``pose_to_3d`` (slice :64-190) evaluated in float64 under that order, ``project_points_fisheye`` = the KAT-1-pinned
    oracle projection.  Everything else (float32 state rounding, float32 forward-difference perturbation, gating, the
    explicit inverses) is whatever the reference text does under this container's numpy."""
    import pandas as pd
    import sympy as sp
    from scipy.stats import linregress
    from time import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import camera as ocam, synth as osynth
    fk_ns = {"sp": sp, "np": np, "sin": np.sin, "cos": np.cos}
    exec(compile(_ref_lines(64, 190), "all_optimizations.py[64:190]", "exec"), fk_ns)
    pose_to_3d = fk_ns["pose_to_3d"]
    names = ["x_0", "y_0", "z_0", "phi_0", "theta_0", "psi_0", "phi_1", "theta_1", "psi_1", "theta_2",
             "phi_3", "theta_3", "psi_3", "theta_4", "psi_4", "theta_5", "psi_5"] + [f"theta_{i}" for i in range(6, 14)]
    sym_names = [str(v).replace("\\", "").replace("{", "").replace("}", "") for v in fk_ns["sym_list"]]   # "\\phi_{0}" -> "phi_0"
    slot = {"x_0": "x", "y_0": "y", "z_0": "z"}
    order = [sym_names.index(slot.get(nm, nm)) for nm in names]          # position of each EKF parameter in sym_list
    markers = ["l_eye", "r_eye", "nose", "neck_base", "spine", "tail_base", "tail1", "tail2",
               "l_shoulder", "l_front_knee", "l_front_ankle", "r_shoulder", "r_front_knee",
               "r_front_ankle", "l_hip", "l_back_knee", "l_back_ankle", "r_hip", "r_back_knee", "r_back_ankle"]

    class _Misc:
        @staticmethod
        def get_pose_params():
            return {nm: i for i, nm in enumerate(names)}

        @staticmethod
        def get_markers():
            return list(markers)

        @staticmethod
        def get_3d_marker_coords(x):
            q = np.zeros(45)
            q[order] = np.asarray(x, dtype=np.float64)
            return np.asarray(pose_to_3d(*q), dtype=np.float64)

    out = dict(param_names=np.array(names), order45=np.array(order))
    import io, contextlib
    for tag, (n_tot, kind, seed, cams, sf1, ef) in (("a", (16, "sprint", 5, [0, 1, 2, 3, 4, 5], 3, 15)),
                                                    ("b", (12, "loop", 6, [0, 2, 5], 1, 12))):
        seq = osynth.make_sequence(n_tot, kind, seed=20210313 + seed)
        det = seq["det"][:, cams].copy()
        rng = np.random.default_rng(77 + seed)
        bad = rng.uniform(size=det.shape[:3]) < 0.04          # confident but wrong detections: exercise the 3-sigma gate
        det[..., :2] += bad[..., None] * rng.uniform(60, 140, det[..., :2].shape) * rng.choice([-1, 1], det[..., :2].shape)
        det[..., 2] = np.where(bad, 0.93, det[..., 2])
        k_arr, d_arr, r_arr, t_arr = seq["K"][cams], seq["D"][cams], seq["R"][cams], seq["t"][cams]
        C = len(cams)
        rows = [dict(frame=n, camera=c, marker=markers[l], x=det[n, c, l, 0], y=det[n, c, l, 1],
                     likelihood=det[n, c, l, 2]) for c in range(C) for n in range(n_tot) for l in range(20)]
        points_2d_df = pd.DataFrame(rows, columns=["frame", "camera", "marker", "x", "y", "likelihood"])
        nose = seq["pos_true"][:, 2] + rng.normal(0, 0.01, (n_tot, 3))
        points_3d_df = pd.DataFrame(dict(frame=np.arange(n_tot, dtype=np.float64), marker="nose", x=nose[:, 0],
                                         y=nose[:, 1], z=nose[:, 2]))
        ns = dict(np=np, pd=pd, linregress=linregress, time=time, misc=_Misc,
                  project_points_fisheye=ocam.project_points_fisheye, k_arr=k_arr, d_arr=d_arr, r_arr=r_arr, t_arr=t_arr,
                  cam_res=(2704, 1520), fps=120.0, n_cams=C, points_2d_df=points_2d_df, points_3d_df=points_3d_df,
                  start_frame=sf1, end_frame=ef, dlc_thresh=0.5, t0=time())
        with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for lo, hi in ((582, 593), (603, 603), (606, 611), (615, 649), (668, 679), (684, 684), (699, 845)):
                exec(compile(_ref_lines(lo, hi), f"all_optimizations.py[{lo}:{hi}]", "exec"), ns)
        sf = ns["start_frame"]                                                     # 0-based after :606
        assert ns["pixels_arr"].shape == (n_tot, C * 40) and ns["n_states"] == 75
        assert np.array_equal(ns["pixels_arr"], det[..., :2].reshape(n_tot, -1))  # :668-674 == the dense layout
        assert np.array_equal(ns["likelihood_arr"], det[..., 2].reshape(n_tot, -1))
        out.update({f"{tag}_det": det[sf:ef], f"{tag}_K": k_arr, f"{tag}_D": d_arr, f"{tag}_R": r_arr, f"{tag}_t": t_arr,
                    f"{tag}_nose_frames": np.arange(n_tot, dtype=np.float64)[sf:ef], f"{tag}_nose_xyz": nose[sf:ef],
                    f"{tag}_start_frame": sf, f"{tag}_P0": ns["P_pred_hist"][0], f"{tag}_Q": ns["Q"], f"{tag}_F": ns["F"],
                    f"{tag}_est": ns["states_est_hist"], f"{tag}_pred": ns["states_pred_hist"],
                    f"{tag}_smooth": ns["smooth_states_est_hist"], f"{tag}_outliers": ns["outliers_ignored"],
                    f"{tag}_P_est_last": ns["P_est_hist"][-1], f"{tag}_smooth_P_1": ns["smooth_P_est_hist"][1],
                    f"{tag}_P_est_head": ns["P_est_hist"][:5]})
        print("ekf", tag, "frames", ef - sf, "cams", C, "outliers", ns["outliers_ignored"],
              "pred dtype", ns["states"].dtype)
    np.savez_compressed(os.path.join(OUT, "ekf_ref.npz"), **out)


def gen_kat1():
    base = os.path.join(REF, "data", "sunday_amelia", "extrinsic_calib")
    out = {}
    for tag in ("rotating", "static"):
        s = json.load(open(os.path.join(base, f"4_cam_scene_{tag}.json")))
        out[f"{tag}_K"] = np.array([c["k"] for c in s["cameras"]])
        out[f"{tag}_D"] = np.array([c["d"] for c in s["cameras"]]).reshape(-1, 4)
        out[f"{tag}_R"] = np.array([c["r"] for c in s["cameras"]])
        out[f"{tag}_t"] = np.array([c["t"] for c in s["cameras"]]).reshape(-1, 3, 1)
    for i in range(1, 5):
        p = json.load(open(os.path.join(base, "points", f"points_cam{i}.json")))
        names = list(p["points"].keys())
        out[f"cam{i}_fnames"] = np.array(names)
        # utils.load_points (utils.py:37) loads these as float32
        out[f"cam{i}_points"] = np.array(list(p["points"].values()), dtype=np.float32)
        out["board_shape"] = np.array(p["board_shape"])
    # recorded in src/calib_with_gui.ipynb cell 29 outputs
    out["recorded"] = np.array([[-8.4537e-05, 0.18570246, 5.4156e+01],
                                [9.63562157113121e-05, 0.11887400393186973, 2.3636e+01]])
    # KAT-2, same cell: SBA end state [final cost, nfev, after-mean, after-std] (rotating, static)
    out["recorded_sba"] = np.array([[5.3361e+01, 690, 0.000692245847955209, 0.18554028389811675],
                                    [2.2845e+01, 50, 0.0013407166365445877, 0.1167727653391298]])
    np.savez_compressed(os.path.join(OUT, "kat1_sunday_amelia.npz"), **out)
    print("kat1:", {k: v.shape for k, v in out.items() if "points" in k})


def gen_kat34():
    out = {}
    for tag, f, skel in (("traj", "data/results/traj_results.pickle", "new_human"),
                         ("run1", "data/old_results/run1.pickle", "human")):
        d = pickle.load(open(os.path.join(REF, f), "rb"))
        for k in ("positions", "x", "dx", "ddx"):
            out[f"{tag}_{k}"] = np.asarray(d[k], dtype=np.float64)
        sk = pickle.load(open(os.path.join(REF, "skeletons", f"{skel}.pickle"), "rb"))
        out[f"{tag}_skeleton_json"] = np.array(json.dumps(sk))
    sk = pickle.load(open(os.path.join(REF, "skeletons", "cheetah.pickle"), "rb"))
    out["cheetah_skeleton_json"] = np.array(json.dumps(sk))
    np.savez_compressed(os.path.join(OUT, "kat34_build_runs.npz"), **out)
    print("kat34:", {k: v.shape for k, v in out.items()})


def gen_dlc_tables():
    """utils.create_dlc_points_2d_file (utils.py:105-120) on synthetic wide DLC tables: pandas.read_hdf is
    monkey-patched (pytables is absent) so the reference's own reshaping code produces the expected rows."""
    import pandas as pd
    sys.path.insert(0, os.path.join(REF, "src"))
    from calib import utils as ref_utils  # noqa: E402
    rng = np.random.default_rng(21)
    parts = ["nose", "l_eye", "r_eye", "neck_base", "spine"]
    wide = []
    for cam in range(3):
        n = 6
        cols = pd.MultiIndex.from_product([["DLC_resnet50_test"], parts, ["x", "y", "likelihood"]],
                                          names=["scorer", "bodyparts", "coords"])
        arr = rng.uniform(0, 1000, (n, len(parts) * 3))
        arr[:, 2::3] = rng.uniform(0, 1, (n, len(parts)))
        wide.append(pd.DataFrame(arr, columns=cols, index=np.arange(n)))
    orig = pd.read_hdf
    pd.read_hdf = lambda path, *a, **k: wide[int(path)].copy()
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            long_df = ref_utils.create_dlc_points_2d_file(["0", "1", "2"])
    finally:
        pd.read_hdf = orig
    out = dict(parts=parts, wide=[w.to_numpy().tolist() for w in wide],
               long_columns=list(long_df.columns),
               long_frame=[int(v) for v in long_df["frame"]], long_camera=[int(v) for v in long_df["camera"]],
               long_marker=list(long_df["marker"]),
               long_xyl=long_df[["x", "y", "likelihood"]].astype(float).values.tolist())
    json.dump(out, open(os.path.join(OUT, "dlc_tables.json"), "w"))
    print("dlc_tables:", len(long_df), "rows, columns", list(long_df.columns))


def gen_dummy_scene():
    s = json.load(open(os.path.join(REF, "configs", "dummy_scene.json")))
    json.dump(s, open(os.path.join(OUT, "dummy_scene.json"), "w"))


if __name__ == "__main__":
    assert os.path.isdir(REF), "reference tree not present: golden fixtures can only be regenerated in the build container"
    install_stubs()
    parts = sys.argv[1:] or ["helpers", "fk", "index", "kat1", "kat34", "scene", "dlc", "fte_model", "fte_stationary", "fte_lbfgs", "ekf",
                             "skel_fte"]
    for name, fn in (("helpers", gen_ref_helpers), ("fk", gen_cheetah_fk), ("index", gen_index_path),
                     ("fte_model", gen_fte_model), ("skel_fte", gen_skel_fte), ("fte_stationary", gen_fte_stationary), ("fte_lbfgs", gen_fte_lbfgs), ("ekf", gen_ekf), ("kat1", gen_kat1), ("kat34", gen_kat34), ("scene", gen_dummy_scene), ("dlc", gen_dlc_tables)):
        if name in parts:
            fn()
