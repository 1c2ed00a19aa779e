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
    parts = sys.argv[1:] or ["helpers", "fk", "index", "kat1", "kat34", "scene", "dlc"]
    for name, fn in (("helpers", gen_ref_helpers), ("fk", gen_cheetah_fk), ("index", gen_index_path),
                     ("kat1", gen_kat1), ("kat34", gen_kat34), ("scene", gen_dummy_scene), ("dlc", gen_dlc_tables)):
        if name in parts:
            fn()
