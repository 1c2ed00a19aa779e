"""Skeleton-driven Full Trajectory Estimation: the drop-in for the reference's ``src/build.py``.

Reference: ``load_skeleton`` (build.py:18-26), ``build_model(skel_dict, project_dir)`` (:28-304: sympy poses from the
skeleton dictionary, the shipped scene / DeepLabCut tables, a Pyomo NLP), ``solve_optimisation(model, exe_path,
project_dir, poses)`` (:306-335: IPOPT, then ``save_data`` -> ``data/results/traj_results.pickle``), ``convert_to_dict``
(:343-365), ``save_data`` (:367-378).  Same names, same argument meaning, same files read and written; the model object is
a plain container of arrays instead of a Pyomo model, and the solve is the projected Levenberg-Marquardt of
``csrc/skel_fte.hip`` on the GPU (``exe_path`` - the IPOPT binary - is accepted and ignored).  What is solved, with the
reference's own index quirks, is written down in oracle/skel_fte.py and DESIGN.md section 8.
"""
import ctypes as C
import glob
import os
import pickle

import numpy as np
import torch

from . import _lib, calib, io, skeleton
from ._lib import SkelFteInfo, SkelFteParams, SkelOp, check, lib, ptr, stream_ptr

MODEL_WEIGHT = 0.002        # build.py:186-191
R_MEAS = 3.0                # :142
LIK_THRESH = 0.4            # :145, :187
H_STEP = 1.0 / 120.0        # :131
START_FRAME, N_FRAMES = 60, 100     # :132-133


def load_skeleton(skel_file):
    """build.py:18-26."""
    with open(skel_file, "rb") as handle:
        return pickle.load(handle)


def active_states(skel):
    """Indices (in the full state [x y z | phi | theta | psi], build.py:68) of the states a pose depends on: x, y, z and the
    enabled angles of every part that is the PARENT of a link (a part's own rotation only moves its children, :77)."""
    dofs = {k: list(v) for k, v in skel["dofs"].items()}
    for joint in skel["markers"]:
        dofs[joint] = [1, 1, 1]
    parts = list(dofs.keys())
    L = len(skel["positions"])
    parents = {link[0] for link in skel["links"] if len(link) == 2}
    act = [0, 1, 2]
    for axis in range(3):
        for i, part in enumerate(parts):
            if part in parents and dofs[part][axis]:
                act.append(3 + axis * L + i)
    return np.array(sorted(act), dtype=np.int32)


def bounds_table(skel, n_frames):
    """The ConstraintList of build.py:263-266, as written: |x[n, i]| <= pi/2 for the frames n = 1 .. N-1 and the 1-based state
    index i = 3 .. 3 len(positions) - 1 (the z coordinate is in, the last four angles and the last frame are out)."""
    L = len(skel["positions"])
    lo = np.full((n_frames, 3 + 3 * L), -np.inf)
    hi = np.full((n_frames, 3 + 3 * L), np.inf)
    lo[:n_frames - 1, 2:3 * L - 1] = -np.pi / 2
    hi[:n_frames - 1, 2:3 * L - 1] = np.pi / 2
    return lo, hi


def marker_pairing(skel, names, how="reference"):
    """Which detections feed pose slot l.  "reference": the marker at the same POSITION in the skeleton's marker list
    (``get_meas_from_df(n, c, l, d)`` looks up ``markers[l-1]`` while the projection uses ``pos_funcs[l-1]``, build.py:113-128
    with :288-292) - for the shipped skeletons that is NOT the part of the same name; "name": the part of the same name.
    The marker called "neck" carries no measurement (:120-121, :289)."""
    markers = list(skel["markers"])
    if how == "reference":
        return [(markers[l] if l < len(markers) and markers[l] != "neck" else None) for l in range(len(names))]
    if how == "name":
        return [n if n in markers and n != "neck" else None for n in names]
    raise ValueError("pairing must be 'reference' or 'name'")


class SkeletonModel:
    """What ``build_model`` returns in place of the Pyomo ConcreteModel: the arrays of the NLP."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    @property
    def N(self):
        return int(self.meas.shape[0])

    @property
    def P(self):
        return 3 + 3 * self.prog["n_angles"]


def build_model(skel_dict, project_dir=None, *, scene=None, dlc_tables=None, n_frames=N_FRAMES, start_frame=START_FRAME,
                h=H_STEP, pairing="reference", lik_thresh=LIK_THRESH, r_meas=R_MEAS, model_weight=MODEL_WEIGHT):
    """build.py:28-304.  ``project_dir`` is the reference's: ``data/4_cam_scene_static_sba.json`` and ``data/*.h5`` are read
    from it (:97-109); alternatively pass ``scene = (k_arr, d_arr, r_arr, t_arr)`` and ``dlc_tables`` = one
    ``(bodyparts, values[frames, K, 3])`` per camera.  Returns ``(model, pose_to_3d)`` as the reference does."""
    prog = skeleton.compile_skeleton(skel_dict)
    names = prog["names"]
    if scene is None:
        scene = io.load_scene(os.path.join(project_dir, "data", "4_cam_scene_static_sba.json"))[:4]
    k_arr, d_arr, r_arr, t_arr = (np.asarray(a, dtype=np.float64) for a in scene)
    d_arr = d_arr.reshape((-1, 4))                                           # :100
    if dlc_tables is None:
        paths = sorted(glob.glob(os.path.join(project_dir, "data", "*.h5")))  # :106
        dlc_tables = [io.read_dlc_table(p) for p in paths]
    C_ = len(k_arr)
    if len(dlc_tables) != C_:
        raise ValueError(f"{len(dlc_tables)} detection tables for {C_} cameras")
    # a table is (bodyparts, values[rows, K, 3]) or (bodyparts, values, frame index[rows]): the reference looks a detection up by
    # the VALUE of its frame index (utils.py:105-120 -> `frame == n - 1`), not by its row number, and every table by its own
    # body-part order
    tabs = []
    for tb in dlc_tables:
        parts, vals = list(tb[0]), np.asarray(tb[1], dtype=np.float64)
        idx = np.arange(vals.shape[0], dtype=np.int64) if len(tb) < 3 or tb[2] is None else np.asarray(tb[2], dtype=np.int64)
        if idx.shape != (vals.shape[0],) or np.unique(idx).size != idx.size:
            raise ValueError("a detection table needs one distinct frame index per row")
        tabs.append((parts, vals, idx))
    want = np.arange(start_frame, start_frame + n_frames, dtype=np.int64)
    rows = []
    for parts, vals, idx in tabs:
        order = np.argsort(idx)
        pos = np.searchsorted(idx[order], want)
        if (pos >= idx.size).any() or (idx[order][np.minimum(pos, idx.size - 1)] != want).any():
            raise ValueError("frame window outside the detection tables")
        rows.append(order[pos])
    # ---- measurements and weights per pose slot (:113-128, :184-206)
    pair = marker_pairing(skel_dict, names, pairing)
    Lp = len(names)
    meas = np.full((n_frames, C_, Lp, 2), np.nan)
    w = np.zeros((n_frames, C_, Lp))
    for c, (parts, vals, _idx) in enumerate(tabs):
        for l, mk in enumerate(pair):
            if mk is None or mk not in parts:
                continue
            sl = vals[rows[c], parts.index(mk)]
            meas[:, c, l] = sl[:, :2]
            w[:, c, l] = np.where(sl[:, 2] > lik_thresh, 1.0 / r_meas, 0.0)
    # ---- initial point: line through the triangulated "forehead" over ALL frames the cameras share (:143-166), a regression
    #      against the frame VALUE, evaluated at 0 .. N-1 (:157)
    init_x = np.zeros((n_frames, 3 + 3 * prog["n_angles"]))
    if all("forehead" in parts for parts, _v, _i in tabs) and C_ >= 2:
        common = tabs[0][2]
        for _p, _v, idx in tabs[1:]:
            common = np.intersect1d(common, idx)
        cols = []
        for parts, vals, idx in tabs:
            order = np.argsort(idx)
            cols.append(vals[order[np.searchsorted(idx[order], common)], parts.index("forehead")][:, None, :])
        det = np.stack(cols, axis=1)                                                                            # [F, C, 1, 3]
        tri = calib.triangulate_pairs_dense(det, lik_thresh, k_arr, d_arr, r_arr, t_arr, return_masks=False)
        tri = np.asarray(tri.cpu().numpy() if isinstance(tri, torch.Tensor) else tri)[:, 0]
        ok = np.isfinite(tri).all(1)
        if ok.sum() >= 2:
            f = common.astype(np.float64)[ok]
            coef, *_ = np.linalg.lstsq(np.stack([f, np.ones_like(f)], 1), tri[ok], rcond=None)
            fe = np.arange(n_frames, dtype=np.float64)
            init_x[:, 0:3] = fe[:, None] * coef[0][None, :] + coef[1][None, :]
    lo, hi = bounds_table(skel_dict, n_frames)
    model = SkeletonModel(skel=skel_dict, prog=prog, names=names, active=active_states(skel_dict), meas=meas, weights=w,
                          K=k_arr, D=d_arr, R=r_arr, t=t_arr, h=float(h), lo=lo, hi=hi, init_x=init_x,
                          start_frame=int(start_frame), model_weight=float(model_weight), pairing=pair, x=None, info=None)

    def pose_to_3d(*states):
        return np.asarray(skeleton.skeleton_fk(prog, np.asarray(states, dtype=np.float64)[None, :]))[0]
    return model, pose_to_3d


def _ops_array(prog):
    ops = (SkelOp * max(len(prog["ops"]), 1))()
    for i, (child, parent, angle, mask, untransposed, off) in enumerate(prog["ops"]):
        ops[i].child, ops[i].parent, ops[i].angle = child, parent, angle
        ops[i].flags = mask | (8 if untransposed else 0)
        ops[i].off[0], ops[i].off[1], ops[i].off[2] = off
    return ops


def solve_model(model, x0=None, max_iter=200, lam0=1e-3, ftol=1e-10, xtol=1e-10, gtol=1e-8, l1_eps=1e-2, lam_max=1e16):
    """The GPU solve of a ``SkeletonModel`` (acino_skel_fte_solve).  Returns (results, info): ``results`` has the layout of
    ``convert_to_dict`` (positions [N, n_pose, 3], x / dx / ddx [N, P]); states outside ``model.active`` keep their initial
    values - which must be 0, as in the reference's initialisation (:215-222)."""
    _lib.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())
    act = np.asarray(model.active, dtype=np.int32)
    x_full = np.array(model.init_x if x0 is None else x0, dtype=np.float64, copy=True)
    N, P = x_full.shape
    if (N, P) != (model.N, model.P):
        raise ValueError(f"x0 must be [{model.N}, {model.P}]")
    inactive = np.setdiff1d(np.arange(P), act)
    if np.any(x_full[:, inactive] != 0):
        raise ValueError("states that move no pose must start (and stay) at 0")
    prog = model.prog
    p = SkelFteParams()
    p.n_frames, p.n_cams, p.n_pose, p.n_ops = N, int(model.meas.shape[1]), len(model.names), len(prog["ops"])
    p.n_angles, p.n_active, p.max_iter = prog["n_angles"], len(act), int(max_iter)
    p.h, p.model_weight, p.l1_eps = float(model.h), float(model.model_weight), float(l1_eps)
    p.lam0, p.ftol, p.xtol, p.gtol, p.lam_max = float(lam0), float(ftol), float(xtol), float(gtol), float(lam_max)
    nbytes = lib().acino_skel_fte_workspace_bytes(C.byref(p))
    if nbytes == 0:
        raise ValueError("problem outside the kernel limits (n_active <= 64)")
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
    ws_ptr = (ws.data_ptr() + 255) // 256 * 256
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)   # noqa: E731
    meas, w = t(np.nan_to_num(model.meas, nan=0.0)), t(np.where(np.isfinite(model.meas).all(-1), model.weights, 0.0))
    cams = torch.as_tensor(calib.fisheye_records(model.K, model.D, model.R, model.t), device=dev)
    lo, hi = t(model.lo[:, act]), t(model.hi[:, act])
    x = t(x_full[:, act])
    pos = torch.empty((N, len(model.names), 3), dtype=torch.float64, device=dev)
    info = SkelFteInfo()
    act_c = (C.c_int32 * len(act))(*[int(a) for a in act])
    check(lib().acino_skel_fte_solve(C.byref(p), _ops_array(prog), act_c, ptr(meas), ptr(w), ptr(cams), ptr(lo), ptr(hi),
                                     ptr(x), ptr(pos), C.c_void_p(ws_ptr), nbytes, C.byref(info), stream_ptr()))
    x_full[:, act] = x.cpu().numpy()
    hh = float(model.h)
    dx, ddx = np.zeros_like(x_full), np.zeros_like(x_full)
    if N >= 2:
        dx[1:] = (x_full[1:] - x_full[:-1]) / hh
    if N >= 3:
        ddx[2:] = (dx[2:] - dx[1:-1]) / hh
        ddx[1] = ddx[0] = ddx[2]
        dx[0] = dx[1] - hh * ddx[1]
    results = dict(positions=pos.cpu().numpy(), x=x_full, dx=dx, ddx=ddx)
    return results, info.as_dict()


def convert_to_dict(m, poses=None):
    """build.py:343-365: the result dictionary of a solved model."""
    if m.x is None:
        raise ValueError("the model has not been solved")
    return dict(positions=m.x["positions"], x=m.x["x"], dx=m.x["dx"], ddx=m.x["ddx"])


def save_data(file_data, file_path, poses=None, dict=True):
    """build.py:367-378."""
    if dict:
        file_data = convert_to_dict(file_data, poses)
    os.makedirs(os.path.dirname(file_path), exist_ok=True)
    with open(file_path, "wb") as f:
        pickle.dump(file_data, f)
    print(f"save {file_path}")


def solve_optimisation(model, exe_path=None, project_dir=None, poses=None, **solver_kw):
    """build.py:306-335: solve, then save ``data/results/traj_results.pickle`` under ``project_dir`` (when given).
    ``exe_path`` named the IPOPT executable; there is none here."""
    results, info = solve_model(model, **solver_kw)
    model.x, model.info = results, info
    if project_dir is not None:
        save_data(model, file_path=os.path.join(project_dir, "data", "results", "traj_results.pickle"), poses=poses)
    return results, info
