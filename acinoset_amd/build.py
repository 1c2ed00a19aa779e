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


def _rows_by_frame_value(idx, want):
    """Rows of a detection table (frame index ``idx``, one distinct value per row) that hold the frames ``want`` - looked up by
    VALUE, as the reference does (utils.py:105-120 -> ``frame == n - 1``).  A wanted frame that the table does not hold is an
    error, never the next frame's row."""
    idx = np.asarray(idx, dtype=np.int64)
    want = np.asarray(want, dtype=np.int64)
    order = np.argsort(idx)
    pos = np.searchsorted(idx[order], want)
    if idx.size == 0 or (pos >= idx.size).any() or (idx[order][np.minimum(pos, idx.size - 1)] != want).any():
        raise ValueError("frame window outside the detection tables")
    return order[pos]


def _forehead_union(tabs, lik_thresh, k_arr, d_arr, r_arr, t_arr):
    """The triangulated forehead over every frame that AT LEAST TWO cameras hold (the reference's pairwise triangulation uses
    every frame a camera pair shares, calib.py:394-423) - frames, points [F, 3] (NaN where no pair sees it).  A camera without
    the frame contributes a row of likelihood 0."""
    frames = np.unique(np.concatenate([idx for _p, _v, idx in tabs]))
    held = np.stack([np.isin(frames, idx) for _p, _v, idx in tabs], axis=1)
    frames, held = frames[held.sum(1) >= 2], held[held.sum(1) >= 2]
    cols = []
    for c, (parts, vals, idx) in enumerate(tabs):
        col = np.zeros((frames.size, 1, 3))
        col[held[:, c], 0] = vals[_rows_by_frame_value(idx, frames[held[:, c]]), parts.index("forehead")]
        cols.append(col)
    tri = calib.triangulate_pairs_dense(np.stack(cols, axis=1), lik_thresh, k_arr, d_arr, r_arr, t_arr, return_masks=False)
    return frames, np.asarray(tri.cpu().numpy() if isinstance(tri, torch.Tensor) else tri)[:, 0]


def build_model(skel_dict, project_dir=None, *, scene=None, dlc_tables=None, n_frames=N_FRAMES, start_frame=START_FRAME,
                h=H_STEP, pairing="reference", lik_thresh=LIK_THRESH, r_meas=R_MEAS, model_weight=MODEL_WEIGHT, initial_line=True):
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
    rows = [_rows_by_frame_value(idx, want) for _parts, _vals, idx in tabs]
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
    # ---- initial point: line through the triangulated "forehead" over ALL frames that a camera pair shares (:143-166), a
    #      regression against the frame VALUE, evaluated at 0 .. N-1 (:157)
    init_x = np.zeros((n_frames, 3 + 3 * prog["n_angles"]))
    if initial_line and all("forehead" in parts for parts, _v, _i in tabs) and C_ >= 2:   # (initial_line=False: the caller brings x0)
        common, tri = _forehead_union(tabs, lik_thresh, k_arr, d_arr, r_arr, t_arr)
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


def _finite_diff_states(x_full, hh):
    """dx / ddx of convert_to_dict (build.py:343-365 takes them from the model's backward-Euler variables)."""
    N = x_full.shape[0]
    dx, ddx = np.zeros_like(x_full), np.zeros_like(x_full)
    if N >= 2:
        dx[1:] = (x_full[1:] - x_full[:-1]) / hh
    if N >= 3:
        ddx[2:] = (dx[2:] - dx[1:-1]) / hh
        ddx[1] = ddx[0] = ddx[2]
        dx[0] = dx[1] - hh * ddx[1]
    return dx, ddx


def solve_models(models, x0=None, max_iter=200, lam0=1e-3, ftol=1e-10, xtol=1e-10, gtol=1e-8, l1_eps=1e-2, lam_max=1e16):
    """The GPU solve of SEVERAL ``SkeletonModel`` s of the same skeleton, cameras and length in one call
    (acino_skel_fte_solve_batch: one workgroup per clip in the banded factorisation, a Levenberg-Marquardt controller per clip
    on the device).  ``x0``: None or one [N, P] array per model.  Returns ``[(results, info), ...]`` in the order of ``models``.
    In a batch a clip that fails numerically does not fail the call: its ``info["status_name"]`` is "numeric" (its results are
    the last accepted iterate) and the other clips' results stand."""
    _lib.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())
    m0 = models[0]
    B, N, P = len(models), m0.N, m0.P
    act = np.asarray(m0.active, dtype=np.int32)
    prog = m0.prog
    for m in models:
        if (m.N, m.P) != (N, P) or repr(m.prog["ops"]) != repr(prog["ops"]) or list(m.active) != list(m0.active) or m.meas.shape != m0.meas.shape:
            raise ValueError("the models of one batch share skeleton, active states, cameras and length")
        if not all(np.array_equal(np.asarray(getattr(m, k)), np.asarray(getattr(m0, k))) for k in ("K", "D", "R", "t")):
            raise ValueError("the models of one batch share the cameras")
        if (m.h, m.model_weight) != (m0.h, m0.model_weight):
            raise ValueError("the models of one batch share h and the model weight")
    xs = [np.array(m.init_x if x0 is None or x0[i] is None else x0[i], dtype=np.float64, copy=True) for i, m in enumerate(models)]
    inactive = np.setdiff1d(np.arange(P), act)
    for xf in xs:
        if xf.shape != (N, P):
            raise ValueError(f"x0 must be [{N}, {P}]")
        if np.any(xf[:, inactive] != 0):
            raise ValueError("states that move no pose must start (and stay) at 0")
    p = SkelFteParams()
    p.n_frames, p.n_cams, p.n_pose, p.n_ops = N, int(m0.meas.shape[1]), len(m0.names), len(prog["ops"])
    p.n_angles, p.n_active, p.max_iter = prog["n_angles"], len(act), int(max_iter)
    p.h, p.model_weight, p.l1_eps = float(m0.h), float(m0.model_weight), float(l1_eps)
    p.lam0, p.ftol, p.xtol, p.gtol, p.lam_max = float(lam0), float(ftol), float(xtol), float(gtol), float(lam_max)
    nbytes = lib().acino_skel_fte_workspace_bytes_batch(C.byref(p), B)
    if nbytes == 0:
        raise ValueError("problem outside the kernel limits (n_active <= 64)")
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
    ws_ptr = (ws.data_ptr() + 255) // 256 * 256
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)   # noqa: E731
    meas = t(np.stack([np.nan_to_num(m.meas, nan=0.0) for m in models]))
    w = t(np.stack([np.where(np.isfinite(m.meas).all(-1), m.weights, 0.0) for m in models]))
    cams = torch.as_tensor(calib.fisheye_records(m0.K, m0.D, m0.R, m0.t), device=dev)
    lo, hi = t(np.stack([m.lo[:, act] for m in models])), t(np.stack([m.hi[:, act] for m in models]))
    x = t(np.stack([xf[:, act] for xf in xs]))
    pos = torch.empty((B, N, len(m0.names), 3), dtype=torch.float64, device=dev)
    infos = (SkelFteInfo * B)()
    act_c = (C.c_int32 * len(act))(*[int(a) for a in act])
    check(lib().acino_skel_fte_solve_batch(C.byref(p), B, _ops_array(prog), act_c, ptr(meas), ptr(w), ptr(cams), ptr(lo), ptr(hi),
                                           ptr(x), ptr(pos), C.c_void_p(ws_ptr), nbytes, infos, stream_ptr()))
    xh, ph = x.cpu().numpy(), pos.cpu().numpy()
    out = []
    for i, (m, xf) in enumerate(zip(models, xs)):
        xf[:, act] = xh[i]
        dx, ddx = _finite_diff_states(xf, float(m.h))
        out.append((dict(positions=ph[i], x=xf, dx=dx, ddx=ddx), infos[i].as_dict()))
    return out


def solve_model(model, x0=None, max_iter=200, lam0=1e-3, ftol=1e-10, xtol=1e-10, gtol=1e-8, l1_eps=1e-2, lam_max=1e16):
    """The GPU solve of a ``SkeletonModel`` (acino_skel_fte_solve).  Returns (results, info): ``results`` has the layout of
    ``convert_to_dict`` (positions [N, n_pose, 3], x / dx / ddx [N, P]); states outside ``model.active`` keep their initial
    values - which must be 0, as in the reference's initialisation (:215-222).  A numeric failure raises (one clip: the
    failure is the call's)."""
    return solve_models([model], None if x0 is None else [x0], max_iter=max_iter, lam0=lam0, ftol=ftol, xtol=xtol, gtol=gtol,
                        l1_eps=l1_eps, lam_max=lam_max)[0]


def solve_model_parallel(model, x0=None, window=N_FRAMES, outer_max=40, xtol_outer=1e-7, first_max_iter=30, later_max_iter=30,
                         **solver_kw):
    """ONE long clip solved with every compute unit: alternating Schwarz on the nonlinear problem.  The clip is covered by
    windows of ``window`` frames that overlap by half; an outer iteration solves ALL windows in one batched call
    (``solve_models``: one workgroup and one controller per window) with the three frames at either inner end of a window
    PINNED (lo = hi) to the current global iterate - three frames are what the third-difference term reaches across - and
    every frame then takes its value from the window in which it lies deepest.  A free frame of a window sees exactly the
    residual rows and smoothness rows it sees in the whole-clip problem, so a fixed point of the outer iteration is a
    stationary point of the whole-clip objective; the error of the boundary values decays over the half-window overlap
    from one outer iteration to the next.  Each outer iteration reports the whole-clip cost and projected-gradient norm
    (one assembly of the full model).  Returns ``(results, info)`` as ``solve_model``; ``info`` carries the outer history.
    (The single-workgroup solve of the same clip, ``solve_model``, walks the banded factorisation frame by frame - 34 us
    per frame and iteration whatever the GPU's size.)  Windows are solved only ``first_max_iter`` / ``later_max_iter`` LM
    iterations per outer iteration: boundary values that are still wrong are not worth converging against.  Measured
    (MI355X, the shipped detections): 400 frames in 10 outer iterations, 1.1 s against 1.9 s for ``solve_model``, ending at
    a LOWER cost (an L1 objective on real detections has many stationary points); the whole 6 240-frame video does NOT
    reach ``xtol_outer`` in 40 outer iterations (9 s; the cost still falls by ~0.4 % per outer iteration): where a stretch of
    the video has no detections the trajectory is held by the smoothness term alone (weight 0.002 / h^4 = 4e5) and
    information crosses it half a window per outer iteration - ``solve_video`` (free windows) is the practical entry for that."""
    if "max_iter" in solver_kw:
        raise TypeError("solve_model_parallel: the inner iteration budget is first_max_iter / later_max_iter (and outer_max), not max_iter")
    N, P = model.N, model.P
    act = np.asarray(model.active, dtype=np.int64)
    if N < 2 * window:
        return solve_model(model, x0=x0, **solver_kw)
    starts = video_windows(0, N - 1, window, window // 2)
    x = np.array(model.init_x if x0 is None else x0, dtype=np.float64, copy=True)
    x[:, act] = np.clip(x[:, act], model.lo[:, act], model.hi[:, act])
    depth_w = np.minimum(np.arange(window), window - 1 - np.arange(window)).astype(np.float64)
    history = []
    res_full, info_full = None, None
    for outer in range(outer_max):
        subs = []
        for st in starts:
            lo, hi = model.lo[st:st + window].copy(), model.hi[st:st + window].copy()
            if st > 0:
                lo[:3, act] = hi[:3, act] = x[st:st + 3, act]
            if st + window < N:
                lo[-3:, act] = hi[-3:, act] = x[st + window - 3:st + window, act]
            subs.append(SkeletonModel(**{**model.__dict__, "meas": model.meas[st:st + window], "weights": model.weights[st:st + window],
                                         "lo": lo, "hi": hi, "init_x": x[st:st + window].copy(), "x": None, "info": None}))
        solved = solve_models(subs, max_iter=first_max_iter if outer == 0 else later_max_iter, **solver_kw)
        xn = x.copy()
        depth = np.full(N, -1.0)
        for st, (res, _i) in zip(starts, solved):
            sl = slice(st, st + window)
            take = depth_w > depth[sl]
            xn[sl][take] = res["x"][take]
            depth[sl] = np.maximum(depth[sl], depth_w)
        change = float(np.abs(xn - x).max())
        x = xn
        res_full, info_full = solve_model(model, x0=x, max_iter=0, **solver_kw)      # whole-clip cost and gradient norm at x
        history.append(dict(outer=outer + 1, change=change, cost=info_full["cost_final"], gnorm_inf=info_full["gnorm_inf"],
                            window_iterations=int(sum(i["iterations"] for _r, i in solved))))
        if change <= xtol_outer:
            break
    info = dict(info_full)
    info.update(outer_iterations=len(history), history=history, windows=len(starts), window_frames=window,
                status_name="outer_xtol" if history[-1]["change"] <= xtol_outer else "outer_max")
    return res_full, info


def video_windows(first_frame, last_frame, window, overlap):
    """First frames of the windows ``solve_video`` cuts first_frame .. last_frame into: ``window`` frames each, consecutive
    windows ``overlap`` frames apart from abutting, the last one pulled back so that it ends on ``last_frame``."""
    if not 0 <= overlap < window:
        raise ValueError("0 <= overlap < window")
    if last_frame - first_frame + 1 < window:
        raise ValueError(f"{last_frame - first_frame + 1} frames, windows of {window}")
    starts = list(range(first_frame, last_frame - window + 2, window - overlap))
    if starts[-1] + window - 1 < last_frame:
        starts.append(last_frame - window + 1)
    return starts


def window_residual_px(model, info):
    """Mean absolute reprojection residual (px) at the end state of a solved window: the L1 objective is sum w |e| with w = 1 / R
    on every detection above the threshold (x and y are two rows each), plus a smoothness term that is small beside it."""
    n_rows = 2 * int((np.asarray(model.weights) > 0).sum())
    return float(info["cost_final"]) * R_MEAS / max(n_rows, 1)


def solve_video(skel_dict, project_dir=None, *, scene=None, dlc_tables=None, first_frame=None, last_frame=None, window=N_FRAMES,
                overlap=20, warm_px=15.0, warm_passes=3, **kw):
    """A whole video as the reference would have to do it - windows of ``window`` frames (build.py:131-133: N = 100), here
    ALL of them in one batched GPU solve: consecutive windows overlap by ``overlap`` frames and every frame is taken from
    the window in which it lies deepest.  An extension (the reference solves one window per run): the initial point of a
    window is the triangulated forehead of its OWN frames, gaps interpolated (the reference fits one straight line through the
    forehead of the whole video and evaluates it at 0 .. N-1 whatever ``start_frame`` is, :143-166 - fine for its one window
    near the start, metres away for a window later in a video in which the subject turns round).

    Warm starts: a window that starts with all joint angles 0 can settle in a wrong local minimum of the L1 objective (the
    body facing the other way: mean residuals of 20 .. 140 px where its neighbours end at 2 .. 6).  After the first batched solve
    every window whose mean absolute residual exceeds ``warm_px`` is solved AGAIN from its better neighbour's end state - the
    shared ``overlap`` frames copied, the neighbour's joint angles at the nearest shared frame held over the rest, positions
    from the triangulated forehead - and the better of the two end states is kept; up to ``warm_passes`` passes, each one
    batched call over the windows concerned (a repaired window can repair its other neighbour in the next pass).

    ``dx`` / ``ddx`` are taken per window and selected with the same depth rule as ``x``: at a seam between two windows the
    stitched ``x`` jumps between two independently converged solutions, and differences ACROSS a seam would read as spikes of
    jump / h and jump / h^2.  ``kw``: build_model's (``pairing``, ``h``, ...) and solve_models' (``max_iter``, ...) keywords.
    Returns ``(results, infos, starts)``: ``results`` as convert_to_dict over frames first_frame .. last_frame (plus
    ``start_frame`` and ``seams``: the first frame, relative to first_frame, of every stretch taken from a new window), one
    info per window (with ``mean_abs_residual_px`` and ``warm_started_from``)."""
    build_kw = {k: kw.pop(k) for k in ("h", "pairing", "lik_thresh", "r_meas", "model_weight") if k in kw}
    if dlc_tables is None:
        paths = sorted(glob.glob(os.path.join(project_dir, "data", "*.h5")))
        dlc_tables = [io.read_dlc_table(p) for p in paths]
    idx = [np.arange(np.asarray(tb[1]).shape[0]) if len(tb) < 3 or tb[2] is None else np.asarray(tb[2]) for tb in dlc_tables]
    f0 = max(int(i.min()) for i in idx) if first_frame is None else int(first_frame)
    f1 = min(int(i.max()) for i in idx) if last_frame is None else int(last_frame)
    total = f1 - f0 + 1
    if total < window:
        raise ValueError(f"{total} frames, windows of {window}")
    starts = video_windows(f0, f1, window, overlap)
    # the forehead of every frame, triangulated once (the reference's initial point uses the same marker, build.py:143-166)
    head = None
    tabs3 = [(list(tb[0]), np.asarray(tb[1], dtype=np.float64), ix) for tb, ix in zip(dlc_tables, idx)]
    want = np.arange(f0, f1 + 1)
    rows3 = [_rows_by_frame_value(ix, want) for _parts, _vals, ix in tabs3]       # (every table must hold f0 .. f1: ValueError)
    if all("forehead" in parts for parts, _v, _i in tabs3) and len(tabs3) >= 2:
        if scene is None:
            scene = io.load_scene(os.path.join(project_dir, "data", "4_cam_scene_static_sba.json"))[:4]
        k_arr, d_arr, r_arr, t_arr = (np.asarray(a, dtype=np.float64) for a in scene)
        cols = [vals[rw, parts.index("forehead")][:, None, :] for (parts, vals, _ix), rw in zip(tabs3, rows3)]
        tri = calib.triangulate_pairs_dense(np.stack(cols, axis=1), build_kw.get("lik_thresh", LIK_THRESH), k_arr,
                                            d_arr.reshape((-1, 4)), r_arr, t_arr, return_masks=False)
        tri = np.asarray(tri.cpu().numpy() if isinstance(tri, torch.Tensor) else tri)[:, 0]
        ok = np.isfinite(tri).all(1)
        if ok.sum() >= 2:
            fr = np.arange(total, dtype=np.float64)
            head = np.stack([np.interp(fr, fr[ok], tri[ok, j]) for j in range(3)], axis=1)
    models, x0s = [], []
    for st in starts:
        m, _ = build_model(skel_dict, project_dir, scene=scene, dlc_tables=dlc_tables, n_frames=window, start_frame=st,
                           initial_line=head is None, **build_kw)
        x0 = m.init_x.copy()
        if head is not None:
            x0[:, :3] = head[st - f0:st - f0 + window]
        models.append(m)
        x0s.append(x0)
    solved = solve_models(models, x0s, **kw)
    px = [window_residual_px(m, i) for m, (_r, i) in zip(models, solved)]
    warm_from = [None] * len(starts)
    act = np.asarray(models[0].active, dtype=np.int64)
    for _pass in range(int(warm_passes)):
        redo, x0r = [], []
        for i, st in enumerate(starts):
            if not (px[i] > warm_px):
                continue
            cand = [j for j in (i - 1, i + 1) if 0 <= j < len(starts) and px[j] < px[i] and px[j] <= warm_px]
            if not cand:
                continue
            j = min(cand, key=lambda q: px[q])
            xj, sj = solved[j][0]["x"], starts[j]
            lo_g, hi_g = max(st, sj), min(st, sj) + window            # shared frames [lo_g, hi_g) (global numbering)
            if hi_g <= lo_g:
                continue
            x0 = x0s[i].copy()
            edge = xj[(hi_g - 1 if sj < st else lo_g) - sj]           # the neighbour's state at the shared frame nearest to the rest
            x0[:, act[act >= 3]] = edge[act[act >= 3]][None, :]        # its joint angles held over the window ...
            x0[lo_g - st:hi_g - st] = xj[lo_g - sj:hi_g - sj]          # ... and the shared frames as the neighbour left them
            x0[:, act] = np.clip(x0[:, act], models[i].lo[:, act], models[i].hi[:, act])
            redo.append((i, j))
            x0r.append(x0)
        if not redo:
            break
        again = solve_models([models[i] for i, _j in redo], x0r, **kw)
        improved = False
        for (i, j), (res, info) in zip(redo, again):
            p_new = window_residual_px(models[i], info)
            if info["status_name"] != "numeric" and p_new < px[i]:
                solved[i], px[i], warm_from[i], improved = (res, info), p_new, j, True
        if not improved:
            break
    Lp, P = len(models[0].names), models[0].P
    pos, x, dx, ddx = np.zeros((total, Lp, 3)), np.zeros((total, P)), np.zeros((total, P)), np.zeros((total, P))
    depth = np.full(total, -1.0)
    owner = np.full(total, -1)
    for w_i, (st, (res, _info)) in enumerate(zip(starts, solved)):
        d = np.minimum(np.arange(window), window - 1 - np.arange(window)).astype(np.float64)
        sl = slice(st - f0, st - f0 + window)
        take = d > depth[sl]
        pos[sl][take], x[sl][take] = res["positions"][take], res["x"][take]
        dx[sl][take], ddx[sl][take] = res["dx"][take], res["ddx"][take]
        owner[sl][take] = w_i
        depth[sl] = np.maximum(depth[sl], d)
    infos = []
    for i, (_r, info) in enumerate(solved):
        info = dict(info)
        info["mean_abs_residual_px"], info["warm_started_from"] = px[i], warm_from[i]
        infos.append(info)
    seams = [int(n) for n in np.nonzero(np.diff(owner) != 0)[0] + 1]
    return dict(positions=pos, x=x, dx=dx, ddx=ddx, start_frame=f0, seams=seams), infos, starts


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
