"""Full Trajectory Estimation: the drop-in for the reference's Pyomo model + ``opt.solve(m)``.

Reference: src/all_optimizations.py:22-566 (``fte``).  The reference builds a Pyomo NLP
(:283-500) and hands it to IPOPT (:503-522); here the same objective, in its equivalent reduced form
(DESIGN.md), is minimised by a projected Levenberg-Marquardt that runs entirely on the GPU:
residuals / analytic Jacobians / normal-equation assembly (fte_assemble.hip), the block-tridiagonal
Gauss-Newton solve by block cyclic reduction on the fp64 matrix cores (bcr.hip) and the accept /
reject controller (fte_api.hip).  ``fte_solve`` returns the reference's ``fte.pickle`` layout
(:548-559): ``{positions [N,20,3], x [N,25], dx [N,25], ddx [N,25], start_frame}``.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib, calib
from ._lib import N_ACTIVE, N_MARKERS, N_STATES, FteParams, FteState, check, lib, ptr, stream_ptr

MARKERS = ["l_eye", "r_eye", "nose", "neck_base", "spine", "tail_base", "tail_mid", "tail_tip",
           "l_shoulder", "l_front_knee", "l_front_ankle", "r_shoulder", "r_front_knee", "r_front_ankle",
           "l_hip", "l_back_knee", "l_back_ankle", "r_hip", "r_back_knee", "r_back_ankle"]
PHI, THETA, PSI = 3, 17, 31          # state layout [x y z | phi_0..13 | theta_0..13 | psi_0..13]  (:182-185)
# the 25 states with Q != 0 (:245-252); convert_m (:530-556) drops the other 20 in this order
ACTIVE = np.array([0, 1, 2, PHI + 0, PHI + 1, PHI + 3] + [THETA + i for i in range(14)] +
                  [PSI + 0, PSI + 1, PSI + 3, PSI + 4, PSI + 5])
Q_SIGMA = np.array([4, 7, 5,
                    13, 32, 0, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                    9, 18, 43, 53, 90, 118, 247, 186, 194, 164, 295, 243, 334, 149,
                    26, 12, 0, 34, 43, 51, 0, 0, 0, 0, 0, 0, 0, 0], dtype=np.float64)
R_MEAS = 5.0                          # measurement std-dev, px (:243)
# acino_fte_params::precision: "f64" everywhere, or BASELINE config 5's "bf16 residuals with fp32 accumulate"
PRECISIONS = {"f64": 0, "bf16": 1, "bf16_residuals": 2}
REDESC = (3.0, 10.0, 20.0)            # redescending a, b, c (:25-27)


def bounds45():
    """Box bounds of :403-483 (1-based Pyomo indices -> 0-based)."""
    lo = np.full(N_STATES, -np.inf)
    hi = np.full(N_STATES, np.inf)
    for p in (4, 18, 5, 19, 33, 20, 21, 7, 35):
        lo[p - 1], hi[p - 1] = -np.pi / 6, np.pi / 6
    for p in (22, 36, 23, 37):
        lo[p - 1], hi[p - 1] = -np.pi / 1.5, np.pi / 1.5
    for p in (24, 26, 28, 30):
        lo[p - 1], hi[p - 1] = -np.pi / 2, np.pi / 2
    for p in (25, 27):
        lo[p - 1], hi[p - 1] = -np.pi, 0.0
    for p in (29, 31):
        lo[p - 1], hi[p - 1] = 0.0, np.pi
    return lo, hi


def make_params(n_frames, n_cams, Ts, dlc_thresh=0.5, r_meas=R_MEAS, Q=None, redesc=REDESC, lam0=1e-3,
                ftol=1e-10, xtol=1e-10, gtol=1e-8, n_global=None, n_offset=0, pin_left=False, pin_right=False,
                lam_max=1e16, clamp_lambda=False, shared_gpu=False, clip_len=0, precision="f64", bcr_levels=0,
                trunc_tol=1e-10, own_first=0, own_count=0, chunk_nodes=0, refine_sweeps=0):
    p = FteParams()
    p.n_frames, p.n_cams = int(n_frames), int(n_cams)
    p.n_global = int(n_frames if n_global is None else n_global)
    p.n_offset = int(n_offset)
    p.pin_left, p.pin_right = int(bool(pin_left)), int(bool(pin_right))
    p.dlc_thresh, p.inv_r_meas = float(dlc_thresh), 1.0 / float(r_meas)
    p.redesc_a, p.redesc_b, p.redesc_c = (float(v) for v in redesc)
    Qs = Q_SIGMA ** 2 if Q is None else np.asarray(Q, dtype=np.float64)
    if Qs.shape != (N_STATES,):
        raise ValueError("Q must have 45 entries")
    if set(np.nonzero(Qs)[0].tolist()) != set(ACTIVE.tolist()):
        raise ValueError("the cheetah kernels are specialised to the reference's 25 active states "
                         "(Q must be non-zero exactly where all_optimizations.py:245-252 is)")
    wq = (1.0 / Qs[ACTIVE]) / float(Ts) ** 4
    lo, hi = bounds45()
    for i in range(N_ACTIVE):
        p.q_w[i] = wq[i]
        p.lo[i] = lo[ACTIVE[i]]
        p.hi[i] = hi[ACTIVE[i]]
    p.lam0, p.ftol, p.xtol, p.gtol = float(lam0), float(ftol), float(xtol), float(gtol)
    p.lam_max, p.clamp_lambda = float(lam_max), int(bool(clamp_lambda))
    p.shared_gpu = int(bool(shared_gpu))
    p.clip_len = int(clip_len)
    if precision not in PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
    p.precision = PRECISIONS[precision]
    p.bcr_levels, p.trunc_tol = int(bcr_levels), float(trunc_tol)
    p.own_first, p.own_count = int(own_first), int(own_count)
    p.chunk_nodes, p.refine_sweeps = int(chunk_nodes), int(refine_sweeps)
    return p


def solver_plan(params):
    """Layout the library chooses for these parameters (acino_fte_plan): nodes per run of the chunked solver (0: block
    cyclic reduction over the whole chain), runs, separators, reduction levels of the reduced chain."""
    out = (C.c_int32 * 4)()
    check(lib().acino_fte_plan(C.byref(params), out))
    return dict(m=int(out[0]), n_chunks=int(out[1]), n_sep=int(out[2]), levels=int(out[3]))


def auto_bcr_levels(params, min_distance_frames=384):
    """Smallest K >= 1 after which the nodes that remain are >= min_distance_frames apart (3 * m * 2^K frames, m = nodes
    per run of the chunked solver, 1 without it), or 0 (complete reduction) when the chain has no level beyond it."""
    plan = solver_plan(params)
    K = 1
    while 3 * max(plan["m"], 1) * 2 ** K < min_distance_frames:
        K += 1
    return K if K <= plan["levels"] - 2 else 0      # (worth it from two saved levels on)


class FTEContext:
    """Owns the device buffers of one FTE problem (one shard of a sequence on one GPU)."""

    def __init__(self, det, k_arr, d_arr, r_arr, t_arr, Ts, **kw):
        _lib.require_gpu()
        dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        self.det = calib._to_dev(det, dev)
        if self.det.dim() != 4 or self.det.shape[2] != N_MARKERS or self.det.shape[3] != 3:
            raise ValueError("det must be [N, C, 20, 3] = (x, y, likelihood)")
        self.N, self.C = int(self.det.shape[0]), int(self.det.shape[1])
        self.Ts = float(Ts)
        self.cams = torch.as_tensor(calib.fisheye_records(k_arr, d_arr, r_arr, t_arr), device=dev)
        if self.cams.shape[0] != self.C:
            raise ValueError("camera count mismatch between det and the rig")
        self._kw = self._resolve_defaults(dict(kw))
        self._kw0 = dict(self._kw)                # as asked for: an escalation (status 7 -> one more level) edits self._kw
        self._escalated = False
        self._graph = False
        self._create()

    # Linear-solver defaults of a single-GPU context (no pinned separators): chunked substructuring, the separator chain
    # reduced until the remaining nodes are >= TRUNC_DISTANCE frames apart, their dropped couplings re-introduced by
    # REFINE_SWEEPS block-Jacobi sweeps - verified on the device every iteration (state["trunc_eps"] <= trunc_tol, status 7
    # "truncation" otherwise, on which solve() continues with one more level).  bcr_levels = 0 asks for the complete
    # reduction, chunk_nodes = -1 for block cyclic reduction over the whole chain (the round-1/2 solver).
    # (a sweep costs ~2.5 us since the sweeps trade their vectors as tagged words, a reduction level ~27 us: one level less and
    #  four sweeps more than rounds 4-6 had - 160 frames, 3 sweeps.  scripts/levels_probe.py: at 10 000 frames of the benchmark
    #  sequence one level + 7 sweeps leave a verified bound <= 3.2e-16 in every iteration of the solve, 6 sweeps 3.5e-14; slow
    #  gaits, whose smoothness prior couples further, are refused at this distance and escalate, as they did at 160.)
    TRUNC_DISTANCE = 80
    REFINE_SWEEPS = 7
    TRUNC_TOL = 1e-12

    @classmethod
    def _resolve_defaults(cls, kw):
        """The linear-solver defaults, resolved ONCE and kept in the keyword set every rebuild of the context starts from
        (an escalation only replaces ``bcr_levels``: refinement sweeps, tolerance and precision travel with it)."""
        pinned = bool(kw.get("pin_left")) or bool(kw.get("pin_right"))
        if "bcr_levels" not in kw and not pinned:
            kw["bcr_levels"] = "auto"
            kw.setdefault("trunc_distance", cls.TRUNC_DISTANCE)
            kw.setdefault("refine_sweeps", cls.REFINE_SWEEPS)
            kw.setdefault("trunc_tol", cls.TRUNC_TOL)
        return kw

    def _create(self):
        kw = dict(self._kw)
        auto = kw.get("bcr_levels") == "auto"
        if auto:
            kw["bcr_levels"] = 0
        dist = kw.pop("trunc_distance", 384)
        self.params = make_params(self.N, self.C, self.Ts, **kw)
        if auto:
            self.params.bcr_levels = auto_bcr_levels(self.params, dist)
        if self.params.bcr_levels == 0:
            self.params.refine_sweeps = 0
        nbytes = lib().acino_fte_workspace_bytes(C.byref(self.params))
        self.workspace = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        if os.environ.get("ACINO_POISON_WORKSPACE"):      # debug: every read-before-write of the workspace becomes a NaN
            self.workspace.view(torch.float64)[: (nbytes + 256) // 8].fill_(float("nan"))
        base = self.workspace.data_ptr()
        self._ws_ptr = (base + 255) // 256 * 256
        self._h = C.c_void_p()
        check(lib().acino_fte_create(C.byref(self._h), C.byref(self.params), ptr(self.det), ptr(self.cams),
                                     C.c_void_p(self._ws_ptr), nbytes, stream_ptr()))
        if self._graph:
            check(lib().acino_fte_enable_graph(self._h, 1))

    def _escalate(self, bound=None):
        """After status 7 (the truncated solve's verified error bound exceeded trunc_tol; the refused step was never
        applied): the same problem with more reduction levels (the complete reduction once the chain is exhausted),
        restarted from the current iterate.  ``bound``: the refused step's state["trunc_eps"], see _next_levels."""
        x = self.result()[0]
        levels = self._next_levels(bound)
        self.close()
        self._rebuild_with_levels(levels)
        check(lib().acino_fte_set_x(self._h, ptr(x), stream_ptr()))
        return levels

    def _next_levels(self, bound=None):
        """One level more - or, when the refused bound says how far off the truncation was, as many as it takes: the dropped
        couplings square with every level, so a bound b (~rho^(sweeps + 1)) becomes ~b^(2^j) after j more levels; the
        smallest j that brings it a decade under the tolerance.  (A bound of 1.0 means the sweeps did not contract by 1/2:
        no size to extrapolate from.)"""
        cur, jump = int(self.params.bcr_levels), 1
        tol = float(self.params.trunc_tol)
        if bound is not None and cur > 0 and 0.0 < bound < 1.0 and 0.0 < tol < 1.0:
            need = math.log(0.1 * tol) / math.log(bound)
            if need > 1.0:
                jump = max(1, int(math.ceil(math.log2(need) - 1e-9)))
        levels = cur + jump
        if cur == 0 or levels >= solver_plan(self.params)["levels"]:
            levels = 0
        return levels

    def _rebuild_with_levels(self, levels):
        """Same problem, same refinement settings / tolerance / precision, another number of reduction levels."""
        self._kw = dict(self._kw, bcr_levels=levels, refine_sweeps=int(self.params.refine_sweeps) or
                        int(self._kw.get("refine_sweeps", 0)), trunc_tol=float(self.params.trunc_tol))
        self._kw.pop("trunc_distance", None)
        self._escalated = True
        self._create()

    def reset_solver(self):
        """Back to the solver settings the context was created with, if an escalation changed them (contexts kept by
        ``reuse_context``: what a solve does must not depend on what earlier solves in the same context ran into)."""
        if self._escalated:
            self.close()
            self._kw = dict(self._kw0)
            self._escalated = False
            self._create()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().acino_fte_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- single-GPU driver -----------------------------------------------------------------------
    def set_x(self, x_active):
        x = calib._to_dev(x_active, self.device)
        if tuple(x.shape) != (self.N, N_ACTIVE):
            raise ValueError(f"x0 must be [{self.N}, 25] active states")
        self._x0 = x
        check(lib().acino_fte_set_x(self._h, ptr(x), stream_ptr()))

    def set_precision(self, precision):
        """Switch the assembly arithmetic (PRECISIONS) and re-evaluate the current iterate in it; the controller keeps
        its damping (after a lambda overflow: lam0) and goes back to "running" - used to polish a mixed-precision solve
        with fp64 iterations."""
        check(lib().acino_fte_set_precision(self._h, PRECISIONS[precision]))
        self._kw["precision"] = precision            # (a later rebuild - escalation - keeps the switched arithmetic)
        check(lib().acino_fte_reevaluate(self._h, stream_ptr()))

    def enable_graph(self, on=True):
        """Replay the LM step as a hipGraph (takes effect on a non-default stream)."""
        self._graph = bool(on)
        check(lib().acino_fte_enable_graph(self._h, int(bool(on))))

    def graphs_active(self):
        """Bit mask: bits 0..3 the four sharded phases, bit 4 the whole single-shard step."""
        return int(lib().acino_fte_graphs_active(self._h))

    def step(self):
        check(lib().acino_fte_step(self._h, stream_ptr()))

    def solve(self, max_iter):
        """Up to max_iter LM iterations IN TOTAL.  A step the truncated linear solve could not verify (status 7) is never
        applied: the context is rebuilt with more reduction levels (_next_levels) and the solve continues from the current iterate
        (the rebuilt controller starts from lam0 again); a refusal that uses up the last iteration is returned as status 7."""
        st = FteState()
        done = 0
        while True:
            check(lib().acino_fte_solve(self._h, max(int(max_iter) - done, 0), C.byref(st), stream_ptr()))
            info = st.as_dict()
            info["iter"] += done
            info["bcr_levels"] = int(self.params.bcr_levels)
            if info["status"] != 7 or info["iter"] >= int(max_iter):
                return info
            done = info["iter"]
            self._escalate(info.get("trunc_eps"))

    def state(self):
        st = FteState()
        check(lib().acino_fte_get_state(self._h, C.byref(st), stream_ptr()))
        return st.as_dict()

    def result(self):
        dev = self.device
        x = torch.empty((self.N, N_ACTIVE), dtype=torch.float64, device=dev)
        pos = torch.empty((self.N, N_MARKERS, 3), dtype=torch.float64, device=dev)
        dx = torch.empty_like(x)
        ddx = torch.empty_like(x)
        check(lib().acino_fte_get_result(self._h, self.Ts, ptr(x), ptr(pos), ptr(dx), ptr(ddx), stream_ptr()))
        return x, pos, dx, ddx

    PROF_CLASSES = ("elim", "elim_deep", "update0", "update", "update_deep", "backsub0", "backsub", "trial", "assemble",
                    "totals", "control", "backsub_tail", "trunc_check", "chunk_sweep", "sep_combine", "chunk_backsub", "refine")
    PROF_KERNELS = dict(elim="k_bcr_elim", elim_deep="k_sep_level", update0="k_bcr_update0", update="k_bcr_update",
                        update_deep="k_bcr_update_deep", backsub0="k_bcr_backsub0", backsub="k_bcr_backsub",
                        backsub_tail="k_bcr_backsub_tail", trial="k_trial", assemble="k_fte_assemble<true, 0, 1>", totals="k_totals", control="k_control",
                        trunc_check="k_bcr_trunc_check", chunk_sweep="k_chunk_sweep", sep_combine="k_sep_combine",
                        chunk_backsub="k_chunk_backsub", refine="k_sep_tail")

    def profile_begin(self):
        check(lib().acino_fte_profile_begin(self._h))

    def profile_end(self):
        """Per kernel class: summed HIP-event time (ms), launches, work units (chain nodes / frames)."""
        n = len(self.PROF_CLASSES)
        ms = (C.c_double * n)()
        cnt = (C.c_int * n)()
        units = (C.c_int64 * n)()
        check(lib().acino_fte_profile_end(self._h, ms, cnt, units, stream_ptr()))
        return {k: dict(ms=ms[i], launches=cnt[i], units=units[i]) for i, k in enumerate(self.PROF_CLASSES)}

    def cost(self, x_active):
        x = calib._to_dev(x_active, self.device)
        out = torch.zeros(1, dtype=torch.float64, device=self.device)
        check(lib().acino_fte_cost(self._h, ptr(x), ptr(out), stream_ptr()))
        return float(out.item())

    def grad_hess(self):
        g = torch.empty((self.N, N_ACTIVE), dtype=torch.float64, device=self.device)
        h = torch.empty((self.N, N_ACTIVE, N_ACTIVE), dtype=torch.float64, device=self.device)
        check(lib().acino_fte_get_grad_hess(self._h, ptr(g), ptr(h), stream_ptr()))
        return g, h


def cheetah_fk(q):
    """pose_to_3d of :170-186: q[N,45] full state -> positions[N,20,3]."""
    _lib.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())
    qd = calib._to_dev(q, dev).reshape(-1, N_STATES)
    pos = torch.empty((qd.shape[0], N_MARKERS, 3), dtype=torch.float64, device=dev)
    check(lib().acino_cheetah_fk(ptr(qd), qd.shape[0], ptr(pos), stream_ptr()))
    return calib._ret(pos, q)


def nose_line_init(det, k_arr, d_arr, r_arr, t_arr, dlc_thresh, n_frames=None, start_frame=0, det_first_frame=None):
    """Initial guess of :262-277,333-337: adjacent-pair triangulation of the detections above the
    likelihood threshold, least-squares line (linregress) through the nose (marker 2) over frames,
    psi_0 = atan2(y_slope, x_slope), every other state 0.  Returns x0[N,45] (numpy) for the frames
    start_frame .. start_frame + N - 1.

    The reference regresses over ALL triangulated frames of the video (:268-271) and evaluates the line on the window
    (:272-276, :334-337): pass the whole video's detections with ``det_first_frame=0`` and the window through
    ``start_frame`` / ``n_frames`` for exactly that.  By default (``det_first_frame=None``) ``det`` IS the window -
    its first row is frame ``start_frame`` - and the line is fitted to the window's own frames."""
    tri = calib.triangulate_pairs_dense(det, dlc_thresh, k_arr, d_arr, r_arr, t_arr, return_masks=False)
    nose = tri[:, 2] if isinstance(tri, np.ndarray) else tri[:, 2].cpu().numpy()
    first = start_frame if det_first_frame is None else det_first_frame
    N = (nose.shape[0] if det_first_frame is None else nose.shape[0] - (start_frame - first)) if n_frames is None else n_frames
    ok = np.isfinite(nose).all(1)
    if ok.sum() < 2:
        raise ValueError("fewer than two triangulated nose points: cannot fit the initial line")
    return nose_line_from_points(np.arange(nose.shape[0], dtype=np.float64)[ok] + first, nose[ok], N, start_frame)


def nose_line_from_points(frames, nose_xyz, n_frames, start_frame=0):
    """The regression itself (:268-277, :333-337) on a table of triangulated nose points (frame, xyz)."""
    f = np.asarray(frames, dtype=np.float64)
    A = np.stack([f, np.ones_like(f)], 1)
    coef, *_ = np.linalg.lstsq(A, np.asarray(nose_xyz, dtype=np.float64), rcond=None)
    out_frames = np.arange(start_frame, start_frame + n_frames, dtype=np.float64)
    x0 = np.zeros((n_frames, N_STATES))
    x0[:, 0:3] = out_frames[:, None] * coef[0][None, :] + coef[1][None, :]
    x0[:, PSI + 0] = np.arctan2(coef[0][1], coef[0][0])
    return x0


def triangulation_init(det, k_arr, d_arr, r_arr, t_arr, dlc_thresh):
    """Per-frame initial guess for LONG sequences (an extension: the reference's straight nose line
    cannot follow a trajectory that turns): head position from the triangulated head markers, heading
    from the neck_base->nose direction, gaps filled by linear interpolation; other states 0."""
    tri = calib.triangulate_pairs_dense(det, dlc_thresh, k_arr, d_arr, r_arr, t_arr, return_masks=False)
    tri = tri if isinstance(tri, np.ndarray) else tri.cpu().numpy()
    N = tri.shape[0]
    seen = np.isfinite(tri[:, 0:3]).all(-1)              # eyes + nose: mean of the ones that were triangulated
    cnt = seen.sum(1)
    head = np.where(seen[..., None], tri[:, 0:3], 0.0).sum(1) / np.maximum(cnt, 1)[:, None]
    head[cnt == 0] = np.nan                              # (no np.nanmean: frames without any head marker are expected, not a warning)
    fwd = tri[:, 2] - tri[:, 3]                          # neck_base -> nose
    idx = np.arange(N)
    x0 = np.zeros((N, N_STATES))
    for j in range(3):
        ok = np.isfinite(head[:, j])
        if ok.sum() == 0:
            raise ValueError("no triangulated head marker in the whole sequence")
        x0[:, j] = np.interp(idx, idx[ok], head[ok, j])
    ok = np.isfinite(fwd).all(1)
    if ok.sum() >= 1:
        psi = np.unwrap(np.arctan2(fwd[ok, 1], fwd[ok, 0]))
        x0[:, PSI + 0] = np.interp(idx, idx[ok], psi)
    return x0


# Contexts kept alive between solves (fte_solve(..., reuse_context=True)): a second sequence of the same shape on the same rig
# finds its workspace, its uploaded constants and its captured hipGraph in place - what is left outside the LM loop is the
# copy of the detections, the initial evaluation and the outputs.  One context per key; clear_context_cache() frees them.
_CTX_CACHE = {}


def clear_context_cache():
    for ctx in _CTX_CACHE.values():
        ctx.close()
    _CTX_CACHE.clear()


def _context_for(det, k_arr, d_arr, r_arr, t_arr, Ts, reuse, kw):
    if not reuse:
        return FTEContext(det, k_arr, d_arr, r_arr, t_arr, Ts, **kw), False
    import hashlib
    cams = np.ascontiguousarray(calib.fisheye_records(k_arr, d_arr, r_arr, t_arr))
    key = (tuple(det.shape), str(det.device), float(Ts), hashlib.sha256(cams.tobytes()).hexdigest(),
           tuple(sorted((k, repr(v)) for k, v in kw.items())))
    ctx = _CTX_CACHE.get(key)
    if ctx is None or not ctx._h.value:
        for old in list(_CTX_CACHE.values()):            # (one workspace at a time: a 10 000-frame context holds ~1 GB)
            old.close()
        _CTX_CACHE.clear()
        ctx = FTEContext(det.clone(), k_arr, d_arr, r_arr, t_arr, Ts, **kw)
        ctx.enable_graph(True)
        _CTX_CACHE[key] = ctx
    else:
        ctx.reset_solver()                                # (a no-op unless an earlier solve escalated the reduction levels)
        ctx.det.copy_(det)                                # (the library reads the detections through this tensor's pointer)
    return ctx, True


def triangulation_init_active(det, k_arr, d_arr, r_arr, t_arr, dlc_thresh, raise_now=True):
    """``triangulation_init`` for a detections tensor that lives on the GPU: the same initial guess, formed on the device and
    returned as the 25 active states [N, 25].  Two launches (pairwise triangulation, ``acino_fte_triangulation_init``: head
    mean, unwrapped heading, interpolation over the frames that lack them) and no host round trip - the numpy form copied
    the 4.8 MB triangulation of a 10 000-frame sequence to the host (3 ms of a 15 ms solve), a torch form of the same
    arithmetic took ~60 small launches and three synchronisations (2 ms).  ``raise_now=False`` returns ``(xa, flag)`` with the
    "no head marker in the whole sequence" flag left on the device for the caller to test later."""
    tri = calib.triangulate_pairs_dense(det, dlc_thresh, k_arr, d_arr, r_arr, t_arr, return_masks=False)
    n = int(tri.shape[0])
    xa = torch.empty((n, N_ACTIVE), dtype=torch.float64, device=tri.device)
    nbytes = int(lib().acino_fte_triangulation_init_scratch_bytes(n))
    scratch = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=tri.device)
    flag = torch.zeros(1, dtype=torch.int32, device=tri.device)
    psi_col = int(np.nonzero(ACTIVE == PSI + 0)[0][0])
    check(lib().acino_fte_triangulation_init(ptr(tri), n, int(tri.shape[1]), ptr(xa), N_ACTIVE, psi_col, ptr(scratch),
                                                scratch.numel() * 8, ptr(flag), stream_ptr()))
    if not raise_now:
        return xa, flag
    if int(flag.item()):
        raise ValueError("no triangulated head marker in the whole sequence")
    return xa


def fte_solve(meas, likelihood, k_arr, d_arr, r_arr, t_arr, Ts, x0=None, dlc_thresh=0.5, start_frame=0,
              max_iter=100, init="nose_line", return_numpy=True, reuse_context=False, **kw):
    """The FTE solve call.

    meas[N,C,20,2] pixel detections, likelihood[N,C,20], cameras as in the scene file (k_arr[C,3,3],
    d_arr[C,4(,1)], r_arr[C,3,3], t_arr[C,3(,1)]), Ts = 1/fps.  x0[N,45] optional initial state
    (default: the reference's nose-line initialisation).  Returns (results, info) where results has the
    reference's fte.pickle layout and info the solver status (iterations, final cost, |g|_inf, ...).
    ``reuse_context``: keep the context (workspace, constants, captured graph) for the next call with the same shapes, rig
    and options (see _CTX_CACHE above)."""
    meas_t = meas if isinstance(meas, torch.Tensor) else torch.as_tensor(np.asarray(meas, dtype=np.float64))
    lik_t = likelihood if isinstance(likelihood, torch.Tensor) else torch.as_tensor(np.asarray(likelihood, dtype=np.float64))
    det = torch.cat([meas_t.to(torch.float64), lik_t.to(torch.float64).unsqueeze(-1).to(meas_t.device)], dim=-1)
    _lib.require_gpu()
    det = det.to(torch.device("cuda", torch.cuda.current_device()))
    xa0 = init_flag = None
    if x0 is None:
        if init == "nose_line":
            x0 = nose_line_init(det, k_arr, d_arr, r_arr, t_arr, dlc_thresh, start_frame=start_frame)
        elif init == "triangulation":
            xa0, init_flag = triangulation_init_active(det, k_arr, d_arr, r_arr, t_arr, dlc_thresh, raise_now=False)   # (stays on the device)
        else:
            raise ValueError("init must be 'nose_line' or 'triangulation'")
    if xa0 is None:
        x0 = np.asarray(x0.cpu().numpy() if isinstance(x0, torch.Tensor) else x0, dtype=np.float64)
        if x0.shape != (det.shape[0], N_STATES):
            raise ValueError("x0 must be [N, 45]")
        inactive = np.setdiff1d(np.arange(N_STATES), ACTIVE)
        if np.any(x0[:, inactive] != 0):
            raise ValueError("states with Q == 0 must start (and stay) at 0 (all_optimizations.py:543)")
        xa0 = x0[:, ACTIVE]
    ctx, cached = _context_for(det, k_arr, d_arr, r_arr, t_arr, Ts, reuse_context, dict(kw, dlc_thresh=dlc_thresh))
    kw.pop("trunc_distance", None)
    try:
        ctx.set_x(xa0)
        info = ctx.solve(max_iter)
        x, pos, dx, ddx = ctx.result()
    except Exception:
        # (the initial guess's flag is read after the solve - no synchronisation in front of it -, but whatever a solve from an
        #  all-zero start ran into must not hide the real cause)
        if init_flag is not None and int(init_flag.item()):
            raise ValueError("no triangulated head marker in the whole sequence") from None
        raise
    finally:
        if not cached:
            ctx.close()
    if init_flag is not None and int(init_flag.item()):
        raise ValueError("no triangulated head marker in the whole sequence")
    if info["status"] == 5:
        raise RuntimeError("FTE: block factorisation hit a non-positive pivot")
    conv = (lambda a: a.cpu().numpy()) if return_numpy else (lambda a: a)
    results = dict(positions=conv(pos), x=conv(x), dx=conv(dx), ddx=conv(ddx), start_frame=start_frame)
    return results, info


def _derivatives(x_clip, Ts):
    dx, ddx = torch.empty_like(x_clip), torch.empty_like(x_clip)
    check(lib().acino_fte_derivatives(ptr(x_clip), x_clip.shape[0], float(Ts), ptr(dx), ptr(ddx), stream_ptr()))
    return dx, ddx


def fte_solve_clips(dets, k_arr, d_arr, r_arr, t_arr, Ts, x0s=None, dlc_thresh=0.5, start_frames=None, max_iter=100,
                    init="nose_line", return_numpy=True, **kw):
    """Equal-length clips of one rig solved as ONE problem (BASELINE config 5's batched FTE at full width): the clips
    are laid end to end on the frame axis, the smoothness prior is cut at the clip boundaries (``clip_len``), and the
    block-cyclic reduction runs over the whole chain - every launch is as wide as all clips together, so the narrow
    levels that dominate a single short clip almost vanish.  One Levenberg-Marquardt controller acts on the SUM of the
    clips' costs (the problem is block diagonal: each clip converges to its own optimum, but damping and accept/reject
    are shared, so iterates differ from per-clip solves until convergence).  Returns a list of (results, info)."""
    B = len(dets)
    if B == 0:
        return []
    dets_t = [d if isinstance(d, torch.Tensor) else torch.as_tensor(np.asarray(d, dtype=np.float64)) for d in dets]
    S = int(dets_t[0].shape[0])
    if any(int(d.shape[0]) != S for d in dets_t):
        raise ValueError("fte_solve_clips needs clips of equal length (use fte_solve_batch otherwise)")
    start_frames = list(start_frames) if start_frames is not None else [0] * B
    inactive = np.setdiff1d(np.arange(N_STATES), ACTIVE)
    x0_all = np.zeros((B * S, N_STATES))
    for b, det in enumerate(dets_t):
        if x0s is not None and x0s[b] is not None:
            x0 = x0s[b]
        elif init == "nose_line":
            x0 = nose_line_init(det, k_arr, d_arr, r_arr, t_arr, dlc_thresh, start_frame=start_frames[b])
        elif init == "triangulation":
            x0 = triangulation_init(det, k_arr, d_arr, r_arr, t_arr, dlc_thresh)
        else:
            raise ValueError("init must be 'nose_line' or 'triangulation'")
        x0 = np.asarray(x0.cpu().numpy() if isinstance(x0, torch.Tensor) else x0, dtype=np.float64)
        if x0.shape != (S, N_STATES):
            raise ValueError(f"x0 of clip {b} must be [{S}, 45]")
        if np.any(x0[:, inactive] != 0):
            raise ValueError("states with Q == 0 must start (and stay) at 0 (all_optimizations.py:543)")
        x0_all[b * S:(b + 1) * S] = x0
    dev = torch.device("cuda", torch.cuda.current_device())
    det_all = torch.cat([d.to(device=dev, dtype=torch.float64) for d in dets_t], dim=0)
    polish = kw.pop("polish_f64", False)
    ctx = FTEContext(det_all, k_arr, d_arr, r_arr, t_arr, Ts, dlc_thresh=dlc_thresh, clip_len=S, **kw)
    try:
        ctx.set_x(x0_all[:, ACTIVE])
        info = ctx.solve(max_iter)
        if polish and kw.get("precision", "f64") != "f64" and info["status"] in (0, 1, 2, 3, 4):
            # mixed-precision solve finished - by a stopping test or because no damping gives descent any more against
            # the fp64-summed cost (lambda overflow: the natural end of an inexact gradient) - : a few fp64 iterations
            # from its end point (same controller; damping kept, or back to lam0 after an overflow)
            n_mixed = info["iter"]
            ctx.set_precision("f64")
            info = ctx.solve(max(max_iter - n_mixed, 5))      # (a mixed solve that ran out of iterations is polished too)
            info["iter_mixed"] = n_mixed
        x, pos, _dx, _ddx = ctx.result()
        if info["status"] == 5:
            raise RuntimeError("FTE: block factorisation hit a non-positive pivot")
        conv = (lambda a: a.cpu().numpy()) if return_numpy else (lambda a: a)
        out = []
        for b in range(B):
            xb = x[b * S:(b + 1) * S].contiguous()
            dxb, ddxb = _derivatives(xb, Ts)                     # per clip: no differences across a clip boundary
            out.append((dict(positions=conv(pos[b * S:(b + 1) * S]), x=conv(xb), dx=conv(dxb), ddx=conv(ddxb),
                             start_frame=start_frames[b]), dict(info, clips=B, cost_is_sum_over_clips=True)))
        return out
    finally:
        ctx.close()


def fte_solve_batch(dets, k_arr, d_arr, r_arr, t_arr, Ts, x0s=None, dlc_thresh=0.5, start_frames=None, max_iter=100,
                    init="nose_line", n_streams=8, peek_every=8, return_numpy=True, **kw):
    """Several independent sequences (BASELINE config 5's "batched FTE": one rig, many clips) solved concurrently
    on ONE GPU.  Every sequence gets its own context and runs on one of ``n_streams`` HIP streams; a Levenberg-
    Marquardt step never synchronises with the host (the accept/reject controller is a device kernel and the step is
    one hipGraph launch), so the streams interleave and the narrow levels of one sequence's block-cyclic reduction
    overlap with the wide levels of another's (HIP multiplexes the streams onto 4 hardware queues by default; 8 streams
    keep all of them busy whatever the stream-to-queue assignment).  ``dets``: list of det[N_b, C, 20, 3] (lengths may differ).
    Returns a list of (results, info) exactly as ``fte_solve`` would for each sequence alone.  The reference solves
    clips one after another (src/all_optimizations.py:22, one ``fte()`` call per data directory)."""
    _lib.require_gpu()
    B = len(dets)
    if B == 0:
        return []
    start_frames = list(start_frames) if start_frames is not None else [0] * B
    streams = [torch.cuda.Stream() for _ in range(max(1, min(int(n_streams), B)))]
    inactive = np.setdiff1d(np.arange(N_STATES), ACTIVE)
    ctxs = []
    try:
        for b, det in enumerate(dets):
            det = det if isinstance(det, torch.Tensor) else torch.as_tensor(np.asarray(det, dtype=np.float64))
            if x0s is not None and x0s[b] is not None:
                x0 = x0s[b]
            elif init == "nose_line":
                x0 = nose_line_init(det, k_arr, d_arr, r_arr, t_arr, dlc_thresh, start_frame=start_frames[b])
            elif init == "triangulation":
                x0 = triangulation_init(det, k_arr, d_arr, r_arr, t_arr, dlc_thresh)
            else:
                raise ValueError("init must be 'nose_line' or 'triangulation'")
            x0 = np.asarray(x0.cpu().numpy() if isinstance(x0, torch.Tensor) else x0, dtype=np.float64)
            if x0.shape != (det.shape[0], N_STATES):
                raise ValueError(f"x0 of sequence {b} must be [N, 45]")
            if np.any(x0[:, inactive] != 0):
                raise ValueError("states with Q == 0 must start (and stay) at 0 (all_optimizations.py:543)")
            s = streams[b % len(streams)]
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                ctx = FTEContext(det, k_arr, d_arr, r_arr, t_arr, Ts, dlc_thresh=dlc_thresh, shared_gpu=True, **kw)
                ctxs.append(ctx)
                ctx.enable_graph(True)
                ctx.set_x(x0[:, ACTIVE])
        infos = [None] * B
        running = list(range(B))
        done_iter = 0
        while running and done_iter < max_iter:
            chunk = min(int(peek_every), max_iter - done_iter)
            for _ in range(chunk):
                for b in running:
                    with torch.cuda.stream(streams[b % len(streams)]):
                        ctxs[b].step()
            done_iter += chunk
            still = []
            for b in running:
                with torch.cuda.stream(streams[b % len(streams)]):
                    infos[b] = ctxs[b].state()
                if infos[b]["status"] == 0:
                    still.append(b)
            running = still
        out = []
        conv = (lambda a: a.cpu().numpy()) if return_numpy else (lambda a: a)
        for b in range(B):
            with torch.cuda.stream(streams[b % len(streams)]):
                if infos[b] is None:
                    infos[b] = ctxs[b].state()
                if infos[b]["status"] == 5:
                    raise RuntimeError(f"FTE: block factorisation hit a non-positive pivot (sequence {b})")
                x, pos, dx, ddx = ctxs[b].result()
                out.append((dict(positions=conv(pos), x=conv(x), dx=conv(dx), ddx=conv(ddx), start_frame=start_frames[b]),
                            infos[b]))
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
        return out
    finally:
        for ctx in ctxs:
            ctx.close()
