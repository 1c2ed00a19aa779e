"""Oracle generic-skeleton Full Trajectory Estimation (numpy fp64).  Test infrastructure - see oracle/__init__.py.

Restates the NLP that ``build_model`` (src/build.py:28-304) builds from a pickled skeleton - the reference's only FTE that
is driven by a skeleton file - in the same reduced form as oracle/fte.py:

    variables (:208-213): x, dx, ddx, slack_model [N, P],  P = 3 + 3 L  (:134; L = len(positions)), poses, slack_meas
    equalities: poses = pose_to_3d(x_n) (:244-250); backward Euler x_n = x_{n-1} + h dx_n, dx_n = dx_{n-1} + h ddx_n
      (n > 1, :252-268); ddx_n = ddx_{n-1} + slack_model_n (n > 1, :276-281); projection - meas - slack_meas = 0
      (:284-293; the marker called "neck" is skipped, :289).
    inequalities (:270-273): |x[n, i]| <= pi/2 for n = 1 .. N-1 and the 1-BASED state index i = 3 .. 3 len(positions) - 1,
      i.e. 0-based states 2 .. 3L-2 - the z coordinate included, the last four angles and the LAST FRAME not (kept as
      written: the fixture recovers exactly this box table from the reference's own ConstraintList).
    objective (:295-309):  sum_n sum_p 0.002 slack_model[n, p]^2  +  sum |w_ncl slack_meas[n, c, l, d]|     (an L1 loss;
      the redescending loss is commented out at :307), w = 1/R (R = 3, :142) where likelihood > 0.4, else 0 (:184-190).

dx_1, ddx_1 are free, so slack_2 = slack_3 = 0 at the optimum and (KAT-4, on the reference's stored runs of this very
model) slack_n = (x_n - 3 x_{n-1} + 3 x_{n-2} - x_{n-3}) / h^2:

    min_x  sum |w_ncl (pi_c(pose_l(x_n))_d - z_ncld)|  +  sum_{n >= 3, p} (0.002 / h^4) (third difference of x_p)^2
    s.t.   lo[n, p] <= x[n, p] <= hi[n, p]

A state that no pose depends on (the three angles of a part that is never the PARENT of a link: its rotation moves
nothing, build.py:77) only sees the smoothness term and stays at its initial value 0: the reduced problem carries the
``active`` states only - x, y, z and the enabled angles of parent parts.  The measurement pairing is the CALLER's: the
reference pairs pose slot l (``pos_funcs[l-1]``, pose_dict order) with the detections of ``markers[l-1]`` (the
skeleton's marker LIST order) - :288-292 with :113-128 - which are different orders for the shipped skeletons;
``reference_pairing`` reproduces that, ``name_pairing`` pairs by name.

Solved by the same projected Levenberg-Marquardt as oracle/fte.py (``lm_solve``); the L1 term enters the Gauss-Newton
curvature as IRLS weights w^2 / max(|e|, l1_eps) (e = scaled residual), the cost and the gradient are those of |e| itself.
"""
import numpy as np

from . import camera
from .fk import _rot
from .fte import FTEProblem, lm_solve  # noqa: F401  (lm_solve re-exported: the solver is shared)

MODEL_WEIGHT = 0.002        # build.py:186-191
R_MEAS = 3.0                # :142
LIK_THRESH = 0.4            # :187 (and :145 for the triangulation)


def _parts_and_dofs(skel):
    dofs = {k: list(v) for k, v in skel["dofs"].items()}
    for joint in skel["markers"]:                                    # :36-37
        dofs[joint] = [1, 1, 1]
    return list(dofs.keys()), dofs


def pose_names(skel):
    """Output order of pose_to_3d = insertion order of pose_dict (:70-83)."""
    names = []
    for link in skel["links"]:
        for part in (link[:1] if len(link) == 1 else link):
            if part not in names:
                names.append(part)
    return names


def active_states(skel):
    """0-based indices of the states the poses depend on: x, y, z and the enabled angles of every part that is the parent
    of a link (state layout [x y z | phi_0.. | theta_0.. | psi_0..], angle index = order of ``dofs``, :46-68)."""
    parts, dofs = _parts_and_dofs(skel)
    L = len(skel["positions"])
    parents = {link[0] for link in skel["links"] if len(link) == 2}
    act = [0, 1, 2]
    for axis in range(3):                                            # phi block, theta block, psi block
        for i, part in enumerate(parts):
            if part in parents and dofs[part][axis]:
                act.append(3 + axis * L + i)
    return np.array(sorted(act))


def skeleton_fk_jac(skel, q):
    """q[N, 3 + 3L] -> (positions[N, n_pose, 3], J[N, n_pose, 3, 3 + 3L]) with the bookkeeping of oracle/skeleton_fk.py
    (product rule through R_loc = Rz Rx Ry of the parent's own angles)."""
    q = np.atleast_2d(np.asarray(q, dtype=np.float64))
    links, positions = skel["links"], skel["positions"]
    parts, dofs = _parts_and_dofs(skel)
    L = len(positions)
    N, P = q.shape[0], 3 + 3 * L
    assert q.shape[1] == P
    eye = np.broadcast_to(np.eye(3), (N, 3, 3))
    rot, drot, transposed = {}, {}, {}
    for i, part in enumerate(parts):
        Ry, dRy = _rot("y", q[:, 3 + L + i]) if dofs[part][1] else (eye, None)
        Rx, dRx = _rot("x", q[:, 3 + i]) if dofs[part][0] else (eye, None)
        Rz, dRz = _rot("z", q[:, 3 + 2 * L + i]) if dofs[part][2] else (eye, None)
        rot[part] = Rz @ Rx @ Ry
        d = {}
        if dRx is not None:
            d[3 + i] = Rz @ dRx @ Ry
        if dRy is not None:
            d[3 + L + i] = Rz @ Rx @ dRy
        if dRz is not None:
            d[3 + 2 * L + i] = dRz @ Rx @ Ry
        drot[part] = d
        transposed[part] = True                                      # rot_dict[part + "_i"] = R^T  (:62)
    root = q[:, :3]
    droot = np.zeros((N, 3, P))
    droot[:, np.arange(3), np.arange(3)] = 1.0
    pose, dpose = {}, {}
    for link in links:
        if len(link) == 1:
            pose[link[0]], dpose[link[0]] = root, droot
            continue
        a, b = link
        if a not in pose:
            pose[a], dpose[a] = root, droot
        off = np.asarray(positions[b], dtype=np.float64) - np.asarray(positions[a], dtype=np.float64)
        transposed[b] = not transposed[b]                            # :76
        M = np.swapaxes(rot[a], 1, 2) if transposed[a] else rot[a]
        pose[b] = pose[a] + M @ off
        dj = dpose[a].copy()
        for col, dR in drot[a].items():
            dM = np.swapaxes(dR, 1, 2) if transposed[a] else dR
            dj[:, :, col] += dM @ off
        dpose[b] = dj
    names = list(pose.keys())
    return np.stack([pose[k] for k in names], 1), np.stack([dpose[k] for k in names], 1), names


def bounds_table(skel, n_frames):
    """lo, hi [N, P] of :270-273 (see the module docstring for the index quirks)."""
    L = len(skel["positions"])
    P = 3 + 3 * L
    lo = np.full((n_frames, P), -np.inf)
    hi = np.full((n_frames, P), np.inf)
    lo[:n_frames - 1, 2:3 * L - 1] = -np.pi / 2
    hi[:n_frames - 1, 2:3 * L - 1] = np.pi / 2
    return lo, hi


def reference_pairing(skel):
    """pose slot l <- detections of markers[l] (by POSITION in the skeleton's marker list; :113-128, :288-292); a slot whose
    paired marker is called "neck" has no measurement (:120-121, :289)."""
    names = pose_names(skel)
    markers = list(skel["markers"])
    return [(markers[l] if l < len(markers) and markers[l] != "neck" else None) for l in range(len(names))]


def name_pairing(skel):
    names = pose_names(skel)
    return [n if n in skel["markers"] and n != "neck" else None for n in names]


class SkelFTEProblem(FTEProblem):
    """meas[N, C, Lp, 2] and weights[N, C, Lp] are indexed by POSE SLOT (the caller has applied a pairing); weights are the
    reference's 1/R or 0."""

    def __init__(self, skel, meas, weights, K, D, R, t, h, lo=None, hi=None, model_weight=MODEL_WEIGHT, l1_eps=1e-2,
                 n_global=None, n_offset=0):
        self.skel = skel
        self.meas = np.asarray(meas, dtype=np.float64)
        self.N, self.C, self.L, _ = self.meas.shape
        self.names = pose_names(skel)
        assert self.L == len(self.names), "one measurement slot per pose"
        self.K = np.asarray(K, dtype=np.float64)
        self.D = np.asarray(D, dtype=np.float64).reshape(self.C, 4)
        self.R = np.asarray(R, dtype=np.float64)
        self.t = np.asarray(t, dtype=np.float64).reshape(self.C, 3)
        self.Ts = float(h)
        finite = np.isfinite(self.meas).all(-1)
        self.w = np.where(finite, np.asarray(weights, dtype=np.float64), 0.0)
        self.meas = np.where(finite[..., None], self.meas, 0.0)
        self.ACT = active_states(skel)
        self.P_full = 3 + 3 * len(skel["positions"])
        self.P = len(self.ACT)
        self.q_w = np.full(self.P, model_weight / self.Ts ** 4)
        blo, bhi = bounds_table(skel, self.N) if lo is None else (np.asarray(lo, float), np.asarray(hi, float))
        self.lo, self.hi = blo[:, self.ACT], bhi[:, self.ACT]
        self.l1_eps = float(l1_eps)
        self.n_global = self.N if n_global is None else int(n_global)
        self.n_offset = int(n_offset)

    def full_state(self, xa):
        q = np.zeros(xa.shape[:-1] + (self.P_full,))
        q[..., self.ACT] = xa
        return q

    def measurement_terms(self, xa, need_jac=True, chunk=2048, per_frame=False):
        N, P = xa.shape
        cost, cost_n = 0.0, np.zeros(N)
        g = np.zeros((N, P))
        H = np.zeros((N, P, P)) if need_jac else None
        n_behind = 0
        for s in range(0, N, chunk):
            sl = slice(s, min(N, s + chunk))
            pos, Jfk, _ = skeleton_fk_jac(self.skel, self.full_state(xa[sl]))
            G = Jfk[..., self.ACT]                                   # [n, L, 3, P]
            for ci in range(self.C):
                uv, Jpi, zc = camera.pt3d_to_2d(pos, self.K[ci], self.D[ci], self.R[ci], self.t[ci], with_jac=True)
                w = self.w[sl, ci]
                n_behind += int(((zc < 1e-6) & (w > 0)).sum())
                sing = np.abs(zc) < 1e-9                            # (the singular plane itself, as in oracle/fte.py)
                w = np.where(sing, 0.0, w)
                res = np.where(sing[..., None], 0.0, uv - self.meas[sl, ci])
                e = w[..., None] * res                               # scaled residual [n, L, 2]
                ae = np.abs(e)
                cost += float(ae.sum())
                if per_frame:
                    cost_n[sl] += ae.sum(axis=(1, 2))
                if need_jac:
                    J = np.einsum("nlij,nljp->nlip", Jpi, G)
                    g[sl] += np.einsum("nlip,nli->np", J, w[..., None] * np.sign(e))
                    hw = (w[..., None] ** 2) / np.maximum(ae, self.l1_eps)
                    H[sl] += np.einsum("nlip,nli,nliq->npq", J, hw, J)
        return (cost_n if per_frame else cost), g, H, n_behind

    def outputs(self, xa, x0_full=None):
        """The result dict of convert_to_dict (:343-365): positions, x, dx, ddx with all P columns."""
        q = np.zeros((self.N, self.P_full)) if x0_full is None else np.array(x0_full, dtype=np.float64, copy=True)
        q[:, self.ACT] = xa
        pos = skeleton_fk_jac(self.skel, q)[0]
        h, N = self.Ts, self.N
        dx, ddx = np.zeros_like(q), np.zeros_like(q)
        if N >= 2:
            dx[1:] = (q[1:] - q[:-1]) / h
        if N >= 3:
            ddx[2:] = (dx[2:] - dx[1:-1]) / h
            ddx[1] = ddx[0] = ddx[2]
            dx[0] = dx[1] - h * ddx[1]
        return dict(positions=pos, x=q, dx=dx, ddx=ddx)


def line_init(frames, xyz, n_frames, P_full, start_frame=0):
    """:149-166, :217-222: least-squares line through the triangulated "forehead" points; x, y, z from the line evaluated
    at 0 .. N-1 (``frame_est = np.arange(N)`` - NOT shifted by start_frame, as written), every angle 0."""
    f = np.asarray(frames, dtype=np.float64)
    A = np.stack([f, np.ones_like(f)], 1)
    coef, *_ = np.linalg.lstsq(A, np.asarray(xyz, dtype=np.float64), rcond=None)
    fe = np.arange(n_frames, dtype=np.float64)
    x0 = np.zeros((n_frames, P_full))
    x0[:, 0:3] = fe[:, None] * coef[0][None, :] + coef[1][None, :]
    return x0


def read_dlc_h5(path):
    """The shipped DeepLabCut tables (data/*.h5) without pytables: uncompressed PyTables records of one int64 index and
    3 K float64 values (x, y, likelihood per body part), contiguous from the first record.  Returns
    (index[N], values[N, K, 3]); the body-part order is that of the file's column index (read by the caller)."""
    raw = open(path, "rb").read()
    for n_parts in range(1, 64):
        rec = 8 + 24 * n_parts
        # the table is the largest region of the file that is a whole number of records starting with index 0, 1, 2, ...
        for off in range(0, 16384, 8):
            n = (len(raw) - off) // rec
            if n < 8:
                continue
            idx = np.frombuffer(raw, dtype="<i8", count=1, offset=off)[0]
            if idx != 0:
                continue
            v = np.frombuffer(raw[off:off + 3 * rec], dtype=np.uint8)
            i1 = np.frombuffer(raw, dtype="<i8", count=1, offset=off + rec)[0]
            i2 = np.frombuffer(raw, dtype="<i8", count=1, offset=off + 2 * rec)[0]
            if i1 == 1 and i2 == 2:
                # count the run of consecutive indices
                k = 0
                while off + (k + 1) * rec <= len(raw) and np.frombuffer(raw, dtype="<i8", count=1, offset=off + k * rec)[0] == k:
                    k += 1
                if k >= 100:
                    dt = np.dtype([("i", "<i8"), ("v", "<f8", (3 * n_parts,))])
                    tab = np.frombuffer(raw, dtype=dt, count=k, offset=off)
                    return tab["i"].copy(), tab["v"].reshape(k, n_parts, 3).copy()
    raise ValueError(f"{path}: no uncompressed DLC table found")
