"""Oracle Full-Trajectory-Estimation (numpy/scipy fp64).  Test infrastructure.

Restates the reference NLP of src/all_optimizations.py:283-500 in its equivalent reduced
form (the elimination is pinned by KAT-4):

    min_x  sum_{n,c,l,d} rho( w_ncl * (pi_c(FK_l(x_n))_d - z_ncld) )
         + sum_{n>=3} sum_p (w_p / Ts^4) (x_n - 3 x_{n-1} + 3 x_{n-2} - x_{n-3})_p^2
    s.t.   lo <= x_n <= hi   (the 21 angle boxes of all_optimizations.py:403-483)

solved by a projected Levenberg-Marquardt on the block-banded Gauss-Newton system
(IRLS weights rho'(e)/e, Marquardt scaling, Nielsen's lambda update) - the SAME algorithm,
step for step, as the HIP path (acinoset_amd/fte.py), so trajectories can be compared
iterate by iterate.  IPOPT itself (tol=1e-1, L-BFGS) is absent; see oracle/__init__.py.
"""
import numpy as np
from scipy.linalg import solveh_banded

from . import camera, fk, loss

C3 = np.array([-1.0, 3.0, -3.0, 1.0])   # third-difference stencil (x_n - 3x_{n-1} + 3x_{n-2} - x_{n-3})


class FTEProblem:
    """Holds inputs in the dense layout of the drop-in boundary."""

    def __init__(self, meas, likelihood, K, D, R, t, Ts, dlc_thresh=0.5, R_meas=5.0, Q=None,
                 redesc=(3.0, 10.0, 20.0), n_global=None, n_offset=0):
        self.meas = np.asarray(meas, dtype=np.float64)            # [N,C,L,2]
        lik = np.asarray(likelihood, dtype=np.float64)            # [N,C,L]
        self.N, self.C, self.L, _ = self.meas.shape
        self.K = np.asarray(K, dtype=np.float64)
        self.D = np.asarray(D, dtype=np.float64).reshape(self.C, 4)
        self.R = np.asarray(R, dtype=np.float64)
        self.t = np.asarray(t, dtype=np.float64).reshape(self.C, 3)
        self.Ts = float(Ts)
        finite = np.isfinite(self.meas).all(-1)
        # binary weights (all_optimizations.py:302-308): 1/R if likelihood > thresh else 0
        self.w = np.where((lik > dlc_thresh) & finite, 1.0 / R_meas, 0.0)
        self.meas = np.where(finite[..., None], self.meas, 0.0)
        Qs = fk.Q_SIGMA ** 2 if Q is None else np.asarray(Q, dtype=np.float64)
        wq = np.where(Qs != 0.0, 1.0 / np.where(Qs != 0, Qs, 1.0), 0.0)   # :310-315
        self.q_w = (wq / self.Ts ** 4)[fk.ACTIVE]                           # per active state
        self.redesc = tuple(float(v) for v in redesc)
        lo, hi = fk.bounds45()
        self.lo, self.hi = lo[fk.ACTIVE], hi[fk.ACTIVE]
        self.n_global = self.N if n_global is None else int(n_global)
        self.n_offset = int(n_offset)
        self.P = len(fk.ACTIVE)

    # ------------------------------------------------------------------ pieces
    def full_state(self, xa):
        q = np.zeros(xa.shape[:-1] + (fk.N_STATES,))
        q[..., fk.ACTIVE] = xa
        return q

    def measurement_terms(self, xa, need_jac=True, chunk=2048, per_frame=False):
        """Returns (cost, g[N,P], H[N,P,P], n_behind) of the measurement term (per_frame: cost as an array [N])."""
        a, b, c = self.redesc
        N, P = xa.shape
        cost = 0.0
        cost_n = np.zeros(N)
        g = np.zeros((N, P))
        H = np.zeros((N, P, P)) if need_jac else None
        n_behind = 0
        for s in range(0, N, chunk):
            sl = slice(s, min(N, s + chunk))
            q = self.full_state(xa[sl])
            if need_jac:
                pos, Jfk = fk.cheetah_fk(q, with_jac=True)
                G = Jfk[..., fk.ACTIVE]                       # [n,L,3,P]
            else:
                pos = fk.cheetah_fk(q)
            for ci in range(self.C):
                if need_jac:
                    uv, Jpi, zc = camera.pt3d_to_2d(pos, self.K[ci], self.D[ci], self.R[ci], self.t[ci], with_jac=True)
                else:
                    uv = camera.pt3d_to_2d(pos, self.K[ci], self.D[ci], self.R[ci], self.t[ci])
                    zc = (pos @ self.R[ci][2]) + self.t[ci][2]
                w = self.w[sl, ci]                            # [n,L]
                # the reference's pt3d_to_2d (:193-209) has NO cut at z_cam <= 0: a marker behind a camera keeps its
                # (mirrored) projection and is penalised like any other residual.  Only the singular plane itself
                # (|z_cam| < 1e-9, where x/z is undefined) is dropped; n_behind counts weighted detections with
                # z_cam < 1e-6 as a diagnostic.
                n_behind += int(((zc < 1e-6) & (w > 0)).sum())
                sing = np.abs(zc) < 1e-9
                w = np.where(sing, 0.0, w)
                res = np.where(sing[..., None], 0.0, uv - self.meas[sl, ci])
                sres = w[..., None] * res                     # scaled residual [n,L,2]
                rho, drho, h = loss.redescending_dloss(sres, a, b, c)
                cost += float(rho.sum())
                if per_frame:
                    cost_n[sl] += rho.sum(axis=(1, 2))
                if need_jac:
                    J = np.einsum("nlij,nljp->nlip", Jpi, G)  # [n,L,2,P]
                    gs = w[..., None] * drho * np.sign(sres)  # d rho / d(raw residual)
                    g[sl] += np.einsum("nlip,nli->np", J, gs)
                    hw = (w[..., None] ** 2) * h
                    H[sl] += np.einsum("nlip,nli,nliq->npq", J, hw, J)
        return (cost_n if per_frame else cost), g, H, n_behind

    def s_band(self):
        """(D3^T D3)[n, n+k], k=0..3, for LOCAL frames, honouring global sequence ends."""
        N, NG, off = self.N, self.n_global, self.n_offset
        if getattr(self, "_band", None) is not None:
            return self._band
        band = np.zeros((4, N))
        for k in range(4):
            for i in range(N):
                n = i + off
                if n + k >= NG:
                    continue
                jlo, jhi = max(0, n + k - 3), min(n, NG - 4)
                tot = 0.0
                for j in range(jlo, jhi + 1):
                    tot += C3[3 - (n - j)] * C3[3 - (n + k - j)]
                band[k, i] = tot
        self._band = band
        return band

    def smooth_terms(self, xa, halo_l=None, halo_r=None, per_frame=False):
        """Third-difference smoothness: cost and gradient on local frames.

        Stencil row j couples frames j..j+3: d_j = -x_j + 3x_{j+1} - 3x_{j+2} + x_{j+3}
        (= x_n - 3x_{n-1} + 3x_{n-2} - x_{n-3} with n=j+3).  Rows are owned by the rank owning
        frame j+3... here (single shard or halo-extended) every row touching a local frame
        contributes to that frame's gradient; cost counts rows whose LAST frame is local.
        """
        N = self.N
        pads_l = halo_l if halo_l is not None else np.zeros((0, self.P))
        pads_r = halo_r if halo_r is not None else np.zeros((0, self.P))
        X = np.vstack([pads_l, xa, pads_r])
        o = pads_l.shape[0]
        M = X.shape[0]
        if M < 4:
            return 0.0, np.zeros_like(xa)
        d = -X[:-3] + 3 * X[1:-2] - 3 * X[2:-1] + X[3:]                 # rows j=0..M-4 (extended index)
        qd = d * self.q_w
        gX = np.zeros_like(X)
        gX[:-3] += -2 * qd
        gX[1:-2] += 6 * qd
        gX[2:-1] += -6 * qd
        gX[3:] += 2 * qd
        last = np.arange(3, M)                                          # last frame of each row
        own = (last >= o) & (last < o + N)
        if per_frame:                                                   # cost of the row whose LAST frame is local frame i
            cost_n = np.zeros(N)
            cost_n[last[own] - o] = (qd[own] * d[own]).sum(axis=1)
            return cost_n, gX[o:o + N]
        cost = float((qd[own] * d[own]).sum())
        return cost, gX[o:o + N]

    def evaluate(self, xa, need_jac=True, halo_l=None, halo_r=None, per_frame=False):
        cm, g, H, nb = self.measurement_terms(xa, need_jac, per_frame=per_frame)
        cs, gs = self.smooth_terms(xa, halo_l, halo_r, per_frame=per_frame)
        return cm + cs, g + gs, H, nb

    # ------------------------------------------------------------------ active set
    GRAD_ZERO_REL = 1e-14

    def active_set(self, x, g, H):
        """Bound-active variables: at a bound with the gradient pushing outward.  A gradient entry below 1e-14 * H_ii
        (a Newton step of 1e-14 rad / m in that variable alone) counts as ZERO: an entry that vanishes analytically -
        a joint none of whose markers is detected in that frame - is exactly 0.0 here and +-1e-20 in the HIP kernel's
        subtree sums, and a bare sign test would make the active set depend on that rounding noise."""
        idx = np.arange(self.P)
        diag = H[:, idx, idx] + 2 * self.q_w[None, :] * self.s_band()[0][:, None]
        tol = self.GRAD_ZERO_REL * diag
        return ((x <= self.lo) & (g > tol)) | ((x >= self.hi) & (g < -tol))

    # ------------------------------------------------------------------ linear algebra
    def solve_banded(self, H, g, lam, fixed):
        """Solve (H_gn + lam*diag(H_gn)) delta = -g over the whole (local = global) sequence,
        H_gn = blockdiag(H_n) + 2 q (x) D3^T D3, with `fixed` variables pinned to delta=0."""
        N, P = g.shape
        band = self.s_band()
        n_tot = N * P
        bw = 3 * P + P - 1
        ab = np.zeros((bw + 1, n_tot))                                   # LAPACK lower banded
        Hd = H.copy()
        idx = np.arange(P)
        Hd[:, idx, idx] += 2 * self.q_w[None, :] * band[0][:, None]
        # (a diagonal entry that is exactly 0 - a state no camera observes in a clip of < 4 frames, which has no
        #  third-difference row either - gets lam * 1e-30: decoupled variable, positive pivot, step exactly 0)
        diag = np.maximum(Hd[:, idx, idx], 1e-30)
        Hd[:, idx, idx] += lam * diag
        fx = fixed
        Hd = np.where(fx[:, :, None] | fx[:, None, :], 0.0, Hd)
        Hd[:, idx, idx] = np.where(fx, 1.0, Hd[:, idx, idx])
        for r in range(P):
            for cc in range(r + 1):
                ab[r - cc, cc::P][:N] = Hd[:, r, cc]
        for k in range(1, 4):
            v = (2 * self.q_w[None, :] * band[k][:, None])               # coupling (n, n+k), diagonal in p
            v[:N - k] = np.where(fx[:N - k] | fx[k:], 0.0, v[:N - k])
            flat = v.reshape(-1)
            ab[k * P, :n_tot - k * P] = flat[:n_tot - k * P]
        rhs = np.where(fx, 0.0, -g).reshape(-1)
        delta = solveh_banded(ab, rhs, lower=True, check_finite=False)
        return delta.reshape(N, P), diag


def nose_line_init(tri_nose_frames, tri_nose_xyz, n_frames, start_frame=0):
    """Initial guess of src/all_optimizations.py:268-277,333-337: least-squares line through the
    triangulated nose positions, psi_0 = atan2(y_slope, x_slope), everything else 0."""
    f = np.asarray(tri_nose_frames, dtype=np.float64)
    A = np.stack([f, np.ones_like(f)], 1)
    coef, *_ = np.linalg.lstsq(A, np.asarray(tri_nose_xyz, dtype=np.float64), rcond=None)
    slope, icpt = coef[0], coef[1]
    frames = np.arange(start_frame, start_frame + n_frames, dtype=np.float64)
    x0 = np.zeros((n_frames, fk.N_STATES))
    x0[:, 0:3] = frames[:, None] * slope[None, :] + icpt[None, :]
    x0[:, 31] = np.arctan2(slope[1], slope[0])
    return x0


def lm_solve(prob, x0_active, max_iter=50, lam0=1e-3, ftol=1e-10, xtol=1e-10, gtol=1e-8, verbose=False,
             history=None):
    """Projected LM, identical control flow to the HIP path's device-side controller."""
    lo, hi = prob.lo, prob.hi
    x = np.clip(np.asarray(x0_active, dtype=np.float64), lo, hi)
    F, g, H, nb = prob.evaluate(x)
    lam, nu = lam0, 2.0
    status = "max_iter"
    it = 0
    n_acc = 0
    for it in range(1, max_iter + 1):
        fixed = prob.active_set(x, g, H)
        pg = np.where(fixed, 0.0, g)
        gnorm = float(np.abs(pg).max())
        if gnorm <= gtol:
            status = "gtol"
            break
        delta, diag = prob.solve_banded(H, g, lam, fixed)
        xt = np.clip(x + delta, lo, hi)
        Ft, gt, Ht, nbt = prob.evaluate(xt)
        pred = 0.5 * float((delta * (lam * diag * delta - pg)).sum())
        gain = (F - Ft) / pred if pred > 0 else -1.0
        step = float(np.abs(xt - x).max())
        if history is not None:
            history.append(dict(it=it, F=F, Ft=Ft, lam=lam, gain=gain, step=step, gnorm=gnorm))
        if verbose:
            print(f"it {it:3d} F={F:.9e} Ft={Ft:.9e} lam={lam:.2e} gain={gain:+.3f} step={step:.2e} |g|={gnorm:.2e}")
        if Ft < F:
            dF = F - Ft
            x, F, g, H, nb = xt, Ft, gt, Ht, nbt
            n_acc += 1
            lam = lam * max(1.0 / 3.0, 1.0 - (2.0 * gain - 1.0) ** 3)
            nu = 2.0
            if dF <= ftol * abs(F):
                status = "ftol"
                break
            if step <= xtol:
                status = "xtol"
                break
        else:
            lam *= nu
            nu *= 2.0
            if lam > 1e16:
                status = "lambda_overflow"
                break
    return x, dict(cost=F, iterations=it, accepted=n_acc, status=status, lam=lam, n_behind=nb,
                   gnorm=float(np.abs(np.where(prob.active_set(x, g, H), 0.0, g)).max()))


def fte_outputs(prob, x_active, x0_full, start_frame=0):
    """Result dict of src/all_optimizations.py:530-559: positions[N,20,3], x/dx/ddx [N,25]."""
    q = np.array(x0_full, dtype=np.float64, copy=True)
    q[:, fk.ACTIVE] = x_active
    pos = fk.cheetah_fk(q)
    Ts = prob.Ts
    x = x_active
    N = x.shape[0]
    dx = np.zeros_like(x)
    ddx = np.zeros_like(x)
    if N >= 2:
        dx[1:] = (x[1:] - x[:-1]) / Ts
    if N >= 3:
        ddx[2:] = (dx[2:] - dx[1:-1]) / Ts
        ddx[1] = ddx[2]
        ddx[0] = ddx[2]
        dx[0] = dx[1] - Ts * ddx[1]
    return dict(positions=pos, x=x.copy(), dx=dx, ddx=ddx, start_frame=start_frame)
