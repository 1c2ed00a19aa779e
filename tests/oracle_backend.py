"""CPU stand-in for acinoset_amd.dist.HipBackend built on the numpy oracle (tests only).

Implements the same backend interface with dense linear algebra so that ShardedFTE's host logic
(shard plan, separator exchange, halo wiring, global accept/reject) can be exercised under gloo
without a GPU, and so the HIP shard kernels have an independent model to be compared with."""
import numpy as np
import torch

from oracle import fk
from oracle import fte as ofte

BS, NP = 80, 25
FIX_SCALE = 2.0 ** 70


class OracleBackend:
    def __init__(self, det_local, K, D, R, t, Ts, n_global, n_offset, rank, world, lam0=1e-3, ftol=1e-10, xtol=1e-10,
                 gtol=1e-8, dlc_thresh=0.5):
        det_local = np.asarray(det_local, dtype=np.float64)
        self.rank, self.world = rank, world
        self.pin_left, self.pin_right = rank > 0, rank + 1 < world
        self.prob = ofte.FTEProblem(det_local[..., :2], det_local[..., 2], K, D, R, t, Ts, dlc_thresh=dlc_thresh,
                                    n_global=n_global, n_offset=n_offset)
        self.N = det_local.shape[0]
        self.lam0, self.ftol, self.xtol, self.gtol = lam0, ftol, xtol, gtol
        self.device = torch.device("cpu")

    def new(self, *shape):
        return torch.zeros(shape, dtype=torch.float64)

    # ---- iterate handling ------------------------------------------------------------------------
    def load_x(self, x_local):
        x = np.clip(np.asarray(x_local, dtype=np.float64), self.prob.lo, self.prob.hi)
        self.x = [x.copy(), x.copy()]
        self.halo = [[None, None], [None, None]]
        self.ev = [None, None]
        self.cur = 0
        self.st = dict(cost=0.0, lam=self.lam0, nu=2.0, iter=0, accepted=0, status=0)
        self.partial = np.zeros(8)

    def _buf(self, which):
        return self.cur ^ which

    def export_edges(self, which, out):
        x = self.x[self._buf(which)]
        out[0:3] = torch.as_tensor(x[:3])
        out[3:6] = torch.as_tensor(x[-3:])

    def set_halo(self, which, left, right):
        conv = lambda h: None if h is None else np.asarray(h, dtype=np.float64).copy()
        self.halo[self._buf(which)] = [conv(left), conv(right)]

    def eval(self, which):
        if which == 1 and self.st["status"] != 0:
            return
        b = self._buf(which)
        hl, hr = self.halo[b]
        F, g, H, nb = self.prob.evaluate(self.x[b], halo_l=hl, halo_r=hr)
        self.ev[b] = (F, g, H)
        self.partial[0] = F
        self.partial[4] = nb
        if which == 0:
            self.partial[1:4] = 0

    def export_partials(self, out):
        out[:] = torch.as_tensor(self.partial)

    def control(self, total, init):
        total = np.asarray(total, dtype=np.float64)
        st = self.st
        if st["status"] != 0:
            return
        if init:
            st["cost"] = total[0]
            return
        F, Ft, pred, step, gnorm = st["cost"], total[0], total[1], total[2], total[3]
        st["iter"] += 1
        if gnorm <= self.gtol:
            st["status"] = 3
            return
        gain = (F - Ft) / pred if pred > 0 else -1.0
        if Ft < F:
            dF = F - Ft
            self.cur ^= 1
            st["cost"] = Ft
            st["accepted"] += 1
            st["lam"] *= max(1.0 / 3.0, 1.0 - (2.0 * gain - 1.0) ** 3)
            st["nu"] = 2.0
            if dF <= self.ftol * abs(Ft):
                st["status"] = 1
            elif step <= self.xtol:
                st["status"] = 2
        else:
            st["lam"] *= st["nu"]
            st["nu"] *= 2.0
            if st["lam"] > 1e16:
                st["status"] = 4

    # ---- local reduction onto the separators -------------------------------------------------------
    def _fixed(self):
        x = self.x[self.cur]
        _F, g, H = self.ev[self.cur]
        return self.prob.active_set(x, g, H)

    def reduce_local(self):
        if self.st["status"] != 0:
            return
        P, N, prob = NP, self.N, self.prob
        x = self.x[self.cur]
        F, g, H = self.ev[self.cur]
        lam = self.st["lam"]
        fixed = self._fixed()
        nl = 3 if self.pin_left else 0                       # separator frames of the left rank come first
        n_tot = (nl + N) * P
        A = np.zeros((n_tot, n_tot))
        b = np.zeros(n_tot)
        ng, off = prob.n_global, prob.n_offset

        def band(n_glob, k):
            if n_glob < 0 or n_glob + k >= ng:
                return 0.0
            tot = 0.0
            for j in range(max(0, n_glob + k - 3), min(n_glob, ng - 4) + 1):
                tot += ofte.C3[3 - (n_glob - j)] * ofte.C3[3 - (n_glob + k - j)]
            return tot

        self.diag0 = np.zeros((N, P))
        for i in range(N):
            r0 = (nl + i) * P
            A[r0:r0 + P, r0:r0 + P] = H[i]
            d0 = np.diag(H[i]) + 2 * prob.q_w * band(off + i, 0)
            self.diag0[i] = d0
            dd = d0 * (1 + lam)
            dd = np.where(fixed[i], dd * FIX_SCALE, dd)
            A[r0 + np.arange(P), r0 + np.arange(P)] = dd
            b[r0:r0 + P] = np.where(fixed[i], 0.0, -g[i])
        # couplings between frames (own-own and own-left separator)
        for i in range(-nl, N):
            for k in range(1, 4):
                j = i + k
                if j < 0 or j >= N:
                    continue
                if i < 0 and j < 0:
                    continue                                  # separator-internal coupling belongs to its owner
                v = 2 * prob.q_w * band(off + i, k)
                ri, rj = (nl + i) * P, (nl + j) * P
                A[ri + np.arange(P), rj + np.arange(P)] += v
                A[rj + np.arange(P), ri + np.arange(P)] += v
        self.gn = float(np.abs(np.where(fixed, 0.0, g)).max()) if N else 0.0
        sep_idx = []
        if self.pin_left:
            sep_idx += list(range(0, 3 * P))
        if self.pin_right:
            sep_idx += list(range((nl + N - 3) * P, (nl + N) * P))
        sep_idx = np.array(sep_idx, dtype=int)
        int_idx = np.setdiff1d(np.arange(n_tot), sep_idx)
        self._A, self._b, self._sep_idx, self._int_idx, self._nl = A, b, sep_idx, int_idx, nl
        AII = A[np.ix_(int_idx, int_idx)]
        self._AII_inv_b = np.linalg.solve(AII, b[int_idx])
        if sep_idx.size:
            AIS = A[np.ix_(int_idx, sep_idx)]
            self._AII_inv_AIS = np.linalg.solve(AII, AIS)
            self._S = A[np.ix_(sep_idx, sep_idx)] - AIS.T @ self._AII_inv_AIS
            self._bS = b[sep_idx] - AIS.T @ self._AII_inv_b

    def export_separators(self, sep):
        if self.st["status"] != 0:
            return
        P3 = 3 * NP
        S, bS = self._S, self._bS
        o = 0
        if self.pin_left:
            rec = sep[self.rank - 1].numpy()
            D = np.zeros((BS, BS))
            D[:P3, :P3] = S[:P3, :P3]
            rec[:BS * BS] = D.reshape(-1)
            rec[2 * BS * BS:2 * BS * BS + P3] = bS[:P3]
            o = P3
        if self.pin_right:
            rec = sep[self.rank].numpy()
            D = np.eye(BS)
            D[:P3, :P3] = S[o:o + P3, o:o + P3]
            rec[:BS * BS] = D.reshape(-1)
            rec[2 * BS * BS:2 * BS * BS + P3] = bS[o:o + P3]
            if self.pin_left:                                 # coupling block(right sep, left sep) -> record of the LEFT separator's C slot
                Cm = np.zeros((BS, BS))
                Cm[:P3, :P3] = S[o:o + P3, :P3]
                sep[self.rank - 1].numpy()[BS * BS:2 * BS * BS] = Cm.reshape(-1)

    def solve_separators(self, sep, sep_x):
        if self.st["status"] != 0:
            return
        n_sep = self.world - 1
        s = sep.numpy()
        A = np.zeros((n_sep * BS, n_sep * BS))
        b = np.zeros(n_sep * BS)
        for k in range(n_sep):
            A[k * BS:(k + 1) * BS, k * BS:(k + 1) * BS] = s[k, :BS * BS].reshape(BS, BS)
            b[k * BS:(k + 1) * BS] = s[k, 2 * BS * BS:]
            if k + 1 < n_sep:
                Cm = s[k + 1, BS * BS:2 * BS * BS].reshape(BS, BS) * 0 + s[k, BS * BS:2 * BS * BS].reshape(BS, BS) * 0
        # coupling records: rec[k].C written by rank k+1 = block(sep k+1, sep k)
        for k in range(n_sep - 1):
            Cm = s[k, BS * BS:2 * BS * BS].reshape(BS, BS)
            A[(k + 1) * BS:(k + 2) * BS, k * BS:(k + 1) * BS] = Cm
            A[k * BS:(k + 1) * BS, (k + 1) * BS:(k + 2) * BS] = Cm.T
        x = np.linalg.solve(A, b)
        sep_x[:] = torch.as_tensor(x.reshape(n_sep, BS))

    def backsub_local(self, sep_x):
        if self.st["status"] != 0:
            return
        P3 = 3 * NP
        xs = []
        if self.pin_left:
            xs.append(sep_x[self.rank - 1].numpy()[:P3])
        if self.pin_right:
            xs.append(sep_x[self.rank].numpy()[:P3])
        full = np.zeros(self._A.shape[0])
        if xs:
            xS = np.concatenate(xs)
            full[self._sep_idx] = xS
            full[self._int_idx] = self._AII_inv_b - self._AII_inv_AIS @ xS
        else:
            full[self._int_idx] = self._AII_inv_b
        self.delta = full[self._nl * NP:].reshape(self.N, NP)

    def trial(self):
        if self.st["status"] != 0:
            return
        x = self.x[self.cur]
        g = self.ev[self.cur][1]
        fixed = self._fixed()
        pg = np.where(fixed, 0.0, g)
        d = self.delta
        xt = np.clip(x + d, self.prob.lo, self.prob.hi)
        self.x[self.cur ^ 1] = xt
        self.partial[1] = 0.5 * float((d * (self.st["lam"] * self.diag0 * d - pg)).sum())
        self.partial[2] = float(np.abs(xt - x).max())
        self.partial[3] = self.gn

    def state(self):
        return dict(self.st, cur=self.cur)

    def result_x(self):
        return torch.as_tensor(self.x[self.cur])


class OracleWindowBackend:
    """CPU stand-in for acinoset_amd.dist.HipWindowBackend (overlapping-window sharding) on the numpy oracle: the window
    [n_offset, n_offset + n) of the sequence is assembled and solved as a principal submatrix of the global system
    (step pinned to 0 outside it), only the owned frames [own_first, own_first + own_count) enter the sums."""

    def __init__(self, det_window, K, D, R, t, Ts, n_global, n_offset, own_first, own_count, lam0=1e-3, ftol=1e-10,
                 xtol=1e-10, gtol=1e-8, dlc_thresh=0.5):
        det_window = np.asarray(det_window, dtype=np.float64)
        self.prob = ofte.FTEProblem(det_window[..., :2], det_window[..., 2], K, D, R, t, Ts, dlc_thresh=dlc_thresh,
                                    n_global=n_global, n_offset=n_offset)
        self.N, self.n_global, self.n_offset = det_window.shape[0], n_global, n_offset
        self.own = slice(own_first, own_first + own_count)
        self.lam0, self.ftol, self.xtol, self.gtol = lam0, ftol, xtol, gtol
        self.device = torch.device("cpu")

    def new(self, *shape):
        return torch.zeros(shape, dtype=torch.float64)

    def load_x(self, x_window):
        x = np.clip(np.asarray(x_window, dtype=np.float64), self.prob.lo, self.prob.hi)
        buf = np.zeros((self.N + 6, NP))                # three stencil rows on either side, as the device buffers
        buf[3:-3] = x
        self.x = [buf.copy(), buf.copy()]
        self.ev = [None, None]
        self.cur = 0
        self.st = dict(cost=0.0, lam=self.lam0, nu=2.0, iter=0, accepted=0, status=0)
        self.partial = np.zeros(8)

    def _buf(self, which):
        return self.cur ^ which

    def copy_frames(self, which, imp, first, n, buf):
        x = self.x[self._buf(which)]
        if imp:
            x[first + 3:first + 3 + n] = np.asarray(buf, dtype=np.float64)
        else:
            buf[:] = torch.as_tensor(x[first + 3:first + 3 + n])

    def eval(self, which):
        if which == 1 and self.st["status"] != 0:
            return
        b = self._buf(which)
        X = self.x[b]
        hl = X[:3] if self.n_offset > 0 else None
        hr = X[-3:] if self.n_offset + self.N < self.n_global else None
        Fn, g, H, nb = self.prob.evaluate(X[3:-3], halo_l=hl, halo_r=hr, per_frame=True)
        self.ev[b] = (g, H)
        self.partial[0] = float(Fn[self.own].sum())
        self.partial[4] = 0
        if which == 0:
            self.partial[1:4] = 0

    def export_partials(self, out):
        out[:] = torch.as_tensor(self.partial)

    control = OracleBackend.control

    def solve_and_trial(self):
        if self.st["status"] != 0:
            return
        X = self.x[self.cur]
        x = X[3:-3]
        g, H = self.ev[self.cur]
        fixed = self.prob.active_set(x, g, H)
        delta, diag = self.prob.solve_banded(H, g, self.st["lam"], fixed)
        pg = np.where(fixed, 0.0, g)
        xt = np.clip(x + delta, self.prob.lo, self.prob.hi)
        T = self.x[self.cur ^ 1]
        T[3:-3] = xt
        o = self.own
        self.partial[1] = 0.5 * float((delta[o] * (self.st["lam"] * diag[o] * delta[o] - pg[o])).sum())
        self.partial[2] = float(np.abs(xt[o] - x[o]).max())
        self.partial[3] = float(np.abs(pg[o]).max())

    def state(self):
        return dict(self.st, cur=self.cur)

    def result_owned(self):
        return torch.as_tensor(self.x[self.cur][3:-3][self.own])
