"""Frame-sharded FTE across the GPUs of one node (one process per GPU, torch.distributed over RCCL).

The reference has no parallelism at all; this is new.  The Gauss-Newton system of
the FTE is block-tridiagonal in super-blocks of 3 frames, so contiguous frame blocks shard naturally:

  * rank g owns super-blocks [m0_g, m1_g); the LAST super-block of every rank but the last is a
    *separator* the rank assembles but does not eliminate, and every rank but the first also sees its
    left neighbour's separator as chain node 0;
  * per LM iteration each rank reduces its interior onto its (<= 2) separators, ONE all-reduce(SUM)
    over xGMI carries the separator system ("temporal-coupling rows": (world-1) x (2*80*80+80) doubles,
    ~0.7 MB at 8 GPUs), every rank solves that tiny chain redundantly and back-substitutes locally;
  * two small all-gathers carry the 3+3 edge frames of the trial iterate (the third-difference stencil
    reaches 3 frames across a boundary) and the 8 scalars of the accept/reject decision, which every
    rank then takes identically on its own device.

The numerical work lives behind a small backend interface; ``HipBackend`` drives libacinoset_hip.so.
(tests/ plug a CPU oracle backend into the same driver to exercise this host logic under gloo.)
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib, fte
from ._lib import BS, N_ACTIVE, SEP_DOUBLES, check, lib, ptr, stream_ptr


def shard_plan(n_frames, world):
    """Frame ranges [(n0, n1)] per rank: boundaries at multiples of 3, super-blocks split evenly."""
    n_sb = (n_frames + 2) // 3
    if world < 1 or n_sb < 2 * world:
        raise ValueError(f"sequence of {n_frames} frames is too short to shard over {world} ranks "
                         "(each rank needs >= 2 super-blocks of 3 frames)")
    bounds = [(n_sb * g) // world for g in range(world + 1)]
    return [(3 * bounds[g], min(3 * bounds[g + 1], n_frames)) for g in range(world)]


class HipBackend:
    """Local shard on this process's GPU: thin calls into the C ABI (include/acinoset_hip.h)."""

    def __init__(self, det_local, k_arr, d_arr, r_arr, t_arr, Ts, n_global, n_offset, rank, world, **kw):
        self.rank, self.world = rank, world
        self.ctx = fte.FTEContext(det_local, k_arr, d_arr, r_arr, t_arr, Ts, n_global=n_global, n_offset=n_offset,
                                  pin_left=rank > 0, pin_right=rank + 1 < world, **kw)
        self.device = self.ctx.device
        self.n_sep = world - 1
        if self.n_sep > 0:
            nb = lib().acino_sep_scratch_bytes(self.n_sep)
            self._scratch = torch.empty(nb + 256, dtype=torch.uint8, device=self.device)
            self._scratch_ptr = (self._scratch.data_ptr() + 255) // 256 * 256
            self._scratch_bytes = nb

    def new(self, *shape):
        return torch.zeros(shape, dtype=torch.float64, device=self.device)

    def load_x(self, x_local):
        self._x0 = x_local.to(self.device, torch.float64).contiguous()
        check(lib().acino_fte_load_x(self.ctx._h, ptr(self._x0), stream_ptr()))

    def export_edges(self, which, out):
        check(lib().acino_fte_export_edges(self.ctx._h, which, ptr(out), stream_ptr()))

    def set_halo(self, which, left, right):
        check(lib().acino_fte_set_halo(self.ctx._h, which, ptr(left), ptr(right), stream_ptr()))

    def eval(self, which):
        check(lib().acino_fte_eval(self.ctx._h, which, stream_ptr()))

    def export_partials(self, out):
        check(lib().acino_fte_export_partials(self.ctx._h, ptr(out), stream_ptr()))

    def control(self, total, init):
        check(lib().acino_fte_control(self.ctx._h, ptr(total), int(init), stream_ptr()))

    def reduce_local(self):
        check(lib().acino_fte_reduce_local(self.ctx._h, stream_ptr()))

    def export_separators(self, sep):
        check(lib().acino_fte_export_separators(self.ctx._h, ptr(sep), self.rank, self.world, stream_ptr()))

    def solve_separators(self, sep, sep_x):
        check(lib().acino_solve_separators(ptr(sep), self.n_sep, ptr(sep_x), C.c_void_p(self._scratch_ptr),
                                           self._scratch_bytes, stream_ptr()))

    def backsub_local(self, sep_x):
        check(lib().acino_fte_backsub_local(self.ctx._h, ptr(sep_x) if sep_x is not None else C.c_void_p(0),
                                            self.rank, self.world, stream_ptr()))

    def trial(self):
        check(lib().acino_fte_trial(self.ctx._h, stream_ptr()))

    def state(self):
        return self.ctx.state()

    def result_x(self):
        return self.ctx.result()[0]

    # sharded iteration as four fused phases (each one C-ABI call = one hipGraph replay when graphs are enabled)
    fused_phases = True

    def phase_reduce(self, sep):
        check(lib().acino_fte_shard_reduce(self.ctx._h, ptr(sep), self.rank, self.world, stream_ptr()))

    def phase_solve(self, sep, sep_x, edges_out):
        check(lib().acino_fte_shard_solve(self.ctx._h, ptr(sep), ptr(sep_x), C.c_void_p(self._scratch_ptr),
                                          self._scratch_bytes, ptr(edges_out), self.rank, self.world, stream_ptr()))

    def phase_eval(self, which, all_edges, partial_out):
        check(lib().acino_fte_shard_eval(self.ctx._h, which, ptr(all_edges), self.rank, self.world, ptr(partial_out),
                                         stream_ptr()))

    def phase_control(self, all_partials, init):
        check(lib().acino_fte_shard_control(self.ctx._h, ptr(all_partials), self.world, int(init), stream_ptr()))

    def enable_graph(self, on=True):
        self.ctx.enable_graph(on)

    # unsharded fast path: the whole iteration is one C-ABI call (and one hipGraph replay when enabled)
    def set_x_single(self, x_local):
        self.ctx.set_x(x_local)

    def step_single(self):
        self.ctx.step()


class TorchComm:
    """torch.distributed collectives: backend "nccl" (= RCCL over xGMI) on the GPU node.  With the "gloo"
    backend (CPU tests, or several ranks sharing one GPU when no multi-GPU node is at hand) device tensors
    are staged through the host."""

    def __init__(self, group=None):
        self.group = group
        self._stage = dist.is_initialized() and dist.get_backend(group) == "gloo"

    def all_gather(self, out, inp):
        # flat views: gloo only accepts the concatenated 1-D layout, RCCL accepts both
        if self._stage and inp.is_cuda:
            o, i = out.cpu().view(-1), inp.detach().cpu().contiguous().view(-1)
            dist.all_gather_into_tensor(o, i, group=self.group)
            out.view(-1).copy_(o)
        else:
            dist.all_gather_into_tensor(out.view(-1), inp.contiguous().view(-1), group=self.group)

    def all_reduce_sum(self, t):
        if self._stage and t.is_cuda:
            h = t.detach().cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)


def combine_partials(gathered):
    """[world, 8] local sums -> global {cost, pred, step_inf, gnorm_inf, n_behind, numeric flags, 0, 0}; fixed (rank)
    order, so every rank computes bit-identical totals."""
    total = torch.zeros(8, dtype=gathered.dtype, device=gathered.device)
    total[0] = gathered[:, 0].sum()
    total[1] = gathered[:, 1].sum()
    total[2] = gathered[:, 2].max()
    total[3] = gathered[:, 3].max()
    total[4] = gathered[:, 4].sum()
    # numeric flags (bit 0 pivot, 1 sync timeout, 2 truncation) are a bit MASK: OR over the ranks, exactly as the device
    # controller combines them (k_control_gathered) - a max would let rank A (bit 0) and rank B (bit 2) derive different
    # statuses from the same totals and part ways in the next collective
    flags = 0
    for v in gathered[:, 5].tolist():
        flags |= int(v) if v > 0 else 0
    total[5] = float(flags)
    return total


class ShardedFTE:
    """One LM solve over a sequence sharded across the process group (strong scaling)."""

    def __init__(self, backend, rank, world, group=None, comm=None):
        self.b, self.rank, self.world, self.group = backend, rank, world, group
        self.comm = comm if comm is not None else TorchComm(group)
        self._edges = backend.new(6, N_ACTIVE)
        self._all_edges = backend.new(world, 6, N_ACTIVE)
        self._partial = backend.new(8)
        self._all_partials = backend.new(world, 8)
        self._sep = backend.new(max(world - 1, 1), SEP_DOUBLES)
        self._sep_x = backend.new(max(world - 1, 1), BS)
        self._timing = None

    # -- per-collective timing (bench.py --gpus N): HIP events on the launch stream around every collective -------
    def collect_timing(self, on=True):
        self._timing = {} if on else None

    def _timed(self, name, fn, *args):
        if self._timing is None or not torch.cuda.is_available():
            return fn(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*args)
        e1.record()
        self._timing.setdefault(name, []).append((e0, e1))
        return out

    def timing_summary(self):
        """{collective: {calls, mean_us, max_us}} since collect_timing(True); synchronises."""
        if not self._timing:
            return {}
        torch.cuda.synchronize()
        out = {}
        for name, pairs in self._timing.items():
            us = [1e3 * a.elapsed_time(b) for a, b in pairs]
            out[name] = dict(calls=len(us), mean_us=float(np.mean(us)), max_us=float(np.max(us)))
        return out

    # -- collectives ---------------------------------------------------------------------------------
    def _exchange_halo(self, which):
        if self.world == 1:
            self.b.set_halo(which, None, None)
            return
        self.b.export_edges(which, self._edges)
        self.comm.all_gather(self._all_edges, self._edges)
        left = self._all_edges[self.rank - 1, 3:6].contiguous() if self.rank > 0 else None
        right = self._all_edges[self.rank + 1, 0:3].contiguous() if self.rank + 1 < self.world else None
        self._halo_keep = (left, right)     # keep alive until the kernels that read them have run
        self.b.set_halo(which, left, right)

    def _global_control(self, init):
        self.b.export_partials(self._partial)
        if self.world > 1:
            self.comm.all_gather(self._all_partials, self._partial)
            total = combine_partials(self._all_partials)
        else:
            total = self._partial
        self._total_keep = total
        self.b.control(total, init)

    # -- driver --------------------------------------------------------------------------------------
    def set_x(self, x_local):
        if self.world == 1 and hasattr(self.b, "set_x_single"):
            self.b.set_x_single(x_local)
            return
        if self.world > 1 and getattr(self.b, "fused_phases", False):
            self.b.load_x(x_local)
            self.b.export_edges(0, self._edges)
            self.comm.all_gather(self._all_edges, self._edges)
            self.b.phase_eval(0, self._all_edges, self._partial)
            self.comm.all_gather(self._all_partials, self._partial)
            self.b.phase_control(self._all_partials, True)
            return
        self.b.load_x(x_local)
        self._exchange_halo(0)
        self.b.eval(0)
        self._global_control(True)

    def step(self):
        if self.world == 1 and hasattr(self.b, "step_single"):
            self.b.step_single()
            return
        if self.world > 1 and getattr(self.b, "fused_phases", False):
            # 4 launches + 3 collectives per iteration; every buffer is persistent, so the phases replay as graphs
            self.b.phase_reduce(self._sep)
            self._timed("all_reduce_separators", self.comm.all_reduce_sum, self._sep)
            self.b.phase_solve(self._sep, self._sep_x, self._edges)
            self._timed("all_gather_edges", self.comm.all_gather, self._all_edges, self._edges)
            self.b.phase_eval(1, self._all_edges, self._partial)
            self._timed("all_gather_scalars", self.comm.all_gather, self._all_partials, self._partial)
            self.b.phase_control(self._all_partials, False)
            return
        self.b.reduce_local()
        if self.world > 1:
            self._sep.zero_()
            self.b.export_separators(self._sep)
            self.comm.all_reduce_sum(self._sep)
            self.b.solve_separators(self._sep, self._sep_x)
            self.b.backsub_local(self._sep_x)
        else:
            self.b.backsub_local(None)
        self.b.trial()
        self._exchange_halo(1)
        self.b.eval(1)
        self._global_control(False)

    def solve(self, max_iter, peek_every=8):
        st = None
        for it in range(max_iter):
            self.step()
            if (it % peek_every) == peek_every - 1:
                st = self.b.state()
                if st["status"] != 0:
                    break
        return self.b.state()

    def gather_x(self, n_max):
        """Full trajectory x[N,25] on every rank: all-gather of the local shards padded to n_max frames
        (n_max = the largest shard of the plan)."""
        x = self.b.result_x()
        if self.world == 1:
            return x
        pad = self.b.new(n_max + 1, N_ACTIVE)
        pad[: x.shape[0]] = x
        pad[n_max, 0] = float(x.shape[0])
        out = self.b.new(self.world, n_max + 1, N_ACTIVE)
        self.comm.all_gather(out, pad)
        return torch.cat([out[g, : int(out[g, n_max, 0].item())] for g in range(self.world)], dim=0)


def make_sharded(det_full, k_arr, d_arr, r_arr, t_arr, Ts, rank, world, group=None, comm=None, **kw):
    """Convenience: slice this rank's frames out of a full det[N,C,20,3] and build the HIP-backed driver."""
    n_global = int(det_full.shape[0])
    plan = shard_plan(n_global, world) if world > 1 else [(0, n_global)]
    n0, n1 = plan[rank]
    backend = HipBackend(det_full[n0:n1], k_arr, d_arr, r_arr, t_arr, Ts, n_global, n0, rank, world, **kw)
    return ShardedFTE(backend, rank, world, group, comm), (n0, n1)


# ---------------------------------------------------------------------------------------------------------------
# Overlapping windows: the inexact-step variant (no separator system, two small collectives per iteration)
# ---------------------------------------------------------------------------------------------------------------
def window_plan(n_frames, world, halo):
    """[(w0, w1, n0, n1)] per rank: owned frames [n0, n1) as shard_plan, window [w0, w1) = owned + `halo` frames on
    either side (clipped to the sequence).  halo is rounded up to a multiple of 3 (windows start on a node boundary)."""
    halo = 3 * ((int(halo) + 2) // 3)
    plan = shard_plan(n_frames, world) if world > 1 else [(0, n_frames)]
    if world > 1 and min(n1 - n0 for n0, n1 in plan) < halo + 3:
        raise ValueError(f"shards of {min(n1 - n0 for n0, n1 in plan)} frames are shorter than the halo ({halo} + 3 frames)")
    return [(max(0, n0 - halo), min(n_frames, n1 + halo), n0, n1) for n0, n1 in plan], halo


class HipWindowBackend:
    """One rank's window on this process's GPU: thin calls into the C ABI (single-GPU context with an owned range)."""

    def __init__(self, det_window, k_arr, d_arr, r_arr, t_arr, Ts, n_global, n_offset, own_first, own_count, **kw):
        self.ctx = fte.FTEContext(det_window, k_arr, d_arr, r_arr, t_arr, Ts, n_global=n_global, n_offset=n_offset,
                                  own_first=own_first, own_count=own_count, **kw)
        self.device = self.ctx.device
        self.own_first, self.own_count = own_first, own_count

    def new(self, *shape):
        return torch.zeros(shape, dtype=torch.float64, device=self.device)

    def escalate(self):
        """After a refused (status 7) step: this window with one more reduction level.  Returns the current iterate of the
        whole window, which the driver loads again (every rank does the same in the same iteration: the refusal travels with
        the partial sums)."""
        n = self.ctx.N
        x = torch.empty((n, N_ACTIVE), dtype=torch.float64, device=self.device)
        self._c(lib().acino_fte_copy_frames, 0, 0, 0, n, ptr(x), stream_ptr())
        levels = self.ctx._next_levels()
        self.ctx.close()
        self.ctx._rebuild_with_levels(levels)      # (refinement sweeps, tolerance and precision travel with the rebuild)
        return x

    def _c(self, fn, *args):
        check(fn(self.ctx._h, *args))

    def load_x(self, x_window):
        self._x0 = calib_to_dev(x_window, self.device)
        self._c(lib().acino_fte_load_x, ptr(self._x0), stream_ptr())

    def solve_and_trial(self):
        L = lib()
        self._c(L.acino_fte_reduce_local, stream_ptr())
        self._c(L.acino_fte_backsub_local, C.c_void_p(0), 0, 1, stream_ptr())
        self._c(L.acino_fte_trial, stream_ptr())

    def copy_frames(self, which, imp, first, n, buf):
        self._c(lib().acino_fte_copy_frames, which, int(imp), first, n, ptr(buf), stream_ptr())

    def eval(self, which):
        self._c(lib().acino_fte_eval, which, stream_ptr())

    def export_partials(self, out):
        self._c(lib().acino_fte_export_partials, ptr(out), stream_ptr())

    def control(self, total, init):
        self._c(lib().acino_fte_control, ptr(total), int(init), stream_ptr())

    def state(self):
        return self.ctx.state()

    def result_owned(self):
        return self.ctx.result()[0][self.own_first:self.own_first + self.own_count]

    supports_graphs = True


class WindowedFTE:
    """One LM solve over a sequence sharded across the process group WITHOUT a separator system.

    The Gauss-Newton matrix is banded and SPD, so the influence of a right-hand side entry on the step decays
    geometrically with the frame distance (measured on the benchmark sequence: 5e-3 at 96 frames, 6e-6 at 192, 2e-9 at
    300 with lambda -> 0, faster with damping).  Every rank therefore solves its OWN window - owned frames plus `halo`
    frames of its neighbours', with the step pinned to 0 outside the window - by the complete single-GPU reduction, keeps
    the step on its owned frames and discards the rest (restricted additive Schwarz).  The step is inexact by
    ~decay(halo); cost, gradient, accept / reject and the damping are EXACT and global: the trial iterate's edge slabs
    travel in one all-gather, the eight partial sums in a second one, and every rank runs the same controller on the
    same totals.  Two small collectives per iteration, no all-reduce, no redundant separator solve.
    The numerical work lives behind a backend (``HipWindowBackend``; tests plug the numpy oracle in under gloo)."""

    def __init__(self, backend, rank, world, own, halo, group=None, comm=None):
        self.b, self.rank, self.world, self.group = backend, rank, world, group
        self.ctx = getattr(backend, "ctx", None)
        self.comm = comm if comm is not None else TorchComm(group)
        self.own_first, self.own_count = own           # local frame indices inside the window
        self.halo = halo
        self.slab = halo + 3                           # + the three stencil rows beyond the window
        self._edges = backend.new(2, self.slab, N_ACTIVE)        # first / last `slab` owned frames of the trial iterate
        self._all_edges = backend.new(world, 2, self.slab, N_ACTIVE)
        self._partial = backend.new(8)
        self._all_partials = backend.new(world, 8)
        self._timing = None
        self._graph_on, self._graphs = False, {}

    collect_timing = ShardedFTE.collect_timing
    _timed = ShardedFTE._timed
    timing_summary = ShardedFTE.timing_summary

    def _export_slabs(self, which):
        first, cnt, s = self.own_first, self.own_count, self.slab
        self.b.copy_frames(which, 0, first, s, self._edges[0])
        self.b.copy_frames(which, 0, first + cnt - s, s, self._edges[1])

    def _import_slabs(self, which):
        first, cnt, s = self.own_first, self.own_count, self.slab
        if self.rank > 0:                              # the left neighbour's LAST slab sits just before my owned frames
            self.b.copy_frames(which, 1, first - s, s, self._all_edges[self.rank - 1, 1])
        if self.rank + 1 < self.world:                 # the right neighbour's FIRST slab just after them
            self.b.copy_frames(which, 1, first + cnt, s, self._all_edges[self.rank + 1, 0])

    def _gather_scalars(self):
        if self.world > 1:
            self._timed("all_gather_scalars", self.comm.all_gather, self._all_partials, self._partial)
            total = combine_partials(self._all_partials)
        else:
            total = self._partial
        self._keep = total
        return total

    def set_x(self, x_window):
        """x_window[n_window, 25]: the initial iterate on this rank's WHOLE window."""
        self.b.load_x(x_window)
        if self.world > 1:     # the three stencil rows beyond the window (and my halo) take the neighbours' values
            self._export_slabs(0)
            self._timed("all_gather_edge_slabs", self.comm.all_gather, self._all_edges, self._edges)
            self._import_slabs(0)
        self.b.eval(0)
        self.b.export_partials(self._partial)
        self.b.control(self._gather_scalars(), True)

    # The iteration between the collectives is two fixed launch sequences on persistent buffers:
    #   A: reduce + back-substitution + trial iterate + export of my two edge slabs        -> all-gather (slabs)
    #   B: import of the neighbours' slabs + residuals / Jacobians / assembly + my sums    -> all-gather (scalars)
    # and the controller.  With enable_graph() each is captured once (torch.cuda.CUDAGraph around the C-ABI calls) and
    # replayed: 3 launches + 2 collectives per iteration instead of ~45 kernel launches.
    def enable_graph(self, on=True):
        self._graph_on = bool(on) and getattr(self.b, "supports_graphs", False)
        self._graphs = {}

    def _phase(self, name, body):
        if not self._graph_on:
            return body()
        g = self._graphs.get(name)
        if g is not None:
            return g.replay()
        body()                                         # eager: this iteration's work
        if not self._graphs.get(name + "_warm"):
            self._graphs[name + "_warm"] = True
            return
        g = torch.cuda.CUDAGraph()                     # second iteration: record the same sequence (a capture executes nothing)
        try:
            with torch.cuda.graph(g, stream=torch.cuda.current_stream(), capture_error_mode="thread_local"):
                body()
            self._graphs[name] = g
        except Exception:                              # capture not possible here (e.g. the default stream): stay eager
            self._graph_on = False

    def _phase_a(self):
        self.b.solve_and_trial()
        if self.world > 1:
            self._export_slabs(1)

    def _phase_b(self):
        self._import_slabs(1)                          # neighbours' owned values replace my halo estimate of the trial
        self.b.eval(1)
        self.b.export_partials(self._partial)

    def step(self):
        self._phase("a", self._phase_a)
        if self.world > 1:
            self._timed("all_gather_edge_slabs", self.comm.all_gather, self._all_edges, self._edges)
        self._phase("b", self._phase_b)
        self.b.control(self._gather_scalars(), False)

    def state(self):
        return self.b.state()

    def solve(self, max_iter, peek_every=8):
        it = 0
        while it < max_iter:
            self.step()
            it += 1
            if (it % peek_every) == 0 or it == max_iter:
                st = self.b.state()
                if st["status"] == 7 and hasattr(self.b, "escalate"):
                    # a rank's truncated solve could not verify its step: every rank has stopped (the flag travels with the
                    # sums), nobody applied it; all continue with one more reduction level from the current iterate
                    x = self.b.escalate()
                    self._graphs = {}
                    self.set_x(x)
                    continue
                if st["status"] != 0:
                    break
        return self.b.state()

    def result_x(self):
        """This rank's OWNED frames of the current iterate."""
        return self.b.result_owned()


def calib_to_dev(a, dev):
    from . import calib
    return calib._to_dev(a, dev)


def make_windowed(det_full, k_arr, d_arr, r_arr, t_arr, Ts, rank, world, halo=192, group=None, comm=None, **kw):
    """This rank's window of a full det[N,C,20,3] and the driver over it.  Returns (driver, (w0, w1, n0, n1)): the
    window and the owned frame range in global indices (set_x takes the initial iterate of [w0, w1))."""
    n_global = int(det_full.shape[0])
    plan, halo = window_plan(n_global, world, halo)
    w0, w1, n0, n1 = plan[rank]
    if w0 > 0 and n0 - w0 < halo:
        raise ValueError("window does not hold the halo")
    backend = HipWindowBackend(det_full[w0:w1], k_arr, d_arr, r_arr, t_arr, Ts, n_global, w0, n0 - w0, n1 - n0, **kw)
    return WindowedFTE(backend, rank, world, (n0 - w0, n1 - n0), halo, group, comm), (w0, w1, n0, n1)
