// FTE solve: context, Levenberg-Marquardt controller (device-resident) and the C ABI.
// Reference: the Pyomo model + IPOPT solve of src/all_optimizations.py:283-556 in reduced form.
#include <algorithm>
#include <map>
#include <new>

#include "bcr.hpp"
#include "chunk.hpp"
#include "seplevel.hpp"

namespace acino {

struct Buffers {
  FteConst* cst;
  acino_fte_state* state;
  double* x[2];       // [(N+6)][25] with 3 halo frames each side
  double* g[2];       // [N][25]
  double* H[2];       // [N][25][25] Gauss-Newton blocks (measurement + smoothness diagonal)
  double* hd[2];      // [N][25] their diagonals, contiguous (the trial kernel would otherwise touch all of H for them)
  double* cost_part;  // [nblk]
  double* pred_part;  // [nblk_t]
  double* step_part;  // [nblk_t]
  double* gn_part;    // [n_nodes]
  double* totals;     // [8]
  int* nbehind;
  int* numeric_err;
  int* sched;         // elim (3/entry), remain (4/entry), tail (4/entry), dropped-coupling pairs (2/entry), tail counter
  double* trunc_eps2; // [n_pairs + 1]
  double* refine_buf; // [3][n_isolated][80] (incomplete reduction with refinement sweeps)
  int* st_flags;      // [n_isolated + n_sep] flags of k_sep_tail (chunked solver with refinement), then the epoch counter
  int n_st_flags;
  unsigned long long* st_ll;   // k_sep_tail's hand-off slots (BcrChain::st_ll)
  size_t n_st_ll;
};

}  // namespace acino

struct acino_fte_ctx {
  double lam0;
  acino::FteConst h;
  acino::Buffers b;
  acino::BcrChain chain;         // the chain of 3-frame nodes
  acino::BcrSchedule sched;      // reduction schedule of `chain`, or - chunked solver - of the separator chain
  acino::ChunkPlan plan;         // chunked solver (csrc/chunk.hip); inactive: block cyclic reduction over `chain`
  acino::BcrChain sepchain;      // chunked solver: the runs' separators
  acino::SepView sep;
  int n_trunc = 0;               // dropped couplings of an incomplete reduction (of whichever chain `sched` reduces)
  const double* d_det;
  int n_blk_asm, n_blk_trial;
  int n_pred;         // entries of pred_part / step_part: blocks of k_trial, or runs of the chunked back-substitution (which folds it in)
  size_t ws_bytes;
  acino::Profiler prof;
  bool graph_on = false;            // replay the LM step as a hipGraph (single-shard contexts, non-null stream)
  hipGraphExec_t gexec = nullptr;
  hipStream_t gstream = nullptr;
  // the four phases of a SHARDED iteration (between the collectives), each captured once per buffer set
  struct SegGraph {
    hipGraphExec_t exec = nullptr;
    hipStream_t stream = nullptr;
    uintptr_t key[6] = {0, 0, 0, 0, 0, 0};
  } seg[4];
};

namespace acino {

static size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

static int chain_nodes(const acino_fte_params* p) { return (p->n_frames + 2) / 3 + (p->pin_left ? 1 : 0); }

struct Carver {
  char* base;
  size_t off;
  template <class T>
  T* take(size_t count) {
    T* p = reinterpret_cast<T*>(base + off);
    off = align_up(off + count * sizeof(T));
    return p;
  }
};

// (sharded ranks - pinned separators - take the chunked solver too since round 4: the pins join the separator chain)
static bool use_chunks(const acino_fte_params* p) { return p->chunk_nodes >= 0; }

static bool fused_levels_enabled() { return getenv("ACINO_NO_FUSED_LEVELS") == nullptr; }

// Host-side layout of a context: the chunk plan and the reduction schedule (of the separator chain when chunked).
struct Layout {
  ChunkPlan plan;
  BcrSchedule sched;
  void build(const acino_fte_params* p) {
    plan.build(chain_nodes(p), use_chunks(p) ? p->chunk_nodes : -1, p->pin_left != 0, p->pin_right != 0);
    // (separator chains take the fused narrow levels of seplevel.hip; ACINO_NO_FUSED_LEVELS=1: the per-phase kernels of bcr.hip)
    if (plan.active()) sched.build(plan.n_sep, p->pin_left != 0, p->pin_right != 0, p->bcr_levels, p->refine_sweeps,
                                   fused_levels_enabled());
    else sched.build(chain_nodes(p), p->pin_left != 0, p->pin_right != 0, p->bcr_levels, p->refine_sweeps);
  }
};

static size_t carve(const acino_fte_params* p, char* base, Buffers* out, BcrChain* ch, const Layout& lay, BcrChain* sepch = nullptr) {
  Carver c{base, 0};
  const size_t N = p->n_frames, T = chain_nodes(p);
  const size_t sched_ints = lay.sched.ints(), n_pairs = lay.sched.pairs.size() / 2;
  const bool chunked = lay.plan.active();
  Buffers b;
  b.cst = c.take<FteConst>(1);
  b.state = c.take<acino_fte_state>(1);
  for (int k = 0; k < 2; ++k) b.x[k] = c.take<double>((N + 2 * HALO) * NP);
  for (int k = 0; k < 2; ++k) b.g[k] = c.take<double>(N * NP);
  for (int k = 0; k < 2; ++k) b.H[k] = c.take<double>(N * HPAIRS);
  for (int k = 0; k < 2; ++k) b.hd[k] = c.take<double>(N * NP);
  b.cost_part = c.take<double>(n_assemble_blocks((int)N) + 1);
  const size_t n_pred = std::max((N * NP + 255) / 256, (size_t)std::max(lay.plan.n_chunks, 0)) + 1;
  b.pred_part = c.take<double>(n_pred);
  b.step_part = c.take<double>(n_pred);
  b.gn_part = c.take<double>(T + 1);
  b.totals = c.take<double>(8);
  b.nbehind = c.take<int>(4);
  b.numeric_err = b.nbehind + 1;
  b.sched = c.take<int>(sched_ints);
  b.trunc_eps2 = c.take<double>(7 * (n_pairs + 1) + 1);   // (refinement: seven [n_isolated] arrays of sweep norms, bcr_backsub)
  b.refine_buf = nullptr;
  if (lay.sched.refine > 0) b.refine_buf = c.take<double>(3 * (size_t)lay.sched.levels.back().n_elim * BS);
  b.st_flags = nullptr;
  b.n_st_flags = 0;
  b.st_ll = nullptr;
  b.n_st_ll = 0;
  if (lay.sched.refine > 0 && chunked) {
    b.n_st_flags = lay.sched.levels.back().n_elim + lay.plan.n_sep;
    b.st_flags = c.take<int>((size_t)b.n_st_flags + 2);
    b.n_st_ll = (2 * (size_t)lay.sched.levels.back().n_elim + (size_t)lay.plan.n_sep) * BS * 2;
    b.st_ll = c.take<unsigned long long>(b.n_st_ll);
  }
  BcrChain chn;
  chn.n_nodes = (int)T;
  chn.D = c.take<double>(T * BS * BS);                       // chunked: G_k of the interior nodes (lower tiles)
  chn.U = chunked ? nullptr : c.take<double>(T * BS * BS);
  chn.Cpl = chunked ? nullptr : c.take<double>(T * BS * BS);
  chn.Wl = c.take<double>(chunked ? T * BS : T * BS * BS);   // chunked: f_k = F_k x_L of the interior nodes, [80] each
  chn.Wr = chunked ? nullptr : c.take<double>(T * BS * BS);
  chn.b = c.take<double>(T * BS);
  chn.d_elim = nullptr;
  chn.d_remain = nullptr;
  chn.d_tail = nullptr;
  chn.d_done = nullptr;
  chn.d_pairs = nullptr;
  chn.n_pairs = 0;
  chn.trunc_eps2 = b.trunc_eps2;
  chn.implicit_couplings = 1;
  chn.dbg = nullptr;
  chn.st = b.state;
  chn.x0 = b.x[0]; chn.x1 = b.x[1];
  chn.g0 = b.g[0]; chn.g1 = b.g[1];
  chn.H0 = b.H[0]; chn.H1 = b.H[1];
  chn.gn_part = b.gn_part;
  BcrChain sc = chn;
  if (chunked) {
    const size_t S = lay.plan.n_sep > 0 ? lay.plan.n_sep : 1;
    sc.n_nodes = lay.plan.n_sep;
    sc.D = c.take<double>(S * BS * BS);
    sc.U = c.take<double>(S * BS * BS);
    const bool fz = lay.sched.fused_levels;
    sc.Cpl = c.take<double>((fz ? 2 : 1) * S * BS * BS);     // fused levels: slot S + i = the coupling the elimination of i created
    sc.Wl = c.take<double>(S * BS * BS);
    sc.Wr = c.take<double>(S * BS * BS);
    sc.b = c.take<double>(S * BS);
    if (fz) {
      sc.SL = c.take<double>(S * BS * BS);
      sc.SR = c.take<double>(S * BS * BS);
      sc.Y = c.take<double>(S * BS);
    }
    sc.AL0 = c.take<double>(S * BS * BS);   // (the sweep's left-run contributions; handed to the reduction by chunk_reduce)
    sc.implicit_couplings = 0;      // dense couplings, plain (non-fused) kernels
    sc.st = nullptr;
    sc.x0 = sc.x1 = sc.g0 = sc.g1 = sc.H0 = sc.H1 = nullptr;
    sc.gn_part = nullptr;
  }
  if (out) *out = b;
  if (ch) *ch = chn;
  if (sepch) *sepch = sc;
  return c.off;
}

// ---- kernels ----------------------------------------------------------------------------------
// (the damped system D = H_gn + lam*diag(H_gn), b = -g is built inside the level-0 BCR kernels: bcr.hip build_node)
// Trial iterate x_t = clip(x + delta) and the model quantities of the step.
__global__ void __launch_bounds__(256)
k_trial(const FteConst* __restrict__ cst, const acino_fte_state* __restrict__ st, double* __restrict__ x0,
        double* __restrict__ x1, const double* __restrict__ g0, const double* __restrict__ g1,
        const double* __restrict__ hd0, const double* __restrict__ hd1, const double* __restrict__ delta_nodes,
        double* __restrict__ pred_part, double* __restrict__ step_part) {
  if (st->status != 0) return;
  const FteConst& K = *cst;
  const int cur = st->cur;
  const double* x = cur ? x1 : x0;
  double* xt = cur ? x0 : x1;
  const double* g = cur ? g1 : g0;
  const double* hd = cur ? hd1 : hd0;     // diag of the Gauss-Newton blocks, written by the assembly
  const double lam = st->lam;
  const int tid = threadIdx.x;
  const int64_t e = (int64_t)blockIdx.x * 256 + tid;
  double pred = 0.0, step = 0.0;
  if (e < (int64_t)K.n_frames * NP) {
    const int n = (int)(e / NP), p = (int)(e % NP);
    const int node = n / 3 + K.pin_left, row = (n % 3) * NP + p;
    const double xv = x[(size_t)(n + HALO) * NP + p], gv = g[e];
    const double d0 = hd[e];
    const double gtol = GRAD_ZERO_REL * d0;
    const bool fixed = (xv <= K.lo[p] && gv > gtol) || (xv >= K.hi[p] && gv < -gtol);
    // A bound-active variable is pinned in the linear system by the 2^70 diagonal boost, which leaves it a step of
    // ~1e-21 instead of the exact 0 of a deleted row.  At a bound of 0.0 (theta_7, 9, 11, 13 - where the nose-line
    // initialisation puts them) that residue is representable: x would leave the bound by 1e-21, count as free in the next
    // iteration and take a different path than the active-set rule prescribes.  Its step is 0, exactly.
    const double d = fixed ? 0.0 : delta_nodes[(size_t)node * BS + row];
    const double pg = fixed ? 0.0 : gv;
    const double xn = fmin(fmax(xv + d, K.lo[p]), K.hi[p]);
    xt[(size_t)(n + HALO) * NP + p] = xn;
    if (n >= K.own_lo && n < K.own_hi) {          // (window sharding: only owned frames enter the global sums)
      pred = 0.5 * d * (lam * fmax(d0, DIAG_FLOOR) * d - pg);
      step = fabs(xn - xv);
    }
  }
  __shared__ double rp[4], rs[4];
  for (int off = 32; off > 0; off >>= 1) {
    pred += __shfl_down(pred, off, 64);
    step = fmax(step, __shfl_down(step, off, 64));
  }
  if ((tid & 63) == 0) {
    rp[tid >> 6] = pred;
    rs[tid >> 6] = step;
  }
  __syncthreads();
  if (tid == 0) {
    pred_part[blockIdx.x] = (rp[0] + rp[1]) + (rp[2] + rp[3]);
    step_part[blockIdx.x] = fmax(fmax(rs[0], rs[1]), fmax(rs[2], rs[3]));
  }
}

// totals = {cost, pred, step_inf, gnorm_inf, n_behind, 0, 0, 0}   (fixed summation order)
struct LmTol {
  double gtol, ftol, xtol, lam_max;
  int clamp_lambda;
};
__device__ void lm_control_local(const LmTol& K, acino_fte_state& S, const double* totals, int numeric_err, int init);
__device__ void lm_control(const FteConst& K, acino_fte_state* st, const double* totals, const int* numeric_err,
                           int init);

// fused_control: -1 none (sharded: the decision needs the cross-rank sums), 0 LM step, 1 initial evaluation
__global__ void __launch_bounds__(1024)
k_totals(acino_fte_state* st, const double* cost_part, int n_cost, const double* pred_part,
         const double* step_part, int n_trial, const double* gn_part, int n_nodes, int* nbehind, double* totals,
         int with_step, const FteConst* __restrict__ cst, const int* __restrict__ numeric_err, int fused_control,
         const double* __restrict__ trunc_eps2, int n_trunc) {
  // (the status word is LOADED first and TESTED behind the loads of the partial sums: tested here, every other load of this
  //  single-workgroup kernel would wait a round trip for it)
  const int st_status = st->status;
  int* numeric_err_rw = const_cast<int*>(numeric_err);
  // the four reductions run together: strided per-thread partials, one shuffle tree per wave, the sixteen waves
  // combined in order - a fixed summation order, one barrier.  1024 threads: this single workgroup is a chain of
  // HBM round trips (one per stride), so the stride count is what it costs - 4 instead of 14 for 10 000 frames
  constexpr int NW = 16;
  __shared__ double sh[NW][10];
  // thread 0 will run the controller: its operands are requested now, beside the reductions
  acino_fte_state S;
  LmTol T{0, 0, 0, 0, 0};
  int ne = 0, nb = 0, n_ref = 0;
  double e2 = 0.0, ttol = 0.0, r_d1 = 0.0, r_d0 = 0.0, r_x = 0.0, r_f1 = 0.0, r_f2 = 0.0;
  if (threadIdx.x == 0) {
    S = *st;
    T = LmTol{cst->gtol, cst->ftol, cst->xtol, cst->lam_max, cst->clamp_lambda};
    ne = *numeric_err;
    nb = *nbehind;
    ttol = cst->trunc_tol;
    n_ref = cst->refine_sweeps;
  }
  double c = 0.0, p = 0.0, s = 0.0, g = 0.0;
  // incomplete reduction: plain truncation -> the largest dropped coupling (eps^2) of this step's solve; with refinement
  // sweeps -> max |update| of the last sweep (r_d1) and of the one before (r_d0), max |x| (r_x); strided like the sums
  if (with_step && n_trunc > 0) {
    if (cst->refine_sweeps > 0) {
      const int n_iso = n_trunc + 1;
      for (int i = threadIdx.x; i < n_iso; i += 64 * NW) {
        r_d1 = fmax(r_d1, trunc_eps2[i]);
        r_x = fmax(r_x, trunc_eps2[n_iso + i]);
        r_d0 = fmax(r_d0, trunc_eps2[2 * n_iso + i]);
        r_f1 = fmax(r_f1, trunc_eps2[4 * n_iso + i]);      // (zero where the sweep count leaves them unwritten)
        r_f2 = fmax(r_f2, trunc_eps2[5 * n_iso + i]);
      }
    } else {
      for (int i = threadIdx.x; i < n_trunc; i += 64 * NW) e2 = fmax(e2, trunc_eps2[i]);
    }
  }
  for (int i = threadIdx.x; i < n_cost; i += 64 * NW) c += cost_part[i];
  if (with_step) {
    for (int i = threadIdx.x; i < n_trial; i += 64 * NW) {
      p += pred_part[i];
      s = fmax(s, step_part[i]);
    }
    for (int i = threadIdx.x; i < n_nodes; i += 64 * NW) g = fmax(g, gn_part[i]);
  }
  if (st_status != 0) return;
  for (int off = 32; off > 0; off >>= 1) {
    c += __shfl_down(c, off, 64);
    p += __shfl_down(p, off, 64);
    s = fmax(s, __shfl_down(s, off, 64));
    g = fmax(g, __shfl_down(g, off, 64));
    e2 = fmax(e2, __shfl_down(e2, off, 64));
    r_d1 = fmax(r_d1, __shfl_down(r_d1, off, 64));
    r_d0 = fmax(r_d0, __shfl_down(r_d0, off, 64));
    r_x = fmax(r_x, __shfl_down(r_x, off, 64));
    r_f1 = fmax(r_f1, __shfl_down(r_f1, off, 64));
    r_f2 = fmax(r_f2, __shfl_down(r_f2, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    double* w = sh[threadIdx.x >> 6];
    w[0] = c;
    w[1] = p;
    w[2] = s;
    w[3] = g;
    w[4] = e2;
    w[5] = r_d1;
    w[6] = r_d0;
    w[7] = r_x;
    w[8] = r_f1;
    w[9] = r_f2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    c = 0.0;
    p = 0.0;
    s = 0.0;
    g = 0.0;
    for (int w = 0; w < NW; ++w) {
      c += sh[w][0];
      p += sh[w][1];
      s = fmax(s, sh[w][2]);
      g = fmax(g, sh[w][3]);
      e2 = fmax(e2, sh[w][4]);
      r_d1 = fmax(r_d1, sh[w][5]);
      r_d0 = fmax(r_d0, sh[w][6]);
      r_x = fmax(r_x, sh[w][7]);
      r_f1 = fmax(r_f1, sh[w][8]);
      r_f2 = fmax(r_f2, sh[w][9]);
    }
    double tot[8] = {c, p, s, g, (double)nb, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 8; ++q) totals[q] = tot[q];
    *nbehind = 0;
    if (with_step && n_trunc > 0) {
      // what the step's relative error is bounded by.  Plain truncation: eps, the measured size of the dropped couplings.
      // With r block-Jacobi sweeps over them (bcr.hip k_bcr_refine): the iteration contracts by rho <= 2 eps per sweep, so
      // the error left after the last sweep is <= rho / (1 - rho) |last update|; rho is measured as a ratio of consecutive
      // updates (one sweep only: rho = 1/2 assumed), and a rho above 1/2 refuses the step.
      double bound;
      if (n_ref > 0) {
        // The contraction is measured where it can be: an update at the rounding level (below 2^-44 of the largest component)
        // is noise, and so is a ratio it takes part in.  The first two sweeps give the clean measurement (the truncated solve
        // is off by ~rho, their updates are ~rho and ~rho^2 of the solution); the last two confirm it as long as they are above
        // the noise - with many sweeps they are not, and their ratio (noise over noise, ~1) must not refuse a converged
        // solve.  Nothing measurable at all (the first update already at rounding): rho = 1/2 assumed, as with one sweep.
        double rho = 0.5;
        if (n_ref >= 2) {
          const double noise = 0x1p-44 * r_x;
          const double d1 = n_ref == 2 ? r_d0 : r_f1, d2 = n_ref == 2 ? r_d1 : (n_ref == 3 ? r_d0 : r_f2);
          const double rho_first = d1 > noise ? d2 / d1 : 0.5;
          const double rho_last = (r_d1 > noise && r_d0 > noise) ? r_d1 / r_d0 : 0.0;
          rho = fmax(rho_first, rho_last);
        }
        bound = (rho <= 0.5 && r_x > 0.0) ? rho / (1.0 - rho) * r_d1 / r_x : (r_d1 == 0.0 ? 0.0 : 1.0);
        S.trunc_eps = bound;
      } else {
        S.trunc_eps = sqrt(e2);
        bound = S.trunc_eps;
      }
      if (!(bound <= ttol)) ne |= 4;                 // (also catches NaN)
      if (fused_control < 0) {
        // split-control path (window / separator drivers): k_control reads the DEVICE flag, so the refusal must be there
        st->trunc_eps = S.trunc_eps;
        if (ne & 4) atomicOr(numeric_err_rw, 4);
      }
    }
    // slot 5: this rank's numeric flags (bit 0 pivot, 1 tail timeout, 2 truncation).  Sharded drivers combine it over the
    // ranks (max) with the sums, so every rank's controller takes the same status in the same iteration
    totals[5] = (double)ne;
    tot[5] = (double)ne;
    if (fused_control >= 0) {
      lm_control_local(T, S, tot, ne, fused_control);
      *st = S;
    }
  }
}

// Accept / reject + Nielsen lambda update; mirrors oracle/fte.py:lm_solve step for step.
// The controller on a LOCAL copy of the state (the caller loads the state and the tolerances early and stores the state
// back once: on one thread every access to global memory is a dependent round trip).
__device__ void lm_control_local(const LmTol& K, acino_fte_state& S, const double* totals, int numeric_err, int init) {
  if (S.status != 0) return;
  if (init) {
    S.cost = totals[0];
    S.cost_trial = totals[0];
    S.n_behind = (int)totals[4];
    return;
  }
  const double F = S.cost, Ft = totals[0], pred = totals[1], step = totals[2], gnorm = totals[3];
  S.cost_trial = Ft;
  S.pred = pred;
  S.step_inf = step;
  S.gnorm_inf = gnorm;
  S.iter += 1;
  if (numeric_err) {      // bit 0: non-positive pivot, bit 1: back-substitution tail timed out, bit 2: dropped couplings too large
    S.status = (numeric_err & 1) ? 5 : ((numeric_err & 2) ? 6 : 7);
    return;
  }
  if (gnorm <= K.gtol) {   // the iterate the step started from was already stationary: keep it
    S.status = 3;
    S.last_accept = 0;
    return;
  }
  const double gain = pred > 0.0 ? (F - Ft) / pred : -1.0;
  S.gain = gain;
  if (Ft < F) {
    const double dF = F - Ft;
    S.cur ^= 1;
    S.cost = Ft;
    S.accepted += 1;
    S.last_accept = 1;
    S.n_behind = (int)totals[4];
    const double t = 2.0 * gain - 1.0;
    S.lam = S.lam * fmax(1.0 / 3.0, 1.0 - t * t * t);
    S.nu = 2.0;
    if (dF <= K.ftol * fabs(Ft)) S.status = 1;
    else if (step <= K.xtol) S.status = 2;
  } else {
    S.last_accept = 0;
    S.lam *= S.nu;
    S.nu *= 2.0;
    if (S.lam > K.lam_max) {
      if (K.clamp_lambda) {
        S.lam = K.lam_max;
        S.nu = 2.0;
      } else {
        S.status = 4;
      }
    }
  }
}

__device__ void lm_control(const FteConst& K, acino_fte_state* st, const double* totals, const int* numeric_err,
                           int init) {
  acino_fte_state S = *st;
  const LmTol T{K.gtol, K.ftol, K.xtol, K.lam_max, K.clamp_lambda};
  // the local flag and the flag that travelled with the (combined) sums: a failure on ANY rank stops every rank
  const int flag = *numeric_err | (totals[5] > 0.0 ? (int)totals[5] : 0);
  lm_control_local(T, S, totals, flag, init);
  *st = S;
}

__global__ void k_control(const FteConst* __restrict__ cst, acino_fte_state* st, const double* __restrict__ totals,
                          const int* __restrict__ numeric_err, int init) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  lm_control(*cst, st, totals, numeric_err, init);
}

// Sharded solve: every rank combines the gathered per-rank sums {cost, pred, step_inf, gnorm_inf, n_behind} in rank
// order (identical on all ranks) and takes the accept/reject decision on its own device.
__global__ void k_control_gathered(const FteConst* __restrict__ cst, acino_fte_state* st,
                                   const double* __restrict__ all_partials, int world,
                                   const int* __restrict__ numeric_err, int init) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int g = 0; g < world; ++g) {
    const double* p = all_partials + 8 * g;
    tot[0] += p[0];
    tot[1] += p[1];
    tot[2] = fmax(tot[2], p[2]);
    tot[3] = fmax(tot[3], p[3]);
    tot[4] += p[4];
    tot[5] = (double)((int)tot[5] | (p[5] > 0.0 ? (int)p[5] : 0));
  }
  lm_control(*cst, st, tot, numeric_err, init);
}

__global__ void k_copy_x_in(const FteConst* __restrict__ cst, const acino_fte_state* __restrict__ st, int which,
                            const double* __restrict__ src, double* x0, double* x1) {
  const FteConst& K = *cst;
  double* dst = (st->cur ^ which) ? x1 : x0;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < (int64_t)K.n_frames * NP) {
    const int p = (int)(e % NP);
    dst[e + HALO * NP] = fmin(fmax(src[e], K.lo[p]), K.hi[p]);
  }
}

__global__ void k_set_halo(const acino_fte_state* __restrict__ st, int which, double* x0, double* x1, int n_frames,
                           const double* __restrict__ hl, const double* __restrict__ hr) {
  double* xbuf = (st->cur ^ which) ? x1 : x0;
  const int t = threadIdx.x;
  if (t < HALO * NP) {
    xbuf[t] = hl ? hl[t] : 0.0;
    xbuf[(size_t)(n_frames + HALO) * NP + t] = hr ? hr[t] : 0.0;
  }
}

// dx_n = (x_n - x_{n-1})/Ts, ddx_n = (dx_n - dx_{n-1})/Ts (all_optimizations.py:369-383); the free first
// rows take the values of the optimum: ddx_0 = ddx_1 = ddx_2, dx_0 = dx_1 - Ts ddx_1.
__global__ void k_derivatives(const double* __restrict__ x, int64_t n, double ts, double* __restrict__ dx,
                              double* __restrict__ ddx) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n * NP) return;
  const int64_t f = e / NP;
  const int p = (int)(e % NP);
  auto X = [&](int64_t k) { return x[k * NP + p]; };
  auto DX = [&](int64_t k) { return (X(k) - X(k - 1)) / ts; };          // k >= 1
  auto DDX = [&](int64_t k) { return (DX(k) - DX(k - 1)) / ts; };       // k >= 2
  double v = 0.0, a = 0.0;
  if (n >= 3) {
    a = f >= 2 ? DDX(f) : DDX(2);
    v = f >= 1 ? DX(f) : DX(1) - ts * DDX(2);
  } else if (n == 2) {
    v = DX(1);
  }
  if (dx) dx[e] = v;
  if (ddx) ddx[e] = a;
}

__global__ void k_copy_x_out(const acino_fte_state* __restrict__ st, int which, const double* x0, const double* x1,
                             int64_t n, double* __restrict__ dst) {
  const double* xh = (st->cur ^ which) ? x1 : x0;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n * NP) dst[e] = xh[e + HALO * NP];
}

__global__ void k_export_HG(const acino_fte_state* __restrict__ st, const double* g0, const double* g1,
                            const double* H0, const double* H1, int64_t n, double* __restrict__ dg,
                            double* __restrict__ dh) {
  const double* g = st->cur ? g1 : g0;
  const double* H = st->cur ? H1 : H0;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (dg && e < n * NP) dg[e] = g[e];
  if (dh)
    for (int64_t k = e; k < n * NP * NP; k += (int64_t)gridDim.x * 256) {      // (out: the full [25][25] block)
      const int rem = (int)(k % (NP * NP));
      dh[k] = H[(k / (NP * NP)) * HPAIRS + hpair(rem / NP, rem % NP)];
    }
}

__global__ void k_export_edges(const acino_fte_state* __restrict__ st, int which, const double* x0, const double* x1,
                               int n_frames, double* __restrict__ edge) {
  const double* xh = (st->cur ^ which) ? x1 : x0;
  const int t = threadIdx.x;
  if (t < HALO * NP) {
    edge[t] = xh[HALO * NP + t];                                              // first 3 frames
    edge[HALO * NP + t] = xh[(size_t)(n_frames + HALO - 3) * NP + t];         // last 3 frames
  }
}

__global__ void __launch_bounds__(256)
k_copy_frames(const acino_fte_state* __restrict__ st, int which, int import, double* x0, double* x1, int first, int n,
              double* __restrict__ buf) {
  double* xh = ((st->cur ^ which) ? x1 : x0) + (size_t)(first + HALO) * NP;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < n * NP) {
    if (import) xh[e] = buf[e];
    else buf[e] = xh[e];
  }
}

__global__ void __launch_bounds__(256)
k_export_sep(BcrChain ch, int node_left, int node_right, double* __restrict__ rec_left, double* __restrict__ rec_right) {
  // rec layout: D[6400] | C[6400] | b[80]
  const size_t MB = (size_t)BS * BS;
  for (int e = threadIdx.x + blockIdx.x * 256; e < (int)MB; e += gridDim.x * 256) {
    if (rec_left) {   // this rank's pinned-left node: Schur update of separator (rank-1), plus coupling to the right separator
      rec_left[e] = ch.D[(size_t)node_left * MB + e];
    }
    if (rec_right) {
      rec_right[e] = ch.D[(size_t)node_right * MB + e];
      if (rec_left) rec_left[MB + e] = ch.Cpl[(size_t)node_left * MB + e];    // block(right sep, left sep)
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < BS) {
    if (rec_left) rec_left[2 * MB + threadIdx.x] = ch.b[(size_t)node_left * BS + threadIdx.x];
    if (rec_right) rec_right[2 * MB + threadIdx.x] = ch.b[(size_t)node_right * BS + threadIdx.x];
  }
}

__global__ void k_import_sep(const double* __restrict__ rec, int n_sep, BcrChain ch) {
  const size_t MB = (size_t)BS * BS;
  const int s = blockIdx.y;
  const double* r = rec + (size_t)s * ACINO_SEP_DOUBLES;
  for (int e = threadIdx.x + blockIdx.x * 256; e < (int)MB; e += gridDim.x * 256) {
    ch.D[(size_t)s * MB + e] = r[e];
    if (s + 1 < n_sep) ch.Cpl[(size_t)s * MB + e] = r[MB + e];   // record s carries block(s+1, s)
  }
  if (blockIdx.x == 0 && threadIdx.x < BS) ch.b[(size_t)s * BS + threadIdx.x] = r[2 * MB + threadIdx.x];
}

__global__ void k_set_node_x(BcrChain ch, int node, const double* __restrict__ xsep) {
  if (threadIdx.x < BS) ch.b[(size_t)node * BS + threadIdx.x] = xsep[threadIdx.x];
}

__global__ void k_copy_vec(const double* __restrict__ src, double* __restrict__ dst, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

static int fill_const(const acino_fte_params* p, const double* h_cams, FteConst* c) {
  memset(c, 0, sizeof(*c));
  c->n_frames = p->n_frames;
  c->n_cams = p->n_cams;
  c->n_global = p->n_global;
  c->n_offset = p->n_offset;
  c->clip_len = p->clip_len;
  c->pin_left = p->pin_left ? 1 : 0;
  c->pin_right = p->pin_right ? 1 : 0;
  c->n_nodes = chain_nodes(p);
  c->dlc_thresh = p->dlc_thresh;
  c->inv_r = p->inv_r_meas;
  c->loss = make_loss(p->redesc_a, p->redesc_b, p->redesc_c);
  for (int i = 0; i < NP; ++i) {
    c->q_w[i] = p->q_w[i];
    c->lo[i] = p->lo[i];
    c->hi[i] = p->hi[i];
  }
  c->ftol = p->ftol;
  c->xtol = p->xtol;
  c->gtol = p->gtol;
  c->lam_max = p->lam_max > 0 ? p->lam_max : 1e16;
  c->clamp_lambda = p->clamp_lambda;
  c->precision = p->precision;
  c->trunc_tol = p->trunc_tol > 0.0 ? p->trunc_tol : 1e-10;
  c->refine_sweeps = p->refine_sweeps;
  c->own_lo = p->own_count > 0 ? p->own_first : 0;
  c->own_hi = p->own_count > 0 ? p->own_first + p->own_count : p->n_frames;
  memcpy(c->cams, h_cams, sizeof(double) * ACINO_CAM_STRIDE * p->n_cams);
  return ACINO_OK;
}

static int validate(const acino_fte_params* p) {
  ACINO_REQUIRE(p != nullptr, "params");
  ACINO_REQUIRE(p->n_frames >= 1, "n_frames >= 1");
  ACINO_REQUIRE(p->n_cams >= 1 && p->n_cams <= ACINO_MAX_CAMS, "n_cams in 1..16");
  ACINO_REQUIRE(p->n_global >= p->n_frames && p->n_offset >= 0 && p->n_offset + p->n_frames <= p->n_global,
                "shard range inside the sequence");
  ACINO_REQUIRE(!p->pin_left || (p->n_offset >= 3 && p->n_offset % 3 == 0), "pinned-left shard must start at a multiple of 3");
  ACINO_REQUIRE(p->clip_len >= 0, "clip_len");
  ACINO_REQUIRE(p->precision >= ACINO_PREC_F64 && p->precision <= ACINO_PREC_BF16_RES, "precision");
  ACINO_REQUIRE(p->bcr_levels >= 0 && p->trunc_tol >= 0.0, "bcr_levels, trunc_tol");
  ACINO_REQUIRE(p->chunk_nodes >= -1 && p->chunk_nodes != 1, "chunk_nodes: -1 (off), 0 (automatic) or >= 2");
  ACINO_REQUIRE(p->refine_sweeps >= 0 && p->refine_sweeps <= 64, "refine_sweeps in 0..64");
  ACINO_REQUIRE(p->own_count >= 0 && (p->own_count == 0 || (p->own_first >= 0 && p->own_first + p->own_count <= p->n_frames &&
                                                           !p->pin_left && !p->pin_right && p->clip_len == 0)),
                "own range must lie inside the window (and windows have no pinned separators / clips)");
  ACINO_REQUIRE(p->bcr_levels == 0 || (!p->pin_left && !p->pin_right),
                "incomplete reduction (bcr_levels > 0): single-GPU contexts only");
  ACINO_REQUIRE(p->clip_len == 0 || (!p->pin_left && !p->pin_right && p->n_offset == 0 && p->n_global == p->n_frames &&
                                     p->n_frames % p->clip_len == 0),
                "clips: single-GPU context whose n_frames is a multiple of clip_len");
  ACINO_REQUIRE(!p->pin_right || (p->n_frames % 3 == 0), "pinned-right shard must hold a multiple of 3 frames");
  ACINO_REQUIRE(!(p->pin_left || p->pin_right) || p->n_frames >= 6, "sharded ranks need >= 6 frames");
  ACINO_REQUIRE(p->redesc_c > p->redesc_b && p->redesc_b > p->redesc_a && p->redesc_a > 0, "redescending a<b<c");
  ACINO_REQUIRE(p->lam0 > 0, "lam0 > 0");
  return ACINO_OK;
}

}  // namespace acino

using namespace acino;

static int eval_iterate(acino_fte_ctx* ctx, int which, bool need_jac, bool with_step, bool respect_status,
                        hipStream_t s, int fused_control = -1) {
  const Buffers& b = ctx->b;
  int rc;
  {
    ProfSpan sp(&ctx->prof, PC_ASSEMBLE, s, ctx->h.n_frames);
    rc = launch_assemble(b.cst, ctx->h, b.state, which, ctx->d_det, b.x, b.H, b.g, b.hd, b.cost_part, b.nbehind,
                         need_jac, respect_status, s);
  }
  if (rc) return rc;
  {
    ProfSpan sp(&ctx->prof, PC_TOTALS, s);
    hipLaunchKernelGGL(k_totals, dim3(1), dim3(1024), 0, s, b.state, b.cost_part, ctx->n_blk_asm, b.pred_part,
                       b.step_part, ctx->n_pred, b.gn_part, ctx->plan.active() ? ctx->plan.n_chunks : ctx->chain.n_nodes,   // (chunked: one entry per run)
                       b.nbehind, b.totals,
                       with_step ? 1 : 0, b.cst, b.numeric_err, fused_control, b.trunc_eps2, ctx->n_trunc);
  }
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

extern "C" {

size_t acino_fte_workspace_bytes(const acino_fte_params* p) {
  if (!p || p->n_frames < 1) return 0;
  Layout lay;
  lay.build(p);
  return carve(p, nullptr, nullptr, nullptr, lay) + 256;
}

int acino_fte_create(acino_fte_ctx** out, const acino_fte_params* p, const double* d_det, const double* d_cams24,
                     void* d_workspace, size_t workspace_bytes, void* stream) {
  ACINO_REQUIRE(out != nullptr, "out");
  *out = nullptr;
  int rc = validate(p);
  if (rc) return rc;
  ACINO_REQUIRE(d_det && d_cams24 && d_workspace, "null buffer");
  ACINO_REQUIRE(((uintptr_t)d_workspace & 255) == 0, "workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  acino_fte_ctx* ctx = new (std::nothrow) acino_fte_ctx();
  if (!ctx) {
    set_error("out of host memory");
    return ACINO_ERR_INVALID_ARG;
  }
  ctx->lam0 = p->lam0;
  Layout lay;
  lay.build(p);
  const size_t need = carve(p, (char*)d_workspace, &ctx->b, &ctx->chain, lay, &ctx->sepchain);
  ctx->plan = lay.plan;
  ctx->sched = std::move(lay.sched);
  if (need > workspace_bytes) {
    set_error("workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    delete ctx;
    return ACINO_ERR_WORKSPACE;
  }
  ctx->ws_bytes = need;
  ctx->d_det = d_det;
  double h_cams[ACINO_MAX_CAMS * ACINO_CAM_STRIDE];
  hipError_t e = hipMemcpyAsync(h_cams, d_cams24, sizeof(double) * ACINO_CAM_STRIDE * p->n_cams,
                                hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) {
    set_error("reading camera records failed: %s", hipGetErrorString(e));
    delete ctx;
    return ACINO_ERR_HIP;
  }
  fill_const(p, h_cams, &ctx->h);
  acino_fte_state st;
  memset(&st, 0, sizeof(st));
  st.lam = p->lam0;
  st.nu = 2.0;
  e = hipMemcpyAsync(ctx->b.cst, &ctx->h, sizeof(FteConst), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(ctx->b.state, &st, sizeof(st), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemsetAsync(ctx->b.nbehind, 0, 4 * sizeof(int), s);
  if (e == hipSuccess) e = hipMemsetAsync(ctx->b.x[0], 0, sizeof(double) * (p->n_frames + 2 * HALO) * NP, s);
  if (e == hipSuccess) e = hipMemsetAsync(ctx->b.x[1], 0, sizeof(double) * (p->n_frames + 2 * HALO) * NP, s);
  if (e == hipSuccess && !ctx->sched.elim.empty())
    e = hipMemcpyAsync(ctx->b.sched, ctx->sched.elim.data(), sizeof(int) * ctx->sched.elim.size(),
                       hipMemcpyHostToDevice, s);
  if (e == hipSuccess && !ctx->sched.remain.empty())
    e = hipMemcpyAsync(ctx->b.sched + ctx->sched.elim.size(), ctx->sched.remain.data(),
                       sizeof(int) * ctx->sched.remain.size(), hipMemcpyHostToDevice, s);
  if (e == hipSuccess && !ctx->sched.tail.empty())
    e = hipMemcpyAsync(ctx->b.sched + ctx->sched.elim.size() + ctx->sched.remain.size(), ctx->sched.tail.data(),
                       sizeof(int) * ctx->sched.tail.size(), hipMemcpyHostToDevice, s);
  if (e == hipSuccess && !ctx->sched.pairs.empty())
    e = hipMemcpyAsync(ctx->b.sched + ctx->sched.elim.size() + ctx->sched.remain.size() + ctx->sched.tail.size(),
                       ctx->sched.pairs.data(), sizeof(int) * ctx->sched.pairs.size(), hipMemcpyHostToDevice, s);
  // (fused narrow levels: their entry tables follow the tail's progress counter)
  const size_t fused_off = ctx->sched.elim.size() + ctx->sched.remain.size() + ctx->sched.tail.size() + ctx->sched.pairs.size() + 4;
  {
    std::vector<int> fz;
    fz.insert(fz.end(), ctx->sched.elim6.begin(), ctx->sched.elim6.end());
    fz.insert(fz.end(), ctx->sched.iso_loc.begin(), ctx->sched.iso_loc.end());
    fz.insert(fz.end(), ctx->sched.fold.begin(), ctx->sched.fold.end());
    if (e == hipSuccess && !fz.empty()) {
      e = hipMemcpyAsync(ctx->b.sched + fused_off, fz.data(), sizeof(int) * fz.size(), hipMemcpyHostToDevice, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);       // (fz is a local)
    }
  }
  if (e == hipSuccess) e = hipMemsetAsync(ctx->b.trunc_eps2, 0, sizeof(double) * (7 * (ctx->sched.pairs.size() / 2 + 1) + 1), s);
  if (e == hipSuccess && ctx->b.st_flags) e = hipMemsetAsync(ctx->b.st_flags, 0, sizeof(int) * ((size_t)ctx->b.n_st_flags + 2), s);
  if (e == hipSuccess && ctx->b.st_ll) e = hipMemsetAsync(ctx->b.st_ll, 0, sizeof(unsigned long long) * ctx->b.n_st_ll, s);
  // (the runs write the contribution AL of every separator that has a run on its right: a right pin has none - zero once)
  if (e == hipSuccess && ctx->plan.active() && ctx->plan.n_sep > 0)
    e = hipMemsetAsync(const_cast<double*>(ctx->sepchain.AL0), 0, sizeof(double) * (size_t)ctx->plan.n_sep * BS * BS, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) {
    set_error("context upload failed: %s", hipGetErrorString(e));
    delete ctx;
    return ACINO_ERR_HIP;
  }
  {
    BcrChain& red = ctx->plan.active() ? ctx->sepchain : ctx->chain;   // the chain the schedule reduces
    red.d_elim = ctx->b.sched;
    red.d_remain = ctx->b.sched + ctx->sched.elim.size();
    if (!ctx->sched.tail.empty() && !p->shared_gpu) {
      red.d_tail = red.d_remain + ctx->sched.remain.size();
      red.d_done = ctx->b.sched + ctx->sched.elim.size() + ctx->sched.remain.size() + ctx->sched.tail.size() +
                  ctx->sched.pairs.size();
    }
    if (!ctx->sched.pairs.empty()) {
      red.d_pairs = ctx->b.sched + ctx->sched.elim.size() + ctx->sched.remain.size() + ctx->sched.tail.size();
      red.n_pairs = (int)(ctx->sched.pairs.size() / 2);
    }
    ctx->n_trunc = red.n_pairs;
    red.refine_buf = ctx->b.refine_buf;
    if (ctx->sched.fused_levels) {
      red.d_elim6 = ctx->b.sched + fused_off;
      red.d_iso_loc = red.d_elim6 + ctx->sched.elim6.size();
      red.d_fold = red.d_iso_loc + ctx->sched.iso_loc.size();
    }
    // one persistent launch for the separator chain's back-substitution: its isolated workgroups wait for each other, so it
    // is kept off GPUs that other spin-waiting kernels may share (shared_gpu: batched clips, several ranks on one device)
    // - and off devices that cannot hold all of its workgroups at once (occupancy x compute units of THIS device: a CU mask, a
    // partitioned GPU): those take the per-level kernels
    red.st_flags = nullptr;
    if (ctx->plan.active() && !p->shared_gpu && !getenv("ACINO_NO_SEP_TAIL") && ctx->b.st_flags) {
      int blocks = 0, capacity = 0;
      int rc = bcr_set_func_attributes();                // (the occupancy query needs the kernel's LDS attribute in place)
      if (!rc) rc = bcr_sep_tail_fit(ctx->sched, &blocks, &capacity);
      if (rc) {
        delete ctx;
        return rc;
      }
      if (const char* e = getenv("ACINO_SEP_TAIL_CAPACITY")) capacity = atoi(e);   // (tests: pretend a smaller device)
      if (blocks > 0 && blocks <= capacity) {
        red.st_flags = ctx->b.st_flags;
        red.n_st_flags = ctx->b.n_st_flags;
        red.st_ll = ctx->b.st_ll;
      }
    }
  }
  if (ctx->plan.active()) {
    ctx->sep = SepView{ctx->sepchain.D, ctx->sepchain.Cpl, const_cast<double*>(ctx->sepchain.AL0), ctx->sepchain.b};
    ctx->sep.flags = ctx->sepchain.st_flags;
    ctx->sep.n_flags = ctx->sepchain.st_flags ? ctx->b.n_st_flags : 0;
  }
  ctx->n_blk_asm = n_assemble_blocks(p->n_frames);
  ctx->n_blk_trial = (int)(((size_t)p->n_frames * NP + 255) / 256);
  ctx->n_pred = ctx->plan.active() ? ctx->plan.n_chunks : ctx->n_blk_trial;
  rc = bcr_set_func_attributes();
  if (!rc) rc = chunk_set_func_attributes();
  if (rc) {
    delete ctx;
    return rc;
  }
  *out = ctx;
  return ACINO_OK;
}

// The linear solver's layout for these parameters: out[0] = nodes per run (0: block cyclic reduction over the whole chain),
// out[1] = runs, out[2] = separators, out[3] = reduction levels of the chain that is reduced (separators or whole chain)
// when nothing is truncated.
int acino_fte_plan(const acino_fte_params* p, int32_t* out) {
  ACINO_REQUIRE(p && out, "null");
  int rc = validate(p);
  if (rc) return rc;
  acino_fte_params q = *p;
  q.bcr_levels = 0;
  q.refine_sweeps = 0;
  Layout lay;
  lay.build(&q);
  out[0] = lay.plan.m;
  out[1] = lay.plan.n_chunks;
  out[2] = lay.plan.n_sep;
  out[3] = (int32_t)lay.sched.levels.size();
  return ACINO_OK;
}

int acino_fte_destroy(acino_fte_ctx* ctx) {
  if (ctx && ctx->gexec) (void)hipGraphExecDestroy(ctx->gexec);
  if (ctx)
    for (auto& g : ctx->seg)
      if (g.exec) (void)hipGraphExecDestroy(g.exec);
  delete ctx;
  return ACINO_OK;
}

int acino_fte_graphs_active(acino_fte_ctx* ctx) {
  if (!ctx) return 0;
  int m = ctx->gexec ? 16 : 0;
  for (int i = 0; i < 4; ++i) m |= ctx->seg[i].exec ? (1 << i) : 0;
  return m;
}

int acino_fte_enable_graph(acino_fte_ctx* ctx, int on) {
  ACINO_REQUIRE(ctx, "null");
  ctx->graph_on = on != 0;
  return ACINO_OK;
}

int acino_fte_load_x(acino_fte_ctx* ctx, const double* d_x0, void* stream) {
  ACINO_REQUIRE(ctx && d_x0, "null");
  hipStream_t s = (hipStream_t)stream;
  acino_fte_state st;
  memset(&st, 0, sizeof(st));
  st.lam = ctx->lam0;
  st.nu = 2.0;
  ACINO_HIP_CHECK(hipMemcpyAsync(ctx->b.state, &st, sizeof(st), hipMemcpyHostToDevice, s));
  ACINO_HIP_CHECK(hipStreamSynchronize(s));   // st lives on this stack frame
  ACINO_HIP_CHECK(hipMemsetAsync(ctx->b.nbehind, 0, 4 * sizeof(int), s));
  hipLaunchKernelGGL(k_copy_x_in, dim3(ctx->n_blk_trial), dim3(256), 0, s, ctx->b.cst, ctx->b.state, 0, d_x0,
                     ctx->b.x[0], ctx->b.x[1]);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

int acino_fte_set_halo(acino_fte_ctx* ctx, int which, const double* d_halo_l, const double* d_halo_r, void* stream) {
  ACINO_REQUIRE(ctx, "null");
  ACINO_REQUIRE(which == 0 || which == 1, "which");
  hipLaunchKernelGGL(k_set_halo, dim3(1), dim3(128), 0, (hipStream_t)stream, ctx->b.state, which, ctx->b.x[0],
                     ctx->b.x[1], ctx->h.n_frames, d_halo_l, d_halo_r);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

__global__ void k_restart_status(acino_fte_state* st, double lam0) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && st->status >= 1 && st->status <= 4) {
    if (st->status == 4) st->lam = lam0;      // ended because no damping gave descent: that damping is not a start value
    st->status = 0;
    st->nu = 2.0;
  }
}

int acino_fte_set_precision(acino_fte_ctx* ctx, int precision) {
  ACINO_REQUIRE(ctx, "null");
  ACINO_REQUIRE(precision >= ACINO_PREC_F64 && precision <= ACINO_PREC_BF16_RES, "precision");
  ctx->h.precision = precision;            // the assembly kernel is chosen on the host; the device block does not need it
  if (ctx->gexec) {
    (void)hipGraphExecDestroy(ctx->gexec);
    ctx->gexec = nullptr;
  }
  for (auto& g : ctx->seg)
    if (g.exec) {
      (void)hipGraphExecDestroy(g.exec);
      g.exec = nullptr;
    }
  return ACINO_OK;
}

int acino_fte_reevaluate(acino_fte_ctx* ctx, void* stream) {
  ACINO_REQUIRE(ctx, "null");
  ACINO_REQUIRE(!ctx->h.pin_left && !ctx->h.pin_right, "single-GPU contexts");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_restart_status, dim3(1), dim3(64), 0, s, ctx->b.state, ctx->lam0);
  ACINO_LAUNCH_CHECK();
  return eval_iterate(ctx, 0, true, false, false, s, 1);      // assembly of the current iterate + cost (init-style control)
}

int acino_fte_copy_frames(acino_fte_ctx* ctx, int which, int import, int first, int n, double* d_buf, void* stream) {
  ACINO_REQUIRE(ctx && d_buf, "null");
  ACINO_REQUIRE(which == 0 || which == 1, "which");
  ACINO_REQUIRE(n >= 0 && first >= -HALO && first + n <= ctx->h.n_frames + HALO, "frame range");
  if (n == 0) return ACINO_OK;
  hipLaunchKernelGGL(k_copy_frames, dim3((unsigned)((n * NP + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     ctx->b.state, which, import ? 1 : 0, ctx->b.x[0], ctx->b.x[1], first, n, d_buf);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

int acino_fte_eval(acino_fte_ctx* ctx, int which, void* stream) {
  ACINO_REQUIRE(ctx, "null");
  ACINO_REQUIRE(which == 0 || which == 1, "which");
  return eval_iterate(ctx, which, true, which == 1, which == 1, (hipStream_t)stream);
}

int acino_fte_export_partials(acino_fte_ctx* ctx, double* d_partial, void* stream) {
  ACINO_REQUIRE(ctx && d_partial, "null");
  hipLaunchKernelGGL(k_copy_vec, dim3(1), dim3(256), 0, (hipStream_t)stream, ctx->b.totals, d_partial, 8);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

int acino_fte_control(acino_fte_ctx* ctx, const double* d_total, int init, void* stream) {
  ACINO_REQUIRE(ctx, "null");
  {
    ProfSpan sp(&ctx->prof, PC_CONTROL, (hipStream_t)stream);
    hipLaunchKernelGGL(k_control, dim3(1), dim3(64), 0, (hipStream_t)stream, ctx->b.cst, ctx->b.state,
                       d_total ? d_total : ctx->b.totals, ctx->b.numeric_err, init);
  }
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

int acino_fte_set_x(acino_fte_ctx* ctx, const double* d_x0, void* stream) {
  int rc = acino_fte_load_x(ctx, d_x0, stream);
  if (rc) return rc;
  return eval_iterate(ctx, 0, true, false, false, (hipStream_t)stream, 1);
}

int acino_fte_reduce_local(acino_fte_ctx* ctx, void* stream) {
  ACINO_REQUIRE(ctx, "null");
  hipStream_t s = (hipStream_t)stream;
  const Buffers& b = ctx->b;
  // the damped system is built inside the level-0 kernels (no separate set-up launch)
  if (ctx->plan.active())
    return chunk_reduce(ctx->chain, ctx->plan, ctx->sep, ctx->sepchain, ctx->sched, b.cst, b.numeric_err, &b.state->status, s,
                        &ctx->prof);
  return bcr_reduce(ctx->chain, ctx->sched, b.cst, b.numeric_err, &b.state->status, s, &ctx->prof);
}

int acino_fte_export_separators(acino_fte_ctx* ctx, double* d_sep, int rank, int world, void* stream) {
  ACINO_REQUIRE(ctx && d_sep, "null");
  ACINO_REQUIRE(world >= 2 && rank >= 0 && rank < world, "rank/world");
  ACINO_REQUIRE((ctx->h.pin_left != 0) == (rank > 0) && (ctx->h.pin_right != 0) == (rank + 1 < world),
                "pins must match the rank position");
  double* rec_left = ctx->h.pin_left ? d_sep + (size_t)(rank - 1) * ACINO_SEP_DOUBLES : nullptr;
  double* rec_right = ctx->h.pin_right ? d_sep + (size_t)rank * ACINO_SEP_DOUBLES : nullptr;
  // (chunked solver: the pins are the first / last node of the separator chain)
  const BcrChain& pc = ctx->plan.active() ? ctx->sepchain : ctx->chain;
  hipLaunchKernelGGL(k_export_sep, dim3(8), dim3(256), 0, (hipStream_t)stream, pc, 0, pc.n_nodes - 1, rec_left, rec_right);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

size_t acino_sep_scratch_bytes(int n_sep) {
  if (n_sep < 1) return 0;
  BcrSchedule sch;
  // (GENERAL 80 x 80 blocks: the fused narrow levels of seplevel.hip need the identity padding of FTE nodes - their right-hand
  //  side rides in the padding column - so this entry point keeps the per-phase kernels)
  sch.build(n_sep, false, false);
  size_t ints = sch.elim.size() + sch.remain.size() + sch.fused_ints() + 8;
  return 6 * align_up((size_t)n_sep * BS * BS * sizeof(double)) + align_up(ints * sizeof(int)) + 1024;   // D U Cpl Wl Wr, b
}

namespace acino {
static const BcrSchedule& sep_schedule(int n_sep) {
  static thread_local std::map<int, BcrSchedule> cache;   // schedules are immutable once built
  auto it = cache.find(n_sep);
  if (it == cache.end()) {
    BcrSchedule sch;
    sch.build(n_sep, false, false);
    it = cache.emplace(n_sep, std::move(sch)).first;
  }
  return it->second;
}
static BcrChain sep_chain(void* d_scratch, int n_sep, const BcrSchedule& sch) {
  Carver c{(char*)d_scratch, 0};
  BcrChain ch;
  ch.n_nodes = n_sep;
  ch.D = c.take<double>((size_t)n_sep * BS * BS);
  ch.U = c.take<double>((size_t)n_sep * BS * BS);
  ch.Cpl = c.take<double>((sch.fused_levels ? 2 : 1) * (size_t)n_sep * BS * BS);
  ch.Wl = c.take<double>((size_t)n_sep * BS * BS);
  ch.Wr = c.take<double>((size_t)n_sep * BS * BS);
  ch.b = c.take<double>((size_t)n_sep * BS);
  if (sch.fused_levels) {
    ch.SL = c.take<double>((size_t)n_sep * BS * BS);
    ch.SR = c.take<double>((size_t)n_sep * BS * BS);
    ch.Y = c.take<double>((size_t)n_sep * BS);
  }
  int* d_sched = c.take<int>(sch.elim.size() + sch.remain.size() + sch.fused_ints() + 8);
  ch.d_elim = d_sched;
  ch.d_remain = d_sched + sch.elim.size();
  if (sch.fused_levels) {
    ch.d_elim6 = ch.d_remain + sch.remain.size();
    ch.d_iso_loc = ch.d_elim6 + sch.elim6.size();
    ch.d_fold = ch.d_iso_loc + sch.iso_loc.size();
  }
  ch.d_tail = nullptr;            // (the separator chain keeps the per-level kernels)
  ch.d_done = nullptr;
  ch.implicit_couplings = 0;
  ch.dbg = nullptr;
  ch.st = nullptr;
  ch.x0 = ch.x1 = ch.g0 = ch.g1 = ch.H0 = ch.H1 = nullptr;
  ch.gn_part = nullptr;
  return ch;
}
// The (static) schedule is written into the caller's scratch by a kernel that carries it as an argument: no
// host-to-device copy, so the write is stream-ordered, capturable and never synchronises (chains of <= 128 separators;
// longer ones fall back to a copy and are not graph-captured).
struct SchedArg {
  int n;
  int v[960];
};
__global__ void k_write_schedule(SchedArg a, int* __restrict__ dst) {
  for (int i = threadIdx.x; i < a.n; i += blockDim.x) dst[i] = a.v[i];
}
static bool sep_schedule_fits_arg(const BcrSchedule& sch) { return sch.elim.size() + sch.remain.size() + sch.fused_ints() <= 960; }
static std::vector<int> sep_schedule_ints(const BcrSchedule& sch) {      // in the order sep_chain lays them out
  std::vector<int> v(sch.elim);
  v.insert(v.end(), sch.remain.begin(), sch.remain.end());
  v.insert(v.end(), sch.elim6.begin(), sch.elim6.end());
  v.insert(v.end(), sch.iso_loc.begin(), sch.iso_loc.end());
  v.insert(v.end(), sch.fold.begin(), sch.fold.end());
  return v;
}
static int sep_upload_schedule(const BcrChain& ch, const BcrSchedule& sch, hipStream_t s) {
  int* d_sched = const_cast<int*>(ch.d_elim);
  const std::vector<int> v = sep_schedule_ints(sch);
  if (sep_schedule_fits_arg(sch)) {
    SchedArg a;
    a.n = (int)v.size();
    std::copy(v.begin(), v.end(), a.v);
    hipLaunchKernelGGL(k_write_schedule, dim3(1), dim3(256), 0, s, a, d_sched);
    ACINO_LAUNCH_CHECK();
    return ACINO_OK;
  }
  ACINO_HIP_CHECK(hipMemcpyAsync(d_sched, v.data(), sizeof(int) * v.size(), hipMemcpyHostToDevice, s));
  ACINO_HIP_CHECK(hipStreamSynchronize(s));       // (v is a local; chains this long are not graph-captured)
  return ACINO_OK;
}
static int sep_launch(const BcrChain& ch, const BcrSchedule& sch, const double* d_sep, int n_sep, double* d_sep_x,
                      hipStream_t s) {
  hipLaunchKernelGGL(k_import_sep, dim3(8, n_sep), dim3(256), 0, s, d_sep, n_sep, ch);
  ACINO_LAUNCH_CHECK();
  int rc = bcr_reduce(ch, sch, nullptr, nullptr, nullptr, s);
  if (rc) return rc;
  rc = bcr_backsub(ch, sch, nullptr, nullptr, s);
  if (rc) return rc;
  hipLaunchKernelGGL(k_copy_vec, dim3((n_sep * BS + 255) / 256), dim3(256), 0, s, ch.b, d_sep_x, n_sep * BS);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}
}  // namespace acino

int acino_solve_separators(const double* d_sep, int n_sep, double* d_sep_x, void* d_scratch, size_t scratch_bytes,
                           void* stream) {
  ACINO_REQUIRE(d_sep && d_sep_x && d_scratch, "null");
  ACINO_REQUIRE(n_sep >= 1 && n_sep <= 1024, "n_sep");
  ACINO_REQUIRE(scratch_bytes >= acino_sep_scratch_bytes(n_sep), "separator scratch too small");
  ACINO_REQUIRE(((uintptr_t)d_scratch & 255) == 0, "scratch must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const BcrSchedule& sch = sep_schedule(n_sep);
  BcrChain ch = sep_chain(d_scratch, n_sep, sch);
  if (int rc = bcr_set_func_attributes()) return rc;
  if (int rc = sep_upload_schedule(ch, sch, s)) return rc;
  return sep_launch(ch, sch, d_sep, n_sep, d_sep_x, s);
}

int acino_fte_backsub_local(acino_fte_ctx* ctx, const double* d_sep_x, int rank, int world, void* stream) {
  ACINO_REQUIRE(ctx, "null");
  hipStream_t s = (hipStream_t)stream;
  if (ctx->h.pin_left) {
    ACINO_REQUIRE(d_sep_x && rank >= 1, "separator solution");
    hipLaunchKernelGGL(k_set_node_x, dim3(1), dim3(128), 0, s, ctx->plan.active() ? ctx->sepchain : ctx->chain, 0,
                       d_sep_x + (size_t)(rank - 1) * BS);
    ACINO_LAUNCH_CHECK();
  }
  if (ctx->h.pin_right) {
    ACINO_REQUIRE(d_sep_x && rank + 1 < world, "separator solution");
    const BcrChain& pc = ctx->plan.active() ? ctx->sepchain : ctx->chain;
    hipLaunchKernelGGL(k_set_node_x, dim3(1), dim3(128), 0, s, pc, pc.n_nodes - 1, d_sep_x + (size_t)rank * BS);
    ACINO_LAUNCH_CHECK();
  }
  if (ctx->plan.active())
    return chunk_backsub(ctx->chain, ctx->plan, ctx->sep, ctx->sepchain, ctx->sched, ctx->b.cst, ctx->b.numeric_err,
                         &ctx->b.state->status, s, &ctx->prof,
                         TrialOut{ctx->b.hd[0], ctx->b.hd[1], ctx->b.pred_part, ctx->b.step_part});
  return bcr_backsub(ctx->chain, ctx->sched, ctx->b.cst, &ctx->b.state->status, s, &ctx->prof, ctx->b.numeric_err);
}

int acino_fte_trial(acino_fte_ctx* ctx, void* stream) {
  ACINO_REQUIRE(ctx, "null");
  if (ctx->plan.active()) return ACINO_OK;      // chunked solver: the back-substitution has formed the trial iterate already
  const Buffers& b = ctx->b;
  {
    ProfSpan sp(&ctx->prof, PC_TRIAL, (hipStream_t)stream, ctx->h.n_frames);
    hipLaunchKernelGGL(k_trial, dim3(ctx->n_blk_trial), dim3(256), 0, (hipStream_t)stream, b.cst, b.state, b.x[0],
                       b.x[1], b.g[0], b.g[1], b.hd[0], b.hd[1], ctx->chain.b, b.pred_part, b.step_part);
  }
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

int acino_fte_export_edges(acino_fte_ctx* ctx, int which, double* d_edge, void* stream) {
  ACINO_REQUIRE(ctx && d_edge, "null");
  ACINO_REQUIRE(ctx->h.n_frames >= 3, "need >= 3 frames");
  hipLaunchKernelGGL(k_export_edges, dim3(1), dim3(128), 0, (hipStream_t)stream, ctx->b.state, which, ctx->b.x[0],
                     ctx->b.x[1], ctx->h.n_frames, d_edge);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

// ---- the four phases of a sharded iteration ------------------------------------------------
// A phase is a fixed launch sequence on caller-owned buffers; with graphs enabled it is captured once per buffer set
// and replayed, so a rank issues 4 graph launches + 3 collectives per iteration instead of ~50 kernel launches.
extern "C++" {
template <class F>
static int run_phase(acino_fte_ctx* ctx, int id, const uintptr_t (&key)[6], hipStream_t s, F&& body) {
  if (!(ctx->graph_on && !ctx->prof.on && s != nullptr)) return body();
  acino_fte_ctx::SegGraph& g = ctx->seg[id];
  if (!g.exec || g.stream != s || memcmp(g.key, key, sizeof(key)) != 0) {
    if (g.exec) {
      (void)hipGraphExecDestroy(g.exec);
      g.exec = nullptr;
    }
    // capture; if anything about it fails (another thread of the process touching the runtime, an unsupported
    // call) the phase simply runs eagerly from now on - nothing captured has executed
    bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
      const int rc = body();
      hipGraph_t graph = nullptr;
      const hipError_t e = hipStreamEndCapture(s, &graph);
      ok = rc == ACINO_OK && e == hipSuccess && graph != nullptr;
      if (ok) ok = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0) == hipSuccess;
      if (graph) (void)hipGraphDestroy(graph);
    }
    if (!ok) {
      (void)hipGetLastError();
      g.exec = nullptr;
      ctx->graph_on = false;
      return body();
    }
    g.stream = s;
    memcpy(g.key, key, sizeof(key));
  }
  ACINO_HIP_CHECK(hipGraphLaunch(g.exec, s));
  return ACINO_OK;
}
}  // extern "C++"

int acino_fte_shard_reduce(acino_fte_ctx* ctx, double* d_sep, int rank, int world, void* stream) {
  ACINO_REQUIRE(ctx && d_sep && world >= 2, "args");
  hipStream_t s = (hipStream_t)stream;
  const uintptr_t key[6] = {(uintptr_t)d_sep, (uintptr_t)rank, (uintptr_t)world, 0, 0, 0};
  return run_phase(ctx, 0, key, s, [&]() -> int {
    ACINO_HIP_CHECK(hipMemsetAsync(d_sep, 0, sizeof(double) * (size_t)(world - 1) * ACINO_SEP_DOUBLES, s));
    if (int rc = acino_fte_reduce_local(ctx, stream)) return rc;
    return acino_fte_export_separators(ctx, d_sep, rank, world, stream);
  });
}

int acino_fte_shard_solve(acino_fte_ctx* ctx, const double* d_sep, double* d_sep_x, void* d_scratch, size_t scratch_bytes,
                          double* d_edge_out, int rank, int world, void* stream) {
  ACINO_REQUIRE(ctx && d_sep && d_sep_x && d_scratch && d_edge_out && world >= 2, "args");
  const int n_sep = world - 1;
  ACINO_REQUIRE(scratch_bytes >= acino_sep_scratch_bytes(n_sep), "separator scratch too small");
  ACINO_REQUIRE(((uintptr_t)d_scratch & 255) == 0, "scratch must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const BcrSchedule& sch = sep_schedule(n_sep);
  const BcrChain ch = sep_chain(d_scratch, n_sep, sch);
  const uintptr_t key[6] = {(uintptr_t)d_sep, (uintptr_t)d_sep_x, (uintptr_t)d_scratch, (uintptr_t)d_edge_out,
                            (uintptr_t)rank, (uintptr_t)world};
  auto body = [&]() -> int {
    if (int rc = sep_upload_schedule(ch, sch, s)) return rc;
    if (int rc = sep_launch(ch, sch, d_sep, n_sep, d_sep_x, s)) return rc;
    if (int rc = acino_fte_backsub_local(ctx, d_sep_x, rank, world, stream)) return rc;
    if (int rc = acino_fte_trial(ctx, stream)) return rc;
    return acino_fte_export_edges(ctx, 1, d_edge_out, stream);
  };
  if (!sep_schedule_fits_arg(sch)) return body();     // (host copy inside: run eagerly)
  return run_phase(ctx, 1, key, s, body);
}

int acino_fte_shard_eval(acino_fte_ctx* ctx, int which, const double* d_all_edges, int rank, int world,
                         double* d_partial_out, void* stream) {
  ACINO_REQUIRE(ctx && d_all_edges && d_partial_out && world >= 2 && rank >= 0 && rank < world, "args");
  ACINO_REQUIRE(which == 0 || which == 1, "which");
  hipStream_t s = (hipStream_t)stream;
  const double* hl = rank > 0 ? d_all_edges + (size_t)(rank - 1) * 6 * NP + 3 * NP : nullptr;   // left rank's last 3 frames
  const double* hr = rank + 1 < world ? d_all_edges + (size_t)(rank + 1) * 6 * NP : nullptr;     // right rank's first 3
  const uintptr_t key[6] = {(uintptr_t)d_all_edges, (uintptr_t)d_partial_out, (uintptr_t)rank, (uintptr_t)world,
                            (uintptr_t)which, 0};
  return run_phase(ctx, 2, key, s, [&]() -> int {
    if (int rc = acino_fte_set_halo(ctx, which, hl, hr, stream)) return rc;
    if (int rc = acino_fte_eval(ctx, which, stream)) return rc;
    return acino_fte_export_partials(ctx, d_partial_out, stream);
  });
}

int acino_fte_shard_control(acino_fte_ctx* ctx, const double* d_all_partials, int world, int init, void* stream) {
  ACINO_REQUIRE(ctx && d_all_partials && world >= 1, "args");
  hipStream_t s = (hipStream_t)stream;
  const uintptr_t key[6] = {(uintptr_t)d_all_partials, (uintptr_t)world, (uintptr_t)init, 0, 0, 0};
  return run_phase(ctx, 3, key, s, [&]() -> int {
    ProfSpan sp(&ctx->prof, PC_CONTROL, s);
    hipLaunchKernelGGL(k_control_gathered, dim3(1), dim3(64), 0, s, ctx->b.cst, ctx->b.state, d_all_partials, world,
                       ctx->b.numeric_err, init);
    ACINO_LAUNCH_CHECK();
    return ACINO_OK;
  });
}

static int step_eager(acino_fte_ctx* ctx, void* stream) {
  int rc = acino_fte_reduce_local(ctx, stream);
  if (rc) return rc;
  rc = acino_fte_backsub_local(ctx, nullptr, 0, 1, stream);
  if (rc) return rc;
  rc = acino_fte_trial(ctx, stream);
  if (rc) return rc;
  return eval_iterate(ctx, 1, true, true, true, (hipStream_t)stream, 0);   // assembly + sums + accept/reject
}

int acino_fte_step(acino_fte_ctx* ctx, void* stream) {
  ACINO_REQUIRE(ctx, "null");
  ACINO_REQUIRE(!ctx->h.pin_left && !ctx->h.pin_right, "sharded contexts are stepped by the multi-GPU driver");
  hipStream_t s = (hipStream_t)stream;
  // The step is a fixed launch sequence (buffer selection and the accept/reject decision live on the
  // device), so it can be captured once and replayed: ~40 launch boundaries shrink to graph-node edges.
  if (ctx->graph_on && !ctx->prof.on && s != nullptr) {
    if (!ctx->gexec || ctx->gstream != s) {
      if (ctx->gexec) {
        (void)hipGraphExecDestroy(ctx->gexec);
        ctx->gexec = nullptr;
      }
      ACINO_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      int rc = step_eager(ctx, stream);
      hipGraph_t graph = nullptr;
      hipError_t e = hipStreamEndCapture(s, &graph);
      if (rc) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
      }
      if (e != hipSuccess) {
        set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e));
        return ACINO_ERR_HIP;
      }
      e = hipGraphInstantiate(&ctx->gexec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (e != hipSuccess) {
        ctx->gexec = nullptr;
        set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
        return ACINO_ERR_HIP;
      }
      ctx->gstream = s;
    }
    ACINO_HIP_CHECK(hipGraphLaunch(ctx->gexec, s));
    return ACINO_OK;
  }
  return step_eager(ctx, stream);
}

int acino_fte_get_state(acino_fte_ctx* ctx, acino_fte_state* out, void* stream) {
  ACINO_REQUIRE(ctx && out, "null");
  ACINO_HIP_CHECK(hipMemcpyAsync(out, ctx->b.state, sizeof(*out), hipMemcpyDeviceToHost, (hipStream_t)stream));
  ACINO_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return ACINO_OK;
}

int acino_fte_solve(acino_fte_ctx* ctx, int max_iter, acino_fte_state* out, void* stream) {
  ACINO_REQUIRE(ctx, "null");
  ACINO_REQUIRE(max_iter >= 0, "max_iter");
  acino_fte_state st;
  // (a context continues where its last solve stopped: the device's iteration counter is cumulative.  What the fall-back below
  //  has left of THIS call's budget is counted from the value at entry - only read where the fall-back can happen.)
  int iter_at_entry = 0;
  if (ctx->sepchain.st_flags && max_iter > 0) {
    if (int rc = acino_fte_get_state(ctx, &st, stream)) return rc;
    iter_at_entry = st.iter;
  }
  for (int it = 0; it < max_iter; ++it) {
    int rc = acino_fte_step(ctx, stream);
    if (rc) return rc;
    // the device stops by itself; the host peeks now and then to stop launching no-ops - not before a stop is plausible
    // (a synchronisation drains the launch queue: ~50 us of idle GPU each; a no-op step costs about as much)
    if (it >= 11 && ((it - 11) & 3) == 0) {
      rc = acino_fte_get_state(ctx, &st, stream);
      if (rc) return rc;
      if (st.status != 0) break;
    }
  }
  int rc = acino_fte_get_state(ctx, &st, stream);
  if (rc) return rc;
  int ne = 0;
  if (st.status == 6) {
    ACINO_HIP_CHECK(hipMemcpyAsync(&ne, ctx->b.numeric_err, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    ACINO_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  }
  if (st.status == 6 && (ne & 8) && ctx->sepchain.st_flags && max_iter > 0) {
    // the single-launch back-substitution of the separator chain (k_sep_tail: numeric_err bit 3) waited in vain for another
    // workgroup (something else holds the compute units it counted on): the refused step changed nothing but the counters -
    // fall back to the per-level kernels for the rest of this context's life and go on from the same iterate.  (The other
    // bounded wait - the d_done tail of bcr.hip - has no such alternative: status 6 is returned.)
    ctx->sepchain.st_flags = nullptr;
    ctx->sep.flags = nullptr;
    ctx->sep.n_flags = 0;
    if (ctx->gexec) {
      (void)hipGraphExecDestroy(ctx->gexec);
      ctx->gexec = nullptr;
    }
    hipStream_t s = (hipStream_t)stream;
    st.status = 0;
    st.iter -= 1;
    ACINO_HIP_CHECK(hipMemcpyAsync(ctx->b.state, &st, sizeof(st), hipMemcpyHostToDevice, s));
    ACINO_HIP_CHECK(hipMemsetAsync(ctx->b.numeric_err, 0, sizeof(int), s));
    ACINO_HIP_CHECK(hipMemsetAsync(ctx->b.st_flags, 0, sizeof(int) * (size_t)ctx->b.n_st_flags, s));
    ACINO_HIP_CHECK(hipStreamSynchronize(s));
    const int done = st.iter - iter_at_entry;          // steps of this call that were applied before the refused one
    return acino_fte_solve(ctx, max_iter > done ? max_iter - done : 1, out, stream);
  }
  if (out) *out = st;
  if (st.status == 5) {
    set_error("non-positive pivot in the block factorisation (system not positive definite)");
    return ACINO_ERR_NUMERIC;
  }
  if (st.status == 6) {
    set_error("the fused back-substitution tail timed out waiting for lower workgroups (scheduling order not as assumed); "
              "create the context with shared_gpu = 1 to use the per-level kernels");
    return ACINO_ERR_HIP;
  }
  // (status 7 - dropped couplings of an incomplete reduction above trunc_tol - is returned in the state: the caller
  //  re-creates the context with more levels and continues from the current iterate, acinoset_amd/fte.py does)
  return ACINO_OK;
}

int acino_fte_cost(acino_fte_ctx* ctx, const double* d_x, double* d_cost, void* stream) {
  ACINO_REQUIRE(ctx && d_x && d_cost, "null");
  hipStream_t s = (hipStream_t)stream;
  // evaluated in the TRIAL buffer (clobbers it; call between LM steps)
  hipLaunchKernelGGL(k_copy_x_in, dim3(ctx->n_blk_trial), dim3(256), 0, s, ctx->b.cst, ctx->b.state, 1, d_x,
                     ctx->b.x[0], ctx->b.x[1]);
  ACINO_LAUNCH_CHECK();
  int rc = eval_iterate(ctx, 1, false, false, false, s);
  if (rc) return rc;
  hipLaunchKernelGGL(k_copy_vec, dim3(1), dim3(256), 0, s, ctx->b.totals, d_cost, 1);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

int acino_fte_get_grad_hess(acino_fte_ctx* ctx, double* d_g, double* d_h, void* stream) {
  ACINO_REQUIRE(ctx, "null");
  const Buffers& b = ctx->b;
  hipLaunchKernelGGL(k_export_HG, dim3(std::max(1, ctx->n_blk_trial)), dim3(256), 0, (hipStream_t)stream, b.state,
                     b.g[0], b.g[1], b.H[0], b.H[1], (int64_t)ctx->h.n_frames, d_g, d_h);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

int acino_fte_get_result(acino_fte_ctx* ctx, double ts, double* d_x, double* d_pos, double* d_dx, double* d_ddx,
                         void* stream) {
  ACINO_REQUIRE(ctx, "null");
  hipStream_t s = (hipStream_t)stream;
  const Buffers& b = ctx->b;
  const int64_t n = ctx->h.n_frames;
  acino_fte_state st;
  int rc = acino_fte_get_state(ctx, &st, stream);
  if (rc) return rc;
  const double* xc = b.x[st.cur];
  if (d_x) {
    hipLaunchKernelGGL(k_copy_x_out, dim3(ctx->n_blk_trial), dim3(256), 0, s, b.state, 0, b.x[0], b.x[1], n, d_x);
    ACINO_LAUNCH_CHECK();
  }
  if (d_pos) {
    rc = launch_fk_active(xc, n, d_pos, s);
    if (rc) return rc;
  }
  if (d_dx || d_ddx) {
    ACINO_REQUIRE(ts > 0, "ts");
    hipLaunchKernelGGL(k_derivatives, dim3(ctx->n_blk_trial), dim3(256), 0, s, xc + HALO * NP, n, ts, d_dx, d_ddx);
    ACINO_LAUNCH_CHECK();
  }
  return ACINO_OK;
}

// Debug: phase timestamps (100 MHz wall clock) go to d_dbg[0..63] of a caller buffer of ACINO_DEBUG_STAMP_ENTRIES (72)
// entries; the selectors are d_dbg[64] = workgroup index and d_dbg[65] = reduction level (k_bcr_elim) / node of the run
// (k_chunk_sweep).
int acino_fte_debug_stamps(acino_fte_ctx* ctx, long long* d_dbg) {
  ACINO_REQUIRE(ctx, "null");
  ctx->chain.dbg = d_dbg;
  ctx->sepchain.dbg = d_dbg;       // (k_sep_level, k_sep_tail: selectors in their own comments)
  return ACINO_OK;
}

// Debug / test aid: copies an internal buffer to d_out (at most n doubles).  what: 0 chain.b (step per node), 1 sep.D,
// 2 sep.b, 3 sep.Cpl, 4 sep.AL, 6 chain.D (G of the interior nodes, lower tiles), 7 chain.Wl (chunked: f_k, [80] per node).
// (5 is not assigned.)
int acino_fte_debug_read(acino_fte_ctx* ctx, int what, double* d_out, int64_t n, void* stream) {
  ACINO_REQUIRE(ctx && d_out && n >= 0, "args");
  const size_t MB = (size_t)BS * BS;
  const size_t T = ctx->chain.n_nodes, S = ctx->plan.active() ? (size_t)ctx->plan.n_sep : 0;
  const double* src = nullptr;
  size_t cnt = 0;
  switch (what) {
    case 0: src = ctx->chain.b; cnt = T * BS; break;
    case 1: src = ctx->sepchain.D; cnt = S * MB; break;
    case 2: src = ctx->sepchain.b; cnt = S * BS; break;
    case 3: src = ctx->sepchain.Cpl; cnt = S * MB; break;
    case 4: src = ctx->sepchain.AL0; cnt = S * MB; break;
    case 6: src = ctx->chain.D; cnt = T * MB; break;
    case 7: src = ctx->chain.Wl; cnt = ctx->plan.active() ? T * BS : T * MB; break;
    default: ACINO_REQUIRE(false, "what");
  }
  if ((size_t)n < cnt) cnt = (size_t)n;
  if (cnt == 0 || !src) return ACINO_OK;
  ACINO_HIP_CHECK(hipMemcpyAsync(d_out, src, cnt * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return ACINO_OK;
}

int acino_fte_profile_begin(acino_fte_ctx* ctx) {
  ACINO_REQUIRE(ctx, "null");
  ctx->prof.on = true;
  ctx->prof.used = 0;
  ctx->prof.spans.clear();
  return ACINO_OK;
}

// Stops profiling, synchronises `stream` and returns per kernel class the summed HIP-event time (ms) and
// the number of launches.  Class order: setup, elim, update, backsub, trial, assemble, totals, control.
int acino_fte_profile_end(acino_fte_ctx* ctx, double* ms_by_class, int* launches_by_class, int64_t* units_by_class,
                          void* stream) {
  ACINO_REQUIRE(ctx && ms_by_class && launches_by_class, "null");
  ctx->prof.on = false;
  ACINO_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  for (int c = 0; c < PC_COUNT; ++c) {
    ms_by_class[c] = 0.0;
    launches_by_class[c] = 0;
    if (units_by_class) units_by_class[c] = 0;
  }
  for (const Profiler::Span& sp : ctx->prof.spans) {
    float ms = 0.f;
    ACINO_HIP_CHECK(hipEventElapsedTime(&ms, ctx->prof.pool[sp.a], ctx->prof.pool[sp.b]));
    ms_by_class[sp.cls] += ms;
    launches_by_class[sp.cls] += 1;
    if (units_by_class) units_by_class[sp.cls] += sp.units;
  }
  ctx->prof.spans.clear();
  ctx->prof.used = 0;
  return ACINO_OK;
}

int acino_fte_derivatives(const double* d_x, int64_t n_frames, double ts, double* d_dx, double* d_ddx, void* stream) {
  ACINO_REQUIRE(d_x && n_frames >= 0 && ts > 0, "args");
  if (n_frames == 0) return ACINO_OK;
  hipLaunchKernelGGL(k_derivatives, dim3((unsigned)((n_frames * NP + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, d_x, n_frames, ts, d_dx, d_ddx);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

int acino_fk_active(const double* d_xa, int64_t n_frames, double* d_pos, void* stream) {
  ACINO_REQUIRE(n_frames >= 0, "n_frames");
  if (n_frames == 0) return ACINO_OK;
  ACINO_REQUIRE(d_xa && d_pos, "null buffer");
  // d_xa[N][25] without halo rows: the kernel indexes (frame + halo) * stride with halo = 0
  return launch_fk_active(d_xa - HALO * NP, n_frames, d_pos, (hipStream_t)stream);
}

}  // extern "C"
