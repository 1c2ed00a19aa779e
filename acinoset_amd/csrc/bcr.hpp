// Block cyclic reduction on a block-tridiagonal SPD chain of 80x80 fp64 blocks (gfx950).
#pragma once
#include <vector>

#include "fte_kernels.hpp"

namespace acino {

struct BcrLevel {
  int n_elim, n_remain;
  int elim_off, remain_off;  // offsets (in entries) into the device schedule arrays
  bool adjacent;             // some eliminated node still has a neighbour at original distance 1 (implicit coupling)
  bool isolated = false;     // last level of an INCOMPLETE reduction: every remaining node solved on its own
  // fused level (k_sep_level, seplevel.hip): elimination AND the Schur products of a narrow level in one launch of T
  // workgroups per node; entries of 6 ints at elim6[6 * e6_off ..]
  bool fused = false;
  int T = 0, e6_off = 0;
};

// Host-side elimination schedule for a chain of n nodes; pinned ends are never eliminated.
struct BcrSchedule {
  std::vector<BcrLevel> levels;
  std::vector<int> elim;    // 3 ints per entry: node, left, right        (-1 = none)
  std::vector<int> remain;  // 4 ints per entry: node, elim-left, elim-right, new right neighbour
  // back-substitution "tail": the deepest tail_levels levels (<= 128 nodes together) as ONE launch, deepest first;
  // 4 ints per entry: node, left, right, number of entries that must be finished before this one may start
  std::vector<int> tail;
  int tail_levels = 0;
  // Incomplete reduction (max_levels > 0): after max_levels levels the couplings between the remaining nodes are
  // DROPPED and every remaining node is factorised on its own.  pairs = (left node, right node) of every dropped
  // coupling block (stored at Cpl[left]); their normalised size is measured on the device (k_bcr_trunc_check).
  std::vector<int> pairs;
  size_t ints() const { return elim.size() + remain.size() + tail.size() + pairs.size() + 4 + fused_ints(); }   // (+ the tail's progress counter)
  // refine > 0 (incomplete reductions only): block-Jacobi sweeps over the isolated nodes that re-introduce the dropped
  // couplings after the truncated solve (k_bcr_refine); the isolated level then stays out of the fused tail.
  int refine = 0;
  // ---- fused narrow levels (chains that carry the S / Y / second coupling arrays: BcrChain::SL != nullptr) ----
  // A fused level never touches the diagonal block of a REMAINING node: the Schur products of an eliminated node i go to
  // running sums SR[l] (what l receives from its right side) and SL[r], the new coupling block(r, l) to slot n + i of the
  // coupling array.  Whoever consumes a node later (its own elimination, the isolated level, the fold of a pin) forms
  // D + AL + SL + SR.  elim6 entry: node, left, right, flags, location of block(node, left), location of block(right, node)
  //   flags: 1 SL[node] valid, 2 SR[node] valid, 4 SR[left] holds earlier contributions (add to them), 8 the same for SL[right]
  bool fused_levels = false;
  std::vector<int> elim6;
  std::vector<int> iso_loc;   // isolated level, 2 ints per entry: flags, location of block(next isolated node, this node)
  // nodes whose sums must be materialised for kernels that know nothing of them (pins: read by the export; the isolated level
  // when k_sep_tail does not apply): 3 ints per entry - node, flags, location of its right coupling (copied to slot `node`; -1: none)
  std::vector<int> fold;
  int n_fold_pins = 0;        // the first n_fold_pins entries of fold are the pins, the rest the isolated level
  size_t fused_ints() const { return elim6.size() + iso_loc.size() + fold.size(); }
  void build(int n, bool pin_left, bool pin_right, int max_levels = 0, int refine_sweeps = 0, bool fused = false);
};

// Device views of one chain.
struct BcrChain {
  int n_nodes;
  double* D;     // [n][80][80] working diagonal blocks (level 0 of an FTE chain: -> G = D^-1); never overwritten by a factor
  double* U;     // [n][80][80] U = L^-T of every eliminated node (own array: the T + 1 workgroups of a narrow-level
                 //             elimination all read D_i, and a late one must not find the factor there)
  double* Cpl;   // [n][80][80] coupling block (right neighbour rows, own cols)
  double* Wl;    // [n][80][80] W_l of eliminated nodes
  double* Wr;    // [n][80][80] W_r of eliminated nodes
  double* b;     // [n][80] rhs -> y -> solution
  const int* d_elim;
  const int* d_remain;
  const int* d_tail;         // device copy of BcrSchedule::tail (null: no fused tail)
  int* d_done;               // progress counter of the tail kernel
  const int* d_pairs = nullptr;   // dropped couplings of an incomplete reduction (2 ints each) ...
  int n_pairs = 0;
  double* trunc_eps2 = nullptr;   // ... and [n_pairs] squared Frobenius norms of L_b^-1 C L_a^-T, written every reduction
  double* refine_buf = nullptr;   // [3][n_isolated][80] x0 and the two iterates of the refinement sweeps
  const double* AL0 = nullptr;    // separator chain of the chunked solver: [n][80][80] left-run contributions (lower tiles; row 79:
                                  // the update of b), added to D / b by the LEVEL-0 kernels of the reduction when set
  int* st_flags = nullptr;        // k_sep_tail: [n_st_flags] flags (zeroed by the consumer of the solution), then the epoch of its
  int n_st_flags = 0;             //             hand-off tags (null: per-level kernels)
  unsigned long long* st_ll = nullptr;   // k_sep_tail: tagged word pairs, [2][n_isolated][80][2] iterates + [n][80][2] solutions
  // fused narrow levels (seplevel.hip; all null: the per-phase kernels of bcr.hip only).  With them Cpl holds 2 n blocks:
  // slot n + i = the coupling created by the elimination of node i.
  double* SL = nullptr;           // [n][80][80] running sum of the Schur contributions a node received from its LEFT side (lower
  double* SR = nullptr;           //             tiles; row 79: the update of b) / from its RIGHT side
  double* Y = nullptr;            // [n][80] y = U^T b of eliminated nodes (b itself stays intact until the back-substitution:
                                  //         the sibling workgroups of a node all read it)
  const int* d_elim6 = nullptr;
  const int* d_iso_loc = nullptr;
  const int* d_fold = nullptr;
  int implicit_couplings;    // 1: level-0 couplings are the analytic smoothness blocks (never stored)
  long long* dbg;            // optional [32] phase timestamps of workgroup 0 (gpu_stamps.py)
  // Fused system build (FTE chains only; all null for the separator chain): the level-0 kernels build
  // D = H_gn + lam*diag(H_gn) (+ 2^70 boost on bound-active variables) and b = -g themselves, so the damped
  // system never makes a round trip through HBM.
  const acino_fte_state* st;
  const double *x0, *x1, *g0, *g1, *H0, *H1;
  double* gn_part;           // [n] max |projected gradient| per node
};

int bcr_reduce(const BcrChain& ch, const BcrSchedule& sch, const FteConst* d_c, int* d_numeric_err,
               const int* d_status, hipStream_t s, Profiler* prof = nullptr);
int bcr_backsub(const BcrChain& ch, const BcrSchedule& sch, const FteConst* d_c, const int* d_status, hipStream_t s,
                Profiler* prof = nullptr, int* d_numeric_err = nullptr);
int bcr_set_func_attributes();
// the single-launch back-substitution of an incomplete reduction with refinement (k_sep_tail) needs every one of its
// workgroups resident: blocks it would launch (0 = not applicable) and the device's capacity for them
int bcr_sep_tail_fit(const BcrSchedule& sch, int* blocks, int* capacity);
// true when level 0 of the schedule runs the narrow-level kernels, which can add the chunk sweep's left-run contributions
// (BcrChain::AL0) themselves - no k_sep_combine launch
bool bcr_level0_adds_al(const BcrSchedule& sch);

}  // namespace acino
