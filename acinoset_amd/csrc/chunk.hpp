// Chunked substructuring of the block-tridiagonal Gauss-Newton chain (chunk.hip): host-side plan and entry points.
#pragma once
#include "bcr.hpp"

namespace acino {

// The chain of n_nodes super-blocks is cut into n_chunks runs of m consecutive nodes.  The LAST node of every run but
// the final one is a SEPARATOR; the others are interior nodes, eliminated in order by one workgroup per run with all
// operands resident in LDS.  The n_sep = n_chunks - 1 separators form a block-tridiagonal chain with dense couplings
// that the block cyclic reduction (bcr.hip) solves.
// Sharded ranks (pinned separators): chain node 0 is the LEFT pin when pin_left (the left neighbour's last three frames: not
// swept, it is the left separator of run 0 and receives its spike) and the chain's last node is the RIGHT pin when pin_right
// (the rank's own last three frames: built by the last run as its right separator, never eliminated).  The separator chain
// is then [left pin] + the runs' separators + [right pin].
struct ChunkPlan {
  int n_nodes = 0, m = 0, n_chunks = 0, n_sep = 0;
  int node0 = 0;        // first swept node (= pin_left)
  int pin_right = 0;
  bool active() const { return n_chunks > 0; }
  // chunk_nodes: nodes per run incl. its separator (>= 2); 0 = automatic; < 0 = no chunking (plain BCR)
  void build(int nodes, int chunk_nodes, bool pin_left = false, bool pin_right = false);
};

// Separator-side buffers written by the sweep (device views, [n_sep] each).
struct SepView {
  double* D;    // [80][80] built node - (right end of the run on its left)          -> + AL by k_sep_combine
  double* Cpl;  // [80][80] block(separator q + 1, separator q)
  double* AL;   // [80][80] -(sum over the run on its right of F^T G F), lower tiles; row 79: its update of b.  An array of its
                // own: level 0 of the separator reduction reads it while sibling workgroups already write W_l / W_r
  double* b;    // [80]
  int* flags = nullptr;   // the flags of k_sep_tail, zeroed again by k_chunk_backsub (n_flags ints, then the epoch counter it advances; may be null)
  int n_flags = 0;
};

// The trial iterate folded into the back-substitution (what k_trial does for the other solvers): the run that solves a node
// also forms x_t = clip(x + delta) of its three frames and the run's share of the predicted reduction / step length.
struct TrialOut {
  const double* hd0;   // diag of the Gauss-Newton blocks, [N][25], buffer 0 / 1 (as the iterate)
  const double* hd1;
  double* pred_part;   // [n_chunks] per-run partial sums -> k_totals
  double* step_part;
};

int chunk_set_func_attributes();
// forward: sweep of every run + separator assembly + reduction of the separator chain
int chunk_reduce(const BcrChain& ch, const ChunkPlan& pl, const SepView& sp, const BcrChain& sepch,
                 const BcrSchedule& sepsch, const FteConst* d_c, int* d_numeric_err, const int* d_status, hipStream_t s,
                 Profiler* prof);
// backward: separator chain back-substitution, then every run from its right end to its left end; x -> ch.b
int chunk_backsub(const BcrChain& ch, const ChunkPlan& pl, const SepView& sp, const BcrChain& sepch,
                  const BcrSchedule& sepsch, const FteConst* d_c, int* d_numeric_err, const int* d_status, hipStream_t s,
                  Profiler* prof, const TrialOut& trial);

}  // namespace acino
