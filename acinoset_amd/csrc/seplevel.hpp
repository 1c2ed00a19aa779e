// Fused narrow levels of the block cyclic reduction (seplevel.hip): host entry points and the kernel's argument block.
#pragma once
#include "bcr.hpp"

namespace acino {

constexpr int SLV_MAXT = 16;     // workgroups per eliminated node, at most
constexpr int SLV_STRIDE = 64;   // ints per workgroup of a split table: strip mask, store mask, tile count, <= 55 tile codes

struct SepLevelArgs {
  const int* ent;       // [n][6] node, left, right, flags, location of block(node, left), location of block(right, node)
  int T;                // workgroups per node
  int nx, per, total;   // XCD mapping (as the narrow-level kernels of bcr.hip)
};

void slv_plan_build(int T, int* out);                 // the split of a node's 55 output tiles over T workgroups (host; tests)
int slv_set_func_attributes();                        // LDS attribute + the split tables on the current device
int slv_workgroups_per_node(int n_elim);
int slv_launch_level(const BcrChain& ch, const BcrLevel& lv, int* d_numeric_err, const int* d_status, hipStream_t s);
int slv_launch_fold(const BcrChain& ch, const int* d_entries, int n, const int* d_status, hipStream_t s);

}  // namespace acino
