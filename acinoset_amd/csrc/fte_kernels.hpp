// Internal declarations shared by the FTE translation units (not part of the C ABI).
#pragma once
#include "common.hpp"

namespace acino {

constexpr int NP = ACINO_N_ACTIVE;   // 25 active states per frame
constexpr int NL = ACINO_N_MARKERS;  // 20 markers
constexpr int BS = ACINO_BS;         // 80 = 3 frames x 25 states + 5 identity pad rows
constexpr int NGRP = 14;             // kinematic frames (rotation groups)
constexpr int FPB = 8;               // frames per workgroup in the assembly kernel
constexpr double FIX_SCALE = 1.1805916207174113e21;  // 2^70: diagonal boost pinning a bound-active variable

// Device-resident constant block of one FTE problem.
struct FteConst {
  int32_t n_frames, n_cams;
  int64_t n_global, n_offset;
  int32_t pin_left, pin_right;
  int32_t n_nodes;         // local chain length (super-blocks incl. a pinned left separator)
  int32_t pad;
  double dlc_thresh, inv_r;
  LossC loss;
  double q_w[NP], lo[NP], hi[NP];
  double ftol, xtol, gtol;
  Cam cams[ACINO_MAX_CAMS];
};

// (D3^T D3)[n, n+k] for global frame n, 0 <= k <= 3, sequence length ng (stencil -1, 3, -3, 1).
__host__ __device__ inline double band_coef(int64_t n, int k, int64_t ng) {
  if (n < 0 || n + k >= ng) return 0.0;
  const double c[4] = {-1.0, 3.0, -3.0, 1.0};
  int64_t jlo = n + k - 3 > 0 ? n + k - 3 : 0;
  int64_t jhi = n < ng - 4 ? n : ng - 4;
  double tot = 0.0;
  for (int64_t j = jlo; j <= jhi; ++j) tot += c[n - j] * c[n + k - j];
  return tot;
}

// x iterate buffers carry 3 halo frames on each side: row (i + 3) holds local frame i.
constexpr int HALO = 3;

int launch_assemble(const FteConst* d_c, const FteConst& h_c, const acino_fte_state* d_st, int which,
                    const double* d_det, double* const x[2], double* const H[2], double* const g[2],
                    double* d_cost_partials, int* d_nbehind, bool need_jac, bool respect_status, hipStream_t s);
int n_assemble_blocks(int n_frames);
int launch_fk(const double* d_q, int64_t n, double* d_pos, hipStream_t s);
int launch_fk_active(const double* d_xa_halo, int64_t n, double* d_pos, hipStream_t s);

}  // namespace acino
