// Cheetah kinematic chain shared by the FTE assembly, the plain FK kernels and the EKF (gfx950, fp64).
// Reference: src/all_optimizations.py:66-190 (rotation conventions :66-91, chain :101-128, marker offsets :138-165).
#pragma once
#include "fte_kernels.hpp"

namespace acino {

// ---- kinematic tree tables (active-state index: 0-2 xyz, 3-5 phi0,phi1,phi3, 6-19 theta0-13, 20-24 psi0,1,3,4,5)
static __device__ const int8_t c_grp_phi[NGRP] = {3, 4, -1, 5, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
static __device__ const int8_t c_grp_psi[NGRP] = {20, 21, -1, 22, 23, 24, -1, -1, -1, -1, -1, -1, -1, -1};
static __device__ const int8_t c_grp_pivot[NGRP] = {20, 20, 3, 4, 5, 6, 8, 9, 11, 12, 14, 15, 17, 18};  // pos index; 20 = head
static __device__ const int8_t c_state_grp[NP] = {-1, -1, -1, 0, 1, 3, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 0, 1, 3, 4, 5};
static __device__ const uint16_t c_ancmask[NGRP] = {
    0x0001, 0x0003, 0x0007, 0x000F, 0x001F, 0x003F, 0x0047, 0x00C7, 0x0107, 0x0307, 0x040F, 0x0C0F, 0x100F, 0x300F};
static __device__ const double c_off[NL][3] = {
    {0, 0.03, 0},          {0, -0.03, 0},         {0.055, 0, -0.055},  {-0.28, 0, 0},       {-0.37, 0, 0},
    {-0.37, 0, 0},         {-0.28, 0, 0},         {-0.36, 0, 0},       {-0.04, 0.08, -0.10}, {0, 0, -0.24},
    {0, 0, -0.28},         {-0.04, -0.08, -0.10}, {0, 0, -0.24},       {0, 0, -0.28},       {0.12, 0.08, -0.06},
    {0, 0, -0.32},         {0, 0, -0.25},         {0.12, -0.08, -0.06}, {0, 0, -0.32},      {0, 0, -0.25}};


// A "frame" type F provides sc[22][2] (sin, cos of the active angles, index a-3) and pos[21][3] (markers 0..19,
// head = 20); when F::kHasOm it also provides om[22][3], filled with the rotation axis of every active angle.
// Apply the group's elementary rotations (reference sign convention) to one column of the parent frame.
template <class FR>
__device__ __forceinline__ void chain_col(FR& F, int grp, const double pc[3], double out[3], int j) {
  const int at = 6 + grp;  // theta_k
  double s = F.sc[at - 3][0], c = F.sc[at - 3][1];
  if constexpr (FR::kHasOm) F.om[at - 3][j] = pc[1];                        // omega_theta = P^T e_y
  double y0 = c * pc[0] - s * pc[2], y1 = pc[1], y2 = s * pc[0] + c * pc[2];
  int ap = c_grp_phi[grp];
  if (ap >= 0) {
    if constexpr (FR::kHasOm) F.om[ap - 3][j] = y0;                         // omega_phi = (Ry P)^T e_x
    double sp = F.sc[ap - 3][0], cp = F.sc[ap - 3][1];
    double n1 = cp * y1 + sp * y2, n2 = -sp * y1 + cp * y2;
    y1 = n1;
    y2 = n2;
  }
  int az = c_grp_psi[grp];
  if (az >= 0) {
    if constexpr (FR::kHasOm) F.om[az - 3][j] = y2;                         // omega_psi = RI_k^T e_z
    double sz = F.sc[az - 3][0], cz = F.sc[az - 3][1];
    double n0 = cz * y0 + sz * y1, n1 = -sz * y0 + cz * y1;
    y0 = n0;
    y1 = n1;
  }
  out[0] = y0;
  out[1] = y1;
  out[2] = y2;
}

template <class FR>
__device__ __forceinline__ void place(FR& F, int m, int parent, const double col[3], int j) {
  F.pos[m][j] = F.pos[parent][j] + col[0] * c_off[m][0] + col[1] * c_off[m][1] + col[2] * c_off[m][2];
}

template <class FR>
__device__ __forceinline__ void fk_columns(FR& F, int j) {
  double e[3] = {j == 0 ? 1.0 : 0.0, j == 1 ? 1.0 : 0.0, j == 2 ? 1.0 : 0.0};
  double c0[3], c1[3], c2[3], c3[3], t[3], t2[3];
  chain_col(F, 0, e, c0, j);
  place(F, 0, 20, c0, j);
  place(F, 1, 20, c0, j);
  place(F, 2, 20, c0, j);
  chain_col(F, 1, c0, c1, j);
  place(F, 3, 20, c1, j);
  chain_col(F, 2, c1, c2, j);
  place(F, 4, 3, c2, j);
  place(F, 8, 3, c2, j);
  place(F, 11, 3, c2, j);
  chain_col(F, 3, c2, c3, j);
  place(F, 5, 4, c3, j);
  place(F, 14, 5, c3, j);
  place(F, 17, 5, c3, j);
  chain_col(F, 4, c3, t, j);
  place(F, 6, 5, t, j);
  chain_col(F, 5, t, t2, j);
  place(F, 7, 6, t2, j);
  chain_col(F, 6, c2, t, j);
  place(F, 9, 8, t, j);
  chain_col(F, 7, t, t2, j);
  place(F, 10, 9, t2, j);
  chain_col(F, 8, c2, t, j);
  place(F, 12, 11, t, j);
  chain_col(F, 9, t, t2, j);
  place(F, 13, 12, t2, j);
  chain_col(F, 10, c3, t, j);
  place(F, 15, 14, t, j);
  chain_col(F, 11, t, t2, j);
  place(F, 16, 15, t2, j);
  chain_col(F, 12, c3, t, j);
  place(F, 18, 17, t, j);
  chain_col(F, 13, t, t2, j);
  place(F, 19, 18, t2, j);
}


}  // namespace acino
