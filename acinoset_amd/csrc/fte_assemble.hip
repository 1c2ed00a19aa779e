// FTE residual / Jacobian / normal-equation assembly (gfx950, fp64).
//
// Reference: cheetah FK src/all_optimizations.py:66-190, projection pt3d_to_2d :193-209, weights
// :243-252,302-315, objective :486-500, loss src/build.py:382-395.
//
// One workgroup handles FPB frames in five LDS-staged phases:
//   A  sin/cos of the 22 active angles                       (frame, angle)
//   B  kinematic chain, column-parallel: RI_k[:,j], marker coordinate j, axis component j
//                                                            (frame, column j)
//   C  fisheye projection + analytic 2x3 Jacobian + robust weights for all cameras; per marker the
//      3x3 M_l = sum_c J^T W J, v_l = sum_c J^T w rho' and their 6x6 "spatial" form
//      Lambda_l = S_l^T M_l S_l, f_l = S_l^T v_l with S_l = [I, -[p_l]x]          (frame, marker)
//   D  subtree sums of Lambda / f over the kinematic tree     (frame, component)
//   E  H[a][b] = xi_a . (Lambda_sub(b) xi_b), g[b] = xi_b . f_sub(b) + smoothness      (frame, state)
// where xi = (pivot x omega, omega) is the joint twist: dp_l/dq_a = omega_a x (p_l - pivot_a).
// J (240x25 per frame) is never materialised; only H_n (25x25) and g_n leave the workgroup.
#include "cheetah_fk.hpp"
#include "dense80.hpp"

namespace acino {

// 6 408 B per frame: 8 frames = 51 KB per workgroup, so THREE workgroups share a CU (12 waves: the projection phase is a
// chain of fp64 transcendentals per thread and lives on latency hiding).  Two arrays share storage with one that is dead
// by the time they are written: the twists (phase C) take the place of sin / cos (phases A-B), the subtree sums (phase D)
// are written over the first NGRP rows of the per-marker blocks, column by column by the thread that has just read it.
struct FrameLds {
  static constexpr bool kHasOm = true;
  union {
    double sc[22][2];     // phases A-B: sin, cos of active angle (index a-3)
    double xi[22][6];     // phase C on: twist (pivot x omega, omega)
  };
  double pos[21][3];      // markers 0..19, head = 20
  double om[22][3];       // rotation axis of active angle in the inertial frame
  double lam[NL][27];     // per-marker 6x6 symmetric (21) + wrench (6); after phase D rows 0..NGRP-1 = subtree sums
};
static_assert(NGRP <= NL, "subtree sums are stored over the per-marker blocks");

// index of (i,j), i<=j, in the packed upper triangle of a 6x6
__device__ __forceinline__ int tri6(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }

// float twin of redescending<> (common.hpp) for the mixed-precision rows of ACINO_PREC_BF16_ROWS
struct LossF {
  float a, b, c, ea, eb, ec, d0, t4, icb;
};
template <bool DERIV>
__device__ __forceinline__ void redescending_f(const LossF& L, float err, float& rho, float& drho, float& h) {
  const float e = fabsf(err);
  const float u = __expf(-e);
  const float da = 1.0f + L.ea * u, db = 1.0f + L.eb * u, dc = 1.0f + L.ec * u;
  const float inv = __frcp_rn(da * db * dc);
  const float sa = inv * (db * dc), sb = inv * (da * dc), sc = inv * (da * db);
  const float cb = L.c - L.b;
  const float t2 = L.a * e - L.a * L.a * 0.5f;
  const float ce = (L.c - e) * L.icb;
  const float t3 = L.a * L.b - L.a * L.a * 0.5f + (L.a * cb * 0.5f) * (1.0f - ce * ce);
  rho = (1.0f - sa) * 0.5f * e * e + (sa - sb) * t2 + (sb - sc) * t3 + sc * L.t4;
  if (DERIV) {
    const float dsa = sa * (1.0f - sa), dsb = sb * (1.0f - sb), dsc = sc * (1.0f - sc);
    drho = -dsa * 0.5f * e * e + (1.0f - sa) * e + (dsa - dsb) * t2 + (sa - sb) * L.a + (dsb - dsc) * t3 +
           (sb - sc) * (L.a * ce) + dsc * L.t4;
    const float hh = e > 1e-6f ? (drho - L.d0) * __frcp_rn(e) : 1.0f;
    h = fminf(fmaxf(hh, 0.0f), 1.0f);
  }
}

// PREC = ACINO_PREC_F64: everything fp64.  PREC = ACINO_PREC_BF16_ROWS (BASELINE config 5): camera-frame coordinates
// in fp64, projection / Jacobian / robust weights in fp32, the scaled residuals and the 2x3 Jacobian ROWS rounded to
// bf16, M_l and v_l accumulated in fp32; from Lambda_l on (phases C-tail, D, E) fp64 as before.  The cost is summed in
// fp64 from the UNROUNDED fp32 residuals (accept / reject decisions need more than 8 bits).
// SPLIT = 2 (short chains: fewer workgroups than the chip holds at once, so only the latency of ONE workgroup counts) deals the
// cameras of a (frame, marker) to two lanes next to each other - contiguous halves of the camera list, summed left + right:
// 8 frames x 20 markers x 2 = 320 threads, three cameras per lane instead of six.  (For long chains it loses: the kernel keeps
// ~165 registers, five-wave workgroups fit twice on a CU where four-wave ones fit three times - NOTES_perf.md round 6.)
template <bool JAC, int PREC, int SPLIT = 1>
__global__ void __launch_bounds__(SPLIT == 2 ? 2 * FPB * NL : 256)
k_fte_assemble(const FteConst* __restrict__ cst, const acino_fte_state* __restrict__ st, int which,
               const double* __restrict__ det, const double* __restrict__ x0, const double* __restrict__ x1,
               double* __restrict__ H0, double* __restrict__ H1, double* __restrict__ g0, double* __restrict__ g1,
               double* __restrict__ hd0, double* __restrict__ hd1, double* __restrict__ cost_partials,
               int* __restrict__ nbehind, int respect_status) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  FrameLds* F = reinterpret_cast<FrameLds*>(smem_raw);
  double* red = reinterpret_cast<double*>(smem_raw + sizeof(FrameLds) * FPB);
  const int tid = threadIdx.x;
  if (respect_status && st->status != 0) return;  // LM already converged: no-op
  // which = 0: the current iterate, 1: the trial iterate (selected on the device, no host sync)
  const int buf = st->cur ^ which;
  const double* __restrict__ xh = buf ? x1 : x0;
  double* __restrict__ D0 = buf ? H1 : H0;
  double* __restrict__ gout = buf ? g1 : g0;
  double* __restrict__ hdout = buf ? hd1 : hd0;      // diag(H), contiguous, for the trial kernel's predicted reduction
  const FteConst& K = *cst;
  const int N = K.n_frames;
  // frames dealt to the XCDs in contiguous ranges, walked backwards: the level-0 elimination (same ranges, forwards)
  // starts with the H blocks written last
  const int blk = xcd_contiguous_rev((int)blockIdx.x, (int)gridDim.x);
  const int f0 = blk * FPB;
  const int nf = min(FPB, N - f0);
  double my_cost = 0.0;

  // ---- A: sincos, head position
  for (int task = tid; task < nf * 25; task += blockDim.x) {
    int f = task / 25, a = task - f * 25;
    double xv = xh[(int64_t)(f0 + f + HALO) * NP + a];
    if (a < 3) {
      F[f].pos[20][a] = xv;
    } else {
      double s, c;
      sincos(xv, &s, &c);
      F[f].sc[a - 3][0] = s;
      F[f].sc[a - 3][1] = c;
    }
  }
  __syncthreads();
  // ---- B: chain, one thread per (frame, column)
  for (int task = tid; task < nf * 3; task += blockDim.x) {
    int f = task / 3, j = task - f * 3;
    fk_columns(F[f], j);
  }
  __syncthreads();
  // ---- C: projection (+ twists: on the otherwise idle threads; SPLIT = 2 has none - all threads, first; the twists go
  //      over sin / cos, dead since the barrier)
  const int nproj = SPLIT * nf * NL;
  if (JAC) {
    const int t0 = SPLIT == 2 ? tid : tid - nproj, dt = SPLIT == 2 ? (int)blockDim.x : ((int)blockDim.x - nproj > 0 ? (int)blockDim.x - nproj : 1);
    for (int task = t0; task < nf * 22; task += dt) {
      if (task < 0) break;
      int f = task / 22, a = task - f * 22;
      int g = c_state_grp[a + 3];
      const double* c = F[f].pos[c_grp_pivot[g]];
      const double* w = F[f].om[a];
      F[f].xi[a][0] = c[1] * w[2] - c[2] * w[1];
      F[f].xi[a][1] = c[2] * w[0] - c[0] * w[2];
      F[f].xi[a][2] = c[0] * w[1] - c[1] * w[0];
      F[f].xi[a][3] = w[0];
      F[f].xi[a][4] = w[1];
      F[f].xi[a][5] = w[2];
    }
  }
  if (tid < nproj) {
    const int half = SPLIT == 2 ? (tid & 1) : 0, fl = SPLIT == 2 ? (tid >> 1) : tid;
    const int f = fl / NL, l = fl - f * NL;
    const int n = f0 + f;
    const bool owned = n >= K.own_lo && n < K.own_hi;      // window sharding: only owned frames enter the cost
    double cost_c = 0.0;
    const double px = F[f].pos[l][0], py = F[f].pos[l][1], pz = F[f].pos[l][2];
    double M[6] = {0, 0, 0, 0, 0, 0}, v[3] = {0, 0, 0};
    float Mf[6] = {0, 0, 0, 0, 0, 0}, vf[3] = {0, 0, 0};
    LossF lossf;
    if (PREC != ACINO_PREC_F64) {
      lossf.a = (float)K.loss.a; lossf.b = (float)K.loss.b; lossf.c = (float)K.loss.c;
      lossf.ea = (float)K.loss.ea; lossf.eb = (float)K.loss.eb; lossf.ec = (float)K.loss.ec;
      lossf.d0 = (float)K.loss.d0; lossf.t4 = (float)K.loss.t4; lossf.icb = (float)K.loss.icb;
    }
    double rho0, dd, hh;
    redescending<false>(K.loss, 0.0, rho0, dd, hh);
    int behind = 0;
    const int C = K.n_cams;
    const int c_lo = (SPLIT == 2 && half) ? (C + 1) / 2 : 0, c_hi = (SPLIT == 2 && !half) ? (C + 1) / 2 : C;   // this lane's cameras
    // software-pipelined detection reads: camera ci+1's (x, y, likelihood) is requested before camera ci is
    // processed - the loop body branches (zero weight, behind camera), which would otherwise expose one HBM
    // latency per camera
    const double* dbase = det + ((int64_t)n * C * NL + l) * 3;
    double nx = 0.0, ny = 0.0, nlik = 0.0;
    if (c_lo < c_hi) {
      const double* d = dbase + (int64_t)c_lo * NL * 3;
      nx = d[0];
      ny = d[1];
      nlik = d[2];
    }
    for (int ci = c_lo; ci < c_hi; ++ci) {
      const Cam& cam = K.cams[ci];
      const double um = nx, vm = ny, lik = nlik;
      if (ci + 1 < c_hi) {
        const double* d = dbase + (int64_t)(ci + 1) * NL * 3;
        nx = d[0];
        ny = d[1];
        nlik = d[2];
      }
      double w = (lik > K.dlc_thresh && isfinite(um) && isfinite(vm)) ? K.inv_r : 0.0;
      double xc = cam.R[0] * px + cam.R[1] * py + cam.R[2] * pz + cam.t[0];
      double yc = cam.R[3] * px + cam.R[4] * py + cam.R[5] * pz + cam.t[1];
      double zc = cam.R[6] * px + cam.R[7] * py + cam.R[8] * pz + cam.t[2];
      // the reference's pt3d_to_2d (all_optimizations.py:193-209) has no cut at z_cam <= 0: a marker behind a camera
      // keeps its mirrored projection and pays (typically the saturated) loss.  Only the singular plane itself is
      // dropped; n_behind counts weighted detections with z_cam < 1e-6 (diagnostic).
      if (zc < 1e-6 && w > 0) ++behind;
      if (fabs(zc) < 1e-9) w = 0.0;
      if (w == 0.0) {
        cost_c += 2.0 * rho0;
        continue;
      }
      if (PREC != ACINO_PREC_F64) {
        // ---- fp32 projection from the fp64 camera-frame point; the pixel offset (c - z) is formed in fp64 first
        const float xf = (float)xc, yf = (float)yc, zf = (float)zc;
        const float izf = __frcp_rn(zf);
        const float a = xf * izf, b = yf * izf;
        const float r2 = a * a + b * b + 1e-12f;
        const float ir = __frsqrt_rn(r2);
        const float r = r2 * ir;
        const float th = atanf(r);
        const float th2 = th * th;
        const float k1 = (float)cam.k1, k2 = (float)cam.k2, k3 = (float)cam.k3, k4 = (float)cam.k4;
        const float fxf = (float)cam.fx, fyf = (float)cam.fy, wf = (float)w;
        const float poly = 1.0f + th2 * (k1 + th2 * (k2 + th2 * (k3 + th2 * k4)));
        const float thD = th * poly;
        const float m = thD * ir;
        const float su_f = wf * (fxf * a * m + (float)(cam.cx - um));
        const float sv_f = wf * (fyf * b * m + (float)(cam.cy - vm));
        // the residual ROW as stored: bf16
        const float su = bf16_round(su_f), sv = bf16_round(sv_f);
        float rho_u, drho_u = 0, h_u = 0, rho_v, drho_v = 0, h_v = 0, dmy0 = 0, dmy1 = 0;
        redescending_f<false>(lossf, su_f, rho_u, dmy0, dmy1);          // cost: unrounded residual, summed in fp64
        redescending_f<false>(lossf, sv_f, rho_v, dmy0, dmy1);
        cost_c += (double)rho_u + (double)rho_v;
        if (JAC) {
          float r0, r1;
          redescending_f<true>(lossf, su, r0, drho_u, h_u);             // weights: from the stored (bf16) residual
          redescending_f<true>(lossf, sv, r1, drho_v, h_v);
          const float dthD = 1.0f + th2 * (3.0f * k1 + th2 * (5.0f * k2 + th2 * (7.0f * k3 + th2 * 9.0f * k4)));
          const float dm_dr = (dthD * __frcp_rn(1.0f + r2) * r - thD) * (ir * ir);
          const float dm_da = dm_dr * a * ir, dm_db = dm_dr * b * ir;
          const float du_da = fxf * (m + a * dm_da), du_db = fxf * a * dm_db;
          const float dv_da = fyf * b * dm_da, dv_db = fyf * (m + b * dm_db);
          const float uc0 = du_da * izf, uc1 = du_db * izf, uc2 = -(du_da * a + du_db * b) * izf;
          const float vc0 = dv_da * izf, vc1 = dv_db * izf, vc2 = -(dv_da * a + dv_db * b) * izf;
          float ju[3], jv[3];
#pragma unroll
          for (int j = 0; j < 3; ++j) {                                   // the Jacobian ROWS as stored: bf16
            ju[j] = uc0 * (float)cam.R[j] + uc1 * (float)cam.R[3 + j] + uc2 * (float)cam.R[6 + j];
            jv[j] = vc0 * (float)cam.R[j] + vc1 * (float)cam.R[3 + j] + vc2 * (float)cam.R[6 + j];
            if (PREC == ACINO_PREC_BF16_ROWS) {
              ju[j] = bf16_round(ju[j]);
              jv[j] = bf16_round(jv[j]);
            }
          }
          const float gu = wf * drho_u * (su > 0 ? 1.0f : (su < 0 ? -1.0f : 0.0f));
          const float gv = wf * drho_v * (sv > 0 ? 1.0f : (sv < 0 ? -1.0f : 0.0f));
          const float hu = wf * wf * h_u, hv = wf * wf * h_v;
          Mf[0] += hu * ju[0] * ju[0] + hv * jv[0] * jv[0];              // fp32 accumulation
          Mf[1] += hu * ju[0] * ju[1] + hv * jv[0] * jv[1];
          Mf[2] += hu * ju[0] * ju[2] + hv * jv[0] * jv[2];
          Mf[3] += hu * ju[1] * ju[1] + hv * jv[1] * jv[1];
          Mf[4] += hu * ju[1] * ju[2] + hv * jv[1] * jv[2];
          Mf[5] += hu * ju[2] * ju[2] + hv * jv[2] * jv[2];
          vf[0] += gu * ju[0] + gv * jv[0];
          vf[1] += gu * ju[1] + gv * jv[1];
          vf[2] += gu * ju[2] + gv * jv[2];
        }
      } else {
        double iz = rcp64(zc);
        double a = xc * iz, b = yc * iz;
        const double r2 = a * a + b * b + 1e-12;
        const double ir = rsqrt(r2);                 // every later "/ r" is a multiplication
        double r = r2 * ir;
        double th = atan(r);
        double th2 = th * th;
        double poly = 1 + th2 * (cam.k1 + th2 * (cam.k2 + th2 * (cam.k3 + th2 * cam.k4)));
        double thD = th * poly;
        double m = thD * ir;
        double su = w * (cam.fx * a * m + cam.cx - um);
        double sv = w * (cam.fy * b * m + cam.cy - vm);
        double rho_u, drho_u = 0, h_u = 0, rho_v, drho_v = 0, h_v = 0;
        redescending<JAC>(K.loss, su, rho_u, drho_u, h_u);
        redescending<JAC>(K.loss, sv, rho_v, drho_v, h_v);
        cost_c += rho_u + rho_v;
        if (JAC) {
          double dthD = 1 + th2 * (3 * cam.k1 + th2 * (5 * cam.k2 + th2 * (7 * cam.k3 + th2 * 9 * cam.k4)));
          double dm_dr = (dthD * rcp64(1 + r2) * r - thD) * (ir * ir);
          double dm_da = dm_dr * a * ir, dm_db = dm_dr * b * ir;
          double du_da = cam.fx * (m + a * dm_da), du_db = cam.fx * a * dm_db;
          double dv_da = cam.fy * b * dm_da, dv_db = cam.fy * (m + b * dm_db);
          double uc0 = du_da * iz, uc1 = du_db * iz, uc2 = -(du_da * a + du_db * b) * iz;
          double vc0 = dv_da * iz, vc1 = dv_db * iz, vc2 = -(dv_da * a + dv_db * b) * iz;
          double ju[3], jv[3];
  #pragma unroll
          for (int j = 0; j < 3; ++j) {
            ju[j] = uc0 * cam.R[j] + uc1 * cam.R[3 + j] + uc2 * cam.R[6 + j];
            jv[j] = vc0 * cam.R[j] + vc1 * cam.R[3 + j] + vc2 * cam.R[6 + j];
          }
          double gu = w * drho_u * (su > 0 ? 1.0 : (su < 0 ? -1.0 : 0.0));
          double gv = w * drho_v * (sv > 0 ? 1.0 : (sv < 0 ? -1.0 : 0.0));
          double hu = w * w * h_u, hv = w * w * h_v;
          M[0] += hu * ju[0] * ju[0] + hv * jv[0] * jv[0];
          M[1] += hu * ju[0] * ju[1] + hv * jv[0] * jv[1];
          M[2] += hu * ju[0] * ju[2] + hv * jv[0] * jv[2];
          M[3] += hu * ju[1] * ju[1] + hv * jv[1] * jv[1];
          M[4] += hu * ju[1] * ju[2] + hv * jv[1] * jv[2];
          M[5] += hu * ju[2] * ju[2] + hv * jv[2] * jv[2];
          v[0] += gu * ju[0] + gv * jv[0];
          v[1] += gu * ju[1] + gv * jv[1];
          v[2] += gu * ju[2] + gv * jv[2];
        }
    
      }
    }
    if (PREC != ACINO_PREC_F64) {
      if (JAC && SPLIT == 2) {       // (the mixed-precision rows accumulate in fp32: so does the sum of the two halves)
#pragma unroll
        for (int k = 0; k < 6; ++k) Mf[k] += __shfl_xor(Mf[k], 1, 64);
#pragma unroll
        for (int k = 0; k < 3; ++k) vf[k] += __shfl_xor(vf[k], 1, 64);
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) M[k] = (double)Mf[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) v[k] = (double)vf[k];
    } else if (JAC && SPLIT == 2) {
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const double o = __shfl_xor(M[k], 1, 64);
        M[k] = half ? o + M[k] : M[k] + o;             // left half + right half, on both lanes
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double o = __shfl_xor(v[k], 1, 64);
        v[k] = half ? o + v[k] : v[k] + o;
      }
    }
    if (owned) my_cost += cost_c;
    if (behind && owned) atomicAdd(nbehind, behind);
    if (JAC && half == 0) {
      // Lambda = [[M, -B], [-B^T, -P B]],  B = M P,  P = [p]x ;  f = [v, p x v]
      const double Mm[3][3] = {{M[0], M[1], M[2]}, {M[1], M[3], M[4]}, {M[2], M[4], M[5]}};
      const double P[3][3] = {{0, -pz, py}, {pz, 0, -px}, {-py, px, 0}};
      double B[3][3], PB[3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) B[i][j] = Mm[i][0] * P[0][j] + Mm[i][1] * P[1][j] + Mm[i][2] * P[2][j];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) PB[i][j] = P[i][0] * B[0][j] + P[i][1] * B[1][j] + P[i][2] * B[2][j];
      double* L = F[f].lam[l];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) L[tri6(i, j)] = Mm[i][j];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) L[tri6(i, 3 + j)] = -B[i][j];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) L[tri6(3 + i, 3 + j)] = -PB[i][j];
      L[21] = v[0];
      L[22] = v[1];
      L[23] = v[2];
      L[24] = py * v[2] - pz * v[1];
      L[25] = pz * v[0] - px * v[2];
      L[26] = px * v[1] - py * v[0];
    }
  }
  __syncthreads();
  if (JAC) {
    // ---- D: subtree sums (children have larger group index than parents)
    for (int task = tid; task < nf * 27; task += blockDim.x) {
      int f = task / 27, q = task - f * 27;
      double(*Lm)[27] = F[f].lam;
      double s13 = Lm[19][q], s12 = Lm[18][q] + s13;
      double s11 = Lm[16][q], s10 = Lm[15][q] + s11;
      double s9 = Lm[13][q], s8 = Lm[12][q] + s9;
      double s7 = Lm[10][q], s6 = Lm[9][q] + s7;
      double s5 = Lm[7][q], s4 = Lm[6][q] + s5;
      double s3 = Lm[5][q] + Lm[14][q] + Lm[17][q] + s4 + s10 + s12;
      double s2 = Lm[4][q] + Lm[8][q] + Lm[11][q] + s3 + s6 + s8;
      double s1 = Lm[3][q] + s2;
      double s0 = Lm[0][q] + Lm[1][q] + Lm[2][q] + s1;
      double(*S)[27] = F[f].lam;      // in place: column q of this frame is this thread's alone, all 20 reads are done
      S[0][q] = s0; S[1][q] = s1; S[2][q] = s2; S[3][q] = s3; S[4][q] = s4; S[5][q] = s5; S[6][q] = s6;
      S[7][q] = s7; S[8][q] = s8; S[9][q] = s9; S[10][q] = s10; S[11][q] = s11; S[12][q] = s12; S[13][q] = s13;
    }
    __syncthreads();
  }
  // ---- E: per (frame, state): smoothness, gradient, Hessian column
  for (int task = tid; task < nf * NP; task += blockDim.x) {
    const int f = task / NP, bq = task - f * NP;
    const int n = f0 + f;
    const int64_t ng = K.n_offset + n;  // global frame index
    const double q = K.q_w[bq];
    const double* xc = xh + (int64_t)(n + HALO) * NP + bq;   // x[n][bq]; neighbours at +-k*NP
    // smoothness cost: rows whose last frame is this one
    if ((K.clip_len > 0 ? ng % K.clip_len : ng) >= 3 && n >= K.own_lo && n < K.own_hi) {
      double d3 = xc[0] - 3.0 * xc[-NP] + 3.0 * xc[-2 * NP] - xc[-3 * NP];
      my_cost += q * d3 * d3;
    }
    if (JAC) {
      double gs = 0.0;
#pragma unroll
      for (int k = -3; k <= 3; ++k) {
        double bc = k >= 0 ? band_coef_clip(ng, k, K.n_global, K.clip_len) : band_coef_clip(ng + k, -k, K.n_global, K.clip_len);
        if (bc != 0.0) gs += bc * xc[k * NP];
      }
      const double b0 = band_coef_clip(ng, 0, K.n_global, K.clip_len);
      const int g = c_state_grp[bq];
      const double* S = F[f].lam[g < 0 ? 0 : g];   // (subtree sums, written over the per-marker blocks by phase D)
      double xb[6];
      if (bq < 3) {
#pragma unroll
        for (int i = 0; i < 6; ++i) xb[i] = (i == bq) ? 1.0 : 0.0;
      } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) xb[i] = F[f].xi[bq - 3][i];
      }
      double Y[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        double acc = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) acc += S[i <= j ? tri6(i, j) : tri6(j, i)] * xb[j];
        Y[i] = acc;
      }
      double gm = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i) gm += xb[i] * S[21 + i];
      gout[(int64_t)n * NP + bq] = gm + 2.0 * q * gs;
      // Hessian column bq of this frame's 25x25 block
      double* Dn = D0 + (int64_t)n * HPAIRS;          // the frame's 325 unordered state pairs (fte_kernels.hpp: hpair)
      const unsigned ancb = g < 0 ? 0u : c_ancmask[g];
#pragma unroll
      for (int a = 0; a < NP; ++a) {
        const int ga = c_state_grp[a];
        // a is ancestor-or-same of bq ?   (root is an ancestor of everything; root-root only with itself)
        bool a_anc_b = ga < 0 ? true : (g >= 0 && ((ancb >> ga) & 1u));
        bool b_anc_a = g < 0 ? true : (ga >= 0 && ((c_ancmask[ga] >> g) & 1u));
        // (two ROOT states are each other's "ancestor": the pair has two candidate writers whose values agree only to rounding -
        //  the thread of the larger state writes, so the stored value does not depend on which store lands last)
        if (a_anc_b && !(b_anc_a && a > bq)) {
          double val;
          if (a < 3) {
            val = Y[a];
          } else {
            const double* xa = F[f].xi[a - 3];
            val = xa[0] * Y[0] + xa[1] * Y[1] + xa[2] * Y[2] + xa[3] * Y[3] + xa[4] * Y[4] + xa[5] * Y[5];
          }
          if (a == bq) {
            val += 2.0 * q * b0;
            hdout[(int64_t)n * NP + bq] = val;
          }
          Dn[hpair(a, bq)] = val;                 // (the pair of a state with an ancestor-or-self: written by the descendant's thread)
        } else if (!b_anc_a && a < bq) {
          Dn[hpair(a, bq)] = 0.0;                 // unrelated branches (one writer)
        }
      }
    }
  }
  // ---- block cost reduction (fixed order -> deterministic)
  for (int off = 32; off > 0; off >>= 1) my_cost += __shfl_down(my_cost, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = my_cost;
  __syncthreads();
  if (tid == 0) {
    double t = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    cost_partials[blk] = t;
  }
}

// ---- plain FK kernels (positions only) -----------------------------------------------------
__global__ void __launch_bounds__(256)
k_fk(const double* __restrict__ q, int64_t n_frames, int stride, int halo, int active_only, double* __restrict__ pos) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  FrameLds* F = reinterpret_cast<FrameLds*>(smem_raw);
  const int tid = threadIdx.x;
  const int64_t f0 = (int64_t)blockIdx.x * FPB;
  const int nf = (int)min((int64_t)FPB, n_frames - f0);
  const int8_t act[NP] = {0, 1, 2, 3, 4, 6, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 34, 35, 36};
  for (int task = tid; task < nf * NP; task += blockDim.x) {
    int f = task / NP, a = task - f * NP;
    double xv = q[(f0 + f + halo) * stride + (active_only ? a : act[a])];
    if (a < 3) {
      F[f].pos[20][a] = xv;
    } else {
      double s, c;
      sincos(xv, &s, &c);
      F[f].sc[a - 3][0] = s;
      F[f].sc[a - 3][1] = c;
    }
  }
  __syncthreads();
  for (int task = tid; task < nf * 3; task += blockDim.x) fk_columns(F[task / 3], task % 3);
  __syncthreads();
  for (int task = tid; task < nf * NL * 3; task += blockDim.x) {
    int f = task / (NL * 3), r = task - f * NL * 3;
    pos[(f0 + f) * NL * 3 + r] = F[f].pos[r / 3][r % 3];
  }
}

int n_assemble_blocks(int n_frames) { return (n_frames + FPB - 1) / FPB; }

int launch_assemble(const FteConst* d_c, const FteConst& h_c, const acino_fte_state* d_st, int which,
                    const double* d_det, double* const x[2], double* const H[2], double* const g[2],
                    double* const hd[2], double* d_cost_partials, int* d_nbehind, bool need_jac, bool respect_status, hipStream_t s) {
  const int nb = n_assemble_blocks(h_c.n_frames);
  if (nb == 0) return ACINO_OK;
  const size_t lds = sizeof(FrameLds) * FPB + 64;
  static PerDeviceOnce attr;
  static int cus_of[64] = {};
  int dev = 0;
  ACINO_HIP_CHECK(hipGetDevice(&dev));
  if (attr.first()) {
#define ACINO_ASM_ATTR(J, P, S)                                                                                       \
  ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fte_assemble<J, P, S>),                           \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
    ACINO_ASM_ATTR(true, ACINO_PREC_F64, 1);        ACINO_ASM_ATTR(false, ACINO_PREC_F64, 1);
    ACINO_ASM_ATTR(true, ACINO_PREC_BF16_ROWS, 1);  ACINO_ASM_ATTR(false, ACINO_PREC_BF16_ROWS, 1);
    ACINO_ASM_ATTR(true, ACINO_PREC_BF16_RES, 1);   ACINO_ASM_ATTR(false, ACINO_PREC_BF16_RES, 1);
    ACINO_ASM_ATTR(true, ACINO_PREC_F64, 2);        ACINO_ASM_ATTR(false, ACINO_PREC_F64, 2);
    ACINO_ASM_ATTR(true, ACINO_PREC_BF16_ROWS, 2);  ACINO_ASM_ATTR(false, ACINO_PREC_BF16_ROWS, 2);
    ACINO_ASM_ATTR(true, ACINO_PREC_BF16_RES, 2);   ACINO_ASM_ATTR(false, ACINO_PREC_BF16_RES, 2);
#undef ACINO_ASM_ATTR
    ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fk),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(FrameLds) * FPB)));
    int cus = 0;
    ACINO_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (dev >= 0 && dev < 64) cus_of[dev] = cus;
  }
  // short chains - a workgroup per CU at most - take the camera-split variant: what counts there is the latency of one
  // workgroup, and its longest phase is the serial loop over the cameras (999 frames: 27.6 -> 23.9 us; with two workgroups on a
  // CU it loses already - 3 331 frames: 30 -> 43 us)
  int split = (dev >= 0 && dev < 64 && nb <= cus_of[dev]) ? 2 : 1;
  if (const char* e = getenv("ACINO_ASM_SPLIT")) split = atoi(e) == 2 ? 2 : 1;
#define ACINO_LAUNCH_ASSEMBLE(J, P)                                                                                   \
  do {                                                                                                                \
    if (split == 2)                                                                                                   \
      hipLaunchKernelGGL((k_fte_assemble<J, P, 2>), dim3(nb), dim3(2 * FPB * NL), lds, s, d_c, d_st, which, d_det, x[0], x[1], \
                         H[0], H[1], g[0], g[1], hd[0], hd[1], d_cost_partials, d_nbehind, respect_status ? 1 : 0);     \
    else                                                                                                              \
      hipLaunchKernelGGL((k_fte_assemble<J, P, 1>), dim3(nb), dim3(256), lds, s, d_c, d_st, which, d_det, x[0], x[1],  \
                         H[0], H[1], g[0], g[1], hd[0], hd[1], d_cost_partials, d_nbehind, respect_status ? 1 : 0);     \
  } while (0)
  if (h_c.precision == ACINO_PREC_BF16_ROWS) {
    if (need_jac) ACINO_LAUNCH_ASSEMBLE(true, ACINO_PREC_BF16_ROWS);
    else ACINO_LAUNCH_ASSEMBLE(false, ACINO_PREC_BF16_ROWS);
  } else if (h_c.precision == ACINO_PREC_BF16_RES) {
    if (need_jac) ACINO_LAUNCH_ASSEMBLE(true, ACINO_PREC_BF16_RES);
    else ACINO_LAUNCH_ASSEMBLE(false, ACINO_PREC_BF16_RES);
  } else {
    if (need_jac) ACINO_LAUNCH_ASSEMBLE(true, ACINO_PREC_F64);
    else ACINO_LAUNCH_ASSEMBLE(false, ACINO_PREC_F64);
  }
#undef ACINO_LAUNCH_ASSEMBLE
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

static int fk_attr() {
  static PerDeviceOnce attr;
  if (attr.first())
    ACINO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fk),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(FrameLds) * FPB)));
  return ACINO_OK;
}

int launch_fk(const double* d_q, int64_t n, double* d_pos, hipStream_t s) {
  if (n == 0) return ACINO_OK;
  if (int rc = fk_attr()) return rc;
  hipLaunchKernelGGL(k_fk, dim3((unsigned)((n + FPB - 1) / FPB)), dim3(256), sizeof(FrameLds) * FPB, s, d_q, n,
                     ACINO_N_STATES, 0, 0, d_pos);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

int launch_fk_active(const double* d_xa_halo, int64_t n, double* d_pos, hipStream_t s) {
  if (n == 0) return ACINO_OK;
  if (int rc = fk_attr()) return rc;
  hipLaunchKernelGGL(k_fk, dim3((unsigned)((n + FPB - 1) / FPB)), dim3(256), sizeof(FrameLds) * FPB, s, d_xa_halo,
                     n, NP, HALO, 1, d_pos);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

}  // namespace acino

extern "C" int acino_cheetah_fk(const double* d_q, int64_t n_frames, double* d_pos, void* stream) {
  using namespace acino;
  ACINO_REQUIRE(n_frames >= 0, "n_frames");
  if (n_frames == 0) return ACINO_OK;
  ACINO_REQUIRE(d_q && d_pos, "null buffer");
  return launch_fk(d_q, n_frames, d_pos, (hipStream_t)stream);
}
