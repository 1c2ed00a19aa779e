// Sparse bundle adjustment of 3-D points (+ optionally the 6-DoF camera extrinsics) on fisheye cameras.
//
// Reference: src/calib/calib.py:307-341 (points only), :345-390 (points + extrinsics) - scipy least_squares(method='trf', loss='cauchy') over
// [rvecs, tvecs, points] with a finite-difference Jacobian and one cv2.fisheye.projectPoints call per
// observation.  Here: analytic Jacobians, Cauchy IRLS weights w = 1/(1 + (r/f)^2), Levenberg-Marquardt on the
// Schur complement of the point blocks (3x3 per point) onto the camera block (6C x 6C), rotations updated by a
// left perturbation R <- exp([dw]x) R (same minimiser, no rvec singularities).
//
// Two paths.  FUSED (up to seven cameras - every rig of the reference): one lane per (point, camera) slot, no coupling table
// in memory; k_sba_fused linearises, forms the point blocks and accumulates the Schur complement on the fp64 matrix cores in
// one pass, k_sba_backsub_fused back-substitutes and prices the trial iterate in one pass (see the block comment in front
// of them).  TABLE (more cameras, or ACINO_SBA_UNFUSED=1 as an independent cross-check): one thread per POINT walks its
// observations (CSR built by the host), the coupling blocks W_pc (6 x 3) live in a dense table [point][camera], camera-block
// sums go through 16 lane-private copies in LDS, the Schur complement is added with LDS atomics.
// History (config 5, 1.28 M points, 6.5 M observations, per LM iteration): table + LDS atomics 6.1 ms; table + matrix-core GEMM
// over the table 2.2 ms (round 4, W written once and read twice: 456 B per observation); fused 1.0 ms of kernels.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "dense80.hpp"

namespace acino {

constexpr int SBA_MAXC = ACINO_MAX_CAMS;

struct SbaIntr {
  double fx, fy, cx, cy;
  double d[12];   // fisheye: k1..k4 ; pinhole (cv2.projectPoints): k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4
};
constexpr int SBA_INTR = 16;

// cv2.fisheye projection of a camera-frame point and d(uv)/d(Xc)
template <bool JAC>
__device__ __forceinline__ void fisheye_cam(const SbaIntr& c, const double Xc[3], double uv[2], double J[2][3]) {
  // (three divisions instead of seven: 1 / z, 1 / r, 1 / (1 + r^2) - an fp64 division is ~12 instructions, four of them at quarter rate)
  const double iz = 1.0 / Xc[2];
  const double a = Xc[0] * iz, b = Xc[1] * iz;
  const double r2 = a * a + b * b;
  const double r = sqrt(r2);
  const double th = atan(r), th2 = th * th;
  const double thd = th * (1 + th2 * (c.d[0] + th2 * (c.d[1] + th2 * (c.d[2] + th2 * c.d[3]))));
  const bool small = !(r > 1e-8);
  const double ir = small ? 0.0 : 1.0 / r;
  const double m = small ? 1.0 : thd * ir;
  uv[0] = c.fx * a * m + c.cx;
  uv[1] = c.fy * b * m + c.cy;
  if (JAC) {
    double dm_da = 0.0, dm_db = 0.0;
    if (!small) {
      const double dthd = 1 + th2 * (3 * c.d[0] + th2 * (5 * c.d[1] + th2 * (7 * c.d[2] + th2 * 9 * c.d[3])));
      const double dm_dr = (dthd / (1 + r2) * r - thd) * (ir * ir);
      dm_da = dm_dr * a * ir;
      dm_db = dm_dr * b * ir;
    }
    const double du_da = c.fx * (m + a * dm_da), du_db = c.fx * a * dm_db;
    const double dv_da = c.fy * b * dm_da, dv_db = c.fy * (m + b * dm_db);
    J[0][0] = du_da * iz;
    J[0][1] = du_db * iz;
    J[0][2] = -(du_da * a + du_db * b) * iz;
    J[1][0] = dv_da * iz;
    J[1][1] = dv_db * iz;
    J[1][2] = -(dv_da * a + dv_db * b) * iz;
  }
}

// cv2.projectPoints (rational + tangential + thin-prism model; the skew entry of K is ignored, as OpenCV does)
template <bool JAC>
__device__ __forceinline__ void pinhole_cam(const SbaIntr& c, const double Xc[3], double uv[2], double J[2][3]) {
  const double* k = c.d;
  const double iz = 1.0 / Xc[2];
  const double a = Xc[0] * iz, b = Xc[1] * iz;
  const double r2 = a * a + b * b, r4 = r2 * r2, r6 = r4 * r2;
  const double num = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
  const double iden = 1.0 / (1 + k[5] * r2 + k[6] * r4 + k[7] * r6);
  const double rad = num * iden;
  const double xd = a * rad + 2 * k[2] * a * b + k[3] * (r2 + 2 * a * a) + k[8] * r2 + k[9] * r4;
  const double yd = b * rad + k[2] * (r2 + 2 * b * b) + 2 * k[3] * a * b + k[10] * r2 + k[11] * r4;
  uv[0] = c.fx * xd + c.cx;
  uv[1] = c.fy * yd + c.cy;
  if (JAC) {
    const double dnum = k[0] + 2 * k[1] * r2 + 3 * k[4] * r4, dden = k[5] + 2 * k[6] * r2 + 3 * k[7] * r4;
    const double drad = (dnum - rad * dden) * iden;                  // d rad / d r2
    const double sx = k[8] + 2 * k[9] * r2, sy = k[10] + 2 * k[11] * r2;
    const double dx_da = rad + 2 * a * a * drad + 2 * k[2] * b + 6 * k[3] * a + 2 * a * sx;
    const double dx_db = 2 * a * b * drad + 2 * k[2] * a + 2 * k[3] * b + 2 * b * sx;
    const double dy_da = 2 * a * b * drad + 2 * k[2] * a + 2 * k[3] * b + 2 * a * sy;
    const double dy_db = rad + 2 * b * b * drad + 6 * k[2] * b + 2 * k[3] * a + 2 * b * sy;
    const double du_da = c.fx * dx_da, du_db = c.fx * dx_db, dv_da = c.fy * dy_da, dv_db = c.fy * dy_db;
    J[0][0] = du_da * iz;
    J[0][1] = du_db * iz;
    J[0][2] = -(du_da * a + du_db * b) * iz;
    J[1][0] = dv_da * iz;
    J[1][1] = dv_db * iz;
    J[1][2] = -(dv_da * a + dv_db * b) * iz;
  }
}

struct SbaBuf {
  int C, P, M, opt_cams;
  int model, prec;       // 0 fisheye, 1 pinhole | ACINO_PREC_F64 or ACINO_PREC_BF16_ROWS
  double fs;
  const double* intr;    // [C][16]
  const double* uv;      // [M][2]
  const int* cam_idx;    // [M]
  const int* pt_start;   // [P+1]
  const int* pt_obs;     // [M] observation ids grouped by point
  double* V;             // [P][6]
  double* gp;            // [P][3]
  double* Vinv;          // [P][6]
  double* Wpc;           // [P][C][18]  (6x3 row-major per (point, camera) slot; zero where the camera does not see the point)
                         //             unfused path only (more than 7 cameras, ACINO_SBA_UNFUSED); null otherwise
  double* part3;         // [SBA_SCHUR_WG][4] per-workgroup partial sums of the fused kernels: cost | predicted reduction | trial cost
  int* slot;             // [P][C] observation id of (point, camera), -1 where the camera does not see the point (fused path)
  double* Spart;         // [SBA_SCHUR_WG + 32][n n + n + 27 C] per-workgroup partial sums [S | rhs | U | g_c] of k_sba_fused
  double* U;             // [C][21]
  double* gc;            // [C][6]
  double* S;             // [6C][6C]
  double* rhs;           // [6C]
  double* dc;            // [6C]
  double* dp;            // [P][3]
  double* scal;          // [8]: 0 cost, 1 pred, 2 max|g|, 3 numeric flag
};

// Evaluate at (Rt, pts): cost; with JAC also V, gp, Wpc, U, gc; optionally the residuals.
// PREC = ACINO_PREC_BF16_ROWS (BASELINE config 5, "bf16 residuals with fp32 accumulate", the SBA half): camera-frame
// point, projection and analytic Jacobians in fp64 as before; every residual and every Jacobian ROW (2x3 point part, 2x6
// camera part) rounded to bf16 before it enters the normal equations; the IRLS weights come from the stored residual; the
// point blocks V_p, g_p, the coupling blocks W_cp and the per-workgroup camera sums U_c, g_c are accumulated in fp32.  The
// cost is summed in fp64 from the unrounded residuals (accept / reject needs more than 8 bits); Schur complement,
// camera solve and the updates stay fp64.
template <bool JAC, int PREC>
__global__ void __launch_bounds__(256)
k_sba_point(SbaBuf B, const double* __restrict__ Rt, const double* __restrict__ pts, double* __restrict__ res_out) {
  typedef typename std::conditional<PREC == ACINO_PREC_F64, double, float>::type acc_t;
  extern __shared__ __attribute__((aligned(16))) char sba_smem[];
  acc_t* sU = reinterpret_cast<acc_t*>(sba_smem);          // [16][C][27]: copy (lane & 15)
  __shared__ double sred[4];
  const int tid = threadIdx.x, slot = tid & 15, nU = B.C * 27;
  if (JAC && B.opt_cams)
    for (int e = tid; e < 16 * nU; e += 256) sU[e] = 0.0;
  __syncthreads();
  const int p = blockIdx.x * 256 + tid;
  double cost = 0.0, gmax = 0.0;
  if (p < B.P) {
    const double X[3] = {pts[3 * p], pts[3 * p + 1], pts[3 * p + 2]};
    acc_t V[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    const double ifs2 = 1.0 / (B.fs * B.fs);
    for (int o = B.pt_start[p]; o < B.pt_start[p + 1]; ++o) {
      const int k = B.pt_obs[o], c = B.cam_idx[k];
      const double* R = Rt + 12 * c;
      const SbaIntr& in = *reinterpret_cast<const SbaIntr*>(B.intr + SBA_INTR * c);
      const double RX[3] = {R[0] * X[0] + R[1] * X[1] + R[2] * X[2], R[3] * X[0] + R[4] * X[1] + R[5] * X[2],
                            R[6] * X[0] + R[7] * X[1] + R[8] * X[2]};
      const double Xc[3] = {RX[0] + R[9], RX[1] + R[10], RX[2] + R[11]};
      double uvp[2], Jpi[2][3];
      if (B.model == 0) fisheye_cam<JAC>(in, Xc, uvp, Jpi);
      else pinhole_cam<JAC>(in, Xc, uvp, Jpi);
      const double r0 = uvp[0] - B.uv[2 * k], r1 = uvp[1] - B.uv[2 * k + 1];
      if (res_out) {
        res_out[2 * k] = r0;
        res_out[2 * k + 1] = r1;
      }
      const double z0 = r0 * r0 * ifs2, z1 = r1 * r1 * ifs2;
      cost += 0.5 * B.fs * B.fs * (log1p(z0) + log1p(z1));
      if (JAC) {
        acc_t w0, w1, rs0, rs1;
        acc_t Jp[2][3], Jc[2][6];
        if (PREC == ACINO_PREC_F64) {
          w0 = 1.0 / (1.0 + z0);
          w1 = 1.0 / (1.0 + z1);
          rs0 = r0;
          rs1 = r1;
        } else {
          rs0 = bf16_round((float)r0);
          rs1 = bf16_round((float)r1);
          const float fi = (float)ifs2;
          w0 = 1.0f / (1.0f + (float)rs0 * (float)rs0 * fi);
          w1 = 1.0f / (1.0f + (float)rs1 * (float)rs1 * fi);
        }
        auto st = [](double v) -> acc_t { return PREC == ACINO_PREC_F64 ? (acc_t)v : (acc_t)bf16_round((float)v); };
#pragma unroll
        for (int d = 0; d < 2; ++d) {
#pragma unroll
          for (int j = 0; j < 3; ++j) Jp[d][j] = st(Jpi[d][0] * R[j] + Jpi[d][1] * R[3 + j] + Jpi[d][2] * R[6 + j]);
          // d(Xc)/d(dw) = -[RX]x ;  d(Xc)/d(dt) = I
          Jc[d][0] = st(Jpi[d][1] * (-RX[2]) + Jpi[d][2] * RX[1]);
          Jc[d][1] = st(Jpi[d][0] * RX[2] - Jpi[d][2] * RX[0]);
          Jc[d][2] = st(-Jpi[d][0] * RX[1] + Jpi[d][1] * RX[0]);
          Jc[d][3] = st(Jpi[d][0]);
          Jc[d][4] = st(Jpi[d][1]);
          Jc[d][5] = st(Jpi[d][2]);
        }
        V[0] += w0 * Jp[0][0] * Jp[0][0] + w1 * Jp[1][0] * Jp[1][0];
        V[1] += w0 * Jp[0][0] * Jp[0][1] + w1 * Jp[1][0] * Jp[1][1];
        V[2] += w0 * Jp[0][0] * Jp[0][2] + w1 * Jp[1][0] * Jp[1][2];
        V[3] += w0 * Jp[0][1] * Jp[0][1] + w1 * Jp[1][1] * Jp[1][1];
        V[4] += w0 * Jp[0][1] * Jp[0][2] + w1 * Jp[1][1] * Jp[1][2];
        V[5] += w0 * Jp[0][2] * Jp[0][2] + w1 * Jp[1][2] * Jp[1][2];
#pragma unroll
        for (int j = 0; j < 3; ++j) g[j] += w0 * rs0 * Jp[0][j] + w1 * rs1 * Jp[1][j];
        if (B.opt_cams) {
          double* W = B.Wpc + 18 * ((size_t)p * B.C + c);
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int j = 0; j < 3; ++j) W[a * 3 + j] = w0 * Jc[0][a] * Jp[0][j] + w1 * Jc[1][a] * Jp[1][j];
          acc_t* su = sU + (slot * B.C + c) * 27;
          int q = 0;
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int bq = a; bq < 6; ++bq) atomicAdd(&su[q++], w0 * Jc[0][a] * Jc[0][bq] + w1 * Jc[1][a] * Jc[1][bq]);
#pragma unroll
          for (int a = 0; a < 6; ++a) atomicAdd(&su[21 + a], w0 * rs0 * Jc[0][a] + w1 * rs1 * Jc[1][a]);
        }
      }
    }
    if (JAC) {
#pragma unroll
      for (int q = 0; q < 6; ++q) B.V[6 * (size_t)p + q] = V[q];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        B.gp[3 * (size_t)p + j] = g[j];
        gmax = fmax(gmax, fabs(g[j]));
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    cost += __shfl_down(cost, off, 64);
    gmax = fmax(gmax, __shfl_down(gmax, off, 64));
  }
  if ((tid & 63) == 0) sred[tid >> 6] = cost;
  __syncthreads();
  if (tid == 0) atomicAdd(&B.scal[0], (sred[0] + sred[1]) + (sred[2] + sred[3]));
  if (JAC && (tid & 63) == 0) atomicMax(reinterpret_cast<unsigned long long*>(&B.scal[2]),
                                         (unsigned long long)__double_as_longlong(gmax));   // gmax >= 0: order-preserving
  if (JAC && B.opt_cams) {
    __syncthreads();
    for (int e = tid; e < nU; e += 256) {
      const int c = e / 27, q = e % 27;
      double v = 0.0;
#pragma unroll
      for (int sl = 0; sl < 16; ++sl) v += (double)sU[sl * nU + e];      // the 16 copies, fixed order
      if (v != 0.0) atomicAdd(q < 21 ? &B.U[21 * c + q] : &B.gc[6 * c + (q - 21)], v);
    }
  }
}

// Per point: Vinv = (V + lam diag V)^-1; Schur contributions S -= W Vinv W'^T, rhs -= W Vinv gp.
__global__ void __launch_bounds__(256) k_sba_schur(SbaBuf B, double lam) {
  extern __shared__ double sS[];   // [6C][6C] + [6C]
  const int n = 6 * B.C, tid = threadIdx.x;
  if (B.opt_cams)
    for (int e = tid; e < n * n + n; e += 256) sS[e] = 0.0;
  __syncthreads();
  const int p = blockIdx.x * 256 + tid;
  if (p < B.P) {
    const double* V = B.V + 6 * (size_t)p;
    const double a = V[0] * (1 + lam), b = V[1], c = V[2], d = V[3] * (1 + lam), e = V[4], f = V[5] * (1 + lam);
    const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    double det = a * c00 + b * c01 + c * c02;
    if (!(fabs(det) > 0.0)) det = 1.0;   // point without observations: harmless (gp = 0)
    const double id = 1.0 / det;
    double Vi[6] = {c00 * id, c01 * id, c02 * id, (a * f - c * c) * id, (b * c - a * e) * id, (a * d - b * b) * id};
#pragma unroll
    for (int q = 0; q < 6; ++q) B.Vinv[6 * (size_t)p + q] = Vi[q];
    if (B.opt_cams) {
      const double* g = B.gp + 3 * (size_t)p;
      const double Vg[3] = {Vi[0] * g[0] + Vi[1] * g[1] + Vi[2] * g[2], Vi[1] * g[0] + Vi[3] * g[1] + Vi[4] * g[2],
                            Vi[2] * g[0] + Vi[4] * g[1] + Vi[5] * g[2]};
      for (int o1 = B.pt_start[p]; o1 < B.pt_start[p + 1]; ++o1) {
        const int k1 = B.pt_obs[o1], c1 = B.cam_idx[k1];
        const double* W1 = B.Wpc + 18 * ((size_t)p * B.C + c1);
        double T[6][3];   // W1 Vinv
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          T[r][0] = W1[r * 3] * Vi[0] + W1[r * 3 + 1] * Vi[1] + W1[r * 3 + 2] * Vi[2];
          T[r][1] = W1[r * 3] * Vi[1] + W1[r * 3 + 1] * Vi[3] + W1[r * 3 + 2] * Vi[4];
          T[r][2] = W1[r * 3] * Vi[2] + W1[r * 3 + 1] * Vi[4] + W1[r * 3 + 2] * Vi[5];
          atomicAdd(&sS[n * n + 6 * c1 + r], -(W1[r * 3] * Vg[0] + W1[r * 3 + 1] * Vg[1] + W1[r * 3 + 2] * Vg[2]));
        }
        for (int o2 = B.pt_start[p]; o2 < B.pt_start[p + 1]; ++o2) {
          const int k2 = B.pt_obs[o2], c2 = B.cam_idx[k2];
          const double* W2 = B.Wpc + 18 * ((size_t)p * B.C + c2);
#pragma unroll
          for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int s = 0; s < 6; ++s)
              atomicAdd(&sS[(6 * c1 + r) * n + 6 * c2 + s],
                        -(T[r][0] * W2[s * 3] + T[r][1] * W2[s * 3 + 1] + T[r][2] * W2[s * 3 + 2]));
        }
      }
    }
  }
  if (B.opt_cams) {
    __syncthreads();
    for (int e = tid; e < n * n; e += 256)
      if (sS[e] != 0.0) atomicAdd(&B.S[e], sS[e]);
    for (int e = tid; e < n; e += 256)
      if (sS[n * n + e] != 0.0) atomicAdd(&B.rhs[e], sS[n * n + e]);
  }
}


constexpr int SBA_SCHUR_WG = 1024;      // workgroups (x 4 waves) of k_sba_fused = records of partial sums
// the records, added in a fixed order in two stages (n_part <= 1024 records of n n + n doubles): stage 0 sums every 32nd record
// into 32 intermediate records (behind the partial records in Spart), stage 1 sums those into [S | rhs]
// (fused path: records of n n + n + 27 C doubles, the tail [U | g_c] goes to B.U)
__global__ void __launch_bounds__(256) k_sba_schur_reduce(SbaBuf B, int n_part, int stage, int tail) {
  const int n = 6 * B.C, tot = n * n + n + tail;
  const int e = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
  if (e >= tot) return;
  double* mid = B.Spart + (size_t)SBA_SCHUR_WG * tot;
  double s = 0.0;
  if (stage == 0) {
    for (int w = g; w < n_part; w += 32) s += B.Spart[(size_t)w * tot + e];
    mid[(size_t)g * tot + e] = s;
  } else {
    for (int w = 0; w < 32; ++w) s += mid[(size_t)w * tot + e];
    if (e < n * n + n) B.S[e] = s;                       // (S | rhs contiguous)
    else B.U[e - (n * n + n)] = s;                       // (U | g_c contiguous)
  }
}

// ================================================================================================================
// The fused path (6 C + 1 <= 48): no coupling table in memory.
//
// k_sba_fused: ONE lane per (point, camera) slot - a wave takes 64 / C points per batch (10 for six cameras, 60 lanes) and the
// lane's camera never changes, so pose, intrinsics and the lane's share of the camera block U_c, g_c stay in registers for
// the whole kernel (no LDS atomics).  Per batch: projection, analytic Jacobian, IRLS weight, W_pc (6 x 3) in registers; the
// point blocks V_p, g_p through a wave-private LDS exchange (fixed summation order over the cameras); every lane inverts its
// point's damped V_p and forms Y_pc = W_pc V_p^-1; W and Y go to the wave's LDS slab in point-major order and the Schur
// complement [S ; rhs^T] = - sum_p [Y_p ; (V_p^-1 g_p)^T] W_p^T is accumulated on the fp64 matrix cores from there: the
// (point, coordinate) pairs of the batch are the k dimension, four per v_mfma_f64_16x16x4 (30 pairs -> 8 k-steps x 6 lower
// tiles for 10 points; the table form spent one k-step of three per point).  Partial sums [S | rhs | U | g_c] per workgroup,
// two-stage fixed-order reduction (k_sba_schur_reduce).  HBM traffic per iteration: slots, detections, points in; V, V^-1,
// g_p out - ~60 B per observation against ~450 for the table form (W written once, read twice).
//
// k_sba_backsub_fused: the same lane mapping; recomputes the observation's Jacobian rows, forms W_pc^T dc without W
// (sum_d w_d J_p[d] (J_c[d] . dc)), the point step, the trial point and - with the trial poses already applied by
// k_sba_apply_cams - the trial cost of the lane's observation: the separate cost pass over the trial iterate is gone.
template <int PREC>
struct SbaObs {
  typedef typename std::conditional<PREC == ACINO_PREC_F64, double, float>::type acc_t;
  acc_t Jp[2][3], Jc[2][6], w[2], rs[2];
  double cost;
};
template <int PREC, bool JAC, int MODEL, bool COST = true>
__device__ __forceinline__ void sba_observe(double fs, const double (&R)[12], const SbaIntr& in, const double (&X)[3],
                                            double u, double v, SbaObs<PREC>& o) {
  typedef typename SbaObs<PREC>::acc_t acc_t;
  const double RX[3] = {R[0] * X[0] + R[1] * X[1] + R[2] * X[2], R[3] * X[0] + R[4] * X[1] + R[5] * X[2],
                        R[6] * X[0] + R[7] * X[1] + R[8] * X[2]};
  const double Xc[3] = {RX[0] + R[9], RX[1] + R[10], RX[2] + R[11]};
  double uvp[2], Jpi[2][3];
  if (MODEL == 0) fisheye_cam<JAC>(in, Xc, uvp, Jpi);
  else pinhole_cam<JAC>(in, Xc, uvp, Jpi);
  const double r0 = uvp[0] - u, r1 = uvp[1] - v;
  const double ifs2 = 1.0 / (fs * fs);
  const double z0 = r0 * r0 * ifs2, z1 = r1 * r1 * ifs2;
  o.cost = COST ? 0.5 * fs * fs * (log1p(z0) + log1p(z1)) : 0.0;
  if (JAC) {
    if (PREC == ACINO_PREC_F64) {
      o.w[0] = 1.0 / (1.0 + z0);
      o.w[1] = 1.0 / (1.0 + z1);
      o.rs[0] = r0;
      o.rs[1] = r1;
    } else {
      o.rs[0] = bf16_round((float)r0);
      o.rs[1] = bf16_round((float)r1);
      const float fi = (float)ifs2;
      o.w[0] = 1.0f / (1.0f + (float)o.rs[0] * (float)o.rs[0] * fi);
      o.w[1] = 1.0f / (1.0f + (float)o.rs[1] * (float)o.rs[1] * fi);
    }
    auto st = [](double x) -> acc_t { return PREC == ACINO_PREC_F64 ? (acc_t)x : (acc_t)bf16_round((float)x); };
#pragma unroll
    for (int d = 0; d < 2; ++d) {
#pragma unroll
      for (int j = 0; j < 3; ++j) o.Jp[d][j] = st(Jpi[d][0] * R[j] + Jpi[d][1] * R[3 + j] + Jpi[d][2] * R[6 + j]);
      o.Jc[d][0] = st(Jpi[d][1] * (-RX[2]) + Jpi[d][2] * RX[1]);      // d(Xc)/d(dw) = -[RX]x ;  d(Xc)/d(dt) = I
      o.Jc[d][1] = st(Jpi[d][0] * RX[2] - Jpi[d][2] * RX[0]);
      o.Jc[d][2] = st(-Jpi[d][0] * RX[1] + Jpi[d][1] * RX[0]);
      o.Jc[d][3] = st(Jpi[d][0]);
      o.Jc[d][4] = st(Jpi[d][1]);
      o.Jc[d][5] = st(Jpi[d][2]);
    }
  }
}

// Input check of BOTH paths: a camera index out of range, or two observations of one point by one camera (the table path keeps
// ONE coupling block per (point, camera): a second observation would overwrite it and leave the Schur complement inconsistent
// with U and V), raise scal[7].  One thread per point, the cameras seen so far as a bit mask (at most 16 cameras).
__global__ void __launch_bounds__(256) k_sba_check(SbaBuf B) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= B.P) return;
  unsigned seen = 0;
  bool bad = false;
  for (int o = B.pt_start[p]; o < B.pt_start[p + 1]; ++o) {
    const int c = B.cam_idx[B.pt_obs[o]];
    if (c < 0 || c >= B.C) {
      bad = true;
    } else {
      bad = bad || ((seen >> c) & 1u);
      seen |= 1u << c;
    }
  }
  if (bad) B.scal[7] = 1.0;
}

// slot[p][c] <- observation id (the table is preset to -1); two observations of one (point, camera) pair raise scal[7]
__global__ void __launch_bounds__(256) k_sba_slots(SbaBuf B) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= B.P) return;
  for (int o = B.pt_start[p]; o < B.pt_start[p + 1]; ++o) {
    const int k = B.pt_obs[o], c = B.cam_idx[k];
    const bool bad = c < 0 || c >= B.C;
    const int old = bad ? 0 : atomicExch(&B.slot[(size_t)p * B.C + c], k);
    if (bad || old != -1) B.scal[7] = 1.0;
  }
}

constexpr int FU_T = 256;
// Operand slab of one wave: A = -[Y ; (V^-1 g)^T] and B = W as [16 k entries][48 rows] each (entry = (point, coordinate) pair of
// a CHUNK of five points, entry 15 and rows beyond 6 C (+ 1) stay zero), + the [64 / C][C][9] exchange of the point blocks.
// Lane (li, lk) reads row 16 t + li of entry 4 s + lk: one base address per lane, everything else immediate offsets, 48 rows
// = the stride that spreads the two k-entries of a half-wave over all 64 banks.
constexpr int FU_PC = 5, FU_E = 16, FU_ROWS = 48;
#ifndef FU_OCC
#define FU_OCC 2
#endif
__host__ __device__ inline int fused_slab_doubles(int C) { return 2 * FU_E * FU_ROWS + (64 / C) * C * 9; }
__host__ __device__ inline size_t fused_lds_bytes(int C) {
  size_t d = (size_t)4 * fused_slab_doubles(C);
  if (d < (size_t)4 * 6 * 256) d = (size_t)4 * 6 * 256;  // the tiles of the four waves
  if (d < (size_t)4 * 64 * 27) d = (size_t)4 * 64 * 27;  // the camera sums of all lanes
  return d * 8;
}

// COST: also the cost of the iterate (first evaluation only: afterwards it is the trial cost of the accepted step)
template <int PREC, int MODEL, bool COST>
__global__ void __launch_bounds__(FU_T, FU_OCC)
k_sba_fused(SbaBuf B, const double* __restrict__ Rt, const double* __restrict__ pts, double lam, int pts_per_wave) {
  typedef typename SbaObs<PREC>::acc_t acc_t;
  extern __shared__ __attribute__((aligned(16))) char fu_smem[];
  __shared__ double sred[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int C = B.C, n = 6 * C, ppp = 64 / C;
  double* Aop = reinterpret_cast<double*>(fu_smem) + (size_t)wave * fused_slab_doubles(C);
  double* Bop = Aop + FU_E * FU_ROWS;
  double* part = Bop + FU_E * FU_ROWS;                   // [ppp][C][9]
  const bool lane_on = lane < ppp * C;
  const int pl = lane_on ? lane / C : 0, c = lane_on ? lane % C : 0;
  // pose and intrinsics of the lane's camera: staged in LDS, read back per batch (20 registers live only while they are used)
  __shared__ double sCam[7][28];                         // (fused path: at most seven cameras)
  for (int e = tid; e < C * 28; e += FU_T) {
    const int cam = e / 28, q = e % 28;
    sCam[cam][q] = q < 12 ? Rt[12 * cam + q] : B.intr[SBA_INTR * cam + (q - 12)];
  }
  for (int e = lane; e < 2 * FU_E * FU_ROWS; e += 64) Aop[e] = 0.0;    // (entry 15, the rows beyond 6 C + 1: zero for good)
  __syncthreads();
  acc_t Uacc[27];
#pragma unroll
  for (int q = 0; q < 27; ++q) Uacc[q] = 0;
  double cost = 0.0, gmax = 0.0;
  d4 acc[6];
#pragma unroll
  for (int t = 0; t < 6; ++t) acc[t] = d4{0, 0, 0, 0};
  const int gw = blockIdx.x * 4 + wave;
  const int p0 = min(gw * pts_per_wave, B.P), p1 = min(p0 + pts_per_wave, B.P);
  const int n_chunk = (ppp + FU_PC - 1) / FU_PC;
  const bool row2 = n + 1 > 32;                          // (five cameras or fewer: tile row 2 is empty)
  const int pch = pl / FU_PC, pin = pl - FU_PC * pch;    // the lane's chunk and its place in it
  const double* ordA = Aop + lk * FU_ROWS + li;          // operands of k-step s, tile row t: + s * 4 * 48 + 16 t
  const double* ordB = Bop + lk * FU_ROWS + li;
  auto ldop = [&](int s_, double (&A)[3], double (&Bo)[3]) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      A[t] = ordA[s_ * 4 * FU_ROWS + 16 * t];
      Bo[t] = ordB[s_ * 4 * FU_ROWS + 16 * t];
    }
  };
  auto products = [&](const double (&A)[3], const double (&Bo)[3]) {
    acc[0] = mfma(A[0], Bo[0], acc[0]);
    acc[1] = mfma(A[1], Bo[0], acc[1]);
    acc[2] = mfma(A[1], Bo[1], acc[2]);
    if (row2) {
      acc[3] = mfma(A[2], Bo[0], acc[3]);
      acc[4] = mfma(A[2], Bo[1], acc[4]);
      acc[5] = mfma(A[2], Bo[2], acc[5]);
    }
  };
  // the first batch's slot, point and detection; afterwards the next batch's are requested a phase ahead
  // (every load unconditional on a valid address: a conditional load is a branch and a full wait per element)
  int kq = 0;
  double Xq[3] = {0, 0, 0}, uq = 0, vq = 0;
  if (p0 < p1) {
    const int p = min(p0 + pl, p1 - 1);
    kq = B.slot[(size_t)p * C + c];
    Xq[0] = pts[3 * (size_t)p];
    Xq[1] = pts[3 * (size_t)p + 1];
    Xq[2] = pts[3 * (size_t)p + 2];
    const int kc = max(kq, 0);
    uq = B.uv[2 * (size_t)kc];
    vq = B.uv[2 * (size_t)kc + 1];
  }
  for (int pb = p0; pb < p1; pb += ppp) {
    const int nb = min(ppp, p1 - pb);
    const bool on = lane_on && pl < nb;
    const int p = pb + (on ? pl : 0);
    const bool valid = on && kq >= 0;
    const double X[3] = {Xq[0], Xq[1], Xq[2]};
    const double u = uq, v = vq;
    // next batch: slot and point now, the detection (which needs the slot) before the matrix-core phase
    const int pn = min(pb + ppp + pl, B.P - 1);
    const int kn = B.slot[(size_t)pn * C + c];
    const double Xn0 = pts[3 * (size_t)pn], Xn1 = pts[3 * (size_t)pn + 1], Xn2 = pts[3 * (size_t)pn + 2];
    SbaObs<PREC> o;
    {
      double R[12];
      SbaIntr in;
      const double* cp = sCam[c];
#pragma unroll
      for (int q = 0; q < 12; ++q) R[q] = cp[q];
      in.fx = cp[12]; in.fy = cp[13]; in.cx = cp[14]; in.cy = cp[15];
#pragma unroll
      for (int q = 0; q < 12; ++q) in.d[q] = (MODEL == 0 && q >= 4) ? 0.0 : cp[16 + q];
      sba_observe<PREC, true, MODEL, COST>(B.fs, R, in, X, u, v, o);
    }
    // what a lane without an observation computed is discarded by SELECTS (it may be inf / NaN)
    const acc_t w0 = valid ? o.w[0] : (acc_t)0, w1 = valid ? o.w[1] : (acc_t)0;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
#pragma unroll
      for (int j = 0; j < 3; ++j) o.Jp[d][j] = valid ? o.Jp[d][j] : (acc_t)0;
#pragma unroll
      for (int a = 0; a < 6; ++a) o.Jc[d][a] = valid ? o.Jc[d][a] : (acc_t)0;
      o.rs[d] = valid ? o.rs[d] : (acc_t)0;
    }
    if (COST) cost += valid ? o.cost : 0.0;
    acc_t W[18];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int j = 0; j < 3; ++j) W[a * 3 + j] = w0 * o.Jc[0][a] * o.Jp[0][j] + w1 * o.Jc[1][a] * o.Jp[1][j];
    if (B.opt_cams) {
      int q = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int bq = a; bq < 6; ++bq) Uacc[q++] += w0 * o.Jc[0][a] * o.Jc[0][bq] + w1 * o.Jc[1][a] * o.Jc[1][bq];
#pragma unroll
      for (int a = 0; a < 6; ++a) Uacc[21 + a] += w0 * o.rs[0] * o.Jc[0][a] + w1 * o.rs[1] * o.Jc[1][a];
    }
    // the point blocks: every lane's share -> LDS -> every lane of the point sums the C shares in camera order
    if (lane_on) {
      double* pp = part + (pl * C + c) * 9;
      pp[0] = w0 * o.Jp[0][0] * o.Jp[0][0] + w1 * o.Jp[1][0] * o.Jp[1][0];
      pp[1] = w0 * o.Jp[0][0] * o.Jp[0][1] + w1 * o.Jp[1][0] * o.Jp[1][1];
      pp[2] = w0 * o.Jp[0][0] * o.Jp[0][2] + w1 * o.Jp[1][0] * o.Jp[1][2];
      pp[3] = w0 * o.Jp[0][1] * o.Jp[0][1] + w1 * o.Jp[1][1] * o.Jp[1][1];
      pp[4] = w0 * o.Jp[0][1] * o.Jp[0][2] + w1 * o.Jp[1][1] * o.Jp[1][2];
      pp[5] = w0 * o.Jp[0][2] * o.Jp[0][2] + w1 * o.Jp[1][2] * o.Jp[1][2];
#pragma unroll
      for (int j = 0; j < 3; ++j) pp[6 + j] = w0 * o.rs[0] * o.Jp[0][j] + w1 * o.rs[1] * o.Jp[1][j];
    }
    acc_t Vs[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) Vs[q] = 0;
    for (int cc = 0; cc < C; ++cc) {
      const double* pp = part + (pl * C + cc) * 9;
#pragma unroll
      for (int q = 0; q < 9; ++q) Vs[q] += (acc_t)pp[q];
    }
    const double V0 = Vs[0], V1 = Vs[1], V2 = Vs[2], V3 = Vs[3], V4 = Vs[4], V5 = Vs[5];
    const double g0 = Vs[6], g1 = Vs[7], g2 = Vs[8];
    // V^-1 of the damped point block (closed form)
    const double a = V0 * (1 + lam), b = V1, cc2 = V2, d = V3 * (1 + lam), e = V4, f = V5 * (1 + lam);
    const double c00 = d * f - e * e, c01 = cc2 * e - b * f, c02 = b * e - cc2 * d;
    double det = a * c00 + b * c01 + cc2 * c02;
    if (!(fabs(det) > 0.0)) det = 1.0;                   // point without observations: harmless (g = 0)
    const double id = 1.0 / det;
    const double Vi[6] = {c00 * id, c01 * id, c02 * id, (a * f - cc2 * cc2) * id, (b * cc2 - a * e) * id, (a * d - b * b) * id};
    if (on && c == 0) {
      double* Vg = B.V + 6 * (size_t)p;
      double* Ig = B.Vinv + 6 * (size_t)p;
      Vg[0] = V0; Vg[1] = V1; Vg[2] = V2; Vg[3] = V3; Vg[4] = V4; Vg[5] = V5;
#pragma unroll
      for (int q = 0; q < 6; ++q) Ig[q] = Vi[q];
      B.gp[3 * (size_t)p] = g0;
      B.gp[3 * (size_t)p + 1] = g1;
      B.gp[3 * (size_t)p + 2] = g2;
      gmax = fmax(gmax, fmax(fabs(g0), fmax(fabs(g1), fabs(g2))));
    }
    // the next batch's detections (its slots have arrived long ago); they arrive behind the matrix-core phase
    kq = kn;
    Xq[0] = Xn0;
    Xq[1] = Xn1;
    Xq[2] = Xn2;
    {
      const int kc = max(kn, 0);
      uq = B.uv[2 * (size_t)kc];
      vq = B.uv[2 * (size_t)kc + 1];
    }
    if (!B.opt_cams) continue;
    // the batch on the matrix cores, five points (15 k entries = four instructions per tile) at a time: the chunk's lanes write
    // their rows of B = W and A = -W V^-1 (lane of camera 0 also the right-hand-side row -V^-1 g), then four k-steps with the
    // operands of step s + 1 requested before the products of step s.  (A wave's LDS operations are ordered: no barrier.)
    double Yn[18];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double x0 = W[r * 3], x1 = W[r * 3 + 1], x2 = W[r * 3 + 2];
      Yn[r * 3] = -(x0 * Vi[0] + x1 * Vi[1] + x2 * Vi[2]);
      Yn[r * 3 + 1] = -(x0 * Vi[1] + x1 * Vi[3] + x2 * Vi[4]);
      Yn[r * 3 + 2] = -(x0 * Vi[2] + x1 * Vi[4] + x2 * Vi[5]);
    }
    const double vgn[3] = {-(Vi[0] * g0 + Vi[1] * g1 + Vi[2] * g2), -(Vi[1] * g0 + Vi[3] * g1 + Vi[4] * g2),
                           -(Vi[2] * g0 + Vi[4] * g1 + Vi[5] * g2)};
    for (int j = 0; j < n_chunk; ++j) {
      const int cnt = min(FU_PC, ppp - FU_PC * j);       // points of this chunk (the batch's last chunk may be short)
      if (cnt < FU_PC)                                   // entries of the points it lacks: zero again
        for (int e = lane; e < (FU_PC - cnt) * 3 * (n + 1); e += 64) {
          const int en = 3 * cnt + e / (n + 1), rw = e % (n + 1);
          Aop[en * FU_ROWS + rw] = 0.0;
          Bop[en * FU_ROWS + rw] = 0.0;
        }
      if (lane_on && pch == j) {
        double* aq = Aop + 3 * pin * FU_ROWS + 6 * c;
        double* bq = Bop + 3 * pin * FU_ROWS + 6 * c;
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
#pragma unroll
          for (int r = 0; r < 6; ++r) {
            aq[jj * FU_ROWS + r] = Yn[r * 3 + jj];
            bq[jj * FU_ROWS + r] = W[r * 3 + jj];
          }
        if (c == 0) {
#pragma unroll
          for (int jj = 0; jj < 3; ++jj) Aop[(3 * pin + jj) * FU_ROWS + n] = vgn[jj];
        }
      }
      double A0[3], B0[3], A1[3], B1[3];
      ldop(0, A0, B0);
      ldop(1, A1, B1);
      __builtin_amdgcn_sched_barrier(0);
      products(A0, B0);
      __builtin_amdgcn_sched_barrier(0);
      ldop(2, A0, B0);
      __builtin_amdgcn_sched_barrier(0);
      products(A1, B1);
      __builtin_amdgcn_sched_barrier(0);
      ldop(3, A1, B1);
      __builtin_amdgcn_sched_barrier(0);
      products(A0, B0);
      __builtin_amdgcn_sched_barrier(0);
      products(A1, B1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // cost, max |g_p|
  for (int off = 32; off > 0; off >>= 1) {
    cost += __shfl_down(cost, off, 64);
    gmax = fmax(gmax, __shfl_down(gmax, off, 64));
  }
  if (lane == 0) {
    sred[wave] = cost;
    atomicMax(reinterpret_cast<unsigned long long*>(&B.scal[2]), (unsigned long long)__double_as_longlong(gmax));   // gmax >= 0: order-preserving
  }
  __syncthreads();                                       // (also: the slabs are free)
  if (tid == 0) B.part3[4 * blockIdx.x] = (sred[0] + sred[1]) + (sred[2] + sred[3]);   // (summed in a fixed order by k_sba_sum_parts)
  if (!B.opt_cams) return;
  // tiles: wave -> LDS -> the workgroup's record.  C layout: register q of lane (li, lk) = entry [row lk + 4 q][column li]
  double* sT = reinterpret_cast<double*>(fu_smem);       // [4][6][256]
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) sT[(wave * 6 + t) * 256 + (lk + 4 * q) * 16 + li] = acc[t][q];
  __syncthreads();
  const int tot = n * n + n + 27 * C;
  double* out = B.Spart + (size_t)blockIdx.x * tot;
  for (int e = tid; e < 6 * 256; e += FU_T) {
    const int t = e >> 8, r = (e >> 4) & 15, cc = e & 15;
    const int ib = t < 1 ? 0 : (t < 3 ? 1 : 2), jb = t < 1 ? 0 : (t < 3 ? t - 1 : t - 3);
    const int Rw = 16 * ib + r, Cw = 16 * jb + cc;
    const double v = (sT[(0 * 6 + t) * 256 + (e & 255)] + sT[(1 * 6 + t) * 256 + (e & 255)]) +
                     (sT[(2 * 6 + t) * 256 + (e & 255)] + sT[(3 * 6 + t) * 256 + (e & 255)]);
    if (Cw >= n) continue;
    if (Rw < n) {
      out[Rw * n + Cw] = v;
      if (ib != jb) out[Cw * n + Rw] = v;                // (diagonal tiles hold both triangles already)
    } else if (Rw == n) {
      out[n * n + Cw] = v;
    }
  }
  __syncthreads();
  // camera sums: lane -> LDS, then the lanes of one camera (c, c + C, ...) of the four waves in a fixed order
  double* sU = reinterpret_cast<double*>(fu_smem);       // [4][64][27]
#pragma unroll
  for (int q = 0; q < 27; ++q) sU[(wave * 64 + lane) * 27 + q] = lane_on ? (double)Uacc[q] : 0.0;
  __syncthreads();
  for (int e = tid; e < 27 * C; e += FU_T) {
    const int cam = e / 27, q = e % 27;
    double v = 0.0;
    for (int w = 0; w < 4; ++w)
      for (int j = 0; j < ppp; ++j) v += sU[(w * 64 + cam + C * j) * 27 + q];
    out[n * n + n + (q < 21 ? 21 * cam + q : 21 * C + 6 * cam + (q - 21))] = v;
  }
}

template <int PREC, int MODEL>
__global__ void __launch_bounds__(FU_T)
k_sba_backsub_fused(SbaBuf B, double lam, const double* __restrict__ Rt, const double* __restrict__ Rt_t,
                    const double* __restrict__ pts, double* __restrict__ pts_t, int pts_per_wave) {
  typedef typename SbaObs<PREC>::acc_t acc_t;
  __shared__ double sx[4][64 * 4];                       // per wave: [ppp][C][3] shares, then [ppp][4] trial points
  __shared__ double sred[2][4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int C = B.C, ppp = 64 / C;
  double* sh = sx[wave];
  const bool lane_on = lane < ppp * C;
  const int pl = lane_on ? lane / C : 0, c = lane_on ? lane % C : 0;
  double R[12], Rn[12], dc[6];
#pragma unroll
  for (int q = 0; q < 12; ++q) {
    R[q] = Rt[12 * c + q];
    Rn[q] = Rt_t[12 * c + q];
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) dc[q] = B.opt_cams ? B.dc[6 * c + q] : 0.0;
  SbaIntr in;
  {
    const double* ip = B.intr + SBA_INTR * c;
    in.fx = ip[0]; in.fy = ip[1]; in.cx = ip[2]; in.cy = ip[3];
#pragma unroll
    for (int q = 0; q < 12; ++q) in.d[q] = (MODEL == 0 && q >= 4) ? 0.0 : ip[4 + q];
  }
  double pred = 0.0, cost_t = 0.0;
  const int gw = blockIdx.x * 4 + wave;
  const int p0 = min(gw * pts_per_wave, B.P), p1 = min(p0 + pts_per_wave, B.P);
  for (int pb = p0; pb < p1; pb += ppp) {
    const int nb = min(ppp, p1 - pb);
    const bool on = lane_on && pl < nb;
    const int p = pb + (on ? pl : 0);
    int k = B.slot[(size_t)p * C + c];
    const bool valid = on && k >= 0;
    k = valid ? k : 0;
    const double X[3] = {pts[3 * (size_t)p], pts[3 * (size_t)p + 1], pts[3 * (size_t)p + 2]};
    const double u = B.uv[2 * (size_t)k], v = B.uv[2 * (size_t)k + 1];
    double s3[3] = {0, 0, 0};
    if (B.opt_cams) {
      SbaObs<PREC> o;
      sba_observe<PREC, true, MODEL>(B.fs, R, in, X, u, v, o);
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        acc_t jd = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) jd += o.Jc[d][a] * (acc_t)dc[a];
        const acc_t wj = o.w[d] * jd;
#pragma unroll
        for (int j = 0; j < 3; ++j) s3[j] += (double)(wj * o.Jp[d][j]);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) s3[j] = valid ? s3[j] : 0.0;
    }
    if (lane_on) {
#pragma unroll
      for (int j = 0; j < 3; ++j) sh[(pl * C + c) * 3 + j] = s3[j];
    }
    double Xt[3] = {X[0], X[1], X[2]};
    double st[3] = {B.gp[3 * (size_t)p], B.gp[3 * (size_t)p + 1], B.gp[3 * (size_t)p + 2]};
    const double g0 = st[0], g1 = st[1], g2 = st[2];
    for (int cc = 0; cc < C; ++cc)
#pragma unroll
      for (int j = 0; j < 3; ++j) st[j] += sh[(pl * C + cc) * 3 + j];
    {
      const double* Vi = B.Vinv + 6 * (size_t)p;
      const double d0 = -(Vi[0] * st[0] + Vi[1] * st[1] + Vi[2] * st[2]), d1 = -(Vi[1] * st[0] + Vi[3] * st[1] + Vi[4] * st[2]),
                   d2 = -(Vi[2] * st[0] + Vi[4] * st[1] + Vi[5] * st[2]);
      Xt[0] += d0;
      Xt[1] += d1;
      Xt[2] += d2;
      if (on && c == 0) {
        pts_t[3 * (size_t)p] = Xt[0];
        pts_t[3 * (size_t)p + 1] = Xt[1];
        pts_t[3 * (size_t)p + 2] = Xt[2];
        const double* V = B.V + 6 * (size_t)p;
        pred += 0.5 * (d0 * (lam * V[0] * d0 - g0) + d1 * (lam * V[3] * d1 - g1) + d2 * (lam * V[5] * d2 - g2));
      }
    }
    // the trial cost of this observation: trial point (every lane of the point formed the same one), trial pose
    SbaObs<ACINO_PREC_F64> ot;
    sba_observe<ACINO_PREC_F64, false, MODEL>(B.fs, Rn, in, Xt, u, v, ot);
    cost_t += valid ? ot.cost : 0.0;
  }
  for (int off = 32; off > 0; off >>= 1) {
    pred += __shfl_down(pred, off, 64);
    cost_t += __shfl_down(cost_t, off, 64);
  }
  if (lane == 0) {
    sred[0][wave] = pred;
    sred[1][wave] = cost_t;
  }
  __syncthreads();
  if (tid == 0) {
    B.part3[4 * blockIdx.x + 1] = (sred[0][0] + sred[0][1]) + (sred[0][2] + sred[0][3]);
    B.part3[4 * blockIdx.x + 2] = (sred[1][0] + sred[1][1]) + (sred[1][2] + sred[1][3]);
  }
}

// scal[dst] <- sum over the workgroups of part3[.][src], in a fixed order (same launch geometry -> same bits): the cost and
// the predicted reduction decide accept / reject, so a solve repeats bit for bit only if they do.  (up to two sums per launch)
__global__ void __launch_bounds__(256) k_sba_sum_parts(SbaBuf B, int n_wg, int src0, int dst0, int src1, int dst1) {
  __shared__ double sh[2][256];
  const int tid = threadIdx.x;
  double a0 = 0.0, a1 = 0.0;
  for (int w = tid; w < n_wg; w += 256) {
    a0 += B.part3[4 * w + src0];
    if (src1 >= 0) a1 += B.part3[4 * w + src1];
  }
  sh[0][tid] = a0;
  sh[1][tid] = a1;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
      sh[0][tid] += sh[0][tid + off];
      sh[1][tid] += sh[1][tid + off];
    }
    __syncthreads();
  }
  if (tid == 0) {
    B.scal[dst0] = sh[0][0];
    if (src1 >= 0) B.scal[dst1] = sh[1][0];
  }
}

// Reduced camera system (U + lam diag U + S) dc = -(gc + rhs_schur): dense Cholesky in LDS, n = 6C <= 96.
__global__ void __launch_bounds__(256) k_sba_cam_solve(SbaBuf B, double lam) {
  extern __shared__ double sA[];   // [n][n+1] + [n]
  const int n = 6 * B.C, ld = n + 1, tid = threadIdx.x;
  double* sb = sA + n * ld;
  for (int e = tid; e < n * n; e += 256) {
    const int r = e / n, c = e % n;
    double v = B.S[e];
    if (r / 6 == c / 6) {
      const int cam = r / 6, a = r % 6, b = c % 6;
      const int lo = a < b ? a : b, hi = a < b ? b : a;
      const int q = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);
      double u = B.U[21 * cam + q];
      if (a == b) u += lam * u + 1e-300;
      v += u;
    }
    sA[r * ld + c] = v;
  }
  for (int e = tid; e < n; e += 256) sb[e] = -(B.gc[e] + B.rhs[e]);
  __syncthreads();
  for (int j = 0; j < n; ++j) {
    if (tid == 0) {
      double d = sA[j * ld + j];
      if (!(d > 0.0)) {
        B.scal[3] = 1.0;
        d = 1.0;
      }
      sA[j * ld + j] = sqrt(d);
    }
    __syncthreads();
    const double dj = sA[j * ld + j];
    for (int i = j + 1 + tid; i < n; i += 256) sA[i * ld + j] /= dj;
    __syncthreads();
    for (int e = tid; e < (n - j - 1) * (n - j - 1); e += 256) {
      const int i = j + 1 + e / (n - j - 1), k = j + 1 + e % (n - j - 1);
      if (k <= i) sA[i * ld + k] -= sA[i * ld + j] * sA[k * ld + j];
    }
    __syncthreads();
  }
  if (tid == 0) {   // tiny triangular solves
    for (int i = 0; i < n; ++i) {
      double s = sb[i];
      for (int k = 0; k < i; ++k) s -= sA[i * ld + k] * sb[k];
      sb[i] = s / sA[i * ld + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = sb[i];
      for (int k = i + 1; k < n; ++k) s -= sA[k * ld + i] * sb[k];
      sb[i] = s / sA[i * ld + i];
    }
    double pred = 0.0;
    for (int i = 0; i < n; ++i) {
      B.dc[i] = sb[i];
      const int cam = i / 6, a = i % 6, q = a * 6 - (a * (a - 1)) / 2;
      pred += 0.5 * sb[i] * (lam * B.U[21 * cam + q] * sb[i] - B.gc[i]);
    }
    B.scal[4] = pred;
  }
}

// The same solve by ONE wave (n <= 48: up to eight cameras): lane i holds row i of the matrix and entry i of the vectors in
// REGISTERS; column j of the factor reaches the other lanes through v_readlane with a constant lane index (a scalar
// broadcast: no LDS, no barrier), the whole factorisation is straight-line code.  (36 columns x 3 workgroup barriers and
// two single-thread substitutions made the 256-thread form 67 us; a one-wave form with the rows in LDS 58 us - a dependent
// LDS round trip per term of the trailing update.)
constexpr int CS_N = 48;
__device__ __forceinline__ double lane_bcast(double v, int lane_const) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane_const);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane_const);
  return __hiloint2double(hi, lo);
}
__global__ void __launch_bounds__(64) k_sba_cam_solve_wave(SbaBuf B, double lam) {
  __shared__ double sL[CS_N * (CS_N + 1)];
  const int n = 6 * B.C, i = threadIdx.x;
  const bool on = i < n;
  const int ic = on ? i : 0;
  // (every load unconditional on a clamped address, the selects afterwards: 48 conditional loads were 48 serialised round trips)
  double a[CS_N], ub[6];
  {
    const int cam = ic / 6, p = ic % 6;
#pragma unroll
    for (int q0 = 0; q0 < 6; ++q0) {
      const int lo = p < q0 ? p : q0, hi = p < q0 ? q0 : p;
      ub[q0] = B.U[21 * cam + lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo)];
    }
#pragma unroll
    for (int c = 0; c < CS_N; ++c) a[c] = B.S[ic * n + min(c, n - 1)];
#pragma unroll
    for (int c = 0; c < CS_N; ++c) {
      double v = a[c];
      const double u = ub[c % 6];
      if (c / 6 == cam) v += (c % 6 == p) ? u + lam * u + 1e-300 : u;
      a[c] = (on && c < n) ? v : (c == i ? 1.0 : 0.0);   // (identity on the padding: lanes >= n never touch the live part)
    }
  }
  double bi = on ? -(B.gc[ic] + B.rhs[ic]) : 0.0;
  bool bad = false;
#pragma unroll
  for (int j = 0; j < CS_N; ++j) {
    if (j < n) {
      double d = lane_bcast(a[j], j);
      if (!(d > 0.0)) {
        bad = true;
        d = 1.0;
      }
      const double dj = sqrt(d), idj = 1.0 / dj;
      const double lij = i == j ? dj : (i > j ? a[j] * idj : 0.0);
      a[j] = lij;
#pragma unroll
      for (int k = j + 1; k < CS_N; ++k) a[k] -= lij * lane_bcast(lij, k);     // (used for k <= i only; harmless elsewhere)
    }
  }
  // L y = b by columns (row i's L[i][j] is in register a[j]); then the factor to LDS and L^T x = y by rows of L
#pragma unroll
  for (int j = 0; j < CS_N; ++j)
    if (j < n) {
      const double yj = lane_bcast(bi, j) / lane_bcast(a[j], j);
      bi = i == j ? yj : (i > j ? bi - a[j] * yj : bi);
    }
#pragma unroll
  for (int c = 0; c < CS_N; ++c)
    if (i < CS_N) sL[i * (CS_N + 1) + c] = a[c];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = CS_N - 1; j >= 0; --j)
    if (j < n) {
      const double xj = lane_bcast(bi, j) / lane_bcast(a[j], j);
      const double lji = sL[j * (CS_N + 1) + (i < CS_N ? i : 0)];
      bi = i == j ? xj : (i < j ? bi - lji * xj : bi);
    }
  double pred = 0.0;
  if (on) {
    B.dc[i] = bi;
    const int cam = i / 6, p = i % 6, q = p * 6 - (p * (p - 1)) / 2;
    pred = 0.5 * bi * (lam * B.U[21 * cam + q] * bi - B.gc[i]);
  }
  for (int off = 32; off > 0; off >>= 1) pred += __shfl_down(pred, off, 64);
  if (i == 0) B.scal[4] = pred;
  if (bad && i == 0) B.scal[3] = 1.0;                    // (d is wave-uniform: every lane sees the same flag)
}

// dp = -Vinv (gp + sum_c W_pc^T dc); trial points; predicted reduction
__global__ void __launch_bounds__(256)
k_sba_backsub(SbaBuf B, double lam, const double* __restrict__ pts, double* __restrict__ pts_t) {
  __shared__ double sred[4];
  const int tid = threadIdx.x, p = blockIdx.x * 256 + tid;
  double pred = 0.0;
  if (p < B.P) {
    double s[3] = {B.gp[3 * (size_t)p], B.gp[3 * (size_t)p + 1], B.gp[3 * (size_t)p + 2]};
    const double g0 = s[0], g1 = s[1], g2 = s[2];
    if (B.opt_cams)
      for (int o = B.pt_start[p]; o < B.pt_start[p + 1]; ++o) {
        const int k = B.pt_obs[o], c = B.cam_idx[k];
        const double* W = B.Wpc + 18 * ((size_t)p * B.C + c);
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          const double dca = B.dc[6 * c + a];
          s[0] += W[a * 3] * dca;
          s[1] += W[a * 3 + 1] * dca;
          s[2] += W[a * 3 + 2] * dca;
        }
      }
    const double* Vi = B.Vinv + 6 * (size_t)p;
    const double d0 = -(Vi[0] * s[0] + Vi[1] * s[1] + Vi[2] * s[2]), d1 = -(Vi[1] * s[0] + Vi[3] * s[1] + Vi[4] * s[2]),
                 d2 = -(Vi[2] * s[0] + Vi[4] * s[1] + Vi[5] * s[2]);
    pts_t[3 * (size_t)p] = pts[3 * (size_t)p] + d0;
    pts_t[3 * (size_t)p + 1] = pts[3 * (size_t)p + 1] + d1;
    pts_t[3 * (size_t)p + 2] = pts[3 * (size_t)p + 2] + d2;
    const double* V = B.V + 6 * (size_t)p;
    pred = 0.5 * (d0 * (lam * V[0] * d0 - g0) + d1 * (lam * V[3] * d1 - g1) + d2 * (lam * V[5] * d2 - g2));
  }
  for (int off = 32; off > 0; off >>= 1) pred += __shfl_down(pred, off, 64);
  if ((tid & 63) == 0) sred[tid >> 6] = pred;
  __syncthreads();
  if (tid == 0) atomicAdd(&B.scal[1], (sred[0] + sred[1]) + (sred[2] + sred[3]));
}

// R_t = exp([dw]x) R, t_t = t + dt
__global__ void k_sba_apply_cams(SbaBuf B, const double* __restrict__ Rt, double* __restrict__ Rt_t) {
  const int c = threadIdx.x;
  if (c >= B.C) return;
  const double* R = Rt + 12 * c;
  double* Ro = Rt_t + 12 * c;
  double w[3] = {0, 0, 0}, dt[3] = {0, 0, 0};
  if (B.opt_cams) {
    for (int j = 0; j < 3; ++j) {
      w[j] = B.dc[6 * c + j];
      dt[j] = B.dc[6 * c + 3 + j];
    }
  }
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
  double A, Bc;   // exp([w]x) = I + A [w]x + Bc [w]x^2
  if (th < 1e-8) {
    A = 1.0 - th2 / 6.0;
    Bc = 0.5 - th2 / 24.0;
  } else {
    A = sin(th) / th;
    Bc = (1.0 - cos(th)) / th2;
  }
  const double Kx[3][3] = {{0, -w[2], w[1]}, {w[2], 0, -w[0]}, {-w[1], w[0], 0}};
  double E[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double k2 = Kx[i][0] * Kx[0][j] + Kx[i][1] * Kx[1][j] + Kx[i][2] * Kx[2][j];
      E[i][j] = (i == j ? 1.0 : 0.0) + A * Kx[i][j] + Bc * k2;
    }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Ro[3 * i + j] = E[i][0] * R[j] + E[i][1] * R[3 + j] + E[i][2] * R[6 + j];
  for (int j = 0; j < 3; ++j) Ro[9 + j] = R[9 + j] + dt[j];
}

static size_t a256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace acino

using namespace acino;

extern "C" {

size_t acino_sizeof_sba_params(void) { return sizeof(acino_sba_params); }
size_t acino_sizeof_sba_info(void) { return sizeof(acino_sba_info); }

// the fused path: at most seven cameras (6 C + 1 rows fit three 16-row tiles); ACINO_SBA_UNFUSED=1 keeps the table form
static bool sba_fused(int n_cams) {
  static const bool unfused = getenv("ACINO_SBA_UNFUSED") != nullptr;
  return 6 * n_cams + 1 <= 48 && !unfused;
}

size_t acino_sba_workspace_bytes(int n_cams, int64_t n_points, int64_t n_obs) {
  if (n_cams < 1 || n_points < 0 || n_obs < 0) return 0;
  const size_t P = (size_t)n_points, M = (size_t)n_obs, n = 6 * (size_t)n_cams;
  size_t b = 0;
  (void)M;
  b += a256(P * 6 * 8) * 2 + a256(P * 3 * 8) * 3;                                      // V, Vinv, gp, dp, pts_t
  b += sba_fused(n_cams) ? a256(P * n_cams * 4) : a256(P * n_cams * 18 * 8);           // slot [P][C]  |  Wpc [P][C][18]
  b += a256((size_t)(SBA_SCHUR_WG + 32) * (n * n + n + 27 * (size_t)n_cams) * 8);      // partial sums (+ 32 intermediate records)
  b += a256((size_t)SBA_SCHUR_WG * 4 * 8);                                               // cost / prediction / trial-cost partials
  b += a256(n_cams * 21 * 8) + a256(n * 8) * 3 + a256(n * n * 8) + a256(n_cams * 12 * 8) + a256(64);
  return b + 1024;
}

// Solves in place: d_Rt[C][12] (R row-major | t) and d_pts[P][3].  d_res_before / d_res_after [M][2] may be NULL.
int acino_sba_solve(const acino_sba_params* prm, const double* d_intr, double* d_Rt, double* d_pts,
                    const double* d_uv, const int32_t* d_cam_idx, const int32_t* d_pt_start, const int32_t* d_pt_obs,
                    void* d_ws, size_t ws_bytes, double* d_res_before, double* d_res_after, acino_sba_info* info,
                    void* stream) {
  return acino_sba_solve_sharded(prm, d_intr, d_Rt, d_pts, d_uv, d_cam_idx, d_pt_start, d_pt_obs, d_ws, ws_bytes,
                                 d_res_before, d_res_after, info, nullptr, nullptr, stream);
}

int acino_sba_solve_sharded(const acino_sba_params* prm, const double* d_intr, double* d_Rt, double* d_pts,
                            const double* d_uv, const int32_t* d_cam_idx, const int32_t* d_pt_start,
                            const int32_t* d_pt_obs, void* d_ws, size_t ws_bytes, double* d_res_before,
                            double* d_res_after, acino_sba_info* info, acino_reduce_fn reduce, void* reduce_user,
                            void* stream) {
  ACINO_REQUIRE(prm && info, "params/info");
  ACINO_REQUIRE(prm->n_cams >= 1 && prm->n_cams <= SBA_MAXC, "n_cams in 1..16");
  ACINO_REQUIRE(prm->n_points >= 1 && prm->n_obs >= 1, "sizes (a rank without points cannot take part)");
  ACINO_REQUIRE(prm->f_scale > 0 && prm->lam0 > 0 && prm->max_iter >= 0, "f_scale, lam0, max_iter");
  ACINO_REQUIRE(prm->camera_model == 0 || prm->camera_model == 1, "camera_model: 0 fisheye, 1 pinhole");
  ACINO_REQUIRE(prm->precision == ACINO_PREC_F64 || prm->precision == ACINO_PREC_BF16_ROWS, "precision: 0 f64, 1 bf16 rows");
  ACINO_REQUIRE(d_intr && d_Rt && d_pts && d_uv && d_cam_idx && d_pt_start && d_pt_obs && d_ws, "null buffer");
  ACINO_REQUIRE(((uintptr_t)d_ws & 255) == 0, "workspace must be 256-byte aligned");
  ACINO_REQUIRE(ws_bytes >= acino_sba_workspace_bytes(prm->n_cams, prm->n_points, prm->n_obs), "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int C = prm->n_cams;
  const size_t P = (size_t)prm->n_points, M = (size_t)prm->n_obs, n = 6 * (size_t)C;
  char* w = (char*)d_ws;
  auto take = [&](size_t bytes) {
    char* p = w;
    w += a256(bytes);
    return p;
  };
  SbaBuf B;
  B.C = C;
  B.P = (int)P;
  B.M = (int)M;
  B.opt_cams = prm->optimize_cameras ? 1 : 0;
  B.model = prm->camera_model;
  B.prec = prm->precision;
  B.fs = prm->f_scale;
  B.intr = d_intr;
  B.uv = d_uv;
  B.cam_idx = d_cam_idx;
  B.pt_start = d_pt_start;
  B.pt_obs = d_pt_obs;
  B.V = (double*)take(P * 6 * 8);
  B.Vinv = (double*)take(P * 6 * 8);
  B.gp = (double*)take(P * 3 * 8);
  B.dp = (double*)take(P * 3 * 8);
  double* pts_t = (double*)take(P * 3 * 8);
  const bool fused = sba_fused(C);
  B.Wpc = nullptr;
  B.slot = nullptr;
  if (fused) B.slot = (int*)take(P * C * 4);
  else B.Wpc = (double*)take(P * C * 18 * 8);
  B.Spart = (double*)take((size_t)(SBA_SCHUR_WG + 32) * (n * n + n + 27 * (size_t)C) * 8);
  B.part3 = (double*)take((size_t)SBA_SCHUR_WG * 4 * 8);
  // (the dense W table: slots of cameras that do not see a point are never written - zero them once)
  if (!fused && B.opt_cams) ACINO_HIP_CHECK(hipMemsetAsync(B.Wpc, 0, P * C * 18 * 8, s));
  B.U = (double*)take((C * 21 + n) * 8);      // [U | gc] contiguous: one reduction
  B.gc = B.U + C * 21;
  B.dc = (double*)take(n * 8);
  B.S = (double*)take((n * n + n) * 8);        // [S | rhs] contiguous: one reduction
  B.rhs = B.S + n * n;
  double* Rt_t = (double*)take(C * 12 * 8);
  B.scal = (double*)take(64);                  // 0 cost, 1 predicted reduction (points), 2 max |g_point|, 3 not-PD flag,
                                               // 4 predicted reduction (cameras; identical on every rank)
  // global sums / maxima over the ranks that share the cameras (no-op for a single process)
  auto greduce = [&](double* d_buf, size_t cnt, int op) -> int {
    if (!reduce) return ACINO_OK;
    ACINO_HIP_CHECK(hipStreamSynchronize(s));
    if (reduce(reduce_user, d_buf, (int64_t)cnt, op, stream) != 0) {
      set_error("SBA: the reduction callback failed");
      return ACINO_ERR_CALLBACK;
    }
    return ACINO_OK;
  };
  const int nblk = (int)((P + 255) / 256);
  const size_t lds_s = (n * n + n) * 8, lds_c = (n * (n + 1) + n) * 8;

  auto eval = [&](const double* Rt, const double* pts, bool jac, double* res, double h[4]) -> int {
    ACINO_HIP_CHECK(hipMemsetAsync(B.scal, 0, 64, s));
    if (jac) {
      ACINO_HIP_CHECK(hipMemsetAsync(B.U, 0, C * 21 * 8, s));
      ACINO_HIP_CHECK(hipMemsetAsync(B.gc, 0, n * 8, s));
      if (B.prec == ACINO_PREC_F64) hipLaunchKernelGGL((k_sba_point<true, ACINO_PREC_F64>), dim3(nblk), dim3(256), 16 * C * 27 * 8, s, B, Rt, pts, res);
      else hipLaunchKernelGGL((k_sba_point<true, ACINO_PREC_BF16_ROWS>), dim3(nblk), dim3(256), 16 * C * 27 * 4, s, B, Rt, pts, res);
    } else {
      hipLaunchKernelGGL((k_sba_point<false, ACINO_PREC_F64>), dim3(nblk), dim3(256), 0, s, B, Rt, pts, res);   // (cost only: fp64)
    }
    ACINO_LAUNCH_CHECK();
    if (int e = greduce(B.scal, 1, 0)) return e;
    if (jac) {
      if (int e = greduce(B.scal + 2, 1, 1)) return e;
      if (B.opt_cams)
        if (int e = greduce(B.U, C * 21 + n, 0)) return e;
    }
    ACINO_HIP_CHECK(hipMemcpyAsync(h, B.scal, 32, hipMemcpyDeviceToHost, s));
    ACINO_HIP_CHECK(hipStreamSynchronize(s));
    return ACINO_OK;
  };

  {
    // ---- input check, both paths; the flag is combined over the ranks (max) BEFORE anyone returns: a rank that left early
    //      would leave the others waiting in the first reduction of the solve
    ACINO_HIP_CHECK(hipMemsetAsync(B.scal, 0, 64, s));
    hipLaunchKernelGGL(k_sba_check, dim3(nblk), dim3(256), 0, s, B);
    ACINO_LAUNCH_CHECK();
    if (int e = greduce(B.scal + 7, 1, 1)) return e;
    double dup = 0.0;
    ACINO_HIP_CHECK(hipMemcpyAsync(&dup, B.scal + 7, 8, hipMemcpyDeviceToHost, s));
    ACINO_HIP_CHECK(hipStreamSynchronize(s));
    ACINO_REQUIRE(dup == 0.0, "SBA: two observations of one point by one camera, or a camera index out of range");
  }
  if (fused) {
    // ---- fused path: per LM iteration ONE pass that linearises and reduces (k_sba_fused + the two-stage sum), the camera
    //      solve, the trial poses, ONE pass that back-substitutes and prices the trial iterate; one host read-back.
    ACINO_HIP_CHECK(hipMemsetAsync(B.scal, 0, 64, s));
    ACINO_HIP_CHECK(hipMemsetAsync(B.slot, 0xFF, P * C * 4, s));
    hipLaunchKernelGGL(k_sba_slots, dim3(nblk), dim3(256), 0, s, B);
    ACINO_LAUNCH_CHECK();
    double hd[4];
    int rc = ACINO_OK;
    if (d_res_before) {
      rc = eval(d_Rt, d_pts, false, d_res_before, hd);
      if (rc) return rc;
    }
    const int ppp = 64 / C;
    const size_t batches = (P + ppp - 1) / ppp;
    const int waves = (int)std::min<size_t>((size_t)SBA_SCHUR_WG * 4, batches);
    const int ppw = (int)((batches + waves - 1) / waves) * ppp, n_wg = (int)((P + (size_t)4 * ppw - 1) / ((size_t)4 * ppw));
    const size_t lds_f = fused_lds_bytes(C);
    const int tail = 27 * C;
    const unsigned rblk = (unsigned)((n * n + n + tail + 255) / 256);
    auto linearise = [&](double lam, bool with_cost) -> int {
      ACINO_HIP_CHECK(hipMemsetAsync(B.scal, 0, 64, s));
#define ACINO_SBA_FUSED(PRECV, MODELV)                                                                                            \
  do {                                                                                                                            \
    if (with_cost) hipLaunchKernelGGL((k_sba_fused<PRECV, MODELV, true>), dim3(n_wg), dim3(FU_T), lds_f, s, B, d_Rt, d_pts, lam, ppw); \
    else hipLaunchKernelGGL((k_sba_fused<PRECV, MODELV, false>), dim3(n_wg), dim3(FU_T), lds_f, s, B, d_Rt, d_pts, lam, ppw);     \
  } while (0)
      if (B.prec == ACINO_PREC_F64) {
        if (B.model == 0) ACINO_SBA_FUSED(ACINO_PREC_F64, 0);
        else ACINO_SBA_FUSED(ACINO_PREC_F64, 1);
      } else {
        if (B.model == 0) ACINO_SBA_FUSED(ACINO_PREC_BF16_ROWS, 0);
        else ACINO_SBA_FUSED(ACINO_PREC_BF16_ROWS, 1);
      }
#undef ACINO_SBA_FUSED
      ACINO_LAUNCH_CHECK();
      if (with_cost) {
        hipLaunchKernelGGL(k_sba_sum_parts, dim3(1), dim3(256), 0, s, B, n_wg, 0, 0, -1, 0);
        ACINO_LAUNCH_CHECK();
      }
      if (B.opt_cams) {
        hipLaunchKernelGGL(k_sba_schur_reduce, dim3(rblk, 32), dim3(256), 0, s, B, n_wg, 0, tail);
        ACINO_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_sba_schur_reduce, dim3(rblk, 1), dim3(256), 0, s, B, n_wg, 1, tail);
        ACINO_LAUNCH_CHECK();
      }
      if (with_cost)
        if (int e = greduce(B.scal, 1, 0)) return e;
      if (int e = greduce(B.scal + 2, 1, 1)) return e;
      if (B.opt_cams) {
        if (int e = greduce(B.U, C * 21 + n, 0)) return e;
        if (int e = greduce(B.S, n * n + n, 0)) return e;
      }
      return ACINO_OK;
    };
    auto read_back = [&](double hs[8], double& gmax) -> int {
      double hgc[6 * SBA_MAXC];
      ACINO_HIP_CHECK(hipMemcpyAsync(hs, B.scal, 64, hipMemcpyDeviceToHost, s));
      if (B.opt_cams) ACINO_HIP_CHECK(hipMemcpyAsync(hgc, B.gc, n * 8, hipMemcpyDeviceToHost, s));
      ACINO_HIP_CHECK(hipStreamSynchronize(s));
      gmax = hs[2];
      if (B.opt_cams)
        for (size_t i = 0; i < n; ++i) gmax = fmax(gmax, fabs(hgc[i]));
      return ACINO_OK;
    };
    double F = 0.0, lam = prm->lam0, nu = 2.0, gmax = 0.0, hs[8];
    bool fresh = false;                                   // hs / gmax describe the CURRENT iterate
    info->iterations = 0;
    info->accepted = 0;
    info->status = 0;
    for (int it = 0; it < prm->max_iter; ++it) {
      if ((rc = linearise(lam, it == 0))) return rc;
      if (B.opt_cams) {
        if (n <= (size_t)CS_N) hipLaunchKernelGGL(k_sba_cam_solve_wave, dim3(1), dim3(64), 0, s, B, lam);
        else hipLaunchKernelGGL(k_sba_cam_solve, dim3(1), dim3(256), lds_c, s, B, lam);
        ACINO_LAUNCH_CHECK();
      }
      hipLaunchKernelGGL(k_sba_apply_cams, dim3(1), dim3(64), 0, s, B, d_Rt, Rt_t);
      ACINO_LAUNCH_CHECK();
#define ACINO_SBA_BACK(PRECV, MODELV) \
  hipLaunchKernelGGL((k_sba_backsub_fused<PRECV, MODELV>), dim3(n_wg), dim3(FU_T), 0, s, B, lam, d_Rt, Rt_t, d_pts, pts_t, ppw)
      if (B.prec == ACINO_PREC_F64) {
        if (B.model == 0) ACINO_SBA_BACK(ACINO_PREC_F64, 0);
        else ACINO_SBA_BACK(ACINO_PREC_F64, 1);
      } else {
        if (B.model == 0) ACINO_SBA_BACK(ACINO_PREC_BF16_ROWS, 0);
        else ACINO_SBA_BACK(ACINO_PREC_BF16_ROWS, 1);
      }
#undef ACINO_SBA_BACK
      ACINO_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_sba_sum_parts, dim3(1), dim3(256), 0, s, B, n_wg, 1, 1, 2, 6);
      ACINO_LAUNCH_CHECK();
      if (int e = greduce(B.scal + 1, 1, 0)) return e;
      if (int e = greduce(B.scal + 6, 1, 0)) return e;
      if ((rc = read_back(hs, gmax))) return rc;
      fresh = true;
      if (it == 0) {
        F = hs[0];
        info->cost_initial = F;
      }
      if (gmax <= prm->gtol) {
        info->status = 3;
        break;
      }
      info->iterations = it + 1;
      const double pred = hs[1] + hs[4];
      // (hs[3] != 0: the damped reduced camera system lost definiteness to round-off along the free gauge - 7 DoF when every
      //  camera moves -: a rejected step)
      const double Ft = hs[3] == 0.0 ? hs[6] : INFINITY;
      const double gain = pred > 0 ? (F - Ft) / pred : -1.0;
      if (Ft < F) {
        const double dF = F - Ft;
        ACINO_HIP_CHECK(hipMemcpyAsync(d_pts, pts_t, P * 3 * 8, hipMemcpyDeviceToDevice, s));
        ACINO_HIP_CHECK(hipMemcpyAsync(d_Rt, Rt_t, C * 12 * 8, hipMemcpyDeviceToDevice, s));
        F = Ft;
        fresh = false;
        info->accepted += 1;
        const double t = 2.0 * gain - 1.0;
        lam *= fmax(1.0 / 3.0, 1.0 - t * t * t);
        nu = 2.0;
        if (dF <= prm->ftol * fabs(F)) {
          info->status = 1;
          break;
        }
      } else {
        lam *= nu;
        nu *= 2.0;
        if (lam > 1e16) {
          info->status = hs[3] != 0.0 ? 5 : 4;
          if (info->status == 5) set_error("SBA: reduced camera system not positive definite at any damping");
          break;
        }
      }
    }
    if (!fresh) {                                          // gradient norm (and, with max_iter = 0, the cost) of the final iterate
      const bool first = prm->max_iter == 0;
      if ((rc = linearise(lam, first))) return rc;
      if ((rc = read_back(hs, gmax))) return rc;
      if (first) {
        F = hs[0];
        info->cost_initial = F;
      }
    }
    if (d_res_after) {
      rc = eval(d_Rt, d_pts, false, d_res_after, hd);
      if (rc) return rc;
    }
    info->cost_final = F;
    info->gnorm_inf = gmax;
    info->lam = lam;
    return info->status == 5 ? ACINO_ERR_NUMERIC : ACINO_OK;
  }

  double h[4];
  int rc = eval(d_Rt, d_pts, true, d_res_before, h);
  if (rc) return rc;
  double F = h[0], lam = prm->lam0, nu = 2.0;
  info->cost_initial = F;
  info->iterations = 0;
  info->accepted = 0;
  info->status = 0;
  double gmax = h[2];
  if (B.opt_cams) {
    double hgc[6 * SBA_MAXC];
    ACINO_HIP_CHECK(hipMemcpyAsync(hgc, B.gc, n * 8, hipMemcpyDeviceToHost, s));
    ACINO_HIP_CHECK(hipStreamSynchronize(s));
    for (size_t i = 0; i < n; ++i) gmax = fmax(gmax, fabs(hgc[i]));
  }
  for (int it = 0; it < prm->max_iter; ++it) {
    if (gmax <= prm->gtol) {
      info->status = 3;
      break;
    }
    info->iterations = it + 1;
    ACINO_HIP_CHECK(hipMemsetAsync(B.scal, 0, 64, s));
    if (B.opt_cams) ACINO_HIP_CHECK(hipMemsetAsync(B.S, 0, (n * n + n) * 8, s));
    hipLaunchKernelGGL(k_sba_schur, dim3(nblk), dim3(256), lds_s, s, B, lam);
    ACINO_LAUNCH_CHECK();
    if (B.opt_cams) {
      if (int e = greduce(B.S, n * n + n, 0)) return e;
      hipLaunchKernelGGL(k_sba_cam_solve, dim3(1), dim3(256), lds_c, s, B, lam);
      ACINO_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_sba_backsub, dim3(nblk), dim3(256), 0, s, B, lam, d_pts, pts_t);
    ACINO_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sba_apply_cams, dim3(1), dim3(64), 0, s, B, d_Rt, Rt_t);
    ACINO_LAUNCH_CHECK();
    if (int e = greduce(B.scal + 1, 1, 0)) return e;
    double hp[5];
    ACINO_HIP_CHECK(hipMemcpyAsync(hp, B.scal, 40, hipMemcpyDeviceToHost, s));
    ACINO_HIP_CHECK(hipStreamSynchronize(s));
    const double pred = hp[1] + hp[4];
    double ht[4] = {INFINITY, 0, 0, 0};
    if (hp[3] == 0.0) {   // else: the damped reduced camera system lost definiteness to round-off along the free
      rc = eval(Rt_t, pts_t, false, nullptr, ht);   // gauge (7 DoF when every camera moves) - a rejected step
      if (rc) return rc;
    }
    const double Ft = ht[0];
    const double gain = pred > 0 ? (F - Ft) / pred : -1.0;
    if (Ft < F) {
      const double dF = F - Ft;
      ACINO_HIP_CHECK(hipMemcpyAsync(d_pts, pts_t, P * 3 * 8, hipMemcpyDeviceToDevice, s));
      ACINO_HIP_CHECK(hipMemcpyAsync(d_Rt, Rt_t, C * 12 * 8, hipMemcpyDeviceToDevice, s));
      rc = eval(d_Rt, d_pts, true, nullptr, h);
      if (rc) return rc;
      F = h[0];
      gmax = h[2];
      if (B.opt_cams) {
        double hgc[6 * SBA_MAXC];
        ACINO_HIP_CHECK(hipMemcpyAsync(hgc, B.gc, n * 8, hipMemcpyDeviceToHost, s));
        ACINO_HIP_CHECK(hipStreamSynchronize(s));
        for (size_t i = 0; i < n; ++i) gmax = fmax(gmax, fabs(hgc[i]));
      }
      info->accepted += 1;
      const double t = 2.0 * gain - 1.0;
      lam *= fmax(1.0 / 3.0, 1.0 - t * t * t);
      nu = 2.0;
      if (dF <= prm->ftol * fabs(F)) {
        info->status = 1;
        break;
      }
    } else {
      lam *= nu;
      nu *= 2.0;
      if (lam > 1e16) {
        info->status = hp[3] != 0.0 ? 5 : 4;
        if (info->status == 5) set_error("SBA: reduced camera system not positive definite at any damping");
        break;
      }
    }
  }
  if (d_res_after) {
    rc = eval(d_Rt, d_pts, false, d_res_after, h);
    if (rc) return rc;
  }
  info->cost_final = F;
  info->gnorm_inf = gmax;
  info->lam = lam;
  return info->status == 5 ? ACINO_ERR_NUMERIC : ACINO_OK;
}

}  // extern "C"
