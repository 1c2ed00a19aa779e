// Sparse bundle adjustment of 3-D points (+ optionally the 6-DoF camera extrinsics) on fisheye cameras.
//
// Reference: src/calib/calib.py:307-341 (points only), :345-390 (points + extrinsics) - scipy least_squares(method='trf', loss='cauchy') over
// [rvecs, tvecs, points] with a finite-difference Jacobian and one cv2.fisheye.projectPoints call per
// observation.  Here: analytic Jacobians, Cauchy IRLS weights w = 1/(1 + (r/f)^2), Levenberg-Marquardt on the
// Schur complement of the point blocks (3x3 per point) onto the camera block (6C x 6C), rotations updated by a
// left perturbation R <- exp([dw]x) R (same minimiser, no rvec singularities).  One thread per POINT walks its
// observations (CSR built by the host); camera-block sums go through 16 lane-private copies in LDS (the lanes of a wave
// are all at the same camera: one shared copy serialises 64 atomics per entry), then one atomic per entry per workgroup.
// The coupling blocks W_pc (6 x 3) live in a DENSE table [point][camera] (zero where the camera does not see the point), so
// the Schur complement S = - sum_p (W_p V_p^-1) W_p^T is a tall-skinny GEMM: k_sba_schur_mfma forms it on the fp64 matrix
// cores, one point per k-step (3 columns + 1 zero), operands built in registers - no LDS, no atomics, partial sums per
// workgroup reduced in a fixed order (round 4: 4.47 -> 0.3 ms per iteration at 1.28 M points; the previous form added 36
// entries per observation pair to one LDS copy with atomics).
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
  const double iz = 1.0 / Xc[2];
  const double a = Xc[0] * iz, b = Xc[1] * iz;
  const double r = sqrt(a * a + b * b);
  const double th = atan(r), th2 = th * th;
  const double thd = th * (1 + th2 * (c.d[0] + th2 * (c.d[1] + th2 * (c.d[2] + th2 * c.d[3]))));
  const bool small = !(r > 1e-8);
  const double m = small ? 1.0 : thd / r;
  uv[0] = c.fx * a * m + c.cx;
  uv[1] = c.fy * b * m + c.cy;
  if (JAC) {
    double dm_da = 0.0, dm_db = 0.0;
    if (!small) {
      const double dthd = 1 + th2 * (3 * c.d[0] + th2 * (5 * c.d[1] + th2 * (7 * c.d[2] + th2 * 9 * c.d[3])));
      const double dm_dr = (dthd / (1 + r * r) * r - thd) / (r * r);
      dm_da = dm_dr * a / r;
      dm_db = dm_dr * b / r;
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
  double* Spart;         // [n_schur_wg][n n + n] per-workgroup partial sums of k_sba_schur_mfma (null: atomics path)
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


// The Schur complement on the matrix cores (6 C + 1 <= 48, i.e. C <= 7 cameras).  With A_p = [W_p V_p^-1 ; (V_p^-1 g_p)^T]
// (37 x 3 for six cameras: rows 6 c + a, then the right-hand-side row) and B_p = W_p (36 x 3, zero rows for cameras that do
// not see the point):  [S ; rhs^T] = - sum_p A_p B_p^T.  One point is one k-step of v_mfma_f64_16x16x4 (k = 0..2 the three
// point coordinates, k = 3 zero); lane (i, k) builds its operand entries from the rows 16 t + i (t = 0, 1, 2) of the dense
// W table and the point's V^-1 (every lane inverts the same 3 x 3: uniform loads), the six lower tiles accumulate in
// registers over the wave's points.  Points are dealt to the waves in contiguous ranges, the next point's values are
// requested before the current point's products.  Partial sums: wave -> LDS -> one [n n + n] record per workgroup; the
// reduction kernel adds the records in a fixed order (deterministic, no atomics).
constexpr int SBA_SCHUR_WG = 1024;      // workgroups (x 4 waves) of k_sba_schur_mfma = records of partial sums
constexpr int SCH_T = 256, SCH_B = 16;     // threads; points per batch of one wave
__global__ void __launch_bounds__(SCH_T) k_sba_schur_mfma(SbaBuf B, double lam, int pts_per_wave) {
  // Each wave streams its points through its own LDS slab in batches of SCH_B: the batch's V (6), g (3) and W (18 C) values
  // are contiguous in memory - copied with all 64 lanes, the NEXT batch's copy in flight (registers) while this batch's
  // points go through the matrix cores.  (One point at a time from memory was latency-bound: 1.03 ms for 1.28 M points.)
  extern __shared__ __attribute__((aligned(16))) char sch_smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int C = B.C, n = 6 * C, wrec = 18 * C, rec = 9 + wrec;            // doubles per point in the slab: V | g | W
  double* slab = reinterpret_cast<double*>(sch_smem) + (size_t)wave * SCH_B * rec;
  const int gw = blockIdx.x * 4 + wave;
  const int p0 = gw * pts_per_wave, p1 = min(p0 + pts_per_wave, B.P);
  d4 acc[6];
#pragma unroll
  for (int t = 0; t < 6; ++t) acc[t] = d4{0, 0, 0, 0};
  int woff[3], rrow[3];                                                    // rows 16 t + li of [W V^-1 ; (V^-1 g)^T]
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int R = 16 * t + li;
    woff[t] = R < n ? 3 * R : -1;                                          // (camera R / 6, parameter R % 6: offset 18 c + 3 r = 3 R)
    rrow[t] = R < n ? 0 : (R == n ? 1 : 2);                                // 0 coupling row, 1 right-hand-side row, 2 padding
  }
  // one batch in registers: V (6 nb <= 96 values: 2 per lane), g (3 nb <= 48: 1), W (18 C values per point: 2 per lane and
  // point for C <= 7).  Every copy is a run of consecutive addresses; no division by a run-time value anywhere (the first
  // form spent ~200 instructions per point on e / wrec, e % wrec).
  double sv[2], sg, sw[SCH_B][2];
  // (every load UNCONDITIONAL on a clamped, valid address: a conditional load compiles to a branch and a full wait per
  //  element - 35 serialised round trips per batch; what is out of range is simply not stashed)
  auto fetch = [&](int pb, int nb) {
#pragma unroll
    for (int q = 0; q < 2; ++q) sv[q] = B.V[6 * (size_t)pb + min(lane + 64 * q, 6 * nb - 1)];
    sg = B.gp[3 * (size_t)pb + min(lane, 3 * nb - 1)];
#pragma unroll
    for (int pt = 0; pt < SCH_B; ++pt)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        sw[pt][q] = B.Wpc[(size_t)wrec * (pb + min(pt, nb - 1)) + min(lane + 64 * q, wrec - 1)];
  };
  auto stash = [&](int nb) {                                               // registers -> slab, point-major records
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = lane + 64 * q;
      if (e < 6 * nb) slab[(e / 6) * rec + e % 6] = sv[q];
    }
    if (lane < 3 * nb) slab[(lane / 3) * rec + 6 + lane % 3] = sg;
#pragma unroll
    for (int pt = 0; pt < SCH_B; ++pt)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int e = lane + 64 * q;
        if (pt < nb && e < wrec) slab[pt * rec + 9 + e] = sw[pt][q];
      }
  };
  int pb = p0;
  if (pb < p1) fetch(pb, min(SCH_B, p1 - pb));
  while (pb < p1) {
    const int nb = min(SCH_B, p1 - pb);
    stash(nb);                                                             // (a wave's LDS operations are ordered: no barrier)
    if (pb + nb < p1) fetch(pb + nb, min(SCH_B, p1 - pb - nb));
    for (int q = 0; q < nb; ++q) {
      const double* r = slab + q * rec;
      // V^-1 of the damped point block (closed form, as k_sba_schur), V^-1 g
      const double a = r[0] * (1 + lam), b = r[1], c = r[2], d = r[3] * (1 + lam), e = r[4], f = r[5] * (1 + lam);
      const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
      double det = a * c00 + b * c01 + c * c02;
      if (!(fabs(det) > 0.0)) det = 1.0;
      const double id = 1.0 / det;
      const double Vi[6] = {c00 * id, c01 * id, c02 * id, (a * f - c * c) * id, (b * c - a * e) * id, (a * d - b * b) * id};
      if (lane < 6) B.Vinv[6 * (size_t)(pb + q) + lane] = lane == 0 ? Vi[0] : (lane == 1 ? Vi[1] : (lane == 2 ? Vi[2] : (lane == 3 ? Vi[3] : (lane == 4 ? Vi[4] : Vi[5]))));
      // column lk of V^-1 (lk = 3: the zero k-step), entry lk of V^-1 g
      const double v0 = lk == 0 ? Vi[0] : (lk == 1 ? Vi[1] : (lk == 2 ? Vi[2] : 0.0));
      const double v1 = lk == 0 ? Vi[1] : (lk == 1 ? Vi[3] : (lk == 2 ? Vi[4] : 0.0));
      const double v2 = lk == 0 ? Vi[2] : (lk == 1 ? Vi[4] : (lk == 2 ? Vi[5] : 0.0));
      const double vg = v0 * r[6] + v1 * r[7] + v2 * r[8];
      double Aop[3], Bop[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const double* w = r + 9 + (woff[t] >= 0 ? woff[t] : 0);
        const double w0 = woff[t] >= 0 ? w[0] : 0.0, w1 = woff[t] >= 0 ? w[1] : 0.0, w2 = woff[t] >= 0 ? w[2] : 0.0;
        Aop[t] = rrow[t] == 1 ? vg : w0 * v0 + w1 * v1 + w2 * v2;
        Bop[t] = lk == 0 ? w0 : (lk == 1 ? w1 : (lk == 2 ? w2 : 0.0));
      }
      acc[0] = mfma(-Aop[0], Bop[0], acc[0]);
      acc[1] = mfma(-Aop[1], Bop[0], acc[1]);
      acc[2] = mfma(-Aop[1], Bop[1], acc[2]);
      acc[3] = mfma(-Aop[2], Bop[0], acc[3]);
      acc[4] = mfma(-Aop[2], Bop[1], acc[4]);
      acc[5] = mfma(-Aop[2], Bop[2], acc[5]);
    }
    pb += nb;
  }
  // wave -> LDS -> one record per workgroup.  C layout: register q of lane (li, lk) = entry [row lk + 4 q][column li] of a tile
  __syncthreads();                                                         // (the slabs are free)
  double* sT = reinterpret_cast<double*>(sch_smem);                        // [4][6][256]
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) sT[(wave * 6 + t) * 256 + (lk + 4 * q) * 16 + li] = acc[t][q];
  __syncthreads();
  double* out = B.Spart + (size_t)blockIdx.x * (n * n + n);
  for (int e = tid; e < 6 * 256; e += SCH_T) {
    const int t = e >> 8, r = (e >> 4) & 15, cc = e & 15;
    const int ib = t < 1 ? 0 : (t < 3 ? 1 : 2), jb = t < 1 ? 0 : (t < 3 ? t - 1 : t - 3);
    const int R = 16 * ib + r, Cc = 16 * jb + cc;
    const double v = (sT[(0 * 6 + t) * 256 + (e & 255)] + sT[(1 * 6 + t) * 256 + (e & 255)]) +
                     (sT[(2 * 6 + t) * 256 + (e & 255)] + sT[(3 * 6 + t) * 256 + (e & 255)]);
    if (Cc >= n) continue;
    if (R < n) {
      out[R * n + Cc] = v;
      if (ib != jb) out[Cc * n + R] = v;                 // (diagonal tiles hold both triangles already)
    } else if (R == n) {
      out[n * n + Cc] = v;
    }
  }
}
// the records, added in a fixed order in two stages (n_part <= 1024 records of n n + n doubles): stage 0 sums every 32nd record
// into 32 intermediate records (behind the partial records in Spart), stage 1 sums those into [S | rhs]
__global__ void __launch_bounds__(256) k_sba_schur_reduce(SbaBuf B, int n_part, int stage) {
  const int n = 6 * B.C, tot = n * n + n;
  const int e = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
  if (e >= tot) return;
  double* mid = B.Spart + (size_t)SBA_SCHUR_WG * tot;
  double s = 0.0;
  if (stage == 0) {
    for (int w = g; w < n_part; w += 32) s += B.Spart[(size_t)w * tot + e];
    mid[(size_t)g * tot + e] = s;
  } else {
    for (int w = 0; w < 32; ++w) s += mid[(size_t)w * tot + e];
    B.S[e] = s;                                          // (S | rhs contiguous)
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

size_t acino_sba_workspace_bytes(int n_cams, int64_t n_points, int64_t n_obs) {
  if (n_cams < 1 || n_points < 0 || n_obs < 0) return 0;
  const size_t P = (size_t)n_points, M = (size_t)n_obs, n = 6 * (size_t)n_cams;
  size_t b = 0;
  (void)M;
  b += a256(P * 6 * 8) * 2 + a256(P * 3 * 8) * 3 + a256(P * n_cams * 18 * 8);        // V, Vinv, gp, dp, pts_t, Wpc [P][C]
  b += a256((size_t)(SBA_SCHUR_WG + 32) * (n * n + n) * 8);                           // partial sums of the Schur kernel (+ 32 intermediate records)
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
  B.Wpc = (double*)take(P * C * 18 * 8);
  B.Spart = (double*)take((size_t)(SBA_SCHUR_WG + 32) * (n * n + n) * 8);
  // (the dense W table: slots of cameras that do not see a point are never written - zero them once)
  if (B.opt_cams) ACINO_HIP_CHECK(hipMemsetAsync(B.Wpc, 0, P * C * 18 * 8, s));
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
    static const bool schur_atomics = getenv("ACINO_SBA_SCHUR_ATOMICS") != nullptr;
    if (B.opt_cams && 6 * C + 1 <= 48 && !schur_atomics) {
      // matrix-core Schur complement: contiguous point ranges per wave, as many workgroups as keep >= 64 points per wave
      const int waves = (int)std::min<size_t>((size_t)SBA_SCHUR_WG * 4, (P + 63) / 64);
      const int ppw = (int)((P + waves - 1) / waves), n_wg = (waves + 3) / 4;
      const size_t lds_m = std::max((size_t)4 * SCH_B * (9 + 18 * C), (size_t)4 * 6 * 256) * 8;
      hipLaunchKernelGGL(k_sba_schur_mfma, dim3(n_wg), dim3(SCH_T), lds_m, s, B, lam, ppw);
      ACINO_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_sba_schur_reduce, dim3((unsigned)((n * n + n + 255) / 256), 32), dim3(256), 0, s, B, n_wg, 0);
      ACINO_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_sba_schur_reduce, dim3((unsigned)((n * n + n + 255) / 256), 1), dim3(256), 0, s, B, n_wg, 1);
    } else {
      if (B.opt_cams) ACINO_HIP_CHECK(hipMemsetAsync(B.S, 0, (n * n + n) * 8, s));
      hipLaunchKernelGGL(k_sba_schur, dim3(nblk), dim3(256), lds_s, s, B, lam);
    }
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
