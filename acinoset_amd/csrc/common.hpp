// Shared host/device helpers for libacinoset_hip (gfx950, fp64).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/acinoset_hip.h"

namespace acino {

void set_error(const char* fmt, ...);

#define ACINO_HIP_CHECK(expr)                                                              \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      ::acino::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                         __LINE__);                                                        \
      return ACINO_ERR_HIP;                                                                \
    }                                                                                      \
  } while (0)

#define ACINO_REQUIRE(cond, msg)                                        \
  do {                                                                  \
    if (!(cond)) {                                                      \
      ::acino::set_error("invalid argument: %s (%s)", msg, #cond);      \
      return ACINO_ERR_INVALID_ARG;                                     \
    }                                                                   \
  } while (0)

#define ACINO_LAUNCH_CHECK()                                                             \
  do {                                                                                   \
    hipError_t _e = hipGetLastError();                                                   \
    if (_e != hipSuccess) {                                                              \
      ::acino::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),      \
                         __FILE__, __LINE__);                                            \
      return ACINO_ERR_HIP;                                                              \
    }                                                                                    \
  } while (0)

// ---- fisheye camera record (24 doubles, see acinoset_hip.h) -------------------------------
struct Cam {
  double fx, fy, cx, cy;
  double k1, k2, k3, k4;
  double R[9];
  double t[3];
  double alpha, pad0, pad1, pad2;
};
static_assert(sizeof(Cam) == ACINO_CAM_STRIDE * sizeof(double), "camera record layout");

// 1 / x for a finite non-zero x: hardware estimate + two Newton steps (5 instructions, error within an ulp, against
// ~30 for the correctly rounded IEEE quotient).  The per-point camera arithmetic (undistortion, DLT, projection,
// robust loss) is bound by fp64 VALU issue and was one third divisions.
__device__ __forceinline__ double rcp64(double x) {
  double y = __builtin_amdgcn_rcp(x);
  double e = fma(-x, y, 1.0);
  y = fma(y, e, y);
  e = fma(-x, y, 1.0);
  return fma(y, e, y);
}

// Kannala-Brandt inverse (OpenCV fisheye::undistortPoints, no R/P).  Returns false when OpenCV would
// flag the point (not converged / theta sign flipped) - caller writes -1e6.
__device__ __forceinline__ bool undistort_fisheye_pt(const Cam& c, double u, double v, int max_iter, double eps,
                                                    double& xn, double& yn) {
  double pwy = (v - c.cy) * rcp64(c.fy);
  double pwx = (u - c.cx) * rcp64(c.fx) - c.alpha * pwy;
  double theta_d = sqrt(pwx * pwx + pwy * pwy);
  theta_d = fmin(fmax(-M_PI / 2.0, theta_d), M_PI / 2.0);
  bool converged = false;
  double theta = theta_d;
  double scale = 0.0;
  if (fabs(theta_d) > eps) {
    for (int j = 0; j < max_iter; ++j) {
      double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
      double k0t2 = c.k1 * t2, k1t4 = c.k2 * t4, k2t6 = c.k3 * t6, k3t8 = c.k4 * t8;
      double fix = (theta * (1 + k0t2 + k1t4 + k2t6 + k3t8) - theta_d) *
                   rcp64(1 + 3 * k0t2 + 5 * k1t4 + 7 * k2t6 + 9 * k3t8);
      theta = theta - fix;
      if (fabs(fix) < eps) {
        converged = true;
        break;
      }
    }
    scale = tan(theta) * rcp64(theta_d);
  } else {
    converged = true;
  }
  bool flipped = (theta_d < 0 && theta > 0) || (theta_d > 0 && theta < 0);
  xn = pwx * scale;
  yn = pwy * scale;
  return converged && !flipped;
}

// cv2.fisheye.projectPoints for one point (r > 1e-8 branch as OpenCV).
// FAST: reciprocals instead of the three IEEE quotients (the bulk residual kernels; within an ulp per operation).
template <bool FAST = false>
__device__ __forceinline__ void project_fisheye_pt(const Cam& c, double X, double Y, double Z, double& u,
                                                  double& v) {
  double xc = c.R[0] * X + c.R[1] * Y + c.R[2] * Z + c.t[0];
  double yc = c.R[3] * X + c.R[4] * Y + c.R[5] * Z + c.t[1];
  double zc = c.R[6] * X + c.R[7] * Y + c.R[8] * Z + c.t[2];
  const double iz = FAST ? rcp64(zc) : 1.0 / zc;
  double a = FAST ? xc * iz : xc / zc, b = FAST ? yc * iz : yc / zc;
  double r = sqrt(a * a + b * b);
  double th = atan(r);
  double th2 = th * th;
  double thd = th * (1 + th2 * (c.k1 + th2 * (c.k2 + th2 * (c.k3 + th2 * c.k4))));
  double cdist = r > 1e-8 ? (FAST ? thd * rcp64(r) : thd / r) : 1.0;
  double xd = a * cdist, yd = b * cdist;
  u = (xd + c.alpha * yd) * c.fx + c.cx;
  v = yd * c.fy + c.cy;
}

// bf16 storage rounding (round-to-nearest-even on the upper 16 bits of the float)
__device__ __forceinline__ float bf16_round(float x) {
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return __uint_as_float(u & 0xffff0000u);
}

__device__ __forceinline__ bool m_finite(double v) { return fabs(v) <= 1.79769313486231570e308; }   // false for NaN too

// Right-singular vector of the smallest singular value of the 4x4 DLT matrix by INVERSE ITERATION on B = A^T A
// (LDL^T of the 4x4, start e4 - the homogeneous solution (X, Y, Z, 1) / |.| always has a w component).  For a
// triangulation the three large singular values are O(1) and the smallest is the noise (sigma4 / sigma3 ~ 1e-3), so
// each step gains (sigma3 / sigma4)^2 ~ 1e5..1e6 and two or three steps reach 1e-16; squaring the matrix costs
// eps * lambda1 / (lambda3 - lambda4) ~ 1e-13 of accuracy on the unit vector, far inside the parity tolerance, because
// the wanted vector is separated from the rest by the GAP, not by the size of sigma4.  About 300 instructions
// against ~4 000 for the Jacobi SVD.  Returns false when six steps have not converged (a pair of rays that do not
// come close to meeting: the caller then falls back to the SVD, which is what the reference computes).
__device__ __forceinline__ bool dlt_null_vector_invit(const double A[4][4], double out[3]) {
  double B[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = i; j < 4; ++j) B[i][j] = A[0][i] * A[0][j] + A[1][i] * A[1][j] + A[2][i] * A[2][j] + A[3][i] * A[3][j];
  const double floor_ = 1e-30 * (B[0][0] + B[1][1] + B[2][2] + B[3][3]) + 1e-300;
  // B = L D L^T, unit lower L stored in l[][], reciprocal pivots in id[]
  double l10, l20, l30, l21, l31, l32, id[4];
  double d0 = fmax(B[0][0], floor_);
  id[0] = rcp64(d0);
  l10 = B[0][1] * id[0];
  l20 = B[0][2] * id[0];
  l30 = B[0][3] * id[0];
  double d1 = fmax(B[1][1] - l10 * B[0][1], floor_);
  id[1] = rcp64(d1);
  const double b21 = B[1][2] - l20 * B[0][1], b31 = B[1][3] - l30 * B[0][1];
  l21 = b21 * id[1];
  l31 = b31 * id[1];
  double d2 = fmax(B[2][2] - l20 * B[0][2] - l21 * b21, floor_);
  id[2] = rcp64(d2);
  const double b32 = B[2][3] - l30 * B[0][2] - l31 * b21;
  l32 = b32 * id[2];
  double d3 = fmax(B[3][3] - l30 * B[0][3] - l31 * b31 - l32 * b32, floor_);
  id[3] = rcp64(d3);
  double x0 = 0.0, x1 = 0.0, x2 = 0.0, x3 = 1.0;
  bool conv = false;
#pragma unroll 1
  for (int it = 0; it < 6 && !conv; ++it) {
    // forward L z = x, scale by D^-1, backward L^T y = z
    const double z0 = x0, z1 = x1 - l10 * z0, z2 = x2 - l20 * z0 - l21 * z1, z3 = x3 - l30 * z0 - l31 * z1 - l32 * z2;
    const double y3 = z3 * id[3];
    const double y2 = z2 * id[2] - l32 * y3;
    const double y1 = z1 * id[1] - l21 * y2 - l31 * y3;
    const double y0 = z0 * id[0] - l10 * y1 - l20 * y2 - l30 * y3;
    // scale-safe normalisation (1 / d3 can be 1e30 for noise-free data)
    const double m = fmax(fmax(fabs(y0), fabs(y1)), fmax(fabs(y2), fabs(y3)));
    const double im = rcp64(m);
    const double s0 = y0 * im, s1 = y1 * im, s2 = y2 * im, s3 = y3 * im;
    const double rn = rsqrt(s0 * s0 + s1 * s1 + s2 * s2 + s3 * s3);
    const double n0 = s0 * rn, n1 = s1 * rn, n2 = s2 * rn, n3 = s3 * rn;
    const double diff = fmax(fmax(fabs(n0 - x0), fabs(n1 - x1)), fmax(fabs(n2 - x2), fabs(n3 - x3)));
    conv = diff < 1e-13;      // the step just taken shrank the error by another (sigma4 / sigma3)^2
    x0 = n0;
    x1 = n1;
    x2 = n2;
    x3 = n3;
  }
  const double iw = rcp64(x3);      // x3 == 0 (a point at infinity) gives NaN here: not finite, the caller falls back
  out[0] = x0 * iw;
  out[1] = x1 * iw;
  out[2] = x2 * iw;
  return conv && m_finite(out[0]) && m_finite(out[1]) && m_finite(out[2]);
}

// Null vector of the 4x4 DLT matrix by one-sided Jacobi (Hestenes) on its columns: on exit the
// columns of A*V are orthogonal; the right-singular vector of the smallest singular value is the
// column of V whose A*V column has the smallest norm.  Returns X/W (dehomogenised).
__device__ __forceinline__ void dlt_null_vector(double A[4][4], double out[3]) {
  if (dlt_null_vector_invit(A, out)) return;
  double V[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int q = p + 1; q < 4; ++q) {
        double al = 0, be = 0, ga = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          al += A[i][p] * A[i][p];
          be += A[i][q] * A[i][q];
          ga += A[i][p] * A[i][q];
        }
        const double ab = al * be;
        const double cosang = ab > 0.0 ? fabs(ga) * rsqrt(ab) : 0.0;     // |cos| of the angle between columns p, q
        if (cosang > 1e-17 && fabs(ga) > 1e-30) {
          off = fmax(off, cosang);
          const double zeta = (be - al) / (2.0 * ga);
          const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          const double cs = rsqrt(1.0 + tt * tt), sn = cs * tt;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            double ap = A[i][p], aq = A[i][q];
            A[i][p] = cs * ap - sn * aq;
            A[i][q] = sn * ap + cs * aq;
            double vp = V[i][p], vq = V[i][q];
            V[i][p] = cs * vp - sn * vq;
            V[i][q] = sn * vp + cs * vq;
          }
        }
      }
    }
    // one-sided Jacobi converges quadratically: a sweep that STARTED below 1e-8 leaves the columns orthogonal to
    // ~1e-16, so no further (pure checking) sweep is needed
    if (off < 1e-8) break;
  }
  double best = 1e300;
  double x = 0, y = 0, z = 0, w = 1;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double nrm = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) nrm += A[i][j] * A[i][j];
    if (nrm < best) {
      best = nrm;
      x = V[0][j];
      y = V[1][j];
      z = V[2][j];
      w = V[3][j];
    }
  }
  out[0] = x / w;
  out[1] = y / w;
  out[2] = z / w;
}

// Two-view DLT from normalised image points (cv2.triangulatePoints + dehomogenise, calib.py:126-129).
__device__ __forceinline__ void triangulate_two_view(const Cam& ca, const Cam& cb, double x1, double y1, double x2,
                                                    double y2, double out[3]) {
  double A[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double p1_0 = k < 3 ? ca.R[k] : ca.t[0], p1_1 = k < 3 ? ca.R[3 + k] : ca.t[1],
           p1_2 = k < 3 ? ca.R[6 + k] : ca.t[2];
    double p2_0 = k < 3 ? cb.R[k] : cb.t[0], p2_1 = k < 3 ? cb.R[3 + k] : cb.t[1],
           p2_2 = k < 3 ? cb.R[6 + k] : cb.t[2];
    A[0][k] = x1 * p1_2 - p1_0;
    A[1][k] = y1 * p1_2 - p1_1;
    A[2][k] = x2 * p2_2 - p2_0;
    A[3][k] = y2 * p2_2 - p2_1;
  }
  dlt_null_vector(A, out);
}

// The opt-in to more than 64 KB of dynamic LDS (hipFuncSetAttribute) is a per-DEVICE property of a kernel: remember it
// per device, not per process (a process may drive more than one GPU).
struct PerDeviceOnce {
  unsigned char seen[64] = {};
  bool first() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    if (seen[dev]) return false;
    seen[dev] = 1;
    return true;
  }
};

// ---- redescending loss (build.py:382-395) ---------------------------------------------------
struct LossC {
  double a, b, c;
  double ea, eb, ec;     // exp(a), exp(b), exp(c)
  double d0;             // rho'(0+)
  double t4;             // a*b - a^2/2 + a*(c-b)/2
  double icb;            // 1 / (c - b)
};

__host__ __device__ inline double loss_step(double start, double x) { return 1.0 / (1.0 + exp(-(x - start))); }

inline LossC make_loss(double a, double b, double c) {
  LossC L;
  L.a = a; L.b = b; L.c = c;
  L.ea = exp(a); L.eb = exp(b); L.ec = exp(c);
  double sa = loss_step(a, 0.0), sb = loss_step(b, 0.0), sc = loss_step(c, 0.0);
  double dsa = sa * (1 - sa), dsb = sb * (1 - sb), dsc = sc * (1 - sc);
  double cb = c - b;
  double t2 = -a * a / 2;
  double t3 = a * b - a * a / 2 + (a * cb / 2) * (1 - (c / cb) * (c / cb));
  L.t4 = a * b - a * a / 2 + a * cb / 2;
  L.icb = 1.0 / cb;
  L.d0 = (dsa - dsb) * t2 + (sa - sb) * a + (dsb - dsc) * t3 + (sb - sc) * (a * c / cb) + dsc * L.t4;
  return L;
}

// rho(e), rho'(e) and the Gauss-Newton curvature weight h = clip((rho'(e) - rho'(0+))/e, 0, 1); e = |err|.
template <bool DERIV>
__device__ __forceinline__ void redescending(const LossC& L, double err, double& rho, double& drho, double& h) {
  double e = fabs(err);
  double u = exp(-e);                       // one exp: sigma(s, e) = 1 / (1 + exp(s) * exp(-e))
  // ... and one division for the three logistic steps (the product stays below 1e15 for e >= 0)
  const double da = 1.0 + L.ea * u, db = 1.0 + L.eb * u, dc = 1.0 + L.ec * u;
  const double inv = rcp64(da * db * dc);
  double sa = inv * (db * dc), sb = inv * (da * dc), sc = inv * (da * db);
  double cb = L.c - L.b;
  double t2 = L.a * e - L.a * L.a / 2;
  double ce = (L.c - e) * L.icb;
  double t3 = L.a * L.b - L.a * L.a / 2 + (L.a * cb / 2) * (1 - ce * ce);
  rho = (1 - sa) / 2 * e * e + (sa - sb) * t2 + (sb - sc) * t3 + sc * L.t4;
  if (DERIV) {
    double dsa = sa * (1 - sa), dsb = sb * (1 - sb), dsc = sc * (1 - sc);
    drho = -dsa / 2 * e * e + (1 - sa) * e + (dsa - dsb) * t2 + (sa - sb) * L.a + (dsb - dsc) * t3 +
           (sb - sc) * (L.a * ce) + dsc * L.t4;
    double hh = e > 1e-12 ? (drho - L.d0) * rcp64(e) : 1.0;
    h = fmin(fmax(hh, 0.0), 1.0);
  }
}

}  // namespace acino
