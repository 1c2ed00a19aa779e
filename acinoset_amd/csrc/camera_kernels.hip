// Point-wise camera model kernels and the dense adjacent-pair triangulation (gfx950, fp64).
// Reference: src/calib/calib.py:52-66,121-136,394-423 (OpenCV algorithms restated in DESIGN.md).
#include <stdarg.h>

#include "common.hpp"

namespace acino {

static thread_local char g_err[512] = "ok";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* last_error() { return g_err; }

// ---- pinhole (rational) model ----------------------------------------------------------------
struct Pin {
  double fx, fy, cx, cy;
  double d[14];
  double R[9];
  double t[3];
  double pad0, pad1;
};
static_assert(sizeof(Pin) == ACINO_PINHOLE_STRIDE * sizeof(double), "pinhole record layout");

__device__ __forceinline__ void undistort_pinhole_pt(const Pin& c, double u, double v, double& x, double& y) {
  const double* k = c.d;
  double x0 = (u - c.cx) / c.fx, y0 = (v - c.cy) / c.fy;
  x = x0;
  y = y0;
  for (int j = 0; j < 5; ++j) {  // OpenCV default criteria (MAX_ITER, 5, 0.01)
    double r2 = x * x + y * y;
    double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
    if (icdist < 0) {
      x = x0;
      y = y0;
      break;
    }
    double dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
    double dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
    x = (x0 - dx) * icdist;
    y = (y0 - dy) * icdist;
  }
}

__global__ void __launch_bounds__(256) k_undistort_fisheye(const double* __restrict__ pts, int64_t m, const double* __restrict__ cam,
                                    double* __restrict__ out, int max_iter, double eps) {
  __shared__ Cam c;
  if (threadIdx.x < ACINO_CAM_STRIDE) reinterpret_cast<double*>(&c)[threadIdx.x] = cam[threadIdx.x];
  __syncthreads();
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    double2 p = reinterpret_cast<const double2*>(pts)[i];
    double x, y;
    bool ok = undistort_fisheye_pt(c, p.x, p.y, max_iter, eps, x, y);
    if (!ok) x = y = -1000000.0;
    reinterpret_cast<double2*>(out)[i] = make_double2(x, y);
  }
}

__global__ void __launch_bounds__(256) k_triangulate_fisheye(const double* __restrict__ p1, const double* __restrict__ p2, int64_t m,
                                      const double* __restrict__ cam_a, const double* __restrict__ cam_b,
                                      double* __restrict__ out) {
  __shared__ Cam c[2];
  if (threadIdx.x < ACINO_CAM_STRIDE) {
    reinterpret_cast<double*>(&c[0])[threadIdx.x] = cam_a[threadIdx.x];
    reinterpret_cast<double*>(&c[1])[threadIdx.x] = cam_b[threadIdx.x];
  }
  __syncthreads();
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    double2 a = reinterpret_cast<const double2*>(p1)[i];
    double2 b = reinterpret_cast<const double2*>(p2)[i];
    double x1, y1, x2, y2;
    if (!undistort_fisheye_pt(c[0], a.x, a.y, 10, 1e-8, x1, y1)) x1 = y1 = -1000000.0;
    if (!undistort_fisheye_pt(c[1], b.x, b.y, 10, 1e-8, x2, y2)) x2 = y2 = -1000000.0;
    double X[3];
    triangulate_two_view(c[0], c[1], x1, y1, x2, y2, X);
    out[3 * i + 0] = X[0];
    out[3 * i + 1] = X[1];
    out[3 * i + 2] = X[2];
  }
}

__global__ void __launch_bounds__(256) k_triangulate_pinhole(const double* __restrict__ p1, const double* __restrict__ p2, int64_t m,
                                      const double* __restrict__ cam_a, const double* __restrict__ cam_b,
                                      double* __restrict__ out) {
  __shared__ Pin c[2];
  if (threadIdx.x < ACINO_PINHOLE_STRIDE) {
    reinterpret_cast<double*>(&c[0])[threadIdx.x] = cam_a[threadIdx.x];
    reinterpret_cast<double*>(&c[1])[threadIdx.x] = cam_b[threadIdx.x];
  }
  __syncthreads();
  Cam ea, eb;  // only R, t are read by the DLT
  for (int j = 0; j < 9; ++j) {
    ea.R[j] = c[0].R[j];
    eb.R[j] = c[1].R[j];
  }
  for (int j = 0; j < 3; ++j) {
    ea.t[j] = c[0].t[j];
    eb.t[j] = c[1].t[j];
  }
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    double2 a = reinterpret_cast<const double2*>(p1)[i];
    double2 b = reinterpret_cast<const double2*>(p2)[i];
    double x1, y1, x2, y2;
    undistort_pinhole_pt(c[0], a.x, a.y, x1, y1);
    undistort_pinhole_pt(c[1], b.x, b.y, x2, y2);
    double X[3];
    triangulate_two_view(ea, eb, x1, y1, x2, y2, X);
    out[3 * i + 0] = X[0];
    out[3 * i + 1] = X[1];
    out[3 * i + 2] = X[2];
  }
}

__global__ void __launch_bounds__(256) k_project_fisheye(const double* __restrict__ obj, int64_t m, const double* __restrict__ cam,
                                  double* __restrict__ out) {
  __shared__ Cam c;
  if (threadIdx.x < ACINO_CAM_STRIDE) reinterpret_cast<double*>(&c)[threadIdx.x] = cam[threadIdx.x];
  __syncthreads();
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    double u, v;
    project_fisheye_pt(c, obj[3 * i], obj[3 * i + 1], obj[3 * i + 2], u, v);
    reinterpret_cast<double2*>(out)[i] = make_double2(u, v);
  }
}

__global__ void __launch_bounds__(256) k_project_pinhole(const double* __restrict__ obj, int64_t m, const double* __restrict__ cam,
                                  double* __restrict__ out) {
  __shared__ Pin c;
  if (threadIdx.x < ACINO_PINHOLE_STRIDE) reinterpret_cast<double*>(&c)[threadIdx.x] = cam[threadIdx.x];
  __syncthreads();
  const double* k = c.d;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
    double xc = c.R[0] * X + c.R[1] * Y + c.R[2] * Z + c.t[0];
    double yc = c.R[3] * X + c.R[4] * Y + c.R[5] * Z + c.t[1];
    double zc = c.R[6] * X + c.R[7] * Y + c.R[8] * Z + c.t[2];
    double x = xc / zc, y = yc / zc;
    double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    double cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
    double icdist2 = 1.0 / (1 + k[5] * r2 + k[6] * r4 + k[7] * r6);
    double xd = x * cdist * icdist2 + k[2] * a1 + k[3] * a2 + k[8] * r2 + k[9] * r4;
    double yd = y * cdist * icdist2 + k[2] * a3 + k[3] * a1 + k[10] * r2 + k[11] * r4;
    reinterpret_cast<double2*>(out)[i] = make_double2(xd * c.fx + c.cx, yd * c.fy + c.cy);
  }
}

// ---- dense adjacent-pair triangulation ------------------------------------------------------
// One thread per (frame, marker), 256 consecutive (frame, marker) items per workgroup pass.  The detections those
// items need are ONE contiguous span of d_det (whole frames: C x L records of 24 bytes each), so the workgroup copies
// the span into LDS with fully coalesced 8-byte loads and the per-thread 24-byte record reads (x, y, likelihood of
// camera c) come from LDS - HBM sees each byte once, in order.  Camera records are staged once per workgroup.
// Undistortion of camera c is reused for pairs (c-1,c) and (c,c+1).  Mean = Kahan sum in pair order / count (pandas
// group_mean), NaN when no pair.  REPROJECT fuses k_reproject_residuals: the mean point is projected into every
// camera with a valid detection while the detections are still in LDS (BASELINE config 2 in one pass).
template <bool REPROJECT>
__global__ void __launch_bounds__(256)
k_triangulate_pairs(const double* __restrict__ det, int64_t n_frames, int n_cams, int n_markers, double thresh,
                    const double* __restrict__ cams, double* __restrict__ tri, uint8_t* __restrict__ npairs,
                    uint8_t* __restrict__ pairmask, double* __restrict__ res, double* __restrict__ sums,
                    int stage_doubles) {
  extern __shared__ __attribute__((aligned(16))) double stage[];
  __shared__ Cam c[ACINO_MAX_CAMS];
  __shared__ double red[4][4];
  for (int i = threadIdx.x; i < n_cams * ACINO_CAM_STRIDE; i += blockDim.x)
    reinterpret_cast<double*>(c)[i] = cams[i];
  const int64_t total = n_frames * n_markers;
  const int64_t frame_doubles = (int64_t)n_cams * n_markers * 3;
  double s_cnt = 0, s_r = 0, s_r2 = 0, s_c = 0;
  for (int64_t idx0 = blockIdx.x * (int64_t)256; idx0 < total; idx0 += (int64_t)gridDim.x * 256) {
    const int64_t nf0 = idx0 / n_markers;
    int64_t nf1 = (idx0 + 255) / n_markers;
    if (nf1 > n_frames - 1) nf1 = n_frames - 1;
    const int64_t span = (nf1 - nf0 + 1) * frame_doubles;
    const bool staged = span <= stage_doubles;
    __syncthreads();                              // the previous pass is done with the stage (and c[] is complete)
    if (staged) {
      const double* src = det + nf0 * frame_doubles;
      for (int64_t e = threadIdx.x; e < span; e += 256) stage[e] = src[e];
    }
    __syncthreads();
    const int64_t idx = idx0 + threadIdx.x;
    if (idx < total) {
      const int64_t n = idx / n_markers;
      const int l = (int)(idx - n * n_markers);
      const double* base = staged ? stage + (n - nf0) * frame_doubles + (int64_t)l * 3
                                  : det + n * frame_doubles + (int64_t)l * 3;
      double sum[3] = {0, 0, 0}, comp[3] = {0, 0, 0};
      int cnt = 0;
      unsigned mask = 0;
      bool pv = false;
      double px = 0, py = 0;
      for (int ci = 0; ci < n_cams; ++ci) {
        const double* d = base + (int64_t)ci * n_markers * 3;
        double u = d[0], v = d[1], lik = d[2];
        bool valid = lik > thresh;
        double x = 0, y = 0;
        if (valid) {
          if (!undistort_fisheye_pt(c[ci], u, v, 10, 1e-8, x, y)) x = y = -1000000.0;
        }
        if (valid && pv) {
          double X[3];
          triangulate_two_view(c[ci - 1], c[ci], px, py, x, y, X);
#pragma unroll
          for (int k = 0; k < 3; ++k) {  // Kahan step
            double yk = X[k] - comp[k];
            double t = sum[k] + yk;
            comp[k] = (t - sum[k]) - yk;
            sum[k] = t;
          }
          ++cnt;
          mask |= 1u << (ci - 1);
        }
        pv = valid;
        px = x;
        py = y;
      }
      double X = __builtin_nan(""), Y = X, Z = X;
      if (cnt > 0) {
        X = sum[0] / cnt;
        Y = sum[1] / cnt;
        Z = sum[2] / cnt;
      }
      double* o = tri + idx * 3;
      o[0] = X;
      o[1] = Y;
      o[2] = Z;
      if (npairs) npairs[idx] = (uint8_t)cnt;
      if (pairmask) pairmask[idx] = (uint8_t)mask;
      if (REPROJECT) {
        const bool fin = isfinite(X) && isfinite(Y) && isfinite(Z);
        for (int ci = 0; ci < n_cams; ++ci) {
          const double* d = base + (int64_t)ci * n_markers * 3;
          double ru = __builtin_nan(""), rv = __builtin_nan("");
          if (fin && d[2] > thresh) {
            double u, v;
            project_fisheye_pt<true>(c[ci], X, Y, Z, u, v);
            ru = u - d[0];
            rv = v - d[1];
            s_cnt += 2;
            s_r += ru + rv;
            s_r2 += ru * ru + rv * rv;
            s_c += 0.5 * (log1p(ru * ru) + log1p(rv * rv));
          }
          reinterpret_cast<double2*>(res)[(n * n_cams + ci) * n_markers + l] = make_double2(ru, rv);
        }
      }
    }
  }
  if (REPROJECT && sums) {
    double v[4] = {s_cnt, s_r, s_r2, s_c};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
      for (int k = 0; k < 4; ++k) red[wave][k] = v[k];
    __syncthreads();
    if (threadIdx.x < 4) {
      double t = 0;
      for (int w = 0; w < 4; ++w) t += red[w][threadIdx.x];
      atomicAdd(&sums[threadIdx.x], t);
    }
  }
}

// Dense adjacent-pair triangulation with the PINHOLE model (the reference's other injected ``triangulate_func``:
// calib.py:52-61 through the seam of calib.py:394-417; app.py:215-218 injects this pair).  Same index path as
// k_triangulate_pairs - pairs (i, i+1), Kahan mean in pair order, NaN when no pair - with cv2.undistortPoints'
// 5-step fixed-point undistortion.  One thread per (frame, marker); the 24-byte records are read from HBM directly
// (the pinhole path is the calibration-time SBA preparation, not the per-frame hot loop).
__global__ void __launch_bounds__(256)
k_triangulate_pairs_pinhole(const double* __restrict__ det, int64_t n_frames, int n_cams, int n_markers, double thresh,
                            const double* __restrict__ cams, double* __restrict__ tri, uint8_t* __restrict__ npairs,
                            uint8_t* __restrict__ pairmask) {
  __shared__ Pin c[ACINO_MAX_PAIR_CAMS];
  __shared__ Cam e[ACINO_MAX_PAIR_CAMS];     // only R, t are read by the DLT
  for (int i = threadIdx.x; i < n_cams * ACINO_PINHOLE_STRIDE; i += blockDim.x)
    reinterpret_cast<double*>(c)[i] = cams[i];
  __syncthreads();
  for (int i = threadIdx.x; i < n_cams * 12; i += blockDim.x) {
    const int ci = i / 12, j = i % 12;
    if (j < 9) e[ci].R[j] = c[ci].R[j];
    else e[ci].t[j - 9] = c[ci].t[j - 9];
  }
  __syncthreads();
  const int64_t total = n_frames * n_markers;
  const int64_t frame_doubles = (int64_t)n_cams * n_markers * 3;
  for (int64_t idx = blockIdx.x * (int64_t)256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t n = idx / n_markers;
    const int l = (int)(idx - n * n_markers);
    const double* base = det + n * frame_doubles + (int64_t)l * 3;
    double sum[3] = {0, 0, 0}, comp[3] = {0, 0, 0};
    int cnt = 0;
    unsigned mask = 0;
    bool pv = false;
    double px = 0, py = 0;
    for (int ci = 0; ci < n_cams; ++ci) {
      const double* d = base + (int64_t)ci * n_markers * 3;
      const double u = d[0], v = d[1], lik = d[2];
      const bool valid = lik > thresh;
      double x = 0, y = 0;
      if (valid) undistort_pinhole_pt(c[ci], u, v, x, y);
      if (valid && pv) {
        double X[3];
        triangulate_two_view(e[ci - 1], e[ci], px, py, x, y, X);
#pragma unroll
        for (int k = 0; k < 3; ++k) {  // Kahan step (pandas group mean)
          double yk = X[k] - comp[k];
          double t = sum[k] + yk;
          comp[k] = (t - sum[k]) - yk;
          sum[k] = t;
        }
        ++cnt;
        mask |= 1u << (ci - 1);
      }
      pv = valid;
      px = x;
      py = y;
    }
    double X = __builtin_nan(""), Y = X, Z = X;
    if (cnt > 0) {
      X = sum[0] / cnt;
      Y = sum[1] / cnt;
      Z = sum[2] / cnt;
    }
    double* o = tri + idx * 3;
    o[0] = X;
    o[1] = Y;
    o[2] = Z;
    if (npairs) npairs[idx] = (uint8_t)cnt;
    if (pairmask) pairmask[idx] = (uint8_t)mask;
  }
}

// Reprojection residual of pts3[N][L][3] in every camera; one thread per (frame, marker).
__global__ void __launch_bounds__(256)
k_reproject_residuals(const double* __restrict__ pts3, const double* __restrict__ det, int64_t n_frames, int n_cams,
                      int n_markers, double thresh, const double* __restrict__ cams, double* __restrict__ res,
                      double* __restrict__ sums) {
  __shared__ Cam c[ACINO_MAX_CAMS];
  __shared__ double red[4][4];
  for (int i = threadIdx.x; i < n_cams * ACINO_CAM_STRIDE; i += blockDim.x)
    reinterpret_cast<double*>(c)[i] = cams[i];
  __syncthreads();
  double s_cnt = 0, s_r = 0, s_r2 = 0, s_c = 0;
  const int64_t total = n_frames * n_markers;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = idx / n_markers;
    const int l = (int)(idx - n * n_markers);
    double X = pts3[idx * 3], Y = pts3[idx * 3 + 1], Z = pts3[idx * 3 + 2];
    bool fin = isfinite(X) && isfinite(Y) && isfinite(Z);
    for (int ci = 0; ci < n_cams; ++ci) {
      const int64_t o = ((n * n_cams + ci) * n_markers + l);
      const double* d = det + o * 3;
      double ru = __builtin_nan(""), rv = __builtin_nan("");
      if (fin && d[2] > thresh) {
        double u, v;
        project_fisheye_pt<true>(c[ci], X, Y, Z, u, v);
        ru = u - d[0];
        rv = v - d[1];
        s_cnt += 2;
        s_r += ru + rv;
        s_r2 += ru * ru + rv * rv;
        s_c += 0.5 * (log1p(ru * ru) + log1p(rv * rv));
      }
      reinterpret_cast<double2*>(res)[o] = make_double2(ru, rv);
    }
  }
  if (sums) {
    double v[4] = {s_cnt, s_r, s_r2, s_c};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
      for (int k = 0; k < 4; ++k) red[wave][k] = v[k];
    __syncthreads();
    if (threadIdx.x < 4) {
      double t = 0;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w][threadIdx.x];
      atomicAdd(&sums[threadIdx.x], t);
    }
  }
}

static inline int grid_for(int64_t work, int block) {
  int64_t g = (work + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 256 * 8) g = 256 * 8;  // 256 CUs x 8 blocks, grid-stride beyond
  return (int)g;
}

}  // namespace acino

using namespace acino;

extern "C" {

const char* acino_last_error_string(void) { return acino::last_error(); }
int acino_abi_version(void) { return ACINO_ABI_VERSION; }
#ifndef ACINO_BUILD_ID
#define ACINO_BUILD_ID "unstamped"
#endif
static const char k_build_id[] = "ACINO_BUILD_ID=" ACINO_BUILD_ID;   // the tag lets build() read it from the file
const char* acino_build_id(void) { return k_build_id + 15; }
size_t acino_sizeof_fte_params(void) { return sizeof(acino_fte_params); }
size_t acino_sizeof_fte_state(void) { return sizeof(acino_fte_state); }
int acino_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
    return ACINO_ERR_NO_DEVICE;
  }
  return n;
}

int acino_undistort_fisheye(const double* d_pts, int64_t m, const double* d_cam24, double* d_out, int max_iter,
                            double eps, void* stream) {
  ACINO_REQUIRE(m >= 0, "m");
  ACINO_REQUIRE(d_cam24 != nullptr, "camera");
  if (m == 0) return ACINO_OK;
  ACINO_REQUIRE(d_pts && d_out, "null buffer");
  hipLaunchKernelGGL(k_undistort_fisheye, dim3(grid_for(m, 256)), dim3(256), 0, (hipStream_t)stream, d_pts, m,
                     d_cam24, d_out, max_iter, eps);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

int acino_triangulate_fisheye(const double* d_pts1, const double* d_pts2, int64_t m, const double* d_cam_a24,
                              const double* d_cam_b24, double* d_out, void* stream) {
  ACINO_REQUIRE(m >= 0, "m");
  ACINO_REQUIRE(d_cam_a24 && d_cam_b24, "camera");
  if (m == 0) return ACINO_OK;
  ACINO_REQUIRE(d_pts1 && d_pts2 && d_out, "null buffer");
  hipLaunchKernelGGL(k_triangulate_fisheye, dim3(grid_for(m, 256)), dim3(256), 0, (hipStream_t)stream, d_pts1,
                     d_pts2, m, d_cam_a24, d_cam_b24, d_out);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

int acino_triangulate_pinhole(const double* d_pts1, const double* d_pts2, int64_t m, const double* d_cam_a32,
                              const double* d_cam_b32, double* d_out, void* stream) {
  ACINO_REQUIRE(m >= 0, "m");
  ACINO_REQUIRE(d_cam_a32 && d_cam_b32, "camera");
  if (m == 0) return ACINO_OK;
  ACINO_REQUIRE(d_pts1 && d_pts2 && d_out, "null buffer");
  hipLaunchKernelGGL(k_triangulate_pinhole, dim3(grid_for(m, 256)), dim3(256), 0, (hipStream_t)stream, d_pts1,
                     d_pts2, m, d_cam_a32, d_cam_b32, d_out);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

int acino_project_fisheye(const double* d_obj, int64_t m, const double* d_cam24, double* d_out, void* stream) {
  ACINO_REQUIRE(m >= 0, "m");
  ACINO_REQUIRE(d_cam24 != nullptr, "camera");
  if (m == 0) return ACINO_OK;
  ACINO_REQUIRE(d_obj && d_out, "null buffer");
  hipLaunchKernelGGL(k_project_fisheye, dim3(grid_for(m, 256)), dim3(256), 0, (hipStream_t)stream, d_obj, m,
                     d_cam24, d_out);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

int acino_project_pinhole(const double* d_obj, int64_t m, const double* d_cam32, double* d_out, void* stream) {
  ACINO_REQUIRE(m >= 0, "m");
  ACINO_REQUIRE(d_cam32 != nullptr, "camera");
  if (m == 0) return ACINO_OK;
  ACINO_REQUIRE(d_obj && d_out, "null buffer");
  hipLaunchKernelGGL(k_project_pinhole, dim3(grid_for(m, 256)), dim3(256), 0, (hipStream_t)stream, d_obj, m,
                     d_cam32, d_out);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

static int launch_pairs(bool reproject, const double* d_det, int64_t n_frames, int n_cams, int n_markers, double thresh,
                        const double* d_cams24, double* d_tri, uint8_t* d_npairs, uint8_t* d_pairmask, double* d_res,
                        double* d_sums, void* stream) {
  // whole frames covering 256 consecutive (frame, marker) items; rigs whose span exceeds 60 KB read HBM directly
  const int64_t frames = 256 / n_markers + 2;
  int64_t stage = frames * n_cams * n_markers * 3;
  if (stage * 8 > 60 * 1024) stage = 0;      // (static LDS - camera records - shares the 64 KB that need no opt-in)
  const int grid = grid_for(n_frames * n_markers, 256);
  if (reproject)
    hipLaunchKernelGGL(k_triangulate_pairs<true>, dim3(grid), dim3(256), (size_t)stage * 8, (hipStream_t)stream, d_det,
                       n_frames, n_cams, n_markers, thresh, d_cams24, d_tri, d_npairs, d_pairmask, d_res, d_sums, (int)stage);
  else
    hipLaunchKernelGGL(k_triangulate_pairs<false>, dim3(grid), dim3(256), (size_t)stage * 8, (hipStream_t)stream, d_det,
                       n_frames, n_cams, n_markers, thresh, d_cams24, d_tri, d_npairs, d_pairmask, d_res, d_sums, (int)stage);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

int acino_triangulate_pairs(const double* d_det, int64_t n_frames, int n_cams, int n_markers, double thresh,
                            const double* d_cams24, double* d_tri, uint8_t* d_npairs, uint8_t* d_pairmask,
                            void* stream) {
  ACINO_REQUIRE(n_frames >= 0 && n_markers >= 0, "sizes");
  ACINO_REQUIRE(n_cams >= 1 && n_cams <= ACINO_MAX_PAIR_CAMS, "n_cams must be 1..8 (pair mask is one byte)");
  if (n_frames == 0 || n_markers == 0) return ACINO_OK;
  ACINO_REQUIRE(d_det && d_cams24 && d_tri, "null buffer");
  return launch_pairs(false, d_det, n_frames, n_cams, n_markers, thresh, d_cams24, d_tri, d_npairs, d_pairmask, nullptr,
                      nullptr, stream);
}

int acino_triangulate_pairs_pinhole(const double* d_det, int64_t n_frames, int n_cams, int n_markers, double thresh,
                                    const double* d_cams32, double* d_tri, uint8_t* d_npairs, uint8_t* d_pairmask,
                                    void* stream) {
  ACINO_REQUIRE(n_frames >= 0 && n_markers >= 0, "sizes");
  ACINO_REQUIRE(n_cams >= 1 && n_cams <= ACINO_MAX_PAIR_CAMS, "n_cams must be 1..8 (pair mask is one byte)");
  if (n_frames == 0 || n_markers == 0) return ACINO_OK;
  ACINO_REQUIRE(d_det && d_cams32 && d_tri, "null buffer");
  hipLaunchKernelGGL(k_triangulate_pairs_pinhole, dim3(grid_for(n_frames * n_markers, 256)), dim3(256), 0,
                     (hipStream_t)stream, d_det, n_frames, n_cams, n_markers, thresh, d_cams32, d_tri, d_npairs,
                     d_pairmask);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

int acino_triangulate_reproject(const double* d_det, int64_t n_frames, int n_cams, int n_markers, double thresh,
                                const double* d_cams24, double* d_tri, uint8_t* d_npairs, uint8_t* d_pairmask,
                                double* d_res, double* d_sums, void* stream) {
  ACINO_REQUIRE(n_frames >= 0 && n_markers >= 0, "sizes");
  ACINO_REQUIRE(n_cams >= 1 && n_cams <= ACINO_MAX_PAIR_CAMS, "n_cams must be 1..8 (pair mask is one byte)");
  if (n_frames == 0 || n_markers == 0) return ACINO_OK;
  ACINO_REQUIRE(d_det && d_cams24 && d_tri && d_res, "null buffer");
  return launch_pairs(true, d_det, n_frames, n_cams, n_markers, thresh, d_cams24, d_tri, d_npairs, d_pairmask, d_res,
                      d_sums, stream);
}

int acino_reproject_residuals(const double* d_pts3, const double* d_det, int64_t n_frames, int n_cams,
                              int n_markers, double thresh, const double* d_cams24, double* d_res, double* d_sums,
                              void* stream) {
  ACINO_REQUIRE(n_frames >= 0 && n_markers >= 0, "sizes");
  ACINO_REQUIRE(n_cams >= 1 && n_cams <= ACINO_MAX_CAMS, "n_cams");
  if (n_frames == 0 || n_markers == 0) return ACINO_OK;
  ACINO_REQUIRE(d_pts3 && d_det && d_cams24 && d_res, "null buffer");
  hipLaunchKernelGGL(k_reproject_residuals, dim3(grid_for(n_frames * n_markers, 256)), dim3(256), 0,
                     (hipStream_t)stream, d_pts3, d_det, n_frames, n_cams, n_markers, thresh, d_cams24, d_res,
                     d_sums);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

}  // extern "C"

// ---- generic-skeleton forward kinematics (build.py:28-86) ---------------------------------------------------
namespace acino {
struct SkelProgram {
  int n_ops, n_pose, n_angles, pad;
  acino_skel_op op[ACINO_SKEL_MAX_OPS];
};

// One thread per frame; the program is tiny and uniform, poses live in the output array (each thread re-reads
// only what it wrote itself).
__global__ void __launch_bounds__(256)
k_skeleton_fk(const double* __restrict__ q, int64_t n_frames, SkelProgram P, double* __restrict__ pos) {
  const int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (n >= n_frames) return;
  const int L = P.n_angles;
  const double* x = q + n * (3 + 3 * L);
  double* out = pos + n * P.n_pose * 3;
  const double rx = x[0], ry = x[1], rz = x[2];
  for (int s = 0; s < P.n_pose; ++s) {
    out[3 * s] = rx;
    out[3 * s + 1] = ry;
    out[3 * s + 2] = rz;
  }
  for (int k = 0; k < P.n_ops; ++k) {
    const acino_skel_op& o = P.op[k];
    // R_loc = Rz(psi) Rx(phi) Ry(theta), reference convention rot_x = [[1,0,0],[0,c,s],[0,-s,c]] etc.
    double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    if (o.flags & 2) {
      double s, c;
      sincos(x[3 + L + o.angle], &s, &c);
      R[0][0] = c; R[0][2] = -s; R[2][0] = s; R[2][2] = c;
    }
    if (o.flags & 1) {   // R = Rx @ R
      double s, c;
      sincos(x[3 + o.angle], &s, &c);
      for (int j = 0; j < 3; ++j) {
        const double r1 = R[1][j], r2 = R[2][j];
        R[1][j] = c * r1 + s * r2;
        R[2][j] = -s * r1 + c * r2;
      }
    }
    if (o.flags & 4) {   // R = Rz @ R
      double s, c;
      sincos(x[3 + 2 * L + o.angle], &s, &c);
      for (int j = 0; j < 3; ++j) {
        const double r0 = R[0][j], r1 = R[1][j];
        R[0][j] = c * r0 + s * r1;
        R[1][j] = -s * r0 + c * r1;
      }
    }
    const double* pp = out + 3 * o.parent;
    const double p0 = pp[0], p1 = pp[1], p2 = pp[2];
    double d[3];
    if (o.flags & 8) {
      for (int i = 0; i < 3; ++i) d[i] = R[i][0] * o.off[0] + R[i][1] * o.off[1] + R[i][2] * o.off[2];
    } else {
      for (int i = 0; i < 3; ++i) d[i] = R[0][i] * o.off[0] + R[1][i] * o.off[1] + R[2][i] * o.off[2];
    }
    double* pc = out + 3 * o.child;
    pc[0] = p0 + d[0];
    pc[1] = p1 + d[1];
    pc[2] = p2 + d[2];
  }
}
}  // namespace acino

extern "C" int acino_skeleton_fk(const double* d_q, int64_t n_frames, int n_angles, int n_pose,
                                 const acino_skel_op* h_ops, int n_ops, double* d_pos, void* stream) {
  using namespace acino;
  ACINO_REQUIRE(n_frames >= 0, "n_frames");
  ACINO_REQUIRE(n_angles >= 1 && n_pose >= 1 && n_pose <= ACINO_SKEL_MAX_OPS + 1, "n_angles, n_pose");
  ACINO_REQUIRE(n_ops >= 0 && n_ops <= ACINO_SKEL_MAX_OPS, "n_ops <= ACINO_SKEL_MAX_OPS");
  if (n_frames == 0) return ACINO_OK;
  ACINO_REQUIRE(d_q && d_pos && (h_ops || n_ops == 0), "null buffer");
  SkelProgram P;
  memset(&P, 0, sizeof(P));
  P.n_ops = n_ops;
  P.n_pose = n_pose;
  P.n_angles = n_angles;
  for (int k = 0; k < n_ops; ++k) {
    const acino_skel_op& o = h_ops[k];
    ACINO_REQUIRE(o.child >= 0 && o.child < n_pose && o.parent >= 0 && o.parent < n_pose, "op slot out of range");
    ACINO_REQUIRE(o.angle >= 0 && o.angle < n_angles, "op angle index out of range");
    P.op[k] = o;
  }
  hipLaunchKernelGGL(k_skeleton_fk, dim3((unsigned)((n_frames + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_q,
                     n_frames, P, d_pos);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}

// ---- initial guess of the FTE solve from the per-frame triangulation, on the device --------------------------------
// (acinoset_amd.fte.triangulation_init: root position = mean of the triangulated eyes and nose, yaw = unwrapped heading of
//  neck_base -> nose, both linearly interpolated over the frames that lack them and held flat at the ends - the same
//  arithmetic as the numpy / torch forms, one launch instead of ~60 small ones and three host synchronisations.)
namespace acino {
constexpr int TI_T = 1024;
// inclusive scan over the workgroup's 1024 values (one per thread); OP: 0 max, 1 min, 2 sum.  carry joins from the left
// (OP 0, 2) or from the right (OP 1, reverse order handled by the caller's indexing).
template <typename T, int OP>
__device__ __forceinline__ T ti_op(T a, T b) {
  return OP == 0 ? (a > b ? a : b) : (OP == 1 ? (a < b ? a : b) : a + b);
}
template <typename T, int OP>
__device__ __forceinline__ T ti_block_scan(T v, T ident, T* sh /*[16]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int off = 1; off < 64; off <<= 1) {
    const T o = __shfl_up(v, off, 64);
    if (lane >= off) v = ti_op<T, OP>(v, o);
  }
  __syncthreads();                                       // (sh is reused between calls)
  if (lane == 63) sh[wave] = v;
  __syncthreads();
  T pre = ident;
  for (int w = 0; w < wave; ++w) pre = ti_op<T, OP>(pre, sh[w]);
  return ti_op<T, OP>(pre, v);
}

__global__ void __launch_bounds__(TI_T)
k_tri_init(const double* __restrict__ tri, int n_frames, int n_markers, double* __restrict__ xa, int n_active, int psi_col,
           int* __restrict__ P, int* __restrict__ Nx, double* __restrict__ val, double* __restrict__ cum, int* __restrict__ flag) {
  __shared__ int shi[16];
  __shared__ double shd[16];
  __shared__ int s_carry_i;
  __shared__ double s_carry_d;
  const int tid = threadIdx.x, N = n_frames;
  const double TWO_PI = 6.283185307179586476925286766559;
  // every other active state starts at 0
  for (long e = tid; e < (long)N * n_active; e += TI_T) xa[e] = 0.0;
  __syncthreads();
  for (int ch = 0; ch < 4; ++ch) {
    // value and validity of the channel per frame: 0..2 head coordinate (mean of the finite ones among markers 0, 1, 2),
    // 3 heading atan2 of marker 2 - marker 3 (all three coordinates finite)
    for (int n = tid; n < N; n += TI_T) {
      const double* t = tri + (size_t)n * n_markers * 3;
      double v = 0.0;
      bool ok;
      if (ch < 3) {
        double s = 0.0;
        int cnt = 0;
        for (int m = 0; m < 3; ++m) {
          const double x = t[3 * m + ch];
          if (!(x != x)) {                               // (torch.nanmean: NaN is skipped, inf is not)
            s += x;
            ++cnt;
          }
        }
        v = cnt ? s / cnt : 0.0;
        ok = cnt > 0 && m_finite(v);
      } else {
        const double fx = t[6] - t[9], fy = t[7] - t[10], fz = t[8] - t[11];
        ok = m_finite(fx) && m_finite(fy) && m_finite(fz);
        v = ok ? atan2(fy, fx) : 0.0;
      }
      val[n] = v;
      P[n] = ok ? n : -1;
      Nx[n] = ok ? n : 0x7fffffff;
    }
    __syncthreads();
    // P[n] <- last valid frame <= n  (forward max-scan, tile by tile with a carry)
    if (tid == 0) s_carry_i = -1;
    __syncthreads();
    for (int base = 0; base < N; base += TI_T) {
      const int n = base + tid;
      int v = n < N ? P[n] : -1;
      v = ti_block_scan<int, 0>(v, -1, shi);
      const int c = s_carry_i;
      v = v > c ? v : c;
      if (n < N) P[n] = v;
      __syncthreads();
      if (tid == TI_T - 1) s_carry_i = v;
      __syncthreads();
    }
    // Nx[n] <- first valid frame >= n  (backward min-scan: the tiles from the right, thread order reversed)
    if (tid == 0) s_carry_i = 0x7fffffff;
    __syncthreads();
    for (int base = 0; base < N; base += TI_T) {
      const int n = N - 1 - (base + tid);
      int v = n >= 0 ? Nx[n] : 0x7fffffff;
      v = ti_block_scan<int, 1>(v, 0x7fffffff, shi);
      const int c = s_carry_i;
      v = v < c ? v : c;
      if (n >= 0) Nx[n] = v;
      __syncthreads();
      if (tid == TI_T - 1) s_carry_i = v;
      __syncthreads();
    }
    if (ch == 3) {
      // np.unwrap over the valid frames: k = round((a_i - a_prev) / 2 pi) per valid frame after the first, a_i -= 2 pi cumsum(k)
      if (tid == 0) s_carry_d = 0.0;
      __syncthreads();
      for (int base = 0; base < N; base += TI_T) {
        const int n = base + tid;
        double k = 0.0;
        if (n < N && n > 0 && P[n] == n) {
          const int pv = P[n - 1];
          if (pv >= 0) k = rint((val[n] - val[pv]) / TWO_PI);
        }
        k = ti_block_scan<double, 2>(k, 0.0, shd);
        k += s_carry_d;
        if (n < N) cum[n] = k;
        __syncthreads();
        if (tid == TI_T - 1) s_carry_d = k;
        __syncthreads();
      }
      for (int n = tid; n < N; n += TI_T)
        if (P[n] == n) val[n] -= TWO_PI * cum[n];
      __syncthreads();
    }
    // np.interp(arange(N), valid frames, values): b = first valid >= n (the last valid one behind the end), a = the valid frame
    // before b (b itself when there is none), weight clamped to [0, 1]
    const int last = P[N - 1];
    if (last < 0) {
      if (ch < 3 && tid == 0) *flag = 1;                 // no triangulated head marker in the whole sequence
    } else {
      const int col = ch < 3 ? ch : psi_col;
      for (int n = tid; n < N; n += TI_T) {
        int b = Nx[n];
        if (b == 0x7fffffff) b = last;
        int a = b > 0 ? P[b - 1] : -1;
        if (a < 0) a = b;
        double w = 0.0;
        if (b > a) w = fmin(fmax((double)(n - a) / fmax((double)(b - a), 1.0), 0.0), 1.0);
        const double va = val[a], vb = val[b];
        xa[(size_t)n * n_active + col] = va + w * (vb - va);
      }
    }
    __syncthreads();
  }
}
}  // namespace acino

extern "C" {
size_t acino_fte_triangulation_init_scratch_bytes(int64_t n_frames) { return n_frames > 0 ? (size_t)n_frames * 24 + 256 : 0; }
int acino_fte_triangulation_init(const double* d_tri, int64_t n_frames, int n_markers, double* d_xa, int n_active,
                                 int psi_column, void* d_scratch, size_t scratch_bytes, int32_t* d_flag, void* stream) {
  ACINO_REQUIRE(n_frames >= 1 && n_frames < (1ll << 30), "n_frames");
  ACINO_REQUIRE(n_markers >= 4 && n_active >= 4 && psi_column >= 3 && psi_column < n_active, "markers / states");
  ACINO_REQUIRE(d_tri && d_xa && d_scratch && d_flag, "null buffer");
  ACINO_REQUIRE(((uintptr_t)d_scratch & 7) == 0 && scratch_bytes >= acino_fte_triangulation_init_scratch_bytes(n_frames), "scratch");
  const size_t N = (size_t)n_frames;
  double* val = (double*)d_scratch;
  double* cum = val + N;
  int* P = (int*)(cum + N);
  int* Nx = P + N;
  hipLaunchKernelGGL(acino::k_tri_init, dim3(1), dim3(acino::TI_T), 0, (hipStream_t)stream, d_tri, (int)n_frames, n_markers, d_xa,
                     n_active, psi_column, P, Nx, val, cum, d_flag);
  ACINO_LAUNCH_CHECK();
  return ACINO_OK;
}
}  // extern "C"
