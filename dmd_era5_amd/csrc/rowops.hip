// K5 (row mean/std, centre, scale), K6 (delay-shift Gram sum), column scaling,
// and the library's error/version plumbing.  All HBM-bound streaming kernels:
// 16 B per lane along the contiguous (row) axis, fp64 accumulation.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "dmdx_common.h"

static thread_local char g_err[512] = "";

void dmdx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int dmdx_version(void) { return DMDX_VERSION; }
extern "C" const char* dmdx_last_error(void) { return g_err; }

namespace {

// One thread owns R consecutive rows (R = 4: float4 accesses; R = 1: scalar
// fallback for unaligned bases / leading dimensions) and walks the n columns.
template <int R>
__global__ __launch_bounds__(256) void row_center_scale_kernel(float* __restrict__ X, int64_t m,
                                                               int64_t n, int64_t ldx,
                                                               float* __restrict__ mean,
                                                               float* __restrict__ sdev,
                                                               int scale) {
  const int64_t r0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * R;
  if (r0 >= m) return;
  const int nr = (m - r0) < R ? (int)(m - r0) : R;
  float* xp = X + r0;
  typedef float vec __attribute__((ext_vector_type(R)));

  auto ld = [&](int64_t j) -> vec {
    vec v;
    if (nr == R) {
      v = *reinterpret_cast<const vec*>(xp + j * ldx);
    } else {
#pragma unroll
      for (int e = 0; e < R; ++e) v[e] = e < nr ? xp[j * ldx + e] : 0.f;
    }
    return v;
  };
  auto st = [&](int64_t j, vec v) {
    if (nr == R) {
      *reinterpret_cast<vec*>(xp + j * ldx) = v;
    } else {
#pragma unroll
      for (int e = 0; e < R; ++e)
        if (e < nr) xp[j * ldx + e] = v[e];
    }
  };

  double s[R];
#pragma unroll
  for (int e = 0; e < R; ++e) s[e] = 0.0;
#pragma unroll 8
  for (int64_t j = 0; j < n; ++j) {
    vec v = ld(j);
#pragma unroll
    for (int e = 0; e < R; ++e) s[e] += (double)v[e];
  }
  vec mu;
#pragma unroll
  for (int e = 0; e < R; ++e) {
    mu[e] = (float)(s[e] / (double)n);
    if (e < nr) mean[r0 + e] = mu[e];
  }

  if (!scale) {
#pragma unroll 8
    for (int64_t j = 0; j < n; ++j) st(j, ld(j) - mu);
    return;
  }
  // std (ddof 0) of the centred row, as numpy computes it: sqrt(mean((c - mean(c))^2))
  double s1[R], s2[R];
#pragma unroll
  for (int e = 0; e < R; ++e) s1[e] = s2[e] = 0.0;
#pragma unroll 8
  for (int64_t j = 0; j < n; ++j) {
    vec c = ld(j) - mu;
#pragma unroll
    for (int e = 0; e < R; ++e) {
      s1[e] += (double)c[e];
      s2[e] += (double)c[e] * (double)c[e];
    }
  }
  vec sd;
#pragma unroll
  for (int e = 0; e < R; ++e) {
    double mc = s1[e] / (double)n;
    double var = s2[e] / (double)n - mc * mc;
    sd[e] = (float)sqrt(var > 0.0 ? var : 0.0);
    if (e < nr) sdev[r0 + e] = sd[e];
  }
#pragma unroll 8
  for (int64_t j = 0; j < n; ++j) st(j, (ld(j) - mu) / sd);
}

// Aligned fast path of K5.  A workgroup owns 4 TX consecutive rows: TX row lanes (float4 each) x TY
// time lanes that walk the columns j = ty, ty + TY, ...; the partial sums of the time lanes meet
// in LDS (fp64).  <64, 4>: a wave reads 1 KiB of one column per load, m / 256 workgroups (507 per
// 129 780-row block); the one-thread-per-row-quad kernel above has m / 1024 (half of the CUs
// without work on a row block: 1.2 TB/s; this one 3.5-4.7 TB/s; <16, 16> and <32, 8> were
// 5-25 % slower).
// Same arithmetic as above: mean in fp64; std (ddof 0) of the fp32-centred values as numpy
// computes it; x <- (x - mean) [/ std].
template <int TX, int TY>
__global__ __launch_bounds__(256) void row_center_scale_tiled_kernel(float* __restrict__ X, int64_t m,
                                                                     int64_t n, int64_t ldx,
                                                                     float* __restrict__ mean,
                                                                     float* __restrict__ sdev,
                                                                     int scale) {
  static_assert(TX * TY == 256, "one workgroup");
  __shared__ double red[2][TY][TX * 4 + 1];
  const int tx = threadIdx.x & (TX - 1), ty = threadIdx.x / TX;
  const int64_t r0 = ((int64_t)blockIdx.x * TX + tx) * 4;
  const bool live = r0 < m;
  const int nr = live ? ((m - r0) < 4 ? (int)(m - r0) : 4) : 0;
  float* xp = X + (live ? r0 : 0);
  typedef float vec __attribute__((ext_vector_type(4)));
  const vec zero = {0.f, 0.f, 0.f, 0.f};
  auto ld = [&](int64_t j) -> vec {
    if (nr == 4) return *reinterpret_cast<const vec*>(xp + j * ldx);
    vec v = zero;                                // the one ragged quad at the end of the rows
    for (int e = 0; e < nr; ++e) v[e] = xp[j * ldx + e];
    return v;
  };
  auto st = [&](int64_t j, vec v) {
    if (nr == 4) {
      *reinterpret_cast<vec*>(xp + j * ldx) = v;
    } else {
      for (int e = 0; e < nr; ++e) xp[j * ldx + e] = v[e];
    }
  };
  auto all_lanes = [&](double (&a)[4], int slot) {   // sum over the 16 time lanes, result in every lane
#pragma unroll
    for (int e = 0; e < 4; ++e) red[slot][ty][4 * tx + e] = a[e];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < TY; ++k) t += red[slot][k][4 * tx + e];
      a[e] = t;
    }
  };

  double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
  for (int64_t j = ty; j < n; j += TY) {
    const vec v = ld(j);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] += (double)v[e];
  }
  all_lanes(s, 0);
  vec mu;
#pragma unroll
  for (int e = 0; e < 4; ++e) mu[e] = (float)(s[e] / (double)n);
  if (ty == 0)
    for (int e = 0; e < nr; ++e) mean[r0 + e] = mu[e];

  if (!scale) {
#pragma unroll 8
    for (int64_t j = ty; j < n; j += TY) st(j, ld(j) - mu);
    return;
  }
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
  for (int64_t j = ty; j < n; j += TY) {
    const vec c = ld(j) - mu;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s1[e] += (double)c[e];
      s2[e] += (double)c[e] * (double)c[e];
    }
  }
  all_lanes(s1, 1);
  __syncthreads();            // slot 0 is about to be reused
  all_lanes(s2, 0);
  vec sd;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const double mc = s1[e] / (double)n;
    const double var = s2[e] / (double)n - mc * mc;
    sd[e] = (float)sqrt(var > 0.0 ? var : 0.0);
  }
  if (ty == 0)
    for (int e = 0; e < nr; ++e) sdev[r0 + e] = sd[e];
#pragma unroll 8
  for (int64_t j = ty; j < n; j += TY) st(j, (ld(j) - mu) / sd);
}

__global__ __launch_bounds__(256) void delay_shift_sum_kernel(const double* __restrict__ G,
                                                              int64_t nd, int64_t ldg, int d,
                                                              double* __restrict__ Gd,
                                                              int64_t ldgd, float* __restrict__ Gd32,
                                                              int64_t ldgd32) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t j = blockIdx.y;
  if (i >= nd) return;
  double s = 0.0;
  for (int k = 0; k < d; ++k) s += G[(i + k) + (j + k) * ldg];
  Gd[i + j * ldgd] = s;
  if (Gd32) Gd32[i + j * ldgd32] = (float)s;
}

__global__ __launch_bounds__(256) void scale_columns_kernel(float* __restrict__ Y, int64_t m,
                                                            int64_t ldy,
                                                            const float* __restrict__ alpha) {
  const float a = alpha[blockIdx.y];
  float* y = Y + (int64_t)blockIdx.y * ldy;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256)
    y[i] *= a;
}

// Phi[i][j] = exp(alpha_j t_i) (and W = diag(t) Phi), the exponential basis of the optimized-DMD
// fit: the exponent is formed and range-reduced in fp64 whatever the output type -- |alpha| t
// reaches 1.4e4 rad over a year of hourly snapshots, which fp32 carries to 1e-3 rad only --, one
// launch instead of the outer product / exp / cast / scale chain of temporaries
template <typename T>
__global__ __launch_bounds__(256) void exp_basis_kernel(const double* __restrict__ alpha, const double* __restrict__ t,
                                                        int64_t n, int64_t r, T* __restrict__ Phi, T* __restrict__ W) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * r) return;
  const int64_t i = idx / r, j = idx - i * r;
  const double ti = t[i];
  const double e = exp(alpha[2 * j] * ti);
  double sn, cs;
  sincos(alpha[2 * j + 1] * ti, &sn, &cs);
  const double re = e * cs, im = e * sn;
  Phi[2 * idx] = (T)re;
  Phi[2 * idx + 1] = (T)im;
  if (W) {
    W[2 * idx] = (T)(ti * re);
    W[2 * idx + 1] = (T)(ti * im);
  }
}

}  // namespace

extern "C" int dmdx_exp_basis(const double* alpha, const double* t, int64_t n, int64_t r, void* Phi, void* W,
                              int single_precision, void* stream) {
  DMDX_CHECK_ARG(alpha && t && Phi, "exp_basis: null pointer");
  DMDX_CHECK_ARG(n >= 1 && r >= 1 && n * r < (int64_t(1) << 40), "exp_basis: bad shape");
  const dim3 grid((unsigned)((n * r + 255) / 256));
  if (single_precision)
    hipLaunchKernelGGL(exp_basis_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, alpha, t, n, r,
                       reinterpret_cast<float*>(Phi), reinterpret_cast<float*>(W));
  else
    hipLaunchKernelGGL(exp_basis_kernel<double>, grid, dim3(256), 0, (hipStream_t)stream, alpha, t, n, r,
                       reinterpret_cast<double*>(Phi), reinterpret_cast<double*>(W));
  DMDX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dmdx_row_center_scale_f32(float* X, int64_t m, int64_t n, int64_t ldx, float* mean,
                                         float* sdev, int scale, void* stream) {
  DMDX_CHECK_ARG(X && mean, "row_center_scale: null pointer");
  DMDX_CHECK_ARG(!scale || sdev, "row_center_scale: scale requested without a std buffer");
  DMDX_CHECK_ARG(m >= 1 && n >= 1 && ldx >= m, "row_center_scale: bad shape/ld");
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (ldx % 4 == 0) && dmdx_aligned16(X);
  if (vec) {
    dim3 grid((unsigned)((m + 255) / 256));
    hipLaunchKernelGGL((row_center_scale_tiled_kernel<64, 4>), grid, dim3(256), 0, st, X, m, n, ldx, mean, sdev,
                       scale);
  } else {
    dim3 grid((unsigned)((m + 255) / 256));
    hipLaunchKernelGGL(row_center_scale_kernel<1>, grid, dim3(256), 0, st, X, m, n, ldx, mean,
                       sdev, scale);
  }
  DMDX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dmdx_delay_shift_sum_f64(const double* G, int64_t n, int64_t ldg, int d, double* Gd,
                                        int64_t ldgd, float* Gd32, int64_t ldgd32, void* stream) {
  DMDX_CHECK_ARG(G && Gd, "delay_shift_sum: null pointer");
  DMDX_CHECK_ARG(d >= 1 && n >= d && ldg >= n, "delay_shift_sum: bad shape n=%lld d=%d",
                 (long long)n, d);
  const int64_t nd = n - d + 1;
  DMDX_CHECK_ARG(ldgd >= nd && (!Gd32 || ldgd32 >= nd) && nd < 65536 * 32768ll,
                 "delay_shift_sum: bad output ld");
  DMDX_CHECK_ARG(nd <= 65535, "delay_shift_sum: n-d+1 > 65535 not supported");
  dim3 grid((unsigned)((nd + 255) / 256), (unsigned)nd);
  hipLaunchKernelGGL(delay_shift_sum_kernel, grid, dim3(256), 0, (hipStream_t)stream, G, nd, ldg,
                     d, Gd, ldgd, Gd32, ldgd32);
  DMDX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dmdx_scale_columns_f32(float* Y, int64_t m, int64_t l, int64_t ldy,
                                      const float* alpha, void* stream) {
  DMDX_CHECK_ARG(Y && alpha, "scale_columns: null pointer");
  DMDX_CHECK_ARG(m >= 1 && l >= 1 && l <= 65535 && ldy >= m, "scale_columns: bad shape");
  int64_t gx = (m + 255) / 256;
  if (gx > 4096) gx = 4096;
  dim3 grid((unsigned)gx, (unsigned)l);
  hipLaunchKernelGGL(scale_columns_kernel, grid, dim3(256), 0, (hipStream_t)stream, Y, m, ldy,
                     alpha);
  DMDX_LAUNCH_CHECK();
  return 0;
}
