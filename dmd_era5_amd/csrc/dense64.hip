// fp64 pieces of the small dense stage (the n x n eigenproblem of the method of snapshots):
//
// K8  dmdx_symm_skinny_f64:  Y = G Q - shift Q,  G symmetric n x n fp64, Q n x b (b <= a few
//     hundred).  The Chebyshev-filtered subspace iteration of svd.top_eigh is a sequence of these
//     products; rocBLAS dgemm takes ~1.0 ms for it at n = 8760 whatever b is (12 TFLOP/s at
//     b = 77), the fp64 MFMA bound is 0.19 / 0.26 ms at 96 / 128 columns (78.6 TFLOP/s) and one
//     read of G (614 MB) is 0.1 ms.
//
//     Layout.  `v_mfma_f64_16x16x4_f64`: A operand = one f64 per lane (row = lane & 15,
//     k = lane >> 4), B likewise (k = lane >> 4, col = lane & 15), D = 4 f64 per lane
//     (row = (lane >> 4) + 4 reg, col = lane & 15).  A wave owns 32 rows of Y and 32 NG columns:
//     one 16-byte load per lane fetches G[k][i0 + 2r .. + 1] -- by symmetry the (row, k)
//     fragments of rows i0 + 2r (even block) and i0 + 2r + 1 (odd block), 4 k-rows x 256
//     contiguous bytes per wave-load, so G streams from HBM in whole lines with no LDS pass (it
//     has no reuse) -- and Q, which every wave needs, is staged through LDS in 32-row chunks
//     (double-buffered, one barrier per chunk) and read back with one conflict-free
//     `ds_read_b128` per 32-column group and k-step (columns are split even / odd the same way).
//     512-thread workgroups (8 waves x 32 rows), K split over gridDim.y so that the grid is one
//     wave of workgroups; every split writes its own partial tile (no atomics: deterministic) and
//     a second kernel sums the splits and subtracts shift * Q.
//
// K9  dmdx_gemm_tn_f64:  C = A^T B for tall fp64 blocks A (n x b1), B (n x b2), n ~ 10^4, b <= a few
//     hundred: the Grams Q^T Q of the CholeskyQR rounds, the Rayleigh-Ritz matrices S^T (G S) and
//     the block projections Q^T W of the eigen stage -- 18 launches per SVD on a gap-free spectrum,
//     each ~250 us through rocBLAS (a 124 x 124 result from 8760 rows: 2.7e8 flops).  Same MFMA and
//     the same even / odd 16-byte fragment loads as K8, both operands straight from global memory
//     (a row of A and B is 1 KB and is re-read by the 2 x 2 waves of a workgroup from L1); 128 x 128
//     output tiles, K split in chunks of 128 rows over gridDim.x, per-split partial tiles
//     (deterministic) summed by a second kernel.
//
// pack / unpack of the upper triangle of a symmetric fp64 matrix: the Gram all-reduce of the
//     row-sharded path moves n (n + 1) / 2 instead of n^2 doubles (307 instead of 614 MB at
//     n = 8760).
#include "dmdx_common.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int SY_TH = 512;
constexpr int SY_ROWS = 256;  // rows of Y per workgroup: 8 waves x 32
constexpr int SY_KC = 32;     // k rows of Q per LDS chunk

struct SymmParams {
  const double* G;
  const double* Q;
  double* P;  // [nsplit][n][b] partial products
  int64_t ldg, ldq;
  int n, b, c_base, kchunk;
};

template <int NG>
__global__ __launch_bounds__(SY_TH) void symm_skinny_partial_kernel(SymmParams p) {
  constexpr int NC = 32 * NG;
  __shared__ __attribute__((aligned(16))) double sQ[2][SY_KC * NC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int n = p.n, b = p.b;
  const int i0 = blockIdx.x * SY_ROWS + wave * 32;
  const int kbeg = blockIdx.y * p.kchunk;
  const int kend = min(n, kbeg + p.kchunk);
  const int nchunks = (kend - kbeg + SY_KC - 1) / SY_KC;
  const bool active = i0 < n;  // wave-uniform; idle waves still stage Q and meet the barriers
  const int ia = min(i0 + 2 * r, n - 2);

  f64x4 acc[2][NG][2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int f = 0; f < 2; ++f) acc[e][g][f] = f64x4{0.0, 0.0, 0.0, 0.0};

  // Q chunk: SY_KC x NC doubles = 16 NC double2; thread t moves pairs t, t + 512, ... (NG of them)
  f64x2 qreg[NG];
  auto load_q = [&](int kc0) {
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int pi = tid + SY_TH * j;
      const int kr = pi / (NC / 2), col = 2 * (pi - kr * (NC / 2));
      const int k = kc0 + kr, c = p.c_base + col;
      f64x2 v{0.0, 0.0};
      if (k < kend && c < b) v = *reinterpret_cast<const f64x2*>(p.Q + (int64_t)k * p.ldq + c);
      qreg[j] = v;
    }
  };
  auto store_q = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int pi = tid + SY_TH * j;
      *reinterpret_cast<f64x2*>(&sQ[buf][2 * pi]) = qreg[j];
    }
  };
  f64x2 a_cur[SY_KC / 4], a_nxt[SY_KC / 4];
  auto load_a = [&](f64x2* a, int kc0) {
#pragma unroll
    for (int s = 0; s < SY_KC / 4; ++s) {
      const int k = kc0 + 4 * s + q;
      const f64x2 v = *reinterpret_cast<const f64x2*>(p.G + (int64_t)min(k, kend - 1) * p.ldg + ia);
      a[s] = (k < kend) ? v : f64x2{0.0, 0.0};
    }
  };

  load_q(kbeg);
  if (active) load_a(a_cur, kbeg);
  store_q(0);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    const bool more = c + 1 < nchunks;
    if (more) {
      load_q(kbeg + (c + 1) * SY_KC);
      if (active) load_a(a_nxt, kbeg + (c + 1) * SY_KC);
    }
    if (active) {
#pragma unroll
      for (int s = 0; s < SY_KC / 4; ++s) {
        const double* row = &sQ[buf][(4 * s + q) * NC + 2 * r];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const f64x2 bb = *reinterpret_cast<const f64x2*>(row + 32 * g);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            acc[e][g][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[s][e], bb[0], acc[e][g][0], 0, 0, 0);
            acc[e][g][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[s][e], bb[1], acc[e][g][1], 0, 0, 0);
          }
        }
      }
    }
    if (more) {
      store_q(buf ^ 1);
#pragma unroll
      for (int s = 0; s < SY_KC / 4; ++s) a_cur[s] = a_nxt[s];
    }
    __syncthreads();
  }
  if (!active) return;
  double* Pt = p.P + (size_t)blockIdx.y * n * b;
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int col = p.c_base + 32 * g + 2 * r;
      if (col >= b) continue;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = i0 + 2 * (q + 4 * reg) + e;
        if (row < n)
          *reinterpret_cast<f64x2*>(Pt + (size_t)row * b + col) = f64x2{acc[e][g][0][reg], acc[e][g][1][reg]};
      }
    }
}

__global__ __launch_bounds__(256) void symm_skinny_reduce_kernel(const double* __restrict__ P, int nsplit, int64_t nb,
                                                                 int b, const double* __restrict__ Q, int64_t ldq,
                                                                 double shift, double* __restrict__ Y, int64_t ldy) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nb) return;
  const int64_t i = idx / b;
  const int c = (int)(idx - i * b);
  double s = 0.0;
  for (int k = 0; k < nsplit; ++k) s += P[(int64_t)k * nb + idx];
  if (shift != 0.0) s -= shift * Q[i * ldq + c];
  Y[i * ldy + c] = s;
}

int symm_plan(int64_t n, int64_t b, int* nsplit, int* kchunk) {
  const int row_tiles = (int)((n + SY_ROWS - 1) / SY_ROWS);
  int ks = 256 / row_tiles;  // one wave of workgroups on the 256 CUs
  if (ks < 1) ks = 1;
  const int max_ks = (int)((n + 255) / 256);  // at least 256 k-rows per split
  if (ks > max_ks) ks = max_ks;
  if (ks > 64) ks = 64;
  int kc = (int)((n + ks - 1) / ks);
  kc = (kc + SY_KC - 1) / SY_KC * SY_KC;
  ks = (int)((n + kc - 1) / kc);
  *nsplit = ks;
  *kchunk = kc;
  return row_tiles;
}

// ---- K9: C = A^T B, tall fp64 operands ------------------------------------------------------
constexpr int TN_KCH = 128;  // rows of A / B per workgroup

struct Tn64Params {
  const double* A;
  const double* B;
  double* P;  // [nsplit][b1][b2]
  int64_t lda, ldb;
  int n, b1, b2;
};

__global__ __launch_bounds__(256) void gemm_tn64_partial_kernel(Tn64Params p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int c1_0 = blockIdx.y * 128 + wr * 64, c2_0 = blockIdx.z * 128 + wc * 64;
  const int kbeg = blockIdx.x * TN_KCH, kend = min(p.n, kbeg + TN_KCH);
  if (c1_0 >= p.b1 || c2_0 >= p.b2) return;  // wave-uniform; no barriers in this kernel
  int ca[2], cb[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    ca[g] = min(c1_0 + 32 * g + 2 * r, p.b1 - 2);
    cb[g] = min(c2_0 + 32 * g + 2 * r, p.b2 - 2);
  }
  f64x4 acc[2][2][2][2];  // [g][e][g'][f]
#pragma unroll
  for (int i = 0; i < 16; ++i) (&acc[0][0][0][0])[i] = f64x4{0.0, 0.0, 0.0, 0.0};
  for (int k0 = kbeg; k0 < kend; k0 += 4) {
    const int k = k0 + q;
    const bool valid = k < kend;
    const int64_t kk = valid ? k : kend - 1;
    f64x2 a[2], b[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      a[g] = *reinterpret_cast<const f64x2*>(p.A + kk * p.lda + ca[g]);
      b[g] = *reinterpret_cast<const f64x2*>(p.B + kk * p.ldb + cb[g]);
      if (!valid) a[g] = f64x2{0.0, 0.0};
    }
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int f = 0; f < 2; ++f)
            acc[g][e][h][f] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[g][e], b[h][f], acc[g][e][h][f], 0, 0, 0);
  }
  double* Pt = p.P + (size_t)blockIdx.x * p.b1 * p.b2;
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c2 = c2_0 + 32 * h + 2 * r;
        if (c2 >= p.b2) continue;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int c1 = c1_0 + 32 * g + 2 * (q + 4 * reg) + e;
          if (c1 < p.b1)
            *reinterpret_cast<f64x2*>(Pt + (size_t)c1 * p.b2 + c2) = f64x2{acc[g][e][h][0][reg], acc[g][e][h][1][reg]};
        }
      }
}

__global__ __launch_bounds__(256) void gemm_tn64_reduce_kernel(const double* __restrict__ P, int nsplit, int64_t nb,
                                                               int b2, double* __restrict__ C, int64_t ldc) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nb) return;
  double s = 0.0;
  for (int k = 0; k < nsplit; ++k) s += P[(int64_t)k * nb + idx];
  const int64_t i = idx / b2;
  C[i * ldc + (idx - i * b2)] = s;
}

// ---- upper triangle <-> packed, row by row: packed[i (2n - i + 1) / 2 + (j - i)] = A[i][j], j >= i
__global__ __launch_bounds__(256) void pack_triu_kernel(const double* __restrict__ A, int64_t n, int64_t lda,
                                                        double* __restrict__ packed) {
  const int64_t i = blockIdx.y;
  const int64_t j = i + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  packed[i * (2 * n - i + 1) / 2 + (j - i)] = A[i * lda + j];
}

__global__ __launch_bounds__(256) void unpack_triu_kernel(const double* __restrict__ packed, int64_t n, int64_t lda,
                                                          double* __restrict__ A) {
  // every element of A from its source in the packed upper triangle (coalesced writes of row i)
  const int64_t i = blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int64_t a = i < j ? i : j, c = i < j ? j : i;
  A[i * lda + j] = packed[a * (2 * n - a + 1) / 2 + (c - a)];
}

}  // namespace

extern "C" size_t dmdx_symm_skinny_workspace_bytes(int64_t n, int64_t b) {
  if (n < 2 || b < 2) return 0;
  int ks, kc;
  symm_plan(n, b, &ks, &kc);
  return (size_t)ks * (size_t)n * (size_t)b * sizeof(double);
}

extern "C" int dmdx_symm_skinny_f64(const double* G, int64_t n, int64_t ldg, const double* Q, int64_t ldq, int64_t b,
                                    double shift, double* Y, int64_t ldy, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  DMDX_CHECK_ARG(G && Q && Y, "symm_skinny: null pointer");
  DMDX_CHECK_ARG(n >= 2 && b >= 2 && n < (int64_t(1) << 30) && b <= 4096, "symm_skinny: bad shape n=%lld b=%lld",
                 (long long)n, (long long)b);
  DMDX_CHECK_ARG(n % 2 == 0 && b % 2 == 0 && ldg % 2 == 0 && ldq % 2 == 0,
                 "symm_skinny: n, b, ldg, ldq must be even (16-byte fragment loads)");
  DMDX_CHECK_ARG(ldg >= n && ldq >= b && ldy >= b, "symm_skinny: leading dimension too small");
  DMDX_CHECK_ARG(dmdx_aligned16(G) && dmdx_aligned16(Q) && dmdx_aligned16(workspace),
                 "symm_skinny: G, Q and the workspace must be 16-byte aligned");
  DMDX_CHECK_ARG(Y != Q, "symm_skinny: Y must not alias Q");
  int ks, kc;
  const int row_tiles = symm_plan(n, b, &ks, &kc);
  const size_t need = (size_t)ks * (size_t)n * (size_t)b * sizeof(double);
  if (workspace == nullptr || workspace_bytes < need) {
    dmdx_set_error("symm_skinny: workspace %zu bytes < required %zu", workspace_bytes, need);
    return DMDX_E_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  SymmParams p{G, Q, reinterpret_cast<double*>(workspace), ldg, ldq, (int)n, (int)b, 0, kc};
  const dim3 grid((unsigned)row_tiles, (unsigned)ks);
  for (int c0 = 0; c0 < b; c0 += 128) {
    p.c_base = c0;
    const int ng = (int)((((b - c0 < 128) ? b - c0 : 128) + 31) / 32);
    switch (ng) {
      case 1: hipLaunchKernelGGL(symm_skinny_partial_kernel<1>, grid, dim3(SY_TH), 0, st, p); break;
      case 2: hipLaunchKernelGGL(symm_skinny_partial_kernel<2>, grid, dim3(SY_TH), 0, st, p); break;
      case 3: hipLaunchKernelGGL(symm_skinny_partial_kernel<3>, grid, dim3(SY_TH), 0, st, p); break;
      default: hipLaunchKernelGGL(symm_skinny_partial_kernel<4>, grid, dim3(SY_TH), 0, st, p); break;
    }
    DMDX_LAUNCH_CHECK();
  }
  const int64_t nb = n * b;
  hipLaunchKernelGGL(symm_skinny_reduce_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, p.P, ks, nb,
                     (int)b, Q, ldq, shift, Y, ldy);
  DMDX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dmdx_pack_triu_f64(const double* A, int64_t n, int64_t lda, double* packed, void* stream) {
  DMDX_CHECK_ARG(A && packed, "pack_triu: null pointer");
  DMDX_CHECK_ARG(n >= 1 && n <= 65535 && lda >= n, "pack_triu: bad shape");
  hipLaunchKernelGGL(pack_triu_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0,
                     (hipStream_t)stream, A, n, lda, packed);
  DMDX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dmdx_unpack_triu_f64(const double* packed, int64_t n, double* A, int64_t lda, void* stream) {
  DMDX_CHECK_ARG(A && packed, "unpack_triu: null pointer");
  DMDX_CHECK_ARG(n >= 1 && n <= 65535 && lda >= n, "unpack_triu: bad shape");
  hipLaunchKernelGGL(unpack_triu_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0,
                     (hipStream_t)stream, packed, n, lda, A);
  DMDX_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t dmdx_gemm_tn_f64_workspace_bytes(int64_t n, int64_t b1, int64_t b2) {
  if (n < 1 || b1 < 2 || b2 < 2) return 0;
  const size_t ks = (size_t)((n + TN_KCH - 1) / TN_KCH);
  return ks * (size_t)b1 * (size_t)b2 * sizeof(double);
}

extern "C" int dmdx_gemm_tn_f64(const double* A, int64_t lda, const double* B, int64_t ldb, int64_t n, int64_t b1,
                                int64_t b2, double* Cm, int64_t ldc, void* workspace, size_t workspace_bytes,
                                void* stream) {
  DMDX_CHECK_ARG(A && B && Cm, "gemm_tn_f64: null pointer");
  DMDX_CHECK_ARG(n >= 1 && n < (int64_t(1) << 30) && b1 >= 2 && b2 >= 2 && b1 <= 8192 && b2 <= 8192,
                 "gemm_tn_f64: bad shape n=%lld b1=%lld b2=%lld", (long long)n, (long long)b1, (long long)b2);
  DMDX_CHECK_ARG(b1 % 2 == 0 && b2 % 2 == 0 && lda % 2 == 0 && ldb % 2 == 0,
                 "gemm_tn_f64: b1, b2, lda, ldb must be even (16-byte fragment loads)");
  DMDX_CHECK_ARG(lda >= b1 && ldb >= b2 && ldc >= b2, "gemm_tn_f64: leading dimension too small");
  DMDX_CHECK_ARG(dmdx_aligned16(A) && dmdx_aligned16(B) && dmdx_aligned16(workspace),
                 "gemm_tn_f64: A, B and the workspace must be 16-byte aligned");
  const size_t need = dmdx_gemm_tn_f64_workspace_bytes(n, b1, b2);
  if (workspace == nullptr || workspace_bytes < need) {
    dmdx_set_error("gemm_tn_f64: workspace %zu bytes < required %zu", workspace_bytes, need);
    return DMDX_E_WORKSPACE;
  }
  const int ks = (int)((n + TN_KCH - 1) / TN_KCH);
  DMDX_CHECK_ARG(ks <= 65535 * 32, "gemm_tn_f64: n too large");
  hipStream_t st = (hipStream_t)stream;
  Tn64Params p{A, B, reinterpret_cast<double*>(workspace), lda, ldb, (int)n, (int)b1, (int)b2};
  const dim3 grid((unsigned)ks, (unsigned)((b1 + 127) / 128), (unsigned)((b2 + 127) / 128));
  hipLaunchKernelGGL(gemm_tn64_partial_kernel, grid, dim3(256), 0, st, p);
  DMDX_LAUNCH_CHECK();
  const int64_t nb = b1 * b2;
  hipLaunchKernelGGL(gemm_tn64_reduce_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, p.P, ks, nb, (int)b2,
                     Cm, ldc);
  DMDX_LAUNCH_CHECK();
  return 0;
}
