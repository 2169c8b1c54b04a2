// K10: Cholesky factorisation + triangular inverse of a small fp64 matrix (n <= 1024) in ONE launch,
// K11: the tall product Y = Q Mt^T that applies the inverse factor -- the dense steps of every
// CholeskyQR round (svd._orth / _chol_rinv), of the Rayleigh-Ritz solves (svd._eigh_desc: Cholesky
// factor -> one-sided Jacobi) and of the graded refinement (svd._graded_eigh).  They replace
// rocSOLVER potrf / trtri and rocBLAS trsm / dgemm there: 73-78 launch-bound library launches per
// randomized SVD (potf2_kernel_small + trtri + trsm pieces), 10 ms of the 65 ms rank-200 eigen
// stage on gap-free spectra, and 0.3-0.8 s of first-call library start-up in a CLI whose every run is
// a first call.  (The part of np.linalg.svd, era5_svd.py:251, and of sklearn's LU / QR
// normalisers, era5_svd.py:258 / extmath.py:349-355, that is left once X is reduced to Gram matrices.)
//
// K10 chol_inv_kernel.  Left-looking blocked Cholesky, 32 x 32 blocks, one workgroup per block row
// (cyclic when there are more rows than workgroups), ONE grid barrier per block column:
//   panel p:  every workgroup forms D = A[p][p] + shift I - sum_{q<p} L[p][q] L[p][q]^T itself (p small
//             block products: cheaper than a second barrier), one wave factors it in LDS (lane =
//             row, wave-synchronous) and inverts the factor; the owner of block row I > p forms
//             L[I][p] = (A[I][p] - sum_{q<p} L[I][q] L[p][q]^T) L[p][p]^-T from its OWN earlier blocks and
//             block row p (visible since the barriers of the earlier panels).
//   then X = L^-1 block column by block column (workgroup J: X[J][J] = L[J][J]^-1,
//             X[I][J] = -L[I][I]^-1 sum_{J<=K<I} L[I][K] X[K][J]), no further barrier.
// Blocks move between workgroups with agent-scope (sc1) loads / stores and the counter barrier of
// K7L (jacobi_svd.hip): every launch must be co-resident (<= 32 workgroups), a bounded spin reports
// -1 instead of hanging.  A non-positive or non-finite pivot sets the status (index + 1 of the
// first one); its column is dropped (pivot 1, zeros below) so that the outputs stay finite: the
// caller shifts and retries.
// info[0] = status, info[1] / info[2] = min / max of diag(L).
//
// K11 gemm_nt64_kernel.  Y (n x b2) = Q (n x b1) Mt^T, Mt (b2 x b1) row-major: Q L^-T of CholeskyQR
// (Mt = L^-1, lower triangular) and S Z of the Rayleigh-Ritz steps (Mt = Z^T, as the Jacobi kernel
// returns it).  v_mfma_f64_16x16x4_f64, both operands straight from global memory with 16-byte
// loads (lane (i, k): Q[row i][k0 + 2 k .. + 1] and Mt[col i][k0 + 2 k .. + 1]: two k-steps per load
// with the same k permutation on both sides), a wave owns 16 rows x 64 columns, no LDS, no barrier.
#include "dmdx_common.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int CB = 32;        // block edge
constexpr int CLD = CB + 2;   // LDS row stride (doubles): 34 makes the MFMA fragment reads (16 rows x 4 k) conflict-free
constexpr int CTH = 256;
constexpr int CMAXN = 1024;
constexpr int CSPIN_LIMIT = 1 << 22;

struct CholParams {
  const double* A;   // n x n symmetric (only the lower triangle is read), row-major
  int64_t lda;
  double shift;
  double* L;         // n x n, lower triangular factor (pre-zeroed by the host)
  int64_t ldl;
  double* X;         // n x n, L^-1 (lower; pre-zeroed), nullable
  int64_t ldx;
  double* info;      // [0] status, [1] min diag(L), [2] max diag(L)
  unsigned* bar;     // [0] barrier counter, [1] timeout flag
  double* Dinv;      // [nbk][32][32] inverses of the diagonal blocks
  int n, nbk;
};

__device__ __forceinline__ double ldc64(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stc64(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// all waves call; false after a timeout (the launch was not co-resident)
__device__ bool chol_grid_barrier(unsigned* bar, unsigned target, int* ok) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have left the CU
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int good = 1, spins = 0;
    while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > CSPIN_LIMIT || __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        __hip_atomic_store(&bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        good = 0;
        break;
      }
    }
    *ok = good;
  }
  __syncthreads();
  return *ok != 0;
}

// thread t of 256 <-> row t >> 3, columns 4 (t & 7) .. + 3 of a 32 x 32 block.
// Block (br, bc) of a row-major matrix: fetched into 4 registers (agent-scope loads: the data was
// written by another workgroup, or by this one through sc1 stores), put into LDS [32][33] later --
// the fetch of the next block is in flight while the current one is multiplied.
__device__ __forceinline__ void fetch_blk(double v[4], const double* M, int64_t ld, int br, int bc, int n, int tid) {
  const int i = tid >> 3, j0 = 4 * (tid & 7);
  const int r = CB * br + i;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = CB * bc + j0 + e;
    v[e] = (r < n && c < n) ? ldc64(M + (int64_t)r * ld + c) : 0.0;
  }
}
__device__ __forceinline__ void put_blk(double* dst, const double v[4], int tid) {
  const int i = tid >> 3, j0 = 4 * (tid & 7);
#pragma unroll
  for (int e = 0; e < 4; ++e) dst[i * CLD + j0 + e] = v[e];
}

// 32 x 32 x 32 block products on v_mfma_f64_16x16x4_f64: wave (wr, wc) owns the 16 x 16 quadrant
// (16 wr .., 16 wc ..) of the result, D layout row = (lane >> 4) + 4 reg, col = lane & 15.
// acc -= A B^T:  A operand lane (i, k) = sA[16 wr + i][4 s + k], B operand lane (k, j) = sB[16 wc + j][4 s + k]
__device__ __forceinline__ void mfma_nt_sub(f64x4& acc, const double* sA, const double* sB, int wr, int wc, int li, int lk) {
#pragma unroll
  for (int s = 0; s < CB / 4; ++s)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-sA[(16 * wr + li) * CLD + 4 * s + lk], sB[(16 * wc + li) * CLD + 4 * s + lk], acc, 0, 0, 0);
}
// acc += sign A B:  B operand lane (k, j) = sB[4 s + k][16 wc + j]
__device__ __forceinline__ void mfma_nn(f64x4& acc, const double* sA, const double* sB, int wr, int wc, int li, int lk, double sign) {
#pragma unroll
  for (int s = 0; s < CB / 4; ++s)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * sA[(16 * wr + li) * CLD + 4 * s + lk], sB[(4 * s + lk) * CLD + 16 * wc + li], acc, 0, 0, 0);
}
// a quadrant in MFMA layout -> LDS block
__device__ __forceinline__ void put_acc(double* dst, const f64x4& acc, int wr, int wc, int li, int lk) {
#pragma unroll
  for (int g = 0; g < 4; ++g) dst[(16 * wr + lk + 4 * g) * CLD + 16 * wc + li] = acc[g];
}

// 1 / sqrt(d) to full double precision: v_rsq_f64 (~2^-26) + two Newton steps (a correctly rounded
// sqrt + divide pair costs ~3x as many dependent cycles on the critical path of every column)
__device__ __forceinline__ double rsqrt64(double d) {
  double y = __builtin_amdgcn_rsq(d);
  const double h = 0.5 * d;
  y = y * fma(-h * y, y, 1.5);
  y = y * fma(-h * y, y, 1.5);
  return y;
}

// The whole workgroup: Cholesky factor of a 32 x 32 block AND the inverse of the factor, one barrier
// per column.  Thread (i, jq) holds the elements (i, 4 jq .. 4 jq + 3) of the block in a[] (lower
// triangle meaningful) and of X (starts as the identity) in registers; per column c the current
// column c and the current row c of X travel through a double-buffered 64-double LDS line:
//   piv = sqrt(a_cc);  l_ic = a_ic / piv;  a_ij -= l_ic l_jc (j > c);  X[c] /= piv;  X[i] -= l_ic X[c] (i > c).
// Leaves L (zeros above the diagonal) in sD, L^-1 in sDi.  Returns the index + 1 of the first
// non-positive / non-finite pivot (replaced by 1), else 0 (the same value in every thread).
__device__ __forceinline__ int factor_diag_wg(double a[4], double* sD, double* sDi, double* buf, int i, int jq) {
  double x[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) x[e] = (i == 4 * jq + e) ? 1.0 : 0.0;
  int bad = 0;
  for (int cq = 0; cq < CB / 4; ++cq) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = 4 * cq + e;
      double* line = buf + 64 * (c & 1);
      if (jq == cq) line[i] = a[e];
      if (i == c) {
#pragma unroll
        for (int f = 0; f < 4; ++f) line[32 + 4 * jq + f] = x[f];
      }
      __syncthreads();
      double d = line[c];
      // a non-positive / non-finite pivot: recorded, and the column is DROPPED (pivot 1, zeros below
      // it), so that what follows is the factorisation of the remaining principal submatrix and
      // stays bounded -- a pivot of 1 under the original column would let the entries grow without limit
      const bool badp = !(d > 0.0) || !(d < 1e300);
      if (badp) {
        if (bad == 0) bad = c + 1;
        d = 1.0;
      }
      const double rinv = rsqrt64(d);
      const double li = (i > c && !badp) ? line[i] * rinv : 0.0;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int col = 4 * jq + f;
        const double lj = (col > c && !badp) ? line[col] * rinv : 0.0;
        a[f] = fma(-li, lj, a[f]);
        const double xr = line[32 + col] * rinv;      // row c of X, final
        x[f] = (i == c) ? xr : fma(-li, xr, x[f]);    // (li = 0 for the rows above c: unchanged)
      }
      if (jq == cq) a[e] = (i > c) ? li : (i == c ? d * rinv : 0.0);
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sD[i * CLD + 4 * jq + e] = a[e];
    sDi[i * CLD + 4 * jq + e] = x[e];
  }
  __syncthreads();
  return bad;
}

__global__ __launch_bounds__(CTH) void chol_inv_kernel(CholParams p) {
  __shared__ double sA[CB * CLD], sB[CB * CLD], sD[CB * CLD], sDi[CB * CLD], colv[128];
  __shared__ int s_ok;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = tid >> 3, jq = tid & 7;             // thread layout: element (i, 4 jq .. 4 jq + 3) of a block
  const int wr = wave >> 1, wc = wave & 1, li = lane & 15, lk = lane >> 4;   // MFMA layout
  const int n = p.n, nbk = p.nbk, G = gridDim.x, wg = blockIdx.x;
  unsigned epoch = 0;
  bool alive = true;
  int first_bad = 0;             // (tracked identically by every workgroup: all of them factor every D)
  double dmin = 1e300, dmax = 0.0;

  for (int pnl = 0; pnl < nbk && alive; ++pnl) {
    const int nb = min(CB, n - CB * pnl);
    // this workgroup's first block row below the panel (G = nbk: its only one)
    int I0 = wg;
    while (I0 <= pnl) I0 += G;
    const bool has_row = I0 < nbk;
    // ---- D = A[p][p] + shift I - sum_q L[p][q] L[p][q]^T   and   T = A[I0][p] - sum_q L[I0][q] L[p][q]^T:
    // the A blocks in thread layout, the sums on the MFMA (its layout), joined through LDS below
    double aD[4], aT[4];
    {
      const int r = CB * pnl + i, rt = CB * I0 + i;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = CB * pnl + 4 * jq + e;
        double v = 0.0;
        if (r < n && c < n) v = (c <= r) ? p.A[(int64_t)r * p.lda + c] : p.A[(int64_t)c * p.lda + r];
        if (r == c) v = (r < n) ? v + p.shift : 1.0;   // rows past n: identity rows
        aD[e] = v;
        aT[e] = (has_row && rt < n && c < n) ? p.A[(int64_t)rt * p.lda + c] : 0.0;
      }
    }
    f64x4 accD = {0.0, 0.0, 0.0, 0.0}, accT = {0.0, 0.0, 0.0, 0.0};
    {
      double nb_[4], na_[4] = {0.0, 0.0, 0.0, 0.0};
      if (pnl > 0) {
        fetch_blk(nb_, p.L, p.ldl, pnl, 0, n, tid);
        if (has_row) fetch_blk(na_, p.L, p.ldl, I0, 0, n, tid);
      }
      for (int q = 0; q < pnl; ++q) {
        __syncthreads();
        put_blk(sB, nb_, tid);
        if (has_row) put_blk(sA, na_, tid);
        __syncthreads();
        if (q + 1 < pnl) {
          fetch_blk(nb_, p.L, p.ldl, pnl, q + 1, n, tid);
          if (has_row) fetch_blk(na_, p.L, p.ldl, I0, q + 1, n, tid);
        }
        mfma_nt_sub(accD, sB, sB, wr, wc, li, lk);
        if (has_row) mfma_nt_sub(accT, sA, sB, wr, wc, li, lk);
      }
    }
    __syncthreads();
    put_acc(sD, accD, wr, wc, li, lk);
    put_acc(sA, accT, wr, wc, li, lk);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      aD[e] += sD[i * CLD + 4 * jq + e];
      aT[e] += sA[i * CLD + 4 * jq + e];
    }
    __syncthreads();
    {
      const int bad = factor_diag_wg(aD, sD, sDi, colv, i, jq);
      if (first_bad == 0 && bad != 0 && CB * pnl + bad <= n) first_bad = CB * pnl + bad;
    }
    for (int c = 0; c < nb; ++c) {   // pivots of this block (every workgroup keeps the same record)
      const double d = sD[c * CLD + c];
      dmin = fmin(dmin, d);
      dmax = fmax(dmax, d);
    }
    // ---- the owner of block row p stores L[p][p] and its inverse
    if (pnl % G == wg) {
      const int r = CB * pnl + i;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = CB * pnl + 4 * jq + e;
        if (r < n && c < n) stc64(p.L + (int64_t)r * p.ldl + c, sD[i * CLD + 4 * jq + e]);
        stc64(p.Dinv + ((size_t)pnl * CB + i) * CB + 4 * jq + e, sDi[i * CLD + 4 * jq + e]);
      }
    }
    // ---- owned block rows I > p: L[I][p] = T L[p][p]^-T
    for (int I = I0; I < nbk; I += G) {
      if (I != I0) {   // (G < nbk only: further rows of this workgroup, their T formed here)
        const int rt = CB * I + i;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = CB * pnl + 4 * jq + e;
          aT[e] = (rt < n && c < n) ? p.A[(int64_t)rt * p.lda + c] : 0.0;
        }
        f64x4 acc2 = {0.0, 0.0, 0.0, 0.0};
        for (int q = 0; q < pnl; ++q) {
          double va[4], vb[4];
          fetch_blk(va, p.L, p.ldl, I, q, n, tid);
          fetch_blk(vb, p.L, p.ldl, pnl, q, n, tid);
          __syncthreads();
          put_blk(sA, va, tid);
          put_blk(sB, vb, tid);
          __syncthreads();
          mfma_nt_sub(acc2, sA, sB, wr, wc, li, lk);
        }
        __syncthreads();
        put_acc(sA, acc2, wr, wc, li, lk);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) aT[e] += sA[i * CLD + 4 * jq + e];
      }
      __syncthreads();
      put_blk(sA, aT, tid);
      __syncthreads();
      // T (L11^-1)^T = -(−T) ...: acc = 0 - (-T) Dinv^T through the subtracting form with a negated operand
      f64x4 accL = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < CB / 4; ++s)
        accL = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[(16 * wr + li) * CLD + 4 * s + lk], sDi[(16 * wc + li) * CLD + 4 * s + lk], accL, 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int r = CB * I + 16 * wr + lk + 4 * g, c = CB * pnl + 16 * wc + li;
        if (r < n && c < n) stc64(p.L + (int64_t)r * p.ldl + c, accL[g]);
      }
    }
    ++epoch;
    if (G == 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    } else {
      alive = chol_grid_barrier(p.bar, epoch * (unsigned)G, &s_ok);
    }
  }
  if (wg == 0 && tid == 0) {
    p.info[0] = alive ? (double)first_bad : -1.0;
    p.info[1] = dmin;
    p.info[2] = dmax;
  }
  if (!alive || p.X == nullptr) return;

  // ---- X = L^-1, block column J by workgroup J % G: X[J][J] = L[J][J]^-1,
  // X[I][J] = -L[I][I]^-1 sum_{J <= K < I} L[I][K] X[K][J]
  for (int J = wg; J < nbk; J += G) {
    double v[4];
    fetch_blk(v, p.Dinv + (size_t)J * CB * CB, CB, 0, 0, CB, tid);
    {
      const int r = CB * J + i;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = CB * J + 4 * jq + e;
        if (r < n && c < n) stc64(p.X + (int64_t)r * p.ldx + c, v[e]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int I = J + 1; I < nbk; ++I) {
      f64x4 accS = {0.0, 0.0, 0.0, 0.0};
      double va[4], vb[4];
      fetch_blk(va, p.L, p.ldl, I, J, n, tid);
      __syncthreads();                                      // (this workgroup's X stores are out: vmcnt above)
      fetch_blk(vb, p.X, p.ldx, J, J, n, tid);
      for (int K = J; K < I; ++K) {
        __syncthreads();
        put_blk(sA, va, tid);
        put_blk(sB, vb, tid);
        __syncthreads();
        if (K + 1 < I) {
          fetch_blk(va, p.L, p.ldl, I, K + 1, n, tid);
          fetch_blk(vb, p.X, p.ldx, K + 1, J, n, tid);
        }
        mfma_nn(accS, sA, sB, wr, wc, li, lk, 1.0);
      }
      fetch_blk(va, p.Dinv + (size_t)I * CB * CB, CB, 0, 0, CB, tid);
      __syncthreads();
      put_acc(sB, accS, wr, wc, li, lk);
      put_blk(sA, va, tid);
      __syncthreads();
      f64x4 accX = {0.0, 0.0, 0.0, 0.0};
      mfma_nn(accX, sA, sB, wr, wc, li, lk, -1.0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int r = CB * I + 16 * wr + lk + 4 * g, c = CB * J + 16 * wc + li;
        if (r < n && c < n) stc64(p.X + (int64_t)r * p.ldx + c, accX[g]);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next row reads this block back
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K11: Y = Q Mt^T
struct NtParams {
  const double* Q;
  const double* Mt;
  double* Y;
  int64_t ldq, ldm, ldy;
  int n, b1, b2;
};

constexpr int NT_NC = 4;     // 16-column MFMA blocks per wave
__global__ __launch_bounds__(256) void gemm_nt64_kernel(NtParams p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int row0 = (blockIdx.x * 4 + wave) * 16;
  const int col0 = blockIdx.y * (16 * NT_NC);
  if (row0 >= p.n) return;
  const int r = min(row0 + li, p.n - 1);
  const double* q = p.Q + (int64_t)r * p.ldq + 2 * lk;
  const double* m[NT_NC];
#pragma unroll
  for (int c = 0; c < NT_NC; ++c) m[c] = p.Mt + (int64_t)min(col0 + 16 * c + li, p.b2 - 1) * p.ldm + 2 * lk;
  f64x4 acc[NT_NC];
#pragma unroll
  for (int c = 0; c < NT_NC; ++c) acc[c] = f64x4{0.0, 0.0, 0.0, 0.0};
  const int b1 = p.b1;   // even
  for (int k0 = 0; k0 < b1; k0 += 8) {
    const bool in = k0 + 2 * lk < b1;
    f64x2 a = in ? *reinterpret_cast<const f64x2*>(q + k0) : f64x2{0.0, 0.0};
    f64x2 b[NT_NC];
#pragma unroll
    for (int c = 0; c < NT_NC; ++c) b[c] = in ? *reinterpret_cast<const f64x2*>(m[c] + k0) : f64x2{0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int c = 0; c < NT_NC; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[c][s], acc[c], 0, 0, 0);
  }
  // D: row = (lane >> 4) + 4 reg, col = lane & 15
#pragma unroll
  for (int c = 0; c < NT_NC; ++c) {
    const int col = col0 + 16 * c + li;
    if (col >= p.b2) continue;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row = row0 + lk + 4 * g;
      if (row < p.n) p.Y[(int64_t)row * p.ldy + col] = acc[c][g];
    }
  }
}

}  // namespace

extern "C" int dmdx_potrf_trtri_max_n(void) { return CMAXN; }

extern "C" size_t dmdx_potrf_trtri_workspace_bytes(int64_t n) {
  if (n < 1) return 0;
  const size_t nbk = (size_t)((n + CB - 1) / CB);
  return 256 + nbk * CB * CB * sizeof(double);
}

extern "C" int dmdx_potrf_trtri_f64(const double* A, int64_t n, int64_t lda, double shift, double* L, int64_t ldl,
                                    double* Linv, int64_t ldi, double* info, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  DMDX_CHECK_ARG(A && L && info, "potrf_trtri: null pointer");
  DMDX_CHECK_ARG(n >= 1 && n <= CMAXN, "potrf_trtri: n = %lld outside [1, %d]", (long long)n, CMAXN);
  DMDX_CHECK_ARG(lda >= n && ldl >= n && (!Linv || ldi >= n), "potrf_trtri: leading dimension smaller than n");
  DMDX_CHECK_ARG(L != A && Linv != A && Linv != L, "potrf_trtri: outputs must not alias the input or each other");
  const size_t need = dmdx_potrf_trtri_workspace_bytes(n);
  if (workspace == nullptr || workspace_bytes < need) {
    dmdx_set_error("potrf_trtri: workspace %zu bytes < required %zu", workspace_bytes, need);
    return DMDX_E_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  DMDX_HIP(hipMemsetAsync(workspace, 0, 256, st));
  // the kernel writes the lower triangles only
  DMDX_HIP(hipMemset2DAsync(L, (size_t)ldl * sizeof(double), 0, (size_t)n * sizeof(double), (size_t)n, st));
  if (Linv) DMDX_HIP(hipMemset2DAsync(Linv, (size_t)ldi * sizeof(double), 0, (size_t)n * sizeof(double), (size_t)n, st));
  CholParams p{};
  p.A = A; p.lda = lda; p.shift = shift;
  p.L = L; p.ldl = ldl;
  p.X = Linv; p.ldx = ldi;
  p.info = info;
  p.bar = reinterpret_cast<unsigned*>(workspace);
  p.Dinv = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + 256);
  p.n = (int)n;
  p.nbk = (int)((n + CB - 1) / CB);
  // one workgroup per block row (<= 32: co-resident on any partition with >= 32 CUs); n <= 64 runs
  // in one workgroup without grid barriers
  const int G = p.nbk <= 2 ? 1 : p.nbk;
  hipLaunchKernelGGL(chol_inv_kernel, dim3((unsigned)G), dim3(CTH), 0, st, p);
  DMDX_LAUNCH_CHECK();
  return 0;
}

extern "C" int dmdx_gemm_nt_f64(const double* Q, int64_t ldq, int64_t n, int64_t b1, const double* Mt, int64_t ldm,
                                int64_t b2, double* Y, int64_t ldy, void* stream) {
  DMDX_CHECK_ARG(Q && Mt && Y, "gemm_nt_f64: null pointer");
  DMDX_CHECK_ARG(n >= 1 && n < (int64_t(1) << 30) && b1 >= 2 && b2 >= 1 && b1 <= 65536 && b2 <= 65536,
                 "gemm_nt_f64: bad shape n=%lld b1=%lld b2=%lld", (long long)n, (long long)b1, (long long)b2);
  DMDX_CHECK_ARG(b1 % 2 == 0 && ldq % 2 == 0 && ldm % 2 == 0, "gemm_nt_f64: b1, ldq, ldm must be even (16-byte loads)");
  DMDX_CHECK_ARG(ldq >= b1 && ldm >= b1 && ldy >= b2, "gemm_nt_f64: leading dimension too small");
  DMDX_CHECK_ARG(dmdx_aligned16(Q) && dmdx_aligned16(Mt), "gemm_nt_f64: Q and Mt must be 16-byte aligned");
  DMDX_CHECK_ARG(Y != Q && Y != Mt, "gemm_nt_f64: Y must not alias an input");
  NtParams p{};
  p.Q = Q; p.Mt = Mt; p.Y = Y;
  p.ldq = ldq; p.ldm = ldm; p.ldy = ldy;
  p.n = (int)n; p.b1 = (int)b1; p.b2 = (int)b2;
  const dim3 grid((unsigned)((n + 63) / 64), (unsigned)((b2 + 16 * NT_NC - 1) / (16 * NT_NC)));
  hipLaunchKernelGGL(gemm_nt64_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
  DMDX_LAUNCH_CHECK();
  return 0;
}
