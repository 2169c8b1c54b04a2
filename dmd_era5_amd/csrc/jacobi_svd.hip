// K7L: one-sided (Hestenes) Jacobi SVD of a square fp64 matrix with 96 < n <= 1024 in ONE launch
// of up to 64 workgroups -- the small dense stage of the rank-200 configurations.
//
// What it is for.  The method of snapshots needs, per SVD, the eigenpairs of (i) the (b x b)
// Rayleigh-Ritz matrices of the top-eigenpair solver and (ii) the graded (l x l) refinement
// matrix T = S M S (s over many decades, M = U'^T U' ~ I), the latter with errors relative to EACH
// eigenvalue.  Both are positive definite: T = C C^T with C the Cholesky factor of (i), or
// C = S L, M = L L^T, for (ii); the left singular vectors of C are the eigenvectors of T and its
// singular values the square roots of the eigenvalues.  One-sided Jacobi computes exactly that
// with high relative accuracy (Demmel & Veselic), and on these nearly orthogonal factors it is
// done after 2-4 sweeps.  The library route costs 7 ms (syevd, n = 312, absolute accuracy only)
// or 37-50 ms (gesvd at n = 250-312) of launch-bound time.
//
// How.  C is held column by column (row c of the (n, n) array = column c of C, contiguous).
// Columns are grouped in blocks of 8; NBLK / 2 workgroups play a round-robin tournament over the
// blocks (NBLK - 1 steps per sweep, every block pair once).  In a step a workgroup copies its two
// blocks (16 columns x n rows) into LDS, rotates all 64 cross pairs -- 8 internal steps of 8
// disjoint pairs, one wave per pair: three dot products over the rows (lanes stride the rows, a
// butterfly of wave shuffles sums them), the rotation that makes the two columns orthogonal,
// applied in place -- (at the first step of a sweep also the 2 x 28 pairs inside the two blocks)
// and writes the blocks back; a grid barrier (a monotonic counter, bounded spin; the column data
// itself moves with cache-bypassing agent-scope loads / stores, so no fences) separates the steps.  A pair is rotated only while
// |a_p . a_q| > tol |a_p| |a_q| (tol = sqrt(n) eps); the sweeps stop after the first one without a
// rotation.  Then: column norms = singular values, ranked in descending order, normalised
// columns = left singular vectors.
#include "dmdx_common.h"

namespace {

constexpr int JW = 8;            // columns per block
constexpr int JTH = 64 * JW;     // one wave per pair of an internal step
constexpr int JMAXN = 1024;
constexpr int JMAX_SWEEPS = 40;
constexpr int JSPIN_LIMIT = 1 << 22;

struct JacobiParams {
  double* C;        // (n, n): row c = column c of the matrix, modified in place
  int64_t ldc;
  int n, nblk;
  double tol;
  unsigned* bar;    // [0] barrier counter, [1] timeout flag, [2 .. 2 + JMAX_SWEEPS) rotations per sweep
  double* snorm;    // n column norms (unsorted)
  double* sigma;    // n singular values, descending
  double* Zt;       // (n, n): row j = left singular vector j
  int64_t ldz;
  int* sweeps_out;  // nullable: sweeps used, -1 on a barrier timeout
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Column data travels between workgroups through global memory with agent-scope (sc1, L1-bypassing
// / write-through) loads and stores only, so the hand-off needs no cache maintenance: every storing
// wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets, ONE lane arrives on the counter
// and polls it, the workgroup meets again, and then loads (MI355X_MICROARCH.md, inter-workgroup
// visibility: the all-sc1 form; one workgroup per CU, hipMalloc memory).  ~3 us less per step
// than the release / acquire fence pair.
__device__ __forceinline__ double ld_cc(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_cc(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// All waves of the workgroup call this.  Returns false after a timeout (some workgroup never
// arrived: the launch was not co-resident), which every later barrier then reports at once.
__device__ bool grid_barrier(unsigned* bar, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have left the CU
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int good = 1, spins = 0;
    while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > JSPIN_LIMIT || __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        __hip_atomic_store(&bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        good = 0;
        break;
      }
    }
    ok = good;
  }
  __syncthreads();
  return ok != 0;
}

// One wave: make LDS columns a and b (n rows each) orthogonal.  NR = ceil(rows / 64) upper bound.
template <int NR>
__device__ __forceinline__ int rotate_pair(double* __restrict__ a, double* __restrict__ b, int n, double tol, int lane) {
  double x[NR], y[NR];
  double saa = 0.0, sbb = 0.0, sab = 0.0;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int r = lane + 64 * i;
    x[i] = (r < n) ? a[r] : 0.0;
    y[i] = (r < n) ? b[r] : 0.0;
    saa = fma(x[i], x[i], saa);
    sbb = fma(y[i], y[i], sbb);
    sab = fma(x[i], y[i], sab);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {   // three independent butterflies, interleaved
    const double ta = __shfl_xor(saa, off, 64), tb = __shfl_xor(sbb, off, 64), tc = __shfl_xor(sab, off, 64);
    saa += ta;
    sbb += tb;
    sab += tc;
  }
  if (!(fabs(sab) > tol * sqrt(saa * sbb)) || !(fabs(sab) > 1e-300)) return 0;
  const double zeta = (sbb - saa) / (2.0 * sab);
  const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
  const double c = 1.0 / sqrt(1.0 + t * t);
  const double s = t * c;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int r = lane + 64 * i;
    if (r < n) {
      a[r] = c * x[i] - s * y[i];
      b[r] = s * x[i] + c * y[i];
    }
  }
  return 1;
}

template <int NR>
__global__ __launch_bounds__(JTH) void jacobi_svd_kernel(JacobiParams p) {
  constexpr int LD = 64 * NR;
  __shared__ double cols[2 * JW * LD];
  __shared__ int flag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = p.n, nblk = p.nblk, nwg = gridDim.x, wg = blockIdx.x;
  unsigned epoch = 0;
  bool alive = true, converged = false;
  int sweep = 0;
  const __amdgpu_buffer_rsrc_t crs = dmdx_sc1_rsrc(p.C);
  const int ldcb = (int)p.ldc * 8;

  for (; sweep < JMAX_SWEEPS && alive; ++sweep) {
    int rotated_wg = 0;
    for (int step = 0; step < nblk - 1 && alive; ++step) {
      int bp, bq;
      if (wg == 0) {
        bp = nblk - 1;
        bq = step;
      } else {
        bp = (step + wg) % (nblk - 1);
        bq = (step - wg + (nblk - 1)) % (nblk - 1);
      }
      // ---- the two blocks into LDS: wave w brings column w of each
      {
        const int cp = bp * JW + wave, cq = bq * JW + wave;
        double* dp = cols + wave * LD;
        double* dq = cols + (JW + wave) * LD;
        // all 2 NR loads in flight together (offsets clamped into the matrix, the value masked afterwards)
        const int op = (cp < n ? cp : 0) * ldcb, oq = (cq < n ? cq : 0) * ldcb;
        double vp[NR], vq[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const int r = lane + 64 * i, rc = r < n ? r : n - 1;
          vp[i] = dmdx_ld_sc1_f64(crs, op + 8 * rc);
          vq[i] = dmdx_ld_sc1_f64(crs, oq + 8 * rc);
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const int r = lane + 64 * i;
          dp[r] = (cp < n && r < n) ? vp[i] : 0.0;
          dq[r] = (cq < n && r < n) ? vq[i] : 0.0;
        }
      }
      __syncthreads();
      int rot = 0;
      if (step == 0) {
        // pairs inside each block once per sweep: round robin of 8 columns, 7 x 4 pairs per block
        const int blk = wave >> 2, i = wave & 3;
        double* base = cols + blk * JW * LD;
        for (int t = 0; t < JW - 1; ++t) {
          int a, b;
          if (i == 0) {
            a = JW - 1;
            b = t;
          } else {
            a = (t + i) % (JW - 1);
            b = (t - i + (JW - 1)) % (JW - 1);
          }
          rot += rotate_pair<NR>(base + a * LD, base + b * LD, n, p.tol, lane);
          __syncthreads();
        }
      }
      for (int t = 0; t < JW; ++t) {
        rot += rotate_pair<NR>(cols + wave * LD, cols + (JW + ((wave + t) & (JW - 1))) * LD, n, p.tol, lane);
        __syncthreads();
      }
      rotated_wg += __syncthreads_count(rot != 0);
      // ---- and back
      {
        const int cp = bp * JW + wave, cq = bq * JW + wave;
        const double* dp = cols + wave * LD;
        const double* dq = cols + (JW + wave) * LD;
        double* sp = p.C + (int64_t)cp * p.ldc;
        double* sq = p.C + (int64_t)cq * p.ldc;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const int r = lane + 64 * i;
          if (cp < n && r < n) st_cc(sp + r, dp[r]);
          if (cq < n && r < n) st_cc(sq + r, dq[r]);
        }
      }
      if (step == nblk - 2 && tid == 0 && rotated_wg != 0)
        __hip_atomic_fetch_add(&p.bar[2 + sweep], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ++epoch;
      alive = grid_barrier(p.bar, epoch * (unsigned)nwg);
    }
    if (!alive) break;
    if (tid == 0)
      flag = (int)__hip_atomic_load(&p.bar[2 + sweep], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int any = flag;
    __syncthreads();
    if (any == 0) {
      converged = true;
      ++sweep;
      break;
    }
  }

  // ---- column norms of this workgroup's two home blocks
  if (alive) {
    for (int h = 0; h < 2; ++h) {
      const int c = (2 * wg + h) * JW + wave;
      if (c < n) {
        double v[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const int r = lane + 64 * i;
          v[i] = dmdx_ld_sc1_f64(crs, c * ldcb + 8 * (r < n ? r : n - 1));
        }
        double ss = 0.0;
#pragma unroll
        for (int i = 0; i < NR; ++i)
          if (lane + 64 * i < n) ss = fma(v[i], v[i], ss);
        ss = wave_sum(ss);
        if (lane == 0) st_cc(&p.snorm[c], sqrt(ss));
      }
    }
    ++epoch;
    alive = grid_barrier(p.bar, epoch * (unsigned)nwg);
  }
  // sweeps used (the rotation-free one included); JMAX_SWEEPS + 1 = the limit was hit without one; -1 = barrier timeout
  if (wg == 0 && tid == 0 && p.sweeps_out) *p.sweeps_out = alive ? (converged ? sweep : JMAX_SWEEPS + 1) : -1;
  if (!alive) return;

  // ---- rank (descending, ties by index) and normalised output
  double* sn = cols;  // n <= 1024 doubles
  for (int i = tid; i < n; i += JTH) sn[i] = ld_cc(&p.snorm[i]);
  __syncthreads();
  for (int h = 0; h < 2; ++h) {
    const int c = (2 * wg + h) * JW + wave;
    if (c >= n) continue;
    const double mine = sn[c];
    int rank = 0;
    for (int j = lane; j < n; j += 64) {
      const double o = sn[j];
      rank += (o > mine) || (o == mine && j < c);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) rank += __shfl_xor(rank, off, 64);
    const double inv = mine > 0.0 ? 1.0 / mine : 0.0;
    double* dst = p.Zt + (int64_t)rank * p.ldz;
    double v[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int r = lane + 64 * i;
      v[i] = dmdx_ld_sc1_f64(crs, c * ldcb + 8 * (r < n ? r : n - 1));
    }
#pragma unroll
    for (int i = 0; i < NR; ++i)
      if (lane + 64 * i < n) dst[lane + 64 * i] = v[i] * inv;
    if (lane == 0) p.sigma[rank] = mine;
  }
}

}  // namespace

extern "C" int dmdx_svd_jacobi_max_n(void) {
  const int cus = dmdx_device_cus();   // n / 16 workgroups must be co-resident, one per CU
  return (cus > 0 && 16 * cus < JMAXN) ? 16 * cus : JMAXN;
}

extern "C" size_t dmdx_svd_jacobi_workspace_bytes(int64_t n) {
  return 256 + (size_t)(n > 0 ? n : 0) * sizeof(double);
}

extern "C" int dmdx_svd_jacobi_f64(double* C, int64_t n, int64_t ldc, double* sigma, double* Zt, int64_t ldz,
                                   int* sweeps, void* workspace, size_t workspace_bytes, void* stream) {
  DMDX_CHECK_ARG(C && sigma && Zt, "svd_jacobi: null pointer");
  DMDX_CHECK_ARG(n >= 2 && n <= JMAXN, "svd_jacobi: n = %lld outside [2, %d]", (long long)n, JMAXN);
  DMDX_CHECK_ARG(ldc >= n && ldz >= n, "svd_jacobi: leading dimension smaller than n");
  DMDX_CHECK_ARG(ldc <= (int64_t(1) << 17), "svd_jacobi: ldc = %lld beyond the 32-bit byte offsets of the column loads", (long long)ldc);
  DMDX_CHECK_ARG(Zt != C, "svd_jacobi: Zt must not alias C");
  const size_t need = dmdx_svd_jacobi_workspace_bytes(n);
  if (workspace == nullptr || workspace_bytes < need) {
    dmdx_set_error("svd_jacobi: workspace %zu bytes < required %zu", workspace_bytes, need);
    return DMDX_E_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  // the workgroups meet at a spin barrier: all of them must be resident at once (one per CU: up to
  // 128 KB of LDS each).  On a partitioned device (CPX: 32 CUs) a larger grid would spin to its
  // limit before reporting -1: refuse it here, the callers fall back to the library at once.
  {
    int nb2 = (int)((n + JW - 1) / JW);
    nb2 += nb2 & 1;
    const int cus = dmdx_device_cus();
    if (cus > 0 && nb2 / 2 > cus) {
      dmdx_set_error("svd_jacobi: n = %lld needs %d co-resident workgroups, the device has %d CUs", (long long)n, nb2 / 2, cus);
      return DMDX_E_UNSUPPORTED;
    }
  }
  DMDX_HIP(hipMemsetAsync(workspace, 0, 256, st));
  JacobiParams p{};
  p.C = C;
  p.ldc = ldc;
  p.n = (int)n;
  int nblk = (int)((n + JW - 1) / JW);
  nblk += nblk & 1;
  if (nblk < 2) nblk = 2;
  p.nblk = nblk;
  p.tol = sqrt((double)n) * 2.220446049250313e-16;
  p.bar = reinterpret_cast<unsigned*>(workspace);
  p.snorm = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + 256);
  p.sigma = sigma;
  p.Zt = Zt;
  p.ldz = ldz;
  p.sweeps_out = sweeps;
  const dim3 grid((unsigned)(nblk / 2));
  if (n <= 128) hipLaunchKernelGGL(jacobi_svd_kernel<2>, grid, dim3(JTH), 0, st, p);
  else if (n <= 256) hipLaunchKernelGGL(jacobi_svd_kernel<4>, grid, dim3(JTH), 0, st, p);
  else if (n <= 512) hipLaunchKernelGGL(jacobi_svd_kernel<8>, grid, dim3(JTH), 0, st, p);
  else hipLaunchKernelGGL(jacobi_svd_kernel<16>, grid, dim3(JTH), 0, st, p);
  DMDX_LAUNCH_CHECK();
  return 0;
}
