// K7: eigendecomposition of a small symmetric fp64 matrix (n <= 96) in ONE launch of one
// workgroup: two-sided cyclic Jacobi with the round-robin ("tournament") ordering, A and V
// resident in LDS (2 x 97 x 96 x 8 B = 149 KB of the CU's 160 KB).
//
// Why: the method of snapshots needs the eigenpairs of two small projected matrices per SVD
// (the (b x b) Rayleigh-Ritz matrix of the top-eigenpair solver and the (l x l) refinement
// matrix; b, l <= 96 for rank <= 75).  rocSOLVER's syevd issues ~10 launches per Householder
// step and is bound by the host's launch rate at these sizes (n = 77: 2 ms of GPU work, 5 ms of
// wall time); one launch of this kernel is ~1 ms and needs no host round trip.
//
// Each step rotates n/2 disjoint index pairs (p, q):  A <- J^T A J,  V <- V J  with
// J = [[c, s], [-s, c]] in the (p, q) plane chosen to zero A[p][q]; a sweep is n-1 steps
// (every pair once).  A pair is rotated only while |a_pq| > 1e-15 sqrt(|a_pp a_qq|) -- the
// relative criterion that gives Jacobi its high relative accuracy on graded (Gram-type)
// matrices -- and the sweeps stop after the first sweep without a rotation (7-10 sweeps in fp64).
#include "dmdx_common.h"

namespace {

constexpr int EN = 96;        // largest matrix
constexpr int ELD = EN + 1;   // LDS row stride in doubles (odd: column walks spread over the banks)
constexpr int ETH = 1024;
constexpr int MAX_SWEEPS = 30;

__global__ __launch_bounds__(ETH) void eigh_jacobi_kernel(const double* __restrict__ Ain, int n, int64_t lda,
                                                         double* __restrict__ w, double* __restrict__ Vout,
                                                         int64_t ldv, int* __restrict__ sweeps_out) {
  __shared__ double As[EN * ELD];
  __shared__ double Vs[EN * ELD];
  __shared__ double rc[EN / 2], rs[EN / 2];
  __shared__ int rp[EN / 2], rq[EN / 2];
  __shared__ int order[EN];

  const int tid = threadIdx.x;
  const int ne = (n + 1) & ~1;  // even size: an odd matrix gets one decoupled zero row/column
  const int half = ne / 2;

  for (int idx = tid; idx < ne * ne; idx += ETH) {
    const int i = idx / ne, j = idx - i * ne;
    double a = 0.0;
    if (i < n && j < n) a = 0.5 * (Ain[(int64_t)i * lda + j] + Ain[(int64_t)j * lda + i]);
    As[i * ELD + j] = a;
    Vs[i * ELD + j] = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();

  int sweep = 0, converged = 0;   // sweeps_out: sweeps used (the rotation-free one included), MAX_SWEEPS + 1 = no convergence
  for (; sweep < MAX_SWEEPS; ++sweep) {
    int nrot = 0;
    for (int step = 0; step < ne - 1; ++step) {
      // ---- rotation parameters of the n/2 pairs of this step (round robin, player ne-1 fixed)
      int rotated = 0;
      if (tid < half) {
        int p, q;
        if (tid == 0) { p = ne - 1; q = step; }
        else { p = (step + tid) % (ne - 1); q = (step - tid + (ne - 1)) % (ne - 1); }
        if (p > q) { const int t = p; p = q; q = t; }
        const double app = As[p * ELD + p], aqq = As[q * ELD + q], apq = As[p * ELD + q];
        double c = 1.0, s = 0.0;
        if (fabs(apq) > 1e-15 * sqrt(fabs(app * aqq)) && fabs(apq) > 1e-300) {
          rotated = 1;
          const double tau = (aqq - app) / (2.0 * apq);
          const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
          c = 1.0 / sqrt(1.0 + t * t);
          s = t * c;
        }
        rp[tid] = p; rq[tid] = q; rc[tid] = c; rs[tid] = s;
      }
      nrot += __syncthreads_count(rotated);
      // ---- rows p, q of A:  (a_p, a_q) <- (c a_p - s a_q, s a_p + c a_q)
      for (int idx = tid; idx < half * ne; idx += ETH) {
        const int k = idx / ne, j = idx - k * ne;
        const int p = rp[k], q = rq[k];
        const double c = rc[k], s = rs[k];
        const double ap = As[p * ELD + j], aq = As[q * ELD + j];
        As[p * ELD + j] = c * ap - s * aq;
        As[q * ELD + j] = s * ap + c * aq;
      }
      __syncthreads();
      // ---- columns p, q of A and of V
      for (int idx = tid; idx < half * ne; idx += ETH) {
        const int k = idx / ne, i = idx - k * ne;
        const int p = rp[k], q = rq[k];
        const double c = rc[k], s = rs[k];
        const double ap = As[i * ELD + p], aq = As[i * ELD + q];
        As[i * ELD + p] = c * ap - s * aq;
        As[i * ELD + q] = s * ap + c * aq;
        const double vp = Vs[i * ELD + p], vq = Vs[i * ELD + q];
        Vs[i * ELD + p] = c * vp - s * vq;
        Vs[i * ELD + q] = s * vp + c * vq;
      }
      __syncthreads();
      if (tid < half && rs[tid] != 0.0) {  // the rotated entry is zero by construction
        As[rp[tid] * ELD + rq[tid]] = 0.0;
        As[rq[tid] * ELD + rp[tid]] = 0.0;
      }
      __syncthreads();
    }
    if (nrot == 0) { converged = 1; ++sweep; break; }
  }

  // ---- eigenvalues descending (rank sort; ties broken by index), eigenvectors in columns
  if (tid < n) {
    const double mine = As[tid * ELD + tid];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const double o = As[j * ELD + j];
      rank += (o > mine) || (o == mine && j < tid);
    }
    order[rank] = tid;
  }
  __syncthreads();
  if (tid < n) w[tid] = As[order[tid] * ELD + order[tid]];
  for (int idx = tid; idx < n * n; idx += ETH) {
    const int i = idx / n, j = idx - i * n;
    Vout[(int64_t)i * ldv + j] = Vs[i * ELD + order[j]];
  }
  if (tid == 0 && sweeps_out) *sweeps_out = converged ? sweep : MAX_SWEEPS + 1;
}

}  // namespace

extern "C" int dmdx_eigh_small_max_n(void) { return EN; }

extern "C" int dmdx_eigh_small_f64(const double* A, int64_t n, int64_t lda, double* w, double* V, int64_t ldv,
                                   int* sweeps, void* stream) {
  DMDX_CHECK_ARG(A && w && V, "eigh_small: null pointer");
  DMDX_CHECK_ARG(n >= 1 && n <= EN, "eigh_small: n = %lld outside [1, %d]", (long long)n, EN);
  DMDX_CHECK_ARG(lda >= n && ldv >= n, "eigh_small: leading dimension smaller than n");
  hipLaunchKernelGGL(eigh_jacobi_kernel, dim3(1), dim3(ETH), 0, (hipStream_t)stream, A, (int)n, lda, w, V, ldv,
                     sweeps);
  DMDX_LAUNCH_CHECK();
  return 0;
}
