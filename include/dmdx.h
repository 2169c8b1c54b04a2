/*
 * dmdx.h -- C ABI of libdmdx.so, the MI355X (gfx950) kernels behind the
 * ERA5 snapshot-matrix SVD hot path.
 *
 * What it replaces in the reference (ClimeTrend/DMD-ERA5, paths relative to
 * the reference root): the reference has no FFI; its hot loop is two
 * third-party CPU calls made from
 *     src/dmd_era5/era5_svd/era5_svd.py:251   np.linalg.svd(X, full_matrices=False)
 *     src/dmd_era5/era5_svd/era5_svd.py:258   sklearn randomized_svd(X, n_components)
 * plus the pre-processing passes that build X
 *     src/dmd_era5/slice_tools/slice_tools.py:171-179  (mean / std / centre / scale)
 *     src/dmd_era5/slice_tools/slice_tools.py:207-211  (delay embedding)
 * Each entry point below names the reference line whose arithmetic it takes
 * over.  The Python host (dmd_era5_amd/) binds these with ctypes and keeps the
 * reference's function signatures (svd_on_era5, main, ...).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless named host_*;
 *   - matrices are COLUMN-MAJOR with a leading dimension in elements,
 *     (ptr, rows, cols, ld); rows > ld is allowed for the input X only: that is
 *     the zero-copy delay-embedding view E[k*m+s, t] = X[s, t+k] = ptr[(k*m+s) + t*ld]
 *     with ld = m (slice_tools.py:207-211);
 *   - stream is a hipStream_t passed as void* (NULL = default stream);
 *   - calls enqueue work on `stream` and return; no hidden synchronisation,
 *     no allocation (workspaces are caller-provided);
 *   - return value 0 = ok, < 0 = error (-(hipError_t) or DMDX_E_*);
 *     dmdx_last_error() gives the message for the calling thread.
 */
#ifndef DMDX_H
#define DMDX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMDX_VERSION 110 /* 0.1.1 */

#define DMDX_E_INVALID (-1000)  /* bad argument (shape, ld, null pointer)  */
#define DMDX_E_WORKSPACE (-1001) /* workspace too small                      */
#define DMDX_E_UNSUPPORTED (-1002)

int dmdx_version(void);
const char* dmdx_last_error(void);

/* ---- K1: Gram matrix G = X^T X  (method of snapshots) ---------------------
 * Takes over the O(m n^2) part of np.linalg.svd (era5_svd.py:251).
 * X: m x n fp32 (ldx), G64: n x n fp64 (ldg), both triangles written.
 * G32 (nullable): fp32 copy of G (ldg32).
 * accumulate != 0: G64 += X^T X (G32, if given, receives the rounded new G64): the
 * snapshot matrix is kept in HBM as row (space) blocks with a short leading
 * dimension -- a 4 MB column stride makes every 128-byte access of a 128-column
 * panel hit a different 2 MB page and thrashes the TLB -- and the Gram is summed
 * over the blocks in fp64.
 * fp32 MFMA products, fp32 chains of at most 4096 rows, fp64 across chains.
 * Deterministic (no atomics).  */
size_t dmdx_syrk_workspace_bytes(int64_t m, int64_t n);
int dmdx_syrk_f32(const float* X, int64_t m, int64_t n, int64_t ldx,
                  double* G64, int64_t ldg, float* G32, int64_t ldg32, int accumulate,
                  void* workspace, size_t workspace_bytes, void* stream);

/* K1 over a list of row blocks in ONE launch per 16 blocks:  G (+)= sum_j X_j^T X_j.
 * X, m, ldx: HOST arrays of nblocks device pointers / row counts / leading dimensions (block j is
 * m[j] x n, column-major, ldx[j]); everything else as dmdx_syrk_f32.  Same arithmetic per block;
 * the K-splits of all blocks are summed by one reduce kernel.  Replaces the per-block loop of
 * launches (last partial round of units, reduce kernel and launch gap per block). */
size_t dmdx_syrk_blocks_workspace_bytes(const int64_t* m, int nblocks, int64_t n);
int dmdx_syrk_blocks_f32(const float* const* X, const int64_t* m, const int64_t* ldx, int nblocks,
                         int64_t n, double* G64, int64_t ldg, float* G32, int64_t ldg32,
                         int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* ---- K3: C = A^T B, A: K x na, B: K x nb (both K-contiguous) ---------------
 * Z = X^T Y of the randomized range finder (extmath.py:351, `A.T @ Q`) and
 * B = Q^T X (extmath.py:577).  C64: na x nb fp64 (ldc); C32 nullable.  */
size_t dmdx_gemm_tn_workspace_bytes(int64_t K, int64_t na, int64_t nb);
int dmdx_gemm_tn_f32(const float* A, int64_t lda, const float* B, int64_t ldb,
                     int64_t K, int64_t na, int64_t nb,
                     double* C64, int64_t ldc, float* C32, int64_t ldc32, int accumulate,
                     void* workspace, size_t workspace_bytes, void* stream);

/* K3 over lists of row blocks in ONE launch per 16 blocks:  C (+)= sum_j A_j^T B_j  (A_j: K[j] x na,
 * lda[j]; B_j: K[j] x nb, ldb[j]; all five arrays live on the HOST).  The randomized path's
 * Z = X^T Y and B = Q^T X are sums over the row blocks of X; per-block launches of this
 * HBM-streaming product are only ~4 rounds of workgroups each. */
size_t dmdx_gemm_tn_blocks_workspace_bytes(const int64_t* K, int nblocks, int64_t na, int64_t nb);
int dmdx_gemm_tn_blocks_f32(const float* const* A, const int64_t* lda, const float* const* B,
                            const int64_t* ldb, const int64_t* K, int nblocks, int64_t na, int64_t nb,
                            double* C64, int64_t ldc, float* C32, int64_t ldc32, int accumulate,
                            void* workspace, size_t workspace_bytes, void* stream);

/* ---- K2: tall-skinny Y = X W -----------------------------------------------
 * X: m x n (ldx, rows > ldx allowed), W: n x l (ldw), Y: m x l (ldy); l is processed in column
 * groups of at most 224, 16-column granular (X is re-read once per group).
 * U = X (V_r S^-1) of the method of snapshots and `A @ Q` of the range finder
 * (extmath.py:349,355).  Fast path (16-byte loads of X, W and stores of Y): m, ldx, ldw, ldy
 * multiples of 4 and X, W, Y 16-byte aligned; anything else takes the scalar-load path (same
 * results, about half the rate).  When n % 4 != 0 give the small W a padded ldw rather than a
 * tight one -- the Python host does (HipKernels.pitch). */
int dmdx_gemm_nn_skinny_f32(const float* X, int64_t m, int64_t n, int64_t ldx,
                            const float* W, int64_t ldw, int64_t l,
                            float* Y, int64_t ldy, void* stream);

/* K2 with the Gram of its output fused in: Y = X W as above (l <= dmdx_gemm_nn_skinny_gram_max_l(),
 * 224) and G (+)= Y^T Y (l x l fp64, ldg, both triangles), formed from the accumulators before
 * they leave the registers: the CholeskyQR rounds of the range finder (the LU / QR normalisers of
 * extmath.py:349-355 in this engine) need that Gram, and computing it separately is another pass
 * over the m x l matrix.  Per-workgroup fp32 partial tiles in the workspace (summed over the
 * workgroup's waves in a fixed order), added up in fp64 (deterministic). */
int dmdx_gemm_nn_skinny_gram_max_l(void);
size_t dmdx_gemm_nn_skinny_gram_workspace_bytes(int64_t m, int64_t l);
int dmdx_gemm_nn_skinny_gram_f32(const float* X, int64_t m, int64_t n, int64_t ldx,
                                 const float* W, int64_t ldw, int64_t l, float* Y, int64_t ldy,
                                 double* G, int64_t ldg, int accumulate,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* ---- K5: per-row (space point) mean / std over time, centre, scale ---------
 * slice_tools.py:171-179.  X: m x n (ldx) modified in place:
 *   mean[i] = sum_j X[i,j] / n ; X[i,:] -= mean[i];
 *   if (scale) { std[i] = sqrt(sum_j X[i,j]^2 / n) of the centred row (ddof 0);
 *                X[i,:] /= std[i]; }
 * mean: m floats (required), std: m floats (required iff scale).
 * Sums are accumulated in fp64.  */
int dmdx_row_center_scale_f32(float* X, int64_t m, int64_t n, int64_t ldx,
                              float* mean, float* std, int scale, void* stream);

/* ---- K6: Gram of the delay-embedded matrix from the plain Gram -------------
 * Gd[i,j] = sum_{k<d} G[i+k, j+k], Gd is (n-d+1)^2  (slice_tools.py:207-211
 * applied to X^T X).  fp64 in, fp64 out (Gd32 nullable fp32 copy).  */
int dmdx_delay_shift_sum_f64(const double* G, int64_t n, int64_t ldg, int d,
                             double* Gd, int64_t ldgd, float* Gd32, int64_t ldgd32,
                             void* stream);

/* ---- small helpers on the same stream --------------------------------------
 * Y[:, j] *= alpha[j]  (m x l, ldy)  -- S^-1 scaling / sign flip of U columns */
int dmdx_scale_columns_f32(float* Y, int64_t m, int64_t l, int64_t ldy,
                           const float* alpha, void* stream);

/* ---- K7: eigenpairs of a small symmetric fp64 matrix, one launch -------------
 * A (n x n, lda, either storage order: only (A + A^T)/2 is used), n <= dmdx_eigh_small_max_n()
 * (96).  w[0..n) eigenvalues in DESCENDING order, V (n x n, row-major, ldv): column j is the
 * unit eigenvector of w[j].  sweeps: nullable device int, number of Jacobi sweeps used (the
 * rotation-free last one included); limit + 1 (31) = no rotation-free sweep within the limit.
 * Replaces the LAPACK syevd calls on the projected matrices of the method of snapshots (the
 * part of np.linalg.svd, era5_svd.py:251, that is left once X is reduced to its Gram matrix). */
int dmdx_eigh_small_max_n(void);
int dmdx_eigh_small_f64(const double* A, int64_t n, int64_t lda, double* w, double* V,
                        int64_t ldv, int* sweeps, void* stream);

/* ---- K7L: one-sided Jacobi SVD of a square fp64 matrix, 2 <= n <= dmdx_svd_jacobi_max_n() (1024),
 * one launch of <= 64 workgroups (they synchronise through a counter in the workspace: the launch
 * must be able to run them all at once, i.e. nothing else may occupy the device for good).
 * C: n x n, COLUMN c at C + c * ldc (contiguous), overwritten.  sigma[0..n): singular values in
 * DESCENDING order; Zt (n x n, ldz): ROW j = unit left singular vector of sigma[j] (zero rows for
 * zero singular values).  sweeps (nullable device int): sweeps used (the rotation-free last one
 * included), 41 = the limit of 40 was hit without one, -1 if the workgroups could not synchronise
 * (results invalid); a grid larger than the device's CU count is refused (DMDX_E_UNSUPPORTED;
 * dmdx_svd_jacobi_max_n() already accounts for it).  With C = chol(T) (or S L for T = S L L^T S) this gives
 * the eigenpairs of the positive definite T = C C^T with errors relative to each eigenvalue:
 * the (b x b) Rayleigh-Ritz matrices and the graded refinement matrix of the method of
 * snapshots beyond K7's n <= 96 -- the LAPACK syevd / gesvd calls left of np.linalg.svd
 * (era5_svd.py:251) once X is reduced to its Gram matrix. */
int dmdx_svd_jacobi_max_n(void);
size_t dmdx_svd_jacobi_workspace_bytes(int64_t n);
int dmdx_svd_jacobi_f64(double* C, int64_t n, int64_t ldc, double* sigma, double* Zt, int64_t ldz,
                        int* sweeps, void* workspace, size_t workspace_bytes, void* stream);

/* ---- K8: Y = G Q - shift Q, G symmetric n x n fp64, Q / Y n x b row-major (ldq, ldy) ------
 * The products of the top-eigenpair solver on the Gram matrix (the part of np.linalg.svd,
 * era5_svd.py:251, left once X is reduced to G): fp64 MFMA, G streamed once, Q staged in LDS,
 * K split over workgroups with per-split partial tiles in the workspace (deterministic), summed
 * and shifted by a second kernel.  n, b, ldg, ldq even; G, Q, workspace 16-byte aligned; only
 * G = G^T is supported (row k of G is read as column k).  Y must not alias Q. */
size_t dmdx_symm_skinny_workspace_bytes(int64_t n, int64_t b);
int dmdx_symm_skinny_f64(const double* G, int64_t n, int64_t ldg, const double* Q, int64_t ldq,
                         int64_t b, double shift, double* Y, int64_t ldy,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ---- K9: C = A^T B for tall fp64 blocks, A n x b1 (lda), B n x b2 (ldb), C b1 x b2 (ldc), all
 * row-major.  The Grams of the CholeskyQR rounds, the Rayleigh-Ritz matrices and the block
 * projections of the top-eigenpair solver (the same part of np.linalg.svd, era5_svd.py:251, as K8).
 * fp64 MFMA, K split over workgroups, per-split partial tiles in the workspace (deterministic).
 * b1, b2, lda, ldb even; A, B, workspace 16-byte aligned. */
size_t dmdx_gemm_tn_f64_workspace_bytes(int64_t n, int64_t b1, int64_t b2);
int dmdx_gemm_tn_f64(const double* A, int64_t lda, const double* B, int64_t ldb, int64_t n,
                     int64_t b1, int64_t b2, double* C, int64_t ldc,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ---- K10: Cholesky factor and its inverse of a small symmetric positive definite fp64 matrix, one launch --
 * A (n x n, lda, row-major; only the lower triangle is read), n <= dmdx_potrf_trtri_max_n() (1024):
 * A + shift I = L L^T, L (n x n, ldl) lower triangular (zeros above), Linv (n x n, ldi, nullable) = L^-1
 * (lower triangular).  info: 3 device doubles -- [0] status: 0 ok, j + 1 = first non-positive or
 * non-finite pivot (it is replaced by 1 and the factorisation goes on, so the outputs stay finite: the
 * caller shifts and retries), -1 = the workgroups of the launch could not synchronise;
 * [1] / [2] = min / max of diag(L) (~ 1 / cond).  One workgroup per 32-row block row (<= 32), one
 * grid barrier per block column (counter in the workspace: the launch must be co-resident).
 * Replaces LAPACK potrf + trtri / trsm behind the CholeskyQR rounds (the LU / QR normalisers of
 * sklearn's randomized_svd, extmath.py:349-355, era5_svd.py:258) and the Cholesky factors fed to the
 * Jacobi kernel (part of np.linalg.svd, era5_svd.py:251). */
int dmdx_potrf_trtri_max_n(void);
size_t dmdx_potrf_trtri_workspace_bytes(int64_t n);
int dmdx_potrf_trtri_f64(const double* A, int64_t n, int64_t lda, double shift, double* L, int64_t ldl,
                         double* Linv, int64_t ldi, double* info, void* workspace, size_t workspace_bytes,
                         void* stream);

/* ---- K11: Y = Q Mt^T for a tall fp64 block Q (n x b1, ldq) and a small Mt (b2 x b1, ldm), Y n x b2 (ldy),
 * all row-major: Q L^-T of a CholeskyQR round (Mt = L^-1 from K10) and S Z of a Rayleigh-Ritz step
 * (Mt = Z^T as K7L returns it).  fp64 MFMA, operands straight from global memory.  b1, ldq, ldm even;
 * Q, Mt 16-byte aligned; Y must not alias an input. */
int dmdx_gemm_nt_f64(const double* Q, int64_t ldq, int64_t n, int64_t b1, const double* Mt, int64_t ldm,
                     int64_t b2, double* Y, int64_t ldy, void* stream);

/* ---- upper triangle of a symmetric fp64 matrix <-> packed row by row ---------------------
 * packed[i (2n - i + 1) / 2 + (j - i)] = A[i][j], j >= i: what the Gram all-reduce of the
 * row-sharded path moves (n (n + 1) / 2 doubles instead of n^2).  unpack writes both triangles. */
int dmdx_pack_triu_f64(const double* A, int64_t n, int64_t lda, double* packed, void* stream);
int dmdx_unpack_triu_f64(const double* packed, int64_t n, double* A, int64_t lda, void* stream);

/* ---- exponential basis of the optimized-DMD fit (BASELINE config 5; the reference announces the fit,
 * README.md:85,139, and holds no code for it) -----------------------------------------------------
 * Phi[i][j] = exp(alpha_j t_i), W = diag(t) Phi (nullable): alpha r complex128 (re, im interleaved),
 * t n fp64, Phi / W n x r row-major complex64 (single_precision != 0) or complex128.  The exponent
 * is formed and range-reduced in fp64 whatever the output type. */
int dmdx_exp_basis(const double* alpha, const double* t, int64_t n, int64_t r, void* Phi, void* W,
                   int single_precision, void* stream);

/* ---- measurement aid (not on the path): sustained core clock of the Gram launches ----------
 * While dev_counters3 (3 device uint64, caller-zeroed) is set, every workgroup of the batched
 * launches (dmdx_syrk_blocks_f32, dmdx_gemm_tn_blocks_f32) and of dmdx_gemm_nn_skinny_f32 adds its
 * core-clock cycles (s_memtime), its 100 MHz reference ticks
 * (s_memrealtime) and 1 to it: clock = 100 MHz * [0] / [1].  NULL (the default) switches the
 * stamps off again; bench.py's calibration block is the only caller. */
int dmdx_set_clock_probe(unsigned long long* dev_counters3);

/* ---- measurement aid (not on the path): register-only fp32 MFMA loop -----------
 * 2 workgroups of 4 waves per CU, 16 * iters v_mfma_f32_32x32x2_f32 per wave; *flops_out
 * (host pointer, nullable) receives the flops of the launch.  bench.py --calibrate times it. */
int dmdx_calib_mfma_f32(int iters, int num_cus, float* sink, double* flops_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DMDX_H */
