// K1 / K3: D = OpA^T OpB for K-contiguous (column-major, tall) operands, on the
// fp32-input MFMA (v_mfma_f32_32x32x2_f32).  K1 (Gram, G = X^T X) is the SYRK
// mode: OpA = OpB = X and only the tiles of one triangle are computed.
//
// Decomposition
//   work unit = (K-split s, 128x128 output tile t); one 256-thread workgroup
//   (4 waves, 2x2, each wave a 64x64 sub-tile = 2x2 MFMA blocks of 32x32) per
//   unit, two workgroups resident per CU.
//   K loop: 32-row chunks, register-staged global -> LDS double buffer
//   (global_load_dwordx4 along K, which is the contiguous axis of both
//   operands; LDS rows padded to 36 floats so the ds_read_b128 fragment reads
//   are bank-conflict free).  One ds_read_b128 per operand block feeds four
//   MFMAs: lane (r, h) reads k = 8t+4h .. 8t+4h+3 of its row, MFMA j uses
//   element j of both operands -- the same k permutation on both sides, so
//   the contraction is unchanged.
//   Numerics: fp32 MFMA chains of at most 32 chunks * 32 = 1024 rows, then the
//   chain is added in fp64 into the unit's own partial tile in HBM (owned
//   read-modify-write, no atomics => deterministic); a second kernel sums the
//   K-splits in fp64 and writes D (both triangles in SYRK mode).
//
// L2 locality: units are ordered tile-fastest inside a K-split, the tiles of
// the triangle are enumerated in 4-row super-rows, column by column, so 32
// consecutive units form a 4x8 patch of tiles; blockIdx is remapped so that
// the 32 blocks that land on one XCD (blockIdx % 8 equal) take one patch.
#include <stdarg.h>
#include <stdio.h>

#include "dmdx_common.h"

namespace {

constexpr int BT = 128;            // output tile edge
constexpr int BK = 32;             // K rows per stage
constexpr int LDT = BK + 4;        // padded LDS row (floats)
constexpr int NTH = 256;
// fp32 chain length = 32 chunks * 32 rows = 1024 rows (hard-wired in the fold schedule)
constexpr int TILE_ELEMS = BT * BT;

struct TnParams {
  const float* A;   // MFMA "A" operand source: D rows
  const float* B;   // MFMA "B" operand source: D cols (fast index of D)
  int64_t lda, ldb;
  int64_t K;
  int nrow, ncol;   // D is nrow x ncol (nrow = cols of A, ncol = cols of B)
  int ntr, ntc;     // tiles along rows / cols
  int ntiles;
  int syrk;
  int nsplit;
  int chunks_total;
  int chunks_per_split;
  double* P;        // [nsplit][ntiles][128*128]
};

// upper-triangle tile enumeration: super-rows of 4 tile rows, column-major
// inside a super-row (see header comment).
__device__ __host__ inline void decode_tri(int t, int nt, int& ta, int& tb) {
  int r0 = 0;
  for (;;) {
    int nrows = nt - r0 < 4 ? nt - r0 : 4;
    int ncols = nt - r0;
    int cnt = nrows * (nrows + 1) / 2 + (ncols - nrows) * nrows;
    if (t < cnt) break;
    t -= cnt;
    r0 += 4;
  }
  int nrows = nt - r0 < 4 ? nt - r0 : 4;
  int head = nrows * (nrows + 1) / 2;
  if (t < head) {
    int c = 0;
    while (t >= c + 1) { t -= c + 1; ++c; }
    tb = r0 + c;
    ta = r0 + t;
  } else {
    t -= head;
    tb = r0 + nrows + t / nrows;
    ta = r0 + t % nrows;
  }
}

__device__ inline void decode_tile(const TnParams& p, int t, int& ta, int& tb) {
  if (p.syrk) {
    decode_tri(t, p.ntr, ta, tb);
  } else {  // 4-row super-rows, column-major inside (same patch idea)
    int per_sr = 4 * p.ntc;
    int sr = t / per_sr;
    int r0 = sr * 4;
    int nrows = p.ntr - r0 < 4 ? p.ntr - r0 : 4;
    int tt = t - sr * per_sr;
    tb = tt / nrows;
    ta = r0 + tt % nrows;
  }
}

// full chunk: no guards at all (columns are clamped to valid ones by the caller)
template <bool ALIGNED>
__device__ inline f32x4 load4_full(const float* p) {
  if (ALIGNED) return *reinterpret_cast<const f32x4*>(p);
  f32x4 v = {p[0], p[1], p[2], p[3]};
  return v;
}
// K tail (last chunk of the last split only): rows >= kend read as zero
__device__ inline f32x4 load4_tail(const float* p, int64_t k, int64_t kend) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (k + e < kend) v[e] = p[e];
  return v;
}

template <bool ALIGNED>
__global__ __launch_bounds__(NTH, 2) void gemm_tn_partial_kernel(TnParams p) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * BT * LDT];
  float* As = lds;                 // [2][BT][LDT]
  float* Bs = lds + 2 * BT * LDT;  // [2][BT][LDT]

  // ---- unit decode (XCD-aware) ----
  const int total = gridDim.x;
  int b = blockIdx.x;
  int g = b >> 8;
  int pos = ((g << 8) + 256 <= total) ? (g << 8) + (b & 7) * 32 + ((b & 255) >> 3) : b;
  const int split = pos / p.ntiles;
  const int tile = pos - split * p.ntiles;
  int ta, tb;
  decode_tile(p, tile, ta, tb);
  const int row0 = ta * BT;  // D rows  <- columns of A
  const int col0 = tb * BT;  // D cols  <- columns of B

  const int c_begin = split * p.chunks_per_split;
  int c_end = c_begin + p.chunks_per_split;
  if (c_end > p.chunks_total) c_end = p.chunks_total;
  const int nchunks = c_end - c_begin;
  const int64_t kend = p.K;
  // index of the one chunk that may be partial (K % 32 != 0), else -1
  const int tail_chunk = (kend % BK) ? p.chunks_total - 1 : -1;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  // ---- staging assignment: 4 x 16-byte pieces per operand per thread.
  // Columns past the matrix edge are clamped onto the last valid column: they
  // only feed rows/columns of D that the reduce kernel never stores.
  const int scol = tid >> 3;  // + 32*i
  const int sq = tid & 7;     // k offset 4*sq
  const float* aptr[4];
  const float* bptr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int ca = row0 + scol + 32 * i;
    int cb = col0 + scol + 32 * i;
    ca = ca < p.nrow ? ca : p.nrow - 1;
    cb = cb < p.ncol ? cb : p.ncol - 1;
    aptr[i] = p.A + (int64_t)ca * p.lda + 4 * sq;
    bptr[i] = p.B + (int64_t)cb * p.ldb + 4 * sq;
  }
  const int sts = (scol * LDT + 4 * sq);  // + 32*i*LDT

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // ---- fp64 partial tile of this unit (owned: only this workgroup touches it).
  // Block q = (mi, ni) of the wave is folded into it every 32 chunks, the four
  // blocks staggered by 8 chunks so that at most one block's 16 old values are
  // in flight at a time (prefetched one chunk ahead of the fold).
  double* Pt = p.P + ((size_t)split * p.ntiles + tile) * TILE_ELEMS;
  const int lane_off = (64 * wr + 4 * lh) * BT + 64 * wc + l31;
  double oldv[16];
#define DMDX_BLOCK_OFF(mi, ni, r) ((32 * (mi) + ((r) & 3) + 8 * ((r) >> 2)) * BT + 32 * (ni))
#define DMDX_PREFETCH(mi, ni)                                                     \
  do {                                                                            \
    int lo_ = lane_off;                                                           \
    asm volatile("" : "+v"(lo_)); /* keep the 16 addresses out of loop-invariant hoisting */ \
    const double* q_ = Pt + lo_;                                                  \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) oldv[r] = q_[DMDX_BLOCK_OFF(mi, ni, r)]; \
  } while (0)
#define DMDX_COMMIT(mi, ni, have_old)                                             \
  do {                                                                            \
    int lo_ = lane_off;                                                           \
    asm volatile("" : "+v"(lo_));                                                 \
    double* q_ = Pt + lo_;                                                        \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                              \
      double v_ = (double)acc[mi][ni][r];                                         \
      if (have_old) v_ += oldv[r];                                                \
      q_[DMDX_BLOCK_OFF(mi, ni, r)] = v_;                                         \
      acc[mi][ni][r] = 0.f;                                                       \
    }                                                                             \
  } while (0)

  if (nchunks <= 0) {  // empty split: the partial tile must still be defined
    DMDX_COMMIT(0, 0, false);
    DMDX_COMMIT(0, 1, false);
    DMDX_COMMIT(1, 0, false);
    DMDX_COMMIT(1, 1, false);
    return;
  }

  f32x4 ra[4], rb[4];
  auto load_stage = [&](int chunk) {
    const int64_t k0 = (int64_t)chunk * BK;
    if (chunk != tail_chunk) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = load4_full<ALIGNED>(aptr[i] + k0);
        rb[i] = load4_full<ALIGNED>(bptr[i] + k0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = load4_tail(aptr[i] + k0, k0 + 4 * sq, kend);
        rb[i] = load4_tail(bptr[i] + k0, k0 + 4 * sq, kend);
      }
    }
  };
  auto store_stage = [&](int st) {
    float* as = As + st * BT * LDT + sts;
    float* bs = Bs + st * BT * LDT + sts;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<f32x4*>(as + 32 * i * LDT) = ra[i];
      *reinterpret_cast<f32x4*>(bs + 32 * i * LDT) = rb[i];
    }
  };

  // MFMA operand fragments, two register sets: the set for k-step t+1 is read
  // from LDS while the 16 MFMAs of k-step t run.
  f32x4 fa0[2], fb0[2], fa1[2], fb1[2];
  const int frag_a = (64 * wr + l31) * LDT + 4 * lh;
  const int frag_b = (64 * wc + l31) * LDT + 4 * lh;
#define DMDX_READ_FRAGS(FA, FB, st, t)                                               \
  do {                                                                               \
    const float* as_ = As + (st) * BT * LDT + frag_a + 8 * (t);                      \
    const float* bs_ = Bs + (st) * BT * LDT + frag_b + 8 * (t);                      \
    FA[0] = *reinterpret_cast<const f32x4*>(as_);                                    \
    FA[1] = *reinterpret_cast<const f32x4*>(as_ + 32 * LDT);                         \
    FB[0] = *reinterpret_cast<const f32x4*>(bs_);                                    \
    FB[1] = *reinterpret_cast<const f32x4*>(bs_ + 32 * LDT);                         \
  } while (0)
#define DMDX_MFMA16(FA, FB)                                                                   \
  do {                                                                                        \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                           \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[0][j], FB[0][j], acc[0][0], 0, 0, 0); \
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[0][j], FB[1][j], acc[0][1], 0, 0, 0); \
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[1][j], FB[0][j], acc[1][0], 0, 0, 0); \
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[1][j], FB[1][j], acc[1][1], 0, 0, 0); \
    }                                                                                         \
  } while (0)

  load_stage(c_begin);
  store_stage(0);
  __syncthreads();
  DMDX_READ_FRAGS(fa0, fb0, 0, 0);

  int cur = 0;
  for (int c = 0; c < nchunks; ++c) {
    const bool has_next = (c + 1 < nchunks);
    if (has_next) load_stage(c_begin + c + 1);
    const int phase = c & 7, fq = (c >> 3) & 3;
    if (phase == 6 && c >= 32) {  // old partial values of block fq, used one chunk later
      switch (fq) {
        case 0: DMDX_PREFETCH(0, 0); break;
        case 1: DMDX_PREFETCH(0, 1); break;
        case 2: DMDX_PREFETCH(1, 0); break;
        default: DMDX_PREFETCH(1, 1); break;
      }
    }

    DMDX_READ_FRAGS(fa1, fb1, cur, 1);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // 4 ds_read
    DMDX_MFMA16(fa0, fb0);
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);  // 16 mfma
    DMDX_READ_FRAGS(fa0, fb0, cur, 2);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    DMDX_MFMA16(fa1, fb1);
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
    DMDX_READ_FRAGS(fa1, fb1, cur, 3);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    DMDX_MFMA16(fa0, fb0);
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
    DMDX_MFMA16(fa1, fb1);

    if (has_next) store_stage(cur ^ 1);
    __syncthreads();
    cur ^= 1;
    if (has_next) DMDX_READ_FRAGS(fa0, fb0, cur, 0);
    if (phase == 7) {
      if (c >= 32) {
        switch (fq) {
          case 0: DMDX_COMMIT(0, 0, true); break;
          case 1: DMDX_COMMIT(0, 1, true); break;
          case 2: DMDX_COMMIT(1, 0, true); break;
          default: DMDX_COMMIT(1, 1, true); break;
        }
      } else {
        switch (fq) {
          case 0: DMDX_COMMIT(0, 0, false); break;
          case 1: DMDX_COMMIT(0, 1, false); break;
          case 2: DMDX_COMMIT(1, 0, false); break;
          default: DMDX_COMMIT(1, 1, false); break;
        }
      }
    }
  }
  // final fold of whatever each block still holds (block q was folded before iff
  // the unit ran at least 8q+8 chunks)
  if (nchunks >= 8) { DMDX_PREFETCH(0, 0); DMDX_COMMIT(0, 0, true); } else { DMDX_COMMIT(0, 0, false); }
  if (nchunks >= 16) { DMDX_PREFETCH(0, 1); DMDX_COMMIT(0, 1, true); } else { DMDX_COMMIT(0, 1, false); }
  if (nchunks >= 24) { DMDX_PREFETCH(1, 0); DMDX_COMMIT(1, 0, true); } else { DMDX_COMMIT(1, 0, false); }
  if (nchunks >= 32) { DMDX_PREFETCH(1, 1); DMDX_COMMIT(1, 1, true); } else { DMDX_COMMIT(1, 1, false); }
#undef DMDX_READ_FRAGS
#undef DMDX_MFMA16
#undef DMDX_PREFETCH
#undef DMDX_COMMIT
#undef DMDX_BLOCK_OFF
}

// Sum the K-splits in fp64 and scatter the tile into D (row-major view:
// D[i*ld + j], which is the column-major C of the C ABI with the operand
// roles swapped by the host wrapper).  SYRK mode also writes the mirror.
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(
    const double* P, int nsplit, int ntiles, int ntr, int ntc, int syrk, int nrow, int ncol,
    double* D64, int64_t ld64, float* D32, int64_t ld32) {
  __shared__ double tr[32][33];
  // one workgroup per (tile, 32x32 sub-block)
  const int tile = blockIdx.x >> 4;
  const int sb = blockIdx.x & 15;
  int ta, tb;
  if (syrk) {
    decode_tri(tile, ntr, ta, tb);
  } else {
    int per_sr = 4 * ntc;
    int sr = tile / per_sr;
    int r0 = sr * 4;
    int nrows = ntr - r0 < 4 ? ntr - r0 : 4;
    int tt = tile - sr * per_sr;
    tb = tt / nrows;
    ta = r0 + tt % nrows;
  }
  const int row0 = ta * BT, col0 = tb * BT;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const bool diag = syrk && (ta == tb);
  const int si = (sb >> 2) * 32, sj = (sb & 3) * 32;
  if (diag && si > sj) return;  // lower sub-blocks of a diagonal tile: mirrored from the upper ones
  if (row0 + si >= nrow || col0 + sj >= ncol) return;
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  const double* src = P + (size_t)tile * TILE_ELEMS + (si + ty) * BT + sj + tx;
  for (int sp = 0; sp < nsplit; ++sp) {
    const double* q = src + (size_t)sp * ntiles * TILE_ELEMS;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += q[8 * k * BT];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = si + ty + 8 * k, j = sj + tx;
    const int gi = row0 + i, gj = col0 + j;
    const bool keep = !(diag && i > j);
    if (keep && gi < nrow && gj < ncol) {
      D64[(int64_t)gi * ld64 + gj] = v[k];
      if (D32) D32[(int64_t)gi * ld32 + gj] = (float)v[k];
    }
    tr[ty + 8 * k][tx] = v[k];
  }
  if (syrk) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // mirrored element: source (i = si+tx, j = sj+ty+8k) -> D[gj][gi]
      const int i = si + tx, j = sj + ty + 8 * k;
      const double s = tr[tx][ty + 8 * k];
      const int gi = row0 + i, gj = col0 + j;
      const bool keep = !(diag && i >= j);
      if (keep && gi < nrow && gj < ncol) {
        D64[(int64_t)gj * ld64 + gi] = s;
        if (D32) D32[(int64_t)gj * ld32 + gi] = (float)s;
      }
    }
  }
}

struct Plan {
  int ntr, ntc, ntiles, nsplit, chunks_total, chunks_per_split;
  size_t ws_bytes;
};

Plan make_plan(int64_t K, int64_t nrow, int64_t ncol, int syrk) {
  Plan pl;
  pl.ntr = (int)((nrow + BT - 1) / BT);
  pl.ntc = (int)((ncol + BT - 1) / BT);
  pl.ntiles = syrk ? pl.ntr * (pl.ntr + 1) / 2 : pl.ntr * pl.ntc;
  pl.chunks_total = (int)((K + BK - 1) / BK);
  if (pl.chunks_total < 1) pl.chunks_total = 1;
  // aim at >= ~20 rounds of 512 resident workgroups, keep >= 8 chunks per split
  int64_t want = (20 * 512 + pl.ntiles - 1) / pl.ntiles;
  int64_t maxs = pl.chunks_total / 8;
  if (maxs < 1) maxs = 1;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 256) want = 256;
  pl.chunks_per_split = (int)((pl.chunks_total + want - 1) / want);
  pl.nsplit = (pl.chunks_total + pl.chunks_per_split - 1) / pl.chunks_per_split;
  pl.ws_bytes = (size_t)pl.nsplit * pl.ntiles * TILE_ELEMS * sizeof(double);
  return pl;
}

int run_tn(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t K, int64_t nrow,
           int64_t ncol, int syrk, double* D64, int64_t ld64, float* D32, int64_t ld32, void* ws,
           size_t ws_bytes, hipStream_t stream) {
  Plan pl = make_plan(K, nrow, ncol, syrk);
  if (ws == nullptr || ws_bytes < pl.ws_bytes) {
    dmdx_set_error("gemm_tn: workspace %zu bytes < required %zu", ws_bytes, pl.ws_bytes);
    return DMDX_E_WORKSPACE;
  }
  TnParams p;
  p.A = A; p.B = B; p.lda = lda; p.ldb = ldb; p.K = K;
  p.nrow = (int)nrow; p.ncol = (int)ncol;
  p.ntr = pl.ntr; p.ntc = pl.ntc; p.ntiles = pl.ntiles; p.syrk = syrk;
  p.nsplit = pl.nsplit; p.chunks_total = pl.chunks_total;
  p.chunks_per_split = pl.chunks_per_split;
  p.P = reinterpret_cast<double*>(ws);
  const bool aligned = (lda % 4 == 0) && (ldb % 4 == 0) && dmdx_aligned16(A) && dmdx_aligned16(B);
  dim3 grid((unsigned)((size_t)pl.nsplit * pl.ntiles));
  if (aligned)
    hipLaunchKernelGGL(gemm_tn_partial_kernel<true>, grid, dim3(NTH), 0, stream, p);
  else
    hipLaunchKernelGGL(gemm_tn_partial_kernel<false>, grid, dim3(NTH), 0, stream, p);
  DMDX_LAUNCH_CHECK();
  hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(pl.ntiles * 16), dim3(256), 0, stream, p.P, pl.nsplit,
                     pl.ntiles, pl.ntr, pl.ntc, syrk, (int)nrow, (int)ncol, D64, ld64, D32, ld32);
  DMDX_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" {

size_t dmdx_syrk_workspace_bytes(int64_t m, int64_t n) {
  if (m < 0 || n <= 0) return 0;
  return make_plan(m, n, n, 1).ws_bytes;
}

int dmdx_syrk_f32(const float* X, int64_t m, int64_t n, int64_t ldx, double* G64, int64_t ldg,
                  float* G32, int64_t ldg32, void* workspace, size_t workspace_bytes,
                  void* stream) {
  DMDX_CHECK_ARG(X && G64, "syrk: null pointer");
  DMDX_CHECK_ARG(m >= 1 && n >= 1 && n < (1 << 30), "syrk: bad shape m=%lld n=%lld", (long long)m,
                 (long long)n);
  DMDX_CHECK_ARG(ldx >= 1 && ldg >= n && (!G32 || ldg32 >= n), "syrk: bad leading dimension");
  return run_tn(X, ldx, X, ldx, m, n, n, 1, G64, ldg, G32, ldg32, workspace, workspace_bytes,
                (hipStream_t)stream);
}

size_t dmdx_gemm_tn_workspace_bytes(int64_t K, int64_t na, int64_t nb) {
  if (K < 0 || na <= 0 || nb <= 0) return 0;
  return make_plan(K, nb, na, 0).ws_bytes;
}

int dmdx_gemm_tn_f32(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t K,
                     int64_t na, int64_t nb, double* C64, int64_t ldc, float* C32, int64_t ldc32,
                     void* workspace, size_t workspace_bytes, void* stream) {
  DMDX_CHECK_ARG(A && B && C64, "gemm_tn: null pointer");
  DMDX_CHECK_ARG(K >= 1 && na >= 1 && nb >= 1 && na < (1 << 30) && nb < (1 << 30),
                 "gemm_tn: bad shape");
  DMDX_CHECK_ARG(lda >= 1 && ldb >= 1 && ldc >= na && (!C32 || ldc32 >= na),
                 "gemm_tn: bad leading dimension");
  // column-major C[a + b*ldc] == row-major D[b][a]: D rows <- B columns, D cols <- A columns
  return run_tn(B, ldb, A, lda, K, nb, na, 0, C64, ldc, C32, ldc32, workspace, workspace_bytes,
                (hipStream_t)stream);
}

}  // extern "C"
