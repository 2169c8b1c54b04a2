// K1 / K3: D = OpA^T OpB for K-contiguous (column-major, tall) operands, on the
// fp32-input MFMA (v_mfma_f32_32x32x2_f32).  K1 (Gram, G = X^T X) is the SYRK
// mode: OpA = OpB = X and only the tiles of one triangle are computed.
//
// Decomposition
//   work unit = (K-split s, 128x128 output tile t); one 256-thread workgroup
//   (4 waves, 2x2, each wave a 64x64 sub-tile = 2x2 MFMA blocks of 32x32) per
//   unit, two workgroups resident per CU.
//   K loop: 32-row chunks through two LDS stages.  Aligned operands are staged by
//   LDS-DMA (global_load_lds_dwordx4 along K, the contiguous axis of both
//   operands; SGPR base + 32-bit per-lane byte offset, no VALU per piece); the LDS
//   panels are unpadded [column][32 k] with the 16-byte k-chunks XOR-swizzled by
//   (column >> 1) & 7 on the DMA source and on the fragment reads, which makes the
//   ds_read_b128 fragment reads bank-conflict free.  Unaligned operands and the
//   K tail go through registers.  One ds_read_b128 per operand block feeds four
//   MFMAs: lane (r, h) reads k = 8t+4h .. 8t+4h+3 of its column, MFMA j uses
//   element j of both operands -- the same k permutation on both sides, so
//   the contraction is unchanged.
//   Numerics: fp32 MFMA chains of at most 128 chunks * 32 = 4096 rows; the chain
//   results of a unit (at most 16) are summed in a second fp32 register
//   accumulator (blocked summation: the error bound stays that of a 4096-row
//   chain) and stored once, as fp64, into the unit's own partial tile (no
//   atomics => deterministic); a second kernel sums the K-splits in fp64 and
//   writes D (both triangles in SYRK mode).  Row blocks are accumulated into D
//   in fp64 (`accumulate`).
//
//   Chunk pipeline: both stages filled up front; chunk c runs k-steps 0..2, then the
//   barrier, then k-step 3 (its fragments are already in registers), the refill of the
//   freed stage with chunk c+2 and the first fragment reads of chunk c+1.
//
// L2 locality: units are ordered tile-fastest inside a K-split, the tiles of
// the triangle are enumerated in 8-row super-rows, column by column, so 64
// consecutive units form an 8x8 patch of tiles; blockIdx is remapped so that
// the 64 blocks of a group of 512 that land on one XCD (blockIdx % 8 equal)
// take one patch (xcd_unit).
//
// Entry points: dmdx_syrk_f32 / dmdx_gemm_tn_f32 (one pair of operands per launch) and
// dmdx_syrk_blocks_f32 / dmdx_gemm_tn_blocks_f32 (the sum over up to 16 row blocks per
// launch, syrk_batch_kernel); all of them run tn_unit, the reduce kernel sums the K-splits.
// Environment knobs (tuning / diagnosis only): DMDX_TN_MAX_CPS (chunks per unit, default 2048),
// DMDX_TN_ROUNDS (rounds of 512 workgroups the K-splits aim at), DMDX_TN_ABLATE (timing-only
// ablations of the single-launch kernel: results are wrong).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "dmdx_common.h"

thread_local unsigned long long* dmdx_clock_probe_ptr = nullptr;  // dmdx_set_clock_probe (measurement aid)

int dmdx_device_cus() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  if (cached[dev] == 0) {
    int v = 0;
    cached[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : -1;
  }
  return cached[dev];
}

namespace {

constexpr int BT = 128;            // output tile edge
constexpr int BK = 32;             // K rows per stage
constexpr int FOLD = 128;          // chunks per fp32 chain (128 * 32 = 4096 rows) before it is folded into acc2
constexpr int NTH = 256;

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
  double* P;        // [slab][ntiles][tile rows x 128] fp64 partial tiles; slab = K-split (of a block)
};

// A batch of row blocks in one launch: the Gram of a row-blocked snapshot matrix (and X^T Y) is
// the sum over its blocks, and one launch per block leaves the GPU draining / refilling 8 times
// (last partial round of units, reduce kernel, launch gap: ~1.5 % at cfg2).  Block j owns the
// grid range [unit_begin[j], unit_begin[j+1]) (padded to multiples of 512 so that the XCD
// patch mapping stays aligned; padding blocks exit at once) and the partial-tile slabs
// [slab_begin[j], slab_begin[j+1]); the reduce kernel sums all slabs.
constexpr int MAXB = 16;
struct TnBatch {
  const float* A[MAXB];   // D rows <- columns of A_j; SYRK: A_j == B_j == X_j
  const float* B[MAXB];   // D cols <- columns of B_j
  int64_t lda[MAXB];
  int64_t ldb[MAXB];
  int64_t K[MAXB];
  int chunks_total[MAXB];
  int chunks_per_split[MAXB];
  int nsplit[MAXB];
  int unit_begin[MAXB + 1];
  int slab_begin[MAXB + 1];
  int nblocks;
  // measurement aid (dmdx_set_clock_probe; null on the product path): every workgroup adds the
  // core-clock cycles and the 100 MHz reference ticks of its lifetime and 1 to clk[0..2]
  unsigned long long* clk;
};

// upper-triangle tile enumeration: super-rows of SR tile rows, column-major
// inside a super-row (see header comment).
constexpr int SR = 8;  // tile rows per super-row: 64 consecutive tiles ~ an 8x8 patch
__device__ __host__ inline void decode_tri(int t, int nt, int& ta, int& tb) {
  int r0 = 0;
  for (;;) {
    int nrows = nt - r0 < SR ? nt - r0 : SR;
    int ncols = nt - r0;
    int cnt = nrows * (nrows + 1) / 2 + (ncols - nrows) * nrows;
    if (t < cnt) break;
    t -= cnt;
    r0 += SR;
  }
  int nrows = nt - r0 < SR ? nt - r0 : SR;
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
  } else {  // SR-row super-rows, column-major inside (same patch idea)
    int per_sr = SR * p.ntc;
    int sr = t / per_sr;
    int r0 = sr * SR;
    int nrows = p.ntr - r0 < SR ? p.ntr - r0 : SR;
    int tt = t - sr * per_sr;
    tb = tt / nrows;
    ta = r0 + tt % nrows;
  }
}

// LDS image of one operand stage: [128 columns][32 k] floats, 128 B per column and NOT
// padded (an LDS-DMA wave-instruction writes 1 KiB linearly = 8 columns).  Bank
// conflicts of the ds_read_b128 fragment reads are removed by an XOR swizzle of the
// 16-byte k-chunk index with f(col) = (col >> 1) & 7, applied on the SOURCE address of
// the DMA (and on the register-staged tail path) and on the read address.
__device__ __forceinline__ int swz(int col) { return (col >> 1) & 7; }

// K tail (last chunk of the last split only): rows >= kend read as zero
__device__ inline f32x4 load4_tail(const float* p, int64_t k, int64_t kend) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (k + e < kend) v[e] = p[e];
  return v;
}

#ifdef DMDX_STAMPS
// diagnostic build only (make stamps): per-segment cycle totals of the chunk loop, summed over
// all waves: [0] DMA issue, [1] fragment reads + MFMA issue, [5] wait for the LDS-DMA (vmcnt),
// [2] barrier, [3] post-barrier (first fragment read, fold), [4] chunks counted,
// [6] / [7] chunk-loop time in core cycles / in 100 MHz ticks (their ratio = sustained clock)
__device__ unsigned long long dmdx_stamp[12];   // [8] unit prologue, [9] partial-tile commit (stores drained), [10] waves counted
#define DMDX_STAMP(var)                                                        \
  do {                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                         \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                         \
  } while (0)
#else
#define DMDX_STAMP(var) do { } while (0)
#endif

#define DMDX_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define DMDX_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// DMA = true : global -> LDS by global_load_lds_dwordx4 (needs 16-byte aligned
//              bases and leading dimensions); the K-tail chunk goes through registers.
// DMA = false: everything register-staged with scalar loads (any alignment).
// ABL: timing-only ablations for diagnosis (results are wrong when ABL != 0):
//   1 no global->LDS staging, 2 no barrier, 8 no chain fold (none of them changes an address).  Selected by DMDX_TN_ABLATE.
// SK ("skinny rows", 0 / 1 / 2 / 3): a (32 SK) x 128 output tile, the four waves side by side (each
//   32 SK rows x 32 columns = SK x 1 MFMA blocks) -- for D with few rows (Z = X^T Y with l <= 32,
//   <= 64 or <= 96 columns of Y): a quarter / half / three quarters of the MFMA work of a 128-row
//   tile that would be padding for the rest.
// One work unit: the K-range of `split` of the TM x 128 output tile at (row0, col0) (D rows <-
// columns of A, D cols <- columns of B), written as fp64 into the partial tile Pt.
// lds: 2 stages of (TM + 128) x 32 floats.
// H16 (round 3): one more 16-row block below the SK 32-row blocks, on v_mfma_f32_16x16x4_f32 (same flop
//   rate, half the rows): tile heights 48 / 80 / 112, so that l = 70 columns of Y (BASELINE config 4:
//   rank 50 + 20 oversamples) run 80 rows instead of 96.  Its fragments are 16-byte reads, one per operand
//   and half chunk (lane (i = lane & 15, kk = lane >> 4) takes the piece 4 u + kk of its column: MFMA e
//   contracts k = 16 u + 4 kk + e, two MFMAs per k-step and 16-column half of the wave's 32 columns).
//   (8-byte reads per k-step -- the first version -- were two-way bank-conflicted: columns 2j and 2j + 1
//   share a bank group under the piece swizzle; PMC: 27 % of the LDS-active cycles.)
template <bool DMA, int ABL, int SK, int H16 = 0>
__device__ __forceinline__ void tn_unit(const TnParams& p, const int split, const int row0, const int col0,
                                        double* Pt, float* lds) {
  static_assert(!H16 || SK >= 1, "a 16-row block only below at least one 32-row block");
#ifdef DMDX_STAMPS
  unsigned long long pt0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pt0)::"memory");
#endif
  constexpr int TM = SK ? 32 * SK + 16 * H16 : BT;  // tile rows    (columns of OpA)
  constexpr int MI = SK ? SK : 2;        // 32-row MFMA blocks per wave
  constexpr int NI = SK ? 1 : 2;         // 32-column MFMA blocks per wave
  constexpr int NPA = (TM + 31) / 32;    // 1 KiB pieces of the A panel per wave (or 16 B pieces per thread)
  constexpr int NPIECE = TM / 8;         // 1 KiB pieces of the A panel in all (H16: wave w takes pieces w, w + 4, ...)
  constexpr int OPA = TM * BK, OPB = BT * BK, STG = OPA + OPB;  // floats per stage
  // stage st: A panel at lds + st * STG, B panel at lds + st * STG + OPA

  const int c_begin = split * p.chunks_per_split;
  int c_end = c_begin + p.chunks_per_split;
  if (c_end > p.chunks_total) c_end = p.chunks_total;
  const int nchunks = c_end - c_begin;
  const int64_t kend = p.K;
  // index of the one chunk that may be partial (K % 32 != 0), else -1
  const int tail_chunk = (kend % BK) ? p.chunks_total - 1 : -1;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = SK ? 0 : wave >> 1, wc = SK ? wave : wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  // ---- staging assignment.  Columns past the matrix edge are clamped onto the last
  // valid column: they only feed rows/columns of D that the reduce kernel never stores.
  // DMA: wave w, instruction i (0..3) fills columns 32w + 8i + (lane >> 3), the lane's
  // 16-byte slot (lane & 7) holds global k-chunk (lane & 7) ^ swz(col).
  // Register path: thread -> column (tid >> 3) + 32 i, k-chunk (tid & 7).
  // addresses = wave-uniform tile base (SGPRs, advanced along K) + 32-bit per-lane element
  // offset: the LDS-DMA then needs no per-instruction 64-bit VALU add
  const int ra0 = row0 < p.nrow ? row0 : p.nrow - 1;
  const int cb0 = col0 < p.ncol ? col0 : p.ncol - 1;
  const float* Abase = p.A + (int64_t)ra0 * p.lda;
  const float* Bbase = p.B + (int64_t)cb0 * p.ldb;
  unsigned aoff[NPA], boff[4];
  int stsa[NPA], stsb[4];  // register path: float offset inside an operand stage
  int kqa[NPA], kqb[4];    // 4 * (global k-chunk of this piece)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int lc = DMA ? 32 * wave + 8 * i + (lane >> 3) : (tid >> 3) + 32 * i;  // local column
    const int q = DMA ? ((lane & 7) ^ swz(lc)) : (tid & 7);                       // global k-chunk
    int cb = col0 + lc;
    cb = cb < p.ncol ? cb : p.ncol - 1;
    boff[i] = (unsigned)((int64_t)(cb - cb0) * p.ldb + 4 * q);  // < 128 * ld: host checks it fits
    stsb[i] = lc * BK + 4 * (DMA ? (lane & 7) : ((tid & 7) ^ swz(lc)));
    kqb[i] = 4 * q;
  }
#pragma unroll
  for (int i = 0; i < NPA; ++i) {
    // (H16: the A panel has TM / 8 pieces, not a multiple of 4: wave w takes the pieces w + 4 i < NPIECE;
    // register path: the columns (tid >> 3) + 32 i < TM)
    const int lc = DMA ? (H16 ? 8 * (wave + 4 * i) : (TM / 4) * wave + 8 * i) + (lane >> 3) : (tid >> 3) + 32 * i;
    const int q = DMA ? ((lane & 7) ^ swz(lc)) : (tid & 7);
    int ca = row0 + (lc < TM ? lc : TM - 1);
    ca = ca < p.nrow ? ca : p.nrow - 1;
    aoff[i] = (unsigned)((int64_t)(ca - ra0) * p.lda + 4 * q);
    stsa[i] = lc * BK + 4 * (DMA ? (lane & 7) : ((tid & 7) ^ swz(lc)));
    kqa[i] = 4 * q;
  }

  // LDS-DMA addressing: wave-uniform byte pointer (SGPR pair, advanced along K) + 32-bit
  // per-lane BYTE offset, the `saddr` form of global_load_lds -- no VALU work per piece.
  // Piece i of an operand lands at LDS base (M0) + 1024 i through the instruction's immediate
  // offset, which the hardware adds to the global address too: offsets carry + 3072 - 1024 i
  // and the base pointer - 3072, so that every offset stays non-negative.
  const char* Adma = reinterpret_cast<const char*>(Abase) - (H16 ? 0 : 3072);
  const char* Bdma = reinterpret_cast<const char*>(Bbase) - 3072;
  unsigned aoffb[NPA], boffb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) boffb[i] = 4u * boff[i] + 3072u - 1024u * i;
#pragma unroll
  for (int i = 0; i < NPA; ++i) aoffb[i] = H16 ? 4u * aoff[i] : 4u * aoff[i] + 3072u - 1024u * i;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // ---- two-level fp32 accumulation inside a unit, fp64 across units.
  // acc holds fp32 chains of at most FOLD*32 = 4096 rows.  Every FOLD chunks block
  // q = (mi, ni) of the wave is added into acc2 (registers) and cleared; the four blocks
  // are staggered by FOLD/4 chunks.  A unit runs <= max_cps chunks, so acc2 sums <= 16
  // chain results: the rounding error stays that of a 4096-row fp32 chain (blocked
  // summation), and the loop touches no memory except the operand stream.  (Folding
  // into an fp64 tile in HBM instead cost 4.6 % -- ~115 VALU/VMEM instructions per fold,
  // each of which waits for an MFMA slot while the SIMD's other wave streams MFMAs.)
  // At the end of the unit acc + acc2 is stored once, as fp64, into the unit's partial tile.
  f32x16 acc2[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[mi][ni][r] = 0.f;
  // the 16-row block: two 16 x 16 results (column halves of the wave's 32 columns)
  f32x4 acch[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, acch2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const int l15 = lane & 15, lkk = lane >> 4;
  const int lane_off = (64 * wr + 4 * lh) * BT + (SK ? 32 : 64) * wc + l31;
#define DMDX_BLOCK_OFF(mi, ni, r) ((32 * (mi) + ((r) & 3) + 8 * ((r) >> 2)) * BT + 32 * (ni))
#define DMDX_FOLD(mi, ni)                                                         \
  do {                                                                            \
    acc2[mi][ni] += acc[mi][ni];                                                  \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;          \
  } while (0)
#define DMDX_COMMIT(mi, ni)                                                       \
  do {                                                                            \
    double* q_ = Pt + lane_off;                                                   \
    _Pragma("unroll") for (int r = 0; r < 16; ++r)                                \
        __builtin_nontemporal_store((double)(acc[mi][ni][r] + acc2[mi][ni][r]),   \
                                    q_ + DMDX_BLOCK_OFF(mi, ni, r));              \
  } while (0)
#define DMDX_COMMIT_H16()                                                         \
  do {                                                                            \
    _Pragma("unroll") for (int hh = 0; hh < 2; ++hh)                              \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                             \
            __builtin_nontemporal_store((double)(acch[hh][r] + acch2[hh][r]),     \
                                        Pt + (32 * SK + 4 * lkk + r) * BT + 32 * wc + 16 * hh + l15); \
  } while (0)
#define DMDX_COMMIT_ALL()                                                         \
  do {                                                                            \
    if constexpr (H16 != 0) DMDX_COMMIT_H16();                                    \
    DMDX_COMMIT(0, 0);                                                            \
    if constexpr (MI >= 2) DMDX_COMMIT(MI >= 2 ? 1 : 0, 0);                       \
    if constexpr (NI == 2) {                                                      \
      DMDX_COMMIT(0, NI - 1);                                                     \
      DMDX_COMMIT(1, NI - 1);                                                     \
    }                                                                             \
    if constexpr (MI == 3) DMDX_COMMIT(MI - 1, 0);                                \
  } while (0)

  if (nchunks <= 0) {  // empty split: the partial tile must still be defined
    DMDX_COMMIT_ALL();
    return;
  }

  // ---- global -> LDS for one chunk into stage st
  auto stage_regs = [&](int chunk, int st, bool tail) {  // register path (tail / unaligned)
    const int64_t k0 = (int64_t)chunk * BK;
    f32x4 ra[NPA], rb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* pb = Bbase + k0 + boff[i];
      rb[i] = tail ? load4_tail(pb, k0 + kqb[i], kend) : f32x4{pb[0], pb[1], pb[2], pb[3]};
    }
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const float* pa = Abase + k0 + aoff[i];
      ra[i] = tail ? load4_tail(pa, k0 + kqa[i], kend) : f32x4{pa[0], pa[1], pa[2], pa[3]};
    }
    // (H16: the last piece / the last group of 32 columns is only half inside the panel)
    const bool a_last_ok = !H16 || (DMA ? wave + 4 * (NPA - 1) < NPIECE : (tid >> 3) + 32 * (NPA - 1) < TM);
    float* as = lds + st * STG;
    float* bs = lds + st * STG + OPA;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(bs + stsb[i]) = rb[i];
#pragma unroll
    for (int i = 0; i < NPA; ++i)
      if (i + 1 < NPA || a_last_ok) *reinterpret_cast<f32x4*>(as + stsa[i]) = ra[i];
  };
// The 8 pieces of the next chunk in one asm block: 2 SALU (M0) + 8 VMEM, no VALU.  A wave
// whose SIMD partner streams MFMAs gets ~one instruction issued per MFMA slot (stamps:
// ~64-95 cycles each), so every VALU address add in this phase cost a whole MFMA slot.
#define DMDX_DMA_OP4(ptr, la, o0, o1, o2, o3)                                                      \
  asm volatile("s_mov_b32 m0, %[la_]\n\ts_nop 0\n\t"                                               \
               "global_load_lds_dwordx4 %[a0], %[ap_]\n\t"                                         \
               "global_load_lds_dwordx4 %[a1], %[ap_] offset:1024\n\t"                             \
               "global_load_lds_dwordx4 %[a2], %[ap_] offset:2048\n\t"                             \
               "global_load_lds_dwordx4 %[a3], %[ap_] offset:3072"                                 \
               :: [la_] "s"(la), [ap_] "s"(ptr), [a0] "v"(o0), [a1] "v"(o1), [a2] "v"(o2), [a3] "v"(o3) \
               : "memory")
#define DMDX_DMA_OP3(ptr, la, o0, o1, o2)                                                          \
  asm volatile("s_mov_b32 m0, %[la_]\n\ts_nop 0\n\t"                                               \
               "global_load_lds_dwordx4 %[a0], %[ap_]\n\t"                                         \
               "global_load_lds_dwordx4 %[a1], %[ap_] offset:1024\n\t"                             \
               "global_load_lds_dwordx4 %[a2], %[ap_] offset:2048"                                 \
               :: [la_] "s"(la), [ap_] "s"(ptr), [a0] "v"(o0), [a1] "v"(o1), [a2] "v"(o2)           \
               : "memory")
#define DMDX_DMA_OP1(ptr, la, o0)                                                                  \
  asm volatile("s_mov_b32 m0, %[la_]\n\ts_nop 0\n\t"                                               \
               "global_load_lds_dwordx4 %[a0], %[ap_]"                                             \
               :: [la_] "s"(la), [ap_] "s"(ptr), [a0] "v"(o0)                                       \
               : "memory")
#define DMDX_DMA_OP2(ptr, la, o0, o1)                                                              \
  asm volatile("s_mov_b32 m0, %[la_]\n\ts_nop 0\n\t"                                               \
               "global_load_lds_dwordx4 %[a0], %[ap_]\n\t"                                         \
               "global_load_lds_dwordx4 %[a1], %[ap_] offset:1024"                                 \
               :: [la_] "s"(la), [ap_] "s"(ptr), [a0] "v"(o0), [a1] "v"(o1)                         \
               : "memory")
#define DMDX_DMA_A(ap, la)                                                                         \
  do {                                                                                             \
    if constexpr (H16 != 0) {                                                                      \
      _Pragma("unroll") for (int i_ = 0; i_ < NPA; ++i_)                                           \
          if (i_ + 1 < NPA || wave + 4 * (NPA - 1) < NPIECE) DMDX_DMA_OP1(ap, (la) + 4096u * i_, aoffb[i_]); \
    } else if constexpr (NPA == 4) DMDX_DMA_OP4(ap, la, aoffb[0], aoffb[1], aoffb[NPA - 2], aoffb[NPA - 1]); \
    else if constexpr (NPA == 3) DMDX_DMA_OP3(ap, la, aoffb[0], aoffb[1], aoffb[NPA - 1]);         \
    else if constexpr (NPA == 2) DMDX_DMA_OP2(ap, la, aoffb[0], aoffb[NPA - 1]);                   \
    else DMDX_DMA_OP1(ap, la, aoffb[0]);                                                           \
  } while (0)
#define DMDX_DMA_B(bp, lb) DMDX_DMA_OP4(bp, lb, boffb[0], boffb[1], boffb[2], boffb[3])
#define DMDX_DMA_NEXT(ap, bp, la, lb) \
  do {                                \
    DMDX_DMA_A(ap, la);               \
    DMDX_DMA_B(bp, lb);               \
  } while (0)
  auto stage_dma = [&](int chunk, int st) {
    const int64_t k0 = (int64_t)chunk * BK;
    float* as = lds + st * STG + (H16 ? 8 * wave : (TM / 4) * wave) * BK;
    float* bs = lds + st * STG + OPA + (32 * wave) * BK;
    DMDX_DMA_NEXT(Adma + 4 * k0, Bdma + 4 * k0, (unsigned)(uintptr_t)DMDX_LDS_PTR(as),
                  (unsigned)(uintptr_t)DMDX_LDS_PTR(bs));
  };
  auto stage = [&](int chunk, int st) {
    if (DMA) {
      if (chunk != tail_chunk) stage_dma(chunk, st);
      else stage_regs(chunk, st, true);
    } else {
      stage_regs(chunk, st, chunk == tail_chunk);
    }
  };

  // ---- MFMA operand fragments, two register sets: the set for k-step t+1 is read from
  // LDS while the 16 MFMAs of k-step t run.  Lane (r = lane&31, h = lane>>5) reads, for
  // k-step t, the 16-byte k-chunk (2t + h) of its column, stored at slot (2t+h) ^ swz.
  f32x4 fa0[MI], fb0[NI], fa1[MI], fb1[NI];
  // the 16-row block: one 16-byte read per operand and HALF chunk (k-steps 2u, 2u + 1): lane (i = lane & 15,
  // kk = lane >> 4) takes the piece 4u + kk of its column, MFMA e (two per k-step) contracts k = 16u + 4kk + e
  f32x4 hA[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  f32x4 hB[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
  int foff[4], foffh[2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    foff[t] = l31 * BK + 4 * ((2 * t + lh) ^ swz(l31));
    if (t < 2) foffh[t] = l15 * BK + 4 * ((4 * t + lkk) ^ swz(l15));
  }
  const int frag_a = 64 * wr * BK, frag_b = (SK ? 32 : 64) * wc * BK;
#define DMDX_READ_FRAGS(FA, FB, st, t)                                               \
  do {                                                                               \
    if constexpr (H16 != 0) {                                                        \
      if constexpr ((t) % 2 == 0) {                                                  \
        const float* ah_ = lds + (st) * STG + 32 * SK * BK + foffh[(t) / 2];         \
        const float* bh_ = lds + (st) * STG + OPA + frag_b + foffh[(t) / 2];         \
        hA[(t) / 2] = *reinterpret_cast<const f32x4*>(ah_);                          \
        hB[(t) / 2][0] = *reinterpret_cast<const f32x4*>(bh_);                       \
        hB[(t) / 2][1] = *reinterpret_cast<const f32x4*>(bh_ + 16 * BK);             \
      }                                                                              \
    }                                                                                \
    const float* as_ = lds + (st) * STG + frag_a + foff[t];                          \
    const float* bs_ = lds + (st) * STG + OPA + frag_b + foff[t];                    \
    FA[0] = *reinterpret_cast<const f32x4*>(as_);                                    \
    if constexpr (MI >= 2) FA[MI >= 2 ? 1 : 0] = *reinterpret_cast<const f32x4*>(as_ + 32 * BK); \
    if constexpr (MI == 3) FA[MI - 1] = *reinterpret_cast<const f32x4*>(as_ + 64 * BK); \
    FB[0] = *reinterpret_cast<const f32x4*>(bs_);                                    \
    if constexpr (NI == 2) FB[NI - 1] = *reinterpret_cast<const f32x4*>(bs_ + 32 * BK); \
  } while (0)

  // ---- chunk pipeline.  Both stages are filled up front; chunk c computes k-steps 0..2 from
  // stage `cur`, then the barrier (all waves are done reading `cur`: the k-step-3 fragments
  // are already in registers; chunk c+1 has landed in the other stage), then the refill of
  // `cur` with chunk c+2 and the first fragment reads of chunk c+1 are issued and k-step 3
  // runs behind them -- the wave leaves the barrier with 16 MFMAs to issue while its LDS
  // reads and LDS-DMA pieces are in flight, and every LDS-DMA has a whole chunk to land.
  // Measured on cfg2 blocks: barrier after k-step 3 + refill before it 85.6 %, this order
  // 87.4 %; pieces issued between the MFMAs of k-step 3 81 %; s_setprio 3 around the
  // refill 68 %.
  stage(c_begin, 0);
  if (nchunks > 1 && !(ABL & 1)) stage(c_begin + 1, 1);
  if (DMA) __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0), see the main loop
  __syncthreads();
  DMDX_READ_FRAGS(fa0, fb0, 0, 0);

#define DMDX_MFMA4(FA, FB, j)                                                                   \
  do {                                                                                          \
    _Pragma("unroll") for (int mi_ = 0; mi_ < MI; ++mi_)                                        \
        _Pragma("unroll") for (int ni_ = 0; ni_ < NI; ++ni_) acc[mi_][ni_] =                    \
            __builtin_amdgcn_mfma_f32_32x32x2f32(FA[mi_][j], FB[ni_][j], acc[mi_][ni_], 0, 0, 0); \
  } while (0)
  // k-step: [MI + NI ds_read of the next fragments] 4 * MI * NI MFMA
#define DMDX_KSTEP(FA, FB, i)                                      \
  do {                                                             \
    DMDX_MFMA4(FA, FB, 0);                                         \
    DMDX_MFMA4(FA, FB, 1);                                         \
    DMDX_MFMA4(FA, FB, 2);                                         \
    DMDX_MFMA4(FA, FB, 3);                                         \
    if constexpr (H16 != 0) {                                      \
      _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)             \
          _Pragma("unroll") for (int hh_ = 0; hh_ < 2; ++hh_) acch[hh_] = \
              __builtin_amdgcn_mfma_f32_16x16x4f32(hA[(i) / 2][2 * ((i) % 2) + s_], hB[(i) / 2][hh_][2 * ((i) % 2) + s_], acch[hh_], 0, 0, 0); \
    }                                                              \
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * MI * NI + 4 * H16, 0);   \
  } while (0)

#ifdef DMDX_STAMPS
  unsigned long long rt0, rt1, ct0;
  asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0), "=s"(ct0)::"memory");
  unsigned long long st0 = 0, st1 = 0, st2 = 0, st2b = 0, st3 = 0, st4 = 0, sa0 = 0, sa1 = 0, sa2 = 0, sa3 = 0, sa5 = 0;
#endif
  int cur = 0;
  for (int c = 0; c < nchunks; ++c) {
    DMDX_STAMP(st0);
    const bool has_next = (c + 1 < nchunks) && !(ABL & 1);
    const int phase = c & (FOLD / 4 - 1), fq = (c / (FOLD / 4)) & 3;
    DMDX_STAMP(st1);
    DMDX_READ_FRAGS(fa1, fb1, cur, 1);
    __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);   // ds_reads of the next fragments (the 16-row block reads at even k-steps)
    DMDX_KSTEP(fa0, fb0, 0);
    DMDX_READ_FRAGS(fa0, fb0, cur, 2);
    __builtin_amdgcn_sched_group_barrier(0x100, MI + NI + 3 * H16, 0);
    DMDX_KSTEP(fa1, fb1, 1);
    DMDX_READ_FRAGS(fa1, fb1, cur, 3);
    __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);
    DMDX_KSTEP(fa0, fb0, 2);
    DMDX_STAMP(st2);

    if (DMA) __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the asm LDS-DMA pieces are invisible to the compiler's counters
    DMDX_STAMP(st2b);
    if (!(ABL & 2)) __syncthreads();
    if (c + 2 < nchunks && !(ABL & 1)) stage(c_begin + c + 2, cur);
    cur ^= 1;
    DMDX_STAMP(st3);
    if (has_next) DMDX_READ_FRAGS(fa0, fb0, cur, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, MI + NI + 3 * H16, 0);
    DMDX_KSTEP(fa1, fb1, 3);
    if (!(ABL & 8) && phase == FOLD / 4 - 1) {
      switch (fq) {
        case 0: DMDX_FOLD(0, 0); break;
        case 1:
          if constexpr (NI == 2) { DMDX_FOLD(0, NI - 1); }
          else if constexpr (MI == 3) { DMDX_FOLD(MI - 1, 0); }
          break;
        case 2: if constexpr (MI >= 2) { DMDX_FOLD(MI >= 2 ? 1 : 0, 0); } break;
        default:
          if constexpr (NI == 2) { DMDX_FOLD(1, NI - 1); }
          if constexpr (H16 != 0) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              acch2[hh] += acch[hh];
              acch[hh] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
          }
          break;
      }
    }
#ifdef DMDX_STAMPS
    DMDX_STAMP(st4);
    sa0 += st1 - st0; sa1 += st2 - st1; sa2 += st3 - st2b; sa3 += st4 - st3; sa5 += st2b - st2;
#endif
  }
#ifdef DMDX_STAMPS
  if (lane == 0) {
    atomicAdd(&dmdx_stamp[0], sa0); atomicAdd(&dmdx_stamp[1], sa1);
    atomicAdd(&dmdx_stamp[2], sa2); atomicAdd(&dmdx_stamp[3], sa3);
    atomicAdd(&dmdx_stamp[5], sa5);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1)::"memory");
    atomicAdd(&dmdx_stamp[6], st4 - ct0);   // core-clock cycles of this wave's chunk loop
    atomicAdd(&dmdx_stamp[7], rt1 - rt0);   // the same interval in 100 MHz reference ticks
    atomicAdd(&dmdx_stamp[4], (unsigned long long)nchunks);
    atomicAdd(&dmdx_stamp[8], ct0 - pt0);
    atomicAdd(&dmdx_stamp[10], 1ull);
  }
#endif
  DMDX_COMMIT_ALL();
#ifdef DMDX_STAMPS
  {
    unsigned long long et1;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(et1)::"memory");
    if (lane == 0) atomicAdd(&dmdx_stamp[9], et1 - st4);
  }
#endif
#undef DMDX_READ_FRAGS
#undef DMDX_MFMA4
#undef DMDX_KSTEP
#undef DMDX_DMA_NEXT
#undef DMDX_DMA_A
#undef DMDX_DMA_B
#undef DMDX_DMA_OP4
#undef DMDX_DMA_OP2
#undef DMDX_DMA_OP1
#undef DMDX_DMA_OP3
#undef DMDX_FOLD
#undef DMDX_COMMIT_ALL
#undef DMDX_COMMIT_H16
#undef DMDX_COMMIT
#undef DMDX_BLOCK_OFF
}

// XCD-aware unit order: groups of 512 consecutive blocks; the 64 blocks of a group that land on
// one XCD (blockIdx % 8) take 64 consecutive units (an 8 x 8 patch of tiles).  `b` counts from
// the start of a range whose first block index is a multiple of 8.
__device__ __forceinline__ int xcd_unit(int b, int total) {
  const int g = b >> 9;
  return ((g << 9) + 512 <= total) ? (g << 9) + (b & 7) * 64 + ((b & 511) >> 3) : b;
}

template <bool DMA, int ABL = 0, int SK = 0, int H16 = 0>
__global__ __launch_bounds__(NTH, 2) void gemm_tn_partial_kernel(TnParams p) {
  constexpr int TM = SK ? 32 * SK + 16 * H16 : BT;
  __shared__ __attribute__((aligned(16))) float lds[2 * (TM + BT) * BK];
  const int pos = xcd_unit(blockIdx.x, gridDim.x);
  const int split = pos / p.ntiles;
  const int tile = pos - split * p.ntiles;
  int ta, tb;
  decode_tile(p, tile, ta, tb);
  double* Pt = p.P + ((size_t)split * p.ntiles + tile) * (TM * BT);
  tn_unit<DMA, ABL, SK, H16>(p, split, ta * TM, tb * BT, Pt, lds);
}

// D (+)= sum_j A_j^T B_j over a batch of row blocks (TnBatch); p carries what the blocks share
// (shape of D, tiles, P).  SYRK: A_j == B_j, triangle tiles.
template <bool DMA, int SK = 0, int H16 = 0>
__global__ __launch_bounds__(NTH, 2) void syrk_batch_kernel(TnParams p, TnBatch bt) {
  constexpr int TM = SK ? 32 * SK + 16 * H16 : BT;
  __shared__ __attribute__((aligned(16))) float lds[2 * (TM + BT) * BK];
  const int b = blockIdx.x;
  int j = 0;
  while (j + 1 < bt.nblocks && b >= bt.unit_begin[j + 1]) ++j;
  const int total = bt.nsplit[j] * p.ntiles;
  const int local = b - bt.unit_begin[j];
  if (local >= total) return;  // padding up to the next multiple of 512
  const int pos = xcd_unit(local, total);
  const int split = pos / p.ntiles;
  const int tile = pos - split * p.ntiles;
  int ta, tb;
  decode_tile(p, tile, ta, tb);
  p.A = bt.A[j];
  p.B = bt.B[j];
  p.lda = bt.lda[j];
  p.ldb = bt.ldb[j];
  p.K = bt.K[j];
  p.chunks_total = bt.chunks_total[j];
  p.chunks_per_split = bt.chunks_per_split[j];
  double* Pt = p.P + ((size_t)(bt.slab_begin[j] + split) * p.ntiles + tile) * (TM * BT);
  // (measurement aid: two scalar clock reads around the unit, three atomics per workgroup; the
  // sustained core clock of the launch = 100 MHz * sum(cycles) / sum(ticks))
  unsigned long long c0 = 0, r0 = 0;
  if (bt.clk != nullptr) {
    c0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  tn_unit<DMA, 0, SK, H16>(p, split, ta * TM, tb * BT, Pt, lds);
  if (bt.clk != nullptr) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
      atomicAdd(&bt.clk[0], c1 - c0);
      atomicAdd(&bt.clk[1], r1 - r0);
      atomicAdd(&bt.clk[2], 1ull);
    }
  }
}

// Sum the K-splits in fp64 and scatter the tile into D (row-major view:
// D[i*ld + j], which is the column-major C of the C ABI with the operand
// roles swapped by the host wrapper).  SYRK mode also writes the mirror.
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(
    const double* P, int nsplit, int ntiles, int ntr, int ntc, int syrk, int nrow, int ncol,
    double* D64, int64_t ld64, float* D32, int64_t ld32, int accumulate, int tm) {
  __shared__ double tr[32][33];
  // one workgroup per (tile, 32x32 sub-block); a tile is tm (32 ... 128, a multiple of 16) rows x 128 columns
  const int nsb = ((tm + 31) / 32) * 4;
  const int tile = blockIdx.x / nsb;
  const int sb = blockIdx.x - tile * nsb;
  const int tile_elems = tm * BT;
  int ta, tb;
  if (syrk) {
    decode_tri(tile, ntr, ta, tb);
  } else {
    int per_sr = SR * ntc;
    int sr = tile / per_sr;
    int r0 = sr * SR;
    int nrows = ntr - r0 < SR ? ntr - r0 : SR;
    int tt = tile - sr * per_sr;
    tb = tt / nrows;
    ta = r0 + tt % nrows;
  }
  const int row0 = ta * tm, col0 = tb * BT;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const bool diag = syrk && (ta == tb);
  const int si = (sb >> 2) * 32, sj = (sb & 3) * 32;
  if (diag && si > sj) return;  // lower sub-blocks of a diagonal tile: mirrored from the upper ones
  if (row0 + si >= nrow || col0 + sj >= ncol) return;
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  const double* src = P + (size_t)tile * tile_elems + (si + ty) * BT + sj + tx;
  for (int sp = 0; sp < nsplit; ++sp) {
    const double* q = src + (size_t)sp * ntiles * tile_elems;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (si + ty + 8 * k < tm) v[k] += q[8 * k * BT];   // (a 48- / 80- / 112-row tile ends inside its last sub-block row)
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = si + ty + 8 * k, j = sj + tx;
    const int gi = row0 + i, gj = col0 + j;
    const bool keep = !(diag && i > j) && i < tm;
    if (keep && gi < nrow && gj < ncol) {
      double o = v[k];
      if (accumulate) o += D64[(int64_t)gi * ld64 + gj];
      D64[(int64_t)gi * ld64 + gj] = o;
      if (D32) D32[(int64_t)gi * ld32 + gj] = (float)o;
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
      const bool keep = !(diag && i >= j) && i < tm;
      if (keep && gi < nrow && gj < ncol) {
        double o = s;
        if (accumulate) o += D64[(int64_t)gj * ld64 + gi];
        D64[(int64_t)gj * ld64 + gi] = o;
        if (D32) D32[(int64_t)gj * ld32 + gi] = (float)o;
      }
    }
  }
}

struct Plan {
  int tm;  // tile rows: 128, or 96 / 64 when that cuts the row padding of a non-SYRK product
  int syrk;  // triangle of tiles + mirror; a Gram of <= 96 columns runs as one 64- / 96-row tile of the
             // plain product X^T X instead (both triangles computed: half / three quarters of the MFMA
             // work of the 128 x 128 tile, no mirror pass)
  int ntr, ntc, ntiles, nsplit, chunks_total, chunks_per_split;
  size_t ws_bytes;
};

// share: number of equally sized row blocks that share the launch (batched SYRK)
Plan make_plan(int64_t K, int64_t nrow, int64_t ncol, int syrk, int share = 1) {
  Plan pl;
  pl.tm = BT;
  const int gram = syrk;
  if (syrk && nrow <= 96) syrk = 0;
  pl.syrk = syrk;
  if (!syrk) {  // fewest padded rows; ties go to the taller tile (fewer re-reads of the B panels)
    int64_t best = (nrow + BT - 1) / BT * BT;
    for (int tm : {112, 96, 80, 64, 48, 32}) {
      // (a Gram keeps to the 32-row blocks: the 16-row block contracts k in another order, and
      // G[i][j] would differ from G[j][i] in the last bits)
      if (gram && tm % 32) continue;
      const int64_t padded = (nrow + tm - 1) / tm * tm;
      if (padded < best && (tm > 48 || nrow <= tm)) { best = padded; pl.tm = tm; }   // 32- / 48-row tiles: single tile row only
    }
  }
  if (const char* e = getenv("DMDX_TN_FORCE_TM")) {   // A/B knob (scripts/ab_k3.py): a tile height for the plain products
    const int f = atoi(e);
    if (!syrk && f >= 32 && f <= 128 && f % 16 == 0 && (f > 48 || nrow <= f)) pl.tm = f;
  }
  pl.ntr = (int)((nrow + pl.tm - 1) / pl.tm);
  pl.ntc = (int)((ncol + BT - 1) / BT);
  pl.ntiles = syrk ? pl.ntr * (pl.ntr + 1) / 2 : pl.ntr * pl.ntc;
  pl.chunks_total = (int)((K + BK - 1) / BK);
  if (pl.chunks_total < 1) pl.chunks_total = 1;
  // aim at >= ~20 rounds of 512 resident workgroups, keep >= 8 chunks per split, and
  // keep units short enough (<= max_cps chunks) that the workgroups sharing panels in
  // L2 / Infinity Cache do not drift apart along K
  // (round 2, cfg2 Gram incl. the reduce kernel, interleaved on one box: 256 -> 564-567 ms, 512 / 1024 /
  // 2048 -> 560-562 ms, 4096 -> 575-577 ms (38 rounds of 512: the last one shows).  2048 halves the
  // partial-tile workspace of 1024: 5 GB instead of 10 GB at cfg2, 9.5 instead of 19 GB at a cfg3 shard)
  int64_t max_cps = 2048;
  if (const char* e = getenv("DMDX_TN_MAX_CPS")) max_cps = atoll(e) > 0 ? atoll(e) : max_cps;
  // Rounds of 512 resident workgroups to aim at.  Every split costs a partial tile (written,
  // then read by the reduce kernel): many rounds only pay where the tiles are many and the
  // operands are re-read from L2 anyway (big SYRK); few-tile products and the HBM-streaming
  // X^T Y / Q^T X products (every X panel is read once) want few, long units
  // (cfg2 randomized: K3 97.6 ms at 20 rounds, 91.5 at 4..8; l x l Grams 12.5 -> 7.6 ms at 2).
  int64_t rounds = gram ? (pl.ntiles >= 64 ? 20 : 2) : 6;
  if (const char* e = getenv("DMDX_TN_ROUNDS")) rounds = atoll(e) > 0 ? atoll(e) : rounds;
  const int64_t per_split = (int64_t)pl.ntiles * share;
  int64_t want = (rounds * 512 + per_split - 1) / per_split;
  int64_t by_len = (pl.chunks_total + max_cps - 1) / max_cps;
  if (want < by_len) want = by_len;
  int64_t maxs = pl.chunks_total / 8;
  if (maxs < 1) maxs = 1;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 256) want = 256;
  // The tail of the launch: its units are equally long and 512 of them are resident, so a launch of
  // r = units / 512 rounds takes ceil(r) of them -- 6.34 rounds (X^T Y of 16 cfg4 blocks, l = 70: 29 tiles x 7
  // splits x 16) cost 7.  With few rounds, take the split count (up to twice the wanted one) that fills the
  // last round best; the first one that fills it to 97 % wins.  (The Gram of cfg2 runs 75 rounds: unchanged.)
  // Only when the wanted count leaves the last round less than 92 % full: measured on 16 cfg4 blocks
  // (n = 3653) l = 70 11.58 -> 11.09 ms, l = 100 15.14 -> 14.24, l = 220 30.07 -> 27.95; on 8 cfg2 blocks
  // (n = 8760, last round 92.4 % full as it is) twice the splits cost 2 % (l = 60 10.60 -> 10.82 ms).
  if (getenv("DMDX_TN_NO_TAILFIT") == nullptr) {
    auto fill = [&](int64_t w) {
      const int64_t cps = (pl.chunks_total + w - 1) / w;
      const int64_t ns = (pl.chunks_total + cps - 1) / cps;
      const double r = (double)(per_split * ns) / 512.0;
      return r / (double)(int64_t)(r + 0.999999);
    };
    int64_t best = want;
    double bf = fill(want);
    if (bf < 0.92)
      for (int64_t w = want + 1; bf < 0.97 && w <= maxs && w <= 256 && w <= 2 * want + 2; ++w)
        if (fill(w) > bf) { bf = fill(w); best = w; }
    want = best;
  }
  pl.chunks_per_split = (int)((pl.chunks_total + want - 1) / want);
  pl.nsplit = (pl.chunks_total + pl.chunks_per_split - 1) / pl.chunks_per_split;
  pl.ws_bytes = (size_t)pl.nsplit * pl.ntiles * pl.tm * BT * sizeof(double);
  return pl;
}

int run_tn(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t K, int64_t nrow,
           int64_t ncol, int syrk, double* D64, int64_t ld64, float* D32, int64_t ld32,
           int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
  Plan pl = make_plan(K, nrow, ncol, syrk);
  if (ws == nullptr || ws_bytes < pl.ws_bytes) {
    dmdx_set_error("gemm_tn: workspace %zu bytes < required %zu", ws_bytes, pl.ws_bytes);
    return DMDX_E_WORKSPACE;
  }
  TnParams p;
  p.A = A; p.B = B; p.lda = lda; p.ldb = ldb; p.K = K;
  p.nrow = (int)nrow; p.ncol = (int)ncol;
  p.ntr = pl.ntr; p.ntc = pl.ntc; p.ntiles = pl.ntiles; p.syrk = pl.syrk;
  p.nsplit = pl.nsplit; p.chunks_total = pl.chunks_total;
  p.chunks_per_split = pl.chunks_per_split;
  p.P = reinterpret_cast<double*>(ws);
  // per-lane offsets inside a 128-column panel are 32-bit: elements on the register path,
  // bytes (+ 4 KiB of slack) on the LDS-DMA path
  if (lda >= (int64_t(1) << 25) || ldb >= (int64_t(1) << 25)) {
    dmdx_set_error("gemm_tn: leading dimension %lld / %lld >= 2^25 not supported", (long long)lda, (long long)ldb);
    return DMDX_E_UNSUPPORTED;
  }
  const bool aligned = (lda % 4 == 0) && (ldb % 4 == 0) && dmdx_aligned16(A) && dmdx_aligned16(B) &&
                       lda < (int64_t(1) << 22) && ldb < (int64_t(1) << 22);
  dim3 grid((unsigned)((size_t)pl.nsplit * pl.ntiles));
  int abl = 0;
  if (const char* e = getenv("DMDX_TN_ABLATE")) abl = atoi(e);
  if (pl.tm != BT) abl = 0;
  if (aligned && abl == 1)
    hipLaunchKernelGGL((gemm_tn_partial_kernel<true, 1>), grid, dim3(NTH), 0, stream, p);
  else if (aligned && abl == 2)
    hipLaunchKernelGGL((gemm_tn_partial_kernel<true, 2>), grid, dim3(NTH), 0, stream, p);
  else if (aligned && abl == 8)
    hipLaunchKernelGGL((gemm_tn_partial_kernel<true, 8>), grid, dim3(NTH), 0, stream, p);
  else if (aligned && abl == 11)
    hipLaunchKernelGGL((gemm_tn_partial_kernel<true, 11>), grid, dim3(NTH), 0, stream, p);
  else if (pl.tm != BT) {  // skinny tiles (LDS-DMA staging needs 16-byte aligned column starts)
#define DMDX_TN_CASE(TMV, SKV, HV)                                                                          \
    case TMV:                                                                                               \
      if (aligned) hipLaunchKernelGGL((gemm_tn_partial_kernel<true, 0, SKV, HV>), grid, dim3(NTH), 0, stream, p);  \
      else hipLaunchKernelGGL((gemm_tn_partial_kernel<false, 0, SKV, HV>), grid, dim3(NTH), 0, stream, p);  \
      break
    switch (pl.tm) {
      DMDX_TN_CASE(32, 1, 0); DMDX_TN_CASE(48, 1, 1); DMDX_TN_CASE(64, 2, 0); DMDX_TN_CASE(80, 2, 1);
      DMDX_TN_CASE(96, 3, 0); DMDX_TN_CASE(112, 3, 1);
      default: dmdx_set_error("gemm_tn: no kernel for tile height %d", pl.tm); return DMDX_E_INVALID;
    }
#undef DMDX_TN_CASE
  } else if (aligned)
    hipLaunchKernelGGL(gemm_tn_partial_kernel<true>, grid, dim3(NTH), 0, stream, p);
  else
    hipLaunchKernelGGL(gemm_tn_partial_kernel<false>, grid, dim3(NTH), 0, stream, p);
  DMDX_LAUNCH_CHECK();
  hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(pl.ntiles * ((pl.tm + 31) / 32) * 4), dim3(256), 0, stream, p.P,
                     pl.nsplit, pl.ntiles, pl.ntr, pl.ntc, pl.syrk, (int)nrow, (int)ncol, D64, ld64, D32, ld32,
                     accumulate, pl.tm);
  DMDX_LAUNCH_CHECK();
  return 0;
}

// ---- batched products: D (+)= sum_j A_j^T B_j, blocks in groups of MAXB per launch
size_t batch_group_ws(const int64_t* K, int nb, int64_t nrow, int64_t ncol, int syrk) {
  size_t slabs = 0;
  Plan pl{};
  for (int j = 0; j < nb; ++j) {
    pl = make_plan(K[j], nrow, ncol, syrk, nb);
    slabs += (size_t)pl.nsplit;
  }
  return slabs * (size_t)pl.ntiles * pl.tm * BT * sizeof(double);
}

int run_batch(const float* const* A, const int64_t* lda, const float* const* B, const int64_t* ldb,
              const int64_t* K, int nblocks, int64_t nrow, int64_t ncol, int syrk, double* D64, int64_t ld64,
              float* D32, int64_t ld32, int accumulate, void* ws, size_t ws_bytes, hipStream_t stream) {
  for (int j0 = 0; j0 < nblocks; j0 += MAXB) {
    const int nb = nblocks - j0 < MAXB ? nblocks - j0 : MAXB;
    const size_t need = batch_group_ws(K + j0, nb, nrow, ncol, syrk);
    if (ws == nullptr || ws_bytes < need) {
      dmdx_set_error("blocks: workspace %zu bytes < required %zu", ws_bytes, need);
      return DMDX_E_WORKSPACE;
    }
    TnBatch bt{};
    TnParams p{};
    bool aligned = true;
    int units = 0, slabs = 0;
    Plan pl{};
    for (int j = 0; j < nb; ++j) {
      const int64_t la = lda[j0 + j], lb = ldb[j0 + j];
      if (la >= (int64_t(1) << 24) || lb >= (int64_t(1) << 24)) {
        dmdx_set_error("blocks: leading dimension >= 2^24 not supported (use smaller row blocks)");
        return DMDX_E_UNSUPPORTED;
      }
      pl = make_plan(K[j0 + j], nrow, ncol, syrk, nb);
      bt.A[j] = A[j0 + j];
      bt.B[j] = B[j0 + j];
      bt.lda[j] = la;
      bt.ldb[j] = lb;
      bt.K[j] = K[j0 + j];
      bt.chunks_total[j] = pl.chunks_total;
      bt.chunks_per_split[j] = pl.chunks_per_split;
      bt.nsplit[j] = pl.nsplit;
      bt.unit_begin[j] = units;
      bt.slab_begin[j] = slabs;
      const int u = pl.nsplit * pl.ntiles;
      units += (j + 1 < nb) ? (u + 511) / 512 * 512 : u;
      slabs += pl.nsplit;
      aligned = aligned && (la % 4 == 0) && (lb % 4 == 0) && dmdx_aligned16(A[j0 + j]) &&
                dmdx_aligned16(B[j0 + j]) && la < (int64_t(1) << 22) && lb < (int64_t(1) << 22);
    }
    bt.unit_begin[nb] = units;
    bt.slab_begin[nb] = slabs;
    bt.nblocks = nb;
    bt.clk = dmdx_clock_probe_ptr;
    p.nrow = (int)nrow;
    p.ncol = (int)ncol;
    p.ntr = pl.ntr; p.ntc = pl.ntc; p.ntiles = pl.ntiles; p.syrk = pl.syrk;
    p.nsplit = slabs;
    p.P = reinterpret_cast<double*>(ws);
    const dim3 grid((unsigned)units);
    switch (pl.tm) {
#define DMDX_TN_CASE(TMV, SKV, HV)                                                                          \
      case TMV:                                                                                             \
        if (aligned) hipLaunchKernelGGL((syrk_batch_kernel<true, SKV, HV>), grid, dim3(NTH), 0, stream, p, bt);    \
        else hipLaunchKernelGGL((syrk_batch_kernel<false, SKV, HV>), grid, dim3(NTH), 0, stream, p, bt);    \
        break
      DMDX_TN_CASE(32, 1, 0); DMDX_TN_CASE(48, 1, 1); DMDX_TN_CASE(64, 2, 0); DMDX_TN_CASE(80, 2, 1);
      DMDX_TN_CASE(96, 3, 0); DMDX_TN_CASE(112, 3, 1); DMDX_TN_CASE(128, 0, 0);
#undef DMDX_TN_CASE
      default: dmdx_set_error("gemm_tn_blocks: no kernel for tile height %d", pl.tm); return DMDX_E_INVALID;
    }
    DMDX_LAUNCH_CHECK();
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(pl.ntiles * ((pl.tm + 31) / 32) * 4), dim3(256), 0, stream, p.P, slabs,
                       pl.ntiles, pl.ntr, pl.ntc, pl.syrk, (int)nrow, (int)ncol, D64, ld64, D32, ld32,
                       (accumulate || j0 > 0) ? 1 : 0, pl.tm);
    DMDX_LAUNCH_CHECK();
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// K3s: D (nb x na) (+)= sum_j Y_j^T X_j for nb <= 32 (the HBM-bound small-l products of the range
// finder at the reference's default n_components = 10, l = 20; the single-column product
// w = A^T mu of the mean deflation).  The generic body above stages 32-row chunks: every LDS-DMA
// piece is 8 columns x 128 B, i.e. every 128-byte request opens another DRAM page of a matrix
// whose columns are 0.5 MB apart, and the pass over cfg2's X runs at 4.85 TB/s where K2 (1 KB
// runs) streams the same bytes at 6.0 TB/s.  Here a chunk is 64 rows -- pieces of 4 columns x
// 256 B -- and the four waves of a workgroup split the ROWS of a chunk (16 each) instead of the
// columns of the tile, so that one workgroup still covers 128 time columns with 40 KB per stage:
//   tile  = 32 (l) x 128 (time columns), unit = (row block, K split, tile), 2 workgroups per CU;
//   stage = Y panel [32 col][64 k] + X panel [128 col][64 k] fp32, 16-byte k-pieces XOR-swizzled
//           with (col & 15) on the DMA source and on the fragment reads (conflict-free b128);
//   wave w contracts rows 16 w .. 16 w + 15 of every chunk: 2 k-steps x (1 + 4) ds_read_b128 and
//           32 MFMAs per chunk; fp32 chains of <= 2048 rows folded into a second accumulator;
//   end of unit: the four waves' 32 x 128 accumulators meet in LDS, one fp64 partial tile per
//           unit (no atomics), a second kernel sums the units of a tile into D.
// Rows beyond the last full 64-row chunk of a block (K % 64) go through the generic path
// (one extra batched launch over the tails, accumulate).
constexpr int XT = 128, XL = 32, XK = 64;
constexpr int XSTG = (XL + XT) * XK;   // floats per stage: 40 KB
constexpr int XFOLD = 128;             // chunks per fp32 chain (16 rows of each per wave: 2048 rows)

struct XtyBatch {
  const float* X[MAXB];   // big operand (D cols): K x na, ldx
  const float* Y[MAXB];   // small operand (D rows): K x nb, ldy
  int64_t ldx[MAXB], ldy[MAXB];
  int chunks[MAXB], cps[MAXB], nsplit[MAXB];
  int unit_begin[MAXB + 1];
  int nblocks, ntiles, na, nb;
  double* P;              // [unit][32][128] fp64
};

__device__ __forceinline__ void xty_dma(unsigned lds_addr, unsigned off, const char* base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :: "s"(lds_addr), "v"(off), "s"(base) : "memory");
}

// T4 < 0: the body described above (32 rows of D on v_mfma_f32_32x32x2_f32).
// T4 = 0 / 1 / 2 (nb <= 16 / 20 / 24, round 3): 16 rows of D on v_mfma_f32_16x16x4_f32 -- lane (i, kk) reads the piece
//   4 wave + kk of its column: MFMA e contracts the wave's rows 16 wave + 4 kk + e -- and T4 groups of 4 more rows on
//   v_mfma_f32_4x4x1_16b_f32 with the sixteen blocks of an instruction on sixteen groups of 4 time columns: A = X[row]
//   [64 h + lane], B = Y[row][16 + 4 q + (lane & 3)], one row per instruction, 4 accumulator registers per 64 time
//   columns and no cross-lane sum.  The reference's default rank (l = 20) executes 20 columns of MFMA work per
//   X element instead of 32: PMC had the 32-column body at 79 % matrix-core occupancy under a 5.5 TB/s stream.
template <int T4>
__global__ __launch_bounds__(NTH, 2) void xty_small_kernel(XtyBatch bt) {
  __shared__ __attribute__((aligned(16))) float lds[2 * XSTG];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int u = blockIdx.x;
  int j = 0;
  while (j + 1 < bt.nblocks && u >= bt.unit_begin[j + 1]) ++j;
  const int local = u - bt.unit_begin[j];
  const int split = local / bt.ntiles, tile = local - split * bt.ntiles;
  const int c_begin = split * bt.cps[j];
  int c_end = c_begin + bt.cps[j];
  if (c_end > bt.chunks[j]) c_end = bt.chunks[j];
  const int nch = c_end - c_begin;
  const int col0 = tile * XT;
  const int64_t ldx = bt.ldx[j], ldy = bt.ldy[j];
  const int cb0 = col0 < bt.na ? col0 : bt.na - 1;
  const char* Xb = reinterpret_cast<const char*>(bt.X[j] + (int64_t)cb0 * ldx) + (int64_t)c_begin * XK * 4;
  const char* Yb = reinterpret_cast<const char*>(bt.Y[j]) + (int64_t)c_begin * XK * 4;

  // per-lane source byte offsets of this wave's pieces (columns past the edge clamp onto the last one)
  unsigned xoff[8], yoff[2];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int lc = 32 * wave + 4 * i + (lane >> 4);
    int cg = col0 + lc;
    cg = cg < bt.na ? cg : bt.na - 1;
    const int g = (lane & 15) ^ (lc & 15);
    xoff[i] = (unsigned)(((int64_t)(cg - cb0) * ldx + 4 * g) * 4);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int lc = 8 * wave + 4 * i + (lane >> 4);
    const int cg = lc < bt.nb ? lc : bt.nb - 1;
    const int g = (lane & 15) ^ (lc & 15);
    yoff[i] = (unsigned)(((int64_t)cg * ldy + 4 * g) * 4);
  }
  auto issue = [&](int c, int st) {
    const char* xb = Xb + (int64_t)c * (XK * 4);
    const char* yb = Yb + (int64_t)c * (XK * 4);
    float* ys = lds + st * XSTG + (8 * wave) * XK;
    float* xs = lds + st * XSTG + XL * XK + (32 * wave) * XK;
#pragma unroll
    for (int i = 0; i < 2; ++i) xty_dma((unsigned)(uintptr_t)DMDX_LDS_PTR(ys + 4 * i * XK), yoff[i], yb);
#pragma unroll
    for (int i = 0; i < 8; ++i) xty_dma((unsigned)(uintptr_t)DMDX_LDS_PTR(xs + 4 * i * XK), xoff[i], xb);
  };

  if constexpr (T4 >= 0) {
    constexpr int NT = T4 > 0 ? T4 : 1;
    const int c15 = lane & 15, kk = lane >> 4;
    f32x4 a16[8], a16b[8], at[2][NT], atb[2][NT];
#pragma unroll
    for (int b = 0; b < 8; ++b) { a16[b] = f32x4{0.f, 0.f, 0.f, 0.f}; a16b[b] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < NT; ++q) { at[h][q] = f32x4{0.f, 0.f, 0.f, 0.f}; atb[h][q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if (nch > 0) issue(0, 0);
    if (nch > 1) issue(1, 1);
    const int ko16 = 4 * ((4 * wave + kk) ^ c15);          // the 16-column operands: piece 4 wave + kk of column c15 (+ 16 b)
    for (int c = 0; c < nch; ++c) {
      const int st = c & 1;
      if (c + 1 < nch) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const float* ys = lds + st * XSTG;
      const float* xs = ys + XL * XK;
      const f32x4 fa = *reinterpret_cast<const f32x4*>(ys + c15 * XK + ko16);
      f32x4 fb[8];
#pragma unroll
      for (int b = 0; b < 8; ++b) fb[b] = *reinterpret_cast<const f32x4*>(xs + (16 * b + c15) * XK + ko16);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int b = 0; b < 8; ++b) a16[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[e], fb[b][e], a16[b], 0, 0, 0);
      if constexpr (T4 > 0) {
#pragma unroll
        for (int p4 = 0; p4 < 4; ++p4) {                     // rows 16 wave + 4 p4 + e
          f32x4 ax[2], by[NT];
#pragma unroll
          for (int h = 0; h < 2; ++h)
            ax[h] = *reinterpret_cast<const f32x4*>(xs + (64 * h + lane) * XK + 4 * ((4 * wave + p4) ^ (lane & 15)));
#pragma unroll
          for (int q = 0; q < NT; ++q) {
            const int yc = 16 + 4 * q + (lane & 3);
            by[q] = *reinterpret_cast<const f32x4*>(ys + yc * XK + 4 * ((4 * wave + p4) ^ (yc & 15)));
          }
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int q = 0; q < NT; ++q) at[h][q] = __builtin_amdgcn_mfma_f32_4x4x1f32(ax[h][e], by[q][e], at[h][q], 0, 0, 0);
        }
      }
      __syncthreads();   // every wave has read stage st for the last time
      if (c + 2 < nch) issue(c + 2, st);
      if ((c & (XFOLD - 1)) == XFOLD - 1) {
#pragma unroll
        for (int b = 0; b < 8; ++b) { a16b[b] += a16[b]; a16[b] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int q = 0; q < NT; ++q) { atb[h][q] += at[h][q]; at[h][q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      }
    }
    // ---- this wave's 32 x 128 image: rows 0..15 from the 16 x 16 results (lane (j, q), register r: row 4 q + r,
    // column 16 b + j), rows 16 + 4 q + (lane & 3) from the 4 x 4 blocks (register i: column 64 h + 4 (lane >> 2) + i),
    // the rest zero
    float* red = lds + wave * (XL * XT);
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(4 * kk + r) * XT + 16 * b + c15] = a16[b][r] + a16b[b][r];
    for (int o = 4 * lane; o < 16 * XT; o += 256)            // rows 16 .. 31
      *reinterpret_cast<f32x4*>(red + 16 * XT + o) = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (T4 > 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the wave's own zeros before its own values: same addresses)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < NT; ++q) {
          const f32x4 v = at[h][q] + atb[h][q];
          *reinterpret_cast<f32x4*>(red + (16 + 4 * q + (lane & 3)) * XT + 64 * h + 4 * (lane >> 2)) = v;
        }
    }
    __syncthreads();
    double* Pt = bt.P + (size_t)u * (XL * XT);
    const float* r0 = lds;
    for (int o = tid; o < XL * XT; o += NTH) {
      const double v = (double)r0[o] + (double)r0[XL * XT + o] + (double)r0[2 * XL * XT + o] + (double)r0[3 * XL * XT + o];
      __builtin_nontemporal_store(v, Pt + o);
    }
    return;
  }
  f32x16 acc[4], acc2[4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[b][r] = 0.f; acc2[b][r] = 0.f; }

  if (nch > 0) issue(0, 0);
  if (nch > 1) issue(1, 1);
  const int sw = l31 & 15;
  for (int c = 0; c < nch; ++c) {
    const int st = c & 1;
    if (c + 1 < nch) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");   // this wave's 10 pieces of chunk c have landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float* ys = lds + st * XSTG;
    const float* xs = ys + XL * XK;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int c16 = 4 * wave + 2 * s2 + lh;
      const int ko = 4 * (c16 ^ sw);
      const f32x4 fa = *reinterpret_cast<const f32x4*>(ys + l31 * XK + ko);
      f32x4 fb[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) fb[b] = *reinterpret_cast<const f32x4*>(xs + (32 * b + l31) * XK + ko);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[b][e], acc[b], 0, 0, 0);
    }
    __syncthreads();   // every wave has read stage st for the last time
    if (c + 2 < nch) issue(c + 2, st);
    if ((c & (XFOLD - 1)) == XFOLD - 1) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        acc2[b] += acc[b];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
      }
    }
  }
  // ---- the four waves' partial tiles meet in LDS (64 KB of the 80), wave w finishes column block w
  float* red = lds;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int q = 0; q < 16; ++q)
      red[wave * (XL * XT) + ((q & 3) + 8 * (q >> 2) + 4 * lh) * XT + 32 * b + l31] = acc[b][q] + acc2[b][q];
  __syncthreads();
  double* Pt = bt.P + (size_t)u * (XL * XT);
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = (q & 3) + 8 * (q >> 2) + 4 * lh;
    const int o = i * XT + 32 * wave + l31;
    const double v = (double)red[o] + (double)red[XL * XT + o] + (double)red[2 * XL * XT + o] + (double)red[3 * XL * XT + o];
    __builtin_nontemporal_store(v, Pt + o);
  }
}

// D[i][col] (+)= sum over the units of tile (col / 128) of their partial tiles
__global__ __launch_bounds__(128) void xty_small_reduce_kernel(XtyBatch bt, double* D64, int64_t ld64, int accumulate) {
  const int tile = blockIdx.x, i = blockIdx.y, jj = threadIdx.x;
  const int col = tile * XT + jj;
  if (i >= bt.nb || col >= bt.na) return;
  double s = 0.0;
  for (int j = 0; j < bt.nblocks; ++j)
    for (int sp = 0; sp < bt.nsplit[j]; ++sp)
      s += bt.P[(size_t)(bt.unit_begin[j] + sp * bt.ntiles + tile) * (XL * XT) + i * XT + jj];
  double* d = D64 + (int64_t)i * ld64 + col;
  *d = accumulate ? *d + s : s;
}

bool xty_small_ok(const float* const* X, const int64_t* ldx, const float* const* Y, const int64_t* ldy,
                  const int64_t* K, int nblocks, int64_t nb, int64_t na, const float* D32) {
  if (getenv("DMDX_NO_K3S")) return false;   // A/B knob
  if (D32 || nb > XL || na < XT) return false;
  for (int j = 0; j < nblocks; ++j) {
    if (K[j] < 8 * XK || ldx[j] % 4 || ldy[j] % 4 || !dmdx_aligned16(X[j]) || !dmdx_aligned16(Y[j])) return false;
    if (ldx[j] >= (int64_t(1) << 22) || ldy[j] >= (int64_t(1) << 22)) return false;
  }
  return true;
}

void xty_small_plan(const int64_t* K, int nb_, int ntiles, int* chunks, int* cps, int* nsplit, int* units) {
  int total = 0;
  for (int j = 0; j < nb_; ++j) {
    chunks[j] = (int)(K[j] / XK);
    int64_t want = (4 * 512 + (int64_t)ntiles * nb_ - 1) / ((int64_t)ntiles * nb_);   // ~4 rounds of 512 workgroups
    const int64_t maxs = chunks[j] / 16 > 0 ? chunks[j] / 16 : 1;
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    // (the tail of the launch, as in make_plan: cfg2 at l = 20 is 69 tiles x 8 blocks x 4 splits = 4.31
    // rounds of 512 that cost 5; 12 splits are 12.94)
    if (getenv("DMDX_TN_NO_TAILFIT") == nullptr) {
      auto fill = [&](int64_t w) {
        const int64_t cps_ = (chunks[j] + w - 1) / w;
        const int64_t ns = cps_ > 0 ? (chunks[j] + cps_ - 1) / cps_ : 0;
        const double r = (double)((int64_t)ntiles * nb_ * ns) / 512.0;
        return r > 0.0 ? r / (double)(int64_t)(r + 0.999999) : 1.0;
      };
      int64_t best = want;
      double bf = fill(want);
      if (bf < 0.92)
        for (int64_t w = want + 1; bf < 0.97 && w <= maxs && w <= 4 * want + 2; ++w)
          if (fill(w) > bf) { bf = fill(w); best = w; }
      want = best;
    }
    cps[j] = (int)((chunks[j] + want - 1) / want);
    // a block of fewer than 64 rows has no full chunk: no units here, all of it goes to the tail launch
    nsplit[j] = cps[j] > 0 ? (chunks[j] + cps[j] - 1) / cps[j] : 0;
    total += nsplit[j] * ntiles;
  }
  *units = total;
}

size_t xty_small_ws(const int64_t* K, int nblocks, int64_t na) {
  const int ntiles = (int)((na + XT - 1) / XT);
  size_t need = 0;
  for (int j0 = 0; j0 < nblocks; j0 += MAXB) {
    const int ng = nblocks - j0 < MAXB ? nblocks - j0 : MAXB;
    int ch[MAXB], cp[MAXB], ns[MAXB], units = 0;
    xty_small_plan(K + j0, ng, ntiles, ch, cp, ns, &units);
    const size_t g = (size_t)units * XL * XT * sizeof(double);
    if (g > need) need = g;
  }
  return need;
}

int run_xty_small(const float* const* X, const int64_t* ldx, const float* const* Y, const int64_t* ldy,
                  const int64_t* K, int nblocks, int64_t nb, int64_t na, double* D64, int64_t ld64, int accumulate,
                  void* ws, size_t ws_bytes, hipStream_t stream) {
  const int ntiles = (int)((na + XT - 1) / XT);
  if (ws == nullptr || ws_bytes < xty_small_ws(K, nblocks, na)) {
    dmdx_set_error("gemm_tn_blocks (small-l path): workspace %zu bytes too small", ws_bytes);
    return DMDX_E_WORKSPACE;
  }
  for (int j0 = 0; j0 < nblocks; j0 += MAXB) {
    const int ng = nblocks - j0 < MAXB ? nblocks - j0 : MAXB;
    XtyBatch bt{};
    int units = 0;
    xty_small_plan(K + j0, ng, ntiles, bt.chunks, bt.cps, bt.nsplit, &units);
    int ub = 0;
    for (int j = 0; j < ng; ++j) {
      bt.X[j] = X[j0 + j];
      bt.Y[j] = Y[j0 + j];
      bt.ldx[j] = ldx[j0 + j];
      bt.ldy[j] = ldy[j0 + j];
      bt.unit_begin[j] = ub;
      ub += bt.nsplit[j] * ntiles;
    }
    bt.unit_begin[ng] = ub;
    bt.nblocks = ng;
    bt.ntiles = ntiles;
    bt.na = (int)na;
    bt.nb = (int)nb;
    bt.P = reinterpret_cast<double*>(ws);
    if (units > 0) {   // (a group of blocks that are all shorter than one chunk: the reduce kernel still defines D)
      // (16 + 4 T4 rows of MFMA work for nb <= 16 / 20 / 24; DMDX_K3S_32=1: the 32-row body for everything, A/B)
      const bool wide = nb > 24 || getenv("DMDX_K3S_32") != nullptr;
      if (wide) hipLaunchKernelGGL(xty_small_kernel<-1>, dim3((unsigned)units), dim3(NTH), 0, stream, bt);
      else if (nb <= 16) hipLaunchKernelGGL(xty_small_kernel<0>, dim3((unsigned)units), dim3(NTH), 0, stream, bt);
      else if (nb <= 20) hipLaunchKernelGGL(xty_small_kernel<1>, dim3((unsigned)units), dim3(NTH), 0, stream, bt);
      else hipLaunchKernelGGL(xty_small_kernel<2>, dim3((unsigned)units), dim3(NTH), 0, stream, bt);
      DMDX_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(xty_small_reduce_kernel, dim3((unsigned)ntiles, (unsigned)nb), dim3(XT), 0, stream, bt, D64, ld64,
                       (accumulate || j0 > 0) ? 1 : 0);
    DMDX_LAUNCH_CHECK();
  }
  // rows past the last full 64-row chunk of every block: the generic path, accumulated on top
  std::vector<const float*> Xt, Yt;
  std::vector<int64_t> lx, ly, Kt;
  for (int j = 0; j < nblocks; ++j) {
    const int64_t tail = K[j] % XK;
    if (tail == 0) continue;
    Xt.push_back(X[j] + (K[j] - tail));
    Yt.push_back(Y[j] + (K[j] - tail));
    lx.push_back(ldx[j]);
    ly.push_back(ldy[j]);
    Kt.push_back(tail);
  }
  if (!Kt.empty())
    return run_batch(Yt.data(), ly.data(), Xt.data(), lx.data(), Kt.data(), (int)Kt.size(), nb, na, 0, D64, ld64, nullptr,
                     0, 1, ws, ws_bytes, stream);
  return 0;
}

}  // namespace

extern "C" {

#ifdef DMDX_STAMPS
int dmdx_debug_read_stamps(unsigned long long* out8, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(dmdx_stamp), 12 * sizeof(unsigned long long));   // (12 entries)
  if (reset) {
    unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(dmdx_stamp), z, sizeof(z));
  }
  return (int)e;
}
#endif

int dmdx_set_clock_probe(unsigned long long* dev_counters3) {
  dmdx_clock_probe_ptr = dev_counters3;
  return 0;
}

size_t dmdx_syrk_workspace_bytes(int64_t m, int64_t n) {
  if (m < 0 || n <= 0) return 0;
  return make_plan(m, n, n, 1).ws_bytes;
}

int dmdx_syrk_f32(const float* X, int64_t m, int64_t n, int64_t ldx, double* G64, int64_t ldg,
                  float* G32, int64_t ldg32, int accumulate, void* workspace,
                  size_t workspace_bytes, void* stream) {
  DMDX_CHECK_ARG(X && G64, "syrk: null pointer");
  DMDX_CHECK_ARG(m >= 1 && n >= 1 && n < (1 << 30), "syrk: bad shape m=%lld n=%lld", (long long)m,
                 (long long)n);
  DMDX_CHECK_ARG(ldx >= 1 && ldg >= n && (!G32 || ldg32 >= n), "syrk: bad leading dimension");
  DMDX_CHECK_ARG(ldx < (1ll << 24), "syrk: ldx >= 2^24 not supported (use row blocks)");
  return run_tn(X, ldx, X, ldx, m, n, n, 1, G64, ldg, G32, ldg32, accumulate, workspace, workspace_bytes,
                (hipStream_t)stream);
}

size_t dmdx_syrk_blocks_workspace_bytes(const int64_t* m, int nblocks, int64_t n) {
  if (!m || nblocks <= 0 || n <= 0) return 0;
  size_t need = 0;
  for (int j0 = 0; j0 < nblocks; j0 += MAXB) {
    for (int j = j0; j < nblocks && j < j0 + MAXB; ++j)
      if (m[j] < 1) return 0;
    const size_t g = batch_group_ws(m + j0, nblocks - j0 < MAXB ? nblocks - j0 : MAXB, n, n, 1);
    if (g > need) need = g;
  }
  return need;
}

int dmdx_syrk_blocks_f32(const float* const* X, const int64_t* m, const int64_t* ldx, int nblocks, int64_t n,
                         double* G64, int64_t ldg, float* G32, int64_t ldg32, int accumulate,
                         void* workspace, size_t workspace_bytes, void* stream) {
  DMDX_CHECK_ARG(X && m && ldx && G64 && nblocks >= 1, "syrk_blocks: null pointer or no blocks");
  DMDX_CHECK_ARG(n >= 1 && n < (1 << 30) && ldg >= n && (!G32 || ldg32 >= n), "syrk_blocks: bad n / ldg");
  for (int j = 0; j < nblocks; ++j)
    DMDX_CHECK_ARG(X[j] && m[j] >= 1 && ldx[j] >= 1, "syrk_blocks: bad block %d", j);
  return run_batch(X, ldx, X, ldx, m, nblocks, n, n, 1, G64, ldg, G32, ldg32, accumulate, workspace,
                   workspace_bytes, (hipStream_t)stream);
}

// D rows beyond one 128-row tile (l > 128 columns of Y in Z = X^T Y): the full 128-row tiles in one
// launch, the remaining rows in a second one with the tile height that pads least (64 / 96) -- when
// that is less MFMA work than any uniform tile height (l = 220: 128 + 96 = 224 rows instead of
// 256).  The second launch streams the big operand once more; these products are MFMA-bound by a
// factor > 3 at such l, so the extra HBM pass is hidden.  Returns the split row, or 0.
static int64_t tn_row_split(int64_t nrow) {
  if (nrow <= BT) return 0;
  const int64_t full = nrow / BT * BT, rem = nrow - full;
  if (rem == 0 || rem > 112) return 0;
  const int64_t rem_pad = rem <= 32 ? 32 : (rem + 15) / 16 * 16;
  int64_t best = (nrow + BT - 1) / BT * BT;
  for (int tm : {112, 96, 80, 64}) {
    const int64_t padded = (nrow + tm - 1) / tm * tm;
    if (padded < best) best = padded;
  }
  return full + rem_pad < best ? full : 0;
}

size_t dmdx_gemm_tn_workspace_bytes(int64_t K, int64_t na, int64_t nb) {
  if (K < 0 || na <= 0 || nb <= 0) return 0;
  if (const int64_t cut = tn_row_split(nb)) {
    const size_t w1 = make_plan(K, cut, na, 0).ws_bytes, w2 = make_plan(K, nb - cut, na, 0).ws_bytes;
    return w1 > w2 ? w1 : w2;
  }
  return make_plan(K, nb, na, 0).ws_bytes;
}

int dmdx_gemm_tn_f32(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t K,
                     int64_t na, int64_t nb, double* C64, int64_t ldc, float* C32, int64_t ldc32,
                     int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  DMDX_CHECK_ARG(A && B && C64, "gemm_tn: null pointer");
  DMDX_CHECK_ARG(K >= 1 && na >= 1 && nb >= 1 && na < (1 << 30) && nb < (1 << 30),
                 "gemm_tn: bad shape");
  DMDX_CHECK_ARG(lda >= 1 && ldb >= 1 && ldc >= na && (!C32 || ldc32 >= na),
                 "gemm_tn: bad leading dimension");
  DMDX_CHECK_ARG(lda < (1ll << 24) && ldb < (1ll << 24),
                 "gemm_tn: leading dimension >= 2^24 not supported (use row blocks)");
  // column-major C[a + b*ldc] == row-major D[b][a]: D rows <- B columns, D cols <- A columns
  if (const int64_t cut = tn_row_split(nb)) {
    const int rc = run_tn(B, ldb, A, lda, K, cut, na, 0, C64, ldc, C32, ldc32, accumulate, workspace,
                          workspace_bytes, (hipStream_t)stream);
    if (rc) return rc;
    return run_tn(B + cut * ldb, ldb, A, lda, K, nb - cut, na, 0, C64 + cut * ldc, ldc,
                  C32 ? C32 + cut * ldc32 : nullptr, ldc32, accumulate, workspace, workspace_bytes,
                  (hipStream_t)stream);
  }
  return run_tn(B, ldb, A, lda, K, nb, na, 0, C64, ldc, C32, ldc32, accumulate, workspace, workspace_bytes,
                (hipStream_t)stream);
}

size_t dmdx_gemm_tn_blocks_workspace_bytes(const int64_t* K, int nblocks, int64_t na, int64_t nb) {
  if (!K || nblocks <= 0 || na <= 0 || nb <= 0) return 0;
  size_t need = 0;
  for (int j0 = 0; j0 < nblocks; j0 += MAXB) {
    for (int j = j0; j < nblocks && j < j0 + MAXB; ++j)
      if (K[j] < 1) return 0;
    const int ng = nblocks - j0 < MAXB ? nblocks - j0 : MAXB;
    size_t g = batch_group_ws(K + j0, ng, nb, na, 0);
    if (const int64_t cut = tn_row_split(nb)) {
      const size_t g1 = batch_group_ws(K + j0, ng, cut, na, 0), g2 = batch_group_ws(K + j0, ng, nb - cut, na, 0);
      g = g1 > g2 ? g1 : g2;
    }
    if (g > need) need = g;
  }
  if (nb <= XL && na >= XT) {   // the small-l path (K3s) has its own partial tiles
    const size_t x = xty_small_ws(K, nblocks, na);
    if (x > need) need = x;
  }
  return need;
}

int dmdx_gemm_tn_blocks_f32(const float* const* A, const int64_t* lda, const float* const* B,
                            const int64_t* ldb, const int64_t* K, int nblocks, int64_t na, int64_t nb,
                            double* C64, int64_t ldc, float* C32, int64_t ldc32, int accumulate,
                            void* workspace, size_t workspace_bytes, void* stream) {
  DMDX_CHECK_ARG(A && B && lda && ldb && K && C64 && nblocks >= 1, "gemm_tn_blocks: null pointer or no blocks");
  DMDX_CHECK_ARG(na >= 1 && nb >= 1 && na < (1 << 30) && nb < (1 << 30) && ldc >= na && (!C32 || ldc32 >= na),
                 "gemm_tn_blocks: bad shape / ldc");
  for (int j = 0; j < nblocks; ++j)
    DMDX_CHECK_ARG(A[j] && B[j] && K[j] >= 1 && lda[j] >= 1 && ldb[j] >= 1, "gemm_tn_blocks: bad block %d", j);
  // column-major C[a + b*ldc] == row-major D[b][a]: D rows <- B columns, D cols <- A columns
  if (xty_small_ok(A, lda, B, ldb, K, nblocks, nb, na, C32))
    return run_xty_small(A, lda, B, ldb, K, nblocks, nb, na, C64, ldc, accumulate, workspace, workspace_bytes,
                         (hipStream_t)stream);
  if (const int64_t cut = tn_row_split(nb)) {
    const int rc = run_batch(B, ldb, A, lda, K, nblocks, cut, na, 0, C64, ldc, C32, ldc32, accumulate, workspace,
                             workspace_bytes, (hipStream_t)stream);
    if (rc) return rc;
    std::vector<const float*> B2((size_t)nblocks);
    for (int j = 0; j < nblocks; ++j) B2[j] = B[j] + cut * ldb[j];
    return run_batch(B2.data(), ldb, A, lda, K, nblocks, nb - cut, na, 0, C64 + cut * ldc, ldc,
                     C32 ? C32 + cut * ldc32 : nullptr, ldc32, accumulate, workspace, workspace_bytes,
                     (hipStream_t)stream);
  }
  return run_batch(B, ldb, A, lda, K, nblocks, nb, na, 0, C64, ldc, C32, ldc32, accumulate, workspace,
                   workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
