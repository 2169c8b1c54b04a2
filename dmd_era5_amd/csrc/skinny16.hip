// K2 (round 3 body): tall-skinny Y = X W on v_mfma_f32_16x16x4_f32, l <= 256 columns in ONE pass
// over X, 16-column granular, with the Gram G = Y^T Y of the result fused in for every l <= 256.
//
// Why a second body.  The 32x32x2 body (skinny.hip) works in 32-column blocks and 128-row waves:
// l = 70 (BASELINE config 4, rank 50 + 20 oversamples) executes 96 columns (27 % padding), l > 128
// re-reads X once per 128-column group, and its fused Gram stops at 96 columns (the wave's 32 x 32
// partials no longer fit).  The 16x16x4 MFMA has the same flop rate (32 cycles for half the flops),
// and its A operand -- lane (i = lane & 15, kk = lane >> 4) holds A[i][kk] -- maps onto 16-byte
// loads of X the same way: a lane loads FOUR consecutive rows of one column, register e is the A
// operand of row block e (rows {row0 + 4 i + e}), the four lane groups kk take four columns 4 apart.
// One wave-load therefore covers 64 rows x 4 columns (256 contiguous bytes per column), a wave
// owns 64 rows x 16 C16 columns of Y in 16 C16 accumulator registers (256 at l = 256), and X goes
// HBM -> VGPR -> MFMA as before (no LDS: it has no reuse).
//
// k order.  W (n x l, L2 resident) is staged through LDS in 32-row chunks in K1's panel layout
// ([column][32 k], 16-byte k-pieces XOR-swizzled with (column >> 1) & 7).  Lane (j, kk) reads the
// piece 4 g + kk of column 16 c + j with one ds_read_b128 (g = 0, 1: the two 16-row halves of a
// chunk): element s is k = 16 g + 4 kk + s, so MFMA s of half g contracts k = 16 g + s + 4 kk over
// the lane groups kk -- and the lane's X column for (g, s) is k0 + 16 g + s + 4 kk: a scalar column
// offset plus the per-lane byte offset 4 kk ldx (raw buffer loads, no address VALU).
// The fragment reads are bank-conflict free with that swizzle (checked lane group by lane group).
//
// Pipeline (as in the 32x32x2 body): the X registers of half g are refilled for the next chunk as
// soon as the half's 16 C16 MFMAs are issued (a whole chunk of prefetch distance), the W chunk
// c + 1 is loaded at the start of chunk c and stored to the other LDS stage before the barrier,
// which sits before the second half's MFMAs.  256 rows per workgroup (4 waves x 64): a 131072-row
// block is 512 workgroups; C16 <= 7 fits 2 workgroups per CU.
//
// Fused Gram.  The accumulators ARE MFMA operands: lane (j, q) holds Y[row0 + 16 q + 4 r + e][16 c + j]
// in acc[e][c][r], which is A[i = j][kk = q] of column block c as well as B[kk = q][j]; 16 MFMAs per
// pair of column blocks (c1 <= c2) contract the wave's 64 rows into a 16 x 16 partial.  The four
// waves of a workgroup sum their partials in LDS (the W stages are free by then) in a FIXED order
// -- round t: wave w adds the pairs p = (w + t) mod 4 (mod 4) -- so no atomics and a bit-wise
// deterministic result; the workgroup writes C16 (C16 + 1) / 2 KB to its own slot, a second kernel
// sums the slots in fp64 into G.
#include <type_traits>

#include <stdlib.h>

#include "dmdx_common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int KB = 32;     // k rows per W chunk
constexpr int RW = 64;     // rows per wave
constexpr int RWG = 256;   // rows per workgroup

// TAIL4 (0 / 1 / 2): that many 4-column blocks behind the C16 16-column blocks, on v_mfma_f32_4x4x1_16b_f32 (16
// independent 4 x 4 x 1 products per instruction, 2 passes: the same flop rate).  Its A operand is the SAME
// register the 16-column blocks use: lane 4 b + i of block b holds X[row 16 (b & 3) + 4 i + e][column of lane group
// kk = b >> 2], so block b is the k-slice kk of four rows; its B operand is W[that k][16 C16 + 4 q + (lane & 3)],
// one 16-byte LDS read per half chunk like the other fragments.  The four kk slices of a row meet in the
// epilogue (two lane exchanges).  l = 20 runs 16 + 4 columns instead of 32, l = 50 48 + 4 instead of 64.
template <int C16, bool ALIGNED, bool GRAM, int WPS, int TAIL4 = 0>
__global__ __launch_bounds__(256, WPS) void skinny16_kernel(
    const float* __restrict__ X, int64_t m, int64_t n, int64_t ldx, const float* __restrict__ W,
    int64_t ldw, int l, float* __restrict__ Y, int64_t ldy, unsigned long long* clk,
    float* __restrict__ gpart) {
  static_assert(!(GRAM && TAIL4), "the fused Gram epilogue knows 16-column blocks only");
  constexpr int NCOL = 16 * C16 + 4 * TAIL4;
  constexpr int NP = C16 * (C16 + 1) / 2;
  constexpr int STG = NCOL * KB;                 // floats per W stage
  constexpr int G_FLOATS = GRAM ? NP * 256 : 0;
  constexpr int NST = 3;                          // W stages (see the counter hand-off below)
  constexpr int LDS_FLOATS = (NST * STG > G_FLOATS ? NST * STG : G_FLOATS) + 4;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  // W chunks are handed from the four storing waves to the four reading waves WITHOUT s_barrier: a
  // monotonic counter in LDS (the last word of the array) counts the waves that have stored their part
  // of a chunk; chunk c is complete at 4 c.  A wave stores chunk c + 1 in the middle of chunk c and
  // polls for it at the end of chunk c: the poll normally succeeds at once, so a wave that lags by
  // less than half a chunk stalls nobody (one wave per SIMD at l > 112: nothing else covers a
  // barrier's skew there -- the barrier + staging cost 9 % at l = 220).  Three stages make the
  // write-after-read safe without a second counter: stage (c + 2) mod 3 is written during chunk
  // c + 1 by waves that have seen chunk c + 1 complete, i.e. after every wave stored it in the middle
  // of chunk c -- when all of them were done reading chunk c - 1's stage.
  unsigned* fill_cnt = reinterpret_cast<unsigned*>(&lds[LDS_FLOATS - 4]);

  unsigned long long pc0 = 0, pr0 = 0;   // measurement aid (dmdx_set_clock_probe; null on the product path)
  if (clk != nullptr) {
    pc0 = __builtin_amdgcn_s_memtime();
    pr0 = __builtin_amdgcn_s_memrealtime();
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kk = lane >> 4;
  const int64_t rowW = (int64_t)blockIdx.x * RWG + wave * RW;
  const int64_t myrow = rowW + 4 * li;
  // loads are never guarded: out-of-range rows / columns are CLAMPED onto valid addresses (their
  // products meet zero-padded W rows or land in accumulator rows that are never stored)
  int64_t crow = myrow;
  if (ALIGNED) {
    if (crow > m - 4) crow = m - 4;   // m % 4 == 0, m >= 4
  } else {
    if (crow >= m) crow = 0;
  }
  const unsigned loffb = 4u * (unsigned)(crow + (int64_t)(4 * kk) * ldx);
  const char* Xbytes = reinterpret_cast<const char*>(X);
  const int64_t ldxb = 4 * ldx;
  const bool fast_ok = 2 * KB * ldxb < (int64_t(1) << 31) && (int64_t)l * ldw * 4 < (int64_t(1) << 31);

  f32x4 acc[4][C16];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int c = 0; c < C16; ++c) acc[e][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acct[4][TAIL4 ? TAIL4 : 1];   // 4-column blocks: [row quad e][block q], register r = row 4 r of the lane's block
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int q = 0; q < (TAIL4 ? TAIL4 : 1); ++q) acct[e][q] = f32x4{0.f, 0.f, 0.f, 0.f};

  // slow path (tail chunks, unaligned operands): kcol = the column of lane group kk = 0
  auto load_x = [&](int64_t kcol) -> f32x4 {
    int64_t kc = kcol + 4 * kk;
    kc = kc < n ? kc : n - 1;
    const float* q = X + kc * ldx + crow;
    if (ALIGNED) {
      return *reinterpret_cast<const f32x4*>(q);
    } else {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (crow + e < m) ? q[e] : 0.f;
      return v;
    }
  };

  // W staging: 128 C16 pieces of 16 bytes per chunk, piece idx = tid + 256 i -> column idx >> 3, k-piece idx & 7
  constexpr int NPW = (8 * NCOL + 255) / 256;
  constexpr int NPW_LAST = (8 * NCOL) % 256;   // threads with a piece in the last batch (0: all of them)
  f32x4 wreg[NPW];
  auto load_w = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int idx = tid + 256 * i;
      const int col = idx >> 3, wq = idx & 7;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (col < l) {
        const float* q = W + (int64_t)col * ldw + k0 + 4 * wq;
        if (ALIGNED && k0 + 4 * wq + 4 <= n) {
          v = *reinterpret_cast<const f32x4*>(q);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k0 + 4 * wq + e < n) v[e] = q[e];
        }
      }
      wreg[i] = v;
    }
  };
  unsigned woffb[NPW];
  int wslot[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int idx = tid + 256 * i;
    const int col = idx >> 3, wq = idx & 7;
    const int cc = col < l ? col : l - 1;   // columns >= l only feed output columns that are never stored
    woffb[i] = (unsigned)(((int64_t)cc * ldw + 4 * wq) * 4);
    wslot[i] = col * KB + 4 * (wq ^ ((col >> 1) & 7));
  }
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, -1, 0x00020000);
  auto load_w_fast = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < NPW; ++i)
      if (NPW_LAST == 0 || i + 1 < NPW || tid < NPW_LAST)
        wreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)woffb[i], (int)(4 * k0), 0));
  };
  auto store_w = [&](int st) {
#pragma unroll
    for (int i = 0; i < NPW; ++i)
      if (NPW_LAST == 0 || i + 1 < NPW || tid < NPW_LAST)
        *reinterpret_cast<f32x4*>(&lds[st * STG + wslot[i]]) = wreg[i];
  };

  // B fragments: lane (j = li, kk) reads the piece 4 g + kk of column 16 c + j.  A chunk is a sequence
  // of NSTEP = 2 NPART steps (half g, part p of the column blocks, at most PB = 4 blocks = 64 MFMAs
  // each); the fragments of step k + 1 are read into the OTHER of two small register sets before the
  // MFMAs of step k are issued (a full set per half, or two, cost 4 C16 / 8 C16 registers: the
  // 2-workgroups-per-CU variants and l > 208 spilled or shuffled accumulators through AGPR moves).
  constexpr int NPART = (C16 + 3) / 4;
  constexpr int PB = (C16 + NPART - 1) / NPART;
  constexpr int NSTEP = 2 * NPART;
  int foff[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) foff[g] = li * KB + 4 * ((4 * g + kk) ^ ((li >> 1) & 7));
  f32x4 bb[2][PB];
  // 4-column blocks: lane (j = lane & 3, kk) reads the piece 4 g + kk of column 16 C16 + 4 q + j, per half g
  int fofft[2][TAIL4 ? TAIL4 : 1];
  f32x4 bt[2][TAIL4 ? TAIL4 : 1];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int q = 0; q < (TAIL4 ? TAIL4 : 1); ++q) {
      const int col = 16 * C16 + 4 * q + (li & 3);
      fofft[g][q] = col * KB + 4 * ((4 * g + kk) ^ ((col >> 1) & 7));
      bt[g][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  // fragments of step k (half k / NPART, blocks PB (k % NPART) ...) from stage st into set k & 1
#define DMDX_READ_STEP(st, k)                                                                  \
  do {                                                                                         \
    constexpr int g_ = (k) / NPART, lo_ = PB * ((k) % NPART);                                  \
    constexpr int hi_ = lo_ + PB < C16 ? lo_ + PB : C16;                                       \
    _Pragma("unroll") for (int c = lo_; c < hi_; ++c)                                          \
        bb[(k) & 1][c - lo_] = *reinterpret_cast<const f32x4*>(&lds[(st) * STG + 16 * c * KB + foff[g_]]); \
    if constexpr (TAIL4 != 0 && (k) % NPART == 0) {                                            \
      _Pragma("unroll") for (int q = 0; q < TAIL4; ++q)                                        \
          bt[g_][q] = *reinterpret_cast<const f32x4*>(&lds[(st) * STG + fofft[g_][q]]);        \
    }                                                                                          \
    __builtin_amdgcn_sched_group_barrier(0x100, hi_ - lo_ + ((k) % NPART == 0 ? TAIL4 : 0), 0); \
  } while (0)
  // the MFMAs of step k: 4 k-steps x 4 row blocks x its column blocks, on the 4 X quads of its half
#if defined(DMDX_K2_ABL) && (DMDX_K2_ABL & 1)   /* timing only: no MFMAs */
#define DMDX_MFMA_STEP(k)                                                                      \
  do {                                                                                         \
    constexpr int g_ = (k) / NPART;                                                            \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) asm volatile("" ::"v"(xq[g_][s]));           \
    _Pragma("unroll") for (int c = 0; c < PB; ++c) asm volatile("" ::"v"(bb[(k) & 1][c]));     \
  } while (0)
#else
#define DMDX_MFMA_STEP(k)                                                                      \
  do {                                                                                         \
    constexpr int g_ = (k) / NPART, lo_ = PB * ((k) % NPART);                                  \
    constexpr int hi_ = lo_ + PB < C16 ? lo_ + PB : C16;                                       \
    _Pragma("unroll") for (int s = 0; s < 4; ++s)                                              \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                          \
            _Pragma("unroll") for (int c = lo_; c < hi_; ++c) acc[e][c] =                      \
                __builtin_amdgcn_mfma_f32_16x16x4f32(xq[g_][s][e], bb[(k) & 1][c - lo_][s], acc[e][c], 0, 0, 0); \
    if constexpr (TAIL4 != 0 && (k) % NPART == 0) {                                            \
      _Pragma("unroll") for (int s = 0; s < 4; ++s)                                            \
          _Pragma("unroll") for (int e = 0; e < 4; ++e)                                        \
              _Pragma("unroll") for (int q = 0; q < TAIL4; ++q) acct[e][q] =                   \
                  __builtin_amdgcn_mfma_f32_4x4x1f32(xq[g_][s][e], bt[g_][q][s], acct[e][q], 0, 0, 0); \
    }                                                                                          \
    __builtin_amdgcn_sched_group_barrier(0x008, 16 * (hi_ - lo_) + ((k) % NPART == 0 ? 16 * TAIL4 : 0), 0); \
  } while (0)
#endif

  const int nchunks = (int)((n + KB - 1) / KB);
  f32x4 xq[2][4];   // X quads of the current chunk: [half g][s] = lane column k0 + 16 g + s + 4 kk

  if (tid == 0) *fill_cnt = 0u;
  load_w(0);
  store_w(0);
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int s = 0; s < 4; ++s) xq[g][s] = load_x(16 * g + s);
  __syncthreads();
  DMDX_READ_STEP(0, 0);

  int cur = 0;
  unsigned want = 0;   // 4 x (index of the chunk whose W is needed next)
  auto step = [&](auto kt, auto fast_tag, const bool has_next, auto& reload, auto& publish, auto& early_store) {
    constexpr int k = decltype(kt)::value;
    // before the MFMAs of step k: the fragments of step k + 1 (the last step's successor is step 0 of
    // the next chunk, in the other stage: behind the barrier)
    if constexpr (k + 1 < NSTEP) {
      DMDX_READ_STEP(cur, k + 1);
    } else {
      publish();
      if (has_next) DMDX_READ_STEP(cur, 0);
    }
    DMDX_MFMA_STEP(k);
    if constexpr ((k + 1) % NPART == 0) reload(k / NPART);   // the half's X registers are free: next chunk's
    // W chunk c + 1 goes into the other stage in the MIDDLE of the chunk (nobody reads that stage between
    // the previous barrier and the next one), so that no LDS store or load wait is pending at the barrier
    if constexpr (k + 1 == NSTEP / 2) early_store();
  };
  auto chunk = [&](int c, auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    const int64_t k0 = (int64_t)c * KB;
    // (a FAST chunk always has a successor: its own columns end at k0 + KB <= n - KB)
    const bool has_next = FAST ? true : (c + 1 < nchunks);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(Xbytes + (k0 + KB) * ldxb), 0, -1, 0x00020000);  // 4 GiB window at column k0 + KB
    auto reload = [&](int g) {   // half g of chunk c + 1
      if (!has_next) return;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if constexpr (FAST) {
#if defined(DMDX_K2_ABL) && (DMDX_K2_ABL & 2)   /* timing only: no X loads */
          f32x4 v = {(float)s, 1.f, 2.f, 3.f};
          asm volatile("" : "+v"(v));
          xq[g][s] = v;
#else
          xq[g][s] = __builtin_bit_cast(
              f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)loffb, (16 * g + s) * (int)ldxb, 0));
#endif
        } else {
          xq[g][s] = load_x(k0 + KB + 16 * g + s);
        }
      }
      if constexpr (FAST) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
    };
    const int nxt = cur == NST - 1 ? 0 : cur + 1;
    auto early_store = [&]() {
      if (has_next) {
        store_w(nxt);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's part of chunk c + 1 is in LDS ...
        if (lane == 0) __hip_atomic_fetch_add(fill_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ... before it says so
      }
    };
    auto publish = [&]() {   // wait (normally not at all) until all four waves have stored chunk c + 1
      want += 4u;
#if !(defined(DMDX_K2_ABL) && (DMDX_K2_ABL & 4))   /* ABL 4, timing only: no hand-off */
      if (has_next) {
        while (__hip_atomic_load(fill_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
      }
#endif
      cur = nxt;
    };
    if (has_next) {
      if constexpr (FAST) load_w_fast(k0 + KB);
      else load_w(k0 + KB);
    }
    if constexpr (FAST) __builtin_amdgcn_sched_group_barrier(0x020, NPW, 0);
    step(std::integral_constant<int, 0>{}, fast_tag, has_next, reload, publish, early_store);
    step(std::integral_constant<int, 1>{}, fast_tag, has_next, reload, publish, early_store);
    if constexpr (NSTEP > 2) {
      step(std::integral_constant<int, 2 % NSTEP>{}, fast_tag, has_next, reload, publish, early_store);
      step(std::integral_constant<int, 3 % NSTEP>{}, fast_tag, has_next, reload, publish, early_store);
    }
    if constexpr (NSTEP > 4) {
      step(std::integral_constant<int, 4 % NSTEP>{}, fast_tag, has_next, reload, publish, early_store);
      step(std::integral_constant<int, 5 % NSTEP>{}, fast_tag, has_next, reload, publish, early_store);
    }
    if constexpr (NSTEP > 6) {
      step(std::integral_constant<int, 6 % NSTEP>{}, fast_tag, has_next, reload, publish, early_store);
      step(std::integral_constant<int, 7 % NSTEP>{}, fast_tag, has_next, reload, publish, early_store);
    }
  };
  int c = 0;
  if (ALIGNED && fast_ok)  // every X column these iterations load (up to k0 + 2 KB - 1) is inside the matrix
    for (; (int64_t)(c + 2) * KB <= n; ++c) chunk(c, std::true_type{});
  for (; c < nchunks; ++c) chunk(c, std::false_type{});
#undef DMDX_READ_STEP
#undef DMDX_MFMA_STEP

  // ---- epilogue: lane (j = li, q = kk) holds, in register r of acc[e][c], MFMA row 4 q + r of row
  // block e -> global row rowW + 16 q + 4 r + e (e = 0..3 contiguous), column 16 c + j
#pragma unroll
  for (int cb = 0; cb < C16; ++cb) {
    const int col = 16 * cb + li;
    if (col >= l) continue;
    float* yc = Y + (int64_t)col * ldy;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = rowW + 16 * kk + 4 * r;
      if (ALIGNED && row + 4 <= m) {
        f32x4 v = {acc[0][cb][r], acc[1][cb][r], acc[2][cb][r], acc[3][cb][r]};
        *reinterpret_cast<f32x4*>(yc + row) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (row + e < m) yc[row + e] = acc[e][cb][r];
      }
    }
  }

  if constexpr (TAIL4 != 0) {
    // block b = lane >> 2 of a 4-column result is the k-slice kk = b >> 2 of the rows 16 (b & 3) + 4 r + e: the four
    // slices sit 16 lanes apart; lanes kk = 0 store the sums
#pragma unroll
    for (int q = 0; q < TAIL4; ++q) {
      const int col = 16 * C16 + 4 * q + (li & 3);
      float* yc = Y + (int64_t)(col < l ? col : 0) * ldy;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        f32x4 v = {acct[0][q][r], acct[1][q][r], acct[2][q][r], acct[3][q][r]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] += __shfl_xor(v[e], 16, 64);
          v[e] += __shfl_xor(v[e], 32, 64);
        }
        const int64_t row = rowW + 16 * (li >> 2) + 4 * r;
        if (kk == 0 && col < l) {
          if (ALIGNED && row + 4 <= m) {
            *reinterpret_cast<f32x4*>(yc + row) = v;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (row + e < m) yc[row + e] = v[e];
          }
        }
      }
    }
  }

  if constexpr (GRAM) {
    // rows past the end hold clamped duplicates, columns >= l copies of column l - 1: not part of Y
    if (rowW + RW > m) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (rowW + 16 * kk + 4 * r + e >= m) {
#pragma unroll
            for (int cb = 0; cb < C16; ++cb) acc[e][cb][r] = 0.f;
          }
    }
    if (16 * (C16 - 1) + li >= l) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e][C16 - 1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();   // the W stages are free
    // pair p of round t belongs to wave (p - t) mod 4: every pair is summed wave 0-first ... in a
    // fixed cyclic order (p mod 4 first), one barrier per round, no atomics
#pragma unroll 1
    for (int t = 0; t < 4; ++t) {
      int p = 0;
#pragma unroll
      for (int c1 = 0; c1 < C16; ++c1)
#pragma unroll
        for (int c2 = c1; c2 < C16; ++c2) {
          if (((p - t) & 3) == wave) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[e][c1][r], acc[e][c2][r], d, 0, 0, 0);
            // d[r'] = G_p[4 q + r'][j]; stored at [p][r'][lane] (64 consecutive floats per register)
            float* gq = &lds[p * 256 + lane];
            if (t == 0) {
#pragma unroll
              for (int r = 0; r < 4; ++r) gq[64 * r] = d[r];
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) gq[64 * r] += d[r];
            }
          }
          ++p;
        }
      __syncthreads();
    }
    float* gp = gpart + (size_t)blockIdx.x * (NP * 256);
    for (int o = 4 * tid; o < NP * 256; o += 1024)
      *reinterpret_cast<f32x4*>(gp + o) = *reinterpret_cast<const f32x4*>(&lds[o]);
  }
  if (clk != nullptr) {
    const unsigned long long pc1 = __builtin_amdgcn_s_memtime(), pr1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
      atomicAdd(&clk[0], pc1 - pc0);
      atomicAdd(&clk[1], pr1 - pr0);
      atomicAdd(&clk[2], 1ull);
    }
  }
}

// sums the per-workgroup 16 x 16 partials of every block pair in fp64 into G (l x l): one workgroup
// per (pair, 32 of its 256 elements); 8 slot lanes x 32 elements: every thread sums its slots s, s + 8, ...
// (coalesced 128-byte reads), the 8 lanes meet in LDS in a fixed order (deterministic)
__global__ __launch_bounds__(256) void skinny16_gram_reduce_kernel(const float* __restrict__ gpart, int nslots, int c16,
                                                                   int l, double* __restrict__ G, int64_t ldg,
                                                                   int accumulate) {
  __shared__ double part[8][32];
  const int np = c16 * (c16 + 1) / 2;
  const int j = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int pair = blockIdx.x >> 3, o = 32 * (blockIdx.x & 7) + j;   // element o of the pair's [r'][q][j] image
  const size_t off = (size_t)pair * 256 + o;
  double s = 0.0;
  for (int k = sl; k < nslots; k += 8) s += (double)gpart[(size_t)k * ((size_t)np * 256) + off];
  part[sl][j] = s;
  __syncthreads();
  if (sl != 0) return;
#pragma unroll
  for (int q = 1; q < 8; ++q) s += part[q][j];
  int c1 = 0, c2 = 0, p = pair;
  for (c1 = 0; c1 < c16; ++c1) {
    if (p < c16 - c1) { c2 = c1 + p; break; }
    p -= c16 - c1;
  }
  const int rr = o >> 6, q = (o >> 4) & 3, jj = o & 15;
  const int gi = 16 * c1 + 4 * q + rr, gj = 16 * c2 + jj;
  if (gi >= l || gj >= l) return;
  if (accumulate) {
    G[(int64_t)gi * ldg + gj] += s;
    if (c1 != c2) G[(int64_t)gj * ldg + gi] += s;
  } else {
    G[(int64_t)gi * ldg + gj] = s;
    if (c1 != c2) G[(int64_t)gj * ldg + gi] = s;
  }
}

template <int C16, bool GRAM, int TAIL4 = 0>
int launch16(const float* X, int64_t m, int64_t n, int64_t ldx, const float* W, int64_t ldw, int l, float* Y,
             int64_t ldy, hipStream_t stream, float* gpart, double* G, int64_t ldg, int accumulate) {
  const bool aligned = (m % 4 == 0) && (m >= 4) && (ldx % 4 == 0) && (ldw % 4 == 0) && (ldy % 4 == 0) &&
                       dmdx_aligned16(X) && dmdx_aligned16(W) && dmdx_aligned16(Y);
  // two workgroups per CU while the accumulators leave room for it (<= 256 registers per lane)
  constexpr int WPS = (C16 <= 7) ? 2 : 1;
  dim3 grid((unsigned)((m + RWG - 1) / RWG));
  if (aligned)
    hipLaunchKernelGGL((skinny16_kernel<C16, true, GRAM, WPS, TAIL4>), grid, dim3(256), 0, stream, X, m, n, ldx, W, ldw, l, Y,
                       ldy, dmdx_clock_probe_ptr, gpart);
  else
    hipLaunchKernelGGL((skinny16_kernel<C16, false, GRAM, WPS, TAIL4>), grid, dim3(256), 0, stream, X, m, n, ldx, W, ldw, l, Y,
                       ldy, dmdx_clock_probe_ptr, gpart);
  DMDX_LAUNCH_CHECK();
  if constexpr (GRAM) {
    constexpr int NP = C16 * (C16 + 1) / 2;
    hipLaunchKernelGGL(skinny16_gram_reduce_kernel, dim3(NP * 8), dim3(256), 0, stream, gpart, (int)grid.x, C16, l, G,
                       ldg, accumulate);
    DMDX_LAUNCH_CHECK();
  }
  return 0;
}

template <bool GRAM>
int dispatch16(int c16, const float* X, int64_t m, int64_t n, int64_t ldx, const float* W, int64_t ldw, int l,
               float* Y, int64_t ldy, hipStream_t st, float* gp, double* G, int64_t ldg, int acc) {
  switch (c16) {
#define DMDX_CASE(C) case C: return launch16<C, GRAM>(X, m, n, ldx, W, ldw, l, Y, ldy, st, gp, G, ldg, acc)
    DMDX_CASE(1); DMDX_CASE(2); DMDX_CASE(3); DMDX_CASE(4); DMDX_CASE(5); DMDX_CASE(6); DMDX_CASE(7); DMDX_CASE(8);
    DMDX_CASE(9); DMDX_CASE(10); DMDX_CASE(11); DMDX_CASE(12); DMDX_CASE(13); DMDX_CASE(14); DMDX_CASE(15);
    DMDX_CASE(16);
#undef DMDX_CASE
  }
  dmdx_set_error("skinny16: unsupported column count %d", l);
  return DMDX_E_INVALID;
}

// 16 c16 + 4 t4 columns (t4 = 1, 2; c16 <= 4: l = 17 .. 72 with l % 16 in 1 .. 8), no fused Gram
int dispatch16_tail(int c16, int t4, const float* X, int64_t m, int64_t n, int64_t ldx, const float* W, int64_t ldw,
                    int l, float* Y, int64_t ldy, hipStream_t st) {
#define DMDX_CASE(C, T) \
  if (c16 == C && t4 == T) return launch16<C, false, T>(X, m, n, ldx, W, ldw, l, Y, ldy, st, nullptr, nullptr, 0, 0)
  DMDX_CASE(1, 1); DMDX_CASE(1, 2); DMDX_CASE(2, 1); DMDX_CASE(2, 2);
  DMDX_CASE(3, 1); DMDX_CASE(3, 2); DMDX_CASE(4, 1); DMDX_CASE(4, 2);
#undef DMDX_CASE
  dmdx_set_error("skinny16: no 4-column-block kernel for %d + %d blocks", c16, t4);
  return DMDX_E_INVALID;
}

}  // namespace

// columns the 16 + 4 granular body runs for l (0: not eligible -- l <= 16, more than 8 columns past a multiple of
// 16, or beyond the instantiated sizes)
static int tail_blocks(int l) {
  if (getenv("DMDX_K2_NO_TAIL") != nullptr) return 0;
  const int rem = l % 16;
  if (l <= 16 || l > 72 || rem == 0 || rem > 8) return 0;
  return (rem + 3) / 4;
}

bool dmdx_skinny16_shape_ok(int64_t m, int64_t ldx) {
  return ldx < (int64_t(1) << 23) && m + 12 * ldx < (int64_t(1) << 29);
}

int dmdx_skinny16_launch(const float* X, int64_t m, int64_t n, int64_t ldx, const float* W, int64_t ldw, int l, float* Y,
                         int64_t ldy, hipStream_t st) {
  if (const int t4 = tail_blocks(l)) return dispatch16_tail(l / 16, t4, X, m, n, ldx, W, ldw, l, Y, ldy, st);
  return dispatch16<false>((l + 15) / 16, X, m, n, ldx, W, ldw, l, Y, ldy, st, nullptr, nullptr, 0, 0);
}

size_t dmdx_skinny16_gram_ws(int64_t m, int64_t l) {
  const size_t c = (size_t)((l + 15) / 16);
  return (size_t)((m + RWG - 1) / RWG) * (c * (c + 1) / 2) * 256 * sizeof(float);
}

int dmdx_skinny16_gram_launch(const float* X, int64_t m, int64_t n, int64_t ldx, const float* W, int64_t ldw, int l,
                              float* Y, int64_t ldy, double* G, int64_t ldg, int accumulate, float* gpart,
                              hipStream_t st) {
  return dispatch16<true>((l + 15) / 16, X, m, n, ldx, W, ldw, l, Y, ldy, st, gpart, G, ldg, accumulate);
}
