// K2: tall-skinny Y = X W on the fp32-input MFMA, X streamed once from HBM.
//
// X: m x n column-major (rows contiguous).  The MFMA A operand wants, per lane
// (i = lane&31, h = lane>>5), the element A[i][k=h]: a lane loads FOUR
// consecutive rows of column k0+2s+h with one global_load_dwordx4 and uses
// register e as the A operand of row-block e, i.e. row-block e holds the rows
// {row0 + 4*i + e}.  That is only a permutation of the rows inside the wave's
// 128-row strip, undone when Y is stored (each lane then owns 4 consecutive
// rows of its column -> one 16-byte store).  X therefore goes HBM -> VGPR ->
// MFMA with full 16 B/lane loads and no LDS round trip (it has no reuse: every
// workgroup owns its rows and all l columns).
// W (n x l, small, L2 resident) is staged through LDS in 32-row chunks, read
// back as the B operand with conflict-free ds_read_b32.
//
// Workgroup = 4 waves = 512 rows; per wave 128 rows x 32*C columns of Y in
// 4*C accumulators; K loop over n in chunks of 32 (16 MFMA k-steps), X loads
// issued half a chunk (8 steps = 8 KiB per wave) ahead of their use.
#include <type_traits>

#include "dmdx_common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int KB = 32;          // k rows per W chunk
constexpr int LDW = KB + 1;     // padded LDS row of the transposed W chunk
constexpr int ROWS_PER_WAVE = 128;
constexpr int ROWS_PER_WG = 512;

template <int C, bool ALIGNED>
__global__ __launch_bounds__(256, (C <= 2 ? 2 : 1)) void skinny_kernel(
    const float* __restrict__ X, int64_t m, int64_t n, int64_t ldx, const float* __restrict__ W,
    int64_t ldw, int l, float* __restrict__ Y, int64_t ldy) {
  __shared__ float Ws[2][32 * C * LDW];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int64_t rowW = (int64_t)blockIdx.x * ROWS_PER_WG + wave * ROWS_PER_WAVE;
  const int64_t myrow = rowW + 4 * l31;
  // Loads are never guarded: out-of-range rows / columns are CLAMPED onto valid
  // addresses instead (their products are multiplied by zero-padded W rows, or
  // land in accumulator rows that are never stored).  FAST requires m % 4 == 0
  // and 16-byte aligned bases / leading dimensions (checked by the host).
  int64_t crow = myrow;
  if (ALIGNED) {
    if (crow > m - 4) crow = m - 4;  // m % 4 == 0, m >= 4: a fully discarded lane
  } else {
    if (crow >= m) crow = 0;         // partial lanes keep their rows (element guards)
  }
  // per-lane 32-bit element offsets (host guarantees m + ldx < 2^29); the column
  // base ku*ldx is wave-uniform and stays in SGPRs (saddr addressing)
  const unsigned loff0 = (unsigned)crow;
  const unsigned loff = (unsigned)(crow + (int64_t)lh * ldx);
  const int64_t ku_max = (n - 1) & ~(int64_t)1;  // last even column index
  // fast path (every column of the iteration in range): raw buffer loads -- a descriptor at
  // the chunk's first column (SGPRs), the column as a scalar byte offset, the lane's rows as
  // this 32-bit per-lane byte offset: no VALU and no 64-bit arithmetic per load
  const unsigned loffb = 4u * loff;
  const char* Xbytes = reinterpret_cast<const char*>(X);
  const int64_t ldxb = 4 * ldx;
  const bool fast_ok = 2 * KB * ldxb < (int64_t(1) << 31);  // scalar byte offsets stay 32-bit

  f32x16 acc[4][C];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[e][c][r] = 0.f;

  // ku: wave-uniform even k (the lane's own column is ku + lh)
  auto load_x = [&](int64_t ku) -> f32x4 {
    const int64_t kc = ku < ku_max ? ku : ku_max;      // scalar clamp
    const float* q = X + kc * ldx;                       // uniform base
    const unsigned off = (kc + lh < n) ? loff : loff0;   // odd-n last column
#if defined(DMDX_K2_ABL) && DMDX_K2_ABL == 2   /* timing only: no X loads */
    if (ALIGNED) {
      f32x4 v = {(float)off, (float)kc, 1.f, 2.f};
      asm volatile("" : "+v"(v));
      return v;
    }
#endif
    if (ALIGNED) {
      return *reinterpret_cast<const f32x4*>(q + off);
    } else {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (crow + e < m) ? q[off + e] : 0.f;
      return v;
    }
  };

  // W staging: C pieces of 16 bytes per thread per chunk
  const int wcol = tid >> 3, wq = tid & 7;
  f32x4 wreg[C];
  auto load_w = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < C; ++i) {
      int col = wcol + 32 * i;
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
  auto store_w = [&](int st) {
#pragma unroll
    for (int i = 0; i < C; ++i) {
      float* d = &Ws[st][(wcol + 32 * i) * LDW + 4 * wq];
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = wreg[i][e];
    }
  };

  const int nchunks = (int)((n + KB - 1) / KB);
  f32x4 xa[8], xb[8];

  load_w(0);
  store_w(0);
#pragma unroll
  for (int s = 0; s < 8; ++s) xa[s] = load_x(2 * s);
  __syncthreads();

  // One k-step = 4*C MFMAs on one X register quad and C values of W.  The W values of
  // step s+1 are read from LDS before the MFMAs of step s are issued (bn), so the LDS
  // latency is covered by the 4*C*64 matrix-pipe cycles of the step.
#define DMDX_READ_B(dst, s_)                                                         \
  do {                                                                               \
    _Pragma("unroll") for (int cc = 0; cc < C; ++cc) dst[cc] = ws[32 * cc * LDW + 2 * (s_)]; \
  } while (0)
#if defined(DMDX_K2_ABL) && DMDX_K2_ABL == 1   /* timing only: no MFMAs */
#define DMDX_STEP(xreg, bv)                                                          \
  do {                                                                               \
    asm volatile("" ::"v"(xreg));                                                    \
    _Pragma("unroll") for (int cc = 0; cc < C; ++cc) asm volatile("" ::"v"(bv[cc])); \
  } while (0)
#else
#define DMDX_STEP(xreg, bv)                                                          \
  do {                                                                               \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                    \
        _Pragma("unroll") for (int cc = 0; cc < C; ++cc) acc[e][cc] =                \
            __builtin_amdgcn_mfma_f32_32x32x2f32(xreg[e], bv[cc], acc[e][cc], 0, 0, 0); \
  } while (0)
#endif

  int cur = 0;
  auto chunk = [&](int c, auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    const int64_t k0 = (int64_t)c * KB;
    const bool has_next = c + 1 < nchunks;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(Xbytes + k0 * ldxb), 0, -1, 0x00020000);  // 4 GiB window at column k0
    auto ld = [&](int koff) -> f32x4 {   // columns k0 + koff, k0 + koff + 1
      if constexpr (FAST) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)loffb, koff * (int)ldxb, 0);
        return __builtin_bit_cast(f32x4, v);
      } else {
        return load_x(k0 + koff);
      }
    };
#pragma unroll
    for (int s = 0; s < 8; ++s) xb[s] = ld(16 + 2 * s);
    if (has_next) load_w(k0 + KB);

    const float* ws = &Ws[cur][l31 * LDW + lh];
    float b0[C], b1[C];
    DMDX_READ_B(b0, 0);
#pragma unroll
    for (int s = 0; s < 8; s += 2) {
      DMDX_READ_B(b1, s + 1);
      __builtin_amdgcn_sched_group_barrier(0x100, (C + 1) / 2, 0);
      DMDX_STEP(xa[s], b0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * C, 0);
      DMDX_READ_B(b0, s + 2);  // s + 2 == 8 is the first step of the second half
      __builtin_amdgcn_sched_group_barrier(0x100, (C + 1) / 2, 0);
      DMDX_STEP(xa[s + 1], b1);
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * C, 0);
    }
    if (has_next) {
#pragma unroll
      for (int s = 0; s < 8; ++s) xa[s] = ld(KB + 2 * s);
    }
#pragma unroll
    for (int s = 0; s < 8; s += 2) {
      DMDX_READ_B(b1, 8 + s + 1);
      __builtin_amdgcn_sched_group_barrier(0x100, (C + 1) / 2, 0);
      DMDX_STEP(xb[s], b0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * C, 0);
      if (s + 2 < 8) {
        DMDX_READ_B(b0, 8 + s + 2);
        __builtin_amdgcn_sched_group_barrier(0x100, (C + 1) / 2, 0);
      }
      DMDX_STEP(xb[s + 1], b1);
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * C, 0);
    }
    if (has_next) store_w(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  };
  int c = 0;
  if (ALIGNED && fast_ok)  // every X column these iterations load (up to k0 + 2 KB - 1) is inside the matrix
    for (; (int64_t)(c + 2) * KB <= n; ++c) chunk(c, std::true_type{});
  for (; c < nchunks; ++c) chunk(c, std::false_type{});
#undef DMDX_READ_B
#undef DMDX_STEP

  // ---- epilogue: lane (j = l31, h) holds, for register r, MFMA row
  // i_m = (r&3) + 8*(r>>2) + 4*h of each row-block e  ->  global rows
  // rowW + 4*i_m + e (e = 0..3 contiguous), column 32*cc + j.
#pragma unroll
  for (int cc = 0; cc < C; ++cc) {
    const int col = 32 * cc + l31;
    if (col >= l) continue;
    float* yc = Y + (int64_t)col * ldy;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int im = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int64_t row = rowW + 4 * im;
      if (ALIGNED && row + 4 <= m) {
        f32x4 v = {acc[0][cc][r], acc[1][cc][r], acc[2][cc][r], acc[3][cc][r]};
        *reinterpret_cast<f32x4*>(yc + row) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (row + e < m) yc[row + e] = acc[e][cc][r];
      }
    }
  }
}

template <int C>
int launch_skinny(const float* X, int64_t m, int64_t n, int64_t ldx, const float* W, int64_t ldw,
                  int l, float* Y, int64_t ldy, hipStream_t stream) {
  const bool aligned = (m % 4 == 0) && (m >= 4) && (ldx % 4 == 0) && (ldw % 4 == 0) &&
                       (ldy % 4 == 0) && dmdx_aligned16(X) && dmdx_aligned16(W) &&
                       dmdx_aligned16(Y);
  dim3 grid((unsigned)((m + ROWS_PER_WG - 1) / ROWS_PER_WG));
  if (aligned)
    hipLaunchKernelGGL((skinny_kernel<C, true>), grid, dim3(256), 0, stream, X, m, n, ldx, W, ldw, l,
                       Y, ldy);
  else
    hipLaunchKernelGGL((skinny_kernel<C, false>), grid, dim3(256), 0, stream, X, m, n, ldx, W, ldw,
                       l, Y, ldy);
  DMDX_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int dmdx_gemm_nn_skinny_f32(const float* X, int64_t m, int64_t n, int64_t ldx,
                                       const float* W, int64_t ldw, int64_t l, float* Y,
                                       int64_t ldy, void* stream) {
  DMDX_CHECK_ARG(X && W && Y, "skinny: null pointer");
  DMDX_CHECK_ARG(m >= 1 && n >= 1 && l >= 1, "skinny: bad shape m=%lld n=%lld l=%lld",
                 (long long)m, (long long)n, (long long)l);
  DMDX_CHECK_ARG(ldx >= 1 && ldw >= n && ldy >= m, "skinny: bad leading dimension");
  DMDX_CHECK_ARG(m + ldx < (1ll << 29), "skinny: m + ldx >= 2^29 not supported");
  hipStream_t st = (hipStream_t)stream;
  // column groups of at most 128 (4 MFMA blocks); X is re-read per group
  for (int64_t c0 = 0; c0 < l; c0 += 128) {
    int lg = (int)((l - c0) < 128 ? (l - c0) : 128);
    const float* Wg = W + c0 * ldw;
    float* Yg = Y + c0 * ldy;
    int rc;
    if (lg <= 32) rc = launch_skinny<1>(X, m, n, ldx, Wg, ldw, lg, Yg, ldy, st);
    else if (lg <= 64) rc = launch_skinny<2>(X, m, n, ldx, Wg, ldw, lg, Yg, ldy, st);
    else if (lg <= 96) rc = launch_skinny<3>(X, m, n, ldx, Wg, ldw, lg, Yg, ldy, st);
    else rc = launch_skinny<4>(X, m, n, ldx, Wg, ldw, lg, Yg, ldy, st);
    if (rc) return rc;
  }
  return 0;
}
