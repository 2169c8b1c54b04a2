// K2: tall-skinny Y = X W on the fp32-input MFMA, X streamed once from HBM.
//
// X: m x n column-major (rows contiguous).  The MFMA A operand wants, per lane
// (i = lane&31, h = lane>>5), one element A[i][k]: a lane loads FOUR consecutive rows of one
// column with a single 16-byte load and uses register e as the A operand of row-block e, i.e.
// row-block e holds the rows {row0 + 4*i + e}.  That is only a permutation of the rows inside
// the wave's 128-row strip, undone when Y is stored (each lane then owns 4 consecutive rows of
// its column -> one 16-byte store).  X therefore goes HBM -> VGPR -> MFMA with full 16 B/lane
// loads and no LDS round trip (it has no reuse: every workgroup owns its rows and all l columns).
// W (n x l, small, L2 resident) is staged through LDS in 32-row chunks with K1's panel layout
// ([column][32 k], 16-byte pieces XOR-swizzled) and read back with one conflict-free
// ds_read_b128 per 4 MFMA k-steps: the 4 MFMAs j = 0..3 of step group t contract
// k = 8t + 4h + j, the same permutation on both operands.
//
// Workgroup = 4 waves = 512 rows; per wave 128 rows x 32*C columns of Y in 4*C accumulators;
// K loop over n in chunks of 32 (4 step groups of 16 C MFMAs); full chunks load X with raw
// buffer loads (descriptor + scalar column offset + 32-bit lane offset: no address VALU), each
// group's registers refilled for the next chunk right after its MFMAs are issued (a whole chunk
// of prefetch distance, pinned with sched_group_barrier -- the scheduler otherwise sinks the
// loads to their uses), barrier before the last step group.  cfg2 pass at l = 62: 9.5 ms
// (3.8 TB/s of X, 81 % of the 64-column MFMA bound); l = 128: 16.6 ms = 140 TFLOP/s.
#include <stdlib.h>

#include <type_traits>

#include "dmdx_common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int KB = 32;          // k rows per W chunk
constexpr int ROWS_PER_WAVE = 128;
constexpr int ROWS_PER_WG = 512;

template <int C, bool ALIGNED, bool GRAM = false>
__global__ __launch_bounds__(256, 1) void skinny_kernel(
    const float* __restrict__ X, int64_t m, int64_t n, int64_t ldx, const float* __restrict__ W,
    int64_t ldw, int l, float* __restrict__ Y, int64_t ldy, unsigned long long* clk,
    float* __restrict__ gpart) {
  // measurement aid (dmdx_set_clock_probe; null on the product path): core-clock cycles and
  // 100 MHz reference ticks of this workgroup's lifetime are added to clk[0..2] at the end
  unsigned long long pc0 = 0, pr0 = 0;
  if (clk != nullptr) {
    pc0 = __builtin_amdgcn_s_memtime();
    pr0 = __builtin_amdgcn_s_memrealtime();
  }
  // W chunk image: [stage][column][32 k], unpadded; the 16-byte k-pieces of a column are stored
  // at piece index q ^ swz(column) so that the ds_read_b128 fragment reads below (32 columns x 2
  // pieces per wave) are bank-conflict free -- the layout of K1's operand panels
  __shared__ __attribute__((aligned(16))) float Ws[2][32 * C * KB];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int64_t rowW = (int64_t)blockIdx.x * ROWS_PER_WG + wave * ROWS_PER_WAVE;
  const int64_t myrow = rowW + 4 * l31;
  // Loads are never guarded: out-of-range rows / columns are CLAMPED onto valid
  // addresses instead (their products are multiplied by zero-padded W rows, or
  // land in accumulator rows that are never stored).  ALIGNED requires m % 4 == 0
  // and 16-byte aligned bases / leading dimensions (checked by the host).
  int64_t crow = myrow;
  if (ALIGNED) {
    if (crow > m - 4) crow = m - 4;  // m % 4 == 0, m >= 4: a fully discarded lane
  } else {
    if (crow >= m) crow = 0;         // partial lanes keep their rows (element guards)
  }
  // k order: the 4 MFMAs j = 0..3 of step group t contract k = 8t + 4h + j (h = lane >> 5) --
  // the same permutation on both operands, so one 16-byte LDS read of W feeds 4 MFMAs (as in K1).
  // The lane's X column for (t, j) is k0 + 8t + j + 4h: a uniform column offset plus this
  // per-lane 32-bit byte offset (host guarantees m + 4 ldx < 2^29).
  const unsigned loffb = 4u * (unsigned)(crow + (int64_t)(4 * lh) * ldx);
  const char* Xbytes = reinterpret_cast<const char*>(X);
  const int64_t ldxb = 4 * ldx;
  // fast path (every column of the iteration inside the matrix): raw buffer loads -- a
  // descriptor at the chunk's first column (SGPRs), the column as a scalar byte offset, the
  // lane's rows as the 32-bit per-lane byte offset: no VALU, no 64-bit arithmetic per load
  const bool fast_ok = 2 * KB * ldxb < (int64_t(1) << 31) &&  // scalar byte offsets stay 32-bit
                       (int64_t)l * ldw * 4 < (int64_t(1) << 31);

  f32x16 acc[4][C];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[e][c][r] = 0.f;

  // slow path (tail chunks, unaligned operands): kcol = uniform column of lane half 0
  auto load_x = [&](int64_t kcol) -> f32x4 {
    const int64_t ka = kcol < n ? kcol : n - 1, kb = kcol + 4 < n ? kcol + 4 : n - 1;  // scalar clamps
    const float* q = X + (lh ? kb : ka) * ldx + crow;
    if (ALIGNED) {
      return *reinterpret_cast<const f32x4*>(q);
    } else {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (crow + e < m) ? q[e] : 0.f;
      return v;
    }
  };

  // W staging: C pieces of 16 bytes per thread per chunk (column wcol + 32 i, k-piece wq)
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
  // fast path (the whole chunk inside the matrix): one unguarded buffer load per piece; the
  // columns >= l are clamped onto column l-1 (they only feed output columns that are never stored)
  unsigned woffb[C];
#pragma unroll
  for (int i = 0; i < C; ++i) {
    const int col = wcol + 32 * i < l ? wcol + 32 * i : l - 1;
    woffb[i] = (unsigned)(((int64_t)col * ldw + 4 * wq) * 4);
  }
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(W), 0, -1, 0x00020000);
  auto load_w_fast = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < C; ++i)
      wreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)woffb[i], (int)(4 * k0), 0));
  };
  const int wslot = 4 * (wq ^ ((wcol >> 1) & 7));  // swz(wcol + 32 i) == swz(wcol)
  auto store_w = [&](int st) {
#pragma unroll
    for (int i = 0; i < C; ++i)
      *reinterpret_cast<f32x4*>(&Ws[st][(wcol + 32 * i) * KB + wslot]) = wreg[i];
  };

  // B fragments of step group t: lane (c = l31, h) reads the piece (2t + h) of column 32 cc + c
  int foff[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) foff[t] = l31 * KB + 4 * ((2 * t + lh) ^ ((l31 >> 1) & 7));
#define DMDX_READ_B(dst, st, t)                                                      \
  do {                                                                               \
    _Pragma("unroll") for (int cc = 0; cc < C; ++cc)                                 \
        dst[cc] = *reinterpret_cast<const f32x4*>(&Ws[st][32 * cc * KB + foff[t]]);  \
  } while (0)
#if defined(DMDX_K2_ABL) && DMDX_K2_ABL == 1   /* timing only: no MFMAs */
#define DMDX_GROUP(xr, bv)                                                           \
  do {                                                                               \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) asm volatile("" ::"v"((xr)[j]));   \
    _Pragma("unroll") for (int cc = 0; cc < C; ++cc) asm volatile("" ::"v"(bv[cc])); \
  } while (0)
#else
  // one step group = 4 k-steps x 4 row-blocks x C column blocks = 16 C MFMAs on 4 X quads
#define DMDX_GROUP(xr, bv)                                                           \
  do {                                                                               \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                    \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                \
            _Pragma("unroll") for (int cc = 0; cc < C; ++cc) acc[e][cc] =            \
                __builtin_amdgcn_mfma_f32_32x32x2f32((xr)[j][e], bv[cc][j], acc[e][cc], 0, 0, 0); \
    __builtin_amdgcn_sched_group_barrier(0x008, 16 * C, 0);                          \
  } while (0)
#endif

  const int nchunks = (int)((n + KB - 1) / KB);
  f32x4 xq[4][4];   // X quads of the current chunk: [step group t][j] = lane columns k0 + 8 t + j + 4 h
  f32x4 b0[C], b1[C];

  load_w(0);
  store_w(0);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) xq[t][j] = load_x(8 * t + j);
  __syncthreads();
  DMDX_READ_B(b0, 0, 0);

  // Chunk c: as soon as the MFMAs of step group t have been issued, the X quads of group t of
  // chunk c + 1 are loaded into the same registers (a whole chunk = 64 C MFMAs ahead of their
  // use).  The W chunk c + 1 is loaded at the start and stored to the other LDS stage before the
  // barrier, which sits BEFORE the last step group: the wave leaves it with 16 C MFMAs to issue
  // while its first W fragment of chunk c + 1 is read.  W fragments of group t + 1 are read while
  // the MFMAs of group t run.
  int cur = 0;
  auto chunk = [&](int c, auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    const int64_t k0 = (int64_t)c * KB;
    // (a FAST chunk always has a successor: its own columns end at k0 + KB <= n - KB)
    const bool has_next = FAST ? true : (c + 1 < nchunks);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(Xbytes + (k0 + KB) * ldxb), 0, -1, 0x00020000);  // 4 GiB window at column k0 + KB
    auto reload = [&](int t) {   // group t of chunk c + 1: lane columns k0 + KB + 8 t + j + 4 h
      if (!has_next) return;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (FAST) {
#if defined(DMDX_K2_ABL) && DMDX_K2_ABL == 2   /* timing only: no X loads */
          f32x4 v = {(float)j, 1.f, 2.f, 3.f};
          asm volatile("" : "+v"(v));
          xq[t][j] = v;
#else
          xq[t][j] = __builtin_bit_cast(
              f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)loffb, (8 * t + j) * (int)ldxb, 0));
#endif
        } else {
          xq[t][j] = load_x(k0 + KB + 8 * t + j);
        }
      }
      // (the VMEM groups pin the loads where they are written: left alone, the scheduler sinks
      // them next to their first use and the prefetch distance is gone)
      if constexpr (FAST) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
    };
    if (has_next) {
      if constexpr (FAST) load_w_fast(k0 + KB);  // FAST: columns k0 .. k0 + 2 KB - 1 exist
      else load_w(k0 + KB);
    }
    if constexpr (FAST) __builtin_amdgcn_sched_group_barrier(0x020, C, 0);
    DMDX_READ_B(b1, cur, 1);
    __builtin_amdgcn_sched_group_barrier(0x100, C, 0);
    DMDX_GROUP(xq[0], b0);
    reload(0);
    DMDX_READ_B(b0, cur, 2);
    __builtin_amdgcn_sched_group_barrier(0x100, C, 0);
    DMDX_GROUP(xq[1], b1);
    reload(1);
    DMDX_READ_B(b1, cur, 3);
    __builtin_amdgcn_sched_group_barrier(0x100, C, 0);
    DMDX_GROUP(xq[2], b0);
    reload(2);
    if (has_next) store_w(cur ^ 1);
    __syncthreads();  // every wave has read stage `cur` for the last time; chunk c + 1 is in the other
    cur ^= 1;
    if (has_next) DMDX_READ_B(b0, cur, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, C, 0);
    DMDX_GROUP(xq[3], b1);
    reload(3);
  };
  int c = 0;
  if (ALIGNED && fast_ok)  // every X column these iterations load (up to k0 + 2 KB - 1) is inside the matrix
    for (; (int64_t)(c + 2) * KB <= n; ++c) chunk(c, std::true_type{});
  for (; c < nchunks; ++c) chunk(c, std::false_type{});
#undef DMDX_READ_B
#undef DMDX_GROUP

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
  // ---- fused Gram of the block just written (CholeskyQR of the range finder: G = Y^T Y would
  // otherwise be another pass over the m x l matrix): the accumulators ARE the MFMA operands --
  // lane (j, h) holds Y[row(e, r, h)][32 cc + j], which is A[i = j][k = h] of block cc as well as
  // B[k = h][j] -- so 64 MFMAs per pair of column blocks contract the wave's 128 rows; the
  // 32 x 32 fp32 partial of every (workgroup, wave, block pair) goes to its own slot (no atomics:
  // deterministic) and a second kernel sums them in fp64.
  if constexpr (GRAM) {
    if (rowW + ROWS_PER_WAVE > m) {   // rows past the end hold clamped duplicates: not part of Y
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = rowW + 4 * ((r & 3) + 8 * (r >> 2) + 4 * lh) + e;
          if (row >= m) {
#pragma unroll
            for (int cc = 0; cc < C; ++cc) acc[e][cc][r] = 0.f;
          }
        }
    }
    constexpr int NPAIR = C * (C + 1) / 2;
    float* gp = gpart + ((size_t)blockIdx.x * 4 + wave) * (NPAIR * 1024);
    int pair = 0;
#pragma unroll
    for (int c1 = 0; c1 < C; ++c1)
#pragma unroll
      for (int c2 = c1; c2 < C; ++c2) {
        f32x16 d;
#pragma unroll
        for (int q = 0; q < 16; ++q) d[q] = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            d = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[e][c1][r], acc[e][c2][r], d, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 16; ++q)
          gp[pair * 1024 + ((q & 3) + 8 * (q >> 2) + 4 * lh) * 32 + l31] = d[q];
        ++pair;
      }
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

// sums the per-(workgroup, wave) 32 x 32 partials of every block pair in fp64 into G (l x l): one
// workgroup per row of 32 output elements, 8 slot lanes x 32 elements; every thread sums its slots
// s, s + 8, ... (coalesced 128-byte reads), the 8 lanes meet in LDS in a fixed order (deterministic)
template <int C>
__global__ __launch_bounds__(256) void skinny_gram_reduce_kernel(const float* __restrict__ gpart, int nslots, int l,
                                                                 double* __restrict__ G, int64_t ldg, int accumulate) {
  constexpr int NPAIR = C * (C + 1) / 2;
  __shared__ double part[8][32];
  const int j = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int pair = blockIdx.x >> 5, i = blockIdx.x & 31;
  const size_t off = (size_t)pair * 1024 + (size_t)i * 32 + j;
  double s = 0.0;
  for (int k = sl; k < nslots; k += 8) s += (double)gpart[(size_t)k * (NPAIR * 1024) + off];
  part[sl][j] = s;
  __syncthreads();
  if (sl != 0) return;
#pragma unroll
  for (int q = 1; q < 8; ++q) s += part[q][j];
  int c1 = 0, c2 = 0, p = pair;
  for (c1 = 0; c1 < C; ++c1) {
    if (p < C - c1) { c2 = c1 + p; break; }
    p -= C - c1;
  }
  const int gi = 32 * c1 + i, gj = 32 * c2 + j;
  if (gi >= l || gj >= l) return;
  if (accumulate) {
    G[(int64_t)gi * ldg + gj] += s;
    if (c1 != c2) G[(int64_t)gj * ldg + gi] += s;
  } else {
    G[(int64_t)gi * ldg + gj] = s;
    if (c1 != c2) G[(int64_t)gj * ldg + gi] = s;
  }
}

template <int C>
int launch_skinny(const float* X, int64_t m, int64_t n, int64_t ldx, const float* W, int64_t ldw,
                  int l, float* Y, int64_t ldy, hipStream_t stream, float* gpart = nullptr, double* G = nullptr,
                  int64_t ldg = 0, int accumulate = 0) {
  const bool aligned = (m % 4 == 0) && (m >= 4) && (ldx % 4 == 0) && (ldw % 4 == 0) &&
                       (ldy % 4 == 0) && dmdx_aligned16(X) && dmdx_aligned16(W) &&
                       dmdx_aligned16(Y);
  dim3 grid((unsigned)((m + ROWS_PER_WG - 1) / ROWS_PER_WG));
  if constexpr (C <= 3) {   // (the fused Gram at C = 4 would spill: 256 accumulator registers + the 32 x 32 tile)
    if (gpart != nullptr) {
      if (aligned)
        hipLaunchKernelGGL((skinny_kernel<C, true, true>), grid, dim3(256), 0, stream, X, m, n, ldx, W, ldw, l,
                           Y, ldy, dmdx_clock_probe_ptr, gpart);
      else
        hipLaunchKernelGGL((skinny_kernel<C, false, true>), grid, dim3(256), 0, stream, X, m, n, ldx, W, ldw,
                           l, Y, ldy, dmdx_clock_probe_ptr, gpart);
    }
  }
  if (gpart == nullptr || C > 3) {
    if (aligned)
      hipLaunchKernelGGL((skinny_kernel<C, true>), grid, dim3(256), 0, stream, X, m, n, ldx, W, ldw, l,
                         Y, ldy, dmdx_clock_probe_ptr, nullptr);
    else
      hipLaunchKernelGGL((skinny_kernel<C, false>), grid, dim3(256), 0, stream, X, m, n, ldx, W, ldw,
                         l, Y, ldy, dmdx_clock_probe_ptr, nullptr);
  }
  DMDX_LAUNCH_CHECK();
  if (gpart != nullptr && C <= 3) {
    constexpr int NPAIR = C * (C + 1) / 2;
    hipLaunchKernelGGL(skinny_gram_reduce_kernel<C>, dim3(NPAIR * 32), dim3(256), 0, stream, gpart,
                       (int)grid.x * 4, l, G, ldg, accumulate);
    DMDX_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace

// Which K2 body runs.  The 16x16x4 body (skinny16.hip: one pass over X up to 224 columns, 16-column
// granular, fused Gram up to 224 columns) is the default; DMDX_K2_IMPL=old selects the 32x32x2 body
// above (A/B measurements; it also remains the fallback for leading dimensions the new body's
// 32-bit lane offsets cannot reach).
static int k2_impl() {   // (read per call: scripts/ab_k2.py flips it inside one process)
  const char* e = getenv("DMDX_K2_IMPL");
  return (e && e[0] == 'o') ? 0 : 1;
}
constexpr int64_t K2_GROUP = 224;   // widest column group of the 16x16x4 body (14 blocks: no register spill)

extern "C" int dmdx_gemm_nn_skinny_f32(const float* X, int64_t m, int64_t n, int64_t ldx,
                                       const float* W, int64_t ldw, int64_t l, float* Y,
                                       int64_t ldy, void* stream) {
  DMDX_CHECK_ARG(X && W && Y, "skinny: null pointer");
  DMDX_CHECK_ARG(m >= 1 && n >= 1 && l >= 1, "skinny: bad shape m=%lld n=%lld l=%lld",
                 (long long)m, (long long)n, (long long)l);
  DMDX_CHECK_ARG(ldx >= 1 && ldw >= n && ldy >= m, "skinny: bad leading dimension");
  DMDX_CHECK_ARG(m + 4 * ldx < (1ll << 29), "skinny: m + 4 ldx >= 2^29 not supported");
  hipStream_t st = (hipStream_t)stream;
  // (l <= 32 is HBM-bound: the 32x32x2 body reads 512-byte runs per column and wave-load, the
  // 16x16x4 body 256-byte runs, and streams X 4-5 % slower there: scripts/ab_k2.py)
  // (17 <= l <= 24 as one 16-column + one or two 4-column blocks of the 16x16x4 body -- 20 or 24 columns of MFMA
  // work instead of 32 -- was measured too: 6.09 against 5.75 ms per cfg2 pass; the 32x32x2 body is at the HBM rate there)
  if (k2_impl() == 1 && l > 32 && dmdx_skinny16_shape_ok(m, ldx)) {
    // column groups of at most 224, equally wide up to the 16-column granule; X is re-read per group
    const int64_t ng = (l + K2_GROUP - 1) / K2_GROUP;
    const int64_t per = ((l + ng - 1) / ng + 15) / 16 * 16;
    for (int64_t c0 = 0; c0 < l; c0 += per) {
      const int lg = (int)((l - c0) < per ? (l - c0) : per);
      const int rc = dmdx_skinny16_launch(X, m, n, ldx, W + c0 * ldw, ldw, lg, Y + c0 * ldy, ldy, st);
      if (rc) return rc;
    }
    return 0;
  }
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

extern "C" int dmdx_gemm_nn_skinny_gram_max_l(void) { return k2_impl() == 1 ? (int)K2_GROUP : 96; }

static size_t old_gram_ws(int64_t m, int64_t l) {
  if (l > 96) return 0;
  const size_t c = (size_t)((l + 31) / 32);
  return (size_t)((m + ROWS_PER_WG - 1) / ROWS_PER_WG) * 4 * (c * (c + 1) / 2) * 1024 * sizeof(float);
}

extern "C" size_t dmdx_gemm_nn_skinny_gram_workspace_bytes(int64_t m, int64_t l) {
  if (m < 1 || l < 1 || l > dmdx_gemm_nn_skinny_gram_max_l()) return 0;
  // (the larger of the two bodies' needs: which one runs also depends on ldx, unknown here)
  const size_t a = k2_impl() == 1 ? dmdx_skinny16_gram_ws(m, l) : 0, b = old_gram_ws(m, l);
  return a > b ? a : b;
}

extern "C" int dmdx_gemm_nn_skinny_gram_f32(const float* X, int64_t m, int64_t n, int64_t ldx, const float* W,
                                            int64_t ldw, int64_t l, float* Y, int64_t ldy, double* G, int64_t ldg,
                                            int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  DMDX_CHECK_ARG(X && W && Y && G, "skinny_gram: null pointer");
  const bool use16 = k2_impl() == 1 && l > 32 && dmdx_skinny16_shape_ok(m, ldx);
  const int64_t lmax = use16 ? K2_GROUP : 96;
  DMDX_CHECK_ARG(m >= 1 && n >= 1 && l >= 1 && l <= lmax, "skinny_gram: bad shape m=%lld n=%lld l=%lld (l <= %lld)",
                 (long long)m, (long long)n, (long long)l, (long long)lmax);
  DMDX_CHECK_ARG(ldx >= 1 && ldw >= n && ldy >= m && ldg >= l, "skinny_gram: bad leading dimension");
  DMDX_CHECK_ARG(m + 4 * ldx < (1ll << 29), "skinny_gram: m + 4 ldx >= 2^29 not supported");
  const size_t need = use16 ? dmdx_skinny16_gram_ws(m, l) : old_gram_ws(m, l);
  if (workspace == nullptr || workspace_bytes < need) {
    dmdx_set_error("skinny_gram: workspace %zu bytes < required %zu", workspace_bytes, need);
    return DMDX_E_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* gp = reinterpret_cast<float*>(workspace);
  const int li = (int)l;
  if (use16) return dmdx_skinny16_gram_launch(X, m, n, ldx, W, ldw, li, Y, ldy, G, ldg, accumulate, gp, st);
  if (l <= 32) return launch_skinny<1>(X, m, n, ldx, W, ldw, li, Y, ldy, st, gp, G, ldg, accumulate);
  if (l <= 64) return launch_skinny<2>(X, m, n, ldx, W, ldw, li, Y, ldy, st, gp, G, ldg, accumulate);
  return launch_skinny<3>(X, m, n, ldx, W, ldw, li, Y, ldy, st, gp, G, ldg, accumulate);
}
