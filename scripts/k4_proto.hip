// Scratch prototype (not part of libdmdx): the fused range-finder pass  Z = X^T (X W)  ("K4", DESIGN.md sections 0 / 7)
// as a cluster kernel, to put a measured number next to the two-pass K2 + K3s it would replace.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/k4_proto scripts/k4_proto.hip
//   ./scripts/k4_proto            (correctness on a small matrix, then timing on one cfg2 row block 129 760 x 8760)
//
// X: one row block, column-major with the space axis contiguous (X[r + t * ld]), mb rows (space) x n columns (time).
// W: n x 32 (l <= 32 padded to 32, row t = 32 floats).  Z: n x 32.
// Work split: strips of 32 rows; strip s belongs to cluster s % 8 = the workgroups with blockIdx % 8 equal (one XCD);
// member c = blockIdx >> 3 of a cluster owns the time columns [144 c, 144 c + 144).  Per strip a member
//   P1  multiplies its 32 x 144 tile (LDS-resident, LDS-DMA staged) by its 144 rows of W  -> a 32 x 32 partial of Y,
//   exchanges: partial -> global (sc1), counter A; sums ITS 16-element slice of all members' partials in a fixed order,
//              slice -> global, counter B; reads the 32 x 32 sum,
//   P2  accumulates Z[its 144 columns] += tile^T Y  in registers over all strips.
// The stages of four strips are in flight per workgroup (loading / P1 done / slice summed / P2), so each counter has a
// whole iteration to fill; wave 3 issues every LDS-DMA piece and no other memory operation, so that its vmcnt counts
// LDS-DMA only and a strip has a whole iteration to land.  Both products run on v_mfma_f32_16x16x4_f32 with l padded to 32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int T = 144, R = 32, LP = 32, NSTG = 4;
constexpr int STG = T * R;              // floats per stage (18 KB)
constexpr int NPIECE = T / 8;           // LDS-DMA wave-instructions per strip (8 columns x 128 B each)
constexpr int SPIN_LIMIT = 1 << 18;

struct K4Params {
  const float* X; long long ld; int n, nstrips;
  const float* W;            // n x 32
  float* P;                  // [8 clusters][4 slots][64 members][1024]   partials, layout [col][32 rows]
  float* Y;                  // [8][4][1024]                               sums
  unsigned* cnt;             // [8][2 counters, 64 B apart]; cnt[1024] = abort flag
  float* Zp;                 // [8][n][32]
  int G;                     // members per cluster = ceil(n / 144)
};

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
// workgroup barrier WITHOUT the vmcnt(0) the compiler puts in front of __syncthreads(): wave 3's LDS-DMA must stay in flight
#define BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ void dma16(unsigned lds_addr, unsigned off, const char* base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(lds_addr), "v"(off), "s"(base) : "memory");
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, -1, 0x00020000);
}

__global__ __launch_bounds__(256, 2) void k4_cluster_kernel(K4Params p) {
  __shared__ __attribute__((aligned(16))) float lds[NSTG * STG + 1024];
  __shared__ int s_ok;
  float* ybuf = lds + NSTG * STG;
  float* pimg = ybuf;   // this member's 32 x 32 partial before it goes out: the same 4 KB (barriers of the counter waits between the two uses)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cluster = blockIdx.x & 7, member = blockIdx.x >> 3;
  if (member >= p.G) return;
  const int G = p.G, n = p.n;
  const int t0 = member * T;
  const int K = (p.nstrips - cluster + 7) / 8;           // strips of this cluster: cluster, cluster + 8, ...
  unsigned* cA = p.cnt + cluster * 32;
  unsigned* cB = cA + 16;
  unsigned* abort_flag = p.cnt + 1024;
  const int li = lane & 15, kk = lane >> 4;

  // ---- W operand of P1, resident: wave (rb = wave & 1, cb = wave >> 1); lane (j = li, kk) holds W[t0 + 4 s + kk][16 cb + j]
  const int rb = wave & 1, cb = wave >> 1;
  float wreg[T / 4];
#pragma unroll
  for (int s = 0; s < T / 4; ++s) {
    const int t = t0 + 4 * s + kk;
    wreg[s] = t < n ? p.W[(size_t)t * LP + 16 * cb + li] : 0.f;
  }
#pragma unroll
  for (int s = 0; s < T / 4; ++s) asm volatile("" :: "v"(wreg[s]));   // (the loads are waited for HERE, not at their first use inside the loop)
  // ---- LDS-DMA: wave w issues the pieces w, w + 4, ...; piece q = columns 8 q .. 8 q + 7; lane -> column 8 q + (lane >> 3),
  // LDS slot lane & 7 holds the 16-byte row piece (lane & 7) ^ (col & 7)
  unsigned doff[NPIECE];                                  // (only wave 3 uses them: it issues every piece and NO other VM operation
#pragma unroll                                            //  inside the loop, so that its vmcnt counts LDS-DMA only)
  for (int q = 0; q < NPIECE; ++q) {
    const int col = 8 * q + (lane >> 3);
    int tc = t0 + col;
    tc = tc < n ? tc : n - 1;                              // (columns past the matrix: duplicates; their W rows are zero)
    doff[q] = (unsigned)(((long long)(tc - t0) * p.ld + 4 * ((lane & 7) ^ (col & 7))) * 4);
  }
  const char* Xb = reinterpret_cast<const char*>(p.X + (long long)t0 * p.ld);
  auto issue = [&](int k) {                                // strip k of this cluster into stage k & 3
    const long long row0 = 32ll * (cluster + 8 * k);
    const char* src = Xb + 4 * row0;
    float* dst = lds + (k & 3) * STG;
#pragma unroll
    for (int q = 0; q < NPIECE; ++q) dma16((unsigned)(uintptr_t)LDS_PTR(dst + 8 * q * R), doff[q], src);
  };
  // bounded wait of thread 0 on a counter; everybody learns the outcome
  auto wait_for = [&](unsigned* c, unsigned target) -> bool {
    if (tid == 0) {
      int good = 1, spins = 0;
      while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          good = 0;
          break;
        }
      }
      s_ok = good;
    }
    BAR();
    const bool ok = s_ok != 0;
    BAR();
    return ok;
  };

  // ---- P2 accumulators: wave (nb = wave & 1, half = wave >> 1): l block nb, time blocks 5 half .. 5 half + 4 (of 9)
  const int nb = wave & 1, half = wave >> 1;
  f32x4 zacc[5];
#pragma unroll
  for (int b = 0; b < 5; ++b) zacc[b] = f32x4{0.f, 0.f, 0.f, 0.f};

  float* Pc = p.P + (size_t)cluster * 4 * 64 * 1024;
  float* Yc = p.Y + (size_t)cluster * 4 * 1024;
  const __amdgpu_buffer_rsrc_t prs = rsrc(Pc), yrs = rsrc(Yc);

  if (K > 0 && wave == 3) issue(0);
  bool alive = true;
  for (int v = 0; v < K + 2 && alive; ++v) {
    if (wave == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // strip v has landed (issued a whole iteration ago)
    BAR();
    if (wave == 3 && v + 1 < K) issue(v + 1);             // into the stage of strip v - 3 (its P2 ran in iteration v - 1)
    // ---- P1 on strip v
    if (v < K) {
      const float* st = lds + (v & 3) * STG;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const int rowp = 4 * rb + (li >> 2), rowe = li & 3;  // row 16 rb + li = piece rowp, element rowe
#pragma unroll
      for (int s = 0; s < T / 4; ++s) {
        const int col = 4 * s + kk;
        const float a = st[col * R + 4 * (rowp ^ (col & 7)) + rowe];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[s], acc, 0, 0, 0);
      }
      // lane (j = li, q = kk), register r: Y_part[row 16 rb + 4 q + r][col 16 cb + j]; image [col][32 rows]
      *reinterpret_cast<f32x4*>(pimg + (16 * cb + li) * 32 + 16 * rb + 4 * kk) = acc;
      BAR();
      if (wave < 3) {                                       // (waves 0 .. 2 carry all global traffic)
        const size_t imgoff = ((size_t)(v & 3) * 64 + member) * 1024;
        for (int o = 4 * tid; o < 1024; o += 768)
          __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(pimg + o), prs, (int)((imgoff + o) * 4), 0, 16 /* sc1 */);
      }
    }
    // ---- my slice(s) of the sum of strip v - 1 (counter A was raised by every member at the end of its iteration v - 1)
    if (v >= 1 && v - 1 < K) {
      alive = wait_for(cA, (unsigned)(G * v));
      if (!alive) break;
      const int slot = (v - 1) & 3;
      // 64 slices of 16 elements: member c takes the slices c, c + G, c + 2 G, ... (one per wave 0 .. 2 and round)
      if (wave < 3)
        for (int slice = member + G * wave; slice < 64; slice += 3 * G) {
          float vals[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int m = kk + 4 * i;
            vals[i] = m < G ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, (int)((((size_t)slot * 64 + m) * 1024 + 16 * slice + li) * 4), 0, 16))
                            : 0.f;
          }
          float sum = 0.f;
#pragma unroll
          for (int i = 0; i < 16; ++i) sum += vals[i];
          sum += __shfl_xor(sum, 16, 64);
          sum += __shfl_xor(sum, 32, 64);
          if (kk == 0)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sum), yrs, (slot * 1024 + 16 * slice + li) * 4, 0, 16);
        }
    }
    // ---- P2 on strip v - 2 (counter B: the slices of strip v - 2 went out in iteration v - 1)
    if (v >= 2 && v - 2 < K) {
      alive = wait_for(cB, (unsigned)(G * (v - 1)));   // (its barriers also separate the image's last read from the Y buffer's write)
      if (!alive) break;
      const int slot = (v - 2) & 3;
      if (wave < 3)
        for (int o = 4 * tid; o < 1024; o += 768) {
          const u32x4 yb = __builtin_amdgcn_raw_buffer_load_b128(yrs, (slot * 1024 + o) * 4, 0, 16);
          *reinterpret_cast<u32x4*>(ybuf + o) = yb;
        }
      BAR();
      const float* st = lds + ((v - 2) & 3) * STG;
      f32x4 ya[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) ya[h] = *reinterpret_cast<const f32x4*>(ybuf + (16 * nb + li) * 32 + 4 * (4 * h + kk));
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        const int blk = 5 * half + b;
        if (blk < T / 16) {
          const int col = 16 * blk + li;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x4 xb = *reinterpret_cast<const f32x4*>(st + col * R + 4 * ((4 * h + kk) ^ (col & 7)));
#pragma unroll
            for (int e = 0; e < 4; ++e) zacc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ya[h][e], xb[e], zacc[b], 0, 0, 0);
          }
        }
      }
    }
    // ---- this iteration's stores are out (waves 0 .. 2), then the arrivals
    if (wave < 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BAR();
    if (tid == 0) {
      if (v < K) __hip_atomic_fetch_add(cA, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v >= 1 && v - 1 < K) __hip_atomic_fetch_add(cB, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (wave == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // ---- Z of this cluster: lane (j = li, q = kk), register r: Z[t0 + 16 blk + j][16 nb + 4 q + r]
  if (alive) {
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      const int blk = 5 * half + b;
      const int t = t0 + 16 * blk + li;
      if (blk < T / 16 && t < n)
        *reinterpret_cast<f32x4*>(p.Zp + ((size_t)cluster * n + t) * LP + 16 * nb + 4 * kk) = zacc[b];
    }
  }
}

// ---- plain reference kernels (fp64 accumulation), for the small case only
__global__ void ref_xw(const float* X, long long ld, int mb, int n, const float* W, double* Y) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (r >= mb) return;
  double s = 0.0;
  for (int t = 0; t < n; ++t) s += (double)X[r + (long long)t * ld] * (double)W[(size_t)t * LP + c];
  Y[(size_t)c * mb + r] = s;
}
__global__ void ref_xty(const float* X, long long ld, int mb, int n, const double* Y, double* Z) {
  const int t = blockIdx.x, c = threadIdx.x;
  double s = 0.0;
  for (int r = 0; r < mb; ++r) s += (double)X[r + (long long)t * ld] * Y[(size_t)c * mb + r];
  Z[(size_t)t * LP + c] = s;
}

static int run_case(int mb, int n, bool check, int reps) {
  const long long ld = mb;
  const int nstrips = mb / R, G = (n + T - 1) / T;
  if (G > 64) { printf("n = %d needs %d members (> 64)\n", n, G); return 1; }
  float *X, *W, *P, *Y, *Zp; unsigned* cnt;
  CHECK(hipMalloc(&X, (size_t)mb * n * 4)); CHECK(hipMalloc(&W, (size_t)n * LP * 4));
  CHECK(hipMalloc(&P, (size_t)8 * 4 * 64 * 1024 * 4)); CHECK(hipMalloc(&Y, (size_t)8 * 4 * 1024 * 4));
  CHECK(hipMalloc(&Zp, (size_t)8 * n * LP * 4)); CHECK(hipMalloc(&cnt, 2048 * 4));
  std::vector<float> hX((size_t)mb * n), hW((size_t)n * LP);
  unsigned st = 12345u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
  for (auto& v : hX) v = rnd();
  for (int t = 0; t < n; ++t) for (int c = 0; c < LP; ++c) hW[(size_t)t * LP + c] = c < 20 ? rnd() : 0.f;   // l = 20, padded
  CHECK(hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  K4Params p{X, ld, n, nstrips, W, P, Y, cnt, Zp, G};
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < reps; ++rep) {
    CHECK(hipMemset(cnt, 0, 2048 * 4));
    CHECK(hipMemset(Zp, 0, (size_t)8 * n * LP * 4));
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k4_cluster_kernel, dim3(512), dim3(256), 0, 0, p);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  unsigned flag = 0; CHECK(hipMemcpy(&flag, cnt + 1024, 4, hipMemcpyDeviceToHost));
  printf("%d x %d (%d strips, %d members per cluster): %.3f ms per launch = %.2f TB/s of X%s\n", mb, n, nstrips, G, best,
         (double)nstrips * R * n * 4 / best / 1e9, flag ? "   *** a wait timed out ***" : "");
  int bad = flag ? 1 : 0;
  if (check && !flag) {
    double *Yr, *Zr; CHECK(hipMalloc(&Yr, (size_t)LP * mb * 8)); CHECK(hipMalloc(&Zr, (size_t)n * LP * 8));
    const int mu = nstrips * R;                       // (rows past the last full strip are not part of the prototype)
    hipLaunchKernelGGL(ref_xw, dim3((mu + 255) / 256, LP), dim3(256), 0, 0, X, ld, mu, n, W, Yr);
    hipLaunchKernelGGL(ref_xty, dim3(n), dim3(LP), 0, 0, X, ld, mu, n, Yr, Zr);
    CHECK(hipDeviceSynchronize());
    std::vector<double> hZr((size_t)n * LP); std::vector<float> hZp((size_t)8 * n * LP);
    CHECK(hipMemcpy(hZr.data(), Zr, hZr.size() * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hZp.data(), Zp, hZp.size() * 4, hipMemcpyDeviceToHost));
    double maxref = 0, maxerr = 0;
    for (size_t i = 0; i < hZr.size(); ++i) {
      double s = 0; for (int c = 0; c < 8; ++c) s += hZp[(size_t)c * n * LP + i];
      maxref = fmax(maxref, fabs(hZr[i])); maxerr = fmax(maxerr, fabs(s - hZr[i]));
    }
    printf("   max |Z - Z_fp64| = %.3e of max |Z| = %.3e  -> %s\n", maxerr, maxref, maxerr <= 2e-5 * maxref ? "ok" : "MISMATCH");
    if (!(maxerr <= 2e-5 * maxref)) bad = 1;
    CHECK(hipFree(Yr)); CHECK(hipFree(Zr));
  }
  CHECK(hipFree(X)); CHECK(hipFree(W)); CHECK(hipFree(P)); CHECK(hipFree(Y)); CHECK(hipFree(Zp)); CHECK(hipFree(cnt));
  return bad;
}

int main(int argc, char** argv) {
  if (run_case(32 * 8 * 3, 300, true, 1)) return 1;        // 24 strips, 3 members
  if (run_case(32 * 8 * 5 + 64, 1000, true, 2)) return 1;  // 42 strips (uneven over the clusters), 7 members
  if (run_case(32 * 64, 8760, true, 2)) return 1;          // 61 members, last one 120 columns wide
  if (argc > 1 && atoi(argv[1]) == 0) return 0;
  return run_case(129760, 8760, false, 5);                 // one cfg2 row block (4055 strips): K2 + K3s take 1.57 ms for it
}
