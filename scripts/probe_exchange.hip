// Scratch probe (not part of libdmdx): what does a hand-off between workgroups cost on this chip?
//   hipcc --offload-arch=gfx950 -O2 -o scripts/probe_exchange scripts/probe_exchange.hip
// A fused range-finder pass  Z += X^T (X W)  ("K4") needs, per strip of rows, the
// sum over all column tiles of the partial products  X[strip, tile] W[tile]  before any tile can go on
// with X[strip, tile]^T Y: a reduce-scatter + gather among the workgroups that hold the tiles.  This
// program times exactly that exchange, without any arithmetic around it:
//   mode 0: counter barrier only                       (one arrival + poll per round)
//   mode 1: partials -> counter -> each member sums its slice -> counter -> everyone reads the sum
// for clusters of CS workgroups that sit on ONE XCD (workgroup id % 8 equal) or are spread over all.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int SPIN_LIMIT = 1 << 20;

__device__ __forceinline__ float ldc(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stc(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct Params {
  unsigned* cnt;      // [clusters][2 counters + 1 abort flag], 64 B apart
  float* P;           // [clusters][2 slots][CS][EL]
  float* Y;           // [clusters][2 slots][EL]
  long long* ticks;   // [workgroups]
  float* sink;        // [workgroups]
  int cs, nclusters, el, rounds, mode, xcd_local, sleep;
};

// one lane arrives and polls; false after a timeout
__device__ bool arrive_wait(unsigned* c, unsigned* abort_flag, unsigned target, int sleep, int* ok) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int good = 1, spins = 0;
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (sleep) __builtin_amdgcn_s_sleep(1);
      if (++spins > SPIN_LIMIT || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        good = 0;
        break;
      }
    }
    *ok = good;
  }
  __syncthreads();
  return *ok != 0;
}

__global__ __launch_bounds__(256) void exchange_kernel(Params p) {
  __shared__ int ok;
  const int w = blockIdx.x;
  int cluster, member;
  if (p.xcd_local) {   // ids congruent mod 8 share an XCD: 32 workgroup slots per XCD, 32 / cs clusters in each
    const int xcd = w & 7, j = w >> 3;
    cluster = xcd + 8 * (j / p.cs);
    member = j % p.cs;
  } else { cluster = w / p.cs; member = w % p.cs; }
  if (cluster >= p.nclusters || member >= p.cs) { if (threadIdx.x == 0) p.ticks[w] = 0; return; }
  unsigned* cA = p.cnt + cluster * 48;
  unsigned* cB = cA + 16;
  unsigned* ab = cA + 32;
  const int tid = threadIdx.x;
  const int el = p.el, cs = p.cs;
  const int per = (el + cs - 1) / cs;
  float acc = 0.f;
  const long long t0 = wall_clock64();
  bool alive = true;
  for (int r = 0; r < p.rounds && alive; ++r) {
    const int slot = r & 1;
    if (p.mode == 1) {
      float* mine = p.P + ((size_t)(cluster * 2 + slot) * cs + member) * el;
      for (int i = tid; i < el; i += 256) stc(mine + i, (float)(member + i + r));
    }
    alive = arrive_wait(cA, ab, (unsigned)(cs * (r + 1)), p.sleep, &ok);
    if (p.mode == 1 && alive) {
      // my slice of the sum, in member order (deterministic)
      if (tid < per) {
        const int i = member * per + tid;
        if (i < el) {
          const float* base = p.P + (size_t)(cluster * 2 + slot) * cs * el + i;
          float s = 0.f;
          for (int m = 0; m < cs; ++m) s += ldc(base + (size_t)m * el);
          stc(p.Y + (size_t)(cluster * 2 + slot) * el + i, s);
        }
      }
      alive = arrive_wait(cB, ab, (unsigned)(cs * (r + 1)), p.sleep, &ok);
      if (alive) {
        const float* y = p.Y + (size_t)(cluster * 2 + slot) * el;
        for (int i = tid; i < el; i += 256) acc += ldc(y + i);
      }
    }
  }
  const long long t1 = wall_clock64();
  if (tid == 0) p.ticks[w] = alive ? (t1 - t0) : -1;
  p.sink[w * 256 + tid] = acc;
}

int main(int argc, char** argv) {
  int dev_cus = 0;
  CHECK(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, 0));
  int rate_khz = 0;
  CHECK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
  printf("CUs %d, wall clock %d kHz\n", dev_cus, rate_khz);
  const int rounds = 2000;
  struct Case { int cs, ncl, el, mode, local, sleep; };
  std::vector<Case> cases;
  for (int sleep = 0; sleep < 2; ++sleep) {
    cases.push_back({32, 1, 640, 0, 1, sleep});
    cases.push_back({32, 8, 640, 0, 1, sleep});
    cases.push_back({32, 8, 640, 0, 0, sleep});
    cases.push_back({256, 1, 640, 0, 0, sleep});
    cases.push_back({32, 1, 640, 1, 1, sleep});
    cases.push_back({32, 8, 640, 1, 1, sleep});
    cases.push_back({32, 8, 640, 1, 0, sleep});
    cases.push_back({35, 7, 640, 1, 0, sleep});
    cases.push_back({32, 8, 2048, 1, 1, sleep});
    cases.push_back({8, 8, 640, 1, 1, sleep});
    cases.push_back({8, 32, 640, 1, 1, sleep});
  }
  for (const Case& c : cases) {
    Params p{};
    p.cs = c.cs; p.nclusters = c.ncl; p.el = c.el; p.rounds = rounds; p.mode = c.mode; p.xcd_local = c.local; p.sleep = c.sleep;
    const int nwg = c.local ? 256 : c.cs * c.ncl;
    if (nwg > dev_cus) { printf("skip (needs %d workgroups)\n", nwg); continue; }
    CHECK(hipMalloc(&p.cnt, c.ncl * 48 * sizeof(unsigned)));
    CHECK(hipMemset(p.cnt, 0, c.ncl * 48 * sizeof(unsigned)));
    CHECK(hipMalloc(&p.P, (size_t)c.ncl * 2 * c.cs * c.el * sizeof(float)));
    CHECK(hipMalloc(&p.Y, (size_t)c.ncl * 2 * c.el * sizeof(float)));
    CHECK(hipMalloc(&p.ticks, nwg * sizeof(long long)));
    CHECK(hipMalloc(&p.sink, (size_t)nwg * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(exchange_kernel, dim3(nwg), dim3(256), 0, 0, p);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> t(nwg);
    CHECK(hipMemcpy(t.data(), p.ticks, nwg * sizeof(long long), hipMemcpyDeviceToHost));
    long long mx = 0; bool bad = false;
    for (long long v : t) { if (v < 0) bad = true; if (v > mx) mx = v; }
    printf("mode %d  cluster %3d x %2d  %s  el %4d  sleep %d : %s %.3f us per round (kernel %.3f ms / %d rounds = %.3f us)\n",
           c.mode, c.cs, c.ncl, c.local ? "one XCD " : "spread  ", c.el, c.sleep, bad ? "TIMEOUT" : "ok",
           1e3 * (double)mx / rate_khz / rounds, ms, rounds, 1e3 * ms / rounds);
    fflush(stdout);
    CHECK(hipFree(p.cnt)); CHECK(hipFree(p.P)); CHECK(hipFree(p.Y)); CHECK(hipFree(p.ticks)); CHECK(hipFree(p.sink));
  }
  return 0;
}
