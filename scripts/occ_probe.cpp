// scratch: are two 320-thread / 64 KB-LDS workgroups co-resident on one CU?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int NT>
__global__ __launch_bounds__(NT, 3) void k(unsigned long long* out, long long spin) {
  __shared__ float pad[16384];
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  pad[threadIdx.x] = (float)spin;
  long long c0 = clock64();
  while (clock64() - c0 < spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t0; out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }
  if (pad[threadIdx.x] < 0) out[0] = 0;
}
template <int NT> void run(const char* name) {
  int n = 512; unsigned long long* d; hipMalloc(&d, 2 * n * sizeof(unsigned long long));
  hipLaunchKernelGGL(k<NT>, dim3(n), dim3(NT), 0, 0, d, 2000000LL);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(2 * n); hipMemcpy(h.data(), d, 2 * n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  unsigned long long tmin = ~0ull; for (int i = 0; i < n; ++i) tmin = std::min(tmin, h[2 * i]);
  int late = 0; for (int i = 0; i < n; ++i) if (h[2 * i] - tmin > 10000) ++late;  // 100 MHz ticks: > 100 us
  int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k<NT>, NT, 0);
  printf("%s: %d of %d blocks started > 100 us after the first (API says %d blocks/CU)\n", name, late, n, occ);
}
int main() { run<256>("256 threads"); run<320>("320 threads"); run<384>("384 threads"); run<512>("512 threads"); return 0; }
