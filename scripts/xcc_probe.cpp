// scratch: which XCD does block b land on?  (HW_REG_XCC_ID)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 2) void k(int* out, int spin) {
  __shared__ float pad[18432];  // 72 KB like the Gram kernel => 2 WG/CU
  unsigned x = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));  // XCC_ID bits [3:0] -> size 4 -> (4-1)<<11
  unsigned cu;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(cu));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = (int)(x & 15); out[2 * blockIdx.x + 1] = (int)cu; }
  pad[threadIdx.x] = spin;
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
  if (pad[threadIdx.x] < 0) out[0] = 0;
}
int main() {
  int n = 2048; int* d; hipMalloc(&d, 2 * n * sizeof(int));
  hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d, 200000);
  std::vector<int> h(2 * n); hipMemcpy(h.data(), d, 2 * n * sizeof(int), hipMemcpyDeviceToHost);
  for (int i = 0; i < 80; ++i) printf("%d ", h[2 * i]); printf("\n");
  int bad = 0; for (int i = 0; i < n; ++i) if (h[2 * i] != h[2 * (i % 8)]) ++bad;
  printf("blocks whose XCC differs from block (b%%8): %d of %d\n", bad, n);
  auto dec = [&](int b) { int w = h[2*b+1]; printf("b=%d xcc=%d wave=%d simd=%d pipe=%d cu=%d sh=%d se=%d\n", b, h[2*b], w&15, (w>>4)&3, (w>>6)&3, (w>>8)&15, (w>>12)&1, (w>>13)&7); };
  for (int b : {0, 8, 16, 256, 264, 512, 520, 768, 1024, 1032}) dec(b);
  // for every block find the other block with same xcc/se/sh/cu
  int same = 0, diffpar = 0;
  for (int a = 0; a < 512; ++a) for (int b2 = a + 1; b2 < 512; ++b2) {
    int wa = h[2*a+1], wb = h[2*b2+1];
    if (h[2*a] == h[2*b2] && (wa >> 8) == (wb >> 8)) { ++same; if ((wa & 1) != (wb & 1)) ++diffpar; }
  }
  printf("co-resident pairs among first 512 blocks: %d, with different wave-slot parity: %d\n", same, diffpar);
  return 0;
}
