// Scratch probe (not part of libdmdx): read bandwidth against the size of the buffer that is read
// over and over -- where do the L2s (8 x 4 MB), the Infinity Cache (256 MB) and HBM show?
//   hipcc --offload-arch=gfx950 -O2 -o scripts/probe_stream scripts/probe_stream.hip
// Also: two sweeps that lag each other by `lag` bytes (the second read of a fused range-finder pass
// would follow the first at such a distance).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// every workgroup reads contiguous 16 KB pieces, piece index strided by the grid
__global__ __launch_bounds__(256) void stream_kernel(const f32x4* __restrict__ x, size_t npieces, int reps, float* sink) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < reps; ++r)
    for (size_t pc = blockIdx.x; pc < npieces; pc += gridDim.x) {
      const f32x4* q = x + pc * 1024 + threadIdx.x;
#pragma unroll
      for (int i = 0; i < 4; ++i) acc += __builtin_nontemporal_load(q + 256 * i);
    }
  sink[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

// the same, but every piece is read a second time `lag` pieces after its first read
template <bool NT>
__global__ __launch_bounds__(256) void lagged_kernel(const f32x4* __restrict__ x, size_t npieces, size_t lag, float* sink) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t pc = blockIdx.x; pc < npieces + lag; pc += gridDim.x) {
    if (pc < npieces) {
      const f32x4* q = x + pc * 1024 + threadIdx.x;
#pragma unroll
      for (int i = 0; i < 4; ++i) acc += NT ? __builtin_nontemporal_load(q + 256 * i) : q[256 * i];
    }
    if (pc >= lag) {
      const f32x4* q = x + (pc - lag) * 1024 + threadIdx.x;
#pragma unroll
      for (int i = 0; i < 4; ++i) acc += NT ? __builtin_nontemporal_load(q + 256 * i) : q[256 * i];
    }
  }
  sink[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
  const size_t maxbytes = size_t(8) << 30;
  f32x4* x; float* sink;
  CHECK(hipMalloc(&x, maxbytes));
  CHECK(hipMemset(x, 0, maxbytes));
  const int grid = 2048;
  CHECK(hipMalloc(&sink, grid * 256 * sizeof(float)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const size_t sizes_mb[] = {16, 64, 128, 8192};
  for (size_t mb : sizes_mb) {
    const size_t bytes = mb << 20, npieces = bytes / 16384;
    const int reps = (int)((size_t(16) << 30) / bytes) + 2;
    hipLaunchKernelGGL(stream_kernel, dim3(grid), dim3(256), 0, 0, x, npieces, 2, sink);   // warm
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(stream_kernel, dim3(grid), dim3(256), 0, 0, x, npieces, reps, sink);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("re-read %5zu MB x %4d : %.2f TB/s\n", mb, reps, (double)bytes * reps / ms / 1e9);
    fflush(stdout);
  }
  const size_t lags_mb[] = {1, 2, 4, 8, 16, 32, 64, 128, 256, 2048};
  for (int nt = 0; nt < 2; ++nt)
  for (size_t lag_mb : lags_mb) {
    const size_t bytes = maxbytes, npieces = bytes / 16384, lag = (lag_mb << 20) / 16384;
    CHECK(hipEventRecord(e0, 0));
    if (nt) hipLaunchKernelGGL(lagged_kernel<true>, dim3(grid), dim3(256), 0, 0, x, npieces, lag, sink);
    else hipLaunchKernelGGL(lagged_kernel<false>, dim3(grid), dim3(256), 0, 0, x, npieces, lag, sink);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s 8 GB read twice, second read %5zu MB behind the first: %.3f ms = %.2f TB/s of loads, %.2f TB/s of distinct bytes\n",
           nt ? "nt loads:   " : "plain loads:", lag_mb, ms, 2.0 * bytes / ms / 1e9, (double)bytes / ms / 1e9);
    fflush(stdout);
  }
  return 0;
}
