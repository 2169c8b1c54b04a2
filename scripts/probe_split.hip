// Scratch probe (not part of libdmdx): can the VALU split an fp32 fragment into three bf16 planes fast enough
// to feed v_mfma_f32_32x32x16_bf16 (6 products per fp32 product: a0b0 + a0b1 + a1b0 + a0b2 + a1b1 + a2b0)?
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probe_split scripts/probe_split.hip
// Per 16-k step of a 64 x 64 wave tile: 32 fp32 values per lane from LDS (8 ds_read_b128), split (and / sub /
// perm), 24 bf16 MFMAs.  Variants: MFMA only, split only, both; 1 or 2 waves per SIMD.  The fp32 path this would
// replace needs 2048 MFMA cycles for the same step (32 x v_mfma_f32_32x32x2_f32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 8 fp32 -> three planes of 8 bf16 (truncation splits: x = p0 + p1 + p2 + O(2^-24 x))
__device__ __forceinline__ void split8(const f32x4 lo, const f32x4 hi, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
  float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  unsigned q0[4], q1[4], q2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned a = __builtin_bit_cast(unsigned, x[2 * i]), b = __builtin_bit_cast(unsigned, x[2 * i + 1]);
    q0[i] = __builtin_amdgcn_perm(b, a, 0x07060302u);          // high halves of (a, b)
    const float ra = x[2 * i] - __builtin_bit_cast(float, a & 0xFFFF0000u);
    const float rb = x[2 * i + 1] - __builtin_bit_cast(float, b & 0xFFFF0000u);
    const unsigned ua = __builtin_bit_cast(unsigned, ra), ub = __builtin_bit_cast(unsigned, rb);
    q1[i] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
    const float sa = ra - __builtin_bit_cast(float, ua & 0xFFFF0000u);
    const float sb = rb - __builtin_bit_cast(float, ub & 0xFFFF0000u);
    q2[i] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, sb), __builtin_bit_cast(unsigned, sa), 0x07060302u);
  }
  p0 = __builtin_bit_cast(bf16x8, (u32x4){q0[0], q0[1], q0[2], q0[3]});
  p1 = __builtin_bit_cast(bf16x8, (u32x4){q1[0], q1[1], q1[2], q1[3]});
  p2 = __builtin_bit_cast(bf16x8, (u32x4){q2[0], q2[1], q2[2], q2[3]});
}

template <int MODE>   // 1 = MFMA only, 2 = split only, 3 = both
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ src, float* out, int iters, long long* cycles) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  bf16x8 A[2][3], B[2][3];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int p = 0; p < 3; ++p) { A[h][p] = (bf16x8)(0); B[h][p] = (bf16x8)(0); }
  unsigned sink = 0;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    const float* base = lds + ((it & 7) * 1024) + lane * 8;
    f32x4 v[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = *reinterpret_cast<const f32x4*>(base + 512 * (i & 1) + 4 * 0 + 256 * (i >> 1) * 0 + 0 + 0 * i);
      v[2 * i + 1] = *reinterpret_cast<const f32x4*>(base + 512 * (i & 1) + 4);
    }
    if (MODE & 2) {
      split8(v[0], v[1], A[0][0], A[0][1], A[0][2]);
      split8(v[2], v[3], A[1][0], A[1][1], A[1][2]);
      split8(v[4], v[5], B[0][0], B[0][1], B[0][2]);
      split8(v[6], v[7], B[1][0], B[1][1], B[1][2]);
      if (!(MODE & 1)) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            const u32x4 a = __builtin_bit_cast(u32x4, A[h][p]), b = __builtin_bit_cast(u32x4, B[h][p]);
            sink ^= a[0] ^ a[1] ^ a[2] ^ a[3] ^ b[0] ^ b[1] ^ b[2] ^ b[3];
          }
      }
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int p = 0; p < 3; ++p) {   // (no split: the raw bits stand in)
          A[h][p] = __builtin_bit_cast(bf16x8, (u32x4){__builtin_bit_cast(unsigned, v[2 * h][p]), 1u, 2u, 3u});
          B[h][p] = __builtin_bit_cast(bf16x8, (u32x4){__builtin_bit_cast(unsigned, v[4 + 2 * h][p]), 1u, 2u, 3u});
        }
    }
    if (MODE & 1) {
#pragma unroll
      for (int ha = 0; ha < 2; ++ha)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          f32x16 c = acc[2 * ha + hb];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ha][0], B[hb][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ha][0], B[hb][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ha][1], B[hb][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ha][0], B[hb][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ha][1], B[hb][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ha][2], B[hb][0], c, 0, 0, 0);
          acc[2 * ha + hb] = c;
        }
    }
  }
  const long long t1 = clock64();
  float s = (float)sink;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[b][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
  float *src, *out; long long* cyc;
  const int grid = 512;
  CHECK(hipMalloc(&src, 8192 * 4)); CHECK(hipMalloc(&out, grid * 256 * 4)); CHECK(hipMalloc(&cyc, grid * 8));
  float h[8192];
  for (int i = 0; i < 8192; ++i) h[i] = 1.0f + 1e-3f * (float)(i % 977);
  CHECK(hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice));
  const int iters = 20000;
  for (int wgs = 1; wgs <= 2; ++wgs)
    for (int mode = 1; mode <= 3; ++mode) {
      const int g = 256 * wgs;
      hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        if (mode == 1) hipLaunchKernelGGL(split_kernel<1>, dim3(g), dim3(256), 0, 0, src, out, iters, cyc);
        if (mode == 2) hipLaunchKernelGGL(split_kernel<2>, dim3(g), dim3(256), 0, 0, src, out, iters, cyc);
        if (mode == 3) hipLaunchKernelGGL(split_kernel<3>, dim3(g), dim3(256), 0, 0, src, out, iters, cyc);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
      }
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      long long c0; CHECK(hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost));
      // per step and wave: 64 x 64 x 16 fp32-equivalent MACs; chip-wide rate in fp32-equivalent TFLOP/s
      const double flops = 2.0 * 64 * 64 * 16 * (double)iters * 4 * g;
      printf("%d workgroup(s) per CU, %s: %.1f us per 1000 steps, clock64 %.0f ticks per step; fp32-equivalent %.0f TFLOP/s (fp32 MFMA peak 157)\n",
             wgs, mode == 1 ? "24 MFMA only " : mode == 2 ? "split only   " : "split + MFMA ", 1e3 * ms / iters * 1000, (double)c0 / iters,
             mode == 2 ? 0.0 : flops / (ms * 1e-3) / 1e12);
    }
  return 0;
}
