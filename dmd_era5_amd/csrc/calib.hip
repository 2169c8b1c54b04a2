// Measurement aid ("confirm the peak with an on-box micro-benchmark"): a
// register-only fp32 MFMA loop.  Not on the hot path; bench.py --calibrate times it to report
// the fp32 matrix rate this particular GPU sustains next to the nominal 157.3 TFLOP/s.
#include "dmdx_common.h"

namespace {

__global__ __launch_bounds__(256, 2) void calib_mfma_kernel(int iters, float* sink) {
  f32x16 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  float a = 1.0f + 1e-3f * (float)(threadIdx.x & 63), b = 0.5f - 1e-3f * (float)(threadIdx.x & 31);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
    asm volatile("" : "+v"(a), "+v"(b));  // keep the operands opaque, the loop un-hoisted
  }
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  if (s == 12345.678f) sink[0] = s;  // never true; keeps the accumulators live
}

}  // namespace

/* Launches 2 workgroups of 4 waves per CU (8 waves per CU = 2 per SIMD, like K1); every wave
 * issues 16 * iters MFMAs of 4096 flops.  *flops_out (host) = total flops of the launch. */
extern "C" int dmdx_calib_mfma_f32(int iters, int num_cus, float* sink, double* flops_out, void* stream) {
  DMDX_CHECK_ARG(iters >= 1 && num_cus >= 1 && sink, "calib_mfma: bad arguments");
  const int wgs = 2 * num_cus;
  hipLaunchKernelGGL(calib_mfma_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, iters, sink);
  DMDX_LAUNCH_CHECK();
  if (flops_out) *flops_out = (double)wgs * 4.0 * 16.0 * (double)iters * 4096.0;
  return 0;
}
