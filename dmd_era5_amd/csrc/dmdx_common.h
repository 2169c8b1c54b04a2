// Shared declarations for the libdmdx.so translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dmdx.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void dmdx_set_error(const char* fmt, ...);

// measurement aid (dmdx_set_clock_probe): 3 device uint64 the stamped kernels (K1 / K3 batch
// launches, K2) add their core-clock cycles, reference ticks and workgroup count to; null = off
extern thread_local unsigned long long* dmdx_clock_probe_ptr;   // (per thread: main() primes the libraries on a side thread)

// compute units of the current device (cached per device; <= 0 if the query fails)
int dmdx_device_cus();

#define DMDX_CHECK_ARG(cond, ...)        \
  do {                                   \
    if (!(cond)) {                       \
      dmdx_set_error(__VA_ARGS__);       \
      return DMDX_E_INVALID;             \
    }                                    \
  } while (0)

#define DMDX_HIP(expr)                                                        \
  do {                                                                        \
    hipError_t e_ = (expr);                                                   \
    if (e_ != hipSuccess) {                                                   \
      dmdx_set_error("%s failed: %s", #expr, hipGetErrorString(e_));          \
      return -(int)e_;                                                        \
    }                                                                         \
  } while (0)

#define DMDX_LAUNCH_CHECK()                                                   \
  do {                                                                        \
    hipError_t e_ = hipGetLastError();                                        \
    if (e_ != hipSuccess) {                                                   \
      dmdx_set_error("kernel launch failed: %s", hipGetErrorString(e_));      \
      return -(int)e_;                                                        \
    }                                                                         \
  } while (0)

static inline bool dmdx_aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

// Agent-scope (sc1: L1-bypassing, coherent across the XCDs' L2s) loads of data another workgroup stored
// with sc1, as plain buffer loads: the compiler keeps any number of them in flight, where every
// `__hip_atomic_load(..., AGENT)` is followed by its own `s_waitcnt vmcnt(0)` (16 column loads of a
// Jacobi step = 16 round trips of ~1 us).  Ordering against the producer is the caller's barrier.
// `rs` = dmdx_sc1_rsrc(base), offsets in bytes (< 2^31).
#ifdef __HIPCC__
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dmdx_sc1_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
}
__device__ __forceinline__ double dmdx_ld_sc1_f64(__amdgpu_buffer_rsrc_t rs, int byte_off) {
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(double, (i32x2)__builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, 16 /* sc1 */));
}
__device__ __forceinline__ float dmdx_ld_sc1_f32(__amdgpu_buffer_rsrc_t rs, int byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, 16 /* sc1 */));
}
#endif

// K2, 16x16x4 body (skinny16.hip): l <= 256 columns in one pass, 16-column granular, optional fused Gram
bool dmdx_skinny16_shape_ok(int64_t m, int64_t ldx);
int dmdx_skinny16_launch(const float* X, int64_t m, int64_t n, int64_t ldx, const float* W, int64_t ldw, int l, float* Y,
                         int64_t ldy, hipStream_t st);
size_t dmdx_skinny16_gram_ws(int64_t m, int64_t l);
int dmdx_skinny16_gram_launch(const float* X, int64_t m, int64_t n, int64_t ldx, const float* W, int64_t ldw, int l,
                              float* Y, int64_t ldy, double* G, int64_t ldg, int accumulate, float* gpart,
                              hipStream_t st);
