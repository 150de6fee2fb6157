// Shared host/device helpers for libhim_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/him.h"

namespace him {

// ---- error plumbing -------------------------------------------------------------------------
char* err_buf();  // thread-local, 512 bytes
inline int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
}  // namespace him

#include <stdarg.h>
namespace him {
inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}
inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(HIM_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return HIM_OK;
}

// ---- exact division of a small dividend by a small constant: one v_mul_hi_u32 -----------------
// q = floor(n/d) for all n with n*d < 2^32 (callers guarantee n < 2^20, d < 2^12).
struct FastDiv {
  uint32_t d, m;
};
inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  f.m = d <= 1 ? 0u : (uint32_t)((1ull << 32) / d + 1ull);
  return f;
}
// branch-free: m == 0 encodes d == 1
__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv f) { return __umulhi(n, f.m) + n * (uint32_t)(f.m == 0); }

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case HIM_ACT_RELU: return v > 0.f ? v : 0.f;
    case HIM_ACT_LRELU: return v > 0.f ? v : v * slope;
    case HIM_ACT_TANH: return tanhf(v);
    case HIM_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum for 256-thread blocks; result valid in every thread.  `sh` holds >= 8 floats.
__device__ __forceinline__ float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace him
