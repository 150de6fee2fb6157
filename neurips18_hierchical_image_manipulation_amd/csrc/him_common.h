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

// compute units of the current device = what a persistent launch sizes its grid by (a query, not cached: no mutable state)
inline int device_cus() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return 256;
  return n > 0 ? n : 256;
}

// ---- LDS-DMA: 16 bytes per lane, global -> LDS, no VGPR staging (global_load_lds_dwordx4) ---------------------------
// The 64 lanes of the wave fill 1 KB of LDS at the WAVE-UNIFORM byte address `lds_dst` in lane order; `gsrc` is per lane.
// Issued through inline asm ON PURPOSE: hipcc (ROCm 7.2) tracks a __builtin_amdgcn_global_load_lds as a pending LDS store
// and puts `s_waitcnt vmcnt(0)` in front of the next ds_read of the same array -- i.e. it drains a multi-stage DMA
// pipeline at the top of every K-step (seen in the ISA of round 2/3's fused Winograd kernel: the request for chunk c+2 was
// waited for before chunk c's first operand read).  An asm statement is invisible to that bookkeeping: completion is
// counted by hand (`s_waitcnt vmcnt(N)` + barrier before any wave reads the data; cdna_hip_programming.md 5.7).
// M0 (the DMA's LDS base) is saved and restored inside the statement.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
// four of them with ONE M0 setup: instruction q moves 1 KB from gsrc + 1024 q (per lane) to lds_dst + 1024 q (the
// instruction offset applies to the global AND the LDS address)
__device__ __forceinline__ void glds16x4(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "global_load_lds_dwordx4 %1, off offset:1024\n\t"
      "global_load_lds_dwordx4 %1, off offset:2048\n\t"
      "global_load_lds_dwordx4 %1, off offset:3072\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
// wave-uniform LDS byte address of a __shared__ pointer
__device__ __forceinline__ unsigned lds_addr_u(const void* p) {
  return (unsigned)__builtin_amdgcn_readfirstlane(
      (int)(unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p);
}

// ---- HimAlgo accessors: a zero field selects the default (include/him.h "Algorithm selection").  Kernel selection is a
// pure function of (descriptor, HimAlgo): no statics, no environment.
// round 4: 256 (was 512) -- with the LDS-DMA GEMM kernel (64 KB LDS, shares a CU) the separate-transform pipeline beats the
// one-workgroup-per-CU fused kernel from 256 channels inside a multi-stream step (box2mask 477 -> 486 images/s, C4 neutral)
inline int algo_wino_min_c(const HimAlgo& a) { return a.wino_min_c == 0 ? 256 : a.wino_min_c; }          // <= 0: off
inline int algo_wino_fused_min_c(const HimAlgo& a) { return a.wino_fused_min_c == 0 ? 64 : a.wino_fused_min_c; }
inline int algo_wino_fused_max_c(const HimAlgo& a) { return a.wino_fused_max_c == 0 ? 255 : a.wino_fused_max_c; }
inline int algo_wino4_min_c(const HimAlgo& a) { return a.wino4_min_c == 0 ? 128 : a.wino4_min_c; }
inline int algo_ksplit_max(const HimAlgo& a) { return a.ksplit_max <= 0 ? 4 : a.ksplit_max; }
inline int algo_wino_fused_chunk(const HimAlgo& a) { return a.wino_fused_chunk == 4 ? 4 : 8; }
inline int algo_tblock(const HimAlgo& a) { return (a.wino_tblock == 128 || a.wino_tblock == 256) ? a.wino_tblock : 64; }
inline bool algo_off(const HimAlgo& a, unsigned bit) { return (a.disable & bit) != 0; }

// him_norm.hip: InstanceNorm forward reading the split-K slabs of the convolution in front of it (him_conv2d_in_act_fwd)
int instnorm_fwd_from_slabs(const float* part, long long slab, int ks, const float* bias, int M, float* xout,
                            const float* residual, float* y, float* mean, float* rstd, int planes, int hw, float eps,
                            int act, float slope, hipStream_t st);
}  // namespace him
