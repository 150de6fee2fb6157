// InstanceNorm2d(affine=False) forward/backward fused with the activation that follows it and the
// ResnetBlock residual add.  HBM-bound: one (n,c) plane per wave64 (small planes, values cached in
// registers: one read) or per 256-thread workgroup (large planes, float4 streams).  Per-plane sums use
// wave64 butterfly shuffles (+ a 4-entry LDS exchange across the waves of a workgroup); the variance is
// the centred second pass (matches torch's biased variance without E[x^2]-mean^2 cancellation).
#include <stdlib.h>

#include <type_traits>

#include "him_common.h"

namespace him {

// block-wide sum for 1024-thread blocks (16 waves); `sh` holds >= 16 floats
__device__ __forceinline__ float block_sum_1024(float v, float* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += sh[i];
  return s;
}

template <int G>
__device__ __forceinline__ float group_sum(float v, float* sh) {
  if constexpr (G == 64) {
    return wave_sum(v);
  } else if constexpr (G == 1024) {
    return block_sum_1024(v, sh);
  } else {
    return block_sum_256(v, sh);
  }
}

__device__ __forceinline__ float act_grad_from_xhat(float xh, int act, float slope) {
  // derivative of act at the normalised value (ReLU / LeakyReLU only follow an IN in this path)
  if (act == HIM_ACT_RELU) return xh > 0.f ? 1.f : 0.f;
  if (act == HIM_ACT_LRELU) return xh > 0.f ? 1.f : slope;
  if (act == HIM_ACT_TANH) {
    const float y = tanhf(xh);
    return 1.f - y * y;
  }
  return 1.f;
}

// G = threads cooperating on one plane (64 or 256); CACHE = elements per thread kept in registers (0: stream)
// G = 1024: the full-resolution planes (131072 elements at 512x256): one 256-thread workgroup per plane left a CU with
// 8 waves of float4 streams -- too little memory parallelism (3.8 TB/s); 16 waves per plane, two planes per CU
// SLABS (him_conv2d_in_act_fwd, few-tile convolutions launched with split-K): the plane's input is not in memory yet --
// it is the sum of `ks` split-K slabs (each shaped like the conv output, `slab` floats apart) + the conv bias, summed in
// gconv_splitk_finish_kernel's order ((s0 + s1) + s2 ... + bias: bit-identical to the separate finish pass), and written
// to `xout` (the raw conv output the backward pass reads) on the way.  The split-K finish launch and its re-read go away.
struct InSlabs {
  const float* part;   // [ks][planes * hw]
  const float* bias;   // [M] or null
  float* xout;         // [planes * hw]
  long long slab;      // floats between slabs
  int ks, M;
};
template <int G, int CACHE, bool SLABS>
__global__ __launch_bounds__(G > 256 ? G : 256) void instnorm_fwd_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ res, float* __restrict__ y,
                                                           float* __restrict__ mean, float* __restrict__ rstd,
                                                           int planes, int hw, float eps, int act, float slope,
                                                           const InSlabs sl) {
  __shared__ float sh[16];
  constexpr int PPB = G >= 256 ? 1 : 256 / G;
  const int plane = blockIdx.x * PPB + (G == 64 ? (threadIdx.x >> 6) : 0);
  if (G == 64 && plane >= planes) return;  // whole wave exits together
  const int tid = G == 64 ? (threadIdx.x & 63) : threadIdx.x;
  // SLABS: xp aliases xo (the apply pass re-reads what the statistics pass stored): no __restrict__ on either
  typedef const float* __restrict__ cfr_t;
  std::conditional_t<SLABS, const float*, cfr_t> xp = (SLABS ? (const float*)sl.xout : x) + (size_t)plane * hw;
  float* __restrict__ yp = y + (size_t)plane * hw;
  const float* __restrict__ rp = res ? res + (size_t)plane * hw : nullptr;
  const float inv_n = 1.f / (float)hw;
  const float* __restrict__ pp = SLABS ? sl.part + (size_t)plane * hw : nullptr;
  float* xo = SLABS ? sl.xout + (size_t)plane * hw : nullptr;
  const float bb = (SLABS && sl.bias) ? sl.bias[plane % sl.M] : 0.f;
  // element idx of the plane from the slabs (SLABS only)
  auto slab1 = [&](int idx) {
    float v = pp[idx];
    for (int z = 1; z < sl.ks; ++z) v += pp[(size_t)z * sl.slab + idx];
    return v + bb;
  };
  auto slab4 = [&](int i4) {
    float4 v = ((const float4*)pp)[i4];
    for (int z = 1; z < sl.ks; ++z) {
      const float4 w = ((const float4*)(pp + (size_t)z * sl.slab))[i4];
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    v.x += bb; v.y += bb; v.z += bb; v.w += bb;
    return v;
  };

  if constexpr (CACHE > 0) {
    float v[CACHE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CACHE; ++i) {
      const int idx = tid + i * G;
      if constexpr (SLABS) {
        v[i] = 0.f;
        if (idx < hw) {
          v[i] = slab1(idx);
          xo[idx] = v[i];
        }
      } else {
        v[i] = idx < hw ? xp[idx] : 0.f;
      }
      s += v[i];
    }
    const float mu = group_sum<G>(s, sh) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < CACHE; ++i) {
      const int idx = tid + i * G;
      const float d = idx < hw ? v[i] - mu : 0.f;
      q += d * d;
    }
    const float var = group_sum<G>(q, sh) * inv_n;
    const float rs = 1.f / sqrtf(var + eps);
    if (tid == 0) {
      mean[plane] = mu;
      rstd[plane] = rs;
    }
#pragma unroll
    for (int i = 0; i < CACHE; ++i) {
      const int idx = tid + i * G;
      if (idx < hw) {
        float o = apply_act((v[i] - mu) * rs, act, slope);
        if (rp) o += rp[idx];
        yp[idx] = o;
      }
    }
  } else {
    const bool vec = (hw & 3) == 0;
    // ONE statistics pass: sums of (x - c) and (x - c)^2 with c = the plane's first element (a value inside the data
    // range, so E[(x-c)^2] - E[x-c]^2 loses at most a few ulps: |mean - c| ~ sigma), then the apply pass.
    const float c0 = SLABS ? slab1(0) : xp[0];
    float s = 0.f, q = 0.f;
    if (vec) {
      const float4* x4 = (const float4*)xp;
      for (int i = tid; i < hw / 4; i += G) {
        float4 a;
        if constexpr (SLABS) {
          a = slab4(i);
          ((float4*)xo)[i] = a;      // the apply pass below re-reads THIS thread's own stores
        } else {
          a = x4[i];
        }
        const float d0 = a.x - c0, d1 = a.y - c0, d2 = a.z - c0, d3 = a.w - c0;
        s += (d0 + d1) + (d2 + d3);
        q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    } else {
      for (int i = tid; i < hw; i += G) {
        float a;
        if constexpr (SLABS) {
          a = slab1(i);
          xo[i] = a;
        } else {
          a = xp[i];
        }
        const float d = a - c0;
        s += d;
        q += d * d;
      }
    }
    const float ms = group_sum<G>(s, sh) * inv_n;
    const float mq = group_sum<G>(q, sh) * inv_n;
    const float mu = c0 + ms;
    const float var = fmaxf(mq - ms * ms, 0.f);
    const float rs = 1.f / sqrtf(var + eps);
    if (tid == 0) {
      mean[plane] = mu;
      rstd[plane] = rs;
    }
    if (vec) {
      const float4* x4 = (const float4*)xp;
      const float4* r4 = (const float4*)rp;
      float4* y4 = (float4*)yp;
      for (int i = tid; i < hw / 4; i += G) {
        const float4 a = x4[i];
        float4 o;
        o.x = apply_act((a.x - mu) * rs, act, slope);
        o.y = apply_act((a.y - mu) * rs, act, slope);
        o.z = apply_act((a.z - mu) * rs, act, slope);
        o.w = apply_act((a.w - mu) * rs, act, slope);
        if (rp) {
          const float4 r = r4[i];
          o.x += r.x;
          o.y += r.y;
          o.z += r.z;
          o.w += r.w;
        }
        y4[i] = o;
      }
    } else {
      for (int i = tid; i < hw; i += G) {
        float o = apply_act((xp[i] - mu) * rs, act, slope);
        if (rp) o += rp[i];
        yp[i] = o;
      }
    }
  }
}

template <int G, int CACHE>
__global__ __launch_bounds__(G > 256 ? G : 256) void instnorm_bwd_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd,
                                                           const float* __restrict__ dy, float* __restrict__ dx,
                                                           int planes, int hw, int act, float slope) {
  __shared__ float sh[16];
  constexpr int PPB = G >= 256 ? 1 : 256 / G;
  const int plane = blockIdx.x * PPB + (G == 64 ? (threadIdx.x >> 6) : 0);
  if (G == 64 && plane >= planes) return;
  const int tid = G == 64 ? (threadIdx.x & 63) : threadIdx.x;
  const float* __restrict__ xp = x + (size_t)plane * hw;
  const float* __restrict__ gp = dy + (size_t)plane * hw;
  float* __restrict__ op = dx + (size_t)plane * hw;
  const float mu = mean[plane], rs = rstd[plane];
  const float inv_n = 1.f / (float)hw;

  if constexpr (CACHE > 0) {
    float xh[CACHE], dz[CACHE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < CACHE; ++i) {
      const int idx = tid + i * G;
      if (idx < hw) {
        xh[i] = (xp[idx] - mu) * rs;
        dz[i] = gp[idx] * act_grad_from_xhat(xh[i], act, slope);
      } else {
        xh[i] = 0.f;
        dz[i] = 0.f;
      }
      s1 += dz[i];
      s2 += dz[i] * xh[i];
    }
    const float m1 = group_sum<G>(s1, sh) * inv_n;
    const float m2 = group_sum<G>(s2, sh) * inv_n;
#pragma unroll
    for (int i = 0; i < CACHE; ++i) {
      const int idx = tid + i * G;
      if (idx < hw) op[idx] = rs * (dz[i] - m1 - xh[i] * m2);
    }
  } else {
    float s1 = 0.f, s2 = 0.f;
    const bool vec = (hw & 3) == 0;
    if (vec) {
      const float4* x4 = (const float4*)xp;
      const float4* g4 = (const float4*)gp;
      for (int i = tid; i < hw / 4; i += G) {
        const float4 a = x4[i], g = g4[i];
        const float h0 = (a.x - mu) * rs, h1 = (a.y - mu) * rs, h2 = (a.z - mu) * rs, h3 = (a.w - mu) * rs;
        const float z0 = g.x * act_grad_from_xhat(h0, act, slope), z1 = g.y * act_grad_from_xhat(h1, act, slope);
        const float z2 = g.z * act_grad_from_xhat(h2, act, slope), z3 = g.w * act_grad_from_xhat(h3, act, slope);
        s1 += (z0 + z1) + (z2 + z3);
        s2 += (z0 * h0 + z1 * h1) + (z2 * h2 + z3 * h3);
      }
    } else {
      for (int i = tid; i < hw; i += G) {
        const float xh = (xp[i] - mu) * rs;
        const float dz = gp[i] * act_grad_from_xhat(xh, act, slope);
        s1 += dz;
        s2 += dz * xh;
      }
    }
    const float m1 = group_sum<G>(s1, sh) * inv_n;
    const float m2 = group_sum<G>(s2, sh) * inv_n;
    if (vec) {
      const float4* x4 = (const float4*)xp;
      const float4* g4 = (const float4*)gp;
      float4* o4 = (float4*)op;
      for (int i = tid; i < hw / 4; i += G) {
        const float4 a = x4[i], g = g4[i];
        const float h0 = (a.x - mu) * rs, h1 = (a.y - mu) * rs, h2 = (a.z - mu) * rs, h3 = (a.w - mu) * rs;
        float4 o;
        o.x = rs * (g.x * act_grad_from_xhat(h0, act, slope) - m1 - h0 * m2);
        o.y = rs * (g.y * act_grad_from_xhat(h1, act, slope) - m1 - h1 * m2);
        o.z = rs * (g.z * act_grad_from_xhat(h2, act, slope) - m1 - h2 * m2);
        o.w = rs * (g.w * act_grad_from_xhat(h3, act, slope) - m1 - h3 * m2);
        o4[i] = o;
      }
    } else {
      for (int i = tid; i < hw; i += G) {
        const float xh = (xp[i] - mu) * rs;
        const float dz = gp[i] * act_grad_from_xhat(xh, act, slope);
        op[i] = rs * (dz - m1 - xh * m2);
      }
    }
  }
}

// InstanceNorm forward straight from the split-K slabs of the convolution in front of it (see InSlabs); same plane-size
// dispatch as him_instnorm_fwd, so the statistics are reduced in the same order as the two-launch path.
int instnorm_fwd_from_slabs(const float* part, long long slab, int ks, const float* bias, int M, float* xout,
                            const float* residual, float* y, float* mean, float* rstd, int planes, int hw, float eps,
                            int act, float slope, hipStream_t st) {
  if (planes <= 0 || hw <= 0 || ks < 1 || !part || !xout) return fail(HIM_E_INVALID, "instnorm (slabs): planes=%d hw=%d ks=%d", planes, hw, ks);
  InSlabs sl;
  sl.part = part; sl.bias = bias; sl.xout = xout; sl.slab = slab; sl.ks = ks; sl.M = M;
  if (hw <= 512) {
    hipLaunchKernelGGL((instnorm_fwd_kernel<64, 8, true>), dim3(cdiv(planes, 4)), dim3(256), 0, st, (const float*)nullptr,
                       residual, y, mean, rstd, planes, hw, eps, act, slope, sl);
  } else if (hw <= 4096) {
    hipLaunchKernelGGL((instnorm_fwd_kernel<256, 16, true>), dim3(planes), dim3(256), 0, st, (const float*)nullptr, residual,
                       y, mean, rstd, planes, hw, eps, act, slope, sl);
  } else if (hw < 32768) {
    hipLaunchKernelGGL((instnorm_fwd_kernel<256, 0, true>), dim3(planes), dim3(256), 0, st, (const float*)nullptr, residual, y,
                       mean, rstd, planes, hw, eps, act, slope, sl);
  } else {
    hipLaunchKernelGGL((instnorm_fwd_kernel<1024, 0, true>), dim3(planes), dim3(1024), 0, st, (const float*)nullptr, residual,
                       y, mean, rstd, planes, hw, eps, act, slope, sl);
  }
  return check_launch("instnorm_fwd (slabs)");
}

}  // namespace him

using namespace him;

extern "C" {

int him_instnorm_fwd(const float* x, const float* residual, float* y, float* mean, float* rstd, int planes,
                     int hw, float eps, int act, float slope, void* stream) {
  if (planes <= 0 || hw <= 0) return fail(HIM_E_INVALID, "instnorm: planes=%d hw=%d", planes, hw);
  hipStream_t st = (hipStream_t)stream;
  InSlabs sl;
  sl.part = nullptr; sl.bias = nullptr; sl.xout = nullptr; sl.slab = 0; sl.ks = 0; sl.M = 1;
  if (hw <= 512) {
    hipLaunchKernelGGL((instnorm_fwd_kernel<64, 8, false>), dim3(cdiv(planes, 4)), dim3(256), 0, st, x, residual, y,
                       mean, rstd, planes, hw, eps, act, slope, sl);
  } else if (hw <= 4096) {
    hipLaunchKernelGGL((instnorm_fwd_kernel<256, 16, false>), dim3(planes), dim3(256), 0, st, x, residual, y, mean,
                       rstd, planes, hw, eps, act, slope, sl);
  } else if (hw < 32768) {
    hipLaunchKernelGGL((instnorm_fwd_kernel<256, 0, false>), dim3(planes), dim3(256), 0, st, x, residual, y, mean, rstd,
                       planes, hw, eps, act, slope, sl);
  } else {
    hipLaunchKernelGGL((instnorm_fwd_kernel<1024, 0, false>), dim3(planes), dim3(1024), 0, st, x, residual, y, mean, rstd,
                       planes, hw, eps, act, slope, sl);
  }
  return check_launch("instnorm_fwd");
}

int him_instnorm_bwd(const float* x, const float* mean, const float* rstd, const float* dy, float* dx, int planes,
                     int hw, int act, float slope, void* stream) {
  if (planes <= 0 || hw <= 0) return fail(HIM_E_INVALID, "instnorm: planes=%d hw=%d", planes, hw);
  hipStream_t st = (hipStream_t)stream;
  if (hw <= 512) {
    hipLaunchKernelGGL((instnorm_bwd_kernel<64, 8>), dim3(cdiv(planes, 4)), dim3(256), 0, st, x, mean, rstd, dy,
                       dx, planes, hw, act, slope);
  } else if (hw <= 4096) {
    hipLaunchKernelGGL((instnorm_bwd_kernel<256, 16>), dim3(planes), dim3(256), 0, st, x, mean, rstd, dy, dx,
                       planes, hw, act, slope);
  } else if (hw < 32768) {
    hipLaunchKernelGGL((instnorm_bwd_kernel<256, 0>), dim3(planes), dim3(256), 0, st, x, mean, rstd, dy, dx,
                       planes, hw, act, slope);
  } else {
    hipLaunchKernelGGL((instnorm_bwd_kernel<1024, 0>), dim3(planes), dim3(1024), 0, st, x, mean, rstd, dy, dx,
                       planes, hw, act, slope);
  }
  return check_launch("instnorm_bwd");
}

}  // extern "C"
