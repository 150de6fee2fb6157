// Loss reductions (HBM-bound): mean |a-b| and mean (x-t)^2, deterministic two-stage sums, and their
// gradients scaled by a DEVICE scalar upstream gradient (no host sync anywhere in the loss graph).
#include <algorithm>

#include "him_common.h"

namespace him {

constexpr int RED_BLOCKS = 1024;

template <int MODE>  // 0: |a-b|, 1: (a-t)^2
__global__ __launch_bounds__(256) void reduce_stage1(const float* __restrict__ a, const float* __restrict__ b,
                                                     size_t n, float target, float* __restrict__ partial) {
  __shared__ float sh[8];
  float s = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  if ((n & 3) == 0 && (((uintptr_t)a | (uintptr_t)(MODE == 0 ? b : a)) & 15) == 0) {
    const float4* a4 = (const float4*)a;
    const float4* b4 = (const float4*)b;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n / 4; i += stride) {
      const float4 x = a4[i];
      if (MODE == 0) {
        const float4 y = b4[i];
        s += (fabsf(x.x - y.x) + fabsf(x.y - y.y)) + (fabsf(x.z - y.z) + fabsf(x.w - y.w));
      } else {
        const float d0 = x.x - target, d1 = x.y - target, d2 = x.z - target, d3 = x.w - target;
        s += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
      if (MODE == 0) {
        s += fabsf(a[i] - b[i]);
      } else {
        const float d = a[i] - target;
        s += d * d;
      }
    }
  }
  s = block_sum_256(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void reduce_stage2(const float* __restrict__ partial, int nb, float inv_n,
                                                     float* __restrict__ out) {
  __shared__ float sh[8];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
  s = block_sum_256(s, sh);
  if (threadIdx.x == 0) out[0] = s * inv_n;
}

__global__ void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n,
                              const float* __restrict__ g, float inv_n, float* __restrict__ da, int accumulate) {
  const float gs = g[0] * inv_n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = a[i] - b[i];
    float v = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);
    if ((accumulate & 2) && !(a[i] > 0.f)) v = 0.f;   // a is a ReLU output: its activation backward folded in
    da[i] = (accumulate & 1) ? da[i] + v : v;
  }
}
__global__ void mse_bwd_kernel(const float* __restrict__ x, size_t n, float target, const float* __restrict__ g,
                               float inv_n, float* __restrict__ dx, int accumulate) {
  const float gs = g[0] * 2.f * inv_n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = (x[i] - target) * gs;
    dx[i] = accumulate ? dx[i] + v : v;
  }
}

static int nblocks(size_t n) {
  size_t b = (n + 1023) / 1024;
  if (b > RED_BLOCKS) b = RED_BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace him

using namespace him;
#define ST ((hipStream_t)stream)

extern "C" {

size_t him_reduce_ws(size_t n) {
  (void)n;
  return RED_BLOCKS * sizeof(float);
}

int him_l1_mean_fwd(const float* a, const float* b, size_t n, float* out, void* ws, size_t ws_bytes, void* stream) {
  if (!n) return fail(HIM_E_INVALID, "l1: empty input");
  if (!ws || ws_bytes < him_reduce_ws(n)) return fail(HIM_E_WORKSPACE, "l1: ws too small");
  const int nb = nblocks(n);
  hipLaunchKernelGGL((reduce_stage1<0>), dim3(nb), dim3(256), 0, ST, a, b, n, 0.f, (float*)ws);
  hipLaunchKernelGGL(reduce_stage2, dim3(1), dim3(256), 0, ST, (const float*)ws, nb, (float)(1.0 / (double)n), out);
  return check_launch("l1_mean_fwd");
}
int him_l1_mean_bwd(const float* a, const float* b, size_t n, const float* g, float* da, int accumulate,
                    void* stream) {
  if (!n) return HIM_OK;
  const int nb = (int)std::min<size_t>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(nb), dim3(256), 0, ST, a, b, n, g, (float)(1.0 / (double)n), da, accumulate);
  return check_launch("l1_mean_bwd");
}
int him_mse_const_fwd(const float* x, size_t n, float target, float* out, void* ws, size_t ws_bytes, void* stream) {
  if (!n) return fail(HIM_E_INVALID, "mse: empty input");
  if (!ws || ws_bytes < him_reduce_ws(n)) return fail(HIM_E_WORKSPACE, "mse: ws too small");
  const int nb = nblocks(n);
  hipLaunchKernelGGL((reduce_stage1<1>), dim3(nb), dim3(256), 0, ST, x, x, n, target, (float*)ws);
  hipLaunchKernelGGL(reduce_stage2, dim3(1), dim3(256), 0, ST, (const float*)ws, nb, (float)(1.0 / (double)n), out);
  return check_launch("mse_const_fwd");
}
int him_mse_const_bwd(const float* x, size_t n, float target, const float* g, float* dx, int accumulate,
                      void* stream) {
  if (!n) return HIM_OK;
  const int nb = (int)std::min<size_t>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(nb), dim3(256), 0, ST, x, n, target, g, (float)(1.0 / (double)n), dx,
                     accumulate);
  return check_launch("mse_const_bwd");
}

}  // extern "C"
