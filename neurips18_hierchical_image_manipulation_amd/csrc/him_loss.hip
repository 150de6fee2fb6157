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

// ---- several L1 terms in one launch (feature matching: 12 pairs, VGG: 5): the per-pair work split, partial sums and
// summation order are those of reduce_stage1<0> / reduce_stage2 / l1_bwd_kernel -> bit-identical results, 3 launches
// instead of 3 per pair.
constexpr int L1_MAX = 16;
struct L1Multi {
  const float* a[L1_MAX];
  const float* b[L1_MAX];
  float* da[L1_MAX];
  unsigned long long n[L1_MAX];
  float inv_n[L1_MAX];
  int blk0[L1_MAX + 1];   // first block of each pair (forward or backward grid)
  int np;
};
__device__ __forceinline__ int l1_pair_of_block(const L1Multi& p, int blk) {
  int q = 0;
  while (q + 1 < p.np && blk >= p.blk0[q + 1]) ++q;
  return q;
}
__global__ __launch_bounds__(256) void l1_multi_stage1(const L1Multi p, float* __restrict__ partial) {
  __shared__ float sh[8];
  const int q = l1_pair_of_block(p, blockIdx.x);
  const int lb = blockIdx.x - p.blk0[q], nbq = p.blk0[q + 1] - p.blk0[q];
  const float* __restrict__ a = p.a[q];
  const float* __restrict__ b = p.b[q];
  const size_t n = p.n[q];
  float s = 0.f;
  const size_t stride = (size_t)nbq * 256;
  if ((n & 3) == 0 && (((uintptr_t)a | (uintptr_t)b) & 15) == 0) {
    const float4* a4 = (const float4*)a;
    const float4* b4 = (const float4*)b;
    for (size_t i = (size_t)lb * 256 + threadIdx.x; i < n / 4; i += stride) {
      const float4 x = a4[i];
      const float4 y = b4[i];
      s += (fabsf(x.x - y.x) + fabsf(x.y - y.y)) + (fabsf(x.z - y.z) + fabsf(x.w - y.w));
    }
  } else {
    for (size_t i = (size_t)lb * 256 + threadIdx.x; i < n; i += stride) s += fabsf(a[i] - b[i]);
  }
  s = block_sum_256(s, sh);
  if (threadIdx.x == 0) partial[(size_t)q * RED_BLOCKS + lb] = s;
}
__global__ __launch_bounds__(256) void l1_multi_stage2(const L1Multi p, const float* __restrict__ partial,
                                                       float* __restrict__ out) {
  __shared__ float sh[8];
  const int q = blockIdx.x, nbq = p.blk0[q + 1] - p.blk0[q];
  float s = 0.f;
  for (int i = threadIdx.x; i < nbq; i += 256) s += partial[(size_t)q * RED_BLOCKS + i];
  s = block_sum_256(s, sh);
  if (threadIdx.x == 0) out[q] = s * p.inv_n[q];
}
__global__ void l1_multi_bwd_kernel(const L1Multi p, const float* __restrict__ g, int accumulate) {
  const int q = l1_pair_of_block(p, blockIdx.x);
  const int lb = blockIdx.x - p.blk0[q], nbq = p.blk0[q + 1] - p.blk0[q];
  const float* __restrict__ a = p.a[q];
  const float* __restrict__ b = p.b[q];
  float* __restrict__ da = p.da[q];
  const size_t n = p.n[q];
  const float gs = g[q] * p.inv_n[q];
  for (size_t i = (size_t)lb * blockDim.x + threadIdx.x; i < n; i += (size_t)nbq * blockDim.x) {
    const float d = a[i] - b[i];
    float v = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);
    if ((accumulate & 2) && !(a[i] > 0.f)) v = 0.f;
    da[i] = (accumulate & 1) ? da[i] + v : v;
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


struct LinComb {
  const float* t[8];
  float* d[8];
  float w[8];
  int n;
  float scale;
};
__global__ void lincomb_fwd_kernel(const LinComb p, float* __restrict__ out) {
  // every product and every sum rounded on its own, as the reference's chain of one-element torch ops: hipcc's default
  // -ffp-contract=fast fused __fadd_rn(acc, __fmul_rn(w, t)) into v_fma_f32 all the same (round 5: a 9-term sum came out one
  // ulp off the fp32 chain, tests/test_ops_gpu.py::test_scalar_loss_arithmetic_...; `#pragma clang fp contract(off)` is
  // ignored under that mode) -- the product passes through an empty asm, which the optimizer cannot look through
  if (threadIdx.x) return;
  float acc = 0.f;
  for (int i = 0; i < p.n; ++i) {
    float prod = __fmul_rn(p.w[i], p.t[i][0]);
    asm volatile("" : "+v"(prod));
    acc = __fadd_rn(acc, prod);
  }
  out[0] = __fmul_rn(acc, p.scale);
}
__global__ void lincomb_bwd_kernel(const LinComb p, const float* __restrict__ g) {
  const int i = threadIdx.x;
  if (i < p.n && p.d[i]) p.d[i][0] = __fmul_rn(__fmul_rn(g[0], p.scale), p.w[i]);
}

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
size_t him_l1_multi_ws(int npairs) { return (size_t)(npairs > 0 ? npairs : 0) * RED_BLOCKS * sizeof(float); }

int him_l1_multi_fwd(const float* const* a, const float* const* b, const size_t* n, int npairs, float* out, void* ws,
                     size_t ws_bytes, void* stream) {
  if (npairs <= 0) return HIM_OK;
  if (!a || !b || !n || !out) return fail(HIM_E_INVALID, "l1 multi: null argument");
  if (!ws || ws_bytes < him_l1_multi_ws(npairs)) return fail(HIM_E_WORKSPACE, "l1 multi: ws too small");
  for (int c0 = 0; c0 < npairs; c0 += L1_MAX) {
    L1Multi p;
    memset(&p, 0, sizeof(p));
    p.np = std::min(L1_MAX, npairs - c0);
    int blk = 0;
    for (int i = 0; i < p.np; ++i) {
      if (!n[c0 + i]) return fail(HIM_E_INVALID, "l1: empty input");
      p.a[i] = a[c0 + i];
      p.b[i] = b[c0 + i];
      p.n[i] = n[c0 + i];
      p.inv_n[i] = (float)(1.0 / (double)n[c0 + i]);
      p.blk0[i] = blk;
      blk += nblocks(n[c0 + i]);
    }
    p.blk0[p.np] = blk;
    float* part = (float*)ws + (size_t)c0 * RED_BLOCKS;
    hipLaunchKernelGGL(l1_multi_stage1, dim3(blk), dim3(256), 0, ST, p, part);
    hipLaunchKernelGGL(l1_multi_stage2, dim3(p.np), dim3(256), 0, ST, p, (const float*)part, out + c0);
  }
  return check_launch("l1_multi_fwd");
}

int him_l1_multi_bwd(const float* const* a, const float* const* b, const size_t* n, int npairs, const float* g,
                     float* const* da, int accumulate, void* stream) {
  if (npairs <= 0) return HIM_OK;
  if (!a || !b || !n || !g || !da) return fail(HIM_E_INVALID, "l1 multi bwd: null argument");
  for (int c0 = 0; c0 < npairs; c0 += L1_MAX) {
    L1Multi p;
    memset(&p, 0, sizeof(p));
    const int cnt = std::min(L1_MAX, npairs - c0);
    int blk = 0, np = 0;
    // pairs without a gradient tensor are skipped; g is indexed by the position inside this chunk, so keep empty slots
    for (int i = 0; i < cnt; ++i) {
      p.a[i] = a[c0 + i];
      p.b[i] = b[c0 + i];
      p.da[i] = da[c0 + i];
      p.n[i] = da[c0 + i] ? n[c0 + i] : 0;
      p.inv_n[i] = n[c0 + i] ? (float)(1.0 / (double)n[c0 + i]) : 0.f;
      p.blk0[i] = blk;
      if (da[c0 + i] && n[c0 + i]) blk += (int)std::min<size_t>((n[c0 + i] + 255) / 256, 8192);
      np = i + 1;
    }
    p.blk0[np] = blk;
    p.np = np;
    if (blk) hipLaunchKernelGGL(l1_multi_bwd_kernel, dim3(blk), dim3(256), 0, ST, p, g + c0, accumulate);
  }
  return check_launch("l1_multi_bwd");
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


// ---- scalar linear combinations of device scalars: the loss arithmetic of models/pix2pixHD_condImg_model.py:218-251 and
// train_mask2image.py:68-76 ((D_fake + D_real) * 0.5, G_GAN + G_GAN_Feat + G_VGG, the per-scale GAN-loss sums, x * lambda)
// as ONE launch forward and ONE backward instead of a chain of one-element ATen kernels.  out = scale * fl(sum_i fl(w_i t_i)),
// summed left to right with every product and sum rounded to fp32 (no contraction) -- the reference's own order.
int him_lincomb_fwd(const float* const* terms, const float* weights, int n, float scale, float* out, void* stream) {
  if (n <= 0 || n > 8 || !terms || !weights || !out) return fail(HIM_E_INVALID, "lincomb: 1..8 terms");
  LinComb p;
  for (int i = 0; i < 8; ++i) {
    p.t[i] = i < n ? terms[i] : nullptr;
    p.w[i] = i < n ? weights[i] : 0.f;
    p.d[i] = nullptr;
  }
  p.n = n;
  p.scale = scale;
  hipLaunchKernelGGL(lincomb_fwd_kernel, dim3(1), dim3(64), 0, ST, p, out);
  return check_launch("lincomb_fwd");
}
/* dterms[i][0] = g[0] * scale * weights[i]  (NULL entries skipped) */
int him_lincomb_bwd(const float* g, const float* weights, int n, float scale, float* const* dterms, void* stream) {
  if (n <= 0 || n > 8 || !g || !weights || !dterms) return fail(HIM_E_INVALID, "lincomb: 1..8 terms");
  LinComb p;
  for (int i = 0; i < 8; ++i) {
    p.t[i] = nullptr;
    p.w[i] = i < n ? weights[i] : 0.f;
    p.d[i] = i < n ? dterms[i] : nullptr;
  }
  p.n = n;
  p.scale = scale;
  hipLaunchKernelGGL(lincomb_bwd_kernel, dim3(1), dim3(64), 0, ST, p, g);
  return check_launch("lincomb_bwd");
}

}  // extern "C"
