// Building blocks of the box2mask generator (reference models/MaskTwoStreamConvSwitch_NET.py, models/layer_util.py
// ConvResnetBlock / DeconvResnetBlock, models/mask_losses.py): BatchNorm2d (training and eval mode), stand-alone
// activations, bilinear x2 upsampling, channel log-softmax, masked NLL and BCE losses.  All HBM-bound: coalesced NCHW
// streams, per-channel statistics by two-stage fixed-order reductions (no atomics, run-to-run deterministic).
#include <algorithm>

#include "him_common.h"

namespace him {

static inline dim3 gs_grid(long long n, int per_block = 256) {
  long long b = (n + per_block - 1) / per_block;
  if (b > 256 * 8 * 4) b = 256 * 8 * 4;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}
#define GS_LOOP(i, n) \
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)(n); i += (long long)gridDim.x * blockDim.x)

static const int BN_SLICES = 32;  // partial sums per channel (over contiguous runs of the B*HW positions)

__device__ __forceinline__ float act_grad_from_z(float z, int act, float slope) {
  if (act == HIM_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (act == HIM_ACT_LRELU) return z > 0.f ? 1.f : slope;
  if (act == HIM_ACT_TANH) {
    const float t = tanhf(z);
    return 1.f - t * t;
  }
  if (act == HIM_ACT_SIGMOID) {
    const float s = 1.f / (1.f + expf(-z));
    return s * (1.f - s);
  }
  return 1.f;
}

// ---- BatchNorm2d ------------------------------------------------------------------------------------------------
// partial[(c*S + s)*2 + {0,1}] = sum, sum of squares of (x - shift_c) over slice s of channel c; shift_c = x[0][c][0].
// A slice is a contiguous range of the channel's B*hw positions; inside one image plane the run is contiguous in memory
// (float4 loads when hw % 4 == 0), so no per-element division.
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, float* __restrict__ partial, int B, int C,
                                                       int hw) {
  __shared__ float sh[8];
  const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y;
  const long long n = (long long)B * hw;
  long long lo = n * s / S, hi = n * (s + 1) / S;
  const bool vec = (hw & 3) == 0;
  if (vec) {
    lo &= ~3ll;
    hi = (s == S - 1) ? n : (hi & ~3ll);
  }
  const float shift = x[(size_t)c * hw];
  float a = 0.f, q = 0.f;
  for (long long p0 = lo; p0 < hi;) {
    const int b = (int)(p0 / hw), r0 = (int)(p0 - (long long)b * hw);
    const int r1 = (int)min((long long)hw, r0 + (hi - p0));
    const float* __restrict__ pl = x + ((size_t)b * C + c) * hw;
    if (vec) {
      for (int r = r0 + threadIdx.x * 4; r < r1; r += 1024) {
        const float4 v = *(const float4*)(pl + r);
        const float d0 = v.x - shift, d1 = v.y - shift, d2 = v.z - shift, d3 = v.w - shift;
        a += (d0 + d1) + (d2 + d3);
        q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    } else {
      for (int r = r0 + threadIdx.x; r < r1; r += 256) {
        const float d = pl[r] - shift;
        a += d;
        q += d * d;
      }
    }
    p0 += r1 - r0;
  }
  a = block_sum_256(a, sh);
  q = block_sum_256(q, sh);
  if (threadIdx.x == 0) {
    partial[((size_t)c * S + s) * 2] = a;
    partial[((size_t)c * S + s) * 2 + 1] = q;
  }
}

// mean / rstd of the batch (training) or from the running statistics (eval); training also updates the running
// statistics the way torch does (momentum m, unbiased variance)
__global__ void bn_finalize_kernel(const float* __restrict__ x, const float* __restrict__ partial, float* __restrict__ mean,
                                   float* __restrict__ rstd, float* __restrict__ run_mean, float* __restrict__ run_var,
                                   int B, int C, int hw, int S, float eps, float momentum, int training) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (!training) {
    mean[c] = run_mean[c];
    rstd[c] = 1.f / sqrtf(run_var[c] + eps);
    return;
  }
  float a = 0.f, q = 0.f;
  for (int s = 0; s < S; ++s) {
    a += partial[((size_t)c * S + s) * 2];
    q += partial[((size_t)c * S + s) * 2 + 1];
  }
  const float n = (float)B * (float)hw;
  const float ms = a / n;
  const float mu = x[(size_t)c * hw] + ms;
  const float var = fmaxf(q / n - ms * ms, 0.f);
  mean[c] = mu;
  rstd[c] = 1.f / sqrtf(var + eps);
  if (run_mean) {
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * (n / fmaxf(n - 1.f, 1.f));
  }
}

// y = act((x - mean)*rstd*gamma + beta) (+ residual); grid (ceil(hw/1024), B*C): the plane index gives the channel
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ residual,
                                                       float* __restrict__ y, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int C, int hw, int act, float slope) {
  const int plane = blockIdx.y, c = plane % C;
  const float g = (gamma ? gamma[c] : 1.f) * rstd[c];
  const float b = (beta ? beta[c] : 0.f) - mean[c] * g;
  const size_t base = (size_t)plane * hw;
  if ((hw & 3) == 0) {
    const int r = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (r >= hw) return;
    const float4 v = *(const float4*)(x + base + r);
    float4 o;
    o.x = apply_act(fmaf(v.x, g, b), act, slope);
    o.y = apply_act(fmaf(v.y, g, b), act, slope);
    o.z = apply_act(fmaf(v.z, g, b), act, slope);
    o.w = apply_act(fmaf(v.w, g, b), act, slope);
    if (residual) {
      const float4 rr = *(const float4*)(residual + base + r);
      o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
    }
    *(float4*)(y + base + r) = o;
  } else {
    for (int r = blockIdx.x * 1024 + threadIdx.x; r < min(hw, (int)(blockIdx.x + 1) * 1024); r += 256) {
      float v = apply_act(fmaf(x[base + r], g, b), act, slope);
      if (residual) v += residual[base + r];
      y[base + r] = v;
    }
  }
}

// partial sums of dz and dz*xhat per channel slice, dz = dy * act'(z)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ partial, int B, int C, int hw, int act,
                                                            float slope) {
  __shared__ float sh[8];
  const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y;
  const long long n = (long long)B * hw;
  long long lo = n * s / S, hi = n * (s + 1) / S;
  const bool vec = (hw & 3) == 0;
  if (vec) {
    lo &= ~3ll;
    hi = (s == S - 1) ? n : (hi & ~3ll);
  }
  const float mu = mean[c], rs = rstd[c], g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
  float a = 0.f, q = 0.f;
  for (long long p0 = lo; p0 < hi;) {
    const int b = (int)(p0 / hw), r0 = (int)(p0 - (long long)b * hw);
    const int r1 = (int)min((long long)hw, r0 + (hi - p0));
    const size_t base = ((size_t)b * C + c) * hw;
    if (vec) {
      for (int r = r0 + threadIdx.x * 4; r < r1; r += 1024) {
        const float4 v = *(const float4*)(x + base + r), gy = *(const float4*)(dy + base + r);
        const float h0 = (v.x - mu) * rs, h1 = (v.y - mu) * rs, h2 = (v.z - mu) * rs, h3 = (v.w - mu) * rs;
        const float z0 = gy.x * act_grad_from_z(h0 * g + bt, act, slope), z1 = gy.y * act_grad_from_z(h1 * g + bt, act, slope);
        const float z2 = gy.z * act_grad_from_z(h2 * g + bt, act, slope), z3 = gy.w * act_grad_from_z(h3 * g + bt, act, slope);
        a += (z0 + z1) + (z2 + z3);
        q += (z0 * h0 + z1 * h1) + (z2 * h2 + z3 * h3);
      }
    } else {
      for (int r = r0 + threadIdx.x; r < r1; r += 256) {
        const float xh = (x[base + r] - mu) * rs;
        const float dz = dy[base + r] * act_grad_from_z(xh * g + bt, act, slope);
        a += dz;
        q += dz * xh;
      }
    }
    p0 += r1 - r0;
  }
  a = block_sum_256(a, sh);
  q = block_sum_256(q, sh);
  if (threadIdx.x == 0) {
    partial[((size_t)c * S + s) * 2] = a;
    partial[((size_t)c * S + s) * 2 + 1] = q;
  }
}

// sums[c*2+{0,1}] = total dz, total dz*xhat; dgamma/dbeta (+)=
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partial, float* __restrict__ sums, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, int C, int S, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float a = 0.f, q = 0.f;
  for (int s = 0; s < S; ++s) {
    a += partial[((size_t)c * S + s) * 2];
    q += partial[((size_t)c * S + s) * 2 + 1];
  }
  sums[c * 2] = a;
  sums[c * 2 + 1] = q;
  if (dgamma) dgamma[c] = accumulate ? dgamma[c] + q : q;
  if (dbeta) dbeta[c] = accumulate ? dbeta[c] + a : a;
}

// training: dx = gamma*rstd*(dz - mean(dz) - xhat*mean(dz*xhat)); eval: dx = gamma*rstd*dz.  grid (ceil(hw/1024), B*C)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ dx, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ sums,
                                                           int C, int hw, float inv_n, int act, float slope, int training) {
  const int plane = blockIdx.y, c = plane % C;
  const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f, rs = rstd[c], mu = mean[c];
  const float m1 = training ? sums[c * 2] * inv_n : 0.f, m2 = training ? sums[c * 2 + 1] * inv_n : 0.f;
  const float gr = g * rs;
  const size_t base = (size_t)plane * hw;
  if ((hw & 3) == 0) {
    const int r = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (r >= hw) return;
    const float4 v = *(const float4*)(x + base + r), gy = *(const float4*)(dy + base + r);
    const float h0 = (v.x - mu) * rs, h1 = (v.y - mu) * rs, h2 = (v.z - mu) * rs, h3 = (v.w - mu) * rs;
    float4 o;
    o.x = gr * (gy.x * act_grad_from_z(h0 * g + bt, act, slope) - m1 - h0 * m2);
    o.y = gr * (gy.y * act_grad_from_z(h1 * g + bt, act, slope) - m1 - h1 * m2);
    o.z = gr * (gy.z * act_grad_from_z(h2 * g + bt, act, slope) - m1 - h2 * m2);
    o.w = gr * (gy.w * act_grad_from_z(h3 * g + bt, act, slope) - m1 - h3 * m2);
    *(float4*)(dx + base + r) = o;
  } else {
    for (int r = blockIdx.x * 1024 + threadIdx.x; r < min(hw, (int)(blockIdx.x + 1) * 1024); r += 256) {
      const float xh = (x[base + r] - mu) * rs;
      const float dz = dy[base + r] * act_grad_from_z(xh * g + bt, act, slope);
      dx[base + r] = gr * (dz - m1 - xh * m2);
    }
  }
}

// ---- stand-alone activation ------------------------------------------------------------------------------------
__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, int act, float slope) {
  GS_LOOP(i, n) y[i] = apply_act(x[i], act, slope);
}

// ---- bilinear x2 upsampling (nn.Upsample(scale_factor=2, mode='bilinear')) -------------------------------------
__device__ __forceinline__ void bilinear_src(int o, int in_size, int out_size, int align, int* i0, int* i1, float* w1) {
  float src;
  if (align) {
    src = out_size > 1 ? (float)o * (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
  } else {
    src = ((float)o + 0.5f) * ((float)in_size / (float)out_size) - 0.5f;
    src = src < 0.f ? 0.f : src;
  }
  const int a = min((int)src, in_size - 1);
  *i0 = a;
  *i1 = min(a + 1, in_size - 1);
  *w1 = src - (float)a;
}

__global__ void upsample2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int planes, int H, int W, int align) {
  const int OH = 2 * H, OW = 2 * W;
  const long long n = (long long)planes * OH * OW;
  GS_LOOP(i, n) {
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
    const long long pl = i / ((long long)OW * OH);
    int y0, y1, x0, x1;
    float wy, wx;
    bilinear_src(oy, H, OH, align, &y0, &y1, &wy);
    bilinear_src(ox, W, OW, align, &x0, &x1, &wx);
    const float* __restrict__ p = x + pl * H * W;
    const float top = p[y0 * W + x0] * (1.f - wx) + p[y0 * W + x1] * wx;
    const float bot = p[y1 * W + x0] * (1.f - wx) + p[y1 * W + x1] * wx;
    y[i] = top * (1.f - wy) + bot * wy;
  }
}

// gather form of the adjoint: every input pixel sums the weights with which the (at most 6 x 6) nearby output pixels
// read it -- no atomics, fixed order
__global__ void upsample2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int planes, int H, int W,
                                     int align) {
  const int OH = 2 * H, OW = 2 * W;
  const long long n = (long long)planes * H * W;
  GS_LOOP(i, n) {
    const int ix = (int)(i % W), iy = (int)((i / W) % H);
    const long long pl = i / ((long long)W * H);
    const float* __restrict__ g = dy + pl * OH * OW;
    float wys[6], wxs[6];
    int oys[6], oxs[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int oy = 2 * iy - 2 + k, ox = 2 * ix - 2 + k;
      int a0, a1;
      float w1;
      float wy = 0.f, wx = 0.f;
      if (oy >= 0 && oy < OH) {
        bilinear_src(oy, H, OH, align, &a0, &a1, &w1);
        wy = (a0 == iy ? 1.f - w1 : 0.f) + (a1 == iy ? w1 : 0.f);
      }
      if (ox >= 0 && ox < OW) {
        bilinear_src(ox, W, OW, align, &a0, &a1, &w1);
        wx = (a0 == ix ? 1.f - w1 : 0.f) + (a1 == ix ? w1 : 0.f);
      }
      wys[k] = wy;
      wxs[k] = wx;
      oys[k] = min(max(oy, 0), OH - 1);
      oxs[k] = min(max(ox, 0), OW - 1);
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      if (wys[a] == 0.f) continue;
      float r = 0.f;
#pragma unroll
      for (int b = 0; b < 6; ++b) r += wxs[b] * g[oys[a] * OW + oxs[b]];
      s += wys[a] * r;
    }
    dx[i] = s;
  }
}

// ---- log-softmax over the channel axis (nn.LogSoftmax(dim=1)) ----------------------------------------------------
__global__ void logsoftmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, int hw) {
  const long long n = (long long)B * hw;
  GS_LOOP(i, n) {
    const int b = (int)(i / hw), r = (int)(i - (long long)b * hw);
    const float* __restrict__ p = x + (size_t)b * C * hw + r;
    float m = p[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, p[(size_t)c * hw]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(p[(size_t)c * hw] - m);
    const float lse = m + logf(s);
    float* __restrict__ o = y + (size_t)b * C * hw + r;
    for (int c = 0; c < C; ++c) o[(size_t)c * hw] = p[(size_t)c * hw] - lse;
  }
}
// dx = dy - exp(y) * sum_c dy
__global__ void logsoftmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, int B,
                                      int C, int hw) {
  const long long n = (long long)B * hw;
  GS_LOOP(i, n) {
    const int b = (int)(i / hw), r = (int)(i - (long long)b * hw);
    const size_t base = (size_t)b * C * hw + r;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += dy[base + (size_t)c * hw];
    for (int c = 0; c < C; ++c) dx[base + (size_t)c * hw] = dy[base + (size_t)c * hw] - expf(y[base + (size_t)c * hw]) * s;
  }
}

// ---- object-gated combination of the two streams' logits (MaskTwoStreamConv_NET.py:213-221, the parser's default net) ---
// comb[b,c,i] = (1 - p[b,i]) * ctx[b,c,i] + p[b,i] * obj[b,i]   with p = sigmoid(obj) broadcast over the C channels.
// The three products / the sum are rounded one by one (no fused multiply-add): the forward is torch's, bit for bit.
__global__ void gate_comb_fwd_kernel(const float* __restrict__ ctx, const float* __restrict__ p, const float* __restrict__ obj,
                                     float* __restrict__ out, int B, int C, int hw) {
  const long long total = (long long)B * C * hw;
  GS_LOOP(i, total) {
    const int px = (int)(i % hw);
    const int b = (int)(i / ((long long)C * hw));
    const float g = p[(size_t)b * hw + px];
    out[i] = __fadd_rn(__fmul_rn(__fsub_rn(1.f, g), ctx[i]), __fmul_rn(g, obj[(size_t)b * hw + px]));
  }
}
// dctx = (1 - p) * dout;  dobj = p * sum_c dout;  dp = sum_c dout * (obj - ctx)      (one thread per pixel walks the channels)
__global__ void gate_comb_bwd_kernel(const float* __restrict__ ctx, const float* __restrict__ p, const float* __restrict__ obj,
                                     const float* __restrict__ dout, float* __restrict__ dctx, float* __restrict__ dp,
                                     float* __restrict__ dobj, int B, int C, int hw) {
  const long long n = (long long)B * hw;
  GS_LOOP(i, n) {
    const int b = (int)(i / hw), r = (int)(i - (long long)b * hw);
    const size_t base = (size_t)b * C * hw + r;
    const float g = p[i], l = obj[i];
    float sd = 0.f, sg = 0.f;
    for (int c = 0; c < C; ++c) {
      const float d = dout[base + (size_t)c * hw];
      dctx[base + (size_t)c * hw] = (1.f - g) * d;
      sd += d;
      sg += d * (l - ctx[base + (size_t)c * hw]);
    }
    dobj[i] = g * sd;
    dp[i] = sg;
  }
}

// ---- losses ------------------------------------------------------------------------------------------------------
// MaskReconLoss (models/mask_losses.py:12-27): NLLLoss2d(ignore_index) with the positions where mask < 0.5 ignored:
// loss = -sum_{valid} logp[label] / #valid.  Stage 1: per-block (sum, count); stage 2 below.
__global__ __launch_bounds__(256) void masked_nll_stage1(const float* __restrict__ logp, const float* __restrict__ label,
                                                         const float* __restrict__ mask, float* __restrict__ partial, int B,
                                                         int C, int hw) {
  __shared__ float sh[8];
  const long long n = (long long)B * hw;
  float a = 0.f, cnt = 0.f;
  GS_LOOP(i, n) {
    const int b = (int)(i / hw), r = (int)(i - (long long)b * hw);
    const int id = (int)label[i];
    if (mask[i] >= 0.5f && id >= 0 && id < C) {
      a -= logp[((size_t)b * C + id) * hw + r];
      cnt += 1.f;
    }
  }
  a = block_sum_256(a, sh);
  cnt = block_sum_256(cnt, sh);
  if (threadIdx.x == 0) {
    partial[blockIdx.x * 2] = a;
    partial[blockIdx.x * 2 + 1] = cnt;
  }
}
__global__ __launch_bounds__(256) void masked_nll_stage2(const float* __restrict__ partial, int nb, float* __restrict__ out) {
  __shared__ float sh[8];
  float a = 0.f, cnt = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) {
    a += partial[i * 2];
    cnt += partial[i * 2 + 1];
  }
  a = block_sum_256(a, sh);
  cnt = block_sum_256(cnt, sh);
  if (threadIdx.x == 0) {
    out[0] = a / cnt;   // 0/0 = nan when nothing is valid, as torch
    out[1] = cnt;
  }
}
// dlogp[b][c][r] = -(g / count) at c = label for valid positions, 0 elsewhere
__global__ void masked_nll_bwd_kernel(const float* __restrict__ label, const float* __restrict__ mask,
                                      const float* __restrict__ g, const float* __restrict__ cnt, float* __restrict__ dlogp,
                                      int B, int C, int hw) {
  const long long n = (long long)B * C * hw;
  const float scale = -g[0] / cnt[0];
  GS_LOOP(i, n) {
    const int r = (int)(i % hw), c = (int)((i / hw) % C), b = (int)(i / ((long long)hw * C));
    const size_t pi = (size_t)b * hw + r;
    const int id = (int)label[pi];
    dlogp[i] = (mask[pi] >= 0.5f && id == c) ? scale : 0.f;
  }
}

// nn.BCELoss (mean): -(t*max(log p, -100) + (1-t)*max(log(1-p), -100))
__global__ __launch_bounds__(256) void bce_stage1(const float* __restrict__ p, const float* __restrict__ t,
                                                  float* __restrict__ partial, size_t n) {
  __shared__ float sh[8];
  float a = 0.f;
  GS_LOOP(i, n) {
    const float pp = p[i], tt = t[i];
    a -= tt * fmaxf(logf(pp), -100.f) + (1.f - tt) * fmaxf(logf(1.f - pp), -100.f);
  }
  a = block_sum_256(a, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = a;
}
__global__ __launch_bounds__(256) void sum_stage2(const float* __restrict__ partial, int nb, float scale, float* __restrict__ out) {
  __shared__ float sh[8];
  float a = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) a += partial[i];
  a = block_sum_256(a, sh);
  if (threadIdx.x == 0) out[0] = a * scale;
}
// dp = g/n * (p - t) / max(p*(1-p), 1e-12)   (torch's binary_cross_entropy_backward)
__global__ void bce_bwd_kernel(const float* __restrict__ p, const float* __restrict__ t, const float* __restrict__ g,
                               float* __restrict__ dp, size_t n, float inv_n) {
  const float s = g[0] * inv_n;
  GS_LOOP(i, n) {
    const float pp = p[i];
    dp[i] = s * (pp - t[i]) / fmaxf(pp * (1.f - pp), 1e-12f);
  }
}

}  // namespace him

using namespace him;

// ---- dilated conv3x3 as d*d dense convs: phase split ("space to batch") --------------------------------------------
// y[(b*d + py)*d + px][c][i][j] = x[b][c][i*d + py][j*d + px]   (inverse: the same map read the other way).
// A stride-1 conv with dilation d and zero padding d on x equals the plain pad-1 conv on every phase image (taps of
// one output only ever meet inputs of the same phase; the phase image's index -1 / H/d is the original's zero padding).
// Threads run along the ORIGINAL row (coalesced on the full-resolution side; the phase side is d-strided within rows of
// <= 32 floats at the shapes of the path).
__global__ __launch_bounds__(256) void space_batch_kernel(const float* __restrict__ src, float* __restrict__ dst, int B,
                                                          int C, int H, int W, int d, int inverse) {
  const long long n = (long long)B * C * H * W;
  const int Hd = H / d, Wd = W / d;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    long long r = i / W;
    const int y = (int)(r % H);
    r /= H;
    const int c = (int)(r % C), b = (int)(r / C);
    const int py = y % d, px = x % d;
    const long long j = ((((long long)(b * d + py) * d + px) * C + c) * Hd + y / d) * Wd + x / d;
    if (inverse) dst[i] = src[j];
    else dst[j] = src[i];
  }
}

// box2mask condition, object half (reference TwoStreamAE_mask.encode_input :127-152): the box mask goes into the channel
// of the object's class, every other of the NC channels is zero:  dst[b][c0 + c][px] = (c == cls[b]) ? mask[b][px] : 0.
// cls: one class id per sample as a float, ON THE DEVICE (no host read-back, no per-sample launches).
__global__ void class_mask_kernel(const float* __restrict__ mask, const float* __restrict__ cls, float* __restrict__ dst,
                                  int B, int NC, int Ctot, int c0, int hw) {
  const long long total = (long long)B * NC * hw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int px = (int)(i % hw);
    const long long r = i / hw;
    const int c = (int)(r % NC), b = (int)(r / NC);
    dst[((size_t)b * Ctot + c0 + c) * hw + px] = (c == (int)cls[b]) ? mask[(size_t)b * hw + px] : 0.f;
  }
}

// lr_control (reference models/Discriminator_NET.py:190-211) evaluated on the device: out2 = {g_lr, d_lr} in {0, 1}
__global__ void lr_control_kernel(const float* __restrict__ d_real, const float* __restrict__ d_fake, float margin,
                                  float* __restrict__ out2) {
  const float r = d_real[0], f = d_fake[0];
  bool upd_d = !(r < margin || f < margin);
  bool upd_g = !(r > 1.f - margin || f > 1.f - margin);
  if (!(upd_d || upd_g)) upd_d = upd_g = true;
  out2[0] = upd_g ? 1.f : 0.f;
  out2[1] = upd_d ? 1.f : 0.f;
}

#define ST ((hipStream_t)stream)

extern "C" {

size_t him_batchnorm_ws(int C) { return ((size_t)C * BN_SLICES * 2 + (size_t)C * 2) * sizeof(float) + 256; }

int him_batchnorm_fwd(const float* x, const float* residual, const float* gamma, const float* beta, float* run_mean,
                      float* run_var, float* y, float* save_mean, float* save_rstd, int B, int C, int hw, float eps,
                      float momentum, int training, int act, float slope, void* ws, size_t ws_bytes, void* stream) {
  if (B <= 0 || C <= 0 || hw <= 0) return fail(HIM_E_INVALID, "batchnorm: B=%d C=%d hw=%d", B, C, hw);
  if ((long long)B * C > 65535) return fail(HIM_E_UNSUPPORTED, "batchnorm: B*C = %lld planes > 65535", (long long)B * C);
  if (!training && (!run_mean || !run_var)) return fail(HIM_E_INVALID, "batchnorm: eval mode needs running statistics");
  if (!ws || ws_bytes < him_batchnorm_ws(C)) return fail(HIM_E_WORKSPACE, "batchnorm: ws too small");
  float* partial = (float*)ws;
  if (training) hipLaunchKernelGGL(bn_stats_kernel, dim3(C, BN_SLICES), dim3(256), 0, ST, x, partial, B, C, hw);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 64)), dim3(64), 0, ST, x, (const float*)partial, save_mean, save_rstd,
                     run_mean, run_var, B, C, hw, BN_SLICES, eps, momentum, training);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(cdiv(hw, 1024), B * C), dim3(256), 0, ST, x, residual, y, (const float*)save_mean,
                     (const float*)save_rstd, gamma, beta, C, hw, act, slope);
  return check_launch("batchnorm_fwd");
}

int him_batchnorm_bwd(const float* x, const float* gamma, const float* beta, const float* save_mean, const float* save_rstd,
                      const float* dy, float* dx, float* dgamma, float* dbeta, int B, int C, int hw, int training, int act,
                      float slope, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  if (B <= 0 || C <= 0 || hw <= 0) return fail(HIM_E_INVALID, "batchnorm: B=%d C=%d hw=%d", B, C, hw);
  if ((long long)B * C > 65535) return fail(HIM_E_UNSUPPORTED, "batchnorm: B*C = %lld planes > 65535", (long long)B * C);
  if (!ws || ws_bytes < him_batchnorm_ws(C)) return fail(HIM_E_WORKSPACE, "batchnorm: ws too small");
  float* partial = (float*)ws;
  float* sums = partial + (size_t)C * BN_SLICES * 2;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, BN_SLICES), dim3(256), 0, ST, x, dy, save_mean, save_rstd, gamma, beta,
                     partial, B, C, hw, act, slope);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 64)), dim3(64), 0, ST, (const float*)partial, sums, dgamma, dbeta, C,
                     BN_SLICES, accumulate);
  if (dx) {
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(cdiv(hw, 1024), B * C), dim3(256), 0, ST, x, dy, dx, save_mean, save_rstd,
                       gamma, beta, (const float*)sums, C, hw, 1.f / ((float)B * (float)hw), act, slope, training);
  }
  return check_launch("batchnorm_bwd");
}

int him_act_fwd(const float* x, float* y, size_t n, int act, float slope, void* stream) {
  if (!n) return HIM_OK;
  hipLaunchKernelGGL(act_fwd_kernel, gs_grid(n), dim3(256), 0, ST, x, y, n, act, slope);
  return check_launch("act_fwd");
}

int him_upsample2_fwd(const float* x, float* y, int planes, int H, int W, int align_corners, void* stream) {
  if (planes <= 0 || H <= 0 || W <= 0) return fail(HIM_E_INVALID, "upsample: planes=%d H=%d W=%d", planes, H, W);
  hipLaunchKernelGGL(upsample2_fwd_kernel, gs_grid((long long)planes * 4 * H * W), dim3(256), 0, ST, x, y, planes, H, W,
                     align_corners);
  return check_launch("upsample2_fwd");
}
int him_upsample2_bwd(const float* dy, float* dx, int planes, int H, int W, int align_corners, void* stream) {
  if (planes <= 0 || H <= 0 || W <= 0) return fail(HIM_E_INVALID, "upsample: planes=%d H=%d W=%d", planes, H, W);
  hipLaunchKernelGGL(upsample2_bwd_kernel, gs_grid((long long)planes * H * W), dim3(256), 0, ST, dy, dx, planes, H, W,
                     align_corners);
  return check_launch("upsample2_bwd");
}

int him_logsoftmax_fwd(const float* x, float* y, int B, int C, int hw, void* stream) {
  if (B <= 0 || C <= 0 || hw <= 0) return fail(HIM_E_INVALID, "logsoftmax: B=%d C=%d hw=%d", B, C, hw);
  hipLaunchKernelGGL(logsoftmax_fwd_kernel, gs_grid((long long)B * hw), dim3(256), 0, ST, x, y, B, C, hw);
  return check_launch("logsoftmax_fwd");
}
int him_logsoftmax_bwd(const float* y, const float* dy, float* dx, int B, int C, int hw, void* stream) {
  if (B <= 0 || C <= 0 || hw <= 0) return fail(HIM_E_INVALID, "logsoftmax: B=%d C=%d hw=%d", B, C, hw);
  hipLaunchKernelGGL(logsoftmax_bwd_kernel, gs_grid((long long)B * hw), dim3(256), 0, ST, y, dy, dx, B, C, hw);
  return check_launch("logsoftmax_bwd");
}

int him_gate_comb_fwd(const float* ctx, const float* p, const float* obj, float* out, int B, int C, int hw, void* stream) {
  if (B <= 0 || C <= 0 || hw <= 0) return fail(HIM_E_INVALID, "gate_comb: B=%d C=%d hw=%d", B, C, hw);
  hipLaunchKernelGGL(gate_comb_fwd_kernel, gs_grid((long long)B * C * hw), dim3(256), 0, ST, ctx, p, obj, out, B, C, hw);
  return check_launch("gate_comb_fwd");
}
int him_gate_comb_bwd(const float* ctx, const float* p, const float* obj, const float* dout, float* dctx, float* dp,
                      float* dobj, int B, int C, int hw, void* stream) {
  if (B <= 0 || C <= 0 || hw <= 0) return fail(HIM_E_INVALID, "gate_comb: B=%d C=%d hw=%d", B, C, hw);
  hipLaunchKernelGGL(gate_comb_bwd_kernel, gs_grid((long long)B * hw), dim3(256), 0, ST, ctx, p, obj, dout, dctx, dp, dobj,
                     B, C, hw);
  return check_launch("gate_comb_bwd");
}

static const int LOSS_BLOCKS = 1024;
size_t him_mask_loss_ws(void) { return (size_t)LOSS_BLOCKS * 2 * sizeof(float) + 256; }

int him_masked_nll_fwd(const float* logp, const float* label, const float* mask, float* out2, int B, int C, int hw, void* ws,
                       size_t ws_bytes, void* stream) {
  if (!ws || ws_bytes < him_mask_loss_ws()) return fail(HIM_E_WORKSPACE, "masked_nll: ws too small");
  const int nb = (int)std::min<long long>(cdiv((long long)B * hw, 256), LOSS_BLOCKS);
  hipLaunchKernelGGL(masked_nll_stage1, dim3(nb), dim3(256), 0, ST, logp, label, mask, (float*)ws, B, C, hw);
  hipLaunchKernelGGL(masked_nll_stage2, dim3(1), dim3(256), 0, ST, (const float*)ws, nb, out2);
  return check_launch("masked_nll_fwd");
}
int him_masked_nll_bwd(const float* label, const float* mask, const float* g, const float* count, float* dlogp, int B, int C,
                       int hw, void* stream) {
  hipLaunchKernelGGL(masked_nll_bwd_kernel, gs_grid((long long)B * C * hw), dim3(256), 0, ST, label, mask, g, count, dlogp, B,
                     C, hw);
  return check_launch("masked_nll_bwd");
}

int him_bce_mean_fwd(const float* p, const float* t, size_t n, float* out, void* ws, size_t ws_bytes, void* stream) {
  if (!n) return fail(HIM_E_INVALID, "bce: empty input");
  if (!ws || ws_bytes < him_mask_loss_ws()) return fail(HIM_E_WORKSPACE, "bce: ws too small");
  const int nb = (int)std::min<long long>(cdiv((long long)n, 256), LOSS_BLOCKS);
  hipLaunchKernelGGL(bce_stage1, dim3(nb), dim3(256), 0, ST, p, t, (float*)ws, n);
  hipLaunchKernelGGL(sum_stage2, dim3(1), dim3(256), 0, ST, (const float*)ws, nb, (float)(1.0 / (double)n), out);
  return check_launch("bce_mean_fwd");
}
int him_bce_mean_bwd(const float* p, const float* t, size_t n, const float* g, float* dp, void* stream) {
  if (!n) return HIM_OK;
  hipLaunchKernelGGL(bce_bwd_kernel, gs_grid(n), dim3(256), 0, ST, p, t, g, dp, n, (float)(1.0 / (double)n));
  return check_launch("bce_mean_bwd");
}

int him_space_to_batch(const float* x, float* y, int B, int C, int H, int W, int d, int inverse, void* stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || d < 1) return fail(HIM_E_INVALID, "space_to_batch: bad shape");
  if (H % d || W % d) return fail(HIM_E_UNSUPPORTED, "space_to_batch: %dx%d not divisible by dilation %d", H, W, d);
  hipLaunchKernelGGL(space_batch_kernel, gs_grid((size_t)B * C * H * W), dim3(256), 0, ST, x, y, B, C, H, W, d, inverse);
  return check_launch("space_to_batch");
}
int him_class_mask(const float* mask, const float* cls, float* dst, int B, int NC, int Ctot, int c0, int hw, void* stream) {
  if (B <= 0 || NC <= 0 || hw <= 0 || c0 < 0 || c0 + NC > Ctot) return fail(HIM_E_INVALID, "class_mask: bad shape");
  hipLaunchKernelGGL(class_mask_kernel, gs_grid((size_t)B * NC * hw), dim3(256), 0, ST, mask, cls, dst, B, NC, Ctot, c0, hw);
  return check_launch("class_mask");
}
int him_lr_control(const float* loss_d_real, const float* loss_d_fake, float margin, float* out2, void* stream) {
  hipLaunchKernelGGL(lr_control_kernel, dim3(1), dim3(1), 0, ST, loss_d_real, loss_d_fake, margin, out2);
  return check_launch("lr_control");
}

}  // extern "C"
