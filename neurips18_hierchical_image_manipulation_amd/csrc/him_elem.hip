// HBM-bound elementwise / pooling / input-encoding kernels (grid-stride, coalesced NCHW streams).
#include <algorithm>

#include "him_common.h"

namespace him {

static inline dim3 gs_grid(long long n, int per_block = 256) {
  long long b = (n + per_block - 1) / per_block;
  if (b > 256 * 8 * 4) b = 256 * 8 * 4;  // 256 CUs x 8 blocks x 4
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}
#define GS_LOOP(i, n) \
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)(n); i += (long long)gridDim.x * blockDim.x)

__global__ void act_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dz,
                               size_t n, int act, float slope) {
  GS_LOOP(i, n) {
    const float yy = y[i], g = dy[i];
    float d;
    if (act == HIM_ACT_RELU) d = yy > 0.f ? g : 0.f;
    else if (act == HIM_ACT_LRELU) d = yy > 0.f ? g : g * slope;
    else if (act == HIM_ACT_TANH) d = g * (1.f - yy * yy);
    else if (act == HIM_ACT_SIGMOID) d = g * yy * (1.f - yy);
    else d = g;
    dz[i] = d;
  }
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, size_t n) {
  GS_LOOP(i, n) o[i] = a[i] + b[i];
}
__global__ void fill_kernel(float* __restrict__ p, size_t n, float v) { GS_LOOP(i, n) p[i] = v; }
__global__ void scale_kernel(float* __restrict__ p, size_t n, float s) { GS_LOOP(i, n) p[i] *= s; }

// one-hot: dst[b][c0+c][i] = (label[b][i] == c)
__global__ void onehot_kernel(const float* __restrict__ label, float* __restrict__ dst, int B, int nc, int Ctot,
                              int c0, int hw) {
  const long long total = (long long)B * nc * hw;
  GS_LOOP(i, total) {
    const int px = (int)(i % hw);
    const long long r = i / hw;
    const int c = (int)(r % nc);
    const int b = (int)(r / nc);
    const int id = (int)label[(size_t)b * hw + px];
    dst[((size_t)b * Ctot + c0 + c) * hw + px] = id == c ? 1.f : 0.f;
  }
}

// AvgPool2d(3, stride 2, pad 1, count_include_pad=False) of one-hot(label) WITHOUT the one-hot tensor: the pooled value of
// class c is (number of window pixels with id c) / (window pixels inside the image) -- sums of 0 / 1 are exact in any
// order, so this equals avgpool3s2_fwd_kernel applied to onehot_kernel's output bit for bit.  One thread per output pixel
// (<= 9 id loads), nc coalesced stores.  dst[b][c0 + c][oy][ox]
__global__ void onehot_pool3s2_kernel(const float* __restrict__ label, float* __restrict__ dst, int B, int nc, int Ctot,
                                      int c0, int H, int W, int OH, int OW) {
  const long long total = (long long)B * OH * OW;
  GS_LOOP(i, total) {
    const int ox = (int)(i % OW);
    const long long r = i / OW;
    const int oy = (int)(r % OH);
    const int b = (int)(r / OH);
    const float* __restrict__ lab = label + (size_t)b * H * W;
    const int y0 = max(oy * 2 - 1, 0), y1 = min(oy * 2 + 2, H), x0 = max(ox * 2 - 1, 0), x1 = min(ox * 2 + 2, W);
    int ids[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int yy = y0 + k / 3, xx = x0 + k % 3;
      ids[k] = (yy < y1 && xx < x1) ? (int)lab[yy * W + xx] : -1;
    }
    const float cnt = (float)((y1 - y0) * (x1 - x0));
    float* __restrict__ o = dst + ((size_t)b * Ctot + c0) * OH * OW + (size_t)oy * OW + ox;
    for (int c = 0; c < nc; ++c) {
      int s = 0;
#pragma unroll
      for (int k = 0; k < 9; ++k) s += ids[k] == c ? 1 : 0;
      o[(size_t)c * OH * OW] = (float)s / cnt;
    }
  }
}

// compact label maps: uint8 ids (as stored on disk / sent over PCIe) -> the float id map the kernels read
__global__ void u8_to_f32_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, size_t n) {
  GS_LOOP(i, n) dst[i] = (float)src[i];
}

// get_masked_image (reference data/base_dataset.py:342-357) for a batch: bbox[b] = (wmin, hmin, wmax, hmax) as floats;
// mask = 1 inside the box (empty unless hmax > hmin and wmax > wmin), obj = mask * image,
// ctx = (1 - mask) * image + mask * cls2fill.  Any of the three outputs may be NULL.
__global__ void masked_image_kernel(const float* __restrict__ image, const float* __restrict__ bbox,
                                    float* __restrict__ mask, float* __restrict__ obj, float* __restrict__ ctx, int B,
                                    int C, int H, int W, float cls2fill) {
  const long long total = (long long)B * C * H * W;
  GS_LOOP(i, total) {
    const int x = (int)(i % W);
    long long r = i / W;
    const int y = (int)(r % H);
    r /= H;
    const int c = (int)(r % C), b = (int)(r / C);
    const int wmin = (int)bbox[b * 4 + 0], hmin = (int)bbox[b * 4 + 1], wmax = (int)bbox[b * 4 + 2],
              hmax = (int)bbox[b * 4 + 3];
    // python slice semantics of masked_tensor[0, hmin:hmax, wmin:wmax] for the non-negative boxes the samplers emit
    const bool in = hmax > hmin && wmax > wmin && y >= hmin && y < hmax && x >= wmin && x < wmax;
    const float m = in ? 1.f : 0.f, v = image[i];
    if (mask && c == 0) mask[((size_t)b * H + y) * W + x] = m;
    if (obj) obj[i] = m * v;
    if (ctx) ctx[i] = (1.f - m) * v + m * cls2fill;
  }
}

__global__ void edges_kernel(const float* __restrict__ t, float* __restrict__ dst, int B, int H, int W, int Ctot,
                             int c0) {
  const long long total = (long long)B * H * W;
  GS_LOOP(i, total) {
    const int x = (int)(i % W);
    const long long r = i / W;
    const int y = (int)(r % H);
    const int b = (int)(r / H);
    const float* p = t + (size_t)b * H * W;
    const float v = p[y * W + x];
    bool e = false;
    if (x > 0) e |= p[y * W + x - 1] != v;
    if (x < W - 1) e |= p[y * W + x + 1] != v;
    if (y > 0) e |= p[(y - 1) * W + x] != v;
    if (y < H - 1) e |= p[(y + 1) * W + x] != v;
    dst[((size_t)b * Ctot + c0) * H * W + (size_t)y * W + x] = e ? 1.f : 0.f;
  }
}

// emb[b][c] = clamp(noise * sum(image*mask)/sum(mask), -1, 1); one block per (b, c)
__global__ __launch_bounds__(256) void masked_mean_kernel(const float* __restrict__ image,
                                                          const float* __restrict__ mask,
                                                          const float* __restrict__ noise, float* __restrict__ emb,
                                                          int hw) {
  __shared__ float sh[8];
  const int b = blockIdx.x / 3, c = blockIdx.x % 3;
  const float* ip = image + ((size_t)b * 3 + c) * hw;
  const float* mp = mask + (size_t)b * hw;
  float s = 0.f, cnt = 0.f;
  for (int i = threadIdx.x; i < hw; i += 256) {
    const float m = mp[i];
    s += ip[i] * m;
    cnt += m;
  }
  s = block_sum_256(s, sh);
  cnt = block_sum_256(cnt, sh);
  if (threadIdx.x == 0) {
    float e = cnt > 0.f ? s / cnt : 0.f;
    if (noise) e *= noise[b * 3 + c];
    e = fminf(fmaxf(e, -1.f), 1.f);
    emb[b * 3 + c] = e;
  }
}

__global__ void tile_embed_kernel(const float* __restrict__ emb, const float* __restrict__ mask,
                                  float* __restrict__ dst, int B, int Ctot, int c0, int hw) {
  const long long total = (long long)B * 3 * hw;
  GS_LOOP(i, total) {
    const int px = (int)(i % hw);
    const long long r = i / hw;
    const int c = (int)(r % 3);
    const int b = (int)(r / 3);
    dst[((size_t)b * Ctot + c0 + c) * hw + px] = emb[b * 3 + c] * mask[(size_t)b * hw + px];
  }
}

__global__ void copy_channels_kernel(const float* __restrict__ src, int Csrc, int cs0, float* __restrict__ dst,
                                     int Cdst, int cd0, int n, int B, int hw, const float* __restrict__ mask,
                                     int mode, int accumulate) {
  const long long total = (long long)B * n * hw;
  GS_LOOP(i, total) {
    const int px = (int)(i % hw);
    const long long r = i / hw;
    const int c = (int)(r % n);
    const int b = (int)(r / n);
    float v = src[((size_t)b * Csrc + cs0 + c) * hw + px];
    if (mode != 0) {
      const float m = mask[(size_t)b * hw + px];
      v *= mode == 1 ? m : 1.f - m;
    }
    float* o = dst + ((size_t)b * Cdst + cd0 + c) * hw + px;
    *o = accumulate ? *o + v : v;
  }
}

__global__ void blend_kernel(const float* __restrict__ a, int Ca, int ca0, const float* __restrict__ bsrc, int Cb,
                             int cb0, const float* __restrict__ m, float* __restrict__ out, int B, int C, int hw) {
  const long long total = (long long)B * C * hw;
  GS_LOOP(i, total) {
    const int px = (int)(i % hw);
    const long long r = i / hw;
    const int c = (int)(r % C);
    const int b = (int)(r / C);
    const float mm = m[(size_t)b * hw + px];
    const float av = a[((size_t)b * Ca + ca0 + c) * hw + px];
    const float bv = bsrc[((size_t)b * Cb + cb0 + c) * hw + px];
    out[i] = (1.f - mm) * av + mm * bv;
  }
}

// AvgPool2d(3, stride 2, pad 1, count_include_pad=False)
__global__ void avgpool3s2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int planes, int H, int W,
                                      int OH, int OW) {
  const long long total = (long long)planes * OH * OW;
  GS_LOOP(i, total) {
    const int ox = (int)(i % OW);
    const long long r = i / OW;
    const int oy = (int)(r % OH);
    const long long pl = r / OH;
    const float* p = x + pl * H * W;
    const int y0 = max(oy * 2 - 1, 0), y1 = min(oy * 2 + 2, H), x0 = max(ox * 2 - 1, 0), x1 = min(ox * 2 + 2, W);
    float s = 0.f;
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) s += p[yy * W + xx];
    y[i] = s / (float)((y1 - y0) * (x1 - x0));
  }
}
__global__ void avgpool3s2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int planes, int H,
                                      int W, int OH, int OW) {
  const long long total = (long long)planes * H * W;
  GS_LOOP(i, total) {
    const int xx = (int)(i % W);
    const long long r = i / W;
    const int yy = (int)(r % H);
    const long long pl = r / H;
    const float* g = dy + pl * OH * OW;
    // windows containing (yy,xx): oy with 2*oy-1 <= yy <= 2*oy+1
    const int oy0 = max((yy) / 2, 0), oy1 = min((yy + 1) / 2, OH - 1);
    const int ox0 = max((xx) / 2, 0), ox1 = min((xx + 1) / 2, OW - 1);
    float s = 0.f;
    for (int oy = oy0; oy <= oy1; ++oy) {
      const int hy = min(oy * 2 + 2, H) - max(oy * 2 - 1, 0);
      for (int ox = ox0; ox <= ox1; ++ox) {
        const int wx = min(ox * 2 + 2, W) - max(ox * 2 - 1, 0);
        s += g[oy * OW + ox] / (float)(hy * wx);
      }
    }
    dx[i] = s;
  }
}

__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int planes, int H, int W,
                                   int k) {
  const int OH = H / k, OW = W / k;
  const long long total = (long long)planes * OH * OW;
  GS_LOOP(i, total) {
    const int ox = (int)(i % OW);
    const long long r = i / OW;
    const int oy = (int)(r % OH);
    const long long pl = r / OH;
    const float* p = x + pl * H * W + (size_t)oy * k * W + ox * k;
    float m = p[0];
    for (int a = 0; a < k; ++a)
      for (int b = 0; b < k; ++b) {
        const float v = p[a * W + b];
        m = (v > m || v != v) ? v : m;
      }
    y[i] = m;
  }
}
// one thread per OUTPUT window: zero-fill the window then route the gradient to the first maximum
// relu_gate: x is a ReLU output whose own activation backward is folded in here: a window whose maximum is not positive
// passes no gradient (the ReLU derivative at the selected element is 0)
__global__ void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                   int planes, int H, int W, int k, int relu_gate) {
  const int OH = H / k, OW = W / k;
  const long long total = (long long)planes * OH * OW;
  GS_LOOP(i, total) {
    const int ox = (int)(i % OW);
    const long long r = i / OW;
    const int oy = (int)(r % OH);
    const long long pl = r / OH;
    const size_t base = pl * H * W + (size_t)oy * k * W + ox * k;
    const float* p = x + base;
    float m = p[0];
    int am = 0;
    for (int a = 0; a < k; ++a)
      for (int b = 0; b < k; ++b) {
        const float v = p[a * W + b];
        if (v > m || v != v) {
          m = v;
          am = a * W + b;
        }
      }
    const float g = (relu_gate && !(m > 0.f)) ? 0.f : dy[i];
    for (int a = 0; a < k; ++a)
      for (int b = 0; b < k; ++b) dx[base + a * W + b] = (a * W + b) == am ? g : 0.f;
  }
}
__global__ void zero_tail_kernel(float* __restrict__ dx, int planes, int H, int W, int k) {
  // rows/cols not covered by any k x k window (H or W not a multiple of k)
  const int CH = (H / k) * k, CW = (W / k) * k;
  const long long total = (long long)planes * H * W;
  GS_LOOP(i, total) {
    const int xx = (int)(i % W);
    const int yy = (int)((i / W) % H);
    if (yy >= CH || xx >= CW) dx[i] = 0.f;
  }
}

}  // namespace him

using namespace him;
#define ST ((hipStream_t)stream)

extern "C" {

int him_act_bwd(const float* y, const float* dy, float* dz, size_t n, int act, float slope, void* stream) {
  if (!n) return HIM_OK;
  hipLaunchKernelGGL(act_bwd_kernel, gs_grid(n), dim3(256), 0, ST, y, dy, dz, n, act, slope);
  return check_launch("act_bwd");
}
int him_add(const float* a, const float* b, float* out, size_t n, void* stream) {
  if (!n) return HIM_OK;
  hipLaunchKernelGGL(add_kernel, gs_grid(n), dim3(256), 0, ST, a, b, out, n);
  return check_launch("add");
}
int him_fill(float* p, size_t n, float value, void* stream) {
  if (!n) return HIM_OK;
  hipLaunchKernelGGL(fill_kernel, gs_grid(n), dim3(256), 0, ST, p, n, value);
  return check_launch("fill");
}
int him_scale(float* p, size_t n, float s, void* stream) {
  if (!n) return HIM_OK;
  hipLaunchKernelGGL(scale_kernel, gs_grid(n), dim3(256), 0, ST, p, n, s);
  return check_launch("scale");
}
int him_u8_to_f32(const unsigned char* src, float* dst, size_t n, void* stream) {
  if (!n) return HIM_OK;
  hipLaunchKernelGGL(u8_to_f32_kernel, gs_grid(n), dim3(256), 0, ST, src, dst, n);
  return check_launch("u8_to_f32");
}
int him_masked_image(const float* image, const float* bbox, float* mask, float* masked_object, float* masked_context,
                     int B, int C, int H, int W, float cls2fill, void* stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return fail(HIM_E_INVALID, "masked_image: bad shape");
  hipLaunchKernelGGL(masked_image_kernel, gs_grid((size_t)B * C * H * W), dim3(256), 0, ST, image, bbox, mask,
                     masked_object, masked_context, B, C, H, W, cls2fill);
  return check_launch("masked_image");
}
int him_onehot(const float* label, float* dst, int B, int label_nc, int Ctot, int c0, int hw, void* stream) {
  if (c0 < 0 || c0 + label_nc > Ctot) return fail(HIM_E_INVALID, "onehot: channel slice out of range");
  hipLaunchKernelGGL(onehot_kernel, gs_grid((long long)B * label_nc * hw), dim3(256), 0, ST, label, dst, B,
                     label_nc, Ctot, c0, hw);
  return check_launch("onehot");
}
int him_onehot_pool3s2(const float* label, float* dst, int B, int label_nc, int Ctot, int c0, int H, int W, int OH, int OW,
                       void* stream) {
  if (c0 < 0 || c0 + label_nc > Ctot) return fail(HIM_E_INVALID, "onehot_pool: channel slice out of range");
  if (OH != (H + 2 - 3) / 2 + 1 || OW != (W + 2 - 3) / 2 + 1) return fail(HIM_E_INVALID, "onehot_pool: bad OH/OW");
  hipLaunchKernelGGL(onehot_pool3s2_kernel, gs_grid((long long)B * OH * OW), dim3(256), 0, ST, label, dst, B, label_nc,
                     Ctot, c0, H, W, OH, OW);
  return check_launch("onehot_pool3s2");
}
int him_edges(const float* inst, float* dst, int B, int H, int W, int Ctot, int c0, void* stream) {
  if (c0 < 0 || c0 + 1 > Ctot) return fail(HIM_E_INVALID, "edges: channel slice out of range");
  hipLaunchKernelGGL(edges_kernel, gs_grid((long long)B * H * W), dim3(256), 0, ST, inst, dst, B, H, W, Ctot, c0);
  return check_launch("edges");
}
int him_masked_mean(const float* image, const float* obj_mask, const float* noise, float* emb, int B, int hw,
                    void* stream) {
  hipLaunchKernelGGL(masked_mean_kernel, dim3(B * 3), dim3(256), 0, ST, image, obj_mask, noise, emb, hw);
  return check_launch("masked_mean");
}
int him_tile_embed(const float* emb, const float* mask, float* dst, int B, int Ctot, int c0, int hw, void* stream) {
  if (c0 < 0 || c0 + 3 > Ctot) return fail(HIM_E_INVALID, "tile_embed: channel slice out of range");
  hipLaunchKernelGGL(tile_embed_kernel, gs_grid((long long)B * 3 * hw), dim3(256), 0, ST, emb, mask, dst, B, Ctot,
                     c0, hw);
  return check_launch("tile_embed");
}
int him_copy_channels(const float* src, int Csrc, int cs0, float* dst, int Cdst, int cd0, int n, int B, int hw,
                      const float* mask, int mask_mode, int accumulate, void* stream) {
  if (cs0 < 0 || cs0 + n > Csrc || cd0 < 0 || cd0 + n > Cdst)
    return fail(HIM_E_INVALID, "copy_channels: slice out of range");
  if (mask_mode != 0 && !mask) return fail(HIM_E_INVALID, "copy_channels: mask_mode without mask");
  hipLaunchKernelGGL(copy_channels_kernel, gs_grid((long long)B * n * hw), dim3(256), 0, ST, src, Csrc, cs0, dst,
                     Cdst, cd0, n, B, hw, mask, mask_mode, accumulate);
  return check_launch("copy_channels");
}
int him_blend(const float* a, int Ca, int ca0, const float* b, int Cb, int cb0, const float* m, float* out, int B,
              int C, int hw, void* stream) {
  if (ca0 < 0 || ca0 + C > Ca || cb0 < 0 || cb0 + C > Cb) return fail(HIM_E_INVALID, "blend: slice out of range");
  hipLaunchKernelGGL(blend_kernel, gs_grid((long long)B * C * hw), dim3(256), 0, ST, a, Ca, ca0, b, Cb, cb0, m, out,
                     B, C, hw);
  return check_launch("blend");
}
int him_avgpool3s2_fwd(const float* x, float* y, int planes, int H, int W, int OH, int OW, void* stream) {
  if (OH != (H + 2 - 3) / 2 + 1 || OW != (W + 2 - 3) / 2 + 1) return fail(HIM_E_INVALID, "avgpool: bad OH/OW");
  hipLaunchKernelGGL(avgpool3s2_fwd_kernel, gs_grid((long long)planes * OH * OW), dim3(256), 0, ST, x, y, planes, H,
                     W, OH, OW);
  return check_launch("avgpool_fwd");
}
int him_avgpool3s2_bwd(const float* dy, float* dx, int planes, int H, int W, int OH, int OW, void* stream) {
  if (OH != (H + 2 - 3) / 2 + 1 || OW != (W + 2 - 3) / 2 + 1) return fail(HIM_E_INVALID, "avgpool: bad OH/OW");
  hipLaunchKernelGGL(avgpool3s2_bwd_kernel, gs_grid((long long)planes * H * W), dim3(256), 0, ST, dy, dx, planes, H,
                     W, OH, OW);
  return check_launch("avgpool_bwd");
}
int him_maxpool_fwd(const float* x, float* y, int planes, int H, int W, int k, void* stream) {
  if (k <= 0 || H / k <= 0 || W / k <= 0) return fail(HIM_E_INVALID, "maxpool: bad k");
  hipLaunchKernelGGL(maxpool_fwd_kernel, gs_grid((long long)planes * (H / k) * (W / k)), dim3(256), 0, ST, x, y,
                     planes, H, W, k);
  return check_launch("maxpool_fwd");
}
static int maxpool_bwd_impl(const float* x, const float* dy, float* dx, int planes, int H, int W, int k, int relu_gate,
                            void* stream) {
  if (k <= 0 || H / k <= 0 || W / k <= 0) return fail(HIM_E_INVALID, "maxpool: bad k");
  if (H % k || W % k) {
    hipLaunchKernelGGL(zero_tail_kernel, gs_grid((long long)planes * H * W), dim3(256), 0, ST, dx, planes, H, W, k);
    int rc = check_launch("maxpool_tail");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(maxpool_bwd_kernel, gs_grid((long long)planes * (H / k) * (W / k)), dim3(256), 0, ST, x, dy, dx,
                     planes, H, W, k, relu_gate);
  return check_launch("maxpool_bwd");
}
int him_maxpool_bwd(const float* x, const float* dy, float* dx, int planes, int H, int W, int k, void* stream) {
  return maxpool_bwd_impl(x, dy, dx, planes, H, W, k, 0, stream);
}
int him_maxpool_relu_bwd(const float* x, const float* dy, float* dx, int planes, int H, int W, int k, void* stream) {
  return maxpool_bwd_impl(x, dy, dx, planes, H, W, k, 1, stream);
}

}  // extern "C"
