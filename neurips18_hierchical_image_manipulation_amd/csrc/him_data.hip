// Loader-side pixel work on the device: crop windows of label / instance maps and photographs arrive as raw bytes,
// everything the reference's loader does to them afterwards (data/base_dataset.py:243-268 -> PIL.Image.resize,
// FLIP_LEFT_RIGHT, ToTensor, Normalize; data/segmentation_dataset.py:86-131 masks) happens here in four launches per
// batch.  Byte / integer work, HBM-bound and tiny next to the training step; the point is that the batch is born on
// the device in the trainer's compact layout and that the bytes equal Pillow's (the index / weight tables come from
// data/resample.py, the arithmetic below is Pillow's 8-bit fixed point: Resample.c ImagingResampleHorizontal_8bpc).
#include "him_common.h"

namespace him {

#define DATA_PRECISION_BITS 22

__device__ __forceinline__ int clip8(int v) {
  v >>= DATA_PRECISION_BITS;  // arithmetic shift, as Pillow's clip8_lookups index
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// out[b][y][x] = src_b[ytab[b][y]][xtab[b][x]]   (a flip is already folded into xtab)
template <typename SRC>
__global__ void data_nearest_kernel(const unsigned char* __restrict__ base, const long long* __restrict__ off,
                                    const int* __restrict__ pitch, const int* __restrict__ xtab,
                                    const int* __restrict__ ytab, void* __restrict__ dst, int dst_kind, int B, int H,
                                    int W) {
  long long n = (long long)B * H * W;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    int x = (int)(i % W);
    long long r = i / W;
    int y = (int)(r % H);
    int b = (int)(r / H);
    const SRC* s = (const SRC*)(base + off[b]);
    SRC v = s[(long long)ytab[b * H + y] * pitch[b] + xtab[b * W + x]];
    switch (dst_kind) {
      case 0: ((float*)dst)[i] = (float)v; break;                       // ToTensor()*255 of an 8-bit map is the id itself
      case 1: ((float*)dst)[i] = __fdiv_rn((float)v, 255.f); break;     // ToTensor() of an 8-bit map
      case 2: ((unsigned char*)dst)[i] = (unsigned char)v; break;       // compact ids for the trainer's uint8 input
      default: ((int*)dst)[i] = (int)v; break;                          // ToTensor() of an integer-mode map
    }
  }
}

// horizontal pass over interleaved RGB bytes: tmp[b][r][i][c] = clip8(2^21 + sum_k w[b][i][k] * src_b[r][first+k][c])
__global__ void data_bicubic_h_kernel(const unsigned char* __restrict__ base, const long long* __restrict__ off,
                                      const int* __restrict__ pitch, const int* __restrict__ rows,
                                      const int* __restrict__ first, const int* __restrict__ count,
                                      const int* __restrict__ weights, int ksize, unsigned char* __restrict__ tmp,
                                      int maxrows, int B, int W) {
  int b = blockIdx.z;
  int r = blockIdx.y;
  if (r >= rows[b]) return;
  const unsigned char* s = base + off[b] + (long long)r * pitch[b] * 3;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W; i += gridDim.x * blockDim.x) {
    int f = first[b * W + i], n = count[b * W + i];
    const int* w = weights + ((long long)b * W + i) * ksize;
    int a0 = 1 << (DATA_PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int k = 0; k < n; ++k) {
      int wk = w[k];
      const unsigned char* p = s + (f + k) * 3;
      a0 += (int)p[0] * wk;
      a1 += (int)p[1] * wk;
      a2 += (int)p[2] * wk;
    }
    unsigned char* o = tmp + (((long long)b * maxrows + r) * W + i) * 3;
    o[0] = (unsigned char)clip8(a0);
    o[1] = (unsigned char)clip8(a1);
    o[2] = (unsigned char)clip8(a2);
  }
}

// vertical pass + FLIP_LEFT_RIGHT + ToTensor + Normalize(.5,.5): dst[b][c][y][x'] = ((v / 255) - .5) / .5
__global__ void data_bicubic_v_kernel(const unsigned char* __restrict__ tmp, int maxrows,
                                      const int* __restrict__ first, const int* __restrict__ count,
                                      const int* __restrict__ weights, int ksize, const int* __restrict__ flip,
                                      float* __restrict__ dst, int normalize, int B, int H, int W) {
  int b = blockIdx.z;
  int y = blockIdx.y;
  int f = first[b * H + y], n = count[b * H + y];
  const int* w = weights + ((long long)b * H + y) * ksize;
  for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < W; x += gridDim.x * blockDim.x) {
    int a0 = 1 << (DATA_PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int k = 0; k < n; ++k) {
      int wk = w[k];
      const unsigned char* p = tmp + (((long long)b * maxrows + f + k) * W + x) * 3;
      a0 += (int)p[0] * wk;
      a1 += (int)p[1] * wk;
      a2 += (int)p[2] * wk;
    }
    int xo = flip[b] ? W - 1 - x : x;
    int v[3] = {clip8(a0), clip8(a1), clip8(a2)};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float t = __fdiv_rn((float)v[c], 255.f);
      if (normalize) t = __fdiv_rn(__fsub_rn(t, 0.5f), 0.5f);
      dst[(((long long)b * 3 + c) * H + y) * W + xo] = t;
    }
  }
}

// get_masked_image for the input and the output window + the instance mask, one pass (segmentation_dataset.py:95-131)
__global__ void data_region_masks_kernel(const float* __restrict__ label, const void* __restrict__ inst, int inst_kind,
                                         const int* __restrict__ boxes, const float* __restrict__ fill,
                                         const int* __restrict__ inst_id, float* __restrict__ mask_in,
                                         float* __restrict__ obj_in, float* __restrict__ ctx_in,
                                         float* __restrict__ mask_out, float* __restrict__ obj_out,
                                         float* __restrict__ inst_mask, int B, int H, int W) {
  long long n = (long long)B * H * W;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    int x = (int)(i % W);
    long long r = i / W;
    int y = (int)(r % H);
    int b = (int)(r / H);
    const int* q = boxes + b * 8;  // (wmin, hmin, wmax, hmax) of the input window, then of the output window
    float mi = (q[3] > q[1] && q[2] > q[0] && y >= q[1] && y < q[3] && x >= q[0] && x < q[2]) ? 1.f : 0.f;
    float mo = (q[7] > q[5] && q[6] > q[4] && y >= q[5] && y < q[7] && x >= q[4] && x < q[6]) ? 1.f : 0.f;
    float l = label[i];
    mask_in[i] = mi;
    obj_in[i] = mi * l;
    ctx_in[i] = (1.f - mi) * l + mi * fill[b];
    mask_out[i] = mo;
    obj_out[i] = mo * l;
    if (inst_mask) {
      float m = 0.f;
      if (inst_id[b * 2]) {  // [b][0] = "an instance was selected", [b][1] = its id
        if (inst_kind == 0) m = ((const float*)inst)[i] == (float)inst_id[b * 2 + 1] ? 1.f : 0.f;
        else m = ((const int*)inst)[i] == inst_id[b * 2 + 1] ? 1.f : 0.f;
      }
      inst_mask[i] = m;
    }
  }
}

static inline dim3 data_grid(long long n) {
  long long g = (n + 255) / 256;
  return dim3((unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g)));
}

}  // namespace him

using namespace him;
#define ST ((hipStream_t)stream)

extern "C" {

int him_data_nearest(const void* base, const long long* off, const int* pitch, const int* xtab, const int* ytab,
                     int src_kind, void* dst, int dst_kind, int B, int H, int W, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0) return fail(HIM_E_INVALID, "data_nearest: bad shape");
  if (dst_kind < 0 || dst_kind > 3) return fail(HIM_E_INVALID, "data_nearest: dst_kind %d", dst_kind);
  dim3 g = data_grid((long long)B * H * W);
  const unsigned char* p = (const unsigned char*)base;
  switch (src_kind) {
    case 0: hipLaunchKernelGGL(data_nearest_kernel<unsigned char>, g, dim3(256), 0, ST, p, off, pitch, xtab, ytab, dst,
                               dst_kind, B, H, W); break;
    case 1: hipLaunchKernelGGL(data_nearest_kernel<unsigned short>, g, dim3(256), 0, ST, p, off, pitch, xtab, ytab, dst,
                               dst_kind, B, H, W); break;
    case 2: hipLaunchKernelGGL(data_nearest_kernel<int>, g, dim3(256), 0, ST, p, off, pitch, xtab, ytab, dst, dst_kind,
                               B, H, W); break;
    default: return fail(HIM_E_INVALID, "data_nearest: src_kind %d", src_kind);
  }
  return check_launch("data_nearest");
}

int him_data_bicubic_h(const void* base, const long long* off, const int* pitch, const int* rows, const int* first,
                       const int* count, const int* weights, int ksize, unsigned char* tmp, int maxrows, int B, int W,
                       void* stream) {
  if (B <= 0 || W <= 0 || maxrows <= 0 || ksize <= 0 || maxrows > 65535 || B > 65535)
    return fail(HIM_E_INVALID, "data_bicubic_h: bad shape");
  hipLaunchKernelGGL(data_bicubic_h_kernel, dim3((W + 255) / 256, maxrows, B), dim3(256), 0, ST,
                     (const unsigned char*)base, off, pitch, rows, first, count, weights, ksize, tmp, maxrows, B, W);
  return check_launch("data_bicubic_h");
}

int him_data_bicubic_v(const unsigned char* tmp, int maxrows, const int* first, const int* count, const int* weights,
                       int ksize, const int* flip, float* dst, int normalize, int B, int H, int W, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || ksize <= 0 || H > 65535 || B > 65535)
    return fail(HIM_E_INVALID, "data_bicubic_v: bad shape");
  hipLaunchKernelGGL(data_bicubic_v_kernel, dim3((W + 255) / 256, H, B), dim3(256), 0, ST, tmp, maxrows, first, count,
                     weights, ksize, flip, dst, normalize, B, H, W);
  return check_launch("data_bicubic_v");
}

int him_data_region_masks(const float* label, const void* inst, int inst_kind, const int* boxes, const float* fill,
                          const int* inst_id, float* mask_in, float* obj_in, float* ctx_in, float* mask_out,
                          float* obj_out, float* inst_mask, int B, int H, int W, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0) return fail(HIM_E_INVALID, "data_region_masks: bad shape");
  if (inst_mask && (!inst || !inst_id)) return fail(HIM_E_INVALID, "data_region_masks: instance mask without a map");
  hipLaunchKernelGGL(data_region_masks_kernel, data_grid((long long)B * H * W), dim3(256), 0, ST, label, inst,
                     inst_kind, boxes, fill, inst_id, mask_in, obj_in, ctx_in, mask_out, obj_out, inst_mask, B, H, W);
  return check_launch("data_region_masks");
}

}  // extern "C"
