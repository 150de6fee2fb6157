// Conv2d / ConvTranspose2d forward, data-gradient and weight-gradient for gfx950 (MI355X).
//
// All three are implicit GEMMs on the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32
// (157 TFLOP/s peak, bit-for-bit an fmaf chain), 256-thread workgroups = 4 wave64, LDS-staged
// double-buffered tiles with the next K-step's global loads in flight during the MFMAs.
//
//   gconv  : D[m][n]  = sum_k A[m][k] * gather(src)[k][n]       n = (b, y, x) output positions
//            forward conv        : A = W (Cout x Cin*KH*KW), gather = im2col with zero/reflect pad
//            data gradient       : A = W regrouped per stride phase (Cin x Cout*taps),
//                                  gather = shifted dY; one launch covers the s*s output phases
//            transposed-conv fwd : identical to the data gradient of its adjoint conv (+bias+act)
//   wgrad  : dW[m][n'] = sum_{k=(b,oy,ox)} dY[m][k] * gather(x)[k][n']   with split-K slabs and a
//            fixed-order slab reduction (deterministic).
//
// Layout facts used below (cdna_hip_programming.md section 3): for mfma_f32_32x32x2f32 lane l holds
// A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; accumulator register r of lane l is
// D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31]  -> putting the spatial index on j makes every accumulator
// register a 128-byte coalesced NCHW row segment.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "him_common.h"

namespace him {

#include "him_gconv_fast.inc"
#include "him_wino_fused.inc"
#include "him_wino_fused2.inc"
#include "him_bgemm.inc"

// ---- weight regrouping for the fast path: out[m][cb][jh][jw][c16] = W[base + m*sm + (16cb+c16)*sc + jh*sh + jw*sw]
struct WT2Phase {
  float* out;
  int JH, JW;
  long long sh, sw, base;
  long long total;  // M*JH*JW*C2p
};
struct WT2P {
  const float* W;
  int M, C2, C2p;
  long long sm, sc;
  WT2Phase ph[4];
};
__global__ void wtrans2_kernel(const WT2P p) {
  // out[m][cb][jh][jw][c16]  (channel block outer, taps inner, 16 channels innermost), zero for padded channels
  const WT2Phase& q = p.ph[blockIdx.y];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < q.total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c16 = (int)(i & 15);
    long long r = i >> 4;
    const int jw = (int)(r % q.JW);
    r /= q.JW;
    const int jh = (int)(r % q.JH);
    r /= q.JH;
    const int CB = p.C2p >> 4;
    const int cb = (int)(r % CB);
    const long long m = r / CB;
    const int c2 = cb * 16 + c16;
    q.out[i] = c2 < p.C2 ? p.W[q.base + m * p.sm + c2 * p.sc + jh * q.sh + jw * q.sw] : 0.f;
  }
}

// Contiguous-run specialisations of the regrouping (stride-1 convs = 95 % of the weight bytes): both read and
// write 16*KK-float (>= 576 B) contiguous runs through an LDS tile instead of 4-byte strided gathers.
//   forward  : out[m][cb][tap][c16] = W[m][16cb + c16][tap]           one wave per (m, cb)
__global__ __launch_bounds__(64) void wt_fwd_kernel(const float* __restrict__ W, float* __restrict__ out, int M, int C2,
                                                    int CB, int KK) {
  __shared__ float sm[16 * 64];
  const int cb = blockIdx.x, m = blockIdx.y;
  const int n = 16 * KK;
  const int cvalid = min(16, C2 - cb * 16);
  const float* __restrict__ src = W + ((size_t)m * C2 + (size_t)cb * 16) * KK;
  for (int i = threadIdx.x; i < n; i += 64) sm[i] = i < cvalid * KK ? src[i] : 0.f;  // [c16][tap]
  __syncthreads();
  float* __restrict__ dst = out + ((size_t)m * CB + cb) * n;
  for (int i = threadIdx.x; i < n; i += 64) {
    const int tap = i >> 4, c16 = i & 15;
    dst[i] = sm[c16 * KK + tap];
  }
}
//   data grad: out[ci][cob][tap][co16] = W[16cob + co16][ci][tap]      one workgroup per (16 ci, cob)
__global__ __launch_bounds__(256) void wt_dgrad_kernel(const float* __restrict__ W, float* __restrict__ out, int Co,
                                                       int Ci, int COB, int KK) {
  __shared__ float sm[16][16 * 49 + 1];
  const int cib = blockIdx.x, cob = blockIdx.y;
  const int civalid = min(16, Ci - cib * 16), covalid = min(16, Co - cob * 16);
  const int n = 16 * KK;  // floats per co row in this tile
  for (int r = threadIdx.x >> 4; r < 16; r += 16) {
    const float* __restrict__ src = W + (((size_t)cob * 16 + r) * Ci + (size_t)cib * 16) * KK;
    for (int i = threadIdx.x & 15; i < n; i += 16) sm[r][i] = (r < covalid && i < civalid * KK) ? src[i] : 0.f;
  }
  __syncthreads();
  for (int ci = threadIdx.x >> 4; ci < civalid; ci += 16) {
    float* __restrict__ dst = out + (((size_t)cib * 16 + ci) * COB + cob) * n;
    for (int i = threadIdx.x & 15; i < n; i += 16) {  // i = tap*16 + co16; lanes of a 16-group write 64 B runs
      const int tap = i >> 4, co16 = i & 15;
      dst[i] = sm[co16][ci * KK + tap];
    }
  }
}

static bool use_fast(const HimAlgo& a, int M, int C2) { return !algo_off(a, HIM_ALGO_GENERIC_CONV) && M > 4 && C2 >= 16; }
static int pad16(int c) { return (c + 15) / 16 * 16; }

#include "him_conv_direct.inc"

template <int MM, int TJ>
static void launch_small_cfg(const GConvP& p, long long maxN, hipStream_t st) {
  const bool split = maxN < 256 * 512 && p.C2 >= 64;  // too few positions to fill 256 CUs: split the channels
  if (split) {
    dim3 grid(cdiv(maxN, 64), p.small_nsplit > 1 ? p.small_nsplit : 1, p.nphase);
    if (p.pad_mode == HIM_PAD_REFLECT)
      hipLaunchKernelGGL((gconv_small_kernel<MM, TJ, true, 4>), grid, dim3(256), 0, st, p);
    else
      hipLaunchKernelGGL((gconv_small_kernel<MM, TJ, false, 4>), grid, dim3(256), 0, st, p);
  } else {
    dim3 grid(cdiv(maxN, 256), 1, p.nphase);
    if (p.pad_mode == HIM_PAD_REFLECT)
      hipLaunchKernelGGL((gconv_small_kernel<MM, TJ, true, 1>), grid, dim3(256), 0, st, p);
    else
      hipLaunchKernelGGL((gconv_small_kernel<MM, TJ, false, 1>), grid, dim3(256), 0, st, p);
  }
}

// returns true when the tiny-M path took the launch
static bool launch_gconv_small(const HimAlgo& a, const GConvP& p, long long maxN, hipStream_t st) {
  if (p.M > 4) return false;
  bool same = true;
  for (int i = 0; i < p.nphase; ++i) same = same && p.ph[i].JH == p.ph[0].JH && p.ph[i].JW == p.ph[0].JW;
  for (int i = 0; i < p.nphase; ++i)
    if (p.ph[i].JH > 8 || p.ph[i].JW > 8) return false;
  const int tj = (same && p.ph[0].JH == p.ph[0].JW) ? p.ph[0].JH : 0;
  if (fewout_tiled_ok(a, p)) {
    switch (p.M * 10 + tj) {
      case 23: launch_fewout_tiled<2, 3>(p, st); return true;
      case 33: launch_fewout_tiled<3, 3>(p, st); return true;
      case 43: launch_fewout_tiled<4, 3>(p, st); return true;
      case 27: launch_fewout_tiled<2, 7>(p, st); return true;
      case 37: launch_fewout_tiled<3, 7>(p, st); return true;
      case 47: launch_fewout_tiled<4, 7>(p, st); return true;
    }
  }
#define HIM_SMALL(MMv)                                      \
  case MMv:                                                 \
    if (tj == 7) launch_small_cfg<MMv, 7>(p, maxN, st);      \
    else if (tj == 4) launch_small_cfg<MMv, 4>(p, maxN, st); \
    else if (tj == 3) launch_small_cfg<MMv, 3>(p, maxN, st); \
    else if (tj == 2) launch_small_cfg<MMv, 2>(p, maxN, st); \
    else launch_small_cfg<MMv, 0>(p, maxN, st);              \
    break;
  switch (p.M) {
    HIM_SMALL(1)
    HIM_SMALL(2)
    HIM_SMALL(3)
    HIM_SMALL(4)
  }
#undef HIM_SMALL
  if (p.small_nsplit > 1 && maxN < 256 * 512 && p.C2 >= 64) {
    const long long total = (long long)p.M * maxN;
    hipLaunchKernelGGL(gconv_small_finish_kernel, dim3(std::min<long long>(cdiv(total, 256), 4096)), dim3(256), 0, st, p);
  }
  return true;
}

template <int WM, int WN, int TM, int TN>
static void launch_gconv_cfg(const GConvP& p, dim3 grid, hipStream_t st) {
  if (p.pad_mode == HIM_PAD_REFLECT)
    hipLaunchKernelGGL((gconv_kernel<WM, WN, TM, TN, true>), grid, dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((gconv_kernel<WM, WN, TM, TN, false>), grid, dim3(256), 0, st, p);
}

static int launch_gconv(const HimAlgo& a, const GConvP& p, hipStream_t st) {
  long long maxN = 0;
  for (int i = 0; i < p.nphase; ++i) {
    long long n = (long long)p.B * p.ph[i].NA * p.ph[i].NC;
    if (n > maxN) maxN = n;
  }
  if (maxN == 0 || p.M <= 0) return HIM_OK;
  if (launch_gconv_small(a, p, maxN, st)) return check_launch("gconv_small");
  if (fewin_tiled_ok(a, p)) {
    launch_fewin_tiled(p, st);
    return check_launch("gconv_fewin_tiled");
  }
  if (p.fast) {
    // The fast kernel gathers through a buffer resource: 31-bit byte offsets.  A larger source tensor (C2 at >= 64 images
    // per GPU on the 64-channel full-resolution planes) is launched in batch slices, each below 2 GiB -- the images of a
    // batch are independent in the forward and the data gradient.
    const unsigned long long img_bytes = (unsigned long long)p.C2 * p.SH * p.SW * 4ull;
    if ((unsigned long long)p.B * img_bytes >= (1ull << 31)) {
      if (p.wbatch || p.ksplit > 1 || img_bytes >= (1ull << 31))
        return fail(HIM_E_UNSUPPORTED, "conv: source tensor of %llu bytes >= 2 GiB cannot be sliced along the batch",
                    (unsigned long long)p.B * img_bytes);
      const int per = (int)(((1ull << 31) - 1) / img_bytes);
      for (int b0 = 0; b0 < p.B; b0 += per) {
        GConvP q = p;
        q.B = std::min(per, p.B - b0);
        q.src = p.src + (size_t)b0 * p.C2 * p.SH * p.SW;
        q.dst = p.dst + (size_t)b0 * p.M * p.DH * p.DW;
        const int rcq = launch_gconv(a, q, st);
        if (rcq) return rcq;
      }
      return HIM_OK;
    }
    // HimAlgo::tile_wb (batched Winograd GEMMs) / tile_nb (direct-form convs); transposed weight tiles: 64x128 only
    const int tile_override = p.atrans ? HIM_TILE_64x128 : (p.wbatch ? a.tile_wb : a.tile_nb);
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    if (p.M <= 64) {   // (64x64 tiles here: no change of the step, round 3)
      dim3 grid(cdiv(maxN, 128) * cdiv(p.M, 64), ks, p.nphase);
      launch_fast_cfg<2, 2, 1, 2>(p, grid, st);
    } else {
      // Tile shape (HimAlgo::tile_wb / tile_nb).  Alone, 128x128 tiles
      // are the fastest (the batched GEMM of the ResnetBlocks: 0.321 ms vs 0.326 ms for 64x128, 0.358 ms for 64x64) --
      // but the training step runs two to four streams, and the 64x128 workgroup (30 KB LDS, ~100 registers: 4-5 per CU
      // instead of 3, twice as many of half the length) shares the CUs with the other streams' kernels better: round 3
      // measured 60.7 ms per step against 61.9 (128x64: 61.0; 64x64: 63.2; `gpurun_out/r03h`).
      const long long tiles128 = (long long)cdiv(maxN, 128) * cdiv(p.M, 128) * p.nphase;
      const bool big = tile_override == HIM_TILE_128x128 ||
                       (tile_override == HIM_TILE_MIXED && (tiles128 >= 512 || (tiles128 >= 200 && tiles128 <= 256)));
      const long long plane0 = (long long)p.ph[0].NA * p.ph[0].NC;
      const long long tiles256 = (maxN / 256) * cdiv(p.M, 128);
      if (tile_override == HIM_TILE_128x256 && p.wbatch && ks == 1 && p.nphase == 1 && plane0 % 256 == 0 && tiles256 % 512 == 0) {
        // experiment (HIM_TILE_128x256): 128x256 tiles for the batched Winograd GEMM, two workgroups per CU.  Alone it
        // matches / beats the 128x128 tiling (weight-gradient GEMM 0.53 -> 0.42 ms) but its 232 VGPRs + 60 KB LDS stop
        // it from sharing a CU with the other stream's kernels: the full step fell from 98 to 69 images/s.
        dim3 grid((unsigned)tiles256, 1, 1);
        launch_fast_cfg<2, 2, 2, 4>(p, grid, st);
      } else if (tile_override == HIM_TILE_64x64) {
        dim3 grid(cdiv(maxN, 64) * cdiv(p.M, 64), ks, p.nphase);
        launch_fast_cfg<2, 2, 1, 1>(p, grid, st);  // 64x64 tiles
      } else if (tile_override == HIM_TILE_128x128_8W) {
        dim3 grid(cdiv(maxN, 128) * cdiv(p.M, 128), ks, p.nphase);
        launch_fast_cfg<2, 4, 2, 1>(p, grid, st);  // 8 waves per 128x128 tile (measured: lockstep, no better than 4)
      } else if (big || (tile_override == HIM_TILE_MIXED && ks > 1)) {
        dim3 grid(cdiv(maxN, 128) * cdiv(p.M, 128), ks, p.nphase);
        launch_fast_cfg<2, 2, 2, 2>(p, grid, st);
      } else if (tile_override == HIM_TILE_128x64 || tile_override == HIM_TILE_MIXED) {
        dim3 grid(cdiv(maxN, 64) * cdiv(p.M, 128), ks, p.nphase);
        launch_fast_cfg<2, 2, 2, 1>(p, grid, st);
      } else {   // HIM_TILE_DEFAULT / HIM_TILE_64x128
        dim3 grid(cdiv(maxN, 128) * cdiv(p.M, 64), ks, p.nphase);
        launch_fast_cfg<2, 2, 1, 2>(p, grid, st);
      }
    }
    int rcf = check_launch("gconv_fast");
    if (rcf || ks == 1 || p.kno_finish) return rcf;
    const long long n = (long long)p.B * p.M * p.DH * p.DW;
    hipLaunchKernelGGL(gconv_splitk_finish_kernel, dim3(std::min<long long>(cdiv(n, 256), 8192)), dim3(256), 0, st,
                       (const float*)p.kpart, p.dst, p.bias, n, ks, p.M, p.DH * p.DW, p.act, p.slope);
    return check_launch("gconv_splitk_finish");
  }
  if (p.M <= 32) {
    dim3 grid(cdiv(maxN, 256), cdiv(p.M, 32), p.nphase);
    launch_gconv_cfg<1, 4, 1, 2>(p, grid, st);
  } else if (p.M <= 64) {
    dim3 grid(cdiv(maxN, 128), cdiv(p.M, 64), p.nphase);
    launch_gconv_cfg<1, 4, 2, 1>(p, grid, st);
  } else {
    const long long tiles128 = (long long)cdiv(maxN, 128) * cdiv(p.M, 128) * p.nphase;
    if (tiles128 >= 768) {
      dim3 grid(cdiv(maxN, 128), cdiv(p.M, 128), p.nphase);
      launch_gconv_cfg<2, 2, 2, 2>(p, grid, st);
    } else {
      dim3 grid(cdiv(maxN, 64), cdiv(p.M, 128), p.nphase);
      launch_gconv_cfg<2, 2, 2, 1>(p, grid, st);
    }
  }
  return check_launch("gconv");
}

// ---- weight regrouping for the data gradient: Wt_phase[ci][(co, jh, jw)] = W[co][ci][ph+s*jh][pw+s*jw]
struct WTransP {
  const float* W;  // [Co][Ci][KH][KW]
  float* Wt;
  int Co, Ci, KH, KW, s;
  int nphase;
  int ph[4], pw[4], JH[4], JW[4];
  long long off[5];  // element offsets of each phase block, off[nphase] = total
};

__global__ void wtrans_kernel(const WTransP p) {
  const long long total = p.off[p.nphase];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int q = 0;
    while (q + 1 < p.nphase && i >= p.off[q + 1]) ++q;
    const long long li = i - p.off[q];
    const int taps = p.JH[q] * p.JW[q];
    const int Kq = p.Co * taps;
    const int ci = (int)(li / Kq);
    const int k = (int)(li - (long long)ci * Kq);
    const int co = k / taps;
    const int r = k - co * taps;
    const int jh = r / p.JW[q], jw = r - jh * p.JW[q];
    const int kh = p.ph[q] + p.s * jh, kw = p.pw[q] + p.s * jw;
    p.Wt[i] = p.W[(((size_t)co * p.Ci + ci) * p.KH + kh) * p.KW + kw];
  }
}

// Fills the phase table of a data-gradient launch for a conv with (KH,KW,stride s,pad) whose input
// grid is IH x IW and whose output-gradient grid is OH x OW.  Returns floats needed for Wt.
static long long setup_dgrad(GConvP& g, WTransP& wt, int Cout, int Cin, int KH, int KW, int s, int pad,
                             int IH, int IW, float* Wt) {
  int q = 0;
  long long off = 0;
  for (int ph = 0; ph < s; ++ph) {
    for (int pw = 0; pw < s; ++pw) {
      const int JH = ph < KH ? (KH - ph + s - 1) / s : 0;
      const int JW = pw < KW ? (KW - pw + s - 1) / s : 0;
      const int ih0 = ((ph - pad) % s + s) % s, iw0 = ((pw - pad) % s + s) % s;
      const int NA = ih0 < IH ? (IH - ih0 + s - 1) / s : 0;
      const int NC = iw0 < IW ? (IW - iw0 + s - 1) / s : 0;
      if (JH == 0 || JW == 0 || NA == 0 || NC == 0) {  // kernel smaller than the stride (conv1x1 s2): these input
        if (NA > 0 && NC > 0) g.kno_finish |= 2;        // positions receive no gradient -> the caller zero-fills
        continue;
      }
      GPhase& P = g.ph[q];
      P.A = Wt + off;
      P.JH = JH;
      P.JW = JW;
      P.K = Cout * JH * JW;
      P.fJHJW = make_fastdiv((uint32_t)(JH * JW > 0 ? JH * JW : 1));
      P.fJW = make_fastdiv((uint32_t)(JW > 0 ? JW : 1));
      P.NA = NA;
      P.NC = NC;
      P.oy0 = ih0;
      P.ox0 = iw0;
      P.offy = (ih0 + pad - ph) / s;
      P.offx = (iw0 + pad - pw) / s;
      wt.ph[q] = ph;
      wt.pw[q] = pw;
      wt.JH[q] = JH;
      wt.JW[q] = JW;
      wt.off[q] = off;
      off += (long long)Cin * Cout * JH * JW;
      ++q;
    }
  }
  wt.off[q] = off;
  wt.nphase = g.nphase = q;
  wt.Co = Cout;
  wt.Ci = Cin;
  wt.KH = KH;
  wt.KW = KW;
  wt.s = s;
  wt.Wt = Wt;
  g.oys = g.oxs = s;
  g.sy = g.sx = 1;
  g.dy = g.dx = -1;
  g.pad_mode = HIM_PAD_ZERO;
  return off;
}

// ---- reflection-pad backward: dx[y][x] = sum of dpad over the padded positions that mirror onto (y,x).
// grid = (ceil(H*W/256), planes): no 64-bit index arithmetic; nslab > 1 sums split-K partial slabs in fixed order.
__global__ __launch_bounds__(256) void reflect_fold_kernel(const float* __restrict__ dpad, float* __restrict__ dx,
                                                           int planes, int H, int W, int p, int nslab,
                                                           size_t slab_stride) {
  const int PH = H + 2 * p, PW = W + 2 * p;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * W) return;
  const int y = i / W, x = i - y * W;
  const int pl = blockIdx.y;
  int ys[3], xs[3], ny = 0, nx = 0;
  ys[ny++] = y + p;
  if (y >= 1 && y <= p) ys[ny++] = p - y;
  if (y >= H - 1 - p && y <= H - 2) ys[ny++] = p + 2 * (H - 1) - y;
  xs[nx++] = x + p;
  if (x >= 1 && x <= p) xs[nx++] = p - x;
  if (x >= W - 1 - p && x <= W - 2) xs[nx++] = p + 2 * (W - 1) - x;
  const float* __restrict__ base = dpad + (size_t)pl * PH * PW;
  float s = 0.f;
  for (int z = 0; z < nslab; ++z) {
    const float* __restrict__ bz = base + (size_t)z * slab_stride;
    float sz = bz[ys[0] * PW + xs[0]];  // the direct position: every thread, coalesced
    if (ny + nx > 2) {
      for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b)
          if (a + b) sz += bz[ys[a] * PW + xs[b]];
    }
    s += sz;
  }
  dx[(size_t)pl * H * W + i] = s;
}

// ---- border-extended gradient for the folded reflect-pad-1 data gradient.  ReflectionPad2d(1) + conv3x3:
// dx[y][x] = sum_t W_t^T sum_{(o_y,o_x) in R_th(y) x R_tw(x)} dy[o_y][o_x] with R_th(y) = {y+1-th} (if inside), plus
// {0} when (y,th) = (1,0) and {H-1} when (y,th) = (H-2,2): the mirrored pad row feeds the neighbour of the border.
// The pair sums are materialised once as two extra rows / columns so that the MFMA kernel still gathers ONE element:
// ext[r][c], r in [0,H+2): rows 0..H-1 = dy, row H = dy[0]+dy[2], row H+1 = dy[H-3]+dy[H-1]; columns likewise.
__global__ __launch_bounds__(256) void reflect_extend_kernel(const float* __restrict__ dy, float* __restrict__ ext,
                                                             int H, int W) {
  const int EH = H + 2, EW = W + 2;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= EH * EW) return;
  const int r = i / EW, c = i - r * EW;
  const float* __restrict__ src = dy + (size_t)blockIdx.y * H * W;
  const int r0 = r < H ? r : (r == H ? 0 : H - 3), r1 = r < H ? -1 : (r == H ? 2 : H - 1);
  const int c0 = c < W ? c : (c == W ? 0 : W - 3), c1 = c < W ? -1 : (c == W ? 2 : W - 1);
  float v = src[r0 * W + c0];
  if (c1 >= 0) v += src[r0 * W + c1];
  if (r1 >= 0) {
    float u = src[r1 * W + c0];
    if (c1 >= 0) u += src[r1 * W + c1];
    v += u;
  }
  ext[(size_t)blockIdx.y * EH * EW + i] = v;
}


#include "him_conv_wino.inc"

// ---- Winograd host side -------------------------------------------------------------------------------------------
// wide 3x3 stride-1 pad-1 layers only (HimAlgo::wino_min_c, default 256 since round 4): below that the x4 transform traffic eats
// the 2.25x multiply saving and the fused single-launch kernel takes over
static bool wino_shape_ok(const HimAlgo& a, int Cout, int Cin, int KH, int KW, int stride, int pad, int H, int W) {
  const int mc = algo_wino_min_c(a);
  return mc > 0 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && Cin >= mc && Cout >= mc && (Cin % 16) == 0 &&
         (Cout % 16) == 0 && H >= 2 && W >= 2 && use_fast(a, Cout, Cin);
}
// Threads per workgroup of the Winograd transform kernels (one tile / channel per thread).  64 = one wave: inside the
// multi-stream step the single-wave workgroups find a free slot next to the MFMA kernels sooner (58.5 vs 58.8 ms per step
// for the input / output transforms alone, 128: 58.65; results are bit-identical); HimAlgo::wino_tblock overrides.
static int wino_tblock(const HimAlgo& a) { return algo_tblock(a); }
static bool wino_wgrad_ok(const HimAlgo& a, int M, int C, int KH, int KW, int stride, int pad, int H, int W) {
  return wino_shape_ok(a, M, C, KH, KW, stride, pad, H, W) && (M % 128) == 0 && (C % 128) == 0;
}
static size_t wino_conv_floats(int B, int Csrc, int Mout, int OH, int OW) {
  const WinoGeom g = wino_geom(B, Csrc, 1, 1, OH, OW, 1);
  return (size_t)16 * (Csrc + Mout) * g.Tp;
}
static size_t wino_wgrad_floats(int B, int M, int C, int OH, int OW) {
  const WinoGeom g = wino_geom(B, C, 1, 1, OH, OW, 1);
  return (size_t)16 * (C + M) * g.Tp + (size_t)16 * M * C;
}
// Cm[z][m][n] = sum_k A[z][m][k] * Bm[z][k][n] for the 16 transform positions z, on the fast MFMA conv kernel: a 1x1
// convolution over 16 "images" [K][1][N] with per-image weight panels.  K % 16 == 0, N % 128 == 0.
// atrans: A is a FORWARD panel U[z][K][M] of the layer whose data gradient this is; the kernel reads it transposed with
// the positions mirrored (GConvP::atrans) -- no second, flipped panel per weight.
static int wino_batched_gemm(const HimAlgo& a, const float* A, const float* Bm, float* Cm, int M, int K, int N, hipStream_t st,
                             bool atrans = false) {
  // Round 4: the dedicated GEMM kernel with LDS-DMA operand loads (him_bgemm.inc) takes every launch that fills the chip
  // with its 128x128 tiles (C2: 1024 workgroups = 2 per CU and XCD-resident batch entries); small planes (config C1: 128
  // tiles) stay on the conv kernel's 64x128 tiles.
  if (!algo_off(a, HIM_ALGO_NO_BGEMM) && bgemm_shape_ok(M, K, N, 16) && 16 * (M / 128) * (N / 128) >= 512)
    return launch_bgemm(A, Bm, Cm, M, K, N, 16, 4, atrans ? 1 : 0, 1, atrans, st);
  GConvP g;
  memset(&g, 0, sizeof(g));
  g.atrans = atrans ? 1 : 0;
  g.src = Bm;
  g.dst = Cm;
  g.M = M;
  g.C2 = K;
  g.B = 16;
  g.SH = 1;
  g.SW = N;
  g.DH = 1;
  g.DW = N;
  g.oys = g.oxs = g.sy = g.sx = g.dy = g.dx = 1;
  g.pad_mode = HIM_PAD_ZERO;
  g.act = HIM_ACT_NONE;
  g.nphase = 1;
  g.fast = 1;
  g.wbatch = M * K;
  GPhase& P = g.ph[0];
  P.A = A;
  P.At = A;
  P.C2p = K;
  P.K = K;
  P.JH = P.JW = 1;
  P.fJHJW = make_fastdiv(1);
  P.fJW = make_fastdiv(1);
  P.NA = 1;
  P.NC = N;
  return launch_gconv(a, g, st);
}
// dst[B][Mout][OH][OW] = act(winograd-conv(src[B][Csrc][H][W], U) + bias); ws holds V and Mo
static int run_wino_conv(const HimAlgo& a, int B, int Csrc, int H, int W, int Mout, int OH, int OW, int po, bool reflect,
                         const float* src, const float* U, const float* bias, int act, float slope, float* dst,
                         float* ws, hipStream_t st, bool fold = false, bool atrans = false, float* keep = nullptr) {
  WinoGeom gi = wino_geom(B, Csrc, H, W, OH, OW, po);
  gi.fold = fold ? 1 : 0;
  float* V = keep ? keep : ws;     // keep: the caller's buffer for the transformed input (him_conv2d_fwd_panel_keep)
  float* Mo = ws + (size_t)16 * Csrc * gi.Tp;
  const int tb = wino_tblock(a);
  const dim3 gin(cdiv(gi.Tp, tb), Csrc);
  if (reflect) hipLaunchKernelGGL((wino_input_kernel<true>), gin, dim3(tb), 0, st, src, V, gi);
  else hipLaunchKernelGGL((wino_input_kernel<false>), gin, dim3(tb), 0, st, src, V, gi);
  int rc = check_launch("wino_input");
  if (rc) return rc;
  rc = wino_batched_gemm(a, U, V, Mo, Mout, Csrc, gi.Tp, st, atrans);
  if (rc) return rc;
  WinoGeom go = gi;
  go.C = Mout;
  hipLaunchKernelGGL(wino_output_kernel, dim3(cdiv(gi.Tp, tb), Mout), dim3(tb), 0, st, (const float*)Mo, dst, bias, go,
                     act, slope);
  return check_launch("wino_output");
}

#include "him_conv_wino4.inc"

#include "him_conv_wgrad.inc"

#include "him_wgrad_fewch.inc"

static const int SMALL_WIN_SLOTS = 256;
static bool small_win_ok(const HimAlgo& a, int M, int KH, int KW, int stride, int pad, int H, int W, int OH, int OW) {
  return !algo_off(a, HIM_ALGO_NO_SMALL_WIN) && M <= 4 && KH == KW && (KH == 3 || KH == 5 || KH == 7) && stride == 1 && pad == KH / 2 && OH == H && OW == W &&
         H > pad && W > pad;
}

static int small_wgrad_slices(int C, int Kdim) {
  int s = (2048 + C - 1) / C;
  const int maxs = cdiv(Kdim, 256 * 8);
  if (s > maxs) s = maxs;
  return s < 1 ? 1 : s;
}
static bool small_wgrad_ok(int M, int KH, int KW) { return M <= 4 && KH == KW && (KH == 7 || KH == 4 || KH == 3); }
static size_t wgrad_slab_bytes(const HimAlgo& a, int M, int C, int KH, int KW, int Kdim, size_t wino_floats = 0) {
  const int Np = C * KH * KW;
  size_t slabs;
  if (wino_floats) return ((wino_floats * sizeof(float) + 255) / 256) * 256;
  const bool fewch_shape = !fewch_off(a) && KH == KW && (KH == 5 || KH == 7) &&
                           ((M <= 4 && C >= 32 && (C % 32) == 0) || (C <= 4 && M >= 32 && (M % 32) == 0));
  if (fewch_shape) {   // upper bound: the runner re-checks stride / padding / plane (else the kernels below, which need less)
    slabs = fewch_ws_floats(M <= 4 ? C : M, KH) * sizeof(float);
    if (small_wgrad_ok(M, KH, KW) || (M <= 4 && KH == 5))
      slabs = std::max(slabs, (size_t)std::max(small_wgrad_slices(C, Kdim), SMALL_WIN_SLOTS) * M * Np * sizeof(float));
    int BM2, BN2;
    wgrad_tile(M, &BM2, &BN2);
    const int s2 = wgrad_splits(M, Np, Kdim, BM2, BN2);
    if (s2 > 1) slabs = std::max(slabs, (size_t)s2 * M * Np * sizeof(float));
  } else if (small_wgrad_ok(M, KH, KW) || (M <= 4 && KH == KW && KH == 5)) {
    slabs = (size_t)std::max(small_wgrad_slices(C, Kdim), SMALL_WIN_SLOTS) * M * Np * sizeof(float);
  } else if (wgrad_fast_ok(a, M, C, 1, 4)) {  /* upper bound; the runner re-checks OH*OW */
    int BM, BN, sp;
    wgrad_fast_cfg(a, M, C, Kdim, KH * KW, &BM, &BN, &sp);
    slabs = (size_t)sp * M * Np * sizeof(float);
    int BM2, BN2;
    wgrad_tile(M, &BM2, &BN2);
    const int s2 = wgrad_splits(M, Np, Kdim, BM2, BN2);
    const size_t alt = s2 > 1 ? (size_t)s2 * M * Np * sizeof(float) : 0;
    if (alt > slabs) slabs = alt;
  } else {
    int BM, BN;
    wgrad_tile(M, &BM, &BN);
    const int s = wgrad_splits(M, Np, Kdim, BM, BN);
    slabs = s > 1 ? (size_t)s * M * Np * sizeof(float) : 0;
  }
  return ((slabs + 255) / 256) * 256;
}
static size_t wgrad_ws_bytes(const HimAlgo& a, int M, int C, int KH, int KW, int Kdim, int biasC, size_t wino_floats = 0) {
  return wgrad_slab_bytes(a, M, C, KH, KW, Kdim, wino_floats) + bias_ws_bytes(biasC);
}
static size_t conv_wino_wgrad_floats(const HimConv2d* d) {
  return wino_wgrad_ok(d->algo, d->Cout, d->Cin, d->KH, d->KW, d->stride, d->pad, d->H, d->W)
             ? wino_wgrad_floats(d->B, d->Cout, d->Cin, d->OH, d->OW)
             : 0;
}

// generic weight gradient: dW[M][C*KH*KW] from dy[B][M][OH][OW] and x[B][C][H][W]
static int run_wgrad(const HimAlgo& a, const float* dy, const float* x, float* dw, int M, int C, int B, int H, int W, int OH,
                     int OW, int KH, int KW, int stride, int pad, int pad_mode, int accumulate, void* ws,
                     size_t ws_bytes, hipStream_t st, const float* kept_v = nullptr) {
  WGradP p;
  p.dy = dy;
  p.x = x;
  p.M = M;
  p.C = C;
  p.B = B;
  p.H = H;
  p.W = W;
  p.OH = OH;
  p.OW = OW;
  p.KH = KH;
  p.KW = KW;
  p.stride = stride;
  p.pad = pad;
  p.pad_mode = pad_mode;
  p.Np = C * KH * KW;
  p.Kdim = B * OH * OW;
  p.fKK = make_fastdiv((uint32_t)(KH * KW));
  p.fKW = make_fastdiv((uint32_t)KW);
  p.fOW = make_fastdiv((uint32_t)OW);
  if (wino_wgrad_ok(a, M, C, KH, KW, stride, pad, H, W) && OH == H && OW == W) {
    // dU = dM x V^T per Winograd position (batched NT GEMM), then dw (+)= G^T dU G
    const size_t need = wino_wgrad_floats(B, M, C, OH, OW) * sizeof(float);
    if (ws_bytes < need || !ws) return fail(HIM_E_WORKSPACE, "wgrad needs %zu ws bytes, got %zu", need, ws_bytes);
    const WinoGeom gx = wino_geom(B, C, H, W, OH, OW, 1);
    WinoGeom gd = gx;
    gd.C = M;
    float* Vt = (float*)ws;                          // [16][Tp][C]
    float* dM = Vt + (size_t)16 * C * gx.Tp;         // [16][M][Tp]
    float* dU = dM + (size_t)16 * M * gx.Tp;         // [16][M][C]
    const int tb = wino_tblock(a);
    int rcw;
    if (kept_v) {
      // the forward's transformed input V[16][C][Tp] IS the column operand (tiles = the reduction index, contiguous):
      // no transposed input transform, both GEMM operands K-contiguous (him_bgemm.inc, layouts (0, 0))
      hipLaunchKernelGGL(wino_dy_kernel, dim3(cdiv(gx.Tp, tb), M), dim3(tb), 0, st, dy, dM, gd);
      rcw = check_launch("wino_dy");
      if (rcw) return rcw;
      rcw = launch_bgemm(dM, kept_v, dU, M, gx.Tp, C, 16, 4, 0, 0, false, st);
    } else {
      const dim3 gin(cdiv(C, tb), gx.Tp);
      if (pad_mode == HIM_PAD_REFLECT) hipLaunchKernelGGL((wino_input_t_kernel<true>), gin, dim3(tb), 0, st, x, Vt, gx);
      else hipLaunchKernelGGL((wino_input_t_kernel<false>), gin, dim3(tb), 0, st, x, Vt, gx);
      hipLaunchKernelGGL(wino_dy_kernel, dim3(cdiv(gx.Tp, tb), M), dim3(tb), 0, st, dy, dM, gd);
      rcw = check_launch("wino_wgrad_transforms");
      if (rcw) return rcw;
      rcw = wino_batched_gemm(a, dM, Vt, dU, M, gx.Tp, C, st);
    }
    if (rcw) return rcw;
    hipLaunchKernelGGL(wino_wgrad_out_kernel, dim3(cdiv(C, 256), M), dim3(256), 0, st, (const float*)dU, dw, M, C,
                       accumulate);
    return check_launch("wino_wgrad_out");
  }
  if (fewch_head_ok(a, M, C, KH, KW, stride, pad, H, W, OH, OW))
    return run_wgrad_fewch(a, true, dy, x, dw, M, C, B, H, W, KH, pad, pad_mode, accumulate, ws, ws_bytes, st);
  if (fewch_stem_ok(a, M, C, KH, KW, stride, pad, H, W, OH, OW))
    return run_wgrad_fewch(a, false, dy, x, dw, M, C, B, H, W, KH, pad, pad_mode, accumulate, ws, ws_bytes, st);
  if (small_win_ok(a, M, KH, KW, stride, pad, H, W, OH, OW)) {
    const int nsx = cdiv(W, 64), rows_per = 64, nyc = cdiv(H, rows_per), ntasks = B * nsx * nyc;
    const int slots = std::min(ntasks, SMALL_WIN_SLOTS);
    const size_t need = (size_t)slots * M * p.Np * sizeof(float);
    if (ws_bytes < need || !ws) return fail(HIM_E_WORKSPACE, "wgrad needs %zu ws bytes, got %zu", need, ws_bytes);
    p.out = (float*)ws;
    const dim3 grid(slots, cdiv(C, 4)), block(256);
    const bool refl = pad_mode == HIM_PAD_REFLECT;
#define HIM_SWK(MMv, KSv)                                                                                         \
  if (M == MMv && KH == KSv) {                                                                                    \
    constexpr int TRv = (MMv * KSv * KSv > 90) ? (KSv + 1) / 2 : KSv;  /* band split once the accumulators pass ~90 */ \
    const dim3 gridz(grid.x, grid.y, cdiv(KSv, TRv));                                                             \
    if (refl) hipLaunchKernelGGL((wgrad_small_win_kernel<MMv, KSv, TRv, true>), gridz, block, 0, st, p, nsx, nyc, rows_per, ntasks); \
    else hipLaunchKernelGGL((wgrad_small_win_kernel<MMv, KSv, TRv, false>), gridz, block, 0, st, p, nsx, nyc, rows_per, ntasks);     \
  }
    HIM_SWK(1, 3) HIM_SWK(2, 3) HIM_SWK(3, 3) HIM_SWK(4, 3)
    HIM_SWK(1, 5) HIM_SWK(2, 5) HIM_SWK(3, 5) HIM_SWK(4, 5)
    HIM_SWK(1, 7) HIM_SWK(2, 7) HIM_SWK(3, 7) HIM_SWK(4, 7)
#undef HIM_SWK
    int rc0 = check_launch("wgrad_small_win");
    if (rc0) return rc0;
    const long long n = (long long)M * p.Np;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(std::min<long long>(cdiv(n, 256), 4096)), dim3(256), 0, st,
                       (const float*)ws, dw, n, slots, accumulate);
    return check_launch("slab_reduce");
  }
  if (small_wgrad_ok(M, KH, KW)) {
    const int slices = small_wgrad_slices(C, p.Kdim);
    const size_t need = (size_t)slices * M * p.Np * sizeof(float);
    if (ws_bytes < need || !ws) return fail(HIM_E_WORKSPACE, "wgrad needs %zu ws bytes, got %zu", need, ws_bytes);
    p.splits = slices;
    p.accumulate = 0;
    p.kchunk = cdiv(p.Kdim, slices);
    p.out = (float*)ws;
    dim3 grid(C, slices), block(256);
    const bool refl = pad_mode == HIM_PAD_REFLECT;
#define HIM_WS(MMv, TJv)                                                                             \
  if (M == MMv && KH == TJv) {                                                                       \
    if (refl) hipLaunchKernelGGL((wgrad_small_kernel<MMv, TJv, true>), grid, block, 0, st, p);        \
    else hipLaunchKernelGGL((wgrad_small_kernel<MMv, TJv, false>), grid, block, 0, st, p);            \
  }
    HIM_WS(1, 7) HIM_WS(2, 7) HIM_WS(3, 7) HIM_WS(4, 7)
    HIM_WS(1, 4) HIM_WS(2, 4) HIM_WS(3, 4) HIM_WS(4, 4)
    HIM_WS(1, 3) HIM_WS(2, 3) HIM_WS(3, 3) HIM_WS(4, 3)
#undef HIM_WS
    int rc0 = check_launch("wgrad_small");
    if (rc0) return rc0;
    const long long n = (long long)M * p.Np;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(std::min<long long>(cdiv(n, 256), 4096)), dim3(256), 0, st,
                       (const float*)ws, dw, n, slices, accumulate);
    return check_launch("slab_reduce");
  }
  // the fast kernel reads both operands through buffer resources (31-bit byte offsets)
  const bool fits31 = (unsigned long long)B * M * OH * OW * 4ull < (1ull << 31) &&
                      (unsigned long long)B * C * H * W * 4ull < (1ull << 31);
  if (wgrad_fast_ok(a, M, C, OH, OW) && fits31) {
    int fBM, fBN, fs;
    wgrad_fast_cfg(a, M, C, p.Kdim, KH * KW, &fBM, &fBN, &fs);
    const size_t need = (size_t)fs * M * p.Np * sizeof(float);
    if (ws_bytes < need || !ws) return fail(HIM_E_WORKSPACE, "wgrad needs %zu ws bytes, got %zu", need, ws_bytes);
    p.splits = fs;
    p.accumulate = 0;
    int kc = cdiv(p.Kdim, fs);
    p.kchunk = ((kc + 31) / 32) * 32;
    p.out = (float*)ws;
    dim3 grid(p.Np / fBN, cdiv(M, fBM), fs), block(256);
    const bool refl = pad_mode == HIM_PAD_REFLECT;
#define HIM_WF(TMv, TNv)                                                                              \
  {                                                                                                   \
    if ((OH * OW) % 4 == 0) {                                                                         \
      if (refl) hipLaunchKernelGGL((wgrad_fast_kernel<TMv, TNv, true, true>), grid, block, 0, st, p);  \
      else hipLaunchKernelGGL((wgrad_fast_kernel<TMv, TNv, false, true>), grid, block, 0, st, p);      \
    } else {                                                                                          \
      if (refl) hipLaunchKernelGGL((wgrad_fast_kernel<TMv, TNv, true, false>), grid, block, 0, st, p); \
      else hipLaunchKernelGGL((wgrad_fast_kernel<TMv, TNv, false, false>), grid, block, 0, st, p);     \
    }                                                                                                 \
  }
    if (fBM == 128 && fBN == 128) HIM_WF(2, 2)
    else if (fBM == 128) HIM_WF(2, 1)
    else if (fBN == 128) HIM_WF(1, 2)
    else HIM_WF(1, 1)
#undef HIM_WF
    int rc0 = check_launch("wgrad_fast");
    if (rc0) return rc0;
    const long long n = (long long)M * p.Np;
    (void)n;
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3(C / 64, M), dim3(256), 0, st, (const float*)ws, dw, M, C, KH * KW, fs,
                       accumulate);
    return check_launch("wgrad_finish");
  }
  int BM, BN;
  wgrad_tile(M, &BM, &BN);
  const int splits = wgrad_splits(M, p.Np, p.Kdim, BM, BN);
  p.splits = splits;
  p.accumulate = accumulate;
  int kchunk = cdiv(p.Kdim, splits);
  kchunk = ((kchunk + 31) / 32) * 32;
  p.kchunk = kchunk;
  if (splits > 1) {
    const size_t need = (size_t)splits * M * p.Np * sizeof(float);
    if (ws_bytes < need || !ws) return fail(HIM_E_WORKSPACE, "wgrad needs %zu ws bytes, got %zu", need, ws_bytes);
    p.out = (float*)ws;
  } else {
    p.out = dw;
  }
  dim3 grid(cdiv(p.Np, BN), cdiv(M, BM), splits), block(256);
  const bool refl = pad_mode == HIM_PAD_REFLECT;
  if (BM == 128) {
    if (refl) hipLaunchKernelGGL((wgrad_kernel<2, 2, 2, 2, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((wgrad_kernel<2, 2, 2, 2, false>), grid, block, 0, st, p);
  } else if (BM == 64) {
    if (refl) hipLaunchKernelGGL((wgrad_kernel<1, 4, 2, 1, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((wgrad_kernel<1, 4, 2, 1, false>), grid, block, 0, st, p);
  } else {
    if (refl) hipLaunchKernelGGL((wgrad_kernel<1, 4, 1, 1, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((wgrad_kernel<1, 4, 1, 1, false>), grid, block, 0, st, p);
  }
  int rc = check_launch("wgrad");
  if (rc) return rc;
  if (splits > 1) {
    const long long n = (long long)M * p.Np;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(std::min<long long>(cdiv(n, 256), 4096)), dim3(256), 0, st,
                       (const float*)ws, dw, n, splits, accumulate);
    rc = check_launch("slab_reduce");
  }
  return rc;
}

static int run_bias_grad(const float* dy, float* db, int B, int C, int hw, int accumulate, void* ws, size_t ws_bytes,
                         hipStream_t st) {
  if (!ws || ws_bytes < bias_ws_bytes(C)) return fail(HIM_E_WORKSPACE, "bias grad ws too small");
  const int nsl = bias_slices(B, hw);
  hipLaunchKernelGGL(bias_grad1_kernel, dim3(C, nsl), dim3(256), 0, st, dy, (float*)ws, B, C, hw);
  hipLaunchKernelGGL(bias_grad2_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, (const float*)ws, db, C, nsl, accumulate);
  return check_launch("bias_grad");
}

static int check_conv(const HimConv2d* d) {
  if (!d) return fail(HIM_E_INVALID, "null descriptor");
  if (d->B <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->H <= 0 || d->W <= 0 || d->KH <= 0 || d->KW <= 0 ||
      d->stride <= 0 || d->pad < 0)
    return fail(HIM_E_INVALID, "conv2d: non-positive dimension");
  if (d->stride > 2) return fail(HIM_E_UNSUPPORTED, "conv2d: stride %d > 2", d->stride);
  const int oh = (d->H + 2 * d->pad - d->KH) / d->stride + 1, ow = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
  if (oh != d->OH || ow != d->OW)
    return fail(HIM_E_INVALID, "conv2d: OH/OW (%d,%d) != expected (%d,%d)", d->OH, d->OW, oh, ow);
  if (d->pad_mode == HIM_PAD_REFLECT && (d->pad >= d->H || d->pad >= d->W))
    return fail(HIM_E_INVALID, "conv2d: reflect pad %d >= input size", d->pad);
  if (d->pad_mode == HIM_PAD_REFLECT && d->stride != 1)
    return fail(HIM_E_UNSUPPORTED, "conv2d: reflect pad needs stride 1");
  if ((long long)d->B * d->Cin * d->H * d->W >= (1ll << 31) || (long long)d->B * d->Cout * d->OH * d->OW >= (1ll << 31))
    return fail(HIM_E_UNSUPPORTED, "conv2d: tensor larger than 2^31 elements (32-bit offsets)");
  if ((long long)d->Cin * d->KH * d->KW >= (1 << 20) || (long long)d->Cout * d->KH * d->KW >= (1 << 20))
    return fail(HIM_E_UNSUPPORTED, "conv2d: reduction length too large for fastdiv");
  return HIM_OK;
}

static void fill_fprop(GConvP& g, const HimConv2d* d, const float* x, const float* w, const float* bias,
                       float* y) {
  memset(&g, 0, sizeof(g));
  g.src = x;
  g.dst = y;
  g.bias = bias;
  g.M = d->Cout;
  g.C2 = d->Cin;
  g.B = d->B;
  g.SH = d->H;
  g.SW = d->W;
  g.DH = d->OH;
  g.DW = d->OW;
  g.oys = g.oxs = 1;
  g.sy = g.sx = d->stride;
  g.dy = g.dx = 1;
  g.pad_mode = d->pad_mode;
  g.act = d->act;
  g.slope = d->slope;
  g.nphase = 1;
  GPhase& P = g.ph[0];
  P.A = w;
  P.K = d->Cin * d->KH * d->KW;
  P.JH = d->KH;
  P.JW = d->KW;
  P.fJHJW = make_fastdiv((uint32_t)(d->KH * d->KW));
  P.fJW = make_fastdiv((uint32_t)d->KW);
  P.NA = d->OH;
  P.NC = d->OW;
  P.oy0 = P.ox0 = 0;
  P.offy = P.offx = -d->pad;
}

static const int SMALL_NSPLIT = 8;
static bool small_split_ok(const HimConv2d* d) {
  return d->Cout <= 4 && d->Cin >= 256 && (long long)d->B * d->OH * d->OW < 256 * 512;
}
// split-K factor of a single-phase fast launch: aim at >= 2 workgroups per CU when the output has few tiles
static int fast_ksplit(const HimAlgo& a, int M, long long N, int nk) {
  if (algo_off(a, HIM_ALGO_NO_SPLITK)) return 1;
  // (the model still counts 128-row tiles for M > 64 although the launch uses 64x128 ones: counting those instead left
  // the step unchanged, 59.7 vs 59.8 ms; no split-K at all: 60.7)
  const long long tiles = (M <= 64 ? (long long)cdiv(N, 128) * cdiv(M, 64) : (long long)cdiv(N, 128) * cdiv(M, 128));
  if (tiles >= 1024) return 1;
  // makespan model in units of one full-K tile on one of 256 CUs: rounds/ks, a 7 % bonus once every CU hosts >= 2
  // independent workgroups (they cover each other's LDS/barrier bubbles), 2 % for the finish pass
  int best = 1;
  double best_cost = 1e30;
  const int cand[] = {1, 2, 3, 4, 6, 8, 12, 16};
  // The model is that of a launch ALONE on the chip; inside the multi-stream step the other streams' kernels fill the CUs
  // a few-tile launch leaves idle, and deep splits only add slab traffic and a longer finish pass: capped at 8 (round 3:
  // 58.5-58.6 vs 58.9 ms per step uncapped; caps of 3 / 4 / 6 within noise of 8, 2: 58.7; profiles/r03_tile_shape_ab.txt;
  // 8 rather than 4 because it leaves every launch of the C1 parity configuration on the split it was validated with).
  const int kmax = algo_ksplit_max(a);
  for (int ks : cand) {
    if (ks > kmax) break;
    if (ks > 1 && nk / ks < 32) break;  // keep >= 32 K-steps per workgroup
    const long long wg = tiles * ks;
    double cost = (double)cdiv(wg, 256) / ks;
    if (wg >= 512) cost *= 0.93;
    if (ks > 1) cost += 0.02;
    if (cost < best_cost - 1e-9) {
      best_cost = cost;
      best = ks;
    }
  }
  return best;
}
static bool wino_fwd_ok(const HimConv2d* d) {
  return wino_shape_ok(d->algo, d->Cout, d->Cin, d->KH, d->KW, d->stride, d->pad, d->H, d->W);
}
// The fused Winograd kernel (him_wino_fused.inc: transforms inside the GEMM kernel) takes the 3x3 stride-1 pad-1 layers
// with 64..512 reduction channels and a multiple of 64 output channels in the FORWARD direction (zero or reflection
// padding) and the data gradient of ZERO-padded layers: all VGG convs but conv1_1, the box2mask ResnetBlocks.
// Measured against the alternatives (tools/conv_bench.py / tools/micro/wino_micro, TFLOP/s direct-form equivalent):
// 64->64 150 vs 116 (direct MFMA kernel), 128->128 189 vs 131, 256->256 203 vs 131, 512->512 220 vs 194
// (separate-transform pipeline, which needs two more launches and 4x the activation in HBM).  The separate-transform
// pipeline keeps the 1024-channel ResnetBlock stack (see wino_fused_max_c), the data gradient of reflection-padded
// layers (border fold) and the weight gradient.
// Upper end of the fused kernel's channel range.  Inside the training step (weight panels streamed from HBM, 67 MB per
// 1024-channel layer) the ResnetBlock forward is faster on the separate-transform pipeline: 10.4 vs 11.8 ms generator
// forward, 121.4 vs 118.9 images/s (A/B with HimAlgo::wino_fused_max_c) -- although the isolated kernel, whose panel stays
// in the Infinity Cache between launches, measures 197 vs 188 TFLOP/s equivalent.  Round 4: with the LDS-DMA GEMM kernel the separate pipeline wins from 256 channels (the
// default range of the fused kernel ends at 255).
static bool wino_fused_ok(const HimAlgo& a, int Co, int Ci, int KH, int KW, int stride, int pad, int B, int H, int W) {
  const int mc = algo_wino_fused_min_c(a);
  // HimAlgo::wino_min_c < 0 turns EVERY Winograd form off (parity runs in the direct form)
  return mc > 0 && algo_wino_min_c(a) > 0 && Ci >= mc && Ci <= algo_wino_fused_max_c(a) && Co >= 64 &&
         wino_fused_shape_ok(Co, Ci, KH, KW, stride, pad, B, H, W);
}
// The persistent, wave-specialised form of the fused kernel (him_wino_fused2.inc) takes the launches of the fused range it
// supports (even W, reduction channels % 8 == 0 and >= 32, NONE / RELU / LRELU epilogue); everything else -- and everything
// under HIM_ALGO_NO_WINO_FUSED2 -- stays on him_wino_fused.inc.  Same weight panel: the choice is per launch, not per panel.
static bool wino_fused2_ok(const HimAlgo& a, int Co, int Ci, int B, int H, int W, int act) {
  return !algo_off(a, HIM_ALGO_NO_WINO_FUSED2) && algo_wino_fused_chunk(a) != 4 && wino_fused2_act_ok(act) &&
         wino_fused2_shape_ok(Co, Ci, 3, 3, 1, 1, B, H, W);
}
static bool wino_fused_fwd_ok(const HimConv2d* d) {
  return wino_fused_ok(d->algo, d->Cout, d->Cin, d->KH, d->KW, d->stride, d->pad, d->B, d->H, d->W);
}
// data gradient of a ZERO-padded 3x3 stride-1 conv = the same convolution with the flipped / transposed filter
static bool wino_fused_dgrad_ok(const HimConv2d* d) {
  return d->pad_mode == HIM_PAD_ZERO && d->OH == d->H && d->OW == d->W &&
         wino_fused_ok(d->algo, d->Cin, d->Cout, d->KH, d->KW, d->stride, d->pad, d->B, d->H, d->W);
}
// F(4x4,3x3) for frozen-weight layers (him_conv_wino4.inc); checked BEFORE the F(2x2,3x3) forms
static bool wino4_fwd_ok(const HimConv2d* d) {
  return wino4_shape_ok(d->algo, d->Cout, d->Cin, d->KH, d->KW, d->stride, d->pad, d->pad_mode, d->B, d->H, d->W) ||
         wino4_shape_ok(d->algo, d->Cout, d->Cin, d->KH, d->KW, d->stride, d->pad, d->pad_mode, d->B, d->H, d->W, true);
}
static bool wino4_dgrad_ok(const HimConv2d* d) {
  return d->OH == d->H && d->OW == d->W &&
         wino4_shape_ok(d->algo, d->Cin, d->Cout, d->KH, d->KW, d->stride, d->pad, d->pad_mode, d->B, d->H, d->W);
}
static size_t fprop_ws_bytes(const HimConv2d* d) {
  if (wino4_fwd_ok(d))
    return (wino4_panel_floats(d->Cout, d->Cin) + wino4_ws_floats(d->B, d->Cout, d->Cin, d->H, d->W)) * sizeof(float) + 256;
  if (wino_fused_fwd_ok(d)) return wino_fused_panel_floats(d->Cout, d->Cin) * sizeof(float) + 256;
  if (wino_fwd_ok(d))
    return ((size_t)16 * d->Cout * d->Cin + wino_conv_floats(d->B, d->Cin, d->Cout, d->OH, d->OW)) * sizeof(float) + 256;
  if (small_split_ok(d)) return (size_t)SMALL_NSPLIT * d->Cout * d->B * d->OH * d->OW * sizeof(float) + 256;
  if (!use_fast(d->algo, d->Cout, d->Cin)) return 0;
  const int ks = fast_ksplit(d->algo, d->Cout, (long long)d->B * d->OH * d->OW, d->KH * d->KW * (pad16(d->Cin) / 16));
  const size_t wts = ((size_t)d->Cout * d->KH * d->KW * pad16(d->Cin) * sizeof(float) + 255) / 256 * 256;
  return wts + (ks > 1 ? (size_t)ks * d->B * d->Cout * d->OH * d->OW * sizeof(float) : 0) + 256;
}
// floats of the regrouped weight panel the forward kernel reads (0: it reads the raw weights)
static size_t fprop_panel_floats(const HimConv2d* d) {
  if (small_split_ok(d) || d->Cout <= 4 || !use_fast(d->algo, d->Cout, d->Cin)) return 0;
  if (wino4_fwd_ok(d)) return wino4_panel_floats(d->Cout, d->Cin);
  if (wino_fused_fwd_ok(d)) return wino_fused_panel_floats(d->Cout, d->Cin);
  if (wino_fwd_ok(d)) return (size_t)16 * d->Cout * d->Cin;
  return (size_t)d->Cout * d->KH * d->KW * pad16(d->Cin);
}
// Conv2d -> InstanceNorm in one call (him_conv2d_in_act_fwd): the descriptors whose forward is a split-K launch of the fast
// implicit-GEMM kernel -- the few-tile layers (PatchGAN blocks, the last generator down-convolutions) -- hand their slabs to
// the InstanceNorm kernel.  Same conditions, in the same order, as run_fprop's dispatch.
static bool in_act_slab_ok(const HimConv2d* d) {
  if (wino4_fwd_ok(d) || wino_fused_fwd_ok(d) || wino_fwd_ok(d) || small_split_ok(d)) return false;
  if (!use_fast(d->algo, d->Cout, d->Cin) || d->act != HIM_ACT_NONE) return false;
  if ((unsigned long long)d->B * d->Cin * d->H * d->W * 4ull >= (1ull << 31)) return false;   // batch-sliced launches
  // launch_gconv's earlier exits (tiny-M kernel, few-channel tiled kernel) never see a fast split-K descriptor: Cout > 4
  // and Cin >= 16 are what use_fast asks for
  return fast_ksplit(d->algo, d->Cout, (long long)d->B * d->OH * d->OW, d->KH * d->KW * (pad16(d->Cin) / 16)) > 1;
}
// panel == nullptr: regroup the weights into the workspace on every call; build_only: write the panel to ws and return
// defer (him_conv2d_in_act_fwd): a split-K launch leaves its raw slabs in the workspace and reports them instead of
// running the finish pass (bias / activation / y are then the caller's: the InstanceNorm kernel reads the slabs)
struct FpropDefer {
  const float* slabs;
  int ks;
};
static int run_fprop(const HimConv2d* d, const float* x, const float* w, const float* bias, float* y, void* ws,
                     size_t ws_bytes, hipStream_t st, const float* panel = nullptr, bool build_only = false,
                     float* keep = nullptr, FpropDefer* defer = nullptr) {
  if (wino4_fwd_ok(d)) {
    const size_t pf = wino4_panel_floats(d->Cout, d->Cin);
    const size_t need = build_only ? pf * sizeof(float) : fprop_ws_bytes(d);
    if (!ws || ws_bytes < need) return fail(HIM_E_WORKSPACE, "conv fwd needs %zu ws bytes, got %zu", need, ws_bytes);
    float* U = (float*)ws;
    if (!panel) {
      hipLaunchKernelGGL((wino4_weight_kernel<0>), dim3(cdiv(d->Cin, 256), d->Cout), dim3(256), 0, st, w, U, d->Cout, d->Cin);
      int rc = check_launch("wino4_weight");
      if (rc || build_only) return rc;
    }
    return run_wino4_conv(d->B, d->Cin, d->H, d->W, d->Cout, x, panel ? panel : U, bias, d->act, d->slope, y, U + pf, st,
                          nullptr, d->pad_mode == HIM_PAD_REFLECT, !algo_off(d->algo, HIM_ALGO_NO_BGEMM_PERSISTENT));
  }
  if (wino_fused_fwd_ok(d)) {
    if (!panel) {
      const size_t need = wino_fused_panel_floats(d->Cout, d->Cin) * sizeof(float);
      if (!ws || ws_bytes < need) return fail(HIM_E_WORKSPACE, "conv fwd needs %zu ws bytes, got %zu", need, ws_bytes);
      hipLaunchKernelGGL((wino_fused_weight_kernel<0>), dim3(cdiv(d->Cin, 256), d->Cout), dim3(256), 0, st, w, (float*)ws,
                         d->Cout, d->Cin);
      int rc = check_launch("wino_fused_weight");
      if (rc || build_only) return rc;
    }
    if (wino_fused2_ok(d->algo, d->Cout, d->Cin, d->B, d->H, d->W, d->act))     // persistent form (round 6), same panel
      return run_wino_fused2(d->B, d->Cin, d->H, d->W, d->Cout, d->pad_mode == HIM_PAD_REFLECT, x,
                             panel ? panel : (const float*)ws, bias, d->act, d->slope, y, st, nullptr, device_cus());
    return run_wino_fused(d->B, d->Cin, d->H, d->W, d->Cout, d->pad_mode == HIM_PAD_REFLECT, x,
                          panel ? panel : (const float*)ws, bias, d->act, d->slope, y, st, nullptr,
                          algo_wino_fused_chunk(d->algo) == 4);
  }
  if (wino_fwd_ok(d)) {
    const size_t need = build_only ? fprop_panel_floats(d) * sizeof(float) : fprop_ws_bytes(d);
    if (!ws || ws_bytes < need) return fail(HIM_E_WORKSPACE, "conv fwd needs %zu ws bytes, got %zu", need, ws_bytes);
    float* U = (float*)ws;
    if (!panel) {
      hipLaunchKernelGGL((wino_weight_kernel<0>), dim3(cdiv(d->Cin, wino_tblock(d->algo)), d->Cout), dim3(wino_tblock(d->algo)), 0, st, w, U, d->Cout,
                         d->Cin);
      int rc = check_launch("wino_weight");
      if (rc || build_only) return rc;
    }
    return run_wino_conv(d->algo, d->B, d->Cin, d->H, d->W, d->Cout, d->OH, d->OW, 1, d->pad_mode == HIM_PAD_REFLECT, x,
                         panel ? panel : U, bias, d->act, d->slope, y, U + (size_t)16 * d->Cout * d->Cin, st, false, false,
                         keep);
  }
  GConvP g;
  fill_fprop(g, d, x, w, bias, y);
  if (small_split_ok(d)) {
    if (!ws || ws_bytes < fprop_ws_bytes(d)) return fail(HIM_E_WORKSPACE, "conv fwd ws too small");
    g.small_part = (float*)ws;
    g.small_nsplit = SMALL_NSPLIT;
  }
  if (use_fast(d->algo, d->Cout, d->Cin)) {
    const size_t need = build_only ? fprop_panel_floats(d) * sizeof(float) : fprop_ws_bytes(d);
    if (!ws || ws_bytes < need) return fail(HIM_E_WORKSPACE, "conv fwd needs %zu ws bytes, got %zu", need, ws_bytes);
    WT2P t;
    memset(&t, 0, sizeof(t));
    const int KK = d->KH * d->KW;
    t.W = w;
    t.M = d->Cout;
    t.C2 = d->Cin;
    t.C2p = pad16(d->Cin);
    t.sm = (long long)d->Cin * KK;
    t.sc = KK;
    t.ph[0].out = (float*)ws;
    t.ph[0].JH = d->KH;
    t.ph[0].JW = d->KW;
    t.ph[0].sh = d->KW;
    t.ph[0].sw = 1;
    t.ph[0].base = 0;
    t.ph[0].total = (long long)d->Cout * KK * t.C2p;
    if (!panel) {
      if (KK <= 64)
        hipLaunchKernelGGL(wt_fwd_kernel, dim3(t.C2p / 16, d->Cout), dim3(64), 0, st, w, (float*)ws, d->Cout, d->Cin,
                           t.C2p / 16, KK);
      else
        hipLaunchKernelGGL(wtrans2_kernel, dim3(std::min<long long>(cdiv(t.ph[0].total, 256), 4096), 1), dim3(256), 0, st, t);
      int rc = check_launch("wtrans2");
      if (rc || build_only) return rc;
    }
    g.fast = 1;
    g.ph[0].At = panel ? panel : (const float*)ws;
    g.ph[0].C2p = t.C2p;
    const int ks = fast_ksplit(d->algo, d->Cout, (long long)d->B * d->OH * d->OW, KK * (t.C2p / 16));
    if (ks > 1) {
      const size_t wts = ((size_t)d->Cout * KK * t.C2p * sizeof(float) + 255) / 256 * 256;
      g.ksplit = ks;
      g.kpart = (float*)((char*)ws + wts);
      if (defer && in_act_slab_ok(d)) {
        g.kno_finish = 1;
        defer->slabs = g.kpart;
        defer->ks = ks;
      }
    }
  }
  return launch_gconv(d->algo, g, st);
}

// data gradient of the conv described by `d` (also the forward of its transposed conv):
// out (B,Cin,H,W) = sum W * g (B,Cout,OH,OW); for reflect mode goes through the padded gradient + fold.
// reflect-pad-1 3x3 stride-1 (every ResnetBlock conv): gather from the border-extended gradient, no padded GEMM columns
static bool dfold_ok(const HimConv2d* d) {
  return !algo_off(d->algo, HIM_ALGO_NO_DFOLD) && d->pad_mode == HIM_PAD_REFLECT && d->pad == 1 && d->KH == 3 && d->KW == 3 && d->stride == 1 &&
         d->H >= 3 && d->W >= 3 && d->OH == d->H && d->OW == d->W && use_fast(d->algo, d->Cin, d->Cout);
}
static bool wino_dgrad_ok(const HimConv2d* d) {
  return wino_shape_ok(d->algo, d->Cin, d->Cout, d->KH, d->KW, d->stride, d->pad, d->H, d->W) && d->OH == d->H && d->OW == d->W;
}
static bool wino_dgrad_fold(const HimConv2d* d) {  // reflect folded into the border tiles' patches (see wino_input_kernel)
  return d->pad_mode == HIM_PAD_REFLECT && (d->H % 2) == 0 && (d->W % 2) == 0 && d->H >= 4 && d->W >= 4 &&
         !algo_off(d->algo, HIM_ALGO_WINO_PADDED_DGRAD);
}
static size_t wino_dgrad_floats(const HimConv2d* d) {  // U' + V + Mo (+ padded gradient for reflect)
  const bool refl = d->pad_mode == HIM_PAD_REFLECT && !wino_dgrad_fold(d);
  const int GH = refl ? d->H + 2 : d->H, GW = refl ? d->W + 2 : d->W;
  return (size_t)16 * d->Cin * d->Cout + wino_conv_floats(d->B, d->Cout, d->Cin, GH, GW) +
         (refl ? (size_t)d->B * d->Cin * GH * GW : 0);
}
static int dgrad_ksplit(const HimConv2d* d) {
  if (d->stride != 1 || !use_fast(d->algo, d->Cin, d->Cout)) return 1;
  const bool refl = d->pad_mode == HIM_PAD_REFLECT && !dfold_ok(d);
  const long long N = (long long)d->B * (refl ? d->H + 2 * d->pad : d->H) * (refl ? d->W + 2 * d->pad : d->W);
  return fast_ksplit(d->algo, d->Cin, N, d->KH * d->KW * (pad16(d->Cout) / 16));
}
static size_t dgrad_ws_bytes(const HimConv2d* d) {
  if (wino4_dgrad_ok(d))
    return (wino4_panel_floats(d->Cin, d->Cout) + wino4_ws_floats(d->B, d->Cin, d->Cout, d->H, d->W)) * sizeof(float) + 256;
  if (wino_fused_dgrad_ok(d)) return wino_fused_panel_floats(d->Cin, d->Cout) * sizeof(float) + 256;
  if (wino_dgrad_ok(d)) return wino_dgrad_floats(d) * sizeof(float) + 256;
  size_t n = (size_t)d->Cin * pad16(d->Cout) * d->KH * d->KW + 64;
  const size_t outn = (size_t)d->B * d->Cin * (d->H + 2 * d->pad) * (d->W + 2 * d->pad);
  if (dfold_ok(d)) n += (size_t)d->B * d->Cout * (d->OH + 2) * (d->OW + 2) + 64;
  else if (d->pad_mode == HIM_PAD_REFLECT) n += outn + 64;
  const int ks = dgrad_ksplit(d);
  if (ks > 1) n += (size_t)ks * outn + 64;
  return n * sizeof(float) + 256;
}
static size_t dgrad_panel_floats(const HimConv2d* d) {
  if (wino4_dgrad_ok(d)) return wino4_panel_floats(d->Cin, d->Cout);
  if (wino_fused_dgrad_ok(d)) return wino_fused_panel_floats(d->Cin, d->Cout);
  if (wino_dgrad_ok(d)) return (size_t)16 * d->Cin * d->Cout;
  return (size_t)d->Cin * (use_fast(d->algo, d->Cin, d->Cout) ? pad16(d->Cout) : d->Cout) * d->KH * d->KW;
}
// panel == nullptr: regroup the weights into the workspace on every call; build_only: write the panel to ws and return
static int run_dgrad(const HimConv2d* d, const float* gy, const float* w, float* out, const float* bias, int act,
                     float slope, void* ws, size_t ws_bytes, hipStream_t st, const float* panel = nullptr,
                     bool build_only = false, const float* relu_mask = nullptr, bool* mask_done = nullptr) {
  const size_t need = build_only ? dgrad_panel_floats(d) * sizeof(float) : dgrad_ws_bytes(d);
  if (!ws || ws_bytes < need) return fail(HIM_E_WORKSPACE, "dgrad needs %zu ws bytes, got %zu", need, ws_bytes);
  if (wino4_dgrad_ok(d)) {   // frozen weights, zero padding: the convolution of gy with the rotated / transposed filter
    const size_t pf = wino4_panel_floats(d->Cin, d->Cout);
    float* U = (float*)ws;
    if (!panel) {
      hipLaunchKernelGGL((wino4_weight_kernel<1>), dim3(cdiv(d->Cout, 256), d->Cin), dim3(256), 0, st, w, U, d->Cin, d->Cout);
      int rcu = check_launch("wino4_weight");
      if (rcu || build_only) return rcu;
    }
    if (mask_done) *mask_done = relu_mask != nullptr;   // the gate rides in the output transform
    return run_wino4_conv(d->B, d->Cout, d->OH, d->OW, d->Cin, gy, panel ? panel : U, bias, act, slope, out, U + pf, st,
                          relu_mask, false, !algo_off(d->algo, HIM_ALGO_NO_BGEMM_PERSISTENT));
  }
  if (wino_fused_dgrad_ok(d)) {   // zero-padded 3x3 stride-1: the convolution of gy with the flipped / transposed filter
    if (!panel) {
      hipLaunchKernelGGL((wino_fused_weight_kernel<1>), dim3(cdiv(d->Cout, 256), d->Cin), dim3(256), 0, st, w, (float*)ws,
                         d->Cin, d->Cout);
      int rcu = check_launch("wino_fused_weight");
      if (rcu || build_only) return rcu;
    }
    if (mask_done) *mask_done = relu_mask != nullptr;   // the gate rides in this kernel's epilogue
    if (wino_fused2_ok(d->algo, d->Cin, d->Cout, d->B, d->OH, d->OW, act))
      return run_wino_fused2(d->B, d->Cout, d->OH, d->OW, d->Cin, false, gy, panel ? panel : (const float*)ws, bias, act,
                             slope, out, st, relu_mask, device_cus());
    return run_wino_fused(d->B, d->Cout, d->OH, d->OW, d->Cin, false, gy, panel ? panel : (const float*)ws, bias, act, slope,
                          out, st, relu_mask, algo_wino_fused_chunk(d->algo) == 4);
  }
  if (wino_dgrad_ok(d) && panel && (bias || act != HIM_ACT_NONE))
    return fail(HIM_E_UNSUPPORTED, "dgrad: Winograd panel with a fused bias/activation epilogue");
  if (wino_dgrad_ok(d) && (build_only || (!bias && act == HIM_ACT_NONE))) {
    float* U = (float*)ws;
    if (!panel) {   // the FORWARD panel: the batched GEMM below reads it transposed (no flipped panel, round 3)
      hipLaunchKernelGGL((wino_weight_kernel<0>), dim3(cdiv(d->Cin, wino_tblock(d->algo)), d->Cout), dim3(wino_tblock(d->algo)), 0, st, w, U, d->Cout,
                         d->Cin);
      int rcu = check_launch("wino_weight");
      if (rcu || build_only) return rcu;
    }
    const bool fold = wino_dgrad_fold(d);
    const bool rf = d->pad_mode == HIM_PAD_REFLECT && !fold;
    const int GH = rf ? d->H + 2 : d->H, GW = rf ? d->W + 2 : d->W;
    float* wsv = U + (size_t)16 * d->Cin * d->Cout;
    float* dpadw = wsv + wino_conv_floats(d->B, d->Cout, d->Cin, GH, GW);
    // reflect: full correlation (offset 2) -> padded gradient -> fold; zero pad: the plain pad-1 correlation
    int rcw = run_wino_conv(d->algo, d->B, d->Cout, d->OH, d->OW, d->Cin, GH, GW, rf ? 2 : 1, false, gy, panel ? panel : U,
                            nullptr, HIM_ACT_NONE, 0.f, rf ? dpadw : out, wsv, st, fold, true);
    if (rcw || !rf) return rcw;
    hipLaunchKernelGGL(reflect_fold_kernel, dim3(cdiv((long long)d->H * d->W, 256), d->B * d->Cin), dim3(256), 0, st,
                       (const float*)dpadw, out, d->B * d->Cin, d->H, d->W, 1, 1, (size_t)0);
    return check_launch("reflect_fold");
  }
  float* Wt = panel ? (float*)panel : (float*)ws;
  const bool dfold = dfold_ok(d);
  const bool refl = d->pad_mode == HIM_PAD_REFLECT && !dfold;
  const bool fast = use_fast(d->algo, d->Cin, d->Cout);
  GConvP g;
  memset(&g, 0, sizeof(g));
  WTransP wt;
  memset(&wt, 0, sizeof(wt));
  wt.W = w;
  const int IH = refl ? d->H + 2 * d->pad : d->H, IW = refl ? d->W + 2 * d->pad : d->W;
  long long nw = setup_dgrad(g, wt, d->Cout, d->Cin, d->KH, d->KW, d->stride, refl ? 0 : d->pad, IH, IW, Wt);
  const bool holes = (g.kno_finish & 2) != 0;  // stride phases without a filter tap: zero gradient there
  g.kno_finish = 0;
  if (holes && (bias || act != HIM_ACT_NONE))
    return fail(HIM_E_UNSUPPORTED, "transposed conv with kernel < stride and a fused bias/activation");
  if (dfold) g.pad_mode = PAD_DFOLD;
  WT2P t2;
  if (fast) {  // regroup tap-major with the Cout axis padded to 16: At_q[ci][(jh*JW+jw)*Cop + co]
    memset(&t2, 0, sizeof(t2));
    const int KK = d->KH * d->KW, Cop = pad16(d->Cout);
    t2.W = w;
    t2.M = d->Cin;
    t2.C2 = d->Cout;
    t2.C2p = Cop;
    t2.sm = KK;
    t2.sc = (long long)d->Cin * KK;
    long long off = 0;
    for (int q = 0; q < g.nphase; ++q) {
      t2.ph[q].out = Wt + off;
      t2.ph[q].JH = g.ph[q].JH;
      t2.ph[q].JW = g.ph[q].JW;
      t2.ph[q].sh = (long long)d->stride * d->KW;
      t2.ph[q].sw = d->stride;
      t2.ph[q].base = (long long)wt.ph[q] * d->KW + wt.pw[q];
      t2.ph[q].total = (long long)d->Cin * g.ph[q].JH * g.ph[q].JW * Cop;
      g.ph[q].At = Wt + off;
      g.ph[q].C2p = Cop;
      off += t2.ph[q].total;
    }
    nw = off;
    g.fast = 1;
  }
  float* dpad = (float*)ws + ((nw + 63) / 64) * 64;
  if (fast && g.nphase == 1) {
    const int ks = dgrad_ksplit(d);
    if (ks > 1) {
      const size_t outn = (size_t)d->B * d->Cin * IH * IW;
      g.ksplit = ks;
      const size_t extn = dfold ? (size_t)d->B * d->Cout * (d->OH + 2) * (d->OW + 2) : 0;
      g.kpart = dpad + (refl ? ((outn + 63) / 64) * 64 : ((extn + 63) / 64) * 64);
      g.kno_finish = refl ? 1 : 0;
    }
  }
  g.src = dfold ? dpad : gy;
  g.dst = refl ? dpad : out;
  g.bias = refl ? nullptr : bias;
  g.M = d->Cin;
  g.C2 = d->Cout;
  g.B = d->B;
  g.SH = dfold ? d->OH + 2 : d->OH;
  g.SW = dfold ? d->OW + 2 : d->OW;
  g.DH = IH;
  g.DW = IW;
  g.act = refl ? HIM_ACT_NONE : act;
  g.slope = slope;
  int rc = HIM_OK;
  if (panel) {
    // weights already regrouped by a build_only call
  } else if (fast) {
    if (d->stride == 1 && d->KH * d->KW <= 49) {
      hipLaunchKernelGGL(wt_dgrad_kernel, dim3(cdiv(d->Cin, 16), pad16(d->Cout) / 16), dim3(256), 0, st, w, Wt, d->Cout,
                         d->Cin, pad16(d->Cout) / 16, d->KH * d->KW);
    } else {
      long long mx = 0;
      for (int q = 0; q < g.nphase; ++q) mx = std::max(mx, t2.ph[q].total);
      hipLaunchKernelGGL(wtrans2_kernel, dim3(std::min<long long>(cdiv(mx, 256), 4096), g.nphase), dim3(256), 0, st, t2);
    }
    rc = check_launch("wtrans2");
  } else {
    hipLaunchKernelGGL(wtrans_kernel, dim3(std::min<long long>(cdiv(nw, 256), 8192)), dim3(256), 0, st, wt);
    rc = check_launch("wtrans");
  }
  if (rc || build_only) return rc;
  if (holes) {
    rc = hipMemsetAsync(out, 0, (size_t)d->B * d->Cin * d->H * d->W * sizeof(float), st) == hipSuccess
             ? HIM_OK
             : fail(HIM_E_LAUNCH, "dgrad: memset failed");
    if (rc) return rc;
  }
  if (dfold) {
    hipLaunchKernelGGL(reflect_extend_kernel, dim3(cdiv((long long)(d->OH + 2) * (d->OW + 2), 256), d->B * d->Cout),
                       dim3(256), 0, st, gy, dpad, d->OH, d->OW);
    rc = check_launch("reflect_extend");
    if (rc) return rc;
  }
  // Reflection-padded few-channel data gradient (the generator's 7x7 head, 3 -> 64 at full resolution: the FIRST kernel of
  // the generator's backward pass): fold inside the tiled kernel (round 5, him_conv_direct.inc FOLD) instead of writing the
  // padded gradient and folding it in a second pass.  Same sums in the same order: bit-identical.
  if (refl && g.ksplit <= 1 && !algo_off(d->algo, HIM_ALGO_NO_FEWIN_FOLD) && fewin_tiled_ok(d->algo, g)) {
    int sy, sx;
    if (fewin_fold_shift(IH, d->pad, 16, &sy) && fewin_fold_shift(IW, d->pad, 64, &sx)) {
      g.fold_p = d->pad;
      g.fold_sy = sy;
      g.fold_sx = sx;
      g.dst = out;
      g.DH = d->H;
      g.DW = d->W;
      if (fewin_tiled_ok(d->algo, g)) return launch_gconv(d->algo, g, st);
      g.fold_p = g.fold_sy = g.fold_sx = 0;
      g.dst = dpad;
      g.DH = IH;
      g.DW = IW;
    }
  }
  rc = launch_gconv(d->algo, g, st);
  if (rc) return rc;
  if (refl) {
    const long long tot = (long long)d->B * d->Cin * d->H * d->W;
    const bool slabs = g.ksplit > 1;
    (void)tot;
    hipLaunchKernelGGL(reflect_fold_kernel, dim3(cdiv((long long)d->H * d->W, 256), d->B * d->Cin), dim3(256), 0, st,
                       (const float*)(slabs ? g.kpart : dpad), out, d->B * d->Cin, d->H, d->W, d->pad,
                       slabs ? g.ksplit : 1, (size_t)d->B * d->Cin * IH * IW);
    rc = check_launch("reflect_fold");
  }
  return rc;
}


#include "him_conv_onehot.inc"

static int adjoint_of(const HimDeconv2d* t, HimConv2d* c) {
  if (!t) return fail(HIM_E_INVALID, "null descriptor");
  const int oh = (t->H - 1) * t->stride - 2 * t->pad + t->KH + t->out_pad;
  const int ow = (t->W - 1) * t->stride - 2 * t->pad + t->KW + t->out_pad;
  if (oh != t->OH || ow != t->OW)
    return fail(HIM_E_INVALID, "deconv2d: OH/OW (%d,%d) != expected (%d,%d)", t->OH, t->OW, oh, ow);
  c->B = t->B;
  c->Cin = t->Cout;  // adjoint conv maps the deconv OUTPUT space ...
  c->H = t->OH;
  c->W = t->OW;
  c->Cout = t->Cin;  // ... onto the deconv INPUT space
  c->KH = t->KH;
  c->KW = t->KW;
  c->stride = t->stride;
  c->pad = t->pad;
  c->pad_mode = HIM_PAD_ZERO;
  c->OH = t->H;
  c->OW = t->W;
  c->act = HIM_ACT_NONE;
  c->slope = 0.f;
  c->algo = t->algo;
  if ((c->H + 2 * c->pad - c->KH) / c->stride + 1 != c->OH || (c->W + 2 * c->pad - c->KW) / c->stride + 1 != c->OW)
    return fail(HIM_E_UNSUPPORTED, "deconv2d: output_padding %d not representable", t->out_pad);
  return check_conv(c);
}

}  // namespace him

using namespace him;

#include "him_resblock.inc"

extern "C" {

void him_algo_resolve(const HimAlgo* in, HimAlgo* out) {
  if (!out) return;
  HimAlgo z;
  memset(&z, 0, sizeof(z));
  const HimAlgo& a = in ? *in : z;
  HimAlgo r = a;
  r.wino_min_c = algo_wino_min_c(a);
  r.wino_fused_min_c = algo_wino_fused_min_c(a);
  r.wino_fused_max_c = algo_wino_fused_max_c(a);
  r.wino4_min_c = algo_wino4_min_c(a);
  r.ksplit_max = algo_off(a, HIM_ALGO_NO_SPLITK) ? 1 : algo_ksplit_max(a);
  r.tile_wb = a.tile_wb == HIM_TILE_DEFAULT ? HIM_TILE_64x128 : a.tile_wb;
  r.tile_nb = a.tile_nb == HIM_TILE_DEFAULT ? HIM_TILE_64x128 : a.tile_nb;
  r.wino_tblock = algo_tblock(a);
  r.wino_fused_chunk = algo_wino_fused_chunk(a);
  r.wgrad_tile = (a.wgrad_tile == 1 || a.wgrad_tile == 2) ? a.wgrad_tile : 0;
  *out = r;
}

// The ONE place the library touches the environment (tools/ A/B runs: env -> HimAlgo -> descriptors).
void him_algo_from_env(HimAlgo* a) {
  if (!a) return;
  memset(a, 0, sizeof(*a));
  auto geti = [](const char* k, int dflt) { const char* e = getenv(k); return e ? atoi(e) : dflt; };
  a->wino_min_c = getenv("HIM_NO_WINOGRAD") ? -1 : geti("HIM_WINO_MIN_C", 0);
  a->wino_fused_min_c = getenv("HIM_NO_WINO_FUSED") ? -1 : geti("HIM_WINO_FUSED_MIN_C", 0);
  a->wino_fused_max_c = geti("HIM_WINO_FUSED_MAX_C", 0);
  a->wino4_min_c = getenv("HIM_NO_WINO4") ? -1 : geti("HIM_WINO4_MIN_C", 0);
  a->ksplit_max = geti("HIM_KSPLIT_MAX", 0);
  const int tile_all = geti("HIM_GCONV_TILE", 0);
  a->tile_wb = geti("HIM_GCONV_TILE_WB", tile_all);
  a->tile_nb = geti("HIM_GCONV_TILE_NB", tile_all);
  a->wino_tblock = geti("HIM_WINO_TBLOCK", 0);
  a->wgrad_splits = geti("HIM_WGRAD_SPLITS", 0);
  a->wino_fused_chunk = geti("HIM_WINO_FUSED_CHUNK", 0);
  a->wgrad_tile = geti("HIM_WGRAD_TILE", 0);
  const struct { const char* k; unsigned bit; } flags[] = {
      {"HIM_NO_SPLITK", HIM_ALGO_NO_SPLITK},           {"HIM_NO_DFOLD", HIM_ALGO_NO_DFOLD},
      {"HIM_WINO_PADDED_DGRAD", HIM_ALGO_WINO_PADDED_DGRAD}, {"HIM_NO_SMALL_WIN", HIM_ALGO_NO_SMALL_WIN},
      {"HIM_NO_FEWOUT_TILED", HIM_ALGO_NO_FEWOUT_TILED}, {"HIM_NO_FEWIN_TILED", HIM_ALGO_NO_FEWIN_TILED},
      {"HIM_NO_FEWCH_MFMA", HIM_ALGO_NO_FEWCH_MFMA},   {"HIM_GENERIC_CONV", HIM_ALGO_GENERIC_CONV},
      {"HIM_NO_RESBLOCK_FUSED", HIM_ALGO_NO_RESBLOCK_FUSED}, {"HIM_NO_BGEMM", HIM_ALGO_NO_BGEMM},
      {"HIM_NO_ONEHOT_RLE", HIM_ALGO_NO_ONEHOT_RLE},   {"HIM_NO_FEWIN_FOLD", HIM_ALGO_NO_FEWIN_FOLD},
      {"HIM_WINO4_TRAIN_FWD", HIM_ALGO_WINO4_TRAIN_FWD}, {"HIM_NO_FEWIN_REFLECT", HIM_ALGO_NO_FEWIN_REFLECT},
      {"HIM_NO_WINO_FUSED2", HIM_ALGO_NO_WINO_FUSED2}, {"HIM_NO_BGEMM_PERSISTENT", HIM_ALGO_NO_BGEMM_PERSISTENT}};
  for (const auto& f : flags)
    if (getenv(f.k)) a->disable |= f.bit;
}

int him_winograd_gemm(const float* a, const float* b, float* c, int M, int K, int N, const HimAlgo* algo, void* stream) {
  if (!a || !b || !c || M <= 4 || K < 16 || (K % 16) || N <= 0 || (N % 128))
    return fail(HIM_E_INVALID, "winograd gemm: need M > 4, K %% 16 == 0, N %% 128 == 0 (got %d, %d, %d)", M, K, N);
  if ((long long)16 * K * N >= (1ll << 31) || (long long)16 * M * N >= (1ll << 31))
    return fail(HIM_E_UNSUPPORTED, "winograd gemm: operand larger than 2^31 elements");
  HimAlgo z;
  memset(&z, 0, sizeof(z));
  return wino_batched_gemm(algo ? *algo : z, a, b, c, M, K, N, (hipStream_t)stream);
}

size_t him_conv2d_fwd_ws(const HimConv2d* d) { return d ? fprop_ws_bytes(d) : 0; }

int him_conv2d_fwd(const HimConv2d* d, const float* x, const float* w, const float* bias, float* y, void* ws,
                   size_t ws_bytes, void* stream) {
  int rc = check_conv(d);
  if (rc) return rc;
  return run_fprop(d, x, w, bias, y, ws, ws_bytes, (hipStream_t)stream);
}

int him_conv2d_in_act_fused(const HimConv2d* d) { return (d && !check_conv(d) && in_act_slab_ok(d)) ? 1 : 0; }

int him_conv2d_in_act_fwd(const HimConv2d* d, const float* x, const float* w, const void* panel, const float* bias,
                          float* y_raw, const float* residual, float* z, float* mean, float* rstd, float eps, int act,
                          float slope, void* ws, size_t ws_bytes, void* stream) {
  int rc = check_conv(d);
  if (rc) return rc;
  if (d->act != HIM_ACT_NONE) return fail(HIM_E_INVALID, "conv_in_act: the activation follows the norm (descriptor act must be none)");
  if (!y_raw || !z || !mean || !rstd) return fail(HIM_E_INVALID, "conv_in_act: null output");
  if (panel && !fprop_panel_floats(d)) return fail(HIM_E_INVALID, "conv_in_act: no panel for this descriptor");
  hipStream_t st = (hipStream_t)stream;
  FpropDefer df;
  df.slabs = nullptr;
  df.ks = 0;
  rc = run_fprop(d, x, panel ? nullptr : w, bias, y_raw, ws, ws_bytes, st, (const float*)panel, false, nullptr, &df);
  if (rc) return rc;
  const int planes = d->B * d->Cout, hw = d->OH * d->OW;
  if (df.ks > 1)
    return instnorm_fwd_from_slabs(df.slabs, (long long)planes * hw, df.ks, bias, d->Cout, y_raw, residual, z, mean, rstd,
                                   planes, hw, eps, act, slope, st);
  return him_instnorm_fwd(y_raw, residual, z, mean, rstd, planes, hw, eps, act, slope, stream);
}

size_t him_conv2d_bwd_data_ws(const HimConv2d* d) { return d ? dgrad_ws_bytes(d) : 0; }

int him_conv2d_bwd_data(const HimConv2d* d, const float* dy, const float* w, float* dx, void* ws,
                        size_t ws_bytes, void* stream) {
  int rc = check_conv(d);
  if (rc) return rc;
  return run_dgrad(d, dy, w, dx, nullptr, HIM_ACT_NONE, 0.f, ws, ws_bytes, (hipStream_t)stream);
}

size_t him_conv2d_panel_bytes(const HimConv2d* d, int kind) {
  if (!d || check_conv(d)) return 0;
  return (kind == HIM_PANEL_FWD ? fprop_panel_floats(d) : dgrad_panel_floats(d)) * sizeof(float);
}

// Which regrouping of the weights the panel of (descriptor, kind) holds: 0 none (the kernel reads the raw weights), 1 the
// implicit-GEMM panel, 2 Winograd F(2x2,3x3) for the separate-transform pipeline (16 positions), 3 the fused Winograd
// kernel's chunked panel, 4 Winograd F(4x4,3x3) (36 positions, frozen weights).  A pure function of the descriptor: the
// SAME weight called with another batch / plane size may land on another layout (F(4x4) needs H, W % 4 == 0 and >= 64 real
// tiles; the tiny-head forms switch on the output size), so a cache of built panels must key on it.
int him_conv2d_panel_layout(const HimConv2d* d, int kind) {
  if (!d || check_conv(d)) return 0;
  if (kind == HIM_PANEL_FWD) {
    if (!fprop_panel_floats(d)) return 0;
    return wino4_fwd_ok(d) ? 4 : wino_fused_fwd_ok(d) ? 3 : wino_fwd_ok(d) ? 2 : 1;
  }
  if (kind != HIM_PANEL_BWD_DATA || !dgrad_panel_floats(d)) return 0;
  return wino4_dgrad_ok(d) ? 4 : wino_fused_dgrad_ok(d) ? 3 : wino_dgrad_ok(d) ? 2 : 1;
}

int him_conv2d_bwd_data_shares_fwd_panel(const HimConv2d* d) {
  if (!d || check_conv(d)) return 0;
  if (wino4_fwd_ok(d) || wino4_dgrad_ok(d)) return 0;
  return (!wino_fused_dgrad_ok(d) && wino_dgrad_ok(d) && !wino_fused_fwd_ok(d) && wino_fwd_ok(d)) ? 1 : 0;
}

int him_conv2d_panel_build(const HimConv2d* d, int kind, const float* w, void* panel, size_t panel_bytes,
                           void* stream) {
  int rc = check_conv(d);
  if (rc) return rc;
  if (kind == HIM_PANEL_FWD) {
    if (!fprop_panel_floats(d)) return fail(HIM_E_UNSUPPORTED, "this conv's forward reads the raw weights: no panel");
    return run_fprop(d, nullptr, w, nullptr, nullptr, panel, panel_bytes, (hipStream_t)stream, nullptr, true);
  }
  if (kind != HIM_PANEL_BWD_DATA) return fail(HIM_E_INVALID, "panel kind %d", kind);
  return run_dgrad(d, nullptr, w, nullptr, nullptr, HIM_ACT_NONE, 0.f, panel, panel_bytes, (hipStream_t)stream, nullptr,
                   true);
}

int him_conv2d_fwd_panel(const HimConv2d* d, const float* x, const void* panel, const float* bias, float* y, void* ws,
                         size_t ws_bytes, void* stream) {
  int rc = check_conv(d);
  if (rc) return rc;
  if (!panel || !fprop_panel_floats(d)) return fail(HIM_E_INVALID, "conv fwd: no panel for this descriptor");
  return run_fprop(d, x, nullptr, bias, y, ws, ws_bytes, (hipStream_t)stream, (const float*)panel);
}

int him_conv2d_bwd_data_panel(const HimConv2d* d, const float* dy, const void* panel, float* dx, void* ws,
                              size_t ws_bytes, void* stream) {
  int rc = check_conv(d);
  if (rc) return rc;
  if (!panel) return fail(HIM_E_INVALID, "conv bwd_data: null panel");
  return run_dgrad(d, dy, nullptr, dx, nullptr, HIM_ACT_NONE, 0.f, ws, ws_bytes, (hipStream_t)stream,
                   (const float*)panel);
}

// dx = dgrad(dy) gated by the ReLU that produced this layer's input: dx[i] = x[i] > 0 ? dx[i] : 0.  Fused into the
// epilogue where the layer runs the fused Winograd kernel; one elementwise pass behind the other kernels.
__global__ void relu_gate_kernel(const float* __restrict__ x, float* __restrict__ dx, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dx[i] = x[i] > 0.f ? dx[i] : 0.f;
}

int him_conv2d_bwd_data_gated(const HimConv2d* d, const float* dy, const float* w, const void* panel, const float* x,
                              float* dx, void* ws, size_t ws_bytes, void* stream) {
  int rc = check_conv(d);
  if (rc) return rc;
  if (!x) return fail(HIM_E_INVALID, "conv bwd_data_gated: null input tensor");
  if (!w && !panel) return fail(HIM_E_INVALID, "conv bwd_data_gated: neither weights nor a panel");
  bool done = false;
  rc = run_dgrad(d, dy, panel ? nullptr : w, dx, nullptr, HIM_ACT_NONE, 0.f, ws, ws_bytes, (hipStream_t)stream,
                 (const float*)panel, false, x, &done);
  if (rc || done) return rc;
  const size_t n = (size_t)d->B * d->Cin * d->H * d->W;
  hipLaunchKernelGGL(relu_gate_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0,
                     (hipStream_t)stream, x, dx, n);
  return check_launch("relu_gate");
}

// ---- kept Winograd input transform (include/him.h) ----
static bool fwd_keep_ok(const HimConv2d* d) {
  if (check_conv(d) || wino4_fwd_ok(d) || wino_fused_fwd_ok(d) || !wino_fwd_ok(d)) return false;
  if (!wino_wgrad_ok(d->algo, d->Cout, d->Cin, d->KH, d->KW, d->stride, d->pad, d->H, d->W) || d->OH != d->H || d->OW != d->W)
    return false;
  const WinoGeom g = wino_geom(d->B, d->Cin, d->H, d->W, d->OH, d->OW, 1);
  return !algo_off(d->algo, HIM_ALGO_NO_BGEMM) && bgemm_shape_ok(d->Cout, g.Tp, d->Cin, 16);
}
size_t him_conv2d_fwd_keep_bytes(const HimConv2d* d) {
  if (!d || !fwd_keep_ok(d)) return 0;
  const WinoGeom g = wino_geom(d->B, d->Cin, d->H, d->W, d->OH, d->OW, 1);
  return (size_t)16 * d->Cin * g.Tp * sizeof(float);
}
int him_conv2d_fwd_panel_keep(const HimConv2d* d, const float* x, const void* panel, const float* bias, float* y, float* keep,
                              void* ws, size_t ws_bytes, void* stream) {
  int rc = check_conv(d);
  if (rc) return rc;
  if (!panel || !fprop_panel_floats(d)) return fail(HIM_E_INVALID, "conv fwd: no panel for this descriptor");
  if (keep && !fwd_keep_ok(d)) return fail(HIM_E_UNSUPPORTED, "conv fwd: this layer has no transformed input to keep");
  return run_fprop(d, x, nullptr, bias, y, ws, ws_bytes, (hipStream_t)stream, (const float*)panel, false, keep);
}
int him_conv2d_bwd_weight_kept(const HimConv2d* d, const float* keep, const float* dy, float* dw, float* dbias, int accumulate,
                               void* ws, size_t ws_bytes, void* stream) {
  int rc = check_conv(d);
  if (rc) return rc;
  if (!keep || !fwd_keep_ok(d)) return fail(HIM_E_INVALID, "conv bwd_weight_kept: no kept transform for this descriptor");
  if (dw) {
    rc = run_wgrad(d->algo, dy, nullptr, dw, d->Cout, d->Cin, d->B, d->H, d->W, d->OH, d->OW, d->KH, d->KW, d->stride, d->pad,
                   d->pad_mode, accumulate, ws, ws_bytes, (hipStream_t)stream, keep);
    if (rc) return rc;
  }
  if (dbias) {
    const size_t off = wgrad_slab_bytes(d->algo, d->Cout, d->Cin, d->KH, d->KW, d->B * d->OH * d->OW, conv_wino_wgrad_floats(d));
    if (ws_bytes < off) return fail(HIM_E_WORKSPACE, "bwd_weight ws too small");
    rc = run_bias_grad(dy, dbias, d->B, d->Cout, d->OH * d->OW, accumulate, (char*)ws + off, ws_bytes - off,
                       (hipStream_t)stream);
  }
  return rc;
}

size_t him_conv2d_bwd_weight_ws(const HimConv2d* d) {
  return d ? wgrad_ws_bytes(d->algo, d->Cout, d->Cin, d->KH, d->KW, d->B * d->OH * d->OW, d->Cout, conv_wino_wgrad_floats(d)) : 0;
}

int him_conv2d_bwd_weight(const HimConv2d* d, const float* x, const float* dy, float* dw, float* dbias,
                          int accumulate, void* ws, size_t ws_bytes, void* stream) {
  int rc = check_conv(d);
  if (rc) return rc;
  if (dw) {
    rc = run_wgrad(d->algo, dy, x, dw, d->Cout, d->Cin, d->B, d->H, d->W, d->OH, d->OW, d->KH, d->KW, d->stride, d->pad,
                   d->pad_mode, accumulate, ws, ws_bytes, (hipStream_t)stream);
    if (rc) return rc;
  }
  if (dbias) {
    const size_t off = wgrad_slab_bytes(d->algo, d->Cout, d->Cin, d->KH, d->KW, d->B * d->OH * d->OW, conv_wino_wgrad_floats(d));
    if (ws_bytes < off) return fail(HIM_E_WORKSPACE, "bwd_weight ws too small");
    rc = run_bias_grad(dy, dbias, d->B, d->Cout, d->OH * d->OW, accumulate, (char*)ws + off, ws_bytes - off,
                       (hipStream_t)stream);
  }
  return rc;
}


size_t him_conv2d_onehot_fwd_ws(const HimConv2d* d, int n_onehot) {
  return (d && !check_conv(d) && onehot_ok(d, n_onehot)) ? onehot_fwd_ws_bytes(d, n_onehot) : 0;
}

// x_dense: x holds ONLY the dense channels, (B, Cin - n_onehot, H, W) contiguous -- no (B, Cin, H, W) buffer exists
static int onehot_fwd_impl(const HimConv2d* d, const float* label, int n_onehot, const float* x, bool x_dense, const float* w,
                           const float* bias, float* y, void* ws, size_t ws_bytes, void* stream) {
  int rc = check_conv(d);
  if (rc) return rc;
  if (!onehot_ok(d, n_onehot))
    return fail(HIM_E_UNSUPPORTED, "onehot conv: needs an odd 'same' stride-1 kernel or 4x4 stride 2 zero-padded, Cout %% 16 == 0");
  if (!ws || ws_bytes < onehot_fwd_ws_bytes(d, n_onehot)) return fail(HIM_E_WORKSPACE, "onehot conv fwd: ws too small");
  hipStream_t st = (hipStream_t)stream;
  const int NC = n_onehot, Cd = d->Cin - NC, KK = d->KH * d->KW, HW = d->H * d->W;
  float* Wt = (float*)ws;
  float* xd = Wt + ((size_t)KK * NC * d->Cout + 63) / 64 * 64;
  hipLaunchKernelGGL(onehot_table_kernel, dim3(cdiv((long long)KK * NC * d->Cout, 256)), dim3(256), 0, st, w, Wt, d->Cout,
                     d->Cin, NC, KK);
  if (Cd > 0) {  // dense channels: the ordinary conv on contiguous copies (bias folded in here)
    float* wd = xd + ((size_t)d->B * Cd * HW + 63) / 64 * 64;
    float* cws = wd + ((size_t)d->Cout * Cd * KK + 63) / 64 * 64;
    if (!x_dense) {
      rc = him_copy_channels(x, d->Cin, NC, xd, Cd, 0, Cd, d->B, HW, nullptr, 0, 0, stream);
      if (rc) return rc;
    }
    hipLaunchKernelGGL(onehot_dense_w_kernel, dim3(cdiv((long long)d->Cout * Cd * KK, 256)), dim3(256), 0, st,
                       (float*)w, wd, d->Cout, d->Cin, NC, KK, 0, 0);
    const HimConv2d dd = onehot_dense_desc(d, NC);
    rc = run_fprop(&dd, x_dense ? x : xd, wd, bias, y, cws, fprop_ws_bytes(&dd), st);
    if (rc) return rc;
  }
  OneHotP p;
  p.label = label;
  p.B = d->B; p.H = d->H; p.W = d->W; p.NC = NC; p.KS = d->KH; p.pad = d->pad;
  p.reflect = d->pad_mode == HIM_PAD_REFLECT;
  p.Cout = d->Cout;
  p.stride = d->stride; p.OH = d->OH; p.OW = d->OW;
  p.npix = d->B * d->OH * d->OW;
  const size_t lds = (size_t)KK * NC * 16 * sizeof(float);
  const dim3 grid(std::min(cdiv(p.npix, 1024), 64), d->Cout / 16);
#define HIM_OH_FWD(KSv)                                                                                              \
  {                                                                                                                  \
    (void)hipFuncSetAttribute((const void*)onehot_conv_fwd_kernel<KSv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((onehot_conv_fwd_kernel<KSv>), grid, dim3(1024), lds, st, p, (const float*)Wt, bias, y,        \
                       Cd > 0 ? 1 : 0, d->act, d->slope);                                                            \
  }
  if (d->KH == 7) HIM_OH_FWD(7)
  else if (d->KH == 5) HIM_OH_FWD(5)
  else if (d->KH == 4) HIM_OH_FWD(4)
  else HIM_OH_FWD(3)
#undef HIM_OH_FWD
  return check_launch("onehot_conv_fwd");
}

int him_conv2d_onehot_fwd(const HimConv2d* d, const float* label, int n_onehot, const float* x, const float* w,
                          const float* bias, float* y, void* ws, size_t ws_bytes, void* stream) {
  return onehot_fwd_impl(d, label, n_onehot, x, false, w, bias, y, ws, ws_bytes, stream);
}
int him_conv2d_onehot_fwd_dense(const HimConv2d* d, const float* label, int n_onehot, const float* xdense, const float* w,
                                const float* bias, float* y, void* ws, size_t ws_bytes, void* stream) {
  if (d && d->Cin > n_onehot && !xdense) return fail(HIM_E_INVALID, "onehot conv: %d dense channels but no dense tensor", d->Cin - n_onehot);
  return onehot_fwd_impl(d, label, n_onehot, xdense, true, w, bias, y, ws, ws_bytes, stream);
}

size_t him_conv2d_onehot_bwd_weight_ws(const HimConv2d* d, int n_onehot) {
  return (d && !check_conv(d) && onehot_ok(d, n_onehot)) ? onehot_wgrad_ws_bytes(d, n_onehot) + bias_ws_bytes(d->Cout) : 0;
}

// parts: HIM_ONEHOT_PART_IDS = the label-id channels' slice of dw (run-length kernel + reduce), HIM_ONEHOT_PART_DENSE = the
// dense channels' slice of dw (MFMA weight gradient) + dbias.  The two write disjoint elements and use disjoint regions of ws.
static int onehot_bwd_weight_impl(const HimConv2d* d, const float* label, int n_onehot, const float* x, bool x_dense,
                                  const float* dy, float* dw, float* dbias, int accumulate, void* ws, size_t ws_bytes,
                                  void* stream, int parts = HIM_ONEHOT_PART_IDS | HIM_ONEHOT_PART_DENSE) {
  float* const dw_ids = (parts & HIM_ONEHOT_PART_IDS) ? dw : nullptr;
  if (!(parts & HIM_ONEHOT_PART_DENSE)) dbias = nullptr;
  float* const dw_all = dw;
  dw = dw_ids;
  int rc = check_conv(d);
  if (rc) return rc;
  if (!onehot_ok(d, n_onehot)) return fail(HIM_E_UNSUPPORTED, "onehot conv: unsupported descriptor");
  if (!ws || ws_bytes < him_conv2d_onehot_bwd_weight_ws(d, n_onehot))
    return fail(HIM_E_WORKSPACE, "onehot conv bwd_weight: ws too small");
  hipStream_t st = (hipStream_t)stream;
  const int NC = n_onehot, Cd = d->Cin - NC, KK = d->KH * d->KW, HW = d->H * d->W;
  float* part = (float*)ws;
  float* xd = part + (size_t)onehot_wgrad_blocks(d) * KK * NC * d->Cout + 64;
  OneHotP p;
  p.label = label;
  p.B = d->B; p.H = d->H; p.W = d->W; p.NC = NC; p.KS = d->KH; p.pad = d->pad;
  p.reflect = d->pad_mode == HIM_PAD_REFLECT;
  p.Cout = d->Cout;
  p.stride = d->stride; p.OH = d->OH; p.OW = d->OW;
  p.npix = d->B * d->OH * d->OW;
  if (dw && onehot_rle_ok(d, NC)) {   // run-length form (him_conv_onehot.inc): dy read once, cost per run of equal class
    int nbands, rows_per;
    onehot_rle_geom(d, &nbands, &rows_per);
    const int nblk = d->B * nbands;
    const size_t lds = onehot_rle_lds_bytes(d, NC);
#define HIM_OH_RLE(KSv, Sv)                                                                                          \
  {                                                                                                                  \
    (void)hipFuncSetAttribute((const void*)onehot_wgrad_rle_kernel<KSv, Sv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((onehot_wgrad_rle_kernel<KSv, Sv>), dim3(nblk, d->Cout / 8), dim3(512), lds, st, p, dy, part, nbands, \
                       rows_per);                                                                                    \
  }
    if (d->stride == 2) HIM_OH_RLE(4, 2)
    else if (d->KH == 7) HIM_OH_RLE(7, 1)
    else if (d->KH == 5) HIM_OH_RLE(5, 1)
    else HIM_OH_RLE(3, 1)
#undef HIM_OH_RLE
    hipLaunchKernelGGL(onehot_wgrad_reduce_kernel, dim3(cdiv((long long)KK * NC * d->Cout, 256)), dim3(256), 0, st,
                       (const float*)part, dw, nblk, d->Cout, d->Cin, NC, KK, accumulate);
    rc = check_launch("onehot_wgrad_rle");
    if (rc) return rc;
  } else if (dw) {
    if (d->stride != 1) return fail(HIM_E_UNSUPPORTED, "onehot conv: the strided weight gradient exists in run-length form only");
    int nsx, nyc, rows_per;
    onehot_wgrad_geom(d, &nsx, &nyc, &rows_per);
    const int nblk = d->B * nsx * nyc;
    const size_t lds = (size_t)4 * KK * NC * 4 * sizeof(float);
#define HIM_OH_WG(KSv)                                                                                              \
  {                                                                                                                 \
    (void)hipFuncSetAttribute((const void*)onehot_wgrad_kernel<KSv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((onehot_wgrad_kernel<KSv>), dim3(nblk, d->Cout / 16), dim3(512), lds, st, p, dy, part, nsx, nyc, \
                       rows_per);                                                                                   \
  }
    if (d->KH == 7) HIM_OH_WG(7)
    else if (d->KH == 5) HIM_OH_WG(5)
    else HIM_OH_WG(3)
#undef HIM_OH_WG
    hipLaunchKernelGGL(onehot_wgrad_reduce_kernel, dim3(cdiv((long long)KK * NC * d->Cout, 256)), dim3(256), 0, st,
                       (const float*)part, dw, nblk, d->Cout, d->Cin, NC, KK, accumulate);
    rc = check_launch("onehot_wgrad");
    if (rc) return rc;
  }
  dw = (parts & HIM_ONEHOT_PART_DENSE) ? dw_all : nullptr;
  if (dw) {
    if (Cd > 0) {
      float* dwd = xd + ((size_t)d->B * Cd * HW + 63) / 64 * 64;
      float* wws = dwd + ((size_t)d->Cout * Cd * KK + 63) / 64 * 64;
      if (!x_dense) {
        rc = him_copy_channels(x, d->Cin, NC, xd, Cd, 0, Cd, d->B, HW, nullptr, 0, 0, stream);
        if (rc) return rc;
      }
      rc = run_wgrad(d->algo, dy, x_dense ? x : xd, dwd, d->Cout, Cd, d->B, d->H, d->W, d->OH, d->OW, d->KH, d->KW, d->stride, d->pad,
                     d->pad_mode, 0, wws, wgrad_slab_bytes(d->algo, d->Cout, Cd, d->KH, d->KW, d->B * d->OH * d->OW), st);
      if (rc) return rc;
      hipLaunchKernelGGL(onehot_dense_w_kernel, dim3(cdiv((long long)d->Cout * Cd * KK, 256)), dim3(256), 0, st, dw, dwd,
                         d->Cout, d->Cin, NC, KK, 1, accumulate);
      rc = check_launch("onehot_dense_w");
      if (rc) return rc;
    }
  }
  if (dbias) {
    const size_t off = onehot_wgrad_ws_bytes(d, NC);
    rc = run_bias_grad(dy, dbias, d->B, d->Cout, d->OH * d->OW, accumulate, (char*)ws + off, ws_bytes - off, st);
  }
  return rc;
}

int him_conv2d_onehot_bwd_weight(const HimConv2d* d, const float* label, int n_onehot, const float* x, const float* dy,
                                 float* dw, float* dbias, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  return onehot_bwd_weight_impl(d, label, n_onehot, x, false, dy, dw, dbias, accumulate, ws, ws_bytes, stream);
}
int him_conv2d_onehot_bwd_weight_dense(const HimConv2d* d, const float* label, int n_onehot, const float* xdense,
                                       const float* dy, float* dw, float* dbias, int accumulate, void* ws, size_t ws_bytes,
                                       void* stream) {
  if (d && d->Cin > n_onehot && dw && !xdense)
    return fail(HIM_E_INVALID, "onehot conv: %d dense channels but no dense tensor", d->Cin - n_onehot);
  return onehot_bwd_weight_impl(d, label, n_onehot, xdense, true, dy, dw, dbias, accumulate, ws, ws_bytes, stream);
}
int him_conv2d_onehot_bwd_weight_part(const HimConv2d* d, const float* label, int n_onehot, const float* x, int x_is_dense,
                                      const float* dy, float* dw, float* dbias, int accumulate, void* ws, size_t ws_bytes,
                                      int parts, void* stream) {
  if (!(parts & (HIM_ONEHOT_PART_IDS | HIM_ONEHOT_PART_DENSE)) || (parts & ~(HIM_ONEHOT_PART_IDS | HIM_ONEHOT_PART_DENSE)))
    return fail(HIM_E_INVALID, "onehot conv bwd_weight: parts = %d", parts);
  if (d && d->Cin > n_onehot && dw && (parts & HIM_ONEHOT_PART_DENSE) && !x)
    return fail(HIM_E_INVALID, "onehot conv: %d dense channels but no input tensor", d->Cin - n_onehot);
  return onehot_bwd_weight_impl(d, label, n_onehot, x, x_is_dense != 0, dy, dw, dbias, accumulate, ws, ws_bytes, stream, parts);
}

size_t him_deconv2d_fwd_ws(const HimDeconv2d* t) {
  HimConv2d c;
  if (adjoint_of(t, &c)) return 0;
  return dgrad_ws_bytes(&c);
}

int him_deconv2d_fwd(const HimDeconv2d* t, const float* x, const float* w, const float* bias, float* y,
                     void* ws, size_t ws_bytes, void* stream) {
  HimConv2d c;
  int rc = adjoint_of(t, &c);
  if (rc) return rc;
  // ConvTranspose2d weight (Cin_t, Cout_t, KH, KW) is exactly the adjoint conv's (Cout_c, Cin_c, KH, KW).
  return run_dgrad(&c, x, w, y, bias, t->act, t->slope, ws, ws_bytes, (hipStream_t)stream);
}

size_t him_deconv2d_bwd_data_ws(const HimDeconv2d* t) {
  HimConv2d c;
  if (adjoint_of(t, &c)) return 0;
  return fprop_ws_bytes(&c);
}

int him_deconv2d_bwd_data(const HimDeconv2d* t, const float* dy, const float* w, float* dx, void* ws, size_t ws_bytes,
                          void* stream) {
  HimConv2d c;
  int rc = adjoint_of(t, &c);
  if (rc) return rc;
  return run_fprop(&c, dy, w, nullptr, dx, ws, ws_bytes, (hipStream_t)stream);
}

// ConvTranspose2d: the forward IS the adjoint conv's data gradient and vice versa, so the panel kinds swap
size_t him_deconv2d_panel_bytes(const HimDeconv2d* t, int kind) {
  HimConv2d c;
  if (adjoint_of(t, &c)) return 0;
  return him_conv2d_panel_bytes(&c, kind == HIM_PANEL_FWD ? HIM_PANEL_BWD_DATA : HIM_PANEL_FWD);
}

int him_deconv2d_panel_build(const HimDeconv2d* t, int kind, const float* w, void* panel, size_t panel_bytes,
                             void* stream) {
  HimConv2d c;
  int rc = adjoint_of(t, &c);
  if (rc) return rc;
  return him_conv2d_panel_build(&c, kind == HIM_PANEL_FWD ? HIM_PANEL_BWD_DATA : HIM_PANEL_FWD, w, panel, panel_bytes,
                                stream);
}

int him_deconv2d_fwd_panel(const HimDeconv2d* t, const float* x, const void* panel, const float* bias, float* y,
                           void* ws, size_t ws_bytes, void* stream) {
  HimConv2d c;
  int rc = adjoint_of(t, &c);
  if (rc) return rc;
  if (!panel) return fail(HIM_E_INVALID, "deconv fwd: null panel");
  return run_dgrad(&c, x, nullptr, y, bias, t->act, t->slope, ws, ws_bytes, (hipStream_t)stream, (const float*)panel);
}

int him_deconv2d_bwd_data_panel(const HimDeconv2d* t, const float* dy, const void* panel, float* dx, void* ws,
                                size_t ws_bytes, void* stream) {
  HimConv2d c;
  int rc = adjoint_of(t, &c);
  if (rc) return rc;
  if (!panel || !fprop_panel_floats(&c)) return fail(HIM_E_INVALID, "deconv bwd_data: no panel for this descriptor");
  return run_fprop(&c, dy, nullptr, nullptr, dx, ws, ws_bytes, (hipStream_t)stream, (const float*)panel);
}

size_t him_deconv2d_bwd_weight_ws(const HimDeconv2d* t) {
  HimConv2d c;
  if (adjoint_of(t, &c)) return 0;
  return wgrad_ws_bytes(c.algo, c.Cout, c.Cin, c.KH, c.KW, c.B * c.OH * c.OW, t->Cout);
}

int him_deconv2d_bwd_weight(const HimDeconv2d* t, const float* x, const float* dy, float* dw, float* dbias,
                            int accumulate, void* ws, size_t ws_bytes, void* stream) {
  HimConv2d c;
  int rc = adjoint_of(t, &c);
  if (rc) return rc;
  // adjoint conv: "input" = deconv output gradient dy, "output gradient" = deconv input x
  if (dw) {
    rc = run_wgrad(c.algo, x, dy, dw, c.Cout, c.Cin, c.B, c.H, c.W, c.OH, c.OW, c.KH, c.KW, c.stride, c.pad, HIM_PAD_ZERO,
                   accumulate, ws, ws_bytes, (hipStream_t)stream);
    if (rc) return rc;
  }
  if (dbias) {
    const size_t off = wgrad_slab_bytes(c.algo, c.Cout, c.Cin, c.KH, c.KW, c.B * c.OH * c.OW);
    if (ws_bytes < off) return fail(HIM_E_WORKSPACE, "bwd_weight ws too small");
    rc = run_bias_grad(dy, dbias, t->B, t->Cout, t->OH * t->OW, accumulate, (char*)ws + off, ws_bytes - off,
                       (hipStream_t)stream);
  }
  return rc;
}

}  // extern "C"
