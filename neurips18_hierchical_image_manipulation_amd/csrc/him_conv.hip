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

static bool use_fast(int M, int C2) {
  static int force_generic = -1;
  if (force_generic < 0) force_generic = getenv("HIM_GENERIC_CONV") ? 1 : 0;
  return !force_generic && M > 4 && C2 >= 16;
}
static int pad16(int c) { return (c + 15) / 16 * 16; }

template <int WM, int WN, int TM, int TN, bool REFLECT>
__global__ __launch_bounds__(256) void gconv_kernel(const GConvP p) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = 16;
  constexpr int LDA = BK + 1;
  constexpr int A_PER_T = BM * BK / 256;
  constexpr int KPT = BK * BN / 256;  // k values per thread in the gathered tile
  static_assert(WM * WN == 4, "4 waves");
  static_assert(BN % 64 == 0, "k must be wave-uniform in the gather");
  __shared__ float sA[2][BM * LDA];
  __shared__ float sB[2][BK * BN];

  const GPhase& ph = p.ph[blockIdx.z];
  const int plane = ph.NA * ph.NC;
  const int Ntot = p.B * plane;
  const int n0 = blockIdx.x * BN;
  if (n0 >= Ntot) return;
  const int m0 = blockIdx.y * BM;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int K = ph.K;
  const float* __restrict__ A = ph.A;
  const float* __restrict__ src = p.src;

  // ---- gather coordinates of this thread's column.  Out-of-range columns/rows are CLAMPED, not predicated:
  // they compute garbage that the epilogue never stores, and the loop body stays branch-free. ----
  const int nl = t % BN;
  const int kg = __builtin_amdgcn_readfirstlane(t / BN);
  const int n = min(n0 + nl, Ntot - 1);
  const int b = n / plane;
  const int rr = n - b * plane;
  const int a = rr / ph.NC;
  const int c = rr - a * ph.NC;
  const int by = a * p.sy + ph.offy, bx = c * p.sx + ph.offx;
  const int SH = p.SH, SW = p.SW;
  const uint32_t SHSW = (uint32_t)SH * SW;
  const uint32_t boff = (uint32_t)b * p.C2 * SHSW;
  const uint32_t JHJW = ph.JH * ph.JW, JW = ph.JW;
  const FastDiv fJHJW = ph.fJHJW, fJW = ph.fJW;
  const int ddy = p.dy, ddx = p.dx;

  const int kkA = t % BK;
  uint32_t rowoff[A_PER_T];
#pragma unroll
  for (int i = 0; i < A_PER_T; ++i) rowoff[i] = (uint32_t)min(m0 + t / BK + i * (256 / BK), p.M - 1) * (uint32_t)K;

  // two register sets: tile kt+2 is in flight while tile kt+1 waits to be written to LDS (2-deep prefetch)
  float ra2[2][A_PER_T], rb2[2][KPT];

  auto loadA = [&](float (&ra)[A_PER_T], int k0) {
    const uint32_t kc = (uint32_t)min(k0 + kkA, K - 1);
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) ra[i] = A[rowoff[i] + kc];
  };
  auto loadB = [&](float (&rb)[KPT], int k0) {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int kk = k0 + kg * KPT + i;  // wave-uniform -> scalar unit
      const bool kval = kk < K;          // zero B rows beyond K kill the (finite) clamped A columns
      const uint32_t kc = (uint32_t)min(kk, K - 1);
      const uint32_t c2 = fdiv(kc, fJHJW);
      const uint32_t r = kc - c2 * JHJW;
      const uint32_t jh = fdiv(r, fJW);
      const uint32_t jw = r - jh * JW;
      int iy = by + (int)jh * ddy, ix = bx + (int)jw * ddx;
      bool ok = kval;
      if (REFLECT) {
        iy = iy < 0 ? -iy : iy;
        iy = iy >= SH ? 2 * (SH - 1) - iy : iy;
        ix = ix < 0 ? -ix : ix;
        ix = ix >= SW ? 2 * (SW - 1) - ix : ix;
      } else {
        const int cy = min(max(iy, 0), SH - 1), cx = min(max(ix, 0), SW - 1);
        ok = ok && (cy == iy) && (cx == ix);
        iy = cy;
        ix = cx;
      }
      const uint32_t off = boff + c2 * SHSW + (uint32_t)iy * (uint32_t)SW + (uint32_t)ix;
      const float v = src[off];
      rb[i] = ok ? v : 0.f;
    }
  };
  auto storeAB = [&](const float (&ra)[A_PER_T], const float (&rb)[KPT], int buf) {
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) sA[buf][(t / BK + i * (256 / BK)) * LDA + kkA] = ra[i];
#pragma unroll
    for (int i = 0; i < KPT; ++i) sB[buf][(kg * KPT + i) * BN + nl] = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, lh = lane >> 5;
  const int nk = (K + BK - 1) / BK;
  loadA(ra2[0], 0);
  loadB(rb2[0], 0);
  storeAB(ra2[0], rb2[0], 0);
  loadA(ra2[1], min(1, nk - 1) * BK);
  loadB(rb2[1], min(1, nk - 1) * BK);
  __syncthreads();
  // iteration kt (parity P = kt & 1): issue tile kt+2 into set P (freed last iteration), run the MFMAs on LDS
  // buffer P, then write set P^1 (tile kt+1, issued a whole iteration ago) into LDS buffer P^1.  Tile indices
  // past the end are clamped: those loads/stores are redundant but keep the body branch-free.
  auto step = [&](auto PAR, int kt) {
    constexpr int P = decltype(PAR)::value;
    const int k2 = min(kt + 2, nk - 1) * BK;
    loadA(ra2[P], k2);
    loadB(rb2[P], k2);
    const float* __restrict__ pa = &sA[P][(wm * TM * 32 + l31) * LDA + lh];
    const float* __restrict__ pb = &sB[P][lh * BN + wn * TN * 32 + l31];
#pragma unroll
    for (int kp = 0; kp < BK / 2; ++kp) {
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = pa[i * 32 * LDA + kp * 2];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = pb[kp * 2 * BN + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    storeAB(ra2[P ^ 1], rb2[P ^ 1], P ^ 1);
    __syncthreads();
  };
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    step(std::integral_constant<int, 0>{}, kt);
    step(std::integral_constant<int, 1>{}, kt + 1);
  }
  if (kt < nk) step(std::integral_constant<int, 0>{}, kt);

  // ---- epilogue: bias + activation, coalesced NCHW stores ----
  const int act = p.act;
  const float slope = p.slope;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nn = n0 + wn * TN * 32 + j * 32 + l31;
    if (nn >= Ntot) continue;
    const int bb = nn / plane;
    const int r2 = nn - bb * plane;
    const int aa = r2 / ph.NC, cc = r2 - aa * ph.NC;
    const int oy = ph.oy0 + p.oys * aa, ox = ph.ox0 + p.oxs * cc;
    float* __restrict__ out = p.dst + ((size_t)bb * p.M * p.DH + oy) * p.DW + ox;
    const size_t mstride = (size_t)p.DH * p.DW;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < p.M) {
          float v = acc[i][j][r];
          if (p.bias) v += p.bias[m];
          out[(size_t)m * mstride] = apply_act(v, act, slope);
        }
      }
    }
  }
}

// ---- tiny-M variant (Cout <= 4 forward heads: G tanh head, PatchGAN logit heads; Cin <= 4 data gradients) ----
// M rows would waste >= 87 % of a 32-wide MFMA tile, so this path is a direct VALU convolution: one thread
// per output position, MM accumulators, wave-uniform weights (scalar loads), coalesced gathers along x.
// TJ = compile-time tap count per axis (0: run-time JH/JW).
// CS = channel slices per workgroup (1: thread = one position, all channels; 4: 64 positions x 4 channel quarters,
// LDS-reduced) -- the latter feeds the few-thousand-position PatchGAN heads whose reduction is 8192 long.
template <int MM, int TJ, bool REFLECT, int CS>
__global__ __launch_bounds__(256) void gconv_small_kernel(const GConvP p) {
  __shared__ float red[CS > 1 ? 256 * MM : 1];
  constexpr int NPB = 256 / CS;  // positions per block
  const GPhase& ph = p.ph[blockIdx.z];
  const int plane = ph.NA * ph.NC;
  const int Ntot = p.B * plane;
  // wave-uniform channel slice (NPB is a multiple of 64): keeps the weight reads on the scalar unit
  const int cs = CS > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x / NPB) : 0;
  // XCD-aware block order (workgroup L runs on XCD L % 8): each XCD gets a CONTIGUOUS band of positions, so vertically
  // adjacent rows -- which share KH-1 of their KH input rows -- are served by the same L2
  int blk = blockIdx.x;
  {
    const int total = gridDim.x, q = total >> 3, r = total & 7, xcd = blk & 7, slot = blk >> 3;
    blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int n = blk * NPB + threadIdx.x % NPB;
  if (blk * NPB >= Ntot) return;
  const int nc = min(n, Ntot - 1);
  const int b = nc / plane;
  const int rr = nc - b * plane;
  const int a = rr / ph.NC;
  const int c = rr - a * ph.NC;
  const int by = a * p.sy + ph.offy, bx = c * p.sx + ph.offx;
  const int SH = p.SH, SW = p.SW;
  const int JH = TJ ? TJ : ph.JH, JW = TJ ? TJ : ph.JW;
  const int K = ph.K, C2 = p.C2;
  const float* __restrict__ A = ph.A;
  const float* __restrict__ src = p.src + (size_t)b * C2 * SH * SW;
  float acc[MM];
#pragma unroll
  for (int m = 0; m < MM; ++m) acc[m] = 0.f;

  constexpr int MAXJ = TJ ? TJ : 8;
  int ixs[MAXJ];
  bool okx[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    int ix = bx + j * p.dx;
    bool ok = j < JW;
    if (REFLECT) {
      ix = ix < 0 ? -ix : ix;
      ix = ix >= SW ? 2 * (SW - 1) - ix : ix;
      ix = min(max(ix, 0), SW - 1);
    } else {
      const int cx = min(max(ix, 0), SW - 1);
      ok = ok && cx == ix;
      ix = cx;
    }
    ixs[j] = ix;
    okx[j] = ok;
  }
  // channel range of this thread: grid.y slices (partials summed by gconv_small_finish_kernel) x CS in-block slices
  const int nsl = gridDim.y * CS, sl = blockIdx.y * CS + cs;
  const int cchunk = (C2 + nsl - 1) / nsl;
  const int cbeg = sl * cchunk, cend = min(C2, cbeg + cchunk);
  // channel OUTER, tap rows inner: the KH rows of one channel plane are read back to back (L1/L2 reuse across the
  // vertical neighbours), instead of sweeping all channels once per tap row
#define HIM_SMALL_ROW()                                                                                  \
  {                                                                                                      \
    int iy = by + jh * p.dy;                                                                             \
    bool oky = true;                                                                                     \
    if (REFLECT) {                                                                                       \
      iy = iy < 0 ? -iy : iy;                                                                            \
      iy = iy >= SH ? 2 * (SH - 1) - iy : iy;                                                            \
      iy = min(max(iy, 0), SH - 1);                                                                      \
    } else {                                                                                             \
      const int cy = min(max(iy, 0), SH - 1);                                                            \
      oky = cy == iy;                                                                                    \
      iy = cy;                                                                                           \
    }                                                                                                    \
    const float* __restrict__ r = pl + iy * SW;                                                          \
    const int kb = kc + jh * JW;                                                                         \
    _Pragma("unroll") for (int j = 0; j < MAXJ; ++j) {                                                   \
      if (TJ == 0 && j >= JW) break;                                                                     \
      float v = r[ixs[j]];                                                                               \
      v = (oky && okx[j]) ? v : 0.f;                                                                     \
      _Pragma("unroll") for (int m = 0; m < MM; ++m) acc[m] = fmaf(A[(size_t)m * K + kb + j], v, acc[m]); \
    }                                                                                                    \
  }
  for (int c2 = cbeg; c2 < cend; ++c2) {
    const float* __restrict__ pl = src + (size_t)c2 * SH * SW;
    const int kc = c2 * JH * JW;
    if (TJ > 0 && TJ <= 4) {  // short tap rows: all KH*KW loads of a channel in flight
#pragma unroll
      for (int jh = 0; jh < TJ; ++jh) HIM_SMALL_ROW()
    } else {
      for (int jh = 0; jh < JH; ++jh) HIM_SMALL_ROW()
    }
  }
#undef HIM_SMALL_ROW
  if (CS > 1) {
#pragma unroll
    for (int m = 0; m < MM; ++m) red[threadIdx.x * MM + m] = acc[m];
    __syncthreads();
    if (cs != 0) return;
#pragma unroll
    for (int m = 0; m < MM; ++m) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < CS; ++q) v += red[(q * NPB + threadIdx.x) * MM + m];
      acc[m] = v;
    }
  }
  if (n < Ntot) {
    if (gridDim.y > 1) {
#pragma unroll
      for (int m = 0; m < MM; ++m) p.small_part[((size_t)blockIdx.y * MM + m) * Ntot + n] = acc[m];
      return;
    }
    const int oy = ph.oy0 + p.oys * a, ox = ph.ox0 + p.oxs * c;
    float* __restrict__ out = p.dst + ((size_t)b * p.M * p.DH + oy) * p.DW + ox;
#pragma unroll
    for (int m = 0; m < MM; ++m) {
      float v = acc[m];
      if (p.bias) v += p.bias[m];
      out[(size_t)m * p.DH * p.DW] = apply_act(v, p.act, p.slope);
    }
  }
}

// sums the channel-slice partials in fixed order, adds bias, applies the activation (single-phase launches only)
__global__ void gconv_small_finish_kernel(const GConvP p) {
  const GPhase& ph = p.ph[0];
  const int plane = ph.NA * ph.NC;
  const int Ntot = p.B * plane;
  const long long total = (long long)p.M * Ntot;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int m = (int)(i / Ntot), n = (int)(i - (long long)m * Ntot);
    float v = 0.f;
    for (int z = 0; z < p.small_nsplit; ++z) v += p.small_part[((size_t)z * p.M + m) * Ntot + n];
    if (p.bias) v += p.bias[m];
    const int b = n / plane, rr = n - b * plane, a = rr / ph.NC, c = rr - a * ph.NC;
    const int oy = ph.oy0 + p.oys * a, ox = ph.ox0 + p.oxs * c;
    p.dst[(((size_t)b * p.M + m) * p.DH + oy) * p.DW + ox] = apply_act(v, p.act, p.slope);
  }
}

// =============================================================================================
// Few output channels (M <= 4), KSxKS, stride 1, "same" geometry on a large plane: the generator's last layer
// (7x7, 64 -> 3, reflection padding) and the data gradient of VGG conv1_1 (3x3, 64 -> 3).  Three outputs cannot feed an
// MFMA tile, so this is VALU work -- but the one-position-per-thread kernel above issues one global load per MM FMAs
// and is load-issue bound (34 TFLOP/s on the 7x7, 9 on the 3x3).  Here a workgroup owns a 16x64 output tile, stages the
// (16+KS-1) x (64+KS-1) input patch of one channel in LDS (next channel's patch prefetched into registers, two LDS
// buffers, one barrier per channel) and every thread keeps 4 adjacent pixels x MM outputs in registers: one LDS row
// segment of 4+KS-1 floats (two / three wide ds_reads) feeds 4*KS*MM FMAs; the weights are wave-uniform scalar loads.
// =============================================================================================
// FLIP: the data-gradient form (taps walk backwards: src = out + P - j), i.e. the same patch with the tap index mirrored.
template <int MM, int KS, bool REFLECT, bool FLIP>
__global__ __launch_bounds__(256) void gconv_fewout_tiled_kernel(const GConvP p) {
  constexpr int TH = 16, TW = 64, P = KS / 2;
  constexpr int PH = TH + KS - 1, PW = TW + KS - 1;
  constexpr int RS = (PW + 3) & ~3;          // row stride: 16-byte aligned rows
  constexpr int NLD = (PH * PW + 255) / 256;  // patch elements per thread
  __shared__ __attribute__((aligned(16))) float tile[2][PH * RS];
  const GPhase& ph = p.ph[0];
  const int H = p.SH, W = p.SW, C = p.C2, K = ph.K;
  const int tiles_x = (W + TW - 1) / TW;
  const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x, b = blockIdx.y;
  const int y0 = by * TH, x0 = bx * TW;
  const int tid = threadIdx.x, ty = tid >> 4, tx = (tid & 15) * 4;
  const float* __restrict__ A = ph.A;
  const float* __restrict__ src = p.src + (size_t)b * C * H * W;

  int off[NLD], lds_at[NLD];
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int i = tid + k * 256;
    const int r = i / PW, c = i - r * PW;
    int iy = y0 - P + r, ix = x0 - P + c;
    bool ok = i < PH * PW;
    if (REFLECT) {
      iy = iy < 0 ? -iy : iy;
      iy = iy >= H ? 2 * (H - 1) - iy : iy;
      ix = ix < 0 ? -ix : ix;
      ix = ix >= W ? 2 * (W - 1) - ix : ix;
      ok = ok && iy >= 0 && iy < H && ix >= 0 && ix < W;   // beyond one reflection: outside every needed output
    } else {
      ok = ok && iy >= 0 && iy < H && ix >= 0 && ix < W;
    }
    off[k] = ok ? iy * W + ix : -1;
    lds_at[k] = i < PH * PW ? r * RS + c : -1;
  }
  float pre[NLD];
#pragma unroll
  for (int k = 0; k < NLD; ++k) pre[k] = off[k] >= 0 ? src[off[k]] : 0.f;

  float acc[MM][4];
#pragma unroll
  for (int m = 0; m < MM; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[m][q] = 0.f;

  for (int c = 0; c < C; ++c) {
    float* __restrict__ t = tile[c & 1];
#pragma unroll
    for (int k = 0; k < NLD; ++k)
      if (lds_at[k] >= 0) t[lds_at[k]] = pre[k];
    __syncthreads();   // one barrier per channel: the buffer written next (c+1) was last read in iteration c-1
    if (c + 1 < C) {
      const float* __restrict__ pl = src + (size_t)(c + 1) * H * W;
#pragma unroll
      for (int k = 0; k < NLD; ++k) pre[k] = off[k] >= 0 ? pl[off[k]] : 0.f;
    }
    const float* __restrict__ Ac = A + c * KS * KS;
#pragma unroll 1   // one tap row at a time: its KS*MM weights fit the scalar registers (all KS*KS*MM at once spill)
    for (int jh = 0; jh < KS; ++jh) {
      const float* __restrict__ row = t + (ty + (FLIP ? KS - 1 - jh : jh)) * RS + tx;
      float v[4 + KS - 1 + 1];
      *(float4*)&v[0] = *(const float4*)&row[0];
      if (KS == 3) {
        *(float2*)&v[4] = *(const float2*)&row[4];
      } else {
        *(float4*)&v[4] = *(const float4*)&row[4];
        *(float2*)&v[8] = *(const float2*)&row[8];
      }
#pragma unroll
      for (int jw = 0; jw < KS; ++jw)
#pragma unroll
        for (int m = 0; m < MM; ++m) {
          const float w = Ac[(size_t)m * K + jh * KS + jw];
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[m][q] = fmaf(w, v[q + (FLIP ? KS - 1 - jw : jw)], acc[m][q]);
        }
    }
  }
  const int oy = y0 + ty, ox = x0 + tx;
  if (oy >= H) return;
#pragma unroll
  for (int m = 0; m < MM; ++m) {
    float o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = apply_act(acc[m][q] + (p.bias ? p.bias[m] : 0.f), p.act, p.slope);
    float* __restrict__ out = p.dst + (((size_t)b * p.M + m) * p.DH + oy) * p.DW + ox;
    if (ox + 3 < W && (W & 3) == 0) {
      *(float4*)out = make_float4(o[0], o[1], o[2], o[3]);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (ox + q < W) out[q] = o[q];
    }
  }
}

template <int MM, int KS>
static void launch_fewout_tiled(const GConvP& p, hipStream_t st) {
  dim3 grid(cdiv(p.SW, 64) * cdiv(p.SH, 16), p.B, 1);
  if (p.dy < 0)     // data gradient of a zero-padded layer (reflection gradients go through the fold kernels)
    hipLaunchKernelGGL((gconv_fewout_tiled_kernel<MM, KS, false, true>), grid, dim3(256), 0, st, p);
  else if (p.pad_mode == HIM_PAD_REFLECT)
    hipLaunchKernelGGL((gconv_fewout_tiled_kernel<MM, KS, true, false>), grid, dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((gconv_fewout_tiled_kernel<MM, KS, false, false>), grid, dim3(256), 0, st, p);
}

// "same"-geometry stride-1 layer on a plane big enough to fill the chip with 16x64 tiles
static bool fewout_tiled_ok(const GConvP& p) {
  static const bool off = getenv("HIM_NO_FEWOUT_TILED") != nullptr;
  if (off || p.nphase != 1 || p.M < 2 || p.M > 4) return false;
  const GPhase& ph = p.ph[0];
  const int ks = ph.JH;
  if (ph.JW != ks || (ks != 3 && ks != 7)) return false;
  if (p.sy != 1 || p.sx != 1 || p.oys != 1 || p.oxs != 1 || ph.oy0 != 0 || ph.ox0 != 0) return false;
  const bool fwd = p.dy == 1 && p.dx == 1 && ph.offy == -(ks / 2) && ph.offx == -(ks / 2);
  const bool bwd = p.dy == -1 && p.dx == -1 && ph.offy == ks / 2 && ph.offx == ks / 2 && p.pad_mode == HIM_PAD_ZERO;
  if (!fwd && !bwd) return false;
  if (ph.NA != p.SH || ph.NC != p.SW || p.DH != p.SH || p.DW != p.SW) return false;
  if (p.SH <= ks / 2 || p.SW <= ks / 2) return false;                 // a single reflection must suffice
  if (p.C2 < 8 || p.B > 65535) return false;
  return (long long)p.B * cdiv(p.SW, 64) * cdiv(p.SH, 16) >= 512;      // enough tiles for 256 CUs
}

// =============================================================================================
// The mirror case: few REDUCTION channels (C2 <= 4), many outputs -- the data gradient of the generator's 7x7 head
// (3 -> 64 on the reflection-padded 262x518 plane, followed by the fold) and of any 3x3 / 7x7 layer with <= 4 outputs.
// K = C2*KS*KS = 147 is too ragged for the tap-major MFMA gather (30 TFLOP/s); same tiling as above with the roles
// swapped: the patch of all C2 channels sits in LDS once, every thread keeps 4 pixels x 16 outputs in registers
// (grid.z walks the groups of 16 outputs) and the 16 weights of a tap are scalar loads.
// Geometry: src y = a + offy + (FLIP ? -jh : jh), zero outside the source plane; output plane NA x NC.
// =============================================================================================
template <int KS, bool FLIP>
__global__ __launch_bounds__(256) void gconv_fewin_tiled_kernel(const GConvP p) {
  constexpr int TH = 16, TW = 64, MG = 16, MAXC = 4;
  constexpr int PH = TH + KS - 1, PW = TW + KS - 1;
  constexpr int RS = (PW + 3) & ~3;
  __shared__ __attribute__((aligned(16))) float tile[MAXC][PH * RS];
  const GPhase& ph = p.ph[0];
  const int H = p.SH, W = p.SW, C = p.C2, K = ph.K, OHt = ph.NA, OWt = ph.NC;
  const int tiles_x = (OWt + TW - 1) / TW;
  const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x, b = blockIdx.y, m0 = blockIdx.z * MG;
  const int y0 = by * TH, x0 = bx * TW;
  const int tid = threadIdx.x, ty = tid >> 4, tx = (tid & 15) * 4;
  const int py0 = y0 + ph.offy - (FLIP ? KS - 1 : 0), px0 = x0 + ph.offx - (FLIP ? KS - 1 : 0);
  const float* __restrict__ src = p.src + (size_t)b * C * H * W;
  for (int i = tid; i < C * PH * PW; i += 256) {
    const int c = i / (PH * PW), r2 = i - c * PH * PW, r = r2 / PW, cc = r2 - r * PW;
    const int iy = py0 + r, ix = px0 + cc;
    tile[c][r * RS + cc] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? src[((size_t)c * H + iy) * W + ix] : 0.f;
  }
  __syncthreads();
  float acc[MG][4];
#pragma unroll
  for (int m = 0; m < MG; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[m][q] = 0.f;
  const float* __restrict__ A = ph.A + (size_t)m0 * K;
  for (int c = 0; c < C; ++c) {
#pragma unroll 1
    for (int jh = 0; jh < KS; ++jh) {
      const float* __restrict__ row = &tile[c][(ty + (FLIP ? KS - 1 - jh : jh)) * RS + tx];
      float v[4 + KS - 1 + 1];
      *(float4*)&v[0] = *(const float4*)&row[0];
      if (KS == 3) {
        *(float2*)&v[4] = *(const float2*)&row[4];
      } else {
        *(float4*)&v[4] = *(const float4*)&row[4];
        *(float2*)&v[8] = *(const float2*)&row[8];
      }
      const float* __restrict__ Ar = A + (c * KS + jh) * KS;
#pragma unroll
      for (int jw = 0; jw < KS; ++jw)
#pragma unroll
        for (int m = 0; m < MG; ++m) {
          const float w = Ar[(size_t)m * K + jw];
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[m][q] = fmaf(w, v[q + (FLIP ? KS - 1 - jw : jw)], acc[m][q]);
        }
    }
  }
  const int oy = y0 + ty, ox = x0 + tx;
  if (oy >= OHt) return;
#pragma unroll
  for (int m = 0; m < MG; ++m) {
    float* __restrict__ out = p.dst + (((size_t)b * p.M + m0 + m) * p.DH + oy) * p.DW + ox;
    const float bb = p.bias ? p.bias[m0 + m] : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (ox + q < OWt) out[q] = apply_act(acc[m][q] + bb, p.act, p.slope);
  }
}

static bool fewin_tiled_ok(const GConvP& p) {
  static const bool off = getenv("HIM_NO_FEWIN_TILED") != nullptr;
  if (off || p.nphase != 1 || p.C2 > 4 || p.M < 16 || (p.M % 16) != 0 || p.ksplit > 1) return false;
  const GPhase& ph = p.ph[0];
  const int ks = ph.JH;
  if (ph.JW != ks || (ks != 3 && ks != 7)) return false;
  if (p.sy != 1 || p.sx != 1 || p.oys != 1 || p.oxs != 1 || ph.oy0 != 0 || ph.ox0 != 0) return false;
  if (!((p.dy == 1 && p.dx == 1) || (p.dy == -1 && p.dx == -1))) return false;
  if (p.pad_mode != HIM_PAD_ZERO || ph.NA != p.DH || ph.NC != p.DW || p.B > 65535) return false;
  return (long long)p.B * cdiv(ph.NC, 64) * cdiv(ph.NA, 16) >= 512;
}

static void launch_fewin_tiled(const GConvP& p, hipStream_t st) {
  const GPhase& ph = p.ph[0];
  dim3 grid(cdiv(ph.NC, 64) * cdiv(ph.NA, 16), p.B, p.M / 16);
  const bool flip = p.dy < 0;
  if (ph.JH == 7) {
    if (flip) hipLaunchKernelGGL((gconv_fewin_tiled_kernel<7, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gconv_fewin_tiled_kernel<7, false>), grid, dim3(256), 0, st, p);
  } else {
    if (flip) hipLaunchKernelGGL((gconv_fewin_tiled_kernel<3, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gconv_fewin_tiled_kernel<3, false>), grid, dim3(256), 0, st, p);
  }
}

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
static bool launch_gconv_small(const GConvP& p, long long maxN, hipStream_t st) {
  if (p.M > 4) return false;
  bool same = true;
  for (int i = 0; i < p.nphase; ++i) same = same && p.ph[i].JH == p.ph[0].JH && p.ph[i].JW == p.ph[0].JW;
  for (int i = 0; i < p.nphase; ++i)
    if (p.ph[i].JH > 8 || p.ph[i].JW > 8) return false;
  const int tj = (same && p.ph[0].JH == p.ph[0].JW) ? p.ph[0].JH : 0;
  if (fewout_tiled_ok(p)) {
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

static int launch_gconv(const GConvP& p, hipStream_t st) {
  long long maxN = 0;
  for (int i = 0; i < p.nphase; ++i) {
    long long n = (long long)p.B * p.ph[i].NA * p.ph[i].NC;
    if (n > maxN) maxN = n;
  }
  if (maxN == 0 || p.M <= 0) return HIM_OK;
  if (launch_gconv_small(p, maxN, st)) return check_launch("gconv_small");
  if (fewin_tiled_ok(p)) {
    launch_fewin_tiled(p, st);
    return check_launch("gconv_fewin_tiled");
  }
  if (p.fast) {
    // the fast kernel gathers through a buffer resource: 31-bit byte offsets (larger tensors: split the batch)
    if ((unsigned long long)p.B * p.C2 * p.SH * p.SW * 4ull >= (1ull << 31))
      return fail(HIM_E_UNSUPPORTED, "conv: source tensor of %llu bytes >= 2 GiB", (unsigned long long)p.B * p.C2 * p.SH * p.SW * 4ull);
    static int tile_override = -2;
    if (tile_override == -2) tile_override = getenv("HIM_GCONV_TILE") ? atoi(getenv("HIM_GCONV_TILE")) : -1;
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    if (p.M <= 64) {
      dim3 grid(cdiv(maxN, 128) * cdiv(p.M, 64), ks, p.nphase);
      launch_fast_cfg<2, 2, 1, 2>(p, grid, st);
    } else {
      const long long tiles128 = (long long)cdiv(maxN, 128) * cdiv(p.M, 128) * p.nphase;
      const bool big = tile_override >= 0 ? tile_override == 1 : (tiles128 >= 512 || (tiles128 >= 200 && tiles128 <= 256));
      const long long plane0 = (long long)p.ph[0].NA * p.ph[0].NC;
      const long long tiles256 = (maxN / 256) * cdiv(p.M, 128);
      if (tile_override == 3 && p.wbatch && ks == 1 && p.nphase == 1 && plane0 % 256 == 0 && tiles256 % 512 == 0) {
        // experiment (HIM_GCONV_TILE=3): 128x256 tiles for the batched Winograd GEMM, two workgroups per CU.  Alone it
        // matches / beats the 128x128 tiling (weight-gradient GEMM 0.53 -> 0.42 ms) but its 232 VGPRs + 60 KB LDS stop
        // it from sharing a CU with the other stream's kernels: the full step fell from 98 to 69 images/s.
        dim3 grid((unsigned)tiles256, 1, 1);
        launch_fast_cfg<2, 2, 2, 4>(p, grid, st);
      } else if (tile_override == 2) {
        dim3 grid(cdiv(maxN, 128) * cdiv(p.M, 128), ks, p.nphase);
        launch_fast_cfg<2, 4, 2, 1>(p, grid, st);  // 8 waves per 128x128 tile (measured: lockstep, no better than 4)
      } else if (big || ks > 1) {
        dim3 grid(cdiv(maxN, 128) * cdiv(p.M, 128), ks, p.nphase);
        launch_fast_cfg<2, 2, 2, 2>(p, grid, st);
      } else {
        dim3 grid(cdiv(maxN, 64) * cdiv(p.M, 128), ks, p.nphase);
        launch_fast_cfg<2, 2, 2, 1>(p, grid, st);
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


// ==============================================================================================
// Winograd F(2x2, 3x3) for the wide 3x3 stride-1 layers (the 1024-channel ResnetBlock stack, VGG conv4/5):
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A      (Lavin & Gray 2016; 2.25x fewer multiplies than the direct form)
// The 16 element-wise products are 16 independent dense contractions over the input channels, i.e. ONE batched GEMM
//   Mo[pos][co][tile] = sum_ci U[pos][co][ci] * V[pos][ci][tile]
// that runs on the same fp32-MFMA kernel as a 1x1 convolution over 16 "images" with per-image weight panels
// (GConvP::wbatch).  The transforms are separate HBM-bound kernels (x4 expansion of the activations).
// Data gradient: same machinery with the flipped/transposed filter (U'); for reflect padding it produces the padded
// gradient (full correlation, offset 2) which reflect_fold_kernel folds.  Weight gradient:
//   dU[pos][co][ci] = sum_tile dM[pos][co][tile] * V[pos][ci][tile],  dM = A dY A^T,  dg = G^T dU G
// = the same batched GEMM with the tiles as reduction index (dM is the row-major panel, V is produced transposed).
// Tile index t = (b*TY + ty)*TX + tx, padded to Tp (multiple of 128) so GEMM tiles never straddle a position slab.
// ==============================================================================================
struct WinoGeom {
  int B, C, H, W;    // tensor being transformed / produced: [B][C][H][W]
  int OH, OW;        // output grid the 2x2 tiles cover
  int TY, TX, T, Tp;
  int po;            // patch origin offset: input row = 2*ty - po + i   (1: pad-1 conv, 2: full correlation)
  int fold;          // data gradient of a REFLECT-padded conv folded into the patches of the border tiles (below)
};
static WinoGeom wino_geom(int B, int C, int H, int W, int OH, int OW, int po) {
  WinoGeom g;
  g.B = B; g.C = C; g.H = H; g.W = W; g.OH = OH; g.OW = OW; g.po = po; g.fold = 0;
  g.TY = (OH + 1) / 2; g.TX = (OW + 1) / 2;
  g.T = B * g.TY * g.TX;
  g.Tp = (g.T + 127) / 128 * 128;
  return g;
}

// V[pos][c][t] = (B^T d B)[pos] of the 4x4 patch of x[b][c] at rows 2ty-po.., cols 2tx-po..; REFLECT: mirrored
// indices (ReflectionPad2d(1)), else zeros outside.  grid (Tp/256, C)
template <bool REFLECT>
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, float* __restrict__ V,
                                                         const WinoGeom g) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (t >= g.Tp) return;
  float d[4][4];
  if (t < g.T) {
    const int per = g.TY * g.TX;
    const int b = t / per, r = t - b * per, ty = r / g.TX, tx = r - ty * g.TX;
    const float* __restrict__ src = x + ((size_t)b * g.C + c) * g.H * g.W;
    int ys[4], xs[4];
    bool oky[4], okx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int iy = 2 * ty - g.po + i, ix = 2 * tx - g.po + i;
      if (REFLECT) {
        iy = iy < 0 ? -iy : iy;
        iy = iy >= g.H ? 2 * (g.H - 1) - iy : iy;
        ix = ix < 0 ? -ix : ix;
        ix = ix >= g.W ? 2 * (g.W - 1) - ix : ix;
      }
      const int cy = min(max(iy, 0), g.H - 1), cx = min(max(ix, 0), g.W - 1);
      oky[i] = REFLECT || cy == iy;
      okx[i] = REFLECT || cx == ix;
      ys[i] = cy * g.W;
      xs[i] = cx;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = src[ys[i] + xs[j]];
        d[i][j] = (oky[i] && okx[j]) ? v : 0.f;
      }
    if (g.fold && (ty == 0 || ty == g.TY - 1 || tx == 0 || tx == g.TX - 1)) {
      // Reflect-pad data gradient without a padded grid.  dx[1] needs W0^T dy[0] on top of the zero-pad gradient and
      // dx[H-2] needs W2^T dy[H-1] (the mirrored pad rows feed the neighbours of the border).  In F(2,3) the LAST
      // patch row of the top tile meets only output 1 / filter row W0, and the FIRST patch row of the bottom tile only
      // output H-2 / filter row W2 -- so adding dy[0] to the top tile's patch row 3 and dy[H-1] to the bottom tile's
      // patch row 0 (columns alike, corners get the product terms) yields exactly the folded gradient (H, W even).
      const int ey = ty == 0 ? 0 : g.H - 1, ei = ty == 0 ? 3 : 0;          // extra source row -> patch row ei
      const int ex = tx == 0 ? 0 : g.W - 1, ej = tx == 0 ? 3 : 0;
      const bool rowx = ty == 0 || ty == g.TY - 1, colx = tx == 0 || tx == g.TX - 1;
      float addr[4], addc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) addr[j] = (rowx && okx[j]) ? src[ey * g.W + xs[j]] : 0.f;   // dy[ey][patch col j]
#pragma unroll
      for (int i = 0; i < 4; ++i) addc[i] = (colx && oky[i]) ? src[ys[i] + ex] : 0.f;         // dy[patch row i][ex]
      const float corner = (rowx && colx) ? src[ey * g.W + ex] : 0.f;
      // (H, W >= 4 is required by the caller: the top/bottom and left/right border tiles are distinct)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float e = 0.f;
          if (i == ei) e += addr[j];
          if (j == ej) e += addc[i];
          if (i == ei && j == ej) e += corner;
          d[i][j] += e;
        }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) d[i][j] = 0.f;
  }
  float tt[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    tt[0][j] = d[0][j] - d[2][j];
    tt[1][j] = d[1][j] + d[2][j];
    tt[2][j] = d[2][j] - d[1][j];
    tt[3][j] = d[1][j] - d[3][j];
  }
  const size_t slab = (size_t)g.C * g.Tp;
  float* __restrict__ o = V + (size_t)c * g.Tp + t;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[(size_t)(i * 4 + 0) * slab] = tt[i][0] - tt[i][2];
    o[(size_t)(i * 4 + 1) * slab] = tt[i][1] + tt[i][2];
    o[(size_t)(i * 4 + 2) * slab] = tt[i][2] - tt[i][1];
    o[(size_t)(i * 4 + 3) * slab] = tt[i][1] - tt[i][3];
  }
}

// Same transform, TRANSPOSED output Vt[pos][t][c] (c fastest) -- the column operand of the weight-gradient GEMM, whose
// reduction runs over the tiles t.  Threads run along c so the stores are coalesced; the 16 patch loads of a thread
// are 2 KB apart across lanes but come from the L2-resident activation.  grid (ceil(C/256), Tp)
template <bool REFLECT>
__global__ __launch_bounds__(256) void wino_input_t_kernel(const float* __restrict__ x, float* __restrict__ Vt,
                                                           const WinoGeom g) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int t = blockIdx.y;
  if (c >= g.C) return;
  float d[4][4];
  if (t < g.T) {
    const int per = g.TY * g.TX;
    const int b = t / per, r = t - b * per, ty = r / g.TX, tx = r - ty * g.TX;
    const float* __restrict__ src = x + ((size_t)b * g.C + c) * g.H * g.W;
    int ys[4], xs[4];
    bool oky[4], okx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int iy = 2 * ty - g.po + i, ix = 2 * tx - g.po + i;
      if (REFLECT) {
        iy = iy < 0 ? -iy : iy;
        iy = iy >= g.H ? 2 * (g.H - 1) - iy : iy;
        ix = ix < 0 ? -ix : ix;
        ix = ix >= g.W ? 2 * (g.W - 1) - ix : ix;
      }
      const int cy = min(max(iy, 0), g.H - 1), cx = min(max(ix, 0), g.W - 1);
      oky[i] = REFLECT || cy == iy;
      okx[i] = REFLECT || cx == ix;
      ys[i] = cy * g.W;
      xs[i] = cx;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = src[ys[i] + xs[j]];
        d[i][j] = (oky[i] && okx[j]) ? v : 0.f;
      }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) d[i][j] = 0.f;
  }
  float tt[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    tt[0][j] = d[0][j] - d[2][j];
    tt[1][j] = d[1][j] + d[2][j];
    tt[2][j] = d[2][j] - d[1][j];
    tt[3][j] = d[1][j] - d[3][j];
  }
  const size_t slab = (size_t)g.C * g.Tp;
  float* __restrict__ o = Vt + (size_t)t * g.C + c;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[(size_t)(i * 4 + 0) * slab] = tt[i][0] - tt[i][2];
    o[(size_t)(i * 4 + 1) * slab] = tt[i][1] + tt[i][2];
    o[(size_t)(i * 4 + 2) * slab] = tt[i][2] - tt[i][1];
    o[(size_t)(i * 4 + 3) * slab] = tt[i][1] - tt[i][3];
  }
}

// y[b][c][2ty+i][2tx+j] = act((A^T Mo A)[i][j] + bias[c]);  Mo[pos][c][t].  grid (Tp/256, C)
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mo, float* __restrict__ y,
                                                          const float* __restrict__ bias, const WinoGeom g, int act,
                                                          float slope) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (t >= g.T) return;
  const size_t slab = (size_t)g.C * g.Tp;
  const float* __restrict__ in = Mo + (size_t)c * g.Tp + t;
  float m[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) m[i][j] = in[(size_t)(i * 4 + j) * slab];
  float r[2][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    r[0][j] = m[0][j] + m[1][j] + m[2][j];
    r[1][j] = m[1][j] - m[2][j] - m[3][j];
  }
  const int per = g.TY * g.TX;
  const int b = t / per, rr = t - b * per, ty = rr / g.TX, tx = rr - ty * g.TX;
  const float bb = bias ? bias[c] : 0.f;
  float* __restrict__ dst = y + ((size_t)b * g.C + c) * g.OH * g.OW;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int oy = 2 * ty + i;
    if (oy >= g.OH) continue;
    const float v0 = r[i][0] + r[i][1] + r[i][2], v1 = r[i][1] - r[i][2] - r[i][3];
    const int ox = 2 * tx;
    dst[oy * g.OW + ox] = apply_act(v0 + bb, act, slope);
    if (ox + 1 < g.OW) dst[oy * g.OW + ox + 1] = apply_act(v1 + bb, act, slope);
  }
}

// dM[pos][c][t] = (A dY A^T)[pos] of the 2x2 output-gradient tile (zeros outside the grid and in the padded columns)
__global__ __launch_bounds__(256) void wino_dy_kernel(const float* __restrict__ dy, float* __restrict__ dM,
                                                      const WinoGeom g) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (t >= g.Tp) return;
  float e[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  if (t < g.T) {
    const int per = g.TY * g.TX;
    const int b = t / per, rr = t - b * per, ty = rr / g.TX, tx = rr - ty * g.TX;
    const float* __restrict__ src = dy + ((size_t)b * g.C + c) * g.OH * g.OW;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int oy = 2 * ty + i, ox = 2 * tx + j;
        if (oy < g.OH && ox < g.OW) e[i][j] = src[oy * g.OW + ox];
      }
  }
  float q[4][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    q[0][j] = e[0][j];
    q[1][j] = e[0][j] + e[1][j];
    q[2][j] = e[0][j] - e[1][j];
    q[3][j] = -e[1][j];
  }
  const size_t slab = (size_t)g.C * g.Tp;
  float* __restrict__ o = dM + (size_t)c * g.Tp + t;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[(size_t)(i * 4 + 0) * slab] = q[i][0];
    o[(size_t)(i * 4 + 1) * slab] = q[i][0] + q[i][1];
    o[(size_t)(i * 4 + 2) * slab] = q[i][0] - q[i][1];
    o[(size_t)(i * 4 + 3) * slab] = -q[i][1];
  }
}

// U[pos][m][c] = (G g G^T)[pos].  FLIP = 0: g = w[m][c] (forward, m = Cout, c = Cin);
// FLIP = 1: g = 180-degree rotation of w[c][m] (data gradient: m = Cin, c = Cout).  grid (ceil(Cc/256), Mm)
template <int FLIP>
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Mm,
                                                          int Cc) {
  const int c = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
  if (c >= Cc) return;
  const float* __restrict__ src = FLIP ? w + ((size_t)c * Mm + m) * 9 : w + ((size_t)m * Cc + c) * 9;
  float g[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) g[i][j] = FLIP ? src[(2 - i) * 3 + (2 - j)] : src[i * 3 + j];
  float sg[4][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    sg[0][j] = g[0][j];
    sg[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
    sg[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
    sg[3][j] = g[2][j];
  }
  const size_t slab = (size_t)Mm * Cc;
  float* __restrict__ o = U + (size_t)m * Cc + c;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[(size_t)(i * 4 + 0) * slab] = sg[i][0];
    o[(size_t)(i * 4 + 1) * slab] = 0.5f * (sg[i][0] + sg[i][1] + sg[i][2]);
    o[(size_t)(i * 4 + 2) * slab] = 0.5f * (sg[i][0] - sg[i][1] + sg[i][2]);
    o[(size_t)(i * 4 + 3) * slab] = sg[i][2];
  }
}
// (An LDS-staged variant with coalesced 9-float filter reads was measured SLOWER -- 33 vs 22 us stand-alone at 1024x1024,
// 110 vs 50 us inside the step: the strided reads of this form are absorbed by L2, the staging only added latency.)

// dw[m][c][3][3] (+)= G^T dU[.][m][c] G.  The 9 results of a thread go through LDS so that the workgroup's 256*9
// contiguous output floats are read-modified-written with coalesced accesses.  grid (ceil(C/256), M)
__global__ __launch_bounds__(256) void wino_wgrad_out_kernel(const float* __restrict__ dU, float* __restrict__ dw,
                                                             int M, int C, int accumulate) {
  __shared__ float so[256 * 9];
  const int c0 = blockIdx.x * 256, t = threadIdx.x, m = blockIdx.y;
  const int c = c0 + t;
  const size_t slab = (size_t)M * C;
  if (c < C) {
    const float* __restrict__ in = dU + (size_t)m * C + c;
    float u[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) u[i][j] = in[(size_t)(i * 4 + j) * slab];
    float e[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      e[0][j] = u[0][j] + 0.5f * (u[1][j] + u[2][j]);
      e[1][j] = 0.5f * (u[1][j] - u[2][j]);
      e[2][j] = 0.5f * (u[1][j] + u[2][j]) + u[3][j];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      so[t * 9 + i * 3 + 0] = e[i][0] + 0.5f * (e[i][1] + e[i][2]);
      so[t * 9 + i * 3 + 1] = 0.5f * (e[i][1] - e[i][2]);
      so[t * 9 + i * 3 + 2] = 0.5f * (e[i][1] + e[i][2]) + e[i][3];
    }
  }
  __syncthreads();
  const int nvalid = min(256, C - c0) * 9;
  float* __restrict__ o = dw + ((size_t)m * C + c0) * 9;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int idx = k * 256 + t;
    if (idx < nvalid) o[idx] = accumulate ? o[idx] + so[idx] : so[idx];
  }
}

// ---- Winograd host side -------------------------------------------------------------------------------------------
static int g_wino_min_c = -2;  // -2: not initialised; <= 0: Winograd off
static int wino_min_c() {
  if (g_wino_min_c == -2) {
    const char* e = getenv("HIM_WINO_MIN_C");
    g_wino_min_c = getenv("HIM_NO_WINOGRAD") ? -1 : (e ? atoi(e) : 512);
  }
  return g_wino_min_c;
}
// wide 3x3 stride-1 pad-1 layers only: below ~512 channels the x4 transform traffic eats the 2.25x multiply saving
static bool wino_shape_ok(int Cout, int Cin, int KH, int KW, int stride, int pad, int H, int W) {
  const int mc = wino_min_c();
  return mc > 0 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && Cin >= mc && Cout >= mc && (Cin % 16) == 0 &&
         (Cout % 16) == 0 && H >= 2 && W >= 2 && use_fast(Cout, Cin);
}
static bool wino_wgrad_ok(int M, int C, int KH, int KW, int stride, int pad, int H, int W) {
  return wino_shape_ok(M, C, KH, KW, stride, pad, H, W) && (M % 128) == 0 && (C % 128) == 0;
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
static int wino_batched_gemm(const float* A, const float* Bm, float* Cm, int M, int K, int N, hipStream_t st) {
  GConvP g;
  memset(&g, 0, sizeof(g));
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
  return launch_gconv(g, st);
}
// dst[B][Mout][OH][OW] = act(winograd-conv(src[B][Csrc][H][W], U) + bias); ws holds V and Mo
static int run_wino_conv(int B, int Csrc, int H, int W, int Mout, int OH, int OW, int po, bool reflect,
                         const float* src, const float* U, const float* bias, int act, float slope, float* dst,
                         float* ws, hipStream_t st, bool fold = false) {
  WinoGeom gi = wino_geom(B, Csrc, H, W, OH, OW, po);
  gi.fold = fold ? 1 : 0;
  float* V = ws;
  float* Mo = V + (size_t)16 * Csrc * gi.Tp;
  const dim3 gin(cdiv(gi.Tp, 256), Csrc);
  if (reflect) hipLaunchKernelGGL((wino_input_kernel<true>), gin, dim3(256), 0, st, src, V, gi);
  else hipLaunchKernelGGL((wino_input_kernel<false>), gin, dim3(256), 0, st, src, V, gi);
  int rc = check_launch("wino_input");
  if (rc) return rc;
  rc = wino_batched_gemm(U, V, Mo, Mout, Csrc, gi.Tp, st);
  if (rc) return rc;
  WinoGeom go = gi;
  go.C = Mout;
  hipLaunchKernelGGL(wino_output_kernel, dim3(cdiv(gi.Tp, 256), Mout), dim3(256), 0, st, (const float*)Mo, dst, bias, go,
                     act, slope);
  return check_launch("wino_output");
}

// ==============================================================================================
// wgrad
// ==============================================================================================
struct WGradP {
  const float* dy;  // [B][M][OH][OW]
  const float* x;   // [B][C][H][W]
  float* out;       // dW [M][Np]  or slabs [splits][M][Np]
  int M, C, B, H, W, OH, OW, KH, KW, stride, pad, pad_mode;
  int Np, Kdim, kchunk, splits, accumulate;
  FastDiv fKK, fKW, fOW;
};

template <int WM, int WN, int TM, int TN, bool REFLECT>
__global__ __launch_bounds__(256) void wgrad_kernel(const WGradP p) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = 32;
  constexpr int LD = BK + 1;
  constexpr int RA = BM / 8, RB = BN / 8;
  static_assert(WM * WN == 4, "4 waves");
  __shared__ float sA[2][BM * LD];
  __shared__ float sB[2][BN * LD];
  __shared__ int tabOff[BN];
  __shared__ int tabD[BN];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int OHW = p.OH * p.OW, HW = p.H * p.W;
  const int kbeg = blockIdx.z * p.kchunk;
  const int kend = min(p.Kdim, kbeg + p.kchunk);

  // per-column (ci,kh,kw) table: loop invariant
  if (t < BN) {
    const int np = n0 + t;
    int off = 0, d = (128 << 16) | 128;
    if (np < p.Np) {
      const int ci = (int)fdiv((uint32_t)np, p.fKK);
      const int r = np - ci * p.KH * p.KW;
      const int kh = (int)fdiv((uint32_t)r, p.fKW);
      const int kw = r - kh * p.KW;
      off = ci * HW;
      d = ((kh - p.pad + 128) << 16) | (kw - p.pad + 128);
    }
    tabOff[t] = off;
    tabD[t] = d;
  }
  __syncthreads();

  const int kkl = t & 31, rg = t >> 5;
  // position of this thread's k index (b, sp) tracked incrementally
  int kcur = kbeg + kkl;
  int b = kcur / OHW;
  int sp = kcur - b * OHW;

  float ra[RA], rb[RB];
  const int H = p.H, W = p.W, stride = p.stride;
  const float* __restrict__ dy = p.dy;
  const float* __restrict__ x = p.x;
  uint32_t mrow[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) mrow[i] = (uint32_t)min(m0 + rg + 8 * i, p.M - 1) * (uint32_t)OHW;

  // branch-free: out-of-range k positions are redirected to element (0,0) and their dY operand zeroed;
  // out-of-range rows/columns are clamped and never stored.
  auto loadAB = [&]() {
    const bool kvalid = kcur < kend;
    const int bb = kvalid ? b : 0;
    const int spp = kvalid ? sp : 0;
    const int oh = (int)fdiv((uint32_t)spp, p.fOW);
    const int ow = spp - oh * p.OW;
    const uint32_t dyb = (uint32_t)bb * (uint32_t)p.M * (uint32_t)OHW + (uint32_t)spp;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const float v = dy[dyb + mrow[i]];
      ra[i] = kvalid ? v : 0.f;
    }
    const uint32_t xb = (uint32_t)bb * (uint32_t)p.C * (uint32_t)HW;
    const int ihb = oh * stride, iwb = ow * stride;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int row = rg + 8 * i;
      const int off = tabOff[row];
      const int d = tabD[row];
      int ih = ihb + (d >> 16) - 128, iw = iwb + (d & 0xffff) - 128;
      bool ok = true;
      if (REFLECT) {
        ih = ih < 0 ? -ih : ih;
        ih = ih >= H ? 2 * (H - 1) - ih : ih;
        iw = iw < 0 ? -iw : iw;
        iw = iw >= W ? 2 * (W - 1) - iw : iw;
      } else {
        const int ch = min(max(ih, 0), H - 1), cw = min(max(iw, 0), W - 1);
        ok = (ch == ih) && (cw == iw);
        ih = ch;
        iw = cw;
      }
      const float v = x[xb + (uint32_t)off + (uint32_t)ih * (uint32_t)W + (uint32_t)iw];
      rb[i] = ok ? v : 0.f;
    }
  };
  auto advance = [&]() {
    kcur += BK;
    sp += BK;
    while (sp >= OHW) {
      sp -= OHW;
      ++b;
    }
  };
  auto storeAB = [&](int buf) {
#pragma unroll
    for (int i = 0; i < RA; ++i) sA[buf][(rg + 8 * i) * LD + kkl] = ra[i];
#pragma unroll
    for (int i = 0; i < RB; ++i) sB[buf][(rg + 8 * i) * LD + kkl] = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, lh = lane >> 5;
  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk > 0) {
    loadAB();
    storeAB(0);
    advance();
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    loadAB();  // past the end: kvalid is false, loads hit element (0,0)
    const float* __restrict__ pa = &sA[buf][(wm * TM * 32 + l31) * LD + lh];
    const float* __restrict__ pb = &sB[buf][(wn * TN * 32 + l31) * LD + lh];
#pragma unroll
    for (int kp = 0; kp < BK / 2; ++kp) {
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = pa[i * 32 * LD + kp * 2];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = pb[j * 32 * LD + kp * 2];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    storeAB(buf ^ 1);
    advance();
    __syncthreads();
  }

  float* __restrict__ out = p.out + (p.splits > 1 ? (size_t)blockIdx.z * p.M * p.Np : (size_t)0);
  const bool accum = p.splits == 1 && p.accumulate;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int np = n0 + wn * TN * 32 + j * 32 + l31;
    if (np >= p.Np) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < p.M) {
          float* o = out + (size_t)m * p.Np + np;
          *o = accum ? (*o + acc[i][j][r]) : acc[i][j][r];
        }
      }
    }
  }
}

// =============================================================================================
// wgrad, fast path (C % 64 == 0, OH*OW % 4 == 0): the filter-column index runs TAP-MAJOR, n' = tap*C + ci, so a
// 64/128-column tile is ONE filter tap x consecutive input channels: the gather coordinates (ih, iw, validity) are
// computed once per K-step per thread and every element is one saddr global_load.  dY tiles are contiguous float4s.
// LDS rows are [row][32 k + 4 pad] with the k <-> (lane>>5) pairing k = 16*(lane>>5) + kp (4 aligned ds_read_b128 per
// operand tile per K-step).  Partial results go to tap-major slabs; wgrad_finish_kernel sums the split-K slabs in
// fixed order and scatters into the reference (Cout,Cin,KH,KW) layout (+= for the gradient arena).
// =============================================================================================
template <int TM, int TN, bool REFLECT, bool AL4>
__global__ __launch_bounds__(256) void wgrad_fast_kernel(const WGradP p) {
  constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32, BK = 32, LD = 36;
  constexpr int A_V4 = BM * BK / 4 / 256;  // 4 or 2
  constexpr int RB = BN / 8;               // gathered rows per thread (16 or 8)
  __shared__ __attribute__((aligned(16))) float sA[2][BM * LD];
  __shared__ __attribute__((aligned(16))) float sB[2][BN * LD];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int OHW = p.OH * p.OW, HW = p.H * p.W, C = p.C;
  const int kbeg = blockIdx.z * p.kchunk;
  const int kend = min(p.Kdim, kbeg + p.kchunk);
  const int nk = (kend - kbeg + BK - 1) / BK;
  // the tile's filter tap (wave-uniform) and first input channel
  const int tap = n0 / C, ci0 = n0 - tap * C;
  const int th = tap / p.KW, tw = tap - th * p.KW;
  const int dh = th - p.pad, dw = tw - p.pad;
  const int H = p.H, W = p.W, stride = p.stride;

  // dY tile: thread -> (row = t/8 + 32 i, k quad = t%8)
  const int arow = t >> 3, akq = t & 7;
  uint32_t arowoff[A_V4];
#pragma unroll
  for (int i = 0; i < A_V4; ++i) arowoff[i] = (uint32_t)min(m0 + arow + 32 * i, p.M - 1) * (uint32_t)OHW;
  // gathered tile: thread -> (k = t%32, row group rg = t/32; rows rg + 8 i)
  const int kkl = t & 31, rg = t >> 5;

  // Both operands are read through BUFFER resources: an invalid element (K tail beyond kend, zero-padding tap) gets
  // voffset 0x80000000 and the range check returns 0.0 -- no select sits between a load and its ds_write, so nothing
  // makes the compiler wait for a load before the K-step's MFMAs have run (run_wgrad guarantees both tensors < 2 GiB).
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.dy, 0, (int)((uint32_t)p.B * (uint32_t)p.M * (uint32_t)OHW * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.x, 0, (int)((uint32_t)p.B * (uint32_t)C * (uint32_t)HW * 4u), 0x00020000);
  const uint32_t xbase = (uint32_t)ci0 * (uint32_t)HW * 4u;   // wave-uniform: first channel of the tile
  const uint32_t step8 = 8u * (uint32_t)HW * 4u;

  float4 ra[A_V4];
  float rb[RB];
  // per-thread cursors of the NEXT tile: A quad position and gather position, as (image, offset in image)
  int ka = kbeg + akq * 4, ba = ka / OHW, spa = ka - ba * OHW;
  int kb = kbeg + kkl, bb = kb / OHW, spb = kb - bb * OHW;

#define HIM_WLOAD()                                                                                           \
  {                                                                                                           \
    if (AL4) {                                                                                                \
      const bool av = ka < kend;                                                                              \
      const uint32_t abase = (uint32_t)ba * (uint32_t)p.M * (uint32_t)OHW + (uint32_t)spa;                     \
      _Pragma("unroll") for (int i = 0; i < A_V4; ++i) {                                                      \
        ra[i] = __builtin_bit_cast(                                                                            \
            float4, __builtin_amdgcn_raw_buffer_load_b128(rdy, av ? (abase + arowoff[i]) * 4u : 0x80000000u, 0, 0)); \
      }                                                                                                       \
    } else { /* plane size not a multiple of 4 (odd PatchGAN planes): a quad may straddle two images */        \
      uint32_t eo[4];                                                                                         \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
        int sp_ = spa + j, b_ = ba;                                                                           \
        const bool wrap = sp_ >= OHW;                                                                         \
        sp_ -= wrap ? OHW : 0;                                                                                \
        b_ += wrap ? 1 : 0;                                                                                   \
        eo[j] = (ka + j < kend) ? ((uint32_t)b_ * (uint32_t)p.M * (uint32_t)OHW + (uint32_t)sp_) * 4u : 0x80000000u; \
      }                                                                                                       \
      _Pragma("unroll") for (int i = 0; i < A_V4; ++i) {                                                      \
        float q_[4];                                                                                          \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                         \
          q_[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(                              \
              rdy, eo[j] == 0x80000000u ? eo[j] : eo[j] + arowoff[i] * 4u, 0, 0));                             \
        ra[i] = make_float4(q_[0], q_[1], q_[2], q_[3]);                                                      \
      }                                                                                                       \
    }                                                                                                         \
    const bool bv = kb < kend;                                                                                \
    const int sp = bv ? spb : 0;                                                                              \
    const int oh = (int)fdiv((uint32_t)sp, p.fOW);                                                            \
    const int ow = sp - oh * p.OW;                                                                            \
    int ih = oh * stride + dh, iw = ow * stride + dw;                                                         \
    bool ok = bv;                                                                                             \
    if (REFLECT) {                                                                                            \
      ih = ih < 0 ? -ih : ih;                                                                                 \
      ih = ih >= H ? 2 * (H - 1) - ih : ih;                                                                   \
      iw = iw < 0 ? -iw : iw;                                                                                 \
      iw = iw >= W ? 2 * (W - 1) - iw : iw;                                                                   \
    } else {                                                                                                  \
      ok = ok && ih >= 0 && ih < H && iw >= 0 && iw < W;                                                      \
    }                                                                                                         \
    const uint32_t voff = ok ? ((uint32_t)bb * (uint32_t)C * (uint32_t)HW + (uint32_t)rg * (uint32_t)HW +      \
                                (uint32_t)ih * (uint32_t)W + (uint32_t)iw) * 4u : 0x80000000u;                 \
    uint32_t so8 = xbase;                                                                                     \
    _Pragma("unroll") for (int i = 0; i < RB; ++i) {                                                          \
      rb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, voff, so8, 0));              \
      so8 += step8;                                                                                           \
    }                                                                                                         \
    ka += BK;                                                                                                 \
    spa += BK;                                                                                                \
    while (spa >= OHW) {                                                                                      \
      spa -= OHW;                                                                                             \
      ++ba;                                                                                                   \
    }                                                                                                         \
    kb += BK;                                                                                                 \
    spb += BK;                                                                                                \
    while (spb >= OHW) {                                                                                      \
      spb -= OHW;                                                                                             \
      ++bb;                                                                                                   \
    }                                                                                                         \
  }
#define HIM_WSTORE(buf_)                                                                                       \
  {                                                                                                           \
    _Pragma("unroll") for (int i = 0; i < A_V4; ++i) *(float4*)&sA[buf_][(arow + 32 * i) * LD + akq * 4] = ra[i]; \
    _Pragma("unroll") for (int i = 0; i < RB; ++i) sB[buf_][(rg + 8 * i) * LD + kkl] = rb[i];                 \
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, lh = lane >> 5;
  if (nk > 0) {
    HIM_WLOAD()
    HIM_WSTORE(0)
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const float4* __restrict__ pa = (const float4*)&sA[buf][(wm * TM * 32 + l31) * LD + lh * 16];
    const float4* __restrict__ pb = (const float4*)&sB[buf][(wn * TN * 32 + l31) * LD + lh * 16];
    float4 af[TM][4], bf[TN][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) af[i][q] = pa[i * 32 * LD / 4 + q];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) bf[j][q] = pb[j * 32 * LD / 4 + q];
    HIM_WLOAD()  // past the end: both operands' validity is false -> zeros, addresses clamped to element 0
#define HIM_WM(Q, CMP)                                                                                         \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] =    \
      __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][Q].CMP, bf[j][Q].CMP, acc[i][j], 0, 0, 0);
    HIM_WM(0, x) HIM_WM(0, y) HIM_WM(0, z) HIM_WM(0, w)
    HIM_WM(1, x) HIM_WM(1, y) HIM_WM(1, z) HIM_WM(1, w)
    HIM_WM(2, x) HIM_WM(2, y) HIM_WM(2, z) HIM_WM(2, w)
    HIM_WM(3, x) HIM_WM(3, y) HIM_WM(3, z) HIM_WM(3, w)
#undef HIM_WM
    HIM_WSTORE(buf ^ 1)
    {
      // K-step instruction order pinned with sched_group_barrier (as in gconv_fast_kernel): the 16 LDS operand reads,
      // then the MFMA stream with the next tile's buffer loads dripped in, the LDS writes only after all but 6 MFMAs
      // -- without it the compiler hoists the writes (and the s_waitcnt on the loads just issued) to the top.
      constexpr int NV = (AL4 ? A_V4 : 4 * A_V4) + RB, NM = 16 * TM * TN, PER = (NM - 10) / NV;
      __builtin_amdgcn_sched_group_barrier(0x100, 4 * (TM + TN), 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
      for (int q_ = 0; q_ < NV; ++q_) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NM - 2 - PER * NV - 6, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, A_V4 + RB, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
    }
    __syncthreads();
  }
#undef HIM_WLOAD
#undef HIM_WSTORE

  // tap-major slab: out[z][m][n']
  float* __restrict__ out = p.out + (size_t)blockIdx.z * p.M * p.Np;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int np = n0 + wn * TN * 32 + j * 32 + l31;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < p.M) out[(size_t)m * p.Np + np] = acc[i][j][r];
      }
    }
  }
}

// dW[m][ci][tap] (+)= sum_z slab[z][m][tap*C + ci]   (fixed summation order; coalesced on the dW side)
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* __restrict__ slabs, float* __restrict__ dw,
                                                           int M, int C, int KK, int splits, int accumulate) {
  // one workgroup per (m, 64-channel chunk): slab rows [tap][64 ci] are read as 256-B runs (summed over the split-K
  // slabs in fixed order), transposed through LDS and written/accumulated as one contiguous [64 ci][KK] run
  __shared__ float sm[64 * 49];
  const int m = blockIdx.y, c0 = blockIdx.x * 64;
  const size_t n = (size_t)M * C * KK;
  const int nel = 64 * KK;
  for (int i = threadIdx.x; i < nel; i += 256) {
    const int tap = i >> 6, ci = i & 63;
    const size_t src = (size_t)m * C * KK + (size_t)tap * C + c0 + ci;
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += slabs[(size_t)z * n + src];
    sm[ci * KK + tap] = v;
  }
  __syncthreads();
  float* __restrict__ dst = dw + ((size_t)m * C + c0) * KK;
  for (int i = threadIdx.x; i < nel; i += 256) dst[i] = accumulate ? dst[i] + sm[i] : sm[i];
}

static bool wgrad_fast_ok(int M, int C, int OH, int OW) {
  static int force_generic = -1;
  if (force_generic < 0) force_generic = getenv("HIM_GENERIC_CONV") ? 1 : 0;
  return !force_generic && M > 4 && (C % 64) == 0 && OH * OW >= 4;   // planes with OH*OW % 4 != 0: scalar dY loads
}
static void wgrad_fast_cfg(int M, int C, int Kdim, int KK, int* BM, int* BN, int* splits) {
  *BM = M > 64 ? 128 : 64;
  *BN = (C % 128) == 0 ? 128 : 64;
  const long long tiles = (long long)cdiv(M, *BM) * ((long long)C * KK / *BN);
  int s = (int)((768 + tiles - 1) / tiles);
  const int maxs = cdiv(Kdim, 32 * 8);
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  if (getenv("HIM_WGRAD_SPLITS")) s = std::max(1, std::min(maxs, atoi(getenv("HIM_WGRAD_SPLITS"))));
  *splits = s;
}

// ---- tiny-M weight gradient (M = Cout <= 4: the G tanh head).  Direct VALU reduction: one workgroup per
// (input channel, slice of the B*OH*OW positions); every thread keeps the MM x TJ x TJ partial filter in
// registers, then a wave64-shuffle + LDS reduction writes one slab per slice (summed in fixed order later).
template <int MM, int TJ, bool REFLECT>
__global__ __launch_bounds__(256) void wgrad_small_kernel(const WGradP p) {
  __shared__ float red[4][MM * TJ * TJ];
  const int c = blockIdx.x;
  const int t = threadIdx.x;
  const int OHW = p.OH * p.OW, HW = p.H * p.W;
  const int kbeg = blockIdx.y * p.kchunk;
  const int kend = min(p.Kdim, kbeg + p.kchunk);
  float acc[MM][TJ][TJ];
#pragma unroll
  for (int m = 0; m < MM; ++m)
#pragma unroll
    for (int a = 0; a < TJ; ++a)
#pragma unroll
      for (int b2 = 0; b2 < TJ; ++b2) acc[m][a][b2] = 0.f;
  const int H = p.H, W = p.W;
  for (int k = kbeg + t; k < kend; k += 256) {
    const int b = k / OHW;
    const int sp = k - b * OHW;
    const int oh = (int)fdiv((uint32_t)sp, p.fOW);
    const int ow = sp - oh * p.OW;
    float g[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) g[m] = p.dy[((size_t)b * p.M + m) * OHW + sp];
    const float* __restrict__ xb = p.x + ((size_t)b * p.C + c) * HW;
    int ixs[TJ];
    bool okx[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      int ix = ow * p.stride - p.pad + j;
      bool ok = true;
      if (REFLECT) {
        ix = ix < 0 ? -ix : ix;
        ix = ix >= W ? 2 * (W - 1) - ix : ix;
      } else {
        const int cx = min(max(ix, 0), W - 1);
        ok = cx == ix;
        ix = cx;
      }
      ixs[j] = ix;
      okx[j] = ok;
    }
#pragma unroll
    for (int a = 0; a < TJ; ++a) {
      int iy = oh * p.stride - p.pad + a;
      bool oky = true;
      if (REFLECT) {
        iy = iy < 0 ? -iy : iy;
        iy = iy >= H ? 2 * (H - 1) - iy : iy;
      } else {
        const int cy = min(max(iy, 0), H - 1);
        oky = cy == iy;
        iy = cy;
      }
      const float* __restrict__ row = xb + (size_t)iy * W;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        float v = row[ixs[j]];
        v = (oky && okx[j]) ? v : 0.f;
#pragma unroll
        for (int m = 0; m < MM; ++m) acc[m][a][j] = fmaf(g[m], v, acc[m][a][j]);
      }
    }
  }
  const int wave = t >> 6, lane = t & 63;
#pragma unroll
  for (int m = 0; m < MM; ++m)
#pragma unroll
    for (int a = 0; a < TJ; ++a)
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const float v = wave_sum(acc[m][a][j]);
        if (lane == 0) red[wave][(m * TJ + a) * TJ + j] = v;
      }
  __syncthreads();
  float* __restrict__ out = p.out + (size_t)blockIdx.y * p.M * p.Np;
  for (int i = t; i < MM * TJ * TJ; i += 256) {
    const int m = i / (TJ * TJ), r = i - m * TJ * TJ;
    out[(size_t)m * p.Np + c * TJ * TJ + r] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
  }
}

__global__ void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, long long n,
                                   int splits, int accumulate) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += slabs[(size_t)z * n + i];
    out[i] = accumulate ? out[i] + s : s;
  }
}

// dbias[c] (+)= sum_{b,sp} dy[b][c][sp]: stage 1 = (channel, slice) partial sums, stage 2 = fixed-order finish.
constexpr int BIAS_SLICES = 32;  // maximum; small planes use fewer (bias_slices())
static int bias_slices(int B, int hw) {
  long long per = ((long long)B * hw + 8191) / 8192;  // ~8k elements per workgroup
  if (per > BIAS_SLICES) per = BIAS_SLICES;
  if (per > hw) per = hw;
  return per < 1 ? 1 : (int)per;
}
// Bias gradients are sums of ~1e6 signed values that largely cancel (the head's: |sum| ~ 1e-2 of sum |dy|): fp32 partial
// sums left 1.6e-4 relative error against 2e-5 of torch's pairwise CPU sum (fp64-anchored step test).  The pass is
// HBM-bound and MI355X adds doubles at full vector rate, so every partial sum is carried in double.
__global__ __launch_bounds__(256) void bias_grad1_kernel(const float* __restrict__ dy, float* __restrict__ part,
                                                         int B, int C, int hw) {
  __shared__ double shd[4];
  const int c = blockIdx.x, sl = blockIdx.y;
  const int chunk = (hw + gridDim.y - 1) / gridDim.y;
  const int beg = sl * chunk, end = min(hw, beg + chunk);
  double s = 0.0;
  for (int b = 0; b < B; ++b) {
    const float* pl = dy + ((size_t)b * C + c) * hw;
    for (int i = beg + threadIdx.x; i < end; i += 256) s += (double)pl[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = s;
  __syncthreads();
  // the slice sums travel as (hi, lo) float pairs: 48 significant bits through the fp32 workspace
  if (threadIdx.x == 0) {
    const double t = (shd[0] + shd[1]) + (shd[2] + shd[3]);
    const float hi = (float)t;
    part[(c * BIAS_SLICES + sl) * 2] = hi;
    part[(c * BIAS_SLICES + sl) * 2 + 1] = (float)(t - (double)hi);
  }
}
__global__ void bias_grad2_kernel(const float* __restrict__ part, float* __restrict__ db, int C, int nsl,
                                  int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int i = 0; i < nsl; ++i) s += (double)part[(c * BIAS_SLICES + i) * 2] + (double)part[(c * BIAS_SLICES + i) * 2 + 1];
  db[c] = accumulate ? db[c] + (float)s : (float)s;
}
static size_t bias_ws_bytes(int C) { return (size_t)C * BIAS_SLICES * 2 * sizeof(float) + 256; }

static int wgrad_splits(int M, int Np, int Kdim, int BM, int BN) {
  const long long tiles = (long long)cdiv(M, BM) * cdiv(Np, BN);
  int splits = (int)((1024 + tiles - 1) / tiles);
  const int maxs = cdiv(Kdim, 32 * 8);  // at least 8 K-steps per split
  if (splits > maxs) splits = maxs;
  if (splits < 1) splits = 1;
  if (splits > 512) splits = 512;
  return splits;
}
static void wgrad_tile(int M, int* BM, int* BN) {
  *BN = 128;
  *BM = M <= 32 ? 32 : (M <= 64 ? 64 : 128);
}


// sum over the 64 lanes of a wave with DPP row shifts + row broadcasts (full-rate VALU, no LDS crossbar); the total
// ends up in lane 63
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));  // row_shr:1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));  // row_shr:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));  // row_shr:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));  // row_shr:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, true));  // row_bcast:15
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, true));  // row_bcast:31
  return v;
}

// Tiny-M weight gradient for "same" odd kernels (the G tanh head conv7x7 64->3): sliding-window, lane-private.
// One wave = one input channel x one BAND of TR filter rows; its 64 lanes are 64 consecutive output columns and walk
// down a strip of rows keeping the TR x KS window of x around their pixel in registers (ONE new row of KS loads per step
// instead of KS*KS gathers), MM*TR*KS private FMA accumulators, a DPP wave reduction once per workgroup.
// The band split (blockIdx.z: filter rows [th0, th0+TR), th0 = min(z*TR, KS-TR); rows a later band recomputes are
// written by the later band only) keeps the 7x7 / 3-output case at ~160 VGPRs = 3 waves per SIMD -- the un-split
// kernel needed 278 registers, i.e. ONE wave per SIMD with nothing to hide its load latency behind.
// part[slot][m][c*KK + t].  grid (slots, ceil(C/4), ceil(KS/TR)); workgroups are persistent over the (image, column
// strip, row chunk) tasks.
template <int MM, int KS, int TR, bool REFLECT>
__global__ __launch_bounds__(256) void wgrad_small_win_kernel(const WGradP p, int nsx, int nyc, int rows_per, int ntasks) {
  constexpr int KK = KS * KS;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = blockIdx.y * 4 + wave;
  if (c >= p.C) return;
  const int th0 = min((int)blockIdx.z * TR, KS - TR);      // first filter row of this band
  const int own0 = (int)blockIdx.z * TR - th0;             // band rows [own0, TR) are written by this band
  const int H = p.H, W = p.W, HW = H * W, pad = p.pad;
  const int padr = pad - th0;                              // row padding as seen by the band
  float acc[MM][TR * KS];
#pragma unroll
  for (int m = 0; m < MM; ++m)
#pragma unroll
    for (int t = 0; t < TR * KS; ++t) acc[m][t] = 0.f;
  for (int task = blockIdx.x; task < ntasks; task += gridDim.x) {
    int q = task;
    const int yc = q % nyc;
    q /= nyc;
    const int sx = q % nsx, b = q / nsx;
    const int px = sx * 64 + lane;
    const bool lane_on = px < W;
    const int y0 = yc * rows_per, y1 = min(y0 + rows_per, H);
    const float* __restrict__ xc = p.x + ((size_t)b * p.C + c) * HW;
    const float* __restrict__ g = p.dy + (size_t)b * p.M * HW + min(px, W - 1);
    int cx[KS];
    bool okc[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      int ix = px - pad + k;
      bool ok = lane_on;
      if (REFLECT) {
        ix = ix < 0 ? -ix : ix;
        ix = ix >= W ? 2 * (W - 1) - ix : ix;
      } else {
        ok = ok && ix >= 0 && ix < W;
      }
      cx[k] = min(max(ix, 0), W - 1);
      okc[k] = ok;
    }
    float win[TR][KS];
#define HIM_SW_ROW(PY, DST)                                                            \
  {                                                                                    \
    int iy = (PY);                                                                     \
    bool oky = true;                                                                   \
    if (REFLECT) {                                                                     \
      iy = iy < 0 ? -iy : iy;                                                          \
      iy = iy >= H ? 2 * (H - 1) - iy : iy;                                            \
    } else {                                                                           \
      oky = iy >= 0 && iy < H;                                                         \
    }                                                                                  \
    const float* __restrict__ rp = xc + min(max(iy, 0), H - 1) * W;                    \
    _Pragma("unroll") for (int k = 0; k < KS; ++k) {                                   \
      const float v = rp[cx[k]];                                                       \
      DST[k] = (oky && okc[k]) ? v : 0.f;                                              \
    }                                                                                  \
  }
#pragma unroll
    for (int r = 0; r < TR; ++r) HIM_SW_ROW(y0 - padr + r, win[r])
    float gv[MM], gn[MM];   // dy of this row / of the next one (loaded a step ahead)
#pragma unroll
    for (int m = 0; m < MM; ++m) {
      const float v = g[(size_t)m * HW + y0 * W];
      gn[m] = lane_on ? v : 0.f;
    }
    float nxt[KS], nx2[KS];   // x rows entering the window one / two steps from now
    HIM_SW_ROW(y0 - padr + TR, nxt)
    for (int pyb = y0; pyb < y1; pyb += TR) {
#pragma unroll
      for (int ph = 0; ph < TR; ++ph) {
        const int py = pyb + ph;
        if (py < y1) {
          HIM_SW_ROW(py - padr + TR + 1, nx2)
#pragma unroll
          for (int m = 0; m < MM; ++m) {
            gv[m] = gn[m];
            const float v = g[(size_t)m * HW + min(py + 1, H - 1) * W];
            gn[m] = lane_on ? v : 0.f;
          }
          // x row (py - padr + th) sits in logical slot th = physical (th + ph) % TR
#pragma unroll
          for (int th = 0; th < TR; ++th)
#pragma unroll
            for (int tw = 0; tw < KS; ++tw)
#pragma unroll
              for (int m = 0; m < MM; ++m) acc[m][th * KS + tw] = fmaf(gv[m], win[(th + ph) % TR][tw], acc[m][th * KS + tw]);
#pragma unroll
          for (int k = 0; k < KS; ++k) {
            win[ph][k] = nxt[k];
            nxt[k] = nx2[k];
          }
        }
      }
    }
#undef HIM_SW_ROW
  }
  float* __restrict__ out = p.out + (size_t)blockIdx.x * p.M * p.Np + (size_t)c * KK;
#pragma unroll
  for (int m = 0; m < MM; ++m)
#pragma unroll
    for (int th = 0; th < TR; ++th)
#pragma unroll
      for (int tw = 0; tw < KS; ++tw) {
        const float sm = wave_sum_dpp(acc[m][th * KS + tw]);
        if (lane == 63 && th >= own0) out[(size_t)m * p.Np + (th0 + th) * KS + tw] = sm;
      }
}

#include "him_wgrad_fewch.inc"

static const int SMALL_WIN_SLOTS = 256;
static bool small_win_ok(int M, int KH, int KW, int stride, int pad, int H, int W, int OH, int OW) {
  static int off = -1;
  if (off < 0) off = getenv("HIM_NO_SMALL_WIN") ? 1 : 0;
  return !off && M <= 4 && KH == KW && (KH == 3 || KH == 5 || KH == 7) && stride == 1 && pad == KH / 2 && OH == H && OW == W &&
         H > pad && W > pad;
}

static int small_wgrad_slices(int C, int Kdim) {
  int s = (2048 + C - 1) / C;
  const int maxs = cdiv(Kdim, 256 * 8);
  if (s > maxs) s = maxs;
  return s < 1 ? 1 : s;
}
static bool small_wgrad_ok(int M, int KH, int KW) { return M <= 4 && KH == KW && (KH == 7 || KH == 4 || KH == 3); }
static size_t wgrad_slab_bytes(int M, int C, int KH, int KW, int Kdim, size_t wino_floats = 0) {
  const int Np = C * KH * KW;
  size_t slabs;
  if (wino_floats) return ((wino_floats * sizeof(float) + 255) / 256) * 256;
  const bool fewch_shape = !fewch_off() && KH == KW && (KH == 5 || KH == 7) &&
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
  } else if (wgrad_fast_ok(M, C, 1, 4)) {  /* upper bound; the runner re-checks OH*OW */
    int BM, BN, sp;
    wgrad_fast_cfg(M, C, Kdim, KH * KW, &BM, &BN, &sp);
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
static size_t wgrad_ws_bytes(int M, int C, int KH, int KW, int Kdim, int biasC, size_t wino_floats = 0) {
  return wgrad_slab_bytes(M, C, KH, KW, Kdim, wino_floats) + bias_ws_bytes(biasC);
}
static size_t conv_wino_wgrad_floats(const HimConv2d* d) {
  return wino_wgrad_ok(d->Cout, d->Cin, d->KH, d->KW, d->stride, d->pad, d->H, d->W)
             ? wino_wgrad_floats(d->B, d->Cout, d->Cin, d->OH, d->OW)
             : 0;
}

// generic weight gradient: dW[M][C*KH*KW] from dy[B][M][OH][OW] and x[B][C][H][W]
static int run_wgrad(const float* dy, const float* x, float* dw, int M, int C, int B, int H, int W, int OH,
                     int OW, int KH, int KW, int stride, int pad, int pad_mode, int accumulate, void* ws,
                     size_t ws_bytes, hipStream_t st) {
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
  if (wino_wgrad_ok(M, C, KH, KW, stride, pad, H, W) && OH == H && OW == W) {
    // dU = dM x V^T per Winograd position (batched NT GEMM), then dw (+)= G^T dU G
    const size_t need = wino_wgrad_floats(B, M, C, OH, OW) * sizeof(float);
    if (ws_bytes < need || !ws) return fail(HIM_E_WORKSPACE, "wgrad needs %zu ws bytes, got %zu", need, ws_bytes);
    const WinoGeom gx = wino_geom(B, C, H, W, OH, OW, 1);
    WinoGeom gd = gx;
    gd.C = M;
    float* Vt = (float*)ws;                          // [16][Tp][C]
    float* dM = Vt + (size_t)16 * C * gx.Tp;         // [16][M][Tp]
    float* dU = dM + (size_t)16 * M * gx.Tp;         // [16][M][C]
    const dim3 gin(cdiv(C, 256), gx.Tp);
    if (pad_mode == HIM_PAD_REFLECT) hipLaunchKernelGGL((wino_input_t_kernel<true>), gin, dim3(256), 0, st, x, Vt, gx);
    else hipLaunchKernelGGL((wino_input_t_kernel<false>), gin, dim3(256), 0, st, x, Vt, gx);
    hipLaunchKernelGGL(wino_dy_kernel, dim3(cdiv(gx.Tp, 256), M), dim3(256), 0, st, dy, dM, gd);
    int rcw = check_launch("wino_wgrad_transforms");
    if (rcw) return rcw;
    rcw = wino_batched_gemm(dM, Vt, dU, M, gx.Tp, C, st);
    if (rcw) return rcw;
    hipLaunchKernelGGL(wino_wgrad_out_kernel, dim3(cdiv(C, 256), M), dim3(256), 0, st, (const float*)dU, dw, M, C,
                       accumulate);
    return check_launch("wino_wgrad_out");
  }
  if (fewch_head_ok(M, C, KH, KW, stride, pad, H, W, OH, OW))
    return run_wgrad_fewch(true, dy, x, dw, M, C, B, H, W, KH, pad, pad_mode, accumulate, ws, ws_bytes, st);
  if (fewch_stem_ok(M, C, KH, KW, stride, pad, H, W, OH, OW))
    return run_wgrad_fewch(false, dy, x, dw, M, C, B, H, W, KH, pad, pad_mode, accumulate, ws, ws_bytes, st);
  if (small_win_ok(M, KH, KW, stride, pad, H, W, OH, OW)) {
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
  if (wgrad_fast_ok(M, C, OH, OW) && fits31) {
    int fBM, fBN, fs;
    wgrad_fast_cfg(M, C, p.Kdim, KH * KW, &fBM, &fBN, &fs);
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
static int fast_ksplit(int M, long long N, int nk) {
  static int off = -1;
  if (off < 0) off = getenv("HIM_NO_SPLITK") ? 1 : 0;
  if (off) return 1;
  const long long tiles = (M <= 64 ? (long long)cdiv(N, 128) * cdiv(M, 64) : (long long)cdiv(N, 128) * cdiv(M, 128));
  if (tiles >= 1024) return 1;
  // makespan model in units of one full-K tile on one of 256 CUs: rounds/ks, a 7 % bonus once every CU hosts >= 2
  // independent workgroups (they cover each other's LDS/barrier bubbles), 2 % for the finish pass
  int best = 1;
  double best_cost = 1e30;
  const int cand[] = {1, 2, 3, 4, 6, 8, 12, 16};
  for (int ks : cand) {
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
  return wino_shape_ok(d->Cout, d->Cin, d->KH, d->KW, d->stride, d->pad, d->H, d->W);
}
// The fused Winograd kernel (him_wino_fused.inc: transforms inside the GEMM kernel) takes the 3x3 stride-1 pad-1 layers
// with 64..512 reduction channels and a multiple of 64 output channels in the FORWARD direction (zero or reflection
// padding) and the data gradient of ZERO-padded layers: all VGG convs but conv1_1, the box2mask ResnetBlocks.
// Measured against the alternatives (tools/conv_bench.py / tools/micro/wino_micro, TFLOP/s direct-form equivalent):
// 64->64 150 vs 116 (direct MFMA kernel), 128->128 189 vs 131, 256->256 203 vs 131, 512->512 220 vs 194
// (separate-transform pipeline, which needs two more launches and 4x the activation in HBM).  The separate-transform
// pipeline keeps the 1024-channel ResnetBlock stack (see wino_fused_max_c), the data gradient of reflection-padded
// layers (border fold) and the weight gradient.
static int wino_fused_min_c() {
  static int v = -2;
  if (v == -2) v = getenv("HIM_NO_WINO_FUSED") ? 0 : (getenv("HIM_WINO_FUSED_MIN_C") ? atoi(getenv("HIM_WINO_FUSED_MIN_C")) : 64);
  return v;
}
// Upper end of the fused kernel's channel range.  Inside the training step (weight panels streamed from HBM, 67 MB per
// 1024-channel layer) the ResnetBlock forward is faster on the separate-transform pipeline: 10.4 vs 11.8 ms generator
// forward, 121.4 vs 118.9 images/s (A/B with HIM_WINO_FUSED_MAX_C) -- although the isolated kernel, whose panel stays in
// the Infinity Cache between launches, measures 197 vs 188 TFLOP/s equivalent.
static int wino_fused_max_c() {
  static int v = -2;
  if (v == -2) v = getenv("HIM_WINO_FUSED_MAX_C") ? atoi(getenv("HIM_WINO_FUSED_MAX_C")) : 512;
  return v;
}
static bool wino_fused_ok(int Co, int Ci, int KH, int KW, int stride, int pad, int B, int H, int W) {
  const int mc = wino_fused_min_c();
  return mc > 0 && Ci >= mc && Ci <= wino_fused_max_c() && Co >= 64 &&
         wino_fused_shape_ok(Co, Ci, KH, KW, stride, pad, B, H, W);
}
static bool wino_fused_fwd_ok(const HimConv2d* d) {
  return wino_fused_ok(d->Cout, d->Cin, d->KH, d->KW, d->stride, d->pad, d->B, d->H, d->W);
}
// data gradient of a ZERO-padded 3x3 stride-1 conv = the same convolution with the flipped / transposed filter
static bool wino_fused_dgrad_ok(const HimConv2d* d) {
  return d->pad_mode == HIM_PAD_ZERO && d->OH == d->H && d->OW == d->W &&
         wino_fused_ok(d->Cin, d->Cout, d->KH, d->KW, d->stride, d->pad, d->B, d->H, d->W);
}
static size_t fprop_ws_bytes(const HimConv2d* d) {
  if (wino_fused_fwd_ok(d)) return wino_fused_panel_floats(d->Cout, d->Cin) * sizeof(float) + 256;
  if (wino_fwd_ok(d))
    return ((size_t)16 * d->Cout * d->Cin + wino_conv_floats(d->B, d->Cin, d->Cout, d->OH, d->OW)) * sizeof(float) + 256;
  if (small_split_ok(d)) return (size_t)SMALL_NSPLIT * d->Cout * d->B * d->OH * d->OW * sizeof(float) + 256;
  if (!use_fast(d->Cout, d->Cin)) return 0;
  const int ks = fast_ksplit(d->Cout, (long long)d->B * d->OH * d->OW, d->KH * d->KW * (pad16(d->Cin) / 16));
  const size_t wts = ((size_t)d->Cout * d->KH * d->KW * pad16(d->Cin) * sizeof(float) + 255) / 256 * 256;
  return wts + (ks > 1 ? (size_t)ks * d->B * d->Cout * d->OH * d->OW * sizeof(float) : 0) + 256;
}
// floats of the regrouped weight panel the forward kernel reads (0: it reads the raw weights)
static size_t fprop_panel_floats(const HimConv2d* d) {
  if (small_split_ok(d) || d->Cout <= 4 || !use_fast(d->Cout, d->Cin)) return 0;
  if (wino_fused_fwd_ok(d)) return wino_fused_panel_floats(d->Cout, d->Cin);
  if (wino_fwd_ok(d)) return (size_t)16 * d->Cout * d->Cin;
  return (size_t)d->Cout * d->KH * d->KW * pad16(d->Cin);
}
// panel == nullptr: regroup the weights into the workspace on every call; build_only: write the panel to ws and return
static int run_fprop(const HimConv2d* d, const float* x, const float* w, const float* bias, float* y, void* ws,
                     size_t ws_bytes, hipStream_t st, const float* panel = nullptr, bool build_only = false) {
  if (wino_fused_fwd_ok(d)) {
    if (!panel) {
      const size_t need = wino_fused_panel_floats(d->Cout, d->Cin) * sizeof(float);
      if (!ws || ws_bytes < need) return fail(HIM_E_WORKSPACE, "conv fwd needs %zu ws bytes, got %zu", need, ws_bytes);
      hipLaunchKernelGGL((wino_fused_weight_kernel<0>), dim3(cdiv(d->Cin, 256), d->Cout), dim3(256), 0, st, w, (float*)ws,
                         d->Cout, d->Cin);
      int rc = check_launch("wino_fused_weight");
      if (rc || build_only) return rc;
    }
    return run_wino_fused(d->B, d->Cin, d->H, d->W, d->Cout, d->pad_mode == HIM_PAD_REFLECT, x,
                          panel ? panel : (const float*)ws, bias, d->act, d->slope, y, st);
  }
  if (wino_fwd_ok(d)) {
    const size_t need = build_only ? fprop_panel_floats(d) * sizeof(float) : fprop_ws_bytes(d);
    if (!ws || ws_bytes < need) return fail(HIM_E_WORKSPACE, "conv fwd needs %zu ws bytes, got %zu", need, ws_bytes);
    float* U = (float*)ws;
    if (!panel) {
      hipLaunchKernelGGL((wino_weight_kernel<0>), dim3(cdiv(d->Cin, 256), d->Cout), dim3(256), 0, st, w, U, d->Cout,
                         d->Cin);
      int rc = check_launch("wino_weight");
      if (rc || build_only) return rc;
    }
    return run_wino_conv(d->B, d->Cin, d->H, d->W, d->Cout, d->OH, d->OW, 1, d->pad_mode == HIM_PAD_REFLECT, x,
                         panel ? panel : U, bias, d->act, d->slope, y, U + (size_t)16 * d->Cout * d->Cin, st);
  }
  GConvP g;
  fill_fprop(g, d, x, w, bias, y);
  if (small_split_ok(d)) {
    if (!ws || ws_bytes < fprop_ws_bytes(d)) return fail(HIM_E_WORKSPACE, "conv fwd ws too small");
    g.small_part = (float*)ws;
    g.small_nsplit = SMALL_NSPLIT;
  }
  if (use_fast(d->Cout, d->Cin)) {
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
    const int ks = fast_ksplit(d->Cout, (long long)d->B * d->OH * d->OW, KK * (t.C2p / 16));
    if (ks > 1) {
      const size_t wts = ((size_t)d->Cout * KK * t.C2p * sizeof(float) + 255) / 256 * 256;
      g.ksplit = ks;
      g.kpart = (float*)((char*)ws + wts);
    }
  }
  return launch_gconv(g, st);
}

// data gradient of the conv described by `d` (also the forward of its transposed conv):
// out (B,Cin,H,W) = sum W * g (B,Cout,OH,OW); for reflect mode goes through the padded gradient + fold.
// reflect-pad-1 3x3 stride-1 (every ResnetBlock conv): gather from the border-extended gradient, no padded GEMM columns
static bool dfold_ok(const HimConv2d* d) {
  static int off = -1;
  if (off < 0) off = getenv("HIM_NO_DFOLD") ? 1 : 0;
  return !off && d->pad_mode == HIM_PAD_REFLECT && d->pad == 1 && d->KH == 3 && d->KW == 3 && d->stride == 1 &&
         d->H >= 3 && d->W >= 3 && d->OH == d->H && d->OW == d->W && use_fast(d->Cin, d->Cout);
}
static bool wino_dgrad_ok(const HimConv2d* d) {
  return wino_shape_ok(d->Cin, d->Cout, d->KH, d->KW, d->stride, d->pad, d->H, d->W) && d->OH == d->H && d->OW == d->W;
}
static bool wino_dgrad_fold(const HimConv2d* d) {  // reflect folded into the border tiles' patches (see wino_input_kernel)
  return d->pad_mode == HIM_PAD_REFLECT && (d->H % 2) == 0 && (d->W % 2) == 0 && d->H >= 4 && d->W >= 4 &&
         !getenv("HIM_WINO_PADDED_DGRAD");
}
static size_t wino_dgrad_floats(const HimConv2d* d) {  // U' + V + Mo (+ padded gradient for reflect)
  const bool refl = d->pad_mode == HIM_PAD_REFLECT && !wino_dgrad_fold(d);
  const int GH = refl ? d->H + 2 : d->H, GW = refl ? d->W + 2 : d->W;
  return (size_t)16 * d->Cin * d->Cout + wino_conv_floats(d->B, d->Cout, d->Cin, GH, GW) +
         (refl ? (size_t)d->B * d->Cin * GH * GW : 0);
}
static int dgrad_ksplit(const HimConv2d* d) {
  if (d->stride != 1 || !use_fast(d->Cin, d->Cout)) return 1;
  const bool refl = d->pad_mode == HIM_PAD_REFLECT && !dfold_ok(d);
  const long long N = (long long)d->B * (refl ? d->H + 2 * d->pad : d->H) * (refl ? d->W + 2 * d->pad : d->W);
  return fast_ksplit(d->Cin, N, d->KH * d->KW * (pad16(d->Cout) / 16));
}
static size_t dgrad_ws_bytes(const HimConv2d* d) {
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
  if (wino_fused_dgrad_ok(d)) return wino_fused_panel_floats(d->Cin, d->Cout);
  if (wino_dgrad_ok(d)) return (size_t)16 * d->Cin * d->Cout;
  return (size_t)d->Cin * (use_fast(d->Cin, d->Cout) ? pad16(d->Cout) : d->Cout) * d->KH * d->KW;
}
// panel == nullptr: regroup the weights into the workspace on every call; build_only: write the panel to ws and return
static int run_dgrad(const HimConv2d* d, const float* gy, const float* w, float* out, const float* bias, int act,
                     float slope, void* ws, size_t ws_bytes, hipStream_t st, const float* panel = nullptr,
                     bool build_only = false, const float* relu_mask = nullptr, bool* mask_done = nullptr) {
  const size_t need = build_only ? dgrad_panel_floats(d) * sizeof(float) : dgrad_ws_bytes(d);
  if (!ws || ws_bytes < need) return fail(HIM_E_WORKSPACE, "dgrad needs %zu ws bytes, got %zu", need, ws_bytes);
  if (wino_fused_dgrad_ok(d)) {   // zero-padded 3x3 stride-1: the convolution of gy with the flipped / transposed filter
    if (!panel) {
      hipLaunchKernelGGL((wino_fused_weight_kernel<1>), dim3(cdiv(d->Cout, 256), d->Cin), dim3(256), 0, st, w, (float*)ws,
                         d->Cin, d->Cout);
      int rcu = check_launch("wino_fused_weight");
      if (rcu || build_only) return rcu;
    }
    if (mask_done) *mask_done = relu_mask != nullptr;   // the gate rides in this kernel's epilogue
    return run_wino_fused(d->B, d->Cout, d->OH, d->OW, d->Cin, false, gy, panel ? panel : (const float*)ws, bias, act, slope,
                          out, st, relu_mask);
  }
  if (wino_dgrad_ok(d) && panel && (bias || act != HIM_ACT_NONE))
    return fail(HIM_E_UNSUPPORTED, "dgrad: Winograd panel with a fused bias/activation epilogue");
  if (wino_dgrad_ok(d) && (build_only || (!bias && act == HIM_ACT_NONE))) {
    float* U = (float*)ws;
    if (!panel) {
      hipLaunchKernelGGL((wino_weight_kernel<1>), dim3(cdiv(d->Cout, 256), d->Cin), dim3(256), 0, st, w, U, d->Cin,
                         d->Cout);
      int rcu = check_launch("wino_weight");
      if (rcu || build_only) return rcu;
    }
    const bool fold = wino_dgrad_fold(d);
    const bool rf = d->pad_mode == HIM_PAD_REFLECT && !fold;
    const int GH = rf ? d->H + 2 : d->H, GW = rf ? d->W + 2 : d->W;
    float* wsv = U + (size_t)16 * d->Cin * d->Cout;
    float* dpadw = wsv + wino_conv_floats(d->B, d->Cout, d->Cin, GH, GW);
    // reflect: full correlation (offset 2) -> padded gradient -> fold; zero pad: the plain pad-1 correlation
    int rcw = run_wino_conv(d->B, d->Cout, d->OH, d->OW, d->Cin, GH, GW, rf ? 2 : 1, false, gy, panel ? panel : U,
                            nullptr, HIM_ACT_NONE, 0.f, rf ? dpadw : out, wsv, st, fold);
    if (rcw || !rf) return rcw;
    hipLaunchKernelGGL(reflect_fold_kernel, dim3(cdiv((long long)d->H * d->W, 256), d->B * d->Cin), dim3(256), 0, st,
                       (const float*)dpadw, out, d->B * d->Cin, d->H, d->W, 1, 1, (size_t)0);
    return check_launch("reflect_fold");
  }
  float* Wt = panel ? (float*)panel : (float*)ws;
  const bool dfold = dfold_ok(d);
  const bool refl = d->pad_mode == HIM_PAD_REFLECT && !dfold;
  const bool fast = use_fast(d->Cin, d->Cout);
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
  rc = launch_gconv(g, st);
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


// ==============================================================================================
// One-hot stems.  The first conv of the generators (conv7x7 38->64 / 49->64 at full resolution: 12.7 % of the
// generator's direct-form FLOPs, twice that with its weight gradient) reads [one-hot(label) | dense channels]
// (reference encode_input, models/pix2pixHD_condImg_model.py:144-174 + the cat at :204).  For the one-hot channels the
// convolution is a table lookup -- exactly ONE of the label_nc products per (pixel, tap) is non-zero and it is 1.0*w:
//   y[co][p]      = bias[co] + sum_t W[co][label(p+t)][t]                   (+ the dense channels' ordinary conv)
//   dW[co][c][t] += sum_{p : label(p+t) = c} dy[co][p]                      (a class-segmented sum of dy)
// Same sums as the dense form up to fp32 summation order; the label ids replace 35/38 of the MFMA work by LDS
// lookups / wave reductions.  Labels are the (B,1,H,W) float id maps of the data set; ids outside [0, NC) select
// nothing (the one-hot column is all zero, as in him_onehot).
// ==============================================================================================
struct OneHotP {
  const float* label;  // [B][H][W] ids as floats
  int B, H, W, NC, KS, pad, reflect, Cout;
  int npix;            // B*H*W
};

// Wt[(t*NC + c)*Cout + co] = w[(co*C + c)*KK + t]
__global__ void onehot_table_kernel(const float* __restrict__ w, float* __restrict__ Wt, int Cout, int C, int NC, int KK) {
  const int total = KK * NC * Cout;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int co = i % Cout, r = i / Cout, c = r % NC, t = r / NC;
    Wt[i] = w[((size_t)co * C + c) * KK + t];
  }
}

__device__ __forceinline__ int onehot_label_at(const OneHotP& p, const float* __restrict__ lab, int y, int x) {
  // returns the class id at (y, x) of the padded image, or -1 (no contribution)
  bool ok = true;
  if (p.reflect) {
    y = y < 0 ? -y : y;
    y = y >= p.H ? 2 * (p.H - 1) - y : y;
    x = x < 0 ? -x : x;
    x = x >= p.W ? 2 * (p.W - 1) - x : x;
  } else {
    ok = y >= 0 && y < p.H && x >= 0 && x < p.W;
  }
  y = min(max(y, 0), p.H - 1);
  x = min(max(x, 0), p.W - 1);
  const int id = (int)lab[y * p.W + x];
  return (ok && id >= 0 && id < p.NC) ? id : -1;
}

// y[b][co0+j][p] (+)= sum_t Wt[t][label(p+t)][co0+j], j < 16; then the activation.  The 16-channel slice of the table
// lives in LDS ([t][c][16], up to 154 KB); workgroups are persistent over pixel tiles.  All KS*KS label loads of a pixel
// are issued before the first lookup (one resident workgroup per CU: nothing else would cover their latency).
// grid (nblk, Cout/16)
template <int KS>
__global__ __launch_bounds__(1024) void onehot_conv_fwd_kernel(const OneHotP p, const float* __restrict__ Wt,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              int add_to_y, int act, float slope) {
  extern __shared__ __attribute__((aligned(16))) float tab[];
  constexpr int KK = KS * KS;
  const int co0 = blockIdx.y * 16;
  for (int i = threadIdx.x; i < KK * p.NC * 4; i += 1024) {
    const int e = i >> 2, q = i & 3;
    *(float4*)&tab[e * 16 + q * 4] = *(const float4*)&Wt[(size_t)e * p.Cout + co0 + q * 4];
  }
  __syncthreads();
  const int HW = p.H * p.W;
  for (int base = blockIdx.x * 1024; base < p.npix; base += gridDim.x * 1024) {
    const int n = base + threadIdx.x;
    if (n >= p.npix) continue;
    const int b = n / HW, r = n - b * HW, yy = r / p.W, xx = r - yy * p.W;
    const float* __restrict__ lab = p.label + (size_t)b * HW;
    float* __restrict__ out = y + ((size_t)b * p.Cout + co0) * HW + r;
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = add_to_y ? out[(size_t)j * HW] : (bias ? bias[co0 + j] : 0.f);
    int cls[KS], nxt[KS];  // one tap row of labels in flight ahead of the lookups of the current row
#pragma unroll
    for (int tw = 0; tw < KS; ++tw) cls[tw] = onehot_label_at(p, lab, yy - p.pad, xx + tw - p.pad);
#pragma unroll 1
    for (int th = 0; th < KS; ++th) {
#pragma unroll
      for (int tw = 0; tw < KS; ++tw) nxt[tw] = onehot_label_at(p, lab, yy + min(th + 1, KS - 1) - p.pad, xx + tw - p.pad);
#pragma unroll
      for (int tw = 0; tw < KS; ++tw) {
      const int c = cls[tw];
      const int t = th * KS + tw;
      const float4* __restrict__ e = (const float4*)&tab[(t * p.NC + max(c, 0)) * 16];
      const float4 e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3];
      const float m = c >= 0 ? 1.f : 0.f;
      acc[0] = fmaf(m, e0.x, acc[0]); acc[1] = fmaf(m, e0.y, acc[1]); acc[2] = fmaf(m, e0.z, acc[2]); acc[3] = fmaf(m, e0.w, acc[3]);
      acc[4] = fmaf(m, e1.x, acc[4]); acc[5] = fmaf(m, e1.y, acc[5]); acc[6] = fmaf(m, e1.z, acc[6]); acc[7] = fmaf(m, e1.w, acc[7]);
      acc[8] = fmaf(m, e2.x, acc[8]); acc[9] = fmaf(m, e2.y, acc[9]); acc[10] = fmaf(m, e2.z, acc[10]); acc[11] = fmaf(m, e2.w, acc[11]);
      acc[12] = fmaf(m, e3.x, acc[12]); acc[13] = fmaf(m, e3.y, acc[13]); acc[14] = fmaf(m, e3.z, acc[14]); acc[15] = fmaf(m, e3.w, acc[15]);
      }
#pragma unroll
      for (int tw = 0; tw < KS; ++tw) cls[tw] = nxt[tw];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) out[(size_t)j * HW] = apply_act(acc[j], act, slope);
  }
}

// part[blk][t][c][co0 + 4*wave + j] = sum over this block's padded positions q with class(q) = c of dy[co][q - t].
// Lane-private accumulation: the 64 lanes of a wave are 64 consecutive columns qx of the PADDED image and walk down a
// strip of rows; a lane keeps the KSxKS window of dy around its position in registers (one new row of KS loads per
// step, coalesced across the lanes) and adds it to KSxKS private accumulators per output channel -- all positions it
// visits while its class stays the same share those accumulators (label maps are piecewise constant).  On a class
// change the lane flushes them into its wave's PRIVATE LDS table with ds_add_f32; 8 waves x 2 output channels (the two
// channels ride in one v_pk_add_f32; 4 per wave needed 450 registers and spent its time moving AGPRs).
// grid (pixel-blocks = B * nsx * nyc, Cout/16); dynamic LDS = 16 * KK*NC floats.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// sum over each DPP row (16 lanes) -- the row total ends up in lanes 15, 31, 47, 63
__device__ __forceinline__ float row_sum_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));  // row_shr:1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));  // row_shr:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));  // row_shr:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));  // row_shr:8
  return v;
}

template <int KS>
__global__ __launch_bounds__(512) void onehot_wgrad_kernel(const OneHotP p, const float* __restrict__ dy,
                                                           float* __restrict__ part, int nsx, int nyc, int rows_per) {
  extern __shared__ __attribute__((aligned(16))) float tab[];
  constexpr int KK = KS * KS, CPW = 2;  // 2 channels per wave (one v_pk_add_f32), 8 waves = the workgroup's 16 channels
  const int pad = p.pad, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int co0 = blockIdx.y * 16 + wave * CPW;
  float* __restrict__ mytab = tab + (size_t)wave * KK * p.NC * CPW;
  for (int i = lane; i < KK * p.NC * CPW; i += 64) mytab[i] = 0.f;
  const int H = p.H, W = p.W, HW = H * W;
  int blk = blockIdx.x;
  const int yc = blk % nyc;
  blk /= nyc;
  const int sx = blk % nsx, b = blk / nsx;
  // strips start 16 columns left of the image so that the 16-lane DPP rows sit on multiples of 16 in image
  // coordinates (label regions of real maps and of the synthetic 16x16 blocks then rarely split a row)
  const int qx = sx * 64 - 16 + lane;
  const bool lane_on = qx >= -pad && qx < W + pad;
  const int qy0 = -pad + yc * rows_per, qy1 = min(qy0 + rows_per, H + pad);
  const float* __restrict__ g = dy + ((size_t)b * p.Cout + co0) * HW;
  const float* __restrict__ lab = p.label + (size_t)b * HW;
  // class of padded position (qy, qx): the one-hot image is what gets padded (reflect: mirrored ids; zero: nothing)
  int lx = qx;
  bool okx = lane_on;
  if (p.reflect) {
    lx = lx < 0 ? -lx : lx;
    lx = lx >= W ? 2 * (W - 1) - lx : lx;
  } else {
    okx = okx && qx >= 0 && qx < W;
  }
  lx = min(max(lx, 0), W - 1);
  int pxs[KS];
  bool pok[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    const int px = qx - pad + k;
    pok[k] = lane_on && px >= 0 && px < W;
    pxs[k] = min(max(px, 0), W - 1);
  }
  f32x2 win[KS][KS], acc[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) acc[t] = f32x2{0.f, 0.f};
  // window rows for the first position: logical slot r holds dy row (qy0 - pad + r)
#pragma unroll
  for (int r = 0; r < KS; ++r) {
    const int py = qy0 - pad + r;
    const bool oky = py >= 0 && py < H;
    const int row = min(max(py, 0), H - 1) * W;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const f32x2 v = f32x2{g[row + pxs[k]], g[(size_t)HW + row + pxs[k]]};
      win[r][k] = (oky && pok[k]) ? v : f32x2{0.f, 0.f};
    }
  }
#define HIM_OH_CLASS(QY, OUT)                                             \
  {                                                                       \
    int ly = (QY);                                                        \
    bool ok = okx;                                                        \
    if (p.reflect) {                                                      \
      ly = ly < 0 ? -ly : ly;                                             \
      ly = ly >= H ? 2 * (H - 1) - ly : ly;                               \
    } else {                                                              \
      ok = ok && (QY) >= 0 && (QY) < H;                                   \
    }                                                                     \
    ly = min(max(ly, 0), H - 1);                                          \
    const int id = (int)lab[ly * W + lx];                                 \
    OUT = (ok && id >= 0 && id < p.NC) ? id : -1;                         \
  }
  // Flush of the private sums.  Fast path (the usual one: label regions are wider than 16 pixels and every lane of the
  // wave changes class on the same row): all 16 lanes of each DPP row leave the same class -> one DPP row sum per
  // accumulator and a single ds_add_f32 lane per row; otherwise every flushing lane adds its own sums.
#define HIM_OH_FLUSH(FLUSHING)                                                                               \
  {                                                                                                          \
    int rc = cur; /* class of the row = max over its 16 lanes (lanes without a class carry -1 and zero sums) */ \
    rc = max(rc, __builtin_amdgcn_update_dpp(-1, rc, 0x111, 0xf, 0xf, false));                               \
    rc = max(rc, __builtin_amdgcn_update_dpp(-1, rc, 0x112, 0xf, 0xf, false));                               \
    rc = max(rc, __builtin_amdgcn_update_dpp(-1, rc, 0x114, 0xf, 0xf, false));                               \
    rc = max(rc, __builtin_amdgcn_update_dpp(-1, rc, 0x118, 0xf, 0xf, false));                               \
    rc = __shfl(rc, lane | 15);                                                                              \
    const bool rowu = __ballot((FLUSHING) && (cur < 0 || cur == rc)) == ~0ull;                               \
    if (rowu) {                                                                                              \
      _Pragma("unroll") for (int t = 0; t < KK; ++t) {                                                       \
        const float s0 = row_sum_dpp(cur < 0 ? 0.f : acc[t].x), s1 = row_sum_dpp(cur < 0 ? 0.f : acc[t].y);  \
        if ((lane & 15) == 15 && rc >= 0) {                                                                  \
          __hip_atomic_fetch_add(&mytab[(t * p.NC + rc) * CPW], s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);     \
          __hip_atomic_fetch_add(&mytab[(t * p.NC + rc) * CPW + 1], s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
        }                                                                                                    \
      }                                                                                                      \
    } else if ((FLUSHING) && cur >= 0) {                                                                     \
      _Pragma("unroll") for (int t = 0; t < KK; ++t) {                                                       \
        __hip_atomic_fetch_add(&mytab[(t * p.NC + cur) * CPW], acc[t].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);     \
        __hip_atomic_fetch_add(&mytab[(t * p.NC + cur) * CPW + 1], acc[t].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
      }                                                                                                      \
    }                                                                                                        \
  }
  int cur = -1, cnext;
  HIM_OH_CLASS(qy0, cnext)
  for (int qyb = qy0; qyb < qy1; qyb += KS) {
#pragma unroll
    for (int ph = 0; ph < KS; ++ph) {
      const int qy = qyb + ph;
      if (qy < qy1) {
        // issue the loads of the row that enters the window at the NEXT position and the next position's class
        const int pyn = qy + 1 + pad;
        const bool okn = pyn >= 0 && pyn < H;
        const int rown = min(max(pyn, 0), H - 1) * W;
        f32x2 nxt[KS];
#pragma unroll
        for (int k = 0; k < KS; ++k) nxt[k] = f32x2{g[rown + pxs[k]], g[(size_t)HW + rown + pxs[k]]};
        const int c = cnext;
        HIM_OH_CLASS(qy + 1, cnext)
        const bool chg = c != cur;
        if (__ballot(chg)) {  // wave-uniform: some lane leaves its class
          HIM_OH_FLUSH(chg)
          if (chg) {
#pragma unroll
            for (int t = 0; t < KK; ++t) acc[t] = f32x2{0.f, 0.f};
            cur = c;
          }
        }
        // acc[th][tw] += dy[q - t] = window row (KS-1-th), column (KS-1-tw); physical row slot = (logical + ph) % KS
#pragma unroll
        for (int th = 0; th < KS; ++th)
#pragma unroll
          for (int tw = 0; tw < KS; ++tw) acc[th * KS + tw] += win[(KS - 1 - th + ph) % KS][KS - 1 - tw];
        // the retired row slot (logical 0 = physical ph) receives the new row
#pragma unroll
        for (int k = 0; k < KS; ++k) win[ph][k] = (okn && pok[k]) ? nxt[k] : f32x2{0.f, 0.f};
      }
    }
  }
  HIM_OH_FLUSH(true)
#undef HIM_OH_CLASS
#undef HIM_OH_FLUSH
  __syncthreads();
  // wave-private table -> part[blk][t][c][Cout] (the CPW channels of this wave)
  float* __restrict__ dst = part + (size_t)blockIdx.x * KK * p.NC * p.Cout;
  for (int i = lane; i < KK * p.NC * CPW; i += 64) dst[(size_t)(i / CPW) * p.Cout + co0 + (i % CPW)] = mytab[i];
}

// dw[co][c][t] (+)= sum_blk part[blk][t][c][co]   (fixed order), c < NC
__global__ void onehot_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nblk, int Cout,
                                           int C, int NC, int KK, int accumulate) {
  const int total = KK * NC * Cout;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < nblk; ++z) s += part[(size_t)z * total + i];
    const int co = i % Cout, r = i / Cout, c = r % NC, t = r / NC;
    float* o = dw + ((size_t)co * C + c) * KK + t;
    *o = accumulate ? *o + s : s;
  }
}

// out[co][c - NC][t] <-> full[co][c][t] for the dense channels c >= NC (dir 0: gather weights; 1: scatter(-add) grads)
__global__ void onehot_dense_w_kernel(float* __restrict__ full, float* __restrict__ dense, int Cout, int C, int NC, int KK,
                                      int dir, int accumulate) {
  const int Cd = C - NC, total = Cout * Cd * KK;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int t = i % KK, r = i / KK, c = r % Cd, co = r / Cd;
    float* f = full + ((size_t)co * C + NC + c) * KK + t;
    if (dir == 0) dense[i] = *f;
    else *f = accumulate ? *f + dense[i] : dense[i];
  }
}

static void onehot_wgrad_geom(const HimConv2d* d, int* nsx, int* nyc, int* rows_per) {
  *nsx = cdiv(d->W + d->pad + 16, 64);  // strips start at column -16
  *rows_per = 66;
  *nyc = cdiv(d->H + 2 * d->pad, *rows_per);
}
static int onehot_wgrad_blocks(const HimConv2d* d) {
  int nsx, nyc, rp;
  onehot_wgrad_geom(d, &nsx, &nyc, &rp);
  return d->B * nsx * nyc;
}
static bool onehot_ok(const HimConv2d* d, int NC) {
  return d->stride == 1 && d->KH == d->KW && (d->KH == 3 || d->KH == 5 || d->KH == 7) && d->pad == d->KH / 2 && NC >= 2 &&
         NC <= d->Cin &&
         (d->Cout % 16) == 0 && (size_t)d->KH * d->KW * NC * 16 * sizeof(float) <= 160 * 1024 && d->OH == d->H &&
         d->OW == d->W;
}
static HimConv2d onehot_dense_desc(const HimConv2d* d, int NC) {
  HimConv2d dd = *d;
  dd.Cin = d->Cin - NC;
  dd.act = HIM_ACT_NONE;
  return dd;
}
// layout of the workspace: [Wt table][dense x copy][dense w copy][dense conv ws]
static size_t onehot_fwd_ws_bytes(const HimConv2d* d, int NC) {
  const int Cd = d->Cin - NC, KK = d->KH * d->KW;
  size_t n = ((size_t)KK * NC * d->Cout + 63) / 64 * 64;
  if (Cd > 0) {
    const HimConv2d dd = onehot_dense_desc(d, NC);
    n += ((size_t)d->B * Cd * d->H * d->W + 63) / 64 * 64 + ((size_t)d->Cout * Cd * KK + 63) / 64 * 64;
    return n * sizeof(float) + fprop_ws_bytes(&dd) + 256;
  }
  return n * sizeof(float) + 256;
}
static size_t onehot_wgrad_ws_bytes(const HimConv2d* d, int NC) {
  const int Cd = d->Cin - NC, KK = d->KH * d->KW;
  size_t n = (size_t)onehot_wgrad_blocks(d) * KK * NC * d->Cout + 64;
  if (Cd > 0) {
    n += ((size_t)d->B * Cd * d->H * d->W + 63) / 64 * 64 + ((size_t)d->Cout * Cd * KK + 63) / 64 * 64;
    return n * sizeof(float) + wgrad_slab_bytes(d->Cout, Cd, d->KH, d->KW, d->B * d->OH * d->OW) + 256;
  }
  return n * sizeof(float) + 256;
}

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
  if ((c->H + 2 * c->pad - c->KH) / c->stride + 1 != c->OH || (c->W + 2 * c->pad - c->KW) / c->stride + 1 != c->OW)
    return fail(HIM_E_UNSUPPORTED, "deconv2d: output_padding %d not representable", t->out_pad);
  return check_conv(c);
}

}  // namespace him

using namespace him;

extern "C" {

int him_set_winograd_min_channels(int c) {
  const int prev = wino_min_c();
  g_wino_min_c = c > 0 ? c : -1;
  return prev;
}

int him_winograd_gemm(const float* a, const float* b, float* c, int M, int K, int N, void* stream) {
  if (!a || !b || !c || M <= 4 || K < 16 || (K % 16) || N <= 0 || (N % 128))
    return fail(HIM_E_INVALID, "winograd gemm: need M > 4, K %% 16 == 0, N %% 128 == 0 (got %d, %d, %d)", M, K, N);
  if ((long long)16 * K * N >= (1ll << 31) || (long long)16 * M * N >= (1ll << 31))
    return fail(HIM_E_UNSUPPORTED, "winograd gemm: operand larger than 2^31 elements");
  return wino_batched_gemm(a, b, c, M, K, N, (hipStream_t)stream);
}

size_t him_conv2d_fwd_ws(const HimConv2d* d) { return d ? fprop_ws_bytes(d) : 0; }

int him_conv2d_fwd(const HimConv2d* d, const float* x, const float* w, const float* bias, float* y, void* ws,
                   size_t ws_bytes, void* stream) {
  int rc = check_conv(d);
  if (rc) return rc;
  return run_fprop(d, x, w, bias, y, ws, ws_bytes, (hipStream_t)stream);
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

size_t him_conv2d_bwd_weight_ws(const HimConv2d* d) {
  return d ? wgrad_ws_bytes(d->Cout, d->Cin, d->KH, d->KW, d->B * d->OH * d->OW, d->Cout, conv_wino_wgrad_floats(d)) : 0;
}

int him_conv2d_bwd_weight(const HimConv2d* d, const float* x, const float* dy, float* dw, float* dbias,
                          int accumulate, void* ws, size_t ws_bytes, void* stream) {
  int rc = check_conv(d);
  if (rc) return rc;
  if (dw) {
    rc = run_wgrad(dy, x, dw, d->Cout, d->Cin, d->B, d->H, d->W, d->OH, d->OW, d->KH, d->KW, d->stride, d->pad,
                   d->pad_mode, accumulate, ws, ws_bytes, (hipStream_t)stream);
    if (rc) return rc;
  }
  if (dbias) {
    const size_t off = wgrad_slab_bytes(d->Cout, d->Cin, d->KH, d->KW, d->B * d->OH * d->OW, conv_wino_wgrad_floats(d));
    if (ws_bytes < off) return fail(HIM_E_WORKSPACE, "bwd_weight ws too small");
    rc = run_bias_grad(dy, dbias, d->B, d->Cout, d->OH * d->OW, accumulate, (char*)ws + off, ws_bytes - off,
                       (hipStream_t)stream);
  }
  return rc;
}


size_t him_conv2d_onehot_fwd_ws(const HimConv2d* d, int n_onehot) {
  return (d && !check_conv(d) && onehot_ok(d, n_onehot)) ? onehot_fwd_ws_bytes(d, n_onehot) : 0;
}

int him_conv2d_onehot_fwd(const HimConv2d* d, const float* label, int n_onehot, const float* x, const float* w,
                          const float* bias, float* y, void* ws, size_t ws_bytes, void* stream) {
  int rc = check_conv(d);
  if (rc) return rc;
  if (!onehot_ok(d, n_onehot)) return fail(HIM_E_UNSUPPORTED, "onehot conv: needs stride 1, odd square kernel, same padding, Cout %% 16 == 0");
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
    rc = him_copy_channels(x, d->Cin, NC, xd, Cd, 0, Cd, d->B, HW, nullptr, 0, 0, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(onehot_dense_w_kernel, dim3(cdiv((long long)d->Cout * Cd * KK, 256)), dim3(256), 0, st,
                       (float*)w, wd, d->Cout, d->Cin, NC, KK, 0, 0);
    const HimConv2d dd = onehot_dense_desc(d, NC);
    rc = run_fprop(&dd, xd, wd, bias, y, cws, fprop_ws_bytes(&dd), st);
    if (rc) return rc;
  }
  OneHotP p;
  p.label = label;
  p.B = d->B; p.H = d->H; p.W = d->W; p.NC = NC; p.KS = d->KH; p.pad = d->pad;
  p.reflect = d->pad_mode == HIM_PAD_REFLECT;
  p.Cout = d->Cout;
  p.npix = d->B * HW;
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
  else HIM_OH_FWD(3)
#undef HIM_OH_FWD
  return check_launch("onehot_conv_fwd");
}

size_t him_conv2d_onehot_bwd_weight_ws(const HimConv2d* d, int n_onehot) {
  return (d && !check_conv(d) && onehot_ok(d, n_onehot)) ? onehot_wgrad_ws_bytes(d, n_onehot) + bias_ws_bytes(d->Cout) : 0;
}

int him_conv2d_onehot_bwd_weight(const HimConv2d* d, const float* label, int n_onehot, const float* x, const float* dy,
                                 float* dw, float* dbias, int accumulate, void* ws, size_t ws_bytes, void* stream) {
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
  p.npix = d->B * HW;
  if (dw) {
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
    if (Cd > 0) {
      float* dwd = xd + ((size_t)d->B * Cd * HW + 63) / 64 * 64;
      float* wws = dwd + ((size_t)d->Cout * Cd * KK + 63) / 64 * 64;
      rc = him_copy_channels(x, d->Cin, NC, xd, Cd, 0, Cd, d->B, HW, nullptr, 0, 0, stream);
      if (rc) return rc;
      rc = run_wgrad(dy, xd, dwd, d->Cout, Cd, d->B, d->H, d->W, d->OH, d->OW, d->KH, d->KW, d->stride, d->pad,
                     d->pad_mode, 0, wws, wgrad_slab_bytes(d->Cout, Cd, d->KH, d->KW, d->B * d->OH * d->OW), st);
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
  return wgrad_ws_bytes(c.Cout, c.Cin, c.KH, c.KW, c.B * c.OH * c.OW, t->Cout);
}

int him_deconv2d_bwd_weight(const HimDeconv2d* t, const float* x, const float* dy, float* dw, float* dbias,
                            int accumulate, void* ws, size_t ws_bytes, void* stream) {
  HimConv2d c;
  int rc = adjoint_of(t, &c);
  if (rc) return rc;
  // adjoint conv: "input" = deconv output gradient dy, "output gradient" = deconv input x
  if (dw) {
    rc = run_wgrad(x, dy, dw, c.Cout, c.Cin, c.B, c.H, c.W, c.OH, c.OW, c.KH, c.KW, c.stride, c.pad, HIM_PAD_ZERO,
                   accumulate, ws, ws_bytes, (hipStream_t)stream);
    if (rc) return rc;
  }
  if (dbias) {
    const size_t off = wgrad_slab_bytes(c.Cout, c.Cin, c.KH, c.KW, c.B * c.OH * c.OW);
    if (ws_bytes < off) return fail(HIM_E_WORKSPACE, "bwd_weight ws too small");
    rc = run_bias_grad(dy, dbias, t->B, t->Cout, t->OH * t->OW, accumulate, (char*)ws + off, ws_bytes - off,
                       (hipStream_t)stream);
  }
  return rc;
}

}  // extern "C"
