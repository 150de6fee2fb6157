// EXPERIMENT (not part of the library): the batched Winograd GEMM with fp32 operands split EXACTLY into three bf16 terms
// (x = hi + mid + lo, each an 8-bit significand) and the partial products run on the bf16 matrix pipe
// (v_mfma_f32_32x32x16_bf16, 16x the fp32-MFMA rate) with fp32 accumulation.  NT = 9: all nine partial products --
// every a*b is then formed EXACTLY (24 x 24 bits as 3 x 3 exact 8 x 8-bit products), only the accumulation is fp32;
// NT = 6: the three products of order 2^-24 and below dropped.  Measures time AND error against fp64, next to the native
// fp32-MFMA kernel (csrc/him_bgemm.inc) on the same data.
//     tools/micro/bgemm_split_micro [M] [K] [N] [iters]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../neurips18_hierchical_image_manipulation_amd/csrc/him_common.h"

namespace him {
char* err_buf() {
  static thread_local char b[512];
  return b;
}
#include "../../neurips18_hierchical_image_manipulation_amd/csrc/him_bgemm.inc"

struct SplitP {
  const unsigned short* A[3];   // [z][M][K] bf16 planes hi, mid, lo
  const unsigned short* B[3];   // [z][N][K]
  float* C;
  int M, N, K, nz;
};
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16s __attribute__((ext_vector_type(16)));

template <int NT>
__global__ __launch_bounds__(256, 2) void bgemm_split_kernel(const SplitP p) {
  constexpr int BM = 128, BN = 128, BK = 16, ST = 3;
  constexpr int PLB = 128 * 32;        // bytes of one plane tile (128 rows x 16 bf16)
  constexpr int STB = 6 * PLB;         // bytes of one stage: A h|m|l, B h|m|l  (24 KB)
  __shared__ __attribute__((aligned(16))) unsigned char lds[ST * STB];   // 72 KB
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned lds_base = lds_addr_u(lds);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int tm = p.M / BM, tn = p.N / BN, per = tm * tn;
  int z, mt, nt;
  {
    const int total = gridDim.x, L = blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = L & 7, slot = L >> 3;
    const int T = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    z = T / per;
    const int rem = T - z * per;
    mt = rem / tn;
    nt = rem - mt * tn;
  }
  const int K = p.K, nk = K / BK;
  // DMA: 24 wave-instructions per stage (6 planes x 4 parts of 32 rows), wave w issues 6: idx = 6 w + q
  const unsigned short* gp[6];
  unsigned ldst[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int idx = 6 * wave + q, plane6 = idx >> 2, part = idx & 3;       // plane6: 0..2 A, 3..5 B
    const int row = 32 * part + (lane >> 1), pos = lane & 1, gh = pos ^ ((row >> 3) & 1);
    const unsigned short* base = plane6 < 3 ? p.A[plane6] + ((size_t)z * p.M + (size_t)mt * BM) * K
                                            : p.B[plane6 - 3] + ((size_t)z * p.N + (size_t)nt * BN) * K;
    gp[q] = base + (size_t)row * K + 8 * gh;
    ldst[q] = (unsigned)(((6 * wave_u + q) >> 2) * PLB + ((6 * wave_u + q) & 3) * 1024);
  }
#define SP_LOAD(ks_)                                                              \
  {                                                                               \
    const int kq = (ks_) < nk ? (ks_) : nk - 1;                                   \
    const unsigned sl = lds_base + (unsigned)(((ks_) % ST) * STB);                \
    _Pragma("unroll") for (int q = 0; q < 6; ++q) glds16(gp[q] + (size_t)kq * BK, sl + ldst[q]); \
  }
  f32x16s acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  int aoff[2], boff[2];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int ra = wm * 64 + blk * 32 + l31, rb = wn * 64 + blk * 32 + l31;
    aoff[blk] = ra * 32 + ((lh ^ ((ra >> 3) & 1)) * 16);
    boff[blk] = 3 * PLB + rb * 32 + ((lh ^ ((rb >> 3) & 1)) * 16);
  }
  SP_LOAD(0)
  SP_LOAD(1)
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    SP_LOAD(ks + 2)
    const unsigned char* __restrict__ sb = lds + (ks % ST) * STB;
    bf16x8 a[3][2], b[3][2];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        a[pl][blk] = *(const bf16x8*)(sb + pl * PLB + aoff[blk]);
        b[pl][blk] = *(const bf16x8*)(sb + pl * PLB + boff[blk]);
      }
    // smallest terms first
#define SP_MM(pa, pb)                                                                                           \
  _Pragma("unroll") for (int x = 0; x < 2; ++x) _Pragma("unroll") for (int y = 0; y < 2; ++y) acc[x][y] =     \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa][x], b[pb][y], acc[x][y], 0, 0, 0);
    if (NT == 9) { SP_MM(2, 2) SP_MM(1, 2) SP_MM(2, 1) }
    SP_MM(0, 2) SP_MM(2, 0) SP_MM(1, 1) SP_MM(0, 1) SP_MM(1, 0) SP_MM(0, 0)
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float* __restrict__ Cb = p.C + (size_t)z * p.M * p.N;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int m0 = mt * BM + wm * 64 + a * 32, n0 = nt * BN + wn * 64 + b * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) Cb[(size_t)(m0 + (r & 3) + 8 * (r >> 2) + 4 * lh) * p.N + n0] = acc[a][b][r];
    }
}
}  // namespace him
using namespace him;

static unsigned short bf16_rn(float x) {
  unsigned u;
  memcpy(&u, &x, 4);
  const unsigned r = u + 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(r >> 16);
}
static float bf16_f(unsigned short h) {
  const unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 1024, K = argc > 2 ? atoi(argv[2]) : 1024, N = argc > 3 ? atoi(argv[3]) : 1024;
  const int iters = argc > 4 ? atoi(argv[4]) : 20, nz = 16;
  const size_t na = (size_t)nz * M * K, nb = (size_t)nz * N * K, nc = (size_t)nz * M * N;
  std::vector<float> ha(na), hb(nb), hc(nc);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
  auto gauss = [&]() { float v = 0; for (int i = 0; i < 6; ++i) v += rnd(); return v; };
  for (auto& v : ha) v = gauss() * 0.05f;
  for (auto& v : hb) v = gauss();
  std::vector<unsigned short> sa[3], sb[3];
  double split_res = 0;
  for (int pl = 0; pl < 3; ++pl) { sa[pl].resize(na); sb[pl].resize(nb); }
  auto split = [&](const std::vector<float>& src, std::vector<unsigned short>* dst) {
    for (size_t i = 0; i < src.size(); ++i) {
      const float x = src[i];
      const unsigned short h = bf16_rn(x);
      const float r1 = x - bf16_f(h);
      const unsigned short m = bf16_rn(r1);
      const float r2 = r1 - bf16_f(m);
      const unsigned short l = bf16_rn(r2);
      dst[0][i] = h; dst[1][i] = m; dst[2][i] = l;
      split_res = fmax(split_res, fabs((double)r2 - (double)bf16_f(l)) / fmax(fabs((double)x), 1e-30));
    }
  };
  split(ha, sa);
  split(hb, sb);
  printf("split residual (x - hi - mid - lo) / |x| max: %.3e\n", split_res);
  float *a, *b, *c;
  unsigned short *da[3], *db[3];
  hipMalloc(&a, na * 4); hipMalloc(&b, nb * 4); hipMalloc(&c, nc * 4);
  hipMemcpy(a, ha.data(), na * 4, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), nb * 4, hipMemcpyHostToDevice);
  for (int pl = 0; pl < 3; ++pl) {
    hipMalloc(&da[pl], na * 2); hipMalloc(&db[pl], nb * 2);
    hipMemcpy(da[pl], sa[pl].data(), na * 2, hipMemcpyHostToDevice);
    hipMemcpy(db[pl], sb[pl].data(), nb * 2, hipMemcpyHostToDevice);
  }
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  SplitP sp;
  for (int pl = 0; pl < 3; ++pl) { sp.A[pl] = da[pl]; sp.B[pl] = db[pl]; }
  sp.C = c; sp.M = M; sp.N = N; sp.K = K; sp.nz = nz;
  const dim3 grid((unsigned)(nz * (M / 128) * (N / 128))), blk(256);
  // sample set + fp64 reference
  const int NS = 4000;
  std::vector<int> sz(NS), sm(NS), sn(NS);
  std::vector<double> ref(NS), mag(NS);
  for (int t = 0; t < NS; ++t) {
    s = s * 1664525u + 1013904223u;
    sz[t] = (s >> 4) % nz; sm[t] = (s >> 9) % M; sn[t] = (s >> 19) % N;
    double r = 0, g = 0;
    for (int k = 0; k < K; ++k) {
      const double pr = (double)ha[((size_t)sz[t] * M + sm[t]) * K + k] * hb[((size_t)sz[t] * N + sn[t]) * K + k];
      r += pr; g += fabs(pr);
    }
    ref[t] = r; mag[t] = g;
  }
  auto report = [&](const char* name, float ms) {
    hipMemcpy(hc.data(), c, nc * 4, hipMemcpyDeviceToHost);
    double worst = 0, sq = 0, rsq = 0, wmag = 0;
    for (int t = 0; t < NS; ++t) {
      const double e = hc[((size_t)sz[t] * M + sm[t]) * N + sn[t]] - ref[t];
      worst = fmax(worst, fabs(e)); sq += e * e; rsq += ref[t] * ref[t];
      wmag = fmax(wmag, fabs(e) / mag[t]);
    }
    printf("%-28s %.4f ms  %.1f TFLOP/s fp32-equivalent | err vs fp64: rel L2 %.3e, max |e| / sum|a b| %.3e  (%s)\n", name, ms,
           2.0 * nz * M * K * N / ms / 1e9, sqrt(sq / rsq), wmag, hipGetErrorString(hipGetLastError()));
  };
  for (int v = 0; v < 3; ++v) {
    hipMemsetAsync(c, 0xff, nc * 4, st);
    auto launch = [&]() {
      if (v == 0) launch_bgemm(a, b, c, M, K, N, nz, 4, 0, 0, false, st);
      else if (v == 1) hipLaunchKernelGGL((bgemm_split_kernel<9>), grid, blk, 0, st, sp);
      else hipLaunchKernelGGL((bgemm_split_kernel<6>), grid, blk, 0, st, sp);
    };
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0, st);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    report(v == 0 ? "native fp32 MFMA (bgemm)" : v == 1 ? "3 x bf16 split, 9 products" : "3 x bf16 split, 6 products", ms / iters);
  }
  return 0;
}
