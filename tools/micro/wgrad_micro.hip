// Stand-alone timing harness of wgrad_fast_kernel (csrc/him_conv_wgrad.inc): one layer's weight gradient, split-K launch only
// (no finish pass):   tools/micro/wgrad_micro [B] [C] [H] [W] [M] [KS] [stride] [pad] [iters]
// -DHIM_WG_DBG=1|2|4|8 switch parts of the K-step off (timing only: 1 no gathered loads, 2 no dY loads, 4 no LDS writes, 8 no MFMAs).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "../../neurips18_hierchical_image_manipulation_amd/csrc/him_common.h"
namespace him {
char* err_buf() { static thread_local char b[512]; return b; }
typedef float f32x16 __attribute__((ext_vector_type(16)));
#include "../../neurips18_hierchical_image_manipulation_amd/csrc/him_conv_wgrad.inc"
}
using namespace him;
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 8, C = argc > 2 ? atoi(argv[2]) : 512, H = argc > 3 ? atoi(argv[3]) : 32, W = argc > 4 ? atoi(argv[4]) : 64;
  const int M = argc > 5 ? atoi(argv[5]) : 1024, KS = argc > 6 ? atoi(argv[6]) : 3, stride = argc > 7 ? atoi(argv[7]) : 2, pad = argc > 8 ? atoi(argv[8]) : 1;
  const int iters = argc > 9 ? atoi(argv[9]) : 20;
  const int OH = (H + 2 * pad - KS) / stride + 1, OW = (W + 2 * pad - KS) / stride + 1;
  HimAlgo a; memset(&a, 0, sizeof(a));
  WGradP p; memset(&p, 0, sizeof(p));
  const size_t nx = (size_t)B * C * H * W, ny = (size_t)B * M * OH * OW;
  std::vector<float> hx(nx), hy(ny);
  unsigned s = 99u;
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
  for (auto& v : hy) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
  float *x, *dy, *out;
  hipMalloc(&x, nx * 4); hipMalloc(&dy, ny * 4);
  hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(dy, hy.data(), ny * 4, hipMemcpyHostToDevice);
  p.dy = dy; p.x = x; p.M = M; p.C = C; p.B = B; p.H = H; p.W = W; p.OH = OH; p.OW = OW; p.KH = KS; p.KW = KS; p.stride = stride; p.pad = pad;
  p.pad_mode = 0; p.Np = C * KS * KS; p.Kdim = B * OH * OW;
  p.fKK = make_fastdiv((uint32_t)(KS * KS)); p.fKW = make_fastdiv((uint32_t)KS); p.fOW = make_fastdiv((uint32_t)OW);
  int BM, BN, fs;
  wgrad_fast_cfg(a, M, C, p.Kdim, KS * KS, &BM, &BN, &fs);
  if (argc > 10) fs = atoi(argv[10]);
  p.splits = fs; p.accumulate = 0;
  int kc = (p.Kdim + fs - 1) / fs; p.kchunk = ((kc + 31) / 32) * 32;
  hipMalloc(&out, (size_t)fs * M * p.Np * 4); p.out = out;
  dim3 grid(p.Np / BN, (M + BM - 1) / BM, fs), block(256);
  auto launch = [&]() {
    if ((OH * OW) % 4 == 0) {
      if (BM == 128 && BN == 128) hipLaunchKernelGGL((wgrad_fast_kernel<2, 2, false, true>), grid, block, 0, 0, p);
      else if (BM == 128) hipLaunchKernelGGL((wgrad_fast_kernel<2, 1, false, true>), grid, block, 0, 0, p);
      else if (BN == 128) hipLaunchKernelGGL((wgrad_fast_kernel<1, 2, false, true>), grid, block, 0, 0, p);
      else hipLaunchKernelGGL((wgrad_fast_kernel<1, 1, false, true>), grid, block, 0, 0, p);
    } else {
      if (BM == 128 && BN == 128) hipLaunchKernelGGL((wgrad_fast_kernel<2, 2, false, false>), grid, block, 0, 0, p);
      else hipLaunchKernelGGL((wgrad_fast_kernel<2, 1, false, false>), grid, block, 0, 0, p);
    }
  };
  for (int i = 0; i < 3; ++i) launch();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
  const double fl = 2.0 * M * (double)p.Np * p.Kdim;
  // spot check against the host (64 samples)
  std::vector<float> ho((size_t)fs * M * p.Np);
  hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0, scale = 0;
  for (int tcase = 0; tcase < 64; ++tcase) {
    s = s * 1664525u + 1013904223u; const int m = (s >> 4) % M; s = s * 1664525u + 1013904223u; const int np = (s >> 4) % p.Np;
    const int tap = np / C, ci = np % C, th = tap / KS, tw = tap % KS;
    double r = 0;
    for (int b = 0; b < B; ++b) for (int oh = 0; oh < OH; ++oh) for (int ow = 0; ow < OW; ++ow) {
      const int ih = oh * stride + th - pad, iw = ow * stride + tw - pad;
      if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
      r += (double)hy[(((size_t)b * M + m) * OH + oh) * OW + ow] * hx[(((size_t)b * C + ci) * H + ih) * W + iw];
    }
    double g = 0; for (int z = 0; z < fs; ++z) g += ho[((size_t)z * M + m) * p.Np + np];
    worst = std::max(worst, fabs(g - r)); scale = std::max(scale, fabs(r));
  }
  printf("wgrad_fast B%d %d->%d %dx%d k%d s%d: tile %dx%d splits %d grid (%d,%d,%d): %.4f ms  %.1f TFLOP/s (%.3f of peak)  max err %.2e of %.2e  dbg %d (%s)\n",
         B, C, M, H, W, KS, stride, BM, BN, fs, grid.x, grid.y, grid.z, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3, worst, scale,
#ifdef HIM_WG_DBG
         HIM_WG_DBG,
#else
         0,
#endif
         hipGetErrorString(hipGetLastError()));
  return 0;
}
