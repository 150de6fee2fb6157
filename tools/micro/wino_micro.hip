// Stand-alone microbenchmark + correctness check of the fused Winograd conv kernel (csrc/him_wino_fused.inc):
//     tools/micro/wino_micro [B] [Ci] [H] [W] [Co] [reflect 0|1] [iters]
// Prints time per launch, direct-form-equivalent and executed TFLOP/s, and the max error of sampled outputs against an
// fp64 direct convolution on the host.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "../../neurips18_hierchical_image_manipulation_amd/csrc/him_common.h"

namespace him {
char* err_buf() {
  static thread_local char b[512];
  return b;
}
#include "../../neurips18_hierchical_image_manipulation_amd/csrc/him_wino_fused.inc"
#include "../../neurips18_hierchical_image_manipulation_amd/csrc/him_wino_fused2.inc"
}  // namespace him
using namespace him;

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 8, Ci = argc > 2 ? atoi(argv[2]) : 256, H = argc > 3 ? atoi(argv[3]) : 64;
  const int W = argc > 4 ? atoi(argv[4]) : 128, Co = argc > 5 ? atoi(argv[5]) : 256, refl = argc > 6 ? atoi(argv[6]) : 0;
  const int iters = argc > 7 ? atoi(argv[7]) : 20;
  const bool chunk4 = argc > 8 && atoi(argv[8]) == 4;      // 4-channel chunks (80 KB of LDS)
  const bool v2 = argc > 8 && atoi(argv[8]) == 2;          // persistent kernel (him_wino_fused2.inc)
  const int wgs = argc > 9 ? atoi(argv[9]) : 0;
  const bool gate = argc > 10 && atoi(argv[10]) != 0;      // ReLU-gate mask epilogue (the gated data gradient's form)
  if (!wino_fused_shape_ok(Co, Ci, 3, 3, 1, 1, B, H, W)) { printf("shape not supported\n"); return 2; }
  const size_t nx = (size_t)B * Ci * H * W, nw = (size_t)Co * Ci * 9, ny = (size_t)B * Co * H * W;
  std::vector<float> hx(nx), hw(nw), hb(Co), hy(ny);
  unsigned s = 777u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& v : hx) v = rnd() * 2.f;
  const float ws = 1.f / sqrtf((float)Ci * 9.f);
  for (auto& v : hw) v = rnd() * 2.f * ws;
  for (auto& v : hb) v = rnd() * 0.2f;
  float *x, *w, *bias, *y, *Uf;
  hipMalloc(&x, nx * 4); hipMalloc(&w, nw * 4); hipMalloc(&bias, Co * 4); hipMalloc(&y, ny * 4);
  hipMalloc(&Uf, wino_fused_panel_floats(Co, Ci) * 4);
  hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
  hipMemcpy(bias, hb.data(), Co * 4, hipMemcpyHostToDevice);
  hipStream_t st;
  hipStreamCreate(&st);
  hipLaunchKernelGGL((wino_fused_weight_kernel<0>), dim3((Ci + 255) / 256, Co), dim3(256), 0, st, w, Uf, Co, Ci);
  float* gmask = nullptr;
  std::vector<float> hm;
  if (gate) {
    hm.resize(ny);
    for (auto& v : hm) v = rnd();
    hipMalloc(&gmask, ny * 4);
    hipMemcpy(gmask, hm.data(), ny * 4, hipMemcpyHostToDevice);
  }
  auto launch = [&]() {
    if (v2) run_wino_fused2(B, Ci, H, W, Co, refl != 0, x, Uf, bias, HIM_ACT_RELU, 0.f, y, st, gmask, wgs);
    else run_wino_fused(B, Ci, H, W, Co, refl != 0, x, Uf, bias, HIM_ACT_RELU, 0.f, y, st, gmask, chunk4);
  };
  for (int i = 0; i < 3; ++i) launch();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  hipError_t err = hipGetLastError();
  hipMemcpy(hy.data(), y, ny * 4, hipMemcpyDeviceToHost);
  if (v2 && argc > 11) {      // determinism / partition independence: the same input under other workgroup counts, bit for bit
    std::vector<float> h2(ny);
    int bad_runs = 0;
    for (int rep = 0; rep < 12; ++rep) {
      const int w2 = rep < 4 ? wgs : (rep % 4 == 0 ? 97 : rep % 4 == 1 ? 64 : rep % 4 == 2 ? 200 : 13);
      hipMemsetAsync(y, 0xff, ny * 4, st);
      run_wino_fused2(B, Ci, H, W, Co, refl != 0, x, Uf, bias, HIM_ACT_RELU, 0.f, y, st, gmask, w2);
      hipStreamSynchronize(st);
      hipMemcpy(h2.data(), y, ny * 4, hipMemcpyDeviceToHost);
      size_t diff = 0, first = 0;
      for (size_t i = 0; i < ny; ++i)
        if (memcmp(&h2[i], &hy[i], 4) != 0) { if (!diff) first = i; ++diff; }
      if (diff) { ++bad_runs; printf("  rep %d (wgs %d): %zu of %zu outputs differ, first at %zu: %g vs %g\n", rep, w2, diff, ny, first, h2[first], hy[first]); }
    }
    printf("  partition / repeat check: %d of 12 runs differ\n", bad_runs);
  }
  double worst = 0, scale = 0;
  auto at = [&](int b, int c, int yy, int xx) -> double {
    if (refl) {
      yy = yy < 0 ? -yy : yy; yy = yy >= H ? 2 * (H - 1) - yy : yy;
      xx = xx < 0 ? -xx : xx; xx = xx >= W ? 2 * (W - 1) - xx : xx;
    } else if (yy < 0 || yy >= H || xx < 0 || xx >= W) return 0.0;
    return hx[(((size_t)b * Ci + c) * H + yy) * W + xx];
  };
  for (int tcase = 0; tcase < 6000; ++tcase) {
    s = s * 1664525u + 1013904223u;
    const int b = (tcase % 5 == 0) ? 0 : (s >> 3) % B, co = (s >> 9) % Co;
    s = s * 1664525u + 1013904223u;
    int oy = (s >> 5) % H, ox = (s >> 15) % W;
    if (tcase % 7 == 0) oy = (tcase % 2) ? 0 : H - 1;      // borders
    if (tcase % 11 == 0) ox = (tcase % 2) ? 0 : W - 1;
    double r = hb[co];
    for (int c = 0; c < Ci; ++c)
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r += (double)hw[((size_t)co * Ci + c) * 9 + i * 3 + j] * at(b, c, oy - 1 + i, ox - 1 + j);
    r = r > 0 ? r : 0;
    if (gate && !(hm[(((size_t)b * Co + co) * H + oy) * W + ox] > 0.f)) r = 0;
    worst = fmax(worst, fabs(r - hy[(((size_t)b * Co + co) * H + oy) * W + ox]));
    scale = fmax(scale, fabs(r));
  }
  const double direct = 2.0 * B * Co * (double)H * W * Ci * 9;
  printf("wino_fused%s B%d %d->%d %dx%d %s: %.4f ms  %.1f TFLOP/s direct-form equivalent (%.1f executed)  max rel err %.2e  (%s)\n",
         v2 ? "2" : "", B, Ci, Co, H, W, refl ? "reflect" : "zero", ms, direct / ms / 1e9, direct / 2.25 / ms / 1e9, worst / scale,
         hipGetErrorString(err));
  return worst / scale < 1e-4 ? 0 : 1;
}
