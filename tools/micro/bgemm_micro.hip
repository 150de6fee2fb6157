// Stand-alone microbenchmark + sampled fp64 check of bgemm_kernel (csrc/him_bgemm.inc): the batched fp32-MFMA GEMM of the
// separate-transform Winograd pipeline with LDS-DMA operand loads, in its three layout variants.
//     make -C tools/micro bgemm_micro       (cross-compiles for gfx950)
//     tools/micro/bgemm_micro [M] [K] [N] [iters] [nz]        (on the GPU box)
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
}  // namespace him
using namespace him;

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 1024, K = argc > 2 ? atoi(argv[2]) : 1024, N = argc > 3 ? atoi(argv[3]) : 1024;
  const int iters = argc > 4 ? atoi(argv[4]) : 20, nz = argc > 5 ? atoi(argv[5]) : 16;
  const int pwgs = argc > 6 ? atoi(argv[6]) : 0;       // > 0: the persistent kernel on that many workgroups
  const size_t na = (size_t)nz * M * K, nb = (size_t)nz * K * N, nc = (size_t)nz * M * N;
  std::vector<float> ha(na), hb(nb), hc(nc);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& v : ha) v = rnd() * 0.05f;
  for (auto& v : hb) v = rnd();
  float *a, *b, *c;
  hipMalloc(&a, na * 4); hipMalloc(&b, nb * 4); hipMalloc(&c, nc * 4);
  hipMemcpy(a, ha.data(), na * 4, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), nb * 4, hipMemcpyHostToDevice);
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  int bad = 0;
  // (al, bl, mirror): element accessors of the SAME host arrays under each layout interpretation
  const int cfg[3][3] = {{0, 1, 0}, {1, 1, 1}, {0, 0, 0}};
  const char* names[3] = {"fwd   A[z][M][K] B[z][K][N]", "dgrad A[s(z)][K][M] B[z][K][N]", "wgrad A[z][M][K] B[z][N][K]"};
  for (int v = 0; v < (nz == 16 ? 3 : 1); ++v) {      // 36 positions (F(4x4)): the forward layout only (the mirror table here is F(2x2)'s)
    const int al = cfg[v][0], bl = cfg[v][1], mir = cfg[v][2];
    hipMemsetAsync(c, 0xff, nc * 4, st);
    for (int i = 0; i < 3; ++i) launch_bgemm(a, b, c, M, K, N, nz, 4, al, bl, mir, st, pwgs);
    hipEventRecord(e0, st);
    for (int i = 0; i < iters; ++i) launch_bgemm(a, b, c, M, K, N, nz, 4, al, bl, mir, st, pwgs);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= iters;
    hipError_t err = hipGetLastError();
    hipMemcpy(hc.data(), c, nc * 4, hipMemcpyDeviceToHost);
    double worst = 0, scale = 0;
    for (int t = 0; t < 6000; ++t) {
      s = s * 1664525u + 1013904223u;
      const int z = (s >> 4) % nz, m = (s >> 9) % M, n = (s >> 19) % N;
      int za = z;
      if (mir) {
        const int r4[4] = {3, 1, 2, 0};
        za = 4 * r4[z >> 2] + r4[z & 3];
      }
      double r = 0;
      for (int k = 0; k < K; ++k) {
        const float av = al == 0 ? ha[((size_t)za * M + m) * K + k] : ha[((size_t)za * K + k) * M + m];
        const float bv = bl == 1 ? hb[((size_t)z * K + k) * N + n] : hb[((size_t)z * N + n) * K + k];
        r += (double)av * bv;
      }
      worst = fmax(worst, fabs(r - hc[((size_t)z * M + m) * N + n]));
      scale = fmax(scale, fabs(r));
    }
    printf("bgemm_micro %-34s [%d]x(%dx%d)x(%dx%d): %.4f ms  %.1f TFLOP/s  max rel err %.2e  (%s)\n", names[v], nz, M, K, K, N,
           ms, 2.0 * nz * M * K * N / ms / 1e9, worst / scale, hipGetErrorString(err));
    if (!(worst / scale < 1e-4)) bad = 1;
  }
  return bad;
}
