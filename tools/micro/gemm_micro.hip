// Stand-alone microbenchmark of gconv_fast_kernel as the batched Winograd GEMM  C[z] = A[z] (M x K) * B[z] (K x N),
// z = 16 transform positions -- the same kernel source the library compiles (csrc/him_gconv_fast.inc), built in seconds
// instead of minutes so that kernel-schedule experiments are cheap:
//     make -C tools/micro            (here: cross-compiles for gfx950)
//     tools/micro/gemm_micro [M] [K] [N] [iters]        (on the GPU box)
// Prints the HIP-event time per launch, executed TFLOP/s and the max error of a sampled fp64 check.
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
#include "../../neurips18_hierchical_image_manipulation_amd/csrc/him_gconv_fast.inc"
}  // namespace him
using namespace him;

#ifndef MICRO_CFG
#define MICRO_CFG 2, 2, 2, 2
#endif

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 1024, K = argc > 2 ? atoi(argv[2]) : 1024, N = argc > 3 ? atoi(argv[3]) : 1024;
  const int iters = argc > 4 ? atoi(argv[4]) : 20;
  const size_t na = (size_t)16 * M * K, nb = (size_t)16 * K * N, nc = (size_t)16 * M * N;
  std::vector<float> ha(na), hb(nb), hc(nc);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& v : ha) v = rnd() * 0.05f;
  for (auto& v : hb) v = rnd();
  float *a, *b, *c;
  hipMalloc(&a, na * 4); hipMalloc(&b, nb * 4); hipMalloc(&c, nc * 4);
  hipMemcpy(a, ha.data(), na * 4, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), nb * 4, hipMemcpyHostToDevice);
  GConvP g;
  memset(&g, 0, sizeof(g));
  g.src = b; g.dst = c; g.M = M; g.C2 = K; g.B = 16; g.SH = 1; g.SW = N; g.DH = 1; g.DW = N;
  g.oys = g.oxs = g.sy = g.sx = g.dy = g.dx = 1;
  g.pad_mode = HIM_PAD_ZERO; g.act = HIM_ACT_NONE; g.nphase = 1; g.fast = 1; g.wbatch = M * K;
  GPhase& P = g.ph[0];
  P.A = a; P.At = a; P.C2p = K; P.K = K; P.JH = P.JW = 1;
  P.fJHJW = make_fastdiv(1); P.fJW = make_fastdiv(1); P.NA = 1; P.NC = N;
  const long long maxN = (long long)16 * N;
  dim3 grid((unsigned)(((maxN + 127) / 128) * ((M + 127) / 128)), 1, 1);
  hipStream_t st;
  hipStreamCreate(&st);
  for (int i = 0; i < 3; ++i) launch_fast_cfg<MICRO_CFG>(g, grid, st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  for (int i = 0; i < iters; ++i) launch_fast_cfg<MICRO_CFG>(g, grid, st);
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  hipError_t err = hipGetLastError();
  hipMemcpy(hc.data(), c, nc * 4, hipMemcpyDeviceToHost);
  double worst = 0, scale = 0;
  for (int t = 0; t < 4000; ++t) {
    s = s * 1664525u + 1013904223u;
    const int z = (s >> 4) % 16, m = (s >> 9) % M, n = (s >> 19) % N;
    double r = 0;
    for (int k = 0; k < K; ++k) r += (double)ha[((size_t)z * M + m) * K + k] * hb[((size_t)z * K + k) * N + n];
    worst = fmax(worst, fabs(r - hc[((size_t)z * M + m) * N + n]));
    scale = fmax(scale, fabs(r));
  }
  printf("gemm_micro [16]x(%dx%d)x(%dx%d): %.4f ms  %.1f TFLOP/s  max rel err %.2e  (%s)\n", M, K, K, N, ms,
         2.0 * 16 * M * K * N / ms / 1e9, worst / scale, hipGetErrorString(err));
  return worst / scale < 1e-4 ? 0 : 1;
}
