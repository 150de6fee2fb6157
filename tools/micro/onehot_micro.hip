// Stand-alone timing harness of onehot_wgrad_rle_kernel (csrc/him_onehot_rle.inc): config C2's stem (8 x 256 x 512, 35
// classes in 16x16 blocks, 64 output channels, 7x7 reflect).  -DHIM_OH_SKIP=1|2|4 switches phases off (timing only).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../neurips18_hierchical_image_manipulation_amd/csrc/him_common.h"
namespace him {
char* err_buf() { static thread_local char b[512]; return b; }
#include "../../neurips18_hierchical_image_manipulation_amd/csrc/him_onehot_rle.inc"
}
using namespace him;
int main(int argc, char** argv) {
  const int B = 8, H = 256, W = 512, NC = 35, Cout = 64, KS = 7, pad = 3;
  const int blockpx = argc > 1 ? atoi(argv[1]) : 16;
  std::vector<float> lab((size_t)B * H * W), hy((size_t)B * Cout * H * W);
  unsigned s = 7u;
  std::vector<int> ids((size_t)B * (H / blockpx + 1) * (W / blockpx + 1));
  for (auto& v : ids) { s = s * 1664525u + 1013904223u; v = (s >> 10) % NC; }
  for (int b = 0; b < B; ++b) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x)
    lab[((size_t)b * H + y) * W + x] = (float)ids[((size_t)b * (H / blockpx + 1) + y / blockpx) * (W / blockpx + 1) + x / blockpx];
  for (auto& v : hy) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
  float *dl, *dy, *part;
  const int nbands = argc > 2 ? atoi(argv[2]) : 4, rows_per = (H + nbands - 1) / nbands, nblk = B * nbands;
  hipMalloc(&dl, lab.size() * 4); hipMalloc(&dy, hy.size() * 4); hipMalloc(&part, (size_t)nblk * KS * KS * NC * Cout * 4);
  hipMemcpy(dl, lab.data(), lab.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dy, hy.data(), hy.size() * 4, hipMemcpyHostToDevice);
  OneHotP p; p.label = dl; p.B = B; p.H = H; p.W = W; p.NC = NC; p.KS = KS; p.pad = pad; p.reflect = 1; p.Cout = Cout; p.npix = B * H * W; p.stride = 1; p.OH = H; p.OW = W;
  const size_t lds = (size_t)8 * (W + 1) * 8 + (size_t)KS * KS * NC * 8 * 4 + (size_t)(KS + 1) * (W + 2 * pad + 2) * 4 + 64;
  hipFuncSetAttribute((const void*)onehot_wgrad_rle_kernel<7, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((onehot_wgrad_rle_kernel<7, 1>), dim3(nblk, Cout / 8), dim3(512), lds, 0, p, dy, part, nbands, rows_per);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((onehot_wgrad_rle_kernel<7, 1>), dim3(nblk, Cout / 8), dim3(512), lds, 0, p, dy, part, nbands, rows_per);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("onehot_wgrad_rle (skip mask %d, %dx%d label blocks, %d bands): %.4f ms  (%s)\n", HIM_OH_SKIP, blockpx, blockpx, nbands, ms / 10, hipGetErrorString(hipGetLastError()));
  return 0;
}
