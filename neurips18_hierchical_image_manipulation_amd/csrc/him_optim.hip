// Flat-arena Adam, spectral-norm power iteration (forward + full backward), and library metadata.
#include <algorithm>
#include <math.h>

#include "him_common.h"

namespace him {

char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

// torch.optim.Adam single-tensor semantics (torch/optim/adam.py, _single_tensor_adam):
//   exp_avg.lerp_(grad, 1-beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
//   denom = exp_avg_sq.sqrt() / sqrt(bias_correction2) + eps;  param.addcdiv_(exp_avg, denom, value=-lr/bc1)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, size_t n, float w1, float beta2, float one_m_beta2,
                            float bc2_sqrt, float eps, float step_size) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    float mi = m[i];
    // at::lerp: weight < 0.5 ? start + weight*(end-start) : end - (end-start)*(1-weight)
    const float diff = gi - mi;
    mi = w1 < 0.5f ? mi + w1 * diff : gi - diff * (1.f - w1);
    float vi = v[i] * beta2;
    vi = vi + one_m_beta2 * (gi * gi);
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
    m[i] = mi;
    v[i] = vi;
  }
}

// ---- spectral norm -----------------------------------------------------------------------------
// s_j = sum_i u_i W_ij   (coalesced over j)
__global__ void sn_colsum_kernel(const float* __restrict__ W, const float* __restrict__ u, int rows, int cols,
                                 float* __restrict__ s) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cols) return;
  float acc = 0.f;
  for (int i = 0; i < rows; ++i) acc += u[i] * W[(size_t)i * cols + j];
  s[j] = acc;
}
// t_i = sum_j W_ij v_j   (one 256-thread block per row)
__global__ __launch_bounds__(256) void sn_rowdot_kernel(const float* __restrict__ W, const float* __restrict__ v,
                                                        int cols, float* __restrict__ t) {
  __shared__ float sh[8];
  const int i = blockIdx.x;
  float acc = 0.f;
  for (int j = threadIdx.x; j < cols; j += 256) acc += W[(size_t)i * cols + j] * v[j];
  acc = block_sum_256(acc, sh);
  if (threadIdx.x == 0) t[i] = acc;
}
// out = in / (||in|| + eps)   single block
__global__ __launch_bounds__(256) void sn_l2n_kernel(const float* __restrict__ in, int n, float eps,
                                                     float* __restrict__ out) {
  __shared__ float sh[8];
  float q = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) q += in[i] * in[i];
  const float nrm = sqrtf(block_sum_256(q, sh));
  for (int i = threadIdx.x; i < n; i += 256) out[i] = in[i] / (nrm + eps);
}
// u' = t/(|t|+eps), sigma = u'.t
__global__ __launch_bounds__(256) void sn_finish_kernel(const float* __restrict__ t, int n, float eps,
                                                        float* __restrict__ u_out, float* __restrict__ sigma) {
  __shared__ float sh[8];
  float q = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) q += t[i] * t[i];
  const float nrm = sqrtf(block_sum_256(q, sh));
  float d = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float ui = t[i] / (nrm + eps);
    u_out[i] = ui;
    d += ui * t[i];
  }
  d = block_sum_256(d, sh);
  if (threadIdx.x == 0) sigma[0] = d;
}
// a = t/(|t|+eps) + t*eps/(|t|+eps)^2   (d sigma / d (W v) including the path through u')
__global__ __launch_bounds__(256) void sn_bwd_a_kernel(const float* __restrict__ t, int n, float eps,
                                                       float* __restrict__ a) {
  __shared__ float sh[8];
  float q = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) q += t[i] * t[i];
  const float nrm = sqrtf(block_sum_256(q, sh));
  const float d = nrm + eps;
  for (int i = threadIdx.x; i < n; i += 256) a[i] = t[i] / d + t[i] * (eps / (d * d));
}
// ds = dv/(|s|+eps) - s (s.dv)/(|s| (|s|+eps)^2)
__global__ __launch_bounds__(256) void sn_bwd_ds_kernel(const float* __restrict__ s, const float* __restrict__ dv,
                                                        int n, float eps, float* __restrict__ ds) {
  __shared__ float sh[8];
  float q = 0.f, d = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    q += s[i] * s[i];
    d += s[i] * dv[i];
  }
  const float nrm = sqrtf(block_sum_256(q, sh));
  d = block_sum_256(d, sh);
  const float den = nrm + eps;
  const float k = nrm > 0.f ? d / (nrm * den * den) : 0.f;
  for (int i = threadIdx.x; i < n; i += 256) ds[i] = dv[i] / den - s[i] * k;
}
__global__ void sn_bwd_dw_kernel(const float* __restrict__ a, const float* __restrict__ v,
                                 const float* __restrict__ u, const float* __restrict__ ds,
                                 const float* __restrict__ g, int rows, int cols, float* __restrict__ dW,
                                 int accumulate) {
  const size_t n = (size_t)rows * cols;
  const float gs = g[0];
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / cols), j = (int)(idx % cols);
    const float val = gs * (a[i] * v[j] + u[i] * ds[j]);
    dW[idx] = accumulate ? dW[idx] + val : val;
  }
}

__global__ void div_scalar_fwd_kernel(const float* __restrict__ W, const float* __restrict__ sigma,
                                      float* __restrict__ out, size_t n) {
  const float s = sigma[0];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = W[i] / s;
}
__global__ __launch_bounds__(256) void div_scalar_bwd1_kernel(const float* __restrict__ W,
                                                              const float* __restrict__ sigma,
                                                              const float* __restrict__ dout, float* __restrict__ dW,
                                                              float* __restrict__ partial, size_t n, int accumulate) {
  __shared__ float sh[8];
  const float s = sigma[0];
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float d = dout[i];
    acc += d * W[i];
    const float val = d / s;
    dW[i] = accumulate ? dW[i] + val : val;
  }
  acc = block_sum_256(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void div_scalar_bwd2_kernel(const float* __restrict__ partial, int nb,
                                                              const float* __restrict__ sigma,
                                                              float* __restrict__ dsigma) {
  __shared__ float sh[8];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) acc += partial[i];
  acc = block_sum_256(acc, sh);
  if (threadIdx.x == 0) dsigma[0] = -acc / (sigma[0] * sigma[0]);
}

}  // namespace him

using namespace him;
#define ST ((hipStream_t)stream)

extern "C" {

const char* him_version(void) { return "him-hip 0.6 (round 6)"; }
const char* him_arch(void) { return "gfx950"; }
const char* him_last_error(void) { return err_buf(); }

int him_adam_step(float* p, const float* g, float* m, float* v, size_t n, double lr, double beta1, double beta2,
                  double eps, int step, void* stream) {
  if (!n) return HIM_OK;
  if (step < 1) return fail(HIM_E_INVALID, "adam: step must be >= 1");
  // Hyper-parameters arrive as the DOUBLES the caller holds (python floats), exactly as torch.optim.Adam sees them:
  // torch derives 1 - beta2, the bias corrections and lr / bc1 in double and only then rounds each scalar to fp32
  // (round 2 took floats: 1.f - 0.999f = 1.0000467e-3 instead of 1e-3, a 4.7e-5 relative bias of exp_avg_sq).
  const double bc1 = 1.0 - pow(beta1, step);
  const double bc2 = 1.0 - pow(beta2, step);
  const float step_size = (float)(lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  const int nb = (int)std::min<size_t>((n + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(adam_kernel, dim3(nb), dim3(256), 0, ST, p, g, m, v, n, (float)(1.0 - beta1), (float)beta2,
                     (float)(1.0 - beta2), bc2_sqrt, (float)eps, step_size);
  return check_launch("adam");
}

size_t him_sn_ws(int rows, int cols) { return ((size_t)2 * rows + 3 * (size_t)cols + 64) * sizeof(float); }

int him_sn_power_iter_fwd(const float* W, const float* u, int rows, int cols, float* v_out, float* u_out,
                          float* sigma_out, void* ws, size_t ws_bytes, void* stream) {
  if (rows <= 0 || cols <= 0) return fail(HIM_E_INVALID, "sn: bad shape");
  if (!ws || ws_bytes < him_sn_ws(rows, cols)) return fail(HIM_E_WORKSPACE, "sn: ws too small");
  float* t = (float*)ws;
  float* s = t + 2 * rows;
  hipLaunchKernelGGL(sn_colsum_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, ST, W, u, rows, cols, s);
  hipLaunchKernelGGL(sn_l2n_kernel, dim3(1), dim3(256), 0, ST, (const float*)s, cols, 1e-12f, v_out);
  hipLaunchKernelGGL(sn_rowdot_kernel, dim3(rows), dim3(256), 0, ST, W, (const float*)v_out, cols, t);
  hipLaunchKernelGGL(sn_finish_kernel, dim3(1), dim3(256), 0, ST, (const float*)t, rows, 1e-12f, u_out, sigma_out);
  return check_launch("sn_fwd");
}

int him_sn_power_iter_bwd(const float* W, const float* u, const float* v_out, const float* u_out, const float* sigma,
                          const float* g_sigma, int rows, int cols, float* dW, int accumulate, void* ws,
                          size_t ws_bytes, void* stream) {
  (void)u_out;
  (void)sigma;
  if (rows <= 0 || cols <= 0) return fail(HIM_E_INVALID, "sn: bad shape");
  if (!ws || ws_bytes < him_sn_ws(rows, cols)) return fail(HIM_E_WORKSPACE, "sn: ws too small");
  float* t = (float*)ws;
  float* a = t + rows;
  float* s = a + rows;
  float* dv = s + cols;
  float* ds = dv + cols;
  hipLaunchKernelGGL(sn_rowdot_kernel, dim3(rows), dim3(256), 0, ST, W, v_out, cols, t);
  hipLaunchKernelGGL(sn_bwd_a_kernel, dim3(1), dim3(256), 0, ST, (const float*)t, rows, 1e-12f, a);
  hipLaunchKernelGGL(sn_colsum_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, ST, W, (const float*)a, rows, cols, dv);
  hipLaunchKernelGGL(sn_colsum_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, ST, W, u, rows, cols, s);
  hipLaunchKernelGGL(sn_bwd_ds_kernel, dim3(1), dim3(256), 0, ST, (const float*)s, (const float*)dv, cols, 1e-12f, ds);
  const size_t n = (size_t)rows * cols;
  hipLaunchKernelGGL(sn_bwd_dw_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, ST,
                     (const float*)a, v_out, u, (const float*)ds, g_sigma, rows, cols, dW, accumulate);
  return check_launch("sn_bwd");
}

int him_div_scalar_fwd(const float* W, const float* sigma, float* out, size_t n, void* stream) {
  if (!n) return HIM_OK;
  hipLaunchKernelGGL(div_scalar_fwd_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0,
                     ST, W, sigma, out, n);
  return check_launch("div_scalar_fwd");
}

int him_div_scalar_bwd(const float* W, const float* sigma, const float* dout, float* dW, float* dsigma, size_t n,
                       int accumulate, void* ws, size_t ws_bytes, void* stream) {
  if (!n) return HIM_OK;
  const int nb = (int)std::min<size_t>((n + 1023) / 1024, 1024);
  if (!ws || ws_bytes < (size_t)nb * sizeof(float)) return fail(HIM_E_WORKSPACE, "div_scalar: ws too small");
  hipLaunchKernelGGL(div_scalar_bwd1_kernel, dim3(nb), dim3(256), 0, ST, W, sigma, dout, dW, (float*)ws, n,
                     accumulate);
  hipLaunchKernelGGL(div_scalar_bwd2_kernel, dim3(1), dim3(256), 0, ST, (const float*)ws, nb, sigma, dsigma);
  return check_launch("div_scalar_bwd");
}

}  // extern "C"
