// Gradient clipping by global norm + AdamW over ONE flat parameter vector.
//
// The reference trains with mmcv's OptimizerHook(grad_clip=dict(max_norm=10, norm_type=2)) and
// torch.optim.AdamW (configs/_base_/schedules).  With all parameters and
// gradients in two flat vectors (nesie_amd/dp.FlatTrainState) both are element-wise passes over
// 2.6 M floats; torch's multi-tensor kernels see a single tensor and cut it into 64 K-element
// chunks = 41 workgroups on a 256-CU chip (0.06 + 0.10 ms).  Here: one pass for the squared norm
// (per-workgroup partials, fixed order), one pass for the update; every workgroup of the second
// pass folds the partials itself (the same order everywhere), so the clip coefficient needs no
// extra launch and no host round trip.
#include "common.h"
#include <math.h>

namespace nesie {

constexpr int OPT_BLOCK = 256, OPT_PARTS = 1024;

__global__ __launch_bounds__(OPT_BLOCK) void sumsq_partials_kernel(long long n, const float *__restrict__ g,
                                                                   float *__restrict__ part,
                                                                   float *__restrict__ step) {
  __shared__ float sh[OPT_BLOCK / 64];
  const long long per = (n + OPT_PARTS - 1) / OPT_PARTS;
  const long long lo = (long long)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  float s = 0.f;
  for (long long i = lo + threadIdx.x; i < hi; i += OPT_BLOCK) s += g[i] * g[i];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    if (blockIdx.x == 0 && step) *step += 1.f;      // the update kernel reads the new count
  }
}

__global__ __launch_bounds__(OPT_BLOCK) void adamw_clip_kernel(
    long long n, float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
    float *__restrict__ v, const float *__restrict__ part, const float *__restrict__ step,
    float lr, float beta1, float beta2, float eps, float wd, float max_norm, float *__restrict__ norm_out,
    const float *__restrict__ hyper) {
  // hyper (optional, device): [lr, weight_decay] read at execution time, so that a captured
  // launch follows a learning-rate schedule without being re-captured
  if (hyper) { lr = hyper[0]; wd = hyper[1]; }
  __shared__ float sh[OPT_BLOCK / 64];
  __shared__ float coef_s;
  // total squared norm: OPT_PARTS partials, 4 per thread, then the block tree (same everywhere)
  float s = 0.f;
  for (int i = threadIdx.x; i < OPT_PARTS; i += OPT_BLOCK) s += part[i];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float total = sqrtf((sh[0] + sh[1]) + (sh[2] + sh[3]));
    float c = max_norm > 0.f ? max_norm / (total + 1e-6f) : 1.f;    // clip_grad_norm_'s coefficient
    coef_s = c < 1.f ? c : 1.f;
    if (blockIdx.x == 0 && norm_out) *norm_out = total;
  }
  __syncthreads();
  const float coef = coef_s;
  const float t = *step;
  const float bc1 = 1.f - powf(beta1, t), bc2 = 1.f - powf(beta2, t);
  const float step_size = lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  const long long per = (n + gridDim.x - 1) / gridDim.x;
  const long long lo = (long long)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  for (long long i = lo + threadIdx.x; i < hi; i += OPT_BLOCK) {
    const float gi = g[i] * coef;
    float pi = p[i];
    pi = pi - pi * (lr * wd);                                         // decoupled weight decay
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * (gi * gi);
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] = pi - step_size * (mi / denom);
    m[i] = mi; v[i] = vi; g[i] = gi;                                  // the clipped gradient stays visible
  }
}

}  // namespace nesie

using namespace nesie;

extern "C" size_t nesie_flat_adamw_workspace_bytes(void) { return OPT_PARTS * sizeof(float); }

static int flat_adamw_impl(const char *W, long long n, float *param, float *grad, float *exp_avg,
                           float *exp_avg_sq, float *step, float lr, float beta1, float beta2,
                           float eps, float weight_decay, const float *hyper, float max_norm,
                           float *grad_norm_out, void *workspace, size_t workspace_bytes,
                           void *stream) {
  NESIE_REQUIRE(n >= 0, W);
  if (n == 0) return NESIE_OK;
  NESIE_REQUIRE(param && grad && exp_avg && exp_avg_sq && step && workspace, W);
  NESIE_REQUIRE(workspace_bytes >= nesie_flat_adamw_workspace_bytes(), W);
  hipStream_t s = (hipStream_t)stream;
  float *part = (float *)workspace;
  hipLaunchKernelGGL(sumsq_partials_kernel, dim3(OPT_PARTS), dim3(OPT_BLOCK), 0, s, n, grad, part, step);
  const int blocks = (int)((n + 4095) / 4096 < 1024 ? (n + 4095) / 4096 : 1024);
  hipLaunchKernelGGL(adamw_clip_kernel, dim3(blocks), dim3(OPT_BLOCK), 0, s, n, param, grad, exp_avg,
                     exp_avg_sq, part, step, lr, beta1, beta2, eps, weight_decay, max_norm,
                     grad_norm_out, hyper);
  return check_launch(W);
}

extern "C" int nesie_flat_adamw_step(long long n, float *param, float *grad, float *exp_avg,
                                     float *exp_avg_sq, float *step, float lr, float beta1,
                                     float beta2, float eps, float weight_decay, float max_norm,
                                     float *grad_norm_out, void *workspace, size_t workspace_bytes,
                                     void *stream) {
  return flat_adamw_impl("flat_adamw_step", n, param, grad, exp_avg, exp_avg_sq, step, lr, beta1,
                         beta2, eps, weight_decay, nullptr, max_norm, grad_norm_out, workspace,
                         workspace_bytes, stream);
}

extern "C" int nesie_flat_adamw_step_dev(long long n, float *param, float *grad, float *exp_avg,
                                         float *exp_avg_sq, float *step, const float *hyper,
                                         float beta1, float beta2, float eps, float max_norm,
                                         float *grad_norm_out, void *workspace,
                                         size_t workspace_bytes, void *stream) {
  const char *W = "flat_adamw_step_dev";
  NESIE_REQUIRE(n == 0 || hyper, W);
  return flat_adamw_impl(W, n, param, grad, exp_avg, exp_avg_sq, step, 0.f, beta1, beta2, eps, 0.f,
                         hyper, max_norm, grad_norm_out, workspace, workspace_bytes, stream);
}
