// The tail of VoteModule.forward behind its last convolution (reference
// mmdet3d/models/model_utils/vote_module.py:106-147, vote_per_seed = 1, with_res_feat, no
// vote_xyz_range): raw (B, 3 + C, N) = [offset (3), residual features (C)] ->
//     vote_points[b][n][:] = seed_points[b][n][:] + raw[b][0:3][n]
//     f[c]                 = seed_feats[b][c][n] + raw[b][3 + c][n]
//     vote_feats[b][:][n]  = f / ||f||_2                       (norm_feats; 1 when off)
// The reference runs it as add / permute / add / norm / div (five ATen launches) and autograd
// runs a dozen more backwards through them; here one kernel per direction, 64 votes per workgroup
// with the C channels walked twice by four waves (they sit in L2: 8 MB at 8 x 1024 x 256).
#include "common.h"

namespace nesie {

constexpr int VF_VOTES = 64, VF_PARTS = 4, VF_BLOCK = VF_VOTES * VF_PARTS;   // 64 votes x 4 channel quarters

// a workgroup = 64 consecutive votes (lane = vote: dense 256-byte rows) x 4 waves, wave w walking
// channels w, w + 4, ...; the four partial sums of a vote meet in LDS
__global__ __launch_bounds__(VF_BLOCK) void vote_finish_fwd_kernel(
    int c, int n, int normalise, const float *__restrict__ raw, const float *__restrict__ seed_points,
    const float *__restrict__ seed_feats, float *__restrict__ vote_points,
    float *__restrict__ vote_feats, float *__restrict__ inv_norm) {
  __shared__ float part[VF_PARTS][VF_VOTES];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * VF_VOTES + lane, bi = blockIdx.y;
  const bool live = i < n;
  const int ii = live ? i : n - 1;
  const float *r = raw + (size_t)bi * (3 + c) * n + ii;
  if (w == 0 && live) {
    const float *sp = seed_points + ((size_t)bi * n + i) * 3;
    float *vp = vote_points + ((size_t)bi * n + i) * 3;
#pragma unroll
    for (int d = 0; d < 3; ++d) vp[d] = sp[d] + r[(size_t)d * n];
  }
  const float *sf = seed_feats + (size_t)bi * c * n + ii;
  float *vf = vote_feats + (size_t)bi * c * n + ii;
  const float *rr = r + (size_t)3 * n;
  float ss = 0.f;
#pragma unroll 8
  for (int k = w; k < c; k += VF_PARTS) {
    const float f = sf[(size_t)k * n] + rr[(size_t)k * n];
    ss += f * f;
  }
  part[w][lane] = ss;
  __syncthreads();
  ss = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
  const float inv = normalise ? 1.f / sqrtf(ss) : 1.f;
  if (w == 0 && live) inv_norm[(size_t)bi * n + i] = inv;
  if (!live) return;
#pragma unroll 8
  for (int k = w; k < c; k += VF_PARTS) vf[(size_t)k * n] = (sf[(size_t)k * n] + rr[(size_t)k * n]) * inv;
}

// g_feats (B, C, N) / g_points (B, N, 3) (either may be NULL) -> d_raw (B, 3 + C, N): rows 0..2 =
// g_points transposed, rows 3.. = d_f = (g - vhat (vhat . g)) / ||f|| (= g without normalisation);
// the gradient of seed_feats IS rows 3.. (the caller hands out a view), that of seed_points is
// g_points itself.
__global__ __launch_bounds__(VF_BLOCK) void vote_finish_bwd_kernel(
    int c, int n, int normalise, const float *__restrict__ g_feats, const float *__restrict__ g_points,
    const float *__restrict__ vote_feats, const float *__restrict__ inv_norm, float *__restrict__ d_raw) {
  __shared__ float part[VF_PARTS][VF_VOTES];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * VF_VOTES + lane, bi = blockIdx.y;
  const bool live = i < n;
  const int ii = live ? i : n - 1;
  float *dr = d_raw + (size_t)bi * (3 + c) * n + ii;
  if (w == 0 && live) {
#pragma unroll
    for (int d = 0; d < 3; ++d) dr[(size_t)d * n] = g_points ? g_points[((size_t)bi * n + i) * 3 + d] : 0.f;
  }
  float *df = dr + (size_t)3 * n;
  if (!g_feats) {                                          // (uniform)
    if (live)
      for (int k = w; k < c; k += VF_PARTS) df[(size_t)k * n] = 0.f;
    return;
  }
  const float *g = g_feats + (size_t)bi * c * n + ii;
  const float *v = vote_feats + (size_t)bi * c * n + ii;
  const float inv = inv_norm[(size_t)bi * n + ii];
  float dot = 0.f;
  if (normalise) {                                         // (uniform)
#pragma unroll 8
    for (int k = w; k < c; k += VF_PARTS) dot += v[(size_t)k * n] * g[(size_t)k * n];
    part[w][lane] = dot;
    __syncthreads();
    dot = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
  }
  if (!live) return;
#pragma unroll 8
  for (int k = w; k < c; k += VF_PARTS) df[(size_t)k * n] = (g[(size_t)k * n] - v[(size_t)k * n] * dot) * inv;
}

}  // namespace nesie

using namespace nesie;

extern "C" int nesie_vote_finish_forward(int b, int c, int n, int normalise, const float *raw,
                                         const float *seed_points, const float *seed_feats,
                                         float *vote_points, float *vote_feats, float *inv_norm,
                                         void *stream) {
  const char *W = "vote_finish_forward";
  NESIE_REQUIRE(b >= 0 && c >= 1 && n >= 0 && b <= 65535, W);
  if (b == 0 || n == 0) return NESIE_OK;
  NESIE_REQUIRE(raw && seed_points && seed_feats && vote_points && vote_feats && inv_norm, W);
  hipLaunchKernelGGL(vote_finish_fwd_kernel, dim3(cdiv(n, VF_VOTES), b), dim3(VF_BLOCK), 0,
                     (hipStream_t)stream, c, n, normalise, raw, seed_points, seed_feats, vote_points,
                     vote_feats, inv_norm);
  return check_launch(W);
}

extern "C" int nesie_vote_finish_backward(int b, int c, int n, int normalise, const float *g_feats,
                                          const float *g_points, const float *vote_feats,
                                          const float *inv_norm, float *d_raw, void *stream) {
  const char *W = "vote_finish_backward";
  NESIE_REQUIRE(b >= 0 && c >= 1 && n >= 0 && b <= 65535, W);
  if (b == 0 || n == 0) return NESIE_OK;
  NESIE_REQUIRE(vote_feats && inv_norm && d_raw, W);
  hipLaunchKernelGGL(vote_finish_bwd_kernel, dim3(cdiv(n, VF_VOTES), b), dim3(VF_BLOCK), 0,
                     (hipStream_t)stream, c, n, normalise, g_feats, g_points, vote_feats, inv_norm, d_raw);
  return check_launch(W);
}
