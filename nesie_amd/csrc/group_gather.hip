// group_points / gather_points forward + backward for gfx950.
//
// Replaces group_points_kernel / group_points_grad_kernel
// (reference mmdet3d/ops/group_points/src/group_points_cuda.cu:56-80, :10-31) and
// gather_points_kernel / gather_points_grad_kernel
// (reference mmdet3d/ops/gather_points/src/gather_points_cuda.cu:8-26, :51-70).
// gather is group with nsample == 1, so one kernel pair serves both.
//
// Forward is pure data movement (HBM/L2 bound): a thread owns one output
// column e = (p, s), reads idx[e] ONCE and walks CH_PER_THREAD channel rows, so
// the index traffic is amortised over the channels and every store is a dense
// 256-byte row segment per wave.  Backward: when a scene's n accumulators for a
// few channel rows fit LDS (pool.hip: group_bwd_lds_kernel) the adds go to LDS and
// each row is written once; otherwise the same walk with float atomics to HBM
// (the reference does the latter; sum order is not deterministic either way).
#include "common.h"

namespace nesie {

constexpr int GG_BLOCK = 256;
constexpr int GG_CH = 8;  // channels walked per thread

// points (B,C,N), idx (B,E) -> out (B,C,E)        E = npoints * nsample
__global__ __launch_bounds__(GG_BLOCK) void group_fwd_kernel(
    int c, int n, int e_total, long long gstride, const float *__restrict__ points,
    const int *__restrict__ idx, float *__restrict__ out) {
  const int e = blockIdx.x * GG_BLOCK + threadIdx.x;
  const int c0 = blockIdx.y * GG_CH;
  const int bi = blockIdx.z;
  if (e >= e_total) return;
  int src = idx[(size_t)bi * e_total + e];
  src = src < 0 ? 0 : (src >= n ? n - 1 : src);  // never fault on a bad index
  const float *p = points + ((size_t)bi * c + c0) * n + src;
  float *o = out + (size_t)bi * gstride + (size_t)c0 * e_total + e;  // gstride: scene pitch
  const int cend = c - c0 < GG_CH ? c - c0 : GG_CH;
  float v[GG_CH];
#pragma unroll
  for (int i = 0; i < GG_CH; ++i)
    if (i < cend) v[i] = p[(size_t)i * n];
#pragma unroll
  for (int i = 0; i < GG_CH; ++i)
    if (i < cend) o[(size_t)i * e_total] = v[i];
}

// grad_out (B,C,E), idx (B,E) -> grad_points (B,C,N) += ...
__global__ __launch_bounds__(GG_BLOCK) void group_bwd_kernel(
    int c, int n, int e_total, long long gstride, const float *__restrict__ grad_out,
    const int *__restrict__ idx, float *__restrict__ grad_points) {
  const int e = blockIdx.x * GG_BLOCK + threadIdx.x;
  const int c0 = blockIdx.y * GG_CH;
  const int bi = blockIdx.z;
  if (e >= e_total) return;
  int dst = idx[(size_t)bi * e_total + e];
  dst = dst < 0 ? 0 : (dst >= n ? n - 1 : dst);  // never fault on a bad index
  const float *g = grad_out + (size_t)bi * gstride + (size_t)c0 * e_total + e;
  float *gp = grad_points + ((size_t)bi * c + c0) * n + dst;
  const int cend = c - c0 < GG_CH ? c - c0 : GG_CH;
  float v[GG_CH];
#pragma unroll
  for (int i = 0; i < GG_CH; ++i)
    if (i < cend) v[i] = g[(size_t)i * e_total];
#pragma unroll
  for (int i = 0; i < GG_CH; ++i)
    if (i < cend) atomicAdd(gp + (size_t)i * n, v[i]);
}

int launch_group_bwd_lds(int b, int c, int n, long long e_total, long long gstride,
                         const float *grad_out, const int *idx, float *grad_points,
                         hipStream_t s);

// gstride = elements between scenes of the GROUPED tensor (out of the forward, grad_out of the
// backward); c * e_total for a dense one, larger when it is a channel slice of a wider tensor
static int launch_group(bool fwd, const char *W, int b, int c, int n, long long e_total,
                        const float *a, const int *idx, float *o, void *stream,
                        long long gstride = -1) {
  if (gstride < 0) gstride = (long long)c * e_total;
  NESIE_REQUIRE(b >= 0 && c >= 0 && n >= 0 && e_total >= 0, W);
  if (b == 0 || c == 0 || e_total == 0) return NESIE_OK;
  NESIE_REQUIRE(n >= 1 && a && idx && o, W);
  NESIE_REQUIRE(e_total < (1ll << 31) && b <= 65535 && cdiv(c, GG_CH) <= 65535, W);
  if (!fwd && n <= 16384 && e_total >= 4 * (long long)n)
    return launch_group_bwd_lds(b, c, n, e_total, gstride, a, idx, o, (hipStream_t)stream);
  dim3 grid(cdiv(e_total, GG_BLOCK), cdiv(c, GG_CH), b);
  if (fwd)
    hipLaunchKernelGGL(group_fwd_kernel, grid, dim3(GG_BLOCK), 0, (hipStream_t)stream, c,
                       n, (int)e_total, gstride, a, idx, o);
  else
    hipLaunchKernelGGL(group_bwd_kernel, grid, dim3(GG_BLOCK), 0, (hipStream_t)stream, c,
                       n, (int)e_total, gstride, a, idx, o);
  return check_launch(W);
}

// channels 0..2 of QueryAndGroup's output: (xyz[idx] - centre) (/ radius), straight from the
// (B,N,3) point array (the reference transposes it, groups it, subtracts, divides and
// concatenates: group_points.py:100-118)
__global__ __launch_bounds__(GG_BLOCK) void group_xyz_kernel(
    int n, int npoints, int nsample, long long gstride, float radius,
    const float *__restrict__ xyz, const float *__restrict__ centres,
    const int *__restrict__ idx, float *__restrict__ out) {
  const int e = blockIdx.x * GG_BLOCK + threadIdx.x;
  const int bi = blockIdx.y;
  const int e_total = npoints * nsample;
  if (e >= e_total) return;
  int src = idx[(size_t)bi * e_total + e];
  src = src < 0 ? 0 : (src >= n ? n - 1 : src);
  const float *p = xyz + ((size_t)bi * n + src) * 3;
  const float *cc = centres + ((size_t)bi * npoints + e / nsample) * 3;
  float *o = out + (size_t)bi * gstride + e;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float v = p[d] - cc[d];
    if (radius > 0.f) v = v / radius;
    o[(size_t)d * e_total] = v;
  }
}

}  // namespace nesie

using namespace nesie;

extern "C" int nesie_query_and_group_forward(int b, int c, int n, int npoints, int nsample,
                                             const float *xyz, const float *centres,
                                             const float *features, const int *idx,
                                             float radius, float *out, void *stream) {
  const char *W = "query_and_group_forward";
  NESIE_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0, W);
  const long long e_total = (long long)npoints * nsample;
  if (b == 0 || e_total == 0) return NESIE_OK;
  NESIE_REQUIRE(n >= 1 && xyz && centres && idx && out && (c == 0 || features), W);
  NESIE_REQUIRE(e_total < (1ll << 31) && b <= 65535, W);
  const long long gstride = (long long)(3 + c) * e_total;
  hipLaunchKernelGGL(group_xyz_kernel, dim3(cdiv(e_total, GG_BLOCK), b), dim3(GG_BLOCK), 0,
                     (hipStream_t)stream, n, npoints, nsample, gstride, radius, xyz, centres, idx,
                     out);
  int st = check_launch(W);
  if (st || c == 0) return st;
  return launch_group(true, W, b, c, n, e_total, features, idx, out + 3 * e_total, stream,
                      gstride);
}

extern "C" int nesie_query_and_group_backward(int b, int c, int n, int npoints, int nsample,
                                              const float *grad_out, const int *idx,
                                              float *grad_features, void *stream) {
  const char *W = "query_and_group_backward";
  NESIE_REQUIRE(npoints >= 0 && nsample >= 0 && c >= 1, W);
  const long long e_total = (long long)npoints * nsample;
  return launch_group(false, W, b, c, n, e_total, grad_out ? grad_out + 3 * e_total : nullptr,
                      idx, grad_features, stream, (long long)(3 + c) * e_total);
}

extern "C" int nesie_group_points_forward(int b, int c, int n, int npoints, int nsample,
                                          const float *points, const int *idx,
                                          float *out, void *stream) {
  NESIE_REQUIRE(npoints >= 0 && nsample >= 0, "group_points_forward");
  return launch_group(true, "group_points_forward", b, c, n,
                      (long long)npoints * nsample, points, idx, out, stream);
}

extern "C" int nesie_group_points_backward(int b, int c, int n, int npoints, int nsample,
                                           const float *grad_out, const int *idx,
                                           float *grad_points, void *stream) {
  NESIE_REQUIRE(npoints >= 0 && nsample >= 0, "group_points_backward");
  return launch_group(false, "group_points_backward", b, c, n,
                      (long long)npoints * nsample, grad_out, idx, grad_points, stream);
}

extern "C" int nesie_gather_points_wrapper(int b, int c, int n, int npoints,
                                           const float *points, const int *idx,
                                           float *out, void *stream) {
  NESIE_REQUIRE(npoints >= 0, "gather_points_wrapper");
  return launch_group(true, "gather_points_wrapper", b, c, n, npoints, points, idx, out,
                      stream);
}

extern "C" int nesie_gather_points_grad_wrapper(int b, int c, int n, int npoints,
                                                const float *grad_out, const int *idx,
                                                float *grad_points, void *stream) {
  NESIE_REQUIRE(npoints >= 0, "gather_points_grad_wrapper");
  return launch_group(false, "gather_points_grad_wrapper", b, c, n, npoints, grad_out,
                      idx, grad_points, stream);
}
