// Max-pool over the neighbourhood axis of a grouped tensor, forward + backward.
//
// Stands in for F.max_pool2d(features, kernel_size=[1, nsample]) in
// BasePointSAModule._pool_features (reference mmdet3d/ops/pointnet_modules/
// point_sa_module.py:136-158) and for torch.max(feature, dim=-1) in MiniPointNet
// (reference mmdet3d/models/dense_heads/side_pooling_module.py:361,368).
// x (R, ns) row-major with ns in {4..64, power of two}: each lane loads one float4, the
// ns/4 lanes of a row reduce with DPP row operations (no LDS, no cross-row traffic), so
// every wave instruction is one dense 1 KiB read: pure HBM streaming.
// Ties resolve to the smallest index, as ATen's max_pool2d / max do; the arg-max is kept
// as one byte per row for the backward, which writes grad * [s == argmax].
#include "common.h"

namespace nesie {

// LPR = lanes per row = ns / 4
template <int LPR>
__global__ __launch_bounds__(256) void group_max_fwd_kernel(
    long long rows, const float4 *__restrict__ x, float *__restrict__ out,
    uint8_t *__restrict__ arg, int nt) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;  // one float4 each
  const long long row = t / LPR;
  const int part = (int)(t % LPR);
  const bool live = row < rows;
  const float4 q = live ? (nt ? ld4<true>((const float *)(x + t)) : x[t])
                        : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  float v; int i;
  row_argmax4<LPR>(q, part, v, i);
  if (live && part == 0) { out[row] = v; arg[row] = (uint8_t)i; }
}

template <int LPR>
__global__ __launch_bounds__(256) void group_max_bwd_kernel(
    long long rows, const float *__restrict__ grad_out, const uint8_t *__restrict__ arg,
    float4 *__restrict__ grad_x, int nt) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long row = t / LPR;
  const int part = (int)(t % LPR);
  if (row >= rows) return;
  const float g = grad_out[row];
  const int a = (int)arg[row] - part * 4;
  const float4 r = make_float4(a == 0 ? g : 0.f, a == 1 ? g : 0.f, a == 2 ? g : 0.f,
                               a == 3 ? g : 0.f);
  if (nt) st4<true>((float *)(grad_x + t), r); else grad_x[t] = r;
}

// grad_x[row, argmax[row]] += grad_out[row]: the pooled gradient added into a dense gradient
// that already exists (a tensor that feeds both the max and another consumer), instead of
// materialising a one-hot tensor and adding the two.
__global__ __launch_bounds__(256) void group_max_bwd_add_kernel(
    long long rows, int ns, const float *__restrict__ grad_out, const uint8_t *__restrict__ arg,
    float *__restrict__ grad_x) {
  const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  grad_x[row * ns + arg[row]] += grad_out[row];
}

// ---- scatter-add of grouped gradients through LDS --------------------------------
// grad_points[b, c, idx[b, e]] += grad_out[b, c, e].  A workgroup owns CH channel rows of
// one scene: their n accumulators sit in LDS, the grouped gradient streams through once as
// float4s with the matching int4 of indices, float adds go to LDS (ds_add_f32) instead of
// HBM atomics, and the rows are written back once, dense.  CH is kept small (the launcher
// aims at >= 512 workgroups of 16-32 KB LDS) so that enough loads are in flight to stream at
// HBM rate; the index array is re-read per workgroup from L2.  Sum order is not fixed (as in
// the reference's atomicAdd kernel, group_points_cuda.cu:10-31).
template <int CH>
__global__ __launch_bounds__(256) void group_bwd_lds_kernel(
    int c, int n, int e_total, long long gstride, int vec, const float *__restrict__ grad_out,
    const int *__restrict__ idx, float *__restrict__ grad_points) {
  extern __shared__ float acc[];  // [CH][n]
  const int bi = blockIdx.y;
  const int c0 = blockIdx.x * CH;
  const int cend = c - c0 < CH ? c - c0 : CH;
  for (int i = threadIdx.x; i < cend * n; i += 256) acc[i] = 0.f;
  __syncthreads();
  const int *ix = idx + (size_t)bi * e_total;
  const float *src = grad_out + (size_t)bi * gstride + (size_t)c0 * e_total;
  if (vec) {  // e_total % 4 == 0 and 16-byte aligned bases: rows and index runs are float4s
    for (int e = threadIdx.x * 4; e < e_total; e += 256 * 4) {
      int4 d = *(const int4 *)(ix + e);
      d.x = d.x < 0 ? 0 : (d.x >= n ? n - 1 : d.x);
      d.y = d.y < 0 ? 0 : (d.y >= n ? n - 1 : d.y);
      d.z = d.z < 0 ? 0 : (d.z >= n ? n - 1 : d.z);
      d.w = d.w < 0 ? 0 : (d.w >= n ? n - 1 : d.w);
      float4 g[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i)
        g[i] = i < cend ? *(const float4 *)(src + (size_t)i * e_total + e) : make_float4(0, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        if (i < cend) {
          atomicAdd(&acc[i * n + d.x], g[i].x);
          atomicAdd(&acc[i * n + d.y], g[i].y);
          atomicAdd(&acc[i * n + d.z], g[i].z);
          atomicAdd(&acc[i * n + d.w], g[i].w);
        }
      }
    }
  } else {
    for (int e = threadIdx.x; e < e_total; e += 256) {
      int dst = ix[e];
      dst = dst < 0 ? 0 : (dst >= n ? n - 1 : dst);
      for (int i = 0; i < cend; ++i) atomicAdd(&acc[i * n + dst], src[(size_t)i * e_total + e]);
    }
  }
  __syncthreads();
  float *dst_rows = grad_points + ((size_t)bi * c + c0) * n;
  for (int i = threadIdx.x; i < cend * n; i += 256) dst_rows[i] += acc[i];
}

}  // namespace nesie

using namespace nesie;

static int pool_dims_ok(const char *W, long long rows, int ns) {
  if (rows < 0 || ns <= 0) { set_error("%s: negative size", W); return NESIE_ERR_INVALID_ARG; }
  if (ns < 4 || ns > 64 || (ns & (ns - 1))) {
    set_error("%s: nsample %d (needs a power of two in 4..64)", W, ns);
    return NESIE_ERR_UNSUPPORTED;
  }
  return NESIE_OK;
}

extern "C" int nesie_group_max_pool_forward(long long rows, int nsample, const float *x,
                                            float *out, uint8_t *argmax, void *stream) {
  const char *W = "group_max_pool_forward";
  int st = pool_dims_ok(W, rows, nsample);
  if (st) return st;
  if (rows == 0) return NESIE_OK;
  NESIE_REQUIRE(x && out && argmax && ((uintptr_t)x & 15) == 0, W);
  const int lpr = nsample / 4;
  const long long threads = rows * lpr;
  NESIE_REQUIRE(threads / 256 + 1 < (1ll << 31), W);
  dim3 grid((unsigned)cdiv(threads, 256));
  hipStream_t s = (hipStream_t)stream;
#define L(N) hipLaunchKernelGGL(group_max_fwd_kernel<N>, grid, dim3(256), 0, s, rows, \
                                (const float4 *)x, out, argmax, nt)
  const int nt = stream_nt(rows * nsample * 4, 2) ? 1 : 0;
  if (lpr == 1) L(1); else if (lpr == 2) L(2); else if (lpr == 4) L(4);
  else if (lpr == 8) L(8); else L(16);
#undef L
  return check_launch(W);
}

extern "C" int nesie_group_max_pool_backward(long long rows, int nsample,
                                             const float *grad_out, const uint8_t *argmax,
                                             float *grad_x, void *stream) {
  const char *W = "group_max_pool_backward";
  int st = pool_dims_ok(W, rows, nsample);
  if (st) return st;
  if (rows == 0) return NESIE_OK;
  NESIE_REQUIRE(grad_out && argmax && grad_x && ((uintptr_t)grad_x & 15) == 0, W);
  const int lpr = nsample / 4;
  const long long threads = rows * lpr;
  NESIE_REQUIRE(threads / 256 + 1 < (1ll << 31), W);
  dim3 grid((unsigned)cdiv(threads, 256));
  hipStream_t s = (hipStream_t)stream;
#define L(N) hipLaunchKernelGGL(group_max_bwd_kernel<N>, grid, dim3(256), 0, s, rows, \
                                grad_out, argmax, (float4 *)grad_x, nt)
  const int nt = stream_nt(rows * nsample * 4, 2) ? 1 : 0;
  if (lpr == 1) L(1); else if (lpr == 2) L(2); else if (lpr == 4) L(4);
  else if (lpr == 8) L(8); else L(16);
#undef L
  return check_launch(W);
}

extern "C" int nesie_group_max_pool_backward_add(long long rows, int nsample,
                                                 const float *grad_out, const uint8_t *argmax,
                                                 float *grad_x, void *stream) {
  const char *W = "group_max_pool_backward_add";
  NESIE_REQUIRE(rows >= 0 && nsample >= 1, W);
  if (rows == 0) return NESIE_OK;
  NESIE_REQUIRE(grad_out && argmax && grad_x && rows / 256 + 1 < (1ll << 31), W);
  hipLaunchKernelGGL(group_max_bwd_add_kernel, dim3((unsigned)cdiv(rows, 256)), dim3(256), 0,
                     (hipStream_t)stream, rows, nsample, grad_out, argmax, grad_x);
  return check_launch(W);
}

// Called by nesie_group_points_backward (group_gather.hip) when n is small enough.
namespace nesie {
int launch_group_bwd_lds(int b, int c, int n, long long e_total, long long gstride,
                         const float *grad_out, const int *idx, float *grad_points,
                         hipStream_t s) {
  int ch = (int)((long long)c * b / 512);  // aim at >= 512 workgroups
  if (ch > 16384 / n) ch = 16384 / n;      // n * ch * 4 bytes <= 64 KB
  ch = ch >= 8 ? 8 : ch >= 4 ? 4 : ch >= 2 ? 2 : 1;
  const size_t lds = (size_t)ch * n * sizeof(float);
  const dim3 grid(cdiv(c, ch), b);
  const int e = (int)e_total;
  const int vec = (e & 3) == 0 && (gstride & 3) == 0 &&
                  (((uintptr_t)grad_out | (uintptr_t)idx) & 15) == 0;
#define L(N) hipLaunchKernelGGL(group_bwd_lds_kernel<N>, grid, dim3(256), lds, s, c, n, e, gstride, vec, \
                                grad_out, idx, grad_points)
  if (ch == 8) L(8); else if (ch == 4) L(4); else if (ch == 2) L(2); else L(1);
#undef L
  return check_launch("group_points_backward");
}
}  // namespace nesie
