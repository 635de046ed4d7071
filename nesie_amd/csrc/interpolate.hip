// three_nn / three_interpolate (+grad) for gfx950.
//
// Replaces three_nn_kernel (reference mmdet3d/ops/interpolate/src/three_nn_cuda.cu:11-65),
// three_interpolate_kernel and three_interpolate_grad_kernel
// (reference .../three_interpolate_cuda.cu:11-35, :61-84).
//
// three_nn: the known set (m <= a few thousand points) is staged through LDS in
// tiles of (x, y, z, -) float4s, so the inner scan reads one LDS broadcast per point instead of
// global memory; each thread owns one query and keeps its three best
// (distance, index) pairs in registers.  The reference keeps the bests in
// double but compares them with a float candidate, which orders exactly like
// float compares with a +inf start (1e40 rounds to +inf on output), so floats
// are used here.  Ties keep the earlier index in the better slot (strict <).
#include "common.h"
#include <math.h>

namespace nesie {

constexpr int NN_BLOCK = 256;
constexpr int NN_TILE = 1024;  // known points per LDS tile (12 KB)

__global__ __launch_bounds__(NN_BLOCK) void three_nn_kernel(
    int n, int m, const float *__restrict__ unknown, const float *__restrict__ known,
    float *__restrict__ dist2, int *__restrict__ idx) {
  __shared__ float4 kk[NN_TILE];  // (x, y, z, -): one ds_read_b128 broadcast per known point
  const int bi = blockIdx.y;
  const int q = blockIdx.x * NN_BLOCK + threadIdx.x;
  const bool live = q < n;
  unknown += (size_t)bi * n * 3;
  known += (size_t)bi * m * 3;
  const int qq = live ? q : n - 1;
  const float ux = unknown[qq * 3 + 0], uy = unknown[qq * 3 + 1], uz = unknown[qq * 3 + 2];
  float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
  int i1 = 0, i2 = 0, i3 = 0;
  for (int t0 = 0; t0 < m; t0 += NN_TILE) {
    const int tn = m - t0 < NN_TILE ? m - t0 : NN_TILE;
    __syncthreads();
    for (int i = threadIdx.x; i < tn; i += NN_BLOCK)
      kk[i] = make_float4(known[(t0 + i) * 3 + 0], known[(t0 + i) * 3 + 1],
                          known[(t0 + i) * 3 + 2], 0.f);
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < tn; ++i) {
      const float4 kp = kk[i];
      const float d = sqdist_nofma(ux - kp.x, uy - kp.y, uz - kp.z);
      if (d < b3) {  // rare once the three bests have settled: one compare on the common path
        const int k = t0 + i;
        if (d < b1) {
          b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k;
        } else if (d < b2) {
          b3 = b2; i3 = i2; b2 = d; i2 = k;
        } else {
          b3 = d; i3 = k;
        }
      }
    }
  }
  if (live) {
    float *od = dist2 + ((size_t)bi * n + q) * 3;
    int *oi = idx + ((size_t)bi * n + q) * 3;
    od[0] = b1; od[1] = b2; od[2] = b3;
    oi[0] = i1; oi[1] = i2; oi[2] = i3;
  }
}

constexpr int TI_BLOCK = 256;
constexpr int TI_CH = 8;

// points (B,C,M), idx/weight (B,N,3) -> out (B,C,N)
__global__ __launch_bounds__(TI_BLOCK) void three_interpolate_kernel(
    int c, int m, int n, const float *__restrict__ points, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ out) {
  const int p = blockIdx.x * TI_BLOCK + threadIdx.x;
  const int c0 = blockIdx.y * TI_CH;
  const int bi = blockIdx.z;
  if (p >= n) return;
  const int *ix = idx + ((size_t)bi * n + p) * 3;
  const float *w = weight + ((size_t)bi * n + p) * 3;
  int j0 = ix[0], j1 = ix[1], j2 = ix[2];
  j0 = j0 < 0 ? 0 : (j0 >= m ? m - 1 : j0);
  j1 = j1 < 0 ? 0 : (j1 >= m ? m - 1 : j1);
  j2 = j2 < 0 ? 0 : (j2 >= m ? m - 1 : j2);
  const float w0 = w[0], w1 = w[1], w2 = w[2];
  const int cend = c - c0 < TI_CH ? c - c0 : TI_CH;
#pragma unroll
  for (int i = 0; i < TI_CH; ++i) {
    if (i < cend) {
      const float *src = points + ((size_t)bi * c + c0 + i) * m;
      // products rounded one by one, summed left to right (.cu:33-34)
      float r = __fadd_rn(__fadd_rn(__fmul_rn(w0, src[j0]), __fmul_rn(w1, src[j1])),
                          __fmul_rn(w2, src[j2]));
      out[((size_t)bi * c + c0 + i) * n + p] = r;
    }
  }
}

// Same blend, written straight into the layout the side-aware quality head consumes:
// query p = (k, s, g) of n = K * segs * seg_len (proposal, face, grid point) goes to
// out[b, s, c_offset + ch, k * seg_len + g] of out (B, segs, c_total, K * seg_len), i.e. one
// (c_total, K, seg_len) block per face with c_offset leading channels left for the caller
// (relative xyz).  Replaces interpolate -> view -> cat -> split -> contiguous
// (side_pooling_module.py:226-243, 304-313).  The features come POINT-major, points_t
// (B, M, C), so the three taps of a query are three dense 256-byte rows per 64 channels
// (lane = channel) instead of 3 * 64 scattered words; a 64 x 64 tile is turned through LDS
// and leaves as dense rows along the query axis (lane = query).
constexpr int TS_Q = 64;  // queries per workgroup (one tile side; per_seg % 64 == 0)

__global__ __launch_bounds__(256) void three_interpolate_segmented_kernel(
    int c, int m, int n, int segs, int seg_len, int c_total, int c_offset,
    const float *__restrict__ points_t, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ out) {
  __shared__ float tile[64][TS_Q + 1];
  __shared__ int sj[TS_Q][3];
  __shared__ float sw[TS_Q][3];
  const int bi = blockIdx.y;
  const int q0 = blockIdx.x * TS_Q;  // output order: s * per_seg + k * seg_len + g
  const int per_seg = n / segs;
  const int sg = q0 / per_seg, r0 = q0 - sg * per_seg;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x < TS_Q) {
    const int r = r0 + threadIdx.x;
    const int k = r / seg_len, g = r - k * seg_len;
    const size_t p = (size_t)bi * n + (size_t)(k * segs + sg) * seg_len + g;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      int j = idx[p * 3 + t];
      sj[threadIdx.x][t] = j < 0 ? 0 : (j >= m ? m - 1 : j);
      sw[threadIdx.x][t] = weight[p * 3 + t];
    }
  }
  __syncthreads();
  const float *feat = points_t + (size_t)bi * m * c;
  float *dst = out + (((size_t)bi * segs + sg) * c_total + c_offset) * per_seg + r0;
  for (int c0 = 0; c0 < c; c0 += 64) {
    // wave wv blends queries wv*16 .. wv*16+15 for channels c0 .. c0+63 (lane = channel)
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int qi = wv * 16 + i;
      const float *f = feat + c0 + lane;
      const float a0 = f[(size_t)sj[qi][0] * c], a1 = f[(size_t)sj[qi][1] * c],
                  a2 = f[(size_t)sj[qi][2] * c];
      // products rounded one by one, summed left to right (three_interpolate_cuda.cu:33-34)
      tile[lane][qi] = __fadd_rn(__fadd_rn(__fmul_rn(sw[qi][0], a0), __fmul_rn(sw[qi][1], a1)),
                                 __fmul_rn(sw[qi][2], a2));
    }
    __syncthreads();
    // wave wv stores channels c0 + wv*16 .. +15 (lane = query): dense 256-byte rows
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int ch = wv * 16 + i;
      dst[(size_t)(c0 + ch) * per_seg + lane] = tile[ch][lane];
    }
    __syncthreads();
  }
}

// Any shape (c % 64 != 0 or per_seg % 64 != 0): one thread per output-order query.
__global__ __launch_bounds__(TI_BLOCK) void three_interpolate_segmented_generic_kernel(
    int c, int m, int n, int segs, int seg_len, int c_total, int c_offset,
    const float *__restrict__ points_t, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ out) {
  const int q = blockIdx.x * TI_BLOCK + threadIdx.x;
  const int c0 = blockIdx.y * TI_CH;
  const int bi = blockIdx.z;
  if (q >= n) return;
  const int per_seg = n / segs;
  const int sg = q / per_seg, r = q - sg * per_seg;
  const int k = r / seg_len, g = r - k * seg_len;
  const int p = (k * segs + sg) * seg_len + g;
  const int *ix = idx + ((size_t)bi * n + p) * 3;
  const float *w = weight + ((size_t)bi * n + p) * 3;
  int j0 = ix[0], j1 = ix[1], j2 = ix[2];
  j0 = j0 < 0 ? 0 : (j0 >= m ? m - 1 : j0);
  j1 = j1 < 0 ? 0 : (j1 >= m ? m - 1 : j1);
  j2 = j2 < 0 ? 0 : (j2 >= m ? m - 1 : j2);
  const float w0 = w[0], w1 = w[1], w2 = w[2];
  const int cend = c - c0 < TI_CH ? c - c0 : TI_CH;
  const float *feat = points_t + (size_t)bi * m * c + c0;
  float *dst = out + (((size_t)bi * segs + sg) * c_total + c_offset + c0) * per_seg + r;
#pragma unroll
  for (int i = 0; i < TI_CH; ++i) {
    if (i < cend) {
      dst[(size_t)i * per_seg] = __fadd_rn(
          __fadd_rn(__fmul_rn(w0, feat[(size_t)j0 * c + i]), __fmul_rn(w1, feat[(size_t)j1 * c + i])),
          __fmul_rn(w2, feat[(size_t)j2 * c + i]));
    }
  }
}

// grad_out (B,C,N) -> grad_points (B,C,M) += w * g   (3 float atomics, .cu:81-83)
__global__ __launch_bounds__(TI_BLOCK) void three_interpolate_grad_kernel(
    int c, int n, int m, const float *__restrict__ grad_out, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ grad_points) {
  const int p = blockIdx.x * TI_BLOCK + threadIdx.x;
  const int c0 = blockIdx.y * TI_CH;
  const int bi = blockIdx.z;
  if (p >= n) return;
  const int *ix = idx + ((size_t)bi * n + p) * 3;
  const float *w = weight + ((size_t)bi * n + p) * 3;
  int j0 = ix[0], j1 = ix[1], j2 = ix[2];
  j0 = j0 < 0 ? 0 : (j0 >= m ? m - 1 : j0);
  j1 = j1 < 0 ? 0 : (j1 >= m ? m - 1 : j1);
  j2 = j2 < 0 ? 0 : (j2 >= m ? m - 1 : j2);
  const float w0 = w[0], w1 = w[1], w2 = w[2];
  const int cend = c - c0 < TI_CH ? c - c0 : TI_CH;
#pragma unroll
  for (int i = 0; i < TI_CH; ++i) {
    if (i < cend) {
      const float g = grad_out[((size_t)bi * c + c0 + i) * n + p];
      float *dst = grad_points + ((size_t)bi * c + c0 + i) * m;
      atomicAdd(dst + j0, __fmul_rn(g, w0));
      atomicAdd(dst + j1, __fmul_rn(g, w1));
      atomicAdd(dst + j2, __fmul_rn(g, w2));
    }
  }
}

}  // namespace nesie

using namespace nesie;

extern "C" int nesie_three_nn_wrapper(int b, int n, int m, const float *unknown,
                                      const float *known, float *dist2, int *idx,
                                      void *stream) {
  const char *W = "three_nn_wrapper";
  NESIE_REQUIRE(b >= 0 && n >= 0 && m >= 0, W);
  if (b == 0 || n == 0) return NESIE_OK;
  NESIE_REQUIRE(unknown && dist2 && idx && (m == 0 || known), W);
  NESIE_REQUIRE(b <= 65535 && (long long)n * 3 < (1ll << 31) && (long long)m * 3 < (1ll << 31), W);
  hipLaunchKernelGGL(three_nn_kernel, dim3(cdiv(n, NN_BLOCK), b), dim3(NN_BLOCK), 0,
                     (hipStream_t)stream, n, m, unknown, known, dist2, idx);
  return check_launch(W);
}

extern "C" int nesie_three_interpolate_wrapper(int b, int c, int m, int n,
                                               const float *points, const int *idx,
                                               const float *weight, float *out,
                                               void *stream) {
  const char *W = "three_interpolate_wrapper";
  NESIE_REQUIRE(b >= 0 && c >= 0 && m >= 0 && n >= 0, W);
  if (b == 0 || c == 0 || n == 0) return NESIE_OK;
  NESIE_REQUIRE(m >= 1 && points && idx && weight && out, W);
  NESIE_REQUIRE(b <= 65535 && cdiv(c, TI_CH) <= 65535, W);
  hipLaunchKernelGGL(three_interpolate_kernel, dim3(cdiv(n, TI_BLOCK), cdiv(c, TI_CH), b),
                     dim3(TI_BLOCK), 0, (hipStream_t)stream, c, m, n, points, idx, weight,
                     out);
  return check_launch(W);
}

extern "C" int nesie_three_interpolate_segmented(int b, int c, int m, int n,
                                                 const float *points_t, const int *idx,
                                                 const float *weight, float *out, int segs,
                                                 int seg_len, int c_total, int c_offset,
                                                 void *stream) {
  const char *W = "three_interpolate_segmented";
  NESIE_REQUIRE(b >= 0 && c >= 0 && m >= 0 && n >= 0 && segs >= 1 && seg_len >= 1, W);
  NESIE_REQUIRE(c_offset >= 0 && c_offset + c <= c_total && n % (segs * seg_len) == 0, W);
  if (b == 0 || c == 0 || n == 0) return NESIE_OK;
  NESIE_REQUIRE(m >= 1 && points_t && idx && weight && out, W);
  NESIE_REQUIRE(b <= 65535 && cdiv(c, TI_CH) <= 65535, W);
  if (c % 64 == 0 && (n / segs) % TS_Q == 0)
    hipLaunchKernelGGL(three_interpolate_segmented_kernel, dim3(n / TS_Q, b), dim3(256), 0,
                       (hipStream_t)stream, c, m, n, segs, seg_len, c_total, c_offset, points_t,
                       idx, weight, out);
  else
    hipLaunchKernelGGL(three_interpolate_segmented_generic_kernel,
                       dim3(cdiv(n, TI_BLOCK), cdiv(c, TI_CH), b), dim3(TI_BLOCK), 0,
                       (hipStream_t)stream, c, m, n, segs, seg_len, c_total, c_offset, points_t,
                       idx, weight, out);
  return check_launch(W);
}

extern "C" int nesie_three_interpolate_grad_wrapper(int b, int c, int n, int m,
                                                    const float *grad_out,
                                                    const int *idx, const float *weight,
                                                    float *grad_points, void *stream) {
  const char *W = "three_interpolate_grad_wrapper";
  NESIE_REQUIRE(b >= 0 && c >= 0 && m >= 0 && n >= 0, W);
  if (b == 0 || c == 0 || n == 0) return NESIE_OK;
  NESIE_REQUIRE(m >= 1 && grad_out && idx && weight && grad_points, W);
  NESIE_REQUIRE(b <= 65535 && cdiv(c, TI_CH) <= 65535, W);
  hipLaunchKernelGGL(three_interpolate_grad_kernel,
                     dim3(cdiv(n, TI_BLOCK), cdiv(c, TI_CH), b), dim3(TI_BLOCK), 0,
                     (hipStream_t)stream, c, n, m, grad_out, idx, weight, grad_points);
  return check_launch(W);
}
