// sort_vertices for gfx950 (rotated-IoU polygon vertex ordering).
//
// Replaces sort_vertices_kernel
// (reference mmdet3d/ops/rotated_iou/cuda_op/sort_vert_kernel.cu:42-134) with the
// comparison of compare_vertices (:15-40) kept term for term.  One thread owns
// one box pair: its 24 candidate vertices and mask bits are pulled into
// registers once (the reference re-reads global memory inside the O(nv * 24)
// selection loops), and the 9 output indices are written with one pass.
// The launch goes to the caller's stream (the reference ignores the stream it
// fetched and uses the legacy default stream, :137-138).
//
// Where the reference is undefined -- compare_vertices falls off its end when
// y1*y2 == 0 in some branches, `pad` is uninitialised when no intersection slot
// is free -- this kernel returns false / uses M-1, as oracle/nesie_oracle.c does.
#include "common.h"

namespace nesie {

constexpr int SV_MAXV = 24;
constexpr int SV_NIDX = 9;
constexpr int SV_OFF = 8;
#define SV_EPS 1e-8  // double, as in the reference

__device__ __forceinline__ bool sv_compare(float x1, float y1, float x2, float y2) {
  if (fabsf(x1 - x2) < SV_EPS && fabsf(y2 - y1) < SV_EPS) return false;
  if (y1 > 0 && y2 < 0) return true;
  if (y1 < 0 && y2 > 0) return false;
  // float sum, then + (double)1e-8, rounded back to float -- no contraction
  float n1 = (float)((double)__fadd_rn(__fmul_rn(x1, x1), __fmul_rn(y1, y1)) + SV_EPS);
  float n2 = (float)((double)__fadd_rn(__fmul_rn(x2, x2), __fmul_rn(y2, y2)) + SV_EPS);
  float a = __fdiv_rn(__fmul_rn(fabsf(x1), x1), n1);
  float c = __fdiv_rn(__fmul_rn(fabsf(x2), x2), n2);
  float diff = __fsub_rn(a, c);
  if (y1 > 0 && y2 > 0) return diff > SV_EPS;
  if (y1 < 0 && y2 < 0) return diff < SV_EPS;
  return false;
}

__global__ __launch_bounds__(256) void sort_vertices_kernel(
    long long total, int m, const float *__restrict__ vertices,
    const uint8_t *__restrict__ mask, const int *__restrict__ num_valid,
    int *__restrict__ idx) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const float *v = vertices + (size_t)i * m * 2;
  const uint8_t *mk = mask + (size_t)i * m;
  float vx[SV_MAXV], vy[SV_MAXV];
  unsigned mbits = 0;
#pragma unroll
  for (int k = 0; k < SV_MAXV; ++k) {
    if (k < m) {
      vx[k] = v[k * 2 + 0];
      vy[k] = v[k * 2 + 1];
      mbits |= (mk[k] ? 1u : 0u) << k;
    } else {
      vx[k] = 0.f; vy[k] = 0.f;
    }
  }
  const int nv = num_valid[i];
  int pad = m - 1;
  {
    unsigned free_slots = ~mbits & (((m >= 32) ? 0xFFFFFFFFu : ((1u << m) - 1u)) & ~0xFFu);
    if (free_slots) pad = __ffs(free_slots) - 1;
  }
  int o[SV_NIDX];
  if (nv < 3) {
#pragma unroll
    for (int j = 0; j < SV_NIDX; ++j) o[j] = pad;
  } else {
    float px = 0.f, py = 0.f;  // previously taken vertex
#pragma unroll
    for (int j = 0; j < SV_OFF; ++j) {
      o[j] = pad;
      if (j < nv) {
        float x_min = 1.f;
        float y_min = (float)(-SV_EPS);
        int i_take = 0;
#pragma unroll
        for (int k = 0; k < SV_MAXV; ++k) {
          if (k < m && ((mbits >> k) & 1u)) {
            const float x = vx[k], y = vy[k];
            bool ok = sv_compare(x, y, x_min, y_min);
            if (j > 0) ok = ok && sv_compare(px, py, x, y);
            if (ok) { x_min = x; y_min = y; i_take = k; }
          }
        }
        o[j] = i_take;
        // the reference re-reads vertices[idx[j-1]]; i_take == 0 with no vertex
        // accepted means vertex 0, exactly as there.
        float tx = vx[0], ty = vy[0];
#pragma unroll
        for (int k = 1; k < SV_MAXV; ++k)
          if (k == i_take) { tx = vx[k]; ty = vy[k]; }
        px = tx; py = ty;
      }
    }
    o[SV_NIDX - 1] = pad;
    // close the polygon (:103) and pad (:106-108); nv <= 8 for rectangles
    const int first = o[0];
#pragma unroll
    for (int j = 0; j < SV_NIDX; ++j) {
      if (j == nv) o[j] = first;
      else if (j > nv) o[j] = pad;
    }
    if (nv == 8) {  // identical boxes (:114-129)
      int counter = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 4; k < SV_OFF; ++k) counter += (o[k] == o[j]) ? 1 : 0;
      if (counter == 4) {
        o[4] = o[0];
#pragma unroll
        for (int j = 5; j < SV_NIDX; ++j) o[j] = pad;
      }
    }
  }
  int *dst = idx + (size_t)i * SV_NIDX;
#pragma unroll
  for (int j = 0; j < SV_NIDX; ++j) dst[j] = o[j];
}

}  // namespace nesie

using namespace nesie;

extern "C" int nesie_sort_vertices_forward(int b, int n, int m, const float *vertices,
                                           const uint8_t *mask, const int *num_valid,
                                           int *idx, void *stream) {
  const char *W = "sort_vertices_forward";
  NESIE_REQUIRE(b >= 0 && n >= 0 && m >= 0, W);
  const long long total = (long long)b * n;
  if (total == 0) return NESIE_OK;
  NESIE_REQUIRE(vertices && mask && num_valid && idx, W);
  if (m > SV_MAXV || m < SV_OFF + 1) {
    set_error("%s: m = %d vertices per pair, built for 9..24", W, m);
    return NESIE_ERR_UNSUPPORTED;
  }
  NESIE_REQUIRE(total / 256 + 1 < (1ll << 31), W);
  hipLaunchKernelGGL(sort_vertices_kernel, dim3(cdiv(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, total, m, vertices, mask, num_valid, idx);
  return check_launch(W);
}
