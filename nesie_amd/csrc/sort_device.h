// Device-side vertex ordering shared by sort_vertices.hip and iou3d.hip.
// Follows sort_vertices_kernel / compare_vertices of the reference
// (mmdet3d/ops/rotated_iou/cuda_op/sort_vert_kernel.cu:15-134) term for term.
#pragma once
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


// vx/vy: the m <= 24 mean-centred candidate vertices, mbits: validity mask, nv: number
// of valid ones.  Writes the 9 indices (closed polygon + padding) into o.
__device__ __forceinline__ void sv_sort_one(const float (&vx)[SV_MAXV], const float (&vy)[SV_MAXV],
                                            unsigned mbits, int nv, int m, int (&o)[SV_NIDX]) {
  int pad = m - 1;
  {
    unsigned free_slots = ~mbits & (((m >= 32) ? 0xFFFFFFFFu : ((1u << m) - 1u)) & ~0xFFu);
    if (free_slots) pad = __ffs(free_slots) - 1;
  }
  if (nv < 3) {
#pragma unroll
    for (int j = 0; j < SV_NIDX; ++j) o[j] = pad;
    return;
  }
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

}  // namespace nesie
