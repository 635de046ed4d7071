// points_in_boxes_batch for gfx950 (vote-target generation).
//
// Replaces points_in_boxes_batch_kernel and check_pt_in_box3d
// (reference mmdet3d/ops/roiaware_pool3d/src/points_in_boxes_cuda.cu:79-105, :24-49).
// A thread owns one point; the scene's boxes are expanded ONCE per workgroup
// into LDS as (cx, cy, cz_mid, l/2, w/2, h/2, cos, sin) so the per-point loop
// has no trigonometry.  cos/sin are evaluated in double and rounded to float,
// the canonical form shared with oracle/nesie_oracle.c (the reference's device
// cosf/sinf lie within their own 2-ulp error of it).  The mixed float/double
// comparisons of the reference are kept: z-window and half extents in double.
// out[(b, p, k)] is written 1 where inside and left untouched otherwise.
#include "common.h"
#include <math.h>

namespace nesie {

constexpr int PIB_BLOCK = 256;
constexpr int PIB_TILE = 256;  // boxes per LDS tile

struct PibBox {
  float cx, cy, czm;       // centre, z already shifted to mid height
  double hl, hw, hh;       // half length (box[4]), half width (box[3]), half height
  float cosa, sina;
};

__global__ __launch_bounds__(PIB_BLOCK) void points_in_boxes_batch_kernel(
    int boxes_num, int pts_num, const float *__restrict__ boxes,
    const float *__restrict__ pts, int *__restrict__ out) {
  __shared__ PibBox sb[PIB_TILE];
  const int bi = blockIdx.y;
  const int p = blockIdx.x * PIB_BLOCK + threadIdx.x;
  const bool live = p < pts_num;
  boxes += (size_t)bi * boxes_num * 7;
  const float *pt = pts + ((size_t)bi * pts_num + (live ? p : pts_num - 1)) * 3;
  const float x = pt[0], y = pt[1], z = pt[2];
  int *o = out + ((size_t)bi * pts_num + (live ? p : 0)) * boxes_num;
  for (int t0 = 0; t0 < boxes_num; t0 += PIB_TILE) {
    const int tn = boxes_num - t0 < PIB_TILE ? boxes_num - t0 : PIB_TILE;
    __syncthreads();
    for (int k = threadIdx.x; k < tn; k += PIB_BLOCK) {
      const float *bx = boxes + (size_t)(t0 + k) * 7;
      const float w = bx[3], l = bx[4], h = bx[5], rz = bx[6];
      PibBox q;
      q.cx = bx[0]; q.cy = bx[1];
      q.czm = (float)((double)bx[2] + (double)h / 2.0);  // cz += h / 2.0
      q.hl = (double)l / 2.0; q.hw = (double)w / 2.0; q.hh = (double)h / 2.0;
      const float rot_angle = (float)((double)rz + M_PI / 2);
      q.cosa = (float)cos((double)rot_angle);
      q.sina = (float)sin((double)rot_angle);
      sb[k] = q;
    }
    __syncthreads();
    if (live) {
      for (int k = 0; k < tn; ++k) {
        const PibBox q = sb[k];
        if ((double)fabsf(__fsub_rn(z, q.czm)) > q.hh) continue;
        const float sx = __fsub_rn(x, q.cx), sy = __fsub_rn(y, q.cy);
        const float lx = __fadd_rn(__fmul_rn(sx, q.cosa), __fmul_rn(sy, -q.sina));
        const float ly = __fadd_rn(__fmul_rn(sx, q.sina), __fmul_rn(sy, q.cosa));
        const bool in = ((double)lx > -q.hl) & ((double)lx < q.hl) &
                        ((double)ly > -q.hw) & ((double)ly < q.hw);
        if (in) o[t0 + k] = 1;
      }
    }
  }
}

}  // namespace nesie

using namespace nesie;

extern "C" int nesie_points_in_boxes_batch(int b, int boxes_num, int pts_num,
                                           const float *boxes, const float *pts, int *out,
                                           void *stream) {
  const char *W = "points_in_boxes_batch";
  NESIE_REQUIRE(b >= 0 && boxes_num >= 0 && pts_num >= 0, W);
  if (b == 0 || boxes_num == 0 || pts_num == 0) return NESIE_OK;
  NESIE_REQUIRE(boxes && pts && out, W);
  NESIE_REQUIRE(b <= 65535, W);
  hipLaunchKernelGGL(points_in_boxes_batch_kernel, dim3(cdiv(pts_num, PIB_BLOCK), b),
                     dim3(PIB_BLOCK), 0, (hipStream_t)stream, boxes_num, pts_num, boxes, pts,
                     out);
  return check_launch(W);
}
