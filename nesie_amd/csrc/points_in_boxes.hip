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


// ---- per-point vote targets (get_targets_single, nesie_head.py:593-654), one launch --------
// For every point: the in-box test above against the scene's first `count` boxes (given in the
// DEPTH frame; the depth -> LiDAR change of depth_box3d.py:263-266 / box_3d_mode.py:124-143 is
// exact: (x, y) -> (y, -x), sizes swapped), then the reference's three vote slots: slot 0 = the
// first box holding the point, slot 1 = the second if any, slot 2 = the LAST when there are three
// or more (its counter clamps at 2, :629-652); empty slots repeat slot 0; a point in no box
// gets zeros and mask 0.  vote = gravity centre (x, y, z + dz * 0.5 in fp32, depth_box3d.py:42-48)
// minus the point.  Replaces ~40 elementwise / reduce launches over the (B, N, T) table.
struct VtBox { PibBox q; float gx, gy, gz; };

__global__ __launch_bounds__(PIB_BLOCK) void vote_targets_kernel(
    int boxes_num, int pts_num, int pt_stride, const float *__restrict__ boxes,
    const long long *__restrict__ count, const float *__restrict__ pts,
    float *__restrict__ votes, long long *__restrict__ mask) {
  constexpr int TILE = 128;
  __shared__ VtBox sb[TILE];
  const int bi = blockIdx.y;
  const int p = blockIdx.x * PIB_BLOCK + threadIdx.x;
  const bool live = p < pts_num;
  boxes += (size_t)bi * boxes_num * 7;
  long long used = count[bi];
  const int nbox = used < 0 ? 0 : (used > boxes_num ? boxes_num : (int)used);
  const float *pt = pts + ((size_t)bi * pts_num + (live ? p : pts_num - 1)) * pt_stride;
  const float xd = pt[0], yd = pt[1], zd = pt[2];
  const float x = yd, y = -xd, z = zd;  // LiDAR frame
  int cnt = 0;
  float c1x = 0.f, c1y = 0.f, c1z = 0.f, c2x = 0.f, c2y = 0.f, c2z = 0.f;
  float clx = 0.f, cly = 0.f, clz = 0.f;
  for (int t0 = 0; t0 < nbox; t0 += TILE) {
    const int tn = nbox - t0 < TILE ? nbox - t0 : TILE;
    __syncthreads();
    for (int k = threadIdx.x; k < tn; k += PIB_BLOCK) {
      const float *bx = boxes + (size_t)(t0 + k) * 7;
      // LiDAR-frame box: (y, -x, z, dy, dx, dz, yaw): w = bx[4] (depth dy), l = bx[3] (depth dx)
      const float w = bx[4], l = bx[3], h = bx[5], rz = bx[6];
      VtBox v;
      v.q.cx = bx[1]; v.q.cy = -bx[0];
      v.q.czm = (float)((double)bx[2] + (double)h / 2.0);
      v.q.hl = (double)l / 2.0; v.q.hw = (double)w / 2.0; v.q.hh = (double)h / 2.0;
      const float rot_angle = (float)((double)rz + M_PI / 2);
      v.q.cosa = (float)cos((double)rot_angle);
      v.q.sina = (float)sin((double)rot_angle);
      v.gx = bx[0]; v.gy = bx[1]; v.gz = __fadd_rn(bx[2], __fmul_rn(h, 0.5f));
      sb[k] = v;
    }
    __syncthreads();
    for (int k = 0; k < tn; ++k) {
      const PibBox q = sb[k].q;
      if ((double)fabsf(__fsub_rn(z, q.czm)) > q.hh) continue;
      const float sx = __fsub_rn(x, q.cx), sy = __fsub_rn(y, q.cy);
      const float lx = __fadd_rn(__fmul_rn(sx, q.cosa), __fmul_rn(sy, -q.sina));
      const float ly = __fadd_rn(__fmul_rn(sx, q.sina), __fmul_rn(sy, q.cosa));
      const bool in = ((double)lx > -q.hl) & ((double)lx < q.hl) &
                      ((double)ly > -q.hw) & ((double)ly < q.hw);
      if (in) {
        const float gx = sb[k].gx, gy = sb[k].gy, gz = sb[k].gz;
        if (cnt == 0) { c1x = gx; c1y = gy; c1z = gz; }
        if (cnt == 1) { c2x = gx; c2y = gy; c2z = gz; }
        clx = gx; cly = gy; clz = gz;
        ++cnt;
      }
    }
  }
  if (!live) return;
  float *o = votes + ((size_t)bi * pts_num + p) * 9;
  const float v1x = __fsub_rn(c1x, xd), v1y = __fsub_rn(c1y, yd), v1z = __fsub_rn(c1z, zd);
  const bool any = cnt > 0, two = cnt >= 2, three = cnt >= 3;
  o[0] = any ? v1x : 0.f; o[1] = any ? v1y : 0.f; o[2] = any ? v1z : 0.f;
  o[3] = two ? __fsub_rn(c2x, xd) : o[0]; o[4] = two ? __fsub_rn(c2y, yd) : o[1];
  o[5] = two ? __fsub_rn(c2z, zd) : o[2];
  o[6] = three ? __fsub_rn(clx, xd) : o[0]; o[7] = three ? __fsub_rn(cly, yd) : o[1];
  o[8] = three ? __fsub_rn(clz, zd) : o[2];
  mask[(size_t)bi * pts_num + p] = any ? 1 : 0;
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

extern "C" int nesie_vote_targets(int b, int boxes_num, int pts_num, int pt_stride,
                                  const float *gt_boxes, const long long *gt_count,
                                  const float *points, float *vote_targets,
                                  long long *vote_target_masks, void *stream) {
  const char *W = "vote_targets";
  NESIE_REQUIRE(b >= 0 && boxes_num >= 0 && pts_num >= 0 && pt_stride >= 3, W);
  if (b == 0 || pts_num == 0) return NESIE_OK;
  NESIE_REQUIRE(points && vote_targets && vote_target_masks && gt_count, W);
  NESIE_REQUIRE(boxes_num == 0 || gt_boxes, W);
  NESIE_REQUIRE(b <= 65535, W);
  hipLaunchKernelGGL(vote_targets_kernel, dim3(cdiv(pts_num, PIB_BLOCK), b), dim3(PIB_BLOCK), 0,
                     (hipStream_t)stream, boxes_num, pts_num, pt_stride, gt_boxes, gt_count,
                     points, vote_targets, vote_target_masks);
  return check_launch(W);
}
