// Training-batch assembly from HBM-resident scenes: the per-sample CPU
// pipeline of the reference -- IndoorPointSample (transforms_3d.py:821-891), GlobalAlignment
// (:410-488), RandomFlip3D (:59-162), GlobalRotScaleTrans (:497-648) on DepthPoints
// (core/points/base_points.py:139-179,186-205,263-269; depth_points.py:28-33) -- as ONE gather
// pass over the points of a batch.  The raw xyz of every scene and the shifted-height column
// (loading.py:424-430, fixed per scene at load time) stay resident; a step reads 16 bytes and
// writes 16 bytes per sampled point.
#include "common.h"

namespace nesie {

// per scene: A[9] t[3] | flip_x flip_y | cos sin | scale | trans[3]  = 20 floats
constexpr int XF = 20;

__global__ __launch_bounds__(256) void scene_assemble_kernel(
    int n, long long pool_rows, const float *__restrict__ pool, const float *__restrict__ height,
    const int *__restrict__ choices, const float *__restrict__ xform, float4 *__restrict__ out) {
  __shared__ float xf[XF];
  const int bi = blockIdx.y;
  if (threadIdx.x < XF) xf[threadIdx.x] = xform[(size_t)bi * XF + threadIdx.x];
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  long long src = choices[(size_t)bi * n + i];
  src = src < 0 ? 0 : (src >= pool_rows ? pool_rows - 1 : src);
  const float *p = pool + src * 3;
  const float x = p[0], y = p[1], z = p[2];
  // GlobalAlignment: p @ R^T (row j of R dotted with p, left to right), then + t
  float ax = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(x, xf[0]), __fmul_rn(y, xf[1])), __fmul_rn(z, xf[2])), xf[9]);
  float ay = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(x, xf[3]), __fmul_rn(y, xf[4])), __fmul_rn(z, xf[5])), xf[10]);
  float az = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(x, xf[6]), __fmul_rn(y, xf[7])), __fmul_rn(z, xf[8])), xf[11]);
  // RandomFlip3D: horizontal negates x, vertical negates y (flip_* = -1 or +1)
  ax = xf[12] < 0.f ? -ax : ax;
  ay = xf[13] < 0.f ? -ay : ay;
  // GlobalRotScaleTrans: p @ [[c, s, 0], [-s, c, 0], [0, 0, 1]], * scale, + trans
  const float c = xf[14], s = xf[15], sc = xf[16];
  const float rx = __fadd_rn(__fmul_rn(ax, c), __fmul_rn(ay, -s));
  const float ry = __fadd_rn(__fmul_rn(ax, s), __fmul_rn(ay, c));
  float4 o;
  o.x = __fadd_rn(__fmul_rn(rx, sc), xf[17]);
  o.y = __fadd_rn(__fmul_rn(ry, sc), xf[18]);
  o.z = __fadd_rn(__fmul_rn(az, sc), xf[19]);
  o.w = __fmul_rn(height[src], sc);
  out[(size_t)bi * n + i] = o;
}

}  // namespace nesie

using namespace nesie;

extern "C" int nesie_scene_assemble(int b, int n, long long pool_rows, const float *pool,
                                    const float *height, const int *choices, const float *xform,
                                    float *out, void *stream) {
  const char *W = "scene_assemble";
  NESIE_REQUIRE(b >= 0 && n >= 0 && pool_rows >= 0, W);
  if (b == 0 || n == 0) return NESIE_OK;
  NESIE_REQUIRE(pool_rows > 0 && pool && height && choices && xform && out, W);
  NESIE_REQUIRE(b <= 65535 && ((uintptr_t)out & 15) == 0, W);
  hipLaunchKernelGGL(scene_assemble_kernel, dim3(cdiv(n, 256), b), dim3(256), 0,
                     (hipStream_t)stream, n, pool_rows, pool, height, choices, xform,
                     (float4 *)out);
  return check_launch(W);
}
