// Differentiable rotated 3-D IoU of box pairs in ONE kernel (value + Jacobian).
//
// Replaces, for the IoU3D loss and the IoU labels of the quality head, the ~100 small
// torch kernels per call of the reference's chain
//   cal_iou_3d -> cal_iou -> box2corners_th / oriented_box_intersection_2d
//   (mmdet3d/ops/rotated_iou/oriented_iou_loss.py:6-109, box_intersection_2d.py:13-184)
// plus its sort_vertices launch.  One thread owns one (prediction, target) pair and walks
// the same formulas in the same order: corners, 4x4 edge intersections with the
// t,u in (0,1) test and the `num + 1e-8` re-division, corner-in-box tests with 1e-6
// slack, mean-centred angular sort (sort_device.h), shoelace area, z overlap.
// Gradients w.r.t. the 7 parameters of the FIRST box are carried forward as dual numbers
// (masks and the vertex order are piecewise constant, exactly as autograd treats them);
// the second box is a constant (targets never require grad on this path).
#include "sort_device.h"
#include <math.h>

namespace nesie {

template <int ND>
struct Dual {
  float v;
  float d[ND > 0 ? ND : 1];
  __device__ Dual() {}
  __device__ explicit Dual(float c) : v(c) {
#pragma unroll
    for (int i = 0; i < ND; ++i) d[i] = 0.f;
  }
  __device__ static Dual var(float c, int k) {
    Dual r(c);
    if (ND > 0) r.d[k] = 1.f;
    return r;
  }
};
#define DUAL_BIN(OP, VAL, DA, DB)                                         \
  template <int ND>                                                        \
  __device__ __forceinline__ Dual<ND> OP(const Dual<ND> &a, const Dual<ND> &b) { \
    Dual<ND> r;                                                            \
    r.v = VAL;                                                             \
    _Pragma("unroll") for (int i = 0; i < ND; ++i) r.d[i] = (DA)*a.d[i] + (DB)*b.d[i]; \
    return r;                                                              \
  }
DUAL_BIN(operator+, a.v + b.v, 1.f, 1.f)
DUAL_BIN(operator-, a.v - b.v, 1.f, -1.f)
DUAL_BIN(operator*, a.v * b.v, b.v, a.v)
DUAL_BIN(operator/, a.v / b.v, 1.f / b.v, -(a.v / b.v) / b.v)
template <int ND>
__device__ __forceinline__ Dual<ND> operator*(const Dual<ND> &a, float c) {
  Dual<ND> r;
  r.v = a.v * c;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * c;
  return r;
}
template <int ND>
__device__ __forceinline__ Dual<ND> operator+(const Dual<ND> &a, float c) {
  Dual<ND> r = a;
  r.v = a.v + c;
  return r;
}
template <int ND>
__device__ __forceinline__ Dual<ND> operator-(const Dual<ND> &a) { return a * -1.f; }
template <int ND>
__device__ __forceinline__ Dual<ND> dsin(const Dual<ND> &a) {
  Dual<ND> r;
  r.v = sinf(a.v);
  const float c = cosf(a.v);
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = c * a.d[i];
  return r;
}
template <int ND>
__device__ __forceinline__ Dual<ND> dcos(const Dual<ND> &a) {
  Dual<ND> r;
  r.v = cosf(a.v);
  const float s = -sinf(a.v);
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = s * a.d[i];
  return r;
}
template <int ND>
__device__ __forceinline__ Dual<ND> dmin(const Dual<ND> &a, const Dual<ND> &b) {
  return a.v <= b.v ? a : b;
}
template <int ND>
__device__ __forceinline__ Dual<ND> dmax(const Dual<ND> &a, const Dual<ND> &b) {
  return a.v >= b.v ? a : b;
}

// corners of (x, y, w, h, alpha): (+,+) (-,+) (-,-) (+,-) halves rotated by +alpha
template <int ND>
__device__ __forceinline__ void corners_of(const Dual<ND> &x, const Dual<ND> &y, const Dual<ND> &w,
                                           const Dual<ND> &h, const Dual<ND> &al,
                                           Dual<ND> (&cx)[4], Dual<ND> (&cy)[4]) {
  const Dual<ND> s = dsin(al), c = dcos(al);
  const Dual<ND> hw = w * 0.5f, hh = h * 0.5f;
  const float sx[4] = {1.f, -1.f, -1.f, 1.f}, sy[4] = {1.f, 1.f, -1.f, -1.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const Dual<ND> x4 = hw * sx[k], y4 = hh * sy[k];
    cx[k] = (x4 * c + y4 * (-s)) + x;   // [x4, y4] @ [[c, s], [-s, c]]
    cy[k] = (x4 * s + y4 * c) + y;
  }
}

// SPLIT (with ND = 1): eight threads per pair, thread `comp` carries the derivative with respect
// to parameter `comp` only (the components of a dual number never mix, so every derivative is
// the same chain of operations as in the 7-wide form: identical bits) -- a quarter of the serial
// work per thread and eight times the threads for a kernel that is pure latency at 2 048 pairs.
template <int ND, bool SPLIT>
__global__ __launch_bounds__(64) void iou3d_kernel(int n, const float *__restrict__ box1,
                                                   const float *__restrict__ box2,
                                                   float *__restrict__ iou,
                                                   float *__restrict__ jac) {
  typedef Dual<ND> D;
  const int gid = blockIdx.x * 64 + threadIdx.x;
  const int i = SPLIT ? gid >> 3 : gid;
  const int comp = gid & 7;
  if (i >= n || (SPLIT && comp == 7)) return;
  const float *p = box1 + (size_t)i * 7, *q = box2 + (size_t)i * 7;
  D b1[7], b2[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    if (SPLIT) { b1[k] = D(p[k]); if (k == comp) b1[k].d[0] = 1.f; }
    else b1[k] = D::var(p[k], k);
    b2[k] = D(q[k]);
  }

  // ---- BEV corners -----------------------------------------------------------------
  D c1x[4], c1y[4], c2x[4], c2y[4];
  corners_of(b1[0], b1[1], b1[3], b1[4], b1[6], c1x, c1y);
  corners_of(b2[0], b2[1], b2[3], b2[4], b2[6], c2x, c2y);

  // ---- candidate vertices: 4 + 4 corners, 16 edge intersections ----------------------
  D vx[SV_MAXV], vy[SV_MAXV];
  unsigned mbits = 0u;
#pragma unroll
  for (int k = 0; k < 4; ++k) { vx[k] = c1x[k]; vy[k] = c1y[k]; vx[4 + k] = c2x[k]; vy[4 + k] = c2y[k]; }
#pragma unroll
  for (int e1 = 0; e1 < 4; ++e1) {
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      const D x1 = c1x[e1], y1 = c1y[e1], x2 = c1x[(e1 + 1) & 3], y2 = c1y[(e1 + 1) & 3];
      const D x3 = c2x[e2], y3 = c2y[e2], x4 = c2x[(e2 + 1) & 3], y4 = c2y[(e2 + 1) & 3];
      const D num = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4);
      const D den_t = (x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4);
      const D den_u = (x1 - x2) * (y1 - y3) - (y1 - y2) * (x1 - x3);
      const bool zero = num.v == 0.f;
      const float t = zero ? -1.f : den_t.v / num.v;
      const float u = zero ? -1.f : -den_u.v / num.v;
      const bool m = (t > 0.f) && (t < 1.f) && (u > 0.f) && (u < 1.f);
      const D t2 = den_t / (num + 1e-8f);
      const int slot = 8 + e1 * 4 + e2;
      if (m) {
        vx[slot] = x1 + t2 * (x2 - x1);
        vy[slot] = y1 + t2 * (y2 - y1);
        mbits |= 1u << slot;
      } else {
        vx[slot] = D(0.f); vy[slot] = D(0.f);  // masked: value 0, zero gradient
      }
    }
  }
  // corner-in-other-box tests (values only)
  {
    const float ax = c2x[0].v, ay = c2y[0].v;
    const float abx = c2x[1].v - ax, aby = c2y[1].v - ay, adx = c2x[3].v - ax, ady = c2y[3].v - ay;
    const float nab = abx * abx + aby * aby, nad = adx * adx + ady * ady;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float amx = c1x[k].v - ax, amy = c1y[k].v - ay;
      const float pab = (abx * amx + aby * amy) / nab, pad = (adx * amx + ady * amy) / nad;
      if (pab > -1e-6f && pab < 1.f + 1e-6f && pad > -1e-6f && pad < 1.f + 1e-6f) mbits |= 1u << k;
    }
  }
  {
    const float ax = c1x[0].v, ay = c1y[0].v;
    const float abx = c1x[1].v - ax, aby = c1y[1].v - ay, adx = c1x[3].v - ax, ady = c1y[3].v - ay;
    const float nab = abx * abx + aby * aby, nad = adx * adx + ady * ady;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float amx = c2x[k].v - ax, amy = c2y[k].v - ay;
      const float pab = (abx * amx + aby * amy) / nab, pad = (adx * amx + ady * amy) / nad;
      if (pab > -1e-6f && pab < 1.f + 1e-6f && pad > -1e-6f && pad < 1.f + 1e-6f) mbits |= 1u << (4 + k);
    }
  }
  // ---- order the valid vertices about their mean -------------------------------------
  const int nv = __popc(mbits);
  float sxm = 0.f, sym = 0.f;
#pragma unroll
  for (int k = 0; k < SV_MAXV; ++k) {
    const float mk = (mbits >> k) & 1u ? 1.f : 0.f;
    sxm += vx[k].v * mk; sym += vy[k].v * mk;
  }
  const float mxv = sxm / (float)nv, myv = sym / (float)nv;  // 0/0 -> NaN when nv == 0, unused
  float nx[SV_MAXV], ny[SV_MAXV];
#pragma unroll
  for (int k = 0; k < SV_MAXV; ++k) { nx[k] = vx[k].v - mxv; ny[k] = vy[k].v - myv; }
  int o[SV_NIDX];
  sv_sort_one(nx, ny, mbits, nv, SV_MAXV, o);

  // ---- shoelace over the 9 gathered (un-centred) vertices -----------------------------
  D total(0.f);
  for (int k = 0; k < SV_NIDX - 1; ++k) {
    const int a = o[k], b = o[k + 1];
    total = total + (vx[a] * vy[b] - vy[a] * vx[b]);
  }
  const D inter = (total.v >= 0.f ? total : -total) * 0.5f;

  // ---- IoU in the plane, then with the z overlap ---------------------------------------
  const D area1 = b1[3] * b1[4], area2 = b2[3] * b2[4];
  const D uni = area1 + area2 - inter;
  const D iou2d = inter / uni;
  const D zmax1 = b1[2] + b1[5] * 0.5f, zmin1 = b1[2] - b1[5] * 0.5f;
  const D zmax2 = b2[2] + b2[5] * 0.5f, zmin2 = b2[2] - b2[5] * 0.5f;
  D zov = dmin(zmax1, zmax2) - dmax(zmin1, zmin2);
  if (zov.v < 0.f) zov = D(0.f);  // clamp_min(0)
  const D inter3 = iou2d * uni * zov;
  const D v1 = b1[3] * b1[4] * b1[5], v2 = b2[3] * b2[4] * b2[5];
  const D out = inter3 / (v1 + v2 - inter3);
  if (SPLIT) {
    if (comp == 0) iou[i] = out.v;
    jac[(size_t)i * 7 + comp] = out.d[0];
    return;
  }
  iou[i] = out.v;
  if (ND > 0) {
#pragma unroll
    for (int k = 0; k < ND; ++k) jac[(size_t)i * 7 + k] = out.d[k];
  }
}

}  // namespace nesie

using namespace nesie;

extern "C" int nesie_iou3d_forward(int n, const float *box1, const float *box2, float *iou,
                                   float *jac, void *stream) {
  const char *W = "iou3d_forward";
  NESIE_REQUIRE(n >= 0, W);
  if (n == 0) return NESIE_OK;
  NESIE_REQUIRE(box1 && box2 && iou, W);
  if (jac)
    hipLaunchKernelGGL((iou3d_kernel<1, true>), dim3(cdiv((long long)n * 8, 64)), dim3(64), 0,
                       (hipStream_t)stream, n, box1, box2, iou, jac);
  else
    hipLaunchKernelGGL((iou3d_kernel<0, false>), dim3(cdiv(n, 64)), dim3(64), 0,
                       (hipStream_t)stream, n, box1, box2, iou, jac);
  return check_launch(W);
}
