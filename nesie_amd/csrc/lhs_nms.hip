// LHS-NMS of the teacher's pseudo boxes, one wave per scene.
//
// Replaces lhs_3d_faster_samecls (reference mmdet3d/models/detectors/votenet_nesie.py:
// 733-779), which the reference runs in numpy on the host after a device->host copy
// inside the training step.  K <= 64 axis-aligned boxes per scene, one per lane: greedy
// over descending score; the same-class boxes overlapping the pick by IoU > thr are
// removed, and the better-scored half of them is kept too.  Volumes and IoUs in double
// (the reference's arrays are float64).  Equal scores order by index (np.argsort leaves
// that unspecified).
#include "common.h"

namespace nesie {

__global__ __launch_bounds__(64) void lhs_nms_kernel(int k, const float *__restrict__ boxes,
                                                     float thr, uint8_t *__restrict__ keep) {
  const int lane = threadIdx.x;
  const float *bx = boxes + ((size_t)blockIdx.x * k + (lane < k ? lane : 0)) * 8;
  const float x1 = bx[0], y1 = bx[1], z1 = bx[2], x2 = bx[3], y2 = bx[4], z2 = bx[5];
  const float score = bx[6], cls = bx[7];
  const double vol = ((double)x2 - x1) * ((double)y2 - y1) * ((double)z2 - z1) + 1e-8;
  unsigned long long alive = k >= 64 ? ~0ull : ((1ull << k) - 1ull);
  unsigned long long kept = 0ull;
  // rank key: larger = later in the ascending (score, index) order
  auto greater = [&](float sa, int ia, float sb, int ib) { return sa > sb || (sa == sb && ia > ib); };
  while (alive) {
    // the alive lane with the largest (score, index)
    int best = -1; float bs = 0.f;
    for (unsigned long long m = alive; m; m &= m - 1) {
      const int j = __builtin_ctzll(m);
      const float sj = __shfl(score, j, 64);
      if (best < 0 || greater(sj, j, bs, best)) { best = j; bs = sj; }
    }
    kept |= 1ull << best;
    const float ix1 = __shfl(x1, best, 64), iy1 = __shfl(y1, best, 64), iz1 = __shfl(z1, best, 64);
    const float ix2 = __shfl(x2, best, 64), iy2 = __shfl(y2, best, 64), iz2 = __shfl(z2, best, 64);
    const float icls = __shfl(cls, best, 64);
    const double ivol = __shfl(vol, best, 64);
    const double l = fmax(0.0, (double)fminf(ix2, x2) - (double)fmaxf(ix1, x1));
    const double w = fmax(0.0, (double)fminf(iy2, y2) - (double)fmaxf(iy1, y1));
    const double h = fmax(0.0, (double)fminf(iz2, z2) - (double)fmaxf(iz1, z1));
    const double inter = l * w * h;
    double o = inter / (ivol + vol - inter);
    o = o * (icls == cls ? 1.0 : 0.0);
    const bool mine = ((alive >> lane) & 1ull) && lane != best && o > (double)thr;
    const unsigned long long over = __ballot(mine);
    const int half = __popcll(over) / 2;
    // my rank among the overlapped: how many of them sort after me
    int after = 0;
    for (unsigned long long m = over; m; m &= m - 1) {
      const int j = __builtin_ctzll(m);
      const float sj = __shfl(score, j, 64);
      if (greater(sj, j, score, lane)) ++after;
    }
    kept |= __ballot(mine && after < half);
    alive &= ~(over | (1ull << best));
  }
  if (lane < k) keep[(size_t)blockIdx.x * k + lane] = (uint8_t)((kept >> lane) & 1ull);
}

}  // namespace nesie

using namespace nesie;

extern "C" int nesie_lhs_nms_samecls(int b, int k, const float *boxes, float thr, uint8_t *keep,
                                     void *stream) {
  const char *W = "lhs_nms_samecls";
  NESIE_REQUIRE(b >= 0 && k >= 0, W);
  if (b == 0 || k == 0) return NESIE_OK;
  NESIE_REQUIRE(boxes && keep, W);
  if (k > 64) { set_error("%s: k = %d boxes per scene, built for <= 64", W, k); return NESIE_ERR_UNSUPPORTED; }
  hipLaunchKernelGGL(lhs_nms_kernel, dim3(b), dim3(64), 0, (hipStream_t)stream, k, boxes, thr, keep);
  return check_launch(W);
}
