// Inference post-processing and evaluation geometry for gfx950:
//   * aligned_3d_nms        -- core/post_processing/box3d_nms.py:129-176, batched over scenes
//   * points_in_boxes_count -- the `box_indices.T.sum(1) > 5` of NesieHead.multiclass_nms_single
//                              (dense_heads/nesie_head.py:740-750) without the (M, T) table
//   * boxes_overlap_bev     -- ops/iou3d/src/iou3d_kernel.cu:127-264 (rotated BEV overlap of
//                              BaseInstance3DBoxes.overlaps, base_box3d.py:387-438)
// The reference runs the NMS as a python loop of ~10 torch launches per pick and per scene;
// here one workgroup per scene sorts, builds the K x K suppression bit matrix in LDS and
// walks it once.
#include "common.h"
#include <math.h>

namespace nesie {

// ---- aligned_3d_nms -----------------------------------------------------------------
// Candidate order: ascending (score, index), read from the end -- a stable
// torch.argsort(scores) (the reference's own sort leaves equal scores unordered).
// Box u leaves the list when a picked box t before it has NOT (iou(t,u) * same <= thr), in
// fp32 with the reference's operation order; NaN (0/0 of two degenerate boxes) fails `<=`
// and therefore removes u, as in the reference.
constexpr int NMS_MAX_K = 512;
constexpr int NMS_BLOCK = 256;

struct NmsBox { float x1, y1, z1, x2, y2, z2, area; int cls; };

__global__ __launch_bounds__(NMS_BLOCK) void aligned_nms_kernel(
    int k, const float *__restrict__ boxes, const float *__restrict__ scores,
    const int *__restrict__ classes, const uint8_t *__restrict__ valid, float thr,
    int *__restrict__ picks, int *__restrict__ count) {
  __shared__ NmsBox sb[NMS_MAX_K];            // candidates, best first
  __shared__ int src[NMS_MAX_K];              // their positions in the input
  __shared__ float sc[NMS_MAX_K];
  __shared__ unsigned char ok[NMS_MAX_K];
  __shared__ unsigned long long sup[NMS_MAX_K * (NMS_MAX_K / 64)];
  __shared__ int n_live;
  const int bi = blockIdx.x, tid = threadIdx.x;
  boxes += (size_t)bi * k * 6;
  scores += (size_t)bi * k;
  classes += (size_t)bi * k;
  if (valid) valid += (size_t)bi * k;
  picks += (size_t)bi * k;
  if (tid == 0) n_live = 0;
  for (int i = tid; i < k; i += NMS_BLOCK) {
    const float s0 = scores[i];
    sc[i] = s0 != s0 ? INFINITY : s0;   // NaN sorts last, as in torch.argsort
    ok[i] = valid ? (valid[i] != 0) : 1;
    picks[i] = -1;
  }
  __syncthreads();
  // rank = number of live candidates that come before me (greater (score, index))
  for (int i = tid; i < k; i += NMS_BLOCK) {
    if (!ok[i]) continue;
    const float s = sc[i];
    int r = 0;
    for (int j = 0; j < k; ++j) r += ok[j] && (sc[j] > s || (sc[j] == s && j > i));
    const float *bx = boxes + (size_t)i * 6;
    NmsBox q;
    q.x1 = bx[0]; q.y1 = bx[1]; q.z1 = bx[2]; q.x2 = bx[3]; q.y2 = bx[4]; q.z2 = bx[5];
    q.area = __fmul_rn(__fmul_rn(__fsub_rn(q.x2, q.x1), __fsub_rn(q.y2, q.y1)),
                       __fsub_rn(q.z2, q.z1));
    q.cls = classes[i];
    sb[r] = q;
    src[r] = i;
    atomicAdd(&n_live, 1);
  }
  __syncthreads();
  const int n = n_live;
  const int words = (n + 63) >> 6;
  // suppression matrix: bit u of row t <=> t removes u (u after t)
  for (int w = tid; w < n * words; w += NMS_BLOCK) {
    const int t = w / words, u0 = (w % words) << 6;
    const NmsBox a = sb[t];
    unsigned long long bits = 0ull;
    const int uend = n - u0 < 64 ? n - u0 : 64;
    for (int j = 0; j < uend; ++j) {
      const int u = u0 + j;
      if (u <= t) continue;
      const NmsBox q = sb[u];
      const float l = fmaxf(0.f, __fsub_rn(fminf(a.x2, q.x2), fmaxf(a.x1, q.x1)));
      const float wd = fmaxf(0.f, __fsub_rn(fminf(a.y2, q.y2), fmaxf(a.y1, q.y1)));
      const float h = fmaxf(0.f, __fsub_rn(fminf(a.z2, q.z2), fmaxf(a.z1, q.z1)));
      const float inter = __fmul_rn(__fmul_rn(l, wd), h);
      float iou = __fdiv_rn(inter, __fsub_rn(__fadd_rn(a.area, q.area), inter));
      iou = __fmul_rn(iou, a.cls == q.cls ? 1.f : 0.f);
      if (!(iou <= thr)) bits |= 1ull << j;
    }
    sup[w] = bits;
  }
  __syncthreads();
  // greedy walk by the first wave: lane w owns word w of the removed set
  if (tid < 64) {
    unsigned long long removed = 0ull;
    int np = 0;
    for (int t = 0; t < n; ++t) {
      const unsigned long long word = __shfl(removed, t >> 6, 64);
      if ((word >> (t & 63)) & 1ull) continue;   // uniform across the wave
      if (tid == 0) picks[np] = src[t];
      ++np;
      if (tid < words) removed |= sup[t * words + tid];
    }
    if (tid == 0) count[bi] = np;
  }
}

// zero-fill as a kernel: a hipMemsetAsync captured into a hipGraph next to the kernel that adds
// into the same buffer replayed WITHOUT the ordering between the two on this stack (counts came
// back partly zeroed on later replays), so the clears below are ordinary launches
__global__ __launch_bounds__(256) void zero_i32_kernel(int n, int *__restrict__ p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0;
}

// ---- points_in_boxes_count ------------------------------------------------------------
// Same inside-test as points_in_boxes.hip (points_in_boxes_cuda.cu:24-49), LiDAR-frame boxes;
// counts[b, t] = number of the scene's points inside box t.
constexpr int PIC_BLOCK = 256;
constexpr int PIC_TILE = 256;

struct PicBox { float cx, cy, czm; double hl, hw, hh; float cosa, sina; };

__global__ __launch_bounds__(PIC_BLOCK) void points_in_boxes_count_kernel(
    int boxes_num, int pts_num, const float *__restrict__ boxes,
    const float *__restrict__ pts, int *__restrict__ counts) {
  __shared__ PicBox sb[PIC_TILE];
  __shared__ int cnt[PIC_TILE];
  const int bi = blockIdx.y;
  const int p = blockIdx.x * PIC_BLOCK + threadIdx.x;
  const bool live = p < pts_num;
  boxes += (size_t)bi * boxes_num * 7;
  const float *pt = pts + ((size_t)bi * pts_num + (live ? p : pts_num - 1)) * 3;
  const float x = pt[0], y = pt[1], z = pt[2];
  for (int t0 = 0; t0 < boxes_num; t0 += PIC_TILE) {
    const int tn = boxes_num - t0 < PIC_TILE ? boxes_num - t0 : PIC_TILE;
    __syncthreads();
    for (int k = threadIdx.x; k < tn; k += PIC_BLOCK) {
      const float *bx = boxes + (size_t)(t0 + k) * 7;
      const float w = bx[3], l = bx[4], h = bx[5], rz = bx[6];
      PicBox q;
      q.cx = bx[0]; q.cy = bx[1];
      q.czm = (float)((double)bx[2] + (double)h / 2.0);
      q.hl = (double)l / 2.0; q.hw = (double)w / 2.0; q.hh = (double)h / 2.0;
      const float rot_angle = (float)((double)rz + M_PI / 2);
      q.cosa = (float)cos((double)rot_angle);
      q.sina = (float)sin((double)rot_angle);
      sb[k] = q;
      cnt[k] = 0;
    }
    __syncthreads();
    for (int k = 0; k < tn; ++k) {
      const PicBox q = sb[k];
      bool in = live && !((double)fabsf(__fsub_rn(z, q.czm)) > q.hh);
      if (in) {
        const float sx = __fsub_rn(x, q.cx), sy = __fsub_rn(y, q.cy);
        const float lx = __fadd_rn(__fmul_rn(sx, q.cosa), __fmul_rn(sy, -q.sina));
        const float ly = __fadd_rn(__fmul_rn(sx, q.sina), __fmul_rn(sy, q.cosa));
        in = ((double)lx > -q.hl) & ((double)lx < q.hl) & ((double)ly > -q.hw) &
             ((double)ly < q.hw);
      }
      const unsigned long long m = __ballot(in);
      if (m && (threadIdx.x & 63) == 0) atomicAdd(&cnt[k], __popcll(m));
    }
    __syncthreads();
    for (int k = threadIdx.x; k < tn; k += PIC_BLOCK)
      if (cnt[k]) atomicAdd(&counts[(size_t)bi * boxes_num + t0 + k], cnt[k]);
  }
}

// ---- rotated BEV overlap ------------------------------------------------------------------
// One thread per (a, b) pair of (x1, y1, x2, y2, angle) rectangles: edge-edge crossings,
// corners of one inside the other (1e-5 slack), angular order about the mean point, fan
// area -- the construction of iou3d_kernel.cu:127-238 with its constants.  cos / sin / atan2
// are taken in double and rounded to float (the canonical form shared with the oracle; the
// reference's device cosf / sinf / atan2f lie within their own error of it).
struct P2 { float x, y; };

__device__ __forceinline__ float cross3(const P2 &p1, const P2 &p2, const P2 &p0) {
  return __fsub_rn(__fmul_rn(__fsub_rn(p1.x, p0.x), __fsub_rn(p2.y, p0.y)),
                   __fmul_rn(__fsub_rn(p2.x, p0.x), __fsub_rn(p1.y, p0.y)));
}

__device__ __forceinline__ bool spans_touch(const P2 &p1, const P2 &p2, const P2 &q1,
                                            const P2 &q2) {
  return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
         fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

// crossing of segment p0-p1 with q0-q1 (iou3d_kernel.cu:80-111)
__device__ __forceinline__ bool edge_crossing(const P2 &p1, const P2 &p0, const P2 &q1,
                                              const P2 &q0, P2 &ans) {
  if (!spans_touch(p0, p1, q0, q1)) return false;
  const float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0);
  const float s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
  if (!(__fmul_rn(s1, s2) > 0.f && __fmul_rn(s3, s4) > 0.f)) return false;
  const float s5 = cross3(q1, p1, p0);
  const float den = __fsub_rn(s5, s1);
  if (fabsf(den) > 1e-8f) {
    ans.x = __fdiv_rn(__fsub_rn(__fmul_rn(s5, q0.x), __fmul_rn(s1, q1.x)), den);
    ans.y = __fdiv_rn(__fsub_rn(__fmul_rn(s5, q0.y), __fmul_rn(s1, q1.y)), den);
  } else {
    const float a0 = __fsub_rn(p0.y, p1.y), b0 = __fsub_rn(p1.x, p0.x);
    const float c0 = __fsub_rn(__fmul_rn(p0.x, p1.y), __fmul_rn(p1.x, p0.y));
    const float a1 = __fsub_rn(q0.y, q1.y), b1 = __fsub_rn(q1.x, q0.x);
    const float c1 = __fsub_rn(__fmul_rn(q0.x, q1.y), __fmul_rn(q1.x, q0.y));
    const float D = __fsub_rn(__fmul_rn(a0, b1), __fmul_rn(a1, b0));
    ans.x = __fdiv_rn(__fsub_rn(__fmul_rn(b0, c1), __fmul_rn(b1, c0)), D);
    ans.y = __fdiv_rn(__fsub_rn(__fmul_rn(a1, c0), __fmul_rn(a0, c1)), D);
  }
  return true;
}

struct Rect {
  float x1, y1, x2, y2, cx, cy, cosa, sina;
  P2 c[5];
};

__device__ __forceinline__ P2 spin(float px, float py, float cx, float cy, float c, float s) {
  const float dx = __fsub_rn(px, cx), dy = __fsub_rn(py, cy);
  P2 r;
  r.x = __fadd_rn(__fadd_rn(__fmul_rn(dx, c), __fmul_rn(dy, s)), cx);
  r.y = __fadd_rn(__fadd_rn(__fmul_rn(-dx, s), __fmul_rn(dy, c)), cy);
  return r;
}

__device__ __forceinline__ void load_rect(const float *b, Rect &r) {
  r.x1 = b[0]; r.y1 = b[1]; r.x2 = b[2]; r.y2 = b[3];
  r.cx = __fdiv_rn(__fadd_rn(r.x1, r.x2), 2.f);
  r.cy = __fdiv_rn(__fadd_rn(r.y1, r.y2), 2.f);
  r.cosa = (float)cos((double)b[4]);
  r.sina = (float)sin((double)b[4]);
  r.c[0] = spin(r.x1, r.y1, r.cx, r.cy, r.cosa, r.sina);
  r.c[1] = spin(r.x2, r.y1, r.cx, r.cy, r.cosa, r.sina);
  r.c[2] = spin(r.x2, r.y2, r.cx, r.cy, r.cosa, r.sina);
  r.c[3] = spin(r.x1, r.y2, r.cx, r.cy, r.cosa, r.sina);
  r.c[4] = r.c[0];
}

// p inside the rotated rectangle r: turned back by cos(-angle), sin(-angle) (:54-78)
__device__ __forceinline__ bool inside_rect(const Rect &r, const P2 &p) {
  const float M = 1e-5f;
  const P2 q = spin(p.x, p.y, r.cx, r.cy, r.cosa, -r.sina);
  return q.x > __fsub_rn(r.x1, M) && q.x < __fadd_rn(r.x2, M) && q.y > __fsub_rn(r.y1, M) &&
         q.y < __fadd_rn(r.y2, M);
}

__global__ __launch_bounds__(64) void boxes_overlap_bev_kernel(
    int num_a, const float *__restrict__ boxes_a, int num_b, const float *__restrict__ boxes_b,
    float *__restrict__ out) {
  const long long pair = (long long)blockIdx.x * 64 + threadIdx.x;
  if (pair >= (long long)num_a * num_b) return;
  const int ia = (int)(pair / num_b), ib = (int)(pair % num_b);
  Rect A, B;
  load_rect(boxes_a + (size_t)ia * 5, A);
  load_rect(boxes_b + (size_t)ib * 5, B);
  P2 pts[16];
  float ang[16];
  int cnt = 0;
  float sx = 0.f, sy = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      P2 hit;
      if (edge_crossing(A.c[i + 1], A.c[i], B.c[j + 1], B.c[j], hit)) {
        sx = __fadd_rn(sx, hit.x); sy = __fadd_rn(sy, hit.y);
        pts[cnt++] = hit;
      }
    }
  for (int k = 0; k < 4; ++k) {
    if (inside_rect(A, B.c[k])) {
      sx = __fadd_rn(sx, B.c[k].x); sy = __fadd_rn(sy, B.c[k].y);
      pts[cnt++] = B.c[k];
    }
    if (inside_rect(B, A.c[k])) {
      sx = __fadd_rn(sx, A.c[k].x); sy = __fadd_rn(sy, A.c[k].y);
      pts[cnt++] = A.c[k];
    }
  }
  float area = 0.f;
  if (cnt > 0) {
    const float mx = __fdiv_rn(sx, (float)cnt), my = __fdiv_rn(sy, (float)cnt);
    for (int i = 0; i < cnt; ++i)
      ang[i] = (float)atan2((double)__fsub_rn(pts[i].y, my), (double)__fsub_rn(pts[i].x, mx));
    // stable ascending order by angle (the reference's adjacent-swap passes)
    for (int i = 1; i < cnt; ++i) {
      const P2 v = pts[i];
      const float av = ang[i];
      int j = i - 1;
      while (j >= 0 && ang[j] > av) { pts[j + 1] = pts[j]; ang[j + 1] = ang[j]; --j; }
      pts[j + 1] = v; ang[j + 1] = av;
    }
    for (int k = 0; k + 1 < cnt; ++k) {
      const float ax = __fsub_rn(pts[k].x, pts[0].x), ay = __fsub_rn(pts[k].y, pts[0].y);
      const float bx = __fsub_rn(pts[k + 1].x, pts[0].x), by = __fsub_rn(pts[k + 1].y, pts[0].y);
      area = __fadd_rn(area, __fsub_rn(__fmul_rn(ax, by), __fmul_rn(ay, bx)));
    }
  }
  out[pair] = (float)((double)fabsf(area) / 2.0);
}

}  // namespace nesie

using namespace nesie;

extern "C" int nesie_aligned_3d_nms(int b, int k, const float *boxes, const float *scores,
                                    const int *classes, const uint8_t *valid, float thr,
                                    int *picks, int *count, void *stream) {
  const char *W = "aligned_3d_nms";
  NESIE_REQUIRE(b >= 0 && k >= 0, W);
  if (b == 0) return NESIE_OK;
  NESIE_REQUIRE(count, W);
  if (k > NMS_MAX_K) {
    set_error("%s: k = %d boxes per scene, built for <= %d", W, k, NMS_MAX_K);
    return NESIE_ERR_UNSUPPORTED;
  }
  if (k == 0) {
    hipLaunchKernelGGL(zero_i32_kernel, dim3(cdiv(b, 256)), dim3(256), 0, (hipStream_t)stream, b, count);
    return check_launch(W);
  }
  NESIE_REQUIRE(boxes && scores && classes && picks, W);
  hipLaunchKernelGGL(aligned_nms_kernel, dim3(b), dim3(NMS_BLOCK), 0, (hipStream_t)stream, k,
                     boxes, scores, classes, valid, thr, picks, count);
  return check_launch(W);
}

extern "C" int nesie_points_in_boxes_count(int b, int boxes_num, int pts_num,
                                           const float *boxes, const float *pts, int *counts,
                                           void *stream) {
  const char *W = "points_in_boxes_count";
  NESIE_REQUIRE(b >= 0 && boxes_num >= 0 && pts_num >= 0, W);
  if (b == 0 || boxes_num == 0) return NESIE_OK;
  NESIE_REQUIRE(counts, W);
  hipLaunchKernelGGL(zero_i32_kernel, dim3(cdiv((long long)b * boxes_num, 256)), dim3(256), 0,
                     (hipStream_t)stream, b * boxes_num, counts);
  if (pts_num == 0) return NESIE_OK;
  NESIE_REQUIRE(boxes && pts && b <= 65535, W);
  hipLaunchKernelGGL(points_in_boxes_count_kernel, dim3(cdiv(pts_num, PIC_BLOCK), b),
                     dim3(PIC_BLOCK), 0, (hipStream_t)stream, boxes_num, pts_num, boxes, pts,
                     counts);
  return check_launch(W);
}

extern "C" int nesie_boxes_overlap_bev(int num_a, const float *boxes_a, int num_b,
                                       const float *boxes_b, float *ans_overlap, void *stream) {
  const char *W = "boxes_overlap_bev";
  NESIE_REQUIRE(num_a >= 0 && num_b >= 0, W);
  if (num_a == 0 || num_b == 0) return NESIE_OK;
  NESIE_REQUIRE(boxes_a && boxes_b && ans_overlap, W);
  const long long pairs = (long long)num_a * num_b;
  NESIE_REQUIRE(pairs / 64 + 1 < (1ll << 31), W);
  hipLaunchKernelGGL(boxes_overlap_bev_kernel, dim3((unsigned)cdiv(pairs, 64)), dim3(64), 0,
                     (hipStream_t)stream, num_a, boxes_a, num_b, boxes_b, ans_overlap);
  return check_launch(W);
}
