// Ball query for gfx950.
//
// Replaces ball_query_kernel (reference mmdet3d/ops/ball_query/src/ball_query_cuda.cu:11-54),
// which gives every centre one thread that scans all N points serially.  Here a
// WAVE owns CPW centres: its 64 lanes test 64 consecutive points at a time
// (one dense 768-byte load shared by the CPW centres), a ballot gives the hit
// mask, and mbcnt gives each hit lane its rank, so hits land in ascending point
// index exactly as the serial scan would record them.  The row is assembled in
// LDS and written out once: slot s = s-th hit for s < cnt, the first hit for
// s >= cnt ("first hit back-fills all slots", .cu:44-48), untouched if cnt == 0.
#include "common.h"
#include "fps_keys.h"

namespace nesie {

constexpr int BQ_BLOCK = 256;  // 4 waves
constexpr int BQ_CPW = 4;      // centres per wave

template <int FORM>
__global__ __launch_bounds__(BQ_BLOCK) void ball_query_kernel(
    int b, int n, int m, float min_radius, float max_radius, int nsample,
    const float *__restrict__ new_xyz, const float *__restrict__ xyz,
    int *__restrict__ idx) {
  extern __shared__ int bq_slots[];  // [4 waves][CPW][nsample]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // scene = blockIdx % b keeps one scene's blocks on one XCD's L2 when b | 8.
  const int scene = blockIdx.x % b;
  const int group = blockIdx.x / b;
  const int c0 = (group * (BQ_BLOCK / 64) + wave) * BQ_CPW;
  int *slots = bq_slots + wave * BQ_CPW * nsample;

  const float max_r2 = __fmul_rn(max_radius, max_radius);
  const float min_r2 = __fmul_rn(min_radius, min_radius);
  xyz += (size_t)scene * n * 3;
  new_xyz += (size_t)scene * m * 3;
  idx += (size_t)scene * m * nsample;

  float cx[BQ_CPW], cy[BQ_CPW], cz[BQ_CPW];
  int cnt[BQ_CPW];
#pragma unroll
  for (int c = 0; c < BQ_CPW; ++c) {
    int ci = c0 + c < m ? c0 + c : m - 1;
    cx[c] = new_xyz[ci * 3 + 0];
    cy[c] = new_xyz[ci * 3 + 1];
    cz[c] = new_xyz[ci * 3 + 2];
    cnt[c] = c0 + c < m ? 0 : nsample;  // out-of-range centres are "full"
  }

  if (c0 < m) {
    for (int base = 0; base < n; base += 64) {
      const int k = base + lane;
      const bool valid = k < n;
      const int kk = valid ? k : n - 1;
      const float x = xyz[kk * 3 + 0], y = xyz[kk * 3 + 1], z = xyz[kk * 3 + 2];
      bool all_full = true;
#pragma unroll
      for (int c = 0; c < BQ_CPW; ++c) {
        if (cnt[c] < nsample) {  // wave-uniform
          float d2 = sqdist_form<FORM>(cx[c] - x, cy[c] - y, cz[c] - z);
          bool hit = valid && (d2 == 0.f || (d2 >= min_r2 && d2 < max_r2));
          unsigned long long mask = __ballot(hit);
          int rank = __builtin_amdgcn_mbcnt_hi(
              (unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
          int pos = cnt[c] + rank;
          if (hit && pos < nsample) slots[c * nsample + pos] = k;
          cnt[c] += __popcll(mask);
          if (cnt[c] < nsample) all_full = false;
        }
      }
      if (all_full) break;
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < BQ_CPW; ++c) {
    if (c0 + c < m && cnt[c] > 0) {
      const int have = cnt[c] < nsample ? cnt[c] : nsample;
      const int first = slots[c * nsample];
      int *row = idx + (size_t)(c0 + c) * nsample;
      for (int s = lane; s < nsample; s += 64)
        row[s] = s < have ? slots[c * nsample + s] : first;
    }
  }
}


// ---- ball query over the spatial index the FPS kernel leaves behind -----------------------
// fps_pruned_kernel sorts a scene by Morton cell into buckets of 64 points and leaves
// (x, y, z, key of the original index) plus one bounding box per bucket (fps_keys.h).  A centre
// only has to look at the buckets whose box comes within max_radius: with
//     d2box = ((ex*ex)+(ey*ey))+(ez*ez),  ex = max(lo.x - c.x, c.x - hi.x, 0), ...
// evaluated by the same fp32 operations as a point distance, rounding monotonicity gives
// d(p, c) >= d2box for every p in the box, so a bucket with d2box >= max_r2 holds no hit
// (a hit needs d2 < max_r2, or d2 == 0 < max_r2) and skipping it leaves the hit set unchanged.
// The reference records hits in ascending point index and stops at nsample (.cu:38-52), i.e.
// it returns the nsample SMALLEST hit indices in ascending order: here one wave per centre
// collects the hits of the ~10 surviving buckets (of 625 at 40 000 points), sorts 64 at a time
// with a bitonic network in registers and keeps the 64 smallest, then writes the row as the
// brute-force kernel does.  nsample <= 64.
__device__ __forceinline__ unsigned bq_sort64(unsigned v, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j >= 1; j >>= 1) {
      const unsigned o = (unsigned)__shfl_xor((int)v, j, 64);
      const bool up = (lane & k) == 0, lower = (lane & j) == 0;
      v = (up == lower) ? (v < o ? v : o) : (v > o ? v : o);
    }
  }
  return v;
}

// best, add: ascending over the lanes; returns the 64 smallest of the 128, ascending
__device__ __forceinline__ unsigned bq_merge_low64(unsigned best, unsigned add, int lane) {
  const unsigned rev = (unsigned)__shfl((int)add, 63 - lane, 64);
  unsigned v = best < rev ? best : rev;  // bitonic, holds the 64 smallest
#pragma unroll
  for (int j = 32; j >= 1; j >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)v, j, 64);
    v = ((lane & j) == 0) ? (v < o ? v : o) : (v > o ? v : o);
  }
  return v;
}

constexpr int BQI_WAVES = 4;

__global__ __launch_bounds__(BQI_WAVES * 64) void ball_query_indexed_kernel(
    int b, int n, int m, float min_radius, float max_radius, int nsample, int L,
    const float *__restrict__ new_xyz, const float4 *__restrict__ pts,
    const float *__restrict__ boxes, long long box_stride, int *__restrict__ idx) {
  __shared__ unsigned pend_all[BQI_WAVES][128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int scene = blockIdx.x % b;  // one scene's workgroups on one XCD's L2 when b | 8
  const int c = (blockIdx.x / b) * BQI_WAVES + wave;
  if (c >= m) return;  // wave-uniform; no block-wide barrier below
  volatile unsigned *pend = pend_all[wave];
  const float max_r2 = __fmul_rn(max_radius, max_radius);
  const float min_r2 = __fmul_rn(min_radius, min_radius);
  pts += (size_t)scene * n;
  const float *box = boxes + (size_t)scene * box_stride;
  const float *ctr = new_xyz + ((size_t)scene * m + c) * 3;
  const float cx = ctr[0], cy = ctr[1], cz = ctr[2];
  const int nb = (n + 63) >> 6;

  unsigned best = 0xFFFFFFFFu;  // lane i: i-th smallest hit index so far
  int npend = 0, total = 0;
  for (int s0 = 0; s0 < nb; s0 += 64) {
    const int bid = s0 + lane;
    bool act = false;
    if (bid < nb) {
      const float ex = fmaxf(fmaxf(box[bid] - cx, cx - box[3 * nb + bid]), 0.f);
      const float ey = fmaxf(fmaxf(box[nb + bid] - cy, cy - box[4 * nb + bid]), 0.f);
      const float ez = fmaxf(fmaxf(box[2 * nb + bid] - cz, cz - box[5 * nb + bid]), 0.f);
      act = sqdist_nofma(ex, ey, ez) < max_r2;
    }
    unsigned long long mask = __builtin_amdgcn_ballot_w64(act);
    while (mask) {
      const int j = __builtin_ctzll(mask);
      mask &= mask - 1;
      const int s = ((s0 + j) << 6) + lane;
      const bool valid = s < n;
      const float4 p = pts[valid ? s : n - 1];
      const float d2 = sqdist_nofma(cx - p.x, cy - p.y, cz - p.z);
      const bool hit = valid && (d2 == 0.f || (d2 >= min_r2 && d2 < max_r2));
      const unsigned long long hm = __builtin_amdgcn_ballot_w64(hit);
      if (hm) {
        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32),
                                                   __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0));
        if (hit) pend[npend + rank] = (unsigned)k_of_key_lo(__float_as_uint(p.w), L);
        const int got = __popcll(hm);
        npend += got; total += got;
        if (npend >= 64) {  // npend < 128 always: at most 63 carried + 64 new
          const unsigned v = bq_sort64(pend[lane], lane);
          best = bq_merge_low64(best, v, lane);
          const unsigned carry = pend[64 + lane];
          npend -= 64;
          if (lane < npend) pend[lane] = carry;
        }
      }
    }
  }
  if (npend > 0) {
    const unsigned v = bq_sort64(lane < npend ? pend[lane] : 0xFFFFFFFFu, lane);
    best = bq_merge_low64(best, v, lane);
  }
  if (total > 0) {  // no hit: the row stays as the caller zeroed it (.cu:38-52)
    const int have = total < nsample ? total : nsample;
    const unsigned first = (unsigned)__builtin_amdgcn_readfirstlane((int)best);
    int *row = idx + ((size_t)scene * m + c) * nsample;
    if (lane < nsample) row[lane] = (int)(lane < have ? best : first);
  }
}

}  // namespace nesie

using namespace nesie;

extern "C" int nesie_ball_query_wrapper(int b, int n, int m, float min_radius,
                                        float max_radius, int nsample,
                                        const float *new_xyz, const float *xyz,
                                        int *idx, void *stream) {
  const char *W = "ball_query_wrapper";
  NESIE_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0, W);
  if (b == 0 || m == 0 || nsample == 0 || n == 0) return NESIE_OK;
  NESIE_REQUIRE(new_xyz && xyz && idx, W);
  NESIE_REQUIRE((long long)n * 3 < (1ll << 31), W);
  const size_t lds = (size_t)(BQ_BLOCK / 64) * BQ_CPW * nsample * sizeof(int);
  if (lds > 64 * 1024) {
    set_error("%s: nsample %d too large for the LDS row buffer", W, nsample);
    return NESIE_ERR_UNSUPPORTED;
  }
  const int per_block = (BQ_BLOCK / 64) * BQ_CPW;
  const long long groups = cdiv(m, per_block);
  NESIE_REQUIRE(groups * b < (1ll << 31), W);
#define BQ(FORM)                                                                                \
  hipLaunchKernelGGL(ball_query_kernel<FORM>, dim3((unsigned)(groups * b)), dim3(BQ_BLOCK), lds, \
                     (hipStream_t)stream, b, n, m, min_radius, max_radius, nsample, new_xyz, xyz, idx)
  if (distance_form() == 1) BQ(1);
  else if (distance_form() == 2) BQ(2);
  else BQ(0);
#undef BQ
  return check_launch(W);
}

// The same operator over the spatial index that nesie_furthest_point_sampling_ws leaves in its
// workspace for THIS xyz (nesie_fps_leaves_index(b, n) != 0): results identical to
// nesie_ball_query_wrapper, ~40x fewer distance evaluations at 40 000 points.
extern "C" int nesie_ball_query_indexed(int b, int n, int m, float min_radius, float max_radius,
                                        int nsample, const float *new_xyz,
                                        const void *fps_workspace, size_t workspace_bytes,
                                        int *idx, void *stream) {
  const char *W = "ball_query_indexed";
  NESIE_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0, W);
  if (b == 0 || m == 0 || nsample == 0 || n == 0) return NESIE_OK;
  NESIE_REQUIRE(new_xyz && fps_workspace && idx, W);
  if (nsample > 64 || !nesie_fps_leaves_index(b, n)) {   // (also: a fused distance form is selected)
    set_error("%s: needs nsample <= 64 and a size for which the FPS kernel leaves its index", W);
    return NESIE_ERR_UNSUPPORTED;
  }
  NESIE_REQUIRE(workspace_bytes >= nesie_fps_workspace_bytes(b, n), W);
  NESIE_REQUIRE(((uintptr_t)fps_workspace & 15) == 0, W);
  const float4 *pts = (const float4 *)fps_workspace;
  const float *boxes = (const float *)((const char *)fps_workspace + (size_t)b * n * 16);
  const long long groups = cdiv(m, BQI_WAVES);
  NESIE_REQUIRE(groups * b < (1ll << 31), W);
  hipLaunchKernelGGL(ball_query_indexed_kernel, dim3((unsigned)(groups * b)),
                     dim3(BQI_WAVES * 64), 0, (hipStream_t)stream, b, n, m, min_radius,
                     max_radius, nsample, fps_ref_log2_block(n), new_xyz, pts, boxes,
                     (long long)n, idx);
  return check_launch(W);
}
