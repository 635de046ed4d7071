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

namespace nesie {

constexpr int BQ_BLOCK = 256;  // 4 waves
constexpr int BQ_CPW = 4;      // centres per wave

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
          float d2 = sqdist_nofma(cx[c] - x, cy[c] - y, cz[c] - z);
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
  hipLaunchKernelGGL(ball_query_kernel, dim3((unsigned)(groups * b)), dim3(BQ_BLOCK),
                     lds, (hipStream_t)stream, b, n, m, min_radius, max_radius,
                     nsample, new_xyz, xyz, idx);
  return check_launch(W);
}
