// Tie-break keys of the reference FPS reduction, shared by fps.hip and ball_query.hip (the
// spatial index fps_pruned_kernel leaves in its workspace stores a point's original index as
// this key).
#pragma once
#include <hip/hip_runtime.h>

namespace nesie {

__host__ __device__ inline int fps_ref_log2_block(int n) {
  // reference opt_n_threads(): largest power of two <= n, capped at 1024.
  int l = 0;
  while ((2 << l) <= n && l < 10) ++l;
  return l;
}

__device__ __forceinline__ unsigned key_lo_of(int k, int L) {
  unsigned t = (unsigned)k & ((1u << L) - 1u);
  unsigned q = (unsigned)k >> L;
  unsigned rb = L == 0 ? 0u : (__brev(t) >> (32 - L));
  return 0xFFFFFFFFu - ((rb << 22) | q);
}

__device__ __forceinline__ int k_of_key_lo(unsigned lo, int L) {
  unsigned v = 0xFFFFFFFFu - lo;
  unsigned rb = v >> 22, q = v & 0x3FFFFFu;
  unsigned t = L == 0 ? 0u : (__brev(rb) >> (32 - L));
  return (int)((q << L) | t);
}


// Spatial index left in the FPS workspace (fps_pruned_kernel, distances-in-LDS mode):
//   float4 pts[b][n]        (x, y, z, key bits) in Morton-cell order, 64 points per bucket
//   float  box[b][6][nb]    at byte offset b*n*16 + scene*n*4: lo.x lo.y lo.z hi.x hi.y hi.z
constexpr int FPS_INDEX_MAX_N = 40192;  // n * 4 bytes of distances must fit the CU's LDS

}  // namespace nesie
