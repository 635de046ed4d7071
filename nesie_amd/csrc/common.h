// Shared helpers for the gfx950 kernels of libnesie_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/nesie_ops.h"

namespace nesie {

void set_error(const char *fmt, ...);
int distance_form();   // 0 (default), 1, 2: see sqdist_form
int cu_count();        // CUs the persistent grids are sized for (nesie_set_cu_count; 256)
// 1: a persistent launch over an operand of this size walks its tiles last-to-first (nesie_lib.hip)
int walk_dir(long long operand_bytes);

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return NESIE_ERR_LAUNCH;
  }
  return NESIE_OK;
}

#define NESIE_REQUIRE(cond, what)                                   \
  do {                                                              \
    if (!(cond)) {                                                  \
      nesie::set_error("%s: requirement failed: %s", what, #cond);  \
      return NESIE_ERR_INVALID_ARG;                                 \
    }                                                               \
  } while (0)

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// true when a tensor of this size should be streamed with non-temporal accesses (bn.hip)
// family: 1 = norm passes, 2 = pooling passes, 4 = blend (NESIE_NT_MASK selects; default 3: same-box A/B showed the blend stores neutral)
bool stream_nt(long long bytes, int family);

// Squared distance in the canonical, contraction-free form
// ((dx*dx) + (dy*dy)) + (dz*dz)   (SURVEY.md appendix A.0).
__device__ __forceinline__ float sqdist_nofma(float dx, float dy, float dz) {
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)),
                   __fmul_rn(dz, dz));
}

// The same distance in a selectable form (nesie_set_distance_form): nvcc's default -fmad=true MAY
// contract the reference's `(x2-x1)*(x2-x1) + (y2-y1)*(y2-y1) + (z2-z1)*(z2-z1)`
// (furthest_point_sample_cuda.cu:65-66, ball_query_cuda.cu:41-42, three_nn_cuda.cu:41); whether and how
// cannot be known without a run of the reference's own CUDA build.
//   FORM 0  ((dx*dx) + (dy*dy)) + (dz*dz)        no contraction: the product's form
//   FORM 1  fma(dz, dz, fma(dx, dx, dy*dy))       LLVM's contraction of the expression as written
//   FORM 2  fma(dz, dz, fma(dy, dy, dx*dx))       the other choice for the inner sum
// FORM != 0 is served by the plain (un-pruned, un-indexed) kernels only: a checking aid for the day
// a CUDA-produced fixture exists, not a fast path.  oracle/nesie_oracle.c carries the same switch.
template <int FORM>
__device__ __forceinline__ float sqdist_form(float dx, float dy, float dz) {
  if (FORM == 1) return __fmaf_rn(dz, dz, __fmaf_rn(dx, dx, __fmul_rn(dy, dy)));
  if (FORM == 2) return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
  return sqdist_nofma(dx, dy, dz);
}

// 64-bit max across the 64 lanes of a wave (result valid in every lane).
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    unsigned long long o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

// ---- DPP helpers for reductions inside a row of <= 16 lanes (no LDS, full rate)
template <int CTRL>
__device__ __forceinline__ unsigned dppm(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}

// (value, index) max with smallest-index tie-break, exchanged with the lane given by CTRL
template <int CTRL>
__device__ __forceinline__ void argmax_step(float &v, int &i) {
  const float ov = __uint_as_float(dppm<CTRL>(__float_as_uint(v)));
  const int oi = (int)dppm<CTRL>((unsigned)i);
  const bool take = ov > v || (ov == v && oi < i);
  v = take ? ov : v;
  i = take ? oi : i;
}

// arg-max over the 4 elements a lane holds, then over the LPR lanes of its row (LPR <= 16):
// smallest index on ties, result in every lane of the row
template <int LPR>
__device__ __forceinline__ void row_argmax4(const float4 q, int part, float &v, int &i) {
  v = q.x; i = part * 4;
  if (q.y > v) { v = q.y; i = part * 4 + 1; }
  if (q.z > v) { v = q.z; i = part * 4 + 2; }
  if (q.w > v) { v = q.w; i = part * 4 + 3; }
  if (LPR >= 2) argmax_step<0xB1>(v, i);    // lane ^ 1
  if (LPR >= 4) argmax_step<0x4E>(v, i);    // lane ^ 2
  if (LPR >= 8) argmax_step<0x141>(v, i);   // row_half_mirror: 7 - lane (within 8)
  if (LPR >= 16) argmax_step<0x140>(v, i);  // row_mirror: 15 - lane (within 16)
}


// Streaming accesses.  NT = non-temporal: a tensor larger than the 256 MB Infinity Cache that is
// not read again soon should not displace what is (tools/clk/stream.hip: 6.65 vs 6.0 TB/s for a
// read + write pass over 268 MB).  The launchers choose NT by tensor size (stream_nt).
typedef float f4v __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 ld4(const float *p) {
  if (NT) {
    const f4v t = __builtin_nontemporal_load((const f4v *)p);
    return make_float4(t.x, t.y, t.z, t.w);
  }
  return *(const float4 *)p;
}
template <bool NT>
__device__ __forceinline__ void st4(float *p, const float4 v) {
  if (NT) {
    const f4v t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, (f4v *)p);
  } else {
    *(float4 *)p = v;
  }
}
}  // namespace nesie
