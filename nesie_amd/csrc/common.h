// Shared helpers for the gfx950 kernels of libnesie_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/nesie_ops.h"

namespace nesie {

void set_error(const char *fmt, ...);

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

// Squared distance in the canonical, contraction-free form
// ((dx*dx) + (dy*dy)) + (dz*dz)   (SURVEY.md appendix A.0).
__device__ __forceinline__ float sqdist_nofma(float dx, float dy, float dz) {
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)),
                   __fmul_rn(dz, dz));
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

}  // namespace nesie
