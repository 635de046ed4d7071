// sort_vertices for gfx950 (rotated-IoU polygon vertex ordering).
//
// Replaces sort_vertices_kernel
// (reference mmdet3d/ops/rotated_iou/cuda_op/sort_vert_kernel.cu:42-134) with the
// comparison of compare_vertices (:15-40) kept term for term.  One thread owns
// one box pair: its 24 candidate vertices and mask bits are pulled into
// registers once (the reference re-reads global memory inside the O(nv * 24)
// selection loops), and the 9 output indices are written with one pass.
// The launch goes to the caller's stream (the reference ignores the stream it
// fetched and uses the legacy default stream, :137-138).
//
// Where the reference is undefined -- compare_vertices falls off its end when
// y1*y2 == 0 in some branches, `pad` is uninitialised when no intersection slot
// is free -- this kernel returns false / uses M-1, as oracle/nesie_oracle.c does.
#include "sort_device.h"

namespace nesie {

__global__ __launch_bounds__(256) void sort_vertices_kernel(
    long long total, int m, const float *__restrict__ vertices,
    const uint8_t *__restrict__ mask, const int *__restrict__ num_valid,
    int *__restrict__ idx) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const float *v = vertices + (size_t)i * m * 2;
  const uint8_t *mk = mask + (size_t)i * m;
  float vx[SV_MAXV], vy[SV_MAXV];
  unsigned mbits = 0;
#pragma unroll
  for (int k = 0; k < SV_MAXV; ++k) {
    if (k < m) {
      vx[k] = v[k * 2 + 0];
      vy[k] = v[k * 2 + 1];
      mbits |= (mk[k] ? 1u : 0u) << k;
    } else {
      vx[k] = 0.f; vy[k] = 0.f;
    }
  }
  const int nv = num_valid[i];
  int o[SV_NIDX];
  sv_sort_one(vx, vy, mbits, nv, m, o);
  int *dst = idx + (size_t)i * SV_NIDX;
#pragma unroll
  for (int j = 0; j < SV_NIDX; ++j) dst[j] = o[j];
}

}  // namespace nesie

using namespace nesie;

extern "C" int nesie_sort_vertices_forward(int b, int n, int m, const float *vertices,
                                           const uint8_t *mask, const int *num_valid,
                                           int *idx, void *stream) {
  const char *W = "sort_vertices_forward";
  NESIE_REQUIRE(b >= 0 && n >= 0 && m >= 0, W);
  const long long total = (long long)b * n;
  if (total == 0) return NESIE_OK;
  NESIE_REQUIRE(vertices && mask && num_valid && idx, W);
  if (m > SV_MAXV || m < SV_OFF + 1) {
    set_error("%s: m = %d vertices per pair, built for 9..24", W, m);
    return NESIE_ERR_UNSUPPORTED;
  }
  NESIE_REQUIRE(total / 256 + 1 < (1ll << 31), W);
  hipLaunchKernelGGL(sort_vertices_kernel, dim3(cdiv(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, total, m, vertices, mask, num_valid, idx);
  return check_launch(W);
}
