// Per-channel sums over (batch, positions): the gradient of a convolution bias that no norm
// follows -- autograd's `grad.sum((0, 2))` for the output convolutions of ReliableConvBboxHead
// (reference mmdet3d/models/model_utils/reliable_conv_bbox_module.py:144-177), the vote module's
// last convolution (vote_module.py:75-79), the score heads' last convolutions
// (side_pooling_module.py:55-78) and the MiniPointNets' post-pool bias (:357, 361-368).  ATen's
// reduction over the two outer axes of a (B, C, P) tensor takes 13 - 18 us at these sizes (a
// strided two-stage reduce); one 256-thread workgroup per channel streams its B rows densely and
// adds them in a fixed order: 2 - 4 us, bitwise reproducible.
#include "common.h"

namespace nesie {

__global__ __launch_bounds__(256) void channel_sum_kernel(int nb, int ng, int c, int p, long long bstride,
                                                          const float *__restrict__ x,
                                                          float *__restrict__ out) {
  __shared__ float part[4];
  const int ch = blockIdx.x % c, g = blockIdx.x / c, tid = threadIdx.x;
  x += (size_t)g * bstride;                       // the group's batch entries: g, g + ng, ...
  bstride *= ng;
  nb /= ng;
  float s = 0.f;
  const int total = nb * p;                       // element e = (batch e / p, position e % p)
  int e = tid;
  for (; e + 3 * 256 < total; e += 4 * 256) {     // four loads in flight, added in element order
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ee = e + 256 * u;
      v[u] = x[(size_t)(ee / p) * bstride + (size_t)ch * p + ee % p];
    }
    s += (v[0] + v[1]) + (v[2] + v[3]);
  }
  for (; e < total; e += 256) s += x[(size_t)(e / p) * bstride + (size_t)ch * p + e % p];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((tid & 63) == 0) part[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) out[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

}  // namespace nesie

using namespace nesie;

extern "C" int nesie_channel_sum(int nb, int ng, int c, long long p, const float *x, long long x_bstride,
                                 float *out, void *stream) {
  const char *W = "channel_sum";
  NESIE_REQUIRE(nb >= 0 && ng >= 1 && nb % ng == 0 && c >= 0 && p >= 0, W);
  if (c == 0) return NESIE_OK;
  NESIE_REQUIRE(out, W);
  if (nb == 0 || p == 0) {
    (void)hipMemsetAsync(out, 0, (size_t)ng * c * sizeof(float), (hipStream_t)stream);
    return NESIE_OK;
  }
  NESIE_REQUIRE(x && (long long)nb * p < (1ll << 31) && p < (1ll << 31) && x_bstride >= (long long)c * p, W);
  hipLaunchKernelGGL(channel_sum_kernel, dim3(ng * c), dim3(256), 0, (hipStream_t)stream, nb, ng, c, (int)p,
                     x_bstride, x, out);
  return check_launch(W);
}
