// Weight gradient and the streaming (skinny) forward of the 1x1 convolutions of the grouped
// MLPs on the fp32 matrix cores of gfx950 (mmcv ConvModule(Conv2d 1x1, BN2d, ReLU):
// mmdet3d/ops/pointnet_modules/point_sa_module.py:277-289; dense_heads/
// side_pooling_module.py:346-358).  The layer kernel proper is pwconv.hip.
#include "common.h"
#include <stdlib.h>

namespace nesie {

typedef float f32x16 __attribute__((ext_vector_type(16)));


// sum over the 32 lanes of each half-wave (result in every lane of the half)
__device__ __forceinline__ float half_wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  v += __shfl_xor(v, 16, 64);
  return v;
}

}  // namespace nesie

using namespace nesie;

// (sum, sum of squares) partials of the streaming layer kernel -> mean / invstd, running
// statistics, folded (scale, bias).  The streaming kernel serves first layers only (their input
// is centred geometry), which keeps the unshifted sums harmless; the layer kernel proper
// (pwconv.hip) carries shifted sums.
namespace nesie {
__global__ void mlp_stat_finalize_kernel(int c_total, int nparts, double n,
                                         const float *__restrict__ partial,
                                         const float *gamma, const float *beta,
                                         float *running_mean, float *running_var, float momentum,
                                         float eps, float *coef, int channel_major) {
  __shared__ double sh[2][256];
  const int c = blockIdx.x;
  double s0 = 0.0, s1 = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) {
    const size_t o = channel_major ? ((size_t)c * nparts + i) * 2 : ((size_t)i * c_total + c) * 2;
    s0 += (double)partial[o];
    s1 += (double)partial[o + 1];
  }
  sh[0][threadIdx.x] = s0;
  sh[1][threadIdx.x] = s1;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + off];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const double mean = sh[0][0] / n;
  double var = sh[1][0] / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  if (running_mean) {
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
    const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
  }
  const double g = gamma ? (double)gamma[c] : 1.0, bt = beta ? (double)beta[c] : 0.0;
  coef[c * 4 + 0] = (float)(g * invstd);
  coef[c * 4 + 1] = (float)(bt - mean * g * invstd);
  coef[c * 4 + 2] = (float)mean;
  coef[c * 4 + 3] = (float)invstd;
}
}  // namespace nesie

extern "C" int nesie_mlp_stat_finalize(int c, long long nparts, double count,
                                       const float *stat_partial, const float *gamma,
                                       const float *beta, float *running_mean,
                                       float *running_var, float momentum, float eps,
                                       float *coef, int channel_major, void *stream) {
  const char *W = "mlp_stat_finalize";
  NESIE_REQUIRE(c >= 1 && nparts >= 1 && count >= 1.0 && stat_partial && coef, W);
  NESIE_REQUIRE((running_mean == nullptr) == (running_var == nullptr), W);
  hipLaunchKernelGGL(nesie::mlp_stat_finalize_kernel, dim3(c), dim3(256), 0, (hipStream_t)stream, c,
                     (int)nparts, count, stat_partial, gamma, beta, running_mean, running_var,
                     momentum, eps, coef, channel_major);
  return check_launch(W);
}

// ---- weight gradient of a 1x1 conv: dW[m][k] = sum_{b,p} dy[b][m][p] * x[b][k][p] -------------
// Both operands are (channels x positions) row-major with the reduction axis contiguous, the
// output is tiny (Cout x Cin) and the reduction is over B*P >= 10^5 positions: a workgroup owns
// a run of positions of one scene, stages (Cout + Cin) x 64-position tiles in LDS (row pitch 65:
// the 32 rows a half-wave reads sit on 32 banks), accumulates the WHOLE Cout x Cin product on
// the matrix cores and leaves one partial per workgroup; a second kernel adds the partials in a
// fixed order (bitwise reproducible).  rocBLAS reaches these shapes through per-scene split-K
// batched GEMMs at 30-70 TFLOP/s; the SA1/SA2 layers are HBM-bound here.
//   optional x_coef [cin][4] = (scale, bias, -, -): the operand is relu?(scale * x + bias)
//   recomputed on load -- the activation a fused forward never stored (mlp_fwd_kernel).
namespace nesie {

constexpr int WG_Q = 64;  // positions per LDS tile
// row pitch 68 floats: rows stay 16-byte aligned (one ds_write_b128 per staged float4, one
// ds_read_b128 per four positions of an MFMA operand row) and the 8 rows a quarter-wave reads
// start 4 banks apart, i.e. cover the 32 banks exactly
constexpr int WG_P = WG_Q + 4;

template <int WM, int WN, int MB, int NB>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(
    int cout, int cin, long long p, long long x_bstride, long long dy_bstride, int run, int vec_ok,
    const float *__restrict__ dy, const float *__restrict__ x,
    const float *__restrict__ x_coef, int x_relu, float *__restrict__ partial,
    const float *__restrict__ bnz, const float *__restrict__ bnb) {
  // bnb != NULL: `dy` is the gradient dA of relu(bn(Z)), Z = `bnz` (same layout); the norm backward's
  // apply pass runs on the load, dZ = a g + (e0 - (z - mean) d1), g = dA [fma(z, scale, shift) > 0],
  // bnb[row] = (scale, shift, a, mean, d1, e0, -, -) (pw_bnb_coef_kernel) -- dZ is never written
  // (nobody else reads it when the layer's input needs no gradient: the first layer of SA1).
  constexpr int MT = WM * MB * 32, NT = WN * NB * 32;  // padded Cout, Cin covered
  extern __shared__ float lds[];                      // [MT + NT][WG_P]
  float *sa = lds, *sb = lds + MT * WG_P;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int bi = blockIdx.y;
  const long long p0 = (long long)blockIdx.x * run;
  const long long p1 = p0 + run < p ? p0 + run : p;
  const float *dyb = dy + (size_t)bi * dy_bstride;
  const float *zb = bnb ? bnz + (size_t)bi * dy_bstride : nullptr;
  const float *xb = x + (size_t)bi * x_bstride;
  auto norm_bwd = [&](float4 g, float4 z, int row) {
    const float *cf = bnb + (size_t)row * 8;
    const float sc = cf[0], bs = cf[1], a = cf[2], mu = cf[3], d1 = cf[4], e0 = cf[5];
    float4 r;
    r.x = __builtin_fmaf(a, __builtin_fmaf(z.x, sc, bs) > 0.f ? g.x : 0.f, __builtin_fmaf(mu - z.x, d1, e0));
    r.y = __builtin_fmaf(a, __builtin_fmaf(z.y, sc, bs) > 0.f ? g.y : 0.f, __builtin_fmaf(mu - z.y, d1, e0));
    r.z = __builtin_fmaf(a, __builtin_fmaf(z.z, sc, bs) > 0.f ? g.z : 0.f, __builtin_fmaf(mu - z.z, d1, e0));
    r.w = __builtin_fmaf(a, __builtin_fmaf(z.w, sc, bs) > 0.f ? g.w : 0.f, __builtin_fmaf(mu - z.w, d1, e0));
    return r;
  };
  f32x16 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int half = lane >> 5, l32 = lane & 31;
  // rows of 64 positions: 16 lanes x float4 per row, 16 rows per pass of the workgroup.  The
  // loads of tile t+1 are issued before the MFMAs of tile t and land in LDS after them.
  constexpr int ROWS = MT + NT, PASSES = ROWS / 16;
  const bool vec = vec_ok != 0;  // float4 rows: p, batch stride and both bases 16-byte aligned
  float4 v[PASSES];
#pragma unroll
  for (int u = 0; u < PASSES; ++u) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load_tile = [&](long long q0) {
    const long long q = q0 + (tid & 15) * 4;
    if (vec && q0 + WG_Q <= p1) {
      // interior tile (a wave-uniform test): PASSES unconditional 16-byte loads issued back to
      // back -- rows past Cout / Cin read a valid row and are zeroed by a select, so that no
      // branch (and no wait) separates the loads
#pragma unroll
      for (int u = 0; u < PASSES; ++u) {
        const bool is_a = u * 16 < MT;                      // compile-time per pass
        // a pass whose 16 rows all lie beyond Cin (Cin = 4 padded to 64 rows: three of four X
        // passes) loads nothing: its LDS rows were zeroed once (a wave-uniform test)
        if (!is_a && u * 16 - MT >= cin) continue;
        const int row = u * 16 + (tid >> 4) - (is_a ? 0 : MT);
        const int lim = is_a ? cout : cin;
        const int rr = row < lim ? row : lim - 1;
        float4 t = *(const float4 *)((is_a ? dyb : xb) + (size_t)rr * p + q);
        if (is_a && bnb) t = norm_bwd(t, *(const float4 *)(zb + (size_t)rr * p + q), rr);   // (uniform test)
        v[u] = row < lim ? t : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (x_coef) {
#pragma unroll
        for (int u = MT / 16; u < PASSES; ++u) {
          const int row = u * 16 + (tid >> 4) - MT;
          if (row < cin) {
            const float sc = x_coef[row * 4 + 0], bs = x_coef[row * 4 + 1];
            v[u].x = v[u].x * sc + bs; v[u].y = v[u].y * sc + bs;
            v[u].z = v[u].z * sc + bs; v[u].w = v[u].w * sc + bs;
            if (x_relu) {
              v[u].x = fmaxf(v[u].x, 0.f); v[u].y = fmaxf(v[u].y, 0.f);
              v[u].z = fmaxf(v[u].z, 0.f); v[u].w = fmaxf(v[u].w, 0.f);
            }
          }
        }
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < PASSES; ++u) {
      const int r = u * 16 + (tid >> 4);
      const bool is_a = r < MT;
      const int row = is_a ? r : r - MT;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < (is_a ? cout : cin)) {
        const float *src = (is_a ? dyb : xb) + (size_t)row * p + q;
        float4 zz = make_float4(0.f, 0.f, 0.f, 0.f);
        const float *zs = (is_a && bnb) ? zb + (size_t)row * p + q : nullptr;
        if (vec && q + 3 < p1) {
          v[u] = *(const float4 *)src;
          if (zs) zz = *(const float4 *)zs;
        } else {
          if (q < p1) { v[u].x = src[0]; if (zs) zz.x = zs[0]; }
          if (q + 1 < p1) { v[u].y = src[1]; if (zs) zz.y = zs[1]; }
          if (q + 2 < p1) { v[u].z = src[2]; if (zs) zz.z = zs[2]; }
          if (q + 3 < p1) { v[u].w = src[3]; if (zs) zz.w = zs[3]; }
        }
        if (zs) {
          v[u] = norm_bwd(v[u], zz, row);
          if (q >= p1) v[u].x = 0.f;      // (positions past the run: dZ of a zero pair is e0-ish, not zero)
          if (q + 1 >= p1) v[u].y = 0.f;
          if (q + 2 >= p1) v[u].z = 0.f;
          if (q + 3 >= p1) v[u].w = 0.f;
        }
        if (!is_a && x_coef) {
          const float sc = x_coef[row * 4 + 0], bs = x_coef[row * 4 + 1];
          v[u].x = v[u].x * sc + bs; v[u].y = v[u].y * sc + bs;
          v[u].z = v[u].z * sc + bs; v[u].w = v[u].w * sc + bs;
          if (x_relu) {
            v[u].x = fmaxf(v[u].x, 0.f); v[u].y = fmaxf(v[u].y, 0.f);
            v[u].z = fmaxf(v[u].z, 0.f); v[u].w = fmaxf(v[u].w, 0.f);
          }
          if (q >= p1) v[u].x = 0.f;
          if (q + 1 >= p1) v[u].y = 0.f;
          if (q + 2 >= p1) v[u].z = 0.f;
          if (q + 3 >= p1) v[u].w = 0.f;
        }
      }
    }
  };
  if (p0 < p1) load_tile(p0);
  for (long long q0 = p0; q0 < p1; q0 += WG_Q) {
    __syncthreads();  // the previous tile's MFMA operand reads are done
#pragma unroll
    for (int u = 0; u < PASSES; ++u) {
      const int r = u * 16 + (tid >> 4);
      if (u * 16 >= MT && u * 16 - MT >= cin && q0 != p0) continue;     // (stays zero: written by the first tile)
      float *dst = (r < MT ? sa + r * WG_P : sb + (r - MT) * WG_P) + (tid & 15) * 4;
      *(float4 *)dst = v[u];
    }
    __syncthreads();
    if (q0 + WG_Q < p1) load_tile(q0 + WG_Q);
    const float *pa = sa + (wm * MB * 32 + l32) * WG_P;
    const float *pb = sb + (wn * NB * 32 + l32) * WG_P;
    // four positions per LDS read: lanes 0-31 feed k = 4j, 4j+2, lanes 32-63 k = 4j+1, 4j+3
#pragma unroll 4
    for (int k4 = 0; k4 < WG_Q / 4; ++k4) {
      float4 a[MB], b[NB];
#pragma unroll
      for (int i = 0; i < MB; ++i) a[i] = *(const float4 *)(pa + i * 32 * WG_P + k4 * 4);
#pragma unroll
      for (int j = 0; j < NB; ++j) b[j] = *(const float4 *)(pb + j * 32 * WG_P + k4 * 4);
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? a[i].y : a[i].x,
                                                           half ? b[j].y : b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? a[i].w : a[i].z,
                                                           half ? b[j].w : b[j].z, acc[i][j], 0, 0, 0);
        }
    }
  }
  // partial[(b * runs + run)][cout][cin]
  float *dst = partial + ((size_t)bi * gridDim.x + blockIdx.x) * cout * cin;
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (wm * MB + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int k = (wn * NB + j) * 32 + l32;
        if (m < cout && k < cin) dst[(size_t)m * cin + k] = acc[i][j][r];
      }
}

// dw[i] = sum over the partials, in a fixed order: 64 consecutive outputs per workgroup (dense
// 256-byte reads), 16 waves striding the partials with 8 independent loads in flight each.
__global__ __launch_bounds__(1024) void conv_wgrad_reduce_kernel(int total, int nparts,
                                                                 const float *__restrict__ partial,
                                                                 float *__restrict__ dw) {
  __shared__ double sh[16][64];      // (double: see pw_wgrad_reduce_kernel)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  double s = 0.0;
  if (i < total) {
    int r = wave;
    for (; r + 7 * 16 < nparts; r += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(r + u * 16) * total + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; r < nparts; r += 16) s += (double)partial[(size_t)r * total + i];
  }
  sh[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && i < total) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += sh[w][lane];
    dw[i] = (float)t;
  }
}

static int wgrad_runs(int b, int cout, int cin, long long p, int *run_len) {
  // ~1024 workgroups, but no more partial bytes than ~32 MB
  long long want = 1024 / (b > 0 ? b : 1);
  const long long cap = (32ll << 20) / ((long long)cout * cin * 4) / (b > 0 ? b : 1);
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  long long run = (p + want - 1) / want;
  run = (run + WG_Q - 1) / WG_Q * WG_Q;
  if (run < WG_Q) run = WG_Q;
  *run_len = (int)run;
  return (int)((p + run - 1) / run);
}

}  // namespace nesie

extern "C" size_t nesie_conv_wgrad_workspace_bytes(int b, int cout, int cin, long long p) {
  if (b <= 0 || cout <= 0 || cin <= 0 || p <= 0) return 0;
  int run;
  const int runs = wgrad_runs(b, cout, cin, p, &run);
  return (size_t)b * runs * cout * cin * sizeof(float);
}

static int conv_wgrad_impl(const char *W, int b, int cout, int cin, long long p, const float *dy,
                           long long dy_bstride, const float *x, long long x_bstride,
                           const float *x_coef, int x_relu, float *dw, void *workspace,
                           size_t workspace_bytes, const float *bnz, const float *bnb, void *stream) {
  NESIE_REQUIRE(b >= 0 && cout >= 1 && cin >= 1 && p >= 0, W);
  NESIE_REQUIRE(dw, W);
  hipStream_t s = (hipStream_t)stream;
  if (b == 0 || p == 0) {
    (void)hipMemsetAsync(dw, 0, (size_t)cout * cin * sizeof(float), s);
    return NESIE_OK;
  }
  NESIE_REQUIRE(dy && x && workspace && x_bstride >= (long long)cin * p && b <= 65535, W);
  NESIE_REQUIRE(dy_bstride >= (long long)cout * p, W);
  NESIE_REQUIRE(workspace_bytes >= nesie_conv_wgrad_workspace_bytes(b, cout, cin, p), W);
  const int vec_ok = (((uintptr_t)dy | (uintptr_t)x) & 15) == 0 && (p & 3) == 0 && (x_bstride & 3) == 0 &&
                     (dy_bstride & 3) == 0;
  const int mb32 = cdiv(cout, 32), nb32 = cdiv(cin, 32);
  int run;
  const int runs = wgrad_runs(b, cout, cin, p, &run);
  float *partial = (float *)workspace;
  const dim3 grid(runs, b);
#define L(WM, WN, MB, NB)                                                                        \
  do {                                                                                           \
    const size_t lds = (size_t)(WM * MB + WN * NB) * 32 * WG_P * sizeof(float);                  \
    auto kern = conv_wgrad_kernel<WM, WN, MB, NB>;                                               \
    static bool attr = false;                                                                    \
    if (!attr && lds > 65536) {                                                                  \
      (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                (int)lds);                                                       \
      attr = true;                                                                               \
    }                                                                                            \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, cout, cin, p, x_bstride, dy_bstride, run,  \
                       vec_ok, dy, x, x_coef, x_relu, partial, bnz, bnb);                        \
  } while (0)
  // (Cout/32) x (Cin/32) output blocks over 4 waves
  if (mb32 <= 2 && nb32 <= 2) L(2, 2, 1, 1);
  else if (mb32 <= 2 && nb32 <= 4) L(2, 2, 1, 2);
  else if (mb32 <= 4 && nb32 <= 1) L(4, 1, 1, 1);
  else if (mb32 <= 4 && nb32 <= 2) L(4, 1, 1, 2);
  else if (mb32 <= 4 && nb32 <= 4) L(4, 1, 1, 4);
  else if (mb32 <= 4 && nb32 <= 5) L(4, 1, 1, 5);
  else if (mb32 <= 4 && nb32 <= 8) L(4, 1, 1, 8);
  else if (mb32 <= 4 && nb32 <= 9) L(4, 1, 1, 9);
  else if (mb32 <= 8 && nb32 <= 2) L(4, 1, 2, 2);
  else if (mb32 <= 8 && nb32 <= 4) L(4, 1, 2, 4);
  else {
    set_error("%s: %d x %d output (built for Cout <= 256 with Cin <= 128, Cout <= 128 with Cin <= 288)", W, cout, cin);
    return NESIE_ERR_UNSUPPORTED;
  }
#undef L
  const int total = cout * cin;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(cdiv(total, 64)), dim3(1024), 0, s, total,
                     b * runs, partial, dw);
  return check_launch(W);
}

extern "C" int nesie_conv_wgrad(int b, int cout, int cin, long long p, const float *dy,
                                long long dy_bstride, const float *x, long long x_bstride,
                                const float *x_coef,
                                int x_relu, float *dw, void *workspace, size_t workspace_bytes,
                                void *stream) {
  return conv_wgrad_impl("conv_wgrad", b, cout, cin, p, dy, dy_bstride, x, x_bstride, x_coef, x_relu, dw,
                         workspace, workspace_bytes, nullptr, nullptr, stream);
}

extern "C" int nesie_conv_wgrad_bn(int b, int cout, int cin, long long p, const float *da, const float *z,
                                   long long z_bstride, const float *bnb, const float *x,
                                   long long x_bstride, const float *x_coef, int x_relu, float *dw,
                                   void *workspace, size_t workspace_bytes, void *stream) {
  const char *W = "conv_wgrad_bn";
  NESIE_REQUIRE(b == 0 || p == 0 || (z && bnb), W);
  return conv_wgrad_impl(W, b, cout, cin, p, da, z_bstride, x, x_bstride, x_coef, x_relu, dw, workspace,
                         workspace_bytes, z, bnb, stream);
}

// ---- streaming form of the layer kernel for skinny layers (Cin <= 64) --------------------------
// When W is tiny and the layer is HBM-bound (SA1: 4->64, 64->64, 64->128 over 10^6 positions)
// the LDS tile + barrier per K step of mlp_fwd_kernel is pure overhead.  Here W (k-major) sits
// in LDS once per workgroup, and every WAVE streams its own 32-position columns: the MFMA B
// operand (lane = (k & 1) * 32 + position) is loaded straight from global memory into its
// register -- two dense 128-byte row segments per instruction -- so X never touches LDS and
// the main loop has no barrier.  The loads of the next column block are in flight during the
// MFMAs of the current one.
namespace nesie {

template <int MB, int KP>  // MB = Cout blocks of 32 (padded), KP = Cin pairs (padded Cin / 2)
__global__ __launch_bounds__(256) void mlp_stream_kernel(
    int cin, int cout, long long p, long long x_bstride, int cols_per_wg,
    const float *__restrict__ w, const float *__restrict__ x,
    const float *__restrict__ in_coef, int in_relu, float *__restrict__ y,
    float *__restrict__ stat_partial) {
  constexpr int LDW = MB * 32 + 32;
  __shared__ float ws[2 * KP * LDW];   // [k][m]
  __shared__ float scs[2 * KP], bis[2 * KP];
  __shared__ float sstat[MB * 32][2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: row bases stay scalar
  const int half = lane >> 5, l32 = lane & 31;
  const int bi = blockIdx.y;
  for (int e = tid; e < 2 * KP * MB * 32; e += 256) {
    const int k = e / (MB * 32), m = e % (MB * 32);
    ws[k * LDW + m] = (k < cin && m < cout) ? w[(size_t)m * cin + k] : 0.f;
  }
  for (int k = tid; k < 2 * KP; k += 256) {
    scs[k] = (in_coef && k < cin) ? in_coef[k * 4 + 0] : 1.f;
    bis[k] = (in_coef && k < cin) ? in_coef[k * 4 + 1] : 0.f;
  }
  if (stat_partial)
    for (int i = tid; i < MB * 32 * 2; i += 256) (&sstat[0][0])[i] = 0.f;
  __syncthreads();
  const float *xb = x + (size_t)bi * x_bstride;
  float *yb = y + (size_t)bi * cout * p;
  const long long c0 = (long long)blockIdx.x * cols_per_wg;
  const long long c1 = c0 + cols_per_wg < p ? c0 + cols_per_wg : p;
  // statistics: sum(y^2) per accumulator element; sum(y) = W . (column sums of the operand),
  // so only the operand's per-lane sums are carried (KP registers instead of 16 * MB)
  float qrow[MB][16], ax[KP];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) qrow[i][r] = 0.f;
#pragma unroll
  for (int kp = 0; kp < KP; ++kp) ax[kp] = 0.f;
  float xn[KP];
  // per-lane 32-bit offsets on top of wave-uniform row bases (one address register per access
  // instead of a 64-bit pointer per row)
  const unsigned ld_off = (unsigned)(half * p + l32);
  const unsigned st_off = (unsigned)(4 * half * p + l32);
  auto load_cols = [&](long long n0) {
    const bool ok = n0 + l32 < c1;
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) {
      const float *row = xb + (size_t)(kp * 2) * p + n0;   // uniform
      xn[kp] = (kp * 2 + half < cin && ok) ? row[ld_off] : 0.f;
    }
  };
  long long n0 = c0 + wave * 32;
  if (n0 < c1) load_cols(n0);
  for (; n0 < c1; n0 += 4 * 32) {
    const bool in_range = n0 + l32 < c1;
    if (in_coef) {
#pragma unroll
      for (int kp = 0; kp < KP; ++kp) {
        const int k = kp * 2 + half;
        float v = xn[kp] * scs[k] + bis[k];
        if (in_relu) v = fmaxf(v, 0.f);
        xn[kp] = (!in_range || k >= cin) ? 0.f : v;
      }
    }
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) ax[kp] += xn[kp];
    // one 32-row block of outputs at a time: a single 16-register accumulator chain
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int kp = 0; kp < KP; ++kp) {
        const float a = ws[(kp * 2 + half) * LDW + i * 32 + l32];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xn[kp], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mu = i * 32 + (r & 3) + 8 * (r >> 2);       // uniform part of the row
        const float v = acc[r];
        float *row = yb + (size_t)mu * p + n0;                // uniform
        if (y && mu + 4 * half < cout && in_range) row[st_off] = v;      // (y null: statistics only)
        qrow[i][r] += v * v;      // out-of-range columns are exact zeros
      }
    }
    if (n0 + 4 * 32 < c1) load_cols(n0 + 4 * 32);
  }
  if (stat_partial) {
    // per-wave slots, then a fixed-order sum over the four waves: the statistics (and with
    // them the whole forward pass) are bitwise reproducible from run to run
    __shared__ float sq_w[4][MB * 32];
    __shared__ float cs_w[4][2 * KP];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float q = half_wave_sum(qrow[i][r]);
        if (l32 == 0) sq_w[wave][m] = q;
      }
    // column sums of the operand per k, then sum(y)[m] = sum_k W[m][k] * colsum[k]
    __shared__ float colsum[2 * KP];
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) {
      const float t = half_wave_sum(ax[kp]);
      if (l32 == 0) cs_w[wave][kp * 2 + half] = t;
    }
    __syncthreads();
    if (tid < 2 * KP) colsum[tid] = (cs_w[0][tid] + cs_w[1][tid]) + (cs_w[2][tid] + cs_w[3][tid]);
    __syncthreads();
    for (int m = tid; m < cout; m += 256) {
      float t = 0.f;
      for (int k = 0; k < cin; ++k) t += ws[k * LDW + m] * colsum[k];
      sstat[m][0] = t;
      sstat[m][1] = (sq_w[0][m] + sq_w[1][m]) + (sq_w[2][m] + sq_w[3][m]);
    }
    __syncthreads();
    float *dst = stat_partial + (((size_t)bi * gridDim.x + blockIdx.x) * cout) * 2;
    for (int i = tid; i < cout * 2; i += 256) dst[i] = (&sstat[0][0])[i];
  }
}


// ---- SA1's first layer without its output tensor: the weight gradient from reductions -----------
// With Z0 = W0 . X4 rebuilt wherever it is needed (pwconv_fwd.h, PW_K4IN / PW_K4Z) the gradient of
// the first activation is never stored either: dZ0 = a g + e0 + (mu - Z0) d1 per channel (the norm
// backward's coefficients, pw_bnb_coef), so
//   dW0[c][j] = sum_pos dZ0[c] X4[j] = a G[c][j] + e0 Sx[j] + d1 (mu Sx[j] - sum_k W0[c][k] M[k][j])
// with G[c][j] = sum g[c] X4[j] (left per slot by the input-gradient launch), Sx[j] = sum X4[j] and
// M = X4 X4^T (4 x 4) -- the moments of the INPUT, 20 sums over 17 MB.  Everything is added in
// double in a fixed order.
constexpr int K4_MOM_WGS = 256;

__global__ __launch_bounds__(256) void k4_moments_kernel(int nb, long long p, long long x_bstride,
                                                         const float *__restrict__ x4, double *__restrict__ part) {
  const long long chunks = (long long)nb * (p / 4);
  double s[4], m[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s[j] = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) m[j][k] = 0.0;
  }
  for (long long c = (long long)blockIdx.x * 256 + threadIdx.x; c < chunks; c += (long long)K4_MOM_WGS * 256) {
    const long long n = c / (p / 4), q = c % (p / 4);
    const float *xb = x4 + (size_t)n * x_bstride + 4 * q;
    float v[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 t = *(const float4 *)(xb + (size_t)j * p);
      v[j][0] = t.x; v[j][1] = t.y; v[j][2] = t.z; v[j][3] = t.w;
    }
    float fs[4], fm[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      fs[j] = (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
#pragma unroll
      for (int k = j; k < 4; ++k)
        fm[j][k] = __builtin_fmaf(v[j][3], v[k][3], __builtin_fmaf(v[j][2], v[k][2], __builtin_fmaf(v[j][1], v[k][1], v[j][0] * v[k][0])));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[j] += (double)fs[j];
#pragma unroll
      for (int k = j; k < 4; ++k) m[j][k] += (double)fm[j][k];
    }
  }
  __shared__ double red[4][20];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double all[20];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    all[j] = s[j];
#pragma unroll
    for (int k = 0; k < 4; ++k) all[4 + 4 * j + k] = k >= j ? m[j][k] : 0.0;
  }
#pragma unroll
  for (int u = 0; u < 20; ++u) {
    double t = all[u];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    if (lane == 0) red[wave][u] = t;
  }
  __syncthreads();
  if (threadIdx.x < 20) {
    const int u = threadIdx.x;
    part[(size_t)blockIdx.x * 20 + u] = (red[0][u] + red[1][u]) + (red[2][u] + red[3][u]);
  }
}

// one 256-thread workgroup per channel: thread t folds moment partial t, wave 0 the channel's slots
__global__ __launch_bounds__(256) void k4_wgrad_finish_kernel(int nslots, const double *__restrict__ mom_part,
                                                              const float *__restrict__ g_part,
                                                              const float *__restrict__ bnb,
                                                              const float *__restrict__ w0, float *__restrict__ dw) {
  static_assert(K4_MOM_WGS == 256, "one partial per thread");
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ double red[4][20];
  __shared__ double gs[4];
  double m[20];
#pragma unroll
  for (int u = 0; u < 20; ++u) m[u] = mom_part[(size_t)tid * 20 + u];
  double g[4] = {0.0, 0.0, 0.0, 0.0};
  if (wave == 0) {
    const float4 *gp = (const float4 *)g_part + (size_t)c * nslots;
    for (int i = lane; i < nslots; i += 64) {
      const float4 q = gp[i];
      g[0] += (double)q.x; g[1] += (double)q.y; g[2] += (double)q.z; g[3] += (double)q.w;
    }
  }
#pragma unroll
  for (int u = 0; u < 20; ++u) {
    double t = m[u];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    if (lane == 0) red[wave][u] = t;
  }
  if (wave == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double t = g[j];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
      if (lane == 0) gs[j] = t;
    }
  }
  __syncthreads();
  if (tid < 4) {
    const int j = tid;
    auto mom = [&](int u) { return (red[0][u] + red[1][u]) + (red[2][u] + red[3][u]); };
    const double a = bnb[c * 8 + 2], mu = bnb[c * 8 + 3], d1 = bnb[c * 8 + 4], e0 = bnb[c * 8 + 5];
    double zx = 0.0;
    for (int k = 0; k < 4; ++k) {
      const double mkj = k <= j ? mom(4 + 4 * k + j) : mom(4 + 4 * j + k);   // (upper triangle stored)
      zx += (double)w0[c * 4 + k] * mkj;
    }
    const double sx = mom(j);
    dw[c * 4 + j] = (float)(a * gs[j] + e0 * sx + d1 * (mu * sx - zx));
  }
}

// The first layer's BatchNorm statistics from the same moments: mean(Z0[c]) = W0[c] . mean(X4),
// var(Z0[c]) = W0[c]^T Cov(X4) W0[c] (the covariance formed first, in double) -- no pass over Z0.
__global__ __launch_bounds__(256) void k4_stat_finalize_kernel(double n, const double *__restrict__ mom_part,
                                                               const float *__restrict__ w0,
                                                               const float *gamma, const float *beta,
                                                               float *running_mean, float *running_var,
                                                               float momentum, float eps, float *__restrict__ coef) {
  static_assert(K4_MOM_WGS == 256, "one partial per thread");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ double red[4][20];
  double m[20];
#pragma unroll
  for (int u = 0; u < 20; ++u) m[u] = mom_part[(size_t)tid * 20 + u];
#pragma unroll
  for (int u = 0; u < 20; ++u) {
    double t = m[u];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    if (lane == 0) red[wave][u] = t;
  }
  __syncthreads();
  if (tid >= 64) return;
  const int c = tid;
  auto mom = [&](int u) { return (red[0][u] + red[1][u]) + (red[2][u] + red[3][u]); };
  double mu[4], cov[4][4];
  for (int j = 0; j < 4; ++j) mu[j] = mom(j) / n;
  for (int j = 0; j < 4; ++j)
    for (int k = j; k < 4; ++k) cov[j][k] = cov[k][j] = mom(4 + 4 * j + k) / n - mu[j] * mu[k];
  double mean = 0.0, var = 0.0;
  for (int j = 0; j < 4; ++j) {
    const double wj = (double)w0[c * 4 + j];
    mean += wj * mu[j];
    for (int k = 0; k < 4; ++k) var += wj * (double)w0[c * 4 + k] * cov[j][k];
  }
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  if (running_mean) {
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
    const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
  }
  const double g = gamma ? (double)gamma[c] : 1.0, bt = beta ? (double)beta[c] : 0.0;
  coef[c * 4 + 0] = (float)(g * invstd);
  coef[c * 4 + 1] = (float)(bt - mean * g * invstd);
  coef[c * 4 + 2] = (float)mean;
  coef[c * 4 + 3] = (float)invstd;
}

static int stream_cols(int b, long long p) {
  // ~2048 workgroups of whole 128-column groups
  long long per = (p * b + 2047) / 2048;
  per = (per + 127) / 128 * 128;
  if (per < 128) per = 128;
  return (int)per;
}

}  // namespace nesie

extern "C" long long nesie_mlp_stream_partials(int b, long long p) {
  return (long long)b * cdiv(p, stream_cols(b, p));
}

extern "C" int nesie_mlp_layer_forward_stream(int b, int cin, int cout, long long p,
                                              const float *x, long long x_bstride,
                                              const float *w, const float *in_coef,
                                              int in_relu, float *y, float *stat_partial,
                                              void *stream) {
  const char *W = "mlp_layer_forward_stream";
  NESIE_REQUIRE(b >= 0 && cin >= 1 && cout >= 1 && p >= 0, W);
  if (b == 0 || p == 0) return NESIE_OK;
  NESIE_REQUIRE(x && w && (y || stat_partial) && x_bstride >= (long long)cin * p && b <= 65535, W);
  NESIE_REQUIRE(p < (1ll << 28), W);
  if (cin > 64 || cout > 128) {
    set_error("%s: %d -> %d (built for Cin <= 64, Cout <= 128)", W, cin, cout);
    return NESIE_ERR_UNSUPPORTED;
  }
  const int cols = stream_cols(b, p);
  const dim3 grid(cdiv(p, cols), b);
  hipStream_t s = (hipStream_t)stream;
  const int mb = cdiv(cout, 32), kp = cdiv(cin, 2);
#define L(MB, KP)                                                                              \
  hipLaunchKernelGGL((mlp_stream_kernel<MB, KP>), grid, dim3(256), 0, s, cin, cout, p,         \
                     x_bstride, cols, w, x, in_coef, in_relu, y, stat_partial)
  if (mb <= 2 && kp <= 2) L(2, 2);
  else if (mb <= 2 && kp <= 32) L(2, 32);
  else if (mb <= 4 && kp <= 32) L(4, 32);
  else {
    set_error("%s: no build for %d -> %d", W, cin, cout);
    return NESIE_ERR_UNSUPPORTED;
  }
#undef L
  return check_launch(W);
}

extern "C" size_t nesie_k4_moments_bytes(void) { return (size_t)nesie::K4_MOM_WGS * 20 * sizeof(double); }

// First and second moments of x4 (nb, 4, p) as per-workgroup partial sums in double:
// mom_part [256][20] = (sum X4[j], j = 0 .. 3; sum X4[j] X4[k], k >= j at 4 + 4 j + k).
extern "C" int nesie_k4_moments(int nb, long long p, const float *x4, long long x4_bstride, void *mom_part,
                                void *stream) {
  const char *W = "k4_moments";
  NESIE_REQUIRE(nb >= 1 && p >= 4 && p % 4 == 0 && x4 && mom_part, W);
  NESIE_REQUIRE(x4_bstride >= 4 * p && (x4_bstride & 3) == 0 && ((uintptr_t)x4 & 15) == 0 && ((uintptr_t)mom_part & 7) == 0, W);
  hipLaunchKernelGGL(nesie::k4_moments_kernel, dim3(nesie::K4_MOM_WGS), dim3(256), 0, (hipStream_t)stream, nb, p,
                     x4_bstride, x4, (double *)mom_part);
  return check_launch(W);
}

// coef [64][4] = (scale, bias, mean, invstd) of BatchNorm(W0 . X4) in training mode from the moments
// of X4 (count = nb * p), running statistics updated like torch.nn.BatchNorm2d.
extern "C" int nesie_k4_stat_finalize(double count, const void *mom_part, const float *w0, const float *gamma,
                                      const float *beta, float *running_mean, float *running_var,
                                      float momentum, float eps, float *coef, void *stream) {
  const char *W = "k4_stat_finalize";
  NESIE_REQUIRE(count >= 1.0 && mom_part && w0 && coef, W);
  NESIE_REQUIRE((running_mean == nullptr) == (running_var == nullptr), W);
  hipLaunchKernelGGL(nesie::k4_stat_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, count,
                     (const double *)mom_part, w0, gamma, beta, running_mean, running_var, momentum, eps, coef);
  return check_launch(W);
}

// dW0 (64, 4) of SA1's first layer from the reductions alone (see k4_moments_kernel): mom_part the
// moments of the layer's input (nesie_k4_moments), bnb [64][8] the norm backward's coefficients
// (nesie_pw_bnb_coef over the bn_part of nesie_pw_dgrad_bn_reduce_k4), g_part [64][nslots][4] of
// the same launch.
extern "C" int nesie_k4_first_layer_wgrad(const void *mom_part, const float *w0, const float *bnb,
                                          const float *g_part, int nslots, float *dw, void *stream) {
  const char *W = "k4_first_layer_wgrad";
  NESIE_REQUIRE(nslots >= 1 && mom_part && w0 && bnb && g_part && dw && ((uintptr_t)g_part & 15) == 0, W);
  hipLaunchKernelGGL(nesie::k4_wgrad_finish_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, nslots,
                     (const double *)mom_part, g_part, bnb, w0, dw);
  return check_launch(W);
}
