// Training-mode BatchNorm (+ fused ReLU) forward and backward for gfx950.
//
// Stands in for the ATen/MIOpen batch_norm + relu pairs the reference builds with
// mmcv ConvModule (conv -> BN -> ReLU: reference mmdet3d/ops/pointnet_modules/
// point_sa_module.py:277-289, point_fp_module.py:28-37, model_utils/vote_module.py:61-74)
// and with nn.BatchNorm + nn.ReLU in side_pooling_module.py:55-78, 346-358.
// x (B, C, P) fp32 contiguous, statistics per channel over the B*P positions.
//
// All kernels are pure HBM streaming (float4 per lane, no re-reads inside a kernel):
//   forward : stats  (read x)            -> finalize (C threads) -> apply (read x, write y)
//   backward: reduce (read dy, y)        -> finalize             -> apply (read dy, x; write dx)
// i.e. 3 tensor passes forward and 5 backward, ReLU and its mask included (the unfused
// pair costs 5 and 10): the reduce rebuilds xhat from y where y > 0 (the only places the
// sums need it), the apply re-derives the mask from x with the forward's own scale/bias.  Sums are taken about a per-channel shift (the channel's first
// element) so E[x^2]-E[x]^2 does not cancel; partials are combined in double.
#include "common.h"
#include <stdlib.h>

namespace nesie {

constexpr int BN_BLOCK = 256;
constexpr int BN_SPAN = 8192;  // floats of one (b, c) row handled by a stats/reduce block

// log2 of the row-bias group (a power of two; 1 when there is no row bias)
// group == 0: row_bias holds ONE value per channel (a conv bias folded into the norm: the sum
// x + bias is formed in registers with the rounding of the separate add, never stored); the
// element index is then shifted out entirely.
__device__ __forceinline__ int group_shift(int group) { return group ? __ffs(group) - 1 : 62; }
__device__ __forceinline__ size_t rb_offset(int b, int c, int c_total, long long p, int group) {
  return group ? ((size_t)b * c_total + c) * (size_t)(p / group) : (size_t)c;
}

__device__ __forceinline__ float block_sum(float v, float *sh) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int w = 0; w < BN_BLOCK / 64; ++w) r += sh[w];
  return r;
}

// partial[(c * nslice + b * sp + s) * 2 + {0,1}] = sum(x - shift), sum((x - shift)^2)
__global__ __launch_bounds__(BN_BLOCK) void bn_stats_kernel(
    int c_total, long long p, int sp, const float *__restrict__ x,
    const float *__restrict__ row_bias, int group, float *__restrict__ partial, int nt) {
  __shared__ float sh[BN_BLOCK / 64];
  const int s = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
  const float *row = x + ((size_t)b * c_total + c) * p;
  // row_bias (optional): x_eff[b,c,i] = x[b,c,i] + row_bias[b,c,i / group]  (a per-proposal
  // term broadcast over its `group` grid points, never materialised)
  const float *rb = row_bias ? row_bias + rb_offset(b, c, c_total, p, group) : nullptr;
  const float shift = x[(size_t)c * p] + (row_bias ? row_bias[rb_offset(0, c, c_total, p, group)] : 0.f);
  const long long lo = (long long)s * BN_SPAN;
  const long long hi = lo + BN_SPAN < p ? lo + BN_SPAN : p;
  const int gs = group_shift(group);
  float a0 = 0.f, a1 = 0.f;
  if ((p & 3) == 0) {
    for (long long i = lo + threadIdx.x * 4; i < hi; i += BN_BLOCK * 4) {
      const float4 v = nt ? ld4<true>(row + i) : *(const float4 *)(row + i);
      const float r = rb ? rb[i >> gs] : 0.f;  // group % 4 == 0: one term per float4
      const float d0 = v.x + r - shift, d1 = v.y + r - shift, d2 = v.z + r - shift,
                  d3 = v.w + r - shift;
      a0 += (d0 + d1) + (d2 + d3);
      a1 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  } else {
    for (long long i = lo + threadIdx.x; i < hi; i += BN_BLOCK) {
      const float d = row[i] + (rb ? rb[i >> gs] : 0.f) - shift;
      a0 += d; a1 += d * d;
    }
  }
  a0 = block_sum(a0, sh);
  a1 = block_sum(a1, sh);
  if (threadIdx.x == 0) {
    const size_t o = ((size_t)c * (gridDim.z * sp) + (size_t)b * sp + s) * 2;
    partial[o] = a0; partial[o + 1] = a1;
  }
}

// coef[c*4 + {0,1,2,3}] = scale, bias, mean, invstd
// Everything the statistics -> coefficients step needs (the per-slice partial sums and where
// the results go).  The step runs inside the apply kernels: every block of a channel folds the
// channel's partials itself (wave 0, fp64; <= a few hundred values from L2) and the first block
// of the channel also writes the saved statistics / running statistics -- no separate
// one-block "finalize" launch between the two streaming passes.
struct BnFwdFin {
  int nslice; double n; const float *x; long long p; const float *row_bias; int group;
  const float *partial; const float *gamma; const float *beta; float *running_mean;
  float *running_var; float momentum; float eps; float *save_mean; float *save_invstd;
  float *coef;
};

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// -> sh[0] = scale, sh[1] = bias (valid for every thread after the barrier inside)
__device__ __forceinline__ void bn_fwd_finalize(const BnFwdFin &f, int c, bool writer, float *sh) {
  if (threadIdx.x < 64) {
    double s0 = 0.0, s1 = 0.0;
    for (int i = threadIdx.x; i < f.nslice; i += 64) {
      s0 += (double)f.partial[((size_t)c * f.nslice + i) * 2];
      s1 += (double)f.partial[((size_t)c * f.nslice + i) * 2 + 1];
    }
    s0 = wave_sum_f64(s0);
    s1 = wave_sum_f64(s1);
    if (threadIdx.x == 0) {
      // sums are taken about the channel's first element (no shift when the producer of the
      // partials had no such element at hand: f.x == NULL)
      const double shift = f.x ? (double)(f.x[(size_t)c * f.p] +
                                          (f.row_bias ? f.row_bias[f.group ? (size_t)c * (f.p / f.group) : (size_t)c] : 0.f))
                               : 0.0;
      const double m = s0 / f.n;
      double var = s1 / f.n - m * m;
      if (var < 0.0) var = 0.0;
      const double mean = shift + m;
      const double invstd = 1.0 / sqrt(var + (double)f.eps);
      const double g = f.gamma ? (double)f.gamma[c] : 1.0, bt = f.beta ? (double)f.beta[c] : 0.0;
      sh[0] = (float)(g * invstd);
      sh[1] = (float)(bt - mean * g * invstd);
      if (writer) {
        if (f.running_mean) {
          f.running_mean[c] = (float)((1.0 - f.momentum) * f.running_mean[c] + f.momentum * mean);
          const double unbiased = f.n > 1.0 ? var * f.n / (f.n - 1.0) : var;
          f.running_var[c] = (float)((1.0 - f.momentum) * f.running_var[c] + f.momentum * unbiased);
        }
        f.save_mean[c] = (float)mean;
        f.save_invstd[c] = (float)invstd;
        f.coef[c * 4 + 0] = sh[0];
        f.coef[c * 4 + 1] = sh[1];
        f.coef[c * 4 + 2] = (float)mean;
        f.coef[c * 4 + 3] = (float)invstd;
      }
    }
  }
  __syncthreads();
}

// the same step as a kernel of its own, for the pooled forward (whose blocks span channels)
__global__ void bn_finalize_kernel(BnFwdFin f, int c_total) {
  __shared__ float sh[2];
  if ((int)blockIdx.x < c_total) bn_fwd_finalize(f, blockIdx.x, true, sh);
}

template <bool RELU, bool NT>
__global__ __launch_bounds__(BN_BLOCK) void bn_apply_kernel(
    int c_total, long long p, const float *__restrict__ x, const float *__restrict__ row_bias,
    int group, BnFwdFin fin, float *__restrict__ y) {
  __shared__ float shc[2];
  const int c = blockIdx.y, b = blockIdx.z;
  bn_fwd_finalize(fin, c, blockIdx.x == 0 && b == 0, shc);
  const float sc = shc[0], bi = shc[1];
  const size_t base = ((size_t)b * c_total + c) * p;
  const float *rb = row_bias ? row_bias + rb_offset(b, c, c_total, p, group) : nullptr;
  const long long lo = (long long)blockIdx.x * BN_SPAN;
  const long long hi = lo + BN_SPAN < p ? lo + BN_SPAN : p;
  const int gs = group_shift(group);
  if ((p & 3) == 0) {
    for (long long i = lo + threadIdx.x * 4; i < hi; i += BN_BLOCK * 4) {
      float4 v = ld4<NT>(x + base + i);
      if (rb) { const float r = rb[i >> gs]; v.x += r; v.y += r; v.z += r; v.w += r; }
      float4 o = make_float4(v.x * sc + bi, v.y * sc + bi, v.z * sc + bi, v.w * sc + bi);
      if (RELU) {
        o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
      }
      st4<NT>(y + base + i, o);
    }
  } else {
    for (long long i = lo + threadIdx.x; i < hi; i += BN_BLOCK) {
      float o = (x[base + i] + (rb ? rb[i >> gs] : 0.f)) * sc + bi;
      y[base + i] = RELU ? fmaxf(o, 0.f) : o;
    }
  }
}

// Evaluation mode: y = relu?(scale[c] * (x + row_bias) + bias[c]) with the coefficients given
// (running statistics folded on the host side of the C ABI): one read + one write.
template <bool RELU, bool NT>
__global__ __launch_bounds__(BN_BLOCK) void affine_apply_kernel(
    int c_total, long long p, const float *__restrict__ x, const float *__restrict__ row_bias,
    int group, const float *__restrict__ coef, float *__restrict__ y) {
  const int c = blockIdx.y, b = blockIdx.z;
  const float sc = coef[c * 4 + 0], bi = coef[c * 4 + 1];
  const size_t base = ((size_t)b * c_total + c) * p;
  const float *rb = row_bias ? row_bias + rb_offset(b, c, c_total, p, group) : nullptr;
  const long long lo = (long long)blockIdx.x * BN_SPAN;
  const long long hi = lo + BN_SPAN < p ? lo + BN_SPAN : p;
  const int gs = group_shift(group);
  if ((p & 3) == 0) {
    for (long long i = lo + threadIdx.x * 4; i < hi; i += BN_BLOCK * 4) {
      float4 v = ld4<NT>(x + base + i);
      if (rb) { const float r = rb[i >> gs]; v.x += r; v.y += r; v.z += r; v.w += r; }
      float4 o = make_float4(v.x * sc + bi, v.y * sc + bi, v.z * sc + bi, v.w * sc + bi);
      if (RELU) {
        o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
      }
      st4<NT>(y + base + i, o);
    }
  } else {
    for (long long i = lo + threadIdx.x; i < hi; i += BN_BLOCK) {
      float o = (x[base + i] + (rb ? rb[i >> gs] : 0.f)) * sc + bi;
      y[base + i] = RELU ? fmaxf(o, 0.f) : o;
    }
  }
}

// backward partials: sum(g), sum(g * xhat), g = RELU ? dy * [y > 0] : dy
template <bool RELU, bool NT>
__global__ __launch_bounds__(BN_BLOCK) void bn_bwd_reduce_kernel(
    int c_total, long long p, int sp, const float *__restrict__ dy,
    const float *__restrict__ x, const float *__restrict__ y,
    const float *__restrict__ save_mean, const float *__restrict__ save_invstd,
    const float *__restrict__ gamma, const float *__restrict__ beta,
    const float *__restrict__ row_bias, int group, float *__restrict__ partial,
    const float *__restrict__ raw_coef, const float *__restrict__ fwd_coef) {
  // save_mean / save_invstd NULL: columns 2 / 3 of fwd_coef (scale, bias, mean, invstd).
  // raw_coef != NULL: the normalised tensor y was never stored (the next layer applied the
  // norm + ReLU on its operand load, pwconv.hip): the ReLU mask is fma(x, scale, bias) > 0,
  // the fused form that kernel used
  __shared__ float sh[BN_BLOCK / 64];
  const int s = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
  const size_t base = ((size_t)b * c_total + c) * p;
  const float *rb = row_bias ? row_bias + rb_offset(b, c, c_total, p, group) : nullptr;
  float mean = save_mean ? save_mean[c] : fwd_coef[c * 4 + 2];
  float invstd = save_invstd ? save_invstd[c] : fwd_coef[c * 4 + 3];
  const float rsc = raw_coef ? raw_coef[c * 4 + 0] : 0.f, rbi = raw_coef ? raw_coef[c * 4 + 1] : 0.f;
  // With the ReLU fused, xhat is only needed where y > 0, and there y = gamma*xhat + beta:
  // xhat = (y - beta) / gamma needs no read of x (one tensor pass less).  Channels whose
  // gamma is ~0 keep the x path.  `src` is the tensor xhat is rebuilt from.
  const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
  const bool from_y = RELU && !raw_coef && fabsf(gm) > 1e-4f;
  const float *__restrict__ src = from_y ? y : x;
  if (from_y) { mean = bt; invstd = 1.f / gm; }
  const long long lo = (long long)s * BN_SPAN;
  const long long hi = lo + BN_SPAN < p ? lo + BN_SPAN : p;
  const int gs = group_shift(group);
  float a0 = 0.f, a1 = 0.f;
  if ((p & 3) == 0) {
    for (long long i = lo + threadIdx.x * 4; i < hi; i += BN_BLOCK * 4) {
      float4 g = *(const float4 *)(dy + base + i);      // read again by the apply pass
      float4 v = from_y ? ld4<NT>(src + base + i) : *(const float4 *)(src + base + i);
      if (rb && !from_y) { const float r = rb[i >> gs]; v.x += r; v.y += r; v.z += r; v.w += r; }
      if (RELU) {
        float4 o;
        if (raw_coef) o = make_float4(__builtin_fmaf(v.x, rsc, rbi), __builtin_fmaf(v.y, rsc, rbi),
                                      __builtin_fmaf(v.z, rsc, rbi), __builtin_fmaf(v.w, rsc, rbi));
        else o = from_y ? v : ld4<NT>(y + base + i);
        g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f;
        g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
      }
      a0 += (g.x + g.y) + (g.z + g.w);
      a1 += (g.x * ((v.x - mean) * invstd) + g.y * ((v.y - mean) * invstd)) +
            (g.z * ((v.z - mean) * invstd) + g.w * ((v.w - mean) * invstd));
    }
  } else {
    for (long long i = lo + threadIdx.x; i < hi; i += BN_BLOCK) {
      float g = dy[base + i];
      float v = src[base + i];
      if (rb && !from_y) v += rb[i >> gs];
      if (RELU) g = (raw_coef ? __builtin_fmaf(v, rsc, rbi) : from_y ? v : y[base + i]) > 0.f ? g : 0.f;
      a0 += g; a1 += g * ((v - mean) * invstd);
    }
  }
  a0 = block_sum(a0, sh);
  a1 = block_sum(a1, sh);
  if (threadIdx.x == 0) {
    const size_t o = ((size_t)c * (gridDim.z * sp) + (size_t)b * sp + s) * 2;
    partial[o] = a0; partial[o + 1] = a1;
  }
}

// coef[c*4 + {0,1,2}] = gamma*invstd, sum(g)/n, sum(g*xhat)/n ; dgamma, dbeta written
struct BnBwdFin {
  int nslice; double n; const float *partial; const float *gamma; const float *save_invstd;
  float *dgamma; float *dbeta;
  int istride;   // save_invstd[c * istride]: 1 for a plain vector, 4 for column 3 of fwd_coef
};

// -> sh[0] = gamma * invstd, sh[1] = sum(dy) / n, sh[2] = sum(dy * xhat) / n
__device__ __forceinline__ void bn_bwd_finalize(const BnBwdFin &f, int c, bool writer, float *sh) {
  if (threadIdx.x < 64) {
    double s0 = 0.0, s1 = 0.0;
    for (int i = threadIdx.x; i < f.nslice; i += 64) {
      s0 += (double)f.partial[((size_t)c * f.nslice + i) * 2];
      s1 += (double)f.partial[((size_t)c * f.nslice + i) * 2 + 1];
    }
    s0 = wave_sum_f64(s0);
    s1 = wave_sum_f64(s1);
    if (threadIdx.x == 0) {
      const double g = f.gamma ? (double)f.gamma[c] : 1.0;
      sh[0] = (float)(g * (double)f.save_invstd[(size_t)c * f.istride]);
      sh[1] = (float)(s0 / f.n);
      sh[2] = (float)(s1 / f.n);
      if (writer) {
        if (f.dbeta) f.dbeta[c] = (float)s0;
        if (f.dgamma) f.dgamma[c] = (float)s1;
      }
    }
  }
  __syncthreads();
}

// kernel form for the pooled backward; coef[c] = (a, k1, k2, -)
__global__ void bn_bwd_finalize_kernel(BnBwdFin f, int c_total, float *coef) {
  __shared__ float sh[3];
  const int c = blockIdx.x;
  if (c >= c_total) return;
  bn_bwd_finalize(f, c, true, sh);
  if (threadIdx.x == 0) { coef[c * 4 + 0] = sh[0]; coef[c * 4 + 1] = sh[1]; coef[c * 4 + 2] = sh[2]; }
}

template <bool RELU, bool NT>
__global__ __launch_bounds__(BN_BLOCK) void bn_bwd_apply_kernel(
    int c_total, long long p, const float *__restrict__ dy, const float *__restrict__ x,
    const float *__restrict__ fwd_coef, BnBwdFin fin,
    const float *__restrict__ row_bias, int group, float *__restrict__ d_row_bias,
    float *__restrict__ dx, int fused_mask) {
  // dx = a * (g - k1 - xhat * k2) for EVERY position, masked ones included, so x is needed
  // everywhere; the ReLU mask is re-derived from x with the forward's own scale/bias
  // (same two fp32 operations as bn_apply_kernel, hence the same bits) instead of reading y.
  __shared__ float shc[3];
  const int c = blockIdx.y, b = blockIdx.z;
  bn_bwd_finalize(fin, c, blockIdx.x == 0 && b == 0, shc);
  const float a = shc[0], k1 = shc[1], k2 = shc[2];
  const float sc = fwd_coef[c * 4 + 0], bi = fwd_coef[c * 4 + 1];
  const float mean = fwd_coef[c * 4 + 2], invstd = fwd_coef[c * 4 + 3];
  const size_t base = ((size_t)b * c_total + c) * p;
  const size_t rbase = (row_bias || d_row_bias) ? rb_offset(b, c, c_total, p, group) : 0;
  const long long lo = (long long)blockIdx.x * BN_SPAN;
  const long long hi = lo + BN_SPAN < p ? lo + BN_SPAN : p;
  const int gs = group_shift(group);
  if ((p & 3) == 0) {
    for (long long i = lo + threadIdx.x * 4; i < hi; i += BN_BLOCK * 4) {
      float4 g = ld4<NT>(dy + base + i);
      float4 v = ld4<NT>(x + base + i);
      if (row_bias) { const float r = row_bias[rbase + (i >> gs)]; v.x += r; v.y += r; v.z += r; v.w += r; }
      if (RELU) {
        if (fused_mask) {   // the forward applied fma(x, scale, bias) (pwconv.hip operand load)
          g.x = __builtin_fmaf(v.x, sc, bi) > 0.f ? g.x : 0.f; g.y = __builtin_fmaf(v.y, sc, bi) > 0.f ? g.y : 0.f;
          g.z = __builtin_fmaf(v.z, sc, bi) > 0.f ? g.z : 0.f; g.w = __builtin_fmaf(v.w, sc, bi) > 0.f ? g.w : 0.f;
        } else {
          g.x = v.x * sc + bi > 0.f ? g.x : 0.f; g.y = v.y * sc + bi > 0.f ? g.y : 0.f;
          g.z = v.z * sc + bi > 0.f ? g.z : 0.f; g.w = v.w * sc + bi > 0.f ? g.w : 0.f;
        }
      }
      float4 r;
      r.x = a * (g.x - k1 - (v.x - mean) * invstd * k2);
      r.y = a * (g.y - k1 - (v.y - mean) * invstd * k2);
      r.z = a * (g.z - k1 - (v.z - mean) * invstd * k2);
      r.w = a * (g.w - k1 - (v.w - mean) * invstd * k2);
      st4<NT>(dx + base + i, r);
      if (d_row_bias) {
        // gradient of the broadcast term = sum of dx over its `group` positions = the
        // group/4 consecutive lanes that hold them (group in {4..256}, power of two)
        float t = (r.x + r.y) + (r.z + r.w);
        for (int off = 1; off < group / 4; off <<= 1) t += __shfl_xor(t, off, 64);
        if (((threadIdx.x & 63) & (group / 4 - 1)) == 0) d_row_bias[rbase + (i >> gs)] = t;
      }
    }
  } else {
    for (long long i = lo + threadIdx.x; i < hi; i += BN_BLOCK) {
      float g = dy[base + i];
      // (the grouped row-bias form requires p % 4 == 0, checked on the host; the per-channel one not)
      const float v = x[base + i] + (row_bias ? row_bias[rbase + (i >> gs)] : 0.f);
      if (RELU) g = (fused_mask ? __builtin_fmaf(v, sc, bi) : v * sc + bi) > 0.f ? g : 0.f;
      dx[base + i] = a * (g - k1 - (v - mean) * invstd * k2);
    }
  }
}

// ---- BatchNorm + ReLU + max over the neighbourhood axis in one pass ------------------
// The last layer of a set-abstraction MLP feeds F.max_pool2d([1, nsample])
// (point_sa_module.py:136-158): only the pooled (B, C, M) tensor is needed downstream, so the
// normalised (B, C, M, ns) tensor is never written.  Same arithmetic per element as
// bn_apply_kernel<true> followed by group_max_fwd_kernel (smallest index on ties).
template <int LPR>
__global__ __launch_bounds__(256) void bn_pool_fwd_kernel(
    long long rows, int m, int c_total, const float4 *__restrict__ x,
    const float *__restrict__ coef, float *__restrict__ out, uint8_t *__restrict__ arg, int nt) {
  // grid.y = (b, c) slab of m rows, grid.x covers its m * LPR float4s: the channel is a scalar
  // per workgroup (no per-thread 64-bit division)
  const int slab = blockIdx.y;
  const int tl = blockIdx.x * 256 + threadIdx.x;          // float4 index inside the slab
  const bool live = tl < m * LPR;
  const int c = slab % c_total;
  const long long t = (long long)slab * m * LPR + tl;
  const long long row = (long long)slab * m + tl / LPR;
  const int part = tl % LPR;
  const float sc = coef[c * 4 + 0], bi = coef[c * 4 + 1];
  float4 q = live ? (nt ? ld4<true>((const float *)(x + t)) : x[t]) : make_float4(0.f, 0.f, 0.f, 0.f);
  q.x = fmaxf(q.x * sc + bi, 0.f); q.y = fmaxf(q.y * sc + bi, 0.f);
  q.z = fmaxf(q.z * sc + bi, 0.f); q.w = fmaxf(q.w * sc + bi, 0.f);
  float v; int i;
  row_argmax4<LPR>(q, part, v, i);
  if (live && part == 0) { out[row] = v; arg[row] = (uint8_t)i; }
}

// The gradient that reaches the normalised tensor is one value per row (at the arg-max, and
// only where the pooled activation is positive), so the two BatchNorm sums need the pooled
// gradient, the pooled activation and ONE gathered x per row -- not the dense tensors.
__global__ __launch_bounds__(BN_BLOCK) void bn_pool_bwd_reduce_kernel(
    int c_total, int m, int ns, int rows_per, const float *__restrict__ gpool,
    const float *__restrict__ pooled, const uint8_t *__restrict__ arg,
    const float *__restrict__ x, const float *__restrict__ fwd_coef,
    float *__restrict__ partial) {
  __shared__ float sh[BN_BLOCK / 64];
  const int s = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
  const float mean = fwd_coef[c * 4 + 2], invstd = fwd_coef[c * 4 + 3];
  const float sc = fwd_coef[c * 4 + 0], bi = fwd_coef[c * 4 + 1];
  // Where the pooled activation is positive it IS the normalised arg-max element,
  // pooled = sc * x + bi, so x comes back from it without touching the dense tensor (one
  // 4-byte gather per 64-byte line otherwise: 0.16 GB of the pair's traffic at SA1).  Channels
  // whose scale is ~0 keep the gather.
  const bool from_pooled = fabsf(sc) > 1e-4f * invstd;
  const float rsc = from_pooled ? 1.f / sc : 0.f;
  const size_t base = ((size_t)b * c_total + c) * m;
  const int lo = s * rows_per, hi = lo + rows_per < m ? lo + rows_per : m;
  float a0 = 0.f, a1 = 0.f;
  for (int i = lo + threadIdx.x; i < hi; i += BN_BLOCK) {
    const float pv = pooled[base + i];
    if (pv > 0.f) {
      const float g = gpool[base + i];
      const float xa = from_pooled ? (pv - bi) * rsc : x[(base + i) * ns + arg[base + i]];
      a0 += g;
      a1 += g * ((xa - mean) * invstd);
    }
  }
  a0 = block_sum(a0, sh);
  a1 = block_sum(a1, sh);
  if (threadIdx.x == 0) {
    const size_t o = ((size_t)c * (gridDim.z * gridDim.x) + (size_t)b * gridDim.x + s) * 2;
    partial[o] = a0; partial[o + 1] = a1;
  }
}

// dx = a * (dy - k1 - xhat * k2) with dy rebuilt from (pooled gradient, arg-max, pooled > 0)
template <int LPR>
__global__ __launch_bounds__(256) void bn_pool_bwd_apply_kernel(
    long long rows, int m, int c_total, const float4 *__restrict__ x,
    const float *__restrict__ gpool, const float *__restrict__ pooled,
    const uint8_t *__restrict__ arg, const float *__restrict__ fwd_coef,
    const float *__restrict__ coef, float4 *__restrict__ dx, int nt) {
  const int slab = blockIdx.y;                              // (b, c): scalar channel
  const int tl = blockIdx.x * 256 + threadIdx.x;
  if (tl >= m * LPR) return;
  const int c = slab % c_total;
  const long long t = (long long)slab * m * LPR + tl;
  const long long row = (long long)slab * m + tl / LPR;
  const int part = tl % LPR;
  const float a = coef[c * 4 + 0], k1 = coef[c * 4 + 1], k2 = coef[c * 4 + 2];
  const float mean = fwd_coef[c * 4 + 2], invstd = fwd_coef[c * 4 + 3];
  const float g = pooled[row] > 0.f ? gpool[row] : 0.f;
  const int ai = (int)arg[row] - part * 4;
  const float4 v = nt ? ld4<true>((const float *)(x + t)) : x[t];
  float4 r;
  r.x = a * ((ai == 0 ? g : 0.f) - k1 - (v.x - mean) * invstd * k2);
  r.y = a * ((ai == 1 ? g : 0.f) - k1 - (v.y - mean) * invstd * k2);
  r.z = a * ((ai == 2 ? g : 0.f) - k1 - (v.z - mean) * invstd * k2);
  r.w = a * ((ai == 3 ? g : 0.f) - k1 - (v.w - mean) * invstd * k2);
  if (nt) st4<true>((float *)(dx + t), r); else dx[t] = r;
}

static inline int bn_sp(long long p) { return (int)((p + BN_SPAN - 1) / BN_SPAN); }

}  // namespace nesie

using namespace nesie;

extern "C" size_t nesie_bn_workspace_bytes(int b, int c, long long p) {
  if (b <= 0 || c <= 0 || p <= 0) return 0;
  const size_t nslice = (size_t)b * bn_sp(p);
  return ((size_t)c * nslice * 2 + (size_t)c * 4) * sizeof(float);
}

static int bn_check(const char *W, int b, int c, long long p, const void *ws, size_t ws_bytes) {
  NESIE_REQUIRE(b >= 0 && c >= 0 && p >= 0, W);
  if (b == 0 || c == 0 || p == 0) return NESIE_OK;
  NESIE_REQUIRE(ws && ws_bytes >= nesie_bn_workspace_bytes(b, c, p), W);
  NESIE_REQUIRE(b <= 65535 && c <= 65535 && bn_sp(p) < (1 << 30), W);
  return NESIE_OK;
}

// Non-temporal accesses for tensors that cannot stay in the 256 MB Infinity Cache anyway.
// NESIE_NT_MB=<n>: threshold in MB (0 = never), a measurement aid.
namespace nesie {
bool stream_nt(long long bytes, int family) {
  static const long long mb = [] { const char *e = getenv("NESIE_NT_MB"); return e ? atoll(e) : 192ll; }();
  static const int mask = [] { const char *e = getenv("NESIE_NT_MASK"); return e ? atoi(e) : 3; }();
  return mb > 0 && (mask & family) && bytes >= (mb << 20);
}
}  // namespace nesie
static bool bn_use_nt(long long elements) { return nesie::stream_nt(elements * 4, 1); }

extern "C" int nesie_bn_relu_forward(int b, int c, long long p, const float *x,
                                     const float *gamma, const float *beta,
                                     float *running_mean, float *running_var,
                                     float momentum, float eps, int relu, float *y,
                                     float *save_mean, float *save_invstd, float *fwd_coef,
                                     const float *row_bias, int group,
                                     const float *pre_partial, int pre_nslice,
                                     void *workspace, size_t workspace_bytes, void *stream) {
  const char *W = "bn_relu_forward";
  int st = bn_check(W, b, c, p, workspace, workspace_bytes);
  if (st || b == 0 || c == 0 || p == 0) return st;
  NESIE_REQUIRE(x && y && save_mean && save_invstd && fwd_coef, W);
  NESIE_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, W);
  if (!row_bias) group = 1;
  // group 0: row_bias is one value per channel (a folded conv bias)
  NESIE_REQUIRE(group >= 0 && (group == 0 || p % group == 0), W);
  if (row_bias && group != 0 && (group < 4 || group > 256 || (group & (group - 1)) || (BN_SPAN % group))) {
    set_error("%s: row_bias group %d (needs 0 = per channel, or a power of two in 4..256)", W, group);
    return NESIE_ERR_UNSUPPORTED;
  }
  hipStream_t s = (hipStream_t)stream;
  const int sp = bn_sp(p), nslice = b * sp;
  float *partial = (float *)workspace, *coef = fwd_coef;  // [C][4]: scale, bias, mean, invstd
  dim3 grid(sp, c, b);
  NESIE_REQUIRE(!pre_partial || (pre_nslice >= 1 && !row_bias), W);
  if (!pre_partial)
    hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(BN_BLOCK), 0, s, c, p, sp, x, row_bias, group,
                       partial, nesie::stream_nt((long long)b * c * p * 4, 8) ? 1 : 0);
  // the producer's partials are unshifted sums: no shift element (x = NULL in the finalize)
  const BnFwdFin fin{pre_partial ? pre_nslice : nslice, (double)b * (double)p,
                     pre_partial ? nullptr : x, p, row_bias, group,
                     pre_partial ? pre_partial : partial, gamma, beta,
                     running_mean, running_var, momentum, eps, save_mean, save_invstd, coef};
  const bool nt = bn_use_nt((long long)b * c * p);
#define L(R, N) hipLaunchKernelGGL((bn_apply_kernel<R, N>), grid, dim3(BN_BLOCK), 0, s, c, p, x, row_bias, group, fin, y)
  if (relu) { if (nt) L(true, true); else L(true, false); }
  else { if (nt) L(false, true); else L(false, false); }
#undef L
  return check_launch(W);
}

extern "C" int nesie_bn_relu_backward(int b, int c, long long p, const float *dy,
                                      const float *x, const float *y, const float *gamma,
                                      const float *beta, const float *save_mean,
                                      const float *save_invstd, const float *fwd_coef,
                                      int relu, float *dx, float *dgamma, float *dbeta,
                                      const float *row_bias, int group, float *d_row_bias,
                                      void *workspace, size_t workspace_bytes, void *stream) {
  const char *W = "bn_relu_backward";
  int st = bn_check(W, b, c, p, workspace, workspace_bytes);
  if (st || b == 0 || c == 0 || p == 0) return st;
  NESIE_REQUIRE(dy && x && dx && fwd_coef && (save_mean == nullptr) == (save_invstd == nullptr), W);
  // save_mean / save_invstd NULL: columns 2 / 3 of fwd_coef.
  // y == NULL with relu: the normalised tensor was never stored (fused forward): the mask is
  // re-derived from x with the forward's fused scale / bias.  d_row_bias without row_bias: only
  // the per-group sums of dx are wanted (the producer had added the row term itself).
  const int raw = relu && !y;
  if (!row_bias && !d_row_bias) group = 1;
  NESIE_REQUIRE(group >= 0 && (group == 0 || p % group == 0), W);
  // group 0 (per-channel bias): no d_row_bias -- the gradient of a bias that feeds a batch norm
  // is identically zero (the mean subtraction removes it)
  NESIE_REQUIRE(group != 0 || (row_bias && !d_row_bias), W);
  NESIE_REQUIRE(group == 0 || !(row_bias || d_row_bias) || (group >= 4 && group <= 256 && !(group & (group - 1)) && (p & 3) == 0), W);
  NESIE_REQUIRE((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)y) & 15) == 0, W);
  hipStream_t s = (hipStream_t)stream;
  const int sp = bn_sp(p), nslice = b * sp;
  float *partial = (float *)workspace, *coef = partial + (size_t)c * nslice * 2;
  dim3 grid(sp, c, b);
  const bool nt = bn_use_nt((long long)b * c * p);
#define LR(R, N) hipLaunchKernelGGL((bn_bwd_reduce_kernel<R, N>), grid, dim3(BN_BLOCK), 0, s, c, p, sp, dy, \
                                    x, y, save_mean, save_invstd, gamma, beta, row_bias, group, partial, \
                                    raw ? fwd_coef : (const float *)nullptr, fwd_coef)
  if (relu) { if (nt) LR(true, true); else LR(true, false); }
  else { if (nt) LR(false, true); else LR(false, false); }
#undef LR
  const BnBwdFin fin{nslice, (double)b * (double)p, partial, gamma, save_invstd ? save_invstd : fwd_coef + 3, dgamma, dbeta, save_invstd ? 1 : 4};
  (void)coef;
#define LA(R, N) hipLaunchKernelGGL((bn_bwd_apply_kernel<R, N>), grid, dim3(BN_BLOCK), 0, s, c, p, dy, x, \
                                    fwd_coef, fin, row_bias, group, d_row_bias, dx, raw)
  if (relu) { if (nt) LA(true, true); else LA(true, false); }
  else { if (nt) LA(false, true); else LA(false, false); }
#undef LA
  return check_launch(W);
}


// The apply pass alone, for a producer that has already left the reduction's partials
// (partial[(c * nslice + i) * 2 + {0,1}] = sum(g), sum(g * xhat); pwconv.hip's
// nesie_pw_dgrad_bn_reduce): x is the RAW conv output of a fused forward (mask = fma(x, scale,
// bias) > 0), dx = gamma invstd (g - sum(g)/n - xhat sum(g xhat)/n); dgamma / dbeta written.
extern "C" int nesie_bn_relu_backward_apply(int b, int c, long long p, const float *dy,
                                            const float *x, const float *gamma,
                                            const float *save_invstd, const float *fwd_coef,
                                            const float *partial, int nslice, float *dx,
                                            float *dgamma, float *dbeta, int group,
                                            float *d_row_bias, void *stream) {
  const char *W = "bn_relu_backward_apply";
  NESIE_REQUIRE(b >= 0 && c >= 0 && p >= 0 && nslice >= 1, W);
  if (b == 0 || c == 0 || p == 0) return NESIE_OK;
  NESIE_REQUIRE(dy && x && dx && fwd_coef && partial, W);   // save_invstd NULL: column 3 of fwd_coef
  if (!d_row_bias) group = 1;
  NESIE_REQUIRE(group >= 1 && p % group == 0, W);
  NESIE_REQUIRE(!d_row_bias || (group >= 4 && group <= 256 && !(group & (group - 1)) && (p & 3) == 0), W);
  NESIE_REQUIRE((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0, W);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(bn_sp(p), c, b);
  const bool nt = bn_use_nt((long long)b * c * p);
  const BnBwdFin fin{nslice, (double)b * (double)p, partial, gamma, save_invstd ? save_invstd : fwd_coef + 3, dgamma, dbeta, save_invstd ? 1 : 4};
  if (nt)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<true, true>), grid, dim3(BN_BLOCK), 0, s, c, p, dy, x,
                       fwd_coef, fin, (const float *)nullptr, group, d_row_bias, dx, 1);
  else
    hipLaunchKernelGGL((bn_bwd_apply_kernel<true, false>), grid, dim3(BN_BLOCK), 0, s, c, p, dy, x,
                       fwd_coef, fin, (const float *)nullptr, group, d_row_bias, dx, 1);
  return check_launch(W);
}

// For producers that compute the (sum, sum of squares) partials themselves
// (partial[(c * nslice + i) * 2 + {0,1}], unshifted): interpolate.hip's blend + norm kernels.
namespace nesie {
int launch_bn_finalize(int c, int nslice, double count, const float *partial,
                       const float *gamma, const float *beta, float *running_mean,
                       float *running_var, float momentum, float eps, float *save_mean,
                       float *save_invstd, float *coef, hipStream_t s) {
  const BnFwdFin fin{nslice, count, nullptr, 0, nullptr, 1, partial, gamma, beta, running_mean,
                     running_var, momentum, eps, save_mean, save_invstd, coef};
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(c), dim3(64), 0, s, fin, c);
  return check_launch("bn_finalize");
}
int launch_bn_bwd_finalize(int c, int nslice, double count, const float *partial,
                           const float *gamma, const float *save_invstd, float *dgamma,
                           float *dbeta, float *coef, hipStream_t s) {
  const BnBwdFin fin{nslice, count, partial, gamma, save_invstd, dgamma, dbeta, 1};
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(c), dim3(64), 0, s, fin, c, coef);
  return check_launch("bn_bwd_finalize");
}
}  // namespace nesie

static int bn_pool_dims(const char *W, int m, int ns) {
  if (m <= 0 || ns < 4 || ns > 64 || (ns & (ns - 1))) {
    set_error("%s: m %d, nsample %d (needs a power of two in 4..64)", W, m, ns);
    return NESIE_ERR_UNSUPPORTED;
  }
  return NESIE_OK;
}

extern "C" int nesie_bn_relu_maxpool_forward(int b, int c, int m, int ns, const float *x,
                                             const float *gamma, const float *beta,
                                             float *running_mean, float *running_var,
                                             float momentum, float eps, float *pooled,
                                             uint8_t *argmax, float *save_mean,
                                             float *save_invstd, float *fwd_coef,
                                             void *workspace, size_t workspace_bytes,
                                             void *stream) {
  const char *W = "bn_relu_maxpool_forward";
  NESIE_REQUIRE(b >= 0 && c >= 0, W);
  if (b == 0 || c == 0) return NESIE_OK;
  int st = bn_pool_dims(W, m, ns);
  if (st) return st;
  const long long p = (long long)m * ns;
  st = bn_check(W, b, c, p, workspace, workspace_bytes);
  if (st) return st;
  NESIE_REQUIRE(x && pooled && argmax && save_mean && save_invstd && fwd_coef, W);
  NESIE_REQUIRE(((uintptr_t)x & 15) == 0, W);
  hipStream_t s = (hipStream_t)stream;
  const int sp = bn_sp(p), nslice = b * sp;
  float *partial = (float *)workspace;
  hipLaunchKernelGGL(bn_stats_kernel, dim3(sp, c, b), dim3(BN_BLOCK), 0, s, c, p, sp, x,
                     (const float *)nullptr, 1, partial,
                     nesie::stream_nt((long long)b * c * p * 4, 8) ? 1 : 0);
  const BnFwdFin fin{nslice, (double)b * (double)p, x, p, nullptr, 1, partial, gamma, beta,
                     running_mean, running_var, momentum, eps, save_mean, save_invstd, fwd_coef};
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(c), dim3(64), 0, s, fin, c);
  const long long rows = (long long)b * c * m;
  const int lpr = ns / 4;
  NESIE_REQUIRE((long long)b * c <= 65535 && (long long)m * lpr < (1ll << 30), W);
  const dim3 grid((unsigned)cdiv((long long)m * lpr, 256), (unsigned)(b * c));
#define L(N) hipLaunchKernelGGL(bn_pool_fwd_kernel<N>, grid, dim3(256), 0, s, rows, m, c, \
                                (const float4 *)x, fwd_coef, pooled, argmax, nt)
  const int nt = stream_nt(rows * ns * 4, 2) ? 1 : 0;
  if (lpr == 1) L(1); else if (lpr == 2) L(2); else if (lpr == 4) L(4);
  else if (lpr == 8) L(8); else L(16);
#undef L
  return check_launch(W);
}

extern "C" int nesie_bn_relu_maxpool_backward(int b, int c, int m, int ns,
                                              const float *grad_pooled, const uint8_t *argmax,
                                              const float *x, const float *pooled,
                                              const float *gamma, const float *save_invstd,
                                              const float *fwd_coef, float *dx, float *dgamma,
                                              float *dbeta, void *workspace,
                                              size_t workspace_bytes, void *stream) {
  const char *W = "bn_relu_maxpool_backward";
  NESIE_REQUIRE(b >= 0 && c >= 0, W);
  if (b == 0 || c == 0) return NESIE_OK;
  int st = bn_pool_dims(W, m, ns);
  if (st) return st;
  const long long p = (long long)m * ns;
  st = bn_check(W, b, c, p, workspace, workspace_bytes);
  if (st) return st;
  NESIE_REQUIRE(grad_pooled && argmax && x && pooled && fwd_coef && dx, W);   // save_invstd NULL: column 3 of fwd_coef
  NESIE_REQUIRE((((uintptr_t)x | (uintptr_t)dx) & 15) == 0, W);
  hipStream_t s = (hipStream_t)stream;
  const int sp = bn_sp(p), nslice = b * sp, rows_per = cdiv(m, sp);
  float *partial = (float *)workspace, *coef = partial + (size_t)c * nslice * 2;
  hipLaunchKernelGGL(bn_pool_bwd_reduce_kernel, dim3(sp, c, b), dim3(BN_BLOCK), 0, s, c, m, ns,
                     rows_per, grad_pooled, pooled, argmax, x, fwd_coef, partial);
  const BnBwdFin fin{nslice, (double)b * (double)p, partial, gamma, save_invstd ? save_invstd : fwd_coef + 3, dgamma, dbeta, save_invstd ? 1 : 4};
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(c), dim3(64), 0, s, fin, c, coef);
  const long long rows = (long long)b * c * m;
  const int lpr = ns / 4;
  NESIE_REQUIRE((long long)b * c <= 65535 && (long long)m * lpr < (1ll << 30), W);
  const dim3 grid((unsigned)cdiv((long long)m * lpr, 256), (unsigned)(b * c));
#define L(N) hipLaunchKernelGGL(bn_pool_bwd_apply_kernel<N>, grid, dim3(256), 0, s, rows, m, c, \
                                (const float4 *)x, grad_pooled, pooled, argmax, fwd_coef, coef, \
                                (float4 *)dx, nt)
  const int nt = stream_nt(rows * ns * 4, 2) ? 1 : 0;
  if (lpr == 1) L(1); else if (lpr == 2) L(2); else if (lpr == 4) L(4);
  else if (lpr == 8) L(8); else L(16);
#undef L
  return check_launch(W);
}

// ---- evaluation mode -----------------------------------------------------------------------
// coef[c] = (gamma / sqrt(var + eps), beta - mean * scale, 0, 0): the folded running statistics,
// recomputed per call (one launch) -- the statistics are written by kernels through raw pointers
// and the parameters through the flat optimiser vector, so no host-side cache key can see them change
__global__ __launch_bounds__(256) void bn_eval_coef_kernel(int c, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta,
                                                           const float *__restrict__ mean,
                                                           const float *__restrict__ var, float eps,
                                                           float4 *__restrict__ coef) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= c) return;
  float scale = rsqrtf(var[i] + eps);
  if (gamma) scale = scale * gamma[i];
  float shift = -mean[i] * scale;
  if (beta) shift = shift + beta[i];
  coef[i] = make_float4(scale, shift, 0.f, 0.f);
}

extern "C" int nesie_bn_eval_coef(int c, const float *gamma, const float *beta,
                                  const float *running_mean, const float *running_var, float eps,
                                  float *coef, void *stream) {
  const char *W = "bn_eval_coef";
  NESIE_REQUIRE(c >= 0, W);
  if (c == 0) return NESIE_OK;
  NESIE_REQUIRE(running_mean && running_var && coef && ((uintptr_t)coef & 15) == 0, W);
  hipLaunchKernelGGL(bn_eval_coef_kernel, dim3(cdiv(c, 256)), dim3(256), 0, (hipStream_t)stream, c,
                     gamma, beta, running_mean, running_var, eps, (float4 *)coef);
  return check_launch(W);
}

extern "C" int nesie_affine_relu_forward(int b, int c, long long p, const float *x,
                                         const float *coef, int relu, const float *row_bias,
                                         int group, float *y, void *stream) {
  const char *W = "affine_relu_forward";
  NESIE_REQUIRE(b >= 0 && c >= 0 && p >= 0, W);
  if (b == 0 || c == 0 || p == 0) return NESIE_OK;
  NESIE_REQUIRE(x && coef && y && b <= 65535 && c <= 65535, W);
  if (!row_bias) group = 1;
  NESIE_REQUIRE(group >= 0 && (group == 0 || p % group == 0), W);
  NESIE_REQUIRE(!row_bias || group == 0 || (group >= 4 && group <= 256 && !(group & (group - 1)) && (p & 3) == 0), W);
  NESIE_REQUIRE((p & 3) != 0 || (((uintptr_t)x | (uintptr_t)y) & 15) == 0, W);
  const dim3 grid(bn_sp(p), c, b);
  hipStream_t s = (hipStream_t)stream;
  const bool nt = bn_use_nt((long long)b * c * p);
#define L(R, N) hipLaunchKernelGGL((affine_apply_kernel<R, N>), grid, dim3(BN_BLOCK), 0, s, c, p, x, row_bias, group, coef, y)
  if (relu) { if (nt) L(true, true); else L(true, false); }
  else { if (nt) L(false, true); else L(false, false); }
#undef L
  return check_launch(W);
}

extern "C" int nesie_affine_relu_maxpool_forward(int b, int c, int m, int ns, const float *x,
                                                 const float *coef, float *pooled,
                                                 uint8_t *argmax, void *stream) {
  const char *W = "affine_relu_maxpool_forward";
  NESIE_REQUIRE(b >= 0 && c >= 0 && m >= 0 && ns >= 1, W);
  if (b == 0 || c == 0 || m == 0) return NESIE_OK;
  if (ns < 4 || ns > 64 || (ns & (ns - 1))) {
    set_error("%s: nsample %d (needs a power of two in 4..64)", W, ns);
    return NESIE_ERR_UNSUPPORTED;
  }
  NESIE_REQUIRE(x && coef && pooled && argmax && ((uintptr_t)x & 15) == 0, W);
  const long long rows = (long long)b * c * m;
  const int lpr = ns / 4;
  NESIE_REQUIRE((long long)b * c <= 65535 && (long long)m * lpr < (1ll << 30), W);
  const dim3 grid((unsigned)cdiv((long long)m * lpr, 256), (unsigned)(b * c));
  hipStream_t s = (hipStream_t)stream;
  const int nt = stream_nt(rows * ns * 4, 2) ? 1 : 0;
#define L(N) hipLaunchKernelGGL(bn_pool_fwd_kernel<N>, grid, dim3(256), 0, s, rows, m, c, \
                                (const float4 *)x, coef, pooled, argmax, nt)
  if (lpr == 1) L(1); else if (lpr == 2) L(2); else if (lpr == 4) L(4);
  else if (lpr == 8) L(8); else L(16);
#undef L
  return check_launch(W);
}
