// Host side of the pointwise-convolution layer kernels: geometry choice, the C ABI entry points,
// the statistics / pooling finish kernels.  The layer kernel template lives in pwconv_fwd.h (its
// design notes are there); each tile geometry is instantiated in its own translation unit
// (pwconv_g*.hip) and the weight-gradient kernels in pwconv_wgrad.hip, so `make -j` builds the
// library in about a minute instead of five.
#include <algorithm>
#include "pwconv_fwd.h"
#include <stdlib.h>

namespace nesie {
// launchers of the built geometries (pwconv_g*.hip; with -DPW_DEV, tools/pwbench: defined below)
// (K sub-tile / 16, sub-tiles along K, row waves, column waves, 16-row sets per wave)
PW_GEOM_DECL(4, 1, 4, 1, 1)    // K <= 64,  Cout <= 64   (HBM-bound: small workgroups, up to four per CU)
PW_GEOM_DECL(4, 1, 4, 1, 2)    // K <= 64,  Cout <= 128
PW_GEOM_DECL(4, 1, 8, 1, 1)    // K <= 64,  Cout <= 128  (A/B: NESIE_PW_K64=8)
PW_GEOM_DECL(8, 1, 4, 1, 1)    // K <= 128, Cout <= 64
PW_GEOM_DECL(8, 1, 8, 1, 1)    // K <= 128, Cout <= 128
PW_GEOM_DECL(9, 1, 8, 1, 1)    // K <= 144
PW_GEOM_DECL(8, 2, 8, 1, 1)    // K <= 256
PW_GEOM_DECL(9, 2, 8, 1, 1)    // K <= 288
PW_GEOM_DECL(8, 4, 8, 1, 1)    // K <= 512

// Chan merge of the per-wave partials -> (scale, bias, mean, invstd) + running statistics.
// One 64-thread block per channel (channel index runs over ng * cout: stacked layers).
__global__ __launch_bounds__(64) void pw_stats_finalize_kernel(
    int cout, int nslots, const float *__restrict__ part, const float *__restrict__ gamma,
    const float *__restrict__ beta, float *running_mean, float *running_var, float momentum,
    float eps, float *__restrict__ coef, const float *__restrict__ chan_bias) {
  const int ch = blockIdx.x, g = ch / cout, m = ch % cout;
  const float4 *pp = (const float4 *)part + ((size_t)g * nslots) * cout + m;
  // ONE pass (round 5; two dependent passes before: the mean first, then the squares about it): a
  // lane merges its slots pairwise with Chan's update -- four slot loads in flight -- and a fixed
  // butterfly merges the 64 lanes' (n, mean, M2) triples, all in fp64
  double n = 0.0, mean = 0.0, m2 = 0.0;
  auto merge = [&](double nb_, double mb_, double m2b_) {
    if (nb_ <= 0.0) return;
    const double nn = n + nb_, d = mb_ - mean;
    mean += d * (nb_ / nn);
    m2 += m2b_ + d * d * (n * nb_ / nn);
    n = nn;
  };
  auto slot = [&](const float4 q) {
    if (q.x > 0.f) {
      const double ni = q.x, mi = (double)q.y + (double)q.z / ni;
      const double m2i = (double)q.w - (double)q.z * (double)q.z / ni;
      merge(ni, mi, m2i > 0.0 ? m2i : 0.0);
    }
  };
  int i = threadIdx.x;
  for (; i + 192 < nslots; i += 256) {
    float4 q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) q[u] = pp[(size_t)(i + 64 * u) * cout];
#pragma unroll
    for (int u = 0; u < 4; ++u) slot(q[u]);
  }
  for (; i < nslots; i += 64) slot(pp[(size_t)i * cout]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double nb_ = __shfl_xor(n, off, 64), mb_ = __shfl_xor(mean, off, 64), m2b_ = __shfl_xor(m2, off, 64);
    // (both lanes of a pair compute the same merged triple: a symmetric form, so that they agree bit for bit)
    const double nn = n + nb_;
    if (nn > 0.0) {
      const double d = mb_ - mean;
      const double mm = (n * mean + nb_ * mb_) / nn;
      m2 = m2 + m2b_ + d * d * (n * nb_ / nn);
      mean = mm;
    }
    n = nn;
  }
  if (threadIdx.x != 0) return;
  const double var = n > 0.0 ? m2 / n : 0.0;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  if (running_mean) {
    // chan_bias: the bias of the convolution in front of the norm.  The layer kernel never adds
    // it (the mean subtraction removes it from every normalised value); the running mean of the
    // biased output differs from the unbiased one by exactly the bias
    const double shown = mean + (chan_bias ? (double)chan_bias[ch] : 0.0);
    running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * shown);
    const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
    running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * unbiased);
  }
  const double gm = gamma ? (double)gamma[ch] : 1.0, bt = beta ? (double)beta[ch] : 0.0;
  coef[ch * 4 + 0] = (float)(gm * invstd);
  coef[ch * 4 + 1] = (float)(bt - mean * gm * invstd);
  coef[ch * 4 + 2] = (float)mean;
  coef[ch * 4 + 3] = (float)invstd;
}

// The pooling tail after the layer kernel: combine the G / PG partial extrema of every group,
// apply the layer's own BatchNorm + ReLU to the extremum the sign of the scale selects
// (max_j relu(s y_j + b) = relu(s (s >= 0 ? max_j y_j : min_j y_j) + b), exactly) and record its
// position inside the group.  rows = nb * channels, coef == NULL: plain max.
__global__ __launch_bounds__(256) void pw_pool_finish_kernel(
    long long total, int channels, int ng, int groups, int nsub, int pg, const float *__restrict__ pmax,
    const float *__restrict__ pmin, const uint8_t *__restrict__ amax,
    const uint8_t *__restrict__ amin, const float *__restrict__ coef, float lo,
    float *__restrict__ pooled, uint8_t *__restrict__ arg, float *__restrict__ zstar) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // (row, group)
  if (i >= total) return;
  const long long row = i / groups;
  const int c = (int)((row / channels) % ng) * channels + (int)(row % channels);   // stacked layers
  const float sc = coef ? coef[c * 4 + 0] : 1.f, bi = coef ? coef[c * 4 + 1] : 0.f;
  const bool use_min = coef && sc < 0.f;
  const float *src = use_min ? pmin : pmax;
  const uint8_t *asrc = use_min ? amin : amax;
  float e = src[i * nsub];
  int at = asrc[i * nsub];
  for (int u = 1; u < nsub; ++u) {
    const float v = src[i * nsub + u];
    const bool better = use_min ? v < e : v > e;   // first extremum wins ties
    at = better ? u * pg + asrc[i * nsub + u] : at;
    e = better ? v : e;
  }
  pooled[i] = coef ? fmaxf(__builtin_fmaf(e, sc, bi), lo) : e;
  arg[i] = (uint8_t)at;
  if (zstar) zstar[i] = e;      // the raw extremum (a pooled tail's backward needs its zhat)
}

// tile geometry of a (K, Cout) layer
struct PwGeom { int kt16, kh, wr, wc, rw, pt, nhalf, per_cu; };

static bool pw_geometry(int k, int cout, PwGeom *o) {
  if (k < 1 || cout < 1 || cout > 512) return false;
  if (k <= 64) { o->kt16 = 4; o->kh = 1; }
  else if (k <= 128) { o->kt16 = 8; o->kh = 1; }
  else if (k <= 144) { o->kt16 = 9; o->kh = 1; }
  else if (k <= 256) { o->kt16 = 8; o->kh = 2; }
  else if (k <= 288) { o->kt16 = 9; o->kh = 2; }
  else if (k > 384 && k <= 512) { o->kt16 = 8; o->kh = 4; }   // (288, 384] would leave sub-tile 3 empty
  else return false;
  // K > 64 (MFMA-bound): 8 waves x 16 rows, 64-position tiles, <= 128 VGPRs: two workgroups per
  // CU.  K <= 64 (HBM-bound, SA1): 4-wave workgroups, up to four per CU.  Cout > 128: two
  // 128-row workgroups per tile.  Measured and rejected in round 3 (tools/pwbench):
  // 32 rows per wave in 4-wave workgroups, one 256-row workgroup of 32-row waves, a
  // two-team ping-pong workgroup.
  o->nhalf = (cout + 127) / 128;   // 128-row workgroups per tile
  if (o->kt16 == 4) {
    static const int k64 = [] { const char *e = getenv("NESIE_PW_K64"); return e ? atoi(e) : 4; }();
    o->wr = 4; o->wc = 1; o->rw = cout <= 64 ? 1 : 2;
    if (k64 == 8 && cout > 64) { o->wr = 8; o->rw = 1; }                       // A/B switch
  } else if (cout <= 64 && o->kt16 == 8 && o->kh == 1) {
    o->wr = 4; o->wc = 1; o->rw = 1;
  } else {
    o->wr = 8; o->wc = 1; o->rw = 1;
  }
  o->pt = 64 * o->wc;
  o->per_cu = pw_per_cu(o->kt16, o->kh, o->wr * o->wc, o->wc, o->rw);
  static const int one = [] { const char *e = getenv("NESIE_PW_ONE_PER_CU"); return e ? atoi(e) : 0; }();
  if (one) o->per_cu = 1;   // A/B switch: 256-workgroup grids
  return true;
}

static size_t pw_lds_bytes(const PwGeom &g) { return (size_t)2 * 16 * g.kt16 * g.pt * sizeof(float); }

}  // namespace nesie

using namespace nesie;

#ifdef PW_STAMP
long long *g_pw_stamps = nullptr;
#endif

extern "C" int nesie_pw_supported(int k, int cout, long long p) {
  PwGeom g;
  return pw_geometry(k, cout, &g) && p % g.pt == 0 && (long long)(k > cout ? k : cout) * p < (1ll << 30) ? 1 : 0;
}

// workgroups per weight group of a launch
static int pw_groups(const PwGeom &g, int nb, int ng, long long p) {
  const long long tiles = (long long)(nb / ng) * cdiv(p, g.pt);
  // NESIE_PW_ROUNDS (A/B, default 1): R shorter rounds of workgroups instead of one -- a CU that is
  // not available to this launch (a long-lived tenant of another stream) then delays 1/R of it
  static const int rounds = [] { const char *e = getenv("NESIE_PW_ROUNDS"); return e ? atoi(e) : 1; }();
  long long nwg = (long long)cu_count() * g.per_cu * (rounds > 0 ? rounds : 1) / (ng * g.nhalf);
  if (nwg < 1) nwg = 1;
  if (nwg > tiles) nwg = tiles;
  // several row blocks per tile: a grid that is a multiple of 8 * nhalf lets the row blocks of a
  // tile stream share an XCD (pwconv_fwd.h, xcd_map); round down when that costs < 10 % of the grid
  if (g.nhalf > 1) {
    const long long unit = 8 / std::__gcd((long long)8, (long long)ng);   // nwg * ng % 8 == 0
    const long long r = nwg / unit * unit;
    if (r > 0 && r * 10 >= nwg * 9) nwg = r;
  }
  return (int)nwg;
}

// number of statistic slots per weight group a forward launch writes
extern "C" int nesie_pw_stat_slots(int nb, int ng, int k, int cout, long long p) {
  PwGeom g;
  if (!pw_geometry(k, cout, &g) || ng < 1) return 0;
  return pw_groups(g, nb, ng, p) * g.wc;
}

#ifdef PW_DEV   // single-translation-unit development build (tools/pwbench): two geometries
PW_GEOM_DEF(8, 2, 8, 1, 1)
PW_GEOM_DEF(8, 1, 8, 1, 1)
#endif

static int pw_forward_impl(const char *W, int nb, int ng, int k, int cout, long long p,
                           const float *x, long long x_bstride, const float *w,
                           long long w_gstride, int w_rstride, int w_cstride,
                           const float *in_coef, int in_relu, const float *row_bias,
                           int rb_group, const float *bias, float *y,
                           long long y_bstride, float *stat_part, int pool_group,
                           int pool_min, float *pool_max_out, float *pool_min_out,
                           uint8_t *arg_max_out, uint8_t *arg_min_out, const float *bn_z,
                           long long bnz_bstride, const float *bn_coef, float *bn_part,
                           void *stream, const float *k4_w = nullptr, int k4_in = 0,
                           float *k4_gpart = nullptr) {
  NESIE_REQUIRE(nb >= 0 && ng >= 1 && k >= 1 && cout >= 1 && p >= 0, W);
  if (nb == 0 || p == 0) return NESIE_OK;
  NESIE_REQUIRE(nb % ng == 0 && x && w, W);
  PwGeom g;
  if (!pw_geometry(k, cout, &g) || p % g.pt != 0 || (long long)(k > cout ? k : cout) * p >= (1ll << 30)) {
    set_error("%s: %d -> %d over %lld positions is outside the built tiles", W, k, cout, p);
    return NESIE_ERR_UNSUPPORTED;
  }
  NESIE_REQUIRE(((uintptr_t)x & 15) == 0 && (x_bstride & 3) == 0, W);
  NESIE_REQUIRE(!y || (((uintptr_t)y & 15) == 0 && (y_bstride & 3) == 0), W);
  int epi = 0, pg = 16;
  if (in_coef) epi |= PW_AFFINE;
  if (y) epi |= PW_STORE;
  if (stat_part) epi |= PW_STATS;
  if (row_bias) {
    NESIE_REQUIRE(rb_group >= 16 && (rb_group & (rb_group - 1)) == 0 && p % rb_group == 0, W);
    epi |= PW_ROWBIAS;
  }
  if (bias) epi |= PW_BIAS;
  if (bn_z) {
    NESIE_REQUIRE((y || k4_gpart) && bn_coef && bn_part && ((uintptr_t)bn_z & 15) == 0 && (bnz_bstride & 3) == 0, W);
    epi |= PW_BNRED;
  }
  if (k4_in) {      // the operand is rebuilt from the four rows of x (pwconv_fwd.h, PW_K4IN)
    NESIE_REQUIRE(k4_w && in_coef && k == 64 && ng == 1 && x_bstride >= 4 * p && ((uintptr_t)k4_w & 15) == 0, W);
    epi |= PW_K4IN;
  }
  if (k4_gpart) {   // Z of the reduction is rebuilt from the four rows of bn_z (PW_K4Z)
    NESIE_REQUIRE(k4_w && bn_z && !y && cout == 64 && ng == 1 && bnz_bstride >= 4 * p && ((uintptr_t)k4_w & 15) == 0 &&
                  ((uintptr_t)k4_gpart & 15) == 0, W);
    epi |= PW_K4Z;
  }
  if (pool_group) {
    NESIE_REQUIRE(pool_group == 16 || pool_group == 32, W);
    NESIE_REQUIRE(pool_max_out && arg_max_out && p % pool_group == 0, W);
    NESIE_REQUIRE(!pool_min || (pool_min_out && arg_min_out), W);
    epi |= PW_POOL | (pool_min ? PW_POOLMIN : 0);
    pg = pool_group;
  }
  PwFwd a;
  a.x = x; a.x_bs = x_bstride; a.p = p; a.nb = nb; a.k = k;
  a.w = w; a.w_gs = w_gstride; a.w_rs = w_rstride; a.w_cs = w_cstride; a.ng = ng; a.cout = cout;
  a.in_coef = in_coef; a.in_lo = in_relu ? 0.f : -__builtin_inff();
  a.y = y; a.y_bs = y_bstride;
  a.row_bias = row_bias; a.rb_shift = row_bias ? __builtin_ctz((unsigned)rb_group) : 0;
  a.bias = bias;
  a.stat_part = stat_part;
  a.pool_max = pool_max_out; a.pool_min = pool_min_out; a.arg_max = arg_max_out; a.arg_min = arg_min_out;
  a.bn_z = bn_z; a.bnz_bs = bnz_bstride; a.bn_coef = bn_coef; a.bn_part = bn_part;
  a.stamps = nullptr;
  a.k4_w = k4_w; a.k4_gpart = k4_gpart;
  static const int w_stage = [] { const char *e = getenv("NESIE_PW_WSTAGE"); return e ? atoi(e) : 1; }();
  a.w_stage = w_stage;
  static const int rev_fwd = [] { const char *e = getenv("NESIE_PW_REV_FWD"); return e ? atoi(e) : 3; }();   // A/B: bit 0 forward products, bit 1 input gradients
  a.rev = ((bn_z ? rev_fwd & 2 : rev_fwd & 1) != 0) ? walk_dir((long long)nb * (k4_in ? 4 : k) * p * 4) : 0;     // (a big operand is read last tile first: nesie_lib.hip)
#ifdef PW_STAMP
  a.stamps = g_pw_stamps;
#endif
  a.tiles_per_batch = cdiv(p, g.pt);
  a.nslots = nesie_pw_stat_slots(nb, ng, k, cout, p);
  a.nwg_g = pw_groups(g, nb, ng, p);
  a.nhalf = g.nhalf;
  const int grid = a.nwg_g * ng * g.nhalf;
  static const bool xcd_on = !(getenv("NESIE_PW_XCD") && atoi(getenv("NESIE_PW_XCD")) == 0);   // A/B switch
  a.xcd_map = (xcd_on && g.nhalf > 1 && grid % (8 * g.nhalf) == 0) ? 1 : 0;
  const size_t lds = pw_lds_bytes(g) + (k4_gpart ? 2048 : 0);    // (PW_K4Z: + the tile's X4, two halves)
  hipStream_t s = (hipStream_t)stream;
  int st = NESIE_ERR_UNSUPPORTED;
#define G(KT16, KH, WR, WC, RW)                                                          \
  if (g.kt16 == KT16 && g.kh == KH && g.wr == WR && g.wc == WC && g.rw == RW)            \
  st = PW_GEOM_NAME(KT16, KH, WR, WC, RW)(a, epi, pg, grid, lds, s)
#ifdef PW_DEV
  G(8, 2, 8, 1, 1); G(8, 1, 8, 1, 1);
#else
  G(4, 1, 4, 1, 1); G(4, 1, 4, 1, 2); G(4, 1, 8, 1, 1); G(8, 1, 4, 1, 1); G(8, 1, 8, 1, 1);
  G(9, 1, 8, 1, 1); G(8, 2, 8, 1, 1); G(9, 2, 8, 1, 1); G(8, 4, 8, 1, 1);
#endif
#undef G
  if (st != NESIE_OK) {
    if (st == NESIE_ERR_UNSUPPORTED && !strstr(nesie_last_error(), "epilogue"))
      set_error("%s: no build for %d -> %d", W, k, cout);
    return st;
  }
  return check_launch(W);
}

extern "C" int nesie_pw_layer_forward(int nb, int ng, int k, int cout, long long p,
                                      const float *x, long long x_bstride, const float *w,
                                      long long w_gstride, int w_rstride, int w_cstride,
                                      const float *in_coef, int in_relu, const float *row_bias,
                                      int rb_group, const float *bias, float *y,
                                      long long y_bstride, float *stat_part, int pool_group,
                                      int pool_min, float *pool_max_out, float *pool_min_out,
                                      uint8_t *arg_max_out, uint8_t *arg_min_out, void *stream) {
  return pw_forward_impl("pw_layer_forward", nb, ng, k, cout, p, x, x_bstride, w, w_gstride,
                         w_rstride, w_cstride, in_coef, in_relu, row_bias, rb_group, bias, y,
                         y_bstride, stat_part, pool_group, pool_min, pool_max_out, pool_min_out,
                         arg_max_out, arg_min_out, nullptr, 0, nullptr, nullptr, stream);
}

// Input gradient of a layer, Y[n] = W[n % ng] . X[n] with W the transposed weight view, whose
// consumer is the backward of relu(bn(Z)) (Z = the previous layer's raw output, same shape as
// Y): the epilogue also leaves, per channel and slot, sum(g) and sum(g * zhat) with
// g = Y [fma(Z, scale, bias) > 0] -- the reduction pass of the BatchNorm backward.
// bn_part: [ng * cout][nesie_pw_stat_slots(...)][2], every slot written.
extern "C" int nesie_pw_dgrad_bn_reduce(int nb, int ng, int k, int cout, long long p,
                                        const float *x, long long x_bstride, const float *w,
                                        long long w_gstride, int w_rstride, int w_cstride,
                                        float *y, long long y_bstride, const float *bn_z,
                                        long long bnz_bstride, const float *bn_coef,
                                        float *bn_part, void *stream) {
  const char *W = "pw_dgrad_bn_reduce";
  NESIE_REQUIRE(y && bn_z && bn_coef && bn_part, W);
  return pw_forward_impl(W, nb, ng, k, cout, p, x, x_bstride, w, w_gstride, w_rstride, w_cstride,
                         nullptr, 0, nullptr, 0, nullptr, y, y_bstride, nullptr, 0, 0, nullptr,
                         nullptr, nullptr, nullptr, bn_z, bnz_bstride, bn_coef, bn_part, stream);
}

// SA1's second layer over the REBUILT output of its first: the operand rows are relu(bn(W0 . X4)),
// formed in the staging from the four rows of x4 (nb, 4, p) -- the 64-row tensor is never written
// or read.  w0 (64, 4) row-major; in_coef [64][4] the folded norm of the first layer.
extern "C" int nesie_pw_layer_forward_k4(int nb, int cout, long long p, const float *x4,
                                         long long x4_bstride, const float *w0, const float *w,
                                         int w_rstride, int w_cstride, const float *in_coef,
                                         float *y, long long y_bstride, float *stat_part,
                                         void *stream) {
  return pw_forward_impl("pw_layer_forward_k4", nb, 1, 64, cout, p, x4, x4_bstride, w, 0, w_rstride,
                         w_cstride, in_coef, 1, nullptr, 0, nullptr, y, y_bstride, stat_part, 0, 0,
                         nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, stream, w0, 1);
}

// ... and the input gradient of that second layer, of which only the REDUCTIONS are wanted: the
// gradient g = dA0 [bn(Z0) > 0] is never stored.  Per channel and slot it leaves sum(g),
// sum(g zhat) (bn_part, as nesie_pw_dgrad_bn_reduce) and sum(g X4[j]), j = 0 .. 3 (g_part
// [64][slots][4]): with the moments of X4 they determine the first layer's weight gradient
// (nesie_k4_first_layer_wgrad).
extern "C" int nesie_pw_dgrad_bn_reduce_k4(int nb, int k, long long p, const float *x,
                                           long long x_bstride, const float *w, int w_rstride,
                                           int w_cstride, const float *x4, long long x4_bstride,
                                           const float *w0, const float *bn_coef, float *bn_part,
                                           float *g_part, void *stream) {
  const char *W = "pw_dgrad_bn_reduce_k4";
  NESIE_REQUIRE(x4 && w0 && bn_coef && bn_part && g_part, W);
  return pw_forward_impl(W, nb, 1, k, 64, p, x, x_bstride, w, 0, w_rstride, w_cstride, nullptr, 0,
                         nullptr, 0, nullptr, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, nullptr,
                         nullptr, x4, x4_bstride, bn_coef, bn_part, stream, w0, 0, g_part);
}

extern "C" int nesie_pw_stats_finalize(int channels, int cout, int nslots, const float *stat_part,
                                       const float *gamma, const float *beta,
                                       float *running_mean, float *running_var, float momentum,
                                       float eps, float *coef, const float *chan_bias,
                                       void *stream) {
  const char *W = "pw_stats_finalize";
  NESIE_REQUIRE(channels >= 1 && cout >= 1 && channels % cout == 0 && nslots >= 1, W);
  NESIE_REQUIRE(stat_part && coef && (running_mean == nullptr) == (running_var == nullptr), W);
  hipLaunchKernelGGL(pw_stats_finalize_kernel, dim3(channels), dim3(64), 0, (hipStream_t)stream,
                     cout, nslots, stat_part, gamma, beta, running_mean, running_var, momentum,
                     eps, coef, chan_bias);
  return check_launch(W);
}

static int pw_pool_finish_impl(int nb, int ng, int channels, long long p, int group, int pool_group,
                               const float *pmax, const float *pmin, const uint8_t *amax,
                               const uint8_t *amin, const float *coef, int relu,
                               float *pooled, uint8_t *argmax, float *zstar, void *stream) {
  const char *W = "pw_pool_finish";
  NESIE_REQUIRE(nb >= 0 && ng >= 1 && channels >= 1 && p >= 0 && group >= 1, W);
  if (nb == 0 || p == 0) return NESIE_OK;
  NESIE_REQUIRE((pool_group == 16 || pool_group == 32) && group % pool_group == 0 && p % group == 0, W);
  NESIE_REQUIRE(group <= 256 && pmax && amax && pooled && argmax && (!coef || (pmin && amin)), W);
  const long long total = (long long)nb * channels * (p / group);
  hipLaunchKernelGGL(pw_pool_finish_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     total, channels, ng, (int)(p / group), group / pool_group, pool_group, pmax, pmin,
                     amax, amin, coef, relu ? 0.f : -__builtin_inff(), pooled, argmax, zstar);
  return check_launch(W);
}

extern "C" int nesie_pw_pool_finish(int nb, int ng, int channels, long long p, int group, int pool_group,
                                    const float *pmax, const float *pmin, const uint8_t *amax,
                                    const uint8_t *amin, const float *coef, int relu,
                                    float *pooled, uint8_t *argmax, void *stream) {
  return pw_pool_finish_impl(nb, ng, channels, p, group, pool_group, pmax, pmin, amax, amin, coef, relu,
                             pooled, argmax, nullptr, stream);
}

// ... and the raw extremum each pooled value came from (nb, channels, p / group)
extern "C" int nesie_pw_pool_finish_z(int nb, int ng, int channels, long long p, int group, int pool_group,
                                      const float *pmax, const float *pmin, const uint8_t *amax,
                                      const uint8_t *amin, const float *coef, int relu,
                                      float *pooled, uint8_t *argmax, float *zstar, void *stream) {
  NESIE_REQUIRE(zstar, "pw_pool_finish_z");
  return pw_pool_finish_impl(nb, ng, channels, p, group, pool_group, pmax, pmin, amax, amin, coef, relu,
                             pooled, argmax, zstar, stream);
}
