// Backward of a POOLED TAIL -- the last layer of a set-abstraction shared MLP followed by its
// BatchNorm + ReLU and the max over the ns samples of a group (reference
// mmdet3d/ops/pointnet_modules/point_sa_module.py:136-158, 277-289) -- WITHOUT the dense pre-pool
// tensor: the forward never writes the raw conv output Z (B, C, P) and the backward never forms
// the dense dZ.
//
// With A (K x P) the layer's operand (the previous layer's normalised activation, rebuilt on load
// from its raw output and folded coefficients), Z = W A, and g the gradient the max-pool hands
// back -- ONE non-zero per channel and group, at the arg-max, masked by the ReLU -- the training-mode
// BatchNorm backward is
//     dZ_i = gamma r (g_i - mean(g) - zhat_i mean(g zhat))          r = invstd, zhat = (z - mu) r
//          = ghat_i + alpha + beta z_i        ghat = gamma r g,  beta = -gamma r^2 mean(g zhat),
//                                             alpha = -gamma r mean(g) + gamma r^2 mu mean(g zhat)
// (per channel).  Both sums run over the arg-max positions only, whose z the pooling pass kept, so
//     dA_i = W^T dZ_i = W^T ghat_i + W^T alpha + (W^T diag(beta) W) A_i
//     dW   = sum_i dZ_i A_i^T = sum_i ghat_i A_i^T + alpha s^T + diag(beta) W M
//            with s = sum_i A_i and M = sum_i A_i A_i^T  (K x K)
// need A (which the previous layer's backward reads anyway) and the sparse ghat, nothing of size
// C x P:
//   * pt_reduce / pt_pack / pt_wcat: the two sums (fp64), alpha / beta, the entries
//     (g masked, position) transposed to [n][group][c], and [W^T diag(gamma r) | W^T diag(beta) W], W^T alpha;
//   * the input gradient is ONE launch of the layer kernel whose first C operand rows are BUILT from
//     the entries (pwconv_fwd.h, PW_SPARSE*) and whose other K rows are A: K + C -> K, with the
//     reduction of the previous layer's norm backward in its epilogue as before;
//   * pt_gram_kernel: M on the matrix cores, s, and the sparse product sum ghat A^T on the vector
//     ALUs straight from the LDS tile -- one pass over A;
//   * pt_reduce_ms / pt_dw: fixed-order fp64 combination into dW.
// Per SA1 step (K = 64, C = 128, 1 M positions): 537 MB of Z are neither written nor read twice, the
// 524 MB dense dZ is neither written nor read twice; no float atomics anywhere.
#include "pwconv_fwd.h"

namespace nesie {

int pw_launch_sparse_6_2_4_1_1(const PwFwd &a, int grid, size_t lds, hipStream_t s);

// sums of g and g zhat over a channel's groups -> dgamma, dbeta, (alpha, beta, gamma r)
__global__ __launch_bounds__(1024) void pt_reduce_kernel(int nb, int c_all, int m, double count,
                                                        const float *__restrict__ g,
                                                        const float *__restrict__ pooled,
                                                        const float *__restrict__ zstar,
                                                        const float *__restrict__ coef,
                                                        const float *__restrict__ gamma,
                                                        float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                        float4 *__restrict__ ab) {
  __shared__ double sh0[1024], sh1[1024];
  const int c = blockIdx.x;
  const float mean = coef[c * 4 + 2], invstd = coef[c * 4 + 3];
  double s0 = 0.0, s1 = 0.0;
  const int total = nb * m;
#pragma unroll 4
  for (int i = threadIdx.x; i < total; i += 1024) {
    const size_t o = ((size_t)(i / m) * c_all + c) * m + (i % m);
    const float gg = pooled[o] > 0.f ? g[o] : 0.f;
    s0 += (double)gg;
    s1 += (double)gg * (double)((zstar[o] - mean) * invstd);
  }
  sh0[threadIdx.x] = s0;
  sh1[threadIdx.x] = s1;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      sh0[threadIdx.x] += sh0[threadIdx.x + w];
      sh1[threadIdx.x] += sh1[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double gm = (double)gamma[c], r = (double)invstd;
    const double mg = sh0[0] / count, mgz = sh1[0] / count;
    dbeta[c] = (float)sh0[0];
    dgamma[c] = (float)sh1[0];
    ab[c] = make_float4((float)(-gm * r * mg + gm * r * r * (double)mean * mgz), (float)(-gm * r * r * mgz),
                        (float)(gm * r), 0.f);
  }
}

// ent[n][group][c] = (g masked by the ReLU, position of the arg-max inside the group)
__global__ __launch_bounds__(256) void pt_pack_kernel(int c_all, int m, const float *__restrict__ g,
                                                      const float *__restrict__ pooled,
                                                      const uint8_t *__restrict__ arg,
                                                      float2 *__restrict__ ent) {
  __shared__ float2 tile[32][33];
  const int n = blockIdx.z, c0 = blockIdx.y * 32, m0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int j = m0 + tx;
    if (j < m) {
      const size_t o = ((size_t)n * c_all + c0 + r) * m + j;
      tile[r][tx] = make_float2(pooled[o] > 0.f ? g[o] : 0.f, __int_as_float((int)arg[o]));
    }
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int j = m0 + r;
    if (j < m) ent[((size_t)n * m + j) * c_all + c0 + tx] = tile[tx][r];
  }
}

// wcat[mrow][0 .. C) = W[c][mrow] gamma_c r_c;  wcat[mrow][C + j] = sum_c W[c][mrow] beta_c W[c][j];
// c0[mrow] = sum_c W[c][mrow] alpha_c
__global__ __launch_bounds__(256) void pt_wcat_kernel(int c_sp, int k, const float *__restrict__ w,
                                                      const float4 *__restrict__ ab,
                                                      float *__restrict__ wcat, float *__restrict__ c0) {
  const int mrow = blockIdx.x, ld = c_sp + k;
  for (int col = threadIdx.x; col <= ld; col += 256) {
    if (col < c_sp) {
      wcat[(size_t)mrow * ld + col] = w[(size_t)col * k + mrow] * ab[col].z;
    } else if (col < ld) {
      const int j = col - c_sp;
      double acc = 0.0;
      for (int c = 0; c < c_sp; ++c)
        acc += (double)w[(size_t)c * k + mrow] * (double)ab[c].y * (double)w[(size_t)c * k + j];
      wcat[(size_t)mrow * ld + col] = (float)acc;
    } else {
      double acc = 0.0;
      for (int c = 0; c < c_sp; ++c) acc += (double)w[(size_t)c * k + mrow] * (double)ab[c].x;
      c0[mrow] = (float)acc;
    }
  }
}

// One pass over A = relu(scale x + bias) (K x P per batch element), 64-position tiles (32-position
// tiles: one barrier per 8 KB, 119 us for SA1's 268 MB):
//   part_m[rank] += A_tile A_tile^T          (fp32 MFMA, as pw_wgrad_kernel with both operands the same tile)
//   part_s[rank] += row sums
//   part_w[rank][c] += g[c] A[:, position of c's entry]    for every entry of the tile's group(s)
// wave (cw, kw) of the sparse part: channels 64 cw + lane, rows KPER kw .. + KPER - 1
template <int K16, int CSP>
__global__ __launch_bounds__(512) void pt_gram_kernel(int nb, long long p, const float *__restrict__ x,
                                                      long long x_bs, const float *__restrict__ coef,
                                                      const float2 *__restrict__ ent, int ns_shift,
                                                      int groups, float *__restrict__ part_m,
                                                      float *__restrict__ part_s,
                                                      float *__restrict__ part_w, int nwg) {
  constexpr int K = 16 * K16, PT = 64, PITCH = PT + 4, CPR = PT / 4, NT = 512;
  constexpr int NX = K * CPR / NT;
  static_assert(K * CPR % NT == 0, "tile");
  constexpr int WM = 2, WN = 4, MB = K16 / WM, NB = K16 / WN;
  constexpr int CW = CSP / 64, KW = 8 / CW, KPER = K / KW;
  static_assert(K16 % WM == 0 && K16 % WN == 0 && CW * KW == 8 && KPER * KW == K, "waves");
  constexpr int TILE = K * PITCH;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, quad = lane >> 4, l16 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int cw = wave % CW, kw = wave / CW;
  const int rank = blockIdx.x;

  unsigned goff[NX], lw[NX];
  float sc[NX], bi[NX], rs[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int c = i * NT + tid, row = c / CPR, cp = c % CPR;
    goff[i] = (unsigned)(((size_t)row * p + cp * 4) * 4);
    lw[i] = (unsigned)((row * PITCH + cp * 4) * 4);
    sc[i] = coef[row * 4];
    bi[i] = coef[row * 4 + 1];
    rs[i] = 0.f;
  }
  const int tpb = (int)(p / PT);
  const int ntiles = nb * tpb;
  f32x4 stg[NX];
  bool pend_ok = false;
  auto load_tile = [&](int t) {
    pend_ok = t < ntiles;
    t = t < ntiles ? t : ntiles - 1;
    const float *xb = x + (size_t)(t / tpb) * x_bs + (size_t)(t % tpb) * PT;   // uniform
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      unsigned o = goff[i];
      asm volatile("" : "+v"(o));
      stg[i] = load16_saddr(o, xb);
    }
  };
  auto write_tile = [&](float *buf) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      f32x4 q = stg[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) q[e] = fmaxf(__builtin_fmaf(q[e], sc[i], bi[i]), 0.f);
      rs[i] += pend_ok ? (q[0] + q[1]) + (q[2] + q[3]) : 0.f;
      *(f32x4 *)((char *)buf + lw[i]) = q;
    }
  };
  f32x4 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float wacc[KPER];
#pragma unroll
  for (int kk = 0; kk < KPER; ++kk) wacc[kk] = 0.f;

  const int ngt = 64 >> ns_shift;                // groups per tile: 1 (ns = 64), 2 or 4
  float *b0 = lds, *b1 = lds + TILE;
  int t = rank;
  if (t < ntiles) {
    load_tile(t);
    write_tile(b0);
    load_tile(t + nwg);
  }
  for (; t < ntiles; t += nwg) {
    const int n = t / tpb, p0 = (t % tpb) * PT;
    // this tile's entries (consumed behind the MFMAs): one per channel and group of the tile
    const float2 *eb = ent + ((size_t)n * groups + (p0 >> ns_shift)) * CSP + cw * 64 + lane;
    float2 e[4];
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) e[gi] = gi < ngt ? eb[(size_t)gi * CSP] : make_float2(0.f, 0.f);
    lgkm_wait<0>();
    __builtin_amdgcn_s_barrier();
    const unsigned la = lds_addr(b0) + (unsigned)(((wm * MB * 16 + l16) * PITCH + 4 * quad) * 4);
    const unsigned lb = lds_addr(b0) + (unsigned)(((wn * NB * 16 + l16) * PITCH + 4 * quad) * 4);
    f32x4 fa[MB], fb[NB];
    static_for<0, PT / 16>([&](auto pgc) {
      constexpr int pg = decltype(pgc)::value;
      static_for<0, MB>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        fa[i] = lds_read_b128<(i * 16 * PITCH + 16 * pg) * 4>(la);
      });
      static_for<0, NB>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        fb[j] = lds_read_b128<(j * 16 * PITCH + 16 * pg) * 4>(lb);
      });
      lgkm_wait<0>();
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, 4>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        static_for<0, MB>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          static_for<0, NB>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][c], fb[j][c], acc[i][j], 0, 0, 0);
          });
        });
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    // sparse product: every entry of the tile's group(s) lies inside the tile
    {
      const float *col = b0 + (kw * KPER) * PITCH;
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        if (gi < ngt) {                           // (uniform)
          const int rel = __float_as_int(e[gi].y) + (gi << ns_shift);
#pragma unroll
          for (int kk = 0; kk < KPER; ++kk) wacc[kk] = __builtin_fmaf(e[gi].x, col[kk * PITCH + rel], wacc[kk]);
        }
      }
    }
    write_tile(b1);
    __builtin_amdgcn_sched_barrier(0);
    load_tile(t + 2 * nwg);
    float *const tb = b0; b0 = b1; b1 = tb;
  }
  float *dst = part_m + (size_t)rank * K * K;
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mm = (wm * MB + i) * 16 + 4 * quad + r, kk = (wn * NB + j) * 16 + l16;
        dst[(size_t)mm * K + kk] = acc[i][j][r];
      }
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    float v = rs[i];
#pragma unroll
    for (int o = 1; o < CPR; o <<= 1) v += __shfl_xor(v, o, 64);
    if ((tid & (CPR - 1)) == 0) part_s[(size_t)rank * K + (i * NT + tid) / CPR] = v;
  }
  float *wd = part_w + ((size_t)rank * CSP + cw * 64 + lane) * K + kw * KPER;
#pragma unroll
  for (int kk = 0; kk < KPER; kk += 4)
    *(f32x4 *)(wd + kk) = (f32x4){wacc[kk], wacc[kk + 1], wacc[kk + 2], wacc[kk + 3]};
}

// ms[0 .. K K) = sum over the partials of M, ms[K K .. K K + K) of s (fp64, fixed order: 16 strided
// slices per output, then the slices in order)
__global__ __launch_bounds__(1024) void pt_reduce_ms_kernel(int k, int nwg, const float *__restrict__ part_m,
                                                            const float *__restrict__ part_s,
                                                            double *__restrict__ ms) {
  __shared__ double sh[16][64];
  const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o, kk = k * k;
  double acc = 0.0;
  if (i < kk) {
#pragma unroll 4
    for (int r = sl; r < nwg; r += 16) acc += (double)part_m[(size_t)r * kk + i];
  } else if (i < kk + k) {
#pragma unroll 4
    for (int r = sl; r < nwg; r += 16) acc += (double)part_s[(size_t)r * k + (i - kk)];
  }
  sh[sl][o] = acc;
  __syncthreads();
  if (sl == 0 && i < kk + k) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sh[q][o];
    ms[i] = t;
  }
}

// dW[c][k] = gamma_c r_c sum_r part_w[r][c][k] + alpha_c s[k] + beta_c sum_j W[c][j] M[j][k]
// (1024 / k strided slices per output, combined in order)
__global__ __launch_bounds__(1024) void pt_dw_kernel(int c_sp, int k, int nwg, const float *__restrict__ part_w,
                                                     const double *__restrict__ ms,
                                                     const float4 *__restrict__ ab,
                                                     const float *__restrict__ w, float *__restrict__ dw) {
  __shared__ double sh0[1024], sh1[1024];
  const int c = blockIdx.x, col = threadIdx.x % k, sl = threadIdx.x / k, nsl = 1024 / k;
  double acc = 0.0, tacc = 0.0;
#pragma unroll 4
  for (int r = sl; r < nwg; r += nsl) acc += (double)part_w[((size_t)r * c_sp + c) * k + col];
  for (int j = sl; j < k; j += nsl) tacc += (double)w[(size_t)c * k + j] * ms[(size_t)j * k + col];
  sh0[threadIdx.x] = acc;
  sh1[threadIdx.x] = tacc;
  __syncthreads();
  if (sl == 0) {
    double a0 = 0.0, a1 = 0.0;
    for (int q = 0; q < nsl; ++q) {
      a0 += sh0[q * k + col];
      a1 += sh1[q * k + col];
    }
    const float4 co = ab[c];
    dw[(size_t)c * k + col] = (float)(a0 * (double)co.z + (double)co.x * ms[(size_t)k * k + col] + (double)co.y * a1);
  }
}

// Input-gradient launch: K + C -> K with the C rows built from the entries.  Built for SA1's tail
// (64 -> 128), the HBM-bound one: there the extra C / K of matrix work hides under the operand
// stream.  (The K = 128 tails are MFMA-bound: 1.5 x the products of the dense form would cost what
// the saved passes gain; a sparse term built with LDS float atomics was 3 - 6 x slower,
// tools/abl/pooled_tail_sparse_term_lds_atomics.patch.)
struct PtGeom { int kt16, kh, wr, wc, rw; size_t lds; int per_cu; };
static bool pt_geometry(int k, int c, PtGeom *o) {
  if (!(k == 64 && c == 128)) return false;
  *o = PtGeom{6, 2, 4, 1, 1, (size_t)2 * 96 * 64 * 4, 0};
  o->per_cu = pw_per_cu(o->kt16, o->kh, o->wr * o->wc, o->wc, o->rw);
  return true;
}
static int pt_ns_shift(int ns) { return ns == 16 ? 4 : ns == 32 ? 5 : ns == 64 ? 6 : -1; }
static int pt_dgrad_groups(const PtGeom &g, int nb, long long p) {
  const long long tiles = (long long)nb * (p / (64 * g.wc));
  long long nwg = (long long)cu_count() * g.per_cu;
  return (int)(nwg < tiles ? nwg : tiles);
}
static int pt_gram_groups(int nb, long long p) {
  const long long tiles = (long long)nb * (p / 64);
  return (int)(tiles < 512 ? tiles : 512);
}

}  // namespace nesie

using namespace nesie;

extern "C" int nesie_pool_tail_supported(int k, int c, long long p, int ns) {
  PtGeom g;
  return pt_geometry(k, c, &g) && pt_ns_shift(ns) >= 0 && p > 0 && p % 64 == 0 && p % ns == 0 &&
                 (long long)(k + c) * p < (1ll << 30)
             ? 1 : 0;
}

// out[0] = reduction slots of the input-gradient launch (bn_part is [k][out[0]][2]),
// out[1] = partials of the weight-gradient pass (part_m [out[1]][k][k], part_s [out[1]][k], part_w [out[1]][c][k])
extern "C" int nesie_pool_tail_sizes(int nb, int k, int c, long long p, int *out) {
  const char *W = "pool_tail_sizes";
  PtGeom g;
  NESIE_REQUIRE(out && pt_geometry(k, c, &g) && nb >= 1 && p >= 64, W);
  out[0] = pt_dgrad_groups(g, nb, p) * g.wc;
  out[1] = pt_gram_groups(nb, p);
  return NESIE_OK;
}

extern "C" int nesie_pool_tail_prepare(int nb, int c, int m, int ns, int k, const float *grad_pooled,
                                       const float *pooled, const float *zstar, const uint8_t *argmax,
                                       const float *coef, const float *gamma, const float *w,
                                       float *dgamma, float *dbeta, float *ab, float *ent, float *wcat,
                                       float *c0, void *stream) {
  const char *W = "pool_tail_prepare";
  NESIE_REQUIRE(nb >= 1 && c >= 32 && c % 32 == 0 && m >= 1 && k >= 1 && pt_ns_shift(ns) >= 0, W);
  NESIE_REQUIRE(grad_pooled && pooled && zstar && argmax && coef && gamma && w && dgamma && dbeta && ab && ent && wcat && c0, W);
  NESIE_REQUIRE(nb <= 65535 && ((uintptr_t)ab & 15) == 0 && ((uintptr_t)ent & 7) == 0, W);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(pt_reduce_kernel, dim3(c), dim3(1024), 0, s, nb, c, m, (double)nb * (double)m * (double)ns,
                     grad_pooled, pooled, zstar, coef, gamma, dgamma, dbeta, (float4 *)ab);
  hipLaunchKernelGGL(pt_pack_kernel, dim3(cdiv(m, 32), c / 32, nb), dim3(256), 0, s, c, m, grad_pooled, pooled,
                     argmax, (float2 *)ent);
  hipLaunchKernelGGL(pt_wcat_kernel, dim3(k), dim3(256), 0, s, c, k, w, (const float4 *)ab, wcat, c0);
  return check_launch(W);
}

extern "C" int nesie_pool_tail_dgrad(int nb, int k, int c, long long p, int ns, const float *z_prev,
                                     long long z_bstride, const float *coef_prev, const float *wcat,
                                     const float *c0, const float *ent, float *da, long long da_bstride,
                                     float *bn_part, void *stream) {
  const char *W = "pool_tail_dgrad";
  PtGeom g;
  NESIE_REQUIRE(nb >= 1 && nesie_pool_tail_supported(k, c, p, ns) && pt_geometry(k, c, &g), W);
  NESIE_REQUIRE(z_prev && coef_prev && wcat && c0 && ent && da && bn_part, W);
  NESIE_REQUIRE((((uintptr_t)z_prev | (uintptr_t)da) & 15) == 0 && ((z_bstride | da_bstride) & 3) == 0 && ((uintptr_t)ent & 7) == 0, W);
  PwFwd a;
  memset(&a, 0, sizeof(a));
  a.x = z_prev; a.x_bs = z_bstride; a.p = p; a.nb = nb; a.k = k + c;
  a.w = wcat; a.w_gs = 0; a.w_rs = k + c; a.w_cs = 1; a.ng = 1; a.cout = k;
  a.in_coef = coef_prev; a.in_lo = 0.f;
  a.y = da; a.y_bs = da_bstride;
  a.bias = c0;
  a.bn_z = z_prev; a.bnz_bs = z_bstride; a.bn_coef = coef_prev; a.bn_part = bn_part;
  a.tiles_per_batch = (int)(p / (64 * g.wc));
  a.nwg_g = pt_dgrad_groups(g, nb, p);
  a.nslots = a.nwg_g * g.wc;
  a.nhalf = 1;
  a.xcd_map = 0;
  a.w_stage = 0;       // (96-row sub-tiles: lane loads)
  a.rev = 0;
  a.k4_w = nullptr; a.k4_gpart = nullptr;           // (its operand is the forward pass's: nothing of it is cached either way)
  a.sp_ent = (const float2 *)ent; a.sp_ns_shift = pt_ns_shift(ns); a.sp_groups = (int)(p / ns);
  const int st = pw_launch_sparse_6_2_4_1_1(a, a.nwg_g, g.lds, (hipStream_t)stream);
  if (st != NESIE_OK) return st;
  return check_launch(W);
}

extern "C" int nesie_pool_tail_wgrad(int nb, int k, int c, long long p, int ns, const float *z_prev,
                                     long long z_bstride, const float *coef_prev, const float *ent,
                                     const float *ab, const float *w, float *part_m, float *part_s,
                                     float *part_w, double *ms, float *dw, void *stream) {
  const char *W = "pool_tail_wgrad";
  NESIE_REQUIRE(nb >= 1 && nesie_pool_tail_supported(k, c, p, ns), W);
  NESIE_REQUIRE(z_prev && coef_prev && ent && ab && w && part_m && part_s && part_w && ms && dw, W);
  NESIE_REQUIRE(((uintptr_t)z_prev & 15) == 0 && (z_bstride & 3) == 0 && ((uintptr_t)part_w & 15) == 0 && (long long)k * p < (1ll << 30), W);
  hipStream_t s = (hipStream_t)stream;
  const int nwg = pt_gram_groups(nb, p);
  const int shift = pt_ns_shift(ns);
  const size_t lds = (size_t)2 * k * 68 * sizeof(float);
#define GRAM(K16, C)                                                                                     \
  hipLaunchKernelGGL((pt_gram_kernel<K16, C>), dim3(nwg), dim3(512), lds, s, nb, p, z_prev, z_bstride,   \
                     coef_prev, (const float2 *)ent, shift, (int)(p / ns), part_m, part_s, part_w, nwg)
  if (k == 64 && c == 128) {
    GRAM(4, 128);
  } else {
    set_error("%s: no build for %d x %d", W, c, k);
    return NESIE_ERR_UNSUPPORTED;
  }
#undef GRAM
  hipLaunchKernelGGL(pt_reduce_ms_kernel, dim3(cdiv(k * k + k, 64)), dim3(1024), 0, s, k, nwg, part_m, part_s, ms);
  hipLaunchKernelGGL(pt_dw_kernel, dim3(c), dim3(1024), 0, s, c, k, nwg, part_w, (const double *)ms,
                     (const float4 *)ab, w, dw);
  return check_launch(W);
}
