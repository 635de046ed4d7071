// Pointwise (1x1) convolution layers of the grouped per-seed MLPs on the fp32 matrix cores of
// gfx950: the shared MLPs of the set-abstraction stack (reference mmdet3d/ops/pointnet_modules/
// point_sa_module.py:277-289 built from mmcv ConvModule(Conv2d 1x1, BN2d, ReLU), pooled at
// :136-158) and the quality head's MiniPointNets (models/dense_heads/side_pooling_module.py:
// 343-370).
//
//     Y[n] = W[n % ng] . act(X[n])        X[n] (K x P) and Y[n] (Cout x P) row-major, positions
//                                         contiguous (NCHW as it stands: no transposes)
//     act(v) = max(scale[k] * v + bias[k], lo)   the PREVIOUS layer's folded BatchNorm + ReLU
//
// Design (weight-stationary, operand streamed by DMA through a 3-deep LDS ring):
//   * a workgroup is persistent and owns a run of (n, position-tile) tiles of ONE weight group;
//     wave (wr, wc) keeps its 16 output rows of W in registers for the whole launch as the A
//     operand of v_mfma_f32_16x16x4_f32 (lane l: W[16 wr + (l & 15)][4 kk + (l >> 4)], K/4
//     registers: 64 at K = 256, so 8-16 waves per CU fit without spilling);
//   * an X tile is K rows x PT positions ~ 32 KB.  Iteration t: tile t+2 is being copied
//     HBM -> LDS by global_load_lds (16 bytes per lane, no VGPR staging), tile t+1 (landed) gets
//     the previous layer's BatchNorm + ReLU applied IN PLACE, once per element, by all threads
//     (2-4 float4 each), tile t feeds the MFMAs.  One barrier per tile.
//   * B operand fetch (lane l: X[4 kk + (l >> 4)][pos + (l & 15)]) = 4 rows x 16 consecutive
//     words; odd rows are stored with their 64-byte halves swapped (the swap is applied to the
//     per-lane SOURCE address of the copy, LDS stays lane-linear) so rows r and r+1 sit on
//     disjoint bank halves: conflict-free ds_read2st64_b32, issued a group ahead with counted
//     lgkmcnt waits (explicit instructions: left to itself hipcc sinks the reads to their uses).
//   * >= 2 accumulator chains per wave (position blocks of 16) cover the 40-cycle dependent
//     latency of the 32-cycle instruction.
//   * epilogue straight from the 4-register accumulators: optional output-side row bias /
//     channel bias, the raw conv output, this layer's own BatchNorm statistics as per-wave
//     SHIFTED sums (count, shift, sum(y - shift), sum((y - shift)^2): no E[x^2] - E[x]^2
//     cancellation; merged in fp64 by pw_stats_finalize_kernel with Chan's formula), and the
//     max / min over each group of 16 or 32 consecutive positions with the position of each
//     (the pooling tail: a 16-position block is exactly one DPP row).
// The K x P operand is read once, Y written once (or never, for a pooled tail): 2 tensor
// passes per layer where conv + statistics + normalise cost 5.
#include "common.h"
#include <string.h>
#include <type_traits>

namespace nesie {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ unsigned lds_addr(const void *p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}

// two words 256 * (U1 - U0) bytes apart in one instruction (offsets in units of 256 bytes)
template <int U0, int U1>
__device__ __forceinline__ f32x2 lds_read2st64(unsigned addr) {
  static_assert(U0 >= 0 && U1 < 256, "ds_read2st64 reach");
  f32x2 v;
  asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(U0), "n"(U1));
  return v;
}

template <int N>
__device__ __forceinline__ void lgkm_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

enum : int {
  PW_STORE = 1,     // write Y
  PW_STATS = 2,     // shifted-sum partials of Y
  PW_POOL = 4,      // max over groups of PG positions (+ position)
  PW_POOLMIN = 8,   // ... and the min (a following BatchNorm's scale may be negative)
  PW_ROWBIAS = 16,  // Y += row_bias[n][m][pos / rb_group] before anything else
  PW_BIAS = 32,     // Y += bias[m]
  PW_AFFINE = 64,   // operand = max(scale * x + bias, lo)
};

struct PwFwd {
  const float *x; long long x_bs; long long p; int nb, k;
  const float *w; long long w_gs; int w_rs, w_cs; int ng, cout;
  const float *in_coef; float in_lo;       // [ng * k][4]; lo = 0 (ReLU) or -inf
  float *y; long long y_bs;
  const float *row_bias; int rb_shift;     // (nb, cout, p >> rb_shift)
  const float *bias;                       // [ng * cout]
  float *stat_part; int nslots;            // [ng][nslots][cout][4]
  float *pool_max, *pool_min; uint8_t *arg_max, *arg_min;  // (nb, cout, p / PG)
  int tiles_per_batch, nwg_g;
};

// max / min over the 16 lanes of a DPP row, result in every lane of the row
template <bool MAX>
__device__ __forceinline__ float row16_reduce(float v) {
#define STEP(CTRL)                                                                            \
  {                                                                                           \
    const float o = __builtin_bit_cast(                                                       \
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true)); \
    v = MAX ? fmaxf(v, o) : fminf(v, o);                                                      \
  }
  STEP(0xB1) STEP(0x4E) STEP(0x141) STEP(0x140)
#undef STEP
  return v;
}

__device__ __forceinline__ float row16_sum(float v) {
#define STEP(CTRL)                                                                            \
  v += __builtin_bit_cast(                                                                    \
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
  STEP(0xB1) STEP(0x4E) STEP(0x141) STEP(0x140)
#undef STEP
  return v;
}

// KQ = padded K / 4; WR x WC waves (16 output rows each x PT / WC positions); PT positions
// per tile; EPI = epilogue / prologue flags; PG = pooling granule (16 or 32)
template <int KQ, int WR, int WC, int PT, int EPI, int PG>
__global__ __launch_bounds__(WR *WC * 64) void pw_fwd_kernel(const PwFwd a) {
  constexpr int NW = WR * WC, NT = NW * 64, KPAD = 4 * KQ, NBLK = PT / 16 / WC;
  constexpr int TILE = KPAD * PT, CPR = PT / 4;          // floats per buffer, 16-byte chunks per row
  constexpr int NI = (KPAD * CPR + 64 * NW - 1) / (64 * NW);  // copy instructions per wave and tile
  constexpr int NX = (KPAD * CPR + NT - 1) / NT;              // transform chunks per thread and tile
  static_assert(NBLK >= 1 && PT % (16 * WC) == 0 && PT >= 32, "tile");
  static_assert(!(EPI & PW_POOL) || PG == 16 || NBLK % 2 == 0, "a 32-position pool needs block pairs");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float2 *const sb = (float2 *)(lds + 3 * TILE);  // [KPAD] (scale, bias) of the operand rows

  const int tid = threadIdx.x, lane = tid & 63, quad = lane >> 4, l16 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const int g = blockIdx.x % a.ng, rank = blockIdx.x / a.ng;
  const int k = a.k, cout = a.cout;
  const long long p = a.p;

  // the rows K..KPAD-1 of the three buffers are never copied into and must read as 0
  if (k < KPAD) {
    const int pad = (KPAD - k) * PT;
    for (int i = tid; i < 3 * pad; i += NT) lds[(i / pad) * TILE + k * PT + i % pad] = 0.f;
  }
  if (EPI & PW_AFFINE)
    for (int i = tid; i < KPAD; i += NT)
      sb[i] = i < k ? make_float2(a.in_coef[((size_t)g * k + i) * 4], a.in_coef[((size_t)g * k + i) * 4 + 1])
                    : make_float2(0.f, 0.f);
  // this wave's 16 rows of W, for the whole launch
  float wreg[KQ];
  {
    const int m = wr * 16 + l16;
    const float *wg = a.w + (size_t)g * a.w_gs + (size_t)m * a.w_rs;
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk) {
      const int kx = 4 * kk + quad;
      wreg[kk] = (m < cout && kx < k) ? wg[(size_t)kx * a.w_cs] : 0.f;
    }
  }
  // per-lane source offsets of the NI copy instructions of a tile (tile-independent).  LDS
  // chunk position cp of row r receives source chunk cp ^ ((r & 1) << 2): odd rows carry their
  // 64-byte halves swapped.
  unsigned coff[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = (i * NW + wave) * 64 + lane;
    const int row = c / CPR, cp = c % CPR;
    const int col = (cp ^ ((row & 1) << 2)) * 4;
    coff[i] = row < k ? (unsigned)((size_t)row * p + col) : 0xFFFFFFFFu;
  }
  const int tpb = a.tiles_per_batch, nwg = a.nwg_g;
  const int ntiles = (a.nb / a.ng) * tpb;
  auto issue = [&](int t, float *buf) {
    const int n = g + a.ng * (t / tpb);
    const long long p0 = (long long)(t % tpb) * PT;
    const float *xb = a.x + (size_t)n * a.x_bs + p0;
    const int left = (int)(p - p0 < PT ? p - p0 : PT);  // positions of this tile that exist
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c0 = (i * NW + wave) * 64;
      const int row = (c0 + lane) / CPR;
      const int col = (((c0 + lane) % CPR) ^ ((row & 1) << 2)) * 4;
      if (coff[i] != 0xFFFFFFFFu && col < left)
        __builtin_amdgcn_global_load_lds((gptr_t *)(xb + coff[i]), (lptr_t *)(buf + c0 * 4), 16, 0, 0);
    }
  };
  // previous layer's BatchNorm + ReLU, in place, once per element
  auto transform = [&](float *buf) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int c = i * NT + tid;
      if (NX * NT == KPAD * CPR || c < KPAD * CPR) {
        const float2 co = sb[c / CPR];
        float4 v = *(float4 *)(buf + c * 4);
        v.x = fmaxf(v.x * co.x + co.y, a.in_lo); v.y = fmaxf(v.y * co.x + co.y, a.in_lo);
        v.z = fmaxf(v.z * co.x + co.y, a.in_lo); v.w = fmaxf(v.w * co.x + co.y, a.in_lo);
        *(float4 *)(buf + c * 4) = v;
      }
    }
  };

  // statistics state: a lane holds 4 channels (rows 4 quad + r) of its position column
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, shift[4] = {0.f, 0.f, 0.f, 0.f};
  int nblk_done = 0;

  float *b0 = lds, *b1 = lds + TILE, *b2 = lds + 2 * TILE;
  int t = rank;
  if (t < ntiles) issue(t, b0);
  if (t + nwg < ntiles) issue(t + nwg, b1);
  __syncthreads();          // pads, coefficients (and, with it, the first two copies: vmcnt(0))
  if (EPI & PW_AFFINE) {
    if (t < ntiles) transform(b0);
  }
  for (; t < ntiles; t += nwg) {
    // b0: tile t (transformed), b1: tile t + nwg (copy issued one iteration ago), b2: free
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 2 * nwg < ntiles) issue(t + 2 * nwg, b2);
    if (EPI & PW_AFFINE) {
      if (t + nwg < ntiles) transform(b1);
    }
    const int n = g + a.ng * (t / tpb);
    const long long p0 = (long long)(t % tpb) * PT;
    const int q0 = wc * NBLK * 16;              // this wave's first position inside the tile
    if (p0 + q0 < p) {
      // ---- MFMA loop: NBLK blocks of 16 positions, K in groups of GK quads
      constexpr int GK = NBLK >= 8 ? 2 : 4, NGRP = (KQ + GK - 1) / GK, UPK = PT / 16;   // 256-byte units per kk
      f32x4 acc[NBLK];
#pragma unroll
      for (int j = 0; j < NBLK; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // lane address: row quad (odd rows: halves swapped), position q0 + 16 j + l16
      unsigned xa[NBLK];
#pragma unroll
      for (int j = 0; j < NBLK; ++j)
        xa[j] = lds_addr(b0) + (unsigned)(quad * PT + ((q0 + 16 * j + l16) ^ ((quad & 1) << 4))) * 4u;
      f32x2 bq[2][NBLK][GK / 2];
      auto load_group = [&](auto gic) {
        constexpr int gi = decltype(gic)::value;
        static_for<0, GK / 2>([&](auto ic) {
          constexpr int i = decltype(ic)::value, kk = gi * GK + 2 * i;
          if constexpr (kk < KQ) {
            // kk + 1 == KQ (odd KQ): the second word is a dummy re-read of kk
            constexpr int k1 = kk + 1 < KQ ? kk + 1 : kk;
            static_for<0, NBLK>([&](auto jc) {
              constexpr int j = decltype(jc)::value;
              bq[gi & 1][j][i] = lds_read2st64<kk * UPK, k1 * UPK>(xa[j]);
            });
          }
        });
      };
      load_group(std::integral_constant<int, 0>{});
      static_for<0, NGRP>([&](auto gic) {
        constexpr int gi = decltype(gic)::value;
        if constexpr (gi + 1 < NGRP) {
          load_group(std::integral_constant<int, gi + 1>{});
          constexpr int nk = KQ - (gi + 1) * GK < GK ? KQ - (gi + 1) * GK : GK;
          lgkm_wait<((nk + 1) / 2) * NBLK>();
        } else {
          lgkm_wait<0>();
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, GK>([&](auto ic) {
          constexpr int i = decltype(ic)::value, kk = gi * GK + i;
          if constexpr (kk < KQ) {
            static_for<0, NBLK>([&](auto jc) {
              constexpr int j = decltype(jc)::value;
              acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[kk], bq[gi & 1][j][i / 2][i & 1], acc[j], 0, 0, 0);
            });
          }
        });
        __builtin_amdgcn_sched_barrier(0);
      });

      // ---- epilogue: lane holds rows mb + r (r = 0..3) of position p0 + q0 + 16 j + l16
      const int mb = wr * 16 + 4 * quad;
#pragma unroll
      for (int j = 0; j < NBLK; ++j) {
        const long long pos = p0 + q0 + 16 * j + l16;
        if (p0 + q0 + 16 * j >= p) break;
        if (EPI & (PW_ROWBIAS | PW_BIAS)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = mb + r;
            if (m < cout) {
              if (EPI & PW_ROWBIAS)
                acc[j][r] += a.row_bias[((size_t)n * cout + m) * (size_t)(p >> a.rb_shift) + (pos >> a.rb_shift)];
              if (EPI & PW_BIAS) acc[j][r] += a.bias[g * cout + m];
            }
          }
        }
        if (EPI & PW_STORE) {
          float *yb = a.y + (size_t)n * a.y_bs + pos;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (mb + r < cout) yb[(size_t)(mb + r) * p] = acc[j][r];
        }
        if (EPI & PW_STATS) {
          if (nblk_done == 0) {
            // shift = the first value this wave sees of each channel (lane 0 of the row)
#pragma unroll
            for (int r = 0; r < 4; ++r) shift[r] = __shfl(acc[j][r], lane & 48, 64);
          }
          ++nblk_done;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float d = acc[j][r] - shift[r];
            s1[r] += d;
            s2[r] += d * d;
          }
        }
      }
      if (EPI & PW_POOL) {
        // PG = 16: a position block is one DPP row; PG = 32: blocks (j, j + 1) of the same lane
        const size_t prow = (size_t)(p / PG);
#pragma unroll
        for (int j = 0; j < NBLK; j += PG / 16) {
          if (p0 + q0 + 16 * j >= p) break;
          const size_t pcol = (size_t)((p0 + q0 + 16 * j) / PG);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int mm = 0; mm < ((EPI & PW_POOLMIN) ? 2 : 1); ++mm) {
              const float v0 = acc[j][r];
              float e = mm ? row16_reduce<false>(v0) : row16_reduce<true>(v0);
              int first;
              if (PG == 32) {
                const float v1 = acc[j + PG / 16 - 1][r];
                const float e1 = mm ? row16_reduce<false>(v1) : row16_reduce<true>(v1);
                e = mm ? fminf(e, e1) : fmaxf(e, e1);
                const unsigned h0 = (unsigned)(__ballot(v0 == e) >> (lane & 48)) & 0xFFFFu;
                const unsigned h1 = (unsigned)(__ballot(v1 == e) >> (lane & 48)) & 0xFFFFu;
                first = h0 ? __ffs(h0) - 1 : 15 + __ffs(h1);
              } else {
                const unsigned h0 = (unsigned)(__ballot(v0 == e) >> (lane & 48)) & 0xFFFFu;
                first = __ffs(h0) - 1;
              }
              if (l16 == 0 && mb + r < cout) {
                const size_t o = ((size_t)n * cout + mb + r) * prow + pcol;
                (mm ? a.pool_min : a.pool_max)[o] = e;
                (mm ? a.arg_min : a.arg_max)[o] = (uint8_t)first;
              }
            }
          }
        }
      }
    }
    float *const tb = b0; b0 = b1; b1 = b2; b2 = tb;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (EPI & PW_STATS) {
    // one partial per wave: (count, shift, sum, sum of squares) of its 16 channels
    const int slot = rank * WC + wc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float u = row16_sum(s1[r]), v = row16_sum(s2[r]);
      const int m = wr * 16 + 4 * quad + r;
      if (l16 == 0 && m < cout) {
        float4 o;
        o.x = (float)nblk_done * 16.f;
        o.y = shift[r];
        o.z = u; o.w = v;
        *(float4 *)(a.stat_part + (((size_t)g * a.nslots + slot) * cout + m) * 4) = o;
      }
    }
  }
}

// Chan merge of the per-wave partials -> (scale, bias, mean, invstd) + running statistics.
// One 64-thread block per channel (channel index runs over ng * cout: stacked layers).
__global__ __launch_bounds__(64) void pw_stats_finalize_kernel(
    int cout, int nslots, const float *__restrict__ part, const float *__restrict__ gamma,
    const float *__restrict__ beta, float *running_mean, float *running_var, float momentum,
    float eps, float *__restrict__ coef) {
  const int ch = blockIdx.x, g = ch / cout, m = ch % cout;
  const float4 *pp = (const float4 *)part + ((size_t)g * nslots) * cout + m;
  double n = 0.0, sm = 0.0;
  for (int i = threadIdx.x; i < nslots; i += 64) {
    const float4 q = pp[(size_t)i * cout];
    if (q.x > 0.f) {
      n += (double)q.x;
      sm += (double)q.x * (double)q.y + (double)q.z;   // n_i * mean_i
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    n += __shfl_xor(n, off, 64);
    sm += __shfl_xor(sm, off, 64);
  }
  const double mean = n > 0.0 ? sm / n : 0.0;
  double m2 = 0.0;
  for (int i = threadIdx.x; i < nslots; i += 64) {
    const float4 q = pp[(size_t)i * cout];
    if (q.x > 0.f) {
      const double ni = q.x, mi = (double)q.y + (double)q.z / ni;
      const double m2i = (double)q.w - (double)q.z * (double)q.z / ni;
      m2 += (m2i > 0.0 ? m2i : 0.0) + ni * (mi - mean) * (mi - mean);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m2 += __shfl_xor(m2, off, 64);
  if (threadIdx.x != 0) return;
  const double var = n > 0.0 ? m2 / n : 0.0;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  if (running_mean) {
    running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * mean);
    const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
    running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * unbiased);
  }
  const double gm = gamma ? (double)gamma[ch] : 1.0, bt = beta ? (double)beta[ch] : 0.0;
  coef[ch * 4 + 0] = (float)(gm * invstd);
  coef[ch * 4 + 1] = (float)(bt - mean * gm * invstd);
  coef[ch * 4 + 2] = (float)mean;
  coef[ch * 4 + 3] = (float)invstd;
}

// tile geometry of a (K, Cout) layer
struct PwGeom { int kq, wr, wc, pt; };

static bool pw_geometry(int k, int cout, PwGeom *o) {
  const int kq = k <= 64 ? 16 : k <= 128 ? 32 : k <= 132 ? 33 : k <= 256 ? 64 : k <= 260 ? 65 : 0;
  const int wr = cout <= 64 ? 4 : cout <= 128 ? 8 : cout <= 256 ? 16 : 0;
  if (!kq || !wr) return false;
  o->kq = kq; o->wr = wr;
  o->pt = kq == 16 ? 128 : kq <= 33 ? 64 : 32;
  // >= 8 waves per workgroup, >= 2 position blocks per wave
  o->wc = wr == 4 ? 2 : 1;
  return true;
}

static size_t pw_lds_bytes(const PwGeom &g) {
  return ((size_t)3 * 4 * g.kq * g.pt + 2 * 4 * g.kq) * sizeof(float);
}

}  // namespace nesie

using namespace nesie;

extern "C" int nesie_pw_supported(int k, int cout, long long p) {
  PwGeom g;
  return pw_geometry(k, cout, &g) && p % 32 == 0 && (long long)k * p < (1ll << 32) ? 1 : 0;
}

// number of statistic slots per weight group a forward launch writes
extern "C" int nesie_pw_stat_slots(int nb, int ng, int k, int cout, long long p) {
  PwGeom g;
  if (!pw_geometry(k, cout, &g) || ng < 1) return 0;
  const long long tiles = (long long)(nb / ng) * cdiv(p, g.pt);
  long long nwg = 256 / ng;
  if (nwg < 1) nwg = 1;
  if (nwg > tiles) nwg = tiles;
  return (int)nwg * g.wc;
}

template <int KQ, int WR, int WC, int PT>
static int pw_launch_epi(const PwFwd &a, int epi, int pg, int grid, size_t lds, hipStream_t s) {
#define GO(E, G)                                                                              \
  do {                                                                                        \
    auto kern = pw_fwd_kernel<KQ, WR, WC, PT, E, G>;                                          \
    static bool attr = false;                                                                 \
    if (!attr) {                                                                              \
      (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                    \
      attr = true;                                                                            \
    }                                                                                         \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WR *WC * 64), lds, s, a);                       \
    return NESIE_OK;                                                                          \
  } while (0)
  const int aff = epi & PW_AFFINE;
  const int base = epi & ~PW_AFFINE;
  constexpr bool P32 = (PT / 16 / WC) % 2 == 0;   // block pairs exist
  // the prologue / epilogue combinations the step uses
  if (aff) {
    if (base == PW_STORE) GO(PW_AFFINE | PW_STORE, 16);
    if (base == (PW_STORE | PW_STATS)) GO(PW_AFFINE | PW_STORE | PW_STATS, 16);
    if (base == (PW_STORE | PW_POOL) && pg == 16) GO(PW_AFFINE | PW_STORE | PW_POOL, 16);
    if (base == PW_POOL && pg == 16) GO(PW_AFFINE | PW_POOL, 16);
    if (base == (PW_STORE | PW_STATS | PW_POOL | PW_POOLMIN) && pg == 16)
      GO(PW_AFFINE | PW_STORE | PW_STATS | PW_POOL | PW_POOLMIN, 16);
    if constexpr (P32) {
      if (base == (PW_STORE | PW_POOL) && pg == 32) GO(PW_AFFINE | PW_STORE | PW_POOL, 32);
      if (base == PW_POOL && pg == 32) GO(PW_AFFINE | PW_POOL, 32);
      if (base == (PW_STORE | PW_STATS | PW_POOL | PW_POOLMIN) && pg == 32)
        GO(PW_AFFINE | PW_STORE | PW_STATS | PW_POOL | PW_POOLMIN, 32);
    }
  } else {
    if (base == PW_STORE) GO(PW_STORE, 16);
    if (base == (PW_STORE | PW_STATS)) GO(PW_STORE | PW_STATS, 16);
    if (base == (PW_STORE | PW_STATS | PW_ROWBIAS)) GO(PW_STORE | PW_STATS | PW_ROWBIAS, 16);
  }
#undef GO
  set_error("pw_layer_forward: epilogue combination 0x%x (pool group %d) is not built", epi, pg);
  return NESIE_ERR_UNSUPPORTED;
}

extern "C" int nesie_pw_layer_forward(int nb, int ng, int k, int cout, long long p,
                                      const float *x, long long x_bstride, const float *w,
                                      long long w_gstride, int w_rstride, int w_cstride,
                                      const float *in_coef, int in_relu, const float *row_bias,
                                      int rb_group, const float *bias, float *y,
                                      long long y_bstride, float *stat_part, int pool_group,
                                      int pool_min, float *pool_max_out, float *pool_min_out,
                                      uint8_t *arg_max_out, uint8_t *arg_min_out, void *stream) {
  const char *W = "pw_layer_forward";
  NESIE_REQUIRE(nb >= 0 && ng >= 1 && k >= 1 && cout >= 1 && p >= 0, W);
  if (nb == 0 || p == 0) return NESIE_OK;
  NESIE_REQUIRE(nb % ng == 0 && x && w, W);
  PwGeom g;
  if (!pw_geometry(k, cout, &g) || p % 32 != 0 || (long long)k * p >= (1ll << 32)) {
    set_error("%s: %d -> %d over %lld positions is outside the built tiles", W, k, cout, p);
    return NESIE_ERR_UNSUPPORTED;
  }
  NESIE_REQUIRE(((uintptr_t)x & 15) == 0 && (x_bstride & 3) == 0, W);
  int epi = 0, pg = 16;
  if (in_coef) epi |= PW_AFFINE;
  if (y) epi |= PW_STORE;
  if (stat_part) epi |= PW_STATS;
  if (row_bias) {
    NESIE_REQUIRE(rb_group >= 16 && (rb_group & (rb_group - 1)) == 0 && p % rb_group == 0, W);
    epi |= PW_ROWBIAS;
  }
  if (bias) epi |= PW_BIAS;
  if (pool_group) {
    NESIE_REQUIRE(pool_group == 16 || pool_group == 32, W);
    NESIE_REQUIRE(pool_max_out && arg_max_out, W);
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
  a.tiles_per_batch = cdiv(p, g.pt);
  a.nslots = nesie_pw_stat_slots(nb, ng, k, cout, p);
  a.nwg_g = a.nslots / g.wc;
  const int grid = a.nwg_g * ng;
  const size_t lds = pw_lds_bytes(g);
  hipStream_t s = (hipStream_t)stream;
  int st = NESIE_ERR_UNSUPPORTED;
#define G(KQ, WR, WC, PT) \
  if (g.kq == KQ && g.wr == WR) st = pw_launch_epi<KQ, WR, WC, PT>(a, epi, pg, grid, lds, s)
  G(16, 4, 2, 128); G(16, 8, 1, 128); G(16, 16, 1, 128);
  G(32, 4, 2, 64); G(32, 8, 1, 64); G(32, 16, 1, 64);
  G(33, 8, 1, 64);
  G(64, 8, 1, 32); G(64, 16, 1, 32);
  G(65, 8, 1, 32);
#undef G
  if (st != NESIE_OK) {
    if (st == NESIE_ERR_UNSUPPORTED && !strstr(nesie_last_error(), "epilogue"))
      set_error("%s: no build for %d -> %d", W, k, cout);
    return st;
  }
  return check_launch(W);
}

extern "C" int nesie_pw_stats_finalize(int channels, int cout, int nslots, const float *stat_part,
                                       const float *gamma, const float *beta,
                                       float *running_mean, float *running_var, float momentum,
                                       float eps, float *coef, void *stream) {
  const char *W = "pw_stats_finalize";
  NESIE_REQUIRE(channels >= 1 && cout >= 1 && channels % cout == 0 && nslots >= 1, W);
  NESIE_REQUIRE(stat_part && coef && (running_mean == nullptr) == (running_var == nullptr), W);
  hipLaunchKernelGGL(pw_stats_finalize_kernel, dim3(channels), dim3(64), 0, (hipStream_t)stream,
                     cout, nslots, stat_part, gamma, beta, running_mean, running_var, momentum,
                     eps, coef);
  return check_launch(W);
}
