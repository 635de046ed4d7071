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
// What shapes the kernel: on gfx950 an fp32 MFMA holds its SIMD's instruction issue for its whole
// duration -- no instruction of the same wave or of the SIMD's other waves overlaps it
// (tools/pwbench modes 10 / 11: an MFMA stream starves its partner wave of VALU, SALU, LDS and
// VMEM issue alike, s_setprio or not; k extra instructions behind an MFMA of the same wave cost
// their full issue time).  Matrix-pipe utilisation is therefore
//     MFMA cycles / (MFMA cycles + issue cycles of EVERYTHING else on the SIMD)
// and the design minimises the instruction count per MFMA rather than trying to hide it:
//   * weight-stationary: a persistent workgroup owns a run of (n, position-tile) tiles of one
//     weight group; wave wr keeps 16 rows of W in registers for the whole launch (K/4 VGPRs);
//   * v_mfma_f32_16x16x4_f32 computes the TRANSPOSED block D[position][channel] (A = X^T
//     fragment, B = W^T fragment: the same register contents, swapped operands), so a lane ends
//     with 4 CONSECUTIVE positions of one output channel: the block is stored with ONE
//     global_store_dwordx4 per lane on a scalar tile base + a per-lane 32-bit offset computed
//     once per launch (a lane of the untransposed block holds 4 channels x 1 position: 4 dword
//     stores and 4x the statistics state);
//   * X tiles are K rows x PT positions = 64 KB, two LDS buffers.  Tile t+1 is loaded
//     HBM -> registers (global_load_dwordx4 on scalar base + constant per-lane offsets, no
//     address arithmetic) at the top of iteration t, rides out the MFMAs of tile t, gets the
//     previous layer's BatchNorm + ReLU applied IN REGISTERS (packed fma, once per element,
//     coefficients of the thread's fixed rows held in registers) and is written to the other
//     buffer: one barrier per tile, no LDS round trip for the transform;
//   * operand fetch (lane l: X[4 kk + (l >> 4)][pos + (l & 15)]) = 4 rows x 16 consecutive
//     words; odd rows are stored with their 64-byte halves swapped so rows r and r+1 sit on
//     disjoint bank halves: conflict-free ds_read2st64_b32 (two k-steps per instruction),
//     issued a group ahead with counted lgkmcnt waits (explicit instructions: left to itself
//     hipcc sinks the reads to their uses and every MFMA pair waits out an LDS round trip);
//   * Cout = 256 runs as two workgroups of 128 rows (the operand tile is fetched twice, the
//     second time from L2 / Infinity Cache) so every geometry is 8 waves of 16 rows;
//   * epilogue straight from the accumulators: optional output-side row bias / channel bias,
//     the raw conv output, this layer's own BatchNorm statistics as per-wave SHIFTED sums
//     (count, shift, sum(y - shift), sum((y - shift)^2): no E[x^2] - E[x]^2 cancellation;
//     merged in fp64 by pw_stats_finalize_kernel with Chan's formula), and the max / min over
//     each group of 16 or 32 consecutive positions with the position of each (pooling tail).
// The K x P operand is read once (twice at Cout = 256), Y written once (or never, for a pooled
// tail): 2 tensor passes per layer where conv + statistics + normalise cost 5.
// (this header: the kernel template and its per-geometry launcher; every geometry is instantiated
// in a translation unit of its own, pwconv_g*.hip, so that the library builds in parallel)
#pragma once
#include "common.h"
#include <string.h>
#include <type_traits>

namespace nesie {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

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

template <int OFF>
__device__ __forceinline__ f32x4 lds_read_b128(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read_b128 reach");
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

// 16 bytes per lane from sbase (wave-uniform) + voff (per-lane byte offset).  A plain load: the
// compiler places the vmcnt wait at the first use (an asm load + a separate asm wait let the
// register allocator copy the destination before the wait: stale words in the first tile).
__device__ __forceinline__ f32x4 load16_saddr(unsigned voff, const void *sbase) {
  return *(const f32x4 *)((const char *)sbase + voff);
}

// *(float4 *)(sbase + voff + IMM) = v
template <int IMM>
__device__ __forceinline__ void store16_saddr(unsigned voff, f32x4 v, const void *sbase) {
  // the trailing s_nop: a store of more than 64 bits needs wait states before its data VGPRs
  // are overwritten, and the hazard recognizer does not look inside asm
  asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" :: "v"(voff), "v"(v), "s"(sbase), "n"(IMM) : "memory");
}

template <int N>
__device__ __forceinline__ void vm_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
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
  PW_BNRED = 128,   // Y is the gradient of relu(bn(Z)): partial sums of g = Y [bn(Z) > 0] and g * zhat
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
  const float *bn_z; long long bnz_bs;     // PW_BNRED: raw conv output Z (nb, cout, p) ...
  const float *bn_coef; float *bn_part;    // ... its [ng * cout][4] (scale, bias, mean, invstd); [ng * cout][nslots][2]
  int tiles_per_batch, nwg_g, nhalf;       // nhalf: workgroups per tile along Cout (128 rows each)
  long long *stamps;   // development only (PW_STAMP builds): per-phase s_memtime of workgroup 0
};
#ifdef PW_STAMP
#define STAMP(slot)                                                                      \
  if (a.stamps && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4) && iter < 24) \
    a.stamps[((wave >> 2) * 24 + iter) * 8 + (slot)] = __builtin_amdgcn_s_memtime();
#define STAMP_CLK(which)                                                     \
  if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) {                      \
    a.stamps[2 * 24 * 8 + 2 * (which)] = __builtin_amdgcn_s_memtime();       \
    a.stamps[2 * 24 * 8 + 2 * (which) + 1] = __builtin_amdgcn_s_memrealtime(); \
  }
#else
#define STAMP(slot)
#define STAMP_CLK(which)
#endif

// KQ = padded K / 4; WR x WC waves (16 output rows each x PT / WC positions); PT positions
// per tile; EPI = epilogue / prologue flags; PG = pooling granule (16 or 32)
template <int KQ, int WR, int WC, int PT, int EPI, int PG>
__global__ __launch_bounds__(WR *WC * 64) void pw_fwd_kernel(const PwFwd a) {
  constexpr int NW = WR * WC, NT = NW * 64, KPAD = 4 * KQ, NBLK = PT / 16 / WC;
  constexpr int TILE = KPAD * PT, CPR = PT / 4;          // floats per buffer, 16-byte chunks per row
  constexpr int NX = (KPAD * CPR + NT - 1) / NT;         // staged chunks per thread and tile
  constexpr bool EVEN = KPAD * CPR == NX * NT;           // every staging slot is a real chunk
  constexpr int CROWS = WR * 16;                         // output rows per workgroup
  static_assert(NBLK >= 1 && PT % (16 * WC) == 0 && PT >= 32 && CPR % 8 == 0, "tile");
  static_assert(!(EPI & PW_POOL) || PG == 16 || NBLK % 2 == 0, "a 32-position pool needs block pairs");
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x, lane = tid & 63, quad = lane >> 4, l16 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;
  // block -> (weight group g, row half, rank inside the group)
  const int half = blockIdx.x % a.nhalf;
  const int g = (blockIdx.x / a.nhalf) % a.ng, rank = blockIdx.x / (a.nhalf * a.ng);
  const int k = a.k;
  const int c0 = half * CROWS;                           // first output row of this workgroup
  const int cout = a.cout, crows = cout - c0 < CROWS ? cout - c0 : CROWS;
  const long long p = a.p;

  // staging slots of this thread: chunk c = i * NT + tid -> row c / CPR, 16-byte column c % CPR.
  // goff: byte offset from the tile's first word; lw: LDS byte offset inside a buffer (odd rows
  // carry their 64-byte halves swapped); (sc, bi): the row's BatchNorm coefficients.
  unsigned goff[NX], lw[NX];
  bool okslot[NX];
  f32x2 sc[(EPI & PW_AFFINE) ? NX : 1], bi[(EPI & PW_AFFINE) ? NX : 1];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int c = i * NT + tid;
    const int row = c / CPR, cp = c % CPR;
    const bool ok = (EVEN || c < KPAD * CPR) && row < k;
    okslot[i] = ok;
    goff[i] = ok ? (unsigned)(((size_t)row * p + cp * 4) * 4) : 0u;   // invalid slots re-read word 0
    lw[i] = (unsigned)((row * PT + ((cp ^ ((row & 1) << 2)) * 4)) * 4);
    if (EPI & PW_AFFINE) {
      const float s0 = ok ? a.in_coef[((size_t)g * k + row) * 4] : 0.f;
      const float b0 = ok ? a.in_coef[((size_t)g * k + row) * 4 + 1] : 0.f;
      sc[i] = (f32x2){s0, s0};
      bi[i] = (f32x2){b0, b0};
    }
  }
  // this wave's 16 rows of W, for the whole launch (lane: row l16, k = 4 kk + quad)
  float wreg[KQ];
  {
    const int m = wr * 16 + l16;
    const float *wg = a.w + (size_t)g * a.w_gs + (size_t)(c0 + m) * a.w_rs;
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk) {
      const int kx = 4 * kk + quad;
      wreg[kk] = (m < crows && kx < k) ? wg[(size_t)kx * a.w_cs] : 0.f;
    }
  }
  const bool all_k = __builtin_amdgcn_readfirstlane(k == KPAD ? 1 : 0) != 0;
  const int tpb = a.tiles_per_batch, nwg = a.nwg_g;
  const int ntiles = (a.nb / a.ng) * tpb;

  // The launch-time loads (W, coefficients) are waited for HERE: left alone, the compiler puts
  // their vmcnt(0) in front of the first use inside the tile loop, where it also drains the
  // operand loads that were just issued for the next tile.
#pragma unroll
  for (int kk = 0; kk < KQ; ++kk) asm volatile("" : "+v"(wreg[kk]));
  if (EPI & PW_AFFINE) {
#pragma unroll
    for (int i = 0; i < NX; ++i) asm volatile("" : "+v"(sc[i]), "+v"(bi[i]));
  }

  // Tiles are whole (the host requires p % PT == 0): the loads of a tile are a straight line of
  // NX instructions with no branch between issue and first use.
  f32x4 stg[NX];
  auto load_tile = [&](int n, long long p0) {
    const float *xb = a.x + (size_t)n * a.x_bs + p0;                  // wave-uniform
#pragma unroll
    for (int i = 0; i < NX; ++i) stg[i] = load16_saddr(goff[i], xb);
  };
  // previous layer's BatchNorm + ReLU in registers (packed fma; the backward's mask test uses
  // the same fused form), then into the LDS buffer
  auto write_chunk = [&](auto ic, float *buf) {
    constexpr int i = decltype(ic)::value;
    f32x4 q = stg[i];
    if (!(EVEN && all_k)) q = okslot[i] ? q : (f32x4){0.f, 0.f, 0.f, 0.f};
    if (EPI & PW_AFFINE) {
      const f32x2 lo = __builtin_elementwise_fma((f32x2){q[0], q[1]}, sc[i], bi[i]);
      const f32x2 hi = __builtin_elementwise_fma((f32x2){q[2], q[3]}, sc[i], bi[i]);
      q[0] = fmaxf(lo[0], a.in_lo); q[1] = fmaxf(lo[1], a.in_lo);
      q[2] = fmaxf(hi[0], a.in_lo); q[3] = fmaxf(hi[1], a.in_lo);
    }
    if (EVEN || (i * NT + tid) < KPAD * CPR) *(f32x4 *)((char *)buf + lw[i]) = q;
  };
  auto write_tile = [&](float *buf) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      f32x4 q = stg[i];
      if (!(EVEN && all_k)) q = okslot[i] ? q : (f32x4){0.f, 0.f, 0.f, 0.f};
      if (EPI & PW_AFFINE) {
        const f32x2 lo = __builtin_elementwise_fma((f32x2){q[0], q[1]}, sc[i], bi[i]);
        const f32x2 hi = __builtin_elementwise_fma((f32x2){q[2], q[3]}, sc[i], bi[i]);
        q[0] = fmaxf(lo[0], a.in_lo); q[1] = fmaxf(lo[1], a.in_lo);
        q[2] = fmaxf(hi[0], a.in_lo); q[3] = fmaxf(hi[1], a.in_lo);
      }
      if (EVEN || (i * NT + tid) < KPAD * CPR) *(f32x4 *)((char *)buf + lw[i]) = q;
    }
  };

  // statistics state: a lane holds ONE channel (row c0 + 16 wr + l16) x 4 positions per block
  float s1 = 0.f, s2 = 0.f, shift = 0.f;
  int nblk_done = 0;
  const int q0 = wc * NBLK * 16;              // this wave's first position inside a tile
  const int m = c0 + wr * 16 + l16;           // this lane's output channel
  // byte offset of (row m, position q0 + 4 quad) from the tile's first output word
  const unsigned roff = (unsigned)(((size_t)m * p + q0 + 4 * quad) * 4);

  // PW_BNRED: the raw output Z of the layer whose activation gradient this launch produces, at
  // this lane's output elements (loaded ahead of the MFMAs of the tile), and the two sums
  f32x4 zv[(EPI & PW_BNRED) ? NBLK : 1];
  float r0 = 0.f, r1 = 0.f;
  float4 zc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (EPI & PW_BNRED) {
    if (m < cout) zc = *(const float4 *)(a.bn_coef + ((size_t)g * cout + m) * 4);
    asm volatile("" : "+v"(zc.x), "+v"(zc.y), "+v"(zc.z), "+v"(zc.w));
  }
  auto load_z = [&](int n, long long p0) {
    const float *zt = a.bn_z + (size_t)n * a.bnz_bs + p0;   // wave-uniform
#pragma unroll
    for (int j = 0; j < NBLK; ++j)
      zv[j] = m < cout ? load16_saddr(roff + 64u * j, zt) : (f32x4){0.f, 0.f, 0.f, 0.f};
  };

  f32x4 acc[NBLK];
  auto epilogue = [&](int n, long long p0) {
    constexpr bool full = true;   // p % PT == 0
    float *ytile = a.y + (size_t)n * a.y_bs + p0;        // wave-uniform
    static_for<0, NBLK>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if (full || p0 + q0 + 16 * j < p) {
        if (EPI & PW_ROWBIAS) {
          if (m < cout) {
            const float rb = a.row_bias[((size_t)n * cout + m) * (size_t)(p >> a.rb_shift) +
                                        ((p0 + q0 + 16 * j) >> a.rb_shift)];
            acc[j] += (f32x4){rb, rb, rb, rb};
          }
        }
        if (EPI & PW_BIAS) {
          if (m < cout) { const float bs = a.bias[g * cout + m]; acc[j] += (f32x4){bs, bs, bs, bs}; }
        }
        if (EPI & PW_STORE) {
          if (m < cout) store16_saddr<64 * j>(roff, acc[j], ytile);
        }
        if (EPI & PW_BNRED) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float z = zv[j][r];
            const float gg = __builtin_fmaf(z, zc.x, zc.y) > 0.f ? acc[j][r] : 0.f;
            r0 += gg;
            r1 += gg * ((z - zc.z) * zc.w);
          }
        }
        if (EPI & PW_STATS) {
          if (nblk_done == 0) shift = __shfl(acc[j][0], l16, 64);   // first value of the channel
          ++nblk_done;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float d = acc[j][r] - shift;
            s1 += d;
            s2 += d * d;
          }
        }
      }
    });
    if (EPI & PW_POOL) {
      // a group = PG consecutive positions = PG / 16 blocks x (4 quads x 4 registers)
      const size_t prow = (size_t)(p / PG);
      static_for<0, NBLK / (PG / 16)>([&](auto jc) {
        constexpr int j = decltype(jc)::value * (PG / 16);
        if (full || p0 + q0 + 16 * j < p) {
          const size_t pcol = (size_t)((p0 + q0 + 16 * j) / PG);
#pragma unroll
          for (int mm = 0; mm < ((EPI & PW_POOLMIN) ? 2 : 1); ++mm) {
            // per lane: best of its 4 (8) values, smallest position on ties
            float e = acc[j][0];
            int at = 4 * quad;
#pragma unroll
            for (int u = 1; u < 4 * (PG / 16); ++u) {
              const float v = acc[j + u / 4][u % 4];
              const bool better = mm ? v < e : v > e;
              at = better ? 16 * (u / 4) + 4 * quad + u % 4 : at;
              e = better ? v : e;
            }
            // across the 4 quads (lanes l16, l16 + 16, + 32, + 48)
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
              const float oe = __shfl_xor(e, off, 64);
              const int oa = __shfl_xor(at, off, 64);
              const bool take = (mm ? oe < e : oe > e) || (oe == e && oa < at);
              e = take ? oe : e;
              at = take ? oa : at;
            }
            if (quad == 0 && m < cout) {
              const size_t o = ((size_t)n * cout + m) * prow + pcol;
              (mm ? a.pool_min : a.pool_max)[o] = e;
              (mm ? a.arg_min : a.arg_max)[o] = (uint8_t)at;
            }
          }
        }
      });
    }
  };

  // ---- main loop: one barrier per tile; b0 holds tile t, tile t+1 is staged through registers
  // into b1 behind the MFMAs of tile t
  float *b0 = lds, *b1 = lds + TILE;
  // tile coordinates advance incrementally: tile t = (batch tq of the group, tile tr of the batch)
  const int dq = nwg / tpb, dr = nwg % tpb;
  auto advance = [&](int &q, int &r) {
    q += dq; r += dr;
    if (r >= tpb) { r -= tpb; ++q; }
  };
  int t = rank;
  int tq = rank / tpb, tr = rank % tpb;          // tile t
  int nq = tq, nr = tr;                          // tile t + 1
  advance(nq, nr);
  __syncthreads();                               // LDS zero fill
  if (t < ntiles) {
    load_tile(g + a.ng * tq, (long long)tr * PT);
    write_tile(b0);
  }
  int iter = 0;
  (void)iter;
  STAMP_CLK(0)
  for (; t < ntiles; t += nwg, ++iter) {
    STAMP(0)
    lgkm_wait<0>();          // this thread's ds_writes of tile t
    __builtin_amdgcn_s_barrier();
    STAMP(1)
    const bool more = t + nwg < ntiles;
#ifdef PW_INTERLEAVE
    const float *nxb = a.x + (size_t)(g + a.ng * nq) * a.x_bs + (long long)nr * PT;   // uniform
#else
    if (EPI & PW_BNRED) load_z(g + a.ng * tq, (long long)tr * PT);   // older than the operand loads
    if (more) load_tile(g + a.ng * nq, (long long)nr * PT);
#endif
    STAMP(2)
    const int n = g + a.ng * tq;
    const long long p0 = (long long)tr * PT;
    constexpr bool do_mfma = true;
    if (do_mfma) {
      // ---- MFMA loop: NBLK blocks of 16 positions, K in groups of GK quads; the LDS reads of
      // group gi + 1 are issued before the MFMAs of group gi, counted lgkmcnt waits
      constexpr int GK = NBLK >= 8 ? 2 : 4, NGRP = (KQ + GK - 1) / GK, UPK = PT / 16;   // 256-byte units per kk
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
              if constexpr (k1 * UPK < 256)
                bq[gi & 1][j][i] = lds_read2st64<kk * UPK, k1 * UPK>(xa[j]);
              else   // beyond the 8-bit reach (the odd tail of K = 132 / 260): rebased by 64 KB
                bq[gi & 1][j][i] = lds_read2st64<kk * UPK - 256, k1 * UPK - 256>(xa[j] + 65536u);
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
              // D[position][channel] += X^T[position][k] . W^T[k][channel]
              acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[gi & 1][j][i / 2][i & 1], wreg[kk], acc[j], 0, 0, 0);
            });
          }
        });
        __builtin_amdgcn_sched_barrier(0);
#ifdef PW_INTERLEAVE
        if (more) {
          constexpr int H = NGRP / 2;
          static_for<0, NX>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr ((i * H) / NX == gi) stg[i] = load16_saddr(goff[i], nxb);
            if constexpr (H + (i * (NGRP - H)) / NX == gi) write_chunk(ic, b1);
          });
        }
        __builtin_amdgcn_sched_barrier(0);
#endif
      });
    }
    STAMP(3)
#ifndef PW_INTERLEAVE
    if (more) write_tile(b1);   // waits for the loads of tile t + 1 (issued before the MFMAs)
#endif
    STAMP(4)
    if (do_mfma) epilogue(n, p0);
    STAMP(5)
    tq = nq; tr = nr;
    advance(nq, nr);
    float *const tb = b0; b0 = b1; b1 = tb;
  }
  STAMP_CLK(1)
  if (EPI & PW_BNRED) {
    const int slot = rank * WC + wc;
    r0 += __shfl_xor(r0, 16, 64); r1 += __shfl_xor(r1, 16, 64);
    r0 += __shfl_xor(r0, 32, 64); r1 += __shfl_xor(r1, 32, 64);
    if (quad == 0 && m < cout)
      *(float2 *)(a.bn_part + (((size_t)g * cout + m) * a.nslots + slot) * 2) = make_float2(r0, r1);
  }
  if (EPI & PW_STATS) {
    // one partial per wave: (count, shift, sum, sum of squares) of its 16 channels; the four
    // quads hold different positions of the same channel
    const int slot = rank * WC + wc;
    s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
    s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
    if (quad == 0 && m < cout) {
      float4 o;
      o.x = (float)nblk_done * 16.f;
      o.y = shift;
      o.z = s1; o.w = s2;
      *(float4 *)(a.stat_part + (((size_t)g * a.nslots + slot) * cout + m) * 4) = o;
    }
  }
}

// ---- per-geometry launcher: picks the built prologue / epilogue combination -----------------
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
#ifdef PW_DEV   // tools/pwbench development build: a few instantiations per geometry
  if (epi == PW_STORE) GO(PW_STORE, 16);
  if (epi == (PW_STORE | PW_BNRED)) GO(PW_STORE | PW_BNRED, 16);
  if (epi == (PW_AFFINE | PW_STORE | PW_STATS)) GO(PW_AFFINE | PW_STORE | PW_STATS, 16);
  if (epi == (PW_AFFINE | PW_STORE | PW_STATS | PW_POOL | PW_POOLMIN) && pg == 16)
    GO(PW_AFFINE | PW_STORE | PW_STATS | PW_POOL | PW_POOLMIN, 16);
  if (epi == (PW_AFFINE | PW_STORE | PW_STATS | PW_POOL | PW_POOLMIN) && pg == 32)
    GO(PW_AFFINE | PW_STORE | PW_STATS | PW_POOL | PW_POOLMIN, 32);
  set_error("dev build");
  return NESIE_ERR_UNSUPPORTED;
#else
  const int aff = epi & PW_AFFINE;
  const int base = epi & ~PW_AFFINE;
  // the prologue / epilogue combinations the step uses
  if (aff) {
    if (base == PW_STORE) GO(PW_AFFINE | PW_STORE, 16);
    if (base == (PW_STORE | PW_STATS)) GO(PW_AFFINE | PW_STORE | PW_STATS, 16);
    if (base == (PW_STORE | PW_POOL) && pg == 16) GO(PW_AFFINE | PW_STORE | PW_POOL, 16);
    if (base == PW_POOL && pg == 16) GO(PW_AFFINE | PW_POOL, 16);
    if (base == (PW_STORE | PW_STATS | PW_POOL | PW_POOLMIN) && pg == 16)
      GO(PW_AFFINE | PW_STORE | PW_STATS | PW_POOL | PW_POOLMIN, 16);
    if (base == (PW_STORE | PW_POOL) && pg == 32) GO(PW_AFFINE | PW_STORE | PW_POOL, 32);
    if (base == PW_POOL && pg == 32) GO(PW_AFFINE | PW_POOL, 32);
    if (base == (PW_STORE | PW_STATS | PW_POOL | PW_POOLMIN) && pg == 32)
      GO(PW_AFFINE | PW_STORE | PW_STATS | PW_POOL | PW_POOLMIN, 32);
  } else {
    if (base == PW_STORE) GO(PW_STORE, 16);
    if (base == (PW_STORE | PW_BNRED)) GO(PW_STORE | PW_BNRED, 16);
    if (base == (PW_STORE | PW_STATS)) GO(PW_STORE | PW_STATS, 16);
    if (base == (PW_STORE | PW_STATS | PW_ROWBIAS)) GO(PW_STORE | PW_STATS | PW_ROWBIAS, 16);
  }
  set_error("pw_layer_forward: epilogue combination 0x%x (pool group %d) is not built", epi, pg);
  return NESIE_ERR_UNSUPPORTED;
#endif
#undef GO
}

#define PW_GEOM_NAME(KQ, WR, WC, PT) pw_launch_##KQ##_##WR##_##WC##_##PT
#define PW_GEOM_DECL(KQ, WR, WC, PT) \
  int PW_GEOM_NAME(KQ, WR, WC, PT)(const PwFwd &a, int epi, int pg, int grid, size_t lds, hipStream_t s);
#define PW_GEOM_DEF(KQ, WR, WC, PT)                                                             \
  namespace nesie {                                                                             \
  int PW_GEOM_NAME(KQ, WR, WC, PT)(const PwFwd &a, int epi, int pg, int grid, size_t lds,       \
                                   hipStream_t s) {                                             \
    return pw_launch_epi<KQ, WR, WC, PT>(a, epi, pg, grid, lds, s);                             \
  }                                                                                             \
  }

}  // namespace nesie
