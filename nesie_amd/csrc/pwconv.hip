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

// The pooling tail after the layer kernel: combine the G / PG partial extrema of every group,
// apply the layer's own BatchNorm + ReLU to the extremum the sign of the scale selects
// (max_j relu(s y_j + b) = relu(s (s >= 0 ? max_j y_j : min_j y_j) + b), exactly) and record its
// position inside the group.  rows = nb * channels, coef == NULL: plain max.
__global__ __launch_bounds__(256) void pw_pool_finish_kernel(
    long long total, int channels, int ng, int groups, int nsub, int pg, const float *__restrict__ pmax,
    const float *__restrict__ pmin, const uint8_t *__restrict__ amax,
    const uint8_t *__restrict__ amin, const float *__restrict__ coef, float lo,
    float *__restrict__ pooled, uint8_t *__restrict__ arg) {
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
}

// tile geometry of a (K, Cout) layer
struct PwGeom { int kq, wr, wc, pt, nhalf; };

static bool pw_geometry(int k, int cout, PwGeom *o) {
  const int kq = k <= 64 ? 16 : k <= 128 ? 32 : k <= 132 ? 33 : k <= 256 ? 64 : k <= 260 ? 65 : 0;
  if (!kq || cout < 1 || cout > 256) return false;
  o->kq = kq;
  o->nhalf = cout > 128 ? 2 : 1;                 // 128-row workgroups
  o->wr = cout <= 64 ? 4 : 8;
  o->wc = o->wr == 4 ? 2 : 1;                    // 8 waves
  // 64 KB operand tiles (32 KB at K <= 64): PT x padded K x 4 bytes
  o->pt = kq == 16 ? (o->wc == 2 ? 256 : 128) : kq <= 33 ? 128 : 64;
  return true;
}

static size_t pw_lds_bytes(const PwGeom &g) { return (size_t)2 * 4 * g.kq * g.pt * sizeof(float); }

}  // namespace nesie

using namespace nesie;

#ifdef PW_STAMP
long long *g_pw_stamps = nullptr;
#endif

extern "C" int nesie_pw_supported(int k, int cout, long long p) {
  PwGeom g;
  return pw_geometry(k, cout, &g) && p % g.pt == 0 && (long long)(k > cout ? k : cout) * p < (1ll << 30) ? 1 : 0;
}

// number of statistic slots per weight group a forward launch writes
extern "C" int nesie_pw_stat_slots(int nb, int ng, int k, int cout, long long p) {
  PwGeom g;
  if (!pw_geometry(k, cout, &g) || ng < 1) return 0;
  const long long tiles = (long long)(nb / ng) * cdiv(p, g.pt);
  long long nwg = 256 / (ng * g.nhalf);
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

static int pw_forward_impl(const char *W, int nb, int ng, int k, int cout, long long p,
                           const float *x, long long x_bstride, const float *w,
                           long long w_gstride, int w_rstride, int w_cstride,
                           const float *in_coef, int in_relu, const float *row_bias,
                           int rb_group, const float *bias, float *y,
                           long long y_bstride, float *stat_part, int pool_group,
                           int pool_min, float *pool_max_out, float *pool_min_out,
                           uint8_t *arg_max_out, uint8_t *arg_min_out, const float *bn_z,
                           long long bnz_bstride, const float *bn_coef, float *bn_part,
                           void *stream) {
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
    NESIE_REQUIRE(y && bn_coef && bn_part && ((uintptr_t)bn_z & 15) == 0 && (bnz_bstride & 3) == 0, W);
    epi |= PW_BNRED;
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
#ifdef PW_STAMP
  a.stamps = g_pw_stamps;
#endif
  a.tiles_per_batch = cdiv(p, g.pt);
  a.nslots = nesie_pw_stat_slots(nb, ng, k, cout, p);
  a.nwg_g = a.nslots / g.wc;
  a.nhalf = g.nhalf;
  const int grid = a.nwg_g * ng * g.nhalf;
  const size_t lds = pw_lds_bytes(g);
  hipStream_t s = (hipStream_t)stream;
  int st = NESIE_ERR_UNSUPPORTED;
#define G(KQ, WR, WC, PT) \
  if (g.kq == KQ && g.wr == WR) st = pw_launch_epi<KQ, WR, WC, PT>(a, epi, pg, grid, lds, s)
#ifdef PW_DEV
  G(64, 8, 1, 64); G(32, 8, 1, 128);
#else
  G(16, 4, 2, 256); G(16, 8, 1, 128);
  G(32, 4, 2, 128); G(32, 8, 1, 128);
  G(33, 4, 2, 128); G(33, 8, 1, 128);
  G(64, 4, 2, 64); G(64, 8, 1, 64);
  G(65, 4, 2, 64); G(65, 8, 1, 64);
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

extern "C" int nesie_pw_pool_finish(int nb, int ng, int channels, long long p, int group, int pool_group,
                                    const float *pmax, const float *pmin, const uint8_t *amax,
                                    const uint8_t *amin, const float *coef, int relu,
                                    float *pooled, uint8_t *argmax, void *stream) {
  const char *W = "pw_pool_finish";
  NESIE_REQUIRE(nb >= 0 && ng >= 1 && channels >= 1 && p >= 0 && group >= 1, W);
  if (nb == 0 || p == 0) return NESIE_OK;
  NESIE_REQUIRE((pool_group == 16 || pool_group == 32) && group % pool_group == 0 && p % group == 0, W);
  NESIE_REQUIRE(group <= 256 && pmax && amax && pooled && argmax && (!coef || (pmin && amin)), W);
  const long long total = (long long)nb * channels * (p / group);
  hipLaunchKernelGGL(pw_pool_finish_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     total, channels, ng, (int)(p / group), group / pool_group, pool_group, pmax, pmin,
                     amax, amin, coef, relu ? 0.f : -__builtin_inff(), pooled, argmax);
  return check_launch(W);
}

// ---- weight gradient ----------------------------------------------------------------------
//     dW[g][co][ci] = sum over the batches n of group g and all positions of
//                     dY[n][co][pos] * act(X[n][ci][pos])
// (the Conv2d weight gradient autograd computes for ConvModule, point_sa_module.py:277-289; act =
// the producer layer's folded BatchNorm + ReLU, recomputed on load: the normalised activation
// was never stored).  The output is small (<= 128 x 320) and the reduction runs over 10^5..10^6
// positions: a persistent workgroup keeps the WHOLE co x ci product in its accumulators (wave
// (wm, wn) owns a (16 MB) x (16 NB) block), walks its run of (batch, 32-position) tiles and
// leaves one partial; pw_wgrad_reduce_kernel adds the partials in a fixed order.  Same
// pipeline as the layer kernel: tile t+1 goes HBM -> registers (transform) -> LDS behind the
// MFMAs of tile t, one barrier per tile.  Both MFMA operands are [row][position] tiles; a lane
// reads FOUR consecutive positions of its row with one ds_read_b128 and feeds component c to
// MFMA c (the position <-> k mapping is the same for both operands, so any bijection works):
// MB + NB reads feed 4 MB NB MFMAs.  Row pitch PT + 4 words: the 16 rows a read touches sit
// 4 banks apart.
namespace nesie {

template <int CO16, int CI16, int WM, int WN, bool AFF>
__global__ __launch_bounds__(512) void pw_wgrad_kernel(
    int nb, int ng, int co, int ci, long long p, const float *__restrict__ dy, long long dy_bs,
    const float *__restrict__ x, long long x_bs, const float *__restrict__ x_coef, float x_lo,
    float *__restrict__ partial, int nwg_g) {
  constexpr int MB = CO16 / WM, NB = CI16 / WN, PT = 32, PITCH = PT + 4, CPR = PT / 4;
  constexpr int ROWS = (CO16 + CI16) * 16, NT = 512;
  constexpr int NX = (ROWS * CPR + NT - 1) / NT;
  constexpr bool EVEN = ROWS * CPR == NX * NT;
  constexpr int DYSLOTS = CO16 * 16 * CPR / NT;          // slots that hold dY rows (CO16 % 4 == 0)
  static_assert(WM * WN == 8 && CO16 % WM == 0 && CI16 % WN == 0 && (CO16 * 16 * CPR) % NT == 0, "tiling");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int TILE = ROWS * PITCH;
  const int tid = threadIdx.x, lane = tid & 63, quad = lane >> 4, l16 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int g = blockIdx.x % ng, rank = blockIdx.x / ng;

  unsigned goff[NX], lw[NX];
  bool okslot[NX];
  f32x2 sc[AFF ? NX : 1], bi[AFF ? NX : 1];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int c = i * NT + tid;
    const int row = c / CPR, cp = c % CPR;
    const bool isdy = i < DYSLOTS;
    const int r = isdy ? row : row - CO16 * 16;
    const bool ok = (EVEN || c < ROWS * CPR) && r < (isdy ? co : ci);
    okslot[i] = ok;
    goff[i] = ok ? (unsigned)(((size_t)r * p + cp * 4) * 4) : 0u;
    lw[i] = (unsigned)((row * PITCH + cp * 4) * 4);
    if (AFF) {
      const float s0 = (ok && !isdy) ? x_coef[((size_t)g * ci + r) * 4] : 0.f;
      const float b0 = (ok && !isdy) ? x_coef[((size_t)g * ci + r) * 4 + 1] : 0.f;
      sc[i] = (f32x2){s0, s0};
      bi[i] = (f32x2){b0, b0};
    }
  }
  if (AFF) {
#pragma unroll
    for (int i = 0; i < NX; ++i) asm volatile("" : "+v"(sc[i]), "+v"(bi[i]));
  }
  const bool all_rows = __builtin_amdgcn_readfirstlane((co == CO16 * 16 && ci == CI16 * 16) ? 1 : 0) != 0;
  const long long tpb = p / PT;
  const long long ntiles = (long long)(nb / ng) * tpb;

  f32x4 stg[NX];
  auto load_tile = [&](long long t) {
    const int n = g + ng * (int)(t / tpb);
    const long long p0 = (t % tpb) * PT;
    const float *dyb = dy + (size_t)n * dy_bs + p0, *xb = x + (size_t)n * x_bs + p0;   // uniform
#pragma unroll
    for (int i = 0; i < NX; ++i) stg[i] = load16_saddr(goff[i], i < DYSLOTS ? dyb : xb);
  };
  auto write_tile = [&](float *buf) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      f32x4 q = stg[i];
      if (AFF && i >= DYSLOTS) {
        const f32x2 lo = __builtin_elementwise_fma((f32x2){q[0], q[1]}, sc[i], bi[i]);
        const f32x2 hi = __builtin_elementwise_fma((f32x2){q[2], q[3]}, sc[i], bi[i]);
        q[0] = fmaxf(lo[0], x_lo); q[1] = fmaxf(lo[1], x_lo);
        q[2] = fmaxf(hi[0], x_lo); q[3] = fmaxf(hi[1], x_lo);
      }
      if (!(EVEN && all_rows)) q = okslot[i] ? q : (f32x4){0.f, 0.f, 0.f, 0.f};
      if (EVEN || (i * NT + tid) < ROWS * CPR) *(f32x4 *)((char *)buf + lw[i]) = q;
    }
  };

  f32x4 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float *b0 = lds, *b1 = lds + TILE;
  long long t = rank;
  if (t < ntiles) {
    load_tile(t);
    write_tile(b0);
  }
  for (; t < ntiles; t += nwg_g) {
    lgkm_wait<0>();
    __builtin_amdgcn_s_barrier();
    const bool more = t + nwg_g < ntiles;
    if (more) load_tile(t + nwg_g);
    // lane (l16 = row inside its block, quad): positions 16 pg + 4 quad .. + 3
    const unsigned la = lds_addr(b0) + (unsigned)(((wm * MB * 16 + l16) * PITCH + 4 * quad) * 4);
    const unsigned lb = lds_addr(b0) + (unsigned)(((CO16 * 16 + wn * NB * 16 + l16) * PITCH + 4 * quad) * 4);
    f32x4 fa[2][MB], fb[2][NB];
    auto load_frags = [&](auto pgc) {
      constexpr int pg = decltype(pgc)::value;
      static_for<0, MB>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        fa[pg][i] = lds_read_b128<(i * 16 * PITCH + 16 * pg) * 4>(la);
      });
      static_for<0, NB>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        fb[pg][j] = lds_read_b128<(j * 16 * PITCH + 16 * pg) * 4>(lb);
      });
    };
    load_frags(std::integral_constant<int, 0>{});
    load_frags(std::integral_constant<int, 1>{});
    static_for<0, 2>([&](auto pgc) {
      constexpr int pg = decltype(pgc)::value;
      if constexpr (pg == 0) lgkm_wait<MB + NB>(); else lgkm_wait<0>();
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, 4>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        static_for<0, MB>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          static_for<0, NB>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[pg][i][c], fb[pg][j][c], acc[i][j], 0, 0, 0);
          });
        });
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    if (more) write_tile(b1);
    float *const tb = b0; b0 = b1; b1 = tb;
  }
  // partial[(g * nwg + rank)][co][ci]: lane (quad, l16) holds rows 4 quad + r, column l16
  float *dst = partial + ((size_t)g * nwg_g + rank) * co * ci;
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = (wm * MB + i) * 16 + 4 * quad + r, k = (wn * NB + j) * 16 + l16;
        if (m < co && k < ci) dst[(size_t)m * ci + k] = acc[i][j][r];
      }
}

// dw[g][i] = sum over the nparts partials of group g, in a fixed order
__global__ __launch_bounds__(1024) void pw_wgrad_reduce_kernel(int total, int nparts,
                                                               const float *__restrict__ partial,
                                                               float *__restrict__ dw) {
  __shared__ float sh[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane, g = blockIdx.y;
  const float *src = partial + (size_t)g * nparts * total;
  float s = 0.f;
  if (i < total) {
    int r = wave;
    for (; r + 7 * 16 < nparts; r += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(r + u * 16) * total + i];
      s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; r < nparts; r += 16) s += src[(size_t)r * total + i];
  }
  sh[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && i < total) {
    float tt = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tt += sh[w][lane];
    dw[(size_t)g * total + i] = tt;
  }
}

static int pw_wgrad_nwg(int nb, int ng, long long p) {
  long long nwg = 256 / ng;
  const long long tiles = (long long)(nb / ng) * (p / 32);
  if (nwg > tiles) nwg = tiles;
  return nwg < 1 ? 1 : (int)nwg;
}

}  // namespace nesie

extern "C" int nesie_pw_wgrad_supported(int co, int ci, long long p) {
  return p % 32 == 0 && ci >= 9 &&
                 ((co <= 64 && ci <= 64) || (co <= 128 && ci <= 320) || (co <= 256 && ci <= 128))
             ? 1 : 0;
}

extern "C" size_t nesie_pw_wgrad_workspace_bytes(int nb, int ng, int co, int ci, long long p) {
  if (nb <= 0 || ng <= 0 || p <= 0) return 0;
  return (size_t)ng * pw_wgrad_nwg(nb, ng, p) * co * ci * sizeof(float);
}

extern "C" int nesie_pw_wgrad(int nb, int ng, int co, int ci, long long p, const float *dy,
                              long long dy_bstride, const float *x, long long x_bstride,
                              const float *x_coef, int x_relu, float *dw, void *workspace,
                              size_t workspace_bytes, void *stream) {
  const char *W = "pw_wgrad";
  NESIE_REQUIRE(nb >= 0 && ng >= 1 && co >= 1 && ci >= 1 && p >= 0 && dw, W);
  hipStream_t s = (hipStream_t)stream;
  if (nb == 0 || p == 0) {
    (void)hipMemsetAsync(dw, 0, (size_t)ng * co * ci * sizeof(float), s);
    return NESIE_OK;
  }
  if (!nesie_pw_wgrad_supported(co, ci, p)) {
    set_error("%s: %d x %d over %lld positions is outside the built tiles", W, co, ci, p);
    return NESIE_ERR_UNSUPPORTED;
  }
  NESIE_REQUIRE(nb % ng == 0 && dy && x && workspace, W);
  NESIE_REQUIRE(workspace_bytes >= nesie_pw_wgrad_workspace_bytes(nb, ng, co, ci, p), W);
  NESIE_REQUIRE((((uintptr_t)dy | (uintptr_t)x) & 15) == 0 && (dy_bstride & 3) == 0 && (x_bstride & 3) == 0, W);
  NESIE_REQUIRE((long long)(co > ci ? co : ci) * p < (1ll << 30), W);
  const int nwg = pw_wgrad_nwg(nb, ng, p);
  float *partial = (float *)workspace;
  const float lo = x_relu ? 0.f : -__builtin_inff();
#define L(CO16, CI16, WM, WN)                                                                    \
  do {                                                                                           \
    const size_t lds = (size_t)2 * (CO16 + CI16) * 16 * 36 * sizeof(float);                      \
    if (x_coef) {                                                                                \
      auto kern = pw_wgrad_kernel<CO16, CI16, WM, WN, true>;                                     \
      static bool attr = false;                                                                  \
      if (!attr) {                                                                               \
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        attr = true;                                                                             \
      }                                                                                          \
      hipLaunchKernelGGL(kern, dim3(nwg * ng), dim3(512), lds, s, nb, ng, co, ci, p, dy,         \
                         dy_bstride, x, x_bstride, x_coef, lo, partial, nwg);                    \
    } else {                                                                                     \
      auto kern = pw_wgrad_kernel<CO16, CI16, WM, WN, false>;                                    \
      static bool attr = false;                                                                  \
      if (!attr) {                                                                               \
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        attr = true;                                                                             \
      }                                                                                          \
      hipLaunchKernelGGL(kern, dim3(nwg * ng), dim3(512), lds, s, nb, ng, co, ci, p, dy,         \
                         dy_bstride, x, x_bstride, x_coef, lo, partial, nwg);                    \
    }                                                                                            \
  } while (0)
  if (co <= 64 && ci <= 64) L(4, 4, 2, 4);
  else if (co <= 128 && ci <= 64) L(8, 4, 4, 2);
  else if (co <= 128 && ci <= 128) L(8, 8, 2, 4);
  else if (co <= 128 && ci <= 192) L(8, 12, 2, 4);
  else if (co <= 128 && ci <= 256) L(8, 16, 2, 4);
  else if (co <= 128 && ci <= 320) L(8, 20, 2, 4);
  else L(16, 8, 4, 2);
#undef L
  const int total = co * ci;
  hipLaunchKernelGGL(pw_wgrad_reduce_kernel, dim3(cdiv(total, 64), ng), dim3(1024), 0, s, total, nwg,
                     partial, dw);
  return check_launch(W);
}
