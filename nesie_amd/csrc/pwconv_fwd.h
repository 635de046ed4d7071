// Pointwise (1x1) convolution layers of the grouped per-seed MLPs on the fp32 matrix cores of
// gfx950: the shared MLPs of the set-abstraction stack (reference mmdet3d/ops/pointnet_modules/
// point_sa_module.py:277-289 built from mmcv ConvModule(Conv2d 1x1, BN2d, ReLU), pooled at
// :136-158), the quality head's MiniPointNets (models/dense_heads/side_pooling_module.py:343-370)
// and the 1-D per-seed / per-proposal stacks (vote_module.py:65-74, reliable_conv_bbox_module.py:
// 112-141, point_fp_module.py:31-37).
//
//     Y[n] = W[n % ng] . act(X[n])        X[n] (K x P) and Y[n] (Cout x P) row-major, positions
//                                         contiguous (NCHW as it stands: no transposes)
//     act(v) = max(scale[k] * v + bias[k], lo)   the PREVIOUS layer's folded BatchNorm + ReLU
//
// What shapes the kernel (tools/pwbench, modes 10-12, re-measured in round 3): on gfx950 an fp32
// MFMA holds its SIMD's instruction issue for its whole 32 cycles.  A second wave on the SIMD gets
// NO issue slot while the first streams MFMAs (its VALU / SALU / LDS / VMEM stream finishes after
// the MFMA wave has ended), and every instruction a wave puts between its own MFMAs costs its
// issue time in full: a conflict-free LDS read ~7 cycles WHATEVER its width (b32, b64, 2 x b32,
// b128 alike), a global_load_dwordx4 ~5, a dense VALU instruction ~3, a packed fp32 fma ~13.
// Matrix-pipe utilisation is therefore
//     MFMA cycles / (MFMA cycles + issue cycles of everything else + cycles in which NO wave of
//                    the SIMD has an instruction to issue)
// and the design (a) minimises the instruction COUNT per MFMA and (b) keeps a burst of loads from
// ever standing between a wave and its MFMAs:
//   * weight-stationary: a persistent workgroup owns a run of (n, position-tile) tiles of one
//     weight group; a wave keeps 16 (RW = 1) or 32 (RW = 2) rows of W in registers for the whole
//     launch;
//   * ONE ds_read_b128 FEEDS FOUR MFMAs (eight at RW = 2).  The operand tile sits in LDS as it
//     sits in memory, [k][position]; lane (l16, quad) reads X[4 kk + quad][64 b + 4 l16 .. + 3]
//     -- four CONSECUTIVE POSITIONS of one k -- and hands component e to MFMA e, whose 16 "rows"
//     are then the positions 64 b + 4 i + e.  Any bijection between MFMA rows and positions is a
//     valid GEMM; this one makes the operand fetch a quarter of the instructions of a
//     one-value-per-lane fetch (round 2: ds_read2st64_b32, two values per instruction) and is
//     bank-conflict-free without a swizzle (row pitch = 64 words: the 16-lane groups of a b128
//     read cover all 64 banks);
//   * v_mfma_f32_16x16x4_f32 computes the TRANSPOSED block D[position][channel] (A = X^T
//     fragment, B = W^T fragment), so with that bijection a lane ends with 16 CONSECUTIVE positions
//     (16 quad .. + 15) of one output channel: per-channel statistics, the 16-position pooling
//     group and the row bias are lane-local, and the block leaves as four 16-byte stores;
//   * K is walked in sub-tiles of KT = 64 / 128 / 144 rows x PT = 64 WC positions (32 KB at
//     128 x 64): K <= 144 is one sub-tile, K <= 288 two, with the accumulators carried across.
//     Two LDS buffers; 64 KB per workgroup at K <= 128, so two workgroups share a CU where the
//     registers allow it (<= 128 VGPRs; tools/isa_regs.py prints the table);
//   * the loads of sub-tile s + 2 are issued INSIDE the MFMA loop of sub-tile s (its last third), one
//     global_load_dwordx4 at a time right behind the LDS write that frees its registers (the
//     write moves sub-tile s + 1, loaded during s - 1, into the other buffer, with the previous
//     layer's BatchNorm + ReLU applied in registers on the way).  A load has a whole sub-tile
//     (>= 4096 MFMA cycles) to arrive, the memory system sees a steady stream instead of one
//     burst per barrier (round 2 issued a tile's loads in one block behind the barrier: with every
//     CU doing so at once the issue itself stalled for 3 000 - 5 000 cycles per tile, pwbench
//     stamps), and one barrier per sub-tile remains;
//   * staging slots are addressed by a wave-uniform base (SGPR) + ONE per-lane offset register,
//     LDS offsets are immediates; rows beyond K are never loaded: their LDS rows are zeroed once
//     (they meet zero weights), a partly valid slot re-reads row K - 1;
//   * epilogue straight from the accumulators: optional output-side row bias / channel bias,
//     the raw conv output, this layer's own BatchNorm statistics as per-wave SHIFTED sums
//     (count, shift, sum(y - shift), sum((y - shift)^2): no E[x^2] - E[x]^2 cancellation;
//     merged in fp64 by pw_stats_finalize_kernel with Chan's formula), the max / min over each
//     group of 16 or 32 consecutive positions with the position of each (pooling tail), and for
//     input-gradient launches the two sums of the BatchNorm backward.
// The K x P operand is read once, Y written once (or never, for a pooled tail): 2 tensor passes
// per layer where conv + statistics + normalise cost 5.  Cout = 256 at K <= 144 is ONE workgroup
// of 8 waves x 32 rows (round 2 ran two 128-row workgroups that each staged the tile).
// (this header: the kernel template and its per-geometry launcher; every geometry is instantiated
// in a translation unit of its own, pwconv_g*.hip, so that the library builds in parallel)
#pragma once
#include "common.h"
#include <stdlib.h>
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

template <int OFF>
__device__ __forceinline__ f32x4 lds_read_b128(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read_b128 reach");
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

template <int OFF>
__device__ __forceinline__ void lds_write_b128(unsigned addr, f32x4 v) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_write_b128 reach");
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}

// 16 bytes per lane from sbase (wave-uniform) + voff (per-lane byte offset).  A plain load: the
// compiler places the vmcnt wait at the first use (an asm load + a separate asm wait let the
// register allocator copy the destination before the wait: stale words in the first tile).
__device__ __forceinline__ f32x4 load16_saddr(unsigned voff, const void *sbase) {
  return *(const f32x4 *)((const char *)sbase + voff);
}

// *(float4 *)(sbase + voff + IMM) = v.  A PLAIN store, like the loads: the compiler's vmcnt
// bookkeeping must see it.  (Round 2 and the first round-3 build issued the stores from inline asm:
// invisible to the wait-count pass, so the counted `s_waitcnt vmcnt(N)` in front of the next
// staging write -- N = the loads issued after the one needed -- also waited for the epilogue's
// four stores, which are YOUNGER than those loads but were not in the count: every tile began by
// sitting out a store acknowledgement from HBM.)
template <int IMM>
__device__ __forceinline__ void store16_saddr(unsigned voff, f32x4 v, void *sbase) {
  *(f32x4 *)((char *)sbase + voff + IMM) = v;
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
  // The first SPC = 128 / 256 operand rows are not loaded but BUILT: row c holds one value per group
  // of ns positions, sp_ent[n][group][c] = (value, position inside the group), zero elsewhere -- the
  // gradient a max-pool hands back (input gradient of a pooled tail, pool_tail.hip).  The loaded
  // operand (x, in_coef) supplies rows SPC .. K - 1; K = KH * KT exactly.
  PW_SPARSE128 = 256,
  PW_SPARSE256 = 512,
  // A 64-row tensor that is the raw output of a 4 -> 64 convolution, Z0 = W0 . X4 (SA1's first
  // layer), is not read but REBUILT from the 4 rows of X4 (k4_dot below: the same four roundings
  // everywhere, so that the forward's ReLU and the backward's masks agree):
  PW_K4IN = 1024,   // ... as the operand: x = X4 (nb, 4, p), k = 64, in_coef = the folded norm of Z0
  PW_K4Z = 2048,    // ... as the Z of PW_BNRED: bn_z = X4; additionally leaves sum(g . X4[j]), j = 0 .. 3, per
                    // channel and slot (k4_gpart) -- with them the FIRST layer's weight gradient needs no pass over g
};

// Z0[m] at one position from the four input rows: ((w0 x0 + w1 x1) + w2 x2) + w3 x3 as an fma chain
__device__ __forceinline__ float k4_dot(const float4 w, float x0, float x1, float x2, float x3) {
  return __builtin_fmaf(w.w, x3, __builtin_fmaf(w.z, x2, __builtin_fmaf(w.y, x1, w.x * x0)));
}

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
  int tiles_per_batch, nwg_g, nhalf;       // nhalf: workgroups per tile along Cout
  int xcd_map;         // the nhalf workgroups of a tile stream sit on ONE XCD (grid % (8 nhalf) == 0)
  const float2 *sp_ent; int sp_ns_shift, sp_groups;   // PW_SPARSE*: entries, log2(ns), groups per batch element
  int w_stage;         // 1: row-major weights come in through LDS (NESIE_PW_WSTAGE=0: lane loads, A/B switch)
  int rev;             // 1: the tile stream runs from the LAST tile of the group to the first (big operands: nesie_lib.hip)
  const float *k4_w;   // PW_K4IN / PW_K4Z: W0 (64, 4) row-major
  float *k4_gpart;     // PW_K4Z: [ng * cout][nslots][4]
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
// per-workgroup wall clock (100 MHz): entry, loop start, loop end, after the last store
#define STAMP_WG(which)                                                                    \
  if (a.stamps && threadIdx.x == 0 && blockIdx.x < 1024)                                    \
    a.stamps[512 + blockIdx.x * 4 + (which)] = __builtin_amdgcn_s_memrealtime();
#else
#define STAMP(slot)
#define STAMP_CLK(which)
#define STAMP_WG(which)
#endif

// workgroups per CU the launch is sized for: limited by the two LDS buffers (160 KB per CU) and
// capped so that each wave keeps a register budget that fits its W rows (512 / waves per SIMD)
constexpr int pw_per_cu(int kt16, int kh, int nw, int wc, int rw) {
  const int lds = 2 * 16 * kt16 * 64 * wc * 4;
  int n = 160 * 1024 / lds;
  const int cap = nw == 4 ? (kt16 * kh * rw <= 4 ? 4 : 2) : (kt16 * kh * rw <= 8 ? 2 : 1);
  return n < cap ? n : cap;
}
// ... and the matching minimum waves per SIMD for __launch_bounds__
constexpr int pw_min_waves(int kt16, int kh, int nw, int wc, int rw) {
  return pw_per_cu(kt16, kh, nw, wc, rw) * nw / 4;
}

// KT16 = rows of a K sub-tile / 16; KH = sub-tiles along K; WR x WC waves, each RW x 16 output
// rows x 64 positions of a PT = 64 WC position tile; EPI = epilogue / prologue flags; PG = pooling
// granule (16 or 32)
// RAGGED: some output rows of some wave lie beyond Cout (Cout not a multiple of the workgroup's
// rows): every per-row access is guarded.  The common case has no guards and no branches.
template <int KT16, int KH, int WR, int WC, int RW, int EPI, int PG, bool RAGGED>
__global__ __launch_bounds__(WR *WC * 64, pw_min_waves(KT16, KH, WR *WC, WC, RW))
void pw_fwd_kernel(const PwFwd a) {
  constexpr int NW = WR * WC, NT = NW * 64, KT = 16 * KT16, KQ = KT / 4, PT = 64 * WC;
  constexpr int TILE = KT * PT, CPR = PT / 4;            // floats per buffer, 16-byte chunks per row
  constexpr int NCH = KT * CPR;                          // chunks of a sub-tile
  constexpr int NX = (NCH + NT - 1) / NT;                // staging slots per thread
  constexpr bool EVEN = NCH == NX * NT;
  constexpr int ROWSTEP = NT / CPR;                      // rows between two slots of a thread
  constexpr int CROWS = WR * RW * 16;                    // output rows per workgroup
  constexpr int SPC = (EPI & PW_SPARSE128) ? 128 : (EPI & PW_SPARSE256) ? 256 : 0;   // built operand rows
  constexpr bool K4Z = (EPI & PW_K4Z) != 0;
  static_assert(SPC % ROWSTEP == 0 && (SPC == 0 || (NCH == NX * NT && SPC < KH * KT)), "sparse rows");
  static_assert(NT % CPR == 0 && (PG == 16 || PG == 32), "tile");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  STAMP_WG(0)

  const int tid = threadIdx.x, lane = tid & 63, quad = lane >> 4, l16 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;
  // block -> (weight group g, row half, rank inside the group)
  // Workgroups go to the 8 XCDs round-robin by block index and every XCD has its own L2.  The nhalf
  // workgroups that walk the SAME tile stream (one per 128-row block of Cout) read the same X tiles:
  // with consecutive block indices they sat on nhalf different XCDs and X crossed the fabric nhalf
  // times (PMC: 807 MB fetched for 201 + 403 MB of operands at Cout = 256).  Block b = 8 slot + xcd:
  // the row blocks of a stream are consecutive SLOTS of one XCD, start together and keep pace, so
  // all but the first find the tile in that XCD's L2.
  int half, sid;
  if (a.xcd_map) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    half = slot % a.nhalf;
    sid = xcd + 8 * (slot / a.nhalf);
  } else {
    half = blockIdx.x % a.nhalf;
    sid = blockIdx.x / a.nhalf;
  }
  const int g = sid % a.ng, rank = sid / a.ng;
  const int k = a.k;
  const int c0 = half * CROWS;                           // first output row of this workgroup
  const int cout = a.cout;
  const long long p = a.p;

  // ---- staging: slot i of a thread = rows SR(i) + tid / CPR of the sub-tile, 16-byte column
  // tid % CPR, SR(i) = i ROWSTEP (a ragged last slot starts at KT - ROWSTEP instead and overlaps
  // its predecessor: every thread is active in every slot, no execution masks).  The loop body
  // is STRAIGHT-LINE code: no slot is skipped and nothing is branched around, so that the
  // compiler's wait-count pass can count the loads and stores in flight exactly (behind a branch
  // it falls back to `s_waitcnt vmcnt(0)` in front of every staging write and every re-load,
  // which made each tile sit out the previous tile's store acknowledgements).
  //  * sub-tiles kh < KH - 1 are whole (the host picks KH = ceil(K / KT)): wave-uniform base +
  //    ONE per-lane offset register;
  //  * the last sub-tile may end inside a slot: per-slot per-lane offsets with the row clamped to
  //    K - 1 (the copies of row K - 1 land in pad rows of the LDS tile and meet zero weights);
  //  * past the workgroup's last tile the loads read one harmless cache line of tile 0 and the
  //    writes fill a buffer nobody reads.
  const int srow = tid / CPR, scol = tid % CPR;
  auto slot_row = [](int i) constexpr { return (EVEN || i + 1 < NX) ? i * ROWSTEP : KT - ROWSTEP; };
  const unsigned voff0 = (unsigned)(((size_t)srow * p + scol * 4) * 4);
  unsigned voff_last[NX];                                 // last sub-tile: rows clamped to K - 1
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    int row = (KH - 1) * KT + slot_row(i) + srow;
    row = row < k ? row : k - 1;
    voff_last[i] = (unsigned)(((size_t)(row - (KH - 1) * KT) * p + scol * 4) * 4);
  }
  const unsigned lw0 = lds_addr(lds) + (unsigned)((srow * PT + scol * 4) * 4);

  // this wave's rows of W, for the whole launch: lane (l16, quad) holds W[m][kh KT + 4 kk + quad]
  // Row-major weights (forward launches: w_cs == 1) come in THROUGH LDS: fetched lane by lane those KH KQ
  // words per lane are KH KQ load instructions that each touch 16 rows -- 8 192 cache-line requests per
  // workgroup, 7.5 us of a 23 us one-tile launch and 10 us of every larger one (tools/pwbench stamps:
  // loop start 9.4 us after entry, 1.9 us with the loads stubbed out).  Instead every wave copies its
  // 16 RW rows of a K sub-tile with 16-byte loads (a row = KT contiguous floats) into a block of its own
  // inside the operand buffers, which are not in use yet, and reads its words back; columns are
  // XOR-swizzled by the row (word k ^ 4 (row & 15)) so that the 16 rows x 4 words of a read hit 64
  // banks.  Same values, same order of the products: bit-identical results.  (The transposed views of the
  // input-gradient launches keep their lane loads: there a load instruction touches 4 lines, not 16, and the
  // same staging with a word-by-word scatter into the block was slower: 13.17 vs 13.02 ms for the step.)
  float wreg[RW][KH * KQ];
  // (sub-tiles of 64 or 128 rows: the swizzle stays inside a row; the 96- and 144-row geometries keep the lane loads)
  const bool w_rows = KT % 64 == 0 && (a.w_stage & 1) && a.w_cs == 1 && (a.w_rs & 3) == 0 && (a.w_gs & 3) == 0 && ((uintptr_t)a.w & 15) == 0;   // (uniform)
  if (w_rows) {
    static_assert(WR * WC * RW * 16 <= 2 * PT, "the waves' blocks fit the two operand buffers");
    constexpr int WROWS = RW * 16, C4 = KT / 4, NL = (WROWS * C4 + 63) / 64;
    float *wl = lds + wave * (WROWS * KT);
    const int m0 = c0 + wr * WROWS;
    const float *wg0 = a.w + (size_t)g * a.w_gs + (size_t)m0 * a.w_rs;
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) {
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int f = i * 64 + lane, row = f / C4, c4 = f % C4, kcol = kh * KT + 4 * c4;
        if (NL * 64 == WROWS * C4 || f < WROWS * C4) {
          f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (m0 + row < cout) {
            const float *src = wg0 + (size_t)row * a.w_rs + kcol;
            if (kcol + 3 < k) {
              v = *(const f32x4 *)src;
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = kcol + e < k ? src[e] : 0.f;
            }
          }
          *(f32x4 *)(wl + row * KT + ((4 * c4) ^ (4 * (row & 15)))) = v;
        }
      }
#pragma unroll
      for (int rw = 0; rw < RW; ++rw)
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk)
          wreg[rw][kh * KQ + kk] = wl[(rw * 16 + l16) * KT + ((4 * kk + quad) ^ (4 * l16))];
    }
    __syncthreads();      // the blocks lie in the operand buffers: nobody stages a tile before every wave has read
  } else {
#pragma unroll
    for (int rw = 0; rw < RW; ++rw) {
      const int m = c0 + (wr * RW + rw) * 16 + l16;
      const float *wg = a.w + (size_t)g * a.w_gs + (size_t)m * a.w_rs;
#pragma unroll
      for (int kk = 0; kk < KH * KQ; ++kk) {
        const int kx = 4 * kk + quad;
        wreg[rw][kk] = (m < cout && kx < k) ? wg[(size_t)kx * a.w_cs] : 0.f;
      }
    }
  }
  // (scale, bias) of the previous layer's folded BatchNorm for the rows of every staging slot
  float2 cof[(EPI & PW_AFFINE) ? KH : 1][(EPI & PW_AFFINE) ? NX : 1];
  if (EPI & PW_AFFINE) {
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        int row = kh * KT + slot_row(i) + srow;
        row = row < k ? row : k - 1;
        if (SPC > 0) row = row >= SPC ? row - SPC : 0;   // (slots of built rows never use theirs)
        cof[kh][i] = *(const float2 *)(a.in_coef + ((size_t)g * (k - SPC) + row) * 4);
      }
  }
  // PW_K4IN: W0 rows of this thread's four operand rows; the slot index doubles as the X4 row it fetches
  float4 k4w[(EPI & PW_K4IN) ? NX : 1];
  if constexpr ((EPI & PW_K4IN) != 0) {
    static_assert(KH == 1 && KT == 64 && NX == 4 && EVEN && SPC == 0, "PW_K4IN: one 64-row sub-tile, four slots");
#pragma unroll
    for (int i = 0; i < NX; ++i) k4w[i] = *(const float4 *)(a.k4_w + (size_t)(slot_row(i) + srow) * 4);
  }
  // built rows: entry of (group of this lane's 16-byte chunk, row srow) relative to the tile's first
  // group and the slot's first row; and the chunk's first position inside its group
  const unsigned sp_voff = SPC > 0 ? (unsigned)(((((4 * scol) >> a.sp_ns_shift) * SPC) + srow) * 8) : 0u;
  const int sp_pos0 = SPC > 0 ? ((4 * scol) & ((1 << a.sp_ns_shift) - 1)) : 0;
  const int tpb = a.tiles_per_batch, nwg = a.nwg_g;
  const int ntiles = (a.nb / a.ng) * tpb;

  // ---- the stream of sub-tiles this workgroup stages: (tile, kh) in execution order
  struct Cursor { int t, q, r, kh; };                    // tile index, (batch of the group, tile of the batch), sub-tile
  const int dq = nwg / tpb, dr = nwg % tpb;
  auto advance = [&](Cursor &c) {
    if (KH > 1 && c.kh + 1 < KH) { ++c.kh; return; }
    c.kh = 0;
    c.t += nwg; c.q += dq; c.r += dr;
    if (c.r >= tpb) { c.r -= tpb; ++c.q; }
  };
  // (batch, tile) a cursor stands for: as it counts, or mirrored (a.rev)
  const int nq = a.nb / a.ng;
  auto QQ = [&](const Cursor &c) { return a.rev ? nq - 1 - c.q : c.q; };
  auto RR = [&](const Cursor &c) { return a.rev ? tpb - 1 - c.r : c.r; };
  f32x4 stg[NX];
  // PW_K4Z: one more staging register -- this thread's 16 bytes of the tile's X4 rows (row (tid / 16) % 4)
  f32x4 stx = {0.f, 0.f, 0.f, 0.f};
  const unsigned k4z_goff = (unsigned)((((size_t)((tid >> 4) & 3)) * p + (tid & 15) * 4) * 4);
  const unsigned k4z_lw = lds_addr(lds) + (unsigned)(2 * TILE * 4) + (unsigned)(((((tid >> 4) & 3) * 64) + (tid & 15) * 4) * 4);
  // loads of slot i of sub-tile c (kh compile-time) into its staging registers; `live` false:
  // the sub-tile does not exist, read the first words of the tensor instead
  auto load_slot = [&](auto ic, auto khc, const Cursor &c, bool live) {
    constexpr int i = decltype(ic)::value, kh = decltype(khc)::value;
    if constexpr (K4Z && i == 0) {
      const float *x4 = a.bn_z + (size_t)(g + a.ng * QQ(c)) * a.bnz_bs + (size_t)RR(c) * PT;   // wave-uniform
      stx = load16_saddr(live ? k4z_goff : 0u, live ? x4 : a.bn_z);
    }
    if constexpr (SPC > 0 && kh * KT + slot_row(i) < SPC) {
      // wave-uniform: first entry of the tile's first group, row kh KT + slot_row(i)
      const float2 *eb = a.sp_ent + ((size_t)(g + a.ng * QQ(c)) * a.sp_groups + (size_t)(((long long)RR(c) * PT) >> a.sp_ns_shift)) * SPC
                         + (kh * KT + slot_row(i));
      const float2 e = *(const float2 *)((const char *)(live ? eb : a.sp_ent) + (live ? sp_voff : 0u));
      stg[i][0] = e.x;
      stg[i][1] = e.y;
      return;
    }
    if constexpr ((EPI & PW_K4IN) != 0) {
      // slot i = row i of X4 at this thread's 16-byte column (every operand row of the column needs all four)
      const float *x4 = a.x + (size_t)(g + a.ng * QQ(c)) * a.x_bs + (size_t)RR(c) * PT + (size_t)i * p;   // wave-uniform
      stg[i] = load16_saddr(live ? (unsigned)(scol * 16) : 0u, live ? x4 : a.x);
      return;
    }
    const float *xb = a.x + (size_t)(g + a.ng * QQ(c)) * a.x_bs + (size_t)RR(c) * PT + (long long)(kh * KT - SPC) * p;
    if constexpr (kh < KH - 1) {
      const float *xs = live ? xb + (size_t)slot_row(i) * p : a.x;   // wave-uniform
      stg[i] = load16_saddr(live ? voff0 : 0u, xs);
    } else {
      stg[i] = load16_saddr(live ? voff_last[i] : 0u, live ? xb : a.x);
    }
  };
  // previous layer's BatchNorm + ReLU in registers, then into LDS buffer `buf` (0 / 1)
  auto write_slot = [&](auto ic, auto khc, int buf) {
    constexpr int i = decltype(ic)::value;
    f32x4 q = stg[i];
    if constexpr ((EPI & PW_K4IN) != 0) {
      // (every slot needs all four rows: the four writes of a sub-tile run together, in front of its re-loads)
#pragma unroll
      for (int e = 0; e < 4; ++e) q[e] = k4_dot(k4w[i], stg[0][e], stg[1][e], stg[2][e], stg[3][e]);
    }
    if constexpr (K4Z) {
      // X4 of the tile whose operand this is -> its 1 KB next to the operand buffer (all 256 threads
      // write; rows repeat four times with the same words)
      if constexpr (i == 0) lds_write_b128<0>(k4z_lw + (unsigned)(buf * 1024), stx);
    }
    if constexpr (SPC > 0 && decltype(khc)::value * KT + slot_row(i) < SPC) {
      const int at = __float_as_int(q[1]) - sp_pos0;
      const float v = q[0];
      q[0] = at == 0 ? v : 0.f;
      q[1] = at == 1 ? v : 0.f;
      q[2] = at == 2 ? v : 0.f;
      q[3] = at == 3 ? v : 0.f;
    } else if constexpr ((EPI & PW_AFFINE) != 0) {
      const float2 co = cof[decltype(khc)::value][i];
      const float lo = a.in_lo;
      q[0] = fmaxf(__builtin_fmaf(q[0], co.x, co.y), lo);
      q[1] = fmaxf(__builtin_fmaf(q[1], co.x, co.y), lo);
      q[2] = fmaxf(__builtin_fmaf(q[2], co.x, co.y), lo);
      q[3] = fmaxf(__builtin_fmaf(q[3], co.x, co.y), lo);
    }
    lds_write_b128<slot_row(i) * PT * 4>(lw0 + (unsigned)(buf * TILE * 4), q);
  };

  // statistics state: a lane holds ONE channel per row set (row c0 + 16 (wr RW + rw) + l16) and 16
  // consecutive positions of it per tile
  float s1[RW], s2[RW], shift[RW];
#pragma unroll
  for (int rw = 0; rw < RW; ++rw) s1[rw] = s2[rw] = shift[rw] = 0.f;
  int ntile_done = 0;
  // Which 16-byte chunk of an operand row a lane fetches decides which output positions it ends
  // with.  PERM (every variant without a pooled tail): lane l16 fetches chunk 4 (l16 % 4) + l16 / 4,
  // so accumulator (e, r) of lane (channel, quad) is position 16 r + 4 quad + e and the r-th
  // 16-byte store of the four quads of a channel is one contiguous 64-byte segment (with the
  // identity order a lane owns 16 consecutive positions and every store instruction writes 64
  // separate 16-byte pieces: the epilogue's eight stores then took 2 000 - 4 000 cycles per tile,
  // pwbench stamps).  A pooled tail WITHOUT a store keeps the identity order: position 16 quad + 4 r + e,
  // the lane's 16 values ARE a pooling group; with a store the four quads of a channel share every
  // group and reduce it with two cross-lane exchanges.
  // (a pooled tail that stores nothing keeps lane-local groups -- unless it also leaves statistics:
  // SA tails, whose sums then run in the order of the storing variant, bit for bit)
  constexpr bool PERM = !(EPI & PW_POOL) || (EPI & (PW_STORE | PW_STATS));
  constexpr int RS = PERM ? 16 : 4;           // positions between accumulator registers r and r + 1
  const int q0 = wc * 64 + (PERM ? 4 : 16) * quad;   // this lane's first position inside a tile
  // byte offset of (row m, position q0) from the tile's first output word, per row set
  unsigned roff[RW];
#pragma unroll
  for (int rw = 0; rw < RW; ++rw)
    roff[rw] = (unsigned)(((size_t)(c0 + (wr * RW + rw) * 16 + l16) * p + q0) * 4);

  // PW_BNRED: the raw output Z of the layer whose activation gradient this launch produces, at
  // this lane's output elements (loaded ahead of the MFMAs of the tile's last sub-tile), and the sums
  static_assert(!K4Z || ((EPI & PW_BNRED) && !RAGGED && WC == 1), "PW_K4Z rides on PW_BNRED");
  f32x4 zv[((EPI & PW_BNRED) && !K4Z) ? RW : 1][4];
  float4 k4zw[K4Z ? RW : 1];               // ... and W0 rows of its channels
  float gx[K4Z ? RW : 1][4];               // ... sums of g . X4[j]
  float r0[RW], r1[RW];
  float4 zc[RW];
#pragma unroll
  for (int rw = 0; rw < RW; ++rw) {
    r0[rw] = r1[rw] = 0.f;
    zc[rw] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI & PW_BNRED) {
      const int m = c0 + (wr * RW + rw) * 16 + l16;
      if (!RAGGED || m < cout) zc[rw] = *(const float4 *)(a.bn_coef + ((size_t)g * cout + m) * 4);
      if constexpr (K4Z) {
        k4zw[rw] = *(const float4 *)(a.k4_w + (size_t)m * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) gx[rw][j] = 0.f;
      }
    }
  }
  auto row_ok = [&](int rw) { return !RAGGED || c0 + (wr * RW + rw) * 16 + l16 < cout; };
  auto load_z = [&](int n, long long p0) {
    const float *zt = a.bn_z + (size_t)n * a.bnz_bs + p0;   // wave-uniform
    if constexpr (K4Z) return;      // (X4 of the tile lies in LDS: staged with the operand)
#pragma unroll
    for (int rw = 0; rw < RW; ++rw) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        zv[rw][r] = row_ok(rw) ? load16_saddr(roff[rw] + 4u * RS * r, zt) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  // output-side additions: the channel bias once, the row bias of a tile at the top of its last
  // sub-tile (a load inside the epilogue would be waited for on the spot)
  float cbias[RW], rbias[RW][4];
#pragma unroll
  for (int rw = 0; rw < RW; ++rw) {
    cbias[rw] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) rbias[rw][r] = 0.f;
    if (EPI & PW_BIAS) {
      if (row_ok(rw)) cbias[rw] = a.bias[g * cout + c0 + (wr * RW + rw) * 16 + l16];
    }
  }
  auto load_row_bias = [&](int n, long long p0) {
    const size_t cols = (size_t)(p >> a.rb_shift);
#pragma unroll
    for (int rw = 0; rw < RW; ++rw)
#pragma unroll
      for (int r = 0; r < 4; ++r) {   // (groups of >= 64 positions: four loads of one word)
        const size_t col = (size_t)((p0 + q0 + RS * r) >> a.rb_shift);
        rbias[rw][r] = row_ok(rw) ? a.row_bias[((size_t)n * cout + c0 + (wr * RW + rw) * 16 + l16) * cols + col] : 0.f;
      }
  };

  // accumulators: acc[rw][e][r] = Y[row set rw][position q0 + RS r + e]
  f32x4 acc[RW][4];
  int k4z_buf = 0;                         // PW_K4Z: the LDS half that holds the current tile's X4
  auto epilogue = [&](int n, long long p0) {
    float *ytile = a.y + (size_t)n * a.y_bs + p0;        // wave-uniform
    static_for<0, RW>([&](auto rwc) {
      constexpr int rw = decltype(rwc)::value;
      const int m = c0 + (wr * RW + rw) * 16 + l16;
      if (EPI & (PW_ROWBIAS | PW_BIAS)) {
        f32x4 add;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          add[r] = ((EPI & PW_ROWBIAS) ? rbias[rw][r] : 0.f) + ((EPI & PW_BIAS) ? cbias[rw] : 0.f);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[rw][e] += add;
      }
      if (EPI & PW_STORE) {
        if (row_ok(rw)) {
          static_for<0, 4>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            store16_saddr<4 * RS * r>(roff[rw], (f32x4){acc[rw][0][r], acc[rw][1][r], acc[rw][2][r], acc[rw][3][r]}, ytile);
          });
        }
      }
      if constexpr (K4Z) {
        // X4 at this lane's positions 16 r + 4 quad + e from the tile's LDS copy (one register set:
        // a second one, read a step ahead, spilled -- the SIMD's other waves cover the LDS latency)
        f32x4 xr[4];
        const unsigned xa = lds_addr(lds) + (unsigned)(2 * TILE * 4) + (unsigned)(k4z_buf * 1024) + (unsigned)(16 * quad);
        static_for<0, 4>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          static_for<0, 4>([&](auto jc) { xr[decltype(jc)::value] = lds_read_b128<decltype(jc)::value * 256 + r * 64>(xa); });
          lgkm_wait<0>();
          // (the reads are asm: pin their destinations BEHIND the wait, or the uses are scheduled in front of it)
          asm volatile("" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x0 = xr[0][e], x1 = xr[1][e], x2 = xr[2][e], x3 = xr[3][e];
            const float z = k4_dot(k4zw[rw], x0, x1, x2, x3);
            const float gg = __builtin_fmaf(z, zc[rw].x, zc[rw].y) > 0.f ? acc[rw][e][r] : 0.f;
            r0[rw] += gg;
            r1[rw] += gg * ((z - zc[rw].z) * zc[rw].w);
            gx[rw][0] = __builtin_fmaf(gg, x0, gx[rw][0]);
            gx[rw][1] = __builtin_fmaf(gg, x1, gx[rw][1]);
            gx[rw][2] = __builtin_fmaf(gg, x2, gx[rw][2]);
            gx[rw][3] = __builtin_fmaf(gg, x3, gx[rw][3]);
          }
          // (... and keep the next reads behind these uses)
          asm volatile("" :: "v"(xr[0]), "v"(xr[1]), "v"(xr[2]), "v"(xr[3]));
        });
      } else if (EPI & PW_BNRED) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float z = zv[rw][r][e];
            const float gg = __builtin_fmaf(z, zc[rw].x, zc[rw].y) > 0.f ? acc[rw][e][r] : 0.f;
            r0[rw] += gg;
            r1[rw] += gg * ((z - zc[rw].z) * zc[rw].w);
          }
      }
      if (EPI & PW_STATS) {
        const float first = __shfl(acc[rw][0][0], l16, 64);   // first value of the channel
        shift[rw] = ntile_done == 0 ? first : shift[rw];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = acc[rw][e][r] - shift[rw];
            s1[rw] += d;
            s2[rw] += d * d;
          }
      }
      if (EPI & PW_POOL) {
        const size_t prow = (size_t)(p / PG);
        if constexpr (!PERM) {
          // the lane's 16 values ARE one 16-position group (position 4 r + e inside it); a
          // 32-position group is the lane pair (quad, quad ^ 1): both lanes end with the pair's
          // extremum and both store it (the same bytes to the same place: no execution mask)
          const size_t pcol = (size_t)((p0 + q0) / PG);
#pragma unroll
          for (int mm = 0; mm < ((EPI & PW_POOLMIN) ? 2 : 1); ++mm) {
            float ev = acc[rw][0][0];
            int at = 0;
#pragma unroll
            for (int u = 1; u < 16; ++u) {            // ascending position: the first extremum wins
              const float v = acc[rw][u % 4][u / 4];
              const bool better = mm ? v < ev : v > ev;
              at = better ? u : at;
              ev = better ? v : ev;
            }
            if (PG == 32) {
              at += 16 * (quad & 1);
              const float oe = __shfl_xor(ev, 16, 64);
              const int oa = __shfl_xor(at, 16, 64);
              const bool take = (mm ? oe < ev : oe > ev) || (oe == ev && oa < at);
              ev = take ? oe : ev;
              at = take ? oa : at;
            }
            if (row_ok(rw)) {
              const size_t o = ((size_t)n * cout + m) * prow + pcol;
              (mm ? a.pool_min : a.pool_max)[o] = ev;
              (mm ? a.arg_min : a.arg_max)[o] = (uint8_t)at;
            }
          }
        } else {
          // accumulator (e, r) is position 16 r + 4 quad + e: group gi = r (PG 16) or r / 2 (PG 32)
          // is spread over the channel's four quads.  Local extremum first (ascending position: the
          // first one wins), then two exchanges (lanes 16 and 32 apart); afterwards every quad
          // holds every group's result and quad gi stores group gi (PG 32: quads 2, 3 repeat 0, 1)
          constexpr int NG = 64 / PG, RPG = PG / 16;           // groups per 64 positions, registers r per group
          const size_t pcol = (size_t)((p0 + 64 * wc) / PG) + (PG == 16 ? quad : (quad & 1));
#pragma unroll
          for (int mm = 0; mm < ((EPI & PW_POOLMIN) ? 2 : 1); ++mm) {
            float ev[NG];
            int at[NG];
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
              ev[gi] = acc[rw][0][gi * RPG];
              at[gi] = 4 * quad;
#pragma unroll
              for (int u = 1; u < 4 * RPG; ++u) {
                const int rr = u / 4, e = u % 4;
                const float v = acc[rw][e][gi * RPG + rr];
                const bool better = mm ? v < ev[gi] : v > ev[gi];
                at[gi] = better ? 16 * rr + 4 * quad + e : at[gi];
                ev[gi] = better ? v : ev[gi];
              }
            }
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
              float oe[NG];
              int oa[NG];
#pragma unroll
              for (int gi = 0; gi < NG; ++gi) { oe[gi] = __shfl_xor(ev[gi], off, 64); oa[gi] = __shfl_xor(at[gi], off, 64); }
#pragma unroll
              for (int gi = 0; gi < NG; ++gi) {
                const bool take = (mm ? oe[gi] < ev[gi] : oe[gi] > ev[gi]) || (oe[gi] == ev[gi] && oa[gi] < at[gi]);
                ev[gi] = take ? oe[gi] : ev[gi];
                at[gi] = take ? oa[gi] : at[gi];
              }
            }
            float sev = ev[0];
            int sat = at[0];
            const int mine = PG == 16 ? quad : (quad & 1);
#pragma unroll
            for (int gi = 1; gi < NG; ++gi) {
              sev = mine == gi ? ev[gi] : sev;
              sat = mine == gi ? at[gi] : sat;
            }
            if (row_ok(rw)) {
              const size_t o = ((size_t)n * cout + m) * prow + pcol;
              (mm ? a.pool_min : a.pool_max)[o] = sev;
              (mm ? a.arg_min : a.arg_max)[o] = (uint8_t)sat;
            }
          }
        }
      }
    });
    if (EPI & PW_STATS) ++ntile_done;
  };

  // ---- pipeline prologue: sub-tile 0 -> buffer 0, sub-tile 1 -> staging registers
  Cursor cur{rank, rank / tpb, rank % tpb, 0};           // the sub-tile being multiplied
  Cursor nxt = cur;                                       // the sub-tile in the staging registers
  using KH0 = std::integral_constant<int, 0>;
  using KH1 = std::integral_constant<int, (KH > 1 ? 1 : 0)>;   // sub-tile 1 is (tile 0, kh 1) or (tile 1, kh 0)
  static_for<0, NX>([&](auto ic) { load_slot(ic, KH0{}, cur, cur.t < ntiles); });
  // The launch-time loads (W, coefficients) are waited for HERE -- behind the first operand loads, which
  // are younger and stay in flight: left alone, the compiler puts their vmcnt(0) in front of the first
  // use inside the tile loop, where it also drains the operand loads.  (Round 3 waited BEFORE the first
  // operand loads: every launch began with two memory latencies back to back.)
#pragma unroll
  for (int rw = 0; rw < RW; ++rw)
#pragma unroll
    for (int kk = 0; kk < KH * KQ; ++kk) asm volatile("" : "+v"(wreg[rw][kk]));
  if (EPI & PW_AFFINE) {
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
      for (int i = 0; i < NX; ++i) asm volatile("" : "+v"(cof[kh][i].x), "+v"(cof[kh][i].y));
  }
  if constexpr ((EPI & PW_K4IN) != 0) {
#pragma unroll
    for (int i = 0; i < NX; ++i) asm volatile("" : "+v"(k4w[i].x), "+v"(k4w[i].y), "+v"(k4w[i].z), "+v"(k4w[i].w));
  }
#pragma unroll
  for (int rw = 0; rw < RW; ++rw) {
    if constexpr (K4Z) asm volatile("" : "+v"(k4zw[rw].x), "+v"(k4zw[rw].y), "+v"(k4zw[rw].z), "+v"(k4zw[rw].w));
    if (EPI & PW_BNRED) asm volatile("" : "+v"(zc[rw].x), "+v"(zc[rw].y), "+v"(zc[rw].z), "+v"(zc[rw].w));
    if (EPI & PW_BIAS) asm volatile("" : "+v"(cbias[rw]));
  }
  static_for<0, NX>([&](auto ic) { write_slot(ic, KH0{}, 0); });
  advance(nxt);
  static_for<0, NX>([&](auto ic) { load_slot(ic, KH1{}, nxt, nxt.t < ntiles); });
  Cursor ld = nxt;                                        // the sub-tile the next loads fetch
  advance(ld);
  int buf = 0;
  int iter = 0;
  (void)iter;
  STAMP_CLK(0)
  STAMP_WG(1)
  // lane address of the operand fetch: row quad, positions 64 wc + 4 chunk .. + 3
  const int chunk = PERM ? 4 * (l16 & 3) + (l16 >> 2) : l16;
  const unsigned xa0 = lds_addr(lds) + (unsigned)((quad * PT + 64 * wc + 4 * chunk) * 4);
  // one K sub-tile on the matrix cores: KQ groups of 4 k, PF operand reads in flight; `stage(kk)`
  // runs behind the MFMAs of group kk
  constexpr int PF = 2;
  auto mfma_subtile = [&](auto khc, int bufsel, auto &&stage) {
    constexpr int kh = decltype(khc)::value;
    const unsigned xa = xa0 + (unsigned)(bufsel * TILE * 4);
    f32x4 frag[PF + 1];
    auto read_frag = [&](auto kkc) {
      constexpr int kk = decltype(kkc)::value;
      constexpr int off = kk * 4 * PT * 4;
      if constexpr (off < 65536) frag[kk % (PF + 1)] = lds_read_b128<off>(xa);
      else frag[kk % (PF + 1)] = lds_read_b128<off - 65536>(xa + 65536u);
    };
    static_for<0, PF>([&](auto kc) { if constexpr (decltype(kc)::value < KQ) read_frag(kc); });
    static_for<0, KQ>([&](auto kkc) {
      constexpr int kk = decltype(kkc)::value;
      // (LDS operations complete in order: waiting for all but the PF youngest reads also waits for
      // a staging ds_write issued in between, which is harmless -- it is older than those reads)
      if constexpr (kk + PF < KQ) {
        read_frag(std::integral_constant<int, kk + PF>{});
        lgkm_wait<PF>();
      } else {
        lgkm_wait<(KQ - 1 - kk)>();
      }
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, 4>([&](auto ec) {
        constexpr int e = decltype(ec)::value;
        static_for<0, RW>([&](auto rwc) {
          constexpr int rw = decltype(rwc)::value;
          // D[position][channel] += X^T[position][k] . W^T[k][channel]
          acc[rw][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[kk % (PF + 1)][e], wreg[rw][kh * KQ + kk], acc[rw][e], 0, 0, 0);
        });
      });
      __builtin_amdgcn_sched_barrier(0);
      stage(kkc);
    });
  };
  auto zero_acc = [&]() {
#pragma unroll
    for (int rw = 0; rw < RW; ++rw)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[rw][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  for (; cur.t < ntiles; ++iter) {
    const int n = g + a.ng * QQ(cur);
    const long long p0 = (long long)RR(cur) * PT;
    static_for<0, KH>([&](auto khc) {
      constexpr int kh = decltype(khc)::value;
      STAMP(0)
      lgkm_wait<0>();          // this thread's ds_writes of the sub-tile
      __builtin_amdgcn_s_barrier();
      STAMP(1)
      const bool have2 = ld.t < ntiles;
      if ((EPI & PW_BNRED) && kh == KH - 1) load_z(n, p0);
      if ((EPI & PW_ROWBIAS) && kh == KH - 1) load_row_bias(n, p0);
      if (kh == 0) zero_acc();
      // Staging sits in the LAST part of the loop (slot i at kk = S0 + 2 i).  The wait-count pass
      // cannot count loads and stores on one counter (mixed event types: it waits for vmcnt(0)), so
      // the first staging write of a sub-tile also waits for the previous tile's stores and this
      // tile's Z / row-bias loads: placed here they are S0 x 128 RW MFMA cycles old and have
      // landed, while the operand loads it really needs are a whole sub-tile old.
      constexpr int S0 = KQ - 2 - 2 * NX;
      static_assert(S0 >= 0, "staging slots do not fit the K loop");
      mfma_subtile(khc, buf, [&](auto kkc) {
        constexpr int kk = decltype(kkc)::value;
        if constexpr (kk >= S0 && (kk - S0) % 2 == 0 && (kk - S0) / 2 < NX) {
          constexpr int i = (kk - S0) / 2;
          // the sub-tile after this one has kh + 1 (mod KH), the one after that kh + 2 (mod KH)
          if constexpr ((EPI & PW_K4IN) != 0) {
            if constexpr (i == 0) {
              static_for<0, NX>([&](auto jc) { write_slot(jc, std::integral_constant<int, 0>{}, buf ^ 1); });
              static_for<0, NX>([&](auto jc) { load_slot(jc, std::integral_constant<int, 0>{}, ld, have2); });
            }
          } else {
            write_slot(std::integral_constant<int, i>{}, std::integral_constant<int, (kh + 1) % KH>{}, buf ^ 1);
            load_slot(std::integral_constant<int, i>{}, std::integral_constant<int, (kh + 2) % KH>{}, ld, have2);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      STAMP(2)
      if constexpr (K4Z) k4z_buf = buf;
      if (kh == KH - 1) epilogue(n, p0);
      STAMP(3)
      cur = nxt;
      nxt = ld;
      advance(ld);
      buf ^= 1;
    });
  }
  STAMP_CLK(1)
  STAMP_WG(2)
  if (EPI & PW_BNRED) {
    const int slot = rank * WC + wc;
#pragma unroll
    for (int rw = 0; rw < RW; ++rw) {
      const int m = c0 + (wr * RW + rw) * 16 + l16;
      float t0 = r0[rw], t1 = r1[rw];
      t0 += __shfl_xor(t0, 16, 64); t1 += __shfl_xor(t1, 16, 64);
      t0 += __shfl_xor(t0, 32, 64); t1 += __shfl_xor(t1, 32, 64);
      if (quad == 0 && m < cout)
        *(float2 *)(a.bn_part + (((size_t)g * cout + m) * a.nslots + slot) * 2) = make_float2(t0, t1);
      if constexpr (K4Z) {
        float4 t;
        float *tv = &t.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float u = gx[rw][j];
          u += __shfl_xor(u, 16, 64);
          u += __shfl_xor(u, 32, 64);
          tv[j] = u;
        }
        if (quad == 0 && m < cout) *(float4 *)(a.k4_gpart + (((size_t)g * cout + m) * a.nslots + slot) * 4) = t;
      }
    }
  }
  if (EPI & PW_STATS) {
    // one partial per wave and row set: (count, shift, sum, sum of squares) of its 16 channels;
    // the four quads hold different positions of the same channel
    const int slot = rank * WC + wc;
#pragma unroll
    for (int rw = 0; rw < RW; ++rw) {
      const int m = c0 + (wr * RW + rw) * 16 + l16;
      float t1 = s1[rw], t2 = s2[rw];
      t1 += __shfl_xor(t1, 16, 64); t2 += __shfl_xor(t2, 16, 64);
      t1 += __shfl_xor(t1, 32, 64); t2 += __shfl_xor(t2, 32, 64);
      if (quad == 0 && m < cout) {
        float4 o;
        o.x = (float)ntile_done * 64.f;
        o.y = shift[rw];
        o.z = t1; o.w = t2;
        *(float4 *)(a.stat_part + (((size_t)g * a.nslots + slot) * cout + m) * 4) = o;
      }
    }
  }
}

// ---- per-geometry launcher: picks the built prologue / epilogue combination -----------------
template <int KT16, int KH, int WR, int WC, int RW>
static int pw_launch_epi(const PwFwd &a, int epi, int pg, int grid, size_t lds, hipStream_t s) {
  const bool ragged = a.cout % (WR * RW * 16) != 0;
#define GO1(E, G, R)                                                                          \
  do {                                                                                        \
    auto kern = pw_fwd_kernel<KT16, KH, WR, WC, RW, E, G, R>;                                 \
    static bool attr = false;                                                                 \
    if (!attr) {                                                                              \
      (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                    \
      attr = true;                                                                            \
      if (getenv("NESIE_PW_OCCUPANCY")) { /* residency report (tools/pw_occupancy.py) */     \
        int nblk = -1;                                                                        \
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, kern, WR * WC * 64, lds);   \
        hipFuncAttributes fa;                                                                 \
        (void)hipFuncGetAttributes(&fa, (const void *)kern);                                  \
        fprintf(stderr, "pw_fwd_kernel<%d,%d,%d,%d,%d,%d,%d,%d>: %d VGPRs, %zu B LDS, %d workgroups/CU (occupancy API), grid %d\n", \
                KT16, KH, WR, WC, RW, E, G, (int)R, fa.numRegs, lds, nblk, grid);             \
      }                                                                                       \
    }                                                                                         \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WR *WC * 64), lds, s, a);                       \
    return NESIE_OK;                                                                          \
  } while (0)
  // GO: built for whole and ragged Cout; GOW: whole only (a ragged Cout there is an error)
#define GO(E, G)                                                                              \
  do {                                                                                        \
    if (ragged) GO1(E, G, true); else GO1(E, G, false);                                       \
  } while (0)
#define GOW(E, G)                                                                             \
  do {                                                                                        \
    if (!ragged) GO1(E, G, false);                                                            \
  } while (0)
#ifdef PW_DEV   // tools/pwbench development build: a few instantiations per geometry
  if (epi == PW_STORE) GO(PW_STORE, 16);
  if (epi == (PW_STORE | PW_BNRED)) GO(PW_STORE | PW_BNRED, 16);
  if (epi == (PW_AFFINE | PW_STORE | PW_STATS)) GO(PW_AFFINE | PW_STORE | PW_STATS, 16);
  if (epi == (PW_AFFINE | PW_STORE | PW_STATS | PW_POOL | PW_POOLMIN) && pg == 16)
    GOW(PW_AFFINE | PW_STORE | PW_STATS | PW_POOL | PW_POOLMIN, 16);
  if (epi == (PW_AFFINE | PW_STORE | PW_STATS | PW_POOL | PW_POOLMIN) && pg == 32)
    GOW(PW_AFFINE | PW_STORE | PW_STATS | PW_POOL | PW_POOLMIN, 32);
  set_error("dev build");
  return NESIE_ERR_UNSUPPORTED;
#else
  // SA1's rebuilt first activation (one geometry: 64 -> 64 with four waves)
  if (epi & (PW_K4IN | PW_K4Z)) {
    if constexpr (KT16 == 4 && KH == 1 && WR == 4 && WC == 1 && RW == 1) {
      if (epi == (PW_K4IN | PW_AFFINE | PW_STORE | PW_STATS)) GOW(PW_K4IN | PW_AFFINE | PW_STORE | PW_STATS, 16);
      if (epi == (PW_K4Z | PW_BNRED)) GOW(PW_K4Z | PW_BNRED, 16);
    }
    set_error("pw_layer_forward: the rebuilt 4 -> 64 operand is built for 64 -> 64 layers only (0x%x)", epi);
    return NESIE_ERR_UNSUPPORTED;
  }
  const int aff = epi & PW_AFFINE;
  const int base = epi & ~PW_AFFINE;
  // the prologue / epilogue combinations the step uses
  if (aff) {
    if (base == PW_STORE) GO(PW_AFFINE | PW_STORE, 16);
    if (base == (PW_STORE | PW_BIAS)) GO(PW_AFFINE | PW_STORE | PW_BIAS, 16);
    if (base == (PW_STORE | PW_STATS)) GO(PW_AFFINE | PW_STORE | PW_STATS, 16);
    if (base == (PW_STORE | PW_POOL) && pg == 16) GOW(PW_AFFINE | PW_STORE | PW_POOL, 16);
    if (base == PW_POOL && pg == 16) GOW(PW_AFFINE | PW_POOL, 16);
    if (base == (PW_STORE | PW_STATS | PW_POOL | PW_POOLMIN) && pg == 16)
      GOW(PW_AFFINE | PW_STORE | PW_STATS | PW_POOL | PW_POOLMIN, 16);
    if (base == (PW_STATS | PW_POOL | PW_POOLMIN) && pg == 16) GOW(PW_AFFINE | PW_STATS | PW_POOL | PW_POOLMIN, 16);
    if (base == (PW_STATS | PW_POOL | PW_POOLMIN) && pg == 32) GOW(PW_AFFINE | PW_STATS | PW_POOL | PW_POOLMIN, 32);
    if (base == (PW_STORE | PW_POOL) && pg == 32) GOW(PW_AFFINE | PW_STORE | PW_POOL, 32);
    if (base == PW_POOL && pg == 32) GOW(PW_AFFINE | PW_POOL, 32);
    if (base == (PW_STORE | PW_STATS | PW_POOL | PW_POOLMIN) && pg == 32)
      GOW(PW_AFFINE | PW_STORE | PW_STATS | PW_POOL | PW_POOLMIN, 32);
  } else {
    if (base == PW_STORE) GO(PW_STORE, 16);
    if (base == (PW_STORE | PW_BIAS)) GO(PW_STORE | PW_BIAS, 16);
    if (base == (PW_STORE | PW_BNRED)) GO(PW_STORE | PW_BNRED, 16);
    if (base == (PW_STORE | PW_STATS)) GO(PW_STORE | PW_STATS, 16);
    if (base == (PW_STORE | PW_STATS | PW_ROWBIAS)) GOW(PW_STORE | PW_STATS | PW_ROWBIAS, 16);
  }
  set_error("pw_layer_forward: epilogue combination 0x%x (pool group %d) is not built", epi, pg);
  return NESIE_ERR_UNSUPPORTED;
#endif
#undef GO
#undef GOW
#undef GO1
}

// the one prologue / epilogue combination of a pooled tail's input gradient (built operand rows)
template <int KT16, int KH, int WR, int WC, int RW, int SPBIT>
static int pw_launch_sparse(const PwFwd &a, int grid, size_t lds, hipStream_t s) {
  auto kern = pw_fwd_kernel<KT16, KH, WR, WC, RW, PW_STORE | PW_BIAS | PW_AFFINE | PW_BNRED | SPBIT, 16, false>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WR * WC * 64), lds, s, a);
  return NESIE_OK;
}

#define PW_GEOM_NAME(KT16, KH, WR, WC, RW) pw_launch_##KT16##_##KH##_##WR##_##WC##_##RW
#define PW_GEOM_DECL(KT16, KH, WR, WC, RW) \
  int PW_GEOM_NAME(KT16, KH, WR, WC, RW)(const PwFwd &a, int epi, int pg, int grid, size_t lds, hipStream_t s);
#define PW_GEOM_DEF(KT16, KH, WR, WC, RW)                                                        \
  namespace nesie {                                                                             \
  int PW_GEOM_NAME(KT16, KH, WR, WC, RW)(const PwFwd &a, int epi, int pg, int grid, size_t lds, \
                                         hipStream_t s) {                                       \
    return pw_launch_epi<KT16, KH, WR, WC, RW>(a, epi, pg, grid, lds, s);                       \
  }                                                                                             \
  }

}  // namespace nesie
