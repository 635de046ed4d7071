// Weight-gradient kernels of the pointwise-convolution layers (see pwconv_fwd.h for the layer
// kernel and pwconv.hip for the host side of the forward family).
#include "pwconv_fwd.h"

using namespace nesie;

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

// BNB: dY is not given but formed on load from the gradient `dy` = dA of relu(bn(Z)) and the raw
// conv output Z (`bnz`, same layout) with the finished reduction coefficients bnb[g*co + r] =
// (scale, shift of the forward's fused multiply-add: mask = fma(z, scale, shift) > 0; a, mean, d1, e0):
//     dZ = a g + (e0 - (z - mean) d1),  g = mask ? dA : 0,  d1 = a invstd mean(g zhat),  e0 = -a mean(g)
// (= gamma invstd (g - mean(g) - zhat mean(g zhat)), the BatchNorm + ReLU backward's apply pass,
// nesie_bn_relu_backward_apply) -- used as the MFMA operand AND written to `dz` for the
// input-gradient launch that follows, so the separate apply pass over (dA, Z) -> dZ disappears.
// dz may be dy itself: a tile's words are read once, by the workgroup that owns the tile, before
// it writes them.  d_rb (optional, BNB): gradient of a row bias that was added to Z per group of
// rb_group (16 or 64) positions, d_rb[n][r][pos / rb_group] = sum of dZ over the group -- the 8
// lanes that hold a row's 32 positions add their words with shuffles; a 64-group spans two tiles
// (of different workgroups): two float atomics into a zero-filled buffer (two addends: the order
// cannot matter).
// TILED launches (gridDim.y x gridDim.z > 1; plain weight gradients only): the product is cut
// into (CO16 * 16) x (CI16 * 16) blocks, blockIdx.y = row block, blockIdx.z = column block, every
// block with its own runs of positions and its own partials -- for layers whose co x ci is large
// and whose position count is small (the 1-D chains: 256 x 512 over 8 x 1024 positions), where one
// whole-product workgroup per 32 positions wrote 128 KB of partial per 32 positions.
// K4 (one build: 64 x 64 with AFF and BNB): the 64 rows of X are the raw output of a 4 -> 64
// convolution, Z0 = W0 . X4, and are REBUILT from the four rows of x (nb, 4, p) on the operand load
// (k4_dot, pwconv_fwd.h: the same roundings as the forward's) -- SA1's first activation is never stored.
// FDG (with K4): the layer's INPUT gradient is formed from the same dZ tile in the same launch,
// dA0 = W^T dZ (16 more MFMAs per wave and tile), and only its reductions leave the kernel -- sum(g),
// sum(g zhat), sum(g X4[j]) with g = dA0 [bn(Z0) > 0] per channel and slot (what
// nesie_pw_dgrad_bn_reduce_k4 leaves): dZ is then needed by nobody and is NOT written.
template <int CO16, int CI16, int WM, int WN, bool AFF, bool BNB, bool K4 = false, bool FDG = false>
__global__ __launch_bounds__(512, FDG ? 4 : 1) void pw_wgrad_kernel(      // (FDG: two workgroups per CU = 128 VGPRs)
    int nb, int ng, int co_all, int ci_all, long long p, const float *__restrict__ dy, long long dy_bs,
    const float *__restrict__ x, long long x_bs, const float *__restrict__ x_coef, int coef_gs,
    float x_lo, float *__restrict__ partial, int nwg_g, const float *__restrict__ bnz,
    const float *__restrict__ bnb, float *dz, float *__restrict__ d_rb, int rb_group, int rev,
    const float *__restrict__ k4_w, const float *__restrict__ fdg_w = nullptr, float *__restrict__ fdg_part = nullptr,
    float *__restrict__ fdg_gpart = nullptr) {
  const int row0 = blockIdx.y * CO16 * 16, col0 = blockIdx.z * CI16 * 16;   // (0, 0) unless tiled
  const int co = co_all - row0 < CO16 * 16 ? co_all - row0 : CO16 * 16;     // this block's extent
  const int ci = ci_all - col0 < CI16 * 16 ? ci_all - col0 : CI16 * 16;
  dy += (size_t)row0 * p;
  x += (size_t)col0 * p;
  if (AFF) x_coef += (size_t)col0 * 4;
  partial += (size_t)(blockIdx.y * gridDim.z + blockIdx.z) * ng * nwg_g * (CO16 * 16) * (CI16 * 16);
  constexpr int MB = CO16 / WM, NB = CI16 / WN, PT = 32, PITCH = PT + 4, CPR = PT / 4;
  constexpr int ROWS = (CO16 + CI16) * 16, NT = 512;
  constexpr int NX = (ROWS * CPR + NT - 1) / NT;
  constexpr bool EVEN = ROWS * CPR == NX * NT;
  constexpr int DYSLOTS = CO16 * 16 * CPR / NT;          // slots that hold dY rows (CO16 % 4 == 0)
  static_assert(WM * WN == 8 && CO16 % WM == 0 && CI16 % WN == 0 && (CO16 * 16 * CPR) % NT == 0, "tiling");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int TILE = ROWS * PITCH + (FDG ? 4 * PITCH : 0);     // (FDG: + the tile's four X4 rows)
  static_assert(!FDG || (K4 && BNB && CO16 == 4 && CI16 == 4), "FDG: the 64 x 64 layer over a rebuilt operand");
  const int tid = threadIdx.x, lane = tid & 63, quad = lane >> 4, l16 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int g = blockIdx.x % ng, rank = blockIdx.x / ng;

  static_assert(!K4 || (AFF && NX - DYSLOTS == 1 && CI16 == 4 && EVEN), "K4: one slot of X rows");
  float4 k4w = make_float4(0.f, 0.f, 0.f, 0.f);      // K4: W0 row of this thread's X row
  unsigned k4off = 0;                                  // ... and its 16-byte column
  bool k4row0 = false;                                 // ... which is row 0's (FDG: it stores the X4 rows)
  f32x4 k4x[K4 ? 3 : 1];                               // ... rows 1 .. 3 of X4 (row 0 sits in the slot's own register)
  unsigned goff[NX], lw[NX];
  bool okslot[NX];
  float sc[AFF ? NX : 1], bi[AFF ? NX : 1];
  float zsc[BNB ? DYSLOTS : 1], zbi[BNB ? DYSLOTS : 1], za[BNB ? DYSLOTS : 1], zmu[BNB ? DYSLOTS : 1], zd1[BNB ? DYSLOTS : 1], ze0[BNB ? DYSLOTS : 1];
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
    if constexpr (K4) {
      if (!isdy) {
        k4w = *(const float4 *)(k4_w + (size_t)r * 4);
        k4off = (unsigned)(cp * 16);
        k4row0 = r == 0;
        asm volatile("" : "+v"(k4w.x), "+v"(k4w.y), "+v"(k4w.z), "+v"(k4w.w));
      }
    }
    if (AFF) {
      const float s0 = (ok && !isdy) ? x_coef[((size_t)g * coef_gs + r) * 4] : 0.f;
      const float b0 = (ok && !isdy) ? x_coef[((size_t)g * coef_gs + r) * 4 + 1] : 0.f;
      sc[i] = s0;
      bi[i] = b0;
    }
    if constexpr (BNB) {
      if (i < DYSLOTS) {
        const float *cf = bnb + ((size_t)g * co + (ok ? r : 0)) * 8;
        zsc[i] = cf[0]; zbi[i] = cf[1]; za[i] = cf[2]; zmu[i] = cf[3]; zd1[i] = cf[4]; ze0[i] = cf[5];
        asm volatile("" : "+v"(zsc[i]), "+v"(zbi[i]), "+v"(za[i]), "+v"(zmu[i]), "+v"(zd1[i]), "+v"(ze0[i]));
      }
    }
  }
  if (AFF) {
#pragma unroll
    for (int i = 0; i < NX; ++i) asm volatile("" : "+v"(sc[i]), "+v"(bi[i]));
  }
  const bool all_rows = __builtin_amdgcn_readfirstlane((co == CO16 * 16 && ci == CI16 * 16) ? 1 : 0) != 0;
  // tiles (batch of the group, 32 positions) in 32-bit arithmetic (the host checks the count): a
  // 64-bit division per tile was 130 instructions of the loop
  const int tpb = (int)(p / PT);
  const int ntiles = (nb / ng) * tpb;

  // Register staging, ONE set: the loads of tile t + 2 are issued right behind the LDS write of
  // tile t + 1 at the bottom of tile t and are waited for at the bottom of tile t + 1 -- a full
  // tile time (8 000 - 11 000 cycles: the SIMD's two waves take turns on the matrix pipe) later.
  // (Issued at the TOP of the tile they were waited for at its bottom, i.e. after only the wave's
  // own 4 096 MFMA cycles for the wave that gets the pipe first: it sat out HBM latency in every
  // tile, 14 000 cycles per tile for 8 192 of MFMA.)  Loads are unconditional (past the end: the
  // last tile again); nothing younger than the awaited loads is in flight at the wait.
  f32x4 stg[NX], stz[BNB ? DYSLOTS : 1];
  size_t pend = 0;           // (batch, position) word offset of the tile in the staging registers
  bool pend_ok = false;      // ... and whether it is a tile of this workgroup (not the repeat past the end)
  int pend_n = 0, pend_p0 = 0;   // ... its batch and first position
  auto load_tile = [&](int t) {
    pend_ok = t < ntiles;
    t = t < ntiles ? t : ntiles - 1;
    t = rev ? ntiles - 1 - t : t;          // (big operands: last tile first, see nesie_lib.hip)
    const int n = g + ng * (t / tpb);
    const long long p0 = (long long)(t % tpb) * PT;
    pend = (size_t)n * dy_bs + p0;
    pend_n = n; pend_p0 = (int)p0;
    const float *dyb = dy + pend, *xb = x + (size_t)n * x_bs + p0;   // uniform
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      // (opaque copy: the zero-extension of the offset must stay in this block, or instruction
      // selection sees a hoisted 64-bit register pair and drops the scalar-base addressing form)
      unsigned o = goff[i];
      asm volatile("" : "+v"(o));
      if constexpr (K4) {
        if (i >= DYSLOTS) {
          unsigned o4 = k4off;
          asm volatile("" : "+v"(o4));
          stg[i] = load16_saddr(o4, xb);
#pragma unroll
          for (int j = 1; j < 4; ++j) k4x[j - 1] = load16_saddr(o4, xb + (size_t)j * p);
          continue;
        }
      }
      stg[i] = load16_saddr(o, i < DYSLOTS ? dyb : xb);
      if constexpr (BNB) {
        if (i < DYSLOTS) stz[i] = load16_saddr(o, bnz + pend);
      }
    }
  };
  auto write_tile = [&](float *buf) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      f32x4 q = stg[i];
      if constexpr (BNB) {
        if (i < DYSLOTS) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float zz = stz[i][e];
            const float gg = __builtin_fmaf(zz, zsc[i], zbi[i]) > 0.f ? q[e] : 0.f;
            q[e] = __builtin_fmaf(za[i], gg, __builtin_fmaf(zmu[i] - zz, zd1[i], ze0[i]));
          }
          if constexpr (!FDG) {
            if (pend_ok && okslot[i]) *(f32x4 *)((char *)(dz + pend) + goff[i]) = q;
          }
          if (d_rb) {      // (wave-uniform)
            float tsum = (q[0] + q[1]) + (q[2] + q[3]);
            tsum += __shfl_xor(tsum, 1, 64);
            tsum += __shfl_xor(tsum, 2, 64);
            if (rb_group == 64) tsum += __shfl_xor(tsum, 4, 64);
            const int c_ = i * NT + tid, row = c_ / CPR, cp = c_ % CPR;
            const bool writer = rb_group == 64 ? cp == 0 : (cp & 3) == 0;
            if (pend_ok && okslot[i] && writer) {
              const int gsh = rb_group == 64 ? 6 : 4;
              float *dst = d_rb + ((size_t)pend_n * co + row) * (size_t)(p >> gsh) + ((pend_p0 + 4 * cp) >> gsh);
              if (rb_group == 64) atomicAdd(dst, tsum); else *dst = tsum;
            }
          }
        }
      }
      if constexpr (K4) {
        if (i >= DYSLOTS) {
          if constexpr (FDG) {
            // the thread of X row 0 also leaves the four X4 rows of its 16-byte column behind the tile
            if (k4row0) {
              float *xt = buf + ROWS * PITCH + (k4off >> 2);
              *(f32x4 *)(xt) = q;
              *(f32x4 *)(xt + PITCH) = k4x[0];
              *(f32x4 *)(xt + 2 * PITCH) = k4x[1];
              *(f32x4 *)(xt + 3 * PITCH) = k4x[2];
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e] = k4_dot(k4w, q[e], k4x[0][e], k4x[1][e], k4x[2][e]);
        }
      }
      if (AFF && i >= DYSLOTS) {
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = fmaxf(__builtin_fmaf(q[e], sc[i], bi[i]), x_lo);
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
  // FDG: wave = (16-channel block cb of the input gradient, 16-position half ph of the tile); a lane
  // owns channel 16 cb + l16 at positions 16 ph + 4 quad .. + 3
  const int cb = wave & 3, ph = wave >> 2;
  float fw[FDG ? 16 : 1];                  // W[m = 4 kk + quad][c]: the B operand, for the whole launch
  float4 fw0 = make_float4(0.f, 0.f, 0.f, 0.f), fco = fw0;   // W0 row and (scale, shift, mean, invstd) of the channel
  float fr0 = 0.f, fr1 = 0.f, fgx[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (FDG) {
    const int c = 16 * cb + l16;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) fw[kk] = fdg_w[(size_t)(4 * kk + quad) * 64 + c];
    fw0 = *(const float4 *)(k4_w + (size_t)c * 4);
    fco = *(const float4 *)(x_coef + (size_t)c * 4);
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) asm volatile("" : "+v"(fw[kk]));
    asm volatile("" : "+v"(fw0.x), "+v"(fw0.y), "+v"(fw0.z), "+v"(fw0.w), "+v"(fco.x), "+v"(fco.y), "+v"(fco.z), "+v"(fco.w));
  }

  float *b0 = lds, *b1 = lds + TILE;
  int t = rank;
  if (t < ntiles) {
    load_tile(t);
    write_tile(b0);
    load_tile(t + nwg_g);
  }
  for (; t < ntiles; t += nwg_g) {
    lgkm_wait<0>();
    __builtin_amdgcn_s_barrier();
    // lane (l16 = row inside its block, quad): positions 16 pg + 4 quad .. + 3
    const unsigned la = lds_addr(b0) + (unsigned)(((wm * MB * 16 + l16) * PITCH + 4 * quad) * 4);
    const unsigned lb = lds_addr(b0) + (unsigned)(((CO16 * 16 + wn * NB * 16 + l16) * PITCH + 4 * quad) * 4);
    // ONE set of fragment registers (the second position group is read behind the MFMAs of the
    // first: its LDS latency is covered by the SIMD's other wave)
    f32x4 fa[MB], fb[NB];
    static_for<0, 2>([&](auto pgc) {
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
    if constexpr (FDG) {
      // dA0[pos][c] = sum_m dZ[m][pos] W[m][c] over the tile in b0: A = dZ^T straight from the LDS rows
      f32x4 da = {0.f, 0.f, 0.f, 0.f};
      const float *az = b0 + quad * PITCH + 16 * ph + l16;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk)
        da = __builtin_amdgcn_mfma_f32_16x16x4f32(az[kk * 4 * PITCH], fw[kk], da, 0, 0, 0);
      const float *xt = b0 + ROWS * PITCH + 16 * ph + 4 * quad;
      const f32x4 x0 = *(const f32x4 *)xt, x1 = *(const f32x4 *)(xt + PITCH), x2 = *(const f32x4 *)(xt + 2 * PITCH),
                  x3 = *(const f32x4 *)(xt + 3 * PITCH);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float z = k4_dot(fw0, x0[r], x1[r], x2[r], x3[r]);
        const float gg = __builtin_fmaf(z, fco.x, fco.y) > 0.f ? da[r] : 0.f;
        fr0 += gg;
        fr1 += gg * ((z - fco.z) * fco.w);
        fgx[0] = __builtin_fmaf(gg, x0[r], fgx[0]);
        fgx[1] = __builtin_fmaf(gg, x1[r], fgx[1]);
        fgx[2] = __builtin_fmaf(gg, x2[r], fgx[2]);
        fgx[3] = __builtin_fmaf(gg, x3[r], fgx[3]);
      }
    }
    write_tile(b1);          // tile t + 1 (after the last tile: a copy of it that nobody reads)
    __builtin_amdgcn_sched_barrier(0);
    load_tile(t + 2 * nwg_g);
    float *const tb = b0; b0 = b1; b1 = tb;
  }
  if constexpr (FDG) {
    // per channel: the four quads of this wave, then one slot per (workgroup, position half)
    float v[6] = {fr0, fr1, fgx[0], fgx[1], fgx[2], fgx[3]};
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      v[u] += __shfl_xor(v[u], 16, 64);
      v[u] += __shfl_xor(v[u], 32, 64);
    }
    if (quad == 0) {
      const int c = 16 * cb + l16, slot = 2 * rank + ph, nslots = 2 * nwg_g;
      *(float2 *)(fdg_part + ((size_t)c * nslots + slot) * 2) = make_float2(v[0], v[1]);
      *(float4 *)(fdg_gpart + ((size_t)c * nslots + slot) * 4) = make_float4(v[2], v[3], v[4], v[5]);
    }
  }
  // partial[(g * nwg + rank)][co][ci]: lane (quad, l16) holds rows 4 quad + r, column l16
  // (tiled: [block][g * nwg + rank][co][ci] with each block's own co x ci, stride the full block)
  const bool tiled = gridDim.y * gridDim.z > 1;
  float *dst = partial + ((size_t)g * nwg_g + rank) * (tiled ? (size_t)(CO16 * 16) * (CI16 * 16) : (size_t)co * ci);
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
// (a launch covers the ci columns [col0, col0 + ci) of dw (ng, co, ld): wide layers run as column blocks)
__global__ __launch_bounds__(1024) void pw_wgrad_reduce_kernel(int total, int nparts,
                                                               const float *__restrict__ partial,
                                                               float *__restrict__ dw, int ci, int ld,
                                                               int col0, int co) {
  // (partials added in DOUBLE since round 5: 512 fp32 addends per element cost ~1e-6 of the sum of their
  // magnitudes, which for the backbone's heavily cancelling weight gradients was a visible share of the
  // distance to a float64 evaluation; the kernel is bandwidth-bound either way)
  __shared__ double sh[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane, g = blockIdx.y;
  const float *src = partial + (size_t)g * nparts * total;
  double s = 0.0;
  if (i < total) {
    int r = wave;
    for (; r + 15 * 16 < nparts; r += 16 * 16) {     // sixteen partials in flight (two trips at 512 partials)
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = src[(size_t)(r + u * 16) * total + i];
#pragma unroll
      for (int u = 0; u < 16; ++u) s += (double)v[u];
    }
    for (; r + 7 * 16 < nparts; r += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(r + u * 16) * total + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; r < nparts; r += 16) s += (double)src[(size_t)r * total + i];
  }
  sh[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && i < total) {
    double tt = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) tt += sh[w][lane];
    dw[((size_t)g * co + i / ci) * ld + col0 + i % ci] = (float)tt;
  }
}

// tiled launches: dw[g][rb * BR + m][cb * BC + k] = sum over the nparts partials of (block, g), fixed order;
// partial[block][g * nparts + r][co_b][ci_b] with co_b x ci_b that block's own extent
__global__ __launch_bounds__(256) void pw_wgrad_reduce_tiled_kernel(int co, int ci, int br, int bc, int nparts,
                                                                    const float *__restrict__ partial,
                                                                    float *__restrict__ dw) {
  const int ncb = (ci + bc - 1) / bc;
  const int blk = blockIdx.y, g = blockIdx.z, rb = blk / ncb, cb = blk % ncb;
  const int co_b = co - rb * br < br ? co - rb * br : br, ci_b = ci - cb * bc < bc ? ci - cb * bc : bc;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= co_b * ci_b) return;
  const float *src = partial + ((size_t)blk * gridDim.z + g) * nparts * br * bc + i;
  double s = 0.0;
  int r = 0;
  for (; r + 3 < nparts; r += 4) {
    const float v0 = src[(size_t)r * br * bc], v1 = src[(size_t)(r + 1) * br * bc],
                v2 = src[(size_t)(r + 2) * br * bc], v3 = src[(size_t)(r + 3) * br * bc];
    s += (double)v0; s += (double)v1; s += (double)v2; s += (double)v3;
  }
  for (; r < nparts; ++r) s += (double)src[(size_t)r * br * bc];
  dw[((size_t)g * co + rb * br + i / ci_b) * ci + cb * bc + i % ci_b] = (float)s;
}

// ---- deferred reductions (round 5) ------------------------------------------------------------
// The partials of a weight gradient are consumed by nobody but the optimiser (and the gradient
// all-reduce): nothing in the backward pass reads dW.  So a launch may leave its reduction PENDING
// (nesie_pw_wgrad_deferred / nesie_pw_wgrad_bn_backward_deferred: the workspace then has to stay
// alive) and nesie_pw_wgrad_flush_deferred adds the partials of every pending gradient in ONE launch
// -- a descriptor table in the kernel arguments, a workgroup finds its descriptor from the block
// index -- with each element summed exactly as its own reduce kernel sums it (bit-identical).  39
// reduce launches of 5 - 8 us per step become one or two.
struct RDesc {
  const float *part; float *dw;
  int kind;                 // 0 plain (pw_wgrad_reduce_kernel), 1 tiled (pw_wgrad_reduce_tiled_kernel)
  int nparts, total, ci, ld, col0, co, ng;
  int br, bc, nblk;         // tiled: block extents and number of blocks
};
constexpr int RD_MAX = 40;
struct RTable { int n; int start[RD_MAX + 1]; RDesc d[RD_MAX]; };

__global__ __launch_bounds__(1024) void pw_wgrad_reduce_batch_kernel(const RTable t) {
  __shared__ double sh[16][64];
  int di = 0;
  while (di + 1 < t.n && (int)blockIdx.x >= t.start[di + 1]) ++di;        // (uniform)
  const RDesc &d = t.d[di];
  const int local = blockIdx.x - t.start[di];
  if (d.kind == 0) {
    const int bpg = (d.total + 63) / 64;
    const int bx = local % bpg, g = local / bpg;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = bx * 64 + lane;
    const float *src = d.part + (size_t)g * d.nparts * d.total;
    const int total = d.total, nparts = d.nparts;
    double s = 0.0;
    if (i < total) {
      int r = wave;
      for (; r + 15 * 16 < nparts; r += 16 * 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = src[(size_t)(r + u * 16) * total + i];
        s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        s += ((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15]));
      }
      for (; r + 7 * 16 < nparts; r += 8 * 16) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(r + u * 16) * total + i];
        s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
      }
      for (; r < nparts; r += 16) s += (double)src[(size_t)r * total + i];
    }
    sh[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && i < total) {
      double tt = 0.0;
#pragma unroll
      for (int w = 0; w < 16; ++w) tt += sh[w][lane];
      d.dw[((size_t)g * d.co + i / d.ci) * d.ld + d.col0 + i % d.ci] = (float)tt;
    }
  } else {
    // one thread per element of a block, the partials in ascending order (four at a time)
    const int per = (d.br * d.bc + 1023) / 1024;
    const int q = local % per, blk = (local / per) % d.nblk, g = local / (per * d.nblk);
    const int ncb = (d.ci + d.bc - 1) / d.bc;
    const int rb = blk / ncb, cb = blk % ncb;
    const int co_b = d.co - rb * d.br < d.br ? d.co - rb * d.br : d.br, ci_b = d.ci - cb * d.bc < d.bc ? d.ci - cb * d.bc : d.bc;
    const int i = q * 1024 + threadIdx.x;
    if (i >= co_b * ci_b) return;
    const size_t bb = (size_t)d.br * d.bc;
    const float *src = d.part + ((size_t)blk * d.ng + g) * d.nparts * bb + i;
    double s = 0.0;
    int r = 0;
    for (; r + 3 < d.nparts; r += 4) {
      const float v0 = src[(size_t)r * bb], v1 = src[(size_t)(r + 1) * bb], v2 = src[(size_t)(r + 2) * bb],
                  v3 = src[(size_t)(r + 3) * bb];
      s += (double)v0; s += (double)v1; s += (double)v2; s += (double)v3;
    }
    for (; r < d.nparts; ++r) s += (double)src[(size_t)r * bb];
    d.dw[((size_t)g * d.co + rb * d.br + i / ci_b) * d.ci + cb * d.bc + i % ci_b] = (float)s;
  }
}

static RTable g_pending;      // host side: the reductions waiting for nesie_pw_wgrad_flush_deferred

static void rd_push(const RDesc &d) {
  const int blocks = d.kind == 0 ? ((d.total + 63) / 64) * d.ng : ((d.br * d.bc + 1023) / 1024) * d.nblk * d.ng;
  const int n = g_pending.n;
  g_pending.d[n] = d;
  if (n == 0) g_pending.start[0] = 0;
  g_pending.start[n + 1] = g_pending.start[n] + blocks;
  g_pending.n = n + 1;
}

static int rd_flush(hipStream_t s) {
  if (g_pending.n == 0) return NESIE_OK;
  hipLaunchKernelGGL(pw_wgrad_reduce_batch_kernel, dim3(g_pending.start[g_pending.n]), dim3(1024), 0, s, g_pending);
  g_pending.n = 0;
  return check_launch("pw_wgrad_flush_deferred");
}

// workgroups per weight group: one per CU; TWO per CU where a block's tiles (<= 74 KB of LDS) and
// registers (<= 128) allow it (NESIE_WGRAD_PER_CU=1: A/B switch)
static int pw_wgrad_nwg(int nb, int ng, long long p, int co, int cw, bool bnb = false) {
  static const int per_cu_max = getenv("NESIE_WGRAD_PER_CU") ? atoi(getenv("NESIE_WGRAD_PER_CU")) : 2;
  // (the fused norm-backward variants need 146-150 VGPRs at 128 x 128: one workgroup per CU there)
  const int per_cu = (co <= 128 && cw <= (bnb ? 64 : 128) && per_cu_max >= 2) ? 2 : 1;
  static const int rounds = [] { const char *e = getenv("NESIE_WGRAD_ROUNDS"); return e ? atoi(e) : 1; }();   // A/B
  long long nwg = (long long)cu_count() * per_cu * (rounds > 0 ? rounds : 1) / ng;
  const long long tiles = (long long)(nb / ng) * (p / 32);
  if (nwg > tiles) nwg = tiles;
  return nwg < 1 ? 1 : (int)nwg;
}

}  // namespace nesie

// widest column block one launch covers (the whole co x block product sits in one workgroup's
// accumulators): wider layers run as several column blocks, each re-reading dY
static int pw_wgrad_block(int co, int ci) {
  if (co <= 128) return ci <= 320 ? ci : 256;
  return ci <= 128 ? ci : 128;
}

extern "C" int nesie_pw_wgrad_supported(int co, int ci, long long p) {
  return p % 32 == 0 && ci >= 9 && co <= 256 && ci <= 1024 ? 1 : 0;
}

// Tiled mode (pw_wgrad_kernel): 64 x 64 blocks of the product, each with its own position runs.
// Chosen for wide products over few positions: at most TILED_MAX_TILES 32-position tiles per
// weight group and more than one whole-product column block's worth of output.
constexpr int TILED_B = 64, TILED_MAX_TILES = 1024;
static bool pw_wgrad_tiled(int nb, int ng, int co, int ci, long long p) {
  static const int on = getenv("NESIE_WGRAD_TILED") ? atoi(getenv("NESIE_WGRAD_TILED")) : 1;   // A/B switch
  if (!on || nb <= 0 || ng <= 0 || nb % ng || p <= 0 || p % 32 || ci < 9 || ci > 1024 || co < 1 || co > 1024) return false;
  const long long tiles = (long long)(nb / ng) * (p / 32);
  return tiles <= TILED_MAX_TILES && (long long)co * ci >= 128 * 192;
}
// position runs per block and weight group: fill the chip about twice over all blocks
static int pw_wgrad_tiled_nwg(int nb, int ng, int co, int ci, long long p) {
  const long long tiles = (long long)(nb / ng) * (p / 32);
  const int blocks = cdiv(co, TILED_B) * cdiv(ci, TILED_B);
  long long nwg = 512 / ((long long)blocks * ng);
  if (nwg < 1) nwg = 1;
  if (nwg > tiles) nwg = tiles;
  return (int)nwg;
}

extern "C" int nesie_pw_wgrad_tiled(int nb, int ng, int co, int ci, long long p) {
  return pw_wgrad_tiled(nb, ng, co, ci, p) ? 1 : 0;     // (also serves co up to 1024, beyond nesie_pw_wgrad_supported)
}

extern "C" size_t nesie_pw_wgrad_workspace_bytes(int nb, int ng, int co, int ci, long long p) {
  if (nb <= 0 || ng <= 0 || p <= 0) return 0;
  const size_t whole = (size_t)ng * pw_wgrad_nwg(nb, ng, p, co, pw_wgrad_block(co, ci)) * co * pw_wgrad_block(co, ci) * sizeof(float);
  const size_t tiled = (size_t)cdiv(co, TILED_B) * cdiv(ci, TILED_B) * ng * pw_wgrad_tiled_nwg(nb, ng, co, ci, p) *
                       TILED_B * TILED_B * sizeof(float);
  return whole > tiled ? whole : tiled;     // (either mode may serve the shape)
}

namespace nesie {
// bnb[c] = (scale, shift, a, mean, d1, e0, -, -) of channel c = g * co + r from the reduction
// partials the input-gradient launch left (partial[(c * nslots + i) * 2 + {0, 1}] = sum g, sum g zhat),
// the layer's folded forward coefficients z_coef[c] = (scale, shift, mean, invstd) and gamma;
// dgamma / dbeta written on the way (fp64 sums, as bn_bwd_finalize in bn.hip)
__global__ __launch_bounds__(64) void pw_bnb_coef_kernel(int channels, int nslots, double count,
                                                         const float *__restrict__ part,
                                                         const float *__restrict__ z_coef,
                                                         const float *__restrict__ gamma,
                                                         float *__restrict__ bnb, float *__restrict__ dgamma,
                                                         float *__restrict__ dbeta) {
  const int c = blockIdx.x;
  double s0 = 0.0, s1 = 0.0;
  const float2 *pc = (const float2 *)part + (size_t)c * nslots;
  int i = threadIdx.x;
  for (; i + 192 < nslots; i += 256) {           // four loads in flight, added in slot order
    float2 q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) q[u] = pc[i + 64 * u];
#pragma unroll
    for (int u = 0; u < 4; ++u) { s0 += (double)q[u].x; s1 += (double)q[u].y; }
  }
  for (; i < nslots; i += 64) {
    const float2 q = pc[i];
    s0 += (double)q.x;
    s1 += (double)q.y;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s0 += __shfl_xor(s0, off, 64);
    s1 += __shfl_xor(s1, off, 64);
  }
  if (threadIdx.x == 0) {
    const float invstd = z_coef[c * 4 + 3];
    const float a = (float)((gamma ? (double)gamma[c] : 1.0) * (double)invstd);
    const float k1 = (float)(s0 / count), k2 = (float)(s1 / count);
    float *o = bnb + (size_t)c * 8;
    o[0] = z_coef[c * 4 + 0]; o[1] = z_coef[c * 4 + 1]; o[2] = a; o[3] = z_coef[c * 4 + 2];
    o[4] = a * invstd * k2; o[5] = -a * k1; o[6] = 0.f; o[7] = 0.f;
    if (dbeta) dbeta[c] = (float)s0;
    if (dgamma) dgamma[c] = (float)s1;
  }
}

static int pw_wgrad_launch(const char *W, int nb, int ng, int co, int ci, long long p, const float *dy,
                           long long dy_bstride, const float *x, long long x_bstride,
                           const float *x_coef, int x_relu, float *dw, void *workspace,
                           size_t workspace_bytes, const float *bnz, const float *bnb, float *dz,
                           float *d_rb, int rb_group, hipStream_t s, bool defer = false,
                           const float *k4_w = nullptr, const float *fdg_w = nullptr, float *fdg_part = nullptr,
                           float *fdg_gpart = nullptr) {
  NESIE_REQUIRE(nb >= 0 && ng >= 1 && co >= 1 && ci >= 1 && p >= 0 && dw, W);
  // (k4_w: x is X4 (nb, 4, p) and the 64 operand rows are rebuilt from it -- one build)
  NESIE_REQUIRE(!k4_w || (co == 64 && ci == 64 && ng == 1 && bnb && x_coef && !d_rb && x_bstride >= 4 * p &&
                          ((uintptr_t)k4_w & 15) == 0), W);
  if (nb == 0 || p == 0) {
    (void)hipMemsetAsync(dw, 0, (size_t)ng * co * ci * sizeof(float), s);
    return NESIE_OK;
  }
  const bool tiled_mode = !bnb && !d_rb && pw_wgrad_tiled(nb, ng, co, ci, p);
  if (!tiled_mode && !nesie_pw_wgrad_supported(co, ci, p)) {
    set_error("%s: %d x %d over %lld positions is outside the built tiles", W, co, ci, p);
    return NESIE_ERR_UNSUPPORTED;
  }
  NESIE_REQUIRE(nb % ng == 0 && dy && x && workspace, W);
  NESIE_REQUIRE(workspace_bytes >= nesie_pw_wgrad_workspace_bytes(nb, ng, co, ci, p), W);
  NESIE_REQUIRE((((uintptr_t)dy | (uintptr_t)x) & 15) == 0 && (dy_bstride & 3) == 0 && (x_bstride & 3) == 0, W);
  NESIE_REQUIRE((long long)(co > ci ? co : ci) * p < (1ll << 30), W);
  NESIE_REQUIRE((long long)(nb / ng) * (p / 32) < (1ll << 30), W);     // (32-bit tile cursor)
  const float lo0 = x_relu ? 0.f : -__builtin_inff();
  // dY is the tensor a launch has just written: a big one is read last tile first (nesie_lib.hip)
  static const int rev_on = [] { const char *e = getenv("NESIE_PW_REV_WGRAD"); return e ? atoi(e) : 1; }();   // A/B
  const int rev = rev_on ? walk_dir((long long)nb * co * p * 4) : 0;
  if (tiled_mode) {
    const int nwg = pw_wgrad_tiled_nwg(nb, ng, co, ci, p);
    const int nrb = cdiv(co, TILED_B), ncb = cdiv(ci, TILED_B);
    float *partial = (float *)workspace;
    const size_t lds = (size_t)2 * (4 + 4) * 16 * 36 * sizeof(float);
#define LT(AFF)                                                                                        \
    do {                                                                                               \
      auto kern = pw_wgrad_kernel<4, 4, 2, 4, AFF, false>;                                             \
      static bool attr = false;                                                                        \
      if (!attr) {                                                                                     \
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        attr = true;                                                                                   \
      }                                                                                                \
      hipLaunchKernelGGL(kern, dim3(nwg * ng, nrb, ncb), dim3(512), lds, s, nb, ng, co, ci, p, dy, dy_bstride, x, \
                         x_bstride, x_coef, ci, lo0, partial, nwg, (const float *)nullptr, (const float *)nullptr, \
                         (float *)nullptr, (float *)nullptr, 0, rev, (const float *)nullptr,            \
                         (const float *)nullptr, (float *)nullptr, (float *)nullptr);                   \
    } while (0)
    if (x_coef) LT(true); else LT(false);
#undef LT
    if (defer) {
      if (g_pending.n == RD_MAX) { const int st = rd_flush(s); if (st) return st; }
      RDesc d{};
      d.part = partial; d.dw = dw; d.kind = 1; d.nparts = nwg; d.ci = ci; d.co = co; d.ng = ng;
      d.br = TILED_B; d.bc = TILED_B; d.nblk = nrb * ncb;
      rd_push(d);
      return check_launch(W);
    }
    hipLaunchKernelGGL(pw_wgrad_reduce_tiled_kernel, dim3(cdiv(TILED_B * TILED_B, 256), nrb * ncb, ng), dim3(256), 0, s,
                       co, ci, TILED_B, TILED_B, nwg, partial, dw);
    return check_launch(W);
  }
  const int block = pw_wgrad_block(co, ci);
  // (the fused norm backward writes dZ while it forms it: one launch must own every column)
  NESIE_REQUIRE(!bnb || (bnz && dz && block == ci && (((uintptr_t)bnz | (uintptr_t)dz) & 15) == 0), W);
  NESIE_REQUIRE(!d_rb || (bnb && (rb_group == 16 || rb_group == 64) && p % rb_group == 0 &&
                          dy_bstride == (long long)co * p), W);
  const int nwg = pw_wgrad_nwg(nb, ng, p, co, block, bnb != nullptr);
  float *partial = (float *)workspace;
  const float lo = x_relu ? 0.f : -__builtin_inff();
#define LK(CO16, CI16, WM, WN, AFF, BNB)                                                         \
  do {                                                                                           \
    const size_t lds = (size_t)2 * (CO16 + CI16) * 16 * 36 * sizeof(float);                      \
    auto kern = pw_wgrad_kernel<CO16, CI16, WM, WN, AFF, BNB>;                                   \
    static bool attr = false;                                                                    \
    if (!attr) {                                                                                 \
      (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      attr = true;                                                                               \
    }                                                                                            \
    hipLaunchKernelGGL(kern, dim3(nwg * ng), dim3(512), lds, s, nb, ng, co, cw, p, dy,           \
                       dy_bstride, xc, x_bstride, cc, ci, lo, partial, nwg, bnz, bnb, dz, d_rb, rb_group, rev, \
                       (const float *)nullptr, (const float *)nullptr, (float *)nullptr, (float *)nullptr); \
  } while (0)
#define L(CO16, CI16, WM, WN)                                                                    \
  do {                                                                                           \
    if (bnb) { if (x_coef) LK(CO16, CI16, WM, WN, true, true); else LK(CO16, CI16, WM, WN, false, true); } \
    else { if (x_coef) LK(CO16, CI16, WM, WN, true, false); else LK(CO16, CI16, WM, WN, false, false); }   \
  } while (0)
  for (int c0 = 0; c0 < ci; c0 += block) {     // column blocks [c0, c0 + cw) of dw; same stream: the
    const int cw = ci - c0 < block ? ci - c0 : block;   // workspace is free again when the next one starts
    const float *xc = x + (size_t)c0 * p;
    const float *cc = x_coef ? x_coef + (size_t)c0 * 4 : nullptr;
    if (k4_w && fdg_w) {     // ... with the input gradient's reductions in the same launch (dz is not written)
      NESIE_REQUIRE(fdg_part && fdg_gpart && ((uintptr_t)fdg_gpart & 15) == 0 && ((uintptr_t)fdg_part & 7) == 0, W);
      const size_t lds = (size_t)2 * ((4 + 4) * 16 * 36 + 4 * 36) * sizeof(float);
      auto kern = pw_wgrad_kernel<4, 4, 2, 4, true, true, true, true>;
      static bool attr = false;
      if (!attr) {
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
      }
      hipLaunchKernelGGL(kern, dim3(nwg * ng), dim3(512), lds, s, nb, ng, co, cw, p, dy, dy_bstride, xc, x_bstride, cc,
                         ci, lo, partial, nwg, bnz, bnb, dz, d_rb, rb_group, rev, k4_w, fdg_w, fdg_part, fdg_gpart);
    } else if (k4_w) {
      const size_t lds = (size_t)2 * (4 + 4) * 16 * 36 * sizeof(float);
      auto kern = pw_wgrad_kernel<4, 4, 2, 4, true, true, true>;
      static bool attr = false;
      if (!attr) {
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
      }
      hipLaunchKernelGGL(kern, dim3(nwg * ng), dim3(512), lds, s, nb, ng, co, cw, p, dy, dy_bstride, xc, x_bstride, cc,
                         ci, lo, partial, nwg, bnz, bnb, dz, d_rb, rb_group, rev, k4_w, (const float *)nullptr,
                         (float *)nullptr, (float *)nullptr);
    } else if (co <= 64 && cw <= 64) L(4, 4, 2, 4);
    else if (co <= 128 && cw <= 64) L(8, 4, 4, 2);
    else if (co <= 128 && cw <= 128) L(8, 8, 2, 4);
    else if (co <= 128 && cw <= 192) L(8, 12, 2, 4);
    else if (co <= 128 && cw <= 256) L(8, 16, 2, 4);
    else if (co <= 128) L(8, 20, 2, 4);
    else L(16, 8, 4, 2);
    const int total = co * cw;
    if (defer && block >= ci) {      // (column blocks share the workspace: only a whole-product launch may wait)
      if (g_pending.n == RD_MAX) { const int st = rd_flush(s); if (st) return st; }
      RDesc d{};
      d.part = partial; d.dw = dw; d.kind = 0; d.nparts = nwg; d.total = total; d.ci = cw; d.ld = ci; d.col0 = c0;
      d.co = co; d.ng = ng;
      rd_push(d);
      continue;
    }
    hipLaunchKernelGGL(pw_wgrad_reduce_kernel, dim3(cdiv(total, 64), ng), dim3(1024), 0, s, total, nwg,
                       partial, dw, cw, ci, c0, co);
  }
#undef L
#undef LK
  return check_launch(W);
}
}  // namespace nesie

extern "C" int nesie_pw_wgrad(int nb, int ng, int co, int ci, long long p, const float *dy,
                              long long dy_bstride, const float *x, long long x_bstride,
                              const float *x_coef, int x_relu, float *dw, void *workspace,
                              size_t workspace_bytes, void *stream) {
  return pw_wgrad_launch("pw_wgrad", nb, ng, co, ci, p, dy, dy_bstride, x, x_bstride, x_coef, x_relu, dw,
                         workspace, workspace_bytes, nullptr, nullptr, nullptr, nullptr, 0, (hipStream_t)stream);
}

// The same launch with its reduction left pending (see "deferred reductions"): dw is written by
// nesie_pw_wgrad_flush_deferred; `workspace` must stay untouched until then.
extern "C" int nesie_pw_wgrad_deferred(int nb, int ng, int co, int ci, long long p, const float *dy,
                                       long long dy_bstride, const float *x, long long x_bstride,
                                       const float *x_coef, int x_relu, float *dw, void *workspace,
                                       size_t workspace_bytes, void *stream) {
  return pw_wgrad_launch("pw_wgrad_deferred", nb, ng, co, ci, p, dy, dy_bstride, x, x_bstride, x_coef, x_relu, dw,
                         workspace, workspace_bytes, nullptr, nullptr, nullptr, nullptr, 0, (hipStream_t)stream, true);
}

extern "C" int nesie_pw_wgrad_flush_deferred(void *stream) { return rd_flush((hipStream_t)stream); }
extern "C" int nesie_pw_wgrad_pending(void) { return g_pending.n; }
extern "C" int nesie_pw_wgrad_drop_deferred(void) {      // (error paths: forget what is pending)
  g_pending.n = 0;
  return NESIE_OK;
}

extern "C" int nesie_pw_wgrad_bn_supported(int co, int ci, long long p) {
  return nesie_pw_wgrad_supported(co, ci, p) && pw_wgrad_block(co, ci) == ci ? 1 : 0;
}

static int pw_wgrad_bn_backward_impl(const char *W, bool defer, int nb, int ng, int co, int ci, long long p, const float *da,
                                          const float *z, long long z_bstride, const float *z_coef,
                                          const float *gamma, const float *part, int nslots,
                                          const float *x, long long x_bstride, const float *x_coef,
                                          int x_relu, float *dz, float *dw, float *dgamma, float *dbeta,
                                          float *coef_ws, float *d_row_bias, int rb_group,
                                          void *workspace, size_t workspace_bytes, void *stream,
                                          const float *k4_w = nullptr, const float *fdg_w = nullptr,
                                          float *fdg_part = nullptr, float *fdg_gpart = nullptr) {
  NESIE_REQUIRE(nb >= 0 && ng >= 1 && co >= 1 && nslots >= 1, W);
  hipStream_t s = (hipStream_t)stream;
  if (nb == 0 || p == 0) {
    if (dgamma) (void)hipMemsetAsync(dgamma, 0, (size_t)ng * co * sizeof(float), s);
    if (dbeta) (void)hipMemsetAsync(dbeta, 0, (size_t)ng * co * sizeof(float), s);
    return pw_wgrad_launch(W, nb, ng, co, ci, p, da, z_bstride, x, x_bstride, x_coef, x_relu, dw, workspace,
                           workspace_bytes, nullptr, nullptr, nullptr, nullptr, 0, s, defer);
  }
  NESIE_REQUIRE(da && z && z_coef && part && dz && coef_ws && nb % ng == 0, W);
  hipLaunchKernelGGL(pw_bnb_coef_kernel, dim3(ng * co), dim3(64), 0, s, ng * co, nslots,
                     (double)(nb / ng) * (double)p, part, z_coef, gamma, coef_ws, dgamma, dbeta);
  return pw_wgrad_launch(W, nb, ng, co, ci, p, da, z_bstride, x, x_bstride, x_coef, x_relu, dw, workspace,
                         workspace_bytes, z, coef_ws, dz, d_row_bias, rb_group, s, defer, k4_w, fdg_w, fdg_part, fdg_gpart);
}

extern "C" int nesie_pw_wgrad_bn_backward(int nb, int ng, int co, int ci, long long p, const float *da,
                                          const float *z, long long z_bstride, const float *z_coef,
                                          const float *gamma, const float *part, int nslots,
                                          const float *x, long long x_bstride, const float *x_coef,
                                          int x_relu, float *dz, float *dw, float *dgamma, float *dbeta,
                                          float *coef_ws, float *d_row_bias, int rb_group,
                                          void *workspace, size_t workspace_bytes, void *stream) {
  return pw_wgrad_bn_backward_impl("pw_wgrad_bn_backward", false, nb, ng, co, ci, p, da, z, z_bstride, z_coef, gamma,
                                   part, nslots, x, x_bstride, x_coef, x_relu, dz, dw, dgamma, dbeta, coef_ws,
                                   d_row_bias, rb_group, workspace, workspace_bytes, stream);
}

// ... and with the weight gradient's reduction left pending (dz, dgamma, dbeta are complete on return)
extern "C" int nesie_pw_wgrad_bn_backward_deferred(int nb, int ng, int co, int ci, long long p, const float *da,
                                                   const float *z, long long z_bstride, const float *z_coef,
                                                   const float *gamma, const float *part, int nslots,
                                                   const float *x, long long x_bstride, const float *x_coef,
                                                   int x_relu, float *dz, float *dw, float *dgamma, float *dbeta,
                                                   float *coef_ws, float *d_row_bias, int rb_group,
                                                   void *workspace, size_t workspace_bytes, void *stream) {
  return pw_wgrad_bn_backward_impl("pw_wgrad_bn_backward_deferred", true, nb, ng, co, ci, p, da, z, z_bstride, z_coef,
                                   gamma, part, nslots, x, x_bstride, x_coef, x_relu, dz, dw, dgamma, dbeta, coef_ws,
                                   d_row_bias, rb_group, workspace, workspace_bytes, stream);
}


// ... of SA1's second layer (64 x 64), whose X is the REBUILT output of the first: x4 (nb, 4, p),
// w0 (64, 4), x_coef the first layer's folded norm.  defer != 0: the reduction stays pending.
extern "C" int nesie_pw_wgrad_bn_backward_k4(int nb, long long p, const float *da, const float *z,
                                             long long z_bstride, const float *z_coef, const float *gamma,
                                             const float *part, int nslots, const float *x4,
                                             long long x4_bstride, const float *w0, const float *x_coef,
                                             float *dz, float *dw, float *dgamma, float *dbeta, float *coef_ws,
                                             void *workspace, size_t workspace_bytes, int defer, void *stream) {
  const char *W = "pw_wgrad_bn_backward_k4";
  NESIE_REQUIRE(nb >= 1 && p >= 32 && x4 && w0 && x_coef, W);
  return pw_wgrad_bn_backward_impl(W, defer != 0, nb, 1, 64, 64, p, da, z, z_bstride, z_coef, gamma, part, nslots, x4,
                                   x4_bstride, x_coef, 1, dz, dw, dgamma, dbeta, coef_ws, nullptr, 0, workspace,
                                   workspace_bytes, stream, w0);
}

// ... and with the reductions of the layer's INPUT gradient from the same launch (pw_wgrad_kernel,
// FDG): w (64, 64) row-major the layer's weight; in_part [64][slots][2], in_gpart [64][slots][4] as
// nesie_pw_dgrad_bn_reduce_k4 leaves them, slots = nesie_pw_wgrad_bn_backward_k4_slots(nb, p).  The
// layer's dZ is consumed inside the launch and NOT written: da is left as it was.
extern "C" int nesie_pw_wgrad_bn_backward_k4_slots(int nb, long long p) {
  return 2 * pw_wgrad_nwg(nb, 1, p, 64, 64, true);
}
extern "C" int nesie_pw_wgrad_bn_backward_k4_fused(int nb, long long p, const float *da, const float *z,
                                                   long long z_bstride, const float *z_coef, const float *gamma,
                                                   const float *part, int nslots, const float *x4,
                                                   long long x4_bstride, const float *w0, const float *x_coef,
                                                   const float *w, float *dw, float *dgamma, float *dbeta,
                                                   float *coef_ws, float *in_part, float *in_gpart,
                                                   void *workspace, size_t workspace_bytes, int defer,
                                                   void *stream) {
  const char *W = "pw_wgrad_bn_backward_k4_fused";
  NESIE_REQUIRE(nb >= 1 && p >= 32 && x4 && w0 && x_coef && w && in_part && in_gpart, W);
  return pw_wgrad_bn_backward_impl(W, defer != 0, nb, 1, 64, 64, p, da, z, z_bstride, z_coef, gamma, part, nslots, x4,
                                   x4_bstride, x_coef, 1, const_cast<float *>(da), dw, dgamma, dbeta, coef_ws, nullptr, 0,
                                   workspace, workspace_bytes, stream, w0, w, in_part, in_gpart);
}

// The reduction coefficients of a BatchNorm + ReLU backward on their own (for a consumer of dZ
// other than the weight gradient: nesie_blend_conv_backward_bn): bnb [channels][8], dgamma, dbeta.
extern "C" int nesie_pw_bnb_coef(int channels, int nslots, double count, const float *part,
                                 const float *z_coef, const float *gamma, float *bnb, float *dgamma,
                                 float *dbeta, void *stream) {
  const char *W = "pw_bnb_coef";
  NESIE_REQUIRE(channels >= 0 && nslots >= 1 && count > 0, W);
  if (channels == 0) return NESIE_OK;
  NESIE_REQUIRE(part && z_coef && bnb, W);
  hipLaunchKernelGGL(pw_bnb_coef_kernel, dim3(channels), dim3(64), 0, (hipStream_t)stream, channels, nslots,
                     count, part, z_coef, gamma, bnb, dgamma, dbeta);
  return check_launch(W);
}
