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

template <int CO16, int CI16, int WM, int WN, bool AFF>
__global__ __launch_bounds__(512) void pw_wgrad_kernel(
    int nb, int ng, int co, int ci, long long p, const float *__restrict__ dy, long long dy_bs,
    const float *__restrict__ x, long long x_bs, const float *__restrict__ x_coef, int coef_gs,
    float x_lo, float *__restrict__ partial, int nwg_g) {
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
  float sc[AFF ? NX : 1], bi[AFF ? NX : 1];
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
      const float s0 = (ok && !isdy) ? x_coef[((size_t)g * coef_gs + r) * 4] : 0.f;
      const float b0 = (ok && !isdy) ? x_coef[((size_t)g * coef_gs + r) * 4 + 1] : 0.f;
      sc[i] = s0;
      bi[i] = b0;
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

  // TWO tiles in flight in registers: the loads of tile t + 2 are issued at the top of tile t and
  // written to LDS at the bottom of tile t + 1, i.e. 1.5 - 2 tile times (6 000 - 16 000 cycles with
  // the SIMD's two waves taking turns on the matrix pipe) later; with one tile in flight the wave
  // that gets the pipe first waited for HBM at the bottom of every tile (14 000 cycles per tile for
  // 8 192 of MFMA: 57 % utilisation).  Loads are unconditional (past the end: the last tile again),
  // so the wait in front of a write is a counted vmcnt(NX), not a drain.
  f32x4 stg[2][NX];
  auto load_tile = [&](auto sc_, int t) {
    constexpr int S = decltype(sc_)::value;
    t = t < ntiles ? t : ntiles - 1;
    const int n = g + ng * (t / tpb);
    const long long p0 = (long long)(t % tpb) * PT;
    const float *dyb = dy + (size_t)n * dy_bs + p0, *xb = x + (size_t)n * x_bs + p0;   // uniform
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      // (opaque copy: the zero-extension of the offset must stay in this block, or instruction
      // selection sees a hoisted 64-bit register pair and drops the scalar-base addressing form)
      unsigned o = goff[i];
      asm volatile("" : "+v"(o));
      stg[S][i] = load16_saddr(o, i < DYSLOTS ? dyb : xb);
    }
  };
  auto write_tile = [&](auto sc_, float *buf) {
    constexpr int S = decltype(sc_)::value;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      f32x4 q = stg[S][i];
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

  float *b0 = lds, *b1 = lds + TILE;
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  int t = rank;
  if (t < ntiles) {
    load_tile(S0{}, t);
    write_tile(S0{}, b0);
    load_tile(S0{}, t + nwg_g);           // stg[0] = tile t + 1 ...
  }
  // one tile: barrier, loads of the tile after next into the set `LD`, MFMAs of the tile in
  // b0, the set `WR` (next tile, loaded one tile ago) to b1
  auto tile = [&](auto ldc, auto wrc) {
    lgkm_wait<0>();
    __builtin_amdgcn_s_barrier();
    load_tile(ldc, t + 2 * nwg_g);
    // lane (l16 = row inside its block, quad): positions 16 pg + 4 quad .. + 3
    const unsigned la = lds_addr(b0) + (unsigned)(((wm * MB * 16 + l16) * PITCH + 4 * quad) * 4);
    const unsigned lb = lds_addr(b0) + (unsigned)(((CO16 * 16 + wn * NB * 16 + l16) * PITCH + 4 * quad) * 4);
    // ONE set of fragment registers (the second position group is read behind the MFMAs of the
    // first: its LDS latency is covered by the SIMD's other wave; two sets + two staging sets
    // spilled)
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
    write_tile(wrc, b1);     // (after the last tile: a copy of it that nobody reads)
    float *const tb = b0; b0 = b1; b1 = tb;
    t += nwg_g;
  };
  while (t + nwg_g < ntiles) {     // pairs (the staging sets swap roles); no exit from the middle:
    tile(S1{}, S0{});              // the accumulators stay in place
    tile(S0{}, S1{});
  }
  if (t < ntiles) tile(S1{}, S0{});
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
// (a launch covers the ci columns [col0, col0 + ci) of dw (ng, co, ld): wide layers run as column blocks)
__global__ __launch_bounds__(1024) void pw_wgrad_reduce_kernel(int total, int nparts,
                                                               const float *__restrict__ partial,
                                                               float *__restrict__ dw, int ci, int ld,
                                                               int col0, int co) {
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
    dw[((size_t)g * co + i / ci) * ld + col0 + i % ci] = tt;
  }
}

static int pw_wgrad_nwg(int nb, int ng, long long p) {
  long long nwg = 256 / ng;
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

extern "C" size_t nesie_pw_wgrad_workspace_bytes(int nb, int ng, int co, int ci, long long p) {
  if (nb <= 0 || ng <= 0 || p <= 0) return 0;
  return (size_t)ng * pw_wgrad_nwg(nb, ng, p) * co * pw_wgrad_block(co, ci) * sizeof(float);
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
  NESIE_REQUIRE((long long)(nb / ng) * (p / 32) < (1ll << 30), W);     // (32-bit tile cursor)
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
      hipLaunchKernelGGL(kern, dim3(nwg * ng), dim3(512), lds, s, nb, ng, co, cw, p, dy,         \
                         dy_bstride, xc, x_bstride, cc, ci, lo, partial, nwg);                   \
    } else {                                                                                     \
      auto kern = pw_wgrad_kernel<CO16, CI16, WM, WN, false>;                                    \
      static bool attr = false;                                                                  \
      if (!attr) {                                                                               \
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        attr = true;                                                                             \
      }                                                                                          \
      hipLaunchKernelGGL(kern, dim3(nwg * ng), dim3(512), lds, s, nb, ng, co, cw, p, dy,         \
                         dy_bstride, xc, x_bstride, cc, ci, lo, partial, nwg);                   \
    }                                                                                            \
  } while (0)
  const int block = pw_wgrad_block(co, ci);
  for (int c0 = 0; c0 < ci; c0 += block) {     // column blocks [c0, c0 + cw) of dw; same stream: the
    const int cw = ci - c0 < block ? ci - c0 : block;   // workspace is free again when the next one starts
    const float *xc = x + (size_t)c0 * p;
    const float *cc = x_coef ? x_coef + (size_t)c0 * 4 : nullptr;
    if (co <= 64 && cw <= 64) L(4, 4, 2, 4);
    else if (co <= 128 && cw <= 64) L(8, 4, 4, 2);
    else if (co <= 128 && cw <= 128) L(8, 8, 2, 4);
    else if (co <= 128 && cw <= 192) L(8, 12, 2, 4);
    else if (co <= 128 && cw <= 256) L(8, 16, 2, 4);
    else if (co <= 128) L(8, 20, 2, 4);
    else L(16, 8, 4, 2);
    const int total = co * cw;
    hipLaunchKernelGGL(pw_wgrad_reduce_kernel, dim3(cdiv(total, 64), ng), dim3(1024), 0, s, total, nwg,
                       partial, dw, cw, ci, c0, co);
  }
#undef L
  return check_launch(W);
}
