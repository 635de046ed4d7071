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
