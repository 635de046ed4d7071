// Furthest point sampling for gfx950.
//
// Replaces furthest_point_sampling_kernel / ..._with_dist_kernel
// (reference mmdet3d/ops/furthest_point_sample/src/furthest_point_sample_cuda.cu:26-141,
// :214-331).  One workgroup owns one scene; the M-1 rounds are serial.
//
// Winner selection.  The reference reduces (value, index) pairs with a
// thread-striped scan followed by an LDS tree whose tie-break is "smallest
// bit-reversed owner thread id, then smallest k" (SURVEY.md appendix A.1).  We
// reduce ONE order-independent 64-bit key per point with a plain unsigned max:
//     key = float_bits(d2) << 32 | (0xFFFFFFFF - (bitrev_L(k mod bs) << 22 | k / bs))
// (bs = the reference's block size for this n, L = log2 bs), so any wave/LDS
// reduction shape gives the reference's winner.  d2 >= +0 so its IEEE bits
// order like the value.
//
// Data placement.  The running-min array `temp` lives in VGPRs for the whole
// kernel (PPT values per lane); for n <= 4096 the coordinates do too, above
// that they are re-streamed from L2 each round with dense 12-byte-per-lane
// loads (480 KB per scene at n = 40000 does not fit one CU's registers + LDS).
#include "common.h"
#include "fps_keys.h"
#include <stdlib.h>

namespace nesie {

// Order-preserving map float -> u32 for ANY sign (the F-FPS distance matrix is
// a^2 + b^2 - 2ab and goes slightly negative); -0 is folded onto +0 first
// because the reference compares them equal.
__device__ __forceinline__ unsigned ordered_bits(float f) {
  f = f + 0.0f;
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Block-wide max of one u64 per thread; NW waves; one barrier per call.
// `red` is [2][NW]; callers alternate `parity` between consecutive calls.
template <int NW>
__device__ __forceinline__ unsigned long long block_max_u64(
    unsigned long long v, unsigned long long (*red)[NW], int parity) {
  v = wave_max_u64(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (NW == 1) return v;
  if (lane == 0) red[parity][wave] = v;
  __syncthreads();
  unsigned long long r = red[parity][0];
#pragma unroll
  for (int w = 1; w < NW; ++w) {
    unsigned long long o = red[parity][w];
    r = o > r ? o : r;
  }
  return r;
}

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
  // old = 0 with bound_ctrl: every permutation used here reads a live lane, and a zero fill is the
  // identity of the unsigned max that follows, so the compiler folds the move into v_max_u32_dpp
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
// max over the 64 lanes, returned wave-uniform
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
  unsigned o;
  o = dpp_u32<0xB1>(v);  v = o > v ? o : v;   // quad_perm [1,0,3,2]
  o = dpp_u32<0x4E>(v);  v = o > v ? o : v;   // quad_perm [2,3,0,1]
  o = dpp_u32<0x141>(v); v = o > v ? o : v;   // row_half_mirror
  o = dpp_u32<0x140>(v); v = o > v ? o : v;   // row_mirror
  unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  a = a > b ? a : b; c = c > d ? c : d;
  return a > c ? a : c;
}
__device__ __forceinline__ float unordered_bits(unsigned u) {  // inverse of ordered_bits
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}
__device__ __forceinline__ float wave_fmax(float f) {  // any sign, wave-uniform
  return unordered_bits(wave_umax(ordered_bits(f)));
}
__device__ __forceinline__ float wave_fmin(float f) {
  return unordered_bits(~wave_umax(~ordered_bits(f)));
}

// (value, key) arg-max over the lanes of a wave, wave-uniform results.  Keys are unique
// among lanes with a non-zero key, so the common case (one lane holds the maximum) needs
// a single DPP reduction; ties take a second one over the low key words.
__device__ __forceinline__ int wave_argmax(bool valid, unsigned val, unsigned lo,
                                           unsigned &vmax, unsigned &lomax) {
  vmax = wave_umax(valid ? val : 0u);
  unsigned long long eq = __builtin_amdgcn_ballot_w64(valid && val == vmax);
  if (__popcll(eq) == 1) {
    const int wl = __builtin_ctzll(eq);
    lomax = (unsigned)__builtin_amdgcn_readlane((int)lo, wl);
    return wl;
  }
  lomax = wave_umax((valid && val == vmax) ? lo : 0u);
  eq = __builtin_amdgcn_ballot_w64(valid && val == vmax && lo == lomax);
  return eq ? __builtin_ctzll(eq) : 0;
}

// The same arg-max when the wave holds G identical groups of `GROUP` consecutive lanes (the
// cross-wave stage: lane i carries candidate i mod GROUP): the reduction stays inside a DPP row,
// so the four cross-row readlanes of wave_umax and their scalar maxima are not needed.
template <int GROUP>
__device__ __forceinline__ unsigned group_umax(unsigned v) {
  unsigned o;
  o = dpp_u32<0xB1>(v);  v = o > v ? o : v;   // quad_perm [1,0,3,2]
  o = dpp_u32<0x4E>(v);  v = o > v ? o : v;   // quad_perm [2,3,0,1]
  if (GROUP >= 8) { o = dpp_u32<0x141>(v); v = o > v ? o : v; }  // row_half_mirror
  if (GROUP >= 16) { o = dpp_u32<0x140>(v); v = o > v ? o : v; } // row_mirror
  return v;
}
template <int GROUP>
__device__ __forceinline__ int group_argmax(bool valid, unsigned val, unsigned lo, unsigned &lomax) {
  static_assert(GROUP == 4 || GROUP == 8 || GROUP == 16, "one DPP row or less");
  const unsigned vmax = group_umax<GROUP>(valid ? val : 0u);
  unsigned long long eq = __builtin_amdgcn_ballot_w64(valid && val == vmax);
  if (__popcll(eq) == 64 / GROUP) {
    const int wl = __builtin_ctzll(eq);
    lomax = (unsigned)__builtin_amdgcn_readlane((int)lo, wl);
    return wl;
  }
  const unsigned lm = group_umax<GROUP>((valid && val == vmax) ? lo : 0u);
  eq = __builtin_amdgcn_ballot_w64(valid && val == vmax && lo == lm);
  const int wl = eq ? __builtin_ctzll(eq) : 0;
  lomax = (unsigned)__builtin_amdgcn_readlane((int)lm, wl);
  return wl;
}

__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// ---- n <= BLOCK*PPT, everything in registers --------------------------------
template <int BLOCK, int PPT>
__global__ __launch_bounds__(BLOCK) void fps_reg_kernel(
    int n, int m, int L, const float *__restrict__ xyz, float *__restrict__ temp,
    int *__restrict__ idx) {
  constexpr int NW = BLOCK / 64;
  __shared__ unsigned red_v[2][NW], red_l[2][NW];
  __shared__ float sx[BLOCK * PPT], sy[BLOCK * PPT], sz[BLOCK * PPT];  // <= 48 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  xyz += (size_t)blockIdx.x * n * 3;
  temp += (size_t)blockIdx.x * n;
  idx += (size_t)blockIdx.x * m;

  float px[PPT], py[PPT], pz[PPT], tp[PPT];
  unsigned klo[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    int k = j * BLOCK + tid;
    bool ok = k < n;
    px[j] = ok ? xyz[k * 3 + 0] : 0.f;
    py[j] = ok ? xyz[k * 3 + 1] : 0.f;
    pz[j] = ok ? xyz[k * 3 + 2] : 0.f;
    tp[j] = ok ? temp[k] : 0.f;
    klo[j] = ok ? key_lo_of(k, L) : 0u;  // key 0 never wins: a real k has lo > 0
    sx[k] = px[j]; sy[k] = py[j]; sz[k] = pz[j];  // k < BLOCK*PPT always
  }
  __syncthreads();
  float x1 = sx[0], y1 = sy[0], z1 = sz[0];
  int my_pick = 0;  // idx[0] = 0
  for (int r = 1; r < m; ++r) {
    // invalid slots hold tp = 0 / klo = 0: their key (0, 0) loses to every real key
    unsigned long long best = 0ull;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      float d = sqdist_nofma(px[j] - x1, py[j] - y1, pz[j] - z1);
      float d2 = fminf(d, tp[j]);
      tp[j] = d2;
      const unsigned long long key =
          ((unsigned long long)__float_as_uint(d2) << 32) | klo[j];
      best = key > best ? key : best;
    }
    const unsigned bv = (unsigned)(best >> 32), bl = (unsigned)best;
    unsigned wv, wl_;
    wave_argmax(bl != 0u, bv, bl, wv, wl_);
    unsigned gl = wl_;
    if constexpr (NW > 1) {
      if (lane == 0) { red_v[r & 1][wave] = wv; red_l[r & 1][wave] = wl_; }
      __syncthreads();
      const unsigned cv = red_v[r & 1][lane & (NW - 1)], cl = red_l[r & 1][lane & (NW - 1)];
      group_argmax<NW>(cl != 0u, cv, cl, gl);  // NW candidates replicated over the lanes
    }
    const int old = k_of_key_lo(gl, L);
    x1 = sx[old]; y1 = sy[old]; z1 = sz[old];  // LDS broadcast, no global read
    // thread (r mod BLOCK) keeps round r's pick; stored BLOCK rounds at a time (no global store
    // inside the dependent rounds)
    my_pick = tid == (r & (BLOCK - 1)) ? old : my_pick;
    if ((r & (BLOCK - 1)) == BLOCK - 1) idx[(r & ~(BLOCK - 1)) + tid] = my_pick;
  }
  if (((m - 1) & (BLOCK - 1)) != BLOCK - 1) {
    const int base = (m - 1) & ~(BLOCK - 1);
    if (base + tid <= m - 1) idx[base + tid] = my_pick;
  }
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    int k = j * BLOCK + tid;
    if (k < n) temp[k] = tp[j];
  }
}

// ---- 1024 <= n <= 1024*PPT: temp in registers, xyz streamed from L2 ----------
// Here the reference block size is 1024 == BLOCK, so k mod bs == tid and
// k / bs == j: the low key word is (base - j).
template <int PPT>
__global__ __launch_bounds__(1024) void fps_stream_kernel(
    int n, int m, const float *__restrict__ xyz, float *__restrict__ temp,
    int *__restrict__ idx) {
  constexpr int BLOCK = 1024, NW = 16, L = 10;
  __shared__ unsigned long long red[2][NW];
  const int tid = threadIdx.x;
  xyz += (size_t)blockIdx.x * n * 3;
  temp += (size_t)blockIdx.x * n;
  idx += (size_t)blockIdx.x * m;

  float tp[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    int k = j * BLOCK + tid;
    tp[j] = k < n ? temp[k] : 0.f;
  }
  const unsigned base_lo = key_lo_of(tid, L);  // q = 0
  const char *sbase = (const char *)xyz;
  const unsigned off0 = (unsigned)tid * 12u;
  const unsigned last_off = (unsigned)(n - 1) * 12u;
  int old = 0;
  if (tid == 0) idx[0] = 0;
  for (int r = 1; r < m; ++r) {
    const float x1 = xyz[old * 3 + 0], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2];
    float bestv = -1.f;
    unsigned bestlo = 0u;
    // CH points per lane are in flight at a time: a memory clobber between the
    // chunks stops the compiler hoisting all 3*PPT loads (which spills).
    constexpr int CH = 8;
#pragma unroll
    for (int j0 = 0; j0 < PPT; j0 += CH) {
      float qx[CH], qy[CH], qz[CH];
      // byte offset of this lane's first point of the chunk; made opaque so the
      // per-slot addresses are re-derived here instead of being hoisted out of
      // the round loop into 2*PPT live registers.
      unsigned off = off0 + (unsigned)j0 * (BLOCK * 12u);
      asm volatile("" : "+v"(off));
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        unsigned o = off + (unsigned)u * (BLOCK * 12u);
        o = o < last_off ? o : last_off;  // slots past n re-read the last point
        const float *p = (const float *)(sbase + o);
        qx[u] = p[0]; qy[u] = p[1]; qz[u] = p[2];
      }
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int j = j0 + u;
        const int k = j * BLOCK + tid;
        float d = sqdist_nofma(qx[u] - x1, qy[u] - y1, qz[u] - z1);
        float d2 = fminf(d, tp[j]);
        // slots are visited with decreasing low word, so a strict '>' on the
        // value keeps the larger key among equal values.
        bool take = (k < n) && d2 > bestv;
        tp[j] = k < n ? d2 : tp[j];
        bestlo = take ? base_lo - (unsigned)j : bestlo;
        bestv = take ? d2 : bestv;
      }
      asm volatile("" ::: "memory");
    }
    unsigned long long best =
        ((unsigned long long)__float_as_uint(bestv) << 32) | bestlo;
    best = block_max_u64<NW>(best, red, r & 1);
    old = k_of_key_lo((unsigned)best, L);
    old = __builtin_amdgcn_readfirstlane(old);
    if (tid == 0) idx[r] = old;
  }
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    int k = j * BLOCK + tid;
    if (k < n) temp[k] = tp[j];
  }
}

// ---- any n: temp stays in global memory (reference-shaped, key reduction) ----
// WITH_DIST: `src` is the (N,N) distance matrix of furthest_point_sample_cuda.cu:214-331.
template <bool WITH_DIST, int FORM = 0>
__global__ __launch_bounds__(1024) void fps_generic_kernel(
    int n, int m, int L, const float *__restrict__ src, float *__restrict__ temp,
    int *__restrict__ idx) {
  constexpr int NW = 16;
  __shared__ unsigned long long red[2][NW];
  const int tid = threadIdx.x;
  src += (size_t)blockIdx.x * (WITH_DIST ? (size_t)n * n : (size_t)n * 3);
  temp += (size_t)blockIdx.x * n;
  idx += (size_t)blockIdx.x * m;
  int old = 0;
  if (tid == 0) idx[0] = 0;
  for (int r = 1; r < m; ++r) {
    float x1 = 0.f, y1 = 0.f, z1 = 0.f;
    if (!WITH_DIST) {
      x1 = src[old * 3 + 0]; y1 = src[old * 3 + 1]; z1 = src[old * 3 + 2];
    }
    unsigned long long best = 0ull;
    for (int k = tid; k < n; k += 1024) {
      float d;
      if (WITH_DIST) d = src[(size_t)old * n + k];
      else d = sqdist_form<FORM>(src[k * 3 + 0] - x1, src[k * 3 + 1] - y1, src[k * 3 + 2] - z1);
      float d2 = fminf(d, temp[k]);
      temp[k] = d2;
      unsigned long long key =
          ((unsigned long long)ordered_bits(d2) << 32) | key_lo_of(k, L);
      best = key > best ? key : best;
    }
    best = block_max_u64<NW>(best, red, r & 1);
    old = k_of_key_lo((unsigned)best, L);
    old = __builtin_amdgcn_readfirstlane(old);
    if (tid == 0) idx[r] = old;
  }
}


// ---- 4096 < n <= 65536: bucket-pruned FPS ---------------------------------------
// The points are first counting-sorted by a 13-bit Morton cell, so every run of 64
// sorted points (a BUCKET, one point per lane of a wave) is spatially compact.
// Each bucket keeps its bounding box and the largest running-min distance of its
// points.  A new sample at c can only lower temp[p] if d(p, c) < temp[p]; with
//     d2box = ((ex*ex)+(ey*ey))+(ez*ez),  ex = max(lo.x - c.x, c.x - hi.x, 0), ...
// computed by the SAME fp32 operations as a point distance, rounding monotonicity
// gives d(p, c) >= d2box for every p in the box, so a bucket with d2box >= bmax is
// skipped with results bit-identical to the full scan.  After a few dozen rounds
// only the buckets around the new sample are touched instead of all n points.
//
// Layout: workspace pts[s] = (x, y, z, temp) as float4 in sorted order + key[s], the tie-break
// key (key_lo_of) of the point's original index;
// bucket b belongs to wave b % 16 and its metadata (box, best value/key, the best
// point's xyz) lives in lane b / 16 of that wave.  Per round: prune test (one lane
// per bucket) -> ballot -> the active buckets are re-evaluated 64 points at a time
// -> per-wave best -> one LDS exchange + one barrier -> every wave derives the
// winner and its coordinates (no global read on the critical path).
__device__ __forceinline__ unsigned part1by2(unsigned x) {  // 3 bits -> bits 0,3,6
  return (x & 1u) | ((x & 2u) << 2) | ((x & 4u) << 4);
}
__device__ __forceinline__ unsigned part1by1(unsigned x) {  // 2 bits -> bits 0,2
  return (x & 1u) | ((x & 2u) << 1);
}

struct FpsCand { unsigned val, lo; float x, y, z; };

constexpr int FPS_CELLS = 8192;

// NW waves per scene; lane j of wave w owns buckets (q*64 + j)*NW + w, q < BPL.
// LT: the running-min distances of the sorted points live in LDS for the whole selection loop
// (n floats, overlaying the phase-0 histogram) instead of in the w component of the workspace.
// On gfx9 stores and loads share vmcnt and return in order: with the distances in global memory
// every round's wait for its bucket loads also waits out the store acknowledgements of the
// previous round's updates.  Fits up to n = 40 192 points in the 160 KB of a CU.
template <int NW, bool LT, bool FULL>
__global__ __launch_bounds__(NW * 64) void fps_pruned_kernel(
    int n, int m, const float *__restrict__ xyz, float *__restrict__ temp,
    int *__restrict__ idx, float4 *__restrict__ ws_pts, unsigned *__restrict__ ws_orig) {
  constexpr int L = 10, BLOCK = NW * 64, BPL = 16 / NW;  // 1024 buckets max
  constexpr int BINS_PER_THREAD = FPS_CELLS / BLOCK;
  extern __shared__ __attribute__((aligned(16))) unsigned fps_dyn[];
  __shared__ unsigned hist_static[LT ? 1 : FPS_CELLS];
  __shared__ unsigned scan_static[LT ? 1 : BLOCK];
  unsigned *hist = LT ? fps_dyn : hist_static;
  unsigned *scan_part = LT ? fps_dyn + FPS_CELLS : scan_static;
  float *lt = (float *)fps_dyn;                      // phase 1 only (LT)
  __shared__ float bb[NW][6];
  __shared__ FpsCand red[2][NW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  xyz += (size_t)blockIdx.x * n * 3;
  temp += (size_t)blockIdx.x * n;
  idx += (size_t)blockIdx.x * m;
  ws_pts += (size_t)blockIdx.x * n;
  ws_orig += (size_t)blockIdx.x * n;

  // ---- phase 0: scene box, Morton histogram, scan, scatter -----------------------
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k = tid; k < n; k += BLOCK) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float v = xyz[k * 3 + a];
      mn[a] = fminf(mn[a], v); mx[a] = fmaxf(mx[a], v);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float lo = wave_fmin(mn[a]), hi = wave_fmax(mx[a]);
    if (lane == 0) { bb[wave][a] = lo; bb[wave][3 + a] = hi; }
  }
  for (int i = tid; i < FPS_CELLS; i += BLOCK) hist[i] = 0u;
  __syncthreads();
  float smin[3], sinv[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float lo_f = bb[0][a], hi_f = bb[0][3 + a];
    for (int w = 1; w < NW; ++w) {
      lo_f = fminf(lo_f, bb[w][a]); hi_f = fmaxf(hi_f, bb[w][3 + a]);
    }
    smin[a] = lo_f;
    const float ext = hi_f - lo_f;
    sinv[a] = ext > 0.f ? (a == 2 ? 8.f : 32.f) / ext : 0.f;
  }
  auto cell_of = [&](float x, float y, float z) -> unsigned {
    int qx = (int)((x - smin[0]) * sinv[0]), qy = (int)((y - smin[1]) * sinv[1]);
    int qz = (int)((z - smin[2]) * sinv[2]);
    qx = qx < 0 ? 0 : (qx > 31 ? 31 : qx);
    qy = qy < 0 ? 0 : (qy > 31 ? 31 : qy);
    qz = qz < 0 ? 0 : (qz > 7 ? 7 : qz);
    unsigned fine = part1by2(qx & 7) | (part1by2(qy & 7) << 1) | (part1by2(qz) << 2);
    unsigned coarse = part1by1(qx >> 3) | (part1by1(qy >> 3) << 1);
    return (coarse << 9) | fine;
  };
  for (int k = tid; k < n; k += BLOCK)
    atomicAdd(&hist[cell_of(xyz[k * 3], xyz[k * 3 + 1], xyz[k * 3 + 2])], 1u);
  __syncthreads();
  {  // exclusive scan of the counters: BINS_PER_THREAD each + Hillis-Steele over the block
    unsigned loc[BINS_PER_THREAD], sum = 0;
#pragma unroll
    for (int i = 0; i < BINS_PER_THREAD; ++i) {
      loc[i] = hist[tid * BINS_PER_THREAD + i]; sum += loc[i];
    }
    scan_part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < BLOCK; off <<= 1) {
      unsigned v = tid >= off ? scan_part[tid - off] : 0u;
      __syncthreads();
      scan_part[tid] += v;
      __syncthreads();
    }
    unsigned run = scan_part[tid] - sum;
#pragma unroll
    for (int i = 0; i < BINS_PER_THREAD; ++i) {
      hist[tid * BINS_PER_THREAD + i] = run; run += loc[i];
    }
  }
  __syncthreads();
  for (int k = tid; k < n; k += BLOCK) {
    float x = xyz[k * 3], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];
    unsigned pos = atomicAdd(&hist[cell_of(x, y, z)], 1u);
    // the tie-break key of the original index (k_of_key_lo inverts it).  LT: the key rides in the
    // w component (one 16-byte load per point in the loop) and the initial distance is parked
    // in the key array until it moves to LDS below.
    const unsigned key = key_lo_of(k, L);
    ws_pts[pos] = make_float4(x, y, z, LT ? __uint_as_float(key) : temp[k]);
    ws_orig[pos] = LT ? __float_as_uint(temp[k]) : key;
  }
  __syncthreads();  // also drains the stores: the block re-reads them below
  if (LT) {         // the histogram is dead: its bytes become the distance array
    for (int s_ = tid; s_ < n; s_ += BLOCK) lt[s_] = __uint_as_float(ws_orig[s_]);
    __syncthreads();
  }

  // ---- phase 1 -------------------------------------------------------------------
  const int nb = (n + 63) >> 6;  // buckets
  float blox[BPL], bloy[BPL], bloz[BPL], bhix[BPL], bhiy[BPL], bhiz[BPL];
  float bx[BPL], by[BPL], bz[BPL];
  unsigned bmax[BPL], blo[BPL];
  bool have[BPL];
#pragma unroll
  for (int q = 0; q < BPL; ++q) {
    blox[q] = bloy[q] = bloz[q] = bhix[q] = bhiy[q] = bhiz[q] = 0.f;
    bx[q] = by[q] = bz[q] = 0.f;
    bmax[q] = blo[q] = 0u;
    have[q] = (q * 64 + lane) * NW + wave < nb;
  }
  float cx = xyz[0], cy = xyz[1], cz = xyz[2];  // idx[0] = 0

  struct Loaded { float4 p; unsigned o; bool valid; int ss; };
  auto fetch = [&](int q, int j) {
    Loaded l;
    const int s = (((q * 64 + j) * NW + wave) << 6) + lane;
    l.valid = FULL || s < n;  // FULL: n is a multiple of 64, every slot holds a point
    l.ss = l.valid ? s : n - 1;
    l.p = ws_pts[l.ss];
    if (LT) { l.o = __float_as_uint(l.p.w); l.p.w = lt[l.ss]; }
    else l.o = ws_orig[l.ss];
    return l;
  };
  // Re-evaluate one bucket against the centre; INIT only (re)builds its box and best.
  auto evaluate = [&](const Loaded &l, int j, bool init, unsigned &o_max, unsigned &o_lo,
                      float &o_x, float &o_y, float &o_z) {
    float d2 = l.p.w;
    if (!init) {
      const float d = sqdist_nofma(l.p.x - cx, l.p.y - cy, l.p.z - cz);
      // fminf() without the NaN-quieting pre-pass (distances are never NaN)
      asm("v_min_f32 %0, %1, %2" : "=v"(d2) : "v"(d), "v"(l.p.w));
      if (LT) {
        // unconditional: cheaper than the compare + exec-mask round trip; the padding lanes of
        // the last bucket hold copies of point n-1 and write the value its own lane writes
        lt[l.ss] = d2;
      } else if (l.valid && d2 != l.p.w) {
        ws_pts[l.ss].w = d2;
      }
    }
    const unsigned lo = l.valid ? l.o : 0u;
    unsigned vmax, lomax;
    const int wl = wave_argmax(l.valid, __float_as_uint(d2), lo, vmax, lomax);
    const float rx = readlane_f(l.p.x, wl), ry = readlane_f(l.p.y, wl);
    const float rz = readlane_f(l.p.z, wl);
    const bool mine = lane == j;  // selects, not a branch: no exec-mask round trip
    o_max = mine ? vmax : o_max; o_lo = mine ? lomax : o_lo;
    o_x = mine ? rx : o_x; o_y = mine ? ry : o_y; o_z = mine ? rz : o_z;
  };

#pragma unroll
  for (int q = 0; q < BPL; ++q) {
    for (int j = 0; j < 64; ++j) {
      if (((q * 64 + j) * NW + wave) >= nb) break;  // wave-uniform
      const Loaded l = fetch(q, j);
      evaluate(l, j, true, bmax[q], blo[q], bx[q], by[q], bz[q]);
      // invalid lanes duplicate the last point: harmless for a box
      const float lx = wave_fmin(l.p.x), ly = wave_fmin(l.p.y), lz = wave_fmin(l.p.z);
      const float hx = wave_fmax(l.p.x), hy = wave_fmax(l.p.y), hz = wave_fmax(l.p.z);
      if (lane == j) {
        blox[q] = lx; bloy[q] = ly; bloz[q] = lz; bhix[q] = hx; bhiy[q] = hy; bhiz[q] = hz;
      }
    }
  }

  if (tid == 0) idx[0] = 0;
  // The wave's candidate survives a round in which the bucket that holds it was not re-evaluated:
  // running minima only fall, so every other bucket's best can only have dropped further.
  FpsCand c{0u, 0u, 0.f, 0.f, 0.f};
  int c_lane = -1;  // lane (bucket slot) the candidate came from; -1 = none yet
  int c_age = 0;    // exchange slots already holding the candidate (0..2)
  unsigned my_key = key_lo_of(0, L);  // idx[0] = 0
  for (int r = 1; r < m; ++r) {
    unsigned long long touched = 0ull;
#pragma unroll
    for (int q = 0; q < BPL; ++q) {
      if (q * 64 * NW >= nb) break;  // no bucket in this slot for any lane
      // prune test, one lane per bucket
      const float ex = fmaxf(fmaxf(blox[q] - cx, cx - bhix[q]), 0.f);
      const float ey = fmaxf(fmaxf(bloy[q] - cy, cy - bhiy[q]), 0.f);
      const float ez = fmaxf(fmaxf(bloz[q] - cz, cz - bhiz[q]), 0.f);
      const float d2box = sqdist_nofma(ex, ey, ez);
      unsigned long long mask = __builtin_amdgcn_ballot_w64(have[q] && d2box < __uint_as_float(bmax[q]));
      touched |= mask;
      // waves with buckets to re-evaluate are the round's critical path: let them win the SIMD's
      // issue arbitration over the waves that only run the fixed prune / exchange steps
      if (mask) __builtin_amdgcn_s_setprio(3);
      while (mask) {  // two buckets per trip so their loads overlap
        const int j0 = __builtin_ctzll(mask);
        mask &= mask - 1;
        const bool two = mask != 0;
        const int j1 = two ? __builtin_ctzll(mask) : j0;
        mask &= mask - 1;
        const Loaded l0 = fetch(q, j0);
        const Loaded l1 = fetch(q, j1);
        evaluate(l0, j0, false, bmax[q], blo[q], bx[q], by[q], bz[q]);
        if (two) evaluate(l1, j1, false, bmax[q], blo[q], bx[q], by[q], bz[q]);
      }
    }
    if (BPL > 1 || c_lane < 0 || ((touched >> c_lane) & 1ull)) {
      // best bucket of this lane, then of this wave
      unsigned mv = have[0] ? bmax[0] : 0u, ml = have[0] ? blo[0] : 0u;
      float mxx = bx[0], myy = by[0], mzz = bz[0];
#pragma unroll
      for (int q = 1; q < BPL; ++q) {
        const bool better = have[q] && (bmax[q] > mv || (bmax[q] == mv && blo[q] > ml));
        mv = better ? bmax[q] : mv; ml = better ? blo[q] : ml;
        mxx = better ? bx[q] : mxx; myy = better ? by[q] : myy; mzz = better ? bz[q] : mzz;
      }
      const int wl = wave_argmax(ml != 0u, mv, ml, c.val, c.lo);
      c.x = readlane_f(mxx, wl); c.y = readlane_f(myy, wl); c.z = readlane_f(mzz, wl);
      c_lane = wl;
      c_age = 0;
    }
    if (c_age < 2) {  // both exchange slots of this wave hold the candidate after two rounds
      if (lane == 0) red[r & 1][wave] = c;
      ++c_age;
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    // every wave reduces the NW candidates (lane i holds candidate i mod NW)
    const FpsCand g = red[r & 1][lane & (NW - 1)];
    unsigned gl;
    // every wave owns buckets (n > 4096 means nb > NW), so all NW candidates are live
    const int gwl = group_argmax<NW>(true, g.val, g.lo, gl);
    cx = readlane_f(g.x, gwl); cy = readlane_f(g.y, gwl); cz = readlane_f(g.z, gwl);
    // thread (r mod BLOCK) keeps round r's winner in a register; stored BLOCK rounds at a time, so
    // no global store (and its acknowledgement in wave 0's vmcnt queue) sits inside a round
    my_key = tid == (r & (BLOCK - 1)) ? gl : my_key;
    if ((r & (BLOCK - 1)) == BLOCK - 1) idx[(r & ~(BLOCK - 1)) + tid] = k_of_key_lo(my_key, L);
  }
  if (((m - 1) & (BLOCK - 1)) != BLOCK - 1) {
    const int base = (m - 1) & ~(BLOCK - 1);
    if (base + tid <= m - 1) idx[base + tid] = k_of_key_lo(my_key, L);
  }
  __syncthreads();
  if (LT) {
    // the key array is dead in this mode (the keys ride in pts.w): leave the bucket boxes there,
    // box[6][nb], so that the sorted scene doubles as a spatial index (nesie_ball_query_indexed)
    float *box = (float *)ws_orig;
#pragma unroll
    for (int q = 0; q < BPL; ++q) {
      const int bid = (q * 64 + lane) * NW + wave;
      if (bid < nb) {
        box[bid] = blox[q]; box[nb + bid] = bloy[q]; box[2 * nb + bid] = bloz[q];
        box[3 * nb + bid] = bhix[q]; box[4 * nb + bid] = bhiy[q]; box[5 * nb + bid] = bhiz[q];
      }
    }
  }
  // running-min distances back to the caller's order
  for (int s = tid; s < n; s += BLOCK)
    temp[k_of_key_lo(LT ? __float_as_uint(ws_pts[s].w) : ws_orig[s], L)] = LT ? lt[s] : ws_pts[s].w;
}

}  // namespace nesie

using namespace nesie;

static int fps_launch(int b, int n, int m, const float *xyz, float *temp, int *idx,
                      void *workspace, size_t workspace_bytes, void *stream);

// running-min distances in LDS (and the spatial index left behind): n floats must fit the CU
static bool fps_lds_mode(int n) {
  static const int enabled = [] {
    const char *e = getenv("NESIE_FPS_LDS");
    return e ? atoi(e) : 1;
  }();
  return enabled && n <= FPS_INDEX_MAX_N;
}

extern "C" int nesie_fps_leaves_index(int b, int n) {
  // (a fused distance form runs on the plain kernel, which builds no index)
  return distance_form() == 0 && nesie_fps_workspace_bytes(b, n) != 0 && fps_lds_mode(n);
}

extern "C" size_t nesie_fps_workspace_bytes(int b, int n) {
  if (b <= 0 || n <= 4096 || n > 65536) return 0;
  return (size_t)b * n * 20;  // float4 sorted points + u32 original index
}

extern "C" int nesie_furthest_point_sampling_ws(int b, int n, int m, const float *xyz,
                                                float *temp, int *idx, void *workspace,
                                                size_t workspace_bytes, void *stream) {
  return fps_launch(b, n, m, xyz, temp, idx, workspace, workspace_bytes, stream);
}

extern "C" int nesie_furthest_point_sampling_wrapper(int b, int n, int m,
                                                     const float *xyz, float *temp,
                                                     int *idx, void *stream) {
  return fps_launch(b, n, m, xyz, temp, idx, nullptr, 0, stream);
}

static int fps_launch(int b, int n, int m, const float *xyz, float *temp, int *idx,
                      void *workspace, size_t workspace_bytes, void *stream) {
  const char *W = "furthest_point_sampling_wrapper";
  NESIE_REQUIRE(b >= 0 && n >= 0 && m >= 0, W);
  if (b == 0 || m <= 0) return NESIE_OK;  // reference kernel returns when m <= 0
  NESIE_REQUIRE(n >= 1 && xyz && temp && idx, W);
  NESIE_REQUIRE((long long)n * 3 < (1ll << 31), W);
  hipStream_t s = (hipStream_t)stream;
  const int L = fps_ref_log2_block(n);
  dim3 grid(b);
  if (distance_form() != 0) {   // checking aid (common.h, sqdist_form): the plain kernel only
    if (distance_form() == 1)
      hipLaunchKernelGGL((fps_generic_kernel<false, 1>), grid, dim3(1024), 0, s, n, m, L, xyz, temp, idx);
    else
      hipLaunchKernelGGL((fps_generic_kernel<false, 2>), grid, dim3(1024), 0, s, n, m, L, xyz, temp, idx);
    return check_launch(W);
  }
#define REG(BLK, P) \
  hipLaunchKernelGGL((fps_reg_kernel<BLK, P>), grid, dim3(BLK), 0, s, n, m, L, xyz, temp, idx)
#define STREAM(P) \
  hipLaunchKernelGGL((fps_stream_kernel<P>), grid, dim3(1024), 0, s, n, m, xyz, temp, idx)
  const size_t need = nesie_fps_workspace_bytes(b, n);
  if (need && workspace && workspace_bytes >= need &&
      ((uintptr_t)workspace & 15) == 0) {
    float4 *wp = (float4 *)workspace;
    unsigned *wo = (unsigned *)((char *)workspace + (size_t)b * n * 16);
    static const int nw = [] {
      const char *e = getenv("NESIE_FPS_WAVES");
      return e ? atoi(e) : 16;
    }();
    const bool lds_temps = fps_lds_mode(n);
    const size_t lt_bytes = (size_t)n * 4 > (FPS_CELLS + 1024) * 4 ? (size_t)n * 4 : (FPS_CELLS + 1024) * 4;
    const bool lds_ok = lds_temps;
#define PRUNED(NWV)                                                                              \
  do {                                                                                           \
    if (lds_ok && n % 64 == 0) {                                                                 \
      auto kern = fps_pruned_kernel<NWV, true, true>;                                            \
      static bool attr = false;                                                                  \
      if (!attr) {                                                                               \
        (void)hipFuncSetAttribute((const void *)kern,                                            \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160768);           \
        attr = true;                                                                             \
      }                                                                                          \
      hipLaunchKernelGGL(kern, grid, dim3(NWV * 64), lt_bytes, s, n, m, xyz, temp, idx, wp, wo); \
    } else if (lds_ok) {                                                                         \
      auto kern = fps_pruned_kernel<NWV, true, false>;                                           \
      static bool attr = false;                                                                  \
      if (!attr) {                                                                               \
        (void)hipFuncSetAttribute((const void *)kern,                                            \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160768);           \
        attr = true;                                                                             \
      }                                                                                          \
      hipLaunchKernelGGL(kern, grid, dim3(NWV * 64), lt_bytes, s, n, m, xyz, temp, idx, wp, wo); \
    } else {                                                                                     \
      hipLaunchKernelGGL((fps_pruned_kernel<NWV, false, false>), grid, dim3(NWV * 64), 0, s, n, m, xyz, \
                         temp, idx, wp, wo);                                                     \
    }                                                                                            \
  } while (0)
    if (nw == 16) PRUNED(16);
    else if (nw == 8) PRUNED(8);
    else PRUNED(4);
#undef PRUNED
    return check_launch(W);
  }
  if (n <= 64) REG(64, 1);
  else if (n <= 256) REG(256, 1);
  else if (n <= 512) REG(256, 2);
  else if (n <= 1024) REG(256, 4);
  else if (n <= 2048) REG(256, 8);
  else if (n <= 4096) REG(256, 16);
  else if (n <= 8192) STREAM(8);
  else if (n <= 16384) STREAM(16);
  else if (n <= 24576) STREAM(24);
  else if (n <= 32768) STREAM(32);
  else if (n <= 40960) STREAM(40);
  else if (n <= 49152) STREAM(48);
  else if (n <= 65536) STREAM(64);
  else
    hipLaunchKernelGGL((fps_generic_kernel<false>), grid, dim3(1024), 0, s, n, m, L,
                       xyz, temp, idx);
#undef REG
#undef STREAM
  return check_launch(W);
}

extern "C" int nesie_furthest_point_sampling_with_dist_wrapper(
    int b, int n, int m, const float *dist, float *temp, int *idx, void *stream) {
  const char *W = "furthest_point_sampling_with_dist_wrapper";
  NESIE_REQUIRE(b >= 0 && n >= 0 && m >= 0, W);
  if (b == 0 || m <= 0) return NESIE_OK;
  NESIE_REQUIRE(n >= 1 && dist && temp && idx, W);
  const int L = fps_ref_log2_block(n);
  hipLaunchKernelGGL((fps_generic_kernel<true>), dim3(b), dim3(1024), 0,
                     (hipStream_t)stream, n, m, L, dist, temp, idx);
  return check_launch(W);
}
