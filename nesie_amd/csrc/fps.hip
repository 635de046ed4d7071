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

namespace nesie {

__host__ __device__ inline int fps_ref_log2_block(int n) {
  // reference opt_n_threads(): largest power of two <= n, capped at 1024.
  int l = 0;
  while ((2 << l) <= n && l < 10) ++l;
  return l;
}

__device__ __forceinline__ unsigned key_lo_of(int k, int L) {
  unsigned t = (unsigned)k & ((1u << L) - 1u);
  unsigned q = (unsigned)k >> L;
  unsigned rb = L == 0 ? 0u : (__brev(t) >> (32 - L));
  return 0xFFFFFFFFu - ((rb << 22) | q);
}

__device__ __forceinline__ int k_of_key_lo(unsigned lo, int L) {
  unsigned v = 0xFFFFFFFFu - lo;
  unsigned rb = v >> 22, q = v & 0x3FFFFFu;
  unsigned t = L == 0 ? 0u : (__brev(rb) >> (32 - L));
  return (int)((q << L) | t);
}

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

// ---- n <= BLOCK*PPT, everything in registers --------------------------------
template <int BLOCK, int PPT>
__global__ __launch_bounds__(BLOCK) void fps_reg_kernel(
    int n, int m, int L, const float *__restrict__ xyz, float *__restrict__ temp,
    int *__restrict__ idx) {
  constexpr int NW = BLOCK / 64;
  __shared__ unsigned long long red[2][NW];
  const int tid = threadIdx.x;
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
  }
  int old = 0;
  if (tid == 0) idx[0] = 0;
  for (int r = 1; r < m; ++r) {
    const float x1 = xyz[old * 3 + 0], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2];
    unsigned long long best = 0ull;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      float d = sqdist_nofma(px[j] - x1, py[j] - y1, pz[j] - z1);
      float d2 = fminf(d, tp[j]);
      tp[j] = d2;
      unsigned long long key =
          ((unsigned long long)__float_as_uint(d2) << 32) | klo[j];
      key = klo[j] ? key : 0ull;
      best = key > best ? key : best;
    }
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
template <bool WITH_DIST>
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
      else d = sqdist_nofma(src[k * 3 + 0] - x1, src[k * 3 + 1] - y1, src[k * 3 + 2] - z1);
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

}  // namespace nesie

using namespace nesie;

extern "C" int nesie_furthest_point_sampling_wrapper(int b, int n, int m,
                                                     const float *xyz, float *temp,
                                                     int *idx, void *stream) {
  const char *W = "furthest_point_sampling_wrapper";
  NESIE_REQUIRE(b >= 0 && n >= 0 && m >= 0, W);
  if (b == 0 || m <= 0) return NESIE_OK;  // reference kernel returns when m <= 0
  NESIE_REQUIRE(n >= 1 && xyz && temp && idx, W);
  NESIE_REQUIRE((long long)n * 3 < (1ll << 31), W);
  hipStream_t s = (hipStream_t)stream;
  const int L = fps_ref_log2_block(n);
  dim3 grid(b);
#define REG(BLK, P) \
  hipLaunchKernelGGL((fps_reg_kernel<BLK, P>), grid, dim3(BLK), 0, s, n, m, L, xyz, temp, idx)
#define STREAM(P) \
  hipLaunchKernelGGL((fps_stream_kernel<P>), grid, dim3(1024), 0, s, n, m, xyz, temp, idx)
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
