// Standalone bench + check of nesie_pw_layer_forward against rocBLAS and an fp64 host reference.
//   hipcc --offload-arch=gfx950 -O3 pwbench.hip -o pwbench -L../../nesie_amd -lnesie_hip -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

extern "C" {
int nesie_pw_layer_forward(int nb, int ng, int k, int cout, long long p, const float *x,
                           long long x_bstride, const float *w, long long w_gstride, int w_rstride,
                           int w_cstride, const float *in_coef, int in_relu, const float *row_bias,
                           int rb_group, const float *bias, float *y, long long y_bstride,
                           float *stat_part, int pool_group, int pool_min, float *pool_max_out,
                           float *pool_min_out, uint8_t *arg_max_out, uint8_t *arg_min_out,
                           void *stream);
int nesie_pw_stat_slots(int nb, int ng, int k, int cout, long long p);
int nesie_pw_stats_finalize(int channels, int cout, int nslots, const float *stat_part,
                            const float *gamma, const float *beta, float *running_mean,
                            float *running_var, float momentum, float eps, float *coef,
                            const float *chan_bias, void *stream);
const char *nesie_last_error(void);
}
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
// matrix-pipe ceiling at the clock the chip holds under an fp32-MFMA load: register operands only
template <int SHAPE, int CHAINS>
__global__ __launch_bounds__(512) void mfma_peak_kernel(float *out, int iters, float a0, float b0) {
  float a = a0 + threadIdx.x * 1e-6f, b = b0 + threadIdx.x * 1e-6f;
  if constexpr (SHAPE == 16) {
    f32x4_t acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  } else {
    f32x16_t acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  }
}

// Co-issue probe: waves 0-3 (one per SIMD) stream fp32 MFMAs; waves 4-7 (their SIMD partners)
// run `what` (0 nothing, 1 v_fma chain-free VALU, 2 global loads, 3 SALU, 4 LDS reads,
// 5 global stores).  Durations by s_memtime.
template <int MF>
__global__ __launch_bounds__(512) void coissue_kernel(float *buf, long long *times, int what, int iters, int prio) {
  __shared__ float sh[4096];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  sh[threadIdx.x] = threadIdx.x; sh[threadIdx.x + 512] = 1.f;
  __syncthreads();
  long long t0 = 0, t1 = 0;
  float sink = 0.f;
  if (wave < 4) {
    f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    f32x16_t c0; for (int r = 0; r < 16; ++r) c0[r] = 0.f;
    float a = 1.f + lane, b = 2.f;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (MF == 16) { a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a1, 0, 0, 0); }
        else { c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0); }
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
    sink = a0[0] + a1[0] + c0[0];
  } else if (what) {
    if (prio) __builtin_amdgcn_s_setprio(3);
    float x0 = lane, x1 = 1.f, x2 = 2.f, x3 = 3.f;
    int si = blockIdx.x;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
      if (what == 1) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { x0 = x0 * 1.0001f + 0.5f; x1 = x1 * 1.0001f + 0.5f; x2 = x2 * 1.0001f + 0.5f; x3 = x3 * 1.0001f + 0.5f; }
      } else if (what == 2) {
#pragma unroll
        for (int u = 0; u < 4; ++u) x0 += buf[((size_t)(i * 4 + u) * 4096 + blockIdx.x * 64 + lane) & 0xFFFFFF];
      } else if (what == 3) {
#pragma unroll
        for (int u = 0; u < 64; ++u) { si = si * 3 + 1; asm volatile("" : "+s"(si)); }
      } else if (what == 4) {
#pragma unroll
        for (int u = 0; u < 8; ++u) x0 += sh[(lane + u * 64 + i) & 1023];
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) buf[(size_t)(1 << 24) + ((size_t)(i * 4 + u) * 4096 + blockIdx.x * 64 + lane)] = x0;
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
    sink = x0 + x1 + x2 + x3 + si;
  }
  if (lane == 0) { times[(blockIdx.x * 8 + wave) * 2] = t0; times[(blockIdx.x * 8 + wave) * 2 + 1] = t1; }
  if (sink == 123.456f) buf[0] = sink;
}

// Own-stream probe: ONE wave per SIMD; per MFMA, NV independent v_fma's + NS SALU + NL LDS reads
// + NG global stores are issued behind it.  cycles per MFMA by s_memtime.
template <int MF, int NV, int NS, int NL, int NG>
__global__ __launch_bounds__(256) void ownstream_kernel(float *buf, long long *times, int iters) {
  __shared__ float sh[4096];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += 256) sh[i] = i;
  __syncthreads();
  f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
  f32x16_t c0; for (int r = 0; r < 16; ++r) c0[r] = 0.f;
  float a = 1.f + lane, b = 2.f, x[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
  int si = blockIdx.x;
  float *dst = buf + ((size_t)blockIdx.x * 4 + wave) * 64 + lane;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MF == 16) { if (u & 1) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a1, 0, 0, 0); else a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a0, 0, 0, 0); }
      else c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < NV; ++v) x[v & 7] = x[v & 7] * 1.0001f + 0.5f;
#pragma unroll
      for (int v = 0; v < NS; ++v) { si = si * 3 + 1; asm volatile("" : "+s"(si)); }
#pragma unroll
      for (int v = 0; v < NL; ++v) x[v & 7] += sh[(lane + (u * 4 + v) * 64 + i) & 4095];
#pragma unroll
      for (int v = 0; v < NG; ++v) dst[(size_t)((i * 8 + u) * NG + v) * 65536 & 0xFFFFFF] = x[v & 7];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float sink = a0[0] + a1[0] + c0[0] + si;
  for (int v = 0; v < 8; ++v) sink += x[v];
  if (lane == 0) { times[(blockIdx.x * 4 + wave) * 2] = t0; times[(blockIdx.x * 4 + wave) * 2 + 1] = t1; }
  if (sink == 123.456f) buf[0] = sink;
}

// Issue-cost probe: one wave per SIMD, per MFMA one extra instruction of kind WHAT whose result is
// not waited for inside the loop (pure issue cost).  1 global_load_dwordx4, 2 ds_write_b128,
// 3 global_load_lds_dwordx4, 4 global_store_dwordx4, 5 global_store_dword, 6 ds_read2st64_b32,
// 7 ds_read_b128, 8 v_pk_fma_f32 x2, 9 s_waitcnt lgkmcnt(15)
template <int WHAT, int EVERY = 1>
__global__ __launch_bounds__(256) void issue_kernel(float *buf, long long *times, int iters) {
  extern __shared__ __attribute__((aligned(16))) float shd[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 8192; i += 256) shd[i] = i;
  __syncthreads();
  f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
  float a = 1.f + lane, b = 2.f;
  typedef float v4 __attribute__((ext_vector_type(4)));
  typedef float v2 __attribute__((ext_vector_type(2)));
  v4 ld[8]; v2 r2 = {1.f, 2.f}, pk = {1.f, 1.f};
  for (int u = 0; u < 8; ++u) ld[u] = (v4){1.f, 2.f, 3.f, 4.f};
  const float *src = buf + ((size_t)blockIdx.x * 4 + wave) * 4096;
  float *dst = buf + (size_t)(1 << 24) + ((size_t)blockIdx.x * 4 + wave) * 4096;
  const unsigned lw = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)shd + wave * 4096;
  const unsigned la = lw + lane * 16;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u & 1) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a1, 0, 0, 0); else a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a0, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      const unsigned go = (unsigned)(((i * 8 + u) & 7) * 1024 + lane * 16);
      if (u % EVERY != 0) continue;
      if (WHAT == 1) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ld[u]) : "v"(go), "s"(src) : "memory");
      if (WHAT == 2) asm volatile("ds_write_b128 %0, %1" :: "v"(la), "v"(ld[u]) : "memory");
      if (WHAT == 3) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(go), "s"(src), "s"(lw) : "memory", "m0");
      if (WHAT == 4) asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(go), "v"(ld[u]), "s"(dst) : "memory");
      if (WHAT == 5) asm volatile("global_store_dword %0, %1, %2" :: "v"(go), "v"(a), "s"(dst) : "memory");
      if (WHAT == 6) asm volatile("ds_read2st64_b32 %0, %1 offset0:0 offset1:4" : "=v"(r2) : "v"(la));
      if (WHAT == 7) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[u]) : "v"(la));
      if (WHAT == 8) { pk = __builtin_elementwise_fma(pk, r2, r2); r2 = __builtin_elementwise_fma(r2, pk, pk); }
      if (WHAT == 9) asm volatile("s_waitcnt lgkmcnt(15)");
      // conflict-free address patterns (lane * 4 / lane * 8): WHAT 6's lane * 16 is a 4-way bank conflict for 4-byte reads
      if (WHAT == 10) asm volatile("ds_read2st64_b32 %0, %1 offset0:0 offset1:4" : "=v"(r2) : "v"(lw + lane * 4));
      if (WHAT == 11) asm volatile("ds_read_b64 %0, %1" : "=v"(r2) : "v"(lw + lane * 8));
      if (WHAT == 12) asm volatile("ds_read_b32 %0, %1" : "=v"(r2[0]) : "v"(lw + lane * 4));
      if (WHAT == 13) asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:64" : "=v"(ld[u]) : "v"(lw + lane * 8));
      __builtin_amdgcn_sched_barrier(0);
    }
    if (WHAT == 1 || WHAT == 3 || WHAT == 4 || WHAT == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (WHAT == 2 || WHAT == 6 || WHAT == 7 || WHAT >= 10) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float sink = a0[0] + a1[0] + r2[0] + pk[0];
  for (int u = 0; u < 8; ++u) sink += ld[u][0];
  if (lane == 0) { times[(blockIdx.x * 4 + wave) * 2] = t0; times[(blockIdx.x * 4 + wave) * 2 + 1] = t1; }
  if (sink == 123.456f) buf[0] = sink;
}

// Pipeline-sharing probe (mode 14): every wave runs T "tiles" = NM fp32 MFMAs on register operands
// followed by NS 16-byte-per-lane stores of its accumulators (no LDS, no barrier: the waves are
// independent pipelines).  How do two such waves share a SIMD?  POLICY: 0 equal priority,
// 1 waves 4-7 at s_setprio 1 throughout, 2 priority 3 around the stores, 3 priority 3 around the
// MFMAs, 4 waves 4-7 start half a tile late (one epilogue + half a loop of s_sleep), 5 = 2 + 4.
// active = waves per workgroup that work (4: one per SIMD, 8: two per SIMD).
template <int NM, int NS, int POLICY>
__global__ __launch_bounds__(512) void pipe_probe_kernel(float *buf, long long *times, int tiles, int active) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wave >= active) return;
  f32x4_t acc[8];
  for (int c = 0; c < 8; ++c) acc[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float a = 1.f + lane * 1e-3f, b = 2.f;
  float *dst = buf + ((size_t)blockIdx.x * 8 + wave) * (size_t)tiles * NS * 256 + lane * 4;
  if (POLICY == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
  if ((POLICY == 4 || POLICY == 5) && wave >= 4)
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_s_sleep(127);       // ~ 4 x 8k cycles? (64 clocks per unit)
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < tiles; ++t) {
    if (POLICY == 3) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int i = 0; i < NM / 8; ++i)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    if (POLICY == 3) __builtin_amdgcn_s_setprio(0);
    if (POLICY == 2 || POLICY == 5) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int i = 0; i < NS; ++i) *(f32x4_t *)(dst + ((size_t)t * NS + i) * 256) = acc[i % 8];
    if (POLICY == 2 || POLICY == 5) __builtin_amdgcn_s_setprio(0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { times[(blockIdx.x * 8 + wave) * 2] = t0; times[(blockIdx.x * 8 + wave) * 2 + 1] = t1; }
}

// The same probe with the layer kernel's OUTPUT ADDRESSING (mode 15): a 128-row x P tensor per batch,
// a tile = 64 positions; lane (l16, quad) of wave w stores rows 32 w + l16 (+ 16), 16 bytes at
// position 4 quad (+ 16 r), i.e. every store instruction writes 16 rows x 64 contiguous bytes, rows
// P * 4 bytes apart.  LOADS > 0: also LOADS 16-byte-per-lane loads per tile in the operand pattern
// (rows P * 4 bytes apart, 256 contiguous bytes per row), consumed by a dummy ds_write.
// LSPREAD: the loads are issued one per NM / 8 / LOADS groups of 8 MFMAs instead of in one burst at
// the top; SSPREAD: the previous tile's stores (from a copy of the accumulators) are issued one per
// group inside the MFMA loop instead of in one burst behind it.
template <int NM, int LOADS, bool LSPREAD, bool SSPREAD>
__global__ __launch_bounds__(256) void pipe_addr_kernel(float *out, const float *in, long long *times, int tiles, int P, int ntile_total) {
  __shared__ float sink[4096];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int l16 = lane & 15, quad = lane >> 4;
  f32x4_t acc[8], old[8];
  for (int c = 0; c < 8; ++c) acc[c] = old[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float a = 1.f + lane * 1e-3f, b = 2.f;
  const int tpb = P / 64;
  f32x4_t ld[LOADS > 0 ? LOADS : 1];
  float *dprev = out;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < tiles; ++t) {
    const int tix = (blockIdx.x + t * gridDim.x) % ntile_total;
    const int n = tix / tpb, col = tix % tpb;
    const float *src = in + (size_t)n * 128 * P + (size_t)col * 64;
    float *dst = out + (size_t)n * 128 * P + (size_t)col * 64;
    if (!LSPREAD) {
#pragma unroll
      for (int i = 0; i < LOADS; ++i)   // thread (row = 16 i + tid / 16, 16-byte column tid % 16)
        ld[i] = *(const f32x4_t *)(src + (size_t)(16 * i + (threadIdx.x >> 4)) * P + (threadIdx.x & 15) * 4);
    }
#pragma unroll
    for (int i = 0; i < NM / 8; ++i) {
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (LSPREAD && LOADS > 0 && i % (NM / 8 / (LOADS > 0 ? LOADS : 1)) == 0 && i / (NM / 8 / (LOADS > 0 ? LOADS : 1)) < LOADS) {
        const int li = i / (NM / 8 / (LOADS > 0 ? LOADS : 1));
        ld[li] = *(const f32x4_t *)(src + (size_t)(16 * li + (threadIdx.x >> 4)) * P + (threadIdx.x & 15) * 4);
      }
      if (SSPREAD && i % 4 == 2 && i / 4 < 8) {
        const int si = i / 4;
        *(f32x4_t *)(dprev + (size_t)(32 * wave + 16 * (si / 4) + l16) * P + 4 * quad + 16 * (si % 4)) = old[si];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < LOADS; ++i) *(f32x4_t *)&sink[((threadIdx.x * 4 + i * 1024) & 4095)] = ld[i];
    if (SSPREAD) {
#pragma unroll
      for (int c = 0; c < 8; ++c) old[c] = acc[c];
      dprev = dst;
    } else {
#pragma unroll
      for (int rw = 0; rw < 2; ++rw)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *(f32x4_t *)(dst + (size_t)(32 * wave + 16 * rw + l16) * P + 4 * quad + 16 * r) = acc[rw * 4 + r];
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { times[(blockIdx.x * 4 + wave) * 2] = t0; times[(blockIdx.x * 4 + wave) * 2 + 1] = t1; }
  if (sink[lane] == 123.456f) out[0] = old[0][0];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static uint32_t rng = 12345;
static float frand() { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) & 0xFFFFFF) / 8388608.0f - 1.0f; }

struct Shape { int nb, ng, k, cout; long long p; int g; const char *what; };

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
  const int only = argc > 2 ? atoi(argv[2]) : -1;   // shape index, -1 = all
  const int mode = argc > 3 ? atoi(argv[3]) : 0;    // 1: plain kernel only, 2: fused only, 3: rocBLAS only (profiling)
  rocblas_handle h; rocblas_create_handle(&h);
  std::vector<Shape> shapes = {
      {48, 6, 256, 128, 8192, 16, "MiniPointNet side 256->128 (+pool16)"},
      {48, 6, 128, 256, 8192, 16, "MiniPointNet side 128->256 (+rowbias)"},
      {8, 1, 256, 128, 32768, 64, "MiniPointNet box 256->128"},
      {8, 1, 128, 256, 32768, 64, "MiniPointNet box 128->256"},
      {8, 1, 64, 64, 131072, 64, "SA1 64->64"},
      {8, 1, 64, 128, 131072, 64, "SA1 64->128"},
      {8, 1, 131, 128, 32768, 32, "SA2 131->128"},
      {8, 1, 128, 128, 32768, 32, "SA2 128->128"},
      {8, 1, 128, 256, 32768, 32, "SA2 128->256"},
      {8, 1, 259, 128, 8192, 16, "SA3 259->128"},
      {8, 1, 128, 256, 8192, 16, "SA3 128->256"},
      {8, 1, 128, 256, 4096, 16, "SA4 128->256"},
      {8, 1, 256, 256, 1024, 16, "FP 256->256"},
  };
  if (const char *e = getenv("PW_SHAPE")) {   // PW_SHAPE="nb ng k cout p g": one custom shape, index 0
    Shape c{}; long long pp = 0;
    if (sscanf(e, "%d %d %d %d %lld %d", &c.nb, &c.ng, &c.k, &c.cout, &pp, &c.g) == 6) { c.p = pp; c.what = "custom"; shapes.insert(shapes.begin(), c); }
  }
  if (mode == 14) {
    const int tiles = 24;
    float *buf; long long *tm;
    CK(hipMalloc(&buf, (size_t)256 * 8 * tiles * 8 * 256 * 4 + 4096)); CK(hipMalloc(&tm, 256 * 16 * 8));
    auto run = [&](auto kern, int active, const char *nm) {
      CK(hipMemset(tm, 0, 256 * 16 * 8));
      for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, buf, tm, tiles, active);
      CK(hipDeviceSynchronize());
      std::vector<long long> h(256 * 16);
      CK(hipMemcpy(h.data(), tm, h.size() * 8, hipMemcpyDeviceToHost));
      // workgroup 7: waves 0 and 4 (SIMD partners); span = first start to last end over its waves
      const long long *q = &h[7 * 16];
      long long lo = q[0], hi = q[1];
      for (int w = 0; w < active; ++w) { if (q[2 * w] < lo) lo = q[2 * w]; if (q[2 * w + 1] > hi) hi = q[2 * w + 1]; }
      const double mf = (double)tiles * 256 * 32 * (active / 4);   // MFMA cycles per SIMD
      printf("%-58s %d waves: wave0 %7lld cyc, wave4 %7lld cyc, span %7lld = %.0f per tile and SIMD-wave, matrix pipe %.0f %%\n", nm, active,
             q[1] - q[0], active > 4 ? q[9] - q[8] : 0ll, hi - lo, (double)(hi - lo) / tiles / (active / 4), 100.0 * mf / (hi - lo));
    };
    run(pipe_probe_kernel<256, 8, 0>, 4, "256 MFMA + 8 stores");
    run(pipe_probe_kernel<256, 0, 0>, 4, "256 MFMA + 0 stores");
    run(pipe_probe_kernel<256, 8, 0>, 8, "256 MFMA + 8 stores, equal priority");
    run(pipe_probe_kernel<256, 0, 0>, 8, "256 MFMA + 0 stores, equal priority");
    run(pipe_probe_kernel<256, 8, 1>, 8, "256 MFMA + 8 stores, waves 4-7 at priority 1");
    run(pipe_probe_kernel<256, 8, 2>, 8, "256 MFMA + 8 stores, priority 3 around the stores");
    run(pipe_probe_kernel<256, 8, 3>, 8, "256 MFMA + 8 stores, priority 3 around the MFMAs");
    run(pipe_probe_kernel<256, 8, 4>, 8, "256 MFMA + 8 stores, waves 4-7 start late");
    run(pipe_probe_kernel<256, 8, 5>, 8, "256 MFMA + 8 stores, late start + priority 3 around stores");
    run(pipe_probe_kernel<128, 4, 0>, 8, "128 MFMA + 4 stores, equal priority");
    run(pipe_probe_kernel<128, 4, 2>, 8, "128 MFMA + 4 stores, priority 3 around the stores");
    return 0;
  }
  if (mode == 15) {
    const int P = 8192, nb = 48, tiles = 24;
    float *out, *in; long long *tm;
    CK(hipMalloc(&out, (size_t)nb * 128 * P * 4)); CK(hipMalloc(&in, (size_t)nb * 128 * P * 4)); CK(hipMalloc(&tm, 256 * 8 * 8));
    CK(hipMemset(in, 0, (size_t)nb * 128 * P * 4));
    auto run = [&](auto kern, const char *nm) {
      for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, in, tm, tiles, P, nb * P / 64);
      CK(hipDeviceSynchronize());
      std::vector<long long> h(256 * 8);
      CK(hipMemcpy(h.data(), tm, h.size() * 8, hipMemcpyDeviceToHost));
      printf("%-50s wave0 %7lld cyc = %.0f per tile (256 MFMAs = 8192)\n", nm, h[7 * 8 + 1] - h[7 * 8], (double)(h[7 * 8 + 1] - h[7 * 8]) / tiles);
    };
    run(pipe_addr_kernel<256, 0, false, false>, "8 stores (burst), no loads");
    run(pipe_addr_kernel<256, 8, false, false>, "8 loads (burst at top) + 8 stores (burst)");
    run(pipe_addr_kernel<256, 8, true, false>, "8 loads (spread) + 8 stores (burst)");
    run(pipe_addr_kernel<256, 8, false, true>, "8 loads (burst at top) + 8 stores (spread, next tile)");
    run(pipe_addr_kernel<256, 8, true, true>, "8 loads (spread) + 8 stores (spread, next tile)");
    run(pipe_addr_kernel<256, 0, false, true>, "no loads, 8 stores (spread, next tile)");
    run(pipe_addr_kernel<256, 8, true, false>, "again: 8 loads (spread) + 8 stores (burst)");
    return 0;
  }
  if (mode == 12) {
    float *buf; long long *tm; CK(hipMalloc(&buf, (size_t)(1 << 24) * 4 * 2)); CK(hipMalloc(&tm, 256 * 8 * 8));
    CK(hipMemset(buf, 0, (size_t)(1 << 24) * 4 * 2));
    const int it = 64;
    auto run = [&](auto kern, const char *nm) {
      for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 65536, 0, buf, tm, it);
      CK(hipDeviceSynchronize());
      std::vector<long long> h(256 * 8);
      CK(hipMemcpy(h.data(), tm, h.size() * 8, hipMemcpyDeviceToHost));
      printf("16x16x4 + %-40s %.1f cycles per MFMA (incl. one drain per 8)\n", nm, (double)(h[7 * 8 + 1] - h[7 * 8]) / (it * 8));
    };
    run(issue_kernel<0>, "nothing");
    run(issue_kernel<1>, "global_load_dwordx4 (saddr)");
    run(issue_kernel<2>, "ds_write_b128");
    run(issue_kernel<3>, "s_mov m0 + global_load_lds_dwordx4");
    run(issue_kernel<4>, "global_store_dwordx4 (saddr)");
    run(issue_kernel<5>, "global_store_dword (saddr)");
    run(issue_kernel<6>, "ds_read2st64_b32");
    run(issue_kernel<7>, "ds_read_b128");
    run(issue_kernel<8>, "2 x v_pk_fma_f32");
    run(issue_kernel<9>, "s_waitcnt lgkmcnt(15)");
    run(issue_kernel<1, 8>, "global_load_dwordx4, one per 8 MFMAs");
    run(issue_kernel<4, 8>, "global_store_dwordx4, one per 8 MFMAs");
    run(issue_kernel<4, 4>, "global_store_dwordx4, one per 4 MFMAs");
    run(issue_kernel<5, 8>, "global_store_dword, one per 8 MFMAs");
    run(issue_kernel<2, 8>, "ds_write_b128, one per 8 MFMAs");
    run(issue_kernel<2, 4>, "ds_write_b128, one per 4 MFMAs");
    run(issue_kernel<10>, "ds_read2st64_b32 conflict-free");
    run(issue_kernel<11>, "ds_read_b64 conflict-free");
    run(issue_kernel<12>, "ds_read_b32 conflict-free");
    run(issue_kernel<13>, "ds_read2_b64 conflict-free");
    return 0;
  }
  if (mode == 11) {
    float *buf; long long *tm; CK(hipMalloc(&buf, (size_t)(1 << 24) * 4 * 2)); CK(hipMalloc(&tm, 256 * 8 * 8));
    const int it = 128;
    auto run = [&](auto kern, const char *nm) {
      hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, buf, tm, it);
      CK(hipDeviceSynchronize());
      std::vector<long long> h(256 * 8);
      CK(hipMemcpy(h.data(), tm, h.size() * 8, hipMemcpyDeviceToHost));
      printf("%-44s %.1f cycles per MFMA\n", nm, (double)(h[7 * 8 + 1] - h[7 * 8]) / (it * 8));
    };
    run(ownstream_kernel<16, 0, 0, 0, 0>, "16x16x4 alone");
    run(ownstream_kernel<16, 2, 0, 0, 0>, "16x16x4 + 2 VALU (v_mul, v_add)");
    run(ownstream_kernel<16, 4, 0, 0, 0>, "16x16x4 + 4 VALU");
    run(ownstream_kernel<16, 8, 0, 0, 0>, "16x16x4 + 8 VALU");
    run(ownstream_kernel<16, 16, 0, 0, 0>, "16x16x4 + 16 VALU");
    run(ownstream_kernel<16, 0, 8, 0, 0>, "16x16x4 + 8 SALU");
    run(ownstream_kernel<16, 0, 0, 1, 0>, "16x16x4 + 1 LDS read");
    run(ownstream_kernel<16, 0, 0, 2, 0>, "16x16x4 + 2 LDS read");
    run(ownstream_kernel<16, 0, 0, 0, 1>, "16x16x4 + 1 global store");
    run(ownstream_kernel<16, 4, 4, 1, 1>, "16x16x4 + 4 VALU 4 SALU 1 LDS 1 store");
    run(ownstream_kernel<32, 0, 0, 0, 0>, "32x32x2 alone");
    run(ownstream_kernel<32, 4, 0, 0, 0>, "32x32x2 + 4 VALU");
    run(ownstream_kernel<32, 8, 0, 0, 0>, "32x32x2 + 8 VALU");
    run(ownstream_kernel<32, 16, 0, 0, 0>, "32x32x2 + 16 VALU");
    run(ownstream_kernel<32, 32, 0, 0, 0>, "32x32x2 + 32 VALU");
    run(ownstream_kernel<32, 0, 16, 0, 0>, "32x32x2 + 16 SALU");
    run(ownstream_kernel<32, 0, 0, 2, 0>, "32x32x2 + 2 LDS read");
    run(ownstream_kernel<32, 0, 0, 0, 2>, "32x32x2 + 2 global store");
    run(ownstream_kernel<32, 8, 8, 2, 2>, "32x32x2 + 8 VALU 8 SALU 2 LDS 2 store");
    return 0;
  }
  if (mode == 10) {
    float *buf; long long *tm; CK(hipMalloc(&buf, (size_t)(1 << 24) * 4 * 3)); CK(hipMalloc(&tm, 256 * 16 * 8));
    CK(hipMemset(buf, 0, (size_t)(1 << 24) * 4));
    const char *nm[6] = {"nothing", "VALU (64 v_mul+v_add / iter)", "global loads (4 / iter)", "SALU (64 / iter)", "LDS reads (8 / iter)", "global stores (4 / iter)"};
    for (int mf = 16; mf <= 32; mf += 16)
      for (int what = 0; what < 6; ++what)
        for (int prio = 0; prio < (what ? 2 : 1); ++prio) {
          const int it = 256;
          if (mf == 16) hipLaunchKernelGGL(coissue_kernel<16>, dim3(256), dim3(512), 0, 0, buf, tm, what, it, prio);
          else hipLaunchKernelGGL(coissue_kernel<32>, dim3(256), dim3(512), 0, 0, buf, tm, what, it, prio);
          CK(hipDeviceSynchronize());
          std::vector<long long> h(256 * 16);
          CK(hipMemcpy(h.data(), tm, h.size() * 8, hipMemcpyDeviceToHost));
          // workgroup 7: MFMA wave 0 and its partner wave 4
          const long long *q = &h[7 * 16];
          printf("mfma %dx: partner runs %-30s prio %d: MFMA wave %7lld cyc (%.1f per MFMA)  partner %7lld cyc  overlap: partner starts %+lld, ends %+lld vs MFMA end\n",
                 mf, nm[what], prio, q[1] - q[0], (double)(q[1] - q[0]) / (it * (mf == 16 ? 16 : 8)), q[9] - q[8], q[8] - q[0], q[9] - q[1]);
        }
    return 0;
  }
  if (mode == 9) {
    float *o; CK(hipMalloc(&o, 1024 * 512 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int it = 2000;
    auto run = [&](auto kern, int threads, int chains, double flop_per_mfma, const char *nm) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, o, it, 0.5f, 0.25f);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double fl = 256.0 * (threads / 64) * it * 16.0 * chains * flop_per_mfma;
        if (rep) printf("%-28s %d waves/CU: %.3f ms  %.1f TFLOP/s\n", nm, threads / 64, ms, fl / ms / 1e9);
      }
    };
    run(mfma_peak_kernel<16, 2>, 256, 2, 2048.0, "16x16x4 f32, 2 chains");
    run(mfma_peak_kernel<16, 2>, 512, 2, 2048.0, "16x16x4 f32, 2 chains");
    run(mfma_peak_kernel<16, 1>, 512, 1, 2048.0, "16x16x4 f32, 1 chain");
    run(mfma_peak_kernel<16, 4>, 256, 4, 2048.0, "16x16x4 f32, 4 chains");
    run(mfma_peak_kernel<32, 1>, 256, 1, 4096.0, "32x32x2 f32, 1 chain");
    run(mfma_peak_kernel<32, 1>, 512, 1, 4096.0, "32x32x2 f32, 1 chain");
    run(mfma_peak_kernel<32, 2>, 256, 2, 4096.0, "32x32x2 f32, 2 chains");
    return 0;
  }
  printf("%-40s %9s %9s %8s %8s %9s %9s %9s\n", "layer", "rocblas", "nesie", "TF(roc)", "TF(own)", "fused ms", "maxerr", "staterr");
  for (size_t si = 0; si < shapes.size(); ++si) {
    const Shape &s = shapes[si];
    if (only >= 0 && (int)si != only) continue;
    const size_t xe = (size_t)s.nb * s.k * s.p, ye = (size_t)s.nb * s.cout * s.p, we = (size_t)s.ng * s.cout * s.k;
    std::vector<float> hx(xe), hw(we), hcoef((size_t)s.ng * s.k * 4);
    for (auto &v : hx) v = frand();
    for (auto &v : hw) v = frand() / sqrtf((float)s.k);
    for (size_t i = 0; i < hcoef.size(); i += 4) { hcoef[i] = 0.5f + 0.5f * fabsf(frand()); hcoef[i + 1] = 0.3f * frand(); if ((i / 4) % 7 == 3) hcoef[i] = -hcoef[i]; }
    float *dx, *dw, *dy, *dy2, *dcoef, *dpart, *dcoef_out, *dpmax, *dpmin; uint8_t *damax, *damin;
    CK(hipMalloc(&dx, xe * 4)); CK(hipMalloc(&dw, we * 4)); CK(hipMalloc(&dy, ye * 4)); CK(hipMalloc(&dy2, ye * 4));
    CK(hipMalloc(&dcoef, hcoef.size() * 4));
    const int slots = nesie_pw_stat_slots(s.nb, s.ng, s.k, s.cout, s.p);
    CK(hipMalloc(&dpart, (size_t)s.ng * slots * s.cout * 16 + 16));
    CK(hipMalloc(&dcoef_out, (size_t)s.ng * s.cout * 16));
    CK(hipMalloc(&dpmax, ye / 16 * 4)); CK(hipMalloc(&dpmin, ye / 16 * 4)); CK(hipMalloc(&damax, ye / 16)); CK(hipMalloc(&damin, ye / 16));
    CK(hipMemcpy(dx, hx.data(), xe * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), we * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcoef, hcoef.data(), hcoef.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto fn) { for (int i = 0; i < 3; ++i) fn(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) fn(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters; };
    const float one = 1.f, zero = 0.f;
    // rocBLAS: per weight group a strided-batched GEMM (column-major view: Y^T = X^T W^T)
    auto roc = [&]() {
      for (int g = 0; g < s.ng; ++g)
        rocblas_sgemm_strided_batched(h, rocblas_operation_none, rocblas_operation_none, (int)s.p, s.cout, s.k, &one,
                                      dx + (size_t)g * s.k * s.p, (int)s.p, (long long)s.ng * s.k * s.p,
                                      dw + (size_t)g * s.cout * s.k, s.k, 0, &zero,
                                      dy2 + (size_t)g * s.cout * s.p, (int)s.p, (long long)s.ng * s.cout * s.p, s.nb / s.ng);
    };
    const long long xbs = getenv("PW_XSTRIDE0") ? 0 : (long long)s.k * s.p;   // every batch reads X[0]: operand from cache
    auto own = [&](const float *coef, int relu, float *y, float *part, int pg, int pmin) {
      static const bool wt = getenv("PW_WT") != nullptr;   // PW_WT: the weights as a transposed view (input-gradient launches)
      int st = nesie_pw_layer_forward(s.nb, s.ng, s.k, s.cout, s.p, dx, xbs, dw, (long long)s.cout * s.k,
                                      wt ? 1 : s.k, wt ? s.cout : 1, coef, relu, nullptr, 0, nullptr, y, (long long)s.cout * s.p, part, pg, pmin,
                                      dpmax, dpmin, damax, damin, 0);
      if (st) { printf("nesie error %d: %s\n", st, nesie_last_error()); exit(1); }
    };
#ifdef PW_STAMP
    if (mode == 7 || mode == 8) {
      extern long long *g_pw_stamps;
      const size_t NST = 512 + 1024 * 4;
      long long *dst; CK(hipMalloc(&dst, NST * 8)); CK(hipMemset(dst, 0, NST * 8));
      for (int rep = 0; rep < 400; ++rep) own(mode == 8 ? dcoef : nullptr, 1, dy, mode == 8 ? dpart : nullptr, 0, 0);
      g_pw_stamps = dst;
      own(mode == 8 ? dcoef : nullptr, 1, dy, mode == 8 ? dpart : nullptr, 0, 0);
      CK(hipDeviceSynchronize());
      g_pw_stamps = nullptr;
      std::vector<long long> hs(NST);
      CK(hipMemcpy(hs.data(), dst, hs.size() * 8, hipMemcpyDeviceToHost));
      {  // per-workgroup wall clock (10 ns ticks), relative to the earliest entry
        long long t0 = -1; int nwgs = 0;
        for (int b = 0; b < 1024; ++b) if (hs[512 + b * 4]) { ++nwgs; if (t0 < 0 || hs[512 + b * 4] < t0) t0 = hs[512 + b * 4]; }
        long long mx[3] = {0, 0, 0}, mn[3] = {1ll << 60, 1ll << 60, 1ll << 60}; double av[3] = {0, 0, 0};
        for (int b = 0; b < 1024; ++b) if (hs[512 + b * 4]) for (int k = 0; k < 3; ++k) {
          const long long v = hs[512 + b * 4 + k] - t0; mx[k] = v > mx[k] ? v : mx[k]; mn[k] = v < mn[k] ? v : mn[k]; av[k] += (double)v / nwgs; }
        if (getenv("PW_XCD_STATS")) {   // loop start / end per XCD (block index mod 8) and per eighth of the grid
          for (int x = 0; x < 8; ++x) {
            double a1 = 0, a2 = 0, m2 = 0; int c = 0;
            for (int b = x; b < 1024; b += 8) if (hs[512 + b * 4]) { a1 += (hs[512 + b * 4 + 1] - t0) * .01; const double e = (hs[512 + b * 4 + 2] - t0) * .01; a2 += e; m2 = e > m2 ? e : m2; ++c; }
            if (c) printf("  xcd %d: %3d workgroups, loop start mean %.2f, loop end mean %.2f max %.2f us\n", x, c, a1 / c, a2 / c, m2);
          }
          for (int part = 0; part < 8; ++part) {
            double a0 = 0, a1 = 0, a2 = 0, m2 = 0; int c = 0;
            for (int b = part * nwgs / 8; b < (part + 1) * nwgs / 8; ++b) if (hs[512 + b * 4]) { a0 += (hs[512 + b * 4] - t0) * .01; a1 += (hs[512 + b * 4 + 1] - t0) * .01; const double e = (hs[512 + b * 4 + 2] - t0) * .01; a2 += e; m2 = e > m2 ? e : m2; ++c; }
            if (c) printf("  blocks %4d..%4d: entry %.2f, loop start %.2f, loop end mean %.2f max %.2f us\n", part * nwgs / 8, (part + 1) * nwgs / 8 - 1, a0 / c, a1 / c, a2 / c, m2);
          }
        }
        printf("%d workgroups; us since the first entry (min / mean / max): entry %.2f %.2f %.2f | loop start %.2f %.2f %.2f | loop end %.2f %.2f %.2f\n", nwgs,
               mn[0] * .01, av[0] * .01, mx[0] * .01, mn[1] * .01, av[1] * .01, mx[1] * .01, mn[2] * .01, av[2] * .01, mx[2] * .01);
      }
      printf("in-kernel clock: %lld shader cycles over %lld ticks of the 100 MHz counter = %.3f GHz\n", hs[2 * 24 * 8 + 2] - hs[2 * 24 * 8],
             hs[2 * 24 * 8 + 3] - hs[2 * 24 * 8 + 1], 0.1 * (double)(hs[2 * 24 * 8 + 2] - hs[2 * 24 * 8]) / (double)(hs[2 * 24 * 8 + 3] - hs[2 * 24 * 8 + 1]));
      const char *nm[8] = {"top", "waited", "barrier", "issued", "xform", "late-epi", "mfma", "epi"};
      for (int h = 0; h < 2; ++h) {
        printf("%s wave %d: per-iteration phase durations (cycles): wait barrier issue xform late-epi mfma epi | total\n", s.what, h * 4);
        for (int it = 0; it < 23; ++it) {
          const long long *q = &hs[(h * 24 + it) * 8], *qn = &hs[(h * 24 + it + 1) * 8];
          if (!q[0] || !qn[0]) break;
          printf("  it %2d:", it);
          for (int k = 1; k < 8; ++k) printf(" %6lld", q[k] - q[k - 1]);
          printf(" | %6lld  (%s)\n", qn[0] - q[0], nm[0]);
        }
      }
      continue;
    }
#endif
    if (mode == 1) { printf("%s plain %.4f ms\n", s.what, timeit([&]() { own(nullptr, 0, dy, nullptr, 0, 0); })); continue; }
    if (mode == 13) { printf("%s pool-only (no Y store) %.4f ms\n", s.what, timeit([&]() { own(dcoef, 1, nullptr, nullptr, 16, 0); })); continue; }
    if (mode == 2) { printf("%s fused %.4f ms\n", s.what, timeit([&]() { own(dcoef, 1, dy, dpart, 0, 0); })); continue; }
    if (mode == 3) { printf("%s rocblas %.4f ms\n", s.what, timeit(roc)); continue; }
    const float t_roc = timeit(roc);
    const float t_own = timeit([&]() { own(nullptr, 0, dy, nullptr, 0, 0); });
    // correctness of the plain product vs rocBLAS and fp64 samples
    std::vector<float> hy(ye), hy2(ye);
    CK(hipMemcpy(hy.data(), dy, ye * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hy2.data(), dy2, ye * 4, hipMemcpyDeviceToHost));
    double maxerr = 0.0;
    for (int it = 0; it < 4000; ++it) {
      rng = rng * 1664525u + 1013904223u; const int n = rng % s.nb;
      rng = rng * 1664525u + 1013904223u; const int m = rng % s.cout;
      rng = rng * 1664525u + 1013904223u; const long long q = it < 64 ? (it < 32 ? it : s.p - 1 - (it - 32)) : rng % s.p;
      double ref = 0.0;
      for (int kk = 0; kk < s.k; ++kk) ref += (double)hw[((size_t)(n % s.ng) * s.cout + m) * s.k + kk] * hx[((size_t)n * s.k + kk) * s.p + q];
      const double e = fabs(ref - hy[((size_t)n * s.cout + m) * s.p + q]);
      if (e > maxerr) maxerr = e;
    }
    double maxd = 0.0;
    for (size_t i = 0; i < ye; ++i) { const double d = fabs((double)hy[i] - hy2[i]); if (d > maxd) maxd = d; }
    if (maxd > 1e-3) {
      size_t bad = 0, hp[16] = {0}, hm[16] = {0}, hn[64] = {0}; int shown = 0;
      for (size_t i = 0; i < ye; ++i) if (fabs((double)hy[i] - hy2[i]) > 1e-3) {
        const long long q = i % s.p; const int mm = (i / s.p) % s.cout, n = i / s.p / s.cout;
        ++bad; ++hp[(q / 16) % 16]; ++hm[mm % 16]; ++hn[n % 64];
        if (shown++ < 6) printf("   mismatch n %d m %d pos %lld: %g vs %g\n", n, mm, q, hy[i], hy2[i]);
      }
      printf("   %zu of %zu differ; by (pos/16)%%16:", bad, ye); for (int i = 0; i < 16; ++i) printf(" %zu", hp[i]);
      printf("\n   by m%%16:"); for (int i = 0; i < 16; ++i) printf(" %zu", hm[i]);
      printf("\n   by n:"); for (int i = 0; i < 16; ++i) printf(" %zu", hn[i]); printf("\n");
    }
    // fused: affine + relu prologue, store + stats epilogue
    const float t_fused = timeit([&]() { own(dcoef, 1, dy, dpart, 0, 0); });
    nesie_pw_stats_finalize(s.ng * s.cout, s.cout, slots, dpart, nullptr, nullptr, nullptr, nullptr, 0.1f, 1e-5f, dcoef_out, nullptr, 0);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hy.data(), dy, ye * 4, hipMemcpyDeviceToHost));
    std::vector<float> hco((size_t)s.ng * s.cout * 4);
    CK(hipMemcpy(hco.data(), dcoef_out, hco.size() * 4, hipMemcpyDeviceToHost));
    // statistics of the stored y (fp64) vs finalize output, and fp64 samples of the fused product
    double staterr = 0.0;
    for (int g = 0; g < s.ng; ++g)
      for (int m = 0; m < s.cout; m += 7) {
        double sm = 0.0, sq = 0.0; size_t cnt = 0;
        for (int n = g; n < s.nb; n += s.ng) { const float *row = &hy[((size_t)n * s.cout + m) * s.p]; for (long long q = 0; q < s.p; ++q) { sm += row[q]; sq += (double)row[q] * row[q]; ++cnt; } }
        const double mean = sm / cnt, var = sq / cnt - mean * mean, inv = 1.0 / sqrt(var + 1e-5);
        const float *c = &hco[((size_t)g * s.cout + m) * 4];
        staterr = fmax(staterr, fabs(c[2] - mean) / (fabs(mean) + 1e-3));
        staterr = fmax(staterr, fabs(c[3] - inv) / inv);
      }
    double ferr = 0.0;
    for (int it = 0; it < 2000; ++it) {
      rng = rng * 1664525u + 1013904223u; const int n = rng % s.nb;
      rng = rng * 1664525u + 1013904223u; const int m = rng % s.cout;
      rng = rng * 1664525u + 1013904223u; const long long q = rng % s.p;
      double ref = 0.0;
      for (int kk = 0; kk < s.k; ++kk) {
        const float *c = &hcoef[((size_t)(n % s.ng) * s.k + kk) * 4];
        const float a = fmaxf(hx[((size_t)n * s.k + kk) * s.p + q] * c[0] + c[1], 0.f);
        ref += (double)hw[((size_t)(n % s.ng) * s.cout + m) * s.k + kk] * a;
      }
      ferr = fmax(ferr, fabs(ref - hy[((size_t)n * s.cout + m) * s.p + q]));
    }
    // pooled tail: affine + store + stats + max/min pool
    const int pg = s.g >= 32 ? 32 : 16;
    const float t_pool = timeit([&]() { own(dcoef, 1, dy, dpart, pg, 1); });
    std::vector<float> hpmax(ye / pg), hpmin(ye / pg); std::vector<uint8_t> hamax(ye / pg), hamin(ye / pg);
    CK(hipMemcpy(hpmax.data(), dpmax, ye / pg * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hpmin.data(), dpmin, ye / pg * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hamax.data(), damax, ye / pg, hipMemcpyDeviceToHost)); CK(hipMemcpy(hamin.data(), damin, ye / pg, hipMemcpyDeviceToHost));
    size_t poolbad = 0;
    for (size_t r = 0; r < ye / pg; ++r) {
      float mx = -INFINITY, mn = INFINITY; int ax = 0, an = 0;
      for (int j = 0; j < pg; ++j) { const float v = hy[r * pg + j]; if (v > mx) { mx = v; ax = j; } if (v < mn) { mn = v; an = j; } }
      if (mx != hpmax[r] || mn != hpmin[r] || ax != hamax[r] || an != hamin[r]) ++poolbad;
    }
    const double fl = 2.0 * s.nb * s.cout * s.k * (double)s.p;
    printf("%-40s %9.4f %9.4f %8.1f %8.1f %9.4f %9.1e %9.1e | vs roc %.1e fused err %.1e pool ms %.4f bad %zu\n", s.what, t_roc, t_own,
           fl / t_roc / 1e9, fl / t_own / 1e9, t_fused, maxerr, staterr, maxd, ferr, t_pool, poolbad);
    fflush(stdout);
    hipFree(dx); hipFree(dw); hipFree(dy); hipFree(dy2); hipFree(dcoef); hipFree(dpart); hipFree(dcoef_out);
    hipFree(dpmax); hipFree(dpmin); hipFree(damax); hipFree(damin);
  }
  return 0;
}
