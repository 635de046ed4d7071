// HBM streaming patterns: y = max(a*x+b, 0) over a 268 MB tensor (read + write) and a pure
// read-reduce, for several work shapes: floats per workgroup (SPAN), loads in flight per
// thread (U), nontemporal accesses.  Answers: what is the best a BN-style pass can reach?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void scale_kernel(long long n4, int span4, const float4 *__restrict__ x,
                                                    float4 *__restrict__ y, float a, float b) {
  const long long lo = (long long)blockIdx.x * span4;
  const long long hi = lo + span4 < n4 ? lo + span4 : n4;
  for (long long i = lo + threadIdx.x; i < hi; i += 256 * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long j = i + (long long)u * 256;
      if (j < hi) {
        if (NT) { f4 t = __builtin_nontemporal_load((const f4 *)(x + j)); v[u] = make_float4(t.x, t.y, t.z, t.w); }
        else v[u] = x[j];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long j = i + (long long)u * 256;
      if (j < hi) {
        float4 o = make_float4(fmaxf(v[u].x * a + b, 0.f), fmaxf(v[u].y * a + b, 0.f),
                               fmaxf(v[u].z * a + b, 0.f), fmaxf(v[u].w * a + b, 0.f));
        if (NT) { f4 t = {o.x, o.y, o.z, o.w}; __builtin_nontemporal_store(t, (f4 *)(y + j)); } else y[j] = o;
      }
    }
  }
}

template <int U>
__global__ __launch_bounds__(256) void sum_kernel(long long n4, int span4, const float4 *__restrict__ x,
                                                  float *__restrict__ out) {
  const long long lo = (long long)blockIdx.x * span4;
  const long long hi = lo + span4 < n4 ? lo + span4 : n4;
  float s = 0.f;
  for (long long i = lo + threadIdx.x; i < hi; i += 256 * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long j = i + (long long)u * 256;
      v[u] = j < hi ? x[j] : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
  }
  for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out + (blockIdx.x & 1023), s);
}

int main() {
  const long long n = 67108864;  // 268 MB
  const int NB = 6;                       // 6 x 268 MB inputs + 6 outputs: every pass is cold
  float *xs[NB], *ys[NB], *o;
  for (int i = 0; i < NB; ++i) { hipMalloc(&xs[i], n * 4); hipMalloc(&ys[i], n * 4); hipMemset(xs[i], 0x3c, n * 4); }
  hipMalloc(&o, 4096); hipMemset(o, 0, 4096);
  int turn = 0;
  float *x = xs[0], *y = ys[0];
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const long long n4 = n / 4;
#define RUN(label, bytes, launch)                                            \
  do {                                                                       \
    for (int w = 0; w < 3; ++w) { x = xs[turn % NB]; y = ys[turn % NB]; ++turn; launch; }  \
    hipEventRecord(e0);                                                      \
    for (int r = 0; r < 20; ++r) { x = xs[turn % NB]; y = ys[turn % NB]; ++turn; launch; } \
    hipEventRecord(e1); hipEventSynchronize(e1);                             \
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;                    \
    printf("%-44s %7.1f us  %6.2f TB/s\n", label, ms * 1e3, (bytes) / ms / 1e9); \
  } while (0)
  const int spans[] = {2048, 8192, 32768, 131072};  // floats per workgroup
  char label[128];
  for (int si = 0; si < 4; ++si) {
    const int span4 = spans[si] / 4;
    const int grid = (int)((n4 + span4 - 1) / span4);
#define SC(U, NT)                                                                     \
    snprintf(label, 128, "scale span=%6d U=%d nt=%d grid=%d", spans[si], U, NT, grid); \
    { auto f = [&] { scale_kernel<U, NT><<<grid, 256, 0, 0>>>(n4, span4, (const float4 *)x, (float4 *)y, 1.5f, 0.1f); }; \
      RUN(label, 2.0 * n * 4, f()); }
    SC(1, false); SC(2, false); SC(4, false); SC(8, false); SC(2, true); SC(4, true);
#define SM(U)                                                                 \
    snprintf(label, 128, "sum   span=%6d U=%d grid=%d", spans[si], U, grid);   \
    { auto f = [&] { sum_kernel<U><<<grid, 256, 0, 0>>>(n4, span4, (const float4 *)x, o); }; \
      RUN(label, 1.0 * n * 4, f()); }
    SM(1); SM(2); SM(4); SM(8);
  }
  RUN("hipMemcpyAsync D2D (read + write)", 2.0 * n * 4, (void)hipMemcpyAsync(y, x, n * 4, hipMemcpyDeviceToDevice, 0));
  return 0;
}
