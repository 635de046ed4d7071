// LDS atomic throughput: float add vs int add vs 64-bit int add, random addresses.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(const int *idx, int iters, float *out) {
  __shared__ unsigned long long acc[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) acc[i] = 0;
  __syncthreads();
  const int *p = idx + (size_t)blockIdx.x * 256 * iters + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    int j = p[(size_t)it * 256];
    if (MODE == 0) atomicAdd((float *)acc + j, 1.5f);
    if (MODE == 1) atomicAdd((int *)acc + j, 3);
    if (MODE == 2) atomicAdd(acc + j, 3ull);
    if (MODE == 3) ((float *)acc)[j] += 1.5f;  // racy plain RMW, for reference
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (float)acc[1];
}
int main() {
  const int blocks = 2048, iters = 512;
  size_t n = (size_t)blocks * 256 * iters;
  int *h = (int *)malloc(n * 4);
  for (size_t i = 0; i < n; ++i) h[i] = rand() & 4095;
  int *d; float *o;
  hipMalloc(&d, n * 4); hipMalloc(&o, blocks * 4);
  hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const char *names[] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "plain rmw"};
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      if (mode == 0) k<0><<<blocks, 256>>>(d, iters, o);
      if (mode == 1) k<1><<<blocks, 256>>>(d, iters, o);
      if (mode == 2) k<2><<<blocks, 256>>>(d, iters, o);
      if (mode == 3) k<3><<<blocks, 256>>>(d, iters, o);
      hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-12s %8.3f ms  %7.1f G atomics/s  (%.2f lanes/clk/CU at 2.4 GHz, 256 CUs)\n", names[mode],
           ms, n / ms / 1e6, n / (ms * 1e-3) / 2.4e9 / 256);
  }
  return 0;
}
