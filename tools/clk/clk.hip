// Measures the shader clock the chip holds while a latency-bound kernel runs on few CUs.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(unsigned long long *out, int iters) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float a = threadIdx.x;
  for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = r1 - r0; }
  if (a == 12345.f) out[0] = 0;
}
int main() {
  unsigned long long *d, h[2];
  hipMalloc(&d, 4096 * 16);
  for (int blocks : {8, 256, 2048}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, 0, d, 2000000);
      hipDeviceSynchronize();
      hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      printf("blocks=%d  cycles=%llu realtime_ticks=%llu  -> %.1f MHz (100MHz ref)  %.2f cyc/iter\n", blocks, h[0], h[1],
             (double)h[0] / (double)h[1] * 100.0, (double)h[0] / 2000000.0);
    }
  }
  return 0;
}
