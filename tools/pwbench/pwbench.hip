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
                            float *running_var, float momentum, float eps, float *coef, void *stream);
const char *nesie_last_error(void);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static uint32_t rng = 12345;
static float frand() { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) & 0xFFFFFF) / 8388608.0f - 1.0f; }

struct Shape { int nb, ng, k, cout; long long p; int g; const char *what; };

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
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
  printf("%-40s %9s %9s %8s %8s %9s %9s %9s\n", "layer", "rocblas", "nesie", "TF(roc)", "TF(own)", "fused ms", "maxerr", "staterr");
  for (const Shape &s : shapes) {
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
    auto own = [&](const float *coef, int relu, float *y, float *part, int pg, int pmin) {
      int st = nesie_pw_layer_forward(s.nb, s.ng, s.k, s.cout, s.p, dx, (long long)s.k * s.p, dw, (long long)s.cout * s.k,
                                      s.k, 1, coef, relu, nullptr, 0, nullptr, y, (long long)s.cout * s.p, part, pg, pmin,
                                      dpmax, dpmin, damax, damin, 0);
      if (st) { printf("nesie error %d: %s\n", st, nesie_last_error()); exit(1); }
    };
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
    // fused: affine + relu prologue, store + stats epilogue
    const float t_fused = timeit([&]() { own(dcoef, 1, dy, dpart, 0, 0); });
    nesie_pw_stats_finalize(s.ng * s.cout, s.cout, slots, dpart, nullptr, nullptr, nullptr, nullptr, 0.1f, 1e-5f, dcoef_out, 0);
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
