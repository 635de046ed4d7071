// side2box: distribution-over-bins side offsets -> box planes -> (centre, size, yaw), forward
// and backward, one thread per (proposal, side).
//
// Stands in for NesieHead.side2box + Integral + the bbox_probs softmax (reference
// mmdet3d/models/dense_heads/nesie_head.py:19-52, 150-209, 255-257): per proposal 6 softmaxes
// over reg_max+1 bins, their expectations scaled to metres and added to / subtracted from the
// aggregated point, the heading from a normalised (sin, cos) pair.  ~75 small ATen launches
// (forward + autograd backward) become two.  reg is channel-major (B, 6*(R+1)+2, K), so the
// threads of a wave read consecutive proposals of one channel: dense rows.
#include "common.h"
#include <math.h>

namespace nesie {

constexpr int DEC_MAXBINS = 33;

// One thread per (proposal, side): a workgroup = 64 proposals x 6 sides, wave s owns side s (its
// lanes read consecutive proposals of one channel row); the six planes of a proposal meet in LDS.
constexpr int DEC_KB = 64;

__global__ __launch_bounds__(6 * DEC_KB) void side_decode_fwd_kernel(
    int kprop, int bins, const float *__restrict__ reg, const float *__restrict__ agg,
    const float *__restrict__ scale, const float *__restrict__ sign, float *__restrict__ probs,
    float *__restrict__ surface, float *__restrict__ bbox) {
  __shared__ float sfs[6][DEC_KB];
  const int kl = threadIdx.x & (DEC_KB - 1), s = threadIdx.x / DEC_KB;
  const int k = blockIdx.x * DEC_KB + kl, bi = blockIdx.y;
  const bool live = k < kprop;
  const int kc = live ? k : kprop - 1;
  const int cch = 6 * bins + 2;
  const float *r = reg + (size_t)bi * cch * kprop + kc;
  const float *rs = r + (size_t)(s * bins) * kprop;
  float mx = -INFINITY;
  for (int j = 0; j < bins; ++j) mx = fmaxf(mx, rs[(size_t)j * kprop]);
  float sum = 0.f;
  for (int j = 0; j < bins; ++j) sum += expf(rs[(size_t)j * kprop] - mx);
  float res = 0.f;
  float *pr = probs + (size_t)bi * 6 * bins * kprop + (size_t)(s * bins) * kprop + kc;
  for (int j = 0; j < bins; ++j) {
    const float pj = expf(rs[(size_t)j * kprop] - mx) / sum;
    if (live) pr[(size_t)j * kprop] = pj;
    res += pj * ((float)j / (float)(bins - 1));
  }
  const float *a = agg + ((size_t)bi * kprop + kc) * 3;
  const float sf = a[s % 3] + sign[s] * (res * scale[s]);
  sfs[s][kl] = sf;
  if (live) surface[((size_t)bi * kprop + k) * 6 + s] = sf;
  __syncthreads();
  if (!live) return;
  float *bo = bbox + ((size_t)bi * kprop + k) * 7;
  if (s < 3) {
    bo[s] = (sfs[s][kl] + sfs[s + 3][kl]) / 2.0f;
  } else {
    const int d = s - 3;
    bo[3 + d] = sfs[d + 3][kl] - sfs[d][kl];
    if (d == 0) {
      const float h0 = r[(size_t)(6 * bins) * kprop], h1 = r[(size_t)(6 * bins + 1) * kprop];
      const float nrm = sqrtf(h0 * h0 + h1 * h1);
      bo[6] = atan2f(h0 / nrm, h1 / nrm);
    }
  }
}

__global__ __launch_bounds__(6 * DEC_KB) void side_decode_bwd_kernel(
    int kprop, int bins, const float *__restrict__ reg, const float *__restrict__ probs,
    const float *__restrict__ scale, const float *__restrict__ sign,
    const float *__restrict__ d_surface, const float *__restrict__ d_bbox,
    float *__restrict__ d_reg, float *__restrict__ d_agg) {
  __shared__ float gs[6][DEC_KB];
  const int kl = threadIdx.x & (DEC_KB - 1), s = threadIdx.x / DEC_KB;
  const int k = blockIdx.x * DEC_KB + kl, bi = blockIdx.y;
  const bool live = k < kprop;
  const int kc = live ? k : kprop - 1;
  const int cch = 6 * bins + 2;
  const float *ds = d_surface ? d_surface + ((size_t)bi * kprop + kc) * 6 : nullptr;
  const float *db = d_bbox ? d_bbox + ((size_t)bi * kprop + kc) * 7 : nullptr;
  // gradient reaching plane s: lo planes (s < 3) take centre / 2 - size, hi planes centre / 2 + size
  float g = ds ? ds[s] : 0.f;
  if (db) {
    const int d = s % 3;
    g += s < 3 ? db[d] * 0.5f - db[3 + d] : db[d] * 0.5f + db[3 + d];
  }
  gs[s][kl] = g;
  const float *pr = probs + (size_t)bi * 6 * bins * kprop + (size_t)(s * bins) * kprop + kc;
  float *dr = d_reg + (size_t)bi * cch * kprop + kc;
  const float dres = g * sign[s] * scale[s];
  float res = 0.f;
  for (int j = 0; j < bins; ++j) res += pr[(size_t)j * kprop] * ((float)j / (float)(bins - 1));
  if (live)
    for (int j = 0; j < bins; ++j) {
      const float pj = pr[(size_t)j * kprop];
      dr[(size_t)(s * bins + j) * kprop] = pj * dres * ((float)j / (float)(bins - 1) - res);
    }
  __syncthreads();
  if (!live) return;
  if (s < 3) {
    d_agg[((size_t)bi * kprop + k) * 3 + s] = gs[s][kl] + gs[s + 3][kl];
  } else if (s == 3) {
    const float *r = reg + (size_t)bi * cch * kprop + k;
    const float h0 = r[(size_t)(6 * bins) * kprop], h1 = r[(size_t)(6 * bins + 1) * kprop];
    const float n2 = h0 * h0 + h1 * h1;
    const float dyaw = db ? db[6] : 0.f;
    dr[(size_t)(6 * bins) * kprop] = dyaw * h1 / n2;
    dr[(size_t)(6 * bins + 1) * kprop] = -dyaw * h0 / n2;
  }
}

}  // namespace nesie

using namespace nesie;

static int dec_check(const char *W, int b, int k, int bins) {
  NESIE_REQUIRE(b >= 0 && k >= 0 && bins >= 2 && b <= 65535, W);
  if (bins > DEC_MAXBINS) {
    set_error("%s: %d bins per side (built for <= %d)", W, bins, DEC_MAXBINS);
    return NESIE_ERR_UNSUPPORTED;
  }
  return NESIE_OK;
}

extern "C" int nesie_side_decode_forward(int b, int k, int bins, const float *reg,
                                         const float *agg, const float *scale, const float *sign,
                                         float *probs, float *surface, float *bbox,
                                         void *stream) {
  const char *W = "side_decode_forward";
  int st = dec_check(W, b, k, bins);
  if (st || b == 0 || k == 0) return st;
  NESIE_REQUIRE(reg && agg && scale && sign && probs && surface && bbox, W);
  hipLaunchKernelGGL(side_decode_fwd_kernel, dim3(cdiv(k, DEC_KB), b), dim3(6 * DEC_KB), 0,
                     (hipStream_t)stream, k, bins, reg, agg, scale, sign, probs, surface, bbox);
  return check_launch(W);
}

extern "C" int nesie_side_decode_backward(int b, int k, int bins, const float *reg,
                                          const float *probs, const float *scale,
                                          const float *sign, const float *d_surface,
                                          const float *d_bbox, float *d_reg, float *d_agg,
                                          void *stream) {
  const char *W = "side_decode_backward";
  int st = dec_check(W, b, k, bins);
  if (st || b == 0 || k == 0) return st;
  NESIE_REQUIRE(reg && probs && scale && sign && d_reg && d_agg, W);
  hipLaunchKernelGGL(side_decode_bwd_kernel, dim3(cdiv(k, DEC_KB), b), dim3(6 * DEC_KB), 0,
                     (hipStream_t)stream, k, bins, reg, probs, scale, sign, d_surface, d_bbox,
                     d_reg, d_agg);
  return check_launch(W);
}

// ---- per-face statistics of the side-bin distributions -----------------------------------------
// SidePooling.dist_feature (side_pooling_module.py:245-264): for every (scene, face, proposal) the
// bins' probabilities, their four largest values (descending) and their unbiased variance,
// face-major and repeated `copies` times along the proposal axis (the jittered half of the
// quality head reads the same statistics): probs (B, 6, bins, K) -> out (6, B, bins + 5, copies K).
// One launch instead of topk + sort + var + cat + permute + repeat.
namespace nesie {
__global__ __launch_bounds__(256) void side_prob_stats_kernel(int b, int bins, int kprop, int copies,
                                                              const float *__restrict__ probs,
                                                              float *__restrict__ out) {
  const int k = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y, bi = blockIdx.z;
  if (k >= kprop) return;
  const float *p = probs + (((size_t)bi * 6 + s) * bins) * kprop + k;
  const size_t row = (size_t)copies * kprop;
  float *o = out + (((size_t)s * b + bi) * (bins + 5)) * row + k;
  float t0 = -INFINITY, t1 = -INFINITY, t2 = -INFINITY, t3 = -INFINITY, sum = 0.f;
  for (int j = 0; j < bins; ++j) {
    const float v = p[(size_t)j * kprop];
    sum += v;
    for (int c = 0; c < copies; ++c) o[(size_t)j * row + (size_t)c * kprop] = v;
    // insert into the running top four (descending)
    if (v > t3) {
      t3 = v;
      if (t3 > t2) { const float x = t2; t2 = t3; t3 = x; }
      if (t2 > t1) { const float x = t1; t1 = t2; t2 = x; }
      if (t1 > t0) { const float x = t0; t0 = t1; t1 = x; }
    }
  }
  const float mean = sum / (float)bins;
  float m2 = 0.f;
  for (int j = 0; j < bins; ++j) {
    const float d = p[(size_t)j * kprop] - mean;
    m2 += d * d;
  }
  const float var = m2 / (float)(bins - 1);
  for (int c = 0; c < copies; ++c) {
    float *oc = o + (size_t)c * kprop;
    oc[(size_t)bins * row] = t0; oc[(size_t)(bins + 1) * row] = t1;
    oc[(size_t)(bins + 2) * row] = t2; oc[(size_t)(bins + 3) * row] = t3;
    oc[(size_t)(bins + 4) * row] = var;
  }
}
}  // namespace nesie

extern "C" int nesie_side_prob_stats(int b, int bins, int kprop, int copies, const float *probs,
                                     float *out, void *stream) {
  const char *W = "side_prob_stats";
  NESIE_REQUIRE(b >= 0 && bins >= 5 && kprop >= 0 && copies >= 1, W);
  if (b == 0 || kprop == 0) return NESIE_OK;
  NESIE_REQUIRE(probs && out && b <= 65535, W);
  hipLaunchKernelGGL(nesie::side_prob_stats_kernel, dim3((kprop + 255) / 256, 6, b), dim3(256), 0,
                     (hipStream_t)stream, b, bins, kprop, copies, probs, out);
  return nesie::check_launch(W);
}

// ---- proposal jitter ----------------------------------------------------------------------------
// NesieHead.jitter_bbox_preds (nesie_head.py:178-209): every proposal gets a perturbed copy
// (centre += size * n_c * sigma, size = max(size + size * (n_s * sigma + size_bias), 1e-8)); the
// quality head scores the 2K boxes [original, jittered].  One launch for the ~14 element-wise /
// concatenation ops: bbox (B,K,7), noise_c / noise_s (B,K,3) ->
//   centre_all (B,2K,3), size_all (B,2K,3), heading_all (B,2K) (zero when zero_heading),
//   jitter_bbox (B,K,7) = (jittered centre, jittered size, heading).
namespace nesie {
__global__ __launch_bounds__(256) void proposal_jitter_kernel(
    int b, int k, const float *__restrict__ bbox, const float *__restrict__ nc,
    const float *__restrict__ ns, float sigma, float size_bias, int zero_heading,
    float *__restrict__ centre_all, float *__restrict__ size_all, float *__restrict__ heading_all,
    float *__restrict__ jitter) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= b * k) return;
  const int bi = p / k, kk = p % k;
  const float *bx = bbox + (size_t)p * 7;
  const size_t o0 = ((size_t)bi * 2 * k + kk) * 3, o1 = ((size_t)bi * 2 * k + k + kk) * 3;
  float *jo = jitter + (size_t)p * 7;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float c = bx[a], s = bx[3 + a];
    const float cj = c + s * nc[(size_t)p * 3 + a] * sigma;
    // the two heads' own operation orders: s + (s n) sigma (Nesie), s + s (n sigma + bias) (SAQE)
    const float nsv = ns[(size_t)p * 3 + a];
    const float sj = fmaxf(size_bias == 0.f ? s + (s * nsv) * sigma : s + s * (nsv * sigma + size_bias), 1e-8f);
    centre_all[o0 + a] = c; centre_all[o1 + a] = cj;
    size_all[o0 + a] = s; size_all[o1 + a] = sj;
    jo[a] = cj; jo[3 + a] = sj;
  }
  const float h = bx[6];
  jo[6] = h;
  heading_all[(size_t)bi * 2 * k + kk] = zero_heading ? 0.f : h;
  heading_all[(size_t)bi * 2 * k + k + kk] = zero_heading ? 0.f : h;
}
}  // namespace nesie

extern "C" int nesie_proposal_jitter(int b, int k, const float *bbox, const float *noise_c,
                                     const float *noise_s, float sigma, float size_bias,
                                     int zero_heading, float *centre_all, float *size_all,
                                     float *heading_all, float *jitter_bbox, void *stream) {
  const char *W = "proposal_jitter";
  NESIE_REQUIRE(b >= 0 && k >= 0, W);
  if (b == 0 || k == 0) return NESIE_OK;
  NESIE_REQUIRE(bbox && noise_c && noise_s && centre_all && size_all && heading_all && jitter_bbox, W);
  hipLaunchKernelGGL(nesie::proposal_jitter_kernel, dim3((b * k + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, b, k, bbox, noise_c, noise_s, sigma, size_bias, zero_heading,
                     centre_all, size_all, heading_all, jitter_bbox);
  return nesie::check_launch(W);
}
