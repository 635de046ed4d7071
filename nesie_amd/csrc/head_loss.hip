// The Nesie head's training targets and its seven per-proposal loss terms, forward and gradient.
//
// Stands in for NesieHead.get_targets (proposal <-> ground-truth assignment and the batch-level
// weights, reference mmdet3d/models/dense_heads/nesie_head.py:566-588, 656-676) and for the loss
// terms of NesieHead.loss (:279-413) with the loss classes they call (CrossEntropyLoss with class
// weights, ChamferDistance l2, SurfaceLoss/MSE, IoU3DLoss, GeneralQualityFocalLoss on
// probabilities, SidePredLoss): ~300 small ATen launches (forward + autograd backward) become
// three.  Every term is a sum over <= a few thousand proposals of closed-form expressions, so ONE
// workgroup evaluates the batch: the normalisers (number of positive / assigned proposals, valid
// boxes) are block-wide integer counts, the seven sums are reduced in a fixed order (bitwise
// reproducible), and the analytic gradients leave with the forward.
#include "common.h"
#include <math.h>
#include <stdint.h>

namespace nesie {

constexpr int HL_BLOCK = 1024;
constexpr int HL_MAXC = 32;      // classes

__device__ __forceinline__ float hl_block_sum(float v, float *sh) {
  // fixed-order tree over the block: lanes by DPP-free shuffles, waves through LDS
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < HL_BLOCK / 64; ++w) t += sh[w];
  return t;
}

__device__ __forceinline__ int hl_block_count(int v, int *sh) {
  __syncthreads();
  if (threadIdx.x == 0) *sh = 0;
  __syncthreads();
  if (v) atomicAdd(sh, v);
  __syncthreads();
  return *sh;
}

// ---- targets ----------------------------------------------------------------------------------
__global__ __launch_bounds__(HL_BLOCK) void head_targets_kernel(
    int b, int k, int t, const float *__restrict__ agg, const float *__restrict__ gt_boxes,
    const long long *__restrict__ gt_labels, const long long *__restrict__ gt_count,
    const float *__restrict__ gt_valid, float pos_thr, float neg_thr,
    long long *__restrict__ assignment, long long *__restrict__ obj_targets,
    float *__restrict__ obj_weights, long long *__restrict__ mask_targets,
    float *__restrict__ bbox_targets, float *__restrict__ center_targets,
    float *__restrict__ box_weights, float *__restrict__ valid_weights) {
  __shared__ int cnt;
  const int np = b * k, ng = b * t;
  int n_pos = 0, n_mask = 0, n_valid = 0;
  for (int p = threadIdx.x; p < np; p += HL_BLOCK) {
    const int bi = p / k;
    const float ax = agg[p * 3], ay = agg[p * 3 + 1], az = agg[p * 3 + 2];
    const int cols = (int)gt_count[bi];
    float best = INFINITY;
    int at = 0;
    for (int j = 0; j < t; ++j) {
      const float *g = gt_boxes + ((size_t)bi * t + j) * 7;
      const float dx = ax - g[0], dy = ay - g[1], dz = az - (g[2] + g[5] * 0.5f);
      float d = (dx * dx + dy * dy) + dz * dz;
      d = j < cols ? d : INFINITY;
      if (d < best) { best = d; at = j; }      // first minimum wins (torch.min)
    }
    const float e = sqrtf(best + 1e-6f);
    const bool pos = e < pos_thr, neg = e > neg_thr;
    assignment[p] = at;
    obj_targets[p] = pos ? 1 : 0;
    obj_weights[p] = (pos || neg) ? 1.f : 0.f;     // normalised below
    box_weights[p] = pos ? 1.f : 0.f;
    mask_targets[p] = gt_labels[(size_t)bi * t + at];
    const float *g = gt_boxes + ((size_t)bi * t + at) * 7;
    float *o = bbox_targets + (size_t)p * 7;
    o[0] = g[0]; o[1] = g[1]; o[2] = g[2] + g[5] * 0.5f;
    o[3] = g[3]; o[4] = g[4]; o[5] = g[5]; o[6] = g[6];
    n_pos += pos ? 1 : 0;
    n_mask += (pos || neg) ? 1 : 0;
  }
  for (int q = threadIdx.x; q < ng; q += HL_BLOCK) {
    const int bi = q / t, j = q % t;
    const float *g = gt_boxes + (size_t)q * 7;
    const bool col = j < (int)gt_count[bi];
    center_targets[q * 3] = col ? g[0] : 0.f;
    center_targets[q * 3 + 1] = col ? g[1] : 0.f;
    center_targets[q * 3 + 2] = col ? g[2] + g[5] * 0.5f : 0.f;
    n_valid += gt_valid[q] != 0.f ? 1 : 0;
  }
  const float f_pos = (float)hl_block_count(n_pos, &cnt) + 1e-6f;
  const float f_mask = (float)hl_block_count(n_mask, &cnt) + 1e-6f;
  const float f_valid = (float)hl_block_count(n_valid, &cnt) + 1e-6f;
  for (int p = threadIdx.x; p < np; p += HL_BLOCK) {
    obj_weights[p] = obj_weights[p] / f_mask;
    box_weights[p] = box_weights[p] / f_pos;
  }
  for (int q = threadIdx.x; q < ng; q += HL_BLOCK) valid_weights[q] = gt_valid[q] / f_valid;
}

// ---- losses -----------------------------------------------------------------------------------
// Layouts are the producers' own (no copies on the way in or out):
//   cls     (B, 2 + C, K)   prediction-head output: rows 0..1 objectness logits, 2.. class logits
//   bbox    (B, K, 7)       decoded boxes; columns 0..2 are the centres
//   surface (B, K, 6)
//   side    (6, B, C, 2K)   side-quality probabilities, plain proposals then jittered copies
//   iou_s   (B, 2K, C)      IoU-quality probabilities, plain then jittered
struct HeadLoss {
  int b, k, t, c;
  const float *cls, *bbox, *surface, *side, *iou_s;
  const float *iou;        // (B*K) IoU of the predicted box with its target
  const float *iou_j;      // (B*K) IoU of the jittered box with its target
  // targets (head_targets_kernel)
  const long long *obj_t, *label;
  const float *obj_w, *box_w, *bbox_t, *centre_t, *valid_w;
  // configuration
  float alpha, w_obj, cw0, cw1, w_sem, w_csrc, w_cdst, w_surf, w_iou, w_qfl, w_side;
  // outputs: loss[7] = objectness, semantic, centre, surface, iou, iou_pred, side; s_* = the
  // gradient of each term w.r.t. its inputs (unit incoming gradient), in the input's layout
  float *loss;
  float *s_cls, *s_centre /* (B*K,3) */, *s_surface, *s_iou /* (B*K) */, *s_iou_s;
  float *s_side_surf, *s_side_iou, *s_side_pred;   // (B*K, 6) each
  int *sem_pick;           // (B*K) arg-max class of the class logits (the column the sigmas read)
  int *kstar;              // (B*T) scratch: nearest proposal of every ground-truth centre
  float *dmin;             // (B*T) scratch
  float *partial;          // (workgroups, 8) scratch
  int *ticket;             // zero before the first launch; the kernel leaves it zero
  // the unsupervised variant (NesieHead.unsup_loss, nesie_head.py:415-509; SAQEHead's, saqe_head.py:706-800):
  // quality (B*K, 6) = the pseudo label's side qualities gathered per proposal: the surface term of
  // side i is weighted box_w * quality[i], the IoU term box_w * mean(quality); HL_UNSUP drops the
  // objectness, IoU-quality and side-quality terms; HL_DETACH_SIGMA treats the uncertainties as
  // constants (no gradient into the side scores)
  const float *quality;
  int flags;
};
// HL_NO_SIGMA: no uncertainty weighting at all (SAQEHead.loss, the pre-training loss: sigma = 0)
enum { HL_UNSUP = 1, HL_DETACH_SIGMA = 2, HL_NO_SIGMA = 4 };

__device__ __forceinline__ float hl_log_clamped(float v) { return fmaxf(logf(v), -100.f); }

// nearest proposal of every ground-truth centre (the destination half of the chamfer term)
__global__ __launch_bounds__(64) void head_loss_nearest_kernel(const HeadLoss a) {
  const int q = blockIdx.x * 64 + threadIdx.x, K = a.k;
  if (q >= a.b * a.t) return;
  const int bi = q / a.t;
  const float gx = a.centre_t[q * 3], gy = a.centre_t[q * 3 + 1], gz = a.centre_t[q * 3 + 2];
  float best = INFINITY;
  int at = 0;
  for (int j = 0; j < K; ++j) {
    const float *cc = a.bbox + (size_t)(bi * K + j) * 7;
    const float dx = cc[0] - gx, dy = cc[1] - gy, dz = cc[2] - gz;
    const float d = (dx * dx + dy * dy) + dz * dz;
    if (d < best) { best = d; at = j; }
  }
  a.kstar[q] = at;
  a.dmin[q] = best;
}

constexpr int HL_PB = 64;        // proposals (threads) per workgroup of the loss kernel

// One thread per proposal; every workgroup leaves seven partial sums and the LAST one to finish
// (a self-resetting ticket) adds the partials in workgroup order: reproducible, one launch.
__global__ __launch_bounds__(HL_PB) void head_loss_kernel(const HeadLoss a) {
  __shared__ int last;
  const int np = a.b * a.k, ng = a.b * a.t, C = a.c, K = a.k, NC = 2 + a.c;
  float l_obj = 0.f, l_sem = 0.f, l_c = 0.f, l_surf = 0.f, l_iou = 0.f, l_qfl = 0.f, l_side = 0.f;
  for (int q = blockIdx.x * HL_PB + threadIdx.x; q < ng; q += gridDim.x * HL_PB)
    l_c += a.w_cdst * (a.dmin[q] * a.valid_w[q]);
  for (int p = blockIdx.x * HL_PB + threadIdx.x; p < np; p += gridDim.x * HL_PB) {
    const int bi = p / K, kk = p % K;
    const float w = a.box_w[p], ow = a.obj_w[p];
    const int y = (int)a.obj_t[p], lab = (int)a.label[p];
    const float *z = a.cls + (size_t)bi * NC * K + kk;        // channel j at z[j * K]
    float *dzp = a.s_cls + (size_t)bi * NC * K + kk;
    // -- objectness: class-weighted softmax cross entropy
    {
      const float z0 = z[0], z1 = z[K];
      const float m = fmaxf(z0, z1);
      const float e0 = expf(z0 - m), e1 = expf(z1 - m), s = e0 + e1;
      const float lse = m + logf(s);
      const float cw = y ? a.cw1 : a.cw0;
      const bool on = !(a.flags & HL_UNSUP);
      l_obj += on ? a.w_obj * ((-cw * ((y ? z1 : z0) - lse)) * ow) : 0.f;
      const float gsc = on ? a.w_obj * ow * cw : 0.f;
      dzp[0] = gsc * (e0 / s - (y ? 0.f : 1.f));
      dzp[K] = gsc * (e1 / s - (y ? 1.f : 0.f));
    }
    // -- semantic cross entropy + the arg-max class (first maximum)
    int pick = 0;
    {
      const float *zs = z + 2 * K;
      float m = zs[0];
      for (int j = 1; j < C; ++j) { if (zs[(size_t)j * K] > m) { m = zs[(size_t)j * K]; pick = j; } }
      float s = 0.f;
      for (int j = 0; j < C; ++j) s += expf(zs[(size_t)j * K] - m);
      const float lse = m + logf(s);
      l_sem += a.w_sem * ((-(zs[(size_t)lab * K] - lse)) * w);
      const float gsc = a.w_sem * w;
      for (int j = 0; j < C; ++j)
        dzp[(size_t)(2 + j) * K] = gsc * (expf(zs[(size_t)j * K] - m) / s - (j == lab ? 1.f : 0.f));
      a.sem_pick[p] = pick;
    }
    // -- centre: nearest (padded) ground-truth centre, and this proposal's share of the
    //    destination half
    const float *cc = a.bbox + (size_t)p * 7;
    {
      float best = INFINITY;
      int at = 0;
      const float *ct = a.centre_t + (size_t)bi * a.t * 3;
      for (int j = 0; j < a.t; ++j) {
        const float dx = cc[0] - ct[j * 3], dy = cc[1] - ct[j * 3 + 1], dz = cc[2] - ct[j * 3 + 2];
        const float d = (dx * dx + dy * dy) + dz * dz;
        if (d < best) { best = d; at = j; }
      }
      l_c += a.w_csrc * (best * w);
      float gx = a.w_csrc * w * 2.f * (cc[0] - ct[at * 3]);
      float gy = a.w_csrc * w * 2.f * (cc[1] - ct[at * 3 + 1]);
      float gz = a.w_csrc * w * 2.f * (cc[2] - ct[at * 3 + 2]);
      for (int j = 0; j < a.t; ++j) {
        if (a.kstar[bi * a.t + j] == kk) {
          const float vw = a.w_cdst * a.valid_w[bi * a.t + j] * 2.f;
          gx += vw * (cc[0] - ct[j * 3]); gy += vw * (cc[1] - ct[j * 3 + 1]);
          gz += vw * (cc[2] - ct[j * 3 + 2]);
        }
      }
      a.s_centre[p * 3] = gx; a.s_centre[p * 3 + 1] = gy; a.s_centre[p * 3 + 2] = gz;
    }
    // -- the six sides: surface regression, side quality, and the uncertainty weights
    const float *tb = a.bbox_t + (size_t)p * 7;
    const size_t side_k = (size_t)2 * K;     // proposals per (side, scene, class) row
    float sig[6], dsig_ds[6], sig_mean = 0.f;
    for (int i = 0; i < 6; ++i) {
      const float s = a.side[(((size_t)i * a.b + bi) * C + pick) * side_k + kk];
      sig[i] = (a.flags & HL_NO_SIGMA) ? 0.f : 0.8f * s * s - 1.8f * s + 1.f;
      dsig_ds[i] = (a.flags & HL_NO_SIGMA) ? 0.f : 1.6f * s - 1.8f;
      sig_mean += sig[i];
    }
    sig_mean = sig_mean / 6.f;
    const bool unsup = (a.flags & HL_UNSUP) != 0, keep_sigma = !(a.flags & HL_DETACH_SIGMA);
    float qmean = 1.f;
    if (a.quality) {
      qmean = 0.f;
      for (int i = 0; i < 6; ++i) qmean += a.quality[p * 6 + i];
      qmean = qmean / 6.f;
    }
    for (int i = 0; i < 6; ++i) {
      const float w_box = w;                       // (the side-quality term keeps the plain weight)
      const float w = a.quality ? w_box * a.quality[p * 6 + i] : w_box;
      const float half = 0.5f * tb[3 + i % 3];
      const float ts = i < 3 ? tb[i] - half : tb[i - 3] + half;
      const float sp = a.surface[p * 6 + i];
      const float e = sp - ts;
      const float l = a.w_surf * ((e * e) * w);
      const float ex = expf(-sig[i]);
      l_surf += ex * l + a.alpha * sig[i] * w;
      a.s_surface[p * 6 + i] = ex * (a.w_surf * (2.f * e) * w);
      a.s_side_surf[p * 6 + i] = keep_sigma ? (-ex * l + a.alpha * w) * dsig_ds[i] : 0.f;
      // side quality: label = min(1, 4 |error|), squared error of the assigned class's score
      const float lbl = fminf(4.f * fabsf(e), 1.f);
      const float sc = a.side[(((size_t)i * a.b + bi) * C + lab) * side_k + kk];
      const float r = sc - lbl;
      l_side += unsup ? 0.f : a.w_side * ((r * r) * w_box);
      a.s_side_pred[p * 6 + i] = unsup ? 0.f : a.w_side * (2.f * r) * w_box;
    }
    // -- IoU regression under the mean uncertainty
    {
      const float wq = a.quality ? w * qmean : w;
      const float li = a.w_iou * ((wq > 0.f ? 1.f - a.iou[p] : 0.f) * wq);
      const float ex = expf(-sig_mean);
      l_iou += ex * li + a.alpha * sig_mean * wq;
      a.s_iou[p] = ex * (a.w_iou * (wq > 0.f ? -1.f : 0.f) * wq);
      const float dmean = keep_sigma ? (-ex * li + a.alpha * wq) / 6.f : 0.f;
      for (int i = 0; i < 6; ++i) a.s_side_iou[p * 6 + i] = dmean * dsig_ds[i];
    }
    // -- IoU prediction: quality focal loss on probabilities, plain and jittered halves
    for (int hj = 0; hj < 2; ++hj) {
      const size_t row = ((size_t)bi * 2 * K + (size_t)hj * K + kk) * C;
      const float *pr = a.iou_s + row;
      float *dp = a.s_iou_s + row;
      const float score = hj ? a.iou_j[p] : a.iou[p];
      float acc = 0.f;
      if (unsup) {
        for (int j = 0; j < C; ++j) dp[j] = 0.f;
        continue;
      }
      for (int j = 0; j < C; ++j) {
        const float q = pr[j];
        const float den = fmaxf((1.f - q) * q, 1e-12f);
        if (j == lab) {
          const float bce = -(score * hl_log_clamped(q) + (1.f - score) * hl_log_clamped(1.f - q));
          const float df = score - q, mod = df * df;
          acc += bce * mod;
          dp[j] = a.w_qfl * w * (((q - score) / den) * mod + bce * (2.f * (q - score)));
        } else {
          const float bce = -hl_log_clamped(1.f - q);
          acc += bce * (q * q);
          dp[j] = a.w_qfl * w * ((q / den) * (q * q) + bce * (2.f * q));
        }
      }
      l_qfl += a.w_qfl * (acc * w);
    }
  }
  float v[7] = {l_obj, l_sem, l_c, l_surf, l_iou, l_qfl, l_side};
#pragma unroll
  for (int i = 0; i < 7; ++i) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v[i] += __shfl_xor(v[i], off, 64);
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 7; ++i) a.partial[(size_t)blockIdx.x * 8 + i] = v[i];
    __threadfence();
    last = atomicAdd(a.ticket, 1) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (threadIdx.x < 7) {
    float t = 0.f;
    for (unsigned w = 0; w < gridDim.x; ++w) t += __builtin_nontemporal_load(a.partial + (size_t)w * 8 + threadIdx.x);
    a.loss[threadIdx.x] = t;
  }
  if (threadIdx.x == 0) *a.ticket = 0;
}

// gradient assembly: every saved per-term gradient times the incoming gradient of its term, in
// the producers' layouts; d_side must arrive zero-filled (only two class columns per proposal and
// side are touched)
struct HeadLossBwd {
  int b, k, c;
  const float *g;            // [7] incoming gradients of the seven terms
  const long long *label;
  const int *sem_pick;
  const float *s_cls, *s_centre, *s_surface, *s_iou, *s_iou_s;
  const float *s_side_surf, *s_side_iou, *s_side_pred;
  float *d_cls, *d_bbox, *d_surface, *d_iou, *d_iou_s, *d_side;
};

__global__ __launch_bounds__(256) void head_loss_bwd_kernel(const HeadLossBwd a) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int K = a.k, C = a.c, NC = 2 + a.c;
  if (p >= a.b * K) return;
  const int bi = p / K, kk = p % K;
  const float g0 = a.g[0], g1 = a.g[1], g2 = a.g[2], g3 = a.g[3], g4 = a.g[4], g5 = a.g[5], g6 = a.g[6];
  const size_t zo = (size_t)bi * NC * K + kk;
  a.d_cls[zo] = g0 * a.s_cls[zo];
  a.d_cls[zo + K] = g0 * a.s_cls[zo + K];
  for (int j = 0; j < C; ++j) a.d_cls[zo + (size_t)(2 + j) * K] = g1 * a.s_cls[zo + (size_t)(2 + j) * K];
  for (int hj = 0; hj < 2; ++hj) {
    const size_t row = ((size_t)bi * 2 * K + (size_t)hj * K + kk) * C;
    for (int j = 0; j < C; ++j) a.d_iou_s[row + j] = g5 * a.s_iou_s[row + j];
  }
  float *db = a.d_bbox + (size_t)p * 7;
  for (int i = 0; i < 3; ++i) db[i] = g2 * a.s_centre[p * 3 + i];
  db[3] = db[4] = db[5] = db[6] = 0.f;
  a.d_iou[p] = g4 * a.s_iou[p];
  const int pick = a.sem_pick[p], lab = (int)a.label[p];
  const size_t side_k = (size_t)2 * K;
  for (int i = 0; i < 6; ++i) {
    a.d_surface[p * 6 + i] = g3 * a.s_surface[p * 6 + i];
    float *base = a.d_side + (((size_t)i * a.b + bi) * C) * side_k + kk;
    base[(size_t)pick * side_k] = g3 * a.s_side_surf[p * 6 + i] + g4 * a.s_side_iou[p * 6 + i];
    base[(size_t)lab * side_k] += g6 * a.s_side_pred[p * 6 + i];
  }
}

// ---- the SAQE head's additional supervised terms (saqe_head.py:331-521 `loss`, :524-703 `sup_loss`) ----
// On top of the terms head_loss_kernel evaluates (objectness on obj_scores, semantic, centre,
// surface, IoU, IoU quality, side quality of the plain proposals):
//   [0] 0.5 (CE(R_obj) + CE(R_obj_jitter))      objectness of the quality head, plain + jittered
//   [1] angle: SmoothL1(sin) + SmoothL1(cos) of the heading, weight box_w (x exp(-angle_sigma) in
//       sup_loss, angle_sigma = 0.8 a^2 - 1.8 a + 1 of the rotate score at the arg-max class, constant)
//   [2] angle quality (loss only): MSE of the rotate score (plain + jittered, arg-max class) against
//       angle_term / max(box_w), weight box_w
//   [3] side quality of the JITTERED proposals: label = min(1, 4 |jittered plane - target plane|)
// Layouts: robj (B, 2K, 2), rot (B, 2K, C) probabilities, jsurf (B, K, 6), side as in HeadLoss.
struct SaqeExtra {
  int b, k, c, sup;
  const float *robj, *rot, *bbox, *bbox_t, *jsurf, *side;
  const long long *obj_t, *label;
  const float *obj_w, *box_w, *box_w_max;
  const int *sem_pick;
  float w_obj, cw0, cw1, w_angle, beta, w_apred, w_side;
  float *loss;                 // [4]
  float *s_robj, *s_angle, *s_rot /* zero-filled */, *s_sidej /* (B*K, 6) */;
  float *partial;
  int *ticket;
};

__device__ __forceinline__ float hl_smooth_l1(float d, float beta, float *grad) {
  const float ad = fabsf(d);
  if (ad < beta) { *grad = d / beta; return 0.5f * ad * ad / beta; }
  *grad = d > 0.f ? 1.f : -1.f;
  return ad - 0.5f * beta;
}

__global__ __launch_bounds__(HL_PB) void saqe_extra_kernel(const SaqeExtra a) {
  __shared__ int last;
  const int np = a.b * a.k, K = a.k, C = a.c;
  float l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
  const float wmax = *a.box_w_max;
  for (int p = blockIdx.x * HL_PB + threadIdx.x; p < np; p += gridDim.x * HL_PB) {
    const int bi = p / K, kk = p % K;
    const float w = a.box_w[p], ow = a.obj_w[p];
    const int y = (int)a.obj_t[p], lab = (int)a.label[p], pick = a.sem_pick[p];
    // -- objectness of the quality head, both halves
    for (int hj = 0; hj < 2; ++hj) {
      const size_t o = ((size_t)bi * 2 * K + (size_t)hj * K + kk) * 2;
      const float z0 = a.robj[o], z1 = a.robj[o + 1];
      const float m = fmaxf(z0, z1);
      const float e0 = expf(z0 - m), e1 = expf(z1 - m), s = e0 + e1;
      const float lse = m + logf(s);
      const float cw = y ? a.cw1 : a.cw0;
      l0 += 0.5f * (a.w_obj * ((-cw * ((y ? z1 : z0) - lse)) * ow));
      const float gsc = 0.5f * a.w_obj * ow * cw;
      a.s_robj[o] = gsc * (e0 / s - (y ? 0.f : 1.f));
      a.s_robj[o + 1] = gsc * (e1 / s - (y ? 1.f : 0.f));
    }
    // -- heading: sin / cos smooth-L1
    const float th = a.bbox[(size_t)p * 7 + 6], tt = a.bbox_t[(size_t)p * 7 + 6];
    const float sp = sinf(th), cp = cosf(th);
    float gs, gc;
    const float ls = hl_smooth_l1(sp - sinf(tt), a.beta, &gs);
    const float lc = hl_smooth_l1(cp - cosf(tt), a.beta, &gc);
    const float ang = a.w_angle * (ls * w) + a.w_angle * (lc * w);
    const size_t rrow = ((size_t)bi * 2 * K + kk) * C;
    const float sc_rot = a.rot[rrow + pick];
    float ex = 1.f;
    if (a.sup) ex = expf(-(0.8f * sc_rot * sc_rot - 1.8f * sc_rot + 1.f));
    l1 += ex * ang;
    a.s_angle[p] = ex * (a.w_angle * w * (gs * cp - gc * sp));
    // -- angle quality (pre-training loss only)
    if (!a.sup) {
      const float lbl = ang / wmax;
      const float r0 = sc_rot - lbl;
      const float scj = a.rot[rrow + (size_t)K * C + pick];
      const float r1 = scj - lbl;
      l2 += a.w_apred * ((r0 * r0) * w) + a.w_apred * ((r1 * r1) * w);
      a.s_rot[rrow + pick] = a.w_apred * (2.f * r0) * w;
      a.s_rot[rrow + (size_t)K * C + pick] = a.w_apred * (2.f * r1) * w;
    }
    // -- side quality of the jittered proposals
    const float *tb = a.bbox_t + (size_t)p * 7;
    const size_t side_k = (size_t)2 * K;
    for (int i = 0; i < 6; ++i) {
      const float half = 0.5f * tb[3 + i % 3];
      const float ts = i < 3 ? tb[i] - half : tb[i - 3] + half;
      const float lbl = fminf(4.f * fabsf(a.jsurf[(size_t)p * 6 + i] - ts), 1.f);
      const float sc = a.side[(((size_t)i * a.b + bi) * C + lab) * side_k + K + kk];
      const float r = sc - lbl;
      l3 += a.w_side * ((r * r) * w);
      a.s_sidej[(size_t)p * 6 + i] = a.w_side * (2.f * r) * w;
    }
  }
  float v[4] = {l0, l1, l2, l3};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v[i] += __shfl_xor(v[i], off, 64);
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a.partial[(size_t)blockIdx.x * 4 + i] = v[i];
    __threadfence();
    last = atomicAdd(a.ticket, 1) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (threadIdx.x < 4) {
    float t = 0.f;
    for (unsigned w = 0; w < gridDim.x; ++w) t += __builtin_nontemporal_load(a.partial + (size_t)w * 4 + threadIdx.x);
    a.loss[threadIdx.x] = t;
  }
  if (threadIdx.x == 0) *a.ticket = 0;
}

// gradients of the four terms in the producers' layouts; d_side (6, B, C, 2K) must arrive zero-filled
__global__ __launch_bounds__(256) void saqe_extra_bwd_kernel(int b, int k, int c, const float *g,
                                                             const long long *label, const float *s_robj,
                                                             const float *s_angle, const float *s_rot,
                                                             const float *s_sidej, float *d_robj,
                                                             float *d_angle, float *d_rot, float *d_side) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= b * k) return;
  const int bi = p / k, kk = p % k;
  const float g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
  for (int hj = 0; hj < 2; ++hj) {
    const size_t o = ((size_t)bi * 2 * k + (size_t)hj * k + kk) * 2;
    d_robj[o] = g0 * s_robj[o];
    d_robj[o + 1] = g0 * s_robj[o + 1];
    const size_t r = ((size_t)bi * 2 * k + (size_t)hj * k + kk) * c;
    for (int j = 0; j < c; ++j) d_rot[r + j] = g2 * s_rot[r + j];
  }
  d_angle[p] = g1 * s_angle[p];
  const int lab = (int)label[p];
  const size_t side_k = (size_t)2 * k;
  for (int i = 0; i < 6; ++i)
    d_side[(((size_t)i * b + bi) * c + lab) * side_k + k + kk] = g3 * s_sidej[(size_t)p * 6 + i];
}

}  // namespace nesie

using namespace nesie;

extern "C" int nesie_head_targets(int b, int k, int t, const float *agg, const float *gt_boxes,
                                  const long long *gt_labels, const long long *gt_count,
                                  const float *gt_valid, float pos_thr, float neg_thr,
                                  long long *assignment, long long *obj_targets,
                                  float *obj_weights, long long *mask_targets,
                                  float *bbox_targets, float *center_targets,
                                  float *box_weights, float *valid_weights, void *stream) {
  const char *W = "head_targets";
  NESIE_REQUIRE(b >= 0 && k >= 1 && t >= 1, W);
  if (b == 0) return NESIE_OK;
  NESIE_REQUIRE(agg && gt_boxes && gt_labels && gt_count && gt_valid && assignment && obj_targets &&
                    obj_weights && mask_targets && bbox_targets && center_targets && box_weights &&
                    valid_weights, W);
  NESIE_REQUIRE((long long)b * k < (1 << 22) && (long long)b * t < (1 << 22), W);
  hipLaunchKernelGGL(head_targets_kernel, dim3(1), dim3(HL_BLOCK), 0, (hipStream_t)stream, b, k, t,
                     agg, gt_boxes, gt_labels, gt_count, gt_valid, pos_thr, neg_thr, assignment,
                     obj_targets, obj_weights, mask_targets, bbox_targets, center_targets,
                     box_weights, valid_weights);
  return check_launch(W);
}

static int head_loss_forward_impl(
    const char *W, const float *quality, int flags,
    int b, int k, int t, int c, const float *cls, const float *bbox, const float *surface,
    const float *side, const float *iou_s, const float *iou, const float *iou_j,
    const long long *obj_t, const long long *label, const float *obj_w, const float *box_w,
    const float *bbox_t, const float *centre_t, const float *valid_w,
    const float *config /* [11] */, float *loss, float *s_cls, float *s_centre,
    float *s_surface, float *s_iou, float *s_iou_s, float *s_side_surf, float *s_side_iou,
    float *s_side_pred, int *sem_pick, int *kstar, float *dmin, float *partial, int *ticket,
    void *stream) {
  NESIE_REQUIRE(b >= 0 && k >= 1 && t >= 1 && c >= 1 && c <= HL_MAXC, W);
  if (b == 0) return NESIE_OK;
  NESIE_REQUIRE(cls && bbox && surface && side && iou_s && iou && iou_j && obj_t && label && obj_w &&
                    box_w && bbox_t && centre_t && valid_w && config && loss, W);
  NESIE_REQUIRE(s_cls && s_centre && s_surface && s_iou && s_iou_s && s_side_surf && s_side_iou &&
                    s_side_pred && sem_pick && kstar && dmin && partial && ticket, W);
  NESIE_REQUIRE((long long)b * k < (1 << 22) && (long long)b * t < (1 << 22), W);
  // config is a HOST array: the eleven scalars travel as kernel arguments
  HeadLoss a;
  a.b = b; a.k = k; a.t = t; a.c = c;
  a.cls = cls; a.bbox = bbox; a.surface = surface; a.side = side; a.iou_s = iou_s;
  a.iou = iou; a.iou_j = iou_j;
  a.obj_t = obj_t; a.label = label; a.obj_w = obj_w; a.box_w = box_w; a.bbox_t = bbox_t;
  a.centre_t = centre_t; a.valid_w = valid_w;
  a.alpha = config[0]; a.w_obj = config[1]; a.cw0 = config[2]; a.cw1 = config[3];
  a.w_sem = config[4]; a.w_csrc = config[5]; a.w_cdst = config[6]; a.w_surf = config[7];
  a.w_iou = config[8]; a.w_qfl = config[9]; a.w_side = config[10];
  a.loss = loss; a.s_cls = s_cls; a.s_centre = s_centre; a.s_surface = s_surface; a.s_iou = s_iou;
  a.s_iou_s = s_iou_s; a.s_side_surf = s_side_surf; a.s_side_iou = s_side_iou;
  a.s_side_pred = s_side_pred; a.sem_pick = sem_pick; a.kstar = kstar; a.dmin = dmin;
  a.partial = partial; a.ticket = ticket;
  a.quality = quality; a.flags = flags;
  hipLaunchKernelGGL(head_loss_nearest_kernel, dim3((b * t + 63) / 64), dim3(64), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(head_loss_kernel, dim3((b * k + HL_PB - 1) / HL_PB), dim3(HL_PB), 0,
                     (hipStream_t)stream, a);
  return check_launch(W);
}

extern "C" int nesie_head_loss_forward(
    int b, int k, int t, int c, const float *cls, const float *bbox, const float *surface,
    const float *side, const float *iou_s, const float *iou, const float *iou_j,
    const long long *obj_t, const long long *label, const float *obj_w, const float *box_w,
    const float *bbox_t, const float *centre_t, const float *valid_w,
    const float *config /* [11] */, float *loss, float *s_cls, float *s_centre,
    float *s_surface, float *s_iou, float *s_iou_s, float *s_side_surf, float *s_side_iou,
    float *s_side_pred, int *sem_pick, int *kstar, float *dmin, float *partial, int *ticket,
    void *stream) {
  return head_loss_forward_impl("head_loss_forward", nullptr, 0, b, k, t, c, cls, bbox, surface, side, iou_s,
                                iou, iou_j, obj_t, label, obj_w, box_w, bbox_t, centre_t, valid_w, config,
                                loss, s_cls, s_centre, s_surface, s_iou, s_iou_s, s_side_surf, s_side_iou,
                                s_side_pred, sem_pick, kstar, dmin, partial, ticket, stream);
}

extern "C" int nesie_head_loss_forward_sigma(
    int sigma_mode /* 0 as nesie_head_loss_forward, 1 constant uncertainties, 2 none */,
    int b, int k, int t, int c, const float *cls, const float *bbox, const float *surface,
    const float *side, const float *iou_s, const float *iou, const float *iou_j,
    const long long *obj_t, const long long *label, const float *obj_w, const float *box_w,
    const float *bbox_t, const float *centre_t, const float *valid_w,
    const float *config /* [11] */, float *loss, float *s_cls, float *s_centre,
    float *s_surface, float *s_iou, float *s_iou_s, float *s_side_surf, float *s_side_iou,
    float *s_side_pred, int *sem_pick, int *kstar, float *dmin, float *partial, int *ticket,
    void *stream) {
  const char *W = "head_loss_forward_sigma";
  NESIE_REQUIRE(sigma_mode >= 0 && sigma_mode <= 2, W);
  const int flags = sigma_mode == 1 ? HL_DETACH_SIGMA : sigma_mode == 2 ? HL_NO_SIGMA : 0;
  return head_loss_forward_impl(W, nullptr, flags, b, k, t, c, cls, bbox, surface, side, iou_s,
                                iou, iou_j, obj_t, label, obj_w, box_w, bbox_t, centre_t, valid_w, config,
                                loss, s_cls, s_centre, s_surface, s_iou, s_iou_s, s_side_surf, s_side_iou,
                                s_side_pred, sem_pick, kstar, dmin, partial, ticket, stream);
}

extern "C" int nesie_head_loss_forward_unsup(
    int b, int k, int t, int c, const float *cls, const float *bbox, const float *surface,
    const float *side, const float *iou_s, const float *iou, const float *quality /* (B*K, 6) */,
    int detach_sigma, const long long *obj_t, const long long *label, const float *obj_w,
    const float *box_w, const float *bbox_t, const float *centre_t, const float *valid_w,
    const float *config /* [11] */, float *loss, float *s_cls, float *s_centre,
    float *s_surface, float *s_iou, float *s_iou_s, float *s_side_surf, float *s_side_iou,
    float *s_side_pred, int *sem_pick, int *kstar, float *dmin, float *partial, int *ticket,
    void *stream) {
  const char *W = "head_loss_forward_unsup";
  NESIE_REQUIRE(b == 0 || quality, W);
  return head_loss_forward_impl(W, quality, HL_UNSUP | (detach_sigma ? HL_DETACH_SIGMA : 0), b, k, t, c, cls,
                                bbox, surface, side, iou_s, iou, iou, obj_t, label, obj_w, box_w, bbox_t,
                                centre_t, valid_w, config, loss, s_cls, s_centre, s_surface, s_iou, s_iou_s,
                                s_side_surf, s_side_iou, s_side_pred, sem_pick, kstar, dmin, partial, ticket,
                                stream);
}

extern "C" int nesie_head_loss_backward(
    int b, int k, int c, const float *g, const long long *label, const int *sem_pick,
    const float *s_cls, const float *s_centre, const float *s_surface, const float *s_iou,
    const float *s_iou_s, const float *s_side_surf, const float *s_side_iou,
    const float *s_side_pred, float *d_cls, float *d_bbox, float *d_surface, float *d_iou,
    float *d_iou_s, float *d_side /* zero-filled */, void *stream) {
  const char *W = "head_loss_backward";
  NESIE_REQUIRE(b >= 0 && k >= 1 && c >= 1 && c <= HL_MAXC, W);
  if (b == 0) return NESIE_OK;
  NESIE_REQUIRE(g && label && sem_pick && s_cls && s_centre && s_surface && s_iou && s_iou_s &&
                    s_side_surf && s_side_iou && s_side_pred, W);
  NESIE_REQUIRE(d_cls && d_bbox && d_surface && d_iou && d_iou_s && d_side, W);
  HeadLossBwd a;
  a.b = b; a.k = k; a.c = c; a.g = g; a.label = label; a.sem_pick = sem_pick;
  a.s_cls = s_cls; a.s_centre = s_centre; a.s_surface = s_surface; a.s_iou = s_iou;
  a.s_iou_s = s_iou_s; a.s_side_surf = s_side_surf; a.s_side_iou = s_side_iou;
  a.s_side_pred = s_side_pred;
  a.d_cls = d_cls; a.d_bbox = d_bbox; a.d_surface = d_surface; a.d_iou = d_iou; a.d_iou_s = d_iou_s;
  a.d_side = d_side;
  hipLaunchKernelGGL(head_loss_bwd_kernel, dim3((b * k + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, a);
  return check_launch(W);
}

// ---- vote loss -----------------------------------------------------------------------------------
// VoteModule.get_loss (vote_module.py:149-180) for one vote per seed: every seed inside an object
// is pulled to the nearest (L1) of its up to three ground-truth centres,
//   loss = w_dst * sum_seeds [mask / (sum mask + 1e-6)] * min_g |vote - (seed + offset_g)|_1 .
// Forward: per-workgroup (sum, count) partials, the last workgroup folds them in order; the sign
// pattern of the winning target is kept for the backward, which is one scaled copy.
namespace nesie {
constexpr int VL_BLOCK = 256;

__global__ __launch_bounds__(VL_BLOCK) void vote_loss_kernel(
    int b, int n, long long npts, int gt_per_seed, const float *__restrict__ seed,
    const float *__restrict__ vote, const long long *__restrict__ seed_idx,
    const long long *__restrict__ mask, const float *__restrict__ targets, float w_dst,
    float *__restrict__ sign_out, float *__restrict__ loss, float *__restrict__ scale_out,
    float *__restrict__ partial, int *__restrict__ ticket) {
  __shared__ float sh[2][VL_BLOCK / 64];
  __shared__ int last;
  const int total = b * n;
  float s = 0.f, cnt = 0.f;
  for (int i = blockIdx.x * VL_BLOCK + threadIdx.x; i < total; i += gridDim.x * VL_BLOCK) {
    const int bi = i / n;
    const long long src = seed_idx[i];
    const float m = (float)mask[(size_t)bi * npts + src];
    const float *tg = targets + ((size_t)bi * npts + src) * (3 * gt_per_seed);
    const float sx = seed[i * 3], sy = seed[i * 3 + 1], sz = seed[i * 3 + 2];
    const float vx = vote[i * 3], vy = vote[i * 3 + 1], vz = vote[i * 3 + 2];
    float best = INFINITY, bx = 0.f, by = 0.f, bz = 0.f;
    for (int g = 0; g < gt_per_seed; ++g) {
      const float dx = vx - (tg[g * 3] + sx), dy = vy - (tg[g * 3 + 1] + sy), dz = vz - (tg[g * 3 + 2] + sz);
      const float d = (fabsf(dx) + fabsf(dy)) + fabsf(dz);
      if (d < best) { best = d; bx = dx; by = dy; bz = dz; }
    }
    s += m * best;
    cnt += m;
    // d|x|/dx with torch's sign(0) = 0
    sign_out[i * 3] = m * (float)((bx > 0.f) - (bx < 0.f));
    sign_out[i * 3 + 1] = m * (float)((by > 0.f) - (by < 0.f));
    sign_out[i * 3 + 2] = m * (float)((bz > 0.f) - (bz < 0.f));
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { s += __shfl_xor(s, off, 64); cnt += __shfl_xor(cnt, off, 64); }
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x * 2] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
    partial[blockIdx.x * 2 + 1] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    __threadfence();
    last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!last || threadIdx.x != 0) return;
  __threadfence();
  float ts = 0.f, tc = 0.f;
  for (unsigned w = 0; w < gridDim.x; ++w) { ts += partial[w * 2]; tc += partial[w * 2 + 1]; }
  const float scale = w_dst / (tc + 1e-6f);
  *loss = ts * scale;
  *scale_out = scale;
  *ticket = 0;
}

__global__ __launch_bounds__(256) void vote_loss_bwd_kernel(long long n3, const float *__restrict__ g,
                                                            const float *__restrict__ scale,
                                                            const float *__restrict__ sign,
                                                            float *__restrict__ d_vote) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n3) d_vote[i] = (*g * *scale) * sign[i];
}
}  // namespace nesie

extern "C" int nesie_vote_loss_forward(int b, int n, long long npts, int gt_per_seed, const float *seed,
                                       const float *vote, const long long *seed_idx,
                                       const long long *mask, const float *targets, float w_dst,
                                       float *sign_out, float *loss, float *scale_out,
                                       float *partial, int *ticket, void *stream) {
  const char *W = "vote_loss_forward";
  NESIE_REQUIRE(b >= 0 && n >= 1 && npts >= 1 && gt_per_seed >= 1, W);
  if (b == 0) return NESIE_OK;
  NESIE_REQUIRE(seed && vote && seed_idx && mask && targets && sign_out && loss && scale_out &&
                    partial && ticket, W);
  const int blocks = (b * n + VL_BLOCK - 1) / VL_BLOCK < 64 ? (b * n + VL_BLOCK - 1) / VL_BLOCK : 64;
  hipLaunchKernelGGL(vote_loss_kernel, dim3(blocks), dim3(VL_BLOCK), 0, (hipStream_t)stream, b, n, npts,
                     gt_per_seed, seed, vote, seed_idx, mask, targets, w_dst, sign_out, loss, scale_out,
                     partial, ticket);
  return check_launch(W);
}

extern "C" int nesie_vote_loss_backward(long long n3, const float *g, const float *scale,
                                        const float *sign, float *d_vote, void *stream) {
  const char *W = "vote_loss_backward";
  NESIE_REQUIRE(n3 >= 0, W);
  if (n3 == 0) return NESIE_OK;
  NESIE_REQUIRE(g && scale && sign && d_vote, W);
  hipLaunchKernelGGL(vote_loss_bwd_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, n3, g, scale, sign, d_vote);
  return check_launch(W);
}

// ---- the SAQE head's additional terms ---------------------------------------------------------------
extern "C" int nesie_saqe_extra_loss_forward(
    int b, int k, int c, int sup, const float *robj, const float *rot, const float *bbox,
    const float *bbox_t, const float *jsurf, const float *side, const long long *obj_t,
    const long long *label, const float *obj_w, const float *box_w, const float *box_w_max,
    const int *sem_pick, const float *config /* [7] */, float *loss, float *s_robj, float *s_angle,
    float *s_rot, float *s_sidej, float *partial, int *ticket, void *stream) {
  const char *W = "saqe_extra_loss_forward";
  NESIE_REQUIRE(b >= 0 && k >= 1 && c >= 1 && c <= HL_MAXC, W);
  if (b == 0) return NESIE_OK;
  NESIE_REQUIRE(robj && rot && bbox && bbox_t && jsurf && side && obj_t && label && obj_w && box_w &&
                    box_w_max && sem_pick && config && loss && s_robj && s_angle && s_rot && s_sidej &&
                    partial && ticket, W);
  NESIE_REQUIRE((long long)b * k < (1 << 22), W);
  SaqeExtra a;
  a.b = b; a.k = k; a.c = c; a.sup = sup;
  a.robj = robj; a.rot = rot; a.bbox = bbox; a.bbox_t = bbox_t; a.jsurf = jsurf; a.side = side;
  a.obj_t = obj_t; a.label = label; a.obj_w = obj_w; a.box_w = box_w; a.box_w_max = box_w_max;
  a.sem_pick = sem_pick;
  a.w_obj = config[0]; a.cw0 = config[1]; a.cw1 = config[2]; a.w_angle = config[3]; a.beta = config[4];
  a.w_apred = config[5]; a.w_side = config[6];
  a.loss = loss; a.s_robj = s_robj; a.s_angle = s_angle; a.s_rot = s_rot; a.s_sidej = s_sidej;
  a.partial = partial; a.ticket = ticket;
  hipLaunchKernelGGL(saqe_extra_kernel, dim3((b * k + HL_PB - 1) / HL_PB), dim3(HL_PB), 0,
                     (hipStream_t)stream, a);
  return check_launch(W);
}

extern "C" int nesie_saqe_extra_loss_backward(int b, int k, int c, const float *g, const long long *label,
                                              const float *s_robj, const float *s_angle,
                                              const float *s_rot, const float *s_sidej, float *d_robj,
                                              float *d_angle, float *d_rot, float *d_side /* zero-filled */,
                                              void *stream) {
  const char *W = "saqe_extra_loss_backward";
  NESIE_REQUIRE(b >= 0 && k >= 1 && c >= 1 && c <= HL_MAXC, W);
  if (b == 0) return NESIE_OK;
  NESIE_REQUIRE(g && label && s_robj && s_angle && s_rot && s_sidej && d_robj && d_angle && d_rot && d_side, W);
  hipLaunchKernelGGL(saqe_extra_bwd_kernel, dim3((b * k + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     b, k, c, g, label, s_robj, s_angle, s_rot, s_sidej, d_robj, d_angle, d_rot, d_side);
  return check_launch(W);
}
