/* nesie_head_ops.h -- C ABI of the Nesie head's training targets and loss terms (part of
 * libnesie_hip.so; conventions as in nesie_ops.h: plain pointers, sizes, a HIP stream handle, int
 * status, nesie_last_error()).
 *
 * Replaces, for the shipped Nesie-VoteNet configuration, the python of
 *   mmdet3d/models/dense_heads/nesie_head.py:566-588, 656-676  (get_targets: assignment, weights)
 *   mmdet3d/models/dense_heads/nesie_head.py:279-413           (loss: seven per-proposal terms)
 * and the loss classes it calls (losses/{chamfer_distance,surface_loss,side_pred_loss,
 * iou3d_loss,gfocal_loss}.py, mmdet CrossEntropyLoss).  The reference has no native entry for
 * these; a maintainer binds them from NesieHead.loss (see nesie_amd/votenet/head_loss.py). */
#ifndef NESIE_HEAD_OPS_H
#define NESIE_HEAD_OPS_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Per-point vote targets of NesieHead.get_targets_single (nesie_head.py:593-654) for a batch:
 *   gt_boxes (B,T,7) DEPTH-frame bottom-centre boxes, gt_count (B) int64 = columns in use,
 *   points (B,N,pt_stride) with xyz in the first three floats
 * -> vote_targets (B,N,9): gravity centre minus point for the first / second / last box holding
 *    the point (empty slots repeat the first; zeros when in no box), vote_target_masks (B,N)
 *    int64 = [in some box].  Same in-box test as nesie_points_in_boxes_batch on the LiDAR-frame
 *    operands of depth_box3d.py:263-266. */
int nesie_vote_targets(int b, int boxes_num, int pts_num, int pt_stride, const float *gt_boxes,
                       const long long *gt_count, const float *points, float *vote_targets,
                       long long *vote_target_masks, void *stream);

/* Proposal <-> ground-truth assignment and the batch-level weights.
 *   agg (B,K,3) aggregated points; gt_boxes (B,T,7) bottom-centre boxes; gt_labels (B,T) int64;
 *   gt_count (B) int64 = box columns in use; gt_valid (B,T) 1 for real boxes.
 * -> assignment (B,K) int64 = nearest used column by squared distance to the box CENTRE (first
 *    minimum), obj_targets (B,K) int64 = [sqrt(d + 1e-6) < pos_thr], obj_weights = [pos or
 *    sqrt(d + 1e-6) > neg_thr] / (their number + 1e-6), box_weights = pos / (number of pos +
 *    1e-6), mask_targets (B,K) int64 = the assigned label, bbox_targets (B,K,7) = (centre, size,
 *    yaw) of the assigned box, center_targets (B,T,3) (zero for unused columns), valid_weights
 *    (B,T) = gt_valid / (number valid + 1e-6). */
int nesie_head_targets(int b, int k, int t, const float *agg, const float *gt_boxes,
                       const long long *gt_labels, const long long *gt_count,
                       const float *gt_valid, float pos_thr, float neg_thr,
                       long long *assignment, long long *obj_targets, float *obj_weights,
                       long long *mask_targets, float *bbox_targets, float *center_targets,
                       float *box_weights, float *valid_weights, void *stream);

/* The seven loss terms, loss[7] = (objectness, semantic, centre, surface, iou, iou_pred, side),
 * and the gradient of each term w.r.t. its inputs (s_* outputs, unit incoming gradient).  Inputs
 * are read in their producers' layouts (no copies):
 *   cls (B, 2 + C, K): rows 0..1 objectness logits, rows 2.. class logits (conv_pred output);
 *   bbox (B,K,7) decoded boxes (columns 0..2 = centre); surface (B,K,6);
 *   side (6, B, C, 2K) side-quality PROBABILITIES, plain proposals then jittered copies;
 *   iou_s (B, 2K, C) IoU-quality probabilities, plain then jittered;
 *   iou / iou_j (B*K) IoU of the predicted / jittered box with its target box.
 *   config: HOST array of 11 floats = alpha, objectness weight, objectness class weights (2),
 *   semantic weight, chamfer source / destination weights, surface, iou, iou_pred, side weights.
 * s_cls (B,2+C,K), s_centre (B*K,3), s_surface (B*K,6), s_iou (B*K), s_iou_s (B,2K,C); the three
 * s_side_* (B*K,6) are the side-score gradients of the surface term and the iou term (both at
 * class sem_pick[p] = arg-max class logit) and of the side term (at class label[p]).
 * kstar (B*T) int32, dmin (B*T), partial (ceil(B*K / 64), 8): scratch; ticket: one int32 that is
 * ZERO before the first call (the kernel leaves it zero). */
int nesie_head_loss_forward(int b, int k, int t, int c, const float *cls, const float *bbox,
                            const float *surface, const float *side, const float *iou_s,
                            const float *iou, const float *iou_j, const long long *obj_t,
                            const long long *label, const float *obj_w, const float *box_w,
                            const float *bbox_t, const float *centre_t, const float *valid_w,
                            const float *config, float *loss, float *s_cls, float *s_centre,
                            float *s_surface, float *s_iou, float *s_iou_s, float *s_side_surf,
                            float *s_side_iou, float *s_side_pred, int *sem_pick, int *kstar,
                            float *dmin, float *partial, int *ticket, void *stream);

/* The unsupervised variant (NesieHead.unsup_loss, nesie_head.py:415-509; SAQEHead.unsup_loss,
 * saqe_head.py:706-800): the same kernel with quality (B*K, 6) = the pseudo label's six side
 * qualities gathered per proposal (zeros for padding).  Semantic and centre terms as above; the
 * surface term of side i is weighted box_w * quality[i], the IoU term box_w * mean(quality); the
 * objectness, IoU-quality and side-quality terms are not part of this loss (their outputs and saved
 * gradients are zero).  detach_sigma != 0: the uncertainties are constants (SAQE: sigma.detach()).
 * The caller applies the reference's un_label_weight (2.0).  Outputs and scratch as above;
 * nesie_head_loss_backward serves both. */
int nesie_head_loss_forward_unsup(int b, int k, int t, int c, const float *cls, const float *bbox,
                                  const float *surface, const float *side, const float *iou_s,
                                  const float *iou, const float *quality, int detach_sigma,
                                  const long long *obj_t, const long long *label, const float *obj_w,
                                  const float *box_w, const float *bbox_t, const float *centre_t,
                                  const float *valid_w, const float *config, float *loss,
                                  float *s_cls, float *s_centre, float *s_surface, float *s_iou,
                                  float *s_iou_s, float *s_side_surf, float *s_side_iou,
                                  float *s_side_pred, int *sem_pick, int *kstar, float *dmin,
                                  float *partial, int *ticket, void *stream);

/* The supervised terms the SAQE head shares with the Nesie head (saqe_head.py:331-703), with the
 * head's treatment of the uncertainties: sigma_mode 1 = exp(-sigma) weights with sigma a CONSTANT
 * (sup_loss: sigma.detach(), config alpha 0), 2 = no uncertainty weighting (loss). */
int nesie_head_loss_forward_sigma(int sigma_mode, int b, int k, int t, int c, const float *cls,
                                  const float *bbox, const float *surface, const float *side,
                                  const float *iou_s, const float *iou, const float *iou_j,
                                  const long long *obj_t, const long long *label, const float *obj_w,
                                  const float *box_w, const float *bbox_t, const float *centre_t,
                                  const float *valid_w, const float *config, float *loss,
                                  float *s_cls, float *s_centre, float *s_surface, float *s_iou,
                                  float *s_iou_s, float *s_side_surf, float *s_side_iou,
                                  float *s_side_pred, int *sem_pick, int *kstar, float *dmin,
                                  float *partial, int *ticket, void *stream);

/* The SAQE head's ADDITIONAL supervised terms (saqe_head.py:331-521 `loss`, :524-703 `sup_loss`):
 *   loss[0] = 0.5 (CE(R_obj) + CE(R_obj_jitter)), class- and proposal-weighted like the objectness term;
 *   loss[1] = sum box_w * w_angle * (SmoothL1(sin th - sin th*) + SmoothL1(cos th - cos th*)), in
 *             sup mode times exp(-angle_sigma), angle_sigma = 0.8 a^2 - 1.8 a + 1 of the rotate score
 *             at the arg-max class (sem_pick, from the forward above), a constant;
 *   loss[2] = (sup == 0 only) MSE of the plain and the jittered rotate score at that class against
 *             angle_term / *box_w_max, weight box_w;
 *   loss[3] = side-quality loss of the JITTERED proposals (label from jsurf = Bbox2Surface of the
 *             jittered boxes, score = side[.., label class, K + k]).
 * robj (B, 2K, 2) logits, rot (B, 2K, C) probabilities, bbox / bbox_t (B*K, 7), jsurf (B*K, 6),
 * side (6, B, C, 2K); config: HOST array of 7 floats = objectness weight, its two class weights,
 * angle weight, SmoothL1 beta, angle-quality weight, side weight.  s_rot must arrive zero-filled.
 * partial (ceil(B*K / 64), 4), ticket as above.  Backward: d_robj, d_angle (B*K, for column 6 of
 * the boxes), d_rot, d_side (zero-filled) = the saved gradients times g[4]. */
int nesie_saqe_extra_loss_forward(int b, int k, int c, int sup, const float *robj, const float *rot,
                                  const float *bbox, const float *bbox_t, const float *jsurf,
                                  const float *side, const long long *obj_t, const long long *label,
                                  const float *obj_w, const float *box_w, const float *box_w_max,
                                  const int *sem_pick, const float *config, float *loss,
                                  float *s_robj, float *s_angle, float *s_rot, float *s_sidej,
                                  float *partial, int *ticket, void *stream);
int nesie_saqe_extra_loss_backward(int b, int k, int c, const float *g, const long long *label,
                                   const float *s_robj, const float *s_angle, const float *s_rot,
                                   const float *s_sidej, float *d_robj, float *d_angle, float *d_rot,
                                   float *d_side, void *stream);

/* Gradient assembly: the saved per-term gradients times the incoming gradients g[7] (device) of
 * the seven terms, in the producers' layouts: d_cls (B,2+C,K), d_bbox (B,K,7) (size and yaw
 * columns zero), d_surface, d_iou (B*K), d_iou_s (B,2K,C), d_side (6,B,C,2K) which must arrive
 * ZERO-FILLED (two class columns per proposal and side are written). */
int nesie_head_loss_backward(int b, int k, int c, const float *g, const long long *label,
                             const int *sem_pick, const float *s_cls, const float *s_centre,
                             const float *s_surface, const float *s_iou, const float *s_iou_s,
                             const float *s_side_surf, const float *s_side_iou,
                             const float *s_side_pred, float *d_cls, float *d_bbox,
                             float *d_surface, float *d_iou, float *d_iou_s, float *d_side,
                             void *stream);

/* VoteModule.get_loss (vote_module.py:149-180) for one vote per seed:
 *   loss = w_dst * sum_seeds [mask / (sum mask + 1e-6)] * min_g |vote - (seed + targets_g)|_1
 * seed, vote (B,N,3); seed_idx (B,N) int64 into the N_pts input points; mask (B,N_pts) int64;
 * targets (B,N_pts,3*gt_per_seed) offsets.  -> loss (device scalar), sign_out (B,N,3) = mask *
 * sign(vote - nearest target), scale_out = w_dst / (sum mask + 1e-6); partial (64, 2) scratch,
 * ticket as in nesie_head_loss_forward.  Backward: d_vote = g * scale * sign. */
int nesie_vote_loss_forward(int b, int n, long long npts, int gt_per_seed, const float *seed,
                            const float *vote, const long long *seed_idx, const long long *mask,
                            const float *targets, float w_dst, float *sign_out, float *loss,
                            float *scale_out, float *partial, int *ticket, void *stream);
int nesie_vote_loss_backward(long long n3, const float *g, const float *scale, const float *sign,
                             float *d_vote, void *stream);

/* SidePooling.dist_feature (side_pooling_module.py:245-264) in one launch: probs (B, 6, bins, K)
 * side-bin distributions -> out (6, B, bins + 5, copies * K) = per face the bins, their four
 * largest values (descending) and their unbiased variance, repeated `copies` times along the
 * proposal axis.  bins >= 5. */
int nesie_side_prob_stats(int b, int bins, int kprop, int copies, const float *probs, float *out,
                          void *stream);

/* NesieHead.jitter_bbox_preds (nesie_head.py:178-209; SAQE variant with size_bias) in one launch:
 * bbox (B,K,7), noise_c / noise_s (B,K,3) -> centre_all, size_all (B,2K,3) = [original,
 * jittered], heading_all (B,2K) (zero when zero_heading), jitter_bbox (B,K,7).
 * centre_j = c + (s n_c) sigma; size_j = max(s + (s n_s) sigma, 1e-8) (size_bias == 0) or
 * max(s + s (n_s sigma + size_bias), 1e-8). */
int nesie_proposal_jitter(int b, int k, const float *bbox, const float *noise_c,
                          const float *noise_s, float sigma, float size_bias, int zero_heading,
                          float *centre_all, float *size_all, float *heading_all,
                          float *jitter_bbox, void *stream);

/* Clip-by-global-norm + AdamW over ONE flat parameter vector (dp.FlatTrainState), two launches,
 * no host round trip: torch.nn.utils.clip_grad_norm_(max_norm, 2) followed by torch.optim.AdamW's
 * update (mmcv OptimizerHook grad_clip + the reference's AdamW schedule).  step: device scalar
 * (float), incremented here; grad is left clipped; grad_norm_out (device scalar) or NULL.
 * max_norm <= 0: no clipping.  workspace: nesie_flat_adamw_workspace_bytes(). */
size_t nesie_flat_adamw_workspace_bytes(void);
int nesie_flat_adamw_step(long long n, float *param, float *grad, float *exp_avg,
                          float *exp_avg_sq, float *step, float lr, float beta1, float beta2,
                          float eps, float weight_decay, float max_norm, float *grad_norm_out,
                          void *workspace, size_t workspace_bytes, void *stream);
/* The same step with the learning rate and the weight decay read from DEVICE memory at execution
 * time (hyper = [lr, weight_decay]): a launch captured in a hipGraph then follows the step-decay
 * schedule of the reference configs (pretrain-010.py:112-114, lr x 0.1 at epochs 24 / 32) without
 * being re-captured -- the scheduler rewrites the two floats between replays. */
int nesie_flat_adamw_step_dev(long long n, float *param, float *grad, float *exp_avg,
                              float *exp_avg_sq, float *step, const float *hyper, float beta1,
                              float beta2, float eps, float max_norm, float *grad_norm_out,
                              void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif
