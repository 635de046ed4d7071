"""SAQE bbox head (``mmdet3d/models/dense_heads/saqe_head.py``), restated as the
differences from :class:`NesieHead` (the reference file is a copy of nesie_head.py with
these edits):

* the prediction layer emits 6*33 side bins + 3 learned log-scales + 12 angle bins
  (``:163-172``); side offsets are scaled by ``exp(log_scale)`` and the heading is the
  expectation over the 12 bins times 2*pi, wrapped to (-pi, pi]  (``:191-218``);
* wider proposal jitter (0.5 sigma, +0.2 size bias) and the jittered boxes' planes are kept
  (``:220-253``);
* quality head = :class:`QualityEstimation` with rotation scores and a second objectness;
* ``loss`` (pre-training): no uncertainty weighting; adds SmoothL1 on sin/cos of the
  heading, an MSE "angle quality" loss, three objectness terms and the jittered side loss
  (``:331-521``); ``sup_loss`` (semi-sup): exp(-sigma.detach()) weighting without the
  alpha*sigma regulariser (``:524-703``); ``unsup_loss`` likewise (``:706-800``).
"""
import math

import torch
import torch.nn as nn
from torch.nn import functional as F

from ..mmdet3d_ops.rotated_iou import cal_iou_3d
from .bbox_module import ReliableConvBboxHead
from .losses import Bbox2Surface, build_loss
from .nesie_head import Integral, NesieHead
from .quality_estimation import QualityEstimation


class AngleIntegral(nn.Module):
    """Expectation of a softmax over reg_max+1 angle bins in [0, 1] (``:54-88``)."""

    def __init__(self, reg_max=16):
        super().__init__()
        self.reg_max = reg_max
        self.register_buffer('project',
                             torch.linspace(0, self.reg_max, self.reg_max + 1) / self.reg_max)

    def forward(self, x):
        x = F.softmax(x.reshape(-1, self.reg_max + 1), dim=1)
        return F.linear(x, self.project.type_as(x)).reshape(-1, 1)


class SAQEHead(NesieHead):
    def __init__(self, num_classes, angle_loss=None, angle_pred_loss=None, grid_conv_cfg=None,
                 pred_layer_cfg=None, **kw):
        super().__init__(num_classes, grid_conv_cfg=grid_conv_cfg, pred_layer_cfg=pred_layer_cfg,
                         **kw)
        self.angle_loss = build_loss(angle_loss)
        self.angle_pred_loss = build_loss(angle_pred_loss)
        self.head_reg_outs = 12
        self.conv_pred = ReliableConvBboxHead(
            **pred_layer_cfg, num_cls_out_channels=self.num_classes + 2,
            num_bbox_out_channels=self.n_reg_outs + 3,
            num_heading_out_channels=self.head_reg_outs, reg_max=self.reg_max)
        self.angle_integral = AngleIntegral(self.head_reg_outs - 1)
        self.grid_conv = QualityEstimation(**grid_conv_cfg)
        self.jitter_sigma, self.jitter_size_bias = 0.5, 0.2

    # ---- decode ---------------------------------------------------------------------
    def side2box(self, aggregated_points, bbox_pred, results):
        B, proposal_num = bbox_pred.shape[:2]
        n = self.n_reg_outs
        res = self.integral(bbox_pred[..., :n]).reshape(B, proposal_num, -1)
        scale3 = torch.exp(bbox_pred[..., n:n + 3])
        scale = torch.cat([scale3, scale3], dim=-1)
        sign = self._side_sign.to(bbox_pred.dtype)
        results['surface_scale'] = scale
        surface = aggregated_points.repeat(1, 1, 2) + sign * (res * scale)
        results['surface_pred'] = surface
        angles = self.angle_integral(bbox_pred[..., n + 3:]).reshape(B, proposal_num) * 2 * math.pi
        angles = torch.where(angles > math.pi, angles - 2 * math.pi, angles)
        lo, hi = surface[..., :3], surface[..., 3:]
        results['bbox_preds'] = torch.cat([(lo + hi) / 2.0, hi - lo, angles.unsqueeze(-1)], dim=-1)
        return results

    def jitter_bbox_preds(self, results, dataset_name):
        bp = results['bbox_preds']
        center, size, heading = bp[..., :3], bp[..., 3:6], bp[..., -1]
        if self.jitter_noise is not None:
            n_c, n_s = (t.to(size) for t in self.jitter_noise)
        else:
            n_c, n_s = torch.randn_like(size), torch.randn_like(size)
        center_jitter = center + size * (n_c * 0.5)
        size_jitter = torch.clamp(size + size * (n_s * 0.5 + 0.2), min=1e-8)
        center_all = torch.cat([center, center_jitter], dim=1)
        size_all = torch.cat([size, size_jitter], dim=1)
        heading_all = torch.cat([heading, heading], dim=1)
        if dataset_name == 'ScanNet':
            heading_all = torch.zeros_like(heading_all)
        results['jitter_bbox_preds'] = torch.cat(
            [center_jitter, size_jitter, heading.unsqueeze(-1)], dim=-1)
        results['jitter_surface_preds'] = Bbox2Surface(results['jitter_bbox_preds'])
        return center_all, size_all, heading_all, results

    def forward(self, feat_dict, sample_mod, dataset_name='ScanNet'):
        results = super().forward(feat_dict, sample_mod, dataset_name)
        k = results['iou_scores'].shape[1]
        rot = results['rotate_scores'].sigmoid()
        results['rotate_scores_jitter'] = rot[:, k:]
        results['rotate_scores'] = rot[:, :k]
        robj = results['R_obj_scores']
        results['_rot_all'], results['_robj_all'] = rot, robj       # for the fused loss kernels
        results['R_obj_scores_jitter'] = robj[:, k:]
        results['R_obj_scores'] = robj[:, :k]
        return results

    # ---- shared pieces of the three losses -----------------------------------------------
    def _objectness(self, bbox_preds, objectness_targets, objectness_weights):
        l1 = self.objectness_loss(bbox_preds['obj_scores'].transpose(2, 1), objectness_targets,
                                  weight=objectness_weights)
        l2 = self.objectness_loss(bbox_preds['R_obj_scores'].transpose(2, 1), objectness_targets,
                                  weight=objectness_weights)
        l3 = self.objectness_loss(bbox_preds['R_obj_scores_jitter'].transpose(2, 1),
                                  objectness_targets, weight=objectness_weights)
        return l1 + (l2 + l3) * 0.5

    def _angle_terms(self, bbox_preds, bbox_targets_cat, w):
        pred_angle = bbox_preds['bbox_preds'][..., -1].reshape(-1)
        target_angle = bbox_targets_cat[..., -1]
        sin_l = self.angle_loss(torch.sin(pred_angle), torch.sin(target_angle), weight=w,
                                reduction_override='none')
        cos_l = self.angle_loss(torch.cos(pred_angle), torch.cos(target_angle), weight=w,
                                reduction_override='none')
        return sin_l + cos_l

    def _side_terms(self, bbox_preds, label_cls, bbox_targets_cat, probs, surface_weight):
        def one(scores, planes):
            side_pred = self._pick_class(scores.reshape(-1, 6, self.num_classes), label_cls)
            return self.side_loss(side_pred, planes.reshape(-1, 6).detach(), bbox_targets_cat,
                                  bbox_preds['surface_scale'].reshape(-1, 6),
                                  bbox_preds['aggregated_points'].reshape(-1, 3), probs,
                                  weight=surface_weight)
        return one(bbox_preds['side_scores'], bbox_preds['surface_pred']) \
            + one(bbox_preds['side_scores_jitter'], bbox_preds['jitter_surface_preds'])

    def _iou_pred(self, bbox_preds, bbox_targets_cat, mask_targets, box_loss_weights):
        targets_b = bbox_targets_cat.view_as(bbox_preds['bbox_preds'])
        label_iou = cal_iou_3d(bbox_preds['bbox_preds'], targets_b).detach().view(-1)
        label_iou_j = cal_iou_3d(bbox_preds['jitter_bbox_preds'], targets_b).detach().view(-1)
        label_cls = mask_targets.reshape(-1)
        w = box_loss_weights.reshape(-1)
        a = self.iou_pred_loss(bbox_preds['iou_scores'].reshape(-1, self.num_classes),
                               (label_cls, label_iou), weight=w)
        b = self.iou_pred_loss(bbox_preds['iou_scores_jitter'].reshape(-1, self.num_classes),
                               (label_cls, label_iou_j), weight=w)
        return a + b, label_cls, targets_b

    def _common(self, bbox_preds, points, gt_bboxes_3d, gt_labels_3d, sem, ins,
                vote_targets=None):
        targets = self.get_targets(points, gt_bboxes_3d, gt_labels_3d, sem, ins, bbox_preds,
                                   vote_targets=vote_targets)
        (vote_targets, vote_target_masks, center_targets, bbox_targets, mask_targets,
         valid_gt_masks, objectness_targets, objectness_weights, box_loss_weights,
         valid_gt_weights, assignment) = targets
        cat = bbox_targets.reshape(-1, 7)
        out = dict(
            vote_loss=self.vote_module.get_loss(
                bbox_preds['seed_points'], bbox_preds['vote_points'], bbox_preds['seed_indices'],
                vote_target_masks, vote_targets),
            objectness_loss=self._objectness(bbox_preds, objectness_targets, objectness_weights))
        s2t, t2s = self.center_loss(bbox_preds['bbox_preds'][..., :3], center_targets,
                                    src_weight=box_loss_weights, dst_weight=valid_gt_weights)
        out['center_loss'] = s2t + t2s
        out['semantic_loss'] = self.semantic_loss(bbox_preds['sem_scores'].transpose(2, 1),
                                                  mask_targets, weight=box_loss_weights)
        surface_weight = box_loss_weights.reshape(-1).unsqueeze(-1).repeat(1, 6)
        probs = bbox_preds['bbox_probs'].permute(0, 3, 1, 2).reshape(-1, 6, self.reg_max + 1)
        surface = self.surface_loss(
            bbox_preds['surface_pred'].reshape(-1, 6), cat,
            bbox_preds['surface_scale'].reshape(-1, 6),
            bbox_preds['aggregated_points'].reshape(-1, 3), probs, weight=surface_weight,
            reduction_override='none')
        iou_weight = box_loss_weights.reshape(-1)
        iou = self.iou_loss(bbox_preds['bbox_preds'].reshape(-1, 7), cat, weight=iou_weight,
                            reduction_override='none').reshape(-1)
        angle = self._angle_terms(bbox_preds, cat, iou_weight)
        out['iou_pred_loss'], label_cls, targets_b = self._iou_pred(
            bbox_preds, cat, mask_targets, box_loss_weights)
        out['side_loss'] = self._side_terms(bbox_preds, label_cls, cat, probs, surface_weight)
        return out, surface, iou, angle, box_loss_weights, targets_b

    def _fused_supervised(self, bbox_preds, points, gt_bboxes_3d, gt_labels_3d, sem, ins, sup,
                          ret_target, vote_targets=None):
        """``loss`` (sup False) / ``sup_loss`` (True) with every per-proposal term in two launches:
        the terms shared with the Nesie head through ``head_loss.HeadLossFn`` (no / constant
        uncertainties) and the SAQE head's own through ``head_loss.SaqeExtraFn``; the vote term and
        the two rotated-IoU evaluations stay."""
        from . import head_loss
        targets = self.get_targets(points, gt_bboxes_3d, gt_labels_3d, sem, ins, bbox_preds,
                                   vote_targets=vote_targets)
        (vote_targets, vote_target_masks, center_targets, bbox_targets, mask_targets,
         valid_gt_masks, objectness_targets, objectness_weights, box_loss_weights,
         valid_gt_weights, assignment) = targets
        vote_loss = self.vote_module.get_loss(
            bbox_preds['seed_points'], bbox_preds['vote_points'], bbox_preds['seed_indices'],
            vote_target_masks, vote_targets)
        boxes = bbox_preds['bbox_preds'].contiguous()
        iou = cal_iou_3d(boxes, bbox_targets)                              # (B, K), with gradient
        iou_jitter = cal_iou_3d(bbox_preds['jitter_bbox_preds'], bbox_targets).detach()
        tg = dict(obj_targets=objectness_targets, mask_targets=mask_targets,
                  obj_weights=objectness_weights, box_weights=box_loss_weights,
                  bbox_targets=bbox_targets, center_targets=center_targets,
                  valid_weights=valid_gt_weights)
        base_cfg, extra_cfg = head_loss.saqe_config_of(self)
        side_all = bbox_preds['_side_all'].contiguous()
        out = head_loss.HeadLossFn.apply(
            bbox_preds['_cls_all'].contiguous(), boxes, bbox_preds['surface_pred'].contiguous(),
            side_all, bbox_preds['_iou_all'].contiguous(), iou, iou_jitter, tg, base_cfg,
            self._loss_ticket, None, False, 1 if sup else 2)
        base, sem_pick = dict(zip(head_loss.TERMS, out[:7])), out[7]
        extra = dict(zip(head_loss.SAQE_TERMS, head_loss.SaqeExtraFn.apply(
            bbox_preds['_robj_all'].contiguous(), bbox_preds['_rot_all'].contiguous(), boxes, side_all,
            bbox_targets.contiguous(), bbox_preds['jitter_surface_preds'].detach().contiguous(), tg,
            sem_pick, extra_cfg, sup, self._loss_ticket)))
        losses = dict(vote_loss=vote_loss,
                      objectness_loss=base['objectness_loss'] + extra['r_objectness_loss'],
                      center_loss=base['center_loss'], semantic_loss=base['semantic_loss'],
                      iou_pred_loss=base['iou_pred_loss'],
                      side_loss=base['side_loss'] + extra['side_jitter_loss'])
        if sup:
            losses.update(surface_loss=base['surface_loss'], angle_loss=extra['angle_loss'],
                          iou_loss=base['iou_loss'])
        else:
            losses.update(surface_loss=base['surface_loss'], iou_loss=base['iou_loss'],
                          angle_loss=extra['angle_loss'], angle_pred_loss=extra['angle_pred_loss'])
        if ret_target:
            losses['targets'] = bbox_targets.view_as(boxes)
        return losses

    # ---- pre-training loss (:331-521) ------------------------------------------------------
    def loss(self, bbox_preds, points, gt_bboxes_3d, gt_labels_3d, pts_semantic_mask=None,
             pts_instance_mask=None, img_metas=None, gt_bboxes_ignore=None, ret_target=False,
             vote_targets=None):
        from . import head_loss
        if head_loss.saqe_usable(self, bbox_preds):
            return self._fused_supervised(bbox_preds, points, gt_bboxes_3d, gt_labels_3d,
                                          pts_semantic_mask, pts_instance_mask, False, ret_target,
                                          vote_targets)
        out, surface, iou, angle, blw, targets_b = self._common(
            bbox_preds, points, gt_bboxes_3d, gt_labels_3d, pts_semantic_mask, pts_instance_mask,
            vote_targets=vote_targets)
        out['surface_loss'] = surface.sum()
        out['iou_loss'] = iou.sum()
        out['angle_loss'] = angle.sum()
        w = blw.reshape(-1)
        angle_score_labels = angle.detach() / blw.max()
        indx = bbox_preds['sem_scores'].max(dim=-1)[1].reshape(-1, 1)
        n_class = bbox_preds['rotate_scores'].shape[-1]
        s = bbox_preds['rotate_scores'].reshape(-1, n_class).gather(1, indx).squeeze(-1)
        sj = bbox_preds['rotate_scores_jitter'].reshape(-1, n_class).gather(1, indx).squeeze(-1)
        out['angle_pred_loss'] = self.angle_pred_loss(s, angle_score_labels, weight=w) \
            + self.angle_pred_loss(sj, angle_score_labels, weight=w)
        if ret_target:
            out['targets'] = targets_b
        return out

    # ---- supervised loss of the semi-supervised stage (:524-703) ------------------------------
    def sup_loss(self, bbox_preds, points, gt_bboxes_3d, gt_labels_3d, pts_semantic_mask=None,
                 pts_instance_mask=None, img_metas=None, gt_bboxes_ignore=None,
                 ret_target=False):
        from . import head_loss
        if head_loss.saqe_usable(self, bbox_preds):
            return self._fused_supervised(bbox_preds, points, gt_bboxes_3d, gt_labels_3d,
                                          pts_semantic_mask, pts_instance_mask, True, ret_target)
        out, surface, iou, angle, blw, targets_b = self._common(
            bbox_preds, points, gt_bboxes_3d, gt_labels_3d, pts_semantic_mask, pts_instance_mask)
        sigma = self._sigma(bbox_preds)
        out['surface_loss'] = (torch.exp(-sigma.detach()) * surface).sum()
        indx = bbox_preds['sem_scores'].max(dim=-1)[1].reshape(-1, 1)
        n_class = bbox_preds['rotate_scores'].shape[-1]
        a = bbox_preds['rotate_scores'].reshape(-1, n_class).gather(1, indx).squeeze(-1)
        angle_sigma = 0.8 * a * a - 1.8 * a + torch.ones_like(a)
        out['angle_loss'] = (torch.exp(-angle_sigma.detach()) * angle).sum()
        out['iou_loss'] = (torch.exp(-sigma.mean(dim=-1).detach()) * iou).sum()
        if ret_target:
            out['targets'] = targets_b
        return out

    # ---- unsupervised loss (:706-800): the Nesie one with detached sigma, no regulariser ------
    def unsup_loss(self, bbox_preds, points, pseudo_boxes, pseudo_label, img_metas=None,
                   pseudo_quality_score=None):
        alpha, self.alpha = self.alpha, 0.0
        sig = self._sigma
        self._sigma = lambda bp: sig(bp).detach()
        self._sigma_is_constant = True        # (read by the fused form, NesieHead._fused_unsup_loss)
        try:
            return super().unsup_loss(bbox_preds, points, pseudo_boxes, pseudo_label, img_metas,
                                      pseudo_quality_score)
        finally:
            self.alpha = alpha
            del self._sigma
            del self._sigma_is_constant
