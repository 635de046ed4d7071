"""Nesie bbox head (``mmdet3d/models/dense_heads/nesie_head.py:19-799``): voting, vote
aggregation, side-distribution box decoding, jittered proposals, the side-aware
quality head, target assignment and the per-side localisation-uncertainty losses.

Differences from the reference, none of which change a number it computes:
  * the B*256-iteration python list comprehensions that pick one class column per
    proposal (:346, :386, :481, :709) are single ``gather`` calls;
  * ``get_targets`` runs batched on the device (one points-in-boxes launch for the
    batch, closed-form vote slots instead of the python loop over GT boxes, :629-652),
    so the step has no device->host synchronisation;
  * the Gaussian jitter is drawn on the device (the reference draws on the host and
    copies, :185-186); tests inject the noise through ``jitter_noise``.
"""
import torch
import torch.nn as nn
from torch.nn import functional as F

from ..mmdet3d_ops import (build_sa_module, furthest_point_sample, points_in_boxes_batch,
                           points_in_boxes_count)
from ..post_processing import batched_aligned_3d_nms
from ..kernels import backend_for
from ..mmdet3d_ops.rotated_iou import cal_iou_3d
from .bbox_module import ReliableConvBboxHead
from .boxes import DepthInstance3DBoxes, depth_to_lidar_boxes, depth_to_lidar_points
from . import head_loss
from .losses import build_loss
from ..streams import fork_join
from .side_pooling import SidePooling
from .vote_module import VoteModule


class Integral(nn.Module):
    """Expectation of a softmax over reg_max+1 bins, scaled to [0, 1] (:19-52)."""

    def __init__(self, reg_max=16):
        super().__init__()
        self.reg_max = reg_max
        self.register_buffer('project',
                             torch.linspace(0, self.reg_max, self.reg_max + 1) / self.reg_max)

    def forward(self, x):
        x = F.softmax(x.reshape(-1, self.reg_max + 1), dim=1)
        return F.linear(x, self.project.type_as(x)).reshape(-1, 6)


class SideDecode(torch.autograd.Function):
    """side2box + Integral + the bbox_probs softmax in one kernel each way (HIP back end).
    reg (B, 6*bins+2, K) channel-major, agg (B, K, 3) -> probs (B, 6, bins, K, no gradient: every
    consumer detaches it), surface (B, K, 6), bbox (B, K, 7)."""

    @staticmethod
    def forward(ctx, reg, agg, scale, sign):
        from ..kernels import backend_for
        reg, agg = reg.contiguous(), agg.contiguous()
        b, cch, k = reg.shape
        bins = (cch - 2) // 6
        probs = reg.new_empty(b, 6, bins, k)
        surface, bbox = reg.new_empty(b, k, 6), reg.new_empty(b, k, 7)
        backend_for(reg).side_decode_forward(reg, agg, scale, sign, probs, surface, bbox)
        ctx.save_for_backward(reg, probs, scale, sign)
        ctx.mark_non_differentiable(probs)
        return probs, surface, bbox

    @staticmethod
    def backward(ctx, _d_probs, d_surface, d_bbox):
        from ..kernels import backend_for
        reg, probs, scale, sign = ctx.saved_tensors
        d_reg = torch.empty_like(reg)
        d_agg = reg.new_empty(reg.shape[0], reg.shape[2], 3)
        backend_for(reg).side_decode_backward(
            reg, probs, scale, sign, None if d_surface is None else d_surface.contiguous(),
            None if d_bbox is None else d_bbox.contiguous(), d_reg, d_agg)
        return d_reg, d_agg, None, None


class GTBatch:
    """Ground truth of a batch padded to T_max boxes, resident on the device.

    boxes (B,T,7) depth frame bottom-centre; labels (B,T) long; count (B,) = number of
    box columns the reference would see (>= 1: an empty scene gets one all-zero fake
    box, nesie_head.py:537-544); valid (B,T) = 1 for real boxes only.
    """

    def __init__(self, boxes, labels, count, valid):
        self.boxes, self.labels, self.count, self.valid = boxes, labels, count, valid

    @staticmethod
    def collate(gt_bboxes_3d, gt_labels_3d, device):
        B = len(gt_labels_3d)
        tensors = [b.tensor if hasattr(b, 'tensor') else torch.as_tensor(b) for b in gt_bboxes_3d]
        counts = [max(int(t.shape[0]), 1) for t in tensors]
        T = max(counts)
        boxes = torch.zeros(B, T, 7)
        labels = torch.zeros(B, T, dtype=torch.long)
        valid = torch.zeros(B, T)
        # padding columns: zero-size boxes far away, so no point is ever inside them
        boxes[:, :, :3] = 1e6
        for i, (t, l) in enumerate(zip(tensors, gt_labels_3d)):
            n = int(t.shape[0])
            if n == 0:
                boxes[i, 0] = 0.0  # the reference's fake box
            else:
                boxes[i, :n] = t[:, :7].float().cpu()
                labels[i, :n] = torch.as_tensor(l).long().cpu()
                valid[i, :n] = 1.0
        return GTBatch(boxes.to(device), labels.to(device),
                       torch.tensor(counts, device=device), valid.to(device))


class NesieHead(nn.Module):
    def __init__(self, num_classes, reg_max=16, reg_channels=128, train_cfg=None,
                 test_cfg=None, vote_module_cfg=None, vote_aggregation_cfg=None,
                 pred_layer_cfg=None, alpha=0.5, objectness_loss=None, center_loss=None,
                 semantic_loss=None, iou_loss=None, iou_pred_loss=None, surface_loss=None,
                 side_loss=None, grid_conv_cfg=None, sizes=(3.0, 3.0, 2.5)):
        super().__init__()
        self.num_classes = num_classes
        self.reg_max = reg_max
        self.reg_channels = reg_channels
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        self.gt_per_seed = vote_module_cfg['gt_per_seed']
        self.num_proposal = vote_aggregation_cfg['num_point']
        self.alpha = alpha
        self.sizes = list(sizes)
        self.objectness_loss = build_loss(objectness_loss)
        self.center_loss = build_loss(center_loss)
        self.iou_loss = build_loss(iou_loss)
        self.iou_pred_loss = build_loss(iou_pred_loss)
        self.surface_loss = build_loss(surface_loss)
        self.side_loss = build_loss(side_loss)
        if semantic_loss is not None:
            self.semantic_loss = build_loss(semantic_loss)
        self.vote_module = VoteModule(**vote_module_cfg)
        self.vote_aggregation = build_sa_module(vote_aggregation_cfg)
        self.fp16_enabled = False
        self.n_reg_outs = 6 * (self.reg_max + 1)
        self.conv_pred = ReliableConvBboxHead(
            **pred_layer_cfg, num_cls_out_channels=self.num_classes + 2,
            num_bbox_out_channels=self.n_reg_outs, num_heading_out_channels=2,
            reg_max=self.reg_max)
        self.integral = Integral(self.reg_max)
        self.grid_conv = SidePooling(**grid_conv_cfg)
        # independent loss terms on forked streams: measured on ROCm 7.2, a hipGraph captured
        # with forked branches replays the whole step 2x slower (43 vs 20 ms), so off
        self.parallel_loss_terms = False
        self.jitter_noise = None  # optional (noise_center, noise_size), each (B,K,3)
        self.register_buffer('_loss_ticket', head_loss.new_ticket(), persistent=False)
        # device-resident constants (no host->device copy inside the step: hipGraph-safe)
        self.register_buffer('_side_scale', torch.tensor(self.sizes + self.sizes), persistent=False)
        self.register_buffer('_side_sign', torch.tensor([-1., -1., -1., 1., 1., 1.]),
                             persistent=False)

    @staticmethod
    def _extract_input(feat_dict):
        return feat_dict['fp_xyz'][-1], feat_dict['fp_features'][-1], feat_dict['fp_indices'][-1]

    def _fused_decode(self, reg_predictions):
        """One-kernel decode when this class's own side2box is in effect (SAQEHead overrides
        it), on the HIP back end, fp32."""
        from ..kernels import backend_for
        return (type(self).side2box is NesieHead.side2box
                and backend_for(reg_predictions).name == 'hip'
                and reg_predictions.dtype == torch.float32 and self.reg_max + 1 <= 33)

    # ---- decode (:150-209) ------------------------------------------------------
    def side2box(self, aggregated_points, bbox_pred, results):
        B, proposal_num = bbox_pred.shape[:2]
        res = self.integral(bbox_pred[..., :self.n_reg_outs]).reshape(B, proposal_num, -1)
        scale = self._side_scale.to(bbox_pred.dtype)  # x y z x y z
        sign = self._side_sign.to(bbox_pred.dtype)
        results['surface_scale'] = scale.expand(B, proposal_num, 6)
        surface = aggregated_points.repeat(1, 1, 2) + sign * (res * scale)
        results['surface_pred'] = surface
        h0 = bbox_pred[..., self.n_reg_outs + 0]
        h1 = bbox_pred[..., self.n_reg_outs + 1]
        norm = torch.pow(torch.pow(h0, 2) + torch.pow(h1, 2), 0.5)
        sin, cos = h0 / norm, h1 / norm
        lo, hi = surface[..., :3], surface[..., 3:]
        results['bbox_preds'] = torch.cat(
            [(lo + hi) / 2.0, hi - lo, torch.atan2(sin, cos).unsqueeze(-1)], dim=-1)
        return results

    def jitter_bbox_preds(self, results, dataset_name):
        bp = results['bbox_preds']
        center, size, heading = bp[..., :3], bp[..., 3:6], bp[..., -1]
        if self.jitter_noise is not None:
            n_c, n_s = (t.to(size) for t in self.jitter_noise)
        else:
            n_c, n_s = torch.randn_like(size), torch.randn_like(size)
        backend = backend_for(bp)
        if backend.name == 'hip' and bp.dtype == torch.float32 and head_loss.ENABLED:
            # every consumer reads these detached (the quality head's grid, the IoU labels): one
            # launch, outside autograd
            with torch.no_grad():
                center_all, size_all, heading_all, jit = backend.proposal_jitter(
                    bp.detach().contiguous(), n_c.contiguous(), n_s.contiguous(), 0.3, 0.0,
                    dataset_name == 'ScanNet')
            results['jitter_bbox_preds'] = jit
            return center_all, size_all, heading_all, results
        center_jitter = center + size * n_c * 0.3
        size_jitter = torch.clamp(size + size * n_s * 0.3, min=1e-8)  # SAQEHead overrides
        heading_jitter = heading
        center_all = torch.cat([center, center_jitter], dim=1)
        size_all = torch.cat([size, size_jitter], dim=1)
        heading_all = torch.cat([heading, heading_jitter], dim=1)
        if dataset_name == 'ScanNet':
            heading_all = torch.zeros_like(heading_all)
        results['jitter_bbox_preds'] = torch.cat(
            [center_jitter, size_jitter, heading_jitter.unsqueeze(-1)], dim=-1)
        return center_all, size_all, heading_all, results

    # ---- forward (:211-275) -----------------------------------------------------
    def forward(self, feat_dict, sample_mod, dataset_name='ScanNet'):
        assert sample_mod in ['vote', 'seed', 'random', 'spec']
        seed_points, seed_features, seed_indices = self._extract_input(feat_dict)
        vote_points, vote_features, vote_offset = self.vote_module(seed_points, seed_features)
        results = dict(seed_points=seed_points, seed_features=seed_features,
                       seed_indices=seed_indices, vote_points=vote_points,
                       vote_features=vote_features, vote_offset=vote_offset)
        if sample_mod == 'vote':
            agg_in = dict(points_xyz=vote_points, features=vote_features)
        elif sample_mod == 'seed':
            sample_indices = furthest_point_sample(seed_points, self.num_proposal)
            agg_in = dict(points_xyz=vote_points, features=vote_features,
                          indices=sample_indices)
        elif sample_mod == 'random':
            batch_size, num_seed = seed_points.shape[:2]
            sample_indices = torch.randint(0, num_seed, (batch_size, self.num_proposal),
                                           dtype=torch.int32, device=seed_points.device)
            agg_in = dict(points_xyz=vote_points, features=vote_features,
                          indices=sample_indices)
        else:
            agg_in = dict(points_xyz=seed_points, features=seed_features,
                          target_xyz=vote_points)
        aggregated_points, features, aggregated_indices = self.vote_aggregation(**agg_in)
        results['aggregated_points'] = aggregated_points
        results['aggregated_features'] = features
        results['aggregated_indices'] = aggregated_indices

        cls_predictions, reg_predictions = self.conv_pred(features)
        origin_proposal_num = cls_predictions.shape[-1]
        cls_preds_trans = cls_predictions.transpose(2, 1)
        results['_cls_all'] = cls_predictions          # (B, 2 + C, K), for the fused loss kernel
        results['obj_scores'] = cls_preds_trans[..., :2]
        results['sem_scores'] = cls_preds_trans[..., 2:]
        B = reg_predictions.shape[0]
        if self._fused_decode(reg_predictions):
            probs, surface, boxes = SideDecode.apply(
                reg_predictions, aggregated_points, self._side_scale, self._side_sign)
            results['surface_scale'] = self._side_scale.expand(B, origin_proposal_num, 6)
            results['surface_pred'], results['bbox_preds'] = surface, boxes
            results['bbox_probs'] = probs
        else:
            results = self.side2box(aggregated_points, reg_predictions.transpose(2, 1), results)
            probs = reg_predictions[:, :self.n_reg_outs, :]
            results['bbox_probs'] = F.softmax(probs.reshape(B, 6, self.reg_max + 1, -1), dim=2)

        if not self.training and self._test_cfg('skip_jitter', False):
            # Opt-in, test time only: get_bboxes reads the ORIGINAL proposals' scores alone, and
            # with evaluation-mode norms every proposal is scored independently of the others, so
            # leaving the jittered copies out (half of the quality head's work) changes no
            # detection.  The reference always scores both halves (:240-262); the `*_jitter`
            # entries are then simply empty.
            bp = results['bbox_preds']
            center, size, heading = bp[..., :3], bp[..., 3:6], bp[..., -1]
            if dataset_name == 'ScanNet':
                heading = torch.zeros_like(heading)
        else:
            center, size, heading, results = self.jitter_bbox_preds(results, dataset_name)
        results = self.grid_conv(center.detach(), size.detach(), heading.detach(), results)

        iou = results['iou_scores'].sigmoid()
        results['iou_scores_jitter'] = iou[:, origin_proposal_num:]
        results['iou_scores'] = iou[:, :origin_proposal_num]
        side_all = results['side_scores'].sigmoid()                   # (6, B, C, 2K)
        results['_iou_all'], results['_side_all'] = iou, side_all     # for the fused loss kernel
        side = side_all.permute(1, 3, 0, 2)  # (B, 2K, 6, C)
        results['side_scores_jitter'] = side[:, origin_proposal_num:]
        results['side_scores'] = side[:, :origin_proposal_num]
        return results

    # ---- test path (:681-788) ---------------------------------------------------
    def _test_cfg(self, key, default=None):
        cfg = self.test_cfg
        if isinstance(cfg, dict):
            return cfg.get(key, default)
        return getattr(cfg, key, default)

    def _nms_selection(self, obj_scores, sem_scores, bbox3d, points_xyz):
        """The device half of multiclass_nms_single (:738-766) for a whole batch, without a
        host synchronisation: bottom-origin boxes (B,K,7), classes (B,K), selected (B,K) bool.
        Three launches stand in for the reference's per-scene (M, K) membership table, its
        python NMS loop and the scatter: a per-box point count, one NMS workgroup per scene,
        one scatter."""
        B, K = obj_scores.shape
        boxes = bbox3d.clone()                                 # origin (0.5,0.5,0.5) -> bottom
        boxes[..., 2] += boxes[..., 5] * -0.5                  # (x, y get + 0.0 * size)
        counts = points_in_boxes_count(depth_to_lidar_points(points_xyz).contiguous(),
                                       depth_to_lidar_boxes(boxes).contiguous())
        nonempty = counts > 5
        corners = DepthInstance3DBoxes.__new__(DepthInstance3DBoxes)
        corners.tensor = boxes.reshape(B * K, 7)
        corner3d = corners.corners.view(B, K, 8, 3)
        minmax = torch.cat([corner3d.min(dim=2)[0], corner3d.max(dim=2)[0]], dim=-1)
        classes = torch.argmax(sem_scores, -1)
        picks, _ = batched_aligned_3d_nms(minmax, obj_scores, classes,
                                          self._test_cfg('nms_thr'), valid=nonempty)
        kept = torch.zeros(B, K + 1, dtype=torch.bool, device=boxes.device)
        kept.scatter_(1, torch.where(picks < 0, picks.new_full((), K), picks).long(),
                      torch.ones_like(picks, dtype=torch.bool))
        selected = kept[:, :K] & (obj_scores > self._test_cfg('score_thr'))
        return boxes, classes, selected

    def _gather_selected(self, boxes, obj_scores, sem_scores, classes, selected):
        """One scene: (K,7), (K), (K,C), (K), (K) bool -> the reference's return triple."""
        box_sel, obj_sel = boxes[selected], obj_scores[selected]
        if self._test_cfg('per_class_proposal'):
            sem_sel = sem_scores[selected]                       # (n, C)
            C = sem_sel.shape[-1]
            labels = torch.arange(C, device=boxes.device, dtype=classes.dtype) \
                .repeat_interleave(box_sel.shape[0])
            return (box_sel.repeat(C, 1), (obj_sel.unsqueeze(1) * sem_sel).t().reshape(-1),
                    labels)
        return box_sel, obj_sel, classes[selected]

    def detect_tensors(self, points, bbox_preds, use_iou_for_nms=True):
        """The device half of get_bboxes (:696-718): scores, bottom-origin boxes, classes and
        the per-proposal `selected` mask for the whole batch -- fixed shapes, no host
        synchronisation (capturable in a hipGraph)."""
        obj_scores = F.softmax(bbox_preds['obj_scores'], dim=-1)[..., -1]
        sem_scores = F.softmax(bbox_preds['sem_scores'], dim=-1)
        if use_iou_for_nms:
            indx = bbox_preds['sem_scores'].max(dim=-1)[1]
            obj_scores = obj_scores * bbox_preds['iou_scores'].gather(2, indx.unsqueeze(-1)).squeeze(-1)
        boxes, classes, selected = self._nms_selection(obj_scores, sem_scores,
                                                       bbox_preds['bbox_preds'], points[..., :3])
        return boxes, obj_scores, sem_scores, classes, selected

    def boxes_from_tensors(self, tensors, input_metas=None):
        """The host-shaped half: per scene, the boolean selection and the reference's
        (boxes, scores, labels) triple (:719-729, 768-788)."""
        boxes, obj_scores, sem_scores, classes, selected = tensors
        results = []
        for b in range(boxes.shape[0]):
            box_sel, score_sel, labels = self._gather_selected(
                boxes[b], obj_scores[b], sem_scores[b], classes[b], selected[b])
            box_type = (input_metas[b] or {}).get('box_type_3d', DepthInstance3DBoxes) \
                if input_metas is not None else DepthInstance3DBoxes
            results.append((box_type(box_sel, box_dim=box_sel.shape[-1], with_yaw=True),
                            score_sel, labels))
        return results

    def get_bboxes(self, points, bbox_preds, input_metas, rescale=False, use_nms=True,
                   use_iou_for_nms=True):
        """Boxes, scores and labels per scene from the head's predictions (:681-729)."""
        if not use_nms:
            return bbox_preds['bbox_preds']
        return self.boxes_from_tensors(self.detect_tensors(points, bbox_preds, use_iou_for_nms),
                                       input_metas)

    def multiclass_nms_single(self, obj_scores, sem_scores, bbox, points, input_meta):
        """Single-scene form with the reference's signature (:731-788)."""
        boxes, classes, selected = self._nms_selection(
            obj_scores.unsqueeze(0), sem_scores.unsqueeze(0), bbox.unsqueeze(0),
            points[..., :3].unsqueeze(0))
        return self._gather_selected(boxes[0], obj_scores, sem_scores, classes[0], selected[0])

    # ---- loss (:278-412) --------------------------------------------------------
    @staticmethod
    def _pick_class(per_class, cls_index):
        """per_class (P,6,C), cls_index (P,) -> (P,6): one class column per proposal."""
        ix = cls_index.view(-1, 1, 1).expand(-1, per_class.shape[1], 1)
        return per_class.gather(2, ix).squeeze(-1)

    def _sigma(self, bbox_preds):
        indx = bbox_preds['sem_scores'].max(dim=-1)[1].reshape(-1)
        n_class = bbox_preds['side_scores'].shape[-1]
        side_pred = bbox_preds['side_scores'].reshape(-1, 6, n_class)
        s = self._pick_class(side_pred, indx).reshape(-1, 6)
        return 0.8 * s * s - 1.8 * s + torch.ones_like(s)

    def loss(self, bbox_preds, points, gt_bboxes_3d, gt_labels_3d, pts_semantic_mask=None,
             pts_instance_mask=None, img_metas=None, gt_bboxes_ignore=None,
             ret_target=False, vote_targets=None):
        """``vote_targets``: optional result of ``vote_targets_of`` for this batch (the
        per-point half of the targets depends on the inputs only, so a training loop may
        compute it ahead of the step, like the backbone's index chain)."""
        targets = self.get_targets(points, gt_bboxes_3d, gt_labels_3d, pts_semantic_mask,
                                   pts_instance_mask, bbox_preds, vote_targets=vote_targets)
        (vote_targets, vote_target_masks, center_targets, bbox_targets, mask_targets,
         valid_gt_masks, objectness_targets, objectness_weights, box_loss_weights,
         valid_gt_weights, assignment) = targets
        bbox_targets_cat = bbox_targets.reshape(-1, 7)
        if head_loss.usable(self, bbox_preds):
            return self._fused_loss(bbox_preds, targets, ret_target)

        # shared by several terms
        surface_weight = box_loss_weights.reshape(-1).unsqueeze(-1).repeat(1, 6)
        probs = bbox_preds['bbox_probs'].permute(0, 3, 1, 2).reshape(-1, 6, self.reg_max + 1)
        sigma = self._sigma(bbox_preds)
        iou_weight = box_loss_weights.reshape(-1)
        label_cls = mask_targets.reshape(-1)
        targets_b = bbox_targets_cat.view_as(bbox_preds['bbox_preds'])

        def branch_votes():
            vote_loss = self.vote_module.get_loss(
                bbox_preds['seed_points'], bbox_preds['vote_points'], bbox_preds['seed_indices'],
                vote_target_masks, vote_targets)
            objectness_loss = self.objectness_loss(
                bbox_preds['obj_scores'].transpose(2, 1), objectness_targets,
                weight=objectness_weights)
            semantic_loss = self.semantic_loss(
                bbox_preds['sem_scores'].transpose(2, 1), mask_targets, weight=box_loss_weights)
            return vote_loss, objectness_loss, semantic_loss

        def branch_surface():
            source2target_loss, target2source_loss = self.center_loss(
                bbox_preds['bbox_preds'][..., :3], center_targets, src_weight=box_loss_weights,
                dst_weight=valid_gt_weights)
            surface_loss = self.surface_loss(
                bbox_preds['surface_pred'].reshape(-1, 6), bbox_targets_cat,
                bbox_preds['surface_scale'].reshape(-1, 6),
                bbox_preds['aggregated_points'].reshape(-1, 3), probs, weight=surface_weight,
                reduction_override='none')
            surface_loss = torch.exp(-sigma) * surface_loss + self.alpha * sigma * surface_weight
            return source2target_loss + target2source_loss, surface_loss.sum()

        def branch_iou():
            iou_loss = self.iou_loss(bbox_preds['bbox_preds'].reshape(-1, 7), bbox_targets_cat,
                                     weight=iou_weight, reduction_override='none').reshape(-1)
            sigma_mean = sigma.mean(dim=-1)
            iou_loss = torch.exp(-sigma_mean) * iou_loss + self.alpha * sigma_mean * iou_weight
            return (iou_loss.sum(),)

        def branch_iou_pred():
            label_iou = cal_iou_3d(bbox_preds['bbox_preds'], targets_b).detach().view(-1)
            label_iou_jitter = cal_iou_3d(bbox_preds['jitter_bbox_preds'],
                                          targets_b).detach().view(-1)
            w = iou_weight
            if getattr(self.iou_pred_loss, 'reduction', None) == 'sum':
                # the plain and the jittered half in one pass (a sum-reduced loss is additive)
                return (self.iou_pred_loss(
                    torch.cat([bbox_preds['iou_scores'].reshape(-1, self.num_classes),
                               bbox_preds['iou_scores_jitter'].reshape(-1, self.num_classes)]),
                    (torch.cat([label_cls, label_cls]),
                     torch.cat([label_iou, label_iou_jitter])), weight=torch.cat([w, w])),)
            loss_iou = self.iou_pred_loss(
                bbox_preds['iou_scores'].reshape(-1, self.num_classes), (label_cls, label_iou),
                weight=w)
            loss_iou_jitter = self.iou_pred_loss(
                bbox_preds['iou_scores_jitter'].reshape(-1, self.num_classes),
                (label_cls, label_iou_jitter), weight=w)
            return (loss_iou + loss_iou_jitter,)

        def branch_side():
            side_pred = bbox_preds['side_scores'].reshape(-1, 6, self.num_classes)
            side_pred = self._pick_class(side_pred, label_cls)
            return (self.side_loss(
                side_pred, bbox_preds['surface_pred'].reshape(-1, 6), bbox_targets_cat,
                bbox_preds['surface_scale'].reshape(-1, 6),
                bbox_preds['aggregated_points'].reshape(-1, 3), probs, weight=surface_weight),)

        # the terms are independent chains of small launches (optionally parallel branches)
        ((vote_loss, objectness_loss, semantic_loss), (center_loss, surface_loss), (iou_loss,),
         (iou_pred_loss,), (side_loss,)) = fork_join(
            [branch_votes, branch_surface, branch_iou, branch_iou_pred, branch_side],
            like=bbox_targets_cat, enabled=self.parallel_loss_terms)

        losses = dict(vote_loss=vote_loss, objectness_loss=objectness_loss,
                      semantic_loss=semantic_loss, center_loss=center_loss,
                      surface_loss=surface_loss, iou_loss=iou_loss,
                      iou_pred_loss=iou_pred_loss, side_loss=side_loss)
        if ret_target:
            losses['targets'] = targets_b
        return losses

    def _fused_loss(self, bbox_preds, targets, ret_target):
        """The same eight terms with the seven per-proposal ones in one launch
        (``head_loss.HeadLossFn``); the vote term and the two rotated-IoU evaluations stay."""
        (vote_targets, vote_target_masks, center_targets, bbox_targets, mask_targets,
         valid_gt_masks, objectness_targets, objectness_weights, box_loss_weights,
         valid_gt_weights, assignment) = targets
        vote_loss = self.vote_module.get_loss(
            bbox_preds['seed_points'], bbox_preds['vote_points'], bbox_preds['seed_indices'],
            vote_target_masks, vote_targets)
        boxes = bbox_preds['bbox_preds']
        iou = cal_iou_3d(boxes, bbox_targets)                              # (B, K), with gradient
        iou_jitter = cal_iou_3d(bbox_preds['jitter_bbox_preds'], bbox_targets).detach()
        tg = dict(obj_targets=objectness_targets, mask_targets=mask_targets,
                  obj_weights=objectness_weights, box_weights=box_loss_weights,
                  bbox_targets=bbox_targets, center_targets=center_targets,
                  valid_weights=valid_gt_weights)
        terms = head_loss.HeadLossFn.apply(
            bbox_preds['_cls_all'].contiguous(), boxes.contiguous(),
            bbox_preds['surface_pred'].contiguous(), bbox_preds['_side_all'].contiguous(),
            bbox_preds['_iou_all'].contiguous(), iou, iou_jitter, tg, head_loss.config_of(self),
            self._loss_ticket)
        losses = dict(vote_loss=vote_loss, **dict(zip(head_loss.TERMS, terms)))
        if ret_target:
            losses['targets'] = bbox_targets.view_as(boxes)
        return losses

    # ---- unsupervised loss (:415-509) -------------------------------------------
    def unsup_loss(self, bbox_preds, points, pseudo_boxes, pseudo_label, img_metas=None,
                   pseudo_quality_score=None):
        targets = self.get_targets(points, pseudo_boxes, pseudo_label, bbox_preds=bbox_preds)
        (_, _, center_targets, bbox_targets, mask_targets, _, _, _, box_loss_weights,
         valid_gt_weights, assignment) = targets
        B, num_proposal = assignment.shape
        bbox_targets_cat = bbox_targets.reshape(-1, 7)
        if torch.is_tensor(pseudo_quality_score):
            # padded (B, T, 6) form (zeros on padding / on the fake box of an empty scene)
            pseudo_quality_side = torch.gather(
                pseudo_quality_score, 1, assignment.unsqueeze(-1).expand(-1, -1, 6))
        else:
            quality = []
            for i in range(B):
                q = pseudo_quality_score[i]
                if q.shape[0] != 0:
                    quality.append(q.to(assignment.device)[assignment[i]])
                else:
                    quality.append(torch.zeros(num_proposal, 6, device=assignment.device))
            pseudo_quality_side = torch.stack(quality)
        if head_loss.usable(self, bbox_preds, unsup=True):
            return self._fused_unsup_loss(bbox_preds, targets, pseudo_quality_side)
        pseudo_quality_mean = pseudo_quality_side.mean(dim=-1)

        s2t, t2s = self.center_loss(bbox_preds['bbox_preds'][..., :3], center_targets,
                                    src_weight=box_loss_weights, dst_weight=valid_gt_weights)
        unsup_center_loss = s2t + t2s
        unsup_semantic_loss = self.semantic_loss(
            bbox_preds['sem_scores'].transpose(2, 1), mask_targets, weight=box_loss_weights)

        iou_weight = (box_loss_weights * pseudo_quality_mean).reshape(-1)
        unsup_iou_loss = self.iou_loss(bbox_preds['bbox_preds'].reshape(-1, 7),
                                       bbox_targets_cat, weight=iou_weight,
                                       reduction_override='none').reshape(-1)
        sigma = self._sigma(bbox_preds)
        sigma_mean = sigma.mean(dim=-1)
        unsup_iou_loss = (torch.exp(-sigma_mean) * unsup_iou_loss
                          + self.alpha * sigma_mean * iou_weight).sum()

        surface_weight = box_loss_weights.reshape(-1).unsqueeze(-1).repeat(1, 6) \
            * pseudo_quality_side.reshape(-1, 6)
        probs = bbox_preds['bbox_probs'].permute(0, 3, 1, 2).reshape(-1, 6, self.reg_max + 1)
        unsup_surface_loss = self.surface_loss(
            bbox_preds['surface_pred'].reshape(-1, 6), bbox_targets_cat,
            bbox_preds['surface_scale'].reshape(-1, 6),
            bbox_preds['aggregated_points'].reshape(-1, 3), probs, weight=surface_weight,
            reduction_override='none')
        unsup_surface_loss = (torch.exp(-sigma) * unsup_surface_loss
                              + self.alpha * sigma * surface_weight).sum()
        un_label_weight = 2.0
        return dict(unsup_semantic_loss=un_label_weight * unsup_semantic_loss,
                    unsup_center_loss=un_label_weight * unsup_center_loss,
                    unsup_iou_loss=un_label_weight * unsup_iou_loss,
                    unsup_surface_loss=un_label_weight * unsup_surface_loss)

    def _fused_unsup_loss(self, bbox_preds, targets, pseudo_quality_side):
        """The four terms in one launch (``head_loss.HeadLossFn`` with the side qualities); the
        rotated-IoU evaluation stays."""
        (_, _, center_targets, bbox_targets, mask_targets, _, objectness_targets,
         objectness_weights, box_loss_weights, valid_gt_weights, _) = targets
        boxes = bbox_preds['bbox_preds']
        iou = cal_iou_3d(boxes, bbox_targets)                              # (B, K), with gradient
        tg = dict(obj_targets=objectness_targets, mask_targets=mask_targets,
                  obj_weights=objectness_weights, box_weights=box_loss_weights,
                  bbox_targets=bbox_targets, center_targets=center_targets,
                  valid_weights=valid_gt_weights)
        terms = dict(zip(head_loss.TERMS, head_loss.HeadLossFn.apply(
            bbox_preds['_cls_all'].contiguous(), boxes.contiguous(),
            bbox_preds['surface_pred'].contiguous(), bbox_preds['_side_all'].contiguous(),
            bbox_preds['_iou_all'].contiguous(), iou, iou.detach(), tg,
            head_loss.unsup_config_of(self), self._loss_ticket,
            pseudo_quality_side.to(boxes.dtype).contiguous(),
            bool(getattr(self, '_sigma_is_constant', False)))))
        un_label_weight = 2.0
        return dict(unsup_semantic_loss=un_label_weight * terms['semantic_loss'],
                    unsup_center_loss=un_label_weight * terms['center_loss'],
                    unsup_iou_loss=un_label_weight * terms['iou_loss'],
                    unsup_surface_loss=un_label_weight * terms['surface_loss'])

    # ---- targets (:511-679), batched on the device ------------------------------
    @staticmethod
    def vote_targets_of(points, gt_bboxes_3d, gt_labels_3d=None):
        """Per-point vote targets (B,N,9) and masks (B,N) of ``get_targets_single``
        (nesie_head.py:593-654), batched on the device.  Depends on the points and the GT
        boxes only -- not on any network output."""
        pts = torch.stack(points) if isinstance(points, (list, tuple)) else points
        device = pts.device
        gt = gt_bboxes_3d if isinstance(gt_bboxes_3d, GTBatch) else \
            GTBatch.collate(gt_bboxes_3d, gt_labels_3d, device)
        B, N = pts.shape[:2]
        T = gt.boxes.shape[1]
        backend = backend_for(pts)
        if (head_loss.ENABLED and backend.name == 'hip' and pts.dtype == torch.float32
                and gt.boxes.dtype == torch.float32 and N > 0):
            # in-box test, slot bookkeeping and the nine offsets in one launch
            return backend.vote_targets(pts.contiguous(), gt.boxes.contiguous(),
                                        gt.count.contiguous())
        xyz = pts[..., :3]
        col = torch.arange(T, device=device).unsqueeze(0)
        is_col = col < gt.count.unsqueeze(1)  # (B,T) columns the reference iterates over

        # gravity centres (depth_box3d.py:42-48)
        centres = torch.cat([gt.boxes[..., :2],
                             (gt.boxes[..., 2] + gt.boxes[..., 5] * 0.5).unsqueeze(-1)], -1)

        # --- vote targets: which boxes hold each point (one launch for the batch) ---
        inbox = points_in_boxes_batch(depth_to_lidar_points(xyz).contiguous(),
                                      depth_to_lidar_boxes(gt.boxes).contiguous())
        inbox = (inbox > 0) & is_col.unsqueeze(1)  # (B,N,T)
        cnt = inbox.sum(-1)
        ib = inbox.to(torch.uint8)
        first = torch.argmax(ib, dim=-1)
        # second = first column after `first` that holds the point (0 if none; unused then)
        second = torch.argmax(ib * (col.unsqueeze(1) > first.unsqueeze(-1)).to(torch.uint8),
                              dim=-1)
        last = (T - 1) - torch.argmax(ib.flip(-1), dim=-1)

        def vote_of(box_idx):
            c = torch.gather(centres, 1, box_idx.unsqueeze(-1).expand(-1, -1, 3))
            return c - xyz
        v1 = vote_of(first)
        # slot 0 = first box; slot 1 = second box if any; slot 2 = last box when the
        # point sits in >= 3 boxes (the reference's counter clamps at 2, :629-652)
        v2 = torch.where((cnt >= 2).unsqueeze(-1), vote_of(second), v1)
        v3 = torch.where((cnt >= 3).unsqueeze(-1), vote_of(last), v1)
        has = (cnt > 0).unsqueeze(-1)
        vote_targets = torch.where(has, torch.cat([v1, v2, v3], dim=-1),
                                   xyz.new_zeros(B, N, 9))
        vote_target_masks = (cnt > 0).long()
        return vote_targets, vote_target_masks

    def get_targets(self, points, gt_bboxes_3d, gt_labels_3d, pts_semantic_mask=None,
                    pts_instance_mask=None, bbox_preds=None, vote_targets=None):
        pts = torch.stack(points) if isinstance(points, (list, tuple)) else points
        device = pts.device
        gt = gt_bboxes_3d if isinstance(gt_bboxes_3d, GTBatch) else \
            GTBatch.collate(gt_bboxes_3d, gt_labels_3d, device)
        T = gt.boxes.shape[1]
        vote_targets, vote_target_masks = vote_targets if vote_targets is not None \
            else self.vote_targets_of(pts, gt)

        # --- proposal <-> GT assignment (:656-676) ---
        aggregated_points = bbox_preds['aggregated_points']
        backend = backend_for(aggregated_points)
        if (head_loss.ENABLED and backend.name == 'hip' and aggregated_points.dtype == torch.float32
                and gt.boxes.dtype == torch.float32):
            # assignment, labels, box targets and the three batch-level weights in one launch
            tg = backend.head_targets(aggregated_points.contiguous(), gt.boxes.contiguous(),
                                      gt.labels.contiguous(), gt.count.contiguous(),
                                      gt.valid.float().contiguous(),
                                      self.train_cfg['pos_distance_thr'],
                                      self.train_cfg['neg_distance_thr'])
            return (vote_targets, vote_target_masks, tg['center_targets'], tg['bbox_targets'],
                    tg['mask_targets'], gt.valid, tg['obj_targets'], tg['obj_weights'],
                    tg['box_weights'], tg['valid_weights'], tg['assignment'])
        col = torch.arange(T, device=device).unsqueeze(0)
        is_col = col < gt.count.unsqueeze(1)
        centres = torch.cat([gt.boxes[..., :2],
                             (gt.boxes[..., 2] + gt.boxes[..., 5] * 0.5).unsqueeze(-1)], -1)
        d = aggregated_points.unsqueeze(2) - centres.unsqueeze(1)
        d = (d * d).sum(-1)  # (B,K,T) squared L2
        d = torch.where(is_col.unsqueeze(1), d, torch.full_like(d, float('inf')))
        distance1, assignment = torch.min(d, dim=2)
        euclidean_distance1 = torch.sqrt(distance1 + 1e-6)
        pos = euclidean_distance1 < self.train_cfg['pos_distance_thr']
        neg = euclidean_distance1 > self.train_cfg['neg_distance_thr']
        objectness_targets = pos.long()
        objectness_masks = (pos | neg).float()
        mask_targets = torch.gather(gt.labels, 1, assignment).long()
        gi = assignment.unsqueeze(-1)
        bbox_targets = torch.cat([torch.gather(centres, 1, gi.expand(-1, -1, 3)),
                                  torch.gather(gt.boxes[..., 3:], 1, gi.expand(-1, -1, 4))], -1)

        # --- batch-level padding / normalisation (:566-588) ---
        center_targets = torch.where(is_col.unsqueeze(-1), centres, torch.zeros_like(centres))
        valid_gt_masks = gt.valid
        objectness_weights = objectness_masks / (torch.sum(objectness_masks) + 1e-6)
        box_loss_weights = objectness_targets.float() / (
            torch.sum(objectness_targets).float() + 1e-6)
        valid_gt_weights = valid_gt_masks.float() / (torch.sum(valid_gt_masks.float()) + 1e-6)
        return (vote_targets, vote_target_masks, center_targets, bbox_targets, mask_targets,
                valid_gt_masks, objectness_targets, objectness_weights, box_loss_weights,
                valid_gt_weights, assignment)
