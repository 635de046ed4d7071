"""Losses of the Nesie head.

Restates ``mmdet3d/models/losses/{chamfer_distance,surface_loss,side_pred_loss,
iou3d_loss,gfocal_loss}.py`` and the mmdet 2.19 loss plumbing they call
(``weighted_loss`` / ``weight_reduce_loss`` / MSE / L1 / SmoothL1 / CrossEntropy;
source not in the reference tree, semantics from SURVEY.md appendix C).
"""
import torch
import torch.nn.functional as F
from torch import nn

from ..mmdet3d_ops.rotated_iou import cal_iou_3d


# ---- mmdet loss plumbing (appendix C) ------------------------------------------
def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        if reduction == 'mean':
            return loss.mean()
        if reduction == 'sum':
            return loss.sum()
        return loss
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction == 'none':
        return loss
    raise ValueError('avg_factor can not be used with reduction="sum"')


class _ElementwiseLoss(nn.Module):
    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction = reduction
        self.loss_weight = loss_weight

    def elementwise(self, pred, target):
        raise NotImplementedError

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        loss = self.elementwise(pred, target)
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction, avg_factor)


class MSELoss(_ElementwiseLoss):
    def elementwise(self, pred, target):
        return F.mse_loss(pred, target, reduction='none')


class L1Loss(_ElementwiseLoss):
    def elementwise(self, pred, target):
        return torch.abs(pred - target)


class SmoothL1Loss(_ElementwiseLoss):
    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__(reduction, loss_weight)
        self.beta = beta

    def elementwise(self, pred, target):
        diff = torch.abs(pred - target)
        return torch.where(diff < self.beta, 0.5 * diff * diff / self.beta,
                           diff - 0.5 * self.beta)


class CrossEntropyLoss(nn.Module):
    """Softmax CE with optional class weights; per-sample weight then reduce."""

    def __init__(self, class_weight=None, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.class_weight = class_weight
        self.reduction = reduction
        self.loss_weight = loss_weight
        self.register_buffer('_cw', None if class_weight is None else
                             torch.tensor(class_weight, dtype=torch.float32), persistent=False)

    def forward(self, cls_score, label, weight=None, avg_factor=None,
                reduction_override=None):
        reduction = reduction_override if reduction_override else self.reduction
        cw = self._cw.to(cls_score.dtype) if self._cw is not None else None
        loss = F.cross_entropy(cls_score, label, weight=cw, reduction='none')
        if weight is not None:
            weight = weight.float()
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction, avg_factor)


# ---- chamfer (behaviour of chamfer_distance.py:8-146) ----------------------------
def _huber_unit(d):
    a = d.abs()
    return torch.where(a < 1.0, 0.5 * a * a, a - 0.5)


_COORD_PENALTY = {'l1': torch.abs, 'l2': torch.square, 'smooth_l1': _huber_unit}
_FOLD = {'none': lambda t: t, 'sum': torch.sum, 'mean': torch.mean}


def chamfer_distance(src, dst, src_weight=1.0, dst_weight=1.0, criterion_mode='l2',
                     reduction='mean'):
    """src (B, N, C), dst (B, M, C).  The cost of a pair is the per-coordinate penalty summed
    over C; every source point is charged its cheapest destination and vice versa.
    -> (source term, destination term, cheapest dst per src (B, N), cheapest src per dst (B, M))."""
    if criterion_mode not in _COORD_PENALTY or reduction not in _FOLD:
        raise NotImplementedError((criterion_mode, reduction))
    pair = _COORD_PENALTY[criterion_mode](src[:, :, None] - dst[:, None]).sum(-1)   # (B, N, M)
    to_dst, to_src = pair.min(dim=2), pair.min(dim=1)
    fold = _FOLD[reduction]
    return (fold(to_dst.values * src_weight), fold(to_src.values * dst_weight),
            to_dst.indices, to_src.indices)


class ChamferDistance(nn.Module):
    """Configured Chamfer term: the two directions scaled by ``loss_src_weight`` /
    ``loss_dst_weight`` (the vote loss uses mode 'l1', reduction 'none', dst weight 10)."""

    def __init__(self, mode='l2', reduction='mean', loss_src_weight=1.0, loss_dst_weight=1.0):
        super().__init__()
        if mode not in _COORD_PENALTY or reduction not in _FOLD:
            raise AssertionError((mode, reduction))
        self.mode, self.reduction = mode, reduction
        self.loss_src_weight, self.loss_dst_weight = loss_src_weight, loss_dst_weight

    def forward(self, source, target, src_weight=1.0, dst_weight=1.0,
                reduction_override=None, return_indices=False, **kwargs):
        if reduction_override is not None and reduction_override not in _FOLD:
            raise AssertionError(reduction_override)
        src_term, dst_term, src_pick, dst_pick = chamfer_distance(
            source, target, src_weight, dst_weight, self.mode, reduction_override or self.reduction)
        out = (src_term * self.loss_src_weight, dst_term * self.loss_dst_weight)
        return out + (src_pick, dst_pick) if return_indices else out


# ---- side surfaces (surface_loss.py:90-100) -------------------------------------
def Bbox2Surface(bbox2):
    """(…,>=6) centre+size -> (…,6) planes (x-,y-,z-,x+,y+,z+)."""
    center = bbox2[..., :3]
    half = 0.5 * bbox2[..., 3:6]
    return torch.cat([center - half, center + half], dim=-1)


class SurfaceLoss(nn.Module):
    """Per-side regression loss against the GT box planes (surface_loss.py:10-88).
    Only the MSE / SmoothL1 forms are used by the shipped configs."""

    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0, func_type='MSELoss'):
        super().__init__()
        self.func_type = func_type
        if func_type == 'MSELoss':
            self.loss_func = MSELoss(reduction, loss_weight)
        elif func_type == 'SmoothL1Loss':
            self.loss_func = SmoothL1Loss(beta, reduction, loss_weight)
        else:
            raise NotImplementedError(f'SurfaceLoss func_type {func_type} is not used by '
                                      'any shipped Nesie/SAQE config')

    def forward(self, pred, target, scale, center, prob, weight=None, avg_factor=None,
                reduction_override=None, **kwargs):
        target = Bbox2Surface(target)
        return self.loss_func(pred, target, weight, avg_factor, reduction_override)


class SidePredLoss(nn.Module):
    """Side-quality loss: label = min(1, 4*|pred side - GT side|) (an L1 with weight 4,
    clamped), loss = MSE(side score, label)  (side_pred_loss.py:10-83)."""

    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0,
                 label_func_type='SmoothL1Loss', loss_func_type='MSELoss'):
        super().__init__()
        self.label_func_type = label_func_type
        if label_func_type == 'MSELoss':
            self.label_func = MSELoss(reduction, 4.0)
        elif label_func_type == 'SmoothL1Loss':
            self.label_func = L1Loss(reduction, 4.0)  # sic: the reference builds an L1
        else:
            raise NotImplementedError
        if loss_func_type == 'MSELoss':
            self.loss_func = MSELoss(reduction, loss_weight)
        elif loss_func_type == 'SmoothL1Loss':
            self.loss_func = SmoothL1Loss(beta, reduction, loss_weight)
        else:
            raise NotImplementedError

    def forward(self, pred_side, pred, target, scale, center, prob, weight=None,
                avg_factor=None, reduction_override=None, **kwargs):
        target = Bbox2Surface(target)
        label_side = self.label_func(pred, target, None, None, 'none').detach()
        label_side = torch.where(label_side > 1, torch.ones_like(label_side), label_side)
        return self.loss_func(pred_side, label_side, weight, avg_factor, reduction_override)


# ---- IoU3D (iou3d_loss.py:12-14, 38-76) -----------------------------------------
class IoU3DLoss(nn.Module):
    def __init__(self, with_yaw=True, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert with_yaw, 'axis-aligned IoU loss is not used by the shipped configs'
        self.reduction = reduction
        self.loss_weight = loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None,
                **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if weight is not None and weight.dim() > 1:
            weight = weight.mean(-1)
        loss = 1 - cal_iou_3d(pred[None, ...], target[None, ...])  # (1, N)
        if weight is not None:
            # reference :53-54 returns pred.sum() * weight.sum() (== 0) when no weight is
            # positive; masking the unweighted entries gives the same zeros without the
            # host sync of `if not torch.any(weight > 0)`.
            loss = torch.where(weight > 0, loss, torch.zeros_like(loss))
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction, avg_factor)


# ---- quality focal loss (gfocal_loss.py:9-51, 80-139) ---------------------------
def quality_focal_loss(pred, target, beta=2.0, use_sigmoid=True):
    label, score = target
    if use_sigmoid:
        func = F.binary_cross_entropy_with_logits
        pred_sigmoid = pred.sigmoid()
    else:
        func = F.binary_cross_entropy
        pred_sigmoid = pred
    zerolabel = pred.new_zeros(pred.shape)
    loss = func(pred, zerolabel, reduction='none') * pred_sigmoid.pow(beta)
    bg_class_ind = pred.size(1)
    # positives (0 <= label < C) are supervised by the IoU score at their class column;
    # written as a dense one-hot select instead of the reference's nonzero()/index_put
    pos = ((label >= 0) & (label < bg_class_ind))
    onehot = F.one_hot(label.clamp(0, bg_class_ind - 1).long(), bg_class_ind).bool() \
        & pos.unsqueeze(1)
    score_b = score.unsqueeze(1).expand_as(pred)
    pos_loss = func(pred, score_b, reduction='none') * (score_b - pred_sigmoid).abs().pow(beta)
    loss = torch.where(onehot, pos_loss, loss)
    return loss.sum(dim=1, keepdim=False)


class GeneralQualityFocalLoss(nn.Module):
    def __init__(self, use_sigmoid=True, beta=2.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.use_sigmoid = use_sigmoid
        self.beta = beta
        self.reduction = reduction
        self.loss_weight = loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        loss = quality_focal_loss(pred, target, beta=self.beta, use_sigmoid=self.use_sigmoid)
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction, avg_factor)


_LOSSES = dict(ChamferDistance=ChamferDistance, CrossEntropyLoss=CrossEntropyLoss,
               IoU3DLoss=IoU3DLoss, GeneralQualityFocalLoss=GeneralQualityFocalLoss,
               SurfaceLoss=SurfaceLoss, SidePredLoss=SidePredLoss, MSELoss=MSELoss,
               L1Loss=L1Loss, SmoothL1Loss=SmoothL1Loss)


def build_loss(cfg):
    cfg = dict(cfg)
    return _LOSSES[cfg.pop('type')](**cfg)
