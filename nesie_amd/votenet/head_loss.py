"""The seven per-proposal loss terms of ``NesieHead.loss`` as one forward and one backward launch.

``nesie_head.py:279-413`` of the reference evaluates objectness / semantic cross entropy, the
centre Chamfer term, the uncertainty-weighted surface and IoU regressions, the IoU-quality focal
loss (plain + jittered proposals) and the side-quality loss through ~150 small tensor ops, and
autograd replays as many for the gradients.  All of them are closed-form sums over the (at most a
few thousand) proposals of the batch: ``csrc/head_loss.hip`` evaluates the sums and their analytic
gradients in one workgroup (``nesie_head_loss_forward``), and the backward only scales the saved
gradients by the incoming ones and lays them out for the producers (``nesie_head_loss_backward``).
The module-by-module evaluation in ``nesie_head.py`` stays the definition (CPU checker, other
configurations); ``tests/test_head_loss_gpu.py`` holds the two against each other.
"""
import torch
from torch.autograd import Function

from ..kernels import backend_for

ENABLED = True   # tests flip this to get the module-by-module evaluation on the device


def new_ticket():
    """The 'last workgroup reduces' counter of the loss kernels: a zeroed int32 the kernels reset
    themselves.  Kept as a NON-persistent module buffer so that it is allocated with the model
    (never inside a hipGraph capture) and follows ``.to(device)``."""
    return torch.zeros(1, dtype=torch.int32)

TERMS = ('objectness_loss', 'semantic_loss', 'center_loss', 'surface_loss', 'iou_loss',
         'iou_pred_loss', 'side_loss')


class HeadLossFn(Function):
    """(cls, bbox, surface, side_all, iou_all, iou, iou_jitter) -> the seven terms in ``TERMS``
    order, as seven 0-dim views of one tensor (so that summing them back-propagates seven scalars
    and not seven zero-padded vectors)."""

    @staticmethod
    def forward(ctx, cls, bbox, surface, side_all, iou_all, iou, iou_j, targets, config, ticket,
                quality=None, detach_sigma=False, sigma_mode=0):
        """quality (B, K, 6): the unsupervised variant -- semantic, centre, surface and IoU terms
        with the pseudo labels' side qualities in the weights, the other three terms zero."""
        backend = backend_for(cls)
        loss, saved = backend.head_loss_forward(cls, bbox, surface, side_all, iou_all,
                                                iou.reshape(-1), iou_j.reshape(-1), targets, config,
                                                ticket, quality=quality, detach_sigma=detach_sigma,
                                                sigma_mode=sigma_mode)
        ctx.saved, ctx.label, ctx.k, ctx.iou_shape = saved, targets['mask_targets'], bbox.shape[1], iou.shape
        if sigma_mode:          # (the SAQE extras read the arg-max classes)
            ctx.mark_non_differentiable(saved['sem_pick'])
            return tuple(loss.unbind(0)) + (saved['sem_pick'],)
        return tuple(loss.unbind(0))

    @staticmethod
    def backward(ctx, *gs):
        gs = gs[:7]
        ref = next(g for g in gs if g is not None)
        g = torch.stack([gi if gi is not None else torch.zeros_like(ref) for gi in gs])
        backend = backend_for(g)
        d = backend.head_loss_backward(g, ctx.label, ctx.saved, ctx.k)
        return (d['cls'], d['bbox'], d['surface'], d['side'], d['iou_s'],
                d['iou'].view(ctx.iou_shape), None, None, None, None, None, None, None)


def config_of(head):
    """The eleven scalars of the shipped configuration, or None when the head's loss modules are
    not the ones the kernel implements."""
    from . import losses as L
    o, s, c = head.objectness_loss, head.semantic_loss, head.center_loss
    su, io, q, sd = head.surface_loss, head.iou_loss, head.iou_pred_loss, head.side_loss
    ok = (isinstance(o, L.CrossEntropyLoss) and o.reduction == 'sum' and o.class_weight is not None
          and len(o.class_weight) == 2
          and isinstance(s, L.CrossEntropyLoss) and s.reduction == 'sum' and s.class_weight is None
          and isinstance(c, L.ChamferDistance) and c.mode == 'l2' and c.reduction == 'sum'
          and isinstance(su, L.SurfaceLoss) and su.func_type == 'MSELoss'
          and isinstance(io, L.IoU3DLoss)
          and isinstance(q, L.GeneralQualityFocalLoss) and not q.use_sigmoid and q.beta == 2.0
          and q.reduction == 'sum'
          and isinstance(sd, L.SidePredLoss) and sd.label_func_type == 'SmoothL1Loss'
          and isinstance(sd.loss_func, L.MSELoss) and sd.loss_func.reduction == 'sum')
    if not ok:
        return None
    return [head.alpha, o.loss_weight, o.class_weight[0], o.class_weight[1], s.loss_weight,
            c.loss_src_weight, c.loss_dst_weight, su.loss_func.loss_weight, io.loss_weight,
            q.loss_weight, sd.loss_func.loss_weight]


def unsup_config_of(head):
    """The scalars the unsupervised variant reads (semantic, centre, surface, IoU), or None."""
    from . import losses as L
    s, c, su, io = head.semantic_loss, head.center_loss, head.surface_loss, head.iou_loss
    ok = (isinstance(s, L.CrossEntropyLoss) and s.reduction == 'sum' and s.class_weight is None
          and isinstance(c, L.ChamferDistance) and c.mode == 'l2' and c.reduction == 'sum'
          and isinstance(su, L.SurfaceLoss) and su.func_type == 'MSELoss'
          and isinstance(io, L.IoU3DLoss))
    if not ok:
        return None
    return [head.alpha, 0.0, 0.0, 0.0, s.loss_weight, c.loss_src_weight, c.loss_dst_weight,
            su.loss_func.loss_weight, io.loss_weight, 0.0, 0.0]


def usable(head, bbox_preds, unsup=False):
    """True when ``NesieHead.loss`` (``unsup``: ``unsup_loss``, also of a subclass) can take the
    fused path for these predictions."""
    if not ENABLED or (type(head).__name__ != 'NesieHead' and not unsup):
        return False
    need = ('_cls_all', '_side_all', '_iou_all')
    if any(k not in bbox_preds for k in need):
        return False
    cls = bbox_preds['_cls_all']
    if backend_for(cls).name != 'hip' or cls.dtype != torch.float32:
        return False
    B, K = bbox_preds['bbox_preds'].shape[:2]
    C = cls.shape[1] - 2
    return (cls.shape[0] == B and cls.shape[2] == K and C <= 32
            and tuple(bbox_preds['_side_all'].shape) == (6, B, C, 2 * K)
            and tuple(bbox_preds['_iou_all'].shape) == (B, 2 * K, C)
            and (unsup_config_of(head) if unsup else config_of(head)) is not None)


SAQE_TERMS = ('r_objectness_loss', 'angle_loss', 'angle_pred_loss', 'side_jitter_loss')


class SaqeExtraFn(Function):
    """(robj, rot, bbox, side_all) -> the SAQE head's four additional supervised terms
    (``nesie_saqe_extra_loss_forward``), ``sup`` = the semi-supervised stage's form."""

    @staticmethod
    def forward(ctx, robj, rot, bbox, side_all, bbox_t, jsurf, targets, sem_pick, config, sup, ticket):
        backend = backend_for(rot)
        loss, saved = backend.saqe_extra_forward(robj, rot, bbox, bbox_t, jsurf, side_all, targets,
                                                 sem_pick, config, sup, ticket)
        ctx.saved, ctx.label, ctx.k, ctx.bshape = saved, targets['mask_targets'], bbox.shape[1], bbox.shape
        return tuple(loss.unbind(0))

    @staticmethod
    def backward(ctx, *gs):
        ref = next(g for g in gs if g is not None)
        g = torch.stack([gi if gi is not None else torch.zeros_like(ref) for gi in gs])
        d = backend_for(g).saqe_extra_backward(g, ctx.label, ctx.saved, ctx.k)
        d_bbox = d['angle'].new_zeros(ctx.bshape)
        d_bbox[..., 6] = d['angle'].view(ctx.bshape[:2])
        return (d['robj'], d['rot'], d_bbox, d['side'], None, None, None, None, None, None, None)


def saqe_config_of(head):
    """(base 11 scalars with alpha 0, the 7 scalars of the extras) of the shipped SAQE configuration,
    or None."""
    from . import losses as L
    base = config_of(head)
    al, ap = getattr(head, 'angle_loss', None), getattr(head, 'angle_pred_loss', None)
    if base is None or not (isinstance(al, L.SmoothL1Loss) and isinstance(ap, L.MSELoss)
                            and ap.reduction == 'sum'):
        return None
    o, sd = head.objectness_loss, head.side_loss
    base = [0.0] + base[1:]
    return base, [o.loss_weight, o.class_weight[0], o.class_weight[1], al.loss_weight, al.beta,
                  ap.loss_weight, sd.loss_func.loss_weight]


def saqe_usable(head, bbox_preds):
    """True when ``SAQEHead.loss`` / ``sup_loss`` can take the fused path."""
    if not ENABLED or type(head).__name__ != 'SAQEHead':
        return False
    if any(k not in bbox_preds for k in ('_cls_all', '_side_all', '_iou_all', '_rot_all', '_robj_all')):
        return False
    cls = bbox_preds['_cls_all']
    if backend_for(cls).name != 'hip' or cls.dtype != torch.float32:
        return False
    B, K = bbox_preds['bbox_preds'].shape[:2]
    C = cls.shape[1] - 2
    return (cls.shape[0] == B and cls.shape[2] == K and C <= 32
            and tuple(bbox_preds['_side_all'].shape) == (6, B, C, 2 * K)
            and tuple(bbox_preds['_iou_all'].shape) == (B, 2 * K, C)
            and tuple(bbox_preds['_rot_all'].shape) == (B, 2 * K, C)
            and tuple(bbox_preds['_robj_all'].shape) == (B, 2 * K, 2)
            and saqe_config_of(head) is not None)


class VoteLossFn(Function):
    """``VoteModule.get_loss`` for one vote per seed as one launch each way."""

    @staticmethod
    def forward(ctx, vote_points, seed_points, seed_indices, mask, targets, w_dst, ticket):
        backend = backend_for(vote_points)
        loss, sign, scale = backend.vote_loss_forward(
            seed_points.contiguous(), vote_points.contiguous(), seed_indices.contiguous(),
            mask.contiguous(), targets.contiguous(), w_dst, ticket)
        ctx.save_for_backward(sign, scale)
        ctx.shape = vote_points.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        sign, scale = ctx.saved_tensors
        d = backend_for(g).vote_loss_backward(g.contiguous(), scale, sign)
        return d.view(ctx.shape), None, None, None, None, None, None


def vote_loss_usable(vote_module, vote_points, seed_indices, mask, targets):
    from . import losses as L
    vl = getattr(vote_module, 'vote_loss', None)
    return (ENABLED and isinstance(vl, L.ChamferDistance) and vl.mode == 'l1' and vl.reduction == 'none'
            and vote_module.vote_per_seed == 1 and backend_for(vote_points).name == 'hip'
            and vote_points.dtype == torch.float32 and seed_indices.dtype == torch.int64
            and mask.dtype == torch.int64 and targets.dtype == torch.float32)
