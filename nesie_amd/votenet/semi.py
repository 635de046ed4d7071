"""Semi-supervised student/teacher step of Nesie (``mmdet3d/models/detectors/votenet_nesie.py:14-324``,
``core/utils/simi_teacher_hook.py:7-92``, the statistics set up in
``core/utils/simi_runner_hook.py:59-67`` / ``runner/simi_epoch_based_runner.py:72-86``).

Same arithmetic as the reference, restructured so the whole step stays on the device:

* pseudo labels keep a fixed (B, 64) shape with a validity mask and are compacted with a
  stable sort, instead of the reference's per-scene python lists (``:279-298``);
* the same-class LHS-NMS runs as a native kernel (``nesie_lhs_nms_samecls``) instead of
  numpy on the host after a device->host copy (``:219-260``);
* the teacher->student box re-augmentation is batched tensor algebra on the device
  instead of per-scene CPU ``DepthInstance3DBoxes`` objects (``:310-324``, ``:596-634``);
* the per-class pseudo-label histogram is a device ``scatter_add`` (``:301-308``).

Quirks of the reference that change numbers are kept as coded and marked ``# sic``.
"""
import math

import torch
from torch import nn
from torch.nn import functional as F

from ..kernels import backend_for
from ..mmdet3d_ops.norm import deferred_bn_counters
from .detector import VoteNet
from .nesie_head import GTBatch

MAX_NUM_OBJ = 64


# ---- augmentation bookkeeping ----------------------------------------------------------
class AugMeta:
    """Per-scene 3-D augmentation of a batch (the ``img_metas`` keys the reference reads:
    ``transformation_3d_flow``, ``pcd_rotation``, ``pcd_scale_factor``, ``pcd_trans`` and the
    flip entries of the flow).  The pipeline order is fixed -- ``flow`` -- and a flip that was
    not drawn for a scene is simply absent from its flow (flag 0 here)."""

    def __init__(self, flip_h, flip_v, rot_mat, scale, trans, flow=('HF', 'VF', 'R', 'S', 'T')):
        self.flip_h, self.flip_v = flip_h.bool(), flip_v.bool()   # (B,)
        self.rot_mat, self.scale, self.trans = rot_mat, scale, trans  # (B,3,3) (B,) (B,3)
        self.flow = tuple(flow)

    @staticmethod
    def identity(batch, device):
        return AugMeta(torch.zeros(batch, dtype=torch.bool, device=device),
                       torch.zeros(batch, dtype=torch.bool, device=device),
                       torch.eye(3, device=device).expand(batch, 3, 3).contiguous(),
                       torch.ones(batch, device=device), torch.zeros(batch, 3, device=device))

    @staticmethod
    def random(batch, device, generator=None, strong=True):
        """Flip / rotate / scale / translate ranges of the ScanNet semi-sup pipeline
        (``configs/Nesie/nesie-votenet-scannet-train-010.py:175-268``)."""
        g = generator
        r = lambda *s: torch.rand(*s, generator=g)  # noqa: E731
        fh, fv = r(batch) < 0.5, r(batch) < 0.5
        ang = (r(batch) - 0.5) * (2 * 0.087266 if strong else 0.0)
        c, s = torch.cos(ang), torch.sin(ang)
        z, o = torch.zeros(batch), torch.ones(batch)
        # mmdet3d stores rot_mat_T of points.rotate(angle) as img_metas['pcd_rotation']
        rot = torch.stack([torch.stack([c, -s, z], -1), torch.stack([s, c, z], -1),
                           torch.stack([z, z, o], -1)], -2)
        return AugMeta(fh.to(device), fv.to(device), rot.to(device), o.to(device),
                       torch.zeros(batch, 3, device=device))

    def apply_points(self, pts):
        """The forward augmentation on (B,N,3+) points (used to build synthetic views)."""
        xyz = pts[..., :3].clone()
        for op in self.flow:
            if op == 'HF':
                xyz[..., 0] = torch.where(self.flip_h[:, None], -xyz[..., 0], xyz[..., 0])
            elif op == 'VF':
                xyz[..., 1] = torch.where(self.flip_v[:, None], -xyz[..., 1], xyz[..., 1])
            elif op == 'R':
                xyz = torch.bmm(xyz, self.rot_mat)   # points @ rot_mat_T
            elif op == 'S':
                xyz = xyz * self.scale[:, None, None]
            elif op == 'T':
                xyz = xyz + self.trans[:, None, :]
        return torch.cat([xyz, pts[..., 3:]], dim=-1)


def _rotate_boxes(boxes, mat):
    """``DepthInstance3DBoxes.rotate(matrix)`` (depth_box3d.py:119-149), batched:
    rot_mat_T = mat^T, centres @ rot_mat_T, yaw -= atan2(rot_mat_T[0,1], rot_mat_T[0,0])."""
    rot_t = mat.transpose(1, 2)
    xyz = torch.bmm(boxes[..., :3], rot_t)
    ang = torch.atan2(rot_t[:, 0, 1], rot_t[:, 0, 0])
    yaw = boxes[..., 6] - ang[:, None]
    return torch.cat([xyz, boxes[..., 3:6], yaw.unsqueeze(-1)], dim=-1)


def _flip_boxes(boxes, flag, direction):
    """``DepthInstance3DBoxes.flip`` (depth_box3d.py:176-199) where ``flag`` is set."""
    f = flag[:, None]
    if direction == 'horizontal':
        x = torch.where(f, -boxes[..., 0], boxes[..., 0])
        yaw = torch.where(f, -boxes[..., 6] + math.pi, boxes[..., 6])
        return torch.cat([x.unsqueeze(-1), boxes[..., 1:6], yaw.unsqueeze(-1)], dim=-1)
    y = torch.where(f, -boxes[..., 1], boxes[..., 1])
    yaw = torch.where(f, -boxes[..., 6], boxes[..., 6])
    return torch.cat([boxes[..., 0:1], y.unsqueeze(-1), boxes[..., 2:6], yaw.unsqueeze(-1)], dim=-1)


def untransform_boxes(boxes, meta):
    """Undo the teacher view's augmentation (``votenet_nesie.py:596-614``): the flow in
    reverse; 'T' by -trans, 'R' by the stored matrix, 'S' by 1/scale."""
    for op in meta.flow[::-1]:
        if op == 'T':
            boxes = torch.cat([boxes[..., :3] - meta.trans[:, None, :], boxes[..., 3:]], -1)
        elif op == 'R':
            boxes = _rotate_boxes(boxes, meta.rot_mat)
        elif op == 'S':  # sic: guarded by 'pcd_trans' in the reference; both keys always exist
            s = (1.0 / meta.scale)[:, None, None]
            boxes = torch.cat([boxes[..., :6] * s, boxes[..., 6:]], -1)
        elif op == 'VF':
            boxes = _flip_boxes(boxes, meta.flip_v, 'vertical')
        elif op == 'HF':
            boxes = _flip_boxes(boxes, meta.flip_h, 'horizontal')
    return boxes


def transform_boxes(boxes, meta):
    """Apply the student view's augmentation (``votenet_nesie.py:616-634``); 'R' uses the
    transposed stored matrix."""
    for op in meta.flow:
        if op == 'T':
            boxes = torch.cat([boxes[..., :3] + meta.trans[:, None, :], boxes[..., 3:]], -1)
        elif op == 'R':
            boxes = _rotate_boxes(boxes, meta.rot_mat.transpose(1, 2))
        elif op == 'S':
            s = meta.scale[:, None, None]
            boxes = torch.cat([boxes[..., :6] * s, boxes[..., 6:]], -1)
        elif op == 'VF':
            boxes = _flip_boxes(boxes, meta.flip_v, 'vertical')
        elif op == 'HF':
            boxes = _flip_boxes(boxes, meta.flip_h, 'horizontal')
    return boxes


# ---- EMA teacher ------------------------------------------------------------------------
class EMATeacher:
    """``SimiTeacherHook``: every parameter has an EMA copy registered on the model as the
    buffer ``ema_<name with dots as underscores>`` (so it rides in the state dict);
    ``update(step)`` after each optimiser step, ``swap()`` around the teacher forward."""

    def __init__(self, model, momentum=0.001, interval=1, warm_up=10):
        assert isinstance(interval, int) and interval > 0 and 0 < momentum < 1
        self.momentum = momentum ** interval
        self.interval, self.warm_up = interval, warm_up
        # Only NAMES are kept: nn.Module.to() / .double() / .cuda() replace buffer tensors (the
        # Parameter objects survive, their .data is swapped), so tensors cached here would be
        # orphans of the original device after model.to('cuda').  The reference resolves its
        # buffers in before_run, i.e. after the model has been moved (simi_teacher_hook.py:39-52).
        self.model = model
        self.names = []
        for name, p in list(model.named_parameters(recurse=True)):
            buf_name = f"ema_{name.replace('.', '_')}"
            model.register_buffer(buf_name, p.data.clone())
            self.names.append((name, buf_name))
        self._bound = None

    def _bind(self):
        """(parameters, EMA buffers) of the model as it is NOW; re-resolved whenever a buffer
        object, device or dtype has changed since the last call."""
        bufs = self.model._buffers
        b = self._bound
        if b is not None and all(bufs[bn] is e and e.device == p.device and e.dtype == p.dtype
                                 for (_, bn), p, e in zip(self.names, b[0], b[1])):
            return b
        named = dict(self.model.named_parameters(recurse=True))
        params = [named[n] for n, _ in self.names]
        emas = [bufs[bn] for _, bn in self.names]
        self._bound = (params, emas)
        self._scratch = None
        return self._bound

    @property
    def params(self):
        return self._bind()[0]

    @property
    def emas(self):
        return self._bind()[1]

    def use_flat(self, state):
        """The EMA copies as views of ONE flat vector laid out like ``state.flat_param``
        (``dp.FlatTrainState``: the parameters are views of it already): ``update`` is then two
        element-wise launches over the vector and ``swap`` three copies of it, element for element
        the per-tensor arithmetic (round 5: 38 multi-tensor launches + 6 multi-tensor copies per
        step otherwise).  A no-op unless every tracked parameter lives in the state."""
        params, emas = self._bind()
        offs = {id(q): o for q, o in zip(state.params, state.offsets)}
        if len(params) != len(state.params) or any(id(q) not in offs for q in params):
            return False
        flat_p = state.flat_param.data
        flat_e = torch.zeros_like(flat_p)
        with torch.no_grad():
            for (name, buf_name), q, e in zip(self.names, params, emas):
                o = offs[id(q)]
                v = flat_e[o:o + q.numel()].view_as(q)
                v.copy_(e)
                self.model._buffers[buf_name] = v
        ends = (0, len(params) - 1)
        self._flat = (state, flat_p, flat_e, torch.empty_like(flat_p), [(i, offs[id(params[i])]) for i in ends])
        self._bound = None
        return True

    def _flat_ok(self):
        """The flat pair while it still IS the storage of parameters and EMA copies (a ``.to()`` /
        ``load_state_dict(assign=True)`` replaces tensors: the per-tensor path takes over again)."""
        f = getattr(self, '_flat', None)
        if f is None:
            return None
        _, flat_p, flat_e, _, ends = f
        params, emas = self._bind()
        es = flat_p.element_size()
        for i, o in ends:
            if params[i].data_ptr() != flat_p.data_ptr() + o * es or emas[i].data_ptr() != flat_e.data_ptr() + o * es:
                self._flat = None
                return None
        return f

    def resync(self):
        """EMA copies <- current parameters (e.g. after loading pre-trained weights)."""
        params, emas = self._bind()
        with torch.no_grad():
            torch._foreach_copy_(emas, [p.data for p in params])

    def update(self, curr_step):
        if curr_step % self.interval != 0:
            return
        m = min(self.momentum, (1 + curr_step) / (self.warm_up + curr_step))
        f = self._flat_ok()
        with torch.no_grad():
            if f is not None:
                f[2].mul_(1 - m).add_(f[1], alpha=m)
                return
            params, emas = self._bind()
            torch._foreach_mul_(emas, 1 - m)
            torch._foreach_add_(emas, [p.data for p in params], alpha=m)

    def swap(self):
        """Parameters <-> EMA copies, as three multi-tensor copies through a scratch list kept
        between calls (one clone launch per parameter otherwise: ~440 tiny copies per step)."""
        f = self._flat_ok()
        if f is not None:
            with torch.no_grad():
                f[3].copy_(f[1])
                f[1].copy_(f[2])
                f[2].copy_(f[3])
            return
        params, emas = self._bind()
        with torch.no_grad():
            data = [p.data for p in params]
            tmp = getattr(self, '_scratch', None)
            if tmp is None or len(tmp) != len(data) or any(
                    t.shape != p.shape or t.device != p.device or t.dtype != p.dtype
                    for t, p in zip(tmp, data)):
                tmp = self._scratch = [torch.empty_like(p) for p in data]
            torch._foreach_copy_(tmp, data)
            torch._foreach_copy_(data, emas)
            torch._foreach_copy_(emas, tmp)


# ---- pseudo-label statistics ---------------------------------------------------------------
class PseudoLabelState:
    """``model.ulb_list`` (per unlabeled scene, per class pseudo-label counts), ``ulb_flag``
    (1 until a scene was pseudo-labelled once) and the labeled/unlabeled set sizes."""

    def __init__(self, num_labeled, num_unlabeled, num_classes, device):
        self.num_labeled, self.num_unlabeled = num_labeled, num_unlabeled
        self.ulb_list = torch.zeros(num_unlabeled, num_classes, device=device)
        self.ulb_flag = torch.ones(num_unlabeled, device=device)

    def classwise_acc(self, thresh_warmup=True):
        """``votenet_nesie.py:133-147`` as coded: the value written to class slot c is the
        c-th LARGEST count (``sorted[i]`` indexed by class id), not class c's count."""
        counter = self.ulb_list.sum(dim=0)
        srt, _ = torch.sort(counter, descending=True)
        if thresh_warmup:
            ulb_count = 10 * self.ulb_flag.sum() * self.num_labeled / self.num_unlabeled
            denom = torch.maximum(srt.max(), ulb_count)
        else:
            denom = srt.max()
        acc = srt / denom  # sic
        return acc / (2.0 - acc)

    def update(self, rows, labels, valid):
        """``ulb_update`` (:301-308): rows (U,) long = positions of the batch's unlabeled
        scenes in the unlabeled set; labels/valid (U,64)."""
        num_classes = self.ulb_list.shape[1]
        hist = torch.zeros(rows.shape[0], num_classes, device=labels.device)
        hist.scatter_add_(1, labels.clamp(0, num_classes - 1), valid.float())
        self.ulb_flag.index_fill_(0, rows, 0.0)   # no host scalar copy: hipGraph-safe
        self.ulb_list.index_copy_(0, rows, hist)


# ---- the detector -----------------------------------------------------------------------------
class VoteNetNesie(VoteNet):
    """Student/teacher step: student forward+loss on all scenes (supervised terms on the
    labeled ones, unsupervised terms against the teacher's filtered pseudo boxes on the
    unlabeled ones), teacher = EMA weights, forward only."""

    # the three places VoteNetSAQE differs (votenet_saqe.py:121,170,201)
    objectness_key = 'obj_scores'
    quality_coef = (5 / 3, 8 / 3)
    sup_loss_name = 'loss'

    def __init__(self, backbone, bbox_head, train_cfg=None, test_cfg=None, ema=None,
                 num_classes=18, head_type='NesieHead'):
        super().__init__(backbone, bbox_head, train_cfg, test_cfg, head_type=head_type)
        self.num_classes = num_classes
        self.teacher = EMATeacher(self, **(ema or dict(momentum=0.001, interval=1, warm_up=10)))
        self.state = None
        self._index_cache = {}

    def init_label_state(self, num_labeled, num_unlabeled, device):
        self.state = PseudoLabelState(num_labeled, num_unlabeled, self.num_classes, device)

    def _batch_index(self, flags, device):
        key = (tuple(bool(f) for f in flags), str(device))
        if key not in self._index_cache:
            sup = [i for i, f in enumerate(key[0]) if f]
            uns = [i for i, f in enumerate(key[0]) if not f]
            self._index_cache[key] = (torch.tensor(sup, dtype=torch.long, device=device),
                                      torch.tensor(uns, dtype=torch.long, device=device))
        return self._index_cache[key]

    @staticmethod
    def _select(bbox_preds, index):
        # ('_*' entries keep their producers' layouts: the scene axis of '_side_all' is axis 1)
        return {k: v.index_select(1 if k == '_side_all' else 0, index) for k, v in bbox_preds.items()}

    # -- pseudo labels (:129-299) -------------------------------------------------------
    def get_pseudo_labels(self, preds, dataset_name='ScanNet'):
        """-> labels (B,64) long, boxes (B,64,7) bottom-centre, quality (B,64,6), valid
        (B,64) bool; valid entries first (stable), the rest is padding."""
        cfg = self.train_cfg
        bp = preds['bbox_preds']
        bp = torch.cat([bp[..., :2], (bp[..., 2] - bp[..., 5] * 0.5).unsqueeze(-1), bp[..., 3:]], -1)
        pred_center, pred_size, pred_heading = bp[..., :3], bp[..., 3:6], bp[..., 6:7]
        B, K = pred_center.shape[:2]
        sem = preds['sem_scores']
        max_cls, argmax_cls = torch.max(sem, dim=2)
        flat = argmax_cls.reshape(-1)
        if cfg.get('use_cbl', True):
            acc = self.state.classwise_acc(cfg.get('thresh_warmup', True))
            # sic: `[acc[flat[i]] for i in flat]` indexes the proposal list BY CLASS ID
            threshold = acc[flat[flat]].reshape(B, K)
            cls_threshold = (0.7 + 0.3 * threshold).clamp(max=0.95)
            iou_threshold = (0.25 + threshold * 0.5).clamp(max=0.35)
        else:
            cls_threshold, iou_threshold = 0.9, 0.25
        cls_mask = max_cls > cls_threshold  # sic: raw class logits, not probabilities
        obj = torch.softmax(preds[self.objectness_key], dim=2)
        pos_obj, neg_obj = obj[..., 1], obj[..., 0]
        objectness_mask = pos_obj > 0.9
        iou_pred = preds['iou_scores'].gather(2, argmax_cls.unsqueeze(-1)).squeeze(-1)
        final_mask = cls_mask & objectness_mask & (iou_pred > iou_threshold)
        side = preds['side_scores'].detach()  # (B,K,6,C)
        s = side.gather(3, argmax_cls[:, :, None, None].expand(-1, -1, 6, 1)).squeeze(-1)
        quality = self.quality_coef[0] * s * s - self.quality_coef[1] * s + torch.ones_like(s)

        score = pos_obj * iou_pred * final_mask
        inds = torch.argsort(score, dim=1, descending=True, stable=True)[:, :MAX_NUM_OBJ]
        k = inds.shape[1]
        take3 = inds.unsqueeze(-1).expand(-1, -1, 3)
        final_sorted = torch.gather(final_mask, 1, inds)
        centre = torch.gather(pred_center, 1, take3)
        size = torch.gather(pred_size, 1, take3)
        heading = torch.gather(pred_heading, 1, inds.unsqueeze(-1))
        cls = torch.gather(argmax_cls, 1, inds)
        # LHS-NMS on upright boxes: the reference builds corners with heading 0 (ScanNet)
        # around the BOTTOM centre as if it were the centre (sic), in camera axes
        # (x, -z, y); min/max per axis, score = pos_obj * iou, class in the last column
        half = size * 0.5
        lo = torch.stack([centre[..., 0] - half[..., 0], -centre[..., 2] - half[..., 2],
                          centre[..., 1] - half[..., 1]], -1)
        hi = torch.stack([centre[..., 0] + half[..., 0], -centre[..., 2] + half[..., 2],
                          centre[..., 1] + half[..., 1]], -1)
        nms_score = torch.gather(pos_obj, 1, inds) * torch.gather(iou_pred, 1, inds)
        nms_in = torch.cat([lo, hi, nms_score.unsqueeze(-1), cls.float().unsqueeze(-1)], -1)
        keep = torch.empty(B, k, dtype=torch.uint8, device=nms_in.device)
        backend_for(nms_in).lhs_nms_samecls(nms_in.contiguous().float(), 0.25, keep)
        valid = final_sorted & keep.bool()

        quality = torch.gather(quality, 1, inds.unsqueeze(-1).expand(-1, -1, 6))
        boxes = torch.cat([centre, size, heading], dim=-1)
        # valid entries first, original order kept (the reference stacks them in order)
        order = torch.argsort((~valid).int(), dim=1, stable=True)
        g = lambda t: torch.gather(t, 1, order.view(B, k, *([1] * (t.dim() - 2))).expand_as(t))  # noqa: E731
        return g(cls), g(boxes), g(quality), torch.gather(valid, 1, order)

    @staticmethod
    def _pseudo_gt(labels, boxes, valid):
        """Padded pseudo ground truth in GTBatch form; a scene without any pseudo box gets
        the reference's all-zero fake box (nesie_head.py:537-544)."""
        cnt = valid.sum(1)
        b = torch.where(valid.unsqueeze(-1), boxes, torch.full_like(boxes, 0.0))
        far = torch.zeros_like(b)
        far[..., :3] = 1e6
        col = torch.arange(b.shape[1], device=b.device).unsqueeze(0)
        fake_col = (cnt == 0).unsqueeze(1) & (col == 0)
        b = torch.where(valid.unsqueeze(-1) | fake_col.unsqueeze(-1), b, far)
        return GTBatch(b, torch.where(valid, labels, torch.zeros_like(labels)),
                       torch.clamp(cnt, min=1), valid.float())

    # -- the step (:69-127) -------------------------------------------------------------------
    def forward_train(self, points_s, points_t, gt_labeled, use_label, meta_s, meta_t,
                      unlabeled_rows, precomputed=None):
        """points_* (B,N,4) student / teacher views of the same B scenes; gt_labeled =
        GTBatch of the labeled scenes (in batch order); use_label = python list of B bools;
        meta_* = AugMeta of each view; unlabeled_rows (U,) long = positions of the batch's
        unlabeled scenes in the unlabeled set.  ``precomputed`` = dict(student=..., teacher=...)
        of ``backbone.sample_and_group_indices`` results for the two views (optional: the
        index chains depend on the input coordinates only)."""
        pre = precomputed or {}
        cfg = self.train_cfg
        name = cfg.get('dataset_name', 'ScanNet')
        with deferred_bn_counters():
            x_s = self.extract_feat(points_s, pre.get('student'))
            if self.keep_head_inputs:
                self.head_inputs = [x_s['fp_features'][-1]]
            preds_s = self.bbox_head(x_s, cfg['sample_mod'], name)
            with torch.no_grad():
                self.teacher.swap()                  # call_hook("switch_to_teacher")
                x_t = self.extract_feat(points_t, pre.get('teacher'))
                preds_t = self.bbox_head(x_t, cfg['sample_mod'], name)
        with torch.no_grad():
            labels, boxes, quality, valid = self.get_pseudo_labels(preds_t, name)
            boxes = transform_boxes(untransform_boxes(boxes, meta_t), meta_s)
            self.teacher.swap()                      # call_hook("switch_to_student")
        sup_idx, uns_idx = self._batch_index(use_label, points_s.device)
        sup_losses = getattr(self.bbox_head, self.sup_loss_name)(self._select(preds_s, sup_idx),
                                         points_s.index_select(0, sup_idx), gt_labeled, None)
        u = lambda t: t.index_select(0, uns_idx)  # noqa: E731
        self.state.update(unlabeled_rows, u(labels), u(valid))
        unsup_losses = self.bbox_head.unsup_loss(
            self._select(preds_s, uns_idx), u(points_s),
            self._pseudo_gt(u(labels), u(boxes), u(valid)), None,
            pseudo_quality_score=u(quality) * u(valid).unsqueeze(-1))
        return {**sup_losses, **unsup_losses}


class VoteNetSAQE(VoteNetNesie):
    """``mmdet3d/models/detectors/votenet_saqe.py``: the same step with the SAQE head's
    ``sup_loss``, the quality head's objectness for the pseudo-label filter and the
    0.8 s^2 - 1.8 s + 1 side quality."""
    objectness_key = 'R_obj_scores'
    quality_coef = (0.8, 1.8)
    sup_loss_name = 'sup_loss'


def build_saqe_votenet_semi(cfg=None):
    """SAQE-VoteNet semi-supervised (``configs/SAQE/saqe-votenet-scannet-train-010.py``)."""
    import copy
    from .detector import saqe_votenet_scannet_cfg
    cfg = copy.deepcopy(cfg or saqe_votenet_scannet_cfg())
    cfg['bbox_head']['iou_pred_loss']['loss_weight'] = 1.0
    cfg['train_cfg'].update(dataset_name='ScanNet', thresh_warmup=True, use_cbl=True)
    return VoteNetSAQE(cfg['backbone'], cfg['bbox_head'], cfg['train_cfg'], cfg['test_cfg'],
                       ema=dict(momentum=0.001, interval=1, warm_up=10), head_type='SAQEHead')


def build_nesie_votenet_semi(cfg=None):
    """Nesie-VoteNet with the semi-supervised wrapper
    (``configs/Nesie/nesie-votenet-scannet-train-010.py``: QFL weight 1.0, EMA 0.001/10)."""
    import copy
    from .detector import nesie_votenet_scannet_cfg
    cfg = copy.deepcopy(cfg or nesie_votenet_scannet_cfg())
    cfg['bbox_head']['iou_pred_loss']['loss_weight'] = 1.0
    cfg['train_cfg'].update(dataset_name='ScanNet', thresh_warmup=True, use_cbl=True)
    return VoteNetNesie(cfg['backbone'], cfg['bbox_head'], cfg['train_cfg'], cfg['test_cfg'],
                        ema=dict(momentum=0.001, interval=1, warm_up=10))
