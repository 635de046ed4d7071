#!/usr/bin/env python
"""Generate golden vectors from the REFERENCE's own Python for the detector-head path.

Runs only in the build container (needs /root/reference; the GPU box never sees it).
The reference package cannot be imported as a whole (`import mmdet3d` ->
ModuleNotFoundError: mmcv, an ordinary error), so its leaf files are loaded BY PATH under
their real dotted names, with
  * tiny stand-ins for the third-party pieces that are not in the reference tree
    (mmcv ConvModule / BaseModule / registries, mmdet weighted_loss / MSE / L1 / SmoothL1 /
    CrossEntropy, multi_apply) written here from their published semantics
     -- deliberately NOT the nesie_amd implementations;
  * the native ops (`mmdet3d.ops`, `mmcv.ops.three_nn`, `sort_vertices`) served by
    nesie_amd.mmdet3d_ops on the CPU oracle back end (the reference has no CPU ops);
  * `Tensor.cuda()` made a no-op (the reference hard-codes `.cuda()`).
What runs from the reference, unmodified: nesie_head.py (forward, loss, get_targets*),
side_pooling_module.py, reliable_conv_bbox_module.py, vote_module.py, the five loss files,
oriented_iou_loss.py + box_intersection_2d.py + cuda_op/cuda_ext.py.

Weights and inputs are regenerated from seeds at test time (nesie_amd's head is built
under torch.manual_seed and its state_dict is loaded into the reference head), so the
fixtures hold only OUTPUTS: tests/golden/nesie_head_golden.pt (< 100 KB).
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("NESIE_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from nesie_amd import kernels  # noqa: E402
from nesie_amd import mmdet3d_ops as my_ops  # noqa: E402
from nesie_amd.votenet.boxes import DepthInstance3DBoxes  # noqa: E402
from tests.golden import golden_inputs  # noqa: E402


# --------------------------------------------------------------------------------------
# third-party stand-ins (mmcv 1.3.17 / mmdet 2.19 semantics)
# --------------------------------------------------------------------------------------
class Registry:
    def __init__(self, name):
        self.name, self.module_dict = name, {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls
        return deco

    def get(self, key):
        return self.module_dict.get(key)

    def __contains__(self, key):
        return key in self.module_dict

    def build(self, cfg):
        cfg = dict(cfg)
        return self.module_dict[cfg.pop('type')](**cfg)


class StubConvModule(nn.Module):
    """conv -> norm -> act; bias='auto' => bias iff no norm; plain F.conv (not the bmm path)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), bias='auto',
                 inplace=True, **kw):
        super().__init__()
        ctype = (conv_cfg or dict(type='Conv2d'))['type']
        if bias == 'auto':
            bias = norm_cfg is None
        self.conv = dict(Conv1d=nn.Conv1d, Conv2d=nn.Conv2d)[ctype](
            in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=bias)
        self.norm_name = None
        if norm_cfg is not None:
            t = norm_cfg['type']
            if t in ('BN1d', 'BN'):
                self.norm_name = 'bn'; self.add_module('bn', nn.BatchNorm1d(out_channels))
            elif t == 'BN2d':
                self.norm_name = 'bn'; self.add_module('bn', nn.BatchNorm2d(out_channels))
            elif t == 'GN':
                self.norm_name = 'gn'
                self.add_module('gn', nn.GroupNorm(norm_cfg['num_groups'], out_channels))
        self.activate = nn.ReLU(inplace=inplace) if act_cfg is not None else None

    def forward(self, x):
        x = self.conv(x)
        if self.norm_name:
            x = getattr(self, self.norm_name)(x)
        if self.activate is not None:
            x = self.activate(x)
        return x


def build_conv_layer(cfg, *args, **kwargs):
    t = (cfg or dict(type='Conv2d'))['type']
    return dict(Conv1d=nn.Conv1d, Conv2d=nn.Conv2d)[t](*args, **kwargs)


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None, *a, **k):
        super().__init__()
        self.init_cfg = init_cfg


def _identity_decorator_factory(*a, **k):
    def deco(fn):
        return fn
    return deco


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return dict(none=lambda l: l, mean=lambda l: l.mean(), sum=lambda l: l.sum())[reduction](loss)
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction == 'none':
        return loss
    raise ValueError


def weighted_loss(loss_func):
    def wrapper(pred, target, weight=None, reduction='mean', avg_factor=None, **kwargs):
        loss = loss_func(pred, target, **kwargs)
        return weight_reduce_loss(loss, weight, reduction, avg_factor)
    return wrapper


def _mk_elementwise(fn):
    class L(nn.Module):
        def __init__(self, *args, **kw):
            super().__init__()
            names = ['reduction', 'loss_weight'] if fn is not _smooth else ['beta', 'reduction', 'loss_weight']
            d = dict(beta=1.0, reduction='mean', loss_weight=1.0)
            d.update(dict(zip(names, args))); d.update(kw)
            self.beta, self.reduction, self.loss_weight = d['beta'], d['reduction'], d['loss_weight']

        def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kw):
            red = reduction_override if reduction_override else self.reduction
            l = fn(pred, target, self.beta)
            return self.loss_weight * weight_reduce_loss(l, weight, red, avg_factor)
    return L


def _mse(p, t, b): return (p - t) ** 2
def _l1(p, t, b): return (p - t).abs()
def _smooth(p, t, b):
    d = (p - t).abs()
    return torch.where(d < b, 0.5 * d * d / b, d - 0.5 * b)


class StubCrossEntropyLoss(nn.Module):
    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None,
                 loss_weight=1.0):
        super().__init__()
        self.reduction, self.class_weight, self.loss_weight = reduction, class_weight, loss_weight

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, **kw):
        red = reduction_override if reduction_override else self.reduction
        cw = cls_score.new_tensor(self.class_weight) if self.class_weight is not None else None
        loss = F.cross_entropy(cls_score, label, weight=cw, reduction='none')
        if weight is not None:
            weight = weight.float()
        return self.loss_weight * weight_reduce_loss(loss, weight, red, avg_factor)


def multi_apply(func, *args, **kwargs):
    from functools import partial
    pfunc = partial(func, **kwargs) if kwargs else func
    return tuple(map(list, zip(*map(pfunc, *args))))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, path, **attrs):
    m = _mod(name, **attrs)
    m.__path__ = [path]
    return m


def install_reference_sandbox():
    torch.Tensor.cuda = lambda self, *a, **k: self  # the reference hard-codes .cuda()
    LOSSES, HEADS = Registry('loss'), Registry('head')
    MSELoss, L1Loss, SmoothL1Loss = _mk_elementwise(_mse), _mk_elementwise(_l1), _mk_elementwise(_smooth)
    for n, c in dict(MSELoss=MSELoss, L1Loss=L1Loss, SmoothL1Loss=SmoothL1Loss,
                     CrossEntropyLoss=StubCrossEntropyLoss).items():
        LOSSES.module_dict[n] = c
    # --- third party ---
    _mod('mmcv', is_tuple_of=lambda seq, t: isinstance(seq, tuple) and all(isinstance(i, t) for i in seq))
    _mod('mmcv.cnn', ConvModule=StubConvModule)
    _mod('mmcv.cnn.bricks', build_conv_layer=build_conv_layer)
    _mod('mmcv.runner', BaseModule=BaseModule, force_fp32=_identity_decorator_factory,
         auto_fp16=_identity_decorator_factory)
    _mod('mmcv.ops', three_nn=my_ops.three_nn)
    _mod('mmcv.utils', Registry=Registry)
    _mod('mmdet')
    _mod('mmdet.core', multi_apply=multi_apply)
    _mod('mmdet.models', HEADS=HEADS, LOSSES=LOSSES)
    _mod('mmdet.models.builder', HEADS=HEADS, LOSSES=LOSSES)
    _mod('mmdet.models.losses', MSELoss=MSELoss, L1Loss=L1Loss, SmoothL1Loss=SmoothL1Loss,
         CrossEntropyLoss=StubCrossEntropyLoss, FocalLoss=None, binary_cross_entropy=None)
    _mod('mmdet.models.losses.utils', weighted_loss=weighted_loss)

    def sort_vertices_forward(vertices, mask, num_valid):
        idx = torch.empty(vertices.shape[0], vertices.shape[1], 9, dtype=torch.int32)
        oracle.OracleKernels().sort_vertices_forward(vertices.contiguous(), mask.contiguous(),
                                                     num_valid.contiguous(), idx)
        return idx
    _mod('sort_vertices', sort_vertices_forward=sort_vertices_forward)
    # --- reference packages as path-only shells (their __init__.py is NOT executed) ---
    r = os.path.join(REF, 'mmdet3d')
    _pkg('mmdet3d', r)
    _pkg('mmdet3d.ops', os.path.join(r, 'ops'), build_sa_module=my_ops.build_sa_module,
         furthest_point_sample=my_ops.furthest_point_sample)
    _pkg('mmdet3d.ops.rotated_iou', os.path.join(r, 'ops', 'rotated_iou'))
    _pkg('mmdet3d.ops.rotated_iou.cuda_op', os.path.join(r, 'ops', 'rotated_iou', 'cuda_op'))
    # min_enclosing_box.py (GIoU only, off the hot path) uses np.int, removed in numpy 2
    _mod('mmdet3d.ops.rotated_iou.min_enclosing_box', smallest_bounding_box=None)
    oi = importlib.import_module('mmdet3d.ops.rotated_iou.oriented_iou_loss')
    sys.modules['mmdet3d.ops.rotated_iou'].cal_iou_3d = oi.cal_iou_3d
    sys.modules['mmdet3d.ops.rotated_iou'].cal_giou_3d = oi.cal_giou_3d
    _mod('mmdet3d.core', DepthInstance3DBoxes=DepthInstance3DBoxes)
    _mod('mmdet3d.core.bbox', AxisAlignedBboxOverlaps3D=object)
    _mod('mmdet3d.core.post_processing', aligned_3d_nms=None)
    _pkg('mmdet3d.models', os.path.join(r, 'models'))
    _mod('mmdet3d.models.builder', build_loss=lambda cfg: LOSSES.build(cfg))
    lp = _pkg('mmdet3d.models.losses', os.path.join(r, 'models', 'losses'))
    for f in ['chamfer_distance', 'surface_loss', 'side_pred_loss', 'iou3d_loss', 'gfocal_loss']:
        importlib.import_module(f'mmdet3d.models.losses.{f}')
    lp.chamfer_distance = sys.modules['mmdet3d.models.losses.chamfer_distance'].chamfer_distance
    mu = _pkg('mmdet3d.models.model_utils', os.path.join(r, 'models', 'model_utils'))
    mu.VoteModule = importlib.import_module('mmdet3d.models.model_utils.vote_module').VoteModule
    _pkg('mmdet3d.models.dense_heads', os.path.join(r, 'models', 'dense_heads'))
    return importlib.import_module('mmdet3d.models.dense_heads.nesie_head')


def semi_goldens():
    """votenet_nesie.py by path: lhs_3d_faster_samecls, get_pseudo_labels,
    transformation_bbox_preds -- with the reference's OWN DepthInstance3DBoxes
    (core/bbox/structures/{utils,base_box3d,depth_box3d}.py loaded by path)."""
    r = os.path.join(REF, 'mmdet3d')
    core = sys.modules['mmdet3d.core']
    _mod('mmdet3d.ops.iou3d', iou3d_cuda=None)
    sys.modules['mmdet3d.ops'].points_in_boxes_batch = my_ops.points_in_boxes_batch
    _mod('mmdet3d.core.points', BasePoints=type('BasePoints', (), {}))
    _pkg('mmdet3d.core.bbox.structures', os.path.join(r, 'core', 'bbox', 'structures'))
    sys.modules['mmdet3d.core.bbox'].__path__ = [os.path.join(r, 'core', 'bbox')]
    dep = importlib.import_module('mmdet3d.core.bbox.structures.depth_box3d')
    RefBoxes = dep.DepthInstance3DBoxes
    core.DepthInstance3DBoxes = RefBoxes
    core.bbox3d2result = None
    core.merge_aug_bboxes_3d = None
    DET = Registry('det')
    sys.modules['mmdet.models'].DETECTORS = DET

    class SingleStage3DDetector(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
    _pkg('mmdet3d.models.detectors', os.path.join(r, 'models', 'detectors'))
    _mod('mmdet3d.models.detectors.single_stage', SingleStage3DDetector=SingleStage3DDetector)
    sys.modules['mmcv'].build_from_cfg = None
    rn = sys.modules['mmcv.runner']
    rn.HOOKS, rn.Hook, rn.Priority, rn.get_priority = Registry('hook'), object, None, None
    vn = importlib.import_module('mmdet3d.models.detectors.votenet_nesie')
    out = {}
    # (a) LHS-NMS
    nb = golden_inputs.nms_boxes()
    picks = torch.zeros(nb.shape[0], nb.shape[1], dtype=torch.uint8)
    for i in range(nb.shape[0]):
        pick = vn.lhs_3d_faster_samecls(nb[i].numpy().astype(np.float64), 0.25, False)
        picks[i, torch.as_tensor(np.array(pick, dtype=np.int64))] = 1
    out['semi/nms_keep'] = picks
    # (b) pseudo labels, (c) re-augmentation
    det = vn.VoteNetNesie.__new__(vn.VoteNetNesie)
    nn.Module.__init__(det)
    ulb_list, ulb_flag, n_lb, n_ulb = golden_inputs.ulb_statistics()
    det.ulb_list, det.ulb_flag = ulb_list, ulb_flag
    det.lb_map, det.ulb_map = list(range(n_lb)), list(range(n_ulb))
    det.CLASSES = list(range(18))
    det.train_cfg = types.SimpleNamespace(thresh_warmup=True, use_cbl=True)
    preds = {k: v.clone() for k, v in golden_inputs.teacher_preds().items()}
    labels, boxes, quality = det.get_pseudo_labels(preds, 'ScanNet')
    out['semi/pl_counts'] = torch.tensor([len(l) for l in labels])
    for i in range(len(labels)):
        out[f'semi/pl_labels/{i}'] = labels[i].clone()
        out[f'semi/pl_boxes/{i}'] = boxes[i].clone()
        out[f'semi/pl_quality/{i}'] = quality[i].clone()
    mt, ms = golden_inputs.aug_metas()

    def metas(m):
        res = []
        for i in range(3):
            flow = (['HF'] if m['flip_h'][i] else []) + (['VF'] if m['flip_v'][i] else []) + ['R', 'S', 'T']
            res.append(dict(transformation_3d_flow=flow, pcd_rotation=m['rot_mat'][i].clone(),
                            pcd_scale_factor=float(m['scale'][i]), pcd_trans=m['trans'][i].clone()))
        return res
    moved = det.transformation_bbox_preds([b.clone() for b in boxes], metas(mt), metas(ms))
    for i in range(3):
        out[f'semi/pl_boxes_student/{i}'] = moved[i].tensor.clone()
    return out


def inference_goldens():
    """The reference's test path, run from its own files (call after semi_goldens(), which
    installs the reference's DepthInstance3DBoxes):
      * core/post_processing/box3d_nms.py::aligned_3d_nms (numba / iou3d imports stubbed:
        the function itself is plain torch);
      * DepthInstance3DBoxes.corners / .overlaps (base_box3d.py, depth_box3d.py) -- the CUDA op
        behind overlaps (`iou3d_cuda.boxes_overlap_bev_gpu`, not buildable here) is served by
        the oracle's restatement, so the overlap AREAS are ours and everything around them is
        the reference's;
      * NesieHead.get_bboxes / multiclass_nms_single (nesie_head.py:681-788);
      * core/evaluation/indoor_eval.py::indoor_eval, with the same stand-in for the BEV op.
    Writes tests/golden/inference_golden.pt."""
    r = os.path.join(REF, 'mmdet3d')
    core = sys.modules['mmdet3d.core']
    RefBoxes = core.DepthInstance3DBoxes
    assert RefBoxes.__module__.startswith('mmdet3d.core.bbox.structures')

    class _Iou3dCuda:
        @staticmethod
        def boxes_overlap_bev_gpu(a, b, out):
            oracle.OracleKernels().boxes_overlap_bev(a.contiguous().float(), b.contiguous().float(), out)
    sys.modules['mmdet3d.ops.iou3d'].iou3d_cuda = _Iou3dCuda
    sys.modules['mmdet3d.core.bbox.structures.base_box3d'].iou3d_cuda = _Iou3dCuda
    _mod('numba', jit=lambda *a, **k: (lambda f: f))
    _mod('mmdet3d.ops.roiaware_pool3d', points_in_boxes_gpu=None, points_in_boxes_cpu=None,
         points_in_boxes_batch=my_ops.points_in_boxes_batch)
    _mod('mmdet3d.ops.iou3d.iou3d_utils', nms_gpu=None, nms_normal_gpu=None)
    sys.modules['mmdet3d.core.post_processing'].__path__ = [os.path.join(r, 'core', 'post_processing')]
    nms_mod = importlib.import_module('mmdet3d.core.post_processing.box3d_nms')
    out = {}
    # (a) aligned_3d_nms
    boxes, scores, classes = golden_inputs.aligned_nms_cases()
    for i in range(boxes.shape[0]):
        out[f'nms/picks/{i}'] = nms_mod.aligned_3d_nms(boxes[i], scores[i], classes[i], 0.25).clone()
    # the masked form of the head: only a subset takes part
    sub = torch.arange(96) % 3 != 0
    out['nms/picks_masked'] = nms_mod.aligned_3d_nms(boxes[0][sub], scores[0][sub],
                                                     classes[0][sub], 0.25).clone()
    # (b) corners
    out['boxes/corners'] = RefBoxes(golden_inputs.corner_boxes()).corners.clone()
    # (c) get_bboxes through the reference head class (no weights involved)
    head_mod = sys.modules['mmdet3d.models.dense_heads.nesie_head']
    head_mod.aligned_3d_nms = nms_mod.aligned_3d_nms
    head = head_mod.NesieHead.__new__(head_mod.NesieHead)
    nn.Module.__init__(head)
    head.num_classes = 18
    pts, preds = golden_inputs.detect_inputs()
    metas = [dict(box_type_3d=RefBoxes) for _ in range(pts.shape[0])]
    for per_class in (True, False):
        head.test_cfg = types.SimpleNamespace(nms_thr=0.25, score_thr=0.05,
                                              per_class_proposal=per_class)
        res = head.get_bboxes(pts, {k: v.clone() for k, v in preds.items()}, metas)
        tag = 'per_class' if per_class else 'single'
        for b, (bx, sc, lb) in enumerate(res):
            out[f'det/{tag}/boxes/{b}'] = bx.tensor.clone()
            out[f'det/{tag}/scores/{b}'] = sc.clone()
            out[f'det/{tag}/labels/{b}'] = lb.clone()
    # (d) overlaps + indoor_eval
    _mod('terminaltables', AsciiTable=type('AsciiTable', (), {
        '__init__': lambda self, data: setattr(self, 'table', ''), }))
    sys.modules['mmcv.utils'].print_log = lambda *a, **k: None
    sys.modules['mmdet3d.core'].__path__ = [os.path.join(r, 'core')]
    _pkg('mmdet3d.core.evaluation', os.path.join(r, 'core', 'evaluation'))
    ev = importlib.import_module('mmdet3d.core.evaluation.indoor_eval')
    gt_annos, dets = golden_inputs.eval_annos()
    dt_annos = [dict(boxes_3d=RefBoxes(b.clone()), scores_3d=s.clone(), labels_3d=l.clone())
                for b, s, l in dets]
    gt0 = RefBoxes(torch.from_numpy(gt_annos[5]['gt_boxes_upright_depth']), origin=(0.5, 0.5, 0.5))
    out['eval/overlaps_scene5'] = RefBoxes.overlaps(dt_annos[5]['boxes_3d'], gt0).clone()
    label2cat = {i: f'cat{i}' for i in range(5)}
    ret = ev.indoor_eval(gt_annos, dt_annos, (0.25, 0.5), label2cat, box_type_3d=RefBoxes,
                         box_mode_3d=2)
    out['eval/keys'] = sorted(ret.keys())
    out['eval/values'] = torch.tensor([ret[k] for k in sorted(ret.keys())], dtype=torch.float64)
    # the same detections with the GT-less class 4 relabelled: finite means
    for d in dt_annos:
        d['labels_3d'] = torch.where(d['labels_3d'] == 4, torch.zeros_like(d['labels_3d']), d['labels_3d'])
    ret = ev.indoor_eval(gt_annos, dt_annos, (0.25, 0.5), label2cat, box_type_3d=RefBoxes,
                         box_mode_3d=2)
    out['eval4/keys'] = sorted(ret.keys())
    out['eval4/values'] = torch.tensor([ret[k] for k in sorted(ret.keys())], dtype=torch.float64)
    ap = ev.average_precision(np.array([[0.1, 0.4, 0.4, 0.9], [0.2, 0.2, 0.5, 1.0]]),
                              np.array([[1.0, 0.5, 0.66, 0.3], [0.9, 0.95, 0.4, 0.2]]))
    out['eval/ap_area'] = torch.from_numpy(ap.copy())
    path = os.path.join(ROOT, 'tests', 'golden', 'inference_golden.pt')
    torch.save(out, path)
    print('wrote', path, os.path.getsize(path), 'bytes;', len(out), 'entries')
    for k in sorted(ret):
        if k.startswith('m'):
            print(k, ret[k])
    return out


def input_goldens():
    """The reference's input pipeline run from its own files on seeded raw scans:
    loading.py::LoadPointsFromFile (on a temporary .bin), transforms_3d.py::GlobalAlignment,
    IndoorPointSample, RandomFlip3D, GlobalRotScaleTrans, with the reference's DepthPoints and
    DepthInstance3DBoxes.  Third-party stand-ins: mmcv.FileClient (reads the file), mmdet's
    RandomFlip base class (restated from mmdet 2.19: one np.random.choice over
    [direction, None] when 'flip' is absent; no image fields here).  The numpy draws are taped
    so that the fixture also pins the ORDER of the random decisions.
    Writes tests/golden/input_golden.pt (sub-sampled rows + checksums)."""
    import tempfile
    r = os.path.join(REF, 'mmdet3d')
    core = sys.modules['mmdet3d.core']
    RefBoxes = core.DepthInstance3DBoxes
    pp = _pkg('mmdet3d.core.points', os.path.join(r, 'core', 'points'))
    for f in ['base_points', 'cam_points', 'depth_points', 'lidar_points']:
        importlib.import_module(f'mmdet3d.core.points.{f}')
    pts_init = importlib.util.spec_from_file_location(
        'mmdet3d.core.points', os.path.join(r, 'core', 'points', '__init__.py'),
        submodule_search_locations=[os.path.join(r, 'core', 'points')])
    pts_mod = importlib.util.module_from_spec(pts_init)
    sys.modules['mmdet3d.core.points'] = pts_mod
    pts_init.loader.exec_module(pts_mod)
    # depth_box3d.py bound the placeholder BasePoints of semi_goldens(); hand it the real class
    sys.modules['mmdet3d.core.bbox.structures.depth_box3d'].BasePoints = pts_mod.BasePoints
    core.VoxelGenerator = None
    sys.modules['mmdet3d.core.bbox'].box_np_ops = None
    PIPE = Registry('pipeline')

    class RandomFlip:   # mmdet/datasets/pipelines/transforms.py (2.19), image-free subset
        def __init__(self, flip_ratio=None, direction='horizontal'):
            self.flip_ratio, self.direction = flip_ratio, direction

        def __call__(self, results):
            if 'flip' not in results:
                direction_list = [self.direction, None]
                non_flip_ratio = 1 - self.flip_ratio
                single_ratio = self.flip_ratio / (len(direction_list) - 1)
                flip_ratio_list = [single_ratio] * (len(direction_list) - 1) + [non_flip_ratio]
                cur_dir = np.random.choice(direction_list, p=flip_ratio_list)
                results['flip'] = cur_dir is not None
            if 'flip_direction' not in results:
                results['flip_direction'] = cur_dir
            return results

    class FileClient:
        def __init__(self, backend='disk'):
            pass

        def get(self, path):
            with open(path, 'rb') as f:
                return f.read()
    sys.modules['mmcv'].FileClient = FileClient
    sys.modules['mmcv.utils'].build_from_cfg = None
    _mod('mmdet.datasets')
    _mod('mmdet.datasets.builder', PIPELINES=PIPE)
    _mod('mmdet.datasets.pipelines', RandomFlip=RandomFlip, LoadAnnotations=object,
         LoadImageFromFile=object)
    _pkg('mmdet3d.datasets', os.path.join(r, 'datasets'))
    _mod('mmdet3d.datasets.builder', OBJECTSAMPLERS=Registry('sampler'))
    _pkg('mmdet3d.datasets.pipelines', os.path.join(r, 'datasets', 'pipelines'))
    _mod('mmdet3d.datasets.pipelines.data_augment_utils', noise_per_object_v3_=None)
    T = importlib.import_module('mmdet3d.datasets.pipelines.transforms_3d')
    Ld = importlib.import_module('mmdet3d.datasets.pipelines.loading')

    out = {}
    for name, seed, n_raw, n_pts, with_yaw, rot, scl, tstd in golden_inputs.INPUT_CASES:
        raw6, align, gt, labels = golden_inputs.raw_scene(seed, n_raw, with_yaw)
        rs = np.random.RandomState(1000 + seed)
        tape = []
        saved = {k: getattr(np.random, k) for k in ('choice', 'rand', 'uniform', 'normal')}

        def taped(k):
            def f(*a, **kw):
                v = getattr(rs, k)(*a, **kw)
                tape.append((k, np.array(v, dtype=object if k == 'choice' and np.ndim(v) == 0 else None)))
                return v
            return f
        for k in saved:
            setattr(np.random, k, taped(k))
        try:
            with tempfile.NamedTemporaryFile(suffix='.bin') as tf:
                raw6.tofile(tf.name)
                res = dict(pts_filename=tf.name, bbox3d_fields=['gt_bboxes_3d'],
                           box_type_3d=RefBoxes, ann_info=dict(axis_align_matrix=align),
                           gt_bboxes_3d=RefBoxes(gt, box_dim=gt.shape[-1], with_yaw=with_yaw,
                                                 origin=(0.5, 0.5, 0.5)))
                res = Ld.LoadPointsFromFile(coord_type='DEPTH', shift_height=True, load_dim=6,
                                            use_dim=[0, 1, 2])(res)
            res = T.GlobalAlignment(rotation_axis=2)(res)
            res = T.IndoorPointSample(num_points=n_pts)(res)
            res = T.RandomFlip3D(sync_2d=False, flip_ratio_bev_horizontal=0.5,
                                 flip_ratio_bev_vertical=0.5)(res)
            res = T.GlobalRotScaleTrans(rot_range=list(rot), scale_ratio_range=list(scl),
                                        translation_std=list(tstd), shift_height=True)(res)
        finally:
            for k, v in saved.items():
                setattr(np.random, k, v)
        pts = res['points'].tensor
        assert pts.shape == (n_pts, 4)
        kinds = [k for k, _ in tape]
        assert kinds == ['choice', 'choice', 'rand', 'rand', 'uniform', 'uniform', 'normal'], kinds
        out[f'input/{name}/rows'] = pts[::64].clone()
        out[f'input/{name}/sum'] = pts.double().sum(0)
        out[f'input/{name}/abs_sum'] = pts.double().abs().sum(0)
        out[f'input/{name}/boxes'] = res['gt_bboxes_3d'].tensor.clone()
        out[f'input/{name}/choices_head'] = torch.from_numpy(np.asarray(tape[0][1][:256], dtype=np.int64))
        out[f'input/{name}/choices_sum'] = torch.tensor(int(np.asarray(tape[0][1], dtype=np.int64).sum()))
        out[f'input/{name}/flips'] = torch.tensor([bool(res['pcd_horizontal_flip']),
                                                   bool(res['pcd_vertical_flip'])])
        out[f'input/{name}/angle_scale'] = torch.tensor([float(tape[4][1]), float(tape[5][1])],
                                                        dtype=torch.float64)
        out[f'input/{name}/trans'] = torch.from_numpy(np.asarray(tape[6][1], dtype=np.float64))
        out[f'input/{name}/flow'] = list(res['transformation_3d_flow'])
    path = os.path.join(ROOT, 'tests', 'golden', 'input_golden.pt')
    torch.save(out, path)
    print('wrote', path, os.path.getsize(path), 'bytes;', len(out), 'entries')
    return out


def main():
    ref_head_mod = install_reference_sandbox()
    out = {}
    with kernels.use_backend(oracle.OracleKernels()):
        # ---- (1) rotated IoU chain, reference torch code + oracle sort ------------------
        ri = sys.modules['mmdet3d.ops.rotated_iou']
        for mode in golden_inputs.IOU_MODES:
            a, b = golden_inputs.iou_boxes(mode)
            out[f'iou3d/{mode}'] = ri.cal_iou_3d(a, b).clone()
        # ---- (2) loss leaf functions ----------------------------------------------------
        L = sys.modules['mmdet3d.models.losses.gfocal_loss']
        pred, label, score, w = golden_inputs.qfl_inputs()
        out['qfl/none'] = L.quality_focal_loss(pred, (label, score), w, beta=2.0, use_sigmoid=False,
                                               reduction='none')
        cd = sys.modules['mmdet3d.models.losses.chamfer_distance']
        s, d = golden_inputs.chamfer_inputs()
        for mode in ('l1', 'l2', 'smooth_l1'):
            r_ = cd.chamfer_distance(s, d, criterion_mode=mode, reduction='none')
            out[f'chamfer/{mode}'] = [t.clone() for t in r_]
        # ---- (3) the whole head: forward + loss, reference code, our weights -------------
        cfg = golden_inputs.head_cfg()
        cfg['bbox_head']['grid_conv_cfg']['mean_size_arr_path'] = os.path.join(
            REF, 'data/scannet/meta_data/scannet_means.npz')
        mine = golden_inputs.build_my_head()
        ref = ref_head_mod.NesieHead(**cfg['bbox_head'], train_cfg=cfg['train_cfg'],
                                     test_cfg=cfg['test_cfg'])
        missing, unexpected = ref.load_state_dict(mine.state_dict(), strict=False)
        assert not unexpected, unexpected
        assert all(k.endswith('num_batches_tracked') or 'project' in k for k in missing), missing
        ref.train()
        feat, points, boxes, labels = golden_inputs.head_inputs()
        noise = golden_inputs.jitter_noise()
        draws = iter(noise)
        real_randn = torch.randn
        torch.randn = lambda *a, **k: next(draws)  # jitter_bbox_preds draws centre then size noise
        try:
            preds = ref(feat, 'vote')
        finally:
            torch.randn = real_randn
        gt = [DepthInstance3DBoxes(b) for b in boxes]
        losses = ref.loss(preds, [p for p in points], gt, [l.clone() for l in labels])
        targets = ref.get_targets([p for p in points], [DepthInstance3DBoxes(b) for b in boxes],
                                  [l.clone() for l in labels], bbox_preds=preds)
        for k, v in losses.items():
            out[f'head/loss/{k}'] = v.detach().clone()
        for k in ['vote_points', 'aggregated_points', 'aggregated_indices', 'obj_scores',
                  'sem_scores', 'surface_pred', 'bbox_preds', 'jitter_bbox_preds', 'iou_scores',
                  'iou_scores_jitter', 'side_scores', 'side_scores_jitter']:
            out[f'head/pred/{k}'] = preds[k].detach().clone()
        names = ['vote_targets', 'vote_target_masks', 'center_targets', 'bbox_targets',
                 'mask_targets', 'valid_gt_masks', 'objectness_targets', 'objectness_weights',
                 'box_loss_weights', 'valid_gt_weights', 'assignment']
        for n, t in zip(names, targets):
            out[f'head/target/{n}'] = (torch.cat(t, 0) if isinstance(t, list) else t).detach().clone()
        # Nesie unsup_loss (GT boxes stand in for pseudo boxes, seeded side qualities)
        q = golden_inputs.pseudo_quality(boxes)
        ul = ref.unsup_loss(preds, [p for p in points], [DepthInstance3DBoxes(b) for b in boxes],
                            [l.clone() for l in labels], None, q)
        for k, v in ul.items():
            out[f'head/unsup/{k}'] = v.detach().clone()
        # ---- (4) SAQE head: saqe_head.py + quelity_estimation_module.py -------------------
        saqe_mod = importlib.import_module('mmdet3d.models.dense_heads.saqe_head')
        scfg = golden_inputs.saqe_head_cfg()
        scfg['bbox_head']['grid_conv_cfg']['mean_size_arr_path'] = os.path.join(
            REF, 'data/scannet/meta_data/scannet_means.npz')
        smine = golden_inputs.build_my_saqe_head()
        sref = saqe_mod.SAQEHead(**scfg['bbox_head'], train_cfg=scfg['train_cfg'],
                                 test_cfg=scfg['test_cfg'])
        missing, unexpected = sref.load_state_dict(smine.state_dict(), strict=False)
        assert not unexpected, unexpected
        assert all(k.endswith('num_batches_tracked') or 'project' in k for k in missing), missing
        sref.train()
        draws = iter(noise)
        torch.randn = lambda *a, **k: next(draws)
        try:
            spreds = sref(feat, 'vote')
        finally:
            torch.randn = real_randn
        mk = lambda: ([p for p in points], [DepthInstance3DBoxes(b) for b in boxes],  # noqa: E731
                      [l.clone() for l in labels])
        for name, fn in (('loss', sref.loss), ('sup_loss', sref.sup_loss)):
            for k, v in fn(spreds, *mk()).items():
                out[f'saqe/{name}/{k}'] = v.detach().clone()
        for k, v in sref.unsup_loss(spreds, *mk(), None, q).items():
            out[f'saqe/unsup/{k}'] = v.detach().clone()
        for k in ['surface_pred', 'surface_scale', 'bbox_preds', 'jitter_bbox_preds',
                  'jitter_surface_preds', 'iou_scores', 'iou_scores_jitter', 'side_scores',
                  'side_scores_jitter', 'rotate_scores', 'rotate_scores_jitter', 'R_obj_scores',
                  'R_obj_scores_jitter']:
            out[f'saqe/pred/{k}'] = spreds[k].detach().clone()
        # vote targets are big (B,N,9): keep a checksum + the non-zero rows count instead
        vt = out.pop('head/target/vote_targets')
        out['head/target/vote_targets_sum'] = vt.double().sum()
        out['head/target/vote_targets_abs_sum'] = vt.double().abs().sum()
        out['head/target/vote_targets_rows'] = vt[:, ::16].clone()
        vm = out.pop('head/target/vote_target_masks')
        out['head/target/vote_target_masks_sum'] = vm.sum()
        out.update(semi_goldens())
        if '--keep-head' not in sys.argv:
            inference_goldens()
            input_goldens()
    if '--inference-only' in sys.argv:
        return
    path = os.path.join(ROOT, 'tests', 'golden', 'nesie_head_golden.pt')
    torch.save(out, path)
    print('wrote', path, os.path.getsize(path), 'bytes;', len(out), 'entries')
    for k in sorted(out):
        if k.startswith('head/loss'):
            print(k, float(out[k]))


if __name__ == '__main__':
    main()
