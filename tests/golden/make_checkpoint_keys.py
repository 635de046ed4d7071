#!/usr/bin/env python
"""Generate tests/golden/reference_state_dict_keys.json: {key: shape} of the state_dict the
REFERENCE's own module classes produce for the Nesie-VoteNet detector AND the SAQE-VoteNet
detector (BASELINE configs[4]: `SAQEHead` + `QualityEstimation`, saqe_head.py:90-253,
quelity_estimation_module.py:10-285), plus the buffer names its EMA hook registers for each.

Runs only in the build container (needs /root/reference).  Loaded BY PATH inside the sandbox of
make_golden.py (third-party stand-ins only; `mmdet3d/__init__.py` never runs):
  mmdet3d/ops/pointnet_modules/{builder,point_sa_module,point_fp_module}.py   (the reference's
      PointSAModule / PointFPModule: `mlps.<i>.layer<j>.{conv,bn}` names, point_sa_module.py:277-289)
  mmdet3d/models/backbones/{base_pointnet,pointnet2_sa_ssg}.py
  mmdet3d/models/dense_heads/nesie_head.py (+ vote_module, reliable_conv_bbox_module,
      side_pooling_module), built with the reference's OWN PointSAModule as vote aggregation
  mmdet3d/models/dense_heads/saqe_head.py (+ quelity_estimation_module), likewise
  mmdet3d/core/utils/simi_teacher_hook.py   (hooks_before_run: `ema_<name with . -> _>` buffers, :39-52)
Only names and shapes are written -- no weights, no source.
"""
import importlib
import json
import os
import sys

import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.golden import make_golden as mg  # noqa: E402
from tests.golden import golden_inputs  # noqa: E402

REF = mg.REF


class _Stub(nn.Module):
    """Parameter-free grouping / sampling operators: only their presence matters for the names."""

    def __init__(self, *a, **k):
        super().__init__()


def main():
    ref_head_mod = mg.install_reference_sandbox()
    r = os.path.join(REF, 'mmdet3d')
    ops = sys.modules['mmdet3d.ops']
    for n in ('GroupAll', 'PAConv', 'Points_Sampler', 'QueryAndGroup'):
        setattr(ops, n, _Stub)
    ops.gather_points = ops.three_interpolate = ops.three_nn = None
    mg._pkg('mmdet3d.ops.pointnet_modules', os.path.join(r, 'ops', 'pointnet_modules'))
    builder = importlib.import_module('mmdet3d.ops.pointnet_modules.builder')
    importlib.import_module('mmdet3d.ops.pointnet_modules.point_sa_module')
    fp = importlib.import_module('mmdet3d.ops.pointnet_modules.point_fp_module')
    ops.build_sa_module, ops.PointFPModule = builder.build_sa_module, fp.PointFPModule
    BACKBONES = mg.Registry('backbone')
    sys.modules['mmdet.models'].BACKBONES = BACKBONES
    mg._pkg('mmdet3d.models.backbones', os.path.join(r, 'models', 'backbones'))
    bb = importlib.import_module('mmdet3d.models.backbones.pointnet2_sa_ssg')

    cfg = golden_inputs.head_cfg()
    cfg['bbox_head']['grid_conv_cfg']['mean_size_arr_path'] = os.path.join(
        REF, 'data/scannet/meta_data/scannet_means.npz')
    # the head module bound build_sa_module at import (our stand-in, for the numeric goldens):
    # point it at the reference's builder for this fixture
    ref_head_mod.build_sa_module = builder.build_sa_module
    for name, m in list(sys.modules.items()):
        if name.startswith('mmdet3d.models') and getattr(m, 'build_sa_module', None) is not None:
            m.build_sa_module = builder.build_sa_module
    from nesie_amd.votenet import nesie_votenet_scannet_cfg
    mine_cfg = nesie_votenet_scannet_cfg()
    bcfg = {k: v for k, v in mine_cfg['backbone'].items() if k != 'type'}

    saqe_mod = importlib.import_module('mmdet3d.models.dense_heads.saqe_head')
    saqe_mod.build_sa_module = builder.build_sa_module
    scfg = golden_inputs.saqe_head_cfg()
    scfg['bbox_head']['grid_conv_cfg']['mean_size_arr_path'] = cfg['bbox_head']['grid_conv_cfg'][
        'mean_size_arr_path']

    class Detector(nn.Module):      # single_stage.py: self.backbone, self.bbox_head
        def __init__(self, head):
            super().__init__()
            self.backbone = bb.PointNet2SASSG(**bcfg)
            self.bbox_head = head

    # the EMA hook's buffers (hooks_before_run only; update / swap are not run)
    mg._mod('mmcv.parallel', is_module_wrapper=lambda m: False)
    rn = sys.modules['mmcv.runner']
    rn.HOOKS, rn.Hook = mg.Registry('hook'), object
    mg._pkg('mmdet3d.core.utils', os.path.join(r, 'core', 'utils'))
    hook_cls = importlib.import_module('mmdet3d.core.utils.simi_teacher_hook').SimiTeacherHook

    def dump(head):
        det = Detector(head)
        assert type(det.bbox_head.vote_aggregation).__module__ == 'mmdet3d.ops.pointnet_modules.point_sa_module'
        assert type(det.backbone.SA_modules[0]).__module__ == 'mmdet3d.ops.pointnet_modules.point_sa_module'
        plain = {k: list(v.shape) for k, v in det.state_dict().items()}
        hook_cls().hooks_before_run(det)
        ema = {k: list(v.shape) for k, v in det.state_dict().items() if k not in plain}
        return plain, ema

    plain, ema = dump(ref_head_mod.NesieHead(**cfg['bbox_head'], train_cfg=cfg['train_cfg'],
                                             test_cfg=cfg['test_cfg']))
    splain, sema = dump(saqe_mod.SAQEHead(**scfg['bbox_head'], train_cfg=scfg['train_cfg'],
                                          test_cfg=scfg['test_cfg']))
    out = {'source': 'reference classes loaded by path (tests/golden/make_checkpoint_keys.py)',
           'backbone_cfg': {k: v for k, v in bcfg.items() if isinstance(v, (int, float, str, list, tuple))},
           'state_dict': plain, 'ema_buffers': ema,
           'saqe_state_dict': splain, 'saqe_ema_buffers': sema}
    path = os.path.join(ROOT, 'tests', 'golden', 'reference_state_dict_keys.json')
    json.dump(out, open(path, 'w'), indent=0, sort_keys=True)
    print('wrote', path, len(plain), 'keys +', len(ema), 'ema buffers; SAQE', len(splain), '+', len(sema), ';',
          os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    with torch.no_grad():
        main()
