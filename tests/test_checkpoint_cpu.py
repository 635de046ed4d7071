"""Reference checkpoint envelope: key names, DDP prefix, EMA buffers."""
import os
from collections import OrderedDict

import pytest
import torch

from nesie_amd import checkpoint
from nesie_amd.votenet import build_nesie_votenet
from nesie_amd.votenet import semi


# names the reference's modules produce (point_sa_module.py:277-289,
# side_pooling_module.py:55-78,346-358, simi_teacher_hook.py:39-52)
REFERENCE_KEYS = [
    'backbone.SA_modules.0.mlps.0.layer0.conv.weight',
    'backbone.SA_modules.0.mlps.0.layer0.bn.weight',
    'backbone.SA_modules.0.mlps.0.layer0.bn.running_mean',
    'backbone.SA_modules.3.mlps.0.layer2.bn.num_batches_tracked',
    'backbone.FP_modules.1.mlps.layer1.conv.weight',
    'bbox_head.vote_module.vote_conv.0.conv.weight',
    'bbox_head.vote_module.conv_out.bias',
    'bbox_head.vote_aggregation.mlps.0.layer0.conv.weight',
    'bbox_head.grid_conv.mlps_before.3.first_conv.0.weight',
    'bbox_head.grid_conv.mlps_before.6.second_conv.3.bias',
    'bbox_head.grid_conv.mlps_head.0.4.running_var',
]


def test_state_dict_uses_the_reference_names():
    keys = set(build_nesie_votenet().state_dict().keys())
    for k in REFERENCE_KEYS:
        assert k in keys, k


def test_reference_envelope_round_trip(tmp_path):
    torch.manual_seed(0)
    src = build_nesie_votenet()
    opt = torch.optim.AdamW(src.parameters(), lr=1e-3)
    paths = checkpoint.save_reference_checkpoint(src, tmp_path, epoch=3, iteration=120,
                                                 optimizer=opt, meta={'CLASSES': ('a',)})
    assert [os.path.basename(p) for p in paths] == ['epoch_3.pth']
    assert os.path.islink(tmp_path / 'latest.pth')
    raw = torch.load(tmp_path / 'latest.pth')
    assert set(raw) == {'meta', 'state_dict', 'optimizer'}
    assert raw['meta']['epoch'] == 3 and raw['meta']['iter'] == 120
    assert all(v.device.type == 'cpu' for v in raw['state_dict'].values())
    # as written from inside DistributedDataParallel: every key carries 'module.'
    raw['state_dict'] = OrderedDict(('module.' + k, v) for k, v in raw['state_dict'].items())
    dst = build_nesie_votenet()
    with torch.no_grad():
        for p in dst.parameters():
            p.add_(1.0)
    meta, optim = checkpoint.load_reference_checkpoint(dst, raw)
    assert meta['iter'] == 120 and optim is not None
    for (ka, a), (kb, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert ka == kb and torch.equal(a, b), ka


def test_pretrain_checkpoint_into_semi_model_and_ema_copy(tmp_path):
    torch.manual_seed(1)
    pre = build_nesie_votenet()
    ckpt = {'meta': {'epoch': 36, 'iter': 1}, 'state_dict': pre.state_dict()}
    model = semi.build_nesie_votenet_semi()
    ema_names = [k for k in model.state_dict() if k.startswith('ema_')]
    assert 'ema_backbone_SA_modules_0_mlps_0_layer0_conv_weight' in ema_names
    checkpoint.load_reference_checkpoint(model, ckpt)            # ema_* may be missing
    model.teacher.resync()
    with torch.no_grad():                                        # make the teacher differ
        model.backbone.SA_modules[0].mlps[0].layer0.conv.weight.mul_(2.0)
    paths = checkpoint.save_reference_checkpoint(model, tmp_path, epoch=1, iteration=7,
                                                 ema_copy=True)
    assert [os.path.basename(p) for p in paths] == ['epoch_1.pth', 'epoch_1_ema.pth']
    k = 'backbone.SA_modules.0.mlps.0.layer0.conv.weight'
    student = torch.load(paths[0])['state_dict'][k]
    teacher = torch.load(paths[1])['state_dict'][k]
    assert torch.equal(student, 2.0 * teacher)                   # swapped in, then back
    assert torch.equal(model.state_dict()[k], student)
    with pytest.raises(RuntimeError, match='does not match'):
        checkpoint.load_reference_checkpoint(pre, {'state_dict': {'nope': torch.zeros(1)}})
