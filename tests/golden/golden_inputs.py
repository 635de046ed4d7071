"""Seeded inputs shared by make_golden.py (reference side, build container only) and
tests/test_golden.py (our side, runs anywhere).  Only OUTPUTS are stored as fixtures."""
import copy

import numpy as np
import torch

from nesie_amd.scenes import make_batch
from nesie_amd.votenet import nesie_votenet_scannet_cfg
from nesie_amd.votenet.nesie_head import NesieHead
from tests import _cases

IOU_MODES = ['random', 'identical', 'disjoint', 'aligned']


def iou_boxes(mode):
    return _cases.box_pairs(91, 64, mode)


def qfl_inputs():
    g = torch.Generator().manual_seed(5)
    pred = torch.rand(40, 18, generator=g) * 0.98 + 0.01
    label = torch.randint(0, 18, (40,), generator=g)
    score = torch.rand(40, generator=g)
    weight = torch.rand(40, generator=g)
    weight[::4] = 0
    return pred, label, score, weight


def chamfer_inputs():
    g = torch.Generator().manual_seed(6)
    return torch.randn(3, 17, 3, generator=g), torch.randn(3, 5, 3, generator=g)


def head_cfg():
    cfg = copy.deepcopy(nesie_votenet_scannet_cfg())
    cfg['bbox_head']['vote_aggregation_cfg'].update(num_point=32, num_sample=8)
    cfg['bbox_head']['grid_conv_cfg'].update(num_proposal=32)
    # wide thresholds so every loss term has positives at random init
    cfg['train_cfg'].update(pos_distance_thr=1.0, neg_distance_thr=1.5)
    return cfg


def build_my_head():
    cfg = head_cfg()
    torch.manual_seed(0)
    head = NesieHead(**cfg['bbox_head'], train_cfg=cfg['train_cfg'], test_cfg=cfg['test_cfg'])
    head.train()
    return head


def head_inputs():
    """feat_dict as the backbone would hand it over, points, GT boxes, GT labels."""
    pts, boxes, labels = make_batch(77, 2, num_points=2048)
    g = torch.Generator().manual_seed(8)
    seed_idx = torch.stack([torch.randperm(2048, generator=g)[:256] for _ in range(2)])
    seed_xyz = torch.gather(pts[..., :3], 1, seed_idx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    seed_feat = (torch.randn(2, 256, 256, generator=g) * 0.5).clamp_min(0).contiguous()
    feat = dict(fp_xyz=[seed_xyz], fp_features=[seed_feat], fp_indices=[seed_idx])
    # overlapping GT in scene 0 (points inside 2..4 boxes) exercises the vote-slot rules
    b0 = boxes[0].clone()
    b0[1] = b0[0]; b0[1, 3:6] *= 1.3
    b0[2] = b0[0]; b0[2, :2] += 0.15
    boxes = [b0, boxes[1]]
    return feat, pts, boxes, labels


def jitter_noise():
    g = torch.Generator().manual_seed(9)
    return torch.randn(2, 32, 3, generator=g), torch.randn(2, 32, 3, generator=g)
