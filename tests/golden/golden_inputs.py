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


def saqe_head_cfg():
    cfg = head_cfg()
    cfg['bbox_head'].update(
        angle_loss=dict(type='SmoothL1Loss', reduction='sum', loss_weight=10.0),
        angle_pred_loss=dict(type='MSELoss', reduction='sum', loss_weight=1.0))
    return cfg


def build_my_saqe_head():
    from nesie_amd.votenet.saqe_head import SAQEHead
    cfg = saqe_head_cfg()
    torch.manual_seed(1)
    head = SAQEHead(**cfg['bbox_head'], train_cfg=cfg['train_cfg'], test_cfg=cfg['test_cfg'])
    head.train()
    return head


def pseudo_quality(boxes):
    """Per-box (K_i, 6) side qualities for the unsup_loss goldens."""
    g = torch.Generator().manual_seed(10)
    return [torch.rand(b.shape[0], 6, generator=g) for b in boxes]


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


# ---- semi-supervised path ----------------------------------------------------------------
def nms_boxes():
    """(3, 64, 8) axis-aligned boxes with clusters of same-class overlaps and score ties."""
    g = torch.Generator().manual_seed(21)
    c = torch.rand(3, 8, 3, generator=g) * 4
    centre = c[:, torch.arange(64) % 8] + (torch.rand(3, 64, 3, generator=g) - 0.5) * 0.4
    half = 0.3 + torch.rand(3, 64, 3, generator=g) * 0.4
    score = torch.rand(3, 64, generator=g)
    score[:, 40:] = 0.0                       # the masked-out tail of the top-64 list
    score[:, 5] = score[:, 6]                 # an exact tie
    cls = torch.randint(0, 3, (3, 64), generator=g).float()
    return torch.cat([centre - half, centre + half, score.unsqueeze(-1), cls.unsqueeze(-1)], -1)


def teacher_preds():
    """A teacher output dict with enough confident proposals to survive the filters."""
    g = torch.Generator().manual_seed(22)
    B, K, C = 3, 256, 18
    centres = torch.rand(B, 12, 3, generator=g) * torch.tensor([6.0, 6.0, 1.5])
    which = torch.randint(0, 12, (B, K), generator=g)
    centre = torch.gather(centres, 1, which.unsqueeze(-1).expand(-1, -1, 3)) \
        + torch.randn(B, K, 3, generator=g) * 0.05
    size = 0.5 + torch.rand(B, K, 3, generator=g)
    heading = (torch.rand(B, K, 1, generator=g) - 0.5) * 0.2
    sem = torch.randn(B, K, C, generator=g)
    sem.scatter_(2, (which % C).unsqueeze(-1), 3.0 + torch.rand(B, K, 1, generator=g))
    obj = torch.stack([torch.zeros(B, K), 1.5 + 3 * torch.rand(B, K, generator=g)], -1)
    obj[:, ::3] = obj[:, ::3].flip(-1)        # a third are confident negatives
    iou = torch.rand(B, K, C, generator=g) * 0.9
    side = torch.rand(B, K, 6, C, generator=g)
    vote = torch.rand(B, 1024, 3, generator=g)
    return dict(bbox_preds=torch.cat([centre, size, heading], -1), sem_scores=sem, obj_scores=obj,
                iou_scores=iou, side_scores=side, vote_points=vote)


def ulb_statistics():
    g = torch.Generator().manual_seed(23)
    ulb_list = torch.randint(0, 4, (108, 18), generator=g).float()
    ulb_list[60:] = 0
    ulb_flag = torch.ones(108)
    ulb_flag[:60] = 0
    return ulb_list, ulb_flag, 12, 108


def aug_metas():
    """Teacher / student augmentation of 3 scenes, as AugMeta fields."""
    g = torch.Generator().manual_seed(24)
    out = []
    for _ in range(2):
        ang = (torch.rand(3, generator=g) - 0.5) * 0.3
        c, s_, z, o = torch.cos(ang), torch.sin(ang), torch.zeros(3), torch.ones(3)
        rot = torch.stack([torch.stack([c, -s_, z], -1), torch.stack([s_, c, z], -1),
                           torch.stack([z, z, o], -1)], -2)
        out.append(dict(flip_h=torch.tensor([True, False, True]) ^ (len(out) == 1),
                        flip_v=torch.tensor([False, True, True]),
                        rot_mat=rot, scale=0.9 + 0.2 * torch.rand(3, generator=g),
                        trans=torch.randn(3, 3, generator=g) * 0.1))
    return out  # [teacher, student]


# ---- test path: NMS, corners, get_bboxes, indoor_eval ---------------------------------------
def aligned_nms_cases():
    """boxes (3,96,6) in clusters (heavy same-class overlap), scores without ties, classes."""
    g = torch.Generator().manual_seed(31)
    c = torch.rand(3, 10, 3, generator=g) * 5
    centre = c[:, torch.arange(96) % 10] + (torch.rand(3, 96, 3, generator=g) - 0.5) * 0.5
    half = 0.25 + torch.rand(3, 96, 3, generator=g) * 0.5
    boxes = torch.cat([centre - half, centre + half], -1)
    boxes[2, 90:] = boxes[2, 90:91]               # identical boxes
    boxes[1, 7, 3:] = boxes[1, 7, :3]             # a degenerate (zero-volume) box
    scores = torch.rand(3, 96, generator=g)
    classes = torch.randint(0, 4, (3, 96), generator=g)
    return boxes, scores, classes


def corner_boxes():
    g = torch.Generator().manual_seed(32)
    b = torch.cat([torch.randn(40, 3, generator=g) * 2, 0.2 + torch.rand(40, 3, generator=g) * 2,
                   (torch.rand(40, 1, generator=g) - 0.5) * 6.2], -1)
    b[:5, 6] = 0
    return b


def detect_inputs():
    """points (3,4096,4) and a prediction dict (K = 64 proposals): gravity-centre boxes around
    the scene's GT boxes (some far away = empty), peaked class scores, random objectness and
    per-class IoU scores."""
    pts, gts, _ = make_batch(78, 3, num_points=4096)
    g = torch.Generator().manual_seed(33)
    B, K, C = 3, 64, 18
    boxes = []
    for b in range(B):
        gt = gts[b]
        pick = torch.arange(K) % gt.shape[0]
        centre = gt[pick, :3].clone()
        centre[:, 2] += gt[pick, 5] * 0.5
        centre += torch.randn(K, 3, generator=g) * 0.08
        size = gt[pick, 3:6] * (0.7 + 0.6 * torch.rand(K, 3, generator=g))
        yaw = gt[pick, 6:7] + (torch.rand(K, 1, generator=g) - 0.5) * 0.3
        centre[K - 6:] += 40.0                                    # off-scene proposals: empty
        boxes.append(torch.cat([centre, size, yaw], -1))
    sem = torch.randn(B, K, C, generator=g)
    sem.scatter_(2, torch.randint(0, 3, (B, K, 1), generator=g), 4.0)
    obj = torch.randn(B, K, 2, generator=g) * 2
    iou = torch.rand(B, K, C, generator=g)
    return pts, dict(bbox_preds=torch.stack(boxes), sem_scores=sem, obj_scores=obj,
                     iou_scores=iou)


def eval_annos():
    """6 scenes, 5 classes (class 4 is predicted but has no ground truth; scene 3 has no GT):
    gt_annos in the dataset's form, detections as (boxes (n,7) bottom-origin, scores, labels)."""
    g = torch.Generator().manual_seed(34)
    gt_annos, dets = [], []
    for s in range(6):
        n = 0 if s == 3 else 3 + s
        centre = torch.rand(n, 3, generator=g) * torch.tensor([5.0, 5.0, 1.0])
        size = 0.4 + torch.rand(n, 3, generator=g)
        yaw = (torch.rand(n, 1, generator=g) - 0.5) * 2.0
        cls = torch.randint(0, 4, (n,), generator=g)
        gt = torch.cat([centre, size, yaw], -1)
        gt_annos.append(dict(gt_num=n, gt_boxes_upright_depth=gt.numpy().copy(),
                             **{'class': cls.numpy().copy()}))
        m = 2 * n + 4
        src = torch.arange(m) % max(n, 1)
        if n:
            dc = centre[src] + torch.randn(m, 3, generator=g) * 0.12
            ds = size[src] * (0.8 + 0.4 * torch.rand(m, 3, generator=g))
            dy = yaw[src] + torch.randn(m, 1, generator=g) * 0.15
            dl = cls[src].clone()
        else:
            dc = torch.rand(m, 3, generator=g) * 5
            ds = 0.4 + torch.rand(m, 3, generator=g)
            dy = torch.zeros(m, 1)
            dl = torch.randint(0, 4, (m,), generator=g)
        dl[-2:] = torch.tensor([4, (s % 4)])
        dc[:, 2] -= ds[:, 2] * 0.5                                 # detections are bottom-origin
        dets.append((torch.cat([dc, ds, dy], -1), torch.rand(m, generator=g), dl))
    return gt_annos, dets


# ---- input side: raw scenes for the loading / augmentation pipeline ------------------------
INPUT_CASES = [
    # name, seed, raw points, sampled, with_yaw, rot_range, scale_range, translation_std
    ('scannet_big', 41, 60000, 40000, False, (-0.087266, 0.087266), (1.0, 1.0), (0, 0, 0)),
    ('scannet_small', 42, 30000, 40000, False, (-0.087266, 0.087266), (1.0, 1.0), (0, 0, 0)),
    ('sunrgbd_like', 43, 50000, 20000, True, (-0.523599, 0.523599), (0.85, 1.15), (0.1, 0.1, 0.05)),
]


def raw_scene(seed, n, with_yaw):
    """An un-aligned raw scan: (n,6) xyz+rgb float32, its axis-align matrix (4,4) float64 (as
    the info files store it), gravity-centre GT boxes (T,6|7) and labels."""
    from nesie_amd.scenes import make_scene
    pts, boxes, labels = make_scene(seed, n)
    g = torch.Generator().manual_seed(seed)
    ang = float(torch.rand(1, generator=g)) * 1.2 - 0.6
    c, s = np.cos(ang), np.sin(ang)
    align = np.eye(4)
    align[:3, :3] = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    align[:3, 3] = (torch.rand(3, generator=g).numpy() - 0.5) * 4
    # raw = R^-1 (aligned - t), so that GlobalAlignment lands near the synthetic room
    xyz = pts[:, :3].double().numpy()
    raw = (xyz - align[:3, 3]) @ align[:3, :3]          # (R^T (p - t))^T = (p - t) R
    rgb = torch.rand(n, 3, generator=g).numpy() * 255
    raw6 = np.concatenate([raw, rgb], 1).astype(np.float32)
    gt = boxes.clone()
    gt[:, 2] += gt[:, 5] * 0.5                            # bottom -> gravity centre
    if with_yaw:
        gt[:, 6] = (torch.rand(gt.shape[0], generator=g) - 0.5) * 2
    else:
        gt = gt[:, :6]
    return raw6, align, gt.numpy().astype(np.float32), labels.numpy()
