"""The fused Nesie-head targets / loss kernels against the module-by-module evaluation (the
definition, the same code the CPU oracle path and the reference goldens pin), on the device."""
import copy

import pytest
import torch

from nesie_amd.votenet import head_loss
from nesie_amd.votenet.nesie_head import GTBatch
from tests import _small

pytestmark = pytest.mark.gpu


def _run(model, pts, gt, fused, weights=None):
    head_loss.ENABLED = fused
    try:
        for p in model.parameters():
            p.grad = None
        x = model.extract_feat(pts)
        preds = model.bbox_head(x, 'vote')
        keep = {}
        for k in ('_cls_all', 'bbox_preds', 'surface_pred', '_side_all', '_iou_all'):
            preds[k].retain_grad()
            keep[k] = preds[k]
        assert head_loss.usable(model.bbox_head, preds) == fused
        losses = model.bbox_head.loss(preds, pts, gt, None)
        w = weights or {k: 1.0 for k in losses}
        sum(losses[k] * w[k] for k in losses).backward()
        return ({k: v.detach().clone() for k, v in losses.items()},
                {k: v.grad.detach().clone() for k, v in keep.items()},
                _small.grads_of(model))
    finally:
        head_loss.ENABLED = True


@pytest.mark.parametrize('thresholds', [(0.3, 0.6), (1.0, 1.5)])
def test_fused_loss_matches_the_module_by_module_loss(hip_device, thresholds):
    model = _small.small_model().to(hip_device)
    model.train_cfg['pos_distance_thr'], model.train_cfg['neg_distance_thr'] = thresholds
    model.bbox_head.train_cfg = model.train_cfg
    pts, boxes, labels = _small.small_batch()
    pts = pts.to(hip_device)
    gt = GTBatch.collate(boxes, labels, hip_device)
    model.bbox_head.jitter_noise = tuple(t.to(hip_device) for t in _small.fixed_noise(2, 32))
    # unequal incoming gradients: the backward must scale every term by its own
    weights = dict(vote_loss=1.0, objectness_loss=0.7, semantic_loss=1.3, center_loss=0.9,
                   surface_loss=1.1, iou_loss=0.8, iou_pred_loss=1.2, side_loss=0.6)
    want = _run(model, pts, gt, False, weights)
    got = _run(model, pts, gt, True, weights)
    assert list(got[0]) == list(want[0])
    for k in want[0]:
        torch.testing.assert_close(got[0][k], want[0][k], rtol=2e-5, atol=1e-6, msg=k)
    for k in want[1]:
        scale = want[1][k].abs().max().item()
        torch.testing.assert_close(got[1][k], want[1][k], rtol=1e-4, atol=2e-5 * max(scale, 1e-6), msg=k)
    flat_w = torch.cat([want[2][n].flatten() for n in sorted(want[2])]).double()
    flat_g = torch.cat([got[2][n].flatten() for n in sorted(want[2])]).double()
    assert ((flat_g - flat_w).norm() / flat_w.norm()).item() < 1e-4


def _run_unsup(model, pts, boxes, labels, quality, fused):
    head_loss.ENABLED = fused
    try:
        for p in model.parameters():
            p.grad = None
        preds = model.bbox_head(model.extract_feat(pts), 'vote')
        keep = {}
        for k in ('_cls_all', 'bbox_preds', 'surface_pred', '_side_all', '_iou_all'):
            preds[k].retain_grad()
            keep[k] = preds[k]
        assert head_loss.usable(model.bbox_head, preds, unsup=True) == fused
        gt = GTBatch.collate(boxes, labels, pts.device)
        losses = model.bbox_head.unsup_loss(preds, pts, gt, None, None, quality)
        w = dict(unsup_semantic_loss=1.3, unsup_center_loss=0.9, unsup_iou_loss=0.8, unsup_surface_loss=1.1)
        sum(losses[k] * w[k] for k in losses).backward()
        return ({k: v.detach().clone() for k, v in losses.items()},
                {k: (v.grad.detach().clone() if v.grad is not None else torch.zeros_like(v)) for k, v in keep.items()},
                _small.grads_of(model))
    finally:
        head_loss.ENABLED = True


@pytest.mark.parametrize('kind', ['nesie', 'saqe'])
def test_fused_unsupervised_loss_matches_the_module_by_module_loss(hip_device, kind):
    """NesieHead.unsup_loss (nesie_head.py:415-509) and SAQEHead.unsup_loss (saqe_head.py:706-800:
    alpha 0, detached uncertainties) through nesie_head_loss_forward_unsup: the four terms, the
    gradients at the head's outputs and every parameter gradient; pseudo labels = the ground truth
    with seeded side qualities, one scene without any pseudo box."""
    if kind == 'saqe':
        from nesie_amd.votenet.detector import build_saqe_votenet, saqe_votenet_scannet_cfg
        cfg, scfg = _small.small_cfg(), saqe_votenet_scannet_cfg()
        cfg['bbox_head'].update(angle_loss=scfg['bbox_head']['angle_loss'],
                                angle_pred_loss=scfg['bbox_head']['angle_pred_loss'])
        cfg['head_type'] = 'SAQEHead'
        torch.manual_seed(0)
        model = build_saqe_votenet(cfg)
        assert type(model.bbox_head).__name__ == 'SAQEHead'
    else:
        model = _small.small_model()
    model = model.to(hip_device).train()
    model.train_cfg['pos_distance_thr'], model.train_cfg['neg_distance_thr'] = 1.0, 1.5
    model.bbox_head.train_cfg = model.train_cfg
    pts, boxes, labels = _small.small_batch(batch=3)
    boxes[1], labels[1] = boxes[1][:0], labels[1][:0]
    pts = pts.to(hip_device)
    model.bbox_head.jitter_noise = tuple(t.to(hip_device) for t in _small.fixed_noise(3, 32))
    T = max(b.shape[0] for b in boxes)
    g = torch.Generator().manual_seed(11)
    quality = torch.zeros(3, max(T, 1), 6)
    for i, b in enumerate(boxes):
        quality[i, :b.shape[0]] = torch.rand(b.shape[0], 6, generator=g)
    quality = quality.to(hip_device)
    want = _run_unsup(model, pts, boxes, labels, quality, False)
    got = _run_unsup(model, pts, boxes, labels, quality, True)
    assert list(got[0]) == list(want[0]) and len(want[0]) == 4
    for k in want[0]:
        assert want[0][k].abs().item() > 0, k
        torch.testing.assert_close(got[0][k], want[0][k], rtol=2e-5, atol=1e-6, msg=k)
    for k in want[1]:
        scale = want[1][k].abs().max().item()
        torch.testing.assert_close(got[1][k], want[1][k], rtol=1e-4, atol=2e-5 * max(scale, 1e-6), msg=k)
    if kind == 'saqe':      # constant uncertainties: nothing reaches the side scores
        assert float(got[1]['_side_all'].abs().max()) == 0.0 == float(want[1]['_side_all'].abs().max())
    flat_w = torch.cat([want[2][n].flatten() for n in sorted(want[2])]).double()
    flat_g = torch.cat([got[2][n].flatten() for n in sorted(want[2])]).double()
    assert ((flat_g - flat_w).norm() / flat_w.norm()).item() < 1e-4


def _saqe_model(hip_device):
    from nesie_amd.votenet.detector import build_saqe_votenet, saqe_votenet_scannet_cfg
    cfg, scfg = _small.small_cfg(), saqe_votenet_scannet_cfg()
    cfg['bbox_head'].update(angle_loss=scfg['bbox_head']['angle_loss'],
                            angle_pred_loss=scfg['bbox_head']['angle_pred_loss'])
    cfg['head_type'] = 'SAQEHead'
    torch.manual_seed(0)
    model = build_saqe_votenet(cfg).to(hip_device).train()
    assert type(model.bbox_head).__name__ == 'SAQEHead'
    return model


@pytest.mark.parametrize('which', ['loss', 'sup_loss'])
def test_fused_saqe_supervised_losses_match_the_module_by_module_losses(hip_device, which):
    """SAQEHead.loss (saqe_head.py:331-521) and sup_loss (:524-703) through head_loss.hip (shared
    terms without / with constant uncertainties) + the SAQE extras kernel (quality-head objectness,
    heading sin / cos, angle quality, jittered side quality): every term, the gradients at the
    head's outputs and every parameter gradient, under unequal incoming gradients."""
    model = _saqe_model(hip_device)
    model.train_cfg['pos_distance_thr'], model.train_cfg['neg_distance_thr'] = 1.0, 1.5
    model.bbox_head.train_cfg = model.train_cfg
    pts, boxes, labels = _small.small_batch(batch=3)
    for b in boxes:                      # headings that are not zero: the angle terms are alive
        b[:, 6] = torch.linspace(-1.2, 1.4, b.shape[0])
    pts = pts.to(hip_device)
    gt = GTBatch.collate(boxes, labels, hip_device)
    model.bbox_head.jitter_noise = tuple(t.to(hip_device) for t in _small.fixed_noise(3, 32))
    weights = dict(vote_loss=1.0, objectness_loss=0.7, semantic_loss=1.3, center_loss=0.9, surface_loss=1.1,
                   iou_loss=0.8, iou_pred_loss=1.2, side_loss=0.6, angle_loss=1.4, angle_pred_loss=0.5)
    keys = ('_cls_all', 'bbox_preds', 'surface_pred', '_side_all', '_iou_all', '_rot_all', '_robj_all')

    def run(fused):
        head_loss.ENABLED = fused
        try:
            for p in model.parameters():
                p.grad = None
            preds = model.bbox_head(model.extract_feat(pts), 'vote')
            keep = {}
            for k in keys:
                preds[k].retain_grad()
                keep[k] = preds[k]
            assert head_loss.saqe_usable(model.bbox_head, preds) == fused
            losses = getattr(model.bbox_head, which)(preds, pts, gt, None)
            sum(losses[k] * weights[k] for k in losses).backward()
            return ({k: v.detach().clone() for k, v in losses.items()},
                    {k: (v.grad.detach().clone() if v.grad is not None else torch.zeros_like(v))
                     for k, v in keep.items()}, _small.grads_of(model))
        finally:
            head_loss.ENABLED = True
    want, got = run(False), run(True)
    assert list(got[0]) == list(want[0])
    assert ('angle_pred_loss' in want[0]) == (which == 'loss')
    for k in want[0]:
        assert want[0][k].abs().item() > 0, k
        torch.testing.assert_close(got[0][k], want[0][k], rtol=2e-5, atol=1e-6, msg=k)
    for k in want[1]:
        scale = want[1][k].abs().max().item()
        torch.testing.assert_close(got[1][k], want[1][k], rtol=1e-4, atol=2e-5 * max(scale, 1e-6), msg=k)
    flat_w = torch.cat([want[2][n].flatten() for n in sorted(want[2])]).double()
    flat_g = torch.cat([got[2][n].flatten() for n in sorted(want[2])]).double()
    assert ((flat_g - flat_w).norm() / flat_w.norm()).item() < 1e-4


def test_targets_kernel_matches_the_tensor_ops(hip_device):
    """nesie_head_targets vs get_targets' tensor-op form: indices exact, weights bit-equal."""
    model = _small.small_model().to(hip_device)
    model.train_cfg['pos_distance_thr'], model.train_cfg['neg_distance_thr'] = 0.6, 1.0
    model.bbox_head.train_cfg = model.train_cfg
    pts, boxes, labels = _small.small_batch(batch=3)
    boxes[1] = boxes[1][:0]          # an empty scene: the reference's all-zero fake box
    labels[1] = labels[1][:0]
    pts = pts.to(hip_device)
    gt = GTBatch.collate(boxes, labels, hip_device)
    with torch.no_grad():
        preds = model.bbox_head(model.extract_feat(pts), 'vote')
        outs = []
        for fused in (False, True):
            head_loss.ENABLED = fused
            try:
                outs.append(model.bbox_head.get_targets(pts, gt, None, bbox_preds=preds))
            finally:
                head_loss.ENABLED = True
    names = ['vote_targets', 'vote_target_masks', 'center_targets', 'bbox_targets', 'mask_targets',
             'valid_gt_masks', 'objectness_targets', 'objectness_weights', 'box_loss_weights',
             'valid_gt_weights', 'assignment']
    assert outs[0][6].sum() > 0
    for n, a, b in zip(names, outs[0], outs[1]):
        assert a.dtype == b.dtype and a.shape == b.shape, n
        assert torch.equal(a, b), n


def test_vote_targets_kernel_on_nested_rotated_boxes(hip_device):
    """nesie_vote_targets vs the tensor-op form where points sit in up to five boxes (first /
    second / LAST slots differ), boxes are rotated, one scene has no box and one has more boxes
    than the kernel's 128-box tile."""
    from nesie_amd.votenet.nesie_head import NesieHead
    g = torch.Generator().manual_seed(5)
    B, N = 4, 6000
    pts = torch.cat([torch.rand(B, N, 3, generator=g) * torch.tensor([6.0, 6.0, 2.5]),
                     torch.rand(B, N, 1, generator=g)], -1)
    boxes, labels = [], []
    for b, t in enumerate([40, 0, 150, 7]):
        centre = torch.rand(t, 3, generator=g) * torch.tensor([6.0, 6.0, 1.0])
        size = 0.3 + 2.5 * torch.rand(t, 3, generator=g)       # large: heavy nesting
        yaw = (torch.rand(t, 1, generator=g) - 0.5) * 6.0
        boxes.append(torch.cat([centre, size, yaw], -1))
        labels.append(torch.randint(0, 18, (t,), generator=g))
    pts = pts.to(hip_device)
    gt = GTBatch.collate(boxes, labels, hip_device)
    outs = []
    for fused in (False, True):
        head_loss.ENABLED = fused
        try:
            outs.append(NesieHead.vote_targets_of(pts, gt))
        finally:
            head_loss.ENABLED = True
    (v0, m0), (v1, m1) = outs
    assert v0.dtype == v1.dtype and m0.dtype == m1.dtype and v0.shape == v1.shape == (B, N, 9)
    assert torch.equal(m0, m1) and torch.equal(v0, v1)
    # the case is not vacuous: all three slots differ somewhere, and some points are in no box
    assert ((v0[..., 0:3] != v0[..., 3:6]).any(-1) & (v0[..., 3:6] != v0[..., 6:9]).any(-1)).any()
    assert (m0 == 0).any() and (m0[1] == 0).all()


@pytest.mark.parametrize('bins,copies', [(17, 2), (33, 1)])
def test_side_prob_stats_match_topk_and_var(hip_device, bins, copies):
    """nesie_side_prob_stats vs SidePooling.dist_feature's tensor form (cat[prob, topk 4, var])."""
    from nesie_amd import kernels
    g = torch.Generator(device=hip_device).manual_seed(bins)
    prob = torch.softmax(torch.randn(3, 6, bins, 200, device=hip_device, generator=g) * 2, dim=2)
    prob[0, 1, :, 5] = 1.0 / bins                      # ties
    want = torch.cat([prob, prob.topk(4, dim=2)[0], prob.var(dim=2, keepdim=True)], dim=2)
    want = want.permute(1, 0, 2, 3).repeat(1, 1, 1, copies)
    got = kernels.backend_for(prob).side_prob_stats(prob, copies)
    assert got.shape == want.shape
    assert torch.equal(got[:, :, :bins + 4], want[:, :, :bins + 4])
    torch.testing.assert_close(got[:, :, bins + 4], want[:, :, bins + 4], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize('sigma,size_bias,zero_heading', [(0.3, 0.0, True), (0.5, 0.2, False)])
def test_proposal_jitter_matches_the_tensor_ops(hip_device, sigma, size_bias, zero_heading):
    """nesie_proposal_jitter vs the element-wise form of jitter_bbox_preds (Nesie and SAQE)."""
    from nesie_amd import kernels
    g = torch.Generator(device=hip_device).manual_seed(9)
    bp = torch.randn(3, 50, 7, device=hip_device, generator=g)
    bp[..., 3:6] = bp[..., 3:6].abs() + 0.05
    n_c = torch.randn(3, 50, 3, device=hip_device, generator=g)
    n_s = torch.randn(3, 50, 3, device=hip_device, generator=g)
    center, size, heading = bp[..., :3], bp[..., 3:6], bp[..., -1]
    cj = center + size * n_c * sigma
    sj = torch.clamp(size + size * n_s * sigma if size_bias == 0 else
                     size + size * (n_s * sigma + size_bias), min=1e-8)
    heading_all = torch.cat([heading, heading], 1)
    want = (torch.cat([center, cj], 1), torch.cat([size, sj], 1),
            torch.zeros_like(heading_all) if zero_heading else heading_all,
            torch.cat([cj, sj, heading.unsqueeze(-1)], -1))
    got = kernels.backend_for(bp).proposal_jitter(bp, n_c, n_s, sigma, size_bias, zero_heading)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
