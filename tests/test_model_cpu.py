"""Detector path on the CPU oracle back end: structure, targets, finite fwd+bwd."""
import pytest
import torch

from nesie_amd import kernels
from nesie_amd.votenet import build_nesie_votenet
from nesie_amd.votenet.boxes import DepthInstance3DBoxes
from nesie_amd.votenet.losses import chamfer_distance
from nesie_amd.votenet.nesie_head import GTBatch
from tests import _small


def test_parameter_count_matches_reference_model():
    # 2 640 477 fp32 parameters = the 10.56 MB gradient message
    model = build_nesie_votenet()
    assert sum(p.numel() for p in model.parameters()) == 2640477
    names = dict(model.named_parameters())
    # checkpoint key names line up with the reference's
    for key in ['backbone.SA_modules.0.mlps.0.layer0.conv.weight',
                'backbone.SA_modules.0.mlps.0.layer0.bn.weight',
                'backbone.FP_modules.1.mlps.layer1.conv.weight',
                'bbox_head.vote_module.vote_conv.0.conv.bias',
                'bbox_head.vote_module.conv_out.weight',
                'bbox_head.vote_aggregation.mlps.0.layer2.bn.bias',
                'bbox_head.conv_pred.shared_convs.layer0.conv.bias',
                'bbox_head.conv_pred.conv_bbox.weight',
                'bbox_head.grid_conv.mlps_before.3.first_conv.0.weight',
                'bbox_head.grid_conv.mlps_head.6.6.bias']:
        assert key in names, key
    assert 'backbone.SA_modules.0.mlps.0.layer0.conv.bias' not in names  # bias='auto'


def _targets_reference_loop(head, points, gt_boxes, gt_labels, aggregated_points, ok):
    """Literal restatement of get_targets_single (nesie_head.py:593-679) for ONE scene."""
    gt = DepthInstance3DBoxes(gt_boxes)
    num_points = points.shape[0]
    vote_targets = points.new_zeros([num_points, 9])
    vote_target_masks = points.new_zeros([num_points], dtype=torch.long)
    vote_target_idx = points.new_zeros([num_points], dtype=torch.long)
    with kernels.use_backend(ok):
        box_indices_all = gt.points_in_boxes(points)
    for i in range(gt_labels.shape[0]):
        indices = torch.nonzero(box_indices_all[:, i], as_tuple=False).squeeze(-1)
        selected = points[indices]
        vote_target_masks[indices] = 1
        tmp = vote_targets[indices]
        votes = gt.gravity_center[i].unsqueeze(0) - selected[:, :3]
        for j in range(3):
            col = torch.nonzero(vote_target_idx[indices] == j, as_tuple=False).squeeze(-1)
            tmp[col, int(j * 3):int(j * 3 + 3)] = votes[col]
            if j == 0:
                tmp[col] = votes[col].repeat(1, 3)
        vote_targets[indices] = tmp
        vote_target_idx[indices] = torch.clamp(vote_target_idx[indices] + 1, max=2)
    center_targets = gt.gravity_center
    distance1, _, assignment, _ = chamfer_distance(
        aggregated_points.unsqueeze(0), center_targets.unsqueeze(0), reduction='none')
    assignment = assignment.squeeze(0)
    dist = torch.sqrt(distance1.squeeze(0) + 1e-6)
    obj_t = (dist < head.train_cfg['pos_distance_thr']).long()
    obj_m = ((dist < head.train_cfg['pos_distance_thr']) |
             (dist > head.train_cfg['neg_distance_thr'])).float()
    bbox_targets = torch.cat((center_targets[assignment], gt.tensor[assignment, 3:]), dim=-1)
    return (vote_targets, vote_target_masks, center_targets, bbox_targets,
            gt_labels[assignment].long(), obj_t, obj_m, assignment)


def test_batched_targets_equal_the_reference_loop(oracle_kernels):
    model = _small.small_model()
    head = model.bbox_head
    pts, boxes, labels = _small.small_batch(batch=3, n=3000)
    # overlapping boxes: points inside 2, 3 and 4 boxes exercise every vote slot rule
    boxes[0] = torch.cat([boxes[0][:1].repeat(4, 1) * torch.tensor([1, 1, 1, 1.0, 1, 1, 1]),
                          boxes[0][4:]], 0)
    boxes[0][1, 3:6] *= 1.3; boxes[0][2, 3:6] *= 0.8; boxes[0][3, :2] += 0.2
    boxes[2] = boxes[2][:0]; labels[2] = labels[2][:0]  # an empty scene -> fake box
    g = torch.Generator().manual_seed(5)
    agg = torch.stack([p[torch.randperm(3000, generator=g)[:32], :3] for p in pts])
    agg[:, :8] = torch.stack([b[:8, :3] if len(b) >= 8 else agg[i, :8]
                              for i, b in enumerate(boxes)])
    with kernels.use_backend(oracle_kernels):
        got = head.get_targets(pts, boxes, labels, bbox_preds=dict(aggregated_points=agg))
    (vt, vm, ct, bt, mt, vg, ot, ow, bw, vgw, asg) = got
    assert (vm.sum(1)[:2] > 0).all()
    n_obj, n_mask, n_valid = 0.0, 0.0, 0.0
    per = []
    for i in range(3):
        if len(labels[i]) == 0:
            b_i, l_i = torch.zeros(1, 7), torch.zeros(1, dtype=torch.long)
        else:
            b_i, l_i = boxes[i], labels[i]
        per.append(_targets_reference_loop(head, pts[i], b_i, l_i, agg[i], oracle_kernels))
        n_obj += per[-1][5].sum().item(); n_mask += per[-1][6].sum().item()
        n_valid += float(len(labels[i]))
    for i in range(3):
        r = per[i]
        torch.testing.assert_close(vt[i], r[0], rtol=0, atol=0)
        assert torch.equal(vm[i], r[1])
        T = r[2].shape[0]
        torch.testing.assert_close(ct[i, :T], r[2], rtol=0, atol=0)
        assert (ct[i, T:] == 0).all()
        torch.testing.assert_close(bt[i], r[3], rtol=0, atol=0)
        assert torch.equal(mt[i], r[4]) and torch.equal(ot[i], r[5]) and torch.equal(asg[i], r[7])
        torch.testing.assert_close(ow[i], r[6] / (n_mask + 1e-6))
        torch.testing.assert_close(bw[i], r[5].float() / (n_obj + 1e-6))
    assert vg[2].sum() == 0 and abs(vgw.sum().item() - 1.0) < 1e-5
    # points in >= 3 boxes exist in scene 0 and use the LAST box for slot 2
    assert ((vt[0, :, 0:3] != vt[0, :, 6:9]).any(1) & (vt[0, :, 3:6] != vt[0, :, 6:9]).any(1)).any()


def test_forward_backward_is_finite_and_deterministic(oracle_kernels):
    model = _small.small_model()
    pts, boxes, labels = _small.small_batch()
    model.bbox_head.jitter_noise = _small.fixed_noise(2, 32)
    with kernels.use_backend(oracle_kernels):
        l1, g1 = _small.train_step_losses(model, pts, boxes, labels)
        l2, g2 = _small.train_step_losses(model, pts, boxes, labels)
    assert set(l1) == {'vote_loss', 'objectness_loss', 'semantic_loss', 'center_loss',
                       'surface_loss', 'iou_loss', 'iou_pred_loss', 'side_loss'}
    for k in l1:
        assert torch.isfinite(l1[k]).all(), k
        torch.testing.assert_close(l1[k], l2[k], rtol=1e-6, atol=1e-7)
    assert len(g1) > 150
    for n in g1:
        assert torch.isfinite(g1[n]).all(), n
    # every parameter that can receive a gradient does (heading conv has none: ScanNet
    # ignores the heading in SidePooling and IoU has no grad through atan2 of a constant 0?)
    missing = [n for n, p in model.named_parameters() if n not in g1]
    assert all('conv_heading' in n for n in missing), missing


def test_first_conv_through_blend_equals_conv_of_grid_features(oracle_kernels):
    """SidePooling.first_conv_through_blend (W_xyz . rel + blend(W_f . F)) vs the reference's
    literal order, first_conv[0](cat[rel, blend(F)]) (side_pooling_module.py:226-243, 304-313,
    346-349): outputs and the gradients of the six conv weights."""
    from nesie_amd.votenet.side_pooling import SidePooling
    torch.manual_seed(5)
    B, K, N, C = 2, 6, 40, 16
    with kernels.use_backend(oracle_kernels):
        sp = SidePooling(num_class=3, num_heading_bin=1, num_size_cluster=3,
                         mean_size_arr_path=None, num_proposal=K, sampling='vote_fps',
                         seed_feat_dim=C)
        xyz = torch.rand(B, N, 3) * 4
        feats_t = torch.randn(B, N, C)
        center = torch.rand(B, K, 3) * 4
        size = torch.rand(B, K, 3) + 0.5
        heading = torch.rand(B, K) * 6.28
        grid = sp.grid_for_side(sp.generate_grid(size), center, heading).view(B, -1, 3).contiguous()
        nets = sp.mlps_before[:6]
        go = [torch.randn(B, 256, K, 16) for _ in range(6)]
        got = sp.first_conv_through_blend(nets, xyz, feats_t, grid, center)[0].unbind(1)
        sum((g * o).sum() for g, o in zip(go, got)).backward()
        g_fused = [n.first_conv[0].weight.grad.clone() for n in nets]
        for n in nets:
            n.first_conv[0].weight.grad = None
        feats = sp.grid_features(xyz, feats_t, grid, center, segs=6)
        want = [n.first_conv[0](feats[:, i]) for i, n in enumerate(nets)]
        sum((g * o).sum() for g, o in zip(go, want)).backward()
    for i in range(6):
        torch.testing.assert_close(got[i], want[i], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(g_fused[i], nets[i].first_conv[0].weight.grad,
                                   rtol=1e-4, atol=1e-4)
