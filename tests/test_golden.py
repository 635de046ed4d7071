"""Our detector-head path against golden OUTPUTS of the reference's own Python
(tests/golden/nesie_head_golden.pt, produced by tests/golden/make_golden.py from
nesie_head.py / side_pooling_module.py / the loss files / oriented_iou_loss.py loaded by
path).  Inputs and weights are regenerated from seeds.  No GPU, no reference tree."""
import os

import pytest
import torch

from nesie_amd import kernels
from nesie_amd.mmdet3d_ops import cal_iou_3d
from nesie_amd.votenet import losses as L
from nesie_amd.votenet.boxes import DepthInstance3DBoxes
from tests.golden import golden_inputs

GOLD = os.path.join(os.path.dirname(__file__), "golden", "nesie_head_golden.pt")


@pytest.fixture(scope="module")
def gold():
    return torch.load(GOLD)


@pytest.mark.parametrize("mode", golden_inputs.IOU_MODES)
def test_rotated_iou_chain(gold, oracle_kernels, mode):
    a, b = golden_inputs.iou_boxes(mode)
    with kernels.use_backend(oracle_kernels):
        got = cal_iou_3d(a, b)
    torch.testing.assert_close(got, gold[f"iou3d/{mode}"], rtol=1e-5, atol=1e-6)


def test_quality_focal_loss_dense_rewrite(gold):
    pred, label, score, w = golden_inputs.qfl_inputs()
    got = L.quality_focal_loss(pred, (label, score), beta=2.0, use_sigmoid=False) * w
    torch.testing.assert_close(got, gold["qfl/none"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("mode", ["l1", "l2", "smooth_l1"])
def test_chamfer_distance(gold, mode):
    s, d = golden_inputs.chamfer_inputs()
    got = L.chamfer_distance(s, d, criterion_mode=mode, reduction="none")
    for g, w in zip(got, gold[f"chamfer/{mode}"]):
        if g.dtype.is_floating_point:
            torch.testing.assert_close(g, w, rtol=1e-6, atol=1e-7)
        else:
            assert torch.equal(g, w)


@pytest.fixture(scope="module")
def head_run(oracle_kernels):
    head = golden_inputs.build_my_head()
    feat, points, boxes, labels = golden_inputs.head_inputs()
    head.jitter_noise = golden_inputs.jitter_noise()
    with kernels.use_backend(oracle_kernels):
        preds = head(feat, "vote")
        losses = head.loss(preds, points, [DepthInstance3DBoxes(b) for b in boxes],
                           [l.clone() for l in labels])
        targets = head.get_targets(points, [DepthInstance3DBoxes(b) for b in boxes],
                                   [l.clone() for l in labels], bbox_preds=preds)
    return preds, losses, targets


def test_head_forward_matches_reference_code(gold, head_run):
    preds, _, _ = head_run
    for key in ["vote_points", "aggregated_points", "obj_scores", "sem_scores", "surface_pred",
                "bbox_preds", "jitter_bbox_preds", "iou_scores", "iou_scores_jitter",
                "side_scores", "side_scores_jitter"]:
        torch.testing.assert_close(preds[key], gold[f"head/pred/{key}"], rtol=1e-4, atol=1e-5,
                                   msg=key)
    assert torch.equal(preds["aggregated_indices"], gold["head/pred/aggregated_indices"])


def test_head_targets_match_reference_loops(gold, head_run):
    _, _, t = head_run
    (vt, vm, ct, bt, mt, vg, ot, ow, bw, vgw, asg) = t
    g = lambda k: gold[f"head/target/{k}"]  # noqa: E731
    assert torch.equal(asg, g("assignment")) and torch.equal(ot, g("objectness_targets"))
    assert torch.equal(mt, g("mask_targets")) and torch.equal(vg.long(), g("valid_gt_masks").long())
    torch.testing.assert_close(ct, g("center_targets"), rtol=0, atol=0)
    torch.testing.assert_close(bt.reshape(-1, 7), g("bbox_targets"), rtol=0, atol=0)
    torch.testing.assert_close(ow, g("objectness_weights"), rtol=1e-6, atol=0)
    torch.testing.assert_close(bw, g("box_loss_weights"), rtol=1e-6, atol=0)
    torch.testing.assert_close(vgw, g("valid_gt_weights"), rtol=1e-6, atol=0)
    assert vm.sum() == g("vote_target_masks_sum")
    torch.testing.assert_close(vt[:, ::16], g("vote_targets_rows"), rtol=0, atol=0)
    assert abs(vt.double().sum() - g("vote_targets_sum")) < 1e-6
    assert abs(vt.double().abs().sum() - g("vote_targets_abs_sum")) < 1e-6


def test_head_losses_match_reference_code(gold, head_run):
    _, losses, _ = head_run
    assert set(losses) == {k.split("/")[-1] for k in gold if k.startswith("head/loss/")}
    for k, v in losses.items():
        want = gold[f"head/loss/{k}"]
        assert want > 0, k
        torch.testing.assert_close(v.detach(), want, rtol=1e-4, atol=1e-6, msg=k)


# ---- semi-supervised path (votenet_nesie.py + the reference's own DepthInstance3DBoxes) ---------
def test_lhs_nms_matches_reference_numpy(gold, oracle_kernels):
    boxes = golden_inputs.nms_boxes().contiguous()
    keep = torch.zeros(3, 64, dtype=torch.uint8)
    oracle_kernels.lhs_nms_samecls(boxes, 0.25, keep)
    want = gold["semi/nms_keep"]
    # np.argsort leaves the order of the equal (zero) scores unspecified, so only the boxes
    # with a unique score are comparable one by one; the tail must agree in count
    scores = boxes[..., 6]
    for i in range(3):
        uniq = torch.tensor([(scores[i] == s).sum() == 1 for s in scores[i]])
        assert torch.equal(keep[i][uniq], want[i][uniq]), i
    assert abs(int(keep.sum()) - int(want.sum())) <= 6


@pytest.fixture(scope="module")
def pseudo(oracle_kernels):
    from nesie_amd.votenet import semi
    cfg = golden_inputs.head_cfg()
    det = semi.VoteNetNesie.__new__(semi.VoteNetNesie)
    torch.nn.Module.__init__(det)
    det.train_cfg = dict(cfg["train_cfg"], thresh_warmup=True, use_cbl=True)
    ulb_list, ulb_flag, n_lb, n_ulb = golden_inputs.ulb_statistics()
    det.state = semi.PseudoLabelState(n_lb, n_ulb, 18, "cpu")
    det.state.ulb_list, det.state.ulb_flag = ulb_list.clone(), ulb_flag.clone()
    preds = {k: v.clone() for k, v in golden_inputs.teacher_preds().items()}
    with kernels.use_backend(oracle_kernels):
        return det.get_pseudo_labels(preds, "ScanNet")


def test_pseudo_labels_match_reference(gold, pseudo):
    labels, boxes, quality, valid = pseudo
    counts = gold["semi/pl_counts"]
    assert counts.sum() > 10
    assert torch.equal(valid.sum(1), counts)
    for i in range(3):
        n = int(counts[i])
        assert valid[i, :n].all() and not valid[i, n:].any()
        assert torch.equal(labels[i, :n], gold[f"semi/pl_labels/{i}"].long())
        torch.testing.assert_close(boxes[i, :n], gold[f"semi/pl_boxes/{i}"], rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(quality[i, :n], gold[f"semi/pl_quality/{i}"], rtol=1e-6, atol=1e-6)


def test_teacher_to_student_box_reaugmentation(gold, pseudo):
    from nesie_amd.votenet import semi
    _, boxes, _, valid = pseudo
    mt, ms = golden_inputs.aug_metas()
    meta_t, meta_s = semi.AugMeta(**mt), semi.AugMeta(**ms)
    moved = semi.transform_boxes(semi.untransform_boxes(boxes, meta_t), meta_s)
    for i in range(3):
        n = int(valid[i].sum())
        want = gold[f"semi/pl_boxes_student/{i}"]
        torch.testing.assert_close(moved[i, :n, :6], want[:, :6], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(torch.sin(moved[i, :n, 6]), torch.sin(want[:, 6]), rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(torch.cos(moved[i, :n, 6]), torch.cos(want[:, 6]), rtol=1e-4, atol=1e-5)


# ---- Nesie unsup_loss and the SAQE head (saqe_head.py + quelity_estimation_module.py) ----------
def test_nesie_unsup_loss_matches_reference(gold, oracle_kernels):
    head = golden_inputs.build_my_head()
    feat, points, boxes, labels = golden_inputs.head_inputs()
    head.jitter_noise = golden_inputs.jitter_noise()
    q = golden_inputs.pseudo_quality(boxes)
    with kernels.use_backend(oracle_kernels):
        preds = head(feat, "vote")
        got = head.unsup_loss(preds, points, [DepthInstance3DBoxes(b) for b in boxes],
                              [l.clone() for l in labels], None, q)
    assert set(got) == {"unsup_semantic_loss", "unsup_center_loss", "unsup_iou_loss",
                        "unsup_surface_loss"}
    for k, v in got.items():
        torch.testing.assert_close(v.detach(), gold[f"head/unsup/{k}"], rtol=1e-4, atol=1e-6, msg=k)


@pytest.fixture(scope="module")
def saqe_run(oracle_kernels):
    head = golden_inputs.build_my_saqe_head()
    feat, points, boxes, labels = golden_inputs.head_inputs()
    head.jitter_noise = golden_inputs.jitter_noise()
    mk = lambda: (points, [DepthInstance3DBoxes(b) for b in boxes], [l.clone() for l in labels])  # noqa: E731
    with kernels.use_backend(oracle_kernels):
        preds = head(feat, "vote")
        return (preds, head.loss(preds, *mk()), head.sup_loss(preds, *mk()),
                head.unsup_loss(preds, *mk(), None, golden_inputs.pseudo_quality(boxes)))


def test_saqe_forward_matches_reference_code(gold, saqe_run):
    preds = saqe_run[0]
    keys = [k.split("/")[-1] for k in gold if k.startswith("saqe/pred/")]
    assert len(keys) == 13
    for key in keys:
        torch.testing.assert_close(preds[key], gold[f"saqe/pred/{key}"], rtol=1e-4, atol=1e-5,
                                   msg=key)


@pytest.mark.parametrize("which,idx", [("loss", 1), ("sup_loss", 2), ("unsup", 3)])
def test_saqe_losses_match_reference_code(gold, saqe_run, which, idx):
    got = saqe_run[idx]
    want = {k.split("/")[-1]: v for k, v in gold.items() if k.startswith(f"saqe/{which}/")}
    assert set(got) == set(want), (sorted(got), sorted(want))
    for k, v in got.items():
        torch.testing.assert_close(v.detach(), want[k], rtol=1e-4, atol=1e-6, msg=f"{which}/{k}")
