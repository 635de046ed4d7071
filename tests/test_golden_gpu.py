"""The HIP head (fused decode / targets / loss kernels, layer kernels, quality head) DIRECTLY against
the golden outputs of the reference's own Python (tests/golden/nesie_head_golden.pt, produced by
tests/golden/make_golden.py from nesie_head.py / saqe_head.py / side_pooling_module.py /
quelity_estimation_module.py / the loss files loaded by path).  tests/test_golden.py checks the
same goldens through the CPU oracle; here no CPU leg sits in between: seeded inputs and weights
go to the device, the product path runs through the C ABI, the outputs meet the reference's.

The one discrete decision inside the head -- furthest-point sampling of the PREDICTED votes --
is replayed from the golden's own picks (oracle/forcing.py: a last-bit difference in a vote
coordinate may legitimately flip a pick); whether the HIP path's own picks agreed is asserted
too, since at this size they do."""
import os

import pytest
import torch

from nesie_amd.votenet.boxes import DepthInstance3DBoxes
from oracle.forcing import ForcedSampler, force_vote_sampling
from tests.golden import golden_inputs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "nesie_head_golden.pt")


@pytest.fixture(scope="module")
def gold():
    return torch.load(GOLD)


def _on(device, feat, points, boxes, labels):
    feat = {k: [t.to(device) for t in v] for k, v in feat.items()}
    return (feat, points.to(device), [DepthInstance3DBoxes(b.to(device)) for b in boxes],
            [l.clone().to(device) for l in labels])


def _replaying(head, key, picks):
    """The head's vote sampler replays ``picks`` on its first call (and reports agreement)."""
    class _M:   # force_vote_sampling wants model.bbox_head
        bbox_head = head
    sampler = force_vote_sampling(_M, key)
    ForcedSampler.book[key] = [picks.cpu()]
    return sampler


@pytest.fixture(scope="module")
def hip_head_run(gold, hip_device):
    head = golden_inputs.build_my_head().to(hip_device)
    feat, points, boxes, labels = _on(hip_device, *golden_inputs.head_inputs())
    head.jitter_noise = tuple(t.to(hip_device) for t in golden_inputs.jitter_noise())
    sampler = _replaying(head, 'golden-gpu-nesie', gold["head/pred/aggregated_indices"])
    preds = head(feat, "vote")
    losses = head.loss(preds, points, boxes, labels)
    targets = head.get_targets(points, boxes, labels, bbox_preds=preds)
    q = [t.to(hip_device) for t in golden_inputs.pseudo_quality(golden_inputs.head_inputs()[2])]
    unsup = head.unsup_loss(preds, points, boxes, labels, None, q)
    torch.cuda.synchronize()
    return preds, losses, targets, unsup, sampler


def test_hip_head_forward_meets_the_reference_goldens(gold, hip_head_run):
    preds, _, _, _, sampler = hip_head_run
    assert sampler.agreed == [True], 'the HIP path picked other votes than the reference run'
    for key in ["vote_points", "aggregated_points", "obj_scores", "sem_scores", "surface_pred",
                "bbox_preds", "jitter_bbox_preds", "iou_scores", "iou_scores_jitter",
                "side_scores", "side_scores_jitter"]:
        torch.testing.assert_close(preds[key].cpu(), gold[f"head/pred/{key}"], rtol=1e-4, atol=1e-5,
                                   msg=key)
    assert torch.equal(preds["aggregated_indices"].cpu(), gold["head/pred/aggregated_indices"])


def test_hip_head_targets_equal_the_reference_loops(gold, hip_head_run):
    """head_targets_kernel / vote_targets_kernel against the targets the reference's python loops
    (nesie_head.py:511-679) produced: integer targets and masks exact, box targets exact,
    batch-level weights to the last bit or 1e-6."""
    (vt, vm, ct, bt, mt, vg, ot, ow, bw, vgw, asg) = [t.cpu() for t in hip_head_run[2]]
    g = lambda k: gold[f"head/target/{k}"]  # noqa: E731
    assert torch.equal(asg.long(), g("assignment").long())
    assert torch.equal(ot.long(), g("objectness_targets").long())
    assert torch.equal(mt.long(), g("mask_targets").long())
    assert torch.equal(vg.long(), g("valid_gt_masks").long())
    torch.testing.assert_close(ct, g("center_targets"), rtol=0, atol=0)
    torch.testing.assert_close(bt.reshape(-1, 7), g("bbox_targets"), rtol=0, atol=0)
    torch.testing.assert_close(ow, g("objectness_weights"), rtol=1e-6, atol=0)
    torch.testing.assert_close(bw, g("box_loss_weights"), rtol=1e-6, atol=0)
    torch.testing.assert_close(vgw, g("valid_gt_weights"), rtol=1e-6, atol=0)
    assert vm.sum() == g("vote_target_masks_sum")
    torch.testing.assert_close(vt[:, ::16], g("vote_targets_rows"), rtol=0, atol=0)
    assert abs(vt.double().sum() - g("vote_targets_sum")) < 1e-6
    assert abs(vt.double().abs().sum() - g("vote_targets_abs_sum")) < 1e-6


def test_hip_head_losses_meet_the_reference_goldens(gold, hip_head_run):
    """The eight loss terms of NesieHead.loss from the fused kernels (head_loss.hip) vs the
    reference's loss dict: 1e-4."""
    _, losses, _, unsup, _ = hip_head_run
    assert set(losses) == {k.split("/")[-1] for k in gold if k.startswith("head/loss/")}
    assert len(losses) == 8
    for k, v in losses.items():
        want = gold[f"head/loss/{k}"]
        assert want > 0, k
        torch.testing.assert_close(v.detach().cpu(), want, rtol=1e-4, atol=1e-6, msg=k)
    assert set(unsup) == {"unsup_semantic_loss", "unsup_center_loss", "unsup_iou_loss",
                          "unsup_surface_loss"}
    for k, v in unsup.items():
        torch.testing.assert_close(v.detach().cpu(), gold[f"head/unsup/{k}"], rtol=1e-4, atol=1e-6,
                                   msg=k)


@pytest.fixture(scope="module")
def hip_saqe_run(gold, hip_device):
    head = golden_inputs.build_my_saqe_head().to(hip_device)
    raw = golden_inputs.head_inputs()
    head.jitter_noise = tuple(t.to(hip_device) for t in golden_inputs.jitter_noise())
    sampler = _replaying(head, 'golden-gpu-saqe', gold["saqe/pred/aggregated_indices"]) \
        if "saqe/pred/aggregated_indices" in gold else None
    feat, points, boxes, labels = _on(hip_device, *raw)
    preds = head(feat, "vote")
    mk = lambda: _on(hip_device, *raw)[1:]  # noqa: E731
    q = [t.to(hip_device) for t in golden_inputs.pseudo_quality(raw[2])]
    out = (preds, head.loss(preds, *mk()), head.sup_loss(preds, *mk()),
           head.unsup_loss(preds, *mk(), None, q), sampler)
    torch.cuda.synchronize()
    return out


def test_hip_saqe_forward_meets_the_reference_goldens(gold, hip_saqe_run):
    preds = hip_saqe_run[0]
    keys = [k.split("/")[-1] for k in gold if k.startswith("saqe/pred/")]
    assert len(keys) == 13
    for key in keys:
        got, want = preds[key].cpu(), gold[f"saqe/pred/{key}"]
        if want.dtype.is_floating_point:
            torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5, msg=key)
        else:
            assert torch.equal(got.long(), want.long()), key


@pytest.mark.parametrize("which,idx", [("loss", 1), ("sup_loss", 2), ("unsup", 3)])
def test_hip_saqe_losses_meet_the_reference_goldens(gold, hip_saqe_run, which, idx):
    """SAQEHead.loss / sup_loss / unsup_loss (saqe_head.py:331-521, 524-703, 706-800) on the HIP
    path vs the reference's dicts: 1e-4."""
    got = hip_saqe_run[idx]
    want = {k.split("/")[-1]: v for k, v in gold.items() if k.startswith(f"saqe/{which}/")}
    assert set(got) == set(want), (sorted(got), sorted(want))
    for k, v in got.items():
        torch.testing.assert_close(v.detach().cpu(), want[k], rtol=1e-4, atol=1e-6, msg=f"{which}/{k}")
