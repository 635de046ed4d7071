"""Detector path: HIP product path on the GPU vs the same model through the CPU oracle."""
import copy

import pytest
import torch

from nesie_amd import kernels
from tests import _small

pytestmark = pytest.mark.gpu


def test_small_model_losses_and_grads_match_cpu_oracle(oracle_kernels, hip_device):
    model = _small.small_model()
    pts, boxes, labels = _small.small_batch()
    model.bbox_head.jitter_noise = _small.fixed_noise(2, 32)
    with kernels.use_backend(oracle_kernels):
        want_l, want_g = _small.train_step_losses(model, pts, boxes, labels)
    gmodel = copy.deepcopy(model).to(hip_device)
    got_l, got_g = _small.train_step_losses(gmodel, pts.to(hip_device), boxes, labels)
    for k in want_l:  # north_star tolerance: 1e-4 for fp32 losses
        torch.testing.assert_close(got_l[k], want_l[k], rtol=1e-4, atol=1e-5, msg=k)
    assert set(got_g) == set(want_g)
    worst = 0.0
    for n in want_g:
        denom = want_g[n].abs().max().item() + 1e-8
        worst = max(worst, (got_g[n] - want_g[n]).abs().max().item() / denom)
    assert worst < 2e-3, worst  # max-normalised gradient error over all parameters


def test_full_config_step_runs_and_is_finite(hip_device):
    from nesie_amd.scenes import make_batch
    from nesie_amd.votenet import build_nesie_votenet
    from nesie_amd.votenet.nesie_head import GTBatch
    torch.manual_seed(0)
    model = build_nesie_votenet().to(hip_device)
    pts, boxes, labels = make_batch(1000, 2)
    gt = GTBatch.collate(boxes, labels, hip_device)
    losses = model.forward_train(pts.to(hip_device), None, gt, None)
    total = model.parse_losses(losses)
    total.backward()
    assert torch.isfinite(total)
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), n
