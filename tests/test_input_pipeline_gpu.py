"""Input side on the MI355X: nesie_scene_assemble through the C ABI against
the oracle (bit-exact on identical operands) and the resident-scene pipeline against the
reference's golden outputs."""
import os

import numpy as np
import pytest
import torch

from nesie_amd import kernels
from nesie_amd.input_pipeline import ResidentScenes
from tests.golden import golden_inputs
from tests.test_input_pipeline_cpu import CASES, build_case, check_case

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "input_golden.pt")


@pytest.mark.parametrize("b,n,rows", [(1, 1, 1), (3, 1000, 777), (8, 40000, 400000)])
def test_scene_assemble_bit_exact_vs_oracle(oracle_kernels, hip_device, b, n, rows):
    g = torch.Generator().manual_seed(n)
    pool = torch.randn(rows, 3, generator=g) * 3
    height = torch.rand(rows, generator=g)
    choices = torch.randint(0, rows, (b, n), generator=g, dtype=torch.int32)
    choices[0, 0] = rows + 5          # out-of-range rows are clamped, not faulted
    ang = torch.rand(b, generator=g) - 0.5
    rot = torch.linalg.qr(torch.randn(b, 3, 3, generator=g))[0]
    xform = torch.cat([rot.reshape(b, 9), torch.randn(b, 3, generator=g),
                       torch.where(torch.rand(b, 2, generator=g) < 0.5, -1.0, 1.0),
                       torch.cos(ang)[:, None], torch.sin(ang)[:, None],
                       0.8 + 0.4 * torch.rand(b, 1, generator=g), torch.randn(b, 3, generator=g)], 1)
    want = torch.empty(b, n, 4)
    oracle_kernels.scene_assemble(pool, height, choices, xform.contiguous(), want)
    got = torch.empty(b, n, 4, device=hip_device)
    kernels.backend_for(got).scene_assemble(pool.to(hip_device), height.to(hip_device),
                                            choices.to(hip_device), xform.to(hip_device).contiguous(), got)
    assert torch.equal(got.cpu(), want)


@pytest.mark.parametrize("name", list(CASES))
def test_resident_pipeline_matches_reference_on_gpu(hip_device, name):
    gold = torch.load(GOLD)
    scenes, draws, labels = build_case(name, hip_device)
    scenes.finalize()
    pts, boxes, lab = scenes.assemble([0], [draws])
    assert pts.is_cuda and boxes[0].is_cuda
    check_case(gold, name, pts[0], boxes[0], draws)


def test_device_drawn_batches_feed_the_training_step(hip_device):
    """A resident synthetic data set -> device-drawn batches -> one forward+backward of the
    small detector; no host round trip between assembly and the step."""
    from tests import _small
    from nesie_amd.votenet.nesie_head import GTBatch
    scenes = ResidentScenes(hip_device)
    for seed in range(4):
        raw6, align, gt, labels = golden_inputs.raw_scene(50 + seed, 6000, False)
        scenes.add_scene(raw6[:, :3], align, gt, labels)
    scenes.finalize()
    g = torch.Generator(device=hip_device).manual_seed(1)
    pts, boxes, labels = scenes.assemble([2, 0], num_points=4096, generator=g)
    pts2, _, _ = scenes.assemble([2, 0], num_points=4096, generator=g)
    assert pts.shape == (2, 4096, 4) and not torch.equal(pts, pts2)
    model = _small.small_model().to(hip_device)
    losses = model.forward_train(pts, None, GTBatch.collate(boxes, labels, hip_device), None)
    total = model.parse_losses(losses)
    total.backward()
    assert torch.isfinite(total)


def test_host_staged_draws_on_the_gpu_equal_the_draw_objects(hip_device):
    """``stage_draws`` (decisions drawn and applied on the host, one pinned buffer, one copy) against
    ``assemble_batch(draws=...)`` with the same generator state on the device: the same rows of the
    pool, points to float rounding of the transform (host cos / sin vs the device's), boxes too."""
    from tests.test_input_pipeline_cpu import draw_like_reference
    ranges = dict(rot_range=(-0.087266, 0.087266), scale_range=(0.9, 1.1), translation_std=(0.1, 0.1, 0.05))
    scenes = ResidentScenes(hip_device)
    for seed, n in [(1, 5000), (2, 900), (3, 2500)]:
        raw6, align, gt, labels = golden_inputs.raw_scene(seed, n, False)
        scenes.add_scene(raw6[:, :3], align, gt, labels)
    scenes.finalize()
    ids, n_pts = [2, 0, 1], 2048
    rng = np.random.RandomState(5)
    draws = [draw_like_reference(rng, int(scenes.counts[s]), n_pts, **ranges) for s in ids]
    want_p, want_g = scenes.assemble_batch(ids, draws)
    staging = scenes.new_staging(len(ids), n_pts)
    assert staging['host'][0].is_pinned()
    scenes.stage_draws(staging, 0, ids, np.random.RandomState(5), **ranges)
    staging['dev'].copy_(staging['host'][0], non_blocking=True)
    got_p, got_g = scenes.assemble_batch(torch.tensor(ids, device=hip_device), staging=staging)
    torch.testing.assert_close(got_p, want_p, rtol=0, atol=2e-6)
    torch.testing.assert_close(got_g.boxes, want_g.boxes, rtol=0, atol=2e-6)
    assert torch.equal(got_g.labels, want_g.labels) and torch.equal(got_g.valid, want_g.valid)
    assert torch.equal(got_g.count, want_g.count)
