"""Input side on the CPU: the resident-scene batch assembly (oracle back end)
against golden OUTPUTS of the reference's own loading / augmentation classes
(tests/golden/input_golden.pt), including the order of the random draws."""
import os

import numpy as np
import pytest
import torch

from nesie_amd import kernels
from nesie_amd.input_pipeline import ResidentScenes, draw_like_reference, load_points_bin
from tests.golden import golden_inputs

GOLD = os.path.join(os.path.dirname(__file__), "golden", "input_golden.pt")
CASES = {c[0]: c for c in golden_inputs.INPUT_CASES}


@pytest.fixture(scope="module")
def gold():
    return torch.load(GOLD)


def build_case(name, device, tmp_path=None):
    _, seed, n_raw, n_pts, with_yaw, rot, scl, tstd = CASES[name]
    raw6, align, gt, labels = golden_inputs.raw_scene(seed, n_raw, with_yaw)
    if tmp_path is not None:                       # through the .bin loader
        f = os.path.join(tmp_path, name + ".bin")
        raw6.tofile(f)
        xyz = load_points_bin(f)
        assert xyz.shape == (n_raw, 3) and xyz.dtype == np.float32
    else:
        xyz = raw6[:, :3]
    scenes = ResidentScenes(device, with_yaw=with_yaw)
    scenes.add_scene(xyz, align, gt, labels)
    draws = draw_like_reference(np.random.RandomState(1000 + seed), n_raw, n_pts,
                                rot_range=rot, scale_range=scl, translation_std=tstd)
    return scenes, draws, labels


def check_case(gold, name, pts, boxes, draws):
    g = lambda k: gold[f"input/{name}/{k}"]  # noqa: E731
    # the random decisions, in the reference's draw order
    assert torch.equal(torch.from_numpy(draws.choices[:256]), g("choices_head"))
    assert int(draws.choices.sum()) == int(g("choices_sum"))
    assert [draws.flip_h, draws.flip_v] == g("flips").tolist()
    assert draws.angle == float(g("angle_scale")[0]) and draws.scale == float(g("angle_scale")[1])
    np.testing.assert_array_equal(draws.trans, g("trans").numpy())
    # the assembled points and boxes
    pts = pts.cpu()
    torch.testing.assert_close(pts[::64], g("rows"), rtol=1e-6, atol=2e-6)
    torch.testing.assert_close(pts.double().sum(0), g("sum"), rtol=1e-6, atol=1e-2)
    torch.testing.assert_close(pts.double().abs().sum(0), g("abs_sum"), rtol=1e-6, atol=1e-2)
    torch.testing.assert_close(boxes.cpu(), g("boxes"), rtol=1e-6, atol=2e-6)


@pytest.mark.parametrize("name", list(CASES))
def test_assembled_sample_matches_reference_pipeline(gold, oracle_kernels, name, tmp_path):
    with kernels.use_backend(oracle_kernels):
        scenes, draws, labels = build_case(name, "cpu", str(tmp_path))
        scenes.finalize()
        pts, boxes, lab = scenes.assemble([0], [draws])
    assert pts.shape == (1, CASES[name][3], 4)
    check_case(gold, name, pts[0], boxes[0], draws)
    assert torch.equal(lab[0], torch.from_numpy(labels).long())
    flow = gold[f"input/{name}/flow"]
    assert ("HF" in flow) == draws.flip_h and ("VF" in flow) == draws.flip_v


def test_batch_of_scenes_equals_the_single_scene_results(gold, oracle_kernels):
    names = ["scannet_big", "scannet_small"]
    with kernels.use_backend(oracle_kernels):
        scenes = ResidentScenes("cpu")
        draws = []
        for n in names:
            _, seed, n_raw, n_pts, with_yaw, rot, scl, tstd = CASES[n]
            raw6, align, gt, labels = golden_inputs.raw_scene(seed, n_raw, with_yaw)
            scenes.add_scene(raw6[:, :3], align, gt, labels)
            draws.append(draw_like_reference(np.random.RandomState(1000 + seed), n_raw, n_pts))
        scenes.finalize()
        assert len(scenes) == 2 and scenes.pool.shape == (90000, 3)
        assert scenes.offsets.tolist() == [0, 60000]
        pts, boxes, _ = scenes.assemble([1, 0], [draws[1], draws[0]])
    check_case(gold, "scannet_small", pts[0], boxes[0], draws[1])
    check_case(gold, "scannet_big", pts[1], boxes[1], draws[0])


def test_device_draws_have_the_reference_distributions(oracle_kernels):
    with kernels.use_backend(oracle_kernels):
        scenes = ResidentScenes("cpu")
        for seed, n in [(1, 5000), (2, 900), (3, 2500)]:
            raw6, align, gt, labels = golden_inputs.raw_scene(seed, n, False)
            scenes.add_scene(raw6[:, :3], align, gt, labels)
        scenes.finalize()
        g = torch.Generator().manual_seed(0)
        ids = [0, 1, 2, 1]
        choices, xform, dec = scenes.draw_on_device(ids, num_points=2000, generator=g,
                                                    scale_range=(0.9, 1.1),
                                                    translation_std=(0.1, 0.1, 0.0))
        assert choices.shape == (4, 2000) and choices.dtype == torch.int32
        for row, s in zip(choices, ids):
            lo, n = int(scenes.offsets[s]), int(scenes.counts[s])
            assert int(row.min()) >= lo and int(row.max()) < lo + n
            if n >= 2000:
                assert row.unique().numel() == 2000          # without replacement
            else:
                assert row.unique().numel() < 2000           # with replacement
        flip_h, flip_v, angle, scale, trans = dec
        assert (angle.abs() <= 0.087266).all() and ((scale >= 0.9) & (scale <= 1.1)).all()
        assert (trans[:, 2] == 0).all() and trans[:, :2].abs().max() > 0
        assert torch.equal(xform[:, 12] < 0, flip_h) and torch.equal(xform[:, 13] < 0, flip_v)
        pts, boxes, labels = scenes.assemble(ids, num_points=2000, generator=g)
        assert pts.shape == (4, 2000, 4) and torch.isfinite(pts).all()
        assert len(boxes) == 4 and boxes[1].shape[1] == 7


def test_identity_transform_returns_the_sampled_rows(oracle_kernels):
    with kernels.use_backend(oracle_kernels):
        scenes = ResidentScenes("cpu")
        xyz = np.random.RandomState(0).randn(300, 3).astype(np.float32)
        scenes.add_scene(xyz)
        scenes.finalize()
        from nesie_amd.input_pipeline import AugmentDraws
        d = AugmentDraws(np.arange(300)[::-1].copy(), False, False, 0.0, 1.0, np.zeros(3))
        pts, boxes, _ = scenes.assemble([0], [d])
    np.testing.assert_array_equal(pts[0, :, :3].numpy(), xyz[::-1])
    floor = np.percentile(xyz[:, 2], 0.99)
    np.testing.assert_array_equal(pts[0, :, 3].numpy(), (xyz[::-1, 2] - floor).astype(np.float32))
    assert boxes[0].shape == (0, 7)


def test_host_staged_draws_give_the_same_batch_as_the_draw_objects(oracle_kernels):
    """``stage_draws`` (numpy draws in the reference's order, packed into one pinned buffer) +
    ``assemble_batch(staging=...)`` = ``assemble_batch(draws=[AugmentDraws ...])`` with the same
    generator state: points, boxes, labels, validity -- bit for bit."""
    ranges = dict(rot_range=(-0.087266, 0.087266), scale_range=(0.9, 1.1), translation_std=(0.1, 0.1, 0.05))
    with kernels.use_backend(oracle_kernels):
        scenes = ResidentScenes("cpu")
        for seed, n in [(1, 5000), (2, 900), (3, 2500)]:
            raw6, align, gt, labels = golden_inputs.raw_scene(seed, n, False)
            scenes.add_scene(raw6[:, :3], align, gt, labels)
        scenes.finalize()
        ids, n_pts = [2, 0, 1], 1024
        rng = np.random.RandomState(77)
        draws = [draw_like_reference(rng, int(scenes.counts[s]), n_pts, **ranges) for s in ids]
        want_p, want_g = scenes.assemble_batch(ids, draws)
        staging = scenes.new_staging(len(ids), n_pts)
        scenes.stage_draws(staging, 1, ids, np.random.RandomState(77), **ranges)
        staging['dev'].copy_(staging['host'][1])
        got_p, got_g = scenes.assemble_batch(ids, staging=staging)
    assert torch.equal(got_p, want_p)
    assert torch.equal(got_g.boxes, want_g.boxes) and torch.equal(got_g.labels, want_g.labels)
    assert torch.equal(got_g.valid, want_g.valid) and torch.equal(got_g.count, want_g.count)
