"""oracle/nesie_oracle.c against independent numpy restatements and against the
reference's own compilable CPU source (oracle/_ref).  No GPU."""
import math

import numpy as np
import pytest
import torch

import oracle
from tests import _cases, _np_ref


def _fps(k, xyz, m):
    b, n, _ = xyz.shape
    temp = torch.full((b, n), 1e10)
    idx = torch.zeros((b, m), dtype=torch.int32)
    k.furthest_point_sampling_wrapper(b, n, m, xyz.contiguous(), temp, idx)
    return idx, temp


def test_block_size_is_floor_log2_capped():
    # opt_n_threads (furthest_point_sample_cuda.cu:11-15) evaluated through log():
    # exact powers of two must not round down.
    L = oracle.lib()
    for n in list(range(1, 5000)) + [1 << i for i in range(11, 21)] + [(1 << i) - 1 for i in range(11, 21)]:
        want = min(1 << (n.bit_length() - 1), 1024)
        assert L.oracle_fps_block_size(n) == want == _np_ref.ref_block_size(n), n


@pytest.mark.parametrize("n,m,kw", [
    (4096, 256, {}),                      # BASELINE config 1 shape (uniform)
    (4096, 256, dict(dup_frac=0.25)),     # duplicates: ties on d2 == 0 and beyond
    (1000, 300, dict(grid=True)),         # lattice: many exactly equal distances
    (37, 37, {}),                         # n not a power of two, m == n
    (64, 80, dict(dup_frac=0.5)),         # m > number of distinct locations
    (1, 3, {}), (2, 2, {}), (3, 5, {}),   # degenerate block sizes 1 and 2
    (1500, 64, dict(dup_frac=0.3)),       # bs = 1024 with a ragged second stripe
])
def test_fps_literal_tree_matches_key_formulation(oracle_kernels, n, m, kw):
    xyz = _cases.cloud(11 + n + m, 2, n, **kw)
    idx, temp = _fps(oracle_kernels, xyz, m)
    for bi in range(2):
        want_idx, want_temp = _np_ref.fps_key(xyz[bi].numpy(), m)
        np.testing.assert_array_equal(idx[bi].numpy(), want_idx)
        np.testing.assert_array_equal(temp[bi].numpy(), want_temp)
    assert (idx[:, 0] == 0).all()


def test_fps_tie_rule_is_bit_reversed_not_lowest_thread(oracle_kernels):
    # 1024 points; after picking 0, points 256 and 512 are the two farthest and
    # equidistant: the tree keeps thread 512 (bitrev 1) over thread 256 (bitrev 2)
    # -- SURVEY.md appendix A.1.
    xyz = torch.zeros(1, 1024, 3)
    xyz[0, 256, 0] = 5.0
    xyz[0, 512, 0] = -5.0
    idx, _ = _fps(oracle_kernels, xyz, 2)
    assert idx[0, 1].item() == 512


def test_fps_m_zero_writes_nothing(oracle_kernels):
    xyz = _cases.cloud(0, 1, 16)
    idx = torch.full((1, 0), -7, dtype=torch.int32)
    oracle_kernels.furthest_point_sampling_wrapper(1, 16, 0, xyz, torch.full((1, 16), 1e10), idx)


def test_fps_with_dist_matches_dfps_on_squared_distances(oracle_kernels):
    xyz = _cases.cloud(5, 2, 200, dup_frac=0.2)
    d = torch.stack([torch.from_numpy(_np_ref.sqdist(x.numpy()[:, None, :], x.numpy()[None, :, :]))
                     for x in xyz]).contiguous()
    temp = torch.full((2, 200), 1e10)
    idx = torch.zeros((2, 50), dtype=torch.int32)
    oracle_kernels.furthest_point_sampling_with_dist_wrapper(2, 200, 50, d, temp, idx)
    want, _ = _fps(oracle_kernels, xyz, 50)
    assert torch.equal(idx, want)


@pytest.mark.parametrize("n,m,r,ns,min_r,kw", [
    (4096, 128, 0.2, 32, 0.0, {}),                   # config-1 radius / nsample
    (4096, 128, 0.2, 32, 0.0, dict(dup_frac=0.25)),
    (512, 64, 0.05, 16, 0.0, {}),                    # mostly empty / single-hit balls
    (512, 64, 0.8, 16, 0.3, {}),                     # dilated (min_radius > 0): d2 == 0 clause
    (100, 10, 10.0, 128, 0.0, {}),                   # nsample > n
])
def test_ball_query(oracle_kernels, n, m, r, ns, min_r, kw):
    xyz = _cases.cloud(3 + n, 2, n, **kw)
    centres = xyz[:, torch.randperm(n, generator=torch.Generator().manual_seed(1))[:m]].contiguous()
    centres[:, -1] += 100.0  # one centre with an empty ball: row stays zero
    idx = torch.zeros((2, m, ns), dtype=torch.int32)
    oracle_kernels.ball_query_wrapper(2, n, m, min_r, r, ns, centres, xyz, idx)
    for bi in range(2):
        want = _np_ref.ball_query(centres[bi].numpy(), xyz[bi].numpy(), min_r, r, ns)
        np.testing.assert_array_equal(idx[bi].numpy(), want)
    assert (idx[:, -1] == 0).all()


@pytest.mark.parametrize("n,m", [(300, 0), (300, 1), (300, 2), (300, 3), (700, 512), (64, 1500)])
def test_three_nn(oracle_kernels, n, m):
    unknown = _cases.cloud(7, 2, n, dup_frac=0.2)
    known = _cases.cloud(8, 2, max(m, 1), dup_frac=0.3)[:, :m].contiguous()
    if m >= 3:
        unknown[:, :3] = known[:, :3]  # exact zero distances
    d2 = torch.empty((2, n, 3)); ix = torch.empty((2, n, 3), dtype=torch.int32)
    oracle_kernels.three_nn_wrapper(2, n, m, unknown, known, d2, ix)
    for bi in range(2):
        wd, wi = _np_ref.three_nn(unknown[bi].numpy(), known[bi].numpy())
        np.testing.assert_array_equal(d2[bi].numpy(), wd)
        np.testing.assert_array_equal(ix[bi].numpy(), wi)


def test_group_gather_interpolate_and_grads(oracle_kernels):
    g = torch.Generator().manual_seed(0)
    b, c, n, m, ns = 2, 5, 50, 7, 4
    pts = torch.randn(b, c, n, generator=g)
    idx = torch.randint(0, n, (b, m, ns), generator=g, dtype=torch.int32)
    out = torch.empty(b, c, m, ns)
    oracle_kernels.group_points_forward(b, c, n, m, ns, pts, idx, out)
    want = torch.gather(pts.unsqueeze(2).expand(-1, -1, m, -1), 3,
                        idx.long().unsqueeze(1).expand(-1, c, -1, -1))
    assert torch.equal(out, want)
    go = torch.randn(b, c, m, ns, generator=g)
    gp = torch.zeros(b, c, n)
    oracle_kernels.group_points_backward(b, c, n, m, ns, go, idx, gp)
    want = torch.zeros(b, c, n).scatter_add_(2, idx.long().view(b, 1, -1).expand(-1, c, -1),
                                             go.view(b, c, -1))
    torch.testing.assert_close(gp, want, rtol=1e-6, atol=1e-6)

    gi = torch.randint(0, n, (b, m), generator=g, dtype=torch.int32)
    o2 = torch.empty(b, c, m)
    oracle_kernels.gather_points_wrapper(b, c, n, m, pts, gi, o2)
    assert torch.equal(o2, torch.gather(pts, 2, gi.long().unsqueeze(1).expand(-1, c, -1)))
    gp2 = torch.zeros(b, c, n)
    oracle_kernels.gather_points_grad_wrapper(b, c, n, m, o2, gi, gp2)
    want = torch.zeros(b, c, n).scatter_add_(2, gi.long().unsqueeze(1).expand(-1, c, -1), o2)
    torch.testing.assert_close(gp2, want, rtol=1e-6, atol=1e-6)

    q = 9
    ti = torch.randint(0, n, (b, q, 3), generator=g, dtype=torch.int32)
    w = torch.rand(b, q, 3, generator=g)
    o3 = torch.empty(b, c, q)
    oracle_kernels.three_interpolate_wrapper(b, c, n, q, pts, ti, w, o3)
    gathered = torch.gather(pts.unsqueeze(2).expand(-1, -1, q, -1), 3,
                            ti.long().unsqueeze(1).expand(-1, c, -1, -1))
    want = ((w[:, None, :, 0] * gathered[..., 0] + w[:, None, :, 1] * gathered[..., 1])
            + w[:, None, :, 2] * gathered[..., 2])
    assert torch.equal(o3, want)
    go3 = torch.randn(b, c, q, generator=g)
    gp3 = torch.zeros(b, c, n)
    oracle_kernels.three_interpolate_grad_wrapper(b, c, q, n, go3, ti, w, gp3)
    want = torch.zeros(b, c, n).scatter_add_(
        2, ti.long().view(b, 1, -1).expand(-1, c, -1),
        (go3.unsqueeze(-1) * w.unsqueeze(1)).reshape(b, c, -1))
    torch.testing.assert_close(gp3, want, rtol=1e-5, atol=1e-6)


def _vertices_for(mode, n, seed):
    """The geometry of the rotated-IoU chain (oracle/rotated_iou.py) on CPU up to sort_v's inputs."""
    from oracle import rotated_iou as R
    a, b = _cases.box_pairs(seed, n, mode)
    c1 = R.bev_corners(a[..., [0, 1, 3, 4, 6]])
    c2 = R.bev_corners(b[..., [0, 1, 3, 4, 6]])
    pts, hit = R.edge_crossings(c1, c2)
    v = torch.cat([c1, c2, pts.reshape(*c1.shape[:2], 16, 2)], 2)
    mask = torch.cat([R.corners_inside(c1, c2), R.corners_inside(c2, c1), hit.reshape(*c1.shape[:2], 16)], 2)
    nv = torch.sum(mask.int(), dim=2).int()
    mean = torch.sum(v * mask.float().unsqueeze(-1), dim=2, keepdim=True) / nv.unsqueeze(-1).unsqueeze(-1)
    return (v - mean).contiguous(), mask.contiguous(), nv.contiguous(), a, b


@pytest.mark.parametrize("mode", ["random", "identical", "disjoint", "aligned"])
def test_sort_vertices(oracle_kernels, mode):
    v, mask, nv, a, b = _vertices_for(mode, 96, 21)
    v = torch.nan_to_num(v)  # disjoint pairs: 0/0 mean
    idx = torch.empty((1, 96, 9), dtype=torch.int32)
    oracle_kernels.sort_vertices_forward(v, mask, nv, idx)
    want = _np_ref.sort_vertices(v[0].numpy(), mask[0].numpy(), nv[0].numpy())
    np.testing.assert_array_equal(idx[0].numpy(), want)
    if mode == "disjoint":
        assert (nv == 0).all() and (idx >= 8).all()


def test_rotated_iou_known_answers(oracle_kernels):
    """Closed-form IoUs through the whole chain with the oracle's sort_vertices."""
    from nesie_amd import kernels
    from nesie_amd.mmdet3d_ops import cal_iou_3d
    a = torch.tensor([[[0., 0., 0., 2., 2., 2., 0.],       # identical -> 1
                       [0., 0., 0., 2., 2., 2., 0.],       # shifted by half -> 1/3
                       [0., 0., 0., 2., 2., 2., 0.],       # disjoint -> 0
                       [0., 0., 0., 2., 2., 2., 0.],       # contained half-size -> 1/8
                       [0., 0., 0., 2., 2., 2., 0.]]])     # 45 degrees, same square
    b = torch.tensor([[[0., 0., 0., 2., 2., 2., 0.],
                       [1., 0., 0., 2., 2., 2., 0.],
                       [5., 5., 0., 2., 2., 2., 0.],
                       [0., 0., 0., 1., 1., 1., 0.],
                       [0., 0., 0., 2., 2., 2., math.pi / 4]]])
    with kernels.use_backend(oracle_kernels):
        iou = cal_iou_3d(a, b)[0]
    oct_area = 8 * (math.sqrt(2) - 1)  # regular octagon from two unit-apothem squares
    want = torch.tensor([1.0, 1 / 3, 0.0, 1 / 8, oct_area / (8 - oct_area)])
    torch.testing.assert_close(iou, want, rtol=1e-5, atol=1e-6)


def test_points_in_boxes_matches_numpy(oracle_kernels):
    boxes = _cases.boxes_lidar(2, 2, 9)
    pts = _cases.cloud(4, 2, 3000)
    out = torch.zeros((2, 3000, 9), dtype=torch.int32)
    oracle_kernels.points_in_boxes_batch(boxes, pts, out)
    for bi in range(2):
        np.testing.assert_array_equal(out[bi].numpy(),
                                      _np_ref.points_in_boxes(boxes[bi].numpy(), pts[bi].numpy()))
    assert 0 < out.sum() < out.numel()


@pytest.mark.skipif(not oracle.ref_points_in_boxes_available(),
                    reason="oracle/_ref not built (reference tree absent at build time)")
@pytest.mark.parametrize("yaw", [False, True])
def test_points_in_boxes_pinned_by_reference_cpu_source(oracle_kernels, yaw):
    """PIN: the reference's own points_in_boxes_cpu.cpp, compiled where it lies."""
    boxes = _cases.boxes_lidar(6, 1, 12, yaw=yaw)
    pts = _cases.cloud(9, 1, 20000)
    # put points exactly on faces / centres as well
    pts[0, :12] = boxes[0, :, :3]
    pts[0, 12:24, 0] = boxes[0, :, 0] + boxes[0, :, 4] / 2
    out = torch.zeros((1, 20000, 12), dtype=torch.int32)
    oracle_kernels.points_in_boxes_batch(boxes, pts, out)
    ref = oracle.ref_points_in_boxes_cpu(boxes[0], pts[0])  # (T, M)
    mism = (out[0].t() != ref).sum().item()
    # the reference evaluates cosf/sinf in float, the oracle rounds double cos/sin:
    # only points within an ulp of a rotated face may differ.
    assert mism <= (0 if not yaw else 2), mism
    assert ref.sum() > 100


def test_fma32_emulation_is_correctly_rounded():
    """tests/_np_ref.fma32 against exact rational arithmetic (fractions) on random and on
    cancellation-heavy operands: the emulation pins the oracle's libm fmaf() forms below."""
    from fractions import Fraction
    rng = np.random.default_rng(5)
    a = rng.standard_normal(400).astype(np.float32) * np.float32(3.7)
    b = rng.standard_normal(400).astype(np.float32)
    c = (-(a.astype(np.float64) * b.astype(np.float64))).astype(np.float32)   # near-total cancellation
    c[::2] = rng.standard_normal(200).astype(np.float32) * np.float32(1e-3)
    got = _np_ref.fma32(a, b, c)
    for x, y, z, g in zip(a, b, c, got):
        exact = Fraction(float(x)) * Fraction(float(y)) + Fraction(float(z))
        lo = np.float32(float(exact))            # float(exact) is the correctly rounded double ...
        # ... which may double-round: decide between its float32 neighbours exactly
        cands = {lo, np.nextafter(lo, np.float32(np.inf)), np.nextafter(lo, np.float32(-np.inf))}
        best = min(cands, key=lambda v: (abs(Fraction(float(v)) - exact), int(np.float32(v).view(np.uint32)) & 1))
        assert np.float32(g) == np.float32(best), (x, y, z, g, best)


@pytest.mark.parametrize('form', [1, 2])
def test_fused_distance_forms_match_their_numpy_restatement(oracle_kernels, form):
    """The distance-form switch (nesie_oracle.c header): with nvcc's default -fmad=true the
    reference's squared distance MAY be a fused chain; forms 1 / 2 restate the two contractions of
    the expression as written.  FPS picks + final running minima, ball-query rows and the 3-NN
    (distances, indices) of the oracle in that form equal the numpy restatement built on an exact
    float32 fma emulation -- and the form changes distances (so the flag is live)."""
    xyz = _cases.cloud(91, 2, 2048, dup_frac=0.2)
    cen = xyz[:, ::16].contiguous()
    try:
        oracle.set_distance_form(0)
        _, d0 = _three_nn_raw(oracle_kernels, cen, xyz)
        oracle.set_distance_form(form)
        _np_ref.FORM = form
        assert oracle.get_distance_form() == form
        idx, temp = _fps(oracle_kernels, xyz, 200)
        bq = torch.zeros(2, cen.shape[1], 16, dtype=torch.int32)
        oracle_kernels.ball_query_wrapper(2, 2048, cen.shape[1], 0.0, 0.13, 16, cen, xyz, bq)
        ni, nd = _three_nn_raw(oracle_kernels, cen, xyz)
        for bi in range(2):
            want_idx, want_temp = _np_ref.fps_key(xyz[bi].numpy(), 200)
            np.testing.assert_array_equal(idx[bi].numpy(), want_idx)
            np.testing.assert_array_equal(temp[bi].numpy(), want_temp)
            np.testing.assert_array_equal(
                bq[bi].numpy(), _np_ref.ball_query(cen[bi].numpy(), xyz[bi].numpy(), 0.0, 0.13, 16))
            wd, wi = _np_ref.three_nn(cen[bi].numpy(), xyz[bi].numpy())
            np.testing.assert_array_equal(nd[bi].numpy(), wd)
            np.testing.assert_array_equal(ni[bi].numpy(), wi)
        assert (nd != d0).any(), 'the fused form changed no distance: the switch is dead'
        assert float((nd - d0).abs().max()) < 1e-6
    finally:
        oracle.set_distance_form(0)
        _np_ref.FORM = 0
    with pytest.raises(ValueError):
        oracle.set_distance_form(3)


def _three_nn_raw(k, unknown, known):
    b, n, m = unknown.shape[0], unknown.shape[1], known.shape[1]
    d = torch.empty(b, n, 3)
    i = torch.empty(b, n, 3, dtype=torch.int32)
    k.three_nn_wrapper(b, n, m, unknown.contiguous(), known.contiguous(), d, i)
    return i, d
