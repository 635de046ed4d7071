"""HIP kernels (through the C ABI) against the CPU oracle on identical seeded inputs.
Index outputs must be bit-exact; fp32 outputs of deterministic kernels too; the
atomic scatter-adds are compared at 1e-5 (sum order differs, as in the reference)."""
import numpy as np
import pytest
import torch

from nesie_amd import kernels
from nesie_amd import mmdet3d_ops as ops
from tests import _cases

pytestmark = pytest.mark.gpu


def both(fn, oracle_kernels, dev, *cpu_args):
    """Run fn on HIP tensors with the product back end and on CPU tensors with the oracle."""
    gpu_args = [a.to(dev) if torch.is_tensor(a) else a for a in cpu_args]
    got = fn(*gpu_args)
    torch.cuda.synchronize()
    with kernels.use_backend(oracle_kernels):
        want = fn(*cpu_args)
    return got, want


def eq(got, want):
    assert got.dtype == want.dtype and got.shape == want.shape
    assert torch.equal(got.cpu(), want), f"max abs diff {(got.cpu().double() - want.double()).abs().max()}"


FPS_CASES = [
    # n, m, batch, kwargs                       kernel variant exercised
    (4096, 1024, 2, {}),                        # BASELINE config 1 (reg kernel, PPT 16)
    (4096, 1024, 2, dict(dup_frac=0.25)),
    (40000, 2048, 2, {}),                       # SA1 full size (stream kernel, PPT 40)
    (40000, 2048, 2, dict(dup_frac=0.3)),       # under-sampled scene => duplicates
    (2048, 1024, 3, {}), (1024, 512, 3, {}), (512, 256, 3, {}), (1024, 256, 3, {}),
    (1000, 300, 2, dict(grid=True)),            # lattice ties
    (37, 37, 2, {}), (64, 80, 2, dict(dup_frac=0.5)), (1, 3, 1, {}), (2, 2, 1, {}),
    (1500, 64, 2, dict(dup_frac=0.3)), (5000, 100, 2, {}), (9000, 64, 1, {}),
    (20000, 64, 1, {}), (30000, 33, 1, dict(dup_frac=0.2)), (45000, 40, 1, {}),
    (60000, 40, 1, {}), (70000, 20, 1, {}),     # > 65536: generic kernel
    # pruned kernel's buffered index stores: a flush plus a tail, exactly one flush, flush + 1
    (5000, 1500, 2, {}), (8192, 1024, 1, {}), (6400, 1025, 1, dict(dup_frac=0.2)),
]


@pytest.mark.parametrize("n,m,b,kw", FPS_CASES)
def test_fps_bit_exact(oracle_kernels, hip_device, n, m, b, kw):
    xyz = _cases.cloud(100 + n + m, b, n, **kw)
    got, want = both(ops.furthest_point_sample, oracle_kernels, hip_device, xyz, m)
    eq(got, want)


@pytest.mark.parametrize("n,m", [(40000, 300), (5000, 100), (65536, 50), (4097, 64)])
def test_fps_without_workspace_uses_the_streaming_kernel(oracle_kernels, hip_device, n, m):
    """nesie_furthest_point_sampling_wrapper (no scratch) must agree as well."""
    from nesie_amd import _lib
    xyz = _cases.cloud(n + 3, 2, n, dup_frac=0.2)
    t_cpu = torch.full((2, n), 1e10); i_cpu = torch.zeros((2, m), dtype=torch.int32)
    oracle_kernels.furthest_point_sampling_wrapper(2, n, m, xyz, t_cpu, i_cpu)
    x = xyz.to(hip_device)
    t_gpu = torch.full((2, n), 1e10, device=hip_device)
    i_gpu = torch.zeros((2, m), dtype=torch.int32, device=hip_device)
    _lib.call("nesie_furthest_point_sampling_wrapper", 2, n, m, x.data_ptr(), t_gpu.data_ptr(),
              i_gpu.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    eq(i_gpu, i_cpu); eq(t_gpu, t_cpu)


def test_fps_temp_buffer_matches(oracle_kernels, hip_device):
    for n, m in [(40000, 64), (3000, 200), (70000, 8)]:
        xyz = _cases.cloud(n, 2, n, dup_frac=0.1)
        t_cpu = torch.full((2, n), 1e10); i_cpu = torch.zeros((2, m), dtype=torch.int32)
        oracle_kernels.furthest_point_sampling_wrapper(2, n, m, xyz, t_cpu, i_cpu)
        t_gpu = torch.full((2, n), 1e10, device=hip_device)
        i_gpu = torch.zeros((2, m), dtype=torch.int32, device=hip_device)
        kernels.backend_for(t_gpu).furthest_point_sampling_wrapper(
            2, n, m, xyz.to(hip_device), t_gpu, i_gpu)
        eq(i_gpu, i_cpu); eq(t_gpu, t_cpu)


def test_fps_with_dist_bit_exact(oracle_kernels, hip_device):
    xyz = _cases.cloud(5, 2, 300, dup_frac=0.2)
    from nesie_amd.mmdet3d_ops.furthest_point_sample import calc_square_dist
    dist = calc_square_dist(xyz, xyz, norm=False).contiguous()
    got, want = both(ops.furthest_point_sample_with_dist, oracle_kernels, hip_device, dist, 77)
    eq(got, want)


BQ_CASES = [
    # n, m, radius, nsample, min_radius, batch, kwargs
    (4096, 1024, 0.2, 32, 0.0, 2, {}),                 # BASELINE config 1
    (40000, 2048, 0.2, 64, 0.0, 2, {}),                # SA1 full size
    (40000, 2048, 0.2, 64, 0.0, 1, dict(dup_frac=0.3)),
    (2048, 1024, 0.4, 32, 0.0, 3, {}), (1024, 512, 0.8, 16, 0.0, 3, {}),
    (512, 256, 1.2, 16, 0.0, 3, {}), (1024, 256, 0.3, 16, 0.0, 3, {}),
    (512, 64, 0.05, 16, 0.0, 2, {}),                   # empty / single-hit balls
    (512, 61, 0.8, 16, 0.3, 2, {}),                    # min_radius > 0, ragged m
    (100, 10, 10.0, 128, 0.0, 2, {}),                  # nsample > n
    (1000, 7, 0.5, 1, 0.0, 1, {}), (65, 3, 5.0, 70, 0.0, 1, {}),
]


@pytest.mark.parametrize("n,m,r,ns,min_r,b,kw", BQ_CASES)
def test_ball_query_bit_exact(oracle_kernels, hip_device, n, m, r, ns, min_r, b, kw):
    xyz = _cases.cloud(7 + n + m, b, n, **kw)
    pick = torch.randperm(n, generator=torch.Generator().manual_seed(1))[:m]
    centres = xyz[:, pick].contiguous()
    centres[:, -1] += 100.0  # an empty ball
    got, want = both(ops.ball_query, oracle_kernels, hip_device, min_r, r, ns, xyz, centres)
    eq(got, want)
    assert (got[:, -1] == 0).all()


@pytest.mark.parametrize("n,m,r,ns,min_r,b,kw", [
    (40000, 2048, 0.2, 64, 0.0, 2, {}),                    # SA1 as trained
    (40000, 2048, 0.2, 64, 0.0, 2, dict(dup_frac=0.3)),    # duplicates: > 64 hits per ball
    (5000, 300, 0.4, 16, 0.0, 2, {}),                      # n not a multiple of 64
    (8192, 512, 0.6, 32, 0.1, 1, dict(dup_frac=0.5)),      # hundreds of hits, inner radius
    (6400, 100, 5.0, 64, 0.0, 1, {}),                      # every point is a hit
    (20000, 700, 0.05, 8, 0.0, 2, {}),                     # mostly one or zero hits
])
def test_ball_query_over_the_fps_index_matches_the_full_scan(oracle_kernels, hip_device, n, m, r,
                                                             ns, min_r, b, kw):
    """FPS leaves the spatially sorted scene; the ball query of the same cloud goes through it
    (nesie_ball_query_indexed) and must return what the full scan and the oracle return."""
    from nesie_amd import _lib
    assert _lib.load().nesie_fps_leaves_index(b, n) == 1
    xyz_cpu = _cases.cloud(11 + n + m, b, n, **kw)
    xyz = xyz_cpu.to(hip_device)
    back = kernels.backend_for(xyz)
    # outside a scope FPS leaves nothing behind: a later ball query cannot meet a stale index
    ops.furthest_point_sample(xyz, m)
    assert back._index_for(xyz, b, n) is None and not back._spatial_index
    with kernels.spatial_index_scope():   # the hand-over of one set-abstraction forward
        picks = ops.furthest_point_sample(xyz, m)
        centres = torch.gather(xyz, 1, picks.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        centres[:, -1] += 100.0  # an empty ball
        assert back._index_for(xyz, b, n) is not None
        got = ops.ball_query(min_r, r, ns, xyz, centres)
        # an in-place change of the cloud retires the index inside the scope too
        keep = xyz.clone()
        xyz.add_(0.25)
        assert back._index_for(xyz, b, n) is None
        xyz.copy_(keep)
    assert not back._spatial_index        # ... and the scope's end drops it
    full = torch.zeros_like(got)
    _lib.call("nesie_ball_query_wrapper", b, n, m, min_r, r, ns, centres.data_ptr(),
              xyz.data_ptr(), full.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert torch.equal(got, full)
    with kernels.use_backend(oracle_kernels):
        want = ops.ball_query(min_r, r, ns, xyz_cpu, centres.cpu())
    eq(got, want)
    assert (got[:, -1] == 0).all()
    # a refilled buffer (same address) outside the scope: the full scan answers for the new cloud
    xyz.add_(0.25)
    moved = ops.ball_query(min_r, r, ns, xyz, centres + 0.25)
    with kernels.use_backend(oracle_kernels):
        want_moved = ops.ball_query(min_r, r, ns, xyz.cpu(), (centres + 0.25).cpu())
    eq(moved, want_moved)


@pytest.mark.parametrize("n,m,b", [(512, 256, 3), (1024, 512, 3), (49152, 1024, 2),
                                   (32768, 1024, 1), (300, 0, 1), (300, 1, 2), (300, 2, 2),
                                   (300, 3, 2), (64, 1500, 2), (1000, 2500, 1)])
def test_three_nn_bit_exact(oracle_kernels, hip_device, n, m, b):
    unknown = _cases.cloud(n, b, n, dup_frac=0.1)
    known = _cases.cloud(m + 1, b, max(m, 1), dup_frac=0.2)[:, :m].contiguous()
    if m >= 3:
        unknown[:, :3] = known[:, :3]
    (gd, gi), (wd, wi) = both(ops.three_nn, oracle_kernels, hip_device, unknown, known)
    eq(gi, wi)
    # three_nn returns torch.sqrt(dist2) (three_nn.py:38): device vs host sqrt may differ
    # by an ulp, so the kernel's own output dist2 is compared bit for bit below.
    torch.testing.assert_close(gd.cpu(), wd, rtol=1e-6, atol=1e-7)
    d2g = torch.empty((b, n, 3), device=hip_device)
    ig = torch.empty((b, n, 3), dtype=torch.int32, device=hip_device)
    kernels.backend_for(d2g).three_nn_wrapper(b, n, m, unknown.to(hip_device),
                                              known.to(hip_device), d2g, ig)
    d2c = torch.empty((b, n, 3)); ic = torch.empty((b, n, 3), dtype=torch.int32)
    oracle_kernels.three_nn_wrapper(b, n, m, unknown, known, d2c, ic)
    eq(ig, ic); eq(d2g, d2c)


def test_group_gather_interpolate_forward_exact_and_grads_close(oracle_kernels, hip_device):
    g = torch.Generator().manual_seed(0)
    for (b, c, n, m, ns) in [(2, 3, 4096, 1024, 32), (2, 131, 2048, 256, 16), (1, 1, 40000, 2048, 64),
                             (3, 9, 50, 7, 4)]:
        pts = torch.randn(b, c, n, generator=g)
        idx = torch.randint(0, n, (b, m, ns), generator=g, dtype=torch.int32)
        got, want = both(ops.grouping_operation, oracle_kernels, hip_device, pts, idx)
        eq(got, want)
        go = torch.randn(b, c, m, ns, generator=g)

        def grad(p, i, gout):
            p = p.clone().requires_grad_(True)
            ops.grouping_operation(p, i).backward(gout)
            return p.grad
        got, want = both(grad, oracle_kernels, hip_device, pts, idx, go)
        torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-5)

        gi = torch.randint(0, n, (b, m), generator=g, dtype=torch.int32)
        got, want = both(ops.gather_points, oracle_kernels, hip_device, pts, gi)
        eq(got, want)

        def ggrad(p, i, gout):
            p = p.clone().requires_grad_(True)
            ops.gather_points(p, i).backward(gout)
            return p.grad
        got, want = both(ggrad, oracle_kernels, hip_device, pts, gi, go[..., 0].contiguous())
        torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-5)

        q = 3 * m + 1
        ti = torch.randint(0, n, (b, q, 3), generator=g, dtype=torch.int32)
        w = torch.rand(b, q, 3, generator=g)
        got, want = both(ops.three_interpolate, oracle_kernels, hip_device, pts, ti, w)
        eq(got, want)
        go3 = torch.randn(b, c, q, generator=g)

        def igrad(p, i, ww, gout):
            p = p.clone().requires_grad_(True)
            ops.three_interpolate(p, i, ww).backward(gout)
            return p.grad
        got, want = both(igrad, oracle_kernels, hip_device, pts, ti, w, go3)
        torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("shape", [(3, 8, 70, 16), (2, 5, 300, 64), (2, 4, 9, 4)])
def test_fused_bn_row_bias_matches_materialised_sum(hip_device, shape):
    """row_bias form of the norm kernels (the per-proposal half of MiniPointNet's second
    conv, side_pooling_module.py:359-368) vs the same kernels on x + row_bias[..., None]."""
    import copy
    from nesie_amd.mmdet3d_ops.norm import FusedBNReLU2d
    g = torch.Generator().manual_seed(shape[2])
    B, C, K, G = shape
    x = torch.randn(shape, generator=g).to(hip_device)
    rb = (torch.randn(B, C, K, generator=g) * 1.5 + 0.5).to(hip_device)
    go = torch.randn(shape, generator=g).to(hip_device)
    a = FusedBNReLU2d(C, relu=True).to(hip_device)
    with torch.no_grad():
        a.weight.copy_(torch.rand(C, generator=g) + 0.5)
        a.bias.copy_(torch.randn(C, generator=g) * 0.3)
    b = copy.deepcopy(a)
    x1, r1 = x.clone().requires_grad_(True), rb.clone().requires_grad_(True)
    y1 = a(x1, row_bias=r1)
    y1.backward(go)
    x2, r2 = x.clone().requires_grad_(True), rb.clone().requires_grad_(True)
    y2 = b(x2 + r2.unsqueeze(-1))
    y2.backward(go)
    # same arithmetic (x + r is rounded once in both), so values agree to summation order
    torch.testing.assert_close(y1, y2, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(a.running_var, b.running_var, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(x1.grad, x2.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(r1.grad, r2.grad, rtol=1e-5, atol=2e-6 * G)
    torch.testing.assert_close(a.weight.grad, b.weight.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(a.bias.grad, b.bias.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("shape", [(8, 16, 1024), (3, 37, 250), (2, 128, 256), (4, 32, 30, 16)])
@pytest.mark.parametrize("relu", [True, False])
def test_fused_bn_folds_a_conv_bias(hip_device, shape, relu):
    """norm(x, chan_bias=b) -- the bias of the convolution in front, added inside the norm's own
    passes -- vs the same kernels on the materialised x + b: the sum is rounded once in both, so
    outputs, running statistics and input gradients are bit-equal; the bias gets no gradient (its
    true gradient is zero: the materialised form returns the rounding residue of sum(dx)).
    Training and evaluation mode, with and without p % 4 == 0."""
    import copy
    from nesie_amd.mmdet3d_ops.norm import FusedBNReLU1d, FusedBNReLU2d
    g = torch.Generator().manual_seed(sum(shape))
    C = shape[1]
    x = torch.randn(shape, generator=g).to(hip_device)
    cb = (torch.randn(C, generator=g) * 2.0 + 0.3).to(hip_device)
    go = torch.randn(shape, generator=g).to(hip_device)
    cls = FusedBNReLU1d if len(shape) == 3 else FusedBNReLU2d
    a = cls(C, relu=relu).to(hip_device)
    with torch.no_grad():
        a.weight.copy_(torch.rand(C, generator=g) + 0.5)
        a.bias.copy_(torch.randn(C, generator=g) * 0.3)
    b = copy.deepcopy(a)
    view = (1, C) + (1,) * (len(shape) - 2)
    x1, c1 = x.clone().requires_grad_(True), cb.clone().requires_grad_(True)
    y1 = a(x1, chan_bias=c1)
    y1.backward(go)
    x2, c2 = x.clone().requires_grad_(True), cb.clone().requires_grad_(True)
    y2 = b(x2 + c2.view(view))
    y2.backward(go)
    assert torch.equal(y1, y2)
    assert torch.equal(a.running_mean, b.running_mean) and torch.equal(a.running_var, b.running_var)
    assert torch.equal(x1.grad, x2.grad)
    assert torch.equal(a.weight.grad, b.weight.grad) and torch.equal(a.bias.grad, b.bias.grad)
    assert c1.grad is not None and float(c1.grad.abs().max()) == 0.0   # identically zero, not None
    assert c2.grad.abs().max().item() <= 1e-4 * go.abs().sum().item()  # the residue of a zero
    a.eval(); b.eval()
    with torch.no_grad():
        assert torch.equal(a(x, chan_bias=cb), b(x + cb.view(view)))


def _blend_case(k, segs, g, c, seed, b=2, m=300):
    gen = torch.Generator().manual_seed(seed)
    n = k * segs * g
    idx = torch.randint(0, m, (b, n, 3), generator=gen, dtype=torch.int32)
    w = torch.rand(b, n, 3, generator=gen)
    w = (w / w.sum(-1, keepdim=True)).contiguous()
    rel = torch.randn(b, n, 3, generator=gen)
    return gen, b, m, n, idx, w, rel


@pytest.mark.parametrize("k,segs,g,c", [(256, 6, 16, 256), (64, 6, 16, 128), (64, 1, 64, 64),
                                        (40, 6, 27, 19), (33, 1, 64, 8), (256, 6, 27, 256)])
def test_three_interpolate_segmented_bit_exact(oracle_kernels, hip_device, k, segs, g, c):
    """Per-face layout of the quality head (side_pooling_module.py:226-243, 304-313) vs the
    oracle's plain three_interpolate followed by the reference's view/cat/split order."""
    gen, b, m, n, idx, w, _ = _blend_case(k, segs, g, c, k + g)
    feats = torch.randn(b, c, m, generator=gen)
    lead = torch.randn(b, segs, 3, k * g, generator=gen)
    with kernels.use_backend(oracle_kernels):
        plain = ops.three_interpolate(feats, idx, w)                     # (b, c, n)
    full = torch.cat([lead.view(b, segs, 3, k, g).permute(0, 2, 3, 1, 4).reshape(b, 3, k, segs * g),
                      plain.view(b, c, k, segs * g)], 1)                 # reference's cat
    want = [t.contiguous() for t in torch.split(full, g, dim=-1)]        # reference's split
    out = torch.empty(b, segs, 3 + c, k * g, device=hip_device)
    out[:, :, :3] = lead.to(hip_device)
    ops.three_interpolate_segmented(feats.transpose(1, 2).contiguous().to(hip_device),
                                    idx.to(hip_device), w.to(hip_device), out, segs, g, 3)
    for s_ in range(segs):
        assert torch.equal(out[:, s_].reshape(b, 3 + c, k, g).cpu(), want[s_]), s_


@pytest.mark.parametrize("k,segs,g,h", [(64, 6, 16, 64), (32, 1, 64, 128), (64, 6, 27, 128),
                                        (512, 6, 16, 256)])
def test_blend_conv_matches_oracle(oracle_kernels, hip_device, k, segs, g, h):
    """BlendConv (first MiniPointNet conv through the 3-NN blend) on the HIP kernels vs the
    oracle's restatement from three_interpolate(+grad) and matmul: forward 1e-5, gradients of
    the table and of the xyz weight columns to summation order."""
    gen, b, m, n, idx, w, rel = _blend_case(k, segs, g, h, k + h)
    table = torch.randn(b, m, segs * h, generator=gen)
    wx = torch.randn(segs, h, 3, generator=gen)
    go = torch.randn(b, segs, h, k * g, generator=gen)
    with kernels.use_backend(oracle_kernels):
        t0, x0 = table.clone().requires_grad_(True), wx.clone().requires_grad_(True)
        want = ops.blend_conv(t0, x0, idx, w, rel, segs, g)
        want.backward(go)
    t1 = table.to(hip_device).requires_grad_(True)
    x1 = wx.to(hip_device).requires_grad_(True)
    got = ops.blend_conv(t1, x1, idx.to(hip_device), w.to(hip_device), rel.to(hip_device), segs, g)
    got.backward(go.to(hip_device))
    torch.testing.assert_close(got.detach().cpu(), want.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(t1.grad.cpu(), t0.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(x1.grad.cpu(), x0.grad, rtol=1e-4, atol=2e-3)
    # deterministic mode: the staged backward (no float atomics) -- same tolerance against the oracle,
    # and a second evaluation gives the same bits
    hip = kernels.backend_for(t1)
    prev = hip.set_deterministic(True)
    try:
        assert hip.blend_backward_writes_table(h, idx.shape[1], segs, m)
        res = []
        for _ in range(2):
            t2 = table.to(hip_device).requires_grad_(True)
            x2 = wx.to(hip_device).requires_grad_(True)
            ops.blend_conv(t2, x2, idx.to(hip_device), w.to(hip_device), rel.to(hip_device), segs, g).backward(go.to(hip_device))
            res.append((t2.grad, x2.grad))
    finally:
        hip.set_deterministic(prev)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    torch.testing.assert_close(res[0][0].cpu(), t0.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(res[0][1].cpu(), x0.grad, rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("k,segs,g,h,hot", [(256, 6, 16, 256, 3), (128, 1, 64, 256, 1), (64, 6, 16, 128, 40)])
def test_staged_blend_backward_is_reproducible_and_equals_the_atomic_form(hip_device, k, segs, g, h, hot):
    """nesie_blend_conv_backward_staged with most taps on a few seeds (`hot`): a seed then collects
    rows from hundreds of 16-query groups, other seeds none at all.  Three evaluations: same bits;
    against a float64 scatter: rounding only; against the atomic form (NESIE_BLEND_STAGED=0's code
    path, zero-filled table): summation order only; rows of seeds without taps are written as zeros
    into a table that arrives full of NaN."""
    hip = kernels.backend_for(torch.empty(1, device=hip_device))
    prev = hip.set_deterministic(True)
    try:
        gen = torch.Generator().manual_seed(k + h)
        b, m = 2, 96
        n = k * segs * g
        hot_idx = torch.randint(0, hot, (b, n, 3), generator=gen)
        cold_idx = torch.randint(m // 2, m, (b, n, 3), generator=gen)
        idx = torch.where(torch.rand(b, n, 3, generator=gen) < 0.7, hot_idx, cold_idx).int().to(hip_device)
        w = torch.rand(b, n, 3, generator=gen).to(hip_device)
        rel = torch.randn(b, n, 3, generator=gen).to(hip_device)
        dy = torch.randn(b, segs, h, n // segs, generator=gen).to(hip_device)
        outs = []
        for _ in range(3):
            d_table = torch.full((b, m, segs * h), float('nan'), device=hip_device)
            d_wx = torch.empty(segs, h, 3, device=hip_device)
            hip.blend_conv_backward(dy, h, idx, w, rel, d_table, d_wx, segs, g)
            outs.append((d_table.clone(), d_wx.clone()))
        assert all(torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) for o in outs[1:])
        assert not torch.isnan(outs[0][0]).any()
        # float64 scatter: query q of face s (output order s, k, g) <- taps of row (k * segs + s) * g + gi
        want = torch.zeros(b, m, segs, h, dtype=torch.float64)
        dyc = dy.double().cpu().view(b, segs, h, k, g)
        idc = idx.cpu().long().view(b, k, segs, g, 3)
        wc = w.double().cpu().view(b, k, segs, g, 3)
        for bi in range(b):
            for s_ in range(segs):
                rows = idc[bi, :, s_].reshape(-1, 3)                           # (k * g, 3)
                contrib = dyc[bi, s_].reshape(h, k * g).t().unsqueeze(1) * wc[bi, :, s_].reshape(-1, 3).unsqueeze(-1)
                want[bi, :, s_].index_add_(0, rows.reshape(-1), contrib.reshape(-1, h))
        err = (outs[0][0].double().cpu().view(b, m, segs, h) - want).abs().max().item()
        assert err <= 3e-6 * want.abs().max().item(), err
        assert float(outs[0][0][:, hot:m // 2].abs().max()) == 0.0          # seeds no tap landed on
        hip.set_deterministic(False)
        d_table = torch.zeros(b, m, segs * h, device=hip_device)
        d_wx = torch.empty(segs, h, 3, device=hip_device)
        hip.blend_conv_backward(dy, h, idx, w, rel, d_table, d_wx, segs, g)
        torch.testing.assert_close(outs[0][0], d_table, rtol=1e-5, atol=1e-5 * float(d_table.abs().max()))
        assert torch.equal(outs[0][1], d_wx)
    finally:
        hip.set_deterministic(prev)


@pytest.mark.parametrize("mode", ["random", "identical", "disjoint", "aligned"])
def test_sort_vertices_bit_exact(oracle_kernels, hip_device, mode):
    from tests.test_oracle import _vertices_for
    v, mask, nv, _, _ = _vertices_for(mode, 4096, 33)
    v = torch.nan_to_num(v)
    got, want = both(ops.sort_v, oracle_kernels, hip_device, v, mask, nv)
    eq(got, want)


@pytest.mark.parametrize("mode", ["random", "identical", "aligned"])
def test_cal_iou_3d_matches_cpu_chain(oracle_kernels, hip_device, mode):
    a, b = _cases.box_pairs(44, 2048, mode)
    got, want = both(ops.cal_iou_3d, oracle_kernels, hip_device, a, b)
    torch.testing.assert_close(got.cpu(), want, rtol=1e-4, atol=1e-5)
    if mode == "identical":
        torch.testing.assert_close(got.cpu(), torch.ones_like(want), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("t,m,b,yaw", [(9, 40000, 2, False), (20, 40000, 1, True), (1, 100, 1, True),
                                       (300, 1000, 2, True)])
def test_points_in_boxes_bit_exact(oracle_kernels, hip_device, t, m, b, yaw):
    boxes = _cases.boxes_lidar(t, b, t, yaw=yaw)
    pts = _cases.cloud(m, b, m)
    pts[:, :min(t, m)] = boxes[:, :min(t, m), :3]
    got, want = both(ops.points_in_boxes_batch, oracle_kernels, hip_device, pts, boxes)
    eq(got, want)
    assert got.sum() > 0


def test_full_size_properties(hip_device):
    """Size-independent properties at BASELINE shapes, no oracle involved."""
    xyz = _cases.cloud(1, 8, 40000, dup_frac=0.2).to(hip_device)
    idx = ops.furthest_point_sample(xyz, 2048)
    assert idx.shape == (8, 2048) and (idx[:, 0] == 0).all()
    assert int(idx.min()) >= 0 and int(idx.max()) < 40000
    # FPS never re-picks a location while unpicked distinct locations remain:
    picked = torch.gather(xyz, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3))
    for bi in range(8):
        assert torch.unique(picked[bi], dim=0).shape[0] == 2048
    # greedy property: min distance to earlier picks is non-increasing
    p = picked[0].double()
    d = torch.cdist(p, p)
    mins = torch.stack([d[j, :j].min() for j in range(1, 2048)])
    assert (mins[1:] <= mins[:-1] + 1e-9).all()
    # ball query rows: ascending hits then first-hit padding, all within radius
    bq = ops.ball_query(0.0, 0.2, 64, xyz, picked.contiguous())
    nb = torch.gather(xyz, 1, bq.long().view(8, -1, 1).expand(-1, -1, 3)).view(8, 2048, 64, 3)
    d2 = ((nb - picked.unsqueeze(2)) ** 2).sum(-1)
    assert (d2 < 0.2 ** 2 + 1e-6).all()
    inc = bq[..., 1:] > bq[..., :-1]
    pad = bq[..., 1:] == bq[..., :1]
    assert (inc | pad).all()


@pytest.mark.parametrize("shape", [(2, 64, 2048, 64), (2, 128, 1024, 32), (3, 256, 512, 16),
                                   (1, 5, 7, 4), (2, 3, 5, 8), (1, 128, 512, 64)])
def test_group_max_pool_matches_aten(oracle_kernels, hip_device, shape):
    """values and arg-max routing identical to F.max_pool2d([1, ns]) incl. relu-style ties."""
    from nesie_amd.mmdet3d_ops.pool import group_max_pool
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g).clamp_min(0.0)  # many exact zeros => ties
    x[..., 1] = x[..., 0]                                # duplicated neighbours => ties
    go = torch.randn(shape[:-1], generator=g)
    xr = x.clone().requires_grad_(True)
    ref = torch.nn.functional.max_pool2d(xr, kernel_size=[1, shape[-1]]).squeeze(-1)
    ref.backward(go)
    xg = x.to(hip_device).requires_grad_(True)
    out = group_max_pool(xg)
    out.backward(go.to(hip_device))
    eq(out.detach(), ref.detach())
    eq(xg.grad, xr.grad)
    with kernels.use_backend(oracle_kernels):  # the CPU path's stand-in agrees too
        xc = x.clone().requires_grad_(True)
        oc = group_max_pool(xc)
        oc.backward(go)
    assert torch.equal(oc.detach(), ref.detach()) and torch.equal(xc.grad, xr.grad)


@pytest.mark.parametrize("shape,relu", [((8, 64, 2048, 16), True), ((2, 128, 1024), True),
                                        ((3, 5, 7), False), ((2, 259, 33), True),
                                        ((4, 256, 512, 4), False), ((1, 1, 1000), True)])
def test_fused_bn_relu_matches_aten(hip_device, shape, relu):
    """nesie_bn_relu_forward/backward vs torch batch_norm(+relu) evaluated on the CPU."""
    from nesie_amd.mmdet3d_ops.norm import FusedBNReLU1d, FusedBNReLU2d
    g = torch.Generator().manual_seed(len(shape) + shape[1])
    x = torch.randn(shape, generator=g) * 2.0 + 3.0  # mean >> 0: the shifted sums matter
    go = torch.randn(shape, generator=g)
    C = shape[1]
    ref_bn = (torch.nn.BatchNorm2d if len(shape) == 4 else torch.nn.BatchNorm1d)(C)
    with torch.no_grad():
        ref_bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        ref_bn.bias.copy_(torch.randn(C, generator=g))
    mine = (FusedBNReLU2d if len(shape) == 4 else FusedBNReLU1d)(C, relu=relu)
    mine.load_state_dict(ref_bn.state_dict())
    mine = mine.to(hip_device)
    xr = x.clone().requires_grad_(True)
    pre = ref_bn(xr)
    yr = torch.relu(pre) if relu else pre
    yr.backward(go)
    xg = x.to(hip_device).requires_grad_(True)
    yg = mine(xg)
    yg.backward(go.to(hip_device))
    torch.testing.assert_close(yg.detach().cpu(), yr.detach(), rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(mine.running_mean.cpu(), ref_bn.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(mine.running_var.cpu(), ref_bn.running_var, rtol=1e-5, atol=1e-6)
    assert int(mine.num_batches_tracked) == int(ref_bn.num_batches_tracked) == 1
    # Backward against the closed form in float64, with the ReLU mask taken from the
    # kernel's own output: positions whose pre-activation is within an ulp of zero may fall
    # on either side of the mask in two fp32 evaluations, which would move dgamma by a
    # whole g*xhat term, so the mask must be shared for a meaningful comparison.
    xd, god = x.double(), go.double()
    dims = [0] + list(range(2, x.dim()))
    n = x.numel() // C
    mean = xd.mean(dims, keepdim=True)
    var = xd.var(dims, unbiased=False, keepdim=True)
    invstd = 1.0 / torch.sqrt(var + mine.eps)
    xhat = (xd - mean) * invstd
    gmask = god * (yg.detach().cpu().double() > 0) if relu else god
    dbeta = gmask.sum(dims)
    dgamma = (gmask * xhat).sum(dims)
    gam = ref_bn.weight.detach().double().view([1, C] + [1] * (x.dim() - 2))
    dx = gam * invstd * (gmask - dbeta.view_as(gam) / n - xhat * dgamma.view_as(gam) / n)
    scale = max(dx.abs().max().item(), 1e-3)
    assert (xg.grad.cpu().double() - dx).abs().max().item() <= 1e-4 * scale  # north_star 1e-4
    torch.testing.assert_close(mine.weight.grad.cpu().double(), dgamma, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(mine.bias.grad.cpu().double(), dbeta, rtol=1e-4, atol=1e-4)
    # and ATen agrees wherever the masks agree (sanity of the closed form itself)
    if not relu:
        torch.testing.assert_close(xr.grad.double(), dx, rtol=1e-3, atol=1e-4 * scale)


@pytest.mark.parametrize("mode", ["random", "identical", "disjoint", "aligned"])
def test_fused_iou3d_matches_torch_chain(oracle_kernels, hip_device, mode):
    """nesie_iou3d_forward (value + Jacobian) vs the reference-shaped torch chain."""
    from oracle.rotated_iou import rotated_iou_3d

    def cal_iou_3d_torch(x, y):   # the chain on whatever device x lives, native sort_vertices
        return rotated_iou_3d(x, y, lambda v, m, n: ops.sort_v(v, m, n))
    a, b = _cases.box_pairs(17, 4096, mode)
    ag = a.to(hip_device).requires_grad_(True)
    got = ops.cal_iou_3d(ag, b.to(hip_device))
    w = torch.linspace(0.5, 1.5, 4096, device=hip_device).view(1, -1)
    (got * w).sum().backward()
    if mode == "identical":
        # exactly coincident rectangles: every mask (t in (0,1), num == 0, corner-in-box with
        # 1e-6 slack) is decided by the last bit of sin/cos, so the torch chain is evaluated
        # on the same device; a handful of pairs may still take the other branch
        ac = a.to(hip_device).requires_grad_(True)
        want = cal_iou_3d_torch(ac, b.to(hip_device))
        (want * w).sum().backward()
        want, ac_grad = want.detach().cpu(), ac.grad.cpu()
        bad = ((got.detach().cpu() - want).abs() > 1e-4).float().mean().item()
        assert bad < 0.01, bad
        assert (got.detach().cpu() > 0.999).float().mean() > 0.98
        return
    with kernels.use_backend(oracle_kernels):
        ac = a.clone().requires_grad_(True)
        want = cal_iou_3d_torch(ac, b)
        (want * w.cpu()).sum().backward()
    torch.testing.assert_close(got.detach().cpu(), want.detach(), rtol=1e-4, atol=1e-5)
    # gradients: compare where the pair is not within rounding of a topology change
    # (a vertex entering/leaving the polygon flips masks that are piecewise constant)
    gerr = (ag.grad.cpu() - ac.grad).abs().amax(-1)[0]
    tol = 1e-3 * ac.grad.abs().amax(-1)[0].clamp_min(1e-2)
    assert (gerr > tol).float().mean().item() < 0.01, (gerr > tol).float().mean().item()
    if mode == "disjoint":
        assert got.abs().max() == 0 and ag.grad.abs().max() == 0


@pytest.mark.parametrize("k", [64, 37, 1])
def test_lhs_nms_bit_exact(oracle_kernels, hip_device, k):
    from tests.golden import golden_inputs
    boxes = golden_inputs.nms_boxes()[:, :k].contiguous()
    g = torch.Generator().manual_seed(k)
    more = torch.cat([boxes, boxes[:, torch.randperm(k, generator=g)]], 0)  # permuted copies
    want = torch.zeros(more.shape[0], k, dtype=torch.uint8)
    oracle_kernels.lhs_nms_samecls(more.contiguous(), 0.25, want)
    got = torch.zeros(more.shape[0], k, dtype=torch.uint8, device=hip_device)
    kernels.backend_for(got).lhs_nms_samecls(more.contiguous().to(hip_device), 0.25, got)
    eq(got, want)
    assert 0 < int(want.sum()) < want.numel() or k == 1


@pytest.mark.parametrize("shape", [(2, 16, 128, 64), (3, 8, 70, 16), (2, 5, 33, 4), (1, 7, 9, 32)])
def test_fused_bn_relu_maxpool_matches_two_step(hip_device, shape):
    """nesie_bn_relu_maxpool_* (tail of an SA MLP: BN2d + ReLU + F.max_pool2d([1, ns]),
    point_sa_module.py:277-289, 136-158) vs the separate norm and pool kernels, which are
    themselves pinned against ATen above: same arithmetic per element, so values and arg-max
    agree exactly and gradients to summation order."""
    import copy
    from nesie_amd.mmdet3d_ops.norm import FusedBNReLU2d
    from nesie_amd.mmdet3d_ops.pool import group_max_pool
    g = torch.Generator().manual_seed(sum(shape))
    B, C, M, ns = shape
    x = (torch.randn(shape, generator=g) * 1.5 + 0.3).to(hip_device)
    go = torch.randn(B, C, M, generator=g).to(hip_device)
    a = FusedBNReLU2d(C, relu=True).to(hip_device)
    with torch.no_grad():
        a.weight.copy_(torch.rand(C, generator=g) + 0.5)
        a.weight[0] = -0.7            # a negative scale flips which element is the max
        a.bias.copy_(torch.randn(C, generator=g) * 0.5)
        a.bias[1] = -50.0             # a channel whose activations are all clipped to zero
    b = copy.deepcopy(a)
    x1 = x.clone().requires_grad_(True)
    y1 = a.forward_max_pool(x1)
    y1.backward(go)
    x2 = x.clone().requires_grad_(True)
    y2 = group_max_pool(b(x2))
    y2.backward(go)
    assert torch.equal(y1, y2)
    assert torch.equal(a.running_mean, b.running_mean) and torch.equal(a.running_var, b.running_var)
    assert int(a.num_batches_tracked) == int(b.num_batches_tracked) == 1
    scale = x2.grad.abs().max().item()
    assert (x1.grad - x2.grad).abs().max().item() <= 1e-5 * max(scale, 1e-3)
    torch.testing.assert_close(a.weight.grad, b.weight.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(a.bias.grad, b.bias.grad, rtol=1e-4, atol=1e-5)
    assert y1[:, 1].abs().max() == 0 and x1.grad[:, 1].abs().max() == 0


@pytest.mark.parametrize("cout,cin,p,b", [(64, 4, 4096, 2), (64, 64, 1000, 3), (128, 131, 640, 2),
                                         (256, 128, 320, 2), (18, 7, 130, 1), (128, 64, 8192, 2)])
def test_conv_wgrad_matches_fp64(hip_device, cout, cin, p, b):
    """nesie_conv_wgrad (weight gradient of a 1x1 conv on the matrix cores, fixed-order
    partial sums) vs an fp64 einsum, with and without the recomputed BN + ReLU operand, and on
    a batch-strided x view.  north_star tolerance 1e-4."""
    hip = kernels.backend_for(torch.empty(1, device=hip_device))
    g = torch.Generator().manual_seed(cout + cin + p)
    dy = torch.randn(b, cout, p, generator=g).to(hip_device)
    xfull = torch.randn(b, cin + 3, p, generator=g).to(hip_device)
    x = xfull[:, 3:]                                   # batch stride (cin + 3) * p
    coef = (torch.rand(cin, 4, generator=g) + 0.5).to(hip_device)
    coef[:, 1] -= 1.0
    dw = torch.empty(cout, cin, device=hip_device)
    hip.conv_wgrad(dy, x, dw)
    want = torch.einsum('bmp,bkp->mk', dy.double(), x.double())
    assert ((dw.double() - want).norm() / want.norm()).item() < 1e-5
    first = dw.clone()
    hip.conv_wgrad(dy, x, dw)
    assert torch.equal(first, dw)                      # fixed summation order
    hip.conv_wgrad(dy, x, dw, x_coef=coef, x_relu=True)
    a = torch.relu(x.double() * coef[:, 0].double().view(1, -1, 1) + coef[:, 1].double().view(1, -1, 1))
    want = torch.einsum('bmp,bkp->mk', dy.double(), a)
    assert ((dw.double() - want).norm() / want.norm()).item() < 1e-5


@pytest.mark.parametrize("cout,cin,p,b", [(64, 4, 4096, 2), (64, 64, 1000, 2), (33, 7, 257, 1)])
def test_mlp_layer_forward_stream_matches_fp64(hip_device, cout, cin, p, b):
    """nesie_mlp_layer_forward_stream (the barrier-free form for skinny HBM-bound layers) vs
    fp64: output, and the (sum, sum of squares) partials of its statistics epilogue."""
    from nesie_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(cout * 3 + cin + p)
    x = torch.randn(b, cin, p, generator=g).to(hip_device)
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(hip_device)
    coef = torch.rand(cin, 4, generator=g).to(hip_device) + 0.5
    coef[:, 1] -= 1.0
    stream = torch.cuda.current_stream().cuda_stream
    for use_coef in (False, True):
        y = torch.full((b, cout, p), float('nan'), device=hip_device)
        nparts = lib.nesie_mlp_stream_partials(b, p)
        part = torch.zeros(nparts, cout, 2, device=hip_device)
        _lib.call('nesie_mlp_layer_forward_stream', b, cin, cout, p, x.data_ptr(), cin * p,
                  w.data_ptr(), coef.data_ptr() if use_coef else 0, 1, y.data_ptr(),
                  part.data_ptr(), stream)
        a = x.double()
        if use_coef:
            a = torch.relu(a * coef[:, 0].double().view(1, -1, 1) + coef[:, 1].double().view(1, -1, 1))
        want = torch.matmul(w.double().unsqueeze(0), a)
        assert (y.double() - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item())
        torch.testing.assert_close(part[..., 0].double().sum(0), want.sum((0, 2)), rtol=1e-4,
                                   atol=1e-3)
        torch.testing.assert_close(part[..., 1].double().sum(0), (want ** 2).sum((0, 2)),
                                   rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("n,m,ns,c,norm", [(2048, 1024, 32, 128, True), (500, 64, 16, 7, False),
                                          (40000, 2048, 64, 1, True)])
def test_query_group_cat_matches_the_literal_chain(oracle_kernels, hip_device, n, m, ns, c, norm):
    """QueryGroupCat (one pass) vs the reference's order through the oracle: transpose -> group
    xyz -> minus centre -> / radius -> group features -> cat (group_points.py:100-128).  Forward
    bit-exact; the feature gradient to summation order (scatter-add)."""
    from nesie_amd.mmdet3d_ops.group_points import QueryAndGroup
    g = torch.Generator().manual_seed(n + ns)
    xyz = torch.rand(2, n, 3, generator=g) * 3
    centres = xyz[:, torch.randperm(n, generator=g)[:m]].contiguous()
    feats = torch.randn(2, c, n, generator=g)
    go = torch.randn(2, 3 + c, m, ns, generator=g)
    qg = QueryAndGroup(0.4, ns, use_xyz=True, normalize_xyz=norm)
    with kernels.use_backend(oracle_kernels):
        idx = qg.ball_indices(xyz, centres)
        f0 = feats.clone().requires_grad_(True)
        want = qg(xyz, centres, f0, idx=idx)
        want.backward(go)
    f1 = feats.to(hip_device).requires_grad_(True)
    got = qg(xyz.to(hip_device), centres.to(hip_device), f1, idx=idx.to(hip_device))
    got.backward(go.to(hip_device))
    assert torch.equal(got.detach().cpu(), want.detach())
    torch.testing.assert_close(f1.grad.cpu(), f0.grad, rtol=1e-4, atol=1e-4)
    # the same backward as a gather-sum through the inverted index (no atomics): run twice,
    # bitwise equal, and equal to the scatter form within summation order
    from nesie_amd.mmdet3d_ops.group_points import inverted_index
    idx_d = idx.to(hip_device)
    csr = inverted_index(idx_d, n)        # (any n since round 5: 40 000 points = 5 windows of 8 192)
    order, sources = csr
    flat = idx.reshape(2, -1).long()
    assert order.dtype == torch.int32 and tuple(sources.shape) == (2, m * ns)
    assert torch.equal(torch.gather(flat, 1, order.cpu().long()), sources.cpu().long())
    assert torch.equal(sources.cpu().long(), flat.sort(1).values)       # grouped by point
    assert torch.equal(order.cpu().long().sort(1).values, torch.arange(m * ns).expand(2, -1))
    # ascending column inside every run = a stable sort by source point: reproducible sums
    assert torch.equal(order.cpu().long(), flat.argsort(dim=1, stable=True))
    grads = []
    for _ in range(2):
        f2 = feats.to(hip_device).requires_grad_(True)
        qg(xyz.to(hip_device), centres.to(hip_device), f2, idx=idx_d, csr=csr).backward(
            go.to(hip_device))
        grads.append(f2.grad.clone())
    assert torch.equal(grads[0], grads[1])          # one owner wave per run, no atomics: same bits
    torch.testing.assert_close(grads[0].cpu(), f0.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("n,m,ns,c", [(512, 256, 16, 259), (1024, 512, 16, 40), (2048, 1024, 32, 9)])
def test_csr_scatter_is_bitwise_reproducible_with_long_runs(hip_device, n, m, ns, c):
    """The grouped scatter-add through the inverted index when a few points sit in hundreds of
    groups (ball query keeps the FIRST nsample hits, so low-index points do): runs of 100 .. 3000
    entries cross many 64-entry chunks.  Every run is summed by one wave in an order fixed by the
    index: three launches give the same bits, equal to a float64 scatter within rounding -- also
    for the weighted three-tap form (three_interpolate's backward)."""
    hip = kernels.backend_for(torch.empty(1, device=hip_device))
    g = torch.Generator().manual_seed(n + c)
    # 60 % of the entries on 6 points, some points with no entry at all
    hot = torch.randint(0, 6, (2, m * ns), generator=g)
    cold = torch.randint(n // 2, n, (2, m * ns), generator=g)
    idx = torch.where(torch.rand(2, m * ns, generator=g) < 0.6, hot, cold).int().view(2, m, ns).contiguous()
    go = torch.randn(2, 3 + c, m, ns, generator=g)
    idx_d, go_d = idx.to(hip_device), go.to(hip_device)
    order, sources = hip.inverted_index(idx_d, n)
    runs = torch.bincount(idx[0].flatten().long(), minlength=n)
    assert int(runs.max()) > 3 * 64 and int((runs == 0).sum()) > 0
    outs = []
    for _ in range(3):
        gf = torch.zeros(2, c, n, device=hip_device)
        hip.query_and_group_backward_csr(go_d, (2, m, ns), order, sources, gf)
        outs.append(gf.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    want = torch.zeros(2, c, n, dtype=torch.float64)
    want.scatter_add_(2, idx.view(2, 1, -1).expand(-1, c, -1).long(), go[:, 3:].reshape(2, c, -1).double())
    err = (outs[0].cpu().double() - want).abs().max().item()
    assert err <= 2e-6 * want.abs().max().item(), err
    # the form that gathers straight from HBM (NESIE_CSR_ROWS=0: rows beyond the LDS budget) adds the
    # same terms in another fixed order: equal to rounding, and bitwise repeatable too
    import os
    os.environ['NESIE_CSR_ROWS'] = '0'
    try:
        two = []
        for _ in range(2):
            gf = torch.zeros(2, c, n, device=hip_device)
            hip.query_and_group_backward_csr(go_d, (2, m, ns), order, sources, gf)
            two.append(gf)
        assert torch.equal(two[0], two[1])
        assert (two[0] - outs[0]).abs().max().item() <= 2e-6 * want.abs().max().item()
    finally:
        del os.environ['NESIE_CSR_ROWS']
    # three weighted taps per column
    nq = m * ns // 3
    idx3 = idx.view(2, -1)[:, :nq * 3].reshape(2, nq, 3).contiguous()
    w3 = torch.rand(2, nq, 3, generator=g)
    go3 = torch.randn(2, c, nq, generator=g)
    o3, s3 = hip.inverted_index(idx3.to(hip_device), n)
    outs = []
    for _ in range(3):
        gp = torch.zeros(2, c, n, device=hip_device)
        hip.three_interpolate_grad_csr(go3.to(hip_device), w3.to(hip_device), o3, s3, gp)
        outs.append(gp.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    want = torch.zeros(2, c, n, dtype=torch.float64)
    contrib = (go3.double().unsqueeze(-1) * w3.double().unsqueeze(1)).reshape(2, c, -1)
    want.scatter_add_(2, idx3.view(2, 1, -1).expand(-1, c, -1).long(), contrib)
    err = (outs[0].cpu().double() - want).abs().max().item()
    assert err <= 2e-6 * want.abs().max().item(), err


@pytest.mark.parametrize("n,e,b", [(40000, 131072, 2), (8193, 5000, 3), (20000, 70000, 1), (2049, 64, 1),
                                   (300, 4000, 2)])
def test_inverted_index_of_any_point_count_is_the_stable_sort(hip_device, n, e, b):
    """nesie_inverted_index beyond 8 192 source points (windows of 8 192 per workgroup, each counting
    the entries of the earlier windows first): (order, sources) = the STABLE sort of the entries by
    source point, exactly -- with a few points in hundreds of entries and points without any."""
    hip = kernels.backend_for(torch.empty(1, device=hip_device))
    g = torch.Generator().manual_seed(n + e)
    hot = torch.randint(0, 5, (b, e), generator=g) * (n // 5)
    idx = torch.where(torch.rand(b, e, generator=g) < 0.3, hot, torch.randint(0, n, (b, e), generator=g)).int()
    order, sources = hip.inverted_index(idx.view(b, e, 1).to(hip_device), n)
    torch.cuda.synchronize()
    for bi in range(b):
        want = np.argsort(idx[bi].numpy(), kind='stable')
        assert np.array_equal(order[bi].cpu().numpy(), want.astype(np.int32))
        assert np.array_equal(sources[bi].cpu().numpy(), idx[bi].numpy()[want])


def test_reference_shaped_backward_calls_are_bitwise_reproducible_by_default(oracle_kernels, hip_device):
    """The drop-in API without any precomputed index (mmdet3d.ops call shapes): the backward of
    grouping_operation, QueryAndGroup, gather_points and three_interpolate -- atomicAdd scatters in
    group_points_cuda.cu:10-31, gather_points_cuda.cu:51-70, three_interpolate_cuda.cu:61-84 -- builds
    the inverted index itself (HipKernels.DETERMINISTIC, the default): three runs give the same bits,
    also at 40 000 source points and with indices that repeat hundreds of times; the atomic kernels
    (set_deterministic(False)) agree to rounding."""
    hip = kernels.backend_for(torch.empty(1, device=hip_device))
    assert hip.DETERMINISTIC, 'the fixed-order backward is the default'
    g = torch.Generator().manual_seed(5)
    for (b, c, n, m, ns) in [(2, 16, 40000, 2048, 64), (2, 131, 2048, 1024, 32), (3, 9, 50, 7, 4)]:
        pts = torch.randn(b, c, n, generator=g).to(hip_device)
        xyz = torch.rand(b, n, 3, generator=g).to(hip_device)
        # ball-query-like rows: low indices everywhere (the first nsample hits)
        idx = torch.where(torch.rand(b, m, ns, generator=g) < 0.5, torch.randint(0, min(n, 8), (b, m, ns), generator=g),
                          torch.randint(0, n, (b, m, ns), generator=g)).int().to(hip_device)
        centres = xyz[:, :m].contiguous()
        gi = torch.randint(0, max(n // 50, 2), (b, m), generator=g, dtype=torch.int32).to(hip_device)   # repeats
        ti = torch.randint(0, n, (b, 3 * m + 1, 3), generator=g, dtype=torch.int32).to(hip_device)
        w = torch.rand(b, 3 * m + 1, 3, generator=g).to(hip_device)
        grouper = ops.QueryAndGroup(0.3, ns, use_xyz=True, normalize_xyz=True)
        go = {k: torch.randn(*s_, generator=g).to(hip_device) for k, s_ in
              dict(group=(b, c, m, ns), qg=(b, 3 + c, m, ns), gather=(b, c, m), interp=(b, c, 3 * m + 1)).items()}

        def grads():
            out = {}
            for name, fn in (('group', lambda p: ops.grouping_operation(p, idx)),
                             ('qg', lambda p: grouper(xyz, centres, p, idx=idx)),
                             ('gather', lambda p: ops.gather_points(p, gi)),
                             ('interp', lambda p: ops.three_interpolate(p, ti, w))):
                p = pts.clone().requires_grad_(True)
                fn(p).backward(go[name])
                out[name] = p.grad.clone()
            torch.cuda.synchronize()
            return out
        runs = [grads() for _ in range(3)]
        for k in runs[0]:
            assert torch.equal(runs[0][k], runs[1][k]) and torch.equal(runs[0][k], runs[2][k]), (k, n)
        prev = hip.set_deterministic(False)
        try:
            atomic = grads()
        finally:
            hip.set_deterministic(prev)
        for k in atomic:
            scale = runs[0][k].abs().max().item()
            assert (atomic[k] - runs[0][k]).abs().max().item() <= 2e-5 * scale, (k, n)


@pytest.mark.parametrize("n,m,c", [(1024, 512, 256), (512, 256, 256), (100, 7, 5)])
def test_three_interpolate_grad_through_inverted_index(oracle_kernels, hip_device, n, m, c):
    """three_interpolate's backward as a scatter through inverted_index(idx, m) vs the
    reference's atomicAdd scatter restated in the oracle (three_interpolate_cuda.cu:61-84)."""
    from nesie_amd.mmdet3d_ops.group_points import inverted_index
    g = torch.Generator().manual_seed(n + m)
    feats = torch.randn(2, c, m, generator=g)
    idx = torch.randint(0, m, (2, n, 3), generator=g, dtype=torch.int32)
    w = torch.rand(2, n, 3, generator=g)
    w = (w / w.sum(-1, keepdim=True)).contiguous()
    go = torch.randn(2, c, n, generator=g)
    with kernels.use_backend(oracle_kernels):
        f0 = feats.clone().requires_grad_(True)
        want = ops.three_interpolate(f0, idx, w)
        want.backward(go)
    idx_d, w_d = idx.to(hip_device), w.to(hip_device)
    csr = inverted_index(idx_d, m)
    assert csr is not None
    f1 = feats.to(hip_device).requires_grad_(True)
    got = ops.three_interpolate(f1, idx_d, w_d, csr)
    got.backward(go.to(hip_device))
    assert torch.equal(got.detach().cpu(), want.detach())
    torch.testing.assert_close(f1.grad.cpu(), f0.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("k,segs,g,h", [(64, 6, 16, 64), (32, 1, 64, 128), (64, 6, 27, 128),
                                        (128, 6, 16, 256)])
def test_blend_conv_bn_matches_blend_then_norm(hip_device, k, segs, g, h):
    """BlendConvBN (conv output and its gradient recomputed, never stored) vs BlendConv followed
    by the fused training BatchNorm + ReLU, both on the HIP kernels pinned above: activations,
    running statistics, and the gradients of the table, the xyz columns, gamma and beta."""
    from nesie_amd.mmdet3d_ops.norm import BNReLUTrain
    gen, b, m, n, idx, w, rel = _blend_case(k, segs, g, h, k + 3 * h)
    dev = hip_device
    table = torch.randn(b, m, segs * h, generator=gen).to(dev)
    wx = torch.randn(segs, h, 3, generator=gen).to(dev)
    gamma = (torch.rand(segs * h, generator=gen) + 0.5).to(dev)
    beta = (torch.randn(segs * h, generator=gen) * 0.3).to(dev)
    go = torch.randn(b, segs, h, k * g, generator=gen).to(dev)
    idx, w, rel = idx.to(dev), w.to(dev), rel.to(dev)

    def run(fused):
        t, x = table.clone().requires_grad_(True), wx.clone().requires_grad_(True)
        ga, be = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        rm, rv = torch.zeros(segs * h, device=dev), torch.ones(segs * h, device=dev)
        if fused:
            out = ops.blend_conv_bn(t, x, ga, be, idx, w, rel, rm, rv, 0.1, 1e-5, segs, g)
        else:
            # (with the statistics partials of the blend epilogue feeding the norm when the
            # shape allows: the production path)
            c0, stats = ops.blend_conv(t, x, idx, w, rel, segs, g, True)
            out = BNReLUTrain.apply(c0.reshape(b, segs * h, k, g), ga, be, rm, rv, 0.1, 1e-5,
                                    True, None, stats if stats.numel() else None
                                    ).view(b, segs, h, k * g)
        out.backward(go)
        return out.detach(), t.grad, x.grad, ga.grad, be.grad, rm, rv

    got, want = run(True), run(False)
    torch.testing.assert_close(got[0], want[0], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(got[5], want[5], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(got[6], want[6], rtol=1e-4, atol=1e-6)
    # a ReLU mask within rounding of zero may flip between the two evaluation orders and moves
    # the per-channel sums of its channel: gradients are compared in norm
    for i, name in ((1, 'table'), (2, 'wx'), (3, 'gamma'), (4, 'beta')):
        err = (got[i] - want[i]).norm().item()
        assert err <= 5e-3 * want[i].norm().item() + 1e-5, (name, err, want[i].norm().item())


def test_blend_statistics_epilogue_feeds_the_norm(hip_device):
    """The (sum, sum of squares) partials the blend kernel leaves per 64-query tile, handed to
    nesie_bn_relu_forward as pre_partial, give the same normalisation and running statistics
    as the norm's own statistics pass over the tensor."""
    from nesie_amd.mmdet3d_ops.norm import BNReLUTrain
    k, segs, g, h = 64, 6, 16, 128
    gen, b, m, n, idx, w, rel = _blend_case(k, segs, g, h, 77)
    dev = hip_device
    table = (torch.randn(b, m, segs * h, generator=gen) + 0.7).to(dev)   # non-zero channel means
    wx = torch.randn(segs, h, 3, generator=gen).to(dev)
    gamma = (torch.rand(segs * h, generator=gen) + 0.5).to(dev)
    beta = (torch.randn(segs * h, generator=gen) * 0.3).to(dev)
    c0, stats = ops.blend_conv(table, wx, idx.to(dev), w.to(dev), rel.to(dev), segs, g, True)
    assert tuple(stats.shape) == (segs * h, b * (k * g // 64), 2)
    x = c0.reshape(b, segs * h, k, g)
    outs = []
    for pre in (stats, None):
        rm, rv = torch.zeros(segs * h, device=dev), torch.ones(segs * h, device=dev)
        y = BNReLUTrain.apply(x, gamma, beta, rm, rv, 0.1, 1e-5, True, None, pre)
        outs.append((y, rm, rv))
    torch.testing.assert_close(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(outs[0][2], outs[1][2], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("m,dups", [(1024, False), (1024, True), (300, True), (2048, False), (3000, False)])
def test_grid_taps_equal_three_nn_on_the_same_points(hip_device, m, dups):
    """The fused grid + tap kernel against three_nn on the grid points it reports: same indices
    (duplicated seeds: equal distances, the smaller index keeps the better slot), same weights."""
    from nesie_amd.mmdet3d_ops import three_nn
    g = torch.Generator().manual_seed(m + dups)
    B, K, gp = 3, 70, 96
    known = torch.rand(B, m, 3, generator=g) * torch.tensor([8.0, 6.0, 2.5])
    if dups:                                   # exact duplicates: equal distances, index decides
        known[:, m // 2:] = known[:, :m - m // 2]
    centre = torch.rand(B, K, 3, generator=g) * torch.tensor([8.0, 6.0, 2.5])
    centre[:, :5] = known[:, :5]               # zero distances
    centre[:, 5] += 30.0                       # a proposal far outside the seed cloud
    size = 0.2 + torch.rand(B, K, 3, generator=g) * 2
    heading = (torch.rand(B, K, generator=g) - 0.5) * 6
    mult = torch.rand(gp, 3, generator=g) * 2 - 1
    plane = torch.zeros(gp, 3)
    dev = lambda t: t.to(hip_device).contiguous()   # noqa: E731
    hip = kernels.backend_for(dev(known))
    idx, weight, rel = hip.grid_taps(dev(centre), dev(size), dev(heading), dev(mult), dev(plane), dev(known))
    world = rel + dev(centre).repeat_interleave(gp, 1)
    d, ref_idx = three_nn(world.contiguous(), dev(known))
    assert torch.equal(idx, ref_idx)
    w = 1.0 / (d + 1e-8)
    torch.testing.assert_close(weight, w / w.sum(-1, keepdim=True), rtol=1e-6, atol=1e-7)


def test_flat_adamw_with_clipping_matches_torch(hip_device):
    """dp.FlatAdamW (nesie_flat_adamw_step) == clip_grad_norm_ + torch.optim.AdamW, several steps,
    with and without the clip active."""
    from nesie_amd import dp
    g = torch.Generator(device=hip_device).manual_seed(4)
    n = 300_001
    p0 = torch.randn(n, device=hip_device, generator=g)
    a = torch.nn.Parameter(p0.clone())
    b = torch.nn.Parameter(p0.clone())
    opt_a = dp.FlatAdamW(a, lr=8e-3, weight_decay=0.01, max_norm=10.0)
    opt_b = torch.optim.AdamW([b], lr=8e-3, weight_decay=0.01)
    for it, scale in enumerate([1e-3, 5.0, 1e-2, 40.0, 1.0]):
        grad = torch.randn(n, device=hip_device, generator=g) * scale
        a.grad, b.grad = grad.clone(), grad.clone()
        want_norm = torch.nn.utils.clip_grad_norm_([b], max_norm=10.0, norm_type=2)
        opt_b.step()
        opt_a.step()
        torch.testing.assert_close(opt_a.grad_norm, want_norm, rtol=1e-5, atol=0)
        torch.testing.assert_close(a.grad, b.grad, rtol=1e-5, atol=1e-9)       # left clipped
        torch.testing.assert_close(a.detach(), b.detach(), rtol=1e-5, atol=1e-7, msg=str(it))
    st_a, st_b = opt_a.state[a], opt_b.state[b]
    assert float(st_a['step']) == float(st_b['step']) == 5
    torch.testing.assert_close(st_a['exp_avg'], st_b['exp_avg'], rtol=1e-5, atol=1e-8)
    torch.testing.assert_close(st_a['exp_avg_sq'], st_b['exp_avg_sq'], rtol=1e-5, atol=1e-10)


def test_captured_flat_adamw_follows_a_learning_rate_change(hip_device):
    """lr and weight decay are read from device memory (nesie_flat_adamw_step_dev): a step
    captured in a hipGraph at lr = 8e-3 and replayed after the scheduler set lr = 8e-4 (the
    reference's step decay, pretrain-010.py:112-114) applies 8e-4 -- compared with eager
    torch.optim.AdamW whose param_group was changed the same way."""
    from nesie_amd import dp
    g = torch.Generator(device=hip_device).manual_seed(5)
    n = 70_001
    p0 = torch.randn(n, device=hip_device, generator=g)
    a, b = torch.nn.Parameter(p0.clone()), torch.nn.Parameter(p0.clone())
    opt_a = dp.FlatAdamW(a, lr=8e-3, weight_decay=0.01, max_norm=10.0)
    opt_b = torch.optim.AdamW([b], lr=8e-3, weight_decay=0.01)
    grads = [torch.randn(n, device=hip_device, generator=g) for _ in range(4)]
    static_g = torch.zeros(n, device=hip_device)
    a.grad = grads[0].clone()
    b.grad = grads[0].clone()
    torch.nn.utils.clip_grad_norm_([b], 10.0)
    opt_a.step()                       # eager: creates the state, loads the library
    opt_b.step()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        a.grad.copy_(static_g)
        opt_a.step()
    for it, grad in enumerate(grads[1:]):
        if it == 1:                    # the scheduler's decay between two replays
            for o in (opt_a, opt_b):
                o.param_groups[0]['lr'] = 8e-4
                o.param_groups[0]['weight_decay'] = 0.02
        static_g.copy_(grad)
        opt_a.sync_hyper()
        graph.replay()
        b.grad = grad.clone()
        torch.nn.utils.clip_grad_norm_([b], 10.0)
        opt_b.step()
        torch.testing.assert_close(a.detach(), b.detach(), rtol=1e-5, atol=1e-7, msg=str(it))
    assert float(opt_a.state[a]['step']) == 4


def test_fresh_flat_adamw_resumes_from_its_own_checkpoint(hip_device, tmp_path):
    """save -> load into a FRESH dp.FlatAdamW (never stepped) -> step: Optimizer.load_state_dict
    leaves ``step`` on the CPU; FlatAdamW normalises it to the float32 device scalar the kernels
    read.  The resumed optimiser then takes the same step as the original."""
    import copy

    from nesie_amd import checkpoint as ck
    from nesie_amd import dp
    torch.manual_seed(6)
    m1 = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3)).to(hip_device)
    s1 = dp.FlatTrainState(m1.parameters())
    o1 = dp.FlatAdamW(s1.flat_param, lr=1e-2, weight_decay=0.05, max_norm=10.0)
    xs = [torch.randn(7, 6, device=hip_device) for _ in range(3)]

    def one(model, st, opt, x):
        st.begin()
        model(x).square().sum().backward()
        st.collect()
        opt.step()
    for x in xs[:2]:
        one(m1, s1, o1, x)
    path = ck.save_reference_checkpoint(m1, tmp_path, 1, 2, optimizer=o1, flat_state=s1)[0]
    m2 = copy.deepcopy(m1)
    saved = torch.load(path, weights_only=False)
    m2.load_state_dict(saved['state_dict'])
    s2 = dp.FlatTrainState(m2.parameters())
    o2 = dp.FlatAdamW(s2.flat_param, lr=1.0, weight_decay=0.0, max_norm=10.0)
    ck.load_per_parameter_optimizer_state(o2, s2, saved['optimizer'])
    assert o2.param_groups[0]['lr'] == 1e-2
    one(m1, s1, o1, xs[2])
    one(m2, s2, o2, xs[2])
    st2 = o2.state[s2.flat_param]
    assert st2['step'].is_cuda and st2['step'].dtype == torch.float32 and float(st2['step']) == 3
    for pa, pb in zip(m1.parameters(), m2.parameters()):
        torch.testing.assert_close(pa, pb, rtol=1e-6, atol=1e-8)


def test_eval_coefficients_follow_native_training_steps_and_the_teacher_swap(hip_device):
    """eval -> one native training step (running statistics written by the kernels through raw
    pointers, parameters through the flat optimiser vector) -> eval again: the second evaluation
    must use the NEW folded statistics (a cache keyed on tensor version counters served the first
    ones).  Same around EMATeacher.swap(), which copies through ``.data``."""
    from nesie_amd import dp
    from nesie_amd.mmdet3d_ops import norm as N
    torch.manual_seed(7)
    bn = N.FusedBNReLU2d(24).to(hip_device)
    flat = dp.FlatTrainState(bn.parameters())
    opt = dp.FlatAdamW(flat.flat_param, lr=0.1, weight_decay=0.0)
    x = torch.randn(4, 24, 32, 16, device=hip_device) * 3 + 1

    def reference():
        return N.eval_coefficients(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
    for it in range(3):
        bn.eval()
        with torch.no_grad():
            got = bn.eval_coef().clone()
            y = bn(x)
        want = reference()
        torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-7, msg=f'eval {it}')
        torch.testing.assert_close(
            y, torch.relu(x * want[:, 0].view(1, -1, 1, 1) + want[:, 1].view(1, -1, 1, 1)),
            rtol=1e-5, atol=1e-5)
        bn.train()
        flat.begin()
        bn(x).square().mean().backward()      # native forward: running stats by raw pointer
        flat.collect()
        opt.step()                            # parameters through the flat vector
    # teacher swap: parameters replaced through .data copies
    model = torch.nn.Module()
    model.bn = bn
    from nesie_amd.votenet.semi import EMATeacher
    teacher = EMATeacher(model)
    with torch.no_grad():
        bn.weight.mul_(3.0)
    bn.eval()
    with torch.no_grad():
        student = bn.eval_coef().clone()
        teacher.swap()
        swapped = bn.eval_coef().clone()
        torch.testing.assert_close(swapped, reference(), rtol=1e-6, atol=1e-7)
        teacher.swap()
        torch.testing.assert_close(bn.eval_coef(), student, rtol=0, atol=0)
    assert (swapped[:, 0] - student[:, 0]).abs().max() > 1e-3


@pytest.mark.parametrize('form', [1, 2])
def test_fused_distance_forms_bit_exact_vs_the_oracle_in_the_same_form(oracle_kernels, hip_device, form):
    """nesie_set_distance_form (include/nesie_ops.h): FPS, ball query, 3-NN and the quality head's
    grid taps with the squared distance as the fused chain an nvcc -fmad=true build of the
    reference may compute, against the oracle with the same switch (itself pinned by an exact numpy
    fma emulation in tests/test_oracle.py): indices and distances bit for bit -- incl. the SA1
    size, where the product's default form runs the bucket-pruned kernel + spatial index and the
    fused forms must fall back to the plain kernels.  Default form restored and re-checked."""
    import oracle
    from nesie_amd import _lib
    hip = kernels.backend_for(torch.empty(1, device=hip_device))
    lib = _lib.load()
    try:
        hip.set_distance_form(form)
        oracle.set_distance_form(form)
        assert hip.get_distance_form() == form and lib.nesie_fps_leaves_index(2, 40000) == 0
        for n, m, kw in [(4096, 512, dict(dup_frac=0.25)), (40000, 300, dict(dup_frac=0.3)), (1000, 200, dict(grid=True))]:
            xyz = _cases.cloud(500 + n, 2, n, **kw)
            got, want = both(ops.furthest_point_sample, oracle_kernels, hip_device, xyz, m)
            eq(got, want)
        xyz = _cases.cloud(77, 2, 40000, dup_frac=0.3)
        cen = xyz[:, ::40].contiguous()
        for r, ns in [(0.2, 64), (0.4, 32)]:
            got, want = both(lambda c, x: ops.ball_query(0.0, r, ns, x, c), oracle_kernels, hip_device, cen, xyz)
            eq(got, want)
        known = xyz[:, :1024].contiguous()
        got, want = both(lambda u, k: ops.three_nn(u, k)[1], oracle_kernels, hip_device, cen, known)
        eq(got, want)
        # the SQUARED distances as the kernels leave them (the python wrapper's sqrt is ATen's)
        n_, m_ = cen.shape[1], known.shape[1]
        d_g = torch.empty(2, n_, 3, device=hip_device); i_g = torch.empty(2, n_, 3, dtype=torch.int32, device=hip_device)
        hip.three_nn_wrapper(2, n_, m_, cen.to(hip_device), known.to(hip_device), d_g, i_g)
        d_c = torch.empty(2, n_, 3); i_c = torch.empty(2, n_, 3, dtype=torch.int32)
        oracle_kernels.three_nn_wrapper(2, n_, m_, cen, known, d_c, i_c)
        eq(d_g, d_c)
    finally:
        hip.set_distance_form(0)
        oracle.set_distance_form(0)
    assert hip.get_distance_form() == 0 and lib.nesie_fps_leaves_index(2, 40000) == 1
    xyz = _cases.cloud(4596, 2, 4096, dup_frac=0.25)
    got, want = both(ops.furthest_point_sample, oracle_kernels, hip_device, xyz, 512)
    eq(got, want)


@pytest.mark.parametrize("n,m,ns,c,norm,dup", [(1024, 256, 16, 256, True, False), (1024, 256, 16, 7, False, True),
                                              (300, 64, 8, 5, True, True)])
def test_sample_and_group_over_predicted_coordinates_matches_the_literal_chain(hip_device, n, m, ns, c, norm, dup):
    """SampleQueryGroupCat (vote aggregation's sample + group as one autograd node with a native
    coordinate gradient) vs the literal chain of the reference on the same device -- transpose,
    gather_points, transpose, ball_query, group_points x 2, sub, div, cat and autograd's backward
    through them (point_sa_module.py:122-131, group_points.py:98-128): centres, grouped tensor and
    ball indices exact, gradients of coordinates and features to summation order -- with gradients
    arriving at BOTH outputs, duplicated points (a point that is several centres) and twice for
    bitwise reproducibility."""
    import importlib
    gp = importlib.import_module('nesie_amd.mmdet3d_ops.group_points')   # (the package re-exports a FUNCTION of that name)
    from nesie_amd.mmdet3d_ops.pointnet_modules import PointSAModule
    g = torch.Generator().manual_seed(n + ns)
    xyz = torch.rand(2, n, 3, generator=g) * 2
    if dup:
        xyz[:, n // 2:] = xyz[:, :n - n // 2]          # every second half point duplicates a first half one
    feats = torch.randn(2, c, n, generator=g)
    sa = PointSAModule(mlp_channels=[c, 16, 16], num_point=m, radius=0.3, num_sample=ns,
                       normalize_xyz=norm).to(hip_device)
    picks = torch.stack([torch.randperm(n, generator=g)[:m] for _ in range(2)]).int()
    if dup:
        picks[:, 1] = picks[:, 0]                       # the same point sampled twice

    class Fixed(torch.nn.Module):
        def forward(self, *_):
            return picks.to(hip_device)
    sa.points_sampler = Fixed()
    go_c = torch.randn(2, m, 3, generator=g).to(hip_device)
    seen = {}
    real = sa._mlp_and_pool

    def tap(mlp, grouped, lead=0):
        seen['grouped'] = grouped
        seen['go'] = torch.randn(grouped.shape, generator=torch.Generator().manual_seed(5)).to(hip_device)
        return (grouped * seen['go']).sum((1, 3))      # a linear read-out: its gradient is `go`
    sa._mlp_and_pool = tap
    outs = []
    try:
        for fused in (False, True, True):
            gp.SAMPLE_GROUP_FUSED = fused
            x = xyz.to(hip_device).requires_grad_(True)
            f = feats.to(hip_device).requires_grad_(True)
            centres, pooled, idx = sa(x * 1.0, f * 1.0)
            (pooled.sum() + (centres * go_c).sum()).backward()
            outs.append((centres.detach(), seen['grouped'].detach(), x.grad.clone(), f.grad.clone()))
    finally:
        gp.SAMPLE_GROUP_FUSED = True
        sa._mlp_and_pool = real
    want, got, again = outs
    assert torch.equal(got[0], want[0])
    # (the literal chain's `/ max_radius` is ATen's tensor / python-scalar = multiply by the reciprocal
    # on the device; the kernel divides, like PyTorch-CPU and the oracle: <= 1 ulp)
    torch.testing.assert_close(got[1], want[1], rtol=3e-7, atol=1e-7)
    assert torch.equal(got[1][:, 3:], want[1][:, 3:])
    for a, b in ((got[2], want[2]), (got[3], want[3])):
        assert b.abs().max().item() > 0
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5 * b.abs().max().item())
    assert torch.equal(got[2], again[2]) and torch.equal(got[3], again[3])


@pytest.mark.parametrize("shape,slice_", [((8, 1536, 512), None), ((8, 20, 256), None), ((3, 7, 33), None),
                                          ((8, 220, 256), (20, 218))])
def test_channel_sum_matches_float64_and_repeats_bitwise(hip_device, shape, slice_):
    """nesie_channel_sum (a conv bias's gradient, autograd's grad.sum((0, 2))): one workgroup per
    channel in a fixed order -- float64 within rounding, bitwise equal on repeats, also on a channel
    slice of a wider tensor (batch-strided view)."""
    hip = kernels.backend_for(torch.empty(1, device=hip_device))
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g).to(hip_device)
    view = x if slice_ is None else x[:, slice_[0]:slice_[1]]
    a, b = hip.channel_sum(view), hip.channel_sum(view)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    want = view.double().sum((0, 2))
    assert (a.double() - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
    if view.shape[0] % 2 == 0:      # two weight groups: batch entries 0, 2, .. and 1, 3, ..
        g2 = hip.channel_sum(view, ng=2)
        want2 = torch.stack([view[0::2].double().sum((0, 2)), view[1::2].double().sum((0, 2))]).flatten()
        assert (g2.double() - want2).abs().max().item() <= 1e-5 * max(1.0, want2.abs().max().item())
