"""The fp32-MFMA pointwise-conv layer kernel (nesie_pw_layer_forward) and the fused shared-MLP
functions built on it, against fp64 / module-by-module evaluations of the reference's
ConvModule(Conv2d 1x1, BN2d, ReLU) chains (point_sa_module.py:277-289, 136-158;
side_pooling_module.py:343-370)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device('cuda:0')


def _hip():
    from nesie_amd import kernels
    return kernels.backend_for(torch.zeros(1, device=_dev()))


def _ref_layer(x, w, ng, coef, relu):
    """fp64: y[n] = W[n % ng] @ act(x[n])."""
    nb, k, p = x.shape
    xd = x.double()
    if coef is not None:
        c = coef.double().view(ng, k, 4)
        idx = torch.arange(nb, device=x.device) % ng
        xd = xd * c[idx, :, 0].unsqueeze(-1) + c[idx, :, 1].unsqueeze(-1)
        if relu:
            xd = xd.clamp_min(0)
    wd = w.double()[torch.arange(nb, device=x.device) % ng]
    return torch.bmm(wd, xd)


CASES = [  # nb, ng, k, cout, p
    (4, 1, 256, 128, 512), (12, 6, 256, 128, 256), (4, 2, 128, 256, 256), (2, 1, 64, 64, 1024),
    (2, 1, 64, 128, 512), (2, 1, 131, 128, 256), (2, 1, 259, 128, 192), (2, 1, 128, 128, 384),
    (3, 1, 100, 200, 256), (2, 1, 128, 64, 256), (2, 1, 256, 256, 128), (2, 1, 260, 90, 64),
    # odd multiples of the 64-position tile, ragged K and Cout
    (2, 1, 256, 128, 192), (2, 1, 128, 128, 192), (2, 1, 131, 64, 192), (2, 1, 100, 40, 384),
    (3, 1, 259, 200, 192), (2, 1, 17, 64, 128), (2, 1, 145, 128, 64), (2, 1, 150, 37, 128),
    # four K sub-tiles (the feature-propagation modules' 512 -> 256), one workgroup per tile at 256 rows
    (2, 1, 512, 256, 128), (2, 1, 400, 100, 64), (4, 2, 128, 256, 512), (2, 1, 72, 256, 128),
]


@pytest.mark.parametrize('nb,ng,k,cout,p', CASES)
def test_layer_forward_matches_fp64(nb, ng, k, cout, p):
    hip = _hip()
    assert hip.pw_supported(k, cout, p)
    g = torch.Generator(device=_dev()).manual_seed(nb * 1000 + k + cout)
    x = torch.randn(nb, k, p, device=_dev(), generator=g)
    w = torch.randn(ng, cout, k, device=_dev(), generator=g) / k ** 0.5
    coef = torch.rand(ng * k, 4, device=_dev(), generator=g) + 0.5
    coef[:, 1] -= 1.0
    coef[::5, 0] *= -1.0
    # plain product
    y = torch.empty(nb, cout, p, device=_dev())
    hip.pw_layer_forward(x, w, ng=ng, y=y)
    ref = _ref_layer(x, w, ng, None, False)
    assert (y.double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    # folded norm + ReLU on the operand, statistics of the output
    slots = hip.pw_stat_slots(nb, ng, k, cout, p)
    part = torch.zeros(ng, slots, cout, 4, device=_dev())
    hip.pw_layer_forward(x, w, ng=ng, in_coef=coef, in_relu=True, y=y, stat_part=part)
    ref = _ref_layer(x, w, ng, coef, True)
    assert (y.double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    gamma = torch.rand(ng * cout, device=_dev(), generator=g) + 0.5
    beta = torch.randn(ng * cout, device=_dev(), generator=g)
    rm, rv = torch.zeros(ng * cout, device=_dev()), torch.ones(ng * cout, device=_dev())
    out_coef = torch.empty(ng * cout, 4, device=_dev())
    hip.pw_stats_finalize(part, gamma, beta, rm, rv, 0.1, 1e-5, out_coef)
    r = ref.view(nb // ng, ng, cout, p).permute(1, 2, 0, 3).reshape(ng * cout, -1)
    mean, var = r.mean(1), r.var(1, unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    torch.testing.assert_close(out_coef[:, 2].double(), mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(out_coef[:, 3].double(), invstd, rtol=2e-5, atol=0)
    torch.testing.assert_close(out_coef[:, 0].double(), gamma.double() * invstd, rtol=2e-5, atol=0)
    torch.testing.assert_close(out_coef[:, 1].double(), beta.double() - mean * gamma.double() * invstd,
                               rtol=1e-4, atol=1e-5)
    n = r.shape[1]
    torch.testing.assert_close(rm.double(), 0.1 * mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rv.double(), 0.9 + 0.1 * var * n / (n - 1), rtol=2e-5, atol=0)


def test_statistics_survive_a_large_mean():
    """|mean| >> sigma: the shifted sums keep the variance (E[x^2] - E[x]^2 in fp32 would not)."""
    hip = _hip()
    nb, k, cout, p = 2, 64, 64, 4096
    g = torch.Generator(device=_dev()).manual_seed(7)
    x = torch.randn(nb, k, p, device=_dev(), generator=g) * 1e-3
    x[:, 0] = 1.0                       # a constant channel carries the mean
    w = torch.randn(1, cout, k, device=_dev(), generator=g) * 0.1
    w[0, :, 0] = 300.0
    y = torch.empty(nb, cout, p, device=_dev())
    part = torch.zeros(1, hip.pw_stat_slots(nb, 1, k, cout, p), cout, 4, device=_dev())
    hip.pw_layer_forward(x, w, y=y, stat_part=part)
    coef = torch.empty(cout, 4, device=_dev())
    hip.pw_stats_finalize(part, None, None, None, None, 0.1, 0.0, coef)
    r = y.double().permute(1, 0, 2).reshape(cout, -1)
    invstd = r.var(1, unbiased=False).rsqrt()
    assert (r.mean(1).abs() / r.std(1)).min() > 1e4
    torch.testing.assert_close(coef[:, 3].double(), invstd, rtol=1e-3, atol=0)


@pytest.mark.parametrize('group', [16, 32, 64])
def test_pooled_tail_matches_norm_relu_max(group):
    """last layer of a set-abstraction MLP: max over the neighbourhood of relu(bn(conv))."""
    hip = _hip()
    nb, k, cout, p = 2, 128, 128, 1024
    g = torch.Generator(device=_dev()).manual_seed(group)
    x = torch.randn(nb, k, p, device=_dev(), generator=g)
    x[:, :, 64:128] = x[:, :, 0:64]          # duplicated columns: ties inside groups
    w = torch.randn(1, cout, k, device=_dev(), generator=g) / k ** 0.5
    y = torch.empty(nb, cout, p, device=_dev())
    pg = 16 if group == 16 else 32
    npg = p // pg
    pool = (torch.empty(nb, cout, npg, device=_dev()), torch.empty(nb, cout, npg, device=_dev()),
            torch.empty(nb, cout, npg, dtype=torch.uint8, device=_dev()),
            torch.empty(nb, cout, npg, dtype=torch.uint8, device=_dev()))
    part = torch.zeros(1, hip.pw_stat_slots(nb, 1, k, cout, p), cout, 4, device=_dev())
    in_coef = torch.rand(k, 4, device=_dev(), generator=g) + 0.5
    in_coef[:, 1] -= 1.0
    hip.pw_layer_forward(x, w, in_coef=in_coef, in_relu=True, y=y, stat_part=part, pool_group=pg,
                         pool_min=True, pool_out=pool)
    ref = _ref_layer(x, w, 1, in_coef, True)
    assert (y.double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    gamma = torch.randn(cout, device=_dev(), generator=g)        # both signs
    beta = torch.randn(cout, device=_dev(), generator=g) * 0.3
    coef = torch.empty(cout, 4, device=_dev())
    hip.pw_stats_finalize(part, gamma, beta, None, None, 0.1, 1e-5, coef)
    pooled = torch.empty(nb, cout, p // group, device=_dev())
    arg = torch.empty(nb, cout, p // group, dtype=torch.uint8, device=_dev())
    hip.pw_pool_finish(1, p, group, pg, pool, coef, True, pooled, arg)
    a = torch.relu(y * coef[:, 0].view(1, -1, 1) + coef[:, 1].view(1, -1, 1)).view(nb, cout, -1, group)
    want = a.max(-1).values
    torch.testing.assert_close(pooled, want, rtol=1e-6, atol=1e-6)
    # the recorded position holds the extremum of the raw output the scale's sign selects
    yv = y.view(nb, cout, -1, group)
    picked = torch.gather(yv, 3, arg.long().unsqueeze(-1)).squeeze(-1)
    ext = torch.where(coef[:, 0].view(1, -1, 1) >= 0, yv.max(-1).values, yv.min(-1).values)
    assert torch.equal(picked, ext)
    first = torch.where(coef[:, 0].view(1, -1, 1, 1) >= 0, yv == ext.unsqueeze(-1), yv == ext.unsqueeze(-1))
    assert torch.equal(arg.long(), first.float().argmax(-1))      # first position on ties


def test_row_bias_bias_and_transposed_weights():
    hip = _hip()
    nb, ng, k, cout, p, grp = 4, 2, 128, 256, 512, 16
    g = torch.Generator(device=_dev()).manual_seed(3)
    x = torch.randn(nb, k, p, device=_dev(), generator=g)
    wt = torch.randn(ng, k, cout, device=_dev(), generator=g) / k ** 0.5       # stored transposed
    rb = torch.randn(nb, cout, p // grp, device=_dev(), generator=g)
    bias = torch.randn(ng * cout, device=_dev(), generator=g)
    y = torch.empty(nb, cout, p, device=_dev())
    part = torch.zeros(ng, hip.pw_stat_slots(nb, ng, k, cout, p), cout, 4, device=_dev())
    hip.pw_layer_forward(x, wt.transpose(1, 2), ng=ng, row_bias=rb, rb_group=grp, y=y, stat_part=part)
    ref = _ref_layer(x, wt.transpose(1, 2), ng, None, False) + rb.double().repeat_interleave(grp, 2)
    assert (y.double() - ref).abs().max().item() < 3e-5
    coef = torch.empty(ng * cout, 4, device=_dev())
    hip.pw_stats_finalize(part, None, None, None, None, 0.1, 1e-5, coef)
    r = ref.view(nb // ng, ng, cout, p).permute(1, 2, 0, 3).reshape(ng * cout, -1)
    torch.testing.assert_close(coef[:, 2].double(), r.mean(1), rtol=1e-5, atol=1e-6)
    # a strided destination (rows 3.. of a wider tensor), as the input-gradient product uses it
    wide = torch.zeros(nb, cout + 3, p, device=_dev())
    hip.pw_layer_forward(x, wt.transpose(1, 2), ng=ng, y=wide[:, 3:])
    ref = _ref_layer(x, wt.transpose(1, 2), ng, None, False)
    assert (wide[:, 3:].double() - ref).abs().max().item() < 3e-5 and wide[:, :3].abs().max().item() == 0


def test_unsupported_shapes_are_refused_not_miscomputed():
    hip = _hip()
    assert not hip.pw_supported(300, 128, 512) and not hip.pw_supported(128, 600, 512)
    assert not hip.pw_supported(600, 128, 512)
    assert not hip.pw_supported(256, 128, 80)        # positions must fill whole 64-wide tiles
    x = torch.randn(1, 256, 80, device=_dev())
    w = torch.randn(1, 128, 256, device=_dev())
    with pytest.raises(RuntimeError):
        hip.pw_layer_forward(x, w, y=torch.empty(1, 128, 80, device=_dev()))


def test_norm_backward_from_the_raw_output():
    """nesie_bn_relu_backward with y = NULL (fused forward) vs autograd of relu(batch_norm(x))."""
    hip = _hip()
    b, c, p, grp = 3, 48, 1024, 16
    g = torch.Generator(device=_dev()).manual_seed(11)
    x = torch.randn(b, c, p, device=_dev(), generator=g) * 2 + 0.5
    gamma = (torch.randn(c, device=_dev(), generator=g)).requires_grad_(True)
    beta = (torch.randn(c, device=_dev(), generator=g) * 0.2).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    out = torch.relu(torch.nn.functional.batch_norm(xr, None, None, gamma, beta, True, 0.1, 1e-5))
    dy = torch.randn(out.shape, device=_dev(), generator=g)
    out.backward(dy)
    mean = x.double().transpose(0, 1).reshape(c, -1).mean(1)
    var = x.double().transpose(0, 1).reshape(c, -1).var(1, unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    coef = torch.stack([gamma.detach().double() * invstd, beta.detach().double() - mean * gamma.detach().double() * invstd,
                        mean, invstd], 1).float().contiguous()
    dx = torch.empty_like(x)
    dgamma, dbeta = torch.empty(c, device=_dev()), torch.empty(c, device=_dev())
    dsum = torch.empty(b, c, p // grp, device=_dev())
    hip.bn_relu_backward(dy, x, None, gamma.detach(), beta.detach(), coef[:, 2].contiguous(),
                         coef[:, 3].contiguous(), coef, True, dx, dgamma, dbeta, d_row_bias=dsum, group=grp)
    torch.testing.assert_close(dx, xr.grad, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(dgamma, gamma.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(dbeta, beta.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(dsum, dx.view(b, c, -1, grp).sum(-1), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('k,cin,p,ng', [(128, 128, 2048, 1), (256, 128, 1024, 1), (128, 64, 4096, 1),
                                          (256, 128, 512, 3), (128, 256, 1024, 2)])
def test_input_gradient_with_the_norm_reduction_in_its_epilogue(k, cin, p, ng):
    """nesie_pw_dgrad_bn_reduce + nesie_bn_relu_backward_apply == the input-gradient product
    followed by autograd's backward of relu(batch_norm(z)) (fp64), per weight group."""
    hip = _hip()
    nb = 2 * ng
    g = torch.Generator(device=_dev()).manual_seed(k + cin + p)
    dy = torch.randn(nb, k, p, device=_dev(), generator=g)
    w = torch.randn(ng, k, cin, device=_dev(), generator=g) / k ** 0.5     # layer weight (Cout=k, Cin)
    z = torch.randn(nb, cin, p, device=_dev(), generator=g) * 1.5 + 0.3
    gamma = torch.randn(ng * cin, device=_dev(), generator=g)
    beta = torch.randn(ng * cin, device=_dev(), generator=g) * 0.3
    # channel (s, m) of the stacked view (B, ng*cin, P)
    zs = z.view(nb // ng, ng * cin, p)
    mean = zs.double().transpose(0, 1).reshape(ng * cin, -1).mean(1)
    var = zs.double().transpose(0, 1).reshape(ng * cin, -1).var(1, unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    coef = torch.stack([gamma.double() * invstd, beta.double() - mean * gamma.double() * invstd,
                        mean, invstd], 1).float().contiguous()
    da = torch.empty(nb, cin, p, device=_dev())
    part = hip.pw_dgrad_bn_reduce(dy, w.transpose(1, 2), z, coef, da, ng=ng)
    dz = torch.empty_like(z)
    dgamma, dbeta = torch.empty(ng * cin, device=_dev()), torch.empty(ng * cin, device=_dev())
    hip.bn_relu_backward_apply(da.view(nb // ng, ng * cin, p), zs, gamma, coef[:, 3].contiguous(),
                               coef, part, dz.view(nb // ng, ng * cin, p), dgamma, dbeta)
    # fp64 referee
    zr = zs.double().clone().requires_grad_(True)
    gr, br = gamma.double().clone().requires_grad_(True), beta.double().clone().requires_grad_(True)
    act = torch.relu(torch.nn.functional.batch_norm(zr, None, None, gr, br, True, 0.1, 1e-5))
    wfull = w.double()[torch.arange(nb, device=_dev()) % ng]               # (nb, k, cin)
    da_ref = torch.bmm(wfull.transpose(1, 2), dy.double())                  # (nb, cin, p)
    act.backward(da_ref.view(nb // ng, ng * cin, p))
    scale = da_ref.abs().max().item()
    assert (da.double() - da_ref).abs().max().item() < 1e-5 * scale * k ** 0.5
    assert (dz.view_as(zr).double() - zr.grad).abs().max().item() < 2e-5 * zr.grad.abs().max().item() * k ** 0.5
    torch.testing.assert_close(dgamma.double(), gr.grad, rtol=1e-4, atol=1e-4 * gr.grad.abs().max().item())
    torch.testing.assert_close(dbeta.double(), br.grad, rtol=1e-4, atol=1e-4 * br.grad.abs().max().item())


@pytest.mark.parametrize('cout,cin', [(128, 256), (256, 128), (128, 259), (128, 131)])
def test_weight_gradient_wide_shapes_and_strided_batches(cout, cin):
    hip = _hip()
    b, p, S = 3, 512, 2
    g = torch.Generator(device=_dev()).manual_seed(cout + cin)
    dy_all = torch.randn(b * S, cout, p, device=_dev(), generator=g)
    x_all = torch.randn(b * S, cin, p, device=_dev(), generator=g)
    coef = torch.rand(cin, 4, device=_dev(), generator=g) + 0.5
    coef[:, 1] -= 1.0
    for s in range(S):
        dy, x = dy_all[s::S], x_all[s::S]
        dw = torch.empty(cout, cin, device=_dev())
        hip.conv_wgrad(dy, x, dw, x_coef=coef, x_relu=True)
        a = torch.relu(x.double() * coef[:, 0].double().view(1, -1, 1) + coef[:, 1].double().view(1, -1, 1))
        ref = torch.bmm(dy.double(), a.transpose(1, 2)).sum(0)
        assert (dw.double() - ref).abs().max().item() < 1e-4 * ref.abs().max().item()


@pytest.mark.parametrize('nb,ng,co,ci,p', [(6, 3, 128, 256, 256), (4, 1, 256, 128, 512), (4, 2, 128, 128, 256),
                                           (2, 1, 128, 131, 512), (2, 1, 128, 259, 256), (2, 1, 64, 64, 1024),
                                           (2, 1, 128, 64, 512), (3, 1, 100, 70, 96), (2, 1, 40, 33, 64),
                                           # wide layers: column blocks of one launch each
                                           (4, 2, 256, 256, 256), (2, 1, 256, 512, 128), (2, 1, 256, 259, 64),
                                           (4, 2, 128, 515, 64), (2, 1, 200, 300, 96),
                                           # the 1-D chains' own shapes (vote module, feature propagation,
                                           # 8 scenes): TILED -- 64 x 64 blocks of the product, split-K
                                           (8, 1, 256, 256, 1024), (8, 1, 259, 256, 1024), (8, 1, 256, 512, 512),
                                           (8, 1, 256, 512, 1024), (12, 6, 128, 166, 512)])
def test_layer_weight_gradient_matches_fp64(nb, ng, co, ci, p):
    """nesie_pw_wgrad: sum over batches and positions of dy . act(x)^T per weight group, with the
    activation recomputed on load, on batch-strided operands."""
    hip = _hip()
    assert hip.pw_wgrad_supported(co, ci, p) or hip.pw_wgrad_tiled(nb, ng, co, ci, p)
    if nb == 8:
        assert hip.pw_wgrad_tiled(nb, ng, co, ci, p)
    g = torch.Generator(device=_dev()).manual_seed(co * 3 + ci + ng)
    dy_all = torch.randn(nb, co + 5, p, device=_dev(), generator=g)
    x_all = torch.randn(nb, ci + 3, p, device=_dev(), generator=g)
    dy, x = dy_all[:, 5:], x_all[:, 3:]              # batch strides wider than the rows used
    coef = torch.rand(ng * ci, 4, device=_dev(), generator=g) + 0.5
    coef[:, 1] -= 1.0
    coef[::3, 0] *= -1.0
    for use_coef in (True, False):
        dw = torch.empty(ng, co, ci, device=_dev())
        hip.pw_wgrad(dy, x, dw, ng=ng, x_coef=coef if use_coef else None)
        xd = x.double()
        if use_coef:
            c = coef.double().view(ng, ci, 4)[torch.arange(nb, device=_dev()) % ng]
            xd = (xd * c[:, :, 0:1] + c[:, :, 1:2]).clamp_min(0)
        full = torch.bmm(dy.double(), xd.transpose(1, 2))          # (nb, co, ci)
        ref = full.view(nb // ng, ng, co, ci).sum(0)
        assert (dw.double() - ref).abs().max().item() < 1e-4 * ref.abs().max().item()
        again = torch.empty_like(dw)                               # fixed-order partial sums: same bits
        hip.pw_wgrad(dy, x, again, ng=ng, x_coef=coef if use_coef else None)
        assert torch.equal(dw, again)


@pytest.mark.parametrize('nb,ng,co,ci,p,in_place', [(6, 3, 128, 256, 256, False), (4, 1, 256, 128, 512, True),
                                                     (4, 2, 128, 128, 256, True), (2, 1, 64, 67, 1024, False),
                                                     (3, 1, 100, 70, 96, True), (2, 1, 128, 320, 128, False)])
def test_weight_gradient_fused_with_the_norm_backward(nb, ng, co, ci, p, in_place):
    """nesie_pw_wgrad_bn_backward = nesie_bn_relu_backward_apply + nesie_pw_wgrad in one pass over
    (dA, Z): dZ (written, optionally over dA), dW, dgamma, dbeta against float64."""
    hip = _hip()
    assert hip.pw_wgrad_bn_supported(co, ci, p)
    g = torch.Generator(device=_dev()).manual_seed(co + 7 * ci + ng)
    da = torch.randn(nb, co, p, device=_dev(), generator=g)
    z = torch.randn(nb, co, p, device=_dev(), generator=g) * 1.5 + 0.7
    x_all = torch.randn(nb, ci + 3, p, device=_dev(), generator=g)
    x = x_all[:, 3:]
    gamma = torch.randn(ng * co, device=_dev(), generator=g)
    beta = torch.randn(ng * co, device=_dev(), generator=g) * 0.3
    xcoef = torch.rand(ng * ci, 4, device=_dev(), generator=g) + 0.5
    xcoef[:, 1] -= 1.0
    grp = torch.arange(nb, device=_dev()) % ng
    # the layer's own forward statistics and folded coefficients, per weight group
    zd = z.double().view(nb // ng, ng, co, p)
    mean = zd.mean((0, 3))
    invstd = (zd.var((0, 3), unbiased=False) + 1e-5).rsqrt()                 # (ng, co)
    scale = gamma.double().view(ng, co) * invstd
    shift = beta.double().view(ng, co) - mean * scale
    zcoef = torch.stack([scale, shift, mean, invstd], -1).view(ng * co, 4).float().contiguous()
    zc = zcoef.double().view(ng, co, 4)
    mask = (torch.addcmul(zcoef[:, 1].view(ng, co)[grp].unsqueeze(-1), z, zcoef[:, 0].view(ng, co)[grp].unsqueeze(-1)) > 0)
    gg = torch.where(mask, da, torch.zeros_like(da)).double()
    zhat = (z.double() - zc[grp][:, :, 2:3]) * zc[grp][:, :, 3:4]
    s0 = gg.view(nb // ng, ng, co, p).sum((0, 3))
    s1 = (gg * zhat).view(nb // ng, ng, co, p).sum((0, 3))
    n = nb // ng * p
    a = gamma.double().view(ng, co) * zc[:, :, 3]
    dz_ref = a[grp].unsqueeze(-1) * (gg - (s0 / n)[grp].unsqueeze(-1) - zhat * (s1 / n)[grp].unsqueeze(-1))
    part = torch.stack([s0, s1], -1).view(ng * co, 1, 2).float().contiguous()   # one slot per channel
    for use_coef in (True, False):
        xd = x.double()
        if use_coef:
            c = xcoef.double().view(ng, ci, 4)[grp]
            xd = (xd * c[:, :, 0:1] + c[:, :, 1:2]).clamp_min(0)
        dw_ref = torch.bmm(dz_ref, xd.transpose(1, 2)).view(nb // ng, ng, co, ci).sum(0)
        src = da.clone()
        dz = src if in_place else torch.empty_like(da)
        dw = torch.empty(ng, co, ci, device=_dev())
        dgamma, dbeta = torch.empty(ng * co, device=_dev()), torch.empty(ng * co, device=_dev())
        hip.pw_wgrad_bn_backward(src, z, zcoef, gamma, part, x, dz, dw, dgamma, dbeta, ng=ng,
                                 x_coef=xcoef if use_coef else None)
        assert (dz.double() - dz_ref).abs().max().item() < 2e-5 * dz_ref.abs().max().item()
        assert (dw.double() - dw_ref).abs().max().item() < 1e-4 * dw_ref.abs().max().item()
        torch.testing.assert_close(dgamma.double(), s1.view(-1), rtol=1e-5, atol=1e-4)
        torch.testing.assert_close(dbeta.double(), s0.view(-1), rtol=1e-5, atol=1e-4)
    # the gradient of a per-group row bias (the forward added it to z): group sums of dz
    for group in (16, 64):
        if p % group:
            continue
        d_rb = torch.zeros(nb, co, p // group, device=_dev())
        src = da.clone()
        dw = torch.empty(ng, co, ci, device=_dev())
        hip.pw_wgrad_bn_backward(src, z, zcoef, gamma, part, x, src, dw, dgamma, dbeta, ng=ng,
                                 x_coef=None, d_row_bias=d_rb, group=group)
        want = dz_ref.view(nb, co, p // group, group).sum(-1)
        assert (d_rb.double() - want).abs().max().item() < 2e-5 * max(want.abs().max().item(), 1e-6)
        assert (src.double() - dz_ref).abs().max().item() < 2e-5 * dz_ref.abs().max().item()


@pytest.mark.parametrize('b,cout,cin,p', [(2, 64, 4, 4096), (3, 64, 4, 1000), (2, 64, 3, 1003), (2, 128, 8, 520)])
def test_skinny_weight_gradient_with_the_norm_backward_on_its_load(b, cout, cin, p):
    """nesie_conv_wgrad_bn (the first layer of SA1: Cin = 4, nobody reads dZ) against float64: vector
    interior, run edges and the scalar path for unaligned p."""
    hip = _hip()
    g = torch.Generator(device=_dev()).manual_seed(cout + cin + p)
    da = torch.randn(b, cout, p, device=_dev(), generator=g)
    z = torch.randn(b, cout, p, device=_dev(), generator=g) * 1.3 - 0.4
    x = torch.randn(b, cin, p, device=_dev(), generator=g)
    gamma = torch.randn(cout, device=_dev(), generator=g)
    beta = torch.randn(cout, device=_dev(), generator=g) * 0.3
    zd = z.double()
    mean, invstd = zd.mean((0, 2)), (zd.var((0, 2), unbiased=False) + 1e-5).rsqrt()
    scale = gamma.double() * invstd
    zcoef = torch.stack([scale, beta.double() - mean * scale, mean, invstd], -1).float().contiguous()
    mask = torch.addcmul(zcoef[:, 1].view(1, -1, 1), z, zcoef[:, 0].view(1, -1, 1)) > 0
    gg = torch.where(mask, da, torch.zeros_like(da)).double()
    zc = zcoef.double()
    zhat = (zd - zc[:, 2].view(1, -1, 1)) * zc[:, 3].view(1, -1, 1)
    s0, s1 = gg.sum((0, 2)), (gg * zhat).sum((0, 2))
    n = b * p
    dz = (gamma.double() * zc[:, 3]).view(1, -1, 1) * (gg - (s0 / n).view(1, -1, 1) - zhat * (s1 / n).view(1, -1, 1))
    want = torch.bmm(dz, x.double().transpose(1, 2)).sum(0)
    part = torch.stack([s0, s1], -1).view(cout, 1, 2).float().contiguous()
    dgamma, dbeta = torch.empty(cout, device=_dev()), torch.empty(cout, device=_dev())
    bnb = hip.pw_bnb_coef(part, zcoef, gamma, float(n), dgamma, dbeta)
    dw = torch.empty(cout, cin, device=_dev())
    hip.conv_wgrad(da, x, dw, bn_z=z, bnb=bnb)
    assert (dw.double() - want).abs().max().item() < 1e-4 * want.abs().max().item()
    torch.testing.assert_close(dgamma.double(), s1, rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(dbeta.double(), s0, rtol=1e-5, atol=1e-4)


def _sa_module(c_in, mlp, ns):
    from nesie_amd.mmdet3d_ops import PointSAModule
    torch.manual_seed(0)
    return PointSAModule(mlp_channels=[c_in] + mlp, num_point=64, radius=0.4, num_sample=ns,
                         normalize_xyz=True).to(_dev())


def test_sa1_rebuilt_first_activation_matches_the_stored_form_and_the_module_path():
    """Round 5: 4 -> 64 -> 64 -> 128 with the first layer's output never stored (fused_mlp.SA1_K4):
    same output, running statistics and all nine parameter gradients as the stored form and as the
    module-by-module path; bitwise equal on a repeat."""
    from nesie_amd.mmdet3d_ops import fused_mlp
    sa = _sa_module(1, [64, 64, 128], 64)
    with torch.no_grad():
        for m in sa.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(-1.0, 1.5)
                m.bias.normal_(0, 0.3)
    g = torch.Generator(device=_dev()).manual_seed(5)
    xyz = torch.rand(3, 2048, 3, device=_dev(), generator=g)
    feats = torch.rand(3, 1, 2048, device=_dev(), generator=g) * 2.5      # (a height: not centred)
    seen = []

    def run(enabled, k4):
        fused_mlp.ENABLED, fused_mlp.SA1_K4 = enabled, k4
        for p_ in sa.parameters():
            p_.grad = None
        for m in sa.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.zero_(); m.running_var.fill_(1.0)
        _, out, _ = sa(xyz, feats)
        seen.append(out.grad_fn)
        (out * torch.linspace(-1, 1, out.numel(), device=_dev()).view_as(out)).sum().backward()
        stats = [m.running_var.clone() for m in sa.modules() if isinstance(m, torch.nn.BatchNorm2d)] \
            + [m.running_mean.clone() for m in sa.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        return out.detach(), [p_.grad.clone() for p_ in sa.parameters()], stats

    keep = fused_mlp.SA1_K4
    try:
        module = run(False, False)
        stored = run(True, False)
        rebuilt = run(True, True)
        again = run(True, True)
    finally:
        fused_mlp.ENABLED, fused_mlp.SA1_K4 = True, keep
    assert torch.equal(rebuilt[0], again[0]) and all(torch.equal(a, b) for a, b in zip(rebuilt[1], again[1]))
    for want in (stored, module):
        torch.testing.assert_close(rebuilt[0], want[0], rtol=1e-4, atol=1e-4)
        for a, b in zip(rebuilt[1], want[1]):
            torch.testing.assert_close(a, b, rtol=1e-3, atol=2e-4 * max(b.abs().max().item(), 1e-3))
        for a, b in zip(rebuilt[2], want[2]):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)


def test_k4_fused_weight_gradient_and_input_gradient_reductions_match_the_two_launches():
    """nesie_pw_wgrad_bn_backward_k4_fused against nesie_pw_wgrad_bn_backward_k4 followed by
    nesie_pw_dgrad_bn_reduce_k4: the same weight gradient bit for bit, dgamma / dbeta bit for bit, the
    input gradient's reductions equal to rounding (other slots), and da left untouched."""
    hip = _hip()
    g = torch.Generator(device=_dev()).manual_seed(21)
    nb, p = 3, 16384
    x4 = torch.randn(nb, 4, p, device=_dev(), generator=g) * 0.5
    x4[:, 3] = x4[:, 3].abs() + 0.5
    w0 = torch.randn(64, 4, device=_dev(), generator=g) * 0.5
    w1 = torch.randn(64, 64, device=_dev(), generator=g) * 0.2
    da = torch.randn(nb, 64, p, device=_dev(), generator=g)
    z1 = torch.randn(nb, 64, p, device=_dev(), generator=g)
    gamma1 = torch.rand(64, device=_dev(), generator=g) + 0.5

    def coef_of(z, gamma, beta):
        zd = z.double()
        mean, invstd = zd.mean((0, 2)), (zd.var((0, 2), unbiased=False) + 1e-5).rsqrt()
        scale = gamma.double() * invstd
        return torch.stack([scale, beta.double() - mean * scale, mean, invstd], -1).float().contiguous()
    z0 = torch.einsum('cj,njp->ncp', w0, x4)
    coef0 = coef_of(z0, torch.rand(64, device=_dev(), generator=g) + 0.5, torch.randn(64, device=_dev(), generator=g) * 0.3)
    coef1 = coef_of(z1, gamma1, torch.randn(64, device=_dev(), generator=g) * 0.3)
    zc = coef1.double()
    gg = torch.where(torch.addcmul(coef1[:, 1].view(1, -1, 1), z1, coef1[:, 0].view(1, -1, 1)) > 0, da, torch.zeros_like(da)).double()
    zhat = (z1.double() - zc[:, 2].view(1, -1, 1)) * zc[:, 3].view(1, -1, 1)
    part1 = torch.stack([gg.sum((0, 2)), (gg * zhat).sum((0, 2))], -1).view(64, 1, 2).float().contiguous()

    def two():
        d = da.clone()
        dw, dg, db = (torch.empty(64, 64, device=_dev()), torch.empty(64, device=_dev()), torch.empty(64, device=_dev()))
        hip.pw_wgrad_bn_backward_k4(d, z1, coef1, gamma1, part1, x4, w0, coef0, dw, dg, db)
        part, gpart = hip.pw_dgrad_bn_reduce_k4(d, w1.t(), x4, w0, coef0)
        return dw, dg, db, part.double().sum(1), gpart.double().sum(1)

    def one():
        d = da.clone()
        dw, dg, db = (torch.empty(64, 64, device=_dev()), torch.empty(64, device=_dev()), torch.empty(64, device=_dev()))
        part, gpart = hip.pw_wgrad_bn_backward_k4_fused(d, z1, coef1, gamma1, part1, x4, w0, coef0, w1, dw, dg, db)
        assert torch.equal(d, da)
        return dw, dg, db, part.double().sum(1), gpart.double().sum(1)
    a, b, b2 = two(), one(), one()
    for u, v in zip(b, b2):
        assert torch.equal(u, v)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for u, v in zip(a[3:], b[3:]):
        torch.testing.assert_close(v, u, rtol=1e-4, atol=1e-4 * u.abs().max().item())


def test_k4_statistics_from_input_moments_against_float64():
    """nesie_k4_moments + nesie_k4_stat_finalize: BatchNorm coefficients and running statistics of
    W0 . X4 without forming it, against float64 statistics of the product."""
    hip = _hip()
    g = torch.Generator(device=_dev()).manual_seed(3)
    nb, p = 3, 4096
    x4 = torch.randn(nb, 4, p, device=_dev(), generator=g)
    x4[:, 3] = x4[:, 3].abs() * 0.7 + 1.3
    w0 = torch.randn(64, 4, device=_dev(), generator=g)
    gamma = torch.randn(64, device=_dev(), generator=g)
    beta = torch.randn(64, device=_dev(), generator=g)
    rm, rv = torch.zeros(64, device=_dev()), torch.ones(64, device=_dev())
    coef = torch.empty(64, 4, device=_dev())
    hip.k4_stat_finalize(hip.k4_moments(x4), w0, gamma, beta, rm, rv, 0.1, 1e-5, float(nb * p), coef)
    z = torch.einsum('cj,njp->ncp', w0.double(), x4.double())
    mean, var = z.mean((0, 2)), z.var((0, 2), unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    want = torch.stack([gamma.double() * invstd, beta.double() - mean * gamma.double() * invstd, mean, invstd], -1)
    torch.testing.assert_close(coef.double(), want, rtol=2e-5, atol=2e-5)
    n = nb * p
    torch.testing.assert_close(rm.double(), 0.1 * mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rv.double(), 0.9 + 0.1 * var * n / (n - 1), rtol=1e-5, atol=1e-6)


def test_k4_first_layer_weight_gradient_from_reductions_against_float64():
    """nesie_pw_dgrad_bn_reduce_k4 + nesie_pw_bnb_coef + nesie_k4_first_layer_wgrad against the
    float64 chain  dA0 = W1^T dZ1,  dZ0 = bn_relu_backward(dA0, Z0),  dW0 = dZ0 X4^T  at SA1's
    positions-per-scene with a biased fourth input row."""
    hip = _hip()
    g = torch.Generator(device=_dev()).manual_seed(11)
    nb, p = 2, 8192
    x4 = torch.randn(nb, 4, p, device=_dev(), generator=g)
    x4[:, 3] = x4[:, 3].abs() + 1.0
    w0 = torch.randn(64, 4, device=_dev(), generator=g) * 0.5
    w1 = torch.randn(64, 64, device=_dev(), generator=g) * 0.2
    dz1 = torch.randn(nb, 64, p, device=_dev(), generator=g)
    gamma = torch.randn(64, device=_dev(), generator=g)
    beta = torch.randn(64, device=_dev(), generator=g) * 0.3
    # Z0 as the kernels rebuild it (one fma chain), then everything else in float64
    z0 = torch.addcmul(torch.addcmul(torch.addcmul(w0[:, 0].view(1, -1, 1) * x4[:, 0:1], w0[:, 1].view(1, -1, 1), x4[:, 1:2]),
                                     w0[:, 2].view(1, -1, 1), x4[:, 2:3]), w0[:, 3].view(1, -1, 1), x4[:, 3:4])
    zd = z0.double()
    mean, invstd = zd.mean((0, 2)), (zd.var((0, 2), unbiased=False) + 1e-5).rsqrt()
    scale = gamma.double() * invstd
    zcoef = torch.stack([scale, beta.double() - mean * scale, mean, invstd], -1).float().contiguous()
    zc = zcoef.double()
    da0 = torch.matmul(w1.double().t(), dz1.double())
    act = zd * zc[:, 0].view(1, -1, 1) + zc[:, 1].view(1, -1, 1)
    gg = torch.where(act > 0, da0, torch.zeros_like(da0))
    zhat = (zd - zc[:, 2].view(1, -1, 1)) * zc[:, 3].view(1, -1, 1)
    n = nb * p
    s0, s1 = gg.sum((0, 2)), (gg * zhat).sum((0, 2))
    dz0 = (gamma.double() * zc[:, 3]).view(1, -1, 1) * (gg - (s0 / n).view(1, -1, 1) - zhat * (s1 / n).view(1, -1, 1))
    want = torch.bmm(dz0, x4.double().transpose(1, 2)).sum(0)
    # knife-edge positions (|act| tiny) may fall on the other side in fp32: exclude none, allow for them
    part, g_part = hip.pw_dgrad_bn_reduce_k4(dz1, w1.t(), x4, w0, zcoef)
    dgamma, dbeta = torch.empty(64, device=_dev()), torch.empty(64, device=_dev())
    bnb = hip.pw_bnb_coef(part, zcoef, gamma, float(n), dgamma, dbeta)
    dw0 = torch.empty(64, 4, device=_dev())
    hip.k4_first_layer_wgrad(hip.k4_moments(x4), w0, bnb, g_part, dw0)
    torch.cuda.synchronize()
    assert (dw0.double() - want).abs().max().item() < 2e-4 * want.abs().max().item()
    torch.testing.assert_close(dbeta.double(), s0, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(dgamma.double(), s1, rtol=1e-4, atol=1e-3)
    gx = torch.einsum('ncp,njp->cj', gg, x4.double())
    torch.testing.assert_close(g_part.double().sum(1), gx, rtol=1e-4, atol=1e-3 * gx.abs().max().item())


@pytest.mark.parametrize('c_in,mlp,ns', [(1, [64, 64, 128], 64), (128, [128, 128, 256], 32),
                                         (256, [128, 128, 256], 16)])
def test_fused_sa_stack_matches_the_module_by_module_path(c_in, mlp, ns):
    from nesie_amd.mmdet3d_ops import fused_mlp
    sa = _sa_module(c_in, mlp, ns)
    with torch.no_grad():
        for m in sa.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(-1.0, 1.5)
                m.bias.normal_(0, 0.3)
    g = torch.Generator(device=_dev()).manual_seed(ns)
    xyz = torch.rand(2, 512, 3, device=_dev(), generator=g)
    feats = torch.randn(2, c_in, 512, device=_dev(), generator=g)

    def run(enabled):
        fused_mlp.ENABLED = enabled
        for p_ in sa.parameters():
            p_.grad = None
        for m in sa.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.zero_(); m.running_var.fill_(1.0)
        f = feats.clone().requires_grad_(True)
        _, out, _ = sa(xyz, f)
        (out * torch.linspace(-1, 1, out.numel(), device=_dev()).view_as(out)).sum().backward()
        stats = [m.running_var.clone() for m in sa.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        return out.detach(), f.grad, [p_.grad.clone() for p_ in sa.parameters()], stats

    try:
        want = run(False)
        got = run(True)
    finally:
        fused_mlp.ENABLED = True
    torch.testing.assert_close(got[0], want[0], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(got[1], want[1], rtol=1e-3, atol=1e-4 * want[1].abs().max().item())
    for a, b in zip(got[2], want[2]):
        torch.testing.assert_close(a, b, rtol=1e-3, atol=2e-4 * max(b.abs().max().item(), 1e-3))
    for a, b in zip(got[3], want[3]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)


def test_fused_sa_stack_carries_the_coordinate_gradient():
    """Vote aggregation groups coordinates the network predicted: the gradient of the grouped
    (x, y, z) channels must reach them (the backbone levels group input coordinates and skip it)."""
    from nesie_amd.mmdet3d_ops import fused_mlp
    sa = _sa_module(256, [128, 128, 128], 16)
    g = torch.Generator(device=_dev()).manual_seed(5)
    xyz0 = torch.rand(2, 512, 3, device=_dev(), generator=g)
    feats = torch.randn(2, 256, 512, device=_dev(), generator=g)

    def run(enabled):
        fused_mlp.ENABLED = enabled
        for p_ in sa.parameters():
            p_.grad = None
        xyz = xyz0.clone().requires_grad_(True)
        f = feats.clone().requires_grad_(True)
        new_xyz, out, _ = sa(xyz * 1.0, f)
        (out * torch.linspace(-1, 1, out.numel(), device=_dev()).view_as(out)).sum().backward()
        return xyz.grad, f.grad, sa.mlps[0][0].conv.weight.grad.clone()

    try:
        want = run(False)
        got = run(True)
    finally:
        fused_mlp.ENABLED = True
    assert want[0].abs().max().item() > 0
    for a, b in zip(got, want):
        torch.testing.assert_close(a, b, rtol=1e-3, atol=2e-4 * b.abs().max().item())


@pytest.mark.parametrize('S,G', [(6, 16), (1, 64)])
def test_fused_mini_pointnets_match_the_module_by_module_path(S, G):
    from nesie_amd.mmdet3d_ops import fused_mlp
    from nesie_amd.votenet.side_pooling import MiniPointNet, grouped_mini_pointnets
    torch.manual_seed(1)
    nets = [MiniPointNet(259, 128).to(_dev()) for _ in range(S)]
    with torch.no_grad():
        for n in nets:
            for m in n.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.uniform_(-1.0, 1.5)
                    m.bias.normal_(0, 0.3)
    B, H, K = 2, 256, 64
    g = torch.Generator(device=_dev()).manual_seed(G)
    c0 = torch.randn(B, S, H, K, G, device=_dev(), generator=g)
    # (sum, sum of squares) partials of c0 in the blend kernel's layout (C, slices, 2)
    flat = c0.permute(1, 2, 0, 3, 4).reshape(S * H, B * K * G // 64, 64).double()
    part = torch.stack([flat.sum(-1), (flat ** 2).sum(-1)], -1).float().contiguous()
    params = [p_ for n in nets for p_ in n.parameters()]

    def run(enabled):
        fused_mlp.ENABLED = enabled
        for p_ in params:
            p_.grad = None
        x = c0.clone().requires_grad_(True)
        out = grouped_mini_pointnets(nets, x, c0_stats=part)
        (out * torch.linspace(-1, 1, out.numel(), device=_dev()).view_as(out)).sum().backward()
        return out.detach(), x.grad, [None if p_.grad is None else p_.grad.clone() for p_ in params]

    try:
        want = run(False)
        got = run(True)
    finally:
        fused_mlp.ENABLED = True
    torch.testing.assert_close(got[0], want[0], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(got[1], want[1], rtol=1e-3, atol=2e-4 * want[1].abs().max().item())
    # conv3's bias passes a per-channel constant into a BatchNorm: its true gradient is zero and
    # both evaluations return rounding noise there, so the floor is set by the other gradients
    scale = max(b.abs().max().item() for b in want[2] if b is not None)
    for a, b in zip(got[2], want[2]):
        assert (a is None) == (b is None)
        if a is not None:
            torch.testing.assert_close(a, b, rtol=1e-3, atol=3e-4 * max(b.abs().max().item(), 1e-2 * scale))


@pytest.mark.parametrize('S,G,second_consumer', [(6, 16, False), (1, 64, False), (6, 16, True)])
def test_deferred_blend_into_fused_mini_pointnets(S, G, second_consumer):
    """side_pooling.DeferredBlendConv -> fused_mlp.BlendMiniHeadFn (blend conv + first norm + second
    conv as ONE autograd node, the norm backward on the blend backward's tile load) against the
    materialised BlendConv tensor through the module-by-module MiniPointNets: outputs, running
    statistics, gradients of the table, the coordinate weights and every parameter.
    ``second_consumer``: the conv output ALSO feeds another loss term -- through ``materialize()``,
    the only way to reach it; the fused node's internal tensors cannot be consumed, hooked or
    accumulated into (the hand-over this replaces travelled on the gradient of an autograd tensor
    and a second consumer silently corrupted it)."""
    import copy
    from nesie_amd.mmdet3d_ops import fused_mlp
    from nesie_amd.votenet.side_pooling import DeferredBlendConv, MiniPointNet, grouped_mini_pointnets
    torch.manual_seed(2)
    dev = _dev()
    nets = [MiniPointNet(259, 128).to(dev) for _ in range(S)]
    with torch.no_grad():
        for n in nets:
            for m in n.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.uniform_(-1.0, 1.5)
                    m.bias.normal_(0, 0.3)
    B, H, K, M = 2, 256, 64, 128
    g = torch.Generator(device=dev).manual_seed(G + S)
    table0 = torch.randn(B, M, S * H, device=dev, generator=g) * 0.5
    wx0 = torch.randn(S, H, 3, device=dev, generator=g)
    n = K * S * G
    idx = torch.randint(0, M, (B, n, 3), device=dev, generator=g, dtype=torch.int32)
    w = torch.rand(B, n, 3, device=dev, generator=g) + 0.05
    w = (w / w.sum(-1, keepdim=True)).contiguous()
    rel = torch.randn(B, n, 3, device=dev, generator=g) * 0.3
    pick = torch.linspace(-1, 1, B * S * 128 * K, device=dev)
    seen = []

    def run(fused):
        local = copy.deepcopy(nets)
        params = [p_ for net in local for p_ in net.parameters()]
        table, wx = table0.clone().requires_grad_(True), wx0.clone().requires_grad_(True)
        d = DeferredBlendConv(table, wx, idx, w, rel, S, G, K)
        fused_mlp.ENABLED = fused
        try:
            if fused:
                assert d.has_stats
                out = grouped_mini_pointnets(local, d)
            else:
                c0, stats = d.materialize()
                out = grouped_mini_pointnets(local, c0, c0_stats=stats)
            loss = (out * pick.view_as(out)).sum()
            if second_consumer:
                loss = loss + d.materialize()[0].square().mean() * 3.0
            loss.backward()
        finally:
            fused_mlp.ENABLED = True
        stats_ = [b_.clone() for net in local for b_ in net.buffers()]
        return out.detach(), table.grad, wx.grad, [None if p_.grad is None else p_.grad.clone() for p_ in params], stats_

    real = fused_mlp.BlendMiniHeadFn.forward

    def spy(*a, **k):
        seen.append(True)
        return real(*a, **k)
    fused_mlp.BlendMiniHeadFn.forward = staticmethod(spy)
    try:
        want = run(False)
        assert not seen
        got = run(True)
        assert seen, 'the deferred conv did not go through BlendMiniHeadFn'
    finally:
        fused_mlp.BlendMiniHeadFn.forward = staticmethod(real)
    torch.testing.assert_close(got[0], want[0], rtol=1e-4, atol=1e-4)
    for a, b in ((got[1], want[1]), (got[2], want[2])):   # (two fp32 orders of the same sums)
        torch.testing.assert_close(a, b, rtol=1e-3, atol=6e-4 * b.abs().max().item())
    scale = max(b.abs().max().item() for b in want[3] if b is not None)
    kinks = 0
    for a, b in zip(got[3], want[3]):
        assert (a is None) == (b is None)
        if a is not None:
            # ReLU knife-edges: the two legs form the second norm's input with different roundings
            # (W_g g + bias on the layer kernel vs ATen's matmuls), and of the ~3 M normalised
            # activations a few sit within rounding of zero: where the legs disagree on such a mask
            # ONE term of a channel's sum appears or vanishes.  A handful of single elements may
            # therefore differ by the size of one term; everything else must agree closely.
            tol = 1e-3 * b.abs() + 3e-4 * max(b.abs().max().item(), 1e-2 * scale)
            off = (a - b).abs() > tol
            assert int(off.sum()) <= 3 and float((a - b).abs().max()) <= 0.15 * scale, \
                (int(off.sum()), float((a - b).abs().max()), scale)
            kinks += int(off.sum())
    assert kinks <= 6, kinks
    for a, b in zip(got[4], want[4]):
        torch.testing.assert_close(a.float(), b.float(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('c_in,mlp,ns', [(1, [64, 64, 128], 64), (256, [128, 128, 256], 16)])
def test_eval_mode_sa_stack_matches_the_module_by_module_path(c_in, mlp, ns):
    from nesie_amd.mmdet3d_ops import fused_mlp
    sa = _sa_module(c_in, mlp, ns)
    with torch.no_grad():
        for m in sa.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(-1.0, 1.5)
                m.bias.normal_(0, 0.3)
                m.running_mean.normal_(0, 0.5)
                m.running_var.uniform_(0.5, 2.0)
    sa.eval()
    g = torch.Generator(device=_dev()).manual_seed(ns)
    xyz = torch.rand(2, 512, 3, device=_dev(), generator=g)
    feats = torch.randn(2, c_in, 512, device=_dev(), generator=g)
    outs = []
    try:
        for enabled in (False, True):
            fused_mlp.ENABLED = enabled
            with torch.no_grad():
                outs.append(sa(xyz, feats)[1])
    finally:
        fused_mlp.ENABLED = True
    torch.testing.assert_close(outs[1], outs[0], rtol=1e-4, atol=1e-4)
    # a changed running statistic must invalidate the cached coefficients
    with torch.no_grad():
        for m in sa.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.add_(0.25)
        again = sa(xyz, feats)[1]
    assert (again - outs[1]).abs().max().item() > 1e-3


@pytest.mark.parametrize('captured', [False, True])
def test_ragged_220_row_output_layer_stays_inside_its_buffers(captured):
    """The prediction head's three output convolutions stacked as ONE 220-row layer (20 + 198 + 2,
    K = 128, 8 x 256 proposals) -- the form whose captured step aborted on replay in round 3 without
    a root cause (bbox_module._fused).  Every buffer the launch touches sits between canaries:
    input, weights, bias, output, and the input-gradient / weight-gradient launches of the same
    shape; eager and inside a hipGraph replayed three times.  No canary may change and the values
    must equal float64."""
    hip = _hip()
    dev = _dev()
    nb, k, cout, p = 8, 128, 220, 256
    g = torch.Generator(device=dev).manual_seed(220)
    CAN = 4096

    def guarded(*shape):
        n = 1
        for s_ in shape:
            n *= s_
        buf = torch.full((n + 2 * CAN,), 12345.678, device=dev)
        return buf, buf[CAN:CAN + n].view(*shape)
    xb, x = guarded(nb, k, p)
    wb, w = guarded(1, cout, k)
    bb, bias = guarded(cout)
    yb, y = guarded(nb, cout, p)
    dxb, dx = guarded(nb, k, p)
    dwb, dw = guarded(1, cout, k)
    x.copy_(torch.randn(nb, k, p, device=dev, generator=g))
    w.copy_(torch.randn(1, cout, k, device=dev, generator=g) / k ** 0.5)
    bias.copy_(torch.randn(cout, device=dev, generator=g))
    dy = torch.randn(nb, cout, p, device=dev, generator=g)

    def launches():
        hip.pw_layer_forward(x, w, y=y, bias=bias)
        hip.pw_layer_forward(dy, w.transpose(1, 2), y=dx)
        hip.pw_wgrad(dy, x, dw, x_coef=None)
    launches()
    torch.cuda.synchronize()
    if captured:
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            launches()
        for _ in range(3):
            y.zero_(); dx.zero_(); dw.zero_()
            gr.replay()
        torch.cuda.synchronize()
    for name, buf, view in (('x', xb, x), ('w', wb, w), ('bias', bb, bias), ('y', yb, y), ('dx', dxb, dx), ('dw', dwb, dw)):
        n = view.numel()
        assert bool((buf[:CAN] == 12345.678).all()) and bool((buf[CAN + n:] == 12345.678).all()), name
    ref = torch.matmul(w[0].double(), x.double()) + bias.double().view(1, -1, 1)
    assert (y.double() - ref).abs().max().item() < 3e-5
    ref_dx = torch.matmul(w[0].double().t(), dy.double())
    assert (dx.double() - ref_dx).abs().max().item() < 3e-5 * max(1.0, ref_dx.abs().max().item())
    ref_dw = torch.bmm(dy.double(), x.double().transpose(1, 2)).sum(0)
    assert (dw[0].double() - ref_dw).abs().max().item() < 1e-4 * ref_dw.abs().max().item()


@pytest.mark.parametrize('nb,m,ns,k,c', [(3, 40, 64, 64, 128), (2, 96, 32, 64, 128), (2, 132, 16, 64, 128),
                                         (1, 10, 64, 64, 128)])
def test_pooled_tail_backward_without_the_dense_tensors_matches_float64(nb, m, ns, k, c):
    """csrc/pool_tail.hip: conv (k -> c) + training BatchNorm + ReLU + max over ns from what the
    forward kept (pooled, arg-max, raw extremum) against autograd in float64 on the literal chain,
    negative norm scales included; and bit-identical on a second run (no atomics)."""
    be, dev = _hip(), _dev()
    p = m * ns
    assert be.pool_tail_supported(k, c, p, ns)
    g = torch.Generator(device=dev).manual_seed(nb * 1000 + ns)
    z_prev = torch.randn(nb, k, p, device=dev, generator=g)
    coef_prev = torch.zeros(k, 4, device=dev)
    coef_prev[:, 0] = torch.empty(k, device=dev).uniform_(-1.2, 1.5, generator=g)   # scale
    coef_prev[:, 1] = torch.randn(k, device=dev, generator=g) * 0.3                  # bias
    coef_prev[:, 2] = torch.randn(k, device=dev, generator=g) * 0.1                  # mean
    coef_prev[:, 3] = torch.empty(k, device=dev).uniform_(0.5, 2.0, generator=g)     # invstd
    w = torch.randn(c, k, device=dev, generator=g) * 0.2
    gamma = torch.empty(c, device=dev).uniform_(-1.0, 1.5, generator=g)
    beta = torch.randn(c, device=dev, generator=g) * 0.3
    gpool = torch.randn(nb, c, m, device=dev, generator=g)
    eps = 1e-5

    # float64 literal chain (the previous layer's activation is an input here: A = relu(s z + b))
    zp = z_prev.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    a = torch.relu(zp * coef_prev[:, 0].double().view(1, k, 1) + coef_prev[:, 1].double().view(1, k, 1))
    a.retain_grad()
    z = torch.einsum('ck,nkp->ncp', wd, a)
    mean, var = z.mean(dim=(0, 2)), z.var(dim=(0, 2), unbiased=False)
    invstd = (var + eps).rsqrt()
    y = torch.relu((z - mean.view(1, c, 1)) * (invstd * gd).view(1, c, 1) + bd.view(1, c, 1))
    pooled64 = y.view(nb, c, m, ns).max(dim=-1).values
    (pooled64 * gpool.double()).sum().backward()

    # what the forward keeps: folded coefficients from the float64 statistics, pooled, arg-max, z*
    coef = torch.stack([invstd * gd, bd - mean * invstd * gd, mean, invstd], 1).detach().float().contiguous()
    zf = z.detach().float().view(nb, c, m, ns)
    sel = torch.where((coef[:, 0] >= 0).view(1, c, 1, 1), zf, -zf)
    argmax = sel.argmax(dim=-1)
    zstar = zf.gather(-1, argmax.unsqueeze(-1)).squeeze(-1).contiguous()
    pooled = torch.relu(zstar * coef[:, 0].view(1, c, 1) + coef[:, 1].view(1, c, 1)).contiguous()
    argmax = argmax.to(torch.uint8).contiguous()

    def run():
        dgamma, dbeta, dw = (torch.full((c,), float('nan'), device=dev), torch.full((c,), float('nan'), device=dev),
                             torch.full((c, k), float('nan'), device=dev))
        da, part = be.pool_tail_backward(gpool, pooled, zstar, argmax, coef, gamma, w, z_prev, coef_prev, ns,
                                         dgamma, dbeta, dw=dw)
        torch.cuda.synchronize()
        return da, part, dgamma, dbeta, dw

    da, part, dgamma, dbeta, dw = run()
    scale = lambda t: t.abs().max().item()  # noqa: E731
    torch.testing.assert_close(dgamma.double(), gd.grad, rtol=1e-4, atol=1e-5 * scale(gd.grad))
    torch.testing.assert_close(dbeta.double(), bd.grad, rtol=1e-4, atol=1e-5 * scale(bd.grad))
    torch.testing.assert_close(da.double(), a.grad, rtol=1e-3, atol=2e-5 * scale(a.grad))
    torch.testing.assert_close(dw.double(), wd.grad, rtol=1e-3, atol=1e-4 * scale(wd.grad))
    # the reduction of the previous layer's norm backward: sum g, sum g zhat with g = dA [A > 0]
    mask = (a.detach() > 0).double()
    g2 = a.grad * mask
    zhat = (z_prev.double() - coef_prev[:, 2].double().view(1, k, 1)) * coef_prev[:, 3].double().view(1, k, 1)
    sums = part.double().sum(dim=1)
    want0, want1 = g2.sum(dim=(0, 2)), (g2 * zhat).sum(dim=(0, 2))
    torch.testing.assert_close(sums[:, 0], want0, rtol=1e-3, atol=1e-4 * scale(want0))
    torch.testing.assert_close(sums[:, 1], want1, rtol=1e-3, atol=1e-4 * scale(want1))
    again = run()
    for t0, t1 in zip((da, part, dgamma, dbeta, dw), again):
        assert torch.equal(t0, t1)


def test_persistent_grids_sized_for_fewer_cus_give_the_same_layer():
    """``HipKernels.cu_budget`` (nesie_set_cu_count): the layer / weight-gradient launches sized for 248
    CUs -- what bench.py does while the next batch's sampling kernels hold one CU per XCD -- walk the
    same tiles with fewer workgroups: outputs bit-identical, statistics and the weight gradient equal
    to rounding, fewer statistic slots; the setting is restored on exit."""
    hip, dev = _hip(), _dev()
    from nesie_amd.kernels import HipKernels
    g = torch.Generator(device=dev).manual_seed(7)
    nb, k, cout, p = 8, 128, 256, 32768
    x = torch.randn(nb, k, p, device=dev, generator=g)
    w = torch.randn(1, cout, k, device=dev, generator=g) / k ** 0.5
    gamma, beta = torch.rand(cout, device=dev, generator=g) + 0.5, torch.randn(cout, device=dev, generator=g)

    def run():
        slots = hip.pw_stat_slots(nb, 1, k, cout, p)
        y, part = torch.empty(nb, cout, p, device=dev), torch.zeros(1, slots, cout, 4, device=dev)
        hip.pw_layer_forward(x, w, y=y, stat_part=part)
        coef = torch.empty(cout, 4, device=dev)
        hip.pw_stats_finalize(part, gamma, beta, None, None, 0.1, 1e-5, coef)
        dw = torch.empty(1, cout, k, device=dev)
        hip.pw_wgrad(y, x, dw)
        torch.cuda.synchronize()
        return slots, y, coef, dw

    full = run()
    assert _lib_cu_count() == 256
    with HipKernels.cu_budget(248):
        assert _lib_cu_count() == 248
        less = run()
    assert _lib_cu_count() == 256
    assert less[0] < full[0]
    assert torch.equal(less[1], full[1])
    torch.testing.assert_close(less[2], full[2], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(less[3], full[3], rtol=1e-4, atol=1e-4 * full[3].abs().max().item())


def _lib_cu_count():
    from nesie_amd import _lib
    return _lib.load().nesie_get_cu_count()


def test_deferred_weight_gradient_reductions_equal_the_immediate_ones_bit_for_bit():
    """nesie_pw_wgrad_deferred + nesie_pw_wgrad_flush_deferred: several weight gradients (whole-product
    launches, a tiled split-K launch, a column-blocked one that must reduce at once) leave their
    partials pending and ONE launch finishes them -- every element summed as its own reduce kernel sums
    it: bitwise equal to the immediate form.  Outside a window ``final`` changes nothing."""
    from nesie_amd.kernels import HipKernels
    hip = _hip()
    g = torch.Generator(device=_dev()).manual_seed(11)
    shapes = [(4, 2, 128, 128, 256), (2, 1, 256, 128, 512), (8, 1, 256, 256, 1024), (6, 3, 128, 256, 256),
              (2, 1, 256, 512, 128), (2, 1, 64, 64, 1024), (12, 6, 128, 166, 512),
              (2, 1, 256, 256, 32768)]            # (the last one: two column blocks over many positions)
    assert not hip.pw_wgrad_tiled(2, 1, 256, 256, 32768)
    cases = []
    for nb, ng, co, ci, p in shapes:
        dy = torch.randn(nb, co, p, device=_dev(), generator=g)
        x = torch.randn(nb, ci, p, device=_dev(), generator=g)
        coef = torch.rand(ng * ci, 4, device=_dev(), generator=g) + 0.5
        coef[:, 1] -= 1.0
        want = torch.empty(ng, co, ci, device=_dev())
        hip.pw_wgrad(dy, x, want, ng=ng, x_coef=coef, final=True)      # no window open: immediate
        cases.append((dy, x, coef, ng, want))
    assert _lib_pending() == 0
    HipKernels.begin_deferred_reductions()
    try:
        got = []
        for dy, x, coef, ng, want in cases:
            dw = torch.full_like(want, float('nan'))
            hip.pw_wgrad(dy, x, dw, ng=ng, x_coef=coef, final=True)
            got.append(dw)
        pending = _lib_pending()
        assert 0 < pending < len(cases)          # (the column-blocked shape did not wait)
        assert any(torch.isnan(d).any() for d in got)
        done = HipKernels.flush_deferred_reductions(close=True)
    finally:
        HipKernels._deferred = None
    torch.cuda.synchronize()
    assert len(done) == len(cases) and _lib_pending() == 0     # (python keeps every workspace of the window alive)
    for dw, (_, _, _, _, want) in zip(got, cases):
        assert torch.equal(dw, want)
    # more gradients than one table holds: the queue flushes itself and keeps going
    HipKernels.begin_deferred_reductions()
    try:
        dy, x, coef, ng, want = cases[0]
        outs = [torch.empty_like(want) for _ in range(45)]
        for dw in outs:
            hip.pw_wgrad(dy, x, dw, ng=ng, x_coef=coef, final=True)
        HipKernels.flush_deferred_reductions(close=True)
    finally:
        HipKernels._deferred = None
    torch.cuda.synchronize()
    assert all(torch.equal(dw, want) for dw in outs)


def _lib_pending():
    from nesie_amd import _lib
    return _lib.load().nesie_pw_wgrad_pending()
