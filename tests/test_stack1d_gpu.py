"""The 1-D per-seed / per-proposal chains on the layer kernel (fused_mlp.Stack1dFn) against the
module-by-module evaluation of the same modules (fused_mlp.ENABLED = False: conv, BatchNorm + ReLU
and bias one operator at a time, as the reference does): VoteModule (vote_module.py:65-74, 85-147),
ReliableConvBboxHead (reliable_conv_bbox_module.py:112-177), PointFPModule
(point_fp_module.py:31-78), the quality head's score heads (side_pooling_module.py:55-78, 318).
Outputs 1e-4 relative, running statistics, gradients of every parameter and of the input."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device('cuda:0')


def _both(run, module, inputs):
    """(outputs, input grads, parameter grads, buffers) of ``run(module, *inputs)`` with and without
    the fused chains."""
    from nesie_amd.mmdet3d_ops import fused_mlp
    out = []
    for enabled in (True, False):
        m = copy.deepcopy(module)
        xs = [x.clone().requires_grad_(x.dtype.is_floating_point) for x in inputs]
        fused_mlp.ENABLED = enabled
        try:
            ys = run(m, *xs)
            ys = ys if isinstance(ys, (tuple, list)) else (ys,)
            g = torch.Generator(device=_dev()).manual_seed(5)
            loss = sum((y * torch.randn(y.shape, device=y.device, generator=g)).sum() for y in ys)
            loss.backward()
        finally:
            fused_mlp.ENABLED = True
        out.append(([y.detach() for y in ys], [x.grad for x in xs if x.requires_grad],
                    {n: p.grad for n, p in m.named_parameters()},
                    {n: b.detach().clone() for n, b in m.named_buffers()}))
    return out


def _close(a, b, tol, what):
    scale = max(float(b.abs().max()), 1e-6)
    err = float((a - b).abs().max()) / scale
    assert err < tol, (what, err)


def _compare(fused, plain, zero_bias=()):
    for i, (a, b) in enumerate(zip(fused[0], plain[0])):
        _close(a, b, 1e-4, f'output {i}')
    for i, (a, b) in enumerate(zip(fused[1], plain[1])):
        _close(a, b, 2e-4, f'input gradient {i}')
    gmax = max(float(g.abs().max()) for g in plain[2].values() if g is not None)
    for n, gp in plain[2].items():
        gf = fused[2][n]
        if n in zero_bias:      # a conv bias in front of a norm: exactly zero, a ZERO TENSOR (not None)
            assert gf is not None and float(gf.abs().max()) == 0.0, n
            assert gp is None or float(gp.abs().max()) < 1e-4 * gmax, n   # (folded there too, or rounding noise)
            continue
        assert gf is not None, n
        _close(gf, gp, 5e-4, f'gradient of {n}')
    for n, bp in plain[3].items():
        bf = fused[3][n]
        if bp.dtype.is_floating_point:
            _close(bf, bp, 1e-5, f'buffer {n}')
        else:
            assert torch.equal(bf, bp), n


def test_vote_module_chain(hip_device):
    from nesie_amd.votenet.vote_module import VoteModule
    torch.manual_seed(0)
    vm = VoteModule(256, conv_channels=(256, 256), norm_feats=True).to(hip_device).train()
    with torch.no_grad():
        for m in vm.vote_conv:                      # non-trivial biases in front of the norms
            m.conv.bias.normal_(0, 0.3)
        vm.conv_out.bias.normal_(0, 0.1)
    g = torch.Generator(device=hip_device).manual_seed(1)
    xyz = torch.rand(4, 1024, 3, device=hip_device, generator=g)
    feats = torch.randn(4, 256, 1024, device=hip_device, generator=g)

    def run(m, f):
        pts, vf, off = m(xyz, f)
        return pts, vf, off
    fused, plain = _both(run, vm, [feats])
    _compare(fused, plain, zero_bias={'vote_conv.0.conv.bias', 'vote_conv.1.conv.bias'})


@pytest.mark.parametrize('norm_feats', [True, False])
def test_vote_module_tail_as_one_kernel_matches_the_op_chain(hip_device, norm_feats):
    """vote_module.VoteFinishFn (csrc/vote.hip: seed + offset, seed + residual features, L2
    normalisation, and one backward kernel) vs the reference's add / permute / add / norm / div
    chain under autograd (vote_module.py:106-147): outputs 1e-6, gradients of the seed features,
    the seed coordinates and every parameter 1e-5 relative, with gradients arriving at all three
    outputs."""
    import copy
    from nesie_amd.votenet import vote_module as vmod
    torch.manual_seed(0)
    vm = vmod.VoteModule(256, conv_channels=(256, 256), norm_feats=norm_feats).to(hip_device).train()
    with torch.no_grad():
        vm.conv_out.bias.normal_(0, 0.1)
    g = torch.Generator(device=hip_device).manual_seed(1)
    xyz0 = torch.rand(4, 1024, 3, device=hip_device, generator=g)
    feats0 = torch.randn(4, 256, 1024, device=hip_device, generator=g)
    go = [torch.randn(4, 1024, 3, device=hip_device, generator=g), torch.randn(4, 256, 1024, device=hip_device, generator=g),
          torch.randn(4, 3, 1024, device=hip_device, generator=g)]
    outs = []
    try:
        for fused in (False, True):
            vmod.FUSED_FINISH = fused
            m = copy.deepcopy(vm)
            xyz, feats = xyz0.clone().requires_grad_(True), feats0.clone().requires_grad_(True)
            res = m(xyz, feats)
            sum((r * w).sum() for r, w in zip(res, go)).backward()
            outs.append(([r.detach() for r in res], xyz.grad, feats.grad, {n: p.grad for n, p in m.named_parameters()}))
    finally:
        vmod.FUSED_FINISH = True
    want, got = outs
    for a, b in zip(got[0], want[0]):
        torch.testing.assert_close(a, b, rtol=2e-6, atol=2e-6)
    for a, b in ((got[1], want[1]), (got[2], want[2])):
        _close(a, b, 1e-5, 'input gradient')
    for n in want[3]:
        if want[3][n] is not None and float(want[3][n].abs().max()) > 0:
            _close(got[3][n], want[3][n], 2e-5, n)


def test_prediction_head_chain(hip_device):
    from nesie_amd.votenet import nesie_votenet_scannet_cfg
    from nesie_amd.votenet.nesie_head import NesieHead
    cfg = nesie_votenet_scannet_cfg()
    torch.manual_seed(0)
    head = NesieHead(**cfg['bbox_head'], train_cfg=cfg['train_cfg'], test_cfg=cfg['test_cfg'])
    pred = head.conv_pred.to(hip_device).train()
    with torch.no_grad():
        for c in (pred.conv_cls, pred.conv_bbox, pred.conv_heading):
            c.weight.normal_(0, 0.05)
            c.bias.normal_(0, 0.1)
    g = torch.Generator(device=hip_device).manual_seed(2)
    feats = torch.randn(4, 128, 256, device=hip_device, generator=g)
    fused, plain = _both(lambda m, f: m(f), pred, [feats])
    assert fused[0][0].shape[1] == 20 and fused[0][1].shape[1] == plain[0][1].shape[1]
    _compare(fused, plain, zero_bias={'shared_convs.layer0.conv.bias', 'shared_convs.layer1.conv.bias'})


def test_feature_propagation_chain(hip_device):
    from nesie_amd.mmdet3d_ops import PointFPModule
    torch.manual_seed(0)
    fp = PointFPModule(mlp_channels=[512, 256, 256]).to(hip_device).train()
    g = torch.Generator(device=hip_device).manual_seed(3)
    target = torch.rand(2, 1024, 3, device=hip_device, generator=g)
    source = torch.rand(2, 512, 3, device=hip_device, generator=g)
    tf = torch.randn(2, 256, 1024, device=hip_device, generator=g)
    sf = torch.randn(2, 256, 512, device=hip_device, generator=g)
    fused, plain = _both(lambda m, a, b: m(target, source, a, b), fp, [tf, sf])
    _compare(fused, plain)


def test_score_heads_chain(hip_device):
    from torch import nn
    from nesie_amd.votenet import side_pooling as sp
    torch.manual_seed(0)

    class Heads(nn.Module):
        def __init__(self):
            super().__init__()
            self.h = nn.ModuleList([sp._score_head(166, 18) for _ in range(6)])
            self.one = sp._score_head(128, 18)

        def forward(self, x, z):
            assert sp.heads_batchable(list(self.h), x[:, 0])
            return (sp.batched_heads(list(self.h), x),
                    sp.batched_heads([self.one], z.unsqueeze(1)).squeeze(1))
    heads = Heads().to(hip_device).train()
    with torch.no_grad():
        for p in heads.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    g = torch.Generator(device=hip_device).manual_seed(4)
    x = torch.randn(4, 6, 166, 512, device=hip_device, generator=g)
    z = torch.randn(4, 128, 512, device=hip_device, generator=g)
    fused, plain = _both(lambda m, a, b: m(a, b), heads, [x, z])
    zero = {n for n, _ in heads.named_parameters() if n.endswith('.0.bias') or n.endswith('.3.bias')}
    _compare(fused, plain, zero_bias=zero)


def test_the_chains_are_taken(hip_device):
    """The product path really runs the fused chain (no silent module-by-module evaluation)."""
    from nesie_amd.mmdet3d_ops import fused_mlp
    from nesie_amd.votenet.vote_module import VoteModule
    calls = []
    real = fused_mlp.Stack1dFn.apply
    fused_mlp.Stack1dFn.apply = lambda *a: (calls.append(1), real(*a))[1]
    try:
        vm = VoteModule(256, conv_channels=(256, 256)).to(hip_device).train()
        vm(torch.rand(2, 1024, 3, device=hip_device), torch.randn(2, 256, 1024, device=hip_device))
    finally:
        fused_mlp.Stack1dFn.apply = real
    assert calls == [1]
