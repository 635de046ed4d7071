"""Side-aware quality head (``mmdet3d/models/dense_heads/side_pooling_module.py:10-370``).

Per proposal: a 4x4x4 grid in the box and 6 face grids (4x4 each) are rotated /
translated into the scene, each grid point gets [relative xyz (3), inverse-distance
3-NN blend of the detached seed features (256)], a MiniPointNet pools each grid to a
128-vector, and small heads emit per-class side scores (6 faces) and IoU scores.

The 3-NN search and the 3-tap feature blend run on the native kernels
(``three_nn`` / ``three_interpolate``); the reference reaches the same numbers with
mmcv's three_nn plus an ``index_select`` that materialises (B, K*G*3, 256) floats.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import grad_slots
from ..kernels import backend_for
from ..mmdet3d_ops import blend_conv, blend_conv_bn, three_interpolate_segmented, three_nn
from ..mmdet3d_ops.pool import group_max_pool, group_max_pool_shared
from ..mmdet3d_ops.norm import FusedBNReLU1d, FusedBNReLU2d
from ..mmdet3d_ops.pointnet_modules import PointwiseConv1d, PointwiseConv2d, pointwise_conv


def rot_gpu(t):
    """Rotation about the upright axis, (...,) -> (...,3,3)  (:326-340)."""
    c, s = torch.cos(t), torch.sin(t)
    z, o = torch.zeros_like(t), torch.ones_like(t)
    return torch.stack([torch.stack([c, s, z], -1), torch.stack([-s, c, z], -1),
                        torch.stack([z, z, o], -1)], -2)


class _TableProduct(torch.autograd.Function):
    """table (B, N, S*H) = features (B, N, C) . W^T (W (S*H, C)): the blend's per-seed products (the
    first conv of the MiniPointNets evaluated over the N seeds, BlendConv).  What autograd would run for
    it, except the weight gradient dW = dTable^T . features: a (S*H) x C product over B*N = 8 192 rows,
    which the BLAS library serves with ONE 32 x 32-tile kernel at S*H = C = 256 (64 workgroups on 256 CUs:
    67 us for 1 GFLOP) -- here it is a batched product over the B scenes + a sum over them (B times the
    workgroups) whenever S*H * C is small."""

    @staticmethod
    def forward(ctx, features, w):
        ctx.save_for_backward(features, w)
        return torch.matmul(features, w.t())

    @staticmethod
    def backward(ctx, d_table):
        features, w = ctx.saved_tensors
        d_feat = d_w = None
        if ctx.needs_input_grad[0]:
            d_feat = torch.matmul(d_table, w)
        if ctx.needs_input_grad[1]:
            B, N, SH = d_table.shape
            if B > 1 and SH * w.shape[1] <= 256 * 256:
                d_w = torch.bmm(d_table.transpose(1, 2), features).sum(0)
            else:
                d_w = torch.mm(d_table.reshape(B * N, SH).t(), features.reshape(B * N, -1))
        return d_feat, d_w


class DeferredBlendConv:
    """The raw first-conv outputs of S MiniPointNets, (B, S, H, K, G), NOT evaluated yet: the
    operands of ``mmdet3d_ops.BlendConv`` (side_pooling_module.py:226-243, 346-349 through the 3-NN
    blend).  The fused MiniPointNet path evaluates them inside ``fused_mlp.BlendMiniHeadFn`` -- one
    autograd node for the blend, the first norm and the second conv, so the conv output and the
    half-finished gradient the blend backward completes never exist as autograd tensors; every
    other consumer calls ``materialize()`` and gets the ordinary differentiable tensor."""

    def __init__(self, table, wx, idx, weight, rel, segs, G, K):
        self.table, self.wx, self.idx, self.weight, self.rel = table, wx, idx, weight, rel
        self.segs, self.G, self.K = segs, G, K
        B, H = table.shape[0], table.shape[2] // segs
        self.shape = (B, segs, H, K, G)
        self.dtype, self.device = table.dtype, table.device

    @property
    def has_stats(self):
        from ..mmdet3d_ops import fused_mlp
        return fused_mlp.blend_mini_head_supported(backend_for(self.table), self.table, self.idx, self.segs)

    def materialize(self):
        """-> (c0 (B, S, H, K, G), statistics partials or None)."""
        out, stats = blend_conv(self.table, self.wx, self.idx, self.weight, self.rel, self.segs,
                                self.G, True)
        return out.view(self.shape), (stats if stats.numel() else None)


def _resolve_c0(nets, c0, c0_stats):
    """A deferred first-conv output stays deferred only for the fused MiniPointNet path."""
    if isinstance(c0, DeferredBlendConv):
        if fused_mini_ok(nets, c0, None):
            return c0, None
        return c0.materialize()
    return c0, c0_stats


class MiniPointNet(nn.Module):
    """(B,C,K,G) -> (B,feature_dim,K): conv-bn-relu-conv, max over G, concat
    [global, local], conv-bn-relu-conv, max over G  (:343-370)."""

    def __init__(self, channels: int, feature_dim: int, hide_dim=256):
        super().__init__()
        self.first_conv = nn.Sequential(
            PointwiseConv2d(channels, hide_dim, 1, bias=False), FusedBNReLU2d(hide_dim),
            nn.Identity(), PointwiseConv2d(hide_dim, hide_dim // 2, 1))  # [2] was the ReLU
        self.second_conv = nn.Sequential(
            PointwiseConv2d(hide_dim, hide_dim, 1, bias=False), FusedBNReLU2d(hide_dim),
            nn.Identity(), PointwiseConv2d(hide_dim, feature_dim, 1))

    def forward(self, points=None, conv0_out=None, a0=None, c0_stats=None):
        """``points`` (B,C,K,G) grid features, or ``conv0_out`` (B,H,K,G) = the first conv
        already applied through the blend (SidePooling.first_conv_through_blend), or ``a0`` =
        that followed by the first norm + ReLU as well (``with_norm=True`` there).

        Same function as the reference's
            f = first_conv(x); g = max_G f; y = second_conv(cat[g expanded, f]); out = max_G y
        evaluated without materialising the (B, 2*128, K, G) concatenation: the first 1x1 conv
        of ``second_conv`` is linear, so  W @ cat[g, f] = W[:, :H] @ g + W[:, H:] @ f  -- the
        global half is ONE column per proposal instead of G identical ones (a quarter of this
        net's MACs), and the biases of the two convs that feed a max commute with it
        (max_G (c + b) = max_G c + b), so they are added on the pooled (B, C, K) tensors.
        Differences from the concatenated form are summation-order rounding only."""
        conv0, bn0, _, conv3 = self.first_conv
        sconv0, sbn0, _, sconv3 = self.second_conv
        if isinstance(conv0_out, DeferredBlendConv):      # (one net: S = 1)
            conv0_out, c0_stats = _resolve_c0([self], conv0_out, c0_stats)
            if isinstance(conv0_out, DeferredBlendConv):
                return fused_mini_pointnets([self], conv0_out, c0_stats).squeeze(1)
            conv0_out = conv0_out.squeeze(1)
        if a0 is None and conv0_out is not None and fused_mini_ok([self], conv0_out.unsqueeze(1), c0_stats):
            return fused_mini_pointnets([self], conv0_out.unsqueeze(1), c0_stats).squeeze(1)
        if a0 is None:   # c0_stats: (sum, sum^2) partials of conv0_out left by its producer
            a0 = bn0(conv0(points)) if conv0_out is None else bn0(conv0_out, pre_partial=c0_stats)
        c = pointwise_conv(a0, conv3.weight)                           # f without its bias
        g, c = group_max_pool_shared(c)                                # (B, H, K): max_G f - b
        half = conv3.out_channels
        w = sconv0.weight.reshape(sconv0.out_channels, -1)
        w_g, w_l = w[:, :half], w[:, half:]
        b3 = conv3.bias if conv3.bias is not None else c.new_zeros(half)
        # global half + everything the bias b contributes:  W_g (g + b) + W_l b
        small = pointwise_conv(g, w_g.unsqueeze(-1)) + (w @ torch.cat([b3, b3])).view(1, -1, 1)
        # the per-proposal term is added inside the norm kernels (never materialised)
        y = sbn0(pointwise_conv(c, w_l.reshape(w_l.shape[0], half, 1, 1)), row_bias=small)
        out = group_max_pool(pointwise_conv(y, sconv3.weight))
        return out + sconv3.bias.view(1, -1, 1) if sconv3.bias is not None else out


def stacked_running_stats(layers):
    """The S layers' running statistics as one (2, S*C) tensor that kernels update in place: the
    per-layer buffers are kept as VIEWS of it (re-made if a .to() / load replaced them), so
    stacking and un-stacking them costs no launch."""
    C = layers[0].num_features
    owner = layers[0]
    pack = getattr(owner, '_stacked_stats', None)
    if pack is None or pack.device != owner.running_mean.device or any(
            l.running_mean.data_ptr() != pack[0, i * C:].data_ptr()
            or l.running_var.data_ptr() != pack[1, i * C:].data_ptr() for i, l in enumerate(layers)):
        with torch.no_grad():
            pack = torch.stack([torch.cat([l.running_mean for l in layers]),
                                torch.cat([l.running_var for l in layers])])
        for i, l in enumerate(layers):
            l.running_mean = pack[0, i * C:(i + 1) * C]
            l.running_var = pack[1, i * C:(i + 1) * C]
        owner._stacked_stats = pack
    return pack


def _stacked_bn(layers, x, row_bias=None, pre_partial=None):
    """ONE fused BatchNorm(+ReLU) over the S*C channels of x (B, S*C, ...) for the S norm
    layers ``layers`` (each C channels): statistics are per channel, so stacking the layers
    along the channel axis is the same arithmetic as calling them one by one.  Running
    statistics and batch counters of every layer are updated as its own forward would."""
    from ..mmdet3d_ops import fused_mlp
    from ..mmdet3d_ops import norm as _norm
    first = layers[0]
    if not first.training:   # evaluation: one scale / bias pass over the stacked channels
        coef = _norm.stacked_eval_coef(layers)
        return _norm.affine_relu_eval(x, coef, first.fuse_relu, row_bias)
    pack = stacked_running_stats(layers)
    rm, rv = pack[0], pack[1]
    gamma, beta = fused_mlp.stack_groups([[l.weight for l in layers], [l.bias for l in layers]])
    gamma, beta = grad_slots.reshaped(gamma, -1), grad_slots.reshaped(beta, -1)
    y = _norm.BNReLUTrain.apply(x, gamma, beta, rm, rv, first.momentum, first.eps, first.fuse_relu,
                                row_bias, pre_partial)
    for l in layers:    # (the running statistics were updated in place through their views)
        _norm.count_batch(l.num_batches_tracked)
    return y


def _stackable_bn(layers):
    f = layers[0]
    mode_ok = all(l.training for l in layers) or \
        (not any(l.training for l in layers) and not torch.is_grad_enabled())
    return mode_ok and all(type(l) is type(f) and l.affine and l.track_running_stats
               and l.momentum is not None and l.momentum == f.momentum and l.eps == f.eps
               and l.fuse_relu == f.fuse_relu and l.num_features == f.num_features
               for l in layers)


def grouped_mini_pointnets(nets, c0, normed=False, c0_stats=None):
    """S structurally identical MiniPointNets on S inputs at once: ``c0`` (B, S, H, K, G) =
    the outputs of their first convs (``normed``: already through the first norm + ReLU) ->
    (B, S, F, K).  Same function as calling
    ``net(conv0_out=c0[:, i])`` for each net (see MiniPointNet.forward for the algebra); every
    1x1 conv is one broadcast batched GEMM over the stacked weights (48 instead of 8 matrices
    per launch at B = 8) and every norm layer one stacked BatchNorm."""
    if not normed:
        c0, c0_stats = _resolve_c0(nets, c0, c0_stats)
    B, S, H, K, G = c0.shape
    if not normed and fused_mini_ok(nets, c0, c0_stats):
        return fused_mini_pointnets(nets, c0, c0_stats)
    f, sc = [n.first_conv for n in nets], [n.second_conv for n in nets]
    stack = lambda ws: torch.stack([w.flatten(1) for w in ws]).unsqueeze(0)  # noqa: E731
    a0 = c0.reshape(B, S, H, K * G) if normed else \
        _stacked_bn([x[1] for x in f], c0.reshape(B, S * H, K, G),
                    pre_partial=c0_stats).view(B, S, H, K * G)
    c = torch.matmul(stack([x[3].weight for x in f]), a0)               # (B,S,half,K*G)
    half = c.shape[2]
    g, c5 = group_max_pool_shared(c.view(B, S, half, K, G))              # (B,S,half,K)
    c = c5.view(B, S, half, K * G)
    w = stack([x[0].weight for x in sc])                                 # (1,S,H2,2*half)
    H2 = w.shape[2]
    b3 = torch.stack([x[3].bias if x[3].bias is not None else c.new_zeros(half) for x in f])
    small = torch.matmul(w[..., :half], g) \
        + torch.matmul(w[0], torch.cat([b3, b3], 1).unsqueeze(-1)).view(1, S, H2, 1)
    y = _stacked_bn([x[1] for x in sc],
                    torch.matmul(w[..., half:], c).view(B, S * H2, K, G),
                    row_bias=small.reshape(B, S * H2, K))
    out = torch.matmul(stack([x[3].weight for x in sc]), y.view(B, S, H2, K * G))
    out = group_max_pool(out.view(B, S, -1, K, G))
    if sc[0][3].bias is not None:
        out = out + torch.stack([x[3].bias for x in sc]).view(1, S, -1, 1)
    return out


def fused_mini_ok(nets, c0, c0_stats):
    """The S = len(nets) MiniPointNets can run on the fused layer kernels (fused_mlp.MiniHeadFn /
    MiniTailFn): training-mode native norms, statistics partials of c0 at hand, built shapes."""
    from ..mmdet3d_ops import fused_mlp
    backend = backend_for(c0.table if isinstance(c0, DeferredBlendConv) else c0)
    bn0s, bn1s = [n.first_conv[1] for n in nets], [n.second_conv[1] for n in nets]
    training = all(l.training for l in bn0s + bn1s)
    evaluating = not any(l.training for l in bn0s + bn1s) and not torch.is_grad_enabled()
    if isinstance(c0, DeferredBlendConv):    # its statistics partials come with its evaluation
        c0_stats = c0.table if c0.has_stats else None
    if not (training or evaluating) or (training and c0_stats is None):
        return False
    if not (_stackable_bn(bn0s) and _stackable_bn(bn1s)):
        return False
    if any(n.first_conv[0].bias is not None or n.second_conv[0].bias is not None for n in nets):
        return False
    ref = c0.table if isinstance(c0, DeferredBlendConv) else c0
    return fused_mlp.mini_pointnets_fused_supported(backend, c0, c0_stats if training else ref,
                                                    c0.shape[-1])


def fused_mini_pointnets(nets, c0, c0_stats):
    """c0 (B, S, H, K, G) raw first-conv outputs -> (B, S, F, K): the same function as
    ``grouped_mini_pointnets`` / ``MiniPointNet.forward`` with every 1x1 conv as ONE launch of the
    native layer kernel (norm + ReLU on the operand load, statistics / max from the
    accumulators: no normalised tensor, no separate statistics or pooling pass)."""
    from ..mmdet3d_ops import fused_mlp
    from ..mmdet3d_ops import norm as _norm
    B, S, H, K, G = c0.shape
    f, sc = [n.first_conv for n in nets], [n.second_conv for n in nets]
    bn0s, bn1s = [x[1] for x in f], [x[1] for x in sc]
    half = f[0][3].out_channels

    def stacked(layers):
        pack = stacked_running_stats(layers)
        for l in layers:
            _norm.count_batch(l.num_batches_tracked)
        return pack[0], pack[1], layers[0].momentum, layers[0].eps

    conv3, conv_g, conv4 = [x[3] for x in f], [x[0] for x in sc], [x[3] for x in sc]
    zero = None
    if conv3[0].bias is None:
        zero = conv3[0].weight.new_zeros(half)
    groups = [[l.weight for l in bn0s], [l.bias for l in bn0s],
              [m.weight.flatten(1) for m in conv3], [m.weight.flatten(1) for m in conv_g],
              [m.bias if m.bias is not None else zero for m in conv3],
              [l.weight for l in bn1s], [l.bias for l in bn1s],
              [m.weight.flatten(1) for m in conv4]]
    if conv4[0].bias is not None:
        groups.append([m.bias for m in conv4])
    gamma0, beta0, w3, w, b3, gamma1, beta1, w4, *b4 = fused_mlp.stack_groups(groups)
    evaluating = not bn0s[0].training
    deferred = c0 if isinstance(c0, DeferredBlendConv) else None
    backend = backend_for(deferred.table if deferred is not None else c0)
    if evaluating:   # test path: the folded running statistics are the operand transforms
        if deferred is not None:
            c0 = deferred.materialize()[0]
        coef0 = _norm.stacked_eval_coef(bn0s)
        c, g, _ = fused_mlp.mini_head_kernels(backend, c0.reshape(B, S, H, K * G).contiguous(),
                                               coef0, w3, G)
    elif deferred is not None:   # blend + first norm + second conv as ONE autograd node
        d = deferred
        c, g = fused_mlp.BlendMiniHeadFn.apply(d.table, d.wx, d.idx, d.weight, d.rel, d.segs, d.G,
                                               stacked(bn0s), G, grad_slots.reshaped(gamma0, -1),
                                               grad_slots.reshaped(beta0, -1), w3)
    else:
        c, g = fused_mlp.MiniHeadFn.apply(c0.reshape(B, S, H, K * G), c0_stats, stacked(bn0s), G,
                                          grad_slots.reshaped(gamma0, -1), grad_slots.reshaped(beta0, -1),
                                          w3)  # w3 (S, half, H)
    H2 = w.shape[1]                                                                  # w (S, H2, 2*half)
    # global half + everything the bias b3 contributes:  W_g (g + b3) + W_l b3.  The two halves
    # of w come from ONE split (its backward is one concatenation; three overlapping uses of w
    # cost two zero-filled slice gradients and two additions)
    w_g, w_l = w.split(half, dim=2)
    gx = g.reshape(B * S, half, g.shape[-1])
    if not evaluating and fused_mlp.stack1d_supported(backend, gx, [(half, H2)], [None], S, which=fused_mlp.HEADS):
        # ... as ONE bias-carrying layer of the layer kernel over the K proposals (weight group =
        # net): W_g g + (W_g b3 + W_l b3); the bias is two tiny products of parameters
        bias = torch.matmul(w, torch.cat([b3, b3], 1).unsqueeze(-1)).reshape(-1)          # (S * H2)
        small = fused_mlp.Stack1dFn.apply(gx, (fused_mlp.Stack1dLayer(True, None),), S, w_g, bias) \
            .view(B, S, H2, -1)
    else:
        small = torch.matmul(w_g.unsqueeze(0), g + b3.view(1, S, half, 1)) \
            + torch.matmul(w_l, b3.unsqueeze(-1)).view(1, S, H2, 1)
    if evaluating:
        y, _ = fused_mlp.mini_tail_first(backend, c, small.contiguous(), w_l, G)
        coef1 = _norm.stacked_eval_coef(bn1s)
        out, _ = fused_mlp.mini_tail_second(backend, y, coef1, w4, G)
    else:
        out = fused_mlp.MiniTailFn.apply(c, small, stacked(bn1s), G, w_l,
                                         grad_slots.reshaped(gamma1, -1), grad_slots.reshaped(beta1, -1),
                                         w4)  # w4 (S, F, H2)
    if b4:
        out = fused_mlp.AddChannelBias.apply(out, b4[0].view(S, -1))
    return out


def mini_pointnets_groupable(nets, c0):
    ref = c0.table if isinstance(c0, DeferredBlendConv) else c0
    if backend_for(ref).name != 'hip' or c0.dtype != torch.float32:
        return False
    G = c0.shape[-1]
    if not (4 <= G <= 64 and G & (G - 1) == 0):   # row-bias norm and max-pool kernels
        return False
    return _stackable_bn([n.first_conv[1] for n in nets]) \
        and _stackable_bn([n.second_conv[1] for n in nets])


def batched_heads(heads, x):
    """S structurally identical score heads on S inputs in one pass: ``heads`` = S
    nn.Sequential of PointwiseConv1d / FusedBNReLU1d / Identity, ``x`` (B, S, Cin, P) ->
    (B, S, Cout, P).  The convs run as one broadcast batched GEMM over the stacked weights and
    every norm layer as ONE BatchNorm over the S*C stacked channels (statistics are per channel,
    so stacking heads along the channel axis changes nothing) -- the same arithmetic as calling
    the heads one by one (side_pooling_module.py:314-321), at a sixth of the launches."""
    from ..mmdet3d_ops import fused_mlp
    B, S = x.shape[:2]
    convs = [layers for layers in zip(*heads) if isinstance(layers[0], PointwiseConv1d)]
    groups = [[l.weight.flatten(1) for l in layers] for layers in convs] \
        + [[l.bias for l in layers] for layers in convs if layers[0].bias is not None]
    stacked = iter(fused_mlp.stack_groups(groups))          # one multi-tensor copy for all of them
    weights = {id(layers[0]): next(stacked) for layers in convs}
    biases = {id(layers[0]): next(stacked) for layers in convs if layers[0].bias is not None}
    steps = [layers for layers in zip(*heads) if not isinstance(layers[0], nn.Identity)]
    fused = _heads_as_stack(heads, x, steps, weights, biases)
    if fused is not None:
        return fused
    held = None   # a conv bias waiting for the norm layer behind it
    for i, layers in enumerate(steps):
        first = layers[0]
        if isinstance(first, PointwiseConv1d):
            x = torch.matmul(weights[id(first)].unsqueeze(0), x)               # (S, Co, Ci) stacks
            if first.bias is not None:
                if i + 1 < len(steps) and isinstance(steps[i + 1][0], FusedBNReLU1d):
                    # added inside the norm's passes (same values); no gradient: it is zero
                    held = biases[id(first)].reshape(-1)
                else:
                    x = x + biases[id(first)].view(1, S, -1, 1)
        elif isinstance(first, FusedBNReLU1d):
            C, P = x.shape[2], x.shape[3]
            x = _stacked_bn(layers, x.reshape(B, S * C, P), row_bias=held).view(B, S, C, P)
            held = None
        else:
            raise TypeError(f'batched_heads: unsupported layer {type(first).__name__}')
    return x


def _heads_as_stack(heads, x, steps, weights, biases):
    """The S stacked heads as ONE fused chain on the layer kernel (``fused_mlp.Stack1dFn``, weight
    group = head): conv -> norm -> ReLU pairs followed by a plain conv.  None when not served."""
    from ..mmdet3d_ops import fused_mlp
    B, S, cin, P = x.shape
    convs, norms = [], []
    i = 0
    while i < len(steps):
        if not isinstance(steps[i][0], PointwiseConv1d):
            return None
        convs.append(steps[i])
        if i + 1 < len(steps) and isinstance(steps[i + 1][0], FusedBNReLU1d):
            norms.append(steps[i + 1])
            i += 2
        else:
            norms.append(None)
            i += 1
    shapes = [(c[0].in_channels, c[0].out_channels) for c in convs]
    if not fused_mlp.stack1d_supported(backend_for(x), x.reshape(B * S, cin, P), shapes,
                                       [None if n is None else n[0] for n in norms], S, which=fused_mlp.HEADS):
        return None
    bn_groups = [g for n in norms if n is not None for g in ([l.weight for l in n], [l.bias for l in n])]
    stacked = iter(fused_mlp.stack_groups(bn_groups)) if bn_groups else iter(())
    gammas, betas, stats = [], [], []
    for n in norms:
        if n is None:
            gammas.append(None); betas.append(None); stats.append(None)
        else:
            gammas.append(grad_slots.reshaped(next(stacked), -1))
            betas.append(grad_slots.reshaped(next(stacked), -1))
            pack = stacked_running_stats(list(n))
            stats.append((pack[0], pack[1]))
    w = [weights[id(c[0])] for c in convs]
    b = [grad_slots.reshaped(biases[id(c[0])], -1) if c[0].bias is not None else None for c in convs]
    y = fused_mlp.stack1d(x.reshape(B * S, cin, P), [list(c) for c in convs],
                          [None if n is None else list(n) for n in norms], S=S, weights=w, biases=b,
                          gammas=gammas, betas=betas, stats=stats)
    return y.view(B, S, y.shape[1], P)


def heads_batchable(heads, x):
    """True when ``batched_heads`` reproduces the per-head loop: native training BatchNorm on
    the HIP back end, fp32, same layer types and hyper-parameters in every head."""
    if backend_for(x).name != 'hip' or x.dtype != torch.float32:
        return False
    for layers in zip(*heads):
        t = type(layers[0])
        if any(type(l) is not t for l in layers):
            return False
        if isinstance(layers[0], FusedBNReLU1d):
            if not _stackable_bn(layers):
                return False
        elif not isinstance(layers[0], (PointwiseConv1d, nn.Identity)):
            return False
    return True


def _score_head(in_ch, out_ch):
    # indices as in the reference Sequential (conv, bn, relu, conv, bn, relu, conv); the
    # ReLUs are folded into the norm layers, Identity keeps the state-dict positions
    return nn.Sequential(PointwiseConv1d(in_ch, 128, 1), FusedBNReLU1d(128), nn.Identity(),
                         PointwiseConv1d(128, 128, 1), FusedBNReLU1d(128), nn.Identity(),
                         PointwiseConv1d(128, out_ch, 1))


class SidePooling(nn.Module):
    def __init__(self, num_class, num_heading_bin, num_size_cluster, mean_size_arr_path,
                 num_proposal, sampling, seed_feat_dim=256, query_feats='seed',
                 iou_class_depend=True):
        super().__init__()
        self.num_class = num_class
        self.num_heading_bin = num_heading_bin
        self.num_size_cluster = num_size_cluster
        # the reference loads scannet_means.npz here (:28) and never reads it again
        self.mean_size_arr = None
        self.num_proposal = num_proposal
        self.sampling = sampling
        self.seed_feat_dim = seed_feat_dim
        self.query_feats = query_feats
        self.iou_class_depend = iou_class_depend
        self.reg_topk = 4
        self.grid_size = g = 4
        self.left_mask = [i // g * g * g + i % g for i in range(g * g)]
        self.right_mask = [i // g * g * g + i % g + g * (g - 1) for i in range(g * g)]
        # the six face selections of grid_for_side as ONE device-resident index (graph-safe)
        face_idx = (list(range(0, g * g)) + list(range(g ** 3 - g * g, g ** 3))
                    + list(range(g - 1, g ** 3, g)) + list(range(0, g ** 3, g))
                    + self.left_mask + self.right_mask)
        self.register_buffer('_face_idx', torch.tensor(face_idx, dtype=torch.long),
                             persistent=False)
        self._register_grid_tables(face_idx)
        self.iou_size = num_class if iou_class_depend else 1
        before, head = [], []
        for _ in range(6):
            before.append(MiniPointNet(seed_feat_dim + 3, 128))
            head.append(_score_head(128 + 33 + 4 + 1, self.iou_size))
        before.append(MiniPointNet(seed_feat_dim + 3, 128))
        head.append(_score_head(128, self.iou_size))
        self.mlps_before = nn.ModuleList(before)
        self.mlps_head = nn.ModuleList(head)

    def stacked_parameter_groups(self):
        """Lists of parameters the step uses STACKED: tensor kind by tensor kind, the six side
        MiniPointNets and the six side score heads (``grouped_mini_pointnets`` / ``batched_heads``).
        A flat training state that lays every list out contiguously (``dp.FlatTrainState(...,
        stack_groups=)``) turns each stack into a view and each stacked gradient into one slot."""
        groups = []
        for modules in (list(self.mlps_before[:6]), list(self.mlps_head[:6])):
            per = [list(m.parameters()) for m in modules]
            if len(per) == 6 and all(len(p) == len(per[0]) for p in per):
                groups += [list(kind) for kind in zip(*per)
                           if all(t.shape == kind[0].shape for t in kind)]
        return groups

    # BlendConvBN (first conv AND its norm + ReLU by recomputation, the conv output never
    # stored) is exact and saves 1.3 GB of activations, but measured break-even on MI355X: the
    # forward gains 0.23 ms, the backward loses 0.3 ms to the recomputed row gathers at the two
    # waves per SIMD its LDS tile allows.  Off; flip to trade time for memory.
    fuse_first_norm = False

    def _register_grid_tables(self, face_idx, plane=None):
        """Box-frame multipliers of the grid points of one proposal, in the order the grids are
        consumed: ``_mult_box`` (g^3,3) for the box grid, ``_mult_side`` for the six face groups
        (with ``_plane_side`` = their +-10 % plane factors in the SAQE variant, else zeros)."""
        g = self.grid_size
        step = torch.linspace(-1, 1, g)
        full = torch.stack([step.view(g, 1, 1).expand(g, g, g), step.view(1, g, 1).expand(g, g, g),
                            step.view(1, 1, g).expand(g, g, g)], -1).reshape(-1, 3)
        side = full[torch.tensor(face_idx)]
        if plane is not None:            # (6,3) axis factors -> three planes per face
            g2 = g * g
            side = side.view(6, g2, 3)
            pl = plane.view(6, 1, 3).expand(6, g2, 3)
            side, plane = (torch.cat([side, side, side], 1).reshape(-1, 3),
                           torch.cat([-pl, torch.zeros_like(pl), pl], 1).reshape(-1, 3))
        else:
            plane = torch.zeros_like(side)
        self.register_buffer('_mult_box', full.contiguous(), persistent=False)
        self.register_buffer('_mult_side', side.contiguous(), persistent=False)
        self.register_buffer('_plane_side', plane.contiguous(), persistent=False)

    def fused_taps(self, origin_xyz, center, size, heading, which):
        """idx, weight, rel of one grid set ('side' or 'box') straight from the proposal
        parameters: one launch instead of generate_grid / grid_for_* / _blend_taps."""
        mult = self._mult_side if which == 'side' else self._mult_box
        plane = self._plane_side if which == 'side' else torch.zeros_like(mult)
        return backend_for(origin_xyz).grid_taps(center.contiguous(), size.contiguous(),
                                                 heading.contiguous(), mult, plane, origin_xyz)

    def extract_features(self, end_points):
        """-> seed xyz (B,N,3), seed features POINT-major (B,N,C) for the blend kernel."""
        return (end_points['seed_points'].detach().contiguous(),
                end_points['seed_features'].detach().transpose(1, 2).contiguous())

    def generate_grid(self, size):
        """(B,K,3) sizes -> (B,K,g^3,3) box-frame grid, x slowest, z fastest (:87-122)."""
        B, K = size.shape[:2]
        g = self.grid_size
        step = torch.linspace(-1, 1, g, device=size.device)
        gx = step.view(g, 1, 1).repeat(1, g, g).view(1, 1, -1).expand(B, K, -1)
        gy = step.view(1, g, 1).repeat(g, 1, g).view(1, 1, -1).expand(B, K, -1)
        gz = step.view(1, 1, g).repeat(g, g, 1).view(1, 1, -1).expand(B, K, -1)
        x_grid = gx * size[:, :, 0:1] / 2
        y_grid = gy * size[:, :, 1:2] / 2
        z_grid = gz * size[:, :, 2:3] / 2
        return torch.cat([x_grid.unsqueeze(-1), y_grid.unsqueeze(-1), z_grid.unsqueeze(-1)],
                         dim=-1)

    def _to_scene(self, grid, center, heading):
        B, K = center.shape[:2]
        rot_mat = rot_gpu(heading).view(-1, 3, 3)
        grid = torch.bmm(grid.reshape(B * K, -1, 3), rot_mat.transpose(1, 2)).view(B, K, -1, 3)
        return grid + center.unsqueeze(2)

    def grid_for_side(self, whole_grid, center, heading):
        """front/back/top/down/left/right face grids, rotated + translated (:124-157)."""
        side_grid = torch.index_select(whole_grid, 2, self._face_idx)
        return self._to_scene(side_grid, center, heading)

    def grid_for_bbox(self, whole_grid, center, heading):
        return self._to_scene(whole_grid, center, heading)

    def _blend_taps(self, origin_xyz, whole_grid, center):
        """3-NN of every grid point among the seeds -> idx (B,n,3) int32, inverse-distance
        weights (B,n,3), grid xyz relative to the proposal centre (B,n,3)  (:204-225)."""
        B, K = center.shape[:2]
        grid_size = whole_grid.shape[1] // K
        _, idx = three_nn(whole_grid, origin_xyz)
        interp_points = torch.gather(origin_xyz, 1, idx.view(B, -1, 1).expand(-1, -1, 3).long())
        expanded = whole_grid.unsqueeze(2).expand(-1, -1, 3, -1).reshape(B, -1, 3)
        dist = interp_points - expanded
        dist = torch.sqrt(torch.sum(dist * dist, dim=2))
        relative_grid = whole_grid - center.unsqueeze(2).expand(-1, -1, grid_size, -1) \
            .reshape(B, -1, 3)
        weight = (1 / (dist + 1e-8)).view(B, -1, 3)
        weight = (weight / torch.sum(weight, dim=2, keepdim=True)).contiguous()
        return idx, weight, relative_grid.contiguous()

    def first_conv_through_blend(self, nets, origin_xyz, origin_features, whole_grid, center,
                                 taps=None, with_norm=False, defer=False):
        """Outputs of ``net.first_conv[0]`` for the S = len(nets) MiniPointNets that read the
        S consecutive point groups of every proposal: ((B,S,H,K,G), normed?, statistics partials
        of the output for the first norm layer or None), evaluated as
        W_xyz . rel + blend(W_f . F) (mmdet3d_ops.BlendConv) instead of
        conv(cat[rel, blend(F)]) -- the conv runs over the N seeds, not the K*S*G grid points.
        ``defer``: return the operands as a ``DeferredBlendConv`` instead of the tensor."""
        B, K = center.shape[:2]
        segs = len(nets)
        idx, weight, rel = taps if taps is not None \
            else self._blend_taps(origin_xyz, whole_grid, center)
        G = idx.shape[1] // (K * segs)
        from ..mmdet3d_ops import fused_mlp
        w = fused_mlp.stack_groups([[net.first_conv[0].weight.flatten(1) for net in nets]])[0]   # (S, H, 3+C)
        H = w.shape[1]
        w_xyz, w_feat = fused_mlp.SplitXyzFeat.apply(w)       # (S, H, 3) contiguous, (S*H, C) view
        table = _TableProduct.apply(origin_features, w_feat)  # (B,N,S*H)
        bns = [net.first_conv[1] for net in nets]
        if with_norm and _stackable_bn(bns) and H % 64 == 0 and H <= 256 and (K * G) % 64 == 0 \
                and origin_features.dtype == torch.float32:
            # ... and the norm + ReLU behind it, the conv output never stored (BlendConvBN)
            from ..mmdet3d_ops import norm as _norm
            rm = torch.cat([l.running_mean for l in bns])
            rv = torch.cat([l.running_var for l in bns])
            out = blend_conv_bn(table, w_xyz, torch.cat([l.weight for l in bns]),
                                torch.cat([l.bias for l in bns]), idx, weight, rel, rm, rv,
                                bns[0].momentum, bns[0].eps, segs, G)
            with torch.no_grad():
                torch._foreach_copy_([l.running_mean for l in bns], list(rm.split(H)))
                torch._foreach_copy_([l.running_var for l in bns], list(rv.split(H)))
                for l in bns:
                    _norm.count_batch(l.num_batches_tracked)
            return out.view(B, segs, H, K, G), True, None
        if defer:   # evaluated by its consumer (BlendMiniHeadFn, or materialize() for any other)
            return DeferredBlendConv(table, w_xyz, idx, weight, rel, segs, G, K), False, None
        # (B, S, H, K*G) and the (sum, sum^2) partials of it for the first norm layer
        out, stats = blend_conv(table, w_xyz, idx, weight, rel, segs, G, True)
        return out.view(B, segs, H, K, G), False, (stats if stats.numel() else None)

    def grid_features(self, origin_xyz, origin_features, whole_grid, center, segs=1):
        """(B,N,3),(B,N,C),(B,K*S*G,3),(B,K,3) -> (B,S,3+C,K,G)  (:183-243).

        The grid points of a proposal come as ``segs`` = S consecutive groups of G (the six
        faces, or one group for the box grid); the result holds one contiguous (3+C, K, G)
        block per group, i.e. what the reference reaches with
        cat([rel_xyz, interpolated]) -> split(G, dim=-1) -> .contiguous() (:304-313)."""
        B, K = center.shape[:2]
        grid_size = whole_grid.shape[1] // K
        idx, weight, relative_grid = self._blend_taps(origin_xyz, whole_grid, center)
        G, C = grid_size // segs, origin_features.shape[2]
        out = origin_features.new_empty(B, segs, 3 + C, K * G)
        out[:, :, :3] = relative_grid.view(B, K, segs, G, 3).permute(0, 2, 4, 1, 3) \
            .reshape(B, segs, 3, K * G)
        three_interpolate_segmented(origin_features, idx, weight, out, segs, G, 3)
        return out.view(B, segs, 3 + C, K, G)

    def dist_feature(self, end_points, prefix='', copies=2):
        """[33 side-bin probabilities, top-4, unbiased variance] per face, duplicated
        for the jittered half -> (6, B, 38, 2K)  (:245-264)."""
        prob = end_points[f'{prefix}bbox_probs'].detach()
        backend = backend_for(prob)
        if (backend.name == 'hip' and self.reg_topk == 4 and prob.dtype == torch.float32
                and prob.dim() == 4 and prob.shape[1] == 6 and prob.shape[2] >= 5):
            return backend.side_prob_stats(prob.contiguous(), copies)
        stat = torch.cat([prob, prob.topk(self.reg_topk, dim=2)[0],
                          prob.var(dim=2, keepdim=True)], dim=2)
        return stat.permute(1, 0, 2, 3).repeat(1, 1, 1, copies)   # copies = 1: no jittered half

    def forward(self, center, size, heading, end_points, prefix=''):
        B, K = size.shape[:2]
        origin_xyz, origin_features = self.extract_features(end_points)
        fused = backend_for(origin_xyz).name == 'hip'
        side_nets = list(self.mlps_before[:6])
        if fused:   # grids + taps in one launch each, first convs (+ norm) through the blend;
            #         the literal form below stays the CPU checker's
            side_c0, side_normed, side_stats = self.first_conv_through_blend(
                side_nets, origin_xyz, origin_features, None, center,
                taps=self.fused_taps(origin_xyz, center, size, heading, 'side'),
                with_norm=self.fuse_first_norm, defer=True)
            bbox_c0, bbox_normed, bbox_stats = self.first_conv_through_blend(
                self.mlps_before[6:7], origin_xyz, origin_features, None, center,
                taps=self.fused_taps(origin_xyz, center, size, heading, 'box'),
                with_norm=self.fuse_first_norm, defer=True)
            if not isinstance(bbox_c0, DeferredBlendConv):
                bbox_c0 = bbox_c0.squeeze(1)   # (a view: indexing [:, 0] costs a zero-filled gradient)
        else:
            whole_grid = self.generate_grid(size)
            side_grid = self.grid_for_side(whole_grid, center, heading).view(B, -1, 3).contiguous()
            bbox_grid = self.grid_for_bbox(whole_grid, center, heading).view(B, -1, 3).contiguous()
            side_feats = self.grid_features(origin_xyz, origin_features, side_grid, center, segs=6)
            bbox_feats = self.grid_features(origin_xyz, origin_features, bbox_grid, center)[:, 0]
        dist_feature = self.dist_feature(end_points, prefix,
                                         copies=K // end_points[f'{prefix}bbox_probs'].shape[-1])
        if fused and mini_pointnets_groupable(side_nets, side_c0):
            pooled = grouped_mini_pointnets(side_nets, side_c0, normed=side_normed,
                                            c0_stats=side_stats)               # (B,6,128,2K)
        elif fused:
            if isinstance(side_c0, DeferredBlendConv):
                side_c0, side_stats = side_c0.materialize()
            key = 'a0' if side_normed else 'conv0_out'
            pooled = torch.stack([side_nets[i](**{key: side_c0[:, i]}) for i in range(6)], 1)
        else:
            pooled = torch.stack([side_nets[i](side_feats[:, i]) for i in range(6)], 1)
        heads = list(self.mlps_head[:6])
        x = torch.cat([pooled, dist_feature.transpose(0, 1)], dim=2)      # (B,6,166,2K)
        if heads_batchable(heads, x[:, 0]):
            side_scores = batched_heads(heads, x).transpose(0, 1).contiguous()
        else:
            side_scores = torch.stack([self.mlps_head[i](x[:, i]) for i in range(6)], 0)
        end_points[f'{prefix}side_scores'] = side_scores
        if fused:
            bbox_feats = self.mlps_before[6](**{'a0' if bbox_normed else 'conv0_out': bbox_c0},
                                             c0_stats=bbox_stats)
        else:
            bbox_feats = self.mlps_before[6](bbox_feats)
        if heads_batchable([self.mlps_head[6]], bbox_feats):     # (one head: the same fused chain)
            iou = batched_heads([self.mlps_head[6]], bbox_feats.unsqueeze(1)).squeeze(1)
        else:
            iou = self.mlps_head[6](bbox_feats)
        end_points[f'{prefix}iou_scores'] = iou.transpose(2, 1)
        return end_points
