"""Mirror of ``mmdet3d/ops/pointnet_modules``: ``PointSAModuleMSG`` / ``PointSAModule``
(point_sa_module.py:12-341), ``PointFPModule`` (point_fp_module.py:10-78),
``build_sa_module`` (builder.py:6-38), plus ``ConvModule`` restated from mmcv 1.3.17
(the source is not in the reference tree): conv -> norm -> act,
``bias='auto'`` means "bias iff no norm", sub-module names ``conv`` / ``bn`` /
``activate`` so reference checkpoints' keys line up, Kaiming-normal(fan_out, relu)
conv weights, unit norm weight, zero biases.
"""
from collections import OrderedDict
from typing import List

import torch
from torch import nn
from torch.nn import functional as F

from ..kernels import backend_for, spatial_index_scope
from . import fused_mlp
from .furthest_point_sample import Points_Sampler
from .gather_points import gather_points
from .group_points import (QueryAndGroup, SampleQueryGroupCat, inverted_index,
                           sample_query_group_supported)
from .interpolate import three_interpolate, three_nn
from .norm import FusedBNReLU1d, FusedBNReLU2d
from .pool import group_max_pool


class _PointwiseConvFn(torch.autograd.Function):
    """out[b] = W @ x[b] with a weight gradient that is split along the position axis.

    The default autograd of bmm(expand(W), x) computes dW as B GEMMs with K = P each, which
    leaves most of the chip idle when P is long (a 256x259 output is ~4 tiles per scene).
    For P >= 32768 the reduction is cut into S = 16 chunks per scene, evaluated as one
    strided-batched GEMM per scene over views (no copies), and the partials are summed."""

    SPLIT_MIN_P, SPLIT, FOLD_MAX_P = 32768, 16, 1023

    @staticmethod
    def forward(ctx, x3, w2):
        ctx.save_for_backward(x3, w2)
        return torch.bmm(w2.unsqueeze(0).expand(x3.shape[0], -1, -1), x3)

    @staticmethod
    def backward(ctx, dy):
        x3, w2 = ctx.saved_tensors
        B, co, P = dy.shape
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.bmm(w2.t().unsqueeze(0).expand(B, -1, -1), dy)
        if ctx.needs_input_grad[1]:
            S = _PointwiseConvFn.SPLIT
            backend = backend_for(dy)
            if (backend.name == 'hip' and co <= 64 and x3.shape[1] <= 64 and P >= 32768
                    and dy.dtype == torch.float32 and x3.stride(2) == 1 and x3.stride(1) == P):
                # HBM-bound skinny reductions (SA1): one pass over (dy, x) on the matrix cores
                # with per-workgroup partials (tools/wgrad_bench.py: 0.12 vs 0.29 ms at 64x4,
                # 0.18 vs 0.22 ms at 64x64, P = 131072, B = 8)
                dw = dy.new_empty(co, x3.shape[1])
                backend.conv_wgrad(dy, x3, dw)
            elif P >= _PointwiseConvFn.SPLIT_MIN_P and P % S == 0:
                pc, ci = P // S, x3.shape[1]
                parts = [torch.bmm(dy[b].view(co, S, pc).permute(1, 0, 2),
                                   x3[b].view(ci, S, pc).permute(1, 2, 0)) for b in range(B)]
                dw = torch.stack(parts).sum((0, 1))
            elif B > 1 and P <= _PointwiseConvFn.FOLD_MAX_P and dy.is_cuda:
                # short rows (proposal-level layers, P = 256 .. 512): B GEMMs with K = P each
                # are latency-bound (4 TFLOP/s measured); one GEMM over the folded B * P axis
                # costs two small transposing copies and runs an order of magnitude faster.
                # From P = 1024 (seed-level layers) the B GEMMs + sum win again: the folded
                # product has so few output tiles that it runs on 8..48 workgroups
                # (tools/debug/wgrad_paths.py, inside a replayed graph: 16-27 vs 33-43 us)
                ci = x3.shape[1]
                dw = torch.mm(dy.transpose(0, 1).reshape(co, B * P),
                              x3.transpose(0, 1).reshape(ci, B * P).t())
            else:
                dw = torch.bmm(dy, x3.transpose(1, 2)).sum(0)
        return dx, dw


class _StreamConvStatsFn(torch.autograd.Function):
    """out[b] = W @ x[b] for a skinny first layer (Cin <= 64 over >= 32768 positions) through the
    streaming matrix-core kernel, which also leaves the (sum, sum of squares) partials of the
    output for the BatchNorm that follows: one pass instead of GEMM + statistics pass
    (measured in round 1: 0.069 vs 0.093 + 0.058 ms for 4 -> 64 at SA1).  Backward = that of
    ``_PointwiseConvFn``."""

    @staticmethod
    def forward(ctx, x3, w2):
        x3 = x3.contiguous()
        backend = backend_for(x3)
        B, _, P = x3.shape
        y = x3.new_empty(B, w2.shape[0], P)
        part = x3.new_empty(backend.mlp_stream_parts(B, P), w2.shape[0], 2)
        backend.mlp_stream_forward(x3, w2.contiguous(), y, part)
        ctx.save_for_backward(x3, w2)
        ctx.mark_non_differentiable(part)
        return y, part

    @staticmethod
    def backward(ctx, dy, _dpart):
        return _PointwiseConvFn.backward(ctx, dy)


def stream_conv_eligible(x, conv, norm):
    """First-layer fast path: native training BatchNorm behind a bias-free skinny 1x1 conv."""
    if conv.bias is not None or not isinstance(norm, (FusedBNReLU1d, FusedBNReLU2d)):
        return False
    cin, cout = conv.in_channels, conv.out_channels
    p = x.numel() // max(x.shape[0] * cin, 1)
    return (x.is_cuda and backend_for(x).name == 'hip' and x.dtype == torch.float32
            and norm.training and cin <= 8 and cout <= 128 and p >= 32768 and p % 4 == 0)


def pointwise_conv(x, weight, bias=None):
    """1x1 Conv1d/Conv2d as ONE strided-batched GEMM  out[b] = W @ x[b].

    x (B, Cin, *spatial) contiguous NCHW is already the row-major (Cin x P) operand and
    W @ x[b] is already the row-major NCHW output, so no layout change is needed (the
    vendor conv path transposes to NHWC and back around its implicit-GEMM kernels).
    Same arithmetic as F.conv{1,2}d with a 1x1 kernel, fp32 in / fp32 accumulate.
    """
    B, cin = x.shape[:2]
    w2 = weight.reshape(weight.shape[0], cin)
    out = _PointwiseConvFn.apply(x.reshape(B, cin, -1), w2)
    if bias is not None:
        out = out + bias.view(1, -1, 1)
    return out.view(B, w2.shape[0], *x.shape[2:])


class PointwiseConv1d(nn.Conv1d):
    """nn.Conv1d(kernel_size=1) whose forward is pointwise_conv (same parameters/keys)."""

    def forward(self, x):
        return pointwise_conv(x, self.weight, self.bias)


class PointwiseConv2d(nn.Conv2d):
    def forward(self, x):
        return pointwise_conv(x, self.weight, self.bias)


class ConvModule(nn.Module):
    """1x1 Conv{1,2}d -> BN{1,2}d (or GN) -> ReLU, mmcv-style."""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=0,
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), bias='auto',
                 inplace=True):
        super().__init__()
        conv_type = (conv_cfg or dict(type='Conv2d'))['type']
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        conv_cls = {'Conv1d': PointwiseConv1d, 'Conv2d': PointwiseConv2d}[conv_type]
        assert kernel_size in (1, (1, 1)) and stride in (1, (1, 1)) and padding == 0
        self.conv = conv_cls(in_channels, out_channels, kernel_size, stride=stride,
                             padding=padding, bias=bias)
        self.norm_name = None
        self.act_fused = False  # True: the norm layer applies the ReLU itself
        if self.with_norm:
            ntype = norm_cfg['type']
            relu_follows = act_cfg is not None
            if ntype in ('BN1d', 'BN'):
                self.norm_name, norm = 'bn', FusedBNReLU1d(out_channels, relu=relu_follows)
                self.act_fused = relu_follows
            elif ntype == 'BN2d':
                self.norm_name, norm = 'bn', FusedBNReLU2d(out_channels, relu=relu_follows)
                self.act_fused = relu_follows
            elif ntype == 'GN':
                self.norm_name, norm = 'gn', nn.GroupNorm(norm_cfg['num_groups'],
                                                          out_channels)
            else:
                raise KeyError(f'unsupported norm type {ntype}')
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            assert act_cfg['type'] == 'ReLU'
            self.activate = nn.ReLU(inplace=inplace)
        self.init_weights()

    @property
    def norm(self):
        return getattr(self, self.norm_name) if self.norm_name else None

    def init_weights(self):
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode='fan_out', nonlinearity='relu')
        if self.conv.bias is not None:
            nn.init.constant_(self.conv.bias, 0)
        if self.with_norm:
            nn.init.constant_(self.norm.weight, 1)
            nn.init.constant_(self.norm.bias, 0)

    def forward(self, x):
        if self.with_norm and stream_conv_eligible(x, self.conv, self.norm):
            B, cin = x.shape[:2]
            w2 = self.conv.weight.reshape(self.conv.out_channels, cin)
            y, part = _StreamConvStatsFn.apply(x.reshape(B, cin, -1), w2)
            # (parts, C, 2) -> the norm kernels' (C, slices, 2) layout: a 1 MB transpose
            x = self.norm(y.view(B, w2.shape[0], *x.shape[2:]),
                          pre_partial=part.permute(1, 0, 2).contiguous())
        elif (self.with_norm and self.conv.bias is not None
              and isinstance(self.norm, (FusedBNReLU1d, FusedBNReLU2d))):
            # the bias add moves into the norm's own passes (a launch and a tensor round trip
            # less; same values), and its identically-zero gradient is not summed up
            x = self.norm(pointwise_conv(x, self.conv.weight), chan_bias=self.conv.bias)
        else:
            x = self.conv(x)
            if self.with_norm:
                x = self.norm(x)
        if self.with_activation and not self.act_fused:
            x = self.activate(x)
        return x


class BasePointSAModule(nn.Module):
    """Set abstraction: sample centres, group a neighbourhood around each (one grouper per
    scale), run the scale's shared MLP over every neighbourhood, pool it, concatenate the scales.
    Constructor keywords and attribute names as in point_sa_module.py:12-211."""

    def __init__(self, num_point, radii, sample_nums, mlp_channels, fps_mod=['D-FPS'],
                 fps_sample_range_list=[-1], dilated_group=False, use_xyz=True,
                 pool_mod='max', normalize_xyz=False, grouper_return_grouped_xyz=False,
                 grouper_return_grouped_idx=False):
        super().__init__()
        scales = len(radii)
        if not (len(sample_nums) == scales == len(mlp_channels)):
            raise AssertionError('radii, sample_nums and mlp_channels need one entry per scale')
        if pool_mod not in ('max', 'avg'):
            raise AssertionError(f'unknown pool_mod {pool_mod!r}')
        if not (isinstance(fps_mod, (list, tuple)) and isinstance(fps_sample_range_list, (list, tuple))
                and len(fps_mod) == len(fps_sample_range_list)):
            raise AssertionError('fps_mod and fps_sample_range_list must be sequences of one length')
        if isinstance(num_point, int):
            num_point = [num_point]
        elif not isinstance(num_point, (list, tuple)):
            raise NotImplementedError('Error type of num_point!')
        self.num_point = num_point
        self.mlp_channels = [list(widths) for widths in mlp_channels]
        self.pool_mod = pool_mod
        self.fps_mod_list, self.fps_sample_range_list = fps_mod, fps_sample_range_list
        self.points_sampler = Points_Sampler(self.num_point, fps_mod, fps_sample_range_list)
        # dilated grouping: scale i only takes neighbours beyond the previous scale's radius
        inner = [0] + list(radii[:-1]) if dilated_group else [0] * scales
        self.groupers = nn.ModuleList(
            QueryAndGroup(outer, count, min_radius=lo, use_xyz=use_xyz, normalize_xyz=normalize_xyz,
                          return_grouped_xyz=grouper_return_grouped_xyz,
                          return_grouped_idx=grouper_return_grouped_idx)
            for outer, count, lo in zip(radii, sample_nums, inner))
        self.mlps = nn.ModuleList()

    def _sample_points(self, points_xyz, features, indices, target_xyz):
        """-> (centres (B, M, 3), their indices or None): given indices win, then given target
        coordinates, else the configured sampler picks."""
        if indices is None and target_xyz is not None:
            return target_xyz.contiguous(), None
        if indices is None:
            indices = self.points_sampler(points_xyz, features)
        elif indices.shape[1] != self.num_point[0]:
            raise AssertionError('indices must name num_point centres')
        channel_major = points_xyz.transpose(1, 2).contiguous()
        return gather_points(channel_major, indices).transpose(1, 2).contiguous(), indices

    def _pool_features(self, features):
        if self.pool_mod == 'max':
            return group_max_pool(features).contiguous()  # == F.max_pool2d([1, nsample])
        elif self.pool_mod == 'avg':
            new_features = F.avg_pool2d(features, kernel_size=[1, features.size(3)])
        else:
            raise NotImplementedError
        return new_features.squeeze(-1).contiguous()

    def _mlp_and_pool(self, mlp, grouped, fixed_lead=0):
        """Shared MLP then pooling; with max pooling the last layer's BN + ReLU + max is one
        fused op (the normalised (B, C, M, ns) tensor is never written).  ``fixed_lead`` = leading
        channels of ``grouped`` that nothing differentiates (grouped INPUT coordinates)."""
        layers = list(mlp)
        last = layers[-1] if layers else None
        if self.pool_mod == 'max' and all(isinstance(l, ConvModule) for l in layers) and \
                fused_mlp.sa_stack_supported(backend_for(grouped), grouped, layers):
            return fused_mlp.sa_stack(grouped, layers, fixed_lead)
        if self.pool_mod == 'max' and all(isinstance(l, ConvModule) for l in layers) and \
                fused_mlp.sa_stack_eval_supported(backend_for(grouped), grouped, layers):
            return fused_mlp.sa_stack_eval(grouped, layers)
        if (self.pool_mod == 'max' and isinstance(last, ConvModule) and last.act_fused
                and isinstance(last.norm, FusedBNReLU2d)):
            x = grouped
            for layer in layers[:-1]:
                x = layer(x)
            return last.norm.forward_max_pool(last.conv(x)).contiguous()
        return self._pool_features(mlp(grouped))

    def sample_and_group_indices(self, points_xyz):
        """The weight-independent half of forward(): FPS indices, sampled centres and the
        ball-query indices of every scale.  A training loop can run this for the NEXT batch
        on a side stream while the current step computes (see bench.py)."""
        with spatial_index_scope():   # FPS hands its sorted scene to the ball queries below
            new_xyz, indices = self._sample_points(points_xyz, None, None, None)
            group_idx = [g.ball_indices(points_xyz, new_xyz) for g in self.groupers]
        # ahead of the step only where a feature gradient is plausible AND the index is cheap: the
        # 40 000-point input level carries input features (no gradient); a backward that does need an
        # index over more points builds it itself (group_points.QueryGroupCat)
        n = points_xyz.shape[1]
        group_csr = [inverted_index(i, n) if n <= 8192 else None for i in group_idx]
        return dict(indices=indices, new_xyz=new_xyz, group_idx=group_idx, group_csr=group_csr)

    def forward(self, points_xyz, features=None, indices=None, target_xyz=None, precomputed=None):
        with spatial_index_scope():   # FPS hands its sorted scene to this forward's ball queries
            return self._forward(points_xyz, features, indices, target_xyz, precomputed)

    def _forward(self, points_xyz, features, indices, target_xyz, precomputed):
        new_features_list = []
        if (precomputed is None and indices is None and target_xyz is None and len(self.groupers) == 1
                and points_xyz.requires_grad and isinstance(self.groupers[0], QueryAndGroup)
                and sample_query_group_supported(points_xyz, features, self.groupers[0])):
            # coordinates computed by the network (vote aggregation): sample + group as ONE autograd
            # node with a native coordinate gradient (group_points.SampleQueryGroupCat)
            g = self.groupers[0]
            indices = self.points_sampler(points_xyz, features)
            new_xyz, grouped, _ = SampleQueryGroupCat.apply(
                points_xyz, features, indices, float(g.min_radius), float(g.max_radius),
                int(g.sample_num), bool(g.normalize_xyz))
            return new_xyz, self._mlp_and_pool(self.mlps[0], grouped, 0), indices
        if precomputed is not None:
            new_xyz, indices = precomputed['new_xyz'], precomputed['indices']
        else:
            new_xyz, indices = self._sample_points(points_xyz, features, indices, target_xyz)
        for i in range(len(self.groupers)):
            if precomputed is not None:
                csr = precomputed.get('group_csr') or [None] * len(self.groupers)
                grouped_results = self.groupers[i](points_xyz, new_xyz, features,
                                                   idx=precomputed['group_idx'][i], csr=csr[i])
            else:
                grouped_results = self.groupers[i](points_xyz, new_xyz, features)
            g = self.groupers[i]
            lead = 3 if (getattr(g, 'use_xyz', False) and features is not None
                         and not points_xyz.requires_grad and not new_xyz.requires_grad) else 0
            new_features_list.append(self._mlp_and_pool(self.mlps[i], grouped_results, lead))
        pooled = new_features_list[0] if len(new_features_list) == 1 \
            else torch.cat(new_features_list, dim=1)
        return new_xyz, pooled, indices


def _shared_mlp(widths, norm_cfg, **conv_kw):
    """1x1 Conv2d -> norm -> ReLU per consecutive width pair, sub-modules ``layer0``, ``layer1`` ..."""
    return nn.Sequential(OrderedDict(
        (f'layer{j}', ConvModule(cin, cout, kernel_size=(1, 1), stride=(1, 1),
                                 conv_cfg=dict(type='Conv2d'), norm_cfg=norm_cfg, **conv_kw))
        for j, (cin, cout) in enumerate(zip(widths, widths[1:]))))


class PointSAModuleMSG(BasePointSAModule):
    """Multi-scale grouping (point_sa_module.py:214-290): one shared MLP per scale; with
    ``use_xyz`` the grouped relative coordinates are three extra input channels."""

    def __init__(self, num_point, radii, sample_nums, mlp_channels, fps_mod=['D-FPS'],
                 fps_sample_range_list=[-1], dilated_group=False,
                 norm_cfg=dict(type='BN2d'), use_xyz=True, pool_mod='max',
                 normalize_xyz=False, bias='auto'):
        super().__init__(num_point, radii, sample_nums, mlp_channels, fps_mod=fps_mod,
                         fps_sample_range_list=fps_sample_range_list,
                         dilated_group=dilated_group, use_xyz=use_xyz, pool_mod=pool_mod,
                         normalize_xyz=normalize_xyz)
        for widths in self.mlp_channels:
            if use_xyz:
                widths[0] += 3
            self.mlps.append(_shared_mlp(widths, norm_cfg, bias=bias))


class PointSAModule(PointSAModuleMSG):
    """The single-scale case (point_sa_module.py:293-341)."""

    def __init__(self, mlp_channels, num_point=None, radius=None, num_sample=None,
                 norm_cfg=dict(type='BN2d'), use_xyz=True, pool_mod='max',
                 fps_mod=['D-FPS'], fps_sample_range_list=[-1], normalize_xyz=False):
        super().__init__(num_point, [radius], [num_sample], [list(mlp_channels)],
                         fps_mod=fps_mod, fps_sample_range_list=fps_sample_range_list,
                         norm_cfg=norm_cfg, use_xyz=use_xyz, pool_mod=pool_mod,
                         normalize_xyz=normalize_xyz)


class PointFPModule(nn.Module):
    """Feature propagation (point_fp_module.py:10-78): every target point takes the
    inverse-distance blend of its three nearest source points' features, joins its own
    features, and the result goes through a shared MLP."""

    def __init__(self, mlp_channels: List[int], norm_cfg: dict = dict(type='BN2d')):
        super().__init__()
        self.fp16_enabled = False
        self.mlps = _shared_mlp(list(mlp_channels), norm_cfg)

    @staticmethod
    def interpolation_taps(target, source):
        """3-NN indices, inverse-distance weights and the inverted index of the indices
        (:56-61).  Coordinates only: a training loop may compute them ahead of the step."""
        dist, idx = three_nn(target, source)
        closeness = 1.0 / (dist + 1e-8)
        weight = (closeness / closeness.sum(dim=2, keepdim=True)).contiguous()
        return idx, weight, inverted_index(idx, source.shape[1])

    def forward(self, target, source, target_feats, source_feats, taps=None):
        if source is None:     # one global feature vector, broadcast to every target point
            spread = source_feats.expand(*source_feats.shape[:2], target.shape[1])
        else:
            idx, weight, csr = taps if taps is not None else self.interpolation_taps(target, source)
            spread = three_interpolate(source_feats, idx, weight, csr)
        joined = spread if target_feats is None else torch.cat([spread, target_feats], dim=1)
        layers = list(self.mlps)
        if all(isinstance(l, ConvModule) and l.act_fused for l in layers):
            shapes = [(l.conv.in_channels, l.conv.out_channels) for l in layers]
            norms = [l.norm for l in layers]
            if fused_mlp.stack1d_supported(backend_for(joined), joined, shapes, norms, which=fused_mlp.FPROP):
                # the shared MLP as one fused chain on the layer kernel (point_fp_module.py:31-37)
                return fused_mlp.stack1d(joined, [l.conv for l in layers], norms)
        return self.mlps(joined.unsqueeze(-1)).squeeze(-1)


SA_MODULES = {'PointSAModule': PointSAModule, 'PointSAModuleMSG': PointSAModuleMSG}


def build_sa_module(cfg, *args, **kwargs):
    """``dict(type=<name in SA_MODULES>, **keywords)`` -> module (builder.py:6-38); None means
    a plain PointSAModule."""
    if cfg is None:
        cfg = dict(type='PointSAModule')
    if not isinstance(cfg, dict):
        raise TypeError('cfg must be a dict')
    if 'type' not in cfg:
        raise KeyError('the cfg dict must contain the key "type"')
    options = dict(cfg)
    kind = options.pop('type')
    if kind not in SA_MODULES:
        raise KeyError(f'Unrecognized module type {kind}')
    return SA_MODULES[kind](*args, **kwargs, **options)
