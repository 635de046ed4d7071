"""Mirror of ``mmdet3d/ops/interpolate`` (three_nn.py:8-45, three_interpolate.py:8-63).
``three_nn`` also stands in for ``mmcv.ops.three_nn`` (side_pooling_module.py:8,204)."""
from typing import Tuple

import torch
from torch.autograd import Function

from ..kernels import backend_for


class ThreeNN(Function):
    """target (B,N,3), source (B,M,3) -> (sqrt(dist2) (B,N,3), idx (B,N,3) int32)."""

    @staticmethod
    def forward(ctx, target: torch.Tensor, source: torch.Tensor):
        assert target.is_contiguous()
        assert source.is_contiguous()
        B, N, _ = target.size()
        m = source.size(1)
        dist2 = target.new_empty((B, N, 3))
        idx = target.new_empty((B, N, 3), dtype=torch.int32)
        backend_for(target).three_nn_wrapper(B, N, m, target, source, dist2, idx)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """features (B,C,M), indices (B,n,3), weight (B,n,3) -> (B,C,n)."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, indices: torch.Tensor,
                weight: torch.Tensor, csr=None) -> torch.Tensor:
        """``csr`` (optional, not in the reference) = inverted_index(indices, M): the backward
        then scatters through it (few atomics) instead of three atomics per target."""
        assert features.is_contiguous()
        assert indices.is_contiguous()
        assert weight.is_contiguous()
        B, c, m = features.size()
        n = indices.size(1)
        ctx.three_interpolate_for_backward = (indices, weight, m, csr)
        output = features.new_empty((B, c, n))
        backend_for(features).three_interpolate_wrapper(B, c, m, n, features, indices,
                                                        weight, output)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        idx, weight, m, csr = ctx.three_interpolate_for_backward
        B, c, n = grad_out.size()
        backend = backend_for(grad_out)
        if csr is None and getattr(backend, 'scatter_index', None) is not None:
            csr = backend.scatter_index(idx, m)     # fixed-order scatter (three_interpolate_cuda.cu:61-84: atomicAdd)
        if csr is not None and hasattr(backend, 'three_interpolate_grad_csr'):
            # a channel slice of a wider gradient (the FP module's cat) goes in as it is: batch-strided
            g = grad_out.data
            if not (g.stride(2) == 1 and g.stride(1) == n):
                g = g.contiguous()
            grad_features = grad_out.new_empty((B, c, m))                    # (written in full)
            backend.three_interpolate_grad_csr(g, weight, csr[0], csr[1], grad_features.data)
        else:
            grad_features = grad_out.new_zeros((B, c, m))
            grad_out_data = grad_out.data.contiguous()
            backend.three_interpolate_grad_wrapper(B, c, n, m, grad_out_data, idx, weight,
                                                   grad_features.data)
        return grad_features, None, None, None


three_interpolate = ThreeInterpolate.apply


def three_interpolate_segmented(features_t, indices, weight, out, segs, seg_len, c_offset):
    """Blend point-major ``features_t`` (B,M,C) at the n = K*segs*seg_len queries ordered
    (proposal, face, grid point) into ``out`` (B, segs, c_total, K*seg_len) at channels
    [c_offset, c_offset+C): the per-face contiguous blocks that
    side_pooling_module.py:226-243, 304-313 reaches through cat -> split -> contiguous.
    Not differentiable (the quality head reads detached seed features, :176-181)."""
    assert not (features_t.requires_grad and torch.is_grad_enabled()), \
        'three_interpolate_segmented serves the detached quality-head path only'
    assert features_t.is_contiguous() and indices.is_contiguous() and weight.is_contiguous()
    assert out.is_contiguous()
    backend_for(features_t).blend_conv_forward(features_t, 0, indices, weight, None, None, out,
                                               segs, seg_len, features_t.shape[2], c_offset)
    return out


class BlendConv(Function):
    """First 1x1 conv of the quality head's MiniPointNets evaluated THROUGH the 3-NN blend.

    The reference feeds cat[rel_xyz (3), blend of the seed features (C)] to
    Conv2d(3+C, H, 1, bias=False) (side_pooling_module.py:226-243, 346-349).  The conv is
    linear, so  W . cat[rel, blend(F)] = W_xyz . rel + blend(W_f . F): ``table`` =
    F^T W_f^T (B, M, segs*H) is one small GEMM over the M seeds instead of one over the
    K*G*segs grid points, and the 259-channel feature tensor is never built.
    table (B, M, segs*H) differentiable, wx (segs, H, 3) differentiable, idx/weight/rel
    (B, n, 3) constants -> (B, segs, H, n/segs)."""

    @staticmethod
    def forward(ctx, table, wx, idx, weight, rel, segs, seg_len, with_stats=False):
        """with_stats: also return the per-tile (sum, sum of squares) of the output per stacked
        channel, (segs*H, B*K*G/64, 2), for the norm layer behind it (its statistics pass over
        the output is then skipped)."""
        table, wx = table.contiguous(), wx.contiguous()
        b, m, pitch = table.shape
        h = pitch // segs
        n = idx.shape[1]
        out = table.new_empty(b, segs, h, n // segs)
        backend = backend_for(table)
        partial = None
        if with_stats and backend.name == 'hip' and h % 64 == 0 and (n // segs) % 64 == 0:
            partial = table.new_empty(segs * h, b * (n // segs // 64), 2)
            backend.blend_conv_forward(table, h, idx, weight, rel, wx, out, segs, seg_len, h, 0,
                                       stat_partial=partial)
        else:
            backend.blend_conv_forward(table, h, idx, weight, rel, wx, out, segs, seg_len, h, 0)
        ctx.save_for_backward(idx, weight, rel)
        ctx.dims = (segs, seg_len, b, m, pitch, h)
        if with_stats:
            if partial is None:
                partial = table.new_empty(0)
            ctx.mark_non_differentiable(partial)
            return out, partial
        return out

    @staticmethod
    def backward(ctx, dy, *unused):
        idx, weight, rel = ctx.saved_tensors
        segs, seg_len, b, m, pitch, h = ctx.dims
        backend = backend_for(dy)
        writes = getattr(backend, 'blend_backward_writes_table', lambda *a: False)(h, idx.shape[1], segs, m)
        d_table = (dy.new_empty if writes else dy.new_zeros)(b, m, pitch)
        d_wx = dy.new_empty(segs, h, 3)          # (written, not accumulated)
        backend_for(dy).blend_conv_backward(dy.contiguous(), h, idx, weight, rel, d_table, d_wx,
                                            segs, seg_len)
        return d_table, d_wx, None, None, None, None, None, None


blend_conv = BlendConv.apply


class BlendConvBN(Function):
    """``BlendConv`` followed by the training BatchNorm + ReLU behind it, fused by
    recomputation: the conv output (and its gradient) is re-evaluated from the small table
    wherever it is needed instead of being stored (side_pooling_module.py:226-243, 346-348).
    gamma / beta / running statistics are over the segs*H stacked channels."""

    @staticmethod
    def forward(ctx, table, wx, gamma, beta, idx, weight, rel, running_mean, running_var,
                momentum, eps, segs, seg_len):
        table, wx = table.contiguous(), wx.contiguous()
        b, m, pitch = table.shape
        h = pitch // segs
        n = idx.shape[1]
        out = table.new_empty(b, segs, h, n // segs)
        save_mean, save_invstd = table.new_empty(segs * h), table.new_empty(segs * h)
        fwd_coef = table.new_empty(segs * h, 4)
        backend_for(table).blend_conv_bn_forward(
            table, idx, weight, rel, wx, gamma, beta, running_mean, running_var, momentum, eps,
            out, save_mean, save_invstd, fwd_coef, segs, seg_len)
        ctx.save_for_backward(table, wx, gamma, idx, weight, rel, save_invstd, fwd_coef)
        ctx.dims = (segs, seg_len, h)
        return out

    @staticmethod
    def backward(ctx, dy):
        table, wx, gamma, idx, weight, rel, save_invstd, fwd_coef = ctx.saved_tensors
        segs, seg_len, h = ctx.dims
        d_table = torch.zeros_like(table)
        d_wx = dy.new_zeros(segs, h, 3)
        dgamma, dbeta = dy.new_empty(segs * h), dy.new_empty(segs * h)
        backend_for(dy).blend_conv_bn_backward(dy.contiguous(), table, idx, weight, rel, wx,
                                               gamma, save_invstd, fwd_coef, d_table, d_wx,
                                               dgamma, dbeta, segs, seg_len)
        return (d_table, d_wx, dgamma, dbeta) + (None,) * 9


blend_conv_bn = BlendConvBN.apply
