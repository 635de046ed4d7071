"""Training-mode BatchNorm with the following ReLU fused, on the native kernels.

The reference builds conv -> BN -> ReLU from mmcv ``ConvModule`` and from
``nn.BatchNorm{1,2}d`` + ``nn.ReLU`` (point_sa_module.py:277-289,
side_pooling_module.py:55-78, 346-358); these classes ARE ``nn.BatchNorm{1,2}d`` (same
parameters, buffers and state-dict keys) whose training forward runs
``nesie_bn_relu_forward`` (3 streaming passes) instead of batch_norm + relu (5), and whose
backward runs ``nesie_bn_relu_backward`` (5 passes instead of 10).  In evaluation mode (no
autograd) the running statistics are folded into one scale / bias pair per channel and the layer
is a single read + write pass (``nesie_affine_relu_forward``); with autograd enabled, and on the
injected CPU back end, evaluation uses the ATen path.
"""
import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function

from ..kernels import backend_for


_counter_sink = None


class deferred_bn_counters:
    """Within this context the fused norms queue their ``num_batches_tracked += 1`` (a
    one-element launch per layer, ~50 per VoteNet step) and the exit applies them all with
    one multi-tensor add.  Same buffer values after the block."""

    def __enter__(self):
        global _counter_sink
        self.prev, _counter_sink = _counter_sink, []
        return self

    def __exit__(self, *exc):
        global _counter_sink
        queued, _counter_sink = _counter_sink, self.prev
        if queued and exc[0] is None:
            seen = {}
            for t in queued:  # a layer called twice in the block counts twice
                seen.setdefault(id(t), [t, 0])[1] += 1
            torch._foreach_add_([t for t, _ in seen.values()], [n for _, n in seen.values()])
        return False


def count_batch(counter):
    """``num_batches_tracked += 1`` (queued inside ``deferred_bn_counters``)."""
    if _counter_sink is not None:
        _counter_sink.append(counter)
    else:
        counter.add_(1)


class BNReLUTrain(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, relu,
                row_bias=None, pre_partial=None):
        """pre_partial (C, nslice, 2): unshifted (sum, sum of squares) partials of x left by its
        producer; the statistics pass over x is then skipped."""
        x = x.contiguous()
        if row_bias is not None:
            row_bias = row_bias.contiguous()
        c = x.shape[1]
        y = torch.empty_like(x)
        save_mean = x.new_empty(c)
        save_invstd = x.new_empty(c)
        fwd_coef = x.new_empty(c, 4)  # scale, bias, mean, invstd as the forward applied them
        backend_for(x).bn_relu_forward(x, weight, bias, running_mean, running_var, momentum,
                                       eps, relu, y, save_mean, save_invstd, fwd_coef,
                                       row_bias=row_bias, pre_partial=pre_partial)
        ctx.relu = relu
        ctx.has_row_bias = row_bias is not None
        saved = (x, y, weight, bias, save_mean, save_invstd, fwd_coef)
        ctx.save_for_backward(*(saved + ((row_bias,) if ctx.has_row_bias else ())))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, bias, save_mean, save_invstd, fwd_coef = ctx.saved_tensors[:7]
        row_bias = ctx.saved_tensors[7] if ctx.has_row_bias else None
        # a per-channel row_bias is a conv bias folded into the norm: the mean subtraction removes
        # it, its gradient is identically zero and is returned as a zero tensor (a summation of dx,
        # as the unfused graph does, returns rounding noise around that zero; a None would make a
        # per-parameter optimiser skip the bias -- no weight decay on it)
        d_row_bias = torch.empty_like(row_bias) if ctx.has_row_bias and row_bias.dim() != 1 else None
        d_chan_bias = torch.zeros_like(row_bias) if ctx.has_row_bias and row_bias.dim() == 1 else None
        dy = dy.contiguous()
        c = x.shape[1]
        dx = torch.empty_like(x)
        dgamma, dbeta = x.new_empty(c), x.new_empty(c)
        backend_for(dy).bn_relu_backward(dy, x, y, weight, bias, save_mean, save_invstd, fwd_coef,
                                         ctx.relu, dx, dgamma, dbeta, row_bias=row_bias,
                                         d_row_bias=d_row_bias)
        return dx, dgamma, dbeta, None, None, None, None, None, \
            (d_row_bias if d_chan_bias is None else d_chan_bias), None


class BNReLUMaxPoolTrain(Function):
    """x (B, C, M, ns) -> max_ns relu(bn(x)) (B, C, M): the tail of a set-abstraction MLP
    (ConvModule's BN2d + ReLU, then F.max_pool2d([1, ns]): point_sa_module.py:277-289,
    136-158) without materialising the normalised tensor, forward or backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps):
        x = x.contiguous()
        b, c, m, _ = x.shape
        pooled = x.new_empty(b, c, m)
        argmax = torch.empty(b, c, m, dtype=torch.uint8, device=x.device)
        save_mean, save_invstd = x.new_empty(c), x.new_empty(c)
        fwd_coef = x.new_empty(c, 4)
        backend_for(x).bn_relu_maxpool_forward(x, weight, bias, running_mean, running_var,
                                               momentum, eps, pooled, argmax, save_mean,
                                               save_invstd, fwd_coef)
        ctx.save_for_backward(x, pooled, argmax, weight, save_invstd, fwd_coef)
        ctx.mark_non_differentiable(argmax)
        return pooled

    @staticmethod
    def backward(ctx, g):
        x, pooled, argmax, weight, save_invstd, fwd_coef = ctx.saved_tensors
        c = x.shape[1]
        dx = torch.empty_like(x)
        dgamma, dbeta = x.new_empty(c), x.new_empty(c)
        backend_for(x).bn_relu_maxpool_backward(g.contiguous(), argmax, x, pooled, weight,
                                                save_invstd, fwd_coef, dx, dgamma, dbeta)
        return dx, dgamma, dbeta, None, None, None, None


def eval_coefficients(weight, bias, running_mean, running_var, eps):
    """(C,4) = (scale, bias, 0, 0) of an evaluation-mode BatchNorm: scale = gamma / sqrt(var +
    eps), bias = beta - mean * scale."""
    scale = torch.rsqrt(running_var + eps)
    if weight is not None:
        scale = scale * weight
    shift = -running_mean * scale
    if bias is not None:
        shift = shift + bias
    zero = torch.zeros_like(scale)
    return torch.stack([scale, shift, zero, zero], dim=1).contiguous()


def stacked_eval_coef(layers):
    """(S*C, 4) evaluation coefficients of S same-width norm layers stacked along the channel
    axis: every layer's launch fills its own slice (no concatenation)."""
    if len(layers) == 1:
        return layers[0].eval_coef()
    c = layers[0].num_features
    table = layers[0].running_mean.new_empty(len(layers) * c, 4)
    for i, l in enumerate(layers):
        l.eval_coef(out=table[i * c:(i + 1) * c])
    return table


def affine_relu_eval(x, coef, relu, row_bias=None):
    x = x.contiguous()
    y = torch.empty_like(x)
    backend_for(x).affine_relu_forward(x, coef, relu, y,
                                       None if row_bias is None else row_bias.contiguous())
    return y


class _FusedBNReLU:
    """Mixin over nn.BatchNorm{1,2}d: ``relu=True`` folds the activation into the norm."""

    def _init_fused(self, relu):
        self.fuse_relu = bool(relu)

    def forward(self, x, row_bias=None, pre_partial=None, chan_bias=None):
        """``row_bias`` (B, C, K), optional: normalise ``x + row_bias[..., None]`` for
        ``x`` (B, C, K, G) without materialising the sum (G a power of two in 4..256).
        ``pre_partial`` (C, nslice, 2), optional: (sum, sum of squares) partials of x left by
        its producer (native training path only; ignored otherwise).
        ``chan_bias`` (C), optional: the bias of the convolution that produced ``x``: normalise
        ``x + chan_bias`` with the sum formed in registers (same rounding as the separate add)."""
        backend = backend_for(x)  # raises for CPU tensors without an injected back end
        native = (backend.name == 'hip' and self.training and x.dtype == torch.float32
                  and self.affine and self.track_running_stats and self.momentum is not None)
        native_eval = self._native_eval(x, backend)
        if chan_bias is not None:
            assert row_bias is None and pre_partial is None
            if native or native_eval:
                row_bias = chan_bias
            else:
                x = x + chan_bias.view(1, -1, *([1] * (x.dim() - 2)))
        elif row_bias is not None:
            g = x.shape[-1]
            if not ((native or native_eval) and x.dim() == 4 and 4 <= g <= 256 and g & (g - 1) == 0):
                x, row_bias = x + row_bias.unsqueeze(-1), None
        if native_eval:
            return affine_relu_eval(x, self.eval_coef(), self.fuse_relu, row_bias)
        if native:
            if _counter_sink is not None:
                _counter_sink.append(self.num_batches_tracked)
            else:
                self.num_batches_tracked.add_(1)
            return BNReLUTrain.apply(x, self.weight, self.bias, self.running_mean,
                                     self.running_var, self.momentum, self.eps, self.fuse_relu,
                                     row_bias, pre_partial if row_bias is None else None)
        y = super().forward(x)
        return F.relu(y) if self.fuse_relu else y


    def eval_coef(self, out=None):
        """(C,4) = (scale, shift, 0, 0) of this layer in evaluation mode, recomputed on every call
        by one native launch (``nesie_bn_eval_coef``).  NOT cached: the running statistics are
        written by the training kernels through raw pointers, the parameters through the flat
        optimiser vector and the EMA swap through ``.data`` copies -- none of which moves a
        tensor version counter, so a cache keyed on them served stale coefficients in any
        train -> eval -> train -> eval loop and across ``EMATeacher.swap``.  ``out`` (C,4),
        optional: a slice of a stacked table to fill in place."""
        c = self.num_features
        coef = self.running_mean.new_empty(c, 4) if out is None else out
        with torch.no_grad():
            backend_for(self.running_mean).bn_eval_coef(
                None if self.weight is None else self.weight.detach(),
                None if self.bias is None else self.bias.detach(),
                self.running_mean, self.running_var, self.eps, coef)
        return coef

    def _native_eval(self, x, backend):
        return (backend.name == 'hip' and not self.training and self.track_running_stats
                and self.running_mean is not None and x.dtype == torch.float32
                and not torch.is_grad_enabled())

    def forward_max_pool(self, x):
        """relu(bn(x)) followed by the max over the last axis of x (B, C, M, ns) -> (B, C, M),
        in one fused pass when the native training path applies; else the two-step form."""
        backend = backend_for(x)
        ns = x.shape[-1]
        native = (backend.name == 'hip' and self.training and x.dtype == torch.float32
                  and self.affine and self.track_running_stats and self.momentum is not None
                  and self.fuse_relu and x.dim() == 4 and 4 <= ns <= 64 and ns & (ns - 1) == 0)
        if (self._native_eval(x, backend) and self.fuse_relu and x.dim() == 4
                and 4 <= ns <= 64 and ns & (ns - 1) == 0):
            x = x.contiguous()
            pooled = x.new_empty(x.shape[:3])
            argmax = torch.empty(x.shape[:3], dtype=torch.uint8, device=x.device)
            backend.affine_relu_maxpool_forward(x, self.eval_coef(), pooled, argmax)
            return pooled
        if not native:
            from .pool import group_max_pool
            return group_max_pool(self.forward(x))
        if _counter_sink is not None:
            _counter_sink.append(self.num_batches_tracked)
        else:
            self.num_batches_tracked.add_(1)
        return BNReLUMaxPoolTrain.apply(x, self.weight, self.bias, self.running_mean,
                                        self.running_var, self.momentum, self.eps)


class FusedBNReLU1d(_FusedBNReLU, nn.BatchNorm1d):
    def __init__(self, num_features, relu=True, **kw):
        nn.BatchNorm1d.__init__(self, num_features, **kw)
        self._init_fused(relu)


class FusedBNReLU2d(_FusedBNReLU, nn.BatchNorm2d):
    def __init__(self, num_features, relu=True, **kw):
        nn.BatchNorm2d.__init__(self, num_features, **kw)
        self._init_fused(relu)
