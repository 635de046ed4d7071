"""conv -> BN -> ReLU chains of the grouped per-seed MLPs on the native layer kernel.

The reference evaluates a shared MLP op by op -- mmcv ``ConvModule(Conv2d 1x1, BN2d, ReLU)`` three
times, then ``F.max_pool2d([1, nsample])`` (point_sa_module.py:277-289, 136-158).  Here a layer is
ONE launch of ``nesie_pw_layer_forward``: the previous layer's folded BatchNorm + ReLU is applied
to the operand on its way to the matrix cores, the layer's own batch statistics (and, for the
last layer, the max / min over the neighbourhood) leave from the accumulators, and the normalised
activations are never written.  Per layer the forward moves its tensor through HBM twice (read
the previous raw conv output, write this one) instead of five times.

The backward is hand-written too: per layer ONE input-gradient launch (the same layer kernel on
the transposed weight view, ``nesie_pw_dgrad_bn_reduce``: its epilogue also leaves the two sums of
the BatchNorm backward), one weight-gradient launch on the matrix cores with the activation
recomputed on load (``nesie_pw_wgrad``), and the BatchNorm + ReLU apply pass from the RAW conv
output and the saved (scale, bias, mean, invstd) (``nesie_bn_relu_backward_apply``; the mask is
re-derived with the same fused multiply-add the forward's operand load used).

Same function as the module-by-module path (which stays the CPU checker's and serves shapes the
kernels are not built for); differences are fp32 summation order and fma-vs-mul/add rounding.
"""
import os as _os

import torch
from torch.autograd import Function

from .. import grad_slots
from ..kernels import backend_for


# Tests flip this to obtain the module-by-module evaluation of the same network on the device.
ENABLED = True

# Ownership hand-over of a freshly allocated gradient buffer between MiniTailFn.backward and
# MiniHeadFn.backward: the producer marks the TENSOR OBJECT (an attribute, not its address -- an
# address outlives the tensor in the caching allocator and could be handed to an unrelated gradient
# later); whatever autograd builds from it for a second consumer is a new object without the mark.
_OWNED_MARK = '_nesie_fresh_grad'


# Layers wider than one workgroup's accumulators (the 1-D chains' 256 x 256 / 256 x 512 at 8 x 1024
# positions) ran in nesie_pw_wgrad as column blocks, one launch + one partial reduction each; at
# these sizes (one 32-position tile per workgroup, 32 MB of partials per block) a transposed copy +
# one rocBLAS GEMM was faster (553 / 557 vs 548 / 548 scenes/s, same-box A/B), so round 3 kept
# rocBLAS there.  Round 4: such shapes run TILED (``pw_wgrad_tiled``: one launch over 64 x 64 blocks
# of the product, split-K with a fixed-order reduction, 8 MB of partials) -- no transposed copies,
# no rocBLAS.  NESIE_WGRAD_WIDE=1 forces the native kernel for every shape it supports,
# NESIE_WGRAD_TILED=0 switches the tiled mode off (A/B).
WIDE_WGRAD = _os.environ.get('NESIE_WGRAD_WIDE', '0') != '0'


def _dst(slot, like, *shape):
    """Where a backward kernel writes a parameter gradient: the parameter's slot of the flat
    gradient vector (``grad_slots.take`` in the forward) or a fresh tensor."""
    return slot.view(*shape) if slot is not None else like.new_empty(*shape)


def _into(slot, g):
    """A gradient that some other op produced, moved into the slot that was taken for it."""
    if slot is None or g is None:
        return g
    slot.view(g.shape).copy_(g)
    return slot.view(g.shape)


def _wgrad(backend, dy, x, x_coef, ng=1, slot=None):
    """dW (ng, Cout, Cin) = sum over n % ng == g of dy[n] @ act(x[n])^T; act = relu(scale * x +
    bias) when x_coef (ng * Cin, 4).  ``slot``: the weight's gradient slot (written in place)."""
    nb, co, p = dy.shape
    ci = x.shape[1]
    if backend.pw_wgrad_tiled(nb, ng, co, ci, p) or (
            backend.pw_wgrad_supported(co, ci, p) and (WIDE_WGRAD or (ci <= 320 if co <= 128 else ci <= 128))):
        dw = _dst(slot, dy, ng, co, ci)
        # (a slot of the flat gradient vector is read by nobody before the optimiser: its reduction
        # may wait for the one batched launch at FlatTrainState.collect())
        backend.pw_wgrad(dy, x, dw, ng=ng, x_coef=x_coef, x_relu=True, final=slot is not None)
        return dw
    if ng == 1 and ci <= 8 and backend.conv_wgrad_supported(co, ci):
        dw = _dst(slot, dy, co, ci)
        backend.conv_wgrad(dy, x, dw, x_coef=x_coef, x_relu=x_coef is not None)
        return dw.view(1, co, ci)
    return _into(slot, _wgrad_aten(backend, dy, x, x_coef, ng))


def _wgrad_aten(backend, dy, x, x_coef, ng):
    nb, co, p = dy.shape
    ci = x.shape[1]
    out = []
    for g in range(ng):      # shapes outside the native kernels (short rows, wide layers)
        dyg, xg = dy[g::ng], x[g::ng]
        cg = None if x_coef is None else x_coef[g * ci:(g + 1) * ci].contiguous()
        if ci <= 8 and backend.conv_wgrad_supported(co, ci):
            dw = dy.new_empty(co, ci)
            backend.conv_wgrad(dyg, xg, dw, x_coef=cg, x_relu=cg is not None)
            out.append(dw)
            continue
        a, d = xg.contiguous(), dyg.contiguous()
        if cg is not None:
            a2 = torch.empty_like(a)
            backend.affine_relu_forward(a, cg, True, a2)
            a = a2
        b = d.shape[0]
        if p <= 2048 and b > 1:
            out.append(torch.mm(d.transpose(0, 1).reshape(co, b * p), a.transpose(0, 1).reshape(ci, b * p).t()))
        else:
            out.append(torch.bmm(d, a.transpose(1, 2)).sum(0))
    return out[0].unsqueeze(0) if ng == 1 else torch.stack(out)


# NESIE_FOLD_NORM_BWD=0: A/B switch -- the BatchNorm + ReLU backward's apply pass runs as its own
# launch (nesie_bn_relu_backward_apply) in front of the weight gradient instead of inside it
FOLD_NORM_BWD = _os.environ.get('NESIE_FOLD_NORM_BWD', '1') != '0'
# the pooled last layer of an SA stack without its dense pre-pool tensor (csrc/pool_tail.hip): the
# forward keeps (pooled, arg-max, raw extremum) only, the backward goes through
# dZ = sparse + alpha + beta Z.  0: the dense form (A/B switch)
POOL_TAIL = _os.environ.get('NESIE_POOL_TAIL', '1') != '0'


def _norm_backward_wgrad(backend, da, z, gamma, coef, part, src, src_coef, need_w, ng=1, need_dz=True,
                         slots=(None, None, None)):
    """The backward of relu(bn(z)) given da (its gradient) and the reduction partials the
    input-gradient launch left, and the weight gradient dz . act(src)^T of the conv that produced z:
    -> (dz, dw | None, dgamma, dbeta).  One launch where the layer kernel's weight gradient serves
    the shape (``nesie_pw_wgrad_bn_backward``: dz is formed on the operand load and written over
    da), the apply pass + ``_wgrad`` otherwise.  ``need_dz`` False (the layer's input needs no
    gradient) and a skinny input (Cin <= 8, the first layer of SA1): ``nesie_conv_wgrad_bn`` forms
    dz on its load and never writes it -> dz None.  ``slots`` = gradient slots of (weight, gamma,
    beta): the kernels write there."""
    nb, co, p = da.shape
    ci = src.shape[1]
    s_w, s_g, s_b = slots
    dgamma, dbeta = _dst(s_g, da, ng * co), _dst(s_b, da, ng * co)
    if FOLD_NORM_BWD and need_w and backend.pw_wgrad_bn_supported(co, ci, p):
        dw = _dst(s_w, da, ng, co, ci)
        backend.pw_wgrad_bn_backward(da, z, coef, gamma, part, src, da, dw, dgamma, dbeta, ng=ng,
                                     x_coef=src_coef, final=slots[0] is not None)
        return da, dw, dgamma, dbeta
    if FOLD_NORM_BWD and need_w and not need_dz and ng == 1 and ci <= 8 and backend.conv_wgrad_supported(co, ci):
        bnb = backend.pw_bnb_coef(part, coef, gamma, float(nb) * float(p), dgamma, dbeta)
        dw = _dst(s_w, da, co, ci)
        backend.conv_wgrad(da, src, dw, x_coef=src_coef, x_relu=src_coef is not None, bn_z=z, bnb=bnb)
        return None, dw.view(1, co, ci), dgamma, dbeta
    dz = torch.empty_like(da)
    b = nb // ng
    backend.bn_relu_backward_apply(da.view(b, ng * co, p), z.view(b, ng * co, p), gamma, None, coef, part,
                                   dz.view(b, ng * co, p), dgamma, dbeta)
    dw = _wgrad(backend, dz, src, src_coef, ng=ng, slot=s_w) if need_w else None
    return dz, dw, dgamma, dbeta


# SA1's first activation (4 -> 64 over 10^6 positions, 268 MB) is rebuilt from the 17 MB input wherever
# it is an operand instead of being stored (include/nesie_ops.h, round 5).  0 = store it (A/B switch).
SA1_K4 = _os.environ.get('NESIE_SA1_K4', '1') != '0'
SA1_K4_FUSED = _os.environ.get('NESIE_SA1_K4_FUSED', '1') != '0'     # (A/B: 0 = two launches with dZ in between)


class SAStackFn(Function):
    """x (B, C0, M, ns) -> max_ns relu(bn_L(conv_L(... relu(bn_1(conv_1(x)))))) (B, C_L, M).
    ``fixed_lead`` = number of leading input channels that are inputs of the step (grouped
    coordinates of the backbone levels: nothing consumes their gradient).  It must be 0 when the
    coordinates were computed by the network (vote aggregation groups the predicted votes)."""

    @staticmethod
    def forward(ctx, x, bufs, fixed_lead, *params):
        backend = backend_for(x)
        x = x.contiguous()
        B, c0, M, ns = x.shape
        P = M * ns
        L = len(params) // 3
        x3 = x.view(B, c0, P)
        ys, coefs = [], []
        coef = None
        pool_group = 16 if ns == 16 else 32
        pool_out = None
        need = ctx.needs_input_grad
        k4_ok = getattr(backend, 'k4_supported', None)
        ctx.k4 = bool(SA1_K4 and L >= 3 and k4_ok is not None and not need[0]
                      and (all(need[3:9]) or not any(need))
                      and k4_ok(c0, params[0].shape[0], params[3].shape[0], P))
        w0c = params[0].reshape(params[0].shape[0], c0) if ctx.k4 else None
        for l in range(L):
            w, gamma, beta = params[3 * l:3 * l + 3]
            rm, rv, momentum, eps = bufs[l]
            cout, cin = w.shape[0], w.shape[1]
            w2 = w.reshape(1, cout, cin)
            src = x3 if l == 0 else ys[-1]
            last = l == L - 1
            tail = bool(last and l > 0 and POOL_TAIL and backend.pool_tail_supported(cin, cout, P, ns))
            # (tail: the raw output is never written; k4: nor is the first layer's)
            y = None if (tail or (ctx.k4 and l == 0)) else x.new_empty(B, cout, P)
            new_coef = x.new_empty(cout, 4)
            if ctx.k4 and l == 1:
                part = x.new_empty(1, backend.pw_stat_slots(B, 1, cin, cout, P), cout, 4)
                backend.pw_layer_forward_k4(x3, w0c, w2[0], coef, y, part)
                backend.pw_stats_finalize(part, gamma, beta, rm, rv, momentum, eps, new_coef)
            elif ctx.k4 and l == 0:
                k4_mom = backend.k4_moments(x3)
                backend.k4_stat_finalize(k4_mom, w0c, gamma, beta, rm, rv, momentum, eps, float(B) * float(P), new_coef)
            elif l == 0 and cin <= 8 and not last:
                part = x.new_empty(backend.mlp_stream_parts(B, P), cout, 2)
                backend.mlp_stream_forward(src, w2[0].contiguous(), y, part)
                backend.mlp_stat_finalize(part, B * P, gamma, beta, rm, rv, momentum, eps, new_coef)
            else:
                part = x.new_empty(1, backend.pw_stat_slots(B, 1, cin, cout, P), cout, 4)
                if last:
                    g = P // pool_group
                    pool_out = (x.new_empty(B, cout, g), x.new_empty(B, cout, g),
                                torch.empty(B, cout, g, dtype=torch.uint8, device=x.device),
                                torch.empty(B, cout, g, dtype=torch.uint8, device=x.device))
                backend.pw_layer_forward(src, w2, in_coef=coef, in_relu=True, y=y, stat_part=part,
                                         pool_group=pool_group if last else 0, pool_min=last,
                                         pool_out=pool_out)
                backend.pw_stats_finalize(part, gamma, beta, rm, rv, momentum, eps, new_coef)
            coef = new_coef
            ys.append(y)
            coefs.append(coef)
        cl = params[3 * (L - 1)].shape[0]
        pooled = x.new_empty(B, cl, M)
        argmax = torch.empty(B, cl, M, dtype=torch.uint8, device=x.device)
        ctx.tail = ys[-1] is None
        if ctx.tail:
            ys[-1] = x.new_empty(B, cl, M)            # the raw extremum behind every pooled value
        backend.pw_pool_finish(1, P, ns, pool_group, pool_out, coef, True, pooled, argmax,
                               zstar=ys[-1] if ctx.tail else None)
        ctx.L, ctx.ns, ctx.fixed_lead = L, ns, int(fixed_lead)
        # gradient slots of (weight, gamma, beta) per layer -- only when a backward will follow
        ctx.slots = [grad_slots.take(t) if ctx.needs_input_grad[3 + j] else None for j, t in enumerate(params)]
        ctx.save_for_backward(x3, pooled, argmax, *ys, *coefs, *params, *([k4_mom] if ctx.k4 else []))
        ctx.mark_non_differentiable(argmax)
        return pooled

    @staticmethod
    def backward(ctx, g):
        L, ns = ctx.L, ctx.ns
        sv = ctx.saved_tensors
        x3, pooled, argmax = sv[:3]
        ys, coefs = sv[3:3 + L], sv[3 + L:3 + 2 * L]
        params = sv[3 + 2 * L:3 + 5 * L]
        backend = backend_for(g)
        B, c0, P = x3.shape
        M = P // ns
        grads = [None] * (3 * L)
        yl = ys[-1]
        cl = yl.shape[1]
        slots = ctx.slots
        dgamma, dbeta = _dst(slots[3 * (L - 1) + 1], g, cl), _dst(slots[3 * (L - 1) + 2], g, cl)
        grads[3 * (L - 1) + 1], grads[3 * (L - 1) + 2] = dgamma, dbeta
        dx = None
        pending = None          # (da, part) of the layer whose norm backward has not been applied yet
        top = L - 1
        if ctx.tail:
            # last layer without its dense tensors: yl is the raw extremum per (channel, group)
            wl = params[3 * (L - 1)]
            need_w = ctx.needs_input_grad[3 + 3 * (L - 1)]
            dwl = _dst(slots[3 * (L - 1)], g, cl, wl.shape[1]) if need_w else None
            pending = backend.pool_tail_backward(g.contiguous(), pooled, yl, argmax, coefs[-1],
                                                 params[3 * (L - 1) + 1], wl.reshape(cl, wl.shape[1]),
                                                 ys[L - 2], coefs[L - 2], ns, dgamma, dbeta, dw=dwl)
            if need_w:
                grads[3 * (L - 1)] = dwl.view_as(wl)
            top = L - 2
        else:
            # last layer: BatchNorm + ReLU + max backward into the dense raw-output gradient
            dy = torch.empty_like(yl)
            backend.bn_relu_maxpool_backward(g.contiguous(), argmax, yl.view(B, cl, M, ns), pooled,
                                             params[3 * (L - 1) + 1], None, coefs[-1],
                                             dy.view(B, cl, M, ns), dgamma, dbeta)
        for l in range(top, -1, -1):
            w = params[3 * l]
            cout, cin = w.shape[0], w.shape[1]
            w2 = w.reshape(cout, cin)
            src = x3 if l == 0 else ys[l - 1]
            src_coef = None if l == 0 else coefs[l - 1]
            need_w = ctx.needs_input_grad[3 + 3 * l]
            if ctx.k4 and l == 1:
                # second layer over the rebuilt first activation: its fused norm backward + weight
                # gradient, then ONLY the reductions of its input gradient -- they determine the
                # first layer's norm backward and weight gradient (nesie_k4_first_layer_wgrad)
                assert pending is not None
                w0 = params[0]
                w0c = w0.reshape(w0.shape[0], c0)
                dgamma, dbeta = _dst(slots[4], g, cout), _dst(slots[5], g, cout)
                dw = _dst(slots[3], g, 1, cout, cin)
                if SA1_K4_FUSED:     # ... both as one launch: the layer's dZ never leaves the chip
                    part, g_part = backend.pw_wgrad_bn_backward_k4_fused(
                        pending[0], ys[1], coefs[1], params[4], pending[1], x3, w0c, coefs[0], w2, dw, dgamma, dbeta,
                        final=slots[3] is not None)
                else:
                    backend.pw_wgrad_bn_backward_k4(pending[0], ys[1], coefs[1], params[4], pending[1], x3, w0c,
                                                    coefs[0], dw, dgamma, dbeta, final=slots[3] is not None)
                    part, g_part = backend.pw_dgrad_bn_reduce_k4(pending[0], w2.t(), x3, w0c, coefs[0])
                grads[3], grads[4], grads[5] = dw.view_as(w), dgamma, dbeta
                dgamma0, dbeta0 = _dst(slots[1], g, cin), _dst(slots[2], g, cin)
                bnb = backend.pw_bnb_coef(part, coefs[0], params[1], float(B) * float(P), dgamma0, dbeta0)
                dw0 = _dst(slots[0], g, cin, c0)
                backend.k4_first_layer_wgrad(sv[3 + 5 * L], w0c, bnb, g_part, dw0)
                grads[0], grads[1], grads[2] = dw0.view_as(w0), dgamma0, dbeta0
                break
            if pending is not None:     # norm backward of this layer, with its weight gradient
                dy, dw, dgamma, dbeta = _norm_backward_wgrad(
                    backend, pending[0], ys[l], params[3 * l + 1], coefs[l], pending[1], src, src_coef,
                    need_w, need_dz=l > 0 or bool(ctx.needs_input_grad[0]), slots=slots[3 * l:3 * l + 3])
                grads[3 * l + 1], grads[3 * l + 2] = dgamma, dbeta
                if dw is not None:
                    grads[3 * l] = dw.view_as(w)
            elif need_w:
                grads[3 * l] = _wgrad(backend, dy, src, src_coef, slot=slots[3 * l]).view_as(w)
            if l == 0:
                if ctx.needs_input_grad[0]:
                    # the layer kernel serves up to 256 output rows: the (at most 3) coordinate
                    # rows in front are a separate skinny product, or zero when nobody reads them
                    lead = 3 if c0 in (131, 259) else 0
                    dx = dy.new_empty(B, c0, P)
                    if backend.pw_supported(cout, c0 - lead, P):
                        backend.pw_layer_forward(dy, w2[:, lead:].t().unsqueeze(0), y=dx[:, lead:])
                        if lead and ctx.fixed_lead >= lead:
                            dx[:, :lead].zero_()
                        elif lead:
                            torch.bmm(w2[:, :lead].t().unsqueeze(0).expand(B, -1, -1), dy,
                                      out=dx[:, :lead])
                    else:
                        dx = torch.bmm(w2.t().unsqueeze(0).expand(B, -1, -1), dy)
                break
            # gradient of the previous layer's activation (+ the reduction of its norm backward)
            da = dy.new_empty(B, cin, P)
            part = backend.pw_dgrad_bn_reduce(dy, w2.t().unsqueeze(0), ys[l - 1], coefs[l - 1], da)
            pending = (da, part)
        if dx is not None:
            dx = dx.view(B, c0, M, ns)
        return (dx, None, None) + tuple(grads)


def sa_stack_supported(backend, x, layers):
    """True when ``SAStackFn`` serves this shared MLP: native training BatchNorm behind bias-free
    1x1 convs, fp32, nsample in {16, 32, 64}, every layer inside the built tiles."""
    from .norm import FusedBNReLU2d
    if not ENABLED or backend.name != 'hip' or x.dtype != torch.float32 or x.dim() != 4 \
            or len(layers) < 2:
        return False
    B, c0, M, ns = x.shape
    if ns not in (16, 32, 64):
        return False
    P = M * ns
    cin = c0
    for i, layer in enumerate(layers):
        norm = getattr(layer, 'norm', None)
        if not (isinstance(norm, FusedBNReLU2d) and layer.act_fused and layer.conv.bias is None
                and norm.training and norm.affine and norm.track_running_stats
                and norm.momentum is not None):
            return False
        cout = layer.conv.out_channels
        if layer.conv.in_channels != cin:
            return False
        first_stream = i == 0 and cin <= 8 and len(layers) > 1
        if not first_stream and not backend.pw_supported(cin, cout, P):
            return False
        if i > 0 and not backend.pw_supported(cout, cin, P):   # input-gradient product
            return False
        cin = cout
    return True


def sa_stack_eval_supported(backend, x, layers):
    """True when ``sa_stack_eval`` serves this shared MLP: evaluation-mode norms with running
    statistics, no autograd, shapes inside the built tiles."""
    from .norm import FusedBNReLU2d
    if not ENABLED or backend.name != 'hip' or x.dtype != torch.float32 or x.dim() != 4 \
            or len(layers) < 2 or torch.is_grad_enabled():
        return False
    B, cin, M, ns = x.shape
    if ns not in (16, 32, 64):
        return False
    for layer in layers:
        norm = getattr(layer, 'norm', None)
        if not (isinstance(norm, FusedBNReLU2d) and layer.act_fused and layer.conv.bias is None
                and not norm.training and norm.track_running_stats and norm.running_mean is not None
                and layer.conv.in_channels == cin
                and backend.pw_supported(cin, layer.conv.out_channels, M * ns)):
            return False
        cin = layer.conv.out_channels
    return True


def sa_stack_eval(x, layers):
    """Evaluation-mode shared MLP + max pooling on the layer kernel: every layer's folded running
    statistics are the next layer's operand transform, the last layer leaves through the pooled
    tail (test path, ``simple_test``: conv + scale/bias pass + pooling pass per layer otherwise)."""
    backend = backend_for(x)
    x = x.contiguous()
    B, c0, M, ns = x.shape
    P = M * ns
    src, coef = x.view(B, c0, P), None
    pool_group = 16 if ns == 16 else 32
    pool_out = None
    for i, layer in enumerate(layers):
        cout, cin = layer.conv.out_channels, layer.conv.in_channels
        w2 = layer.conv.weight.reshape(1, cout, cin)
        y = x.new_empty(B, cout, P)
        if i == len(layers) - 1:
            g = P // pool_group
            pool_out = (x.new_empty(B, cout, g), x.new_empty(B, cout, g),
                        torch.empty(B, cout, g, dtype=torch.uint8, device=x.device),
                        torch.empty(B, cout, g, dtype=torch.uint8, device=x.device))
            part = x.new_empty(1, backend.pw_stat_slots(B, 1, cin, cout, P), cout, 4)
            backend.pw_layer_forward(src, w2, in_coef=coef, in_relu=True, y=y, stat_part=part,
                                     pool_group=pool_group, pool_min=True, pool_out=pool_out)
        else:
            backend.pw_layer_forward(src, w2, in_coef=coef, in_relu=True, y=y)
        src, coef = y, layer.norm.eval_coef()
    pooled = x.new_empty(B, src.shape[1], M)
    argmax = torch.empty(B, src.shape[1], M, dtype=torch.uint8, device=x.device)
    backend.pw_pool_finish(1, P, ns, pool_group, pool_out, coef, True, pooled, argmax)
    return pooled


def sa_stack(x, layers, fixed_lead=0):
    """Shared MLP + max pooling of a set-abstraction module through ``SAStackFn``;
    ``fixed_lead`` = leading channels of x whose gradient nobody consumes (see SAStackFn)."""
    from . import norm as _norm
    params, bufs = [], []
    for layer in layers:
        n = layer.norm
        params += [layer.conv.weight, n.weight, n.bias]
        bufs.append((n.running_mean, n.running_var, n.momentum, n.eps))
        _norm.count_batch(n.num_batches_tracked)
    return SAStackFn.apply(x, bufs, fixed_lead, *params)


class StackGroups(Function):
    """Several groups of S same-shaped tensors -> one stacked (S, ...) tensor per group, all
    filled by ONE multi-tensor copy (a ``torch.stack`` per group is a launch per group).  The
    gradient of a stacked tensor goes back as S views."""

    @staticmethod
    def forward(ctx, sizes, *tensors):
        outs, dst, off = [], [], 0
        for n in sizes:
            group = tensors[off:off + n]
            o = group[0].new_empty(n, *group[0].shape)
            outs.append(o)
            dst += list(o.unbind(0))
            off += n
        torch._foreach_copy_(dst, list(tensors))
        ctx.sizes = sizes
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        out = [None]
        for n, g in zip(ctx.sizes, grads):
            out += [None] * n if g is None else list(g.unbind(0))
        return tuple(out)


class CatRows(Function):
    """Same-width 2-D tensors (rows_i, C) -> their row concatenation (sum rows_i, C) filled by one
    multi-tensor copy; the gradient goes back as row slices.  1-D tensors concatenate likewise."""

    @staticmethod
    def forward(ctx, *tensors):
        rows = [t.shape[0] for t in tensors]
        out = tensors[0].new_empty(sum(rows), *tensors[0].shape[1:])
        torch._foreach_copy_(list(out.split(rows, 0)), [t.detach() for t in tensors])
        ctx.rows = rows
        return out

    @staticmethod
    def backward(ctx, g):
        return tuple(g.split(ctx.rows, 0))


def stack_groups(groups):
    """[[t_0 .. t_{S-1}], ...] -> [stacked (S, ...) per group]; S = 1 needs no copy at all, and
    neither does a group whose members sit side by side in the flat parameter vector
    (``grad_slots.stacked``: a view, with the group's gradient slot attached); the rest share one
    multi-tensor copy."""
    if len(groups[0]) == 1:
        return [g[0].unsqueeze(0) for g in groups]
    out = [grad_slots.stacked(g) for g in groups]
    rest = [i for i, o in enumerate(out) if o is None]
    if rest:
        flat = [t for i in rest for t in groups[i]]
        for i, o in zip(rest, StackGroups.apply(tuple(len(groups[i]) for i in rest), *flat)):
            out[i] = o
    return out


# ---- MiniPointNet (side_pooling_module.py:343-370) -------------------------------------------
# f = conv3(relu(bn0(c0))); g = max_G f; y = relu(bn1(W_g (g + b3) + W_l f + ...)); out = max_G conv4(y)
# split at the two places where small per-proposal tensors leave the big ones (g, and the
# per-proposal term `small` built from it with ordinary torch ops):
#   MiniHeadFn : c0 -> (f without its bias, max_G of it)
#   MiniTailFn : (f, small) -> max_G conv4(relu(bn1(W_l f + small)))

def _pool_group(G):
    return 16 if G == 16 else 32


def mini_head_kernels(backend, c0, coef0, w3, G):
    """c0 (B,S,H0,P) raw, coef0 (S*H0,4) -> c = W3 . relu(coef0 c0) (B,S,half,P), its max over
    groups of G positions g (B,S,half,P/G) and the arg-max."""
    B, S, H0, P = c0.shape
    half = w3.shape[1]
    pg = _pool_group(G)
    c = c0.new_empty(B, S, half, P)
    npg = P // pg
    pool_out = (c0.new_empty(B * S, half, npg), None,
                torch.empty(B * S, half, npg, dtype=torch.uint8, device=c0.device), None)
    backend.pw_layer_forward(c0.view(B * S, H0, P), w3, ng=S, in_coef=coef0, in_relu=True,
                             y=c.view(B * S, half, P), pool_group=pg, pool_min=False,
                             pool_out=pool_out)
    g = c0.new_empty(B, S, half, P // G)
    arg = torch.empty(B, S, half, P // G, dtype=torch.uint8, device=c0.device)
    backend.pw_pool_finish(S, P, G, pg, pool_out, None, False, g.view(B * S, half, -1),
                           arg.view(B * S, half, -1))
    return c, g, arg


def mini_tail_first(backend, c, small, wl, G):
    """y = W_l c + small (row bias per group of G positions) (B,S,H2,P) + its statistics partials."""
    B, S, half, P = c.shape
    H2 = wl.shape[1]
    y = c.new_empty(B, S, H2, P)
    part = c.new_empty(S, backend.pw_stat_slots(B * S, S, half, H2, P), H2, 4)
    backend.pw_layer_forward(c.view(B * S, half, P), wl, ng=S, row_bias=small.view(B * S, H2, -1),
                             rb_group=G, y=y.view(B * S, H2, P), stat_part=part)
    return y, part


def mini_tail_second(backend, y, coef1, w4, G):
    """max over groups of G positions of W4 . relu(coef1 y) -> (B,S,F,P/G) and the arg-max."""
    B, S, H2, P = y.shape
    F = w4.shape[1]
    pg = _pool_group(G)
    npg = P // pg
    pool_out = (y.new_empty(B * S, F, npg), None,
                torch.empty(B * S, F, npg, dtype=torch.uint8, device=y.device), None)
    backend.pw_layer_forward(y.view(B * S, H2, P), w4, ng=S, in_coef=coef1, in_relu=True,
                             pool_group=pg, pool_min=False, pool_out=pool_out)
    out = y.new_empty(B, S, F, P // G)
    arg = torch.empty(B, S, F, P // G, dtype=torch.uint8, device=y.device)
    backend.pw_pool_finish(S, P, G, pg, pool_out, None, False, out.view(B * S, F, -1),
                           arg.view(B * S, F, -1))
    return out, arg


def _mini_head_forward(backend, c0, c0_part, bufs, G, gamma0, beta0, w3):
    """-> (c, g, arg, coef0): the first norm's statistics from the producer's partials, then
    ``mini_head_kernels``."""
    B, S, H0, P = c0.shape
    rm, rv, momentum, eps = bufs
    coef0 = c0.new_empty(S * H0, 4)
    backend.mlp_stat_finalize(c0_part, B * P, gamma0, beta0, rm, rv, momentum, eps, coef0,
                              channel_major=True)
    c, g, arg = mini_head_kernels(backend, c0, coef0, w3, G)
    return c, g, arg, coef0


def _mini_head_backward(backend, dc, dg, c0, coef0, arg, gamma0, w3, G, need_w3, slot_w3=None):
    """-> (da0 (B*S, H0, P) = gradient of relu(bn0(c0)), reduction partials, dw3)."""
    B, S, H0, P = c0.shape
    half = w3.shape[1]
    # (MiniTailFn.backward hands over a buffer it has just allocated and marks the tensor object:
    # that one is ours to write into; anything else is copied first)
    owned = dc is None or (dc.is_contiguous() and dc._base is None and getattr(dc, _OWNED_MARK, False))
    if dc is None:
        dc = c0.new_zeros(B, S, half, P)
    if dg is not None:
        # the pooled gradient joins the dense one at the arg-max -- in a buffer of our OWN:
        # autograd does not hand a backward ownership of its incoming gradients (the same
        # tensor may be another consumer's gradient, a hook's or a retained one)
        if not owned:
            dc = dc.clone(memory_format=torch.contiguous_format)
        dcv = dc.view(B, S, half, P // G, G)
        backend.group_max_pool_backward_add(dg.contiguous(), arg, dcv)
    else:
        dc = dc.contiguous()
    x0 = c0.view(B * S, H0, P)
    dcf = dc.view(B * S, half, P)
    dw3 = None
    if need_w3:
        # per-net weight gradient: the S nets are the S strided batch subsets
        dw3 = _wgrad(backend, dcf, x0, coef0, ng=S, slot=slot_w3)
    da0 = c0.new_empty(B * S, H0, P)
    part = backend.pw_dgrad_bn_reduce(dcf, w3.transpose(1, 2), x0, coef0, da0, ng=S)
    return da0, part, dw3


class MiniHeadFn(Function):
    """c0 (B, S, H0, K*G) raw first-conv outputs of S stacked nets (+ their (sum, sum^2)
    partials) -> c = W3 . relu(bn0(c0)) (B, S, half, K*G) and g = max_G c (B, S, half, K)."""

    @staticmethod
    def forward(ctx, c0, c0_part, bufs, G, gamma0, beta0, w3):
        backend = backend_for(c0)
        c0 = c0.contiguous()
        c, g, arg, coef0 = _mini_head_forward(backend, c0, c0_part, bufs, G, gamma0, beta0, w3)
        ctx.G = G
        ctx.slots = [grad_slots.take(t) if ctx.needs_input_grad[4 + j] else None
                     for j, t in enumerate((gamma0, beta0, w3))]
        ctx.save_for_backward(c0, coef0, arg, gamma0, beta0, w3)
        ctx.mark_non_differentiable(arg)
        return c, g

    @staticmethod
    def backward(ctx, dc, dg):
        c0, coef0, arg, gamma0, beta0, w3 = ctx.saved_tensors
        backend = backend_for(c0)
        B, S, H0, P = c0.shape
        da0, part, dw3 = _mini_head_backward(backend, dc, dg, c0, coef0, arg, gamma0, w3, ctx.G,
                                             ctx.needs_input_grad[6], ctx.slots[2])
        dgamma, dbeta = _dst(ctx.slots[0], c0, S * H0), _dst(ctx.slots[1], c0, S * H0)
        dc0 = torch.empty_like(c0)
        backend.bn_relu_backward_apply(da0.view(B, S * H0, P), c0.view(B, S * H0, P), gamma0,
                                       None, coef0, part, dc0.view(B, S * H0, P), dgamma, dbeta)
        return dc0, None, None, None, dgamma, dbeta, dw3


class BlendMiniHeadFn(Function):
    """``interpolate.BlendConv`` (the first 1x1 conv of S MiniPointNets through the 3-NN blend,
    side_pooling_module.py:226-243, 346-349) and ``MiniHeadFn`` as ONE autograd node:
    (table (B, M, S*H0), wx (S, H0, 3)) -> (c, g) as MiniHeadFn.

    The first norm's backward is applied by the blend backward on its tile load
    (``nesie_blend_conv_backward_bn``): no apply pass, no dZ tensor.  That needs the blend
    backward to see the gradient of the NORMALISED activation together with (c0, reduction
    coefficients) -- state that must not travel through autograd as if it were a gradient (a
    second consumer of c0, a hook or ``retain_grad`` would silently turn it into garbage).  Inside
    one node c0 and da0 are locals: nothing else can consume, hook or accumulate into them."""

    @staticmethod
    def forward(ctx, table, wx, idx, weight, rel, segs, seg_len, bufs, G, gamma0, beta0, w3):
        backend = backend_for(table)
        table, wx = table.contiguous(), wx.contiguous()
        b, m, pitch = table.shape
        h = pitch // segs
        n = idx.shape[1]
        c0 = table.new_empty(b, segs, h, n // segs)
        c0_part = table.new_empty(segs * h, b * (n // segs // 64), 2)
        backend.blend_conv_forward(table, h, idx, weight, rel, wx, c0, segs, seg_len, h, 0,
                                   stat_partial=c0_part)
        c, g, arg, coef0 = _mini_head_forward(backend, c0, c0_part, bufs, G, gamma0, beta0, w3)
        ctx.G, ctx.dims = G, (segs, seg_len, b, m, pitch, h)
        ctx.slots = [grad_slots.take(t) if ctx.needs_input_grad[9 + j] else None
                     for j, t in enumerate((gamma0, beta0, w3))]
        ctx.save_for_backward(c0, coef0, arg, gamma0, w3, idx, weight, rel)
        ctx.mark_non_differentiable(arg)
        return c, g

    @staticmethod
    def backward(ctx, dc, dg):
        c0, coef0, arg, gamma0, w3, idx, weight, rel = ctx.saved_tensors
        segs, seg_len, b, m, pitch, h = ctx.dims
        backend = backend_for(c0)
        B, S, H0, P = c0.shape
        da0, part, dw3 = _mini_head_backward(backend, dc, dg, c0, coef0, arg, gamma0, w3, ctx.G,
                                             ctx.needs_input_grad[11], ctx.slots[2])
        dgamma, dbeta = _dst(ctx.slots[0], c0, S * H0), _dst(ctx.slots[1], c0, S * H0)
        # (the staged backward writes every row; the atomic form adds into zeros)
        d_table = (c0.new_empty if backend.blend_backward_writes_table(h, idx.shape[1], segs, m) else c0.new_zeros)(b, m, pitch)
        d_wx = c0.new_empty(segs, h, 3)          # (written, not accumulated)
        if FOLD_NORM_BWD:
            bnb = backend.pw_bnb_coef(part, coef0, gamma0, float(B) * float(P), dgamma, dbeta)
            backend.blend_conv_backward(da0.view(B, S, H0, P), h, idx, weight, rel, d_table, d_wx,
                                        segs, seg_len, bn_z=c0, bnb=bnb)
        else:       # A/B switch: the separate apply pass
            dc0 = torch.empty_like(c0)
            backend.bn_relu_backward_apply(da0.view(B, S * H0, P), c0.view(B, S * H0, P), gamma0,
                                           None, coef0, part, dc0.view(B, S * H0, P), dgamma, dbeta)
            backend.blend_conv_backward(dc0, h, idx, weight, rel, d_table, d_wx, segs, seg_len)
        return d_table, d_wx, None, None, None, None, None, None, None, dgamma, dbeta, dw3


def blend_mini_head_supported(backend, table, idx, segs):
    """The blend forward leaves statistics partials (what ``BlendMiniHeadFn`` needs) for stacked
    channel counts that are multiples of 64 over whole 64-query tiles."""
    h, n = table.shape[2] // segs, idx.shape[1]
    return backend.name == 'hip' and h % 64 == 0 and (n // segs) % 64 == 0


class MiniTailFn(Function):
    """c (B, S, half, K*G), small (B, S, H2, K), W_l (S, H2, half), bn1, W4 (S, F, H2) ->
    max_G W4 . relu(bn1(W_l c + small)) (B, S, F, K); the (B, S, F, K*G) tensor is never written."""

    @staticmethod
    def forward(ctx, c, small, bufs, G, wl, gamma1, beta1, w4):
        backend = backend_for(c)
        c, small = c.contiguous(), small.contiguous()
        B, S, half, P = c.shape
        H2, F = wl.shape[1], w4.shape[1]
        rm, rv, momentum, eps = bufs
        y, part = mini_tail_first(backend, c, small, wl, G)
        coef1 = c.new_empty(S * H2, 4)
        backend.pw_stats_finalize(part, gamma1, beta1, rm, rv, momentum, eps, coef1)
        out, arg = mini_tail_second(backend, y, coef1, w4, G)
        ctx.G = G
        ctx.slots = [grad_slots.take(t) if ctx.needs_input_grad[5 + j] else None
                     for j, t in enumerate((gamma1, beta1, w4))]
        ctx.save_for_backward(c, y, coef1, arg, wl, gamma1, beta1, w4)
        ctx.mark_non_differentiable(arg)
        return out

    @staticmethod
    def backward(ctx, dout):
        c, y, coef1, arg, wl, gamma1, beta1, w4 = ctx.saved_tensors
        backend = backend_for(c)
        B, S, half, P = c.shape
        H2, F = wl.shape[1], w4.shape[1]
        G = ctx.G
        # gradient of the last conv's output: the pooled gradient at the arg-max, zero elsewhere
        dz = c.new_empty(B, S, F, P // G, G)
        backend.group_max_pool_backward(dout.contiguous(), arg, dz)
        dzf = dz.view(B * S, F, P)
        yf = y.view(B * S, H2, P)
        dw4 = None
        if ctx.needs_input_grad[7]:
            dw4 = _wgrad(backend, dzf, yf, coef1, ng=S, slot=ctx.slots[2])
        da = c.new_empty(B * S, H2, P)
        part = backend.pw_dgrad_bn_reduce(dzf, w4.transpose(1, 2), yf, coef1, da, ng=S)
        dgamma, dbeta = _dst(ctx.slots[0], c, S * H2), _dst(ctx.slots[1], c, S * H2)
        cf = c.view(B * S, half, P)
        dwl = None
        if (FOLD_NORM_BWD and ctx.needs_input_grad[4] and G in (16, 64)
                and backend.pw_wgrad_bn_supported(H2, half, P)):
            # norm backward + row-bias gradient + weight gradient in one launch (dY over da)
            dsmall = (c.new_zeros if G == 64 else c.new_empty)(B, S, H2, P // G)
            dwl = c.new_empty(S, H2, half)
            backend.pw_wgrad_bn_backward(da, yf, coef1, gamma1, part, cf, da, dwl, dgamma, dbeta, ng=S,
                                         x_coef=None, d_row_bias=dsmall.view(B * S, H2, -1), group=G)
            dyf = da
        else:
            dy = torch.empty_like(y)
            dsmall = c.new_empty(B, S, H2, P // G)
            backend.bn_relu_backward_apply(da.view(B, S * H2, P), y.view(B, S * H2, P), gamma1,
                                           None, coef1, part, dy.view(B, S * H2, P), dgamma, dbeta,
                                           d_row_bias=dsmall.view(B, S * H2, -1), group=G)
            dyf = dy.view(B * S, H2, P)
            if ctx.needs_input_grad[4]:
                dwl = _wgrad(backend, dyf, cf, None, ng=S)
        dc = torch.empty_like(c)
        backend.pw_layer_forward(dyf, wl.transpose(1, 2), ng=S, y=dc.view(B * S, half, P))
        setattr(dc, _OWNED_MARK, True)      # nobody else holds this gradient buffer
        return dc, dsmall, None, None, dwl, dgamma, dbeta, dw4


def mini_pointnets_fused_supported(backend, c0, c0_part, G):
    if not ENABLED or backend.name != 'hip' or c0.dtype != torch.float32 or c0_part is None \
            or G not in (16, 64):
        return False
    B, S, H0, P = c0.shape[0], c0.shape[1], c0.shape[2], c0.shape[3] * c0.shape[4]
    return (c0_part.numel() > 0 and backend.pw_supported(H0, H0 // 2, P)
            and backend.pw_supported(H0 // 2, H0, P))


# ---- 1-D per-seed / per-proposal stacks ----------------------------------------------------------
# VoteModule (vote_module.py:65-74, 85-147), the prediction head's trunk and its three output
# convolutions (reliable_conv_bbox_module.py:112-141, 144-177), the feature-propagation MLPs
# (point_fp_module.py:31-37) and the quality head's score heads (side_pooling_module.py:55-78, 318):
# chains of Conv1d(1x1) [-> BatchNorm1d -> ReLU] over (B, C, P) with P = 256 .. 1024.  The reference
# runs them op by op; here a chain is ONE autograd function on the layer kernel, like the
# set-abstraction stacks: per layer one forward launch (previous norm + ReLU folded into the operand
# load, this layer's statistics from the accumulators), and in the backward one input-gradient
# launch with the norm reduction, one weight-gradient launch and one norm apply pass.

class SplitXyzFeat(Function):
    """w (S, H, 3 + C) -- the stacked first-conv weights of S MiniPointNets, whose input is
    cat[rel_xyz (3), blended features (C)] (side_pooling_module.py:226-243, 346-349) -> (w_xyz (S, H, 3)
    contiguous, w_feat (S * H, C) as a strided VIEW).  Two plain slices cost autograd two zero-filled
    (S, H, 3 + C) tensors, two slice copies and an addition on the way back; here the two gradients
    are written side by side with ONE concatenation, straight into the weight's slot of the flat
    gradient vector when there is one."""

    @staticmethod
    def forward(ctx, w):
        S, H, K = w.shape
        ctx.dims = (S, H, K)
        ctx.slot = grad_slots.take(w) if ctx.needs_input_grad[0] else None
        return w[:, :, :3].contiguous(), w[:, :, 3:].reshape(S * H, K - 3)

    @staticmethod
    def backward(ctx, d_xyz, d_feat):
        S, H, K = ctx.dims
        like = d_xyz if d_xyz is not None else d_feat
        if d_xyz is None:
            d_xyz = like.new_zeros(S, H, 3)
        if d_feat is None:
            d_feat = like.new_zeros(S * H, K - 3)
        out = ctx.slot.view(S, H, K) if ctx.slot is not None else like.new_empty(S, H, K)
        torch.cat([d_xyz.view(S, H, 3), d_feat.view(S, H, K - 3)], dim=2, out=out)
        return out


class AddChannelBias(Function):
    """x (B, S, F, K) + bias (S, F) broadcast over batch and positions (the conv bias a max-pool
    commutes with, added after pooling: side_pooling_module.py:357, 361-368).  The backward hands the
    gradient through untouched and sums the bias gradient with one native launch per call
    (``channel_sum``) instead of autograd's sum-to-size reduction."""

    @staticmethod
    def forward(ctx, x, bias):
        ctx.dims = tuple(x.shape)
        return x + bias.view(1, bias.shape[0], -1, 1)

    @staticmethod
    def backward(ctx, g):
        B, S, F, K = ctx.dims
        db = None
        if ctx.needs_input_grad[1]:
            backend = backend_for(g)
            if hasattr(backend, 'channel_sum'):
                # (scene, net) runs over the batch axis of a (B * S, F, K) view; a channel slice of a
                # wider gradient (the cat in front of the score heads) keeps a uniform stride there
                if g.stride(3) == 1 and g.stride(2) == K and g.stride(0) == S * g.stride(1):
                    gv = g.as_strided((B * S, F, K), (g.stride(1), K, 1))
                else:
                    gv = g.contiguous().view(B * S, F, K)
                db = backend.channel_sum(gv, ng=S).view(S, F)
            else:
                db = g.sum((0, 3))
        return g, db


class Stack1dLayer:
    """Static description of one layer: ``bias`` (the conv has one), ``bn`` = None or
    (running_mean, running_var, momentum, eps) (-> conv, BatchNorm, ReLU)."""

    def __init__(self, bias, bn):
        self.bias, self.bn = bool(bias), bn


class Stack1dFn(Function):
    """x (NB, C0, P), batch n uses weight group n % S.  Layers l = 0 .. L-1:
    y_l = W_l . a_{l-1} (+ bias_l), a_l = relu(bn_l(y_l)) for a layer with a norm, a_l = y_l
    without.  Returns a_{L-1}.  Tensors after ``layers``: per layer W (S, Cout, Cin), then bias
    (S * Cout) if the conv has one, then gamma, beta (S * Cout) if it has a norm.

    A conv bias in FRONT of a norm is never added: the mean subtraction removes it from every
    normalised value; the running mean gets it (``pw_stats_finalize(chan_bias=)``) and its
    gradient -- identically zero -- is returned as a zero tensor, as ATen does (a None would make a
    per-parameter optimiser skip the parameter: no weight decay on it, "unused parameter" under
    DDP)."""

    @staticmethod
    def forward(ctx, x, layers, S, *tensors):
        backend = backend_for(x)
        x = x.contiguous()
        NB, c0, P = x.shape
        ts = list(tensors)
        per_layer, pos = [], 0
        for lay in layers:
            w = ts[pos]; pos += 1
            b = None
            if lay.bias:
                b = ts[pos]; pos += 1
            gamma = beta = None
            if lay.bn is not None:
                gamma, beta = ts[pos], ts[pos + 1]; pos += 2
            per_layer.append((w, b, gamma, beta))
        src, coef = x, None
        ys, coefs = [], []
        for lay, (w, b, gamma, beta) in zip(layers, per_layer):
            cout = w.shape[1]
            y = x.new_empty(NB, cout, P)
            if lay.bn is not None:
                rm, rv, momentum, eps = lay.bn
                part = x.new_empty(S, backend.pw_stat_slots(NB, S, w.shape[2], cout, P), cout, 4)
                backend.pw_layer_forward(src, w, ng=S, in_coef=coef, in_relu=True, y=y, stat_part=part)
                coef = x.new_empty(S * cout, 4)
                backend.pw_stats_finalize(part, gamma, beta, rm, rv, momentum, eps, coef,
                                          chan_bias=None if b is None else b.detach().contiguous())
            else:
                backend.pw_layer_forward(src, w, ng=S, in_coef=coef, in_relu=True, y=y,
                                         bias=None if b is None else b.detach().contiguous())
                coef = None
            ys.append(y)
            coefs.append(coef)
            src = y
        out = ys[-1]
        if layers[-1].bn is not None:      # the last activation is somebody's input: materialise it
            cl = out.shape[1]
            act = torch.empty_like(out)
            backend.affine_relu_forward(out.view(NB // S, S * cl, P), coefs[-1], True,
                                        act.view(NB // S, S * cl, P))
            out = act
        ctx.layers, ctx.S, ctx.n_tensors = layers, S, len(tensors)
        # gradient slots of the tensors whose gradient a kernel (or a fill) writes: weights, norm
        # scales / shifts, and the identically-zero bias in front of a norm (a plain conv's bias
        # gradient is an ATen reduction: nothing to gain there)
        ctx.slots, pos = [None] * len(tensors), 0
        take = lambda i: grad_slots.take(ts[i]) if ctx.needs_input_grad[3 + i] else None  # noqa: E731
        for lay in layers:
            ctx.slots[pos] = take(pos); pos += 1
            if lay.bias:
                if lay.bn is not None:
                    ctx.slots[pos] = take(pos)
                pos += 1
            if lay.bn is not None:
                ctx.slots[pos], ctx.slots[pos + 1] = take(pos), take(pos + 1)
                pos += 2
        ctx.save_for_backward(x, *ys, *[c for c in coefs if c is not None], *tensors)
        return out

    @staticmethod
    def backward(ctx, dout):
        layers, S = ctx.layers, ctx.S
        L = len(layers)
        sv = list(ctx.saved_tensors)
        x, ys = sv[0], sv[1:1 + L]
        ncoef = sum(1 for lay in layers if lay.bn is not None)
        coef_list = sv[1 + L:1 + L + ncoef]
        ts = sv[1 + L + ncoef:]
        coefs, ci_ = [], 0
        for lay in layers:
            if lay.bn is not None:
                coefs.append(coef_list[ci_]); ci_ += 1
            else:
                coefs.append(None)
        per_layer, pos, slots = [], 0, []
        for lay in layers:
            sl = {'w': pos}
            w = ts[pos]; pos += 1
            b = gamma = beta = None
            if lay.bias:
                sl['b'] = pos; b = ts[pos]; pos += 1
            if lay.bn is not None:
                sl['g'] = pos; gamma, beta = ts[pos], ts[pos + 1]; pos += 2
            per_layer.append((w, b, gamma, beta))
            slots.append(sl)
        backend = backend_for(dout)
        NB, c0, P = x.shape
        B = NB // S
        grads = [None] * ctx.n_tensors
        need = ctx.needs_input_grad
        tslots = ctx.slots

        def zero_bias(l):      # the bias in front of layer l's norm: gradient identically zero
            # With a slot in the flat gradient vector the answer is None: FlatTrainState.collect()
            # zero-fills every parameter that got no gradient (one multi-tensor fill for all of
            # them), so whatever an earlier step, another path or a loaded vector left in the slot
            # is overwritten in every step.  Without a flat state: a zero tensor, as ATen returns
            # (a None would make a per-parameter optimiser skip the parameter).
            sl = tslots[slots[l]['b']]
            return None if sl is not None else torch.zeros_like(per_layer[l][1])
        # a channel slice of a wider gradient (the outputs of several chains concatenated by their
        # consumer) is batch-strided: the layer kernels take a batch stride, no copy needed
        if not (S == 1 and layers[-1].bn is None and dout.stride(2) == 1 and dout.stride(1) == P):
            dout = dout.contiguous()
        # gradient of the last layer's RAW output
        w, b, gamma, beta = per_layer[-1]
        cl = ys[-1].shape[1]
        if layers[-1].bn is not None:
            coef = coefs[-1]
            dz = torch.empty_like(ys[-1])
            dgamma = _dst(tslots[slots[-1]['g']], dout, S * cl)
            dbeta = _dst(tslots[slots[-1]['g'] + 1], dout, S * cl)
            backend.bn_relu_backward(dout.view(B, S * cl, P), ys[-1].view(B, S * cl, P), None, gamma, beta,
                                     None, None, coef, True, dz.view(B, S * cl, P), dgamma, dbeta)
            grads[slots[-1]['g']], grads[slots[-1]['g'] + 1] = dgamma, dbeta
            if 'b' in slots[-1] and need[3 + slots[-1]['b']]:
                grads[slots[-1]['b']] = zero_bias(L - 1)
        else:
            dz = dout
            if b is not None and need[3 + slots[-1]['b']]:
                # (one workgroup per channel in a fixed order: ATen's two-axis reduction took 13 - 18 us)
                grads[slots[-1]['b']] = backend.channel_sum(dz if S == 1 else dz.view(B, S * cl, P))
        dx = None
        pending = None          # (da, part) of the layer whose norm backward has not been applied yet
        for l in range(L - 1, -1, -1):
            w = per_layer[l][0]
            cout, cin = w.shape[1], w.shape[2]
            src = x if l == 0 else ys[l - 1]
            src_coef = None if l == 0 else coefs[l - 1]
            need_w = need[3 + slots[l]['w']]
            if pending is not None:     # norm backward of this layer, with its weight gradient
                dz, dw, dgamma, dbeta = _norm_backward_wgrad(
                    backend, pending[0], ys[l], per_layer[l][2], coefs[l], pending[1], src, src_coef, need_w, ng=S,
                    slots=(tslots[slots[l]['w']], tslots[slots[l]['g']], tslots[slots[l]['g'] + 1]))
                grads[slots[l]['g']], grads[slots[l]['g'] + 1] = dgamma, dbeta
                if dw is not None:
                    grads[slots[l]['w']] = dw.view(per_layer[l][0].shape)
                if 'b' in slots[l] and need[3 + slots[l]['b']]:
                    grads[slots[l]['b']] = zero_bias(l)
            elif need_w:
                grads[slots[l]['w']] = _wgrad(backend, dz, src, src_coef, ng=S,
                                              slot=tslots[slots[l]['w']]).view(per_layer[l][0].shape)
            if l == 0:
                if need[0]:
                    if backend.pw_supported(cout, cin, P):
                        dx = dz.new_empty(NB, cin, P)
                        backend.pw_layer_forward(dz, w.transpose(1, 2), ng=S, y=dx)
                    else:
                        dx = torch.matmul(w.transpose(1, 2).unsqueeze(0), dz.view(B, S, cout, P)).view(NB, cin, P)
                break
            # gradient of the previous layer's activation (+ the reduction of its norm backward)
            da = dz.new_empty(NB, cin, P)
            part = backend.pw_dgrad_bn_reduce(dz, w.transpose(1, 2), ys[l - 1], coefs[l - 1], da, ng=S)
            pending = (da, part)
        return (dx, None, None) + tuple(grads)


# A/B switch (NESIE_STACK1D, default 31): bit 0 vote module, 1 prediction head, 2 feature
# propagation, 3 score heads, 4 the prediction head's output convs -- which chains run through Stack1dFn
STACK1D_MASK = int(_os.environ.get('NESIE_STACK1D', '31'))
VOTE, PRED, FPROP, HEADS, PRED_OUT = 1, 2, 4, 8, 16   # (PRED_OUT: the prediction head's three output convs)


def stack1d_supported(backend, x, shapes, norms, S=1, which=15):
    """True when ``Stack1dFn`` serves the chain: ``shapes`` = [(Cin, Cout), ...], ``norms`` = the
    norm layer (or None) behind each conv; native training BatchNorm1d/2d with a folded ReLU,
    fp32, every forward and input-gradient product inside the built tiles, only the LAST layer
    may come without a norm."""
    from .norm import FusedBNReLU1d, FusedBNReLU2d
    if not ENABLED or backend.name != 'hip' or x.dtype != torch.float32 or not torch.is_grad_enabled() \
            or not (STACK1D_MASK & which):
        return False
    P = x.shape[-1] if x.dim() == 3 else x.numel() // (x.shape[0] * x.shape[1])
    for i, ((cin, cout), norm) in enumerate(zip(shapes, norms)):
        if norm is None:
            if i != len(shapes) - 1:
                return False
        elif not (isinstance(norm, (FusedBNReLU1d, FusedBNReLU2d)) and norm.fuse_relu and norm.training
                  and norm.affine and norm.track_running_stats and norm.momentum is not None):
            return False
        if not backend.pw_supported(cin, cout, P):
            return False
        if i > 0 and not backend.pw_supported(cout, cin, P):
            return False
    return True


def stack1d(x, convs, norms, S=1, weights=None, biases=None, gammas=None, betas=None, stats=None):
    """Run a conv / norm chain through ``Stack1dFn``.  ``convs`` / ``norms``: one module (or, for
    S > 1, a list of S structurally identical modules) per layer; norms[i] None = plain conv.
    ``weights`` etc.: pre-stacked tensors for S > 1 (see side_pooling.batched_heads)."""
    from . import norm as _norm
    layers, tensors = [], []
    for i, (conv, norm) in enumerate(zip(convs, norms)):
        first_conv = conv[0] if isinstance(conv, (list, tuple)) else conv
        first_norm = norm[0] if isinstance(norm, (list, tuple)) else norm
        w = weights[i] if weights is not None else first_conv.weight.flatten(1).unsqueeze(0)
        tensors.append(w)
        has_bias = first_conv.bias is not None
        if has_bias:
            tensors.append(biases[i] if biases is not None else first_conv.bias)
        bn = None
        if first_norm is not None:
            tensors += [gammas[i] if gammas is not None else first_norm.weight,
                        betas[i] if betas is not None else first_norm.bias]
            rm, rv = stats[i] if stats is not None else (first_norm.running_mean, first_norm.running_var)
            bn = (rm, rv, first_norm.momentum, first_norm.eps)
            for n in (norm if isinstance(norm, (list, tuple)) else [norm]):
                _norm.count_batch(n.num_batches_tracked)
        layers.append(Stack1dLayer(has_bias, bn))
    return Stack1dFn.apply(x, tuple(layers), S, *tensors)
