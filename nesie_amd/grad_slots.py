"""Where a parameter's gradient (and, for stacked parameters, the parameter itself) lives in the
flat training state (``dp.FlatTrainState``), so that

* the hand-written backward kernels write a weight / scale / shift gradient STRAIGHT into its slot
  of the flat gradient vector (autograd then hands over a view of that slot and
  ``FlatTrainState.collect`` has nothing to copy for it), and
* S structurally identical modules whose tensors the step uses stacked -- the six side
  MiniPointNets and the six side score heads of the quality head (side_pooling_module.py:55-78,
  304-321) -- keep each such tensor kind CONTIGUOUS in the flat parameter vector: the stacked
  (S, ...) operand is then a view, not a per-step multi-tensor copy, and its gradient is one slot.

The reference reaches the same tensors through ``torch.stack`` / per-parameter ``.grad``s; this is
bookkeeping only -- no arithmetic changes.  Nothing here is required: without a registered state
every lookup answers None and callers allocate as before.
"""
import weakref

import torch
from torch.autograd import Function

# id(parameter) -> (weakref to it, gradient view, owner key)
_PARAM = {}
# tuple(id(parameter) ...) -> (parameter view (S, ...), gradient view (S, ...), owner key)
_STACKED = {}
# owner key -> set of slot keys that may be taken now (between begin() and collect())
_OPEN = {}
_SLOT = '_nesie_grad_slot'


def register(owner, params, grad_views, stacked):
    """``owner``: the FlatTrainState; ``stacked``: [(params of the group, pview, gview)]."""
    key = id(owner)
    for p, g in zip(params, grad_views):
        _PARAM[id(p)] = (weakref.ref(p), g, key)
    for group, pview, gview in stacked:
        _STACKED[tuple(id(p) for p in group)] = (pview, gview, key)
    _OPEN[key] = set()
    weakref.finalize(owner, _forget, key)


def _forget(key):
    for table in (_PARAM, _STACKED):
        for k in [k for k, v in table.items() if v[2] == key]:
            del table[k]
    _OPEN.pop(key, None)


def begin(owner):
    """Every slot of ``owner`` may be taken once from now on (until ``close``)."""
    key = id(owner)
    _OPEN[key] = {k for k, v in _PARAM.items() if v[2] == key} | {k for k, v in _STACKED.items() if v[2] == key}


def close(owner, params=None):
    """The slots of ``params`` (default: all of ``owner``'s) may not be taken any more: outside a
    begin() / collect() pair autograd ACCUMULATES into the attached ``.grad`` views, and a kernel
    that had written its result into the same memory would be counted twice."""
    key = id(owner)
    if params is None:
        _OPEN[key] = set()
        return
    gone = {id(p) for p in params}
    still = _OPEN.get(key, set())
    _OPEN[key] = {k for k in still if not (k in gone if not isinstance(k, tuple) else any(i in gone for i in k))}


def _base_param(t):
    base = t._base if t._base is not None else t
    hit = _PARAM.get(id(base))
    return (base, hit) if hit is not None and hit[0]() is base else (None, None)


def take(t):
    """The gradient slot of tensor ``t`` -- a registered parameter, a contiguous full-size view of
    one (``weight.flatten(1)``), or a stacked view from ``stacked`` -- reshaped like ``t``, or None.
    A slot is handed out ONCE per begin(): whoever takes it must WRITE (not accumulate) the whole
    gradient of ``t`` into it and return it as that gradient."""
    if not torch.is_tensor(t) or not t.is_contiguous():
        return None
    tag = getattr(t, _SLOT, None)
    if tag is not None:
        entry = _STACKED.get(tag)
        if entry is None or tag not in _OPEN.get(entry[2], ()) or entry[1].numel() != t.numel():
            return None
        _OPEN[entry[2]].discard(tag)
        return entry[1].view(t.shape)
    base, hit = _base_param(t)
    if base is None or id(base) not in _OPEN.get(hit[2], ()) or base.numel() != t.numel():
        return None
    _OPEN[hit[2]].discard(id(base))
    return hit[1].view(t.shape)


class _StackedParams(Function):
    """The (S, ...) view of S parameters that sit side by side in the flat parameter vector, as an
    autograd function of them; the gradient goes back as S views of whatever arrives (the slot,
    when the consumer took it)."""

    @staticmethod
    def forward(ctx, view, *params):
        ctx.shapes = [p.shape for p in params]
        return view.view(view.shape)

    @staticmethod
    def backward(ctx, g):
        return (None,) + tuple(gi.reshape(s) for gi, s in zip(g.unbind(0), ctx.shapes))


def stacked(tensors):
    """``tensors``: S same-shaped tensors, each a registered parameter or a contiguous full-size
    view of one (``weight.flatten(1)``) -> their stack (S, *shape) as a VIEW of the flat parameter
    vector (autograd-connected to the S parameters), or None when they are not one registered
    group in this order."""
    bases = []
    for t in tensors:
        if not torch.is_tensor(t) or not t.is_contiguous():
            return None
        base, hit = _base_param(t)
        if base is None or base.numel() != t.numel():
            return None
        bases.append(base)
    key = tuple(id(b) for b in bases)
    entry = _STACKED.get(key)
    if entry is None:
        return None
    shape = (len(tensors),) + tuple(tensors[0].shape)
    out = _StackedParams.apply(entry[0].view(shape), *bases)
    setattr(out, _SLOT, key)
    return out


def reshaped(t, *shape):
    """``t.reshape(shape)`` that keeps a stacked view's slot tag (a reshape is a new tensor object)."""
    out = t.reshape(*shape)
    tag = getattr(t, _SLOT, None)
    if tag is not None and out.numel() == t.numel():
        setattr(out, _SLOT, tag)
    return out
