"""dp.FlatTrainState's layout with stack groups and nesie_amd.grad_slots (gradients written straight
into the flat vector, stacked parameters as views) on a toy model, no GPU: a stand-in autograd
function plays the part of the hand-written backward kernels."""
import copy

import torch
from torch.autograd import Function

from nesie_amd import checkpoint, dp, grad_slots


class _Lin(Function):
    """y[s] = x @ W[s]^T + b[s]; the backward WRITES dW / db into their slots when it got them."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.slots = [grad_slots.take(t) if ctx.needs_input_grad[1 + j] else None for j, t in enumerate((W, b))]
        ctx.save_for_backward(x, W)
        return torch.einsum('bi,soi->sbo', x, W) + b[:, None, :]

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        outs = []
        for sl, v in zip(ctx.slots, (torch.einsum('sbo,bi->soi', g, x), g.sum(1))):
            outs.append(v if sl is None else sl.copy_(v))
        return torch.einsum('sbo,soi->bi', g, W), outs[0], outs[1]


def _toy():
    torch.manual_seed(0)
    pre = torch.nn.Linear(5, 4)
    nets = [torch.nn.Linear(4, 3) for _ in range(3)]
    params = list(pre.parameters()) + [p for n in nets for p in n.parameters()]
    return pre, nets, params


def test_stack_groups_are_contiguous_views_and_gradients_land_in_their_slots():
    pre, nets, params = _toy()
    ref = copy.deepcopy([pre] + nets)
    state = dp.FlatTrainState(params, stack_groups=[[n.weight for n in nets], [n.bias for n in nets]])
    assert [id(p) for p in state.params] == [id(p) for p in params]            # model order kept
    assert sorted(state.offsets) != state.offsets                               # ... the layout differs
    assert state.split_after(pre.parameters()) == (2, 24)                       # ungrouped block first
    W = grad_slots.stacked([n.weight for n in nets])
    assert W.data_ptr() == nets[0].weight.data_ptr() and tuple(W.shape) == (3, 3, 4)
    for i, n in enumerate(nets):
        assert torch.equal(W[i], n.weight) and n.weight.data_ptr() == W[i].data_ptr()
    assert grad_slots.stacked([nets[1].weight, nets[0].weight, nets[2].weight]) is None   # another order
    assert grad_slots.take(W) is None                                           # nothing open yet
    x = torch.randn(7, 5)

    def loss():
        W = grad_slots.stacked([n.weight for n in nets])
        b = grad_slots.stacked([n.bias for n in nets])
        return _Lin.apply(pre(x), W, b).square().sum()
    state.begin()
    grads = torch.autograd.grad(loss(), state.params)
    for p, g in zip(state.params, grads):
        p.grad = g
    in_place = [p.grad.data_ptr() == v.data_ptr() for p, v in zip(state.params, state.grad_views)]
    assert in_place == [False, False] + [True] * 6          # the kernel-written ones never get copied
    state.collect()
    torch.stack([m(ref[0](x)) for m in ref[1:]]).square().sum().backward()
    want = [p.grad for m in ref for p in m.parameters()]
    for p, off, w in zip(state.params, state.offsets, want):
        torch.testing.assert_close(state.flat[off:off + p.numel()].view_as(p), w, rtol=1e-5, atol=1e-6)
    # outside begin() / collect() the attached .grad views ACCUMULATE: slots must stay closed
    loss().backward()
    for p, off, w in zip(state.params, state.offsets, want):
        torch.testing.assert_close(state.flat[off:off + p.numel()].view_as(p), 2 * w, rtol=1e-5, atol=1e-6)
    # a slot is handed out once per begin(): the second taker allocates, autograd adds, collect() copies
    state.begin()
    W = grad_slots.stacked([n.weight for n in nets])
    b = grad_slots.stacked([n.bias for n in nets])
    h = pre(x)
    total = _Lin.apply(h, W, b).square().sum() + _Lin.apply(h, W, b).square().sum()
    for p, g in zip(state.params, torch.autograd.grad(total, state.params)):
        p.grad = g
    state.collect()
    for p, off, w in zip(state.params, state.offsets, want):
        torch.testing.assert_close(state.flat[off:off + p.numel()].view_as(p), 2 * w, rtol=1e-5, atol=1e-6)
    # no-grad passes (the EMA teacher's forward) take nothing
    state.begin()
    with torch.no_grad():
        _Lin.apply(pre(x), grad_slots.stacked([n.weight for n in nets]), grad_slots.stacked([n.bias for n in nets]))
    assert grad_slots.take(grad_slots.stacked([n.weight for n in nets])) is not None


def test_registry_forgets_a_dead_state_and_reshapes_keep_the_tag():
    pre, nets, params = _toy()
    state = dp.FlatTrainState(params, stack_groups=[[n.bias for n in nets]])
    state.begin()
    b = grad_slots.stacked([n.bias for n in nets])
    flat = grad_slots.reshaped(b, -1)
    slot = grad_slots.take(flat)
    assert slot is not None and tuple(slot.shape) == (9,) and slot.data_ptr() == state.grad_views[3].data_ptr()
    assert grad_slots.take(b) is None                       # the same slot, already out
    w = nets[0].weight.flatten(0)                           # a full-size contiguous view of a parameter
    assert grad_slots.take(w).data_ptr() == state.grad_views[2].data_ptr()
    assert grad_slots.take(nets[1].weight[:2]) is None      # a partial view has no slot
    key = id(state)
    del state, slot
    import gc
    gc.collect()
    assert not any(v[2] == key for v in grad_slots._PARAM.values())
    assert grad_slots.stacked([n.bias for n in nets]) is None


def test_per_parameter_optimizer_state_follows_the_layout(tmp_path):
    """checkpoint.per_parameter_optimizer_state / load_per_parameter_optimizer_state with a state whose
    flat layout is NOT the running sum of the parameter sizes."""
    pre, nets, params = _toy()
    twin = copy.deepcopy([pre] + nets)
    tparams = [p for m in twin for p in m.parameters()]
    state = dp.FlatTrainState(params, stack_groups=[[n.weight for n in nets], [n.bias for n in nets]])
    opt = torch.optim.AdamW([state.flat_param], lr=1e-2, weight_decay=0.05)
    ref = torch.optim.AdamW(tparams, lr=1e-2, weight_decay=0.05)
    x = torch.randn(6, 5)
    for _ in range(2):
        state.begin()
        torch.stack([n(pre(x)) for n in nets]).square().sum().backward()
        state.collect()
        opt.step()
        ref.zero_grad()
        torch.stack([m(twin[0](x)) for m in twin[1:]]).square().sum().backward()
        ref.step()
    for p, t in zip(params, tparams):
        torch.testing.assert_close(p, t, rtol=1e-5, atol=1e-7)
    saved = checkpoint.per_parameter_optimizer_state(opt, state)
    want = ref.state_dict()
    for i in range(len(params)):
        for k in ('exp_avg', 'exp_avg_sq'):
            torch.testing.assert_close(saved['state'][i][k], want['state'][i][k], rtol=1e-6, atol=1e-8)
    fresh = torch.optim.AdamW([state.flat_param], lr=1.0)
    checkpoint.load_per_parameter_optimizer_state(fresh, state, want)
    got, old = fresh.state[state.flat_param], opt.state[state.flat_param]
    torch.testing.assert_close(got['exp_avg'], old['exp_avg'], rtol=1e-6, atol=1e-8)
    torch.testing.assert_close(got['exp_avg_sq'], old['exp_avg_sq'], rtol=1e-6, atol=1e-8)


def _gap_mask(state):
    used = torch.zeros(state.flat.numel(), dtype=torch.bool)
    for p, o in zip(state.params, state.offsets):
        used[o:o + p.numel()] = True
    return ~used


def test_alignment_gaps_hold_zeros_in_both_flat_vectors_and_stay_zero():
    """Every parameter starts on a 16-byte boundary; the gaps are part of the vectors the optimiser,
    the clip and the all-reduce walk, so they must be 0 after construction (torch.empty would hand
    back recycled memory on the GPU) and after optimiser steps."""
    torch.manual_seed(1)
    mods = [torch.nn.Linear(3, 3), torch.nn.Linear(3, 5), torch.nn.Linear(5, 2)]   # numels 9, 3, 15, 5, 10, 2
    params = [p for m in mods for p in m.parameters()]
    state = dp.FlatTrainState(params)
    gaps = _gap_mask(state)
    assert int(gaps.sum()) > 0 and all(o % 4 == 0 for o in state.offsets)
    assert torch.all(state.flat_param.data[gaps] == 0) and torch.all(state.flat[gaps] == 0)
    opt = torch.optim.AdamW([state.flat_param], lr=1e-2, weight_decay=0.1)
    x = torch.randn(4, 3)
    for _ in range(3):
        state.begin()
        mods[2](mods[1](mods[0](x))).square().sum().backward()
        state.collect()
        opt.step()
        assert torch.all(state.flat_param.data[gaps] == 0) and torch.all(state.flat[gaps] == 0)
    assert torch.isfinite(state.flat_param.data).all()


def test_split_after_returns_the_layout_extent_of_a_prefix_with_ragged_sizes():
    """The segment boundary of the two-part all-reduce is a LAYOUT offset: a leading block holding a
    3-element parameter ends on the next 16-byte boundary, not at the sum of its sizes."""
    torch.manual_seed(2)
    pre = torch.nn.Linear(3, 3)      # weight 9 (-> 12), bias 3 (-> 16)
    rest = [torch.nn.Linear(3, 4), torch.nn.Linear(3, 4)]
    params = list(pre.parameters()) + [p for m in rest for p in m.parameters()]
    state = dp.FlatTrainState(params, stack_groups=[[m.weight for m in rest], [m.bias for m in rest]])
    n, extent = state.split_after(pre.parameters())
    assert (n, extent) == (2, 16)
    assert all(o >= extent for o in state.offsets[n:]) and extent % 4 == 0
    # a block that is not a layout prefix is refused: a member of a stack group lies behind the others
    twin = dp.FlatTrainState([p for m in rest for p in m.parameters()] + list(pre.parameters()),
                             stack_groups=[[m.weight for m in rest]])
    try:
        twin.split_after(list(rest[0].parameters()))
        raise SystemExit('expected an assertion')
    except AssertionError:
        pass


def test_collect_overwrites_a_poisoned_slot_of_a_parameter_that_got_no_gradient():
    """A parameter whose backward returns None (the conv bias in front of a norm in Stack1dFn:
    identically zero) is zero-filled by collect() in EVERY step, whatever sat in its slot."""
    pre, nets, params = _toy()
    state = dp.FlatTrainState(params)
    state.flat.fill_(float('nan'))
    x = torch.randn(3, 5)
    state.begin()
    nets[0](pre(x)).sum().backward()          # nets[1], nets[2] are not reached
    state.collect()
    for p, o in zip(state.params, state.offsets):
        g = state.flat[o:o + p.numel()]
        reached = any(p is q for m in (pre, nets[0]) for q in m.parameters())
        assert torch.isfinite(g).all() and (reached or torch.all(g == 0))


def test_deferred_reduction_window_is_a_no_op_without_a_gpu():
    """FlatTrainState on CPU tensors never opens the deferred-reduction window of the HIP back end
    (begin() / collect() leave kernels.HipKernels._deferred alone)."""
    from nesie_amd.kernels import HipKernels
    pre, nets, params = _toy()
    state = dp.FlatTrainState(params)
    before = HipKernels._deferred
    state.begin()
    assert HipKernels._deferred is before and state._deferral is None
    nets[0](pre(torch.randn(2, 5))).sum().backward()
    state.collect()
    assert HipKernels._deferred is before


def test_flat_ema_teacher_equals_the_per_tensor_teacher_bit_for_bit():
    """EMATeacher.use_flat (round 5): update and swap over ONE flat vector give the per-tensor
    arithmetic element for element; a replaced buffer drops back to the per-tensor path."""
    from nesie_amd import dp
    from nesie_amd.votenet.semi import EMATeacher

    def mk():
        torch.manual_seed(0)
        return torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3))
    a, b = mk(), mk()
    ta, tb = EMATeacher(a, momentum=0.01), EMATeacher(b, momentum=0.01)
    dp.FlatTrainState(a.parameters())
    sb = dp.FlatTrainState(b.parameters())
    assert tb.use_flat(sb) and tb._flat_ok() is not None
    for step in range(1, 6):
        for m_ in (a, b):
            with torch.no_grad():
                g = torch.Generator().manual_seed(step)
                for p in m_.parameters():
                    p.data.add_(torch.randn(p.shape, generator=g) * 0.1)
        ta.update(step); tb.update(step)
        ta.swap(); tb.swap()
        assert all(torch.equal(x, y) for x, y in zip(a.parameters(), b.parameters()))
        ta.swap(); tb.swap()
        assert all(torch.equal(x, y) for x, y in zip(ta.emas, tb.emas))
        assert all(torch.equal(x, y) for x, y in zip(a.parameters(), b.parameters()))
    assert tb._flat_ok() is not None
    b._buffers[tb.names[0][1]] = b._buffers[tb.names[0][1]].clone()      # e.g. a state-dict load with assign
    assert tb._flat_ok() is None
    ta.update(7); tb.update(7)
    assert all(torch.equal(x, y) for x, y in zip(ta.emas, tb.emas))

