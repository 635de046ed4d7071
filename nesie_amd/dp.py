"""Data-parallel step plumbing: one process per GPU, scenes sharded by rank, ONE
exchange per iteration -- an all-reduce(mean) of the flat gradient vector
(2 640 477 fp32 = 10.56 MB for Nesie-VoteNet) over RCCL/xGMI.

BatchNorm stays per-rank (the configs use plain BN, not SyncBN) and the EMA teacher is
a deterministic function of the post-all-reduce parameters, so nothing else crosses
ranks.  The message is latency-bound (a ring moves ~18.5 MB per GPU per step), so the
gradients travel as a single flat bucket rather than per-tensor collectives.
"""
import os

import torch
import torch.distributed as dist


def force_process_group():
    return os.environ.get('NESIE_FORCE_PG', '0') not in ('', '0')


def init_distributed(backend=None):
    """Read RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the env (torchrun contract).
    Returns (rank, world_size, local_rank).  backend: 'nccl' (= RCCL on ROCm) on GPUs,
    'gloo' for the CPU tests."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # NESIE_FORCE_PG=1: build the process group even for ONE rank, so that the single-GPU run
    # launches the real collectives (RCCL all-reduce of both gradient segments on the communication
    # stream between the graph replays) -- the only way to execute that path without a second GPU
    if (world > 1 or force_process_group()) and not dist.is_initialized():
        if backend is None:
            # NESIE_DIST_BACKEND=gloo: rehearse the N > 1 step with several ranks on ONE GPU
            # (RCCL refuses two ranks on a device; gloo stages the all-reduce through the host)
            backend = os.environ.get('NESIE_DIST_BACKEND') or \
                ('nccl' if torch.cuda.is_available() else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class FlatGradBucket:
    """All parameters' gradients as views into one contiguous buffer, so the per-step
    exchange is a single all-reduce and the optimiser sees ordinary ``p.grad``s.

    ``params`` keeps the model's ``parameters()`` order; ``offsets[i]`` is where parameter i
    starts in the flat vector.  Without ``stack_groups`` that is the running sum of the sizes.
    ``stack_groups`` = lists of S same-shaped parameters that the step uses STACKED (the six side
    MiniPointNets / score heads of the quality head): each group is laid out contiguously behind
    the ungrouped parameters, so its (S, ...) stack is a view (``grad_slots``)."""

    def __init__(self, params, stack_groups=None):
        self.params = [p for p in params if p.requires_grad]
        ids = {id(p) for p in self.params}
        groups = []
        seen = set()
        for g in stack_groups or ():
            g = list(g)
            if (len(g) > 1 and all(id(p) in ids and id(p) not in seen for p in g)
                    and all(p.shape == g[0].shape for p in g)):
                groups.append(g)
                seen.update(id(p) for p in g)
        # every parameter (and every group) starts on a 16-byte boundary: the layer kernel fetches
        # row-major weights with 16-byte loads (a 259-float bias in front would shift every later
        # weight by 12 bytes); the gaps hold zeros in both flat vectors and stay zero
        align = lambda o: -(-o // 4) * 4  # noqa: E731
        where, off = {}, 0
        for p in self.params:                       # ungrouped, in model order
            if id(p) not in seen:
                off = align(off)
                where[id(p)] = off
                off += p.numel()
        self.group_spans = []                       # (offset, S, numel of one member, shape)
        for g in groups:
            off = align(off)
            self.group_spans.append((off, len(g), g[0].numel(), tuple(g[0].shape), g))
            for p in g:
                where[id(p)] = off
                off += p.numel()
        total = align(off)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        self.offsets = [where[id(p)] for p in self.params]
        for p, o in zip(self.params, self.offsets):
            p.grad = self.flat[o:o + p.numel()].view_as(p)

    def zero_(self):
        self.flat.zero_()
        for p in self.params:  # autograd accumulates in place into the views
            if p.grad is None or p.grad.data_ptr() < self.flat.data_ptr():
                raise RuntimeError('a gradient left the flat bucket (set_to_none?)')

    def all_reduce_mean(self, group=None):
        """In-place mean over ranks; a no-op in a single process."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(dist.get_world_size(group))
        return self.flat

    def nbytes(self):
        return self.flat.numel() * self.flat.element_size()


class FlatTrainState(FlatGradBucket):
    """Parameters AND gradients of a model as two contiguous vectors.

    * Every ``p.data`` becomes a view into ``flat_param``; ``flat_param.grad`` is the bucket's
      flat gradient.  An optimiser built on ``[flat_param]`` (uniform hyper-parameters, as in
      the reference configs: AdamW lr/wd with no paramwise_cfg, grad_clip max_norm=10) runs
      its element-wise update as a handful of launches instead of one multi-tensor sweep per
      operation over ~220 tensors; element for element the arithmetic is the per-tensor
      optimiser's.  ``clip_grad_norm_([flat_param])`` is the same total 2-norm.
    * ``begin()`` drops the ``.grad``s so autograd hands each finished gradient over instead
      of adding it into a zeroed view (one launch per parameter); ``collect()`` gathers them
      into the flat vector with one multi-tensor copy, zero-fills parameters the loss did not
      reach (their ``.grad`` would have stayed zero), and re-attaches the views.
    * Between ``begin()`` and ``collect()`` the hand-written backward kernels may write a
      gradient straight into its slot of the flat vector (``grad_slots.take``): what autograd
      then hands over IS the slot and ``collect()`` copies nothing for it.
    Both calls are capture-safe: under a hipGraph the python bookkeeping runs once at capture
    time and the gradient tensors live in the graph's pool at fixed addresses."""

    def __init__(self, params, stack_groups=None):
        super().__init__(params, stack_groups)
        from . import grad_slots
        ref = self.params[0]
        # zeros, not empty: the 16-byte alignment gaps between parameters are part of the vector the
        # optimiser, the norm and the all-reduce walk -- they must hold 0 (and then stay 0: gradient
        # 0, moments 0, decay of 0)
        flat_p = torch.zeros(self.flat.numel(), dtype=ref.dtype, device=ref.device)
        self.views, self.grad_views = [], []
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                n = p.numel()
                v = flat_p[o:o + n].view_as(p)
                v.copy_(p.data)
                p.data = v
                self.views.append(v)
                self.grad_views.append(p.grad)
        self.flat_param = torch.nn.Parameter(flat_p)
        self.flat_param.grad = self.flat
        self._held = []
        stacked = [(g, flat_p[o:o + s * n].view(s, *shape), self.flat[o:o + s * n].view(s, *shape))
                   for o, s, n, shape, g in self.group_spans]
        grad_slots.register(self, self.params, self.grad_views, stacked)

    def begin(self):
        from . import grad_slots
        self._held = []
        for p in self.params:
            p.grad = None
        grad_slots.begin(self)
        # weight gradients written straight into their slots may leave their partial sums pending
        # until collect() (kernels.HipKernels.begin_deferred_reductions)
        self._deferral, self._collected = None, 0
        if self.flat.is_cuda:
            from .kernels import HipKernels
            HipKernels.begin_deferred_reductions()
            self._deferral = HipKernels

    def split_after(self, first_params):
        """-> (number of parameters, LAYOUT extent in elements) of the leading block
        ``first_params`` (which must be a prefix of the bucket's parameter order AND of its layout,
        e.g. ``model.backbone``: no member of a stack group).  The extent is where the block ends
        in the flat vectors, alignment gaps included and rounded up to the next 16-byte boundary
        (= where the next parameter starts), so [0, extent) and [extent, total) are the two
        all-reduce segments."""
        first = [p for p in first_params if p.requires_grad]
        assert [id(p) for p in first] == [id(p) for p in self.params[:len(first)]], \
            'not a prefix of the bucket'
        if not first:
            return 0, 0
        n = len(first)
        extent = -(-(self.offsets[n - 1] + first[-1].numel()) // 4) * 4
        assert sorted(self.offsets[:n]) == self.offsets[:n], \
            'the leading block is not laid out in parameter order'
        assert all(o >= extent for o in self.offsets[n:]) and \
            all(o >= extent for o, *_ in self.group_spans), \
            'a later parameter or a stack group lies inside the leading block'
        return n, min(extent, self.flat.numel())

    def collect(self, lo=0, hi=None):
        """Gather the gradients of parameters [lo, hi) (default: all) into the flat vector."""
        from . import grad_slots
        if getattr(self, '_deferral', None) is not None:
            # ONE launch finishes every weight gradient whose reduction was left pending (kernels.
            # HipKernels.flush_deferred_reductions); the window closes with the last segment
            n_seg = len(self.params[lo:hi])
            self._collected = getattr(self, '_collected', 0) + n_seg
            last = self._collected >= len(self.params)
            done = self._deferral.flush_deferred_reductions(self.flat.device, close=last)
            if last:
                self._collected, self._deferral = 0, None
            # each of them must arrive as the slot ITSELF: a gradient autograd combined with another
            # one (a second use of the parameter outside the fused functions) was read before it was
            # complete -- fail loudly (NESIE_DEFER_WGRAD=0 switches the deferral off)
            spans = [(d.data_ptr(), d.data_ptr() + d.numel() * d.element_size()) for d in done]
            if spans:
                for p, v in zip(self.params[lo:hi], self.grad_views[lo:hi]):
                    a = v.data_ptr()
                    if p.grad is not None and p.grad.data_ptr() != a and any(s0 <= a < s1 for s0, s1 in spans):
                        raise RuntimeError('a weight gradient with a deferred reduction was consumed before '
                                           'FlatTrainState.collect() (parameter used twice?): set NESIE_DEFER_WGRAD=0')
        src, dst, missing = [], [], []
        for p, v in zip(self.params[lo:hi], self.grad_views[lo:hi]):
            if p.grad is None:
                missing.append(v)
            elif p.grad.data_ptr() == v.data_ptr() and p.grad.is_contiguous() and p.grad.numel() == v.numel():
                self._held.append(p.grad)        # written in place by the backward kernel (grad_slots)
            else:
                src.append(p.grad)
                dst.append(v)
        self._held = self._held + src  # keeps graph-pool gradients alive between capture and replays
        if dst:
            torch._foreach_copy_(dst, src)
        if missing:
            torch._foreach_zero_(missing)
        for p, v in zip(self.params[lo:hi], self.grad_views[lo:hi]):
            p.grad = v
        # their .grad views are attached again: a later backward ACCUMULATES into them, so their
        # slots must not be handed to a kernel any more (until the next begin())
        grad_slots.close(self, self.params[lo:hi])

    def zero_(self):
        self.flat.zero_()


class FlatAdamW(torch.optim.Optimizer):
    """``clip_grad_norm_(max_norm)`` + ``torch.optim.AdamW`` over ``FlatTrainState.flat_param`` as
    two launches of ``nesie_flat_adamw_step_dev``.  Capture-safe: the step count, the learning
    rate and the weight decay live in device memory and the clip coefficient never visits the
    host, so a step captured in a hipGraph follows an LR schedule -- the scheduler writes
    ``param_groups[0]['lr']`` as usual and ``sync_hyper()`` (called by ``step()``, and by whoever
    replays a captured step) carries a changed value to the device.  betas / eps / max_norm are
    launch constants (the reference never schedules them).  State keys are torch's (``step``,
    ``exp_avg``, ``exp_avg_sq``), so ``checkpoint.per_parameter_optimizer_state`` applies
    unchanged."""

    def __init__(self, flat_param, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                 max_norm=None):
        super().__init__([flat_param], dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_norm = max_norm
        self.grad_norm = None      # device scalar: the un-clipped global norm of the last step
        self.hyper = None          # device (2,): [lr, weight_decay] as the kernels read them
        self._hyper_host = None

    def sync_hyper(self):
        """Device copy of (lr, weight_decay) <- ``param_groups[0]`` when they differ from what was
        last written.  Must not be called while a graph is being captured with a changed value
        (the write is a host-to-device copy); ``step()`` calls it, so an eager step needs nothing."""
        group = self.param_groups[0]
        p = group['params'][0]
        want = (float(group['lr']), float(group['weight_decay']))
        if self.hyper is None or self.hyper.device != p.device:
            self.hyper = torch.tensor(want, dtype=torch.float32, device=p.device)
            self._hyper_host = want
        elif want != self._hyper_host:
            assert not (p.is_cuda and torch.cuda.is_current_stream_capturing()), \
                'the learning rate changed inside a graph capture: call sync_hyper() before it'
            self.hyper.copy_(torch.tensor(want, dtype=torch.float32), non_blocking=False)
            self._hyper_host = want
        return self.hyper

    def _device_state(self, p):
        """Adam state of the flat parameter, created on first use and NORMALISED after a
        ``load_state_dict``: ``Optimizer.load_state_dict`` leaves ``step`` on the CPU (it only
        moves it for capturable / fused groups) and checkpoints store it as a CPU scalar
        (checkpoint.per_parameter_optimizer_state); the kernels need a float32 device scalar."""
        st = self.state[p]
        if not st:
            st['step'] = torch.zeros((), dtype=torch.float32, device=p.device)
            st['exp_avg'] = torch.zeros_like(p)
            st['exp_avg_sq'] = torch.zeros_like(p)
        step = st['step']
        if not torch.is_tensor(step):
            step = torch.tensor(float(step))
        if step.device != p.device or step.dtype != torch.float32 or step.dim() != 0:
            st['step'] = step.detach().reshape(()).to(device=p.device, dtype=torch.float32)
        for k in ('exp_avg', 'exp_avg_sq'):
            if st[k].device != p.device or st[k].dtype != p.dtype or not st[k].is_contiguous():
                st[k] = st[k].to(device=p.device, dtype=p.dtype).contiguous()
        return st

    @torch.no_grad()
    def step(self, closure=None):
        from .kernels import backend_for
        group = self.param_groups[0]
        p = group['params'][0]
        st = self._device_state(p)
        if self.grad_norm is None:
            self.grad_norm = torch.zeros((), dtype=torch.float32, device=p.device)
        hyper = self.sync_hyper()
        backend_for(p).flat_adamw_step(p.data, p.grad, st['exp_avg'], st['exp_avg_sq'], st['step'],
                                       group['lr'], group['betas'], group['eps'],
                                       group['weight_decay'], self.max_norm, self.grad_norm,
                                       hyper=hyper)


def backward_head(total, boundary, early_params):
    """Phase 1 of a backward pass cut at ``boundary`` (tensors every path from the loss to the
    remaining parameters runs through -- the backbone's output): the gradients of
    ``early_params`` (the head) land in ``p.grad`` as fresh tensors; -> the boundary gradients."""
    early, boundary = list(early_params), list(boundary)
    g = torch.autograd.grad(total, boundary + early, allow_unused=True)
    for p, gp in zip(early, g[len(boundary):]):
        p.grad = gp
    return g[:len(boundary)]


def backward_rest(boundary, boundary_grads, late_params):
    """Phase 2: carry the boundary gradients down to ``late_params`` (the backbone)."""
    late = list(late_params)
    live = [(b, gb) for b, gb in zip(boundary, boundary_grads) if gb is not None]
    gl = torch.autograd.grad([b for b, _ in live], late, grad_outputs=[gb for _, gb in live],
                             allow_unused=True)
    for p, gp in zip(late, gl):
        p.grad = gp


def backward_in_two_phases(total, boundary, early_params, late_params, between=None):
    """``total.backward()`` in two phases with ``between()`` in the middle.  With the phases in
    two hipGraphs (bench.py) the head's share of the gradient all-reduce travels while the
    backbone's backward pass still computes."""
    gb = backward_head(total, boundary, early_params)
    if between is not None:
        between()
    backward_rest(boundary, gb, late_params)


class SegmentedAllReduce:
    """all-reduce(mean) of segments of the flat gradient on a communication stream of its own:
    ``launch(lo, hi)`` orders the segment's exchange after everything already queued on the
    current stream and returns at once; ``wait()`` makes the current stream wait for every
    exchange launched so far.  A single process (or no process group) does nothing; on the CPU
    (gloo) the exchange is synchronous."""

    def __init__(self, flat, group=None):
        self.flat, self.group = flat, group
        ready = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if ready else 1
        # a one-rank group exchanges nothing, but with NESIE_FORCE_PG the collectives are launched
        # all the same (sum over one rank, division by 1: the gradient is unchanged bit for bit)
        self.active = self.world > 1 or (ready and force_process_group())
        self.stream = torch.cuda.Stream(flat.device) if flat.is_cuda and self.active else None
        self.launched = []
        self.collectives = 0       # all-reduces actually issued

    def launch(self, lo, hi):
        self.launched.append((lo, hi))
        if not self.active:
            return
        self.collectives += 1
        seg = self.flat[lo:hi]
        if self.stream is None:
            dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group)
            seg.div_(self.world)
            return
        self.stream.wait_stream(torch.cuda.current_stream(self.flat.device))
        with torch.cuda.stream(self.stream):
            dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group)
            seg.div_(self.world)

    def wait(self):
        if self.stream is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.stream)
        done, self.launched = self.launched, []
        return done


def shard_range(total, rank, world):
    """Scenes [lo, hi) of a global batch owned by ``rank`` (contiguous blocks)."""
    per = total // world
    assert per * world == total, 'global batch must divide by the world size'
    return rank * per, (rank + 1) * per
