"""Checkpoint wire format of the reference.

The reference saves through mmcv's ``save_checkpoint``
(``mmdet3d/runner/simi_epoch_based_runner.py:149-194``): one ``torch.save`` of

    {'meta': {'epoch': e, 'iter': i, ...},
     'state_dict': OrderedDict(name -> cpu tensor),        # DDP's 'module.' prefix stripped
     'optimizer': optimizer.state_dict()}                  # optional

as ``epoch_{e}.pth`` plus a ``latest.pth`` symlink; ``SimiRunnerHook._save_checkpoint``
(``simi_runner_hook.py:129-152, 166-196``) writes a second ``epoch_{e}_ema.pth`` after swapping
the teacher weights in.  The EMA teacher rides inside the state dict as buffers named
``ema_<param name with dots replaced by underscores>`` (``simi_teacher_hook.py:39-52``).
``nesie_amd.votenet`` keeps the reference's module and parameter names, so a reference
checkpoint loads key for key; these helpers only handle the envelope.
"""
import os
from collections import OrderedDict

import torch


def _state_dict_of(obj):
    if isinstance(obj, dict) and 'state_dict' in obj:
        return obj['state_dict'], obj.get('meta', {}), obj.get('optimizer')
    if isinstance(obj, dict) and obj and all(isinstance(v, torch.Tensor) for v in obj.values()):
        return obj, {}, None
    raise ValueError('not a checkpoint: expected {"state_dict": ...} or a plain state dict')


def load_reference_checkpoint(model, checkpoint, strict=True, map_location='cpu'):
    """Load a reference ``epoch_N.pth`` (path or the loaded dict) into ``model``.

    Handles the mmcv envelope and a leading ``module.`` (checkpoints written from inside
    ``MMDistributedDataParallel``).  A supervised pre-training checkpoint has no ``ema_*``
    buffers: loading it into a semi-supervised model with ``strict=True`` tolerates exactly
    those missing keys (``--load-from`` of the reference, README.md:38-43; call
    ``model.teacher.resync()`` afterwards, as ``SimiTeacherHook.before_run`` does).
    Returns ``(meta, optimizer_state_or_None)``."""
    if isinstance(checkpoint, (str, os.PathLike)):
        checkpoint = torch.load(checkpoint, map_location=map_location)
    state, meta, optim = _state_dict_of(checkpoint)
    state = OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in state.items())
    own = model.state_dict()
    missing = [k for k in own if k not in state]
    unexpected = [k for k in state if k not in own]
    if strict:
        bad_missing = [k for k in missing if not k.startswith('ema_')]
        if bad_missing or unexpected:
            raise RuntimeError(f'checkpoint does not match the model: missing {bad_missing[:5]} '
                               f'(+{max(len(bad_missing) - 5, 0)}), unexpected {unexpected[:5]} '
                               f'(+{max(len(unexpected) - 5, 0)})')
    for k, v in state.items():
        if k in own and tuple(own[k].shape) != tuple(v.shape):
            raise RuntimeError(f'{k}: checkpoint shape {tuple(v.shape)} != model '
                               f'{tuple(own[k].shape)}')
    model.load_state_dict(state, strict=False)
    return meta, optim


def per_parameter_optimizer_state(optimizer, flat_state):
    """State dict of an optimiser built on ``dp.FlatTrainState.flat_param`` (ONE parameter, one
    2.6 M-element moment pair) -> the reference's per-parameter form: one entry per model
    parameter in ``parameters()`` order, ``exp_avg`` / ``exp_avg_sq`` cut at the flat vector's
    offsets, ``step`` repeated, ``param_groups[0]['params'] = [0 .. n-1]`` (what
    ``torch.optim.AdamW(model.parameters()).state_dict()`` holds after the same steps)."""
    sd = optimizer.state_dict()
    if len(sd['param_groups']) != 1 or len(sd['param_groups'][0]['params']) != 1:
        raise ValueError('expected an optimiser over the single flat parameter')
    flat = sd['state'].get(sd['param_groups'][0]['params'][0], {})
    state = {}
    for i, (p, off) in enumerate(zip(flat_state.params, flat_state.offsets)):
        n = p.numel()
        entry = {}
        for k, v in flat.items():
            if torch.is_tensor(v) and v.numel() == flat_state.flat.numel():
                entry[k] = v[off:off + n].view_as(p).detach().cpu().clone()
            else:
                entry[k] = v.detach().cpu().clone() if torch.is_tensor(v) else v
        if entry:
            state[i] = entry
    group = dict(sd['param_groups'][0], params=list(range(len(flat_state.params))))
    return {'state': state, 'param_groups': [group]}


def load_per_parameter_optimizer_state(optimizer, flat_state, reference_state):
    """The inverse: a reference ``checkpoint['optimizer']`` (per-parameter AdamW state) into an
    optimiser built on the flat parameter.  Shapes are checked entry by entry."""
    groups = reference_state['param_groups']
    order = [i for g in groups for i in g['params']]
    if len(order) != len(flat_state.params):
        raise RuntimeError(f'optimizer state holds {len(order)} parameters, the model '
                           f'{len(flat_state.params)}')
    sd = optimizer.state_dict()
    key = sd['param_groups'][0]['params'][0]
    ref = reference_state['state']
    if ref:
        total = flat_state.flat.numel()
        like = flat_state.flat_param
        merged, step = {}, None
        for i, p, off in zip(order, flat_state.params, flat_state.offsets):
            n = p.numel()
            for k, v in ref.get(i, {}).items():
                if torch.is_tensor(v) and v.numel() == n and k != 'step':
                    if tuple(v.shape) != tuple(p.shape):
                        raise RuntimeError(f'optimizer state {k} of parameter {i}: shape '
                                           f'{tuple(v.shape)} != {tuple(p.shape)}')
                    merged.setdefault(k, torch.zeros(total, dtype=like.dtype, device=like.device))
                    merged[k][off:off + n] = v.reshape(-1).to(like.device, like.dtype)
                elif k == 'step':
                    step = v
        if step is not None:
            old = sd['state'].get(key, {}).get('step')
            merged['step'] = (torch.as_tensor(float(step)).to(old.device, old.dtype)
                              if torch.is_tensor(old) else
                              (step.clone() if torch.is_tensor(step) else torch.tensor(float(step))))
        sd['state'] = {key: merged}
    for k, v in groups[0].items():          # learning rate, betas, weight decay ... (uniform)
        if k != 'params':
            sd['param_groups'][0][k] = v
    optimizer.load_state_dict(sd)


def save_reference_checkpoint(model, out_dir, epoch, iteration, optimizer=None, meta=None,
                              ema_copy=False, create_symlink=True, flat_state=None):
    """Write ``epoch_{epoch}.pth`` (+ ``latest.pth``) in the reference's layout; with
    ``ema_copy`` also ``epoch_{epoch}_ema.pth`` holding the teacher's weights under the
    student's names (``model.teacher.swap()`` around the second save, as
    ``SimiRunnerHook._save_checkpoint`` does).  Returns the paths written.

    ``optimizer``: saved in the reference's PER-PARAMETER form.  An optimiser built on
    ``dp.FlatTrainState.flat_param`` needs ``flat_state`` (the FlatTrainState) so that its single
    flat moment pair can be cut at the parameter offsets; without it such an optimiser is refused
    rather than written in a form the reference cannot resume from."""
    os.makedirs(out_dir, exist_ok=True)
    meta = dict(meta or {}, epoch=int(epoch), iter=int(iteration))

    def dump(path):
        ckpt = {'meta': meta,
                'state_dict': OrderedDict((k, v.detach().cpu())
                                          for k, v in model.state_dict().items())}
        if optimizer is not None:
            if flat_state is not None:
                ckpt['optimizer'] = per_parameter_optimizer_state(optimizer, flat_state)
            else:
                n_opt = sum(len(g['params']) for g in optimizer.param_groups)
                n_model = sum(1 for p in model.parameters() if p.requires_grad)
                if n_opt != n_model:
                    raise ValueError(
                        f'the optimiser holds {n_opt} parameter(s), the model {n_model}: an '
                        'optimiser over dp.FlatTrainState.flat_param must be saved with '
                        'flat_state=<the FlatTrainState> (its state is converted to the '
                        "reference's per-parameter layout)")
                ckpt['optimizer'] = optimizer.state_dict()
        torch.save(ckpt, path)

    path = os.path.join(out_dir, f'epoch_{int(epoch)}.pth')
    dump(path)
    written = [path]
    if create_symlink:
        link = os.path.join(out_dir, 'latest.pth')
        if os.path.lexists(link):
            os.remove(link)
        os.symlink(os.path.basename(path), link)
    if ema_copy:
        teacher = getattr(model, 'teacher', None)
        if teacher is None:
            raise ValueError('ema_copy needs a model with an EMA teacher')
        teacher.swap()
        try:
            ema_path = os.path.join(out_dir, f'epoch_{int(epoch)}_ema.pth')
            dump(ema_path)
            written.append(ema_path)
        finally:
            teacher.swap()
    return written
