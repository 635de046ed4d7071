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


def save_reference_checkpoint(model, out_dir, epoch, iteration, optimizer=None, meta=None,
                              ema_copy=False, create_symlink=True):
    """Write ``epoch_{epoch}.pth`` (+ ``latest.pth``) in the reference's layout; with
    ``ema_copy`` also ``epoch_{epoch}_ema.pth`` holding the teacher's weights under the
    student's names (``model.teacher.swap()`` around the second save, as
    ``SimiRunnerHook._save_checkpoint`` does).  Returns the paths written."""
    os.makedirs(out_dir, exist_ok=True)
    meta = dict(meta or {}, epoch=int(epoch), iter=int(iteration))

    def dump(path):
        ckpt = {'meta': meta,
                'state_dict': OrderedDict((k, v.detach().cpu())
                                          for k, v in model.state_dict().items())}
        if optimizer is not None:
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
