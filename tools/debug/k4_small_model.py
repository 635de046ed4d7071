"""Small test model, HIP leg, SA1's first activation stored vs rebuilt: per-module output differences
and the loss terms (which discrete decision does the 1e-7 perturbation tip?)."""
import sys, os, copy
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import _small
from nesie_amd.mmdet3d_ops import fused_mlp

dev = torch.device('cuda:0')
model = _small.small_model()
model.train_cfg['pos_distance_thr'] = 1.0
model.train_cfg['neg_distance_thr'] = 1.5
pts, boxes, labels = _small.small_batch()
noise = _small.fixed_noise(2, 32)
model.bbox_head.jitter_noise = noise
_small.force_vote_sampling(model, 'dbg')
_small.force_grid_taps(model, 'dbg')
outs = {}
for k4 in (False, True):
    fused_mlp.SA1_K4 = k4
    m = copy.deepcopy(model).to(dev)
    rec = {}
    hooks = []
    for name, mod in m.named_modules():
        def hook(mod_, inp, out, name=name):
            if torch.is_tensor(out) and out.is_floating_point():
                rec[name] = out.detach().clone()
            elif isinstance(out, (tuple, list)):
                for i, o in enumerate(out):
                    if torch.is_tensor(o) and o.is_floating_point():
                        rec[f'{name}[{i}]'] = o.detach().clone()
        hooks.append(mod.register_forward_hook(hook))
    losses, grads = _small.train_step_losses(m, pts.to(dev), boxes, labels)
    outs[k4] = (rec, losses, grads)
a, b = outs[False], outs[True]
print('losses', {k: (float(a[1][k].sum()), float(b[1][k].sum())) for k in a[1]})
rows = []
for k in a[0]:
    if k in b[0] and a[0][k].shape == b[0][k].shape:
        d = (a[0][k] - b[0][k]).abs().max().item()
        s = a[0][k].abs().max().item()
        rows.append((d / max(s, 1e-20), k, tuple(a[0][k].shape)))
for r in rows:
    if r[0] > 1e-5:
        print(f'{r[0]:.2e}', r[1], r[2])
names = sorted(a[2])
fa = torch.cat([a[2][n].double().flatten() for n in names]); fb = torch.cat([b[2][n].double().flatten() for n in names])
print('flat gradient: stored vs rebuilt rel. L2', ((fa - fb).norm() / fa.norm()).item())
contrib = sorted((((a[2][n].double() - b[2][n].double()).norm() / fa.norm()).item(), n) for n in names)[-8:]
for e, n in contrib:
    print(f'share {e:.2e}', n)
print('first modules over 1e-5 listed above (module order); grads:')
gw = sorted(((a[2][n] - b[2][n]).abs().max().item() / max(a[2][n].abs().max().item(), 1e-12), n) for n in a[2])[-8:]
for e, n in gw:
    print(f'{e:.2e}', n)
