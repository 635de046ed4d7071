"""Per-parameter gradient gap, HIP path (fused / module-by-module) vs CPU oracle path and fp64,
full-size B=2 40k step, vote sampling forced to the fp64 leg's picks."""
import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle
from nesie_amd import kernels
from nesie_amd.scenes import make_batch
from nesie_amd.votenet import build_nesie_votenet
from nesie_amd.mmdet3d_ops import fused_mlp
from tests import _small, _fp64

torch.manual_seed(0)
model = build_nesie_votenet()
model.train()
pts, boxes, labels = make_batch(4242, 2, 40000)
noise = _small.fixed_noise(2, model.bbox_head.num_proposal)
model.bbox_head.jitter_noise = noise
dev = torch.device('cuda:0')
s = _small.force_vote_sampling(model, "dbg")
l64, g64 = _fp64.train_step_fp64(model, pts, boxes, labels, noise=noise)
print("picks", len(_small.ForcedSampler.book["dbg"]))
legs = {}
m = copy.deepcopy(model)
with kernels.use_backend(oracle.OracleKernels()):
    legs['cpu32'] = _small.train_step_losses(m, pts, boxes, labels)
print('cpu32 own picks agreed:', m.bbox_head.vote_aggregation.points_sampler.agreed)
for enabled in (True, False):
    fused_mlp.ENABLED = enabled
    gmodel = copy.deepcopy(model).to(dev)
    legs[f'gpu fused={enabled}'] = _small.train_step_losses(gmodel, pts.to(dev), boxes, labels)
    print(enabled, 'own picks agreed:', gmodel.bbox_head.vote_aggregation.points_sampler.agreed)
names = [n for n, _ in model.named_parameters() if n in g64]
w = torch.cat([g64[n].flatten() for n in names])
for leg, (l, g) in legs.items():
    f = torch.cat([g[n].flatten().double().cpu() for n in names])
    print(leg, 'flat rel to fp64', ((f - w).norm() / w.norm()).item(),
          {k: round(float(v.sum()) - l64[k], 6) for k, v in l.items()})
gmax = w.abs().max().item()
for n in names:
    row = []
    for leg, (l, g) in legs.items():
        denom = max(g64[n].abs().max().item(), 1e-3 * gmax)
        row.append((g[n].double().cpu() - g64[n]).abs().max().item() / denom)
    if max(row) > 1e-3:
        print(f'{n:66s} ' + ' '.join(f'{r:.2e}' for r in row))
