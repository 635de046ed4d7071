"""Per-parameter gradient gap between the fused layer kernels and the module-by-module path,
both on the GPU, full-size B=2 40k step (model order)."""
import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nesie_amd.scenes import make_batch
from nesie_amd.votenet import build_nesie_votenet
from nesie_amd.mmdet3d_ops import fused_mlp
from tests import _small

torch.manual_seed(0)
model = build_nesie_votenet()
model.train()
pts, boxes, labels = make_batch(4242, 2, 40000)
model.bbox_head.jitter_noise = _small.fixed_noise(2, model.bbox_head.num_proposal)
dev = torch.device('cuda:0')
res = {}
for enabled in (True, False):
    fused_mlp.ENABLED = enabled
    gmodel = copy.deepcopy(model).to(dev)
    res[enabled] = _small.train_step_losses(gmodel, pts.to(dev), boxes, labels)
for k in res[True][0]:
    print(k, float(res[True][0][k].sum()), float(res[False][0][k].sum()))
for n, _ in model.named_parameters():
    if n in res[True][1]:
        a, b = res[True][1][n].double(), res[False][1][n].double()
        print(f'{n:70s} rel {((a - b).norm() / (b.norm() + 1e-30)).item():.3e}  |g| {b.norm().item():.3e}')
