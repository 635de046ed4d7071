import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oracle
from nesie_amd import kernels
from nesie_amd.votenet.nesie_head import GTBatch
from tests import _small
dev = torch.device("cuda:0")
model = _small.small_model()
model.train_cfg['pos_distance_thr'] = 1.0; model.train_cfg['neg_distance_thr'] = 1.5
pts, boxes, labels = _small.small_batch()
model.bbox_head.jitter_noise = _small.fixed_noise(2, 32)
names = ['bbox_head.grid_conv.mlps_before.0.first_conv.3.weight', 'bbox_head.grid_conv.mlps_head.6.0.weight',
         'bbox_head.grid_conv.mlps_before.6.first_conv.0.weight']
def per_term(m, p):
    gt = GTBatch.collate(boxes, labels, p.device)
    losses = m.forward_train(p, None, gt, None)
    out = {}
    params = dict(m.named_parameters())
    for k, v in losses.items():
        gs = torch.autograd.grad(v, [params[n] for n in names], retain_graph=True, allow_unused=True)
        out[k] = [None if g is None else g.detach().cpu() for g in gs]
    return out
with kernels.use_backend(oracle.OracleKernels()):
    c = per_term(model, pts)
gmodel = copy.deepcopy(model).to(dev)
g = per_term(gmodel, pts.to(dev))
for k in c:
    for i, n in enumerate(names):
        if c[k][i] is None: continue
        a, b = c[k][i], g[k][i]
        print(k, n.split('grid_conv.')[1], "max|g| %.3e  err %.3e" % (a.abs().max().item(), (a - b).abs().max().item()))
