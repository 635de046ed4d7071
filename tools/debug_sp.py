import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oracle
from nesie_amd import kernels
from nesie_amd.votenet import side_pooling as SP
from nesie_amd.votenet.nesie_head import GTBatch
from tests import _small
dev = torch.device("cuda:0")
model = _small.small_model()
pts, boxes, labels = _small.small_batch()
model.bbox_head.jitter_noise = _small.fixed_noise(2, 32)
ok = oracle.OracleKernels()
with kernels.use_backend(ok):
    x = model.extract_feat(pts)
    res = model.bbox_head(x, 'vote')
    center, size, heading, _ = model.bbox_head.jitter_bbox_preds(dict(res), 'ScanNet')
ep = {k: res[k].detach() for k in ['seed_points', 'seed_features', 'bbox_probs']}
center, size, heading = center.detach(), size.detach(), heading.detach()
gc = model.bbox_head.grid_conv
torch.manual_seed(1); tgt = torch.rand(2, 64, 18)
def run(gc, ep, c, s, h, tgt):
    for p in gc.parameters(): p.grad = None
    out = gc(c, s, h, dict(ep))
    loss = ((out['iou_scores'].sigmoid() - tgt) ** 2).sum() + (out['side_scores'].sigmoid() ** 2).sum()
    loss.backward()
    return loss.item(), {n: p.grad.detach().cpu().double().clone() for n, p in gc.named_parameters()}
with kernels.use_backend(ok):
    l32, g32 = run(gc, ep, center, size, heading, tgt)
gcg = copy.deepcopy(gc).to(dev)
lg, gg = run(gcg, {k: v.to(dev) for k, v in ep.items()}, center.to(dev), size.to(dev), heading.to(dev), tgt.to(dev))
# fp64 on CPU: three_nn via the oracle in fp32 (indices), blend in fp64 torch
def nn64(q, k):
    d = torch.empty(q.shape[0], q.shape[1], 3); i = torch.empty(q.shape[0], q.shape[1], 3, dtype=torch.int32)
    ok.three_nn_wrapper(q.shape[0], q.shape[1], k.shape[1], q.float().contiguous(), k.float().contiguous(), d, i)
    return d.sqrt().double(), i
def interp64(f, idx, w):
    B, C, M = f.shape; n = idx.shape[1]
    g = torch.gather(f.unsqueeze(2).expand(-1, -1, n, -1), 3, idx.long().unsqueeze(1).expand(-1, C, -1, -1))
    return (g * w.unsqueeze(1)).sum(-1)
SP.three_nn, SP.three_interpolate = nn64, interp64
gc64 = copy.deepcopy(gc).double()
l64, g64 = run(gc64, {k: v.double() for k, v in ep.items()}, center.double(), size.double(), heading.double(), tgt.double())
print(l32, lg, l64)
gmax = max(v.abs().max().item() for v in g64.values())
def rep(a, b, tag):
    rows = sorted((((b[n] - a[n]).abs().max().item() / max(a[n].abs().max().item(), 1e-3 * gmax), n) for n in a), reverse=True)
    print(tag, ["%.2e %s" % r for r in rows[:4]])
rep(g64, g32, "cpu32 vs cpu64"); rep(g64, gg, "gpu32 vs cpu64"); rep(g32, gg, "gpu32 vs cpu32")
