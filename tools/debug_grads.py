import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oracle
from nesie_amd import kernels
from tests import _small
dev = torch.device("cuda:0")
model = _small.small_model()
model.train_cfg['pos_distance_thr'] = 1.0; model.train_cfg['neg_distance_thr'] = 1.5
pts, boxes, labels = _small.small_batch()
model.bbox_head.jitter_noise = _small.fixed_noise(2, 32)
with kernels.use_backend(oracle.OracleKernels()):
    want_l, want_g = _small.train_step_losses(model, pts, boxes, labels)
gmodel = copy.deepcopy(model).to(dev)
got_l, got_g = _small.train_step_losses(gmodel, pts.to(dev), boxes, labels)
got_l2, got_g2 = _small.train_step_losses(gmodel, pts.to(dev), boxes, labels)
torch.backends.cudnn.enabled = False
got_l3, got_g3 = _small.train_step_losses(gmodel, pts.to(dev), boxes, labels)
for k in want_l: print(k, want_l[k].item(), got_l[k].item(), got_l3[k].item())
gmax = max(g.abs().max().item() for g in want_g.values())
def report(a, b, tag):
    rows = []
    for n in a:
        denom = max(a[n].abs().max().item(), 1e-4 * gmax)
        rows.append(((b[n] - a[n]).abs().max().item() / denom, denom, n))
    rows.sort(reverse=True)
    print(tag)
    for r in rows[:6]: print("   %.3e  max|g|=%.3e  %s" % r)
report(want_g, got_g, "cpu vs gpu")
report(got_g, got_g2, "gpu vs gpu (rerun)")
report(want_g, got_g3, "cpu vs gpu (miopen off)")
