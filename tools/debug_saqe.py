import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, oracle
from nesie_amd import kernels
from nesie_amd.votenet import build_saqe_votenet
from nesie_amd.votenet.detector import saqe_votenet_scannet_cfg
from tests import _small
dev = torch.device("cuda:0")
cfg = _small.small_cfg(); scfg = saqe_votenet_scannet_cfg()
cfg['bbox_head'].update(angle_loss=scfg['bbox_head']['angle_loss'], angle_pred_loss=scfg['bbox_head']['angle_pred_loss'])
cfg['head_type'] = 'SAQEHead'; cfg['train_cfg'].update(pos_distance_thr=1.0, neg_distance_thr=1.5)
torch.manual_seed(0)
model = build_saqe_votenet(cfg)
pts, boxes, labels = _small.small_batch()
model.bbox_head.jitter_noise = _small.fixed_noise(2, 32)
with kernels.use_backend(oracle.OracleKernels()):
    wl, wg = _small.train_step_losses(model, pts, boxes, labels)
gm = copy.deepcopy(model).to(dev)
gl, gg = _small.train_step_losses(gm, pts.to(dev), boxes, labels)
gl2, gg2 = _small.train_step_losses(gm, pts.to(dev), boxes, labels)
for k in wl: print(k, wl[k].item(), gl[k].item())
gmax = max(g.abs().max().item() for g in wg.values())
def rep(a, b, tag):
    rows = sorted((((b[n]-a[n]).norm().item(), (b[n]-a[n]).abs().max().item()/max(a[n].abs().max().item(), 1e-3*gmax), n) for n in a), reverse=True)
    print(tag); [print("   l2err %.3e  relmax %.2e  %s" % r) for r in rows[:8]]
rep(wg, gg, "cpu vs gpu"); rep(gg, gg2, "gpu vs gpu")
