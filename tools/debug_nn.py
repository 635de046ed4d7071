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
rec = {}
def tap(backend, tag):
    for meth in ["three_nn_wrapper", "ball_query_wrapper", "furthest_point_sampling_wrapper"]:
        inner = getattr(backend, meth)
        def w(*a, inner=inner, meth=meth):
            inner(*a)
            rec.setdefault((tag, meth), []).append(a[-1].detach().cpu().clone())
        setattr(backend, meth, w)
ok = oracle.OracleKernels(); tap(ok, "cpu")
with kernels.use_backend(ok):
    want_l, want_g = _small.train_step_losses(model, pts, boxes, labels)
gmodel = copy.deepcopy(model).to(dev)
hip = kernels.backend_for(torch.empty(1, device=dev)); tap(hip, "gpu")
got_l, got_g = _small.train_step_losses(gmodel, pts.to(dev), boxes, labels)
for meth in ["three_nn_wrapper", "ball_query_wrapper", "furthest_point_sampling_wrapper"]:
    for i, (a, b) in enumerate(zip(rec[("cpu", meth)], rec[("gpu", meth)])):
        print(meth, i, tuple(a.shape), "mismatching entries:", (a != b).sum().item())
