import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from nesie_amd.votenet import nesie_votenet_scannet_cfg
from tests import _small
dev = torch.device('cuda:0')
ocfg = nesie_votenet_scannet_cfg()['optimizer']
lr, wd = ocfg['lr'], ocfg['weight_decay']
noise = _small.fixed_noise(2, 256)
twin, twin_step, _ = bench.build_step(dev, 2, 77, lr, wd, graph=False, workload='pretrain', noise=noise)
model_g, step_g, bucket = bench.build_step(dev, 2, 77, lr, wd, graph=True, workload='pretrain', noise=noise)
inp = twin_step.inputs
with torch.no_grad():
    for pt, pg in zip(twin.parameters(), model_g.parameters()):
        pt.copy_(pg)
    for bt, bg in zip(twin.buffers(), model_g.buffers()):
        bt.copy_(bg)
name = 'backbone.SA_modules.0.mlps.0.layer2.bn.bias'
def eager_grads():
    for p in twin.parameters():
        p.grad = None
    losses = twin.forward_train(inp['points'], None, inp['gt'], None)
    twin.parse_losses(losses).backward()
    return {n: p.grad.detach().clone() for n, p in twin.named_parameters() if p.grad is not None}
g1 = eager_grads(); g2 = eager_grads()
print('eager run-to-run max diff', max((g1[n] - g2[n]).abs().max().item() for n in g1))
# graph leg: replay only g1 by calling step (updates weights too) -> gradient before clip is lost;
# instead re-run the eager path of the graph model
m = model_g
for p in m.parameters():
    p.grad = None
losses = m.forward_train(inp['points'], None, inp['gt'], None)
m.parse_losses(losses).backward()
gm = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
print('graph-model eager vs twin eager max diff', max((g1[n] - gm[n]).abs().max().item() for n in g1))
a = g1[name]
print('twin grad', a[:16])
# now the graph step's gradient: flat grad after the step is clipped in place; compare direction
step_g()
torch.cuda.synchronize()
gg = dict(zip([n for n, _ in m.named_parameters()], [p.grad for p in m.parameters()]))
tot = torch.sqrt(sum((g1[n].double() ** 2).sum() for n in g1)).item()
scale = min(1.0, 10.0 / (tot + 1e-6))
b = gg[name] / scale
print('graph grad', b[:16])
d = (a - b).abs()
print('max abs diff', d.max().item(), 'at', d.argmax().item(), a[d.argmax()].item(), b[d.argmax()].item(), 'max |a|', a.abs().max().item())
for n in g1:
    dd = (g1[n] - gg[n] / scale).abs().max().item() / (g1[n].abs().max().item() + 1e-30)
    if dd > 1e-3:
        print('  differs:', n, dd)
