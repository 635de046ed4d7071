import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nesie_amd.votenet.side_pooling import MiniPointNet, _score_head
torch.manual_seed(0)
dev = torch.device("cuda:0")
net = torch.nn.Sequential()
mp = MiniPointNet(259, 128); head = _score_head(128, 18)
x = torch.randn(2, 259, 64, 16)
x[:, 3:] = x[:, 3:].abs()   # relu-like features
tgt = torch.rand(2, 18, 64)
def run(mp, head, x, tgt):
    for p in list(mp.parameters()) + list(head.parameters()): p.grad = None
    out = head(mp(x)).sigmoid()
    loss = ((out - tgt) ** 2).sum()
    loss.backward()
    return loss.item(), {n: p.grad.clone().cpu().double() for n, p in list(mp.named_parameters()) + [("h." + k, v) for k, v in head.named_parameters()]}
l32, g32 = run(mp, head, x, tgt)
mp64, head64 = copy.deepcopy(mp).double(), copy.deepcopy(head).double()
l64, g64 = run(mp64, head64, x.double(), tgt.double())
mpg, headg = copy.deepcopy(mp).to(dev), copy.deepcopy(head).to(dev)
lg, gg = run(mpg, headg, x.to(dev), tgt.to(dev))
torch.backends.cudnn.enabled = False
lg2, gg2 = run(mpg, headg, x.to(dev), tgt.to(dev))
print(l32, l64, lg, lg2)
def rep(a, b, tag):
    gmax = max(a[n].abs().max().item() for n in a)
    rows = sorted((((b[n] - a[n]).abs().max().item() / max(a[n].abs().max().item(), 1e-4 * gmax), n) for n in a), reverse=True)
    print(tag, ["%.2e %s" % r for r in rows[:4]])
rep(g64, g32, "cpu32 vs cpu64")
rep(g64, gg, "gpu32 vs cpu64")
rep(g64, gg2, "gpu32(miopen off) vs cpu64")
