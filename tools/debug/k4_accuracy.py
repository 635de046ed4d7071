"""SA1's stack (4 -> 64 -> 64 -> 128, max over 64) with the first activation stored vs rebuilt
(fused_mlp.SA1_K4), both against a float64 evaluation: output and the nine parameter gradients."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nesie_amd.mmdet3d_ops import fused_mlp

dev = torch.device('cuda:0')
B, M, ns = int(sys.argv[1]) if len(sys.argv) > 1 else 2, int(sys.argv[2]) if len(sys.argv) > 2 else 2048, 64
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(B, 4, M, ns, device=dev, generator=g) * 0.4
x[:, 3] = torch.rand(B, M, 1, device=dev, generator=g).expand(B, M, ns) * 2.5 + 0.2     # a height: constant per group
ch = [4, 64, 64, 128]
params = []
for i in range(3):
    params += [torch.randn(ch[i + 1], ch[i], 1, 1, device=dev, generator=g) / ch[i] ** 0.5,
               torch.rand(ch[i + 1], device=dev, generator=g) + 0.5, torch.randn(ch[i + 1], device=dev, generator=g) * 0.2]
wout = torch.randn(B, 128, M, device=dev, generator=g)


def run64():
    ps = [p.double().requires_grad_(True) for p in params]
    h = x.double().view(B, 4, M * ns)
    for i in range(3):
        z = torch.matmul(ps[3 * i].view(ch[i + 1], ch[i]), h)
        mean, var = z.mean((0, 2), keepdim=True), z.var((0, 2), unbiased=False, keepdim=True)
        h = torch.relu((z - mean) / torch.sqrt(var + 1e-5) * ps[3 * i + 1].view(1, -1, 1) + ps[3 * i + 2].view(1, -1, 1))
    out = h.view(B, 128, M, ns).max(-1)[0]
    (out * wout.double()).sum().backward()
    return out.detach(), [p.grad for p in ps]


def run32(k4):
    fused_mlp.SA1_K4 = k4
    ps = [p.clone().requires_grad_(True) for p in params]
    bufs = [(torch.zeros(c, device=dev), torch.ones(c, device=dev), 0.1, 1e-5) for c in ch[1:]]
    out = fused_mlp.SAStackFn.apply(x, bufs, 3, *ps)
    (out * wout).sum().backward()
    return out.detach(), [p.grad for p in ps]


ref = run64()
for k4 in (False, True):
    got = run32(k4)
    eo = ((got[0].double() - ref[0]).norm() / ref[0].norm()).item()
    eg = [((a.double() - b).norm() / b.norm()).item() for a, b in zip(got[1], ref[1])]
    print('k4' if k4 else 'stored', f'out {eo:.2e}', ' '.join(f'{e:.1e}' for e in eg))
