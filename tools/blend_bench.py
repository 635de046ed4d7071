"""The blend backward at the step's shapes (side grid: 6 faces x 512 proposals x 16 points; box
grid: 512 x 64), staged fixed-order form with the norm backward on the tile load (the form the step
runs), on taps with the locality of a real step (~19 distinct seeds per 16-query group).
usage: python tools/blend_bench.py [side|box]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nesie_amd import kernels

dev = torch.device('cuda:0')
hip = kernels.backend_for(torch.empty(1, device=dev))
which = sys.argv[1] if len(sys.argv) > 1 else 'side'
B, c, m, K = 8, 256, 1024, 512
segs, G = (6, 16) if which == 'side' else (1, 64)
n = K * segs * G
g = torch.Generator(device=dev).manual_seed(0)
dy = torch.randn(B, segs, c, K * G, device=dev, generator=g)
z = torch.randn(B, segs, c, K * G, device=dev, generator=g)
bnb = torch.rand(segs * c, 8, device=dev, generator=g)
w = torch.rand(B, n, 3, device=dev, generator=g)
rel = torch.randn(B, n, 3, device=dev, generator=g)
d_table = torch.zeros(B, m, segs * c, device=dev)
d_wx = torch.zeros(segs, c, 3, device=dev)
pool = 19 if which == 'side' else 40      # distinct seeds of a proposal's face / of its whole box grid
base = torch.randint(0, m, (B, K * segs, 1, pool), device=dev, generator=g)
pick = torch.randint(0, pool, (B, K * segs, G, 3), device=dev, generator=g)
idx = torch.gather(base.expand(-1, -1, G, -1), 3, pick).reshape(B, n, 3).int().contiguous()


def f():
    hip.blend_conv_backward(dy, c, idx, w, rel, d_table, d_wx, segs, G, bn_z=z, bnb=bnb)


for _ in range(3):
    f()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    f()
e.record()
torch.cuda.synchronize()
gb = (2 * dy.numel() * 4) / 1e9
print(f'{which}: {s.elapsed_time(e) / 10 * 1e3:.1f} us per call (rows + slot index + gather + d_wx sum), '
      f'{gb:.2f} GB of (dA, Z) read; NESIE_BLEND_ABL={os.environ.get("NESIE_BLEND_ABL", "0")}', flush=True)
