"""A few hundred graph-replayed training steps on freshly assembled batches: the loss must stay
finite and go down (synthetic ScanNet-shaped scenes, random-init weights, the bench optimiser)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nesie_amd.votenet import nesie_votenet_scannet_cfg

dev = torch.device('cuda:0')
cfg = nesie_votenet_scannet_cfg()['optimizer']
model, step, bucket = bench.build_step(dev, 8, 1000, 1e-3, cfg['weight_decay'], graph=True, resident=64)
hist = []
for i in range(301):
    loss = step()
    if i % 50 == 0:
        torch.cuda.synchronize()
        hist.append(float(loss))
        print(f'step {i:4d}  loss {hist[-1]:.4f}', flush=True)
assert all(l == l and l < 1e6 for l in hist), hist
print('decreased' if hist[-1] < hist[0] else 'NOT decreased', hist[0], '->', hist[-1])
