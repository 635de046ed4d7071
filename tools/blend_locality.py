"""How many DISTINCT seeds do the 3-NN taps of one (proposal, face) touch in a real step?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nesie_amd.votenet import nesie_votenet_scannet_cfg
import nesie_amd.votenet.side_pooling as sp

dev = torch.device('cuda:0')
cfg = nesie_votenet_scannet_cfg()['optimizer']
model, step, bucket = bench.build_step(dev, 8, 1000, cfg['lr'], cfg['weight_decay'], graph=False)
gc = model.bbox_head.grid_conv
seen = []
orig = gc._blend_taps
def spy(origin_xyz, whole_grid, center):
    out = orig(origin_xyz, whole_grid, center)
    seen.append((out[0].clone(), center.shape[1], whole_grid.shape[1]))
    return out
gc._blend_taps = spy
for _ in range(3):
    step()
torch.cuda.synchronize()
for idx, K, n in seen[-2:]:
    B = idx.shape[0]
    per = n // K                      # grid points per proposal
    segs = 6 if per == 96 else 1
    G = per // segs
    groups = idx.view(B, K, segs, G * 3)                       # taps of one (proposal, face)
    s = groups.sort(-1).values
    distinct = 1 + (s[..., 1:] != s[..., :-1]).sum(-1).float()
    print(f'n={n} K={K} segs={segs} G={G}: taps/group={G*3}  distinct seeds/group: mean '
          f'{distinct.mean():.1f}  median {distinct.median():.0f}  p90 {distinct.flatten().kthvalue(int(distinct.numel()*0.9)).values:.0f}  max {distinct.max():.0f}')
    # per seed: how many taps land on it (load imbalance of a per-destination pass)
    cnt = torch.zeros(B, segs, 1024, device=dev)
    face = (torch.arange(n, device=dev) // G) % segs
    cnt.view(B, -1).scatter_add_(1, (face.view(1, n, 1) * 1024 + idx.long()).view(B, -1),
                                 torch.ones(B, n * 3, device=dev))
    nz = cnt[cnt > 0]
    print(f'   seeds hit: {nz.numel() / (B * segs):.0f} of 1024 per (scene, face); taps per hit seed: '
          f'mean {nz.mean():.0f}  median {nz.median():.0f}  max {nz.max():.0f}')
