"""How many of the 1 024 seeds survive the per-proposal pruning radius of grid_taps_pruned_kernel on the
proposals of a real (random-init) supervised step?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from nesie_amd.kernels import HipKernels

calls = []
orig = HipKernels.grid_taps
def wrap(self, centre, size, heading, mult, plane, known):
    calls.append([t.detach().clone() for t in (centre, size, heading, mult, plane, known)])
    return orig(self, centre, size, heading, mult, plane, known)
HipKernels.grid_taps = wrap
dev = torch.device('cuda:0')
model, step, bucket = bench.build_step(dev, 8, 1000, 1e-3, 0.01, graph=False)
step()
torch.cuda.synchronize()
for centre, size, heading, mult, plane, known in calls:
    B, K = centre.shape[:2]
    gp = mult.shape[0]
    f = mult.view(1, 1, gp, 3) * size.view(B, K, 1, 3) / 2
    f = f + f * plane.view(1, 1, gp, 3)
    rad = f.norm(dim=-1).max(-1)[0]                       # (B, K): distance of the farthest grid point from the centre
    d = torch.cdist(centre, known)                        # (B, K, m)
    d3 = d.kthvalue(3, dim=-1)[0]
    keep = (d <= (d3 + 2 * rad).unsqueeze(-1)).sum(-1).float()
    print(f'gp {gp}: K {K}, size median {size.median().item():.2f} max {size.max().item():.2f}; radius median {rad.median().item():.2f}; '
          f'd3 median {d3.median().item():.2f}; survivors of {known.shape[1]}: median {keep.median().item():.0f} mean {keep.mean().item():.0f} '
          f'p90 {keep.flatten().kthvalue(int(0.9 * keep.numel()))[0].item():.0f}')
