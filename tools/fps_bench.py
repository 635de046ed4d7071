"""D-FPS kernel time and time per dependent round: 40000 -> 2048 (8 scenes and 1 scene: the pruned
kernel) and the register-resident kernel's shapes (SA2..SA4 and the vote aggregation)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nesie_amd.mmdet3d_ops import furthest_point_sample
from nesie_amd.scenes import make_batch

dev = torch.device('cuda:0')
pts, _, _ = make_batch(1000, 8, 40000)
cloud = pts[..., :3].contiguous().to(dev)
for b, n, m in ((8, 40000, 2048), (1, 40000, 2048), (8, 2048, 1024), (8, 1024, 512), (8, 512, 256),
                (8, 1024, 256)):
    xyz = cloud[:b, :n].contiguous()
    for _ in range(3):
        idx = furthest_point_sample(xyz, m)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        idx = furthest_point_sample(xyz, m)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'B={b} {n}->{m}: {ms:.3f} ms per call, {ms * 1e3 / (m - 1):.3f} us per round, '
          f'checksum {int(idx.long().sum())}')
