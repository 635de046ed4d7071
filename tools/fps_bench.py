"""D-FPS 40000 -> 2048 (8 scenes and 1 scene): kernel time and time per dependent round."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nesie_amd.mmdet3d_ops import furthest_point_sample
from nesie_amd.scenes import make_batch

dev = torch.device('cuda:0')
for b in (8, 1):
    pts, _, _ = make_batch(1000, b, 40000)
    xyz = pts[..., :3].contiguous().to(dev)
    for _ in range(3):
        idx = furthest_point_sample(xyz, 2048)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        idx = furthest_point_sample(xyz, 2048)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'B={b}: {ms:.3f} ms per call, {ms * 1e3 / 2047:.3f} us per round, checksum {int(idx.long().sum())}')
