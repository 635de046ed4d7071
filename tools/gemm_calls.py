"""Every GEMM one eager training step launches (forward and backward), grouped by operand
shapes: calls, device time, TFLOP/s.  usage: python tools/gemm_calls.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from nesie_amd.votenet import nesie_votenet_scannet_cfg

dev = torch.device('cuda:0')
cfg = nesie_votenet_scannet_cfg()['optimizer']
model, step, bucket = bench.build_step(dev, 8, 1000, cfg['lr'], cfg['weight_decay'], graph=False)
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key not in ('aten::bmm', 'aten::mm', 'aten::addmm', 'aten::baddbmm'):
        continue
    shp = [s for s in e.input_shapes if s]
    t = e.device_time_total / 1e3   # ms
    a, b = (shp[-2], shp[-1]) if e.key in ('aten::bmm', 'aten::mm') else (shp[1], shp[2])
    if len(a) == 3:
        fl = 2.0 * a[0] * a[1] * a[2] * b[2]
    else:
        fl = 2.0 * a[0] * a[1] * b[1]
    rows.append((t, e.count, e.key, a, b, fl * e.count / 1e12))
rows.sort(key=lambda r: -r[0])
tot = sum(r[0] for r in rows)
print(f'total GEMM device time {tot:.3f} ms over {sum(r[1] for r in rows)} calls')
for t, c, k, a, b, tf in rows:
    byts = 4.0 * c * ((a[-1] * a[-2] + b[-1] * b[-2] + a[-2] * b[-1]) * (a[0] if len(a) == 3 else 1))
    print(f'{t:7.3f} ms {c:3d}x {k:11s} {str(a):22s} x {str(b):22s} {tf / (t / 1e3) if t else 0:6.1f} TF/s  {byts / t / 1e9 if t else 0:6.2f} TB/s(min traffic)')
