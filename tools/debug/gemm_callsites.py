"""Which python lines launch rocBLAS / hipBLASLt GEMMs in one eager training step (count, time)."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from nesie_amd.votenet import nesie_votenet_scannet_cfg

dev = torch.device('cuda:0')
cfg = nesie_votenet_scannet_cfg()['optimizer']
model, step, bucket = bench.build_step(dev, 8, 1000, cfg['lr'], cfg['weight_decay'], graph=False,
                                       workload=(sys.argv[1] if len(sys.argv) > 1 else 'pretrain'))
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step()
    torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
        continue
    ks = [k for k in ev.kernels if k.name.startswith('Cijk')]
    if not ks or not ev.name.startswith('aten::'):
        continue
    chain, par = [], ev.cpu_parent
    while par is not None:
        chain.append(par.name)
        par = par.cpu_parent
    frames = [f for f in (ev.stack or []) if f.startswith('nesie_amd/') or f.startswith('bench.py')]
    site = ' < '.join(n.replace('autograd::engine::evaluate_function: ', 'bwd ') for n in chain[:2]) + ' @ ' + (frames[0] if frames else '?')
    a = acc[(ev.name, str(ev.input_shapes)[:80], site[:140])]
    a[0] += len(ks); a[1] += sum(k.duration for k in ks)
for (op, shp, site), (n, us) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f'{n:3d} {us:7.1f} us  {op:14s} {shp:80s} {site}')
