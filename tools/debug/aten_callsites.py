"""Which python lines launch the small ATen kernels of one eager training step (count, time)."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from nesie_amd.votenet import nesie_votenet_scannet_cfg

dev = torch.device('cuda:0')
cfg = nesie_votenet_scannet_cfg()['optimizer']
model, step, bucket = bench.build_step(dev, 8, 1000, cfg['lr'], cfg['weight_decay'], graph=False, workload=(sys.argv[1] if len(sys.argv) > 1 else 'pretrain'))
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step()
    torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith('aten::'):
        continue
    if not ev.kernels:
        continue
    k_us = sum(k.duration for k in ev.kernels)
    names = ' '.join(k.name for k in ev.kernels)
    if 'nesie::' in names or 'Cijk' in names:
        continue
    chain, par = [], ev.cpu_parent
    while par is not None:
        chain.append(par.name)
        par = par.cpu_parent
    site = ' < '.join(n.replace('autograd::engine::evaluate_function: ', 'bwd ') for n in chain[:3])
    frames = [f for f in (ev.stack or []) if f.startswith('nesie_amd/') or f.startswith('bench.py')]
    site = (site + ' @ ' if site else 'py ') + (frames[0] if frames else '?')
    if 'emcpy' in names or 'copyBuffer' in names:
        site = '[memcpy] ' + site
    a = acc[(ev.name, site[:150])]
    a[0] += len(ev.kernels); a[1] += k_us
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
tot = sum(v[0] for v in acc.values())
print('ATen launches with a python site:', tot)
for (op, site), (n, us) in rows[:70]:
    print(f'{n:4d} {us:8.1f} us  {op:22s} {site}')
