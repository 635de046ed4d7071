"""blend_conv_backward under three index patterns: all taps on one seed (every tap hits a live
slot), sorted-by-seed queries, uniformly random seeds (every tap misses)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nesie_amd import kernels

dev = torch.device('cuda:0')
hip = kernels.backend_for(torch.empty(1, device=dev))
B, c, m, K, segs, G = 8, 256, 1024, 512, 6, 16
n = K * segs * G
g = torch.Generator(device=dev).manual_seed(0)
dy = torch.randn(B, segs, c, K * G, device=dev, generator=g)
w = torch.rand(B, n, 3, device=dev, generator=g)
rel = torch.randn(B, n, 3, device=dev, generator=g)
d_table = torch.zeros(B, m, segs * c, device=dev)
d_wx = torch.zeros(segs, c, 3, device=dev)


def run(idx, label):
    def f():
        hip.blend_conv_backward(dy, c, idx, w, rel, d_table, d_wx, segs, G)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            f()
    gr.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    gr.replay()
    e.record()
    torch.cuda.synchronize()
    print('%-28s %.3f ms for six faces' % (label, s.elapsed_time(e) / 10), flush=True)


run(torch.zeros(B, n, 3, dtype=torch.int32, device=dev), 'one seed (all hits)')
rnd = torch.randint(0, m, (B, n, 3), device=dev, generator=g, dtype=torch.int32)
run(rnd, 'random seeds (all misses)')
# per (proposal, face): 16 grid points drawing from a pool of P seeds
for pool in (4, 8, 16, 32):
    base = torch.randint(0, m, (B, K * segs, 1, pool), device=dev, generator=g)
    pick = torch.randint(0, pool, (B, K * segs, G, 3), device=dev, generator=g)
    idx = torch.gather(base.expand(-1, -1, G, -1), 3, pick).reshape(B, n, 3).int().contiguous()
    run(idx, f'{pool} seeds per face')
