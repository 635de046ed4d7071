"""Debug aid: does a captured graph gain from running the weight-gradient launches on a second
stream beside the input-gradient chain?  Six layers of (input gradient 128 -> 128, weight gradient
128 x 128) over 8 x 32768 positions, serial vs forked, replay time per graph."""
import sys

import torch

sys.path.insert(0, '.')
from nesie_amd.kernels import backend_for  # noqa: E402

dev = torch.device('cuda:0')
hip = backend_for(torch.empty(1, device=dev))
nb, c, p, L = 8, 128, 32768, 6
g = torch.Generator(device=dev).manual_seed(0)
dz = [torch.randn(nb, c, p, device=dev, generator=g) for _ in range(L + 1)]
x = [torch.randn(nb, c, p, device=dev, generator=g) for _ in range(L)]
w = torch.randn(1, c, c, device=dev, generator=g) / c ** 0.5
coef = torch.rand(c, 4, device=dev, generator=g) + 0.5
dw = [torch.empty(1, c, c, device=dev) for _ in range(L)]
side = torch.cuda.Stream(dev)


def chain(fork):
    main = torch.cuda.current_stream()
    for l in range(L):
        if fork:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                hip.pw_wgrad(dz[l], x[l], dw[l], ng=1, x_coef=coef, x_relu=True)
        else:
            hip.pw_wgrad(dz[l], x[l], dw[l], ng=1, x_coef=coef, x_relu=True)
        hip.pw_layer_forward(dz[l], w, in_coef=None, y=dz[l + 1])
    if fork:
        main.wait_stream(side)


for fork in (False, True, False, True):
    chain(fork)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(graph, stream=s):
            chain(fork)
    torch.cuda.synchronize()
    for _ in range(3):
        graph.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        graph.replay()
    b.record()
    torch.cuda.synchronize()
    print('forked' if fork else 'serial', f'{a.elapsed_time(b) / 20:.4f} ms per replay')
