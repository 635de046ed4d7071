"""Debug aid: how much slower do representative main-stream kernels run while the sampling kernel of
the index chain (D-FPS 40000 -> 2048, one 1024-thread workgroup per scene, 160 KB of LDS) is resident
on a side stream?  Each kernel is timed with HIP events over back-to-back launches, side stream idle
vs busy."""
import sys

import torch

sys.path.insert(0, '.')
from nesie_amd import kernels  # noqa: E402
from nesie_amd.mmdet3d_ops.furthest_point_sample import FurthestPointSampling  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    be = kernels.backend_for(torch.zeros(1, device=dev))
    g = torch.Generator(device=dev).manual_seed(0)
    pts = torch.rand(8, 40000, 3, device=dev, generator=g) * 8
    side = torch.cuda.Stream(dev, priority=-1)

    def rnd(*shape):
        return torch.randn(*shape, device=dev, generator=g)

    cases = {}
    x128, w128, y128 = rnd(8, 128, 32768), rnd(1, 128, 128), torch.empty(8, 128, 32768, device=dev)
    cases['layer 128->128, 8x32768 (2 wg/CU, 64 KB LDS)'] = lambda: be.pw_layer_forward(x128, w128, y=y128)
    x64, w64, y64 = rnd(8, 64, 131072), rnd(1, 64, 64), torch.empty(8, 64, 131072, device=dev)
    cases['layer 64->64, 8x131072 (4 wg/CU)'] = lambda: be.pw_layer_forward(x64, w64, y=y64)
    x256, w256, y256 = rnd(8, 256, 1024), rnd(1, 256, 256), torch.empty(8, 256, 1024, device=dev)
    cases['layer 256->256, 8x1024 (1-D chain, one tile per workgroup)'] = lambda: be.pw_layer_forward(x256, w256, y=y256)
    dy, xw, dw = rnd(8, 256, 32768), rnd(8, 128, 32768), torch.empty(1, 256, 128, device=dev)
    cases['weight gradient 256x128, 8x32768'] = lambda: be.pw_wgrad(dy, xw, dw)
    a, b = rnd(64 << 20), torch.empty(64 << 20, device=dev)
    cases['ATen add, 256 MB'] = lambda: torch.add(a, 1.0, out=b)
    s = rnd(1 << 16)
    cases['ATen add, 256 KB (launch-bound)'] = lambda: s.add_(1.0)

    def timed(fn, n=30):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    def busy(fn, n=30):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(8):                          # ~ 18 ms of resident sampling workgroups
                FurthestPointSampling.apply(pts, 2048)
        # let the sampling workgroups become resident before the timed launches start
        torch.cuda._sleep(2_000_000)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    FurthestPointSampling.apply(pts, 2048)
    torch.cuda.synchronize()
    for cus in [int(a) for a in sys.argv[1:]] or (256, 248):
        print(f'persistent grids sized for {cus} CUs (HipKernels.cu_budget)')
        print(f'{"kernel":62s} {"alone us":>9s} {"beside FPS us":>13s} {"ratio":>6s}')
        with kernels.HipKernels.cu_budget(cus):
            for name, fn in cases.items():
                n = 30 if 'launch-bound' not in name and '1-D' not in name else 200
                t0, t1 = timed(fn, n), busy(fn, n)
                print(f'{name:62s} {t0:9.1f} {t1:13.1f} {t1 / t0:6.2f}')


if __name__ == '__main__':
    main()
