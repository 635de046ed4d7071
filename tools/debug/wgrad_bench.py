"""Debug aid: nesie_pw_wgrad at the step's largest shapes -- time per launch, TFLOP/s, error vs float64."""
import sys

import torch

sys.path.insert(0, '.')
from nesie_amd.kernels import backend_for  # noqa: E402

dev = torch.device('cuda:0')
hip = backend_for(torch.empty(1, device=dev))
for nb, ng, co, ci, p, aff in [(48, 6, 128, 256, 8192, True), (48, 6, 256, 128, 8192, False),
                               (48, 6, 256, 128, 8192, True), (8, 1, 128, 256, 32768, True),
                               (8, 1, 128, 128, 65536, True), (8, 1, 64, 64, 131072, True),
                               (8, 1, 128, 320, 16384, False), (8, 1, 128, 128, 1024, True)]:
    g = torch.Generator(device=dev).manual_seed(1)
    dy = torch.randn(nb, co, p, device=dev, generator=g)
    x = torch.randn(nb, ci, p, device=dev, generator=g)
    coef = torch.rand(ng * ci, 4, device=dev, generator=g) + 0.5
    coef[:, 1] -= 1.0
    dw = torch.empty(ng, co, ci, device=dev)
    run = lambda: hip.pw_wgrad(dy, x, dw, ng=ng, x_coef=coef if aff else None, x_relu=aff)  # noqa: E731
    for _ in range(3):
        run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        run()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    want = torch.zeros(ng, co, ci, dtype=torch.float64, device=dev)
    for n in range(nb):
        a = x[n].double()
        if aff:
            c = coef.view(ng, ci, 4)[n % ng].double()
            a = torch.relu(a * c[:, 0:1] + c[:, 1:2])
        want[n % ng] += dy[n].double() @ a.t()
    err = float((dw.double() - want).abs().max() / want.abs().max())
    print((nb, ng, co, ci, p, aff), f'{ms:.4f} ms  {2 * nb * co * ci * p / ms / 1e9:.1f} TFLOP/s  rel err {err:.2e}')

print('---- fused norm backward + weight gradient vs the two separate launches')
for nb, ng, co, ci, p in [(48, 6, 128, 256, 8192), (48, 6, 256, 128, 8192), (8, 1, 64, 64, 131072),
                          (8, 1, 128, 128, 65536), (8, 1, 128, 256, 32768)]:
    g = torch.Generator(device=dev).manual_seed(1)
    da = torch.randn(nb, co, p, device=dev, generator=g)
    z = torch.randn(nb, co, p, device=dev, generator=g)
    x = torch.randn(nb, ci, p, device=dev, generator=g)
    xcoef = torch.rand(ng * ci, 4, device=dev, generator=g) + 0.5
    zcoef = torch.rand(ng * co, 4, device=dev, generator=g) + 0.5
    gamma = torch.randn(ng * co, device=dev, generator=g)
    part = torch.randn(ng * co, 512, 2, device=dev, generator=g)
    dz = torch.empty_like(da)
    dw = torch.empty(ng, co, ci, device=dev)
    dg, db = torch.empty(ng * co, device=dev), torch.empty(ng * co, device=dev)

    def separate():
        hip.bn_relu_backward_apply(da.view(nb // ng, ng * co, p), z.view(nb // ng, ng * co, p), gamma, None, zcoef,
                                   part, dz.view(nb // ng, ng * co, p), dg, db)
        hip.pw_wgrad(dz, x, dw, ng=ng, x_coef=xcoef, x_relu=True)

    def fused():
        hip.pw_wgrad_bn_backward(da, z, zcoef, gamma, part, x, dz, dw, dg, db, ng=ng, x_coef=xcoef)
    res = []
    for fn in (separate, fused):
        for _ in range(3):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        res.append(s.elapsed_time(e) / 20)
    print((nb, ng, co, ci, p), f'separate {res[0]:.4f} ms  fused {res[1]:.4f} ms')
