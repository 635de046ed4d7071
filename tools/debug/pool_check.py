"""Debug aid: the layer kernel's store + pool variants at the quality head's full-size shapes --
stored output against a float64 product, pooled extremum / position against the stored output."""
import sys

import torch

sys.path.insert(0, '.')
from nesie_amd.kernels import backend_for  # noqa: E402

dev = torch.device('cuda:0')
hip = backend_for(torch.empty(1, device=dev))
bad = 0
for nb, ng, k, cout, p, pg, stats in [(8, 1, 256, 128, 32768, 32, False), (48, 6, 256, 128, 8192, 16, False),
                                      (8, 1, 128, 256, 32768, 32, True), (8, 1, 64, 128, 65536, 32, True),
                                      (8, 1, 128, 128, 16384, 16, True)]:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(nb, k, p, device=dev, generator=g)
    w = torch.randn(ng, cout, k, device=dev, generator=g) / k ** 0.5
    coef = torch.rand(ng * k, 4, device=dev, generator=g) + 0.5
    coef[:, 1] -= 1.0
    y = torch.empty(nb, cout, p, device=dev)
    npg = p // pg
    pool = (torch.empty(nb, cout, npg, device=dev), torch.empty(nb, cout, npg, device=dev) if stats else None,
            torch.empty(nb, cout, npg, dtype=torch.uint8, device=dev),
            torch.empty(nb, cout, npg, dtype=torch.uint8, device=dev) if stats else None)
    part = torch.zeros(ng, hip.pw_stat_slots(nb, ng, k, cout, p), cout, 4, device=dev) if stats else None
    hip.pw_layer_forward(x, w, ng=ng, in_coef=coef, in_relu=True, y=y, stat_part=part, pool_group=pg,
                         pool_min=stats, pool_out=pool)
    torch.cuda.synchronize()
    err = 0.0
    for n in range(nb):
        c = coef.view(ng, k, 4)[n % ng]
        a = torch.relu(x[n].double() * c[:, 0:1].double() + c[:, 1:2].double())
        err = max(err, float((w[n % ng].double() @ a - y[n].double()).abs().max()))
    yv = y.view(nb, cout, npg, pg)
    mx, am = yv.max(-1)
    okv = torch.equal(mx, pool[0])
    picked = torch.gather(yv, 3, pool[2].long().unsqueeze(-1)).squeeze(-1)
    oka = torch.equal(picked, mx)
    okm = True
    if stats:
        okm = torch.equal(yv.min(-1).values, pool[1]) and \
            torch.equal(torch.gather(yv, 3, pool[3].long().unsqueeze(-1)).squeeze(-1), pool[1])
    print((nb, ng, k, cout, p, pg, stats), 'y err', f'{err:.2e}', 'max', okv, 'arg', oka, 'min', okm,
          'first-arg', float((pool[2].long() == am).float().mean()))
    bad += (not okv) + (not oka) + (not okm) + (err > 1e-4)
sys.exit(1 if bad else 0)
