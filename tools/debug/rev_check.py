"""Debug aid: the layer kernel on operands big enough for the reversed tile order, against a
float64 product -- per shape: max error of the output, of the statistics, of the pooled values."""
import os, sys
sys.path.insert(0, '.')
import torch
from nesie_amd import kernels
dev = torch.device('cuda:0')
hip = kernels.backend_for(torch.empty(1, device=dev))
g = torch.Generator(device=dev).manual_seed(0)
for nb, ng, k, cout, p in [(8, 1, 64, 64, 131072), (8, 1, 131, 128, 32768), (8, 1, 128, 128, 32768), (8, 1, 128, 256, 32768),
                           (48, 6, 256, 128, 8192), (48, 6, 128, 256, 8192), (8, 1, 259, 128, 8192)]:
    x = torch.randn(nb, k, p, device=dev, generator=g)
    w = torch.randn(ng, cout, k, device=dev, generator=g) * 0.1
    y = torch.empty(nb, cout, p, device=dev)
    part = torch.empty(ng, hip.pw_stat_slots(nb, ng, k, cout, p), cout, 4, device=dev)
    hip.pw_layer_forward(x, w, ng=ng, y=y, stat_part=part)
    torch.cuda.synchronize()
    err = 0.0
    for n in range(0, nb, max(1, nb // 4)):
        ref = (w[n % ng].double() @ x[n].double())
        err = max(err, float((y[n].double() - ref).abs().max()))
    cnt = part[..., 0].sum(1)
    print(f'{nb}x{p} {k}->{cout} ng {ng}: operand {nb * k * p * 4 / 1e6:.0f} MB  max |y - ref| {err:.2e}  stat counts {cnt.min().item():.0f}..{cnt.max().item():.0f} (want {nb // ng * p})', flush=True)
print('--- pooled epilogue (store + stats + max/min over 16 positions) and input gradient with the norm reduction')
for nb, ng, k, cout, p in [(8, 1, 128, 256, 32768), (8, 1, 128, 128, 32768), (2, 1, 128, 256, 32768)]:
    x = torch.randn(nb, k, p, device=dev, generator=g)
    w = torch.randn(ng, cout, k, device=dev, generator=g) * 0.1
    y = torch.empty(nb, cout, p, device=dev)
    part = torch.empty(ng, hip.pw_stat_slots(nb, ng, k, cout, p), cout, 4, device=dev)
    npg = p // 16
    pool_out = (torch.empty(nb, cout, npg, device=dev), torch.empty(nb, cout, npg, device=dev),
                torch.empty(nb, cout, npg, dtype=torch.uint8, device=dev), torch.empty(nb, cout, npg, dtype=torch.uint8, device=dev))
    coef = torch.rand(ng * k, 4, device=dev, generator=g) + 0.5
    hip.pw_layer_forward(x, w, ng=ng, in_coef=coef, in_relu=True, y=y, stat_part=part, pool_group=16, pool_min=True, pool_out=pool_out)
    torch.cuda.synchronize()
    yy = y.view(nb, cout, npg, 16)
    ok_max = torch.equal(pool_out[0], yy.max(-1).values), torch.equal(pool_out[1], yy.min(-1).values)
    act = torch.relu(x.double() * coef[:, 0].double().view(1, -1, 1) + coef[:, 1].double().view(1, -1, 1))
    err = max(float((y[n].double() - w[0].double() @ act[n]).abs().max()) for n in (0, nb - 1))
    print(f'{nb}x{p} {k}->{cout}: pooled max/min equal the stored output: {ok_max}, max |y - ref| {err:.2e}', flush=True)
    # input gradient + norm reduction
    z = torch.randn(nb, k, p, device=dev, generator=g)
    zc = torch.rand(ng * k, 4, device=dev, generator=g)
    da = torch.empty(nb, k, p, device=dev)
    dy = torch.randn(nb, cout, p, device=dev, generator=g)
    partb = hip.pw_dgrad_bn_reduce(dy, w.transpose(1, 2), z, zc, da, ng=ng)
    torch.cuda.synchronize()
    ref = w[0].double().t() @ dy[nb - 1].double()
    gg = torch.where((z.double() * zc[:, 0].double().view(1, -1, 1) + zc[:, 1].double().view(1, -1, 1)) > 0, da.double(), torch.zeros((), dtype=torch.float64, device=dev))
    s0 = gg.sum((0, 2))
    print(f'   dgrad: max |da - ref| {float((da[nb - 1].double() - ref).abs().max()):.2e}; sum g: kernel {float(partb[:, :, 0].double().sum(1)[3]):.4f} vs {float(s0[3]):.4f}', flush=True)
print('--- weight gradients (plain, and fused with the norm backward in place)')
for nb, ng, co, ci, p in [(8, 1, 128, 131, 32768), (8, 1, 256, 128, 32768), (8, 1, 128, 128, 32768), (8, 1, 64, 64, 131072)]:
    dy = torch.randn(nb, co, p, device=dev, generator=g)
    x = torch.randn(nb, ci, p, device=dev, generator=g)
    dw = torch.empty(ng, co, ci, device=dev)
    hip.pw_wgrad(dy, x, dw, ng=ng, x_coef=None, x_relu=False)
    ref = torch.einsum('ncp,nkp->ck', dy.double(), x.double())
    print(f'{nb}x{p} {co}x{ci}: plain  max |dw - ref| / max |ref| {float((dw[0].double() - ref).abs().max() / ref.abs().max()):.2e}', flush=True)
    if hip.pw_wgrad_bn_supported(co, ci, p):
        z = torch.randn(nb, co, p, device=dev, generator=g)
        zc = torch.rand(ng * co, 4, device=dev, generator=g) + 0.1
        zc[:, 1] -= 0.5
        gamma = torch.rand(ng * co, device=dev, generator=g) + 0.5
        da = torch.randn(nb, co, p, device=dev, generator=g)
        gg = torch.where(torch.addcmul(zc[:, 1].view(1, -1, 1), z, zc[:, 0].view(1, -1, 1)) > 0, da, torch.zeros_like(da)).double()
        zhat = (z.double() - zc[:, 2].double().view(1, -1, 1)) * zc[:, 3].double().view(1, -1, 1)
        s0, s1 = gg.sum((0, 2)), (gg * zhat).sum((0, 2))
        n = nb * p
        a = gamma.double() * zc[:, 3].double()
        dz_ref = a.view(1, -1, 1) * (gg - (s0 / n).view(1, -1, 1) - zhat * (s1 / n).view(1, -1, 1))
        part = torch.stack([s0, s1], -1).view(ng * co, 1, 2).float().contiguous()
        dw2 = torch.empty(ng, co, ci, device=dev)
        dgamma, dbeta = torch.empty(ng * co, device=dev), torch.empty(ng * co, device=dev)
        da_io = da.clone()
        hip.pw_wgrad_bn_backward(da_io, z, zc, gamma, part, x, da_io, dw2, dgamma, dbeta, ng=ng, x_coef=None, x_relu=False)
        torch.cuda.synchronize()
        dw_ref = torch.einsum('ncp,nkp->ck', dz_ref, x.double())
        print(f'   fused in place: max |dz - ref| {float((da_io.double() - dz_ref).abs().max()):.2e}  max |dw - ref| / max {float((dw2[0].double() - dw_ref).abs().max() / dw_ref.abs().max()):.2e}', flush=True)
print('--- pooled tail WITHOUT store (identity lane order), pool granule 32, K = 64 geometry')
for nb, ng, k, cout, p, pg in [(8, 1, 64, 128, 131072, 32), (2, 1, 64, 128, 131072, 32), (48, 6, 256, 128, 8192, 16)]:
    x = torch.randn(nb, k, p, device=dev, generator=g)
    w = torch.randn(ng, cout, k, device=dev, generator=g) * 0.1
    coef = torch.rand(ng * k, 4, device=dev, generator=g) + 0.5
    npg = p // pg
    mk = lambda: (torch.empty(nb, cout, npg, device=dev), torch.empty(nb, cout, npg, device=dev),
                  torch.empty(nb, cout, npg, dtype=torch.uint8, device=dev), torch.empty(nb, cout, npg, dtype=torch.uint8, device=dev))
    po = mk()
    part = torch.empty(ng, hip.pw_stat_slots(nb, ng, k, cout, p), cout, 4, device=dev)
    hip.pw_layer_forward(x, w, ng=ng, in_coef=coef, in_relu=True, y=None, stat_part=part, pool_group=pg, pool_min=True, pool_out=po)
    torch.cuda.synchronize()
    act = torch.relu(x * coef[:, 0].view(ng, 1, k, 1).expand(ng, nb // ng, k, 1).transpose(0, 1).reshape(nb, k, 1) + coef[:, 1].view(ng, 1, k, 1).expand(ng, nb // ng, k, 1).transpose(0, 1).reshape(nb, k, 1))
    bad = 0
    for n in (0, 1, nb - 1):
        ref = (w[n % ng].double() @ act[n].double()).view(cout, npg, pg)
        bad = max(bad, float((po[0][n].double() - ref.max(-1).values).abs().max()), float((po[1][n].double() - ref.min(-1).values).abs().max()))
    cnt = part[..., 0].sum(1)
    print(f'{nb}x{p} {k}->{cout} pg {pg}: max |pooled extremum - ref| {bad:.2e}; stat counts {cnt.min().item():.0f}..{cnt.max().item():.0f} (want {nb // ng * p})', flush=True)
print('--- plain input gradient into a channel slice of a wider tensor (transposed weight view)')
for nb, k, cout, p, lead in [(8, 128, 128, 32768, 3), (2, 128, 128, 32768, 3), (8, 128, 256, 32768, 3)]:
    dy = torch.randn(nb, k, p, device=dev, generator=g)
    w2 = torch.randn(k, cout + lead, device=dev, generator=g) * 0.1          # (cout_layer = k rows, c0 = cout + lead columns)
    dx = torch.full((nb, cout + lead, p), float('nan'), device=dev)
    hip.pw_layer_forward(dy, w2[:, lead:].t().unsqueeze(0), y=dx[:, lead:])
    torch.cuda.synchronize()
    err = max(float((dx[n, lead:].double() - w2[:, lead:].double().t() @ dy[n].double()).abs().max()) for n in (0, 1, nb - 1))
    print(f'{nb}x{p} {k}->{cout} into [:, {lead}:]: max |dx - ref| {err:.2e}, NaNs left in the slice {int(torch.isnan(dx[:, lead:]).sum())}, lead rows untouched {bool(torch.isnan(dx[:, :lead]).all())}', flush=True)
print('--- statistics partials of the K <= 64 geometry: merged mean / variance against the stored output')
for nb, k, cout, p in [(8, 64, 64, 131072), (8, 64, 128, 131072), (2, 64, 64, 131072)]:
    x = torch.randn(nb, k, p, device=dev, generator=g) + 0.3
    w = torch.randn(1, cout, k, device=dev, generator=g) * 0.1
    coef = torch.rand(k, 4, device=dev, generator=g) + 0.5
    y = torch.empty(nb, cout, p, device=dev)
    part = torch.empty(1, hip.pw_stat_slots(nb, 1, k, cout, p), cout, 4, device=dev)
    hip.pw_layer_forward(x, w, in_coef=coef, in_relu=True, y=y, stat_part=part)
    rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
    out = torch.empty(cout, 4, device=dev)
    hip.pw_stats_finalize(part, torch.ones(cout, device=dev), torch.zeros(cout, device=dev), rm, rv, 0.1, 1e-5, out)
    torch.cuda.synchronize()
    mean = y.double().mean((0, 2))
    var = y.double().var((0, 2), unbiased=False)
    print(f'{nb}x{p} {k}->{cout}: max |mean - ref| {float((out[:, 2].double() - mean).abs().max()):.2e}  max rel invstd err {float(((out[:, 3].double() - (var + 1e-5).rsqrt()) / (var + 1e-5).rsqrt()).abs().max()):.2e}', flush=True)
