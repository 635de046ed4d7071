"""nesie_mlp_layer_forward vs torch.bmm (+ separate BN kernels) on the step's layer shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nesie_amd import _lib

dev = torch.device('cuda:0')
lib = _lib.load()
S = lambda: torch.cuda.current_stream().cuda_stream


def fwd(x, w, coef, relu, y, part):
    b, cin, p = x.shape
    cout = w.shape[0]
    _lib.call('nesie_mlp_layer_forward', b, cin, cout, p, x.data_ptr(), cin * p, w.data_ptr(),
              coef.data_ptr() if coef is not None else 0, int(relu), y.data_ptr(),
              part.data_ptr() if part is not None else 0, S())


def fwd_stream(x, w, coef, relu, y, part):
    b, cin, p = x.shape
    cout = w.shape[0]
    _lib.call('nesie_mlp_layer_forward_stream', b, cin, cout, p, x.data_ptr(), cin * p,
              w.data_ptr(), coef.data_ptr() if coef is not None else 0, int(relu), y.data_ptr(),
              part.data_ptr() if part is not None else 0, S())


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


shapes = [(64, 4, 131072), (64, 64, 131072), (128, 64, 131072), (128, 131, 32768),
          (128, 128, 32768), (256, 128, 32768), (128, 259, 8192), (256, 259, 8192),
          (128, 256, 8192), (256, 128, 8192), (256, 259, 32768), (128, 256, 32768),
          (256, 256, 1024), (128, 128, 4096), (18, 128, 512)]
B = 8
print('%-22s %9s %9s %9s %9s %8s %8s' % ('cout,cin,P', 'bmm ms', 'mfma ms', 'fused ms', 'TF(mfma)',
                                          'GB/s', 'maxerr'))
for cout, cin, p in shapes:
    g = torch.Generator(device=dev).manual_seed(cout * 7 + cin)
    x = torch.randn(B, cin, p, device=dev, generator=g)
    w = torch.randn(cout, cin, device=dev, generator=g) / cin ** 0.5
    coef = torch.rand(cin, 4, device=dev, generator=g) + 0.5
    coef[:, 1] -= 1.0
    y = torch.empty(B, cout, p, device=dev)
    nparts = lib.nesie_mlp_stat_partials(B, cout, p)
    part = torch.empty(nparts, cout, 2, device=dev)
    we = w.unsqueeze(0).expand(B, -1, -1)
    t_bmm = timeit(lambda: torch.bmm(we, x))
    t_plain = timeit(lambda: fwd(x, w, None, 0, y, None))
    ref = torch.bmm(we.double(), x.double())
    err = (y.double() - ref).abs().max().item()
    t_fused = timeit(lambda: fwd(x, w, coef, 1, y, part))
    a = torch.relu(x.double() * coef[:, 0].double().view(1, -1, 1) + coef[:, 1].double().view(1, -1, 1))
    ref2 = torch.bmm(we.double(), a)
    err2 = (y.double() - ref2).abs().max().item()
    s_err = (part[..., 0].double().sum(0) - ref2.sum((0, 2))).abs().max().item() / max(1.0, ref2.sum((0, 2)).abs().max().item())
    q_err = ((part[..., 1].double().sum(0) - (ref2 ** 2).sum((0, 2))).abs() / (ref2 ** 2).sum((0, 2))).max().item()
    if cin <= 64 and cout <= 64:
        np2 = lib.nesie_mlp_stream_partials(B, p)
        part2 = torch.empty(np2, cout, 2, device=dev)
        t_s = timeit(lambda: fwd_stream(x, w, coef, 1, y, part2))
        e3 = (y.double() - ref2).abs().max().item()
        s3 = (part2[..., 0].double().sum(0) - ref2.sum((0, 2))).abs().max().item() / max(1.0, ref2.sum((0, 2)).abs().max().item())
        q3 = ((part2[..., 1].double().sum(0) - (ref2 ** 2).sum((0, 2))).abs() / (ref2 ** 2).sum((0, 2))).max().item()
        print('   stream fused: %.4f ms  %.0f GB/s  err %.1e sum %.1e sq %.1e' % (
            t_s, 4.0 * B * p * (cin + cout) / t_s / 1e6, e3, s3, q3), flush=True)
    fl = 2.0 * B * cout * cin * p
    by = 4.0 * B * p * (cin + cout)
    print('%-22s %9.4f %9.4f %9.4f %9.1f %8.0f %8.1e  fused err %.1e sum %.1e sq %.1e' % (
        f'{cout},{cin},{p}', t_bmm, t_plain, t_fused, fl / t_plain / 1e9, by / t_fused / 1e6, err,
        err2, s_err, q_err), flush=True)
