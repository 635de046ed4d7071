"""nesie_conv_wgrad vs the torch/rocBLAS weight gradient on the step's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nesie_amd import kernels
from nesie_amd.mmdet3d_ops.pointnet_modules import _PointwiseConvFn

dev = torch.device('cuda:0')
hip = kernels.backend_for(torch.empty(1, device=dev))


def timeit(f, n=10):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n):
            f()
    gr.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); gr.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


B = 8
shapes = [(64, 4, 131072), (64, 64, 131072), (128, 64, 131072), (128, 131, 32768),
          (128, 128, 32768), (256, 128, 32768), (128, 128, 8192), (256, 128, 8192),
          (128, 64, 4096)]
print('%-20s %9s %9s %8s %8s %9s' % ('cout,cin,P', 'torch ms', 'mfma ms', 'TF', 'GB/s', 'rel err'))
for cout, cin, p in shapes:
    g = torch.Generator(device=dev).manual_seed(cout + cin)
    dy = torch.randn(B, cout, p, device=dev, generator=g)
    x = torch.randn(B, cin, p, device=dev, generator=g)
    w2 = torch.randn(cout, cin, device=dev, generator=g)
    dw = torch.empty(cout, cin, device=dev)

    class Ctx:
        saved_tensors = (x, w2)
        needs_input_grad = (False, True)
    t_ref = timeit(lambda: _PointwiseConvFn.backward(Ctx, dy))
    t_new = timeit(lambda: hip.conv_wgrad(dy, x, dw))
    ref = torch.einsum('bmp,bkp->mk', dy.double(), x.double())
    err = ((dw.double() - ref).norm() / ref.norm()).item()
    fl = 2.0 * B * cout * cin * p
    by = 4.0 * B * p * (cin + cout)
    print('%-20s %9.4f %9.4f %8.1f %8.0f %9.1e' % (f'{cout},{cin},{p}', t_ref, t_new,
          fl / t_new / 1e9, by / t_new / 1e6, err), flush=True)
