"""Is SA1's rebuilt-first-activation path taken in the model's step?  Counts the k4 entry points in one
eager supervised step (B = 2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from nesie_amd.kernels import HipKernels
from nesie_amd.mmdet3d_ops import fused_mlp

calls = {}
for name in ('pw_layer_forward_k4', 'pw_dgrad_bn_reduce_k4', 'pw_wgrad_bn_backward_k4', 'k4_first_layer_wgrad', 'mlp_stream_forward'):
    orig = getattr(HipKernels, name)
    def wrap(self, *a, _o=orig, _n=name, **k):
        calls[_n] = calls.get(_n, 0) + 1
        return _o(self, *a, **k)
    setattr(HipKernels, name, wrap)
orig_fwd = fused_mlp.SAStackFn.forward
def fwd(ctx, x, bufs, fixed_lead, *params):
    out = orig_fwd(ctx, x, bufs, fixed_lead, *params)
    print('SAStackFn', tuple(x.shape), 'needs', ctx.needs_input_grad[:9], 'k4', ctx.k4, [tuple(p.shape) for p in params[::3]])
    return out
fused_mlp.SAStackFn.forward = staticmethod(fwd)
dev = torch.device('cuda:0')
model, step, bucket = bench.build_step(dev, 2, 1000, 1e-3, 0.01, graph=False)
step()
torch.cuda.synchronize()
print(calls)
