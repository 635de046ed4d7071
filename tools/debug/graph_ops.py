"""Capture single back-end calls in a hipGraph, one child process per case (a runtime segfault in
hipStreamEndCapture kills the process): which call cannot be captured?"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CASES = ['dgrad_220', 'dgrad_259', 'dgrad_256', 'fwd_220_bias', 'fwd_259_bias', 'sum', 'wgrad_220', 'apply']


def child(which):
    import torch
    from nesie_amd import kernels
    dev = torch.device('cuda:0')
    hip = kernels.backend_for(torch.zeros(1, device=dev))
    B, P = 2, 256

    def op():
        if which.startswith('dgrad_'):
            k = int(which.split('_')[1])
            dz = torch.randn(B, k, P, device=dev)
            w = torch.randn(1, k, 128, device=dev)
            z = torch.randn(B, 128, P, device=dev)
            coef = torch.rand(128, 4, device=dev)
            da = torch.empty(B, 128, P, device=dev)
            return hip.pw_dgrad_bn_reduce(dz, w.transpose(1, 2), z, coef, da)
        if which.startswith('fwd_'):
            co = int(which.split('_')[1])
            x = torch.randn(B, 128, P, device=dev)
            w = torch.randn(1, co, 128, device=dev)
            y = torch.empty(B, co, P, device=dev)
            coef = torch.rand(128, 4, device=dev)
            hip.pw_layer_forward(x, w, in_coef=coef, y=y, bias=torch.randn(co, device=dev))
            return y
        if which == 'sum':
            return torch.randn(B, 220, P, device=dev).sum((0, 2))
        if which == 'wgrad_220':
            dw = torch.empty(1, 220, 128, device=dev)
            hip.pw_wgrad(torch.randn(B, 220, P, device=dev), torch.randn(B, 128, P, device=dev), dw,
                         ng=1, x_coef=torch.rand(128, 4, device=dev), x_relu=True)
            return dw
        if which == 'apply':
            c = 128
            da, z = torch.randn(B, c, P, device=dev), torch.randn(B, c, P, device=dev)
            part = torch.randn(c, 8, 2, device=dev)
            dz = torch.empty_like(da)
            dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
            hip.bn_relu_backward_apply(da, z, torch.rand(c, device=dev), None, torch.rand(c, 4, device=dev), part, dz, dg, db)
            return dz
    for _ in range(2):
        op()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = op()
    g.replay()
    torch.cuda.synchronize()
    print(which, 'ok')


if __name__ == '__main__':
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for w in CASES:
            r = subprocess.run([sys.executable, __file__, w], capture_output=True, text=True, timeout=300)
            print(w, 'rc', r.returncode, r.stdout.strip()[-100:])
