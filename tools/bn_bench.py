"""BatchNorm(+ReLU) forward / backward passes at the step's big shapes, timed per kernel with
HIP events around single-kernel regions.  usage: python tools/bn_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nesie_amd.kernels import backend_for

dev = torch.device('cuda:0')


def timed(fn, it=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


def case(name, shape, row_bias):
    B, C = shape[:2]
    x = torch.randn(*shape, device=dev)
    rb = torch.randn(B, C, shape[2], device=dev) if row_bias else None
    g, b_ = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    y, sm, si, coef = torch.empty_like(x), x.new_empty(C), x.new_empty(C), x.new_empty(C, 4)
    hip = backend_for(x)
    fwd = timed(lambda: hip.bn_relu_forward(x, g, b_, rm, rv, 0.1, 1e-5, True, y, sm, si, coef, row_bias=rb))
    dy, dx = torch.randn_like(x), torch.empty_like(x)
    dg, db = x.new_empty(C), x.new_empty(C)
    drb = torch.empty_like(rb) if row_bias else None
    bwd = timed(lambda: hip.bn_relu_backward(dy, x, y, g, b_, sm, si, coef, True, dx, dg, db,
                                             row_bias=rb, d_row_bias=drb))
    mb = x.numel() * 4 / 1e6
    print(f'{name:34s} {mb:6.0f} MB/tensor  fwd (3 passes) {fwd * 1e3:7.1f} us = {3 * mb / fwd / 1e3:5.2f} TB/s   '
          f'bwd (5 passes) {bwd * 1e3:7.1f} us = {5 * mb / bwd / 1e3:5.2f} TB/s')


case('SA1 layer (8,64,2048,64)', (8, 64, 2048, 64), False)
case('SA1 last (8,128,2048,64)', (8, 128, 2048, 64), False)
case('MiniPointNet (8,1536,512,16)', (8, 1536, 512, 16), False)
case('MiniPointNet + row_bias', (8, 1536, 512, 16), True)
case('bbox net (8,256,512,64)', (8, 256, 512, 64), False)
case('bbox net + row_bias', (8, 256, 512, 64), True)
case('SA2 layer (8,128,1024,32)', (8, 128, 1024, 32), False)
