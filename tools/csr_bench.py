"""The inverted-index scatter (QueryAndGroup's feature gradient) at the step's shapes: rows in LDS
(group_bwd_csr_rows_kernel) vs gathered from HBM (group_bwd_csr_kernel, NESIE_CSR_ROWS=0), on
ball-query indices of a synthetic scene (low-index points sit in many balls).
usage: python tools/csr_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nesie_amd import kernels
from nesie_amd.mmdet3d_ops import ball_query, furthest_point_sample
from nesie_amd.scenes import make_batch

dev = torch.device('cuda:0')
hip = kernels.backend_for(torch.empty(1, device=dev))
pts, _, _ = make_batch(7, 8, 40000)
xyz = pts[..., :3].contiguous().to(dev)


def level(xyz, m):
    idx = furthest_point_sample(xyz, m)
    return torch.gather(xyz, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()


l1 = level(xyz, 2048)
l2 = level(l1, 1024)
l3 = level(l2, 512)
l4 = level(l3, 256)
cases = [('SA2', l1, l2, 0.4, 32, 128), ('SA3', l2, l3, 0.8, 16, 256), ('SA4', l3, l4, 1.2, 16, 256)]
for name, src, ctr, r, ns, c in cases:
    n, m = src.shape[1], ctr.shape[1]
    idx = ball_query(0.0, r, ns, src, ctr)
    order, sources = hip.inverted_index(idx, n)
    runs = torch.bincount(idx[0].flatten().long(), minlength=n)
    go = torch.randn(8, 3 + c, m, ns, device=dev)
    res = {}
    for mode in ('0', '1'):
        os.environ['NESIE_CSR_ROWS'] = mode
        gf = torch.zeros(8, c, n, device=dev)
        for _ in range(3):
            hip.query_and_group_backward_csr(go, idx.shape, order, sources, gf)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            hip.query_and_group_backward_csr(go, idx.shape, order, sources, gf)
        e.record()
        torch.cuda.synchronize()
        res[mode] = s.elapsed_time(e) / 20 * 1e3
    del os.environ['NESIE_CSR_ROWS']
    print(f'{name}: n {n} m {m} ns {ns} c {c}  longest run {int(runs.max())}, runs > 64: {int((runs > 64).sum())}  '
          f'HBM gather {res["0"]:.1f} us, LDS rows {res["1"]:.1f} us  ({go.numel() * 4 / 1e6:.0f} MB)', flush=True)
