"""Weight gradient of a small 1x1 conv, dW = sum_b dy[b] x[b]^T: folded single GEMM (two
transposing copies) vs B batched GEMMs + sum, per layer shape of the module-by-module layers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device('cuda:0')
B = 8
def t(fn, n=20):
    """us per call inside a replayed hipGraph of n calls (what the training step sees)."""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3
for co, ci, p in ((256, 512, 512), (256, 256, 512), (256, 512, 1024), (256, 256, 1024), (256, 256, 1024),
                  (259, 256, 1024), (128, 128, 256), (128, 256, 256), (256, 131, 1024), (768, 259, 1024)):
    dy = torch.randn(B, co, p, device=dev); x = torch.randn(B, ci, p, device=dev)
    fold = lambda: torch.mm(dy.transpose(0, 1).reshape(co, B * p), x.transpose(0, 1).reshape(ci, B * p).t())
    bmm = lambda: torch.bmm(dy, x.transpose(1, 2)).sum(0)
    for s in (2, 4):
        pass
    def split(S):
        # K split inside every batch as well: (B*S) GEMMs with K = p / S
        return lambda: torch.bmm(dy.view(B, co, S, p // S).permute(0, 2, 1, 3).reshape(B * S, co, p // S),
                                 x.view(B, ci, S, p // S).permute(0, 2, 3, 1).reshape(B * S, p // S, ci)).sum(0)
    print(f'{co}x{ci} P={p}: fold {t(fold):6.1f} us   bmm+sum {t(bmm):6.1f} us')
