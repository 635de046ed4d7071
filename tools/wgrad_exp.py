"""Experiment: split-K weight gradient of a 1x1 conv through batched rocBLAS GEMMs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device('cuda:0')

def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

def wgrad_plain(dy, x):      # what autograd does for bmm(expand(W), x)
    return torch.bmm(dy, x.transpose(1, 2)).sum(0)

def wgrad_split(dy, x, S):
    B, Co, P = dy.shape
    Ci = x.shape[1]
    Pc = P // S
    parts = []
    for b in range(B):
        a = dy[b].view(Co, S, Pc).permute(1, 0, 2)          # (S, Co, Pc) strides (Pc, P, 1)
        c = x[b].view(Ci, S, Pc).permute(1, 2, 0)            # (S, Pc, Ci) strides (Pc, 1, P)
        parts.append(torch.bmm(a, c))
    return torch.stack(parts).sum((0, 1))

def wgrad_flat(dy, x):       # one big GEMM after making K contiguous: (Co, B*P) @ (B*P, Ci)
    B, Co, P = dy.shape
    return dy.permute(1, 0, 2).reshape(Co, B * P) @ x.permute(1, 0, 2).reshape(x.shape[1], B * P).t()

shapes = [(8, 256, 259, 512 * 16), (8, 128, 256, 512 * 16), (8, 256, 256, 512 * 16), (8, 128, 256, 512 * 64),
          (8, 256, 259, 512 * 64), (8, 128, 131, 1024 * 32), (8, 64, 64, 2048 * 64), (8, 128, 64, 2048 * 64)]
for (B, Co, Ci, P) in shapes:
    dy = torch.randn(B, Co, P, device=dev); x = torch.randn(B, Ci, P, device=dev)
    ref = wgrad_plain(dy, x)
    t0 = timeit(lambda: wgrad_plain(dy, x))
    res = []
    for S in (4, 16, 64):
        out = wgrad_split(dy, x, S)
        err = (out - ref).abs().max().item() / ref.abs().max().item()
        res.append((S, timeit(lambda: wgrad_split(dy, x, S)), err))
    tf = timeit(lambda: wgrad_flat(dy, x))
    fl = 2.0 * B * Co * Ci * P
    print(f"Co={Co} Ci={Ci} P={P}: plain {t0:.3f} ms ({fl/t0/1e9:.0f} TF)  " +
          "  ".join(f"S={S}: {t:.3f} ms ({fl/t/1e9:.0f} TF, err {e:.1e})" for S, t, e in res) + f"  flat+copy {tf:.3f} ms")
