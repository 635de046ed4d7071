"""Per-layer GEMM times (fwd / dgrad / wgrad as autograd runs them) at the model's shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device('cuda:0'); B = 8
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it
layers = [("SA1.0", 64, 4, 131072, 1, False), ("SA1.1", 64, 64, 131072, 1, True), ("SA1.2", 128, 64, 131072, 1, True),
          ("SA2.0", 128, 131, 32768, 1, True), ("SA2.1", 128, 128, 32768, 1, True), ("SA2.2", 256, 128, 32768, 1, True),
          ("SA3.0", 128, 259, 8192, 1, True), ("SA3.1", 128, 128, 8192, 1, True), ("SA3.2", 256, 128, 8192, 1, True),
          ("SA4.0", 128, 259, 4096, 1, True), ("SA4.1", 128, 128, 4096, 1, True), ("SA4.2", 256, 128, 4096, 1, True),
          ("agg.0", 128, 259, 4096, 1, True), ("agg.1", 128, 128, 4096, 1, True), ("agg.2", 128, 128, 4096, 1, True),
          ("mpn16.c0", 256, 259, 8192, 6, False), ("mpn16.c3", 128, 256, 8192, 6, True), ("mpn16.s0l", 256, 128, 8192, 6, True), ("mpn16.s3", 128, 256, 8192, 6, True),
          ("mpn64.c0", 256, 259, 32768, 1, False), ("mpn64.c3", 128, 256, 32768, 1, True), ("mpn64.s0l", 256, 128, 32768, 1, True), ("mpn64.s3", 128, 256, 32768, 1, True)]
tot = [0, 0, 0]; totf = 0
for name, co, ci, p, mult, dgrad in layers:
    w = torch.randn(co, ci, device=dev); x = torch.randn(B, ci, p, device=dev); dy = torch.randn(B, co, p, device=dev)
    we = w.unsqueeze(0).expand(B, -1, -1)
    f = t(lambda: torch.bmm(we, x))
    d = t(lambda: torch.bmm(we.transpose(1, 2), dy)) if dgrad else 0.0
    g = t(lambda: torch.bmm(dy, x.transpose(1, 2)).sum(0))
    fl = 2.0 * B * co * ci * p / 1e9
    print(f"{name:10s} co={co:3d} ci={ci:3d} P={p:6d} x{mult}: fwd {f:.3f} ms ({fl/f:5.0f} TF)  dgrad {d:.3f} ({(fl/d if d else 0):5.0f})  wgrad {g:.3f} ({fl/g:5.0f})")
    tot[0] += f * mult; tot[1] += d * mult; tot[2] += g * mult; totf += fl * mult
print("totals ms: fwd %.2f dgrad %.2f wgrad %.2f ; GFLOP per pass %.0f" % (*tot, totf))
