"""Per-op timing of the libnesie_hip.so kernels at the Nesie-VoteNet ScanNet shapes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nesie_amd import mmdet3d_ops as ops
from tests import _cases

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 8))


def timeit(name, fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(f"{name:50s} {ms:9.3f} ms", flush=True)
    return ms


xyz = _cases.cloud(1, B, 40000, dup_frac=0.1).to(dev)
tot = 0
layers = [(40000, 2048, .2, 64), (2048, 1024, .4, 32), (1024, 512, .8, 16), (512, 256, 1.2, 16)]
cur = xyz
for (n, m, r, ns) in layers:
    tot += timeit(f"fps {n}->{m}", lambda: ops.furthest_point_sample(cur, m))
    idx = ops.furthest_point_sample(cur, m)
    new = ops.gather_points(cur.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
    tot += timeit(f"ball_query n={n} m={m} r={r} ns={ns}", lambda: ops.ball_query(0.0, r, ns, cur, new))
    bq = ops.ball_query(0.0, r, ns, cur, new)
    C = {40000: 1, 2048: 128, 1024: 256, 512: 256}[n]
    f = torch.randn(B, C, n, device=dev)
    tot += timeit(f"group C={C}", lambda: ops.grouping_operation(f, bq))
    g = torch.randn(B, C, m, ns, device=dev)
    f.requires_grad_(True)
    out = ops.grouping_operation(f, bq)
    tot += timeit(f"group bwd C={C}", lambda: torch.autograd.grad(out, f, g, retain_graph=True))
    cur = new
seeds = _cases.cloud(2, B, 1024).to(dev)
for nq in (49152, 32768):
    q = _cases.cloud(3, B, nq).to(dev)
    tot += timeit(f"three_nn n={nq} m=1024", lambda: ops.three_nn(q, seeds))
print("sum ms", tot)
