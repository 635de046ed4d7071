"""Review item 10 (measure-only): the error of a split-precision product -- bf16 x 3 (hi.hi + hi.lo +
lo.hi, fp32 accumulation), the form a bf16-MFMA variant of the layer kernel would compute -- against
float64, beside plain fp32 accumulation, on the shapes of tests/test_pwconv_gpu.py's
test_layer_forward_matches_fp64 (128 -> 128 and 128 -> 256 over 512 positions; unit-variance operands).
CPU emulation: it prices the ARITHMETIC (what the matrix cores would be fed), not a kernel."""
import torch

torch.manual_seed(0)


def split(t):
    hi = t.to(torch.bfloat16).to(torch.float32)
    lo = (t - hi).to(torch.bfloat16).to(torch.float32)
    return hi, lo


for k, co, p in ((128, 128, 512), (128, 256, 512), (256, 256, 512)):
    w = torch.randn(co, k) / k ** 0.5
    x = torch.randn(k, p)
    ref = w.double() @ x.double()
    scale = ref.abs().max().item()
    fp32 = (w @ x).double()
    wh, wl = split(w)
    xh, xl = split(x)
    b3 = ((wh @ xh) + (wh @ xl) + (wl @ xh)).double()
    wl2 = (w - wh - wl)
    xl2 = (x - xh - xl)
    b6 = ((wh @ xh) + (wh @ xl) + (wl @ xh) + (wl @ xl) + (wh @ xl2.to(torch.bfloat16).float()) + (wl2.to(torch.bfloat16).float() @ xh)).double()
    e = lambda y: ((y - ref).abs().max().item() / scale, ((y - ref).norm() / ref.norm()).item())
    print(f'{k:4d} -> {co:4d} x {p}: fp32 max {e(fp32)[0]:.2e} rel-L2 {e(fp32)[1]:.2e} | bf16x3 max {e(b3)[0]:.2e} rel-L2 {e(b3)[1]:.2e} '
          f'({e(b3)[1] / e(fp32)[1]:.0f} x) | bf16x6 max {e(b6)[0]:.2e} rel-L2 {e(b6)[1]:.2e} ({e(b6)[1] / e(fp32)[1]:.1f} x)')
