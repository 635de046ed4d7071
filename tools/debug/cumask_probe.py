"""Probe: does a HIP stream created with a CU mask (hipExtStreamCreateWithCUMask) confine (a) eager
kernels and (b) a hipGraph captured on and replayed into it?  Times a chip-filling elementwise
kernel on the default stream, on a stream masked to 32 of 256 CUs, and the same inside a graph."""
import ctypes
import time

import torch

hip = ctypes.CDLL('libamdhip64.so')


def masked_stream(n_cus, total=256):
    words = (total + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    # CUs are numbered round-robin over the 8 XCDs: take the first n_cus / 8 of every XCD
    for cu in range(total):
        if (cu // 8) < n_cus // 8:
            mask[cu // 32] |= 1 << (cu % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), words, mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def timed(fn, stream, n=20):
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(n):
            fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def main():
    dev = torch.device('cuda:0')
    x = torch.randn(64 << 20, device=dev)
    y = torch.empty_like(x)

    def work():
        torch.sin(x, out=y)          # ALU + bandwidth over 256 MB

    d = torch.cuda.current_stream()
    print(f'default stream        : {timed(work, d):.3f} ms')
    for n in (32, 64):
        ms = masked_stream(n)
        print(f'{n:3d}-CU masked stream   : {timed(work, ms):.3f} ms (eager)')
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=ms):
            work()
        print(f'{n:3d}-CU masked, graphed  : {timed(g.replay, ms):.3f} ms (graph captured on and replayed into it)')
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=side):
        work()
    print(f'plain stream, graphed  : {timed(g.replay, side):.3f} ms')


if __name__ == '__main__':
    main()
