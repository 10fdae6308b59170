"""What a CU-masked stream (virtex_amd.streams.masked_stream: hipExtStreamCreateWithCUMask) gets on MI355X: an MFMA-bound GEMM and an
HBM-bound copy on streams restricted to the first n compute units, n = 256 (plain stream) ... 16.  Says (1) whether the mask is
honoured, (2) how many CUs saturate HBM -- the number that decides whether partitioning the chip between the compute stream and the
weight-gradient stream can pay."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from virtex_amd import streams  # noqa: E402

dev = torch.device("cuda", 0)
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
x = torch.empty(1 << 29, device=dev, dtype=torch.uint8)
y = torch.empty_like(x)


def timed(st, fn, n=10):
    with torch.cuda.stream(st):
        fn(); fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


print(f"{'CUs':>10s} {'GEMM TFLOP/s':>14s} {'copy GB/s (r+w)':>16s}")
for lo, hi in ((0, 256), (0, 192), (0, 128), (0, 96), (0, 64), (0, 32), (0, 16), (64, 256), (128, 256)):
    st = torch.cuda.Stream(device=dev) if (lo, hi) == (0, 256) else streams.masked_stream(dev, lo, hi)
    tg = timed(st, lambda: torch.mm(a, b))
    tc = timed(st, lambda: y.copy_(x))
    print(f"[{lo:3d},{hi:3d}) {2 * 8192 ** 3 / tg / 1e12:14.1f} {2 * x.numel() / tc / 1e9:16.1f}", flush=True)
