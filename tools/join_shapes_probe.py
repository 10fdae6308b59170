"""The residual-join input gradients (1x1 + residual + mask bits + fused BatchNorm backward) at the four stage shapes of
ResNet-50 at bs 256, a few launches each: run under  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
and read with tools/pmc_by_grid.py -- where does a wave of the tiled kernel spend its time at each image size?
Also prints event timings (us, TB/s) per shape."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from virtex_amd import ops

dt = torch.bfloat16
B = 256
for (K, N, H) in [(64, 256, 56), (128, 512, 28), (256, 512, 28), (256, 1024, 14), (512, 2048, 7)]:
    M = B * H * H
    dy = torch.randn(M, K, device="cuda").to(dt); wt = (torch.randn(N, K, device="cuda") / K ** 0.5).to(dt)
    x = torch.randn(M, N, device="cuda").to(dt); res = torch.randn(M, N, device="cuda").to(dt)
    bits = torch.randint(0, 256, (M * N // 8,), device="cuda", dtype=torch.uint8)
    mean = torch.zeros(N, device="cuda"); rstd = torch.ones(N, device="cuda")
    bn = ops.BnBwd(x, mean, rstd, ybits=bits)
    for _ in range(3):
        ops.gemm_nt_bnbwd(dy, wt, bn, residual=res)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.gemm_nt_bnbwd(dy, wt, bn, residual=res)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5 * 1e-3
    byts = 2.0 * M * K + 3 * 2.0 * M * N + M * N / 8
    print(f"K={K:4d} N={N:4d} @{H:2d}: M={M:7d} {byts / 1e6:7.0f} MB  {t * 1e6:6.1f} us  {byts / t / 1e12:.2f} TB/s  {2.0 * M * N * K / t / 1e12:6.0f} TF/s", flush=True)
    del dy, x, res, bits
