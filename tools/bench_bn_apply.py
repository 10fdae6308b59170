"""Fused BatchNorm backward apply (read x, dz; write dx) at the large ResNet-50 shapes, B = 256: GB/s per unroll factor."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from virtex_amd import ops, _lib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_layers import timeit
B, dt = 256, torch.bfloat16
for (H, C) in [(112, 64), (56, 256), (56, 64), (28, 512), (28, 128), (14, 1024), (14, 256), (7, 2048)]:
    x = torch.randn(B, H, H, C, device="cuda").to(dt); dz = torch.randn(B, H, H, C, device="cuda").to(dt)
    g = torch.rand(C, device="cuda") + 0.5; mean = torch.zeros(C, device="cuda"); rstd = torch.ones(C, device="cuda")
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    P = B * H * H
    parts = torch.randn(64 * 2 * C, device="cuda")
    st = ops.BnStats(parts, 64, None)
    row = []
    for u in (1, 2, 4):
        _lib.call("vtx_set_bn_apply_unroll", _lib.c_int(u))
        t = timeit(lambda: ops.bn_bwd_fused(x, dz, g, mean, rstd, dg, db, st), iters=20, warm=3)
        row.append(f"unr{u}: {t*1e6:7.1f} us {x.numel()*6/t/1e9:6.0f} GB/s")
    _lib.call("vtx_set_bn_apply_unroll", _lib.c_int(0))
    print(f"bn_bwd_fused {H:3d}x{H:<3d} C={C:<5d} | " + " | ".join(row), flush=True)
