"""The flat BatchNorm apply kernels at the ResNet-50 shapes, B = 256: fused backward apply (read x, dz; write dx), forward apply
(read x; write y) and the join pass (read x, identity; write y + mask bits) -- strided form (rounds 1-2) against the adjacent
form (round 3) per unroll factor and grid cap.  us / GB/s of algorithmic traffic."""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from virtex_amd import ops, _lib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_layers import timeit
B, dt = 256, torch.bfloat16


def sw(name, v):
    _lib.call("vtx_set_switch", name.encode(), ctypes.c_int(v))


VARIANTS = [("strided", 0, 0, 8192), ("adj u1", 1, 1, 8192), ("adj u2", 1, 2, 8192), ("adj u4", 1, 4, 8192), ("adj auto", 1, 0, 8192),
            ("adj u4 g4096", 1, 4, 4096), ("adj u4 g16384", 1, 4, 16384), ("adj u2 g16384", 1, 2, 16384)]
for (H, C) in [(112, 64), (56, 256), (56, 64), (28, 512), (28, 128), (14, 1024), (14, 256), (7, 2048), (7, 512)]:
    x = torch.randn(B, H, H, C, device="cuda").to(dt); dz = torch.randn(B, H, H, C, device="cuda").to(dt)
    g = torch.rand(C, device="cuda") + 0.5; beta = torch.zeros(C, device="cuda")
    mean = torch.zeros(C, device="cuda"); rstd = torch.ones(C, device="cuda")
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    parts = torch.randn(64 * 2 * C, device="cuda")
    st = ops.BnStats(parts, 64, None)
    fst = ops.BnStats(torch.rand(64 * 2 * C, device="cuda"), 64, torch.zeros(C, device="cuda"))
    rows = {"bwd fused": [], "fwd apply": [], "fwd join": []}
    for name, adj, unr, grid in VARIANTS:
        sw("bn_adj", adj); sw("bn_grid", grid); _lib.call("vtx_set_bn_apply_unroll", _lib.c_int(unr))
        t = timeit(lambda: ops.bn_bwd_fused(x, dz, g, mean, rstd, dg, db, st), iters=12, warm=3)
        rows["bwd fused"].append(f"{name}: {t*1e6:6.1f}us {x.numel()*6/t/1e9:5.0f}")
        t = timeit(lambda: ops.bn_fwd(x, g, beta, rm, rv, None, relu=True, stats=fst), iters=12, warm=3)
        rows["fwd apply"].append(f"{name}: {t*1e6:6.1f}us {x.numel()*4/t/1e9:5.0f}")
        t = timeit(lambda: ops.bn_fwd(x, g, beta, rm, rv, None, relu=True, residual=dz, stats=fst, want_bits=True), iters=12, warm=3)
        rows["fwd join"].append(f"{name}: {t*1e6:6.1f}us {x.numel()*6.125/t/1e9:5.0f}")
    for k, r in rows.items():
        print(f"{k} {H:3d}x{H:<3d} C={C:<5d}| " + " | ".join(r), flush=True)
sw("bn_adj", 1); sw("bn_grid", 8192); _lib.call("vtx_set_bn_apply_unroll", _lib.c_int(0))
