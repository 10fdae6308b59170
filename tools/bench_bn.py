"""BatchNorm layer microbenchmark: every distinct (pixels, channels) shape of ResNet-50 at B=256, forward
(stats + finalize + apply) and backward (reduce + finalize + apply), with the achieved HBM bandwidth on the
algorithmic bytes (fwd 6 B/elem [+2 residual], bwd 10 B/elem [+2 mask tensor, +2 dz])."""
import sys
import torch
sys.path.insert(0, ".")
from virtex_amd import ops
sys.path.insert(0, "tools")
from bench_layers import timeit

B, dt = 256, torch.bfloat16
# (H, C, count, kind)  kind: "relu" (bn1/bn2), "res" (bn3: residual + ReLU, mask = y, dz out), "plain" (downsample)
SHAPES = [(112, 64, 1, "relu"), (56, 64, 6, "relu"), (56, 256, 3, "res"), (56, 256, 1, "plain"), (56, 128, 1, "relu"),
          (28, 128, 7, "relu"), (28, 512, 4, "res"), (28, 512, 1, "plain"), (28, 256, 1, "relu"), (14, 256, 11, "relu"),
          (14, 1024, 6, "res"), (14, 1024, 1, "plain"), (14, 512, 1, "relu"), (7, 512, 5, "relu"), (7, 2048, 3, "res"),
          (7, 2048, 1, "plain")]
tf = tb = 0.0
bf = bb = 0.0
print(f"{'shape':22s} {'cnt':>3s} | {'fwd us':>8s} {'GB/s':>6s} | {'bwd us':>8s} {'GB/s':>6s}")
for (H, C, cnt, kind) in SHAPES:
    x = torch.randn(B, H, H, C, device="cuda").to(dt)
    r = torch.randn(B, H, H, C, device="cuda").to(dt) if kind == "res" else None
    dy = torch.randn(B, H, H, C, device="cuda").to(dt)
    g = torch.rand(C, device="cuda") + 0.5; b = torch.randn(C, device="cuda") * 0.1
    rm = torch.zeros(C, device="cuda"); rv = torch.ones(C, device="cuda"); nbt = torch.zeros((), dtype=torch.long, device="cuda")
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    y, mean, rstd = ops.bn_fwd(x, g, b, rm, rv, nbt, relu=kind != "plain", residual=r)
    t_f = timeit(lambda: ops.bn_fwd(x, g, b, rm, rv, nbt, relu=kind != "plain", residual=r), iters=10, warm=2)
    if kind == "res":
        fn = lambda: ops.bn_bwd(x, dy, y, g, mean, rstd, dg, db, want_dz=True)
    elif kind == "relu":
        fn = lambda: ops.bn_bwd(x, dy, None, g, mean, rstd, dg, db, relu_beta=b)
    else:
        fn = lambda: ops.bn_bwd(x, dy, None, g, mean, rstd, dg, db)
    t_b = timeit(fn, iters=10, warm=2)
    n = x.numel()
    by_f = n * (6 + (2 if kind == "res" else 0))
    by_b = n * (10 + (6 if kind == "res" else 0))      # res: mask tensor read twice, dz written once
    print(f"bn {H:3d}x{H:<3d} C={C:<5d}{kind:6s} {cnt:3d} | {t_f*1e6:8.1f} {by_f/t_f/1e9:6.0f} | {t_b*1e6:8.1f} {by_b/t_b/1e9:6.0f}", flush=True)
    tf += cnt * t_f; tb += cnt * t_b; bf += cnt * by_f; bb += cnt * by_b
print(f"BatchNorm totals per step: fwd {tf*1e3:.2f} ms ({bf/tf/1e9:.0f} GB/s), bwd {tb*1e3:.2f} ms ({bb/tb/1e9:.0f} GB/s)")
