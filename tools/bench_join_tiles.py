"""The residual-join input gradients (1x1, K = block width / 4, N = block width, fused BatchNorm backward with the ReLU mask as
bits and the identity gradient as residual) of stages 1-3 at bs = 256 under the tile candidates of the generation-2 kernel --
is the A operand (dy, read once per COLUMN tile) worth a 256-wide tile?  PMC says these launches fetch 20-30 % more than their
algorithmic bytes (profiles/r04_pmc_join_variants.txt)."""
import ctypes, sys, torch
sys.path.insert(0, ".")
from virtex_amd import ops, _lib

B, dt = 256, torch.bfloat16
lib = _lib.lib()
names = {-1: "auto", 1: "256x128", 2: "128x128", 3: "128x64", 4: "64x128", 5: "64x64", 6: "128x128w8"}


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for (K, N, H) in [(64, 256, 56), (128, 512, 28), (256, 1024, 14), (512, 2048, 7)]:
    M = B * H * H
    dy = torch.randn(M, K, device="cuda").to(dt); wt = (torch.randn(N, K, device="cuda") / K ** 0.5).to(dt)
    x = torch.randn(M, N, device="cuda").to(dt); res = torch.randn(M, N, device="cuda").to(dt)
    bits = torch.randint(0, 256, (M * N // 8,), device="cuda", dtype=torch.uint8)
    mean = torch.zeros(N, device="cuda"); rstd = torch.ones(N, device="cuda")
    bn = ops.BnBwd(x, mean, rstd, ybits=bits)
    byts = 2.0 * M * K + 3 * 2.0 * M * N + M * N / 8
    ref = None
    out = []
    for c in names:
        lib.vtx_set_tile_override(ctypes.c_int(c))
        o, st = ops.gemm_nt_bnbwd(dy, wt, bn, residual=res)
        gen = lib.vtx_last_contraction_generation()
        if ref is None:
            ref = o.float()
        err = (o.float() - ref).abs().max().item()
        t = timeit(lambda: ops.gemm_nt_bnbwd(dy, wt, bn, residual=res))
        out.append(f"{names[c]}(g{gen})={t * 1e6:.0f}us/{byts / t / 1e12:.2f}TB/s" + ("" if err == 0 else f" !diff {err:.2e}"))
    lib.vtx_set_tile_override(ctypes.c_int(-1))
    print(f"K={K:4d} N={N:4d} @{H:2d}  {byts / 1e6:7.0f} MB | " + "  ".join(out), flush=True)
