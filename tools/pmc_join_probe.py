"""Which operand of the fused-BatchNorm-backward input gradient is over-fetched?  The stage-1 conv1 input gradient
(M = 256*56*56 rows, K = 64, N = 256) in five variants, told apart in a PMC pass by their grids (M shrinks by one row tile per
variant): run under  rocprofv3 --pmc FETCH_SIZE  and read with tools/pmc_by_grid.py."""
import sys
import torch
sys.path.insert(0, ".")
from virtex_amd import ops

dt = torch.bfloat16
K, N = 64, 256
M0 = 256 * 56 * 56
torch.manual_seed(0)
dy = torch.randn(M0, K, device="cuda").to(dt)
wt = (torch.randn(N, K, device="cuda") / 8).to(dt)
x = torch.randn(M0, N, device="cuda").to(dt)
res = torch.randn(M0, N, device="cuda").to(dt)
ymask = torch.randn(M0, N, device="cuda").to(dt)
bits = torch.randint(0, 256, (M0 * N // 8,), device="cuda", dtype=torch.uint8)
mean = torch.zeros(N, device="cuda"); rstd = torch.ones(N, device="cuda")
gamma = torch.ones(N, device="cuda"); beta = torch.zeros(N, device="cuda")
variants = [("bits + residual (the step)", dict(ybits=True), True),
            ("bits, no residual", dict(ybits=True), False),
            ("mask recomputed from x, no residual", dict(gamma=gamma, beta=beta), False),
            ("no ReLU, no residual", dict(), False),
            ("bf16 mask tensor + residual", dict(ymask=True), True)]
for i, (label, kw, use_res) in enumerate(variants):
    M = M0 - 256 * i
    kw = dict(kw)
    if kw.pop("ybits", None):
        kw["ybits"] = bits[: M * N // 8]
    if kw.pop("ymask", None):
        kw["ymask"] = ymask[:M]
    bn = ops.BnBwd(x[:M], mean, rstd, **kw)
    for _ in range(3):
        out, stats = ops.gemm_nt_bnbwd(dy[:M], wt, bn, residual=res[:M] if use_res else None)
    torch.cuda.synchronize()
    alg = 2 * M * K + 2 * M * N + (2 * M * N if use_res else 0) + (M * N // 8 if "ybits" in kw else 0) + (2 * M * N if "ymask" in kw else 0)
    print(f"grid {(M // 256) * (N // 128):5d}: {label:40s} algorithmic reads {alg / 1e6:7.1f} MB, writes {2 * M * N / 1e6:6.1f} MB", flush=True)
