"""The HBM-bound 1x1 convolutions of stages 1-2 at bs=256 (tensors of 100-400 MB: larger than the Infinity Cache, so a
loop over one layer is NOT cache-fed) with the epilogues the step runs: forward with BatchNorm statistics, input
gradient with the fused BatchNorm backward (mask recomputed from x: conv3/conv2 side; mask = block output + residual
join: conv1 side).  Per block-tile candidate: time, algorithmic GB/s."""
import ctypes, sys, torch
sys.path.insert(0, ".")
from virtex_amd import ops, _lib

B, dt = 256, torch.bfloat16
lib = _lib.lib()
names = {-1: "auto", 1: "256x128", 2: "128x128", 3: "128x64", 4: "64x128", 6: "128x128w8"}
CANDS = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [-1, 1, 2, 6]


def timeit(fn, iters=8, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def sweep(label, fn, N, byts):
    out = []
    for c in CANDS:
        if c in (1, 2, 4, 6) and N <= 64: continue
        lib.vtx_set_tile_override(ctypes.c_int(c))
        t = timeit(fn)
        out.append(f"{names[c]}={t*1e6:.0f}us/{byts/t/1e12:.2f}TB/s")
    lib.vtx_set_tile_override(ctypes.c_int(-1))
    print(f"{label:40s} {byts/1e6:7.0f} MB | " + "  ".join(out), flush=True)


for (C, KO, H) in [(64, 256, 56), (256, 64, 56), (128, 512, 28), (512, 128, 28), (256, 128, 56), (256, 1024, 14)]:
    M = B * H * H
    x = torch.randn(M, C, device="cuda").to(dt); w = (torch.randn(KO, C, device="cuda") / C ** 0.5).to(dt)
    wt = w.t().contiguous(); dy = torch.randn(M, KO, device="cuda").to(dt)
    shift = torch.zeros(KO, device="cuda")
    sweep(f"{C}->{KO}@{H} fwd+stats", lambda: ops.gemm_nt(x, w, bn_shift=shift), KO, 2.0 * (M * C + M * KO + KO * C))
    # input gradient (rows of dy x wt) with the BatchNorm backward of the layer that produced x
    xin = torch.randn(M, C, device="cuda").to(dt)                 # that BatchNorm's input
    mean = torch.zeros(C, device="cuda"); rstd = torch.ones(C, device="cuda")
    gamma = torch.ones(C, device="cuda"); beta = torch.zeros(C, device="cuda")
    bn = ops.BnBwd(xin, mean, rstd, gamma=gamma, beta=beta)
    sweep(f"{KO}->{C}@{H} dgrad+bnbwd(remask)", lambda: ops.gemm_nt_bnbwd(dy, wt, bn), C, 2.0 * (M * KO + 3 * M * C + KO * C))
    if C >= 256:                                                   # the residual join of a block input (C = block width)
        ymask = torch.randn(M, C, device="cuda").to(dt); res = torch.randn(M, C, device="cuda").to(dt)
        bn2 = ops.BnBwd(xin, mean, rstd, ymask=ymask)
        sweep(f"{KO}->{C}@{H} dgrad+bnbwd(join)", lambda: ops.gemm_nt_bnbwd(dy, wt, bn2, residual=res), C,
              2.0 * (M * KO + 5 * M * C + KO * C))
