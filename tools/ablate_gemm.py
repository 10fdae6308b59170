import sys, torch, ctypes
sys.path.insert(0, ".")
from virtex_amd import ops, _lib
dt = torch.bfloat16
M, N, K = 7680, 4096, 1024
a = torch.randn(M, K, device="cuda").to(dt); b = torch.randn(N, K, device="cuda").to(dt)
out = torch.empty(M, N, device="cuda", dtype=dt)
def t():
    for _ in range(3): ops.gemm_nt(a, b, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gemm_nt(a, b, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
names = {0: "full", 1: "no MFMA", 2: "no frag reads", 4: "no DMA", 8: "no barrier/wait", 3: "no MFMA+no reads", 6: "no reads+no DMA", 7: "only barrier", 14: "only MFMA", 15: "nothing"}
for cand, cn in ((0, "256x256"), (1, "256x128"), (2, "128x128")):
    _lib.lib().vtx_set_tile_override(ctypes.c_int(cand))
    for bits, n in names.items():
        _lib.lib().vtx_set_ablation(ctypes.c_int(bits))
        us = t()
        print(f"{cn} abl {bits:2d} {n:22s}: {us:7.1f} us  ({2*M*N*K/us/1e6:7.1f} TF/s equiv)", flush=True)
