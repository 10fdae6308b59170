import sys, torch, ctypes
sys.path.insert(0, ".")
from virtex_amd import ops, _lib
dt = torch.bfloat16
for (M, N, K) in [(1024, 4096, 7680), (64, 576, 802816)]:
    a = torch.randn(K, M, device="cuda").to(dt); b = torch.randn(K, N, device="cuda").to(dt)
    out = torch.zeros(M, N, device="cuda")
    def t(split):
        for _ in range(2): ops.gemm_tn_acc(a, b, out, split_k=split)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.gemm_tn_acc(a, b, out, split_k=split)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10 * 1e3
    for cand, cn in ((1, "256x128"), (2, "128x128"), (3, "128x64"), (5, "64x64")):
        _lib.lib().vtx_set_tile_override(ctypes.c_int(cand))
        for split in (1, 2, 4, 8, 16, 64, 256):
            line = f"TN M={M} N={N} K={K} {cn} split {split:3d}:"
            for bits in (0, 1, 4, 14, 15):
                _lib.lib().vtx_set_ablation(ctypes.c_int(bits))
                us = t(split)
                line += f"  abl{bits}={us:7.1f}us"
            _lib.lib().vtx_set_ablation(ctypes.c_int(0))
            print(line + f"  -> {2*M*N*K/t(split)/1e6:6.1f} TF/s", flush=True)
