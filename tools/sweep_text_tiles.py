"""Tile candidates for the text-head GEMMs at the token counts of BASELINE configs 4 / 5 (bs 128 / 64): forward (NT) and input
gradient per shape, every generation-2 candidate, the 64-deep two-stage variants (11, 12) and generation 3 (20, 21).
    python tools/sweep_text_tiles.py [batch] [hidden] [ffn]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from virtex_amd import ops, _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
Hd = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
F = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
T, S, V = 30, 49, 10000
dt = torch.bfloat16
lib = _lib.lib()
names = {-1: "auto", 1: "256x128", 2: "128x128", 3: "128x64", 4: "64x128", 5: "64x64", 11: "256x128k64", 12: "128x128k64", 20: "g3_256x256", 21: "g3_256x128"}


def timeit(fn, iters=8, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def sweep(label, fn, M, N, K):
    res = {}
    for c in names:
        if c in (1, 11, 20, 21) and M < 256:
            continue
        lib.vtx_set_tile_override(ctypes.c_int(c))
        try:
            res[c] = timeit(fn)
        except Exception:
            res[c] = float("inf")
    lib.vtx_set_tile_override(ctypes.c_int(-1))
    best = min((v, k) for k, v in res.items() if k != -1)
    flag = "" if res[-1] <= best[0] * 1.05 else f"   <-- auto loses {100 * (res[-1] / best[0] - 1):.0f}%"
    print(f"{label:22s} M={M:5d} N={N:5d} K={K:5d} auto {res[-1] * 1e6:6.1f} us {2.0 * M * N * K / res[-1] / 1e12:5.0f} TF/s | best {names[best[1]]:10s} {best[0] * 1e6:6.1f} | " +
          " ".join(f"{names[k]}={v * 1e6:.0f}" for k, v in res.items() if k != -1) + flag, flush=True)


for (name, M, N, K) in [("vis_proj", B * S, Hd, 2048), ("in_proj", B * T, 3 * Hd, Hd), ("out_proj", B * T, Hd, Hd), ("kv_proj", B * S, 2 * Hd, Hd),
                        ("ffn1", B * T, F, Hd), ("ffn2", B * T, Hd, F), ("vocab", B * T, V, Hd)]:
    a = torch.randn(M, K, device="cuda").to(dt); b = (torch.randn(N, K, device="cuda") / K ** 0.5).to(dt)
    bt = b.t().contiguous(); dy = torch.randn(M, N, device="cuda").to(dt)
    sweep(name + " fwd", lambda: ops.gemm_nt(a, b), M, N, K)
    sweep(name + " dgrad", lambda: ops.gemm_nt(dy, bt), M, K, N)
