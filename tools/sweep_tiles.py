"""For every distinct GEMM-shaped op of the step: time each block-tile candidate and report the best next to
the automatic choice (data for the tile picker's scoring table)."""
import ctypes, sys, torch
sys.path.insert(0, ".")
from virtex_amd import ops, _lib
sys.path.insert(0, "tools")
from bench_layers import CONVS, timeit  # noqa

B, dt = 256, torch.bfloat16
lib = _lib.lib()
names = {-1: "auto", 0: "256x256", 1: "256x128", 2: "128x128", 3: "128x64", 4: "64x128", 5: "64x64", 11: "256x128k64s2", 12: "128x128k64s2"}
CANDS = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [-1, 0, 1, 2, 3, 4, 5]
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""          # "kc": skip the weight gradients; "wgrad": only them
ONLY_KC = ONLY not in ("", "wgrad")
def sweep(label, fn, M, N):
    if ONLY_KC and "wgrad" in label:
        return
    if ONLY == "wgrad" and "wgrad" not in label:
        return
    res = {}
    for c in CANDS:
        if c in (0, 1) and M < 256: continue
        if c in (0,) and N <= 128: continue
        if c in (0, 1, 2, 4) and N <= 64 and c != -1: continue
        lib.vtx_set_tile_override(ctypes.c_int(c))
        try:
            res[c] = timeit(fn, iters=6, warm=2)
        except Exception as e:
            res[c] = float("inf")
    lib.vtx_set_tile_override(ctypes.c_int(-1))
    best = min((v, k) for k, v in res.items() if k != -1)
    flag = "" if res[-1] <= best[0] * 1.05 else f"   <-- auto loses {100*(res[-1]/best[0]-1):.0f}%"
    print(f"{label:44s} auto {res[-1]*1e6:7.1f} us | best {names[best[1]]:8s} {best[0]*1e6:7.1f} us |" +
          " ".join(f"{names[k]}={v*1e6:.0f}" for k, v in res.items() if k != -1) + flag, flush=True)

for (C, KO, k, s, H, cnt) in CONVS:
    if k == 7:
        continue                      # the packed stem has one sensible tile (N = 64)
    pad = {1: 0, 3: 1, 7: 3}[k]
    OH = (H + 2 * pad - k) // s + 1
    x = torch.randn(B, H, H, C, device="cuda").to(dt)
    w = (torch.randn(KO, k, k, C, device="cuda") / (k * k * C) ** 0.5).to(dt)
    wt = w.permute(3, 1, 2, 0).contiguous()
    dy = torch.randn(B, OH, OH, KO, device="cuda").to(dt)
    dw = torch.zeros(KO, k, k, C, device="cuda")
    gemm = (k == 1 and s == 1)
    tag = f"conv {C}->{KO} k{k}s{s}@{H}"
    if gemm:
        sweep(tag + " fwd", lambda: ops.gemm_nt(x.view(-1, C), w.view(KO, C)), B * OH * OH, KO)
        sweep(tag + " dgrad", lambda: ops.gemm_nt(dy.view(-1, KO), wt.view(C, KO)), B * H * H, C)
        sweep(tag + " wgrad", lambda: ops.gemm_tn_acc(dy.view(-1, KO), x.view(-1, C), dw.view(KO, C)), KO, C)
    else:
        sweep(tag + " fwd", lambda: ops.conv2d_fwd(x, w, s, pad), B * OH * OH, KO)
        if C != 8:
            sweep(tag + " dgrad", lambda: ops.conv2d_dgrad(dy, wt, x.shape, s, pad), B * H * H // (s * s), C)
        sweep(tag + " wgrad", lambda: ops.conv2d_wgrad(x, dy, dw, s, pad), KO, k * k * C)
T, S, Hd, F, V = 30, 49, 1024, 4096, 10000
for (name, M, N, K) in [("vis_proj", B * S, Hd, 2048), ("in_proj", B * T, 3 * Hd, Hd), ("out_proj", B * T, Hd, Hd),
                        ("kv_proj", B * S, 2 * Hd, Hd), ("ffn1", B * T, F, Hd), ("ffn2", B * T, Hd, F), ("vocab", B * T, V, Hd)]:
    a = torch.randn(M, K, device="cuda").to(dt); b = torch.randn(N, K, device="cuda").to(dt)
    bt = b.t().contiguous(); dyy = torch.randn(M, N, device="cuda").to(dt); dwt = torch.zeros(N, K, device="cuda")
    sweep(f"gemm {name} fwd", lambda: ops.gemm_nt(a, b), M, N)
    sweep(f"gemm {name} dgrad", lambda: ops.gemm_nt(dyy, bt), M, K)
    sweep(f"gemm {name} wgrad", lambda: ops.gemm_tn_acc(dyy, a, dwt), N, K)
