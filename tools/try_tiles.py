import sys, torch, ctypes
sys.path.insert(0, ".")
from virtex_amd import ops, _lib
dt = torch.bfloat16
def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
cands = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 6, 7]
for (M, N, K) in [(7680, 4096, 1024), (7680, 1024, 4096), (7680, 10000, 1024), (12544, 1024, 2048), (50176, 1024, 256), (50176, 256, 1024), (802816, 256, 64), (802816, 64, 256)]:
    a = torch.randn(M, K, device="cuda").to(dt); b = torch.randn(N, K, device="cuda").to(dt)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    row = []
    for c in cands:
        _lib.lib().vtx_set_tile_override(ctypes.c_int(c))
        try:
            us = t(lambda: ops.gemm_nt(a, b, out=out))
            row.append(f"c{c}={us:7.1f}us {2*M*N*K/us/1e6:6.0f}TF")
        except Exception as e:
            row.append(f"c{c}=ERR")
    _lib.lib().vtx_set_tile_override(ctypes.c_int(-1))
    print(f"M={M:7d} N={N:5d} K={K:5d} | " + " | ".join(row), flush=True)
