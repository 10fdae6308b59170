import sys, torch
sys.path.insert(0, ".")
from virtex_amd import ops
dt = torch.bfloat16
for (M, N, K) in [(7680, 4096, 1024), (802816, 256, 64), (50176, 1024, 256)]:
    a = torch.randn(M, K, device="cuda").to(dt); b = torch.randn(N, K, device="cuda").to(dt)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    for _ in range(5): ops.gemm_nt(a, b, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.gemm_nt(a, b, out=out)
    e1.record(); torch.cuda.synchronize()
    plain = e0.elapsed_time(e1) / 50 * 1e3
    ops.profile_start()
    for _ in range(50): ops.gemm_nt(a, b, out=out)
    r = ops.profile_stop()[0]
    print(f"M={M} N={N} K={K}: back-to-back {plain:.1f} us/launch, instrumented per-launch {r['seconds']/r['launches']*1e6:.1f} us")
