"""Micro-benchmarks of individual C-ABI kernels on a real MI355X (prints one line each)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from virtex_amd import ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def bench_gemm():
    dev = "cuda"
    for dt in (torch.bfloat16, torch.float32):
        for (M, N, K) in [(7680, 4096, 1024), (7680, 1024, 4096), (12544, 1024, 2048), (7680, 10000, 1024),
                          (802816, 64, 64), (802816, 256, 64), (200704, 512, 128), (12544, 2048, 512)]:
            a = torch.randn(M, K, device=dev).to(dt)
            b = torch.randn(N, K, device=dev).to(dt)
            out = torch.empty(M, N, device=dev, dtype=dt)
            t = timeit(lambda: ops.gemm_nt(a, b, out=out))
            t2 = timeit(lambda: torch.matmul(a, b.t()))
            print(f"gemm_nt {str(dt)[6:]:8s} M={M} N={N} K={K}: {t*1e6:9.1f} us {2*M*N*K/t/1e12:7.1f} TF/s | torch(hipBLASLt) {2*M*N*K/t2/1e12:7.1f} TF/s", flush=True)
        for (M, N, K) in [(1024, 4096, 7680), (10000, 1024, 7680), (64, 576, 802816), (512, 512, 12544)]:
            a = torch.randn(K, M, device=dev).to(dt)
            b = torch.randn(K, N, device=dev).to(dt)
            out = torch.zeros(M, N, device=dev)
            t = timeit(lambda: ops.gemm_tn_acc(a, b, out))
            print(f"gemm_tn {str(dt)[6:]:8s} M={M} N={N} K={K}: {t*1e6:9.1f} us {2*M*N*K/t/1e12:7.1f} TF/s", flush=True)


def bench_ln():
    for dt in (torch.bfloat16, torch.float32):
        x = torch.randn(7680 * 8, 1024, device="cuda").to(dt)
        y = torch.randn_like(x)
        g = torch.ones(1024, device="cuda"); b = torch.zeros(1024, device="cuda")
        t = timeit(lambda: ops.layernorm_residual_fwd(x, y, g, b, 1e-5))
        nbytes = 3 * x.numel() * x.element_size()
        print(f"ln_fwd {str(dt)[6:]:8s} rows={x.shape[0]}: {t*1e6:8.1f} us {nbytes/t/1e9:8.1f} GB/s", flush=True)


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    which = sys.argv[1:] or ["gemm", "ln"]
    for w in which:
        globals()["bench_" + w]()
