"""Calibration only (NOT on the product path): what the vendor libraries shipped with PyTorch-ROCm (hipBLASLt / MIOpen)
reach on this MI355X for the step's GEMM / convolution shapes, bf16.  Tells whether a hand-written kernel's TF/s is far
from what is practically attainable for the SHAPE, as opposed to the 2.5 PF datasheet peak."""
import torch, torch.nn.functional as F
dt = torch.bfloat16
def t(fn, it=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3
print("hipBLASLt (torch.mm, A[M,K] @ B[N,K]^T)")
for (M, N, K) in [(7680, 4096, 1024), (7680, 1024, 4096), (7680, 10000, 1024), (7680, 3072, 1024), (7680, 1024, 1024), (12544, 1024, 2048),
                  (8192, 8192, 8192), (50176, 1024, 256), (50176, 256, 1024), (802816, 256, 64), (802816, 64, 256)]:
    a = torch.randn(M, K, device="cuda", dtype=dt); b = torch.randn(N, K, device="cuda", dtype=dt)
    s = t(lambda: torch.mm(a, b.t()))
    print(f"  M={M:7d} N={N:5d} K={K:5d}  {s*1e6:8.1f} us  {2*M*N*K/s/1e12:7.0f} TF/s  {(M*K+N*K+M*N)*2/s/1e9:7.0f} GB/s", flush=True)
print("MIOpen (F.conv2d, channels_last bf16), B=256")
for (C, KO, k, s_, H) in [(64, 64, 3, 1, 56), (128, 128, 3, 1, 28), (256, 256, 3, 1, 14), (512, 512, 3, 1, 7), (128, 128, 3, 2, 56), (3, 64, 7, 2, 224)]:
    x = torch.randn(256, C, H, H, device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last)
    w = torch.randn(KO, C, k, k, device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last)
    pad = k // 2
    try:
        s = t(lambda: F.conv2d(x, w, stride=s_, padding=pad), it=10)
        OH = (H + 2 * pad - k) // s_ + 1
        print(f"  conv {C:4d}->{KO:4d} k{k} s{s_} @{H:3d}  {s*1e6:8.1f} us  {2*256*OH*OH*KO*k*k*C/s/1e12:7.0f} TF/s", flush=True)
    except Exception as e:
        print("  conv", C, KO, k, "failed:", str(e)[:80])
