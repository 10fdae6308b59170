"""Weight-gradient launches for rocprofv3 counter passes (round 3): the streaming 3x3 kernel at 56x56 / 28x28, the
implicit-GEMM 3x3 kernel at 14x14, the k-major contraction kernel on a text shape and on a late 1x1 convolution."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from virtex_amd import ops
dt, dev, B = torch.bfloat16, "cuda", 256
def r(*s): return torch.randn(*s, device=dev).to(dt)
cases = []
for (C, H) in [(64, 56), (128, 28), (256, 14)]:
    cases.append((r(B, H, H, C), r(B, H, H, C), torch.zeros(C, 3, 3, C, device=dev)))
at, bt, dw = r(7680, 1024), r(7680, 4096), torch.zeros(1024, 4096, device=dev)          # ffn1 weight gradient
a2, b2, dw2 = r(B * 14 * 14, 1024), r(B * 14 * 14, 256), torch.zeros(1024, 256, device=dev)   # 256 -> 1024 @ 14x14
for _ in range(3):
    for x, dy, d in cases:
        ops.conv2d_wgrad(x, dy, d, 1, 1)
    ops.gemm_tn_acc(at, bt, dw)
    ops.gemm_tn_acc(a2, b2, dw2)
torch.cuda.synchronize()
