"""A handful of representative contraction launches for rocprofv3 counter passes (tools/pmc_kernels.sh)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from virtex_amd import ops
dt = torch.bfloat16
dev = "cuda"
B = 256
def r(*s): return torch.randn(*s, device=dev).to(dt)
# 1. big NT GEMM (ffn1)
a, b = r(7680, 1024), r(4096, 1024)
# 2. 3x3 conv 128->128 @28
x, w = r(B, 28, 28, 128), r(128, 3, 3, 128)
# 3. HBM-bound pointwise 64->256 @56
pa, pw = r(B * 56 * 56, 64), r(256, 64)
# 4. weight gradient (k-major operands)
at, bt = r(7680, 1024), r(7680, 4096)
dw = torch.zeros(1024, 4096, device=dev)
# 5. latency-bound small GEMM (out_proj)
sa, sb = r(7680, 1024), r(1024, 1024)
for _ in range(4):
    ops.gemm_nt(a, b)
    ops.conv2d_fwd(x, w, 1, 1)
    ops.gemm_nt(pa, pw)
    ops.gemm_tn_acc(at, bt, dw)
    ops.gemm_nt(sa, sb)
torch.cuda.synchronize()
