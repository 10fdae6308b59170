import sys, torch
sys.path.insert(0, ".")
from virtex_amd import ops
mode = sys.argv[1] if len(sys.argv) > 1 else "nt"
dt = torch.bfloat16
if mode == "nt":
    M, N, K = 7680, 4096, 1024
    a = torch.randn(M, K, device="cuda").to(dt); b = torch.randn(N, K, device="cuda").to(dt)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    for _ in range(5): ops.gemm_nt(a, b, out=out)
else:
    M, N, K = 1024, 4096, 7680
    a = torch.randn(K, M, device="cuda").to(dt); b = torch.randn(K, N, device="cuda").to(dt)
    out = torch.zeros(M, N, device="cuda")
    for _ in range(5): ops.gemm_tn_acc(a, b, out)
torch.cuda.synchronize()
