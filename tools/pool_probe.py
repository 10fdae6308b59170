"""The stem's pooling tails (vtx_bn_fwd_maxpool, vtx_bn_bwd_maxpool) at bs 256 with the XCD-major block order (switch pool_xcd = 1)
and the plain one (0): event timing per call; under rocprofv3 --pmc FETCH_SIZE (POOL_XCD=0/1 fixes the order for a whole run) the
bytes each order fetches (tools/r06_s11.sh)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from virtex_amd import _lib, ops

dt = torch.bfloat16
B, H, C = 256, 112, 64
x = torch.randn(B, H, H, C, device="cuda").to(dt)
gamma = torch.rand(C, device="cuda") + 0.5; beta = torch.randn(C, device="cuda") * 0.1
rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
nbt = torch.zeros((), dtype=torch.int64, device="cuda")
fixed = os.environ.get("POOL_XCD")
orders = [int(fixed)] * 2 if fixed is not None else [1, 0, 1, 0]
res = {}
for order in orders:
    _lib.lib().vtx_set_switch(b"pool_xcd", ctypes.c_int(order))
    for _ in range(3):
        pooled, arg, mean, rstd = ops.bn_fwd_maxpool(x, gamma, beta, rm, rv, nbt)
    dpool = torch.randn_like(pooled)
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    for _ in range(2):
        dx = ops.bn_bwd_maxpool(x, dpool, arg, gamma, beta, mean, rstd, dg, db)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(10):
        pooled, arg, mean, rstd = ops.bn_fwd_maxpool(x, gamma, beta, rm, rv, nbt)
    e[1].record()
    for _ in range(10):
        dx = ops.bn_bwd_maxpool(x, dpool, arg, gamma, beta, mean, rstd, dg, db)
    e[2].record(); torch.cuda.synchronize()
    res[order] = (pooled.clone(), arg.clone(), dx.clone())
    print(f"pool_xcd = {order}: forward tail {e[0].elapsed_time(e[1]) * 100:7.1f} us, backward tail (reduce + apply) {e[1].elapsed_time(e[2]) * 100:7.1f} us per call", flush=True)
_lib.lib().vtx_set_switch(b"pool_xcd", ctypes.c_int(1))
if len(res) == 2:
    print("results identical:", all(torch.equal(a, b) for a, b in zip(res[0], res[1])))
