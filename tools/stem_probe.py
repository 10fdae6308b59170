"""The streaming stem convolution (csrc/stem.hip) at bs 256 with the XCD-major strip order (switch stem_stream = 1) and the plain
one (= 2): event timing per launch; run under rocprofv3 --pmc FETCH_SIZE for the bytes each order fetches (tools/r06_s8.sh)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from virtex_amd import _lib, ops

dt = torch.bfloat16
B = 256
x = torch.randn(B, 230, 230, 4, device="cuda").to(dt)
w = (torch.randn(64, 7, 8, 4, device="cuda") / 14).to(dt)
shift = torch.zeros(64, device="cuda")
outs = {}
for order in (1, 2, 1, 2):
    _lib.lib().vtx_set_switch(b"stem_stream", ctypes.c_int(order))
    for _ in range(3):
        y, st = ops.conv2d_fwd(x, w, 2, 0, bn_shift=shift)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        y, st = ops.conv2d_fwd(x, w, 2, 0, bn_shift=shift)
    e1.record(); torch.cuda.synchronize()
    outs[order] = y
    print(f"stem_stream = {order} ({'XCD-major' if order == 1 else 'plain'} strip order): {e0.elapsed_time(e1) / 10 * 1e3:7.1f} us per launch", flush=True)
_lib.lib().vtx_set_switch(b"stem_stream", ctypes.c_int(1))
print("outputs identical:", torch.equal(outs[1], outs[2]))
