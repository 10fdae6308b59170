"""The k-major weight-gradient class (dW = dy^T x over the pixels: both operands pixel-major) at the 1x1 shapes of ResNet-50 at
bs 256 -- the largest family of the step (`contraction_v2<256,128,4,2,PlainMC,PlainMC,EpiStore<float>>`
at 0.23 of MFMA / 0.36 of HBM with clean traffic: why?).  Per shape: event timing (us, TB/s over the operands, TFLOP/s), the split
the policy chose, and the same under the library's switches (split-K block target, generation-3 threshold, tile efficiency).
Run under  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY  and read with tools/pmc_by_grid.py for
the wave-time split (PROBE_PLAIN=1: default switches only, a few launches per shape)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from virtex_amd import _lib, ops

dt = torch.bfloat16
B = 256
SHAPES = [  # (cout = M, cin = N, image side): conv1 / conv3 / shortcut 1x1 layers, largest pixel counts first
    (64, 256, 56), (256, 64, 56), (64, 64, 56), (128, 256, 56), (128, 512, 28), (512, 128, 28), (512, 256, 28),
    (256, 512, 28), (256, 1024, 14), (1024, 256, 14), (1024, 512, 14), (512, 1024, 14), (512, 2048, 7), (2048, 512, 7), (2048, 1024, 7)]


def sw(name, v):
    _lib.lib().vtx_set_switch(name.encode(), ctypes.c_int(v))


def timed(dy, x, dw, reps=5):
    for _ in range(2):
        ops.gemm_tn_acc(dy, x, dw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm_tn_acc(dy, x, dw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


plain = os.environ.get("PROBE_PLAIN") == "1"
variants = [("default", [])] if plain else [
    ("default", []), ("splitk_blocks=256", [("splitk_blocks", 256)]), ("splitk_blocks=1024", [("splitk_blocks", 1024)]),
    ("gen3_mc=100", [("gen3_mc", 100)]), ("gen3_mc off", [("gen3_mc", 0)]), ("mc_eff128=100", [("mc_eff128", 100)]), ("mc_eff128=40", [("mc_eff128", 40)])]
defaults = {"splitk_blocks": 512, "gen3_mc": 800, "mc_eff128": 70}
print(f"{'shape (cout x cin @ side)':28s} {'MB':>6s}  " + "  ".join(f"{n:>18s}" for n, _ in variants) + "   (us incl. the split-K reduction; best TB/s)")
tot = [0.0] * len(variants)
for (M, N, H) in SHAPES:
    P = B * H * H
    dy = torch.randn(P, M, device="cuda").to(dt); x = torch.randn(P, N, device="cuda").to(dt)
    dw = torch.zeros(M, N, device="cuda")
    byts = 2.0 * P * (M + N)
    row = []
    for vi, (name, sets) in enumerate(variants):
        for k, v in defaults.items():
            sw(k, v)
        for k, v in sets:
            sw(k, v)
        t = timed(dy, x, dw, reps=3 if plain else 5)
        row.append(t); tot[vi] += t
    for k, v in defaults.items():
        sw(k, v)
    best = min(row)
    print(f"{M:5d} x {N:5d} @ {H:2d}  P={P:7d} {byts / 1e6:6.0f}  " + "  ".join(f"{t * 1e6:18.1f}" for t in row) +
          f"   {byts / best / 1e12:.2f} TB/s  {2.0 * P * M * N / best / 1e12:5.0f} TF/s", flush=True)
    del dy, x, dw
print(f"{'sum over the shapes (us)':42s}  " + "  ".join(f"{t * 1e6:18.1f}" for t in tot))
