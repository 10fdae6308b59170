"""Generation-3 contraction kernels (gemm_v3.h) against what the tile picker takes today, per shape of the step at bs = 256:
text-head GEMMs, 3x3 / strided / late 1x1 convolutions, forward and input gradient.  Per shape: microseconds and TFLOP/s
for  auto (generation 2, the picker's tile) | 20 (generation 3, 256x256 blocks) | 21 (generation 3, 256x128 blocks).

    python tools/bench_gen3.py [--race]      # --race: 30 launches per forced variant must be bit-identical, and equal to
                                             #         the fp32 torch product of the same operands within bf16 rounding
"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from virtex_amd import _lib, ops

B = 256
dt = torch.bfloat16
lib = _lib.lib()


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def three(fn):
    out = []
    for c in (-1, 20, 21):
        lib.vtx_set_tile_override(ctypes.c_int(c))
        try:
            fn()
            gen = lib.vtx_last_contraction_generation()
            out.append((timeit(fn), gen))
        finally:
            lib.vtx_set_tile_override(ctypes.c_int(-1))
    return out


def show(name, flops, res):
    cells = " | ".join(f"{t*1e6:7.1f} us {flops/t/1e12:5.0f} TF g{g}" for (t, g) in res)
    best = min(range(3), key=lambda i: res[i][0])
    print(f"{name:44s} | {cells} | best {('auto', 'v3-256', 'v3-128')[best]} x{res[0][0]/res[best][0]:.2f}", flush=True)
    return res


def race(name, fn, ref_fn):
    for c in (20, 21):
        lib.vtx_set_tile_override(ctypes.c_int(c))
        try:
            first = fn().clone()
            bad = 0
            for _ in range(30):
                bad += int(not torch.equal(fn(), first))
            ref = ref_fn()
            err = ((first.float() - ref).norm() / ref.norm()).item()
            print(f"race {name:38s} cand {c}: {bad}/30 launches differ, rel err vs fp32 torch {err:.2e}", flush=True)
        finally:
            lib.vtx_set_tile_override(ctypes.c_int(-1))


def main():
    do_race = "--race" in sys.argv
    T, S, H, F, V = 30, 49, 1024, 4096, 10000
    GEMMS = [("vis_proj", B * S, H, 2048, 2), ("self in_proj", B * T, 3 * H, H, 2), ("out_proj/q", B * T, H, H, 6),
             ("kv_proj", B * S, 2 * H, H, 2), ("ffn1", B * T, F, H, 2), ("ffn2", B * T, H, F, 2), ("vocab", B * T, V, H, 6)]
    tot = [0.0, 0.0, 0.0]
    print(f"{'shape':44s} | {'auto (generation 2)':24s} | {'20: v3 256x256':24s} | {'21: v3 256x128':24s} |")
    for (name, M, N, K, cnt) in GEMMS:
        a = torch.randn(M, K, device="cuda").to(dt); b = torch.randn(N, K, device="cuda").to(dt)
        bt = b.t().contiguous(); dy = torch.randn(M, N, device="cuda").to(dt)
        fl = 2.0 * M * N * K
        r1 = show(f"gemm {name:12s} fwd   {M}x{N}x{K}", fl, three(lambda: ops.gemm_nt(a, b)))
        r2 = show(f"gemm {name:12s} dgrad {M}x{K}x{N}", fl, three(lambda: ops.gemm_nt(dy, bt)))
        for i in range(3):
            tot[i] += cnt * (r1[i][0] + r2[i][0]) / 2 * (2 if name != "vocab" else 1)
        if do_race:
            race(f"{name} fwd", lambda: ops.gemm_nt(a, b), lambda: a.float() @ b.float().t())
    print(f"text-head forward + input-gradient GEMMs per step (weighted): auto {tot[0]*1e3:.2f} ms, v3-256 {tot[1]*1e3:.2f}, v3-128 {tot[2]*1e3:.2f}")
    # (Cin, Cout, k, stride, Hin, count)
    CONVS = [(128, 128, 3, 1, 28, 3), (256, 256, 3, 2, 28, 1), (256, 256, 3, 1, 14, 5), (512, 512, 3, 2, 14, 1), (512, 512, 3, 1, 7, 2),
             (512, 256, 1, 1, 28, 1), (256, 1024, 1, 1, 14, 6), (1024, 256, 1, 1, 14, 5), (1024, 512, 1, 1, 14, 1),
             (512, 1024, 1, 2, 28, 1), (512, 2048, 1, 1, 7, 3), (2048, 512, 1, 1, 7, 2), (1024, 2048, 1, 2, 14, 1)]
    ctot = [0.0, 0.0, 0.0]
    for (C, KO, k, s, Hh, cnt) in CONVS:
        pad = k // 2
        OH = (Hh + 2 * pad - k) // s + 1
        x = torch.randn(B, Hh, Hh, C, device="cuda").to(dt)
        w = (torch.randn(KO, k, k, C, device="cuda") / (k * k * C) ** 0.5).to(dt)
        wt = w.permute(3, 1, 2, 0).contiguous()
        dy = torch.randn(B, OH, OH, KO, device="cuda").to(dt)
        fl = 2.0 * B * OH * OH * KO * k * k * C
        if k == 1 and s == 1:
            f = lambda: ops.gemm_nt(x.view(-1, C), w.view(KO, C))
            d = lambda: ops.gemm_nt(dy.view(-1, KO), wt.view(C, KO))
        else:
            f = lambda: ops.conv2d_fwd(x, w, s, pad)
            d = lambda: ops.conv2d_dgrad(dy, wt, x.shape, s, pad)
        r1 = show(f"conv {C:4d}->{KO:4d} k{k} s{s} @{Hh:2d} fwd", fl, three(f))
        r2 = show(f"conv {C:4d}->{KO:4d} k{k} s{s} @{Hh:2d} dgrad", fl, three(d))
        for i in range(3):
            ctot[i] += cnt * (r1[i][0] + r2[i][0])
        if do_race and k == 3:
            xr = x.float().permute(0, 3, 1, 2); wr = w.float().permute(0, 3, 1, 2)
            race(f"conv {C}->{KO} k{k} s{s} @{Hh} fwd", f,
                 lambda: torch.nn.functional.conv2d(xr, wr, stride=s, padding=pad).permute(0, 2, 3, 1))
    print(f"these convolutions forward + input gradient per step: auto {ctot[0]*1e3:.2f} ms, v3-256 {ctot[1]*1e3:.2f}, v3-128 {ctot[2]*1e3:.2f}")


if __name__ == "__main__":
    main()
