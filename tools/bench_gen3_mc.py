"""Generation-3 weight-gradient kernels (gemm_v3mc.h) against generation 2, per weight-gradient shape of the step at bs = 256:
auto (generation 2 with its own split-K policy) | 20 (256x256 blocks) | 21 (256x128 blocks), each including its split-K
reduction.  --race: 20 launches per forced variant bit-identical, and equal to the fp32 torch product."""
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
            out.append((timeit(fn), lib.vtx_last_contraction_generation()))
        finally:
            lib.vtx_set_tile_override(ctypes.c_int(-1))
    return out


def show(name, flops, res, cnt):
    cells = " | ".join(f"{t*1e6:7.1f} us {flops/t/1e12:5.0f} TF g{g}" for (t, g) in res)
    best = min(range(3), key=lambda i: res[i][0])
    print(f"{name:46s} x{cnt:2d} | {cells} | best {('auto', 'v3-256', 'v3-128')[best]} x{res[0][0]/res[best][0]:.2f}", flush=True)


def main():
    race = "--race" in sys.argv
    T, S, H, F, V = 30, 49, 1024, 4096, 10000
    tot = [0.0, 0.0, 0.0]
    GEMMS = [("vis_proj", B * S, H, 2048, 2), ("self in_proj", B * T, 3 * H, H, 2), ("out_proj/q", B * T, H, H, 6),
             ("kv_proj", B * S, 2 * H, H, 2), ("ffn1", B * T, F, H, 2), ("ffn2", B * T, H, F, 2), ("vocab", B * T, V, H, 2)]
    print(f"{'shape':46s}     | {'auto (generation 2)':24s} | {'20: v3 256x256':24s} | {'21: v3 256x128':24s} |")
    for (name, K, M, N, cnt) in GEMMS:                      # dW[M][N] += dy[K][M]^T x[K][N]
        dy = torch.randn(K, M, device="cuda").to(dt); x = torch.randn(K, N, device="cuda").to(dt)
        dw = torch.zeros(M, N, device="cuda")
        fn = lambda: ops.gemm_tn_acc(dy, x, dw)
        r = three(fn)
        show(f"wgrad {name:12s} {M}x{N} K={K}", 2.0 * M * N * K, r, cnt)
        for i in range(3):
            tot[i] += cnt * r[i][0]
        if race:
            for c in (20, 21):
                lib.vtx_set_tile_override(ctypes.c_int(c))
                outs = []
                for _ in range(20):
                    dw.zero_(); fn(); outs.append(dw.clone())
                lib.vtx_set_tile_override(ctypes.c_int(-1))
                bad = sum(int(not torch.equal(o, outs[0])) for o in outs)
                ref = dy.float().t() @ x.float()
                print(f"   race cand {c}: {bad}/20 differ, rel err {((outs[0] - ref).norm() / ref.norm()).item():.2e}", flush=True)
    print(f"text weight gradients per step: auto {tot[0]*1e3:.2f} ms, v3-256 {tot[1]*1e3:.2f}, v3-128 {tot[2]*1e3:.2f}")
    CONVS = [(256, 512, 1, 2, 56, 1), (512, 128, 1, 1, 28, 3), (512, 256, 1, 1, 28, 1), (128, 512, 1, 1, 28, 4), (256, 1024, 1, 1, 14, 6), (1024, 256, 1, 1, 14, 5),
             (1024, 512, 1, 1, 14, 1), (512, 1024, 1, 2, 28, 1), (512, 2048, 1, 1, 7, 3), (2048, 512, 1, 1, 7, 2), (1024, 2048, 1, 2, 14, 1),
             (256, 256, 3, 2, 28, 1), (256, 256, 3, 1, 14, 5), (512, 512, 3, 2, 14, 1), (512, 512, 3, 1, 7, 2), (128, 128, 3, 1, 28, 3)]
    ctot = [0.0, 0.0, 0.0]
    for (C, KO, k, s, Hh, cnt) in CONVS:
        pad = k // 2
        OH = (Hh + 2 * pad - k) // s + 1
        x = torch.randn(B, Hh, Hh, C, device="cuda").to(dt)
        dy = torch.randn(B, OH, OH, KO, device="cuda").to(dt)
        dw = torch.zeros(KO, k, k, C, device="cuda")
        fl = 2.0 * B * OH * OH * KO * k * k * C
        if k == 1 and s == 1:
            fn = lambda: ops.gemm_tn_acc(dy.view(-1, KO), x.view(-1, C), dw.view(KO, C))
        else:
            fn = lambda: ops.conv2d_wgrad(x, dy, dw, s, pad)
        r = three(fn)
        show(f"wgrad conv {C:4d}->{KO:4d} k{k} s{s} @{Hh:2d}", fl, r, cnt)
        for i in range(3):
            ctot[i] += cnt * r[i][0]
        if race and k == 3:
            lib.vtx_set_tile_override(ctypes.c_int(20))
            dw.zero_(); fn(); a = dw.clone(); dw.zero_(); fn()
            same = torch.equal(a, dw)
            lib.vtx_set_tile_override(ctypes.c_int(-1))
            dw.zero_(); fn()
            print(f"   cand 20 repeatable {same}, rel err vs generation 2 {((a - dw).norm() / dw.norm()).item():.2e}", flush=True)
    print(f"these convolution weight gradients per step: auto {ctot[0]*1e3:.2f} ms, v3-256 {ctot[1]*1e3:.2f}, v3-128 {ctot[2]*1e3:.2f}")


if __name__ == "__main__":
    main()
