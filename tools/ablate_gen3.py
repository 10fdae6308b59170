"""Where the time of a generation-3 launch goes (needs the measurement build: python -m virtex_amd.build --variant ablate
--define VTX_ABLATE; run with VIRTEX_AMD_LIB=.../libvirtex_amd_ablate.so).
 1. ablations: full | no K loop (epilogue only) | no epilogue | loop without DMA | without MFMA | without fragment reads
 2. per-wave shader-clock stamps (kernel entry, K loop start, K loop end, epilogue end) of one launch: distribution over the
    blocks of prologue / loop / epilogue cycles and of the block start times."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from virtex_amd import _lib, ops

lib = _lib.lib()
dt = torch.bfloat16


def t(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


NAMES = [(0, "full"), (64, "no K loop"), (128, "no epilogue"), (4, "no DMA in loop"), (1, "no MFMA"), (2, "no frag reads"),
         (4 + 128, "no DMA, no epilogue"), (1 + 2 + 128, "DMA + barriers only"), (1 + 4 + 128, "reads + barriers only"),
         (2 + 4 + 128, "MFMA + barriers only"), (1 + 2 + 4 + 128, "barriers only")]


def stamps(fn, nblocks, label):
    buf = torch.zeros(nblocks * 8 * 4, dtype=torch.int64, device="cuda")
    lib.vtx_set_debug_buffer(ctypes.c_void_p(buf.data_ptr()))
    fn(); torch.cuda.synchronize()
    lib.vtx_set_debug_buffer(ctypes.c_void_p(0))
    s = buf.view(nblocks, 8, 4).cpu().double()
    t0 = s[:, :, 0].min()
    start = (s[:, :, 0].min(1).values - t0)
    pro = (s[:, :, 1] - s[:, :, 0]).mean(1); loop = (s[:, :, 2] - s[:, :, 1]).mean(1); epi = (s[:, :, 3] - s[:, :, 2]).max(1).values
    end = (s[:, :, 3].max(1).values - t0)
    q = lambda x: "min %8.0f  med %8.0f  max %8.0f" % (x.min().item(), x.median().item(), x.max().item())
    print(f"stamps {label}: {nblocks} blocks (shader-clock cycles)")
    print("   block start after first : " + q(start))
    print("   prologue (entry -> loop): " + q(pro))
    print("   K loop                  : " + q(loop))
    print("   epilogue (slowest wave) : " + q(epi))
    print("   block end after first   : " + q(end), flush=True)
    order = torch.argsort(start)
    n2 = (start > start.median() + 0.5 * loop.median()).sum().item()
    print(f"   blocks starting more than half a loop after the median start (second round): {n2}")


def main():
    B, T = 256, 30
    shapes = [("ffn1 fwd", B * T, 4096, 1024), ("ffn2 fwd", B * T, 1024, 4096), ("out_proj", B * T, 1024, 1024), ("vocab fwd", B * T, 10000, 1024)]
    for (name, M, N, K) in shapes:
        a = torch.randn(M, K, device="cuda").to(dt); b = torch.randn(N, K, device="cuda").to(dt)
        out = torch.empty(M, N, device="cuda", dtype=dt)
        fn = lambda: ops.gemm_nt(a, b, out=out)
        for cand, bn in ((20, 256), (21, 128)):
            lib.vtx_set_tile_override(ctypes.c_int(cand))
            for bits, nm in NAMES:
                lib.vtx_set_ablation(ctypes.c_int(bits))
                us = t(fn)
                print(f"{name:10s} {M}x{N}x{K} cand {cand} abl {bits:3d} {nm:24s}: {us:7.1f} us ({2.0*M*N*K/us/1e6:6.0f} TF/s equiv)", flush=True)
            lib.vtx_set_ablation(ctypes.c_int(0))
            nblocks = ((M + 255) // 256) * ((N + bn - 1) // bn)
            stamps(fn, nblocks, f"{name} cand {cand}")
            lib.vtx_set_tile_override(ctypes.c_int(-1))
    # one 3x3 convolution (gather loader) the same way
    x = torch.randn(B, 14, 14, 256, device="cuda").to(dt); w = (torch.randn(256, 3, 3, 256, device="cuda") / 48).to(dt)
    fn = lambda: ops.conv2d_fwd(x, w, 1, 1)
    for cand, bn in ((20, 256), (21, 128)):
        lib.vtx_set_tile_override(ctypes.c_int(cand))
        for bits, nm in NAMES[:6]:
            lib.vtx_set_ablation(ctypes.c_int(bits))
            us = t(fn)
            print(f"conv3x3 256@14 cand {cand} abl {bits:3d} {nm:24s}: {us:7.1f} us", flush=True)
        lib.vtx_set_ablation(ctypes.c_int(0))
        stamps(fn, 196 * (256 // bn), f"conv3x3 256@14 cand {cand}")
        lib.vtx_set_tile_override(ctypes.c_int(-1))


def stats_epilogues():
    """the convolution epilogues of the real step: statistics of the stored output (forward), fused BatchNorm backward (dgrad)"""
    B = 256
    for (C, KO, k, H) in ((256, 256, 3, 14), (512, 512, 3, 7), (1024, 256, 1, 14), (256, 1024, 1, 14)):
        pad = k // 2
        x = torch.randn(B, H, H, C, device="cuda").to(dt); w = (torch.randn(KO, k, k, C, device="cuda") / (k * k * C) ** 0.5).to(dt)
        wt = w.permute(3, 1, 2, 0).contiguous()
        dy = torch.randn(B, H, H, KO, device="cuda").to(dt)
        shift = torch.zeros(KO, device="cuda")
        xin = torch.randn(B, H, H, C, device="cuda").to(dt)       # the BatchNorm input of the layer below (same shape as dx)
        mean = torch.zeros(C, device="cuda"); rstd = torch.ones(C, device="cuda"); gamma = torch.ones(C, device="cuda"); beta = torch.zeros(C, device="cuda")
        bn = ops.BnBwd(xin, mean, rstd, gamma=gamma, beta=beta)
        if k == 1:
            f = lambda: ops.gemm_nt(x.view(-1, C), w.view(KO, C), bn_shift=shift)
            d = lambda: ops.gemm_nt_bnbwd(dy.view(-1, KO), wt.view(C, KO), ops.BnBwd(xin.view(-1, C), mean, rstd, gamma=gamma, beta=beta))
        else:
            f = lambda: ops.conv2d_fwd(x, w, 1, pad, bn_shift=shift)
            d = lambda: ops.conv2d_dgrad(dy, wt, x.shape, 1, pad, bn=bn)
        M = B * H * H
        for (nm, fn, N) in (("fwd+stats", f, KO), ("dgrad+bnbwd", d, C)):
            for cand, bn_ in ((-1, 0), (20, 256), (21, 128)):
                lib.vtx_set_tile_override(ctypes.c_int(cand))
                us = t(fn)
                print(f"conv {C}->{KO} k{k} @{H} {nm:12s} cand {cand:2d}: {us:7.1f} us gen {lib.vtx_last_contraction_generation()}", flush=True)
                if cand > 0:
                    stamps(fn, ((M + 255) // 256) * ((N + bn_ - 1) // bn_), f"conv {C}->{KO} k{k} @{H} {nm} cand {cand}")
                lib.vtx_set_tile_override(ctypes.c_int(-1))


if __name__ == "__main__":
    if "--stats" in sys.argv:
        stats_epilogues()
        sys.exit(0)
    main()
