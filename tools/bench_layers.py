"""Per-layer micro-benchmark of every distinct GEMM-shaped op of the step at bs=256 (bf16):
time, TFLOP/s, and the HBM GB/s implied by the algorithmic bytes (in + weights + out)."""
import sys
import torch
sys.path.insert(0, ".")
from virtex_amd import ops

B = 256
dt = torch.bfloat16


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


# (Cin, Cout, k, stride, Hin, count)  -- SURVEY.md B.2
CONVS = [(3, 64, 7, 2, 224, 1), (64, 64, 1, 1, 56, 1), (64, 64, 3, 1, 56, 3), (64, 256, 1, 1, 56, 4), (256, 64, 1, 1, 56, 2),
         (256, 128, 1, 1, 56, 1), (128, 128, 3, 2, 56, 1), (128, 512, 1, 1, 28, 4), (256, 512, 1, 2, 56, 1),
         (512, 128, 1, 1, 28, 3), (128, 128, 3, 1, 28, 3), (512, 256, 1, 1, 28, 1), (256, 256, 3, 2, 28, 1),
         (256, 1024, 1, 1, 14, 6), (512, 1024, 1, 2, 28, 1), (1024, 256, 1, 1, 14, 5), (256, 256, 3, 1, 14, 5),
         (1024, 512, 1, 1, 14, 1), (512, 512, 3, 2, 14, 1), (512, 2048, 1, 1, 7, 3), (1024, 2048, 1, 2, 14, 1),
         (2048, 512, 1, 1, 7, 2), (512, 512, 3, 1, 7, 2)]
def main():
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    print(f"{'layer':34s} {'cnt':>3s} | {'fwd us':>8s} {'TF/s':>6s} {'GB/s':>6s} | {'dgrad us':>8s} {'TF/s':>6s} | {'wgrad us':>8s} {'TF/s':>6s}")
    for (C, KO, k, s, H, cnt) in CONVS:
        pad = {1: 0, 3: 1, 7: 3}[k]
        OH = (H + 2 * pad - k) // s + 1
        stem = k == 7
        if stem:      # what the model runs: 4-channel pixels with a 3-pixel zero frame, 7x8 "valid" filter
            x = torch.randn(B, H + 6, H + 6, 4, device="cuda").to(dt)
            w = (torch.randn(KO, 7, 8, 4, device="cuda") / 147 ** 0.5).to(dt)
            pad, C = 0, 4
        else:
            x = torch.randn(B, H, H, C, device="cuda").to(dt)
            w = (torch.randn(KO, k, k, C, device="cuda") / (k * k * C) ** 0.5).to(dt)
        wt = w.permute(3, 1, 2, 0).contiguous()
        dy = torch.randn(B, OH, OH, KO, device="cuda").to(dt)
        dw = torch.zeros(w.shape, device="cuda")
        flops = 2.0 * B * OH * OH * KO * (147 if stem else k * k * C)
        gemm = (k == 1 and s == 1)
        if gemm:
            f = lambda: ops.gemm_nt(x.view(-1, C), w.view(KO, C))
            d = lambda: ops.gemm_nt(dy.view(-1, KO), wt.view(C, KO))
            g = lambda: ops.gemm_tn_acc(dy.view(-1, KO), x.view(-1, C), dw.view(KO, C))
        else:
            f = lambda: ops.conv2d_fwd(x, w, s, pad)
            d = lambda: ops.conv2d_dgrad(dy, wt, x.shape, s, pad)
            g = lambda: ops.conv2d_wgrad(x, dy, dw, s, pad)
        tf_, td, tg = timeit(f), (timeit(d) if not stem else 0.0), timeit(g)
        byts = (x.numel() + dy.numel() + w.numel()) * 2
        print(f"conv {C:4d}->{KO:4d} k{k} s{s} @{H:3d}        {cnt:3d} | {tf_*1e6:8.1f} {flops/tf_/1e12:6.0f} {byts/tf_/1e9:6.0f} | "
              f"{td*1e6:8.1f} {(flops/td/1e12 if td else 0):6.0f} | {tg*1e6:8.1f} {flops/tg/1e12:6.0f}", flush=True)
        tot["fwd"] += cnt * tf_; tot["dgrad"] += cnt * td; tot["wgrad"] += cnt * tg
    print(f"ResNet-50 conv totals per step: fwd {tot['fwd']*1e3:.2f} ms, dgrad {tot['dgrad']*1e3:.2f} ms, wgrad {tot['wgrad']*1e3:.2f} ms")
    # text head GEMMs (per head; x2 heads)
    T, S, H, F, V = 30, 49, 1024, 4096, 10000
    GEMMS = [("vis_proj", B * S, H, 2048), ("self in_proj", B * T, 3 * H, H), ("out_proj/q", B * T, H, H),
             ("kv_proj", B * S, 2 * H, H), ("ffn1", B * T, F, H), ("ffn2", B * T, H, F), ("vocab", B * T, V, H)]
    tt = 0.0
    for (name, M, N, K) in GEMMS:
        a = torch.randn(M, K, device="cuda").to(dt); b = torch.randn(N, K, device="cuda").to(dt)
        bt = b.t().contiguous(); dyy = torch.randn(M, N, device="cuda").to(dt); dwt = torch.zeros(N, K, device="cuda")
        fl = 2.0 * M * N * K
        tf_ = timeit(lambda: ops.gemm_nt(a, b)); td = timeit(lambda: ops.gemm_nt(dyy, bt)); tg = timeit(lambda: ops.gemm_tn_acc(dyy, a, dwt))
        print(f"gemm {name:14s} M={M:6d} N={N:5d} K={K:5d} | fwd {tf_*1e6:7.1f} us {fl/tf_/1e12:5.0f} TF/s | dgrad {td*1e6:7.1f} us {fl/td/1e12:5.0f} | wgrad {tg*1e6:7.1f} us {fl/tg/1e12:5.0f}", flush=True)
        mult = 3 if name == "out_proj/q" else 1
        tt += mult * (tf_ + td + tg)
    print(f"text-head GEMM total per step (2 heads): {2*tt*1e3:.2f} ms")


if __name__ == "__main__":
    main()
