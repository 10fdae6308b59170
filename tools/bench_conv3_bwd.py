"""conv3's backward of a stage-1 Bottleneck at bs = 256 (M = 802 816 rows, 256 -> 64 channels): the fused streaming kernel
(csrc/conv3_bwd.hip) against the three launches it replaces -- results compared, each timed with events on one stream.

    python tools/bench_conv3_bwd.py [--batch 256]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from virtex_amd import ops  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    K, N, H = 256, 64, 56
    M = a.batch * H * H
    g = torch.Generator(device=dev).manual_seed(0)
    dt = torch.bfloat16
    dz = (torch.randn(M, K, device=dev, generator=g) * (torch.rand(M, K, device=dev, generator=g) > 0.4)).to(dt)
    x3 = (0.8 * torch.randn(M, K, device=dev, generator=g) + 0.3 * torch.randn(K, device=dev, generator=g)).to(dt)
    x2 = (0.9 * torch.randn(M, N, device=dev, generator=g) + 0.2).to(dt)
    wt = (torch.randn(N, K, device=dev, generator=g) / 16).to(dt)
    gamma3 = 0.5 + torch.rand(K, device=dev, generator=g); gamma2 = 0.5 + torch.rand(N, device=dev, generator=g)
    beta2 = 0.3 * torch.randn(N, device=dev, generator=g)
    mean3 = x3.float().mean(0); rstd3 = (x3.float().var(0, unbiased=False) + 1e-5).rsqrt()
    s1 = dz.float().sum(0); s2 = (dz.float() * ((x3.float() - mean3) * rstd3)).sum(0)
    parts3 = torch.stack([s1, s2]).view(1, 2, K).contiguous()
    st3 = ops.BnStats(parts3, 1, None)
    rm, rv = torch.zeros(N, device=dev), torch.ones(N, device=dev)
    y2, mean2, rstd2 = ops.bn_fwd(x2.view(1, 1, M, N), gamma2, beta2, rm, rv, None, relu=True)
    y2 = y2.view(M, N)
    dg, db = torch.zeros(K, device=dev), torch.zeros(K, device=dev)
    dw_f, dw_r = torch.zeros(K, N, device=dev), torch.zeros(K, N, device=dev)

    def bn2():
        return ops.BnBwd(x2, mean2, rstd2, gamma=gamma2, beta=beta2)

    assert ops.conv3_bwd_fused_supported(dz, wt), "the fused kernel does not take this shape / is switched off"
    dy2, st2, parts, nparts = ops.conv3_bwd_fused(dz, x3, gamma3, mean3, rstd3, dg, db, st3, wt, bn2())
    ops.partials_reduce_acc(parts, nparts, dw_f)
    dx3 = ops.bn_bwd_fused(x3, dz, gamma3, mean3, rstd3, dg, db, st3)
    dy2_r, st2_r = ops.gemm_nt_bnbwd(dx3, wt, bn2())
    ops.gemm_tn_acc(dx3, y2, dw_r)
    torch.cuda.synchronize()
    sums = lambda st: st.parts[: st.strips * 2 * N].view(st.strips, 2, N).double().sum(0)   # noqa: E731
    print(f"M={M} K={K} N={N}: partials {nparts}; dy2 rel {rel(dy2.float(), dy2_r.float()):.2e}, max abs diff "
          f"{(dy2.float() - dy2_r.float()).abs().max().item():.3e}; sums rel {rel(sums(st2), sums(st2_r)):.2e}; dW rel {rel(dw_f, dw_r):.2e}")
    t_apply = timeit(lambda: ops.bn_bwd_fused(x3, dz, gamma3, mean3, rstd3, dg, db, st3))
    t_dgrad = timeit(lambda: ops.gemm_nt_bnbwd(dx3, wt, bn2()))
    t_wgrad = timeit(lambda: ops.gemm_tn_acc(dx3, y2, dw_r))
    t_fused = timeit(lambda: ops.conv3_bwd_fused(dz, x3, gamma3, mean3, rstd3, dg, db, st3, wt, bn2()))
    t_red = timeit(lambda: ops.partials_reduce_acc(parts, nparts, dw_f))
    by_f = 2.0 * M * (2 * K + 2 * N)
    print(f"three launches: bn3 backward (finalize + apply) {t_apply:.1f} us, input gradient + bn2 epilogue {t_dgrad:.1f} us "
          f"(compute stream: {t_apply + t_dgrad:.1f} us), weight gradient {t_wgrad:.1f} us (side stream)")
    print(f"fused: {t_fused:.1f} us incl. the finalize launch = {by_f / t_fused / 1e6:.2f} TB/s over {by_f / 1e6:.0f} MB, "
          f"partials fold {t_red:.1f} us (side stream)")
    # repeatability (races would show as run-to-run differences)
    ref = (dy2.clone(), parts.clone())
    bad = 0
    for _ in range(10):
        d2, _, p2, _ = ops.conv3_bwd_fused(dz, x3, gamma3, mean3, rstd3, dg, db, st3, wt, bn2())
        torch.cuda.synchronize()
        bad += int(not torch.equal(d2, ref[0])) + int(not torch.equal(p2, ref[1]))
    print(f"10 repeats: {bad} differing results")


if __name__ == "__main__":
    main()
