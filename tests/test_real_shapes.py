"""GPU parity of the contraction kernel at the shapes the benchmarked step actually runs (R_50_L1_H1024, 256 images):
one case per distinct shape class, so that the 8-wave 256x128 tile, the XCD-aware tile order, split-K through the
workspace, the stride-2 parity decomposition, the packed stem and the statistics epilogues are checked WHERE THEY RUN,
not only through forced-tile toy problems.  Reference: torch fp32 on the CPU of the SAME bf16 inputs (products of bf16
values are exact in fp32, so the distance is the kernel's accumulation order plus one bf16 rounding of the output)."""
import pytest
import torch
import torch.nn.functional as F

from backends import rel_err, select
from virtex_amd import ops

B = 256
DT = torch.bfloat16
pytestmark = pytest.mark.gpu


def _g(seed):
    return torch.Generator().manual_seed(seed)


def test_pointwise_64_to_256_at_56x56_forward_and_statistics():
    """HBM-bound 1x1 convolution of stage 1 (M = 802,816): forward with the BatchNorm statistics in the epilogue."""
    dev = select("gpu")
    g = _g(1)
    M, K, N = B * 56 * 56, 64, 256
    a = torch.randn(M, K, generator=g).to(DT); w = (torch.randn(N, K, generator=g) / 8).to(DT)
    shift = 0.1 * torch.randn(N, generator=g)
    y, st = ops.gemm_nt(a.to(dev), w.to(dev), bn_shift=shift.to(dev))
    ref = a.float() @ w.float().t()
    assert rel_err(y.float().cpu(), ref) < 5e-3
    # one strip per block row of the tiled kernel, or one per workgroup of the streaming kernel (expand1x1.hip: this shape is its)
    assert st is not None and (st.strips in ((M + 255) // 256, (M + 127) // 128, (M + 63) // 64) or 0 < st.strips <= 512)
    parts = st.parts[: st.strips * 2 * N].view(st.strips, 2, N).double().cpu()
    yq = y.float().cpu().double()
    d = yq - shift.double()
    assert rel_err(parts[:, 0].sum(0), d.sum(0)) < 1e-4 and rel_err(parts[:, 1].sum(0), (d * d).sum(0)) < 1e-4


def test_pointwise_input_gradient_with_fused_batchnorm_backward_at_56x56():
    """conv1's input gradient of a stage-1 block, 64 -> 256 channels, joined with the identity gradient, masked by the
    previous block's output and reduced for that block's bn3 (the new epilogue at the largest M of the step)."""
    dev = select("gpu")
    g = _g(2)
    M, K, N = B * 56 * 56, 64, 256
    dy = torch.randn(M, K, generator=g).to(DT); wt = (torch.randn(N, K, generator=g) / 8).to(DT)
    res = torch.randn(M, N, generator=g).to(DT)
    x = (0.7 * torch.randn(M, N, generator=g) + 0.2).to(DT)
    ymask = torch.relu(torch.randn(M, N, generator=g)).to(DT)
    mean = x.float().mean(0); rstd = (x.float().var(0, unbiased=False) + 1e-5).rsqrt()
    bn = ops.BnBwd(x.to(dev), mean.to(dev), rstd.to(dev), ymask=ymask.to(dev))
    dz, st = ops.gemm_nt_bnbwd(dy.to(dev), wt.to(dev), bn, residual=res.to(dev))
    z = dy.float() @ wt.float().t() + res.float()
    dz_ref = torch.where(ymask.float() > 0, z, torch.zeros_like(z))
    assert rel_err(dz.float().cpu(), dz_ref) < 5e-3
    xh = (x.float() - mean) * rstd
    parts = st.parts[: st.strips * 2 * N].view(st.strips, 2, N).double().cpu()
    # the sums are taken of the STORED (bf16-rounded) gradient -- what bn_bwd_fused combines them with -- so they match
    # the stored tensor to fp32 summation accuracy and the fp32 reference to bf16 rounding of 800k terms per channel
    dzq = dz.float().cpu().double()
    assert rel_err(parts[:, 0].sum(0), dzq.sum(0)) < 1e-4
    assert rel_err(parts[:, 1].sum(0), (dzq * xh.double()).sum(0)) < 1e-4
    assert rel_err(parts[:, 0].sum(0), dz_ref.double().sum(0)) < 5e-3
    assert rel_err(parts[:, 1].sum(0), (dz_ref.double() * xh.double()).sum(0)) < 5e-3
    gamma = 0.5 + torch.rand(N, generator=g)
    dgamma = torch.zeros(N, device=dev); dbeta = torch.zeros(N, device=dev)
    dx = ops.bn_bwd_fused(x.to(dev), dz, gamma.to(dev), mean.to(dev), rstd.to(dev), dgamma, dbeta, st)
    s1, s2 = dz_ref.sum(0), (dz_ref * xh).sum(0)
    dx_ref = gamma * rstd * (dz_ref - s1 / M - xh * s2 / M)
    assert rel_err(dx.float().cpu(), dx_ref) < 1e-2
    assert rel_err(dgamma.cpu(), s2) < 5e-3 and rel_err(dbeta.cpu(), s1) < 5e-3


def test_conv3x3_at_28x28_forward_dgrad_wgrad():
    """MFMA-bound 3x3 convolution of stage 2 (128 -> 128 @ 28x28): forward, input gradient, weight gradient (split-K)."""
    dev = select("gpu")
    g = _g(3)
    C = KO = 128
    x = torch.randn(B, 28, 28, C, generator=g).to(DT)
    w = (torch.randn(KO, 3, 3, C, generator=g) / 34).to(DT)
    dy = torch.randn(B, 28, 28, KO, generator=g).to(DT)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(); wr = w.float().permute(0, 3, 1, 2).requires_grad_()
    yr = F.conv2d(xr, wr, padding=1)
    yr.backward(dy.float().permute(0, 3, 1, 2))
    y = ops.conv2d_fwd(x.to(dev), w.to(dev), 1, 1)
    assert rel_err(y.float().cpu(), yr.detach().permute(0, 2, 3, 1)) < 5e-3
    dx = ops.conv2d_dgrad(dy.to(dev), w.permute(3, 1, 2, 0).contiguous().to(dev), x.shape, 1, 1)
    assert rel_err(dx.float().cpu(), xr.grad.permute(0, 2, 3, 1)) < 5e-3
    dw = ops.conv2d_wgrad(x.to(dev), dy.to(dev), torch.zeros(KO, 3, 3, C, device=dev), 1, 1)
    assert rel_err(dw.cpu(), wr.grad.permute(0, 2, 3, 1)) < 2e-3


def test_stride2_input_gradient_56_to_28_with_fused_batchnorm_backward():
    """The stride-2 3x3 of stage 2's first block: four parity-class launches scatter rows of the 56x56 gradient and
    each writes its own statistics strips (mask recomputed from the BatchNorm input)."""
    dev = select("gpu")
    g = _g(4)
    C = KO = 128
    x_in = (0.7 * torch.randn(B, 56, 56, C, generator=g) + 0.2).to(DT)
    w = (torch.randn(KO, 3, 3, C, generator=g) / 34).to(DT)
    dy = torch.randn(B, 28, 28, KO, generator=g).to(DT)
    xr = torch.zeros(B, C, 56, 56, requires_grad=True)
    F.conv2d(xr, w.float().permute(0, 3, 1, 2), stride=2, padding=1).backward(dy.float().permute(0, 3, 1, 2))
    z = xr.grad.permute(0, 2, 3, 1).reshape(-1, C)
    wt = w.permute(3, 1, 2, 0).contiguous().to(dev)
    plain = ops.conv2d_dgrad(dy.to(dev), wt, x_in.shape, 2, 1)
    assert rel_err(plain.float().cpu().view(-1, C), z) < 5e-3
    xf = x_in.float().view(-1, C)
    mean = xf.mean(0); rstd = (xf.var(0, unbiased=False) + 1e-5).rsqrt()
    gamma = 0.5 + torch.rand(C, generator=g); beta = 0.2 * torch.randn(C, generator=g)
    xh = (xf - mean) * rstd
    dz_ref = torch.where(xh * gamma + beta > 0, z, torch.zeros_like(z))
    bn = ops.BnBwd(x_in.to(dev), mean.to(dev), rstd.to(dev), gamma=gamma.to(dev), beta=beta.to(dev))
    dz, st = ops.conv2d_dgrad(dy.to(dev), wt, x_in.shape, 2, 1, bn=bn)
    # elements within rounding distance of the ReLU threshold may legitimately fall on the other side: compare sums
    assert rel_err(dz.float().cpu().view(-1, C), dz_ref) < 2e-2
    parts = st.parts[: st.strips * 2 * C].view(st.strips, 2, C).double().cpu()
    dzq = dz.float().cpu().view(-1, C).double()
    assert rel_err(parts[:, 0].sum(0), dzq.sum(0)) < 2e-3
    assert rel_err(parts[:, 1].sum(0), (dzq * xh.double()).sum(0)) < 2e-3


def test_stem_weight_gradient_packed_layout():
    """7x7/s2 stem on the packed 4-channel layout (K = 802,816 pixels per slice group, split-K through the workspace)."""
    dev = select("gpu")
    g = _g(5)
    img = torch.randn(B, 3, 224, 224, generator=g)
    a0 = ops.image_to_nhwc(img.to(dev), DT, 4, halo=3)                    # (B, 230, 230, 4) with a zero frame
    dy = torch.randn(B, 112, 112, 64, generator=g).to(DT)
    dwp = torch.zeros(64, 7, 8, 4, device=dev)
    ops.conv2d_wgrad(a0, dy.to(dev), dwp, 2, 0)
    imq = a0[:, 3:-3, 3:-3, :3].float().cpu().permute(0, 3, 1, 2).contiguous()     # the bf16-rounded image
    wr = torch.zeros(64, 3, 7, 7, requires_grad=True)
    F.conv2d(imq, wr, stride=2, padding=3).backward(dy.float().permute(0, 3, 1, 2))
    got = dwp[:, :, :7, :3].permute(0, 3, 1, 2).cpu()
    assert rel_err(got, wr.grad) < 2e-3        # (the gradients of the padding tap / channel are computed and discarded)


def test_tied_vocabulary_projection_7680x10000x1024():
    """The largest GEMM of a text head (N = 10000 is not a multiple of any tile): fp32 logits, and the weight gradient."""
    dev = select("gpu")
    g = _g(6)
    M, N, K = B * 30, 10000, 1024
    h = torch.randn(M, K, generator=g).to(DT); w = (0.02 * torch.randn(N, K, generator=g)).to(DT)
    bias = 0.1 * torch.randn(N, generator=g)
    logits = ops.gemm_nt(h.to(dev), w.to(dev), bias=bias.to(dev), out_f32=True)
    ref = h.float() @ w.float().t() + bias
    assert rel_err(logits.cpu(), ref) < 1e-4
    d = (0.01 * torch.randn(M, N, generator=g)).to(DT)
    dw = ops.gemm_tn_acc(d.to(dev), h.to(dev), torch.zeros(N, K, device=dev))
    assert rel_err(dw.cpu(), d.float().t() @ h.float()) < 1e-3


def _generation():
    from virtex_amd import _lib
    return _lib.lib().vtx_last_contraction_generation()


def test_buffer_addressing_at_the_top_of_its_range_and_the_fallback_beyond_it():
    """The DMA kernel addresses operands with 32-bit byte offsets (buffer descriptors): an operand just under the 2 GB
    limit exercises offsets close to 2^31 (where the out-of-range marker 0x80000000 begins), one above the limit must be
    taken by the register-staged kernel instead of wrapping.  Both against torch on the device, rows sampled at the
    start, the middle and the very end of the matrix."""
    dev = select("gpu")
    K, N = 1024, 64
    w = (torch.randn(N, K, generator=_g(11)) / 32).to(DT).to(dev)
    for rows, gen in ((950_000, 2), (1_100_000, 1)):            # 1.95 GB -> DMA kernel; 2.25 GB -> fallback
        a = torch.empty(rows, K, dtype=DT, device=dev)
        a.normal_(generator=None)
        y = ops.gemm_nt(a, w)
        assert _generation() == gen
        for lo in (0, rows // 2 - 128, rows - 256):
            ref = a[lo:lo + 256].float() @ w.float().t()
            assert rel_err(y[lo:lo + 256].float(), ref) < 5e-3
        # weight gradient over the same operand (k-major view of `a`, pixel index up to `rows`)
        dy = torch.randn(rows, N, device=dev).to(DT)
        dw = torch.zeros(N, K, device=dev)
        if rows * K * 2 < 2.0e9:
            ops.gemm_tn_acc(dy, a, dw)
            assert _generation() == 2
            ref = torch.zeros(N, K, device=dev)
            for lo in range(0, rows, 50_000):
                ref += dy[lo:lo + 50_000].float().t() @ a[lo:lo + 50_000].float()
            assert rel_err(dw, ref) < 5e-3
        del a, y, dy, dw
        torch.cuda.empty_cache()


def test_conv3x3_shared_tile_kernel_at_stage_1_and_2_shapes():
    """conv3x3_kernel.h at the shapes it takes in the step (>= 100 000 output pixels: 64 filters on 254-pixel tiles, 128 on
    256 x 128): forward with statistics and input gradient against the generic implicit-GEMM kernel."""
    import ctypes
    from virtex_amd import _lib
    dev = select("gpu")
    for (Bn, H, C) in [(32, 56, 64), (128, 28, 128)]:
        g = _g(H + C)
        x = torch.randn(Bn, H, H, C, generator=g).to(DT).to(dev)
        w = (torch.randn(C, 3, 3, C, generator=g) / (9 * C) ** 0.5).to(DT).to(dev)
        dy = torch.randn(Bn, H, H, C, generator=g).to(DT).to(dev)
        wt = w.permute(3, 1, 2, 0).contiguous()
        out = {}
        for sw in (1, 0):
            _lib.call("vtx_set_switch", b"conv3x3_shared", ctypes.c_int(sw))
            try:
                ops.profile_start()
                y, st = ops.conv2d_fwd(x, w, 1, 1, bn_shift=torch.zeros(C, device=dev))
                dx = ops.conv2d_dgrad(dy, wt, x.shape, 1, 1)
                recs = ops.profile_stop()
            finally:
                _lib.call("vtx_set_switch", b"conv3x3_shared", ctypes.c_int(1))
            sums = st.parts[: st.strips * 2 * C].view(st.strips, 2, C).sum(0).cpu()
            out[sw] = (y.float().cpu(), dx.float().cpu(), sums, sum(r["launches"] for r in recs if "Conv3x3SharedA" in r["name"]))
        assert out[1][3] == 2 and out[0][3] == 0
        assert rel_err(out[1][0], out[0][0]) < 2e-3 and rel_err(out[1][1], out[0][1]) < 2e-3
        assert rel_err(out[1][2], out[0][2]) < 2e-3
