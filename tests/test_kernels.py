"""Per-kernel parity: HIP kernel (through the C ABI) vs the torch fp32 op it replaces.
Runs on the CPU fiber emulator (`emu`) and on a real MI355X (`gpu`)."""
import os

import pytest
import torch
import torch.nn.functional as F

from backends import BACKENDS, rel_err, select, tol
from virtex_amd import _lib, ops

DTYPES = [torch.float32, torch.bfloat16]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,H,with_y", [(7, 128, True), (33, 1024, True), (5, 2048, False), (9, 64, True)])
def test_layernorm_residual(backend, dtype, rows, H, with_y):
    dev = select(backend)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(rows, H, generator=g).to(dtype)
    y = torch.randn(rows, H, generator=g).to(dtype) if with_y else None
    gamma = (0.5 + torch.rand(H, generator=g))
    beta = 0.1 * torch.randn(H, generator=g)
    dout = torch.randn(rows, H, generator=g).to(dtype)

    xr = x.float().requires_grad_()
    yr = y.float().requires_grad_() if with_y else None
    gr, br = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    z = xr + yr if with_y else xr
    ref = F.layer_norm(z, (H,), gr, br, 1e-5)
    ref.backward(dout.float())

    xd, yd = x.to(dev), (y.to(dev) if with_y else None)
    out, mean, rstd = ops.layernorm_residual_fwd(xd, yd, gamma.to(dev), beta.to(dev), 1e-5)
    assert torch.allclose(out.float().cpu(), ref.detach(), **tol(dtype))
    dgamma = torch.zeros(H, device=dev); dbeta = torch.zeros(H, device=dev)
    dz, dy = ops.layernorm_residual_bwd(xd, yd, gamma.to(dev), mean, rstd, dout.to(dev), dgamma, dbeta)
    assert rel_err(dz.float().cpu(), xr.grad) < (2e-5 if dtype == torch.float32 else 1e-2)
    assert rel_err(dgamma.cpu(), gr.grad) < (2e-5 if dtype == torch.float32 else 1e-2)
    assert rel_err(dbeta.cpu(), br.grad) < (2e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("backend", BACKENDS)
def test_layernorm_dropout_statistics(backend):
    dev = select(backend)
    rows, H, p = 64, 1024, 0.1
    x = torch.zeros(rows, H, device=dev)
    y = torch.ones(rows, H, device=dev)
    gamma, beta = torch.ones(H, device=dev), torch.zeros(H, device=dev)
    # recover the keep-mask through backward: dy = mask * dz / (1-p)
    out, mean, rstd = ops.layernorm_residual_fwd(x, y, gamma, beta, 1e-5, p, seed=123)
    dg, db = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    dout = torch.randn(rows, H, device=dev)
    dz, dy = ops.layernorm_residual_bwd(x, y, gamma, mean, rstd, dout, dg, db, p, seed=123)
    keep = (dy != 0) | (dz == 0)
    rate = keep.float().mean().item()
    assert abs(rate - (1 - p)) < 0.01
    assert torch.allclose(dy[keep], dz[keep] / (1 - p), rtol=1e-5, atol=1e-7)
    # a different seed gives a different mask; the same seed reproduces it
    _, dy2 = ops.layernorm_residual_bwd(x, y, gamma, mean, rstd, dout, dg, db, p, seed=124)
    assert ((dy2 != 0) != (dy != 0)).float().mean().item() > 0.05
    _, dy3 = ops.layernorm_residual_bwd(x, y, gamma, mean, rstd, dout, dg, db, p, seed=123)
    assert torch.equal(dy3, dy)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N", [(96, 128), (130, 200), (37, 72)])
def test_dropout_bwd_replays_the_mask_of_the_gemm_epilogue(backend, dtype, M, N):
    """vtx_dropout_bwd(seed) keeps exactly the elements the GEMM epilogue kept under the same seed (the join
    x + dropout(y) of a pre-norm sub-layer and its backward), scaled by 1 / (1 - p); ragged sizes included."""
    dev = select(backend)
    p, seed = 0.25, 4242
    a = torch.randn(M, 64, device=dev).to(dtype)
    b = torch.randn(N, 64, device=dev).to(dtype)
    bias = torch.full((N,), 100.0, device=dev)                   # no output element is zero before the mask
    y = ops.gemm_nt(a, b, bias=bias, p_drop=p, seed=seed)
    kept = y != 0
    assert abs(kept.float().mean().item() - (1 - p)) < 0.03
    dx = (torch.randn(M, N, device=dev).abs() + 0.5).to(dtype)
    dy = ops.dropout_bwd(dx, p, seed)
    assert torch.equal(dy != 0, kept)
    assert torch.allclose(dy[kept].float(), dx[kept].float() / (1 - p), **tol(dtype))
    assert ops.dropout_bwd(dx, 0.0, seed) is dx


def _gemm_tol(dtype, K):
    # fp32 MFMA is an exact fmaf chain; bf16 inputs are exact products accumulated in fp32,
    # so against the fp32 product of the SAME (already rounded) inputs both are tight.
    return 2e-5 if dtype == torch.float32 else 1e-2


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(98, 64, 64), (130, 200, 96), (257, 128, 40), (64, 1000, 128), (300, 72, 256)])
def test_gemm_nt(backend, dtype, M, N, K):
    dev = select(backend)
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(dtype)
    b = torch.randn(N, K, generator=g).to(dtype)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g).to(dtype)
    ref_pre = a.float() @ b.float().t() * 0.5 + bias
    ref = F.gelu(ref_pre) + res.float()
    out, pre = ops.gemm_nt(a.to(dev), b.to(dev), bias.to(dev), res.to(dev), act=ops.ACT_GELU,
                           want_preact=True, alpha=0.5)
    assert rel_err(pre.float().cpu(), ref_pre) < _gemm_tol(dtype, K)
    assert rel_err(out.float().cpu(), ref) < _gemm_tol(dtype, K)
    # asymmetric check without epilogue (catches transposes)
    out2 = ops.gemm_nt(a.to(dev), b.to(dev))
    assert rel_err(out2.float().cpu(), a.float() @ b.float().t()) < _gemm_tol(dtype, K)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,split", [(64, 64, 50, 1), (136, 200, 300, 3), (1000, 128, 77, 0), (72, 264, 513, 4)])
def test_gemm_tn_acc(backend, dtype, M, N, K, split):
    dev = select(backend)
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(K, M, generator=g).to(dtype)
    b = torch.randn(K, N, generator=g).to(dtype)
    c0 = torch.randn(M, N, generator=g)
    out = ops.gemm_tn_acc(a.to(dev), b.to(dev), c0.clone().to(dev), alpha=2.0, split_k=split)
    ref = c0 + 2.0 * a.float().t() @ b.float()
    assert rel_err(out.cpu(), ref) < _gemm_tol(dtype, K)


CONV_CASES = [
    # N, H, W, C, KO, R, S, stride, pad
    (2, 9, 9, 16, 32, 3, 3, 1, 1),
    (2, 10, 10, 16, 16, 3, 3, 2, 1),
    (3, 8, 8, 32, 64, 1, 1, 2, 0),
    (2, 8, 8, 64, 16, 1, 1, 1, 0),
    (2, 22, 22, 8, 64, 7, 7, 2, 3),
    (1, 7, 7, 128, 128, 3, 3, 1, 1),
    # channel counts that are multiples of the 32-deep K step: the DMA kernel with buffer-descriptor addressing
    # (tap-uniform K steps, validity masks, the parity-decomposed stride-2 gradient, the counter-based pixel gather)
    (2, 9, 11, 32, 64, 3, 3, 1, 1),
    (2, 10, 12, 64, 32, 3, 3, 2, 1),
    (3, 6, 6, 32, 32, 3, 3, 1, 0),
    (2, 5, 7, 64, 64, 2, 3, 1, 1),
]
# the cases whose bf16 forward / input gradient must run on the DMA kernel (every bf16 weight gradient does)
def _dma_expected(case):
    N, H, W, C, KO, R, S, stride, pad = case
    fwd = C % 32 == 0
    dgrad = KO % 32 == 0 and (stride == 1 or (stride == 2 and H % 2 == 0 and W % 2 == 0 and R <= 4 and S <= 4))
    return fwd, dgrad


def _generation():
    from virtex_amd import _lib
    return _lib.lib().vtx_last_contraction_generation()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_dgrad_wgrad(backend, dtype, case):
    dev = select(backend)
    N, H, W, C, KO, R, S, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, H, W, C, generator=g).to(dtype)
    w = (torch.randn(KO, R, S, C, generator=g) / (R * S * C) ** 0.5).to(dtype)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_()
    wr = w.float().permute(0, 3, 1, 2).requires_grad_()
    yr = F.conv2d(xr, wr, stride=stride, padding=pad)
    dy = torch.randn(yr.shape, generator=g).permute(0, 2, 3, 1).contiguous().to(dtype)
    yr.backward(dy.float().permute(0, 3, 1, 2))
    e = 2e-5 if dtype == torch.float32 else 1e-2

    bf = dtype == torch.bfloat16
    dma_fwd, dma_dgrad = _dma_expected(case)
    y = ops.conv2d_fwd(x.to(dev), w.to(dev), stride, pad)
    assert _generation() == (2 if bf and dma_fwd else 1)
    assert y.shape == dy.shape
    assert rel_err(y.float().cpu(), yr.detach().permute(0, 2, 3, 1)) < e
    wt = w.permute(3, 1, 2, 0).contiguous()
    dx = ops.conv2d_dgrad(dy.to(dev), wt.to(dev), x.shape, stride, pad)
    assert _generation() == (2 if bf and dma_dgrad else 1)
    assert rel_err(dx.float().cpu(), xr.grad.permute(0, 2, 3, 1)) < e
    dw0 = torch.randn(KO, R, S, C, generator=g)
    dw = ops.conv2d_wgrad(x.to(dev), dy.to(dev), dw0.clone().to(dev), stride, pad)
    assert _generation() == (2 if bf else 1)
    assert rel_err(dw.cpu() - dw0, wr.grad.permute(0, 2, 3, 1)) < 2 * e
    for split in (2, 3):                     # split-K slices start their pixel counters in the middle of the batch
        dw = ops.conv2d_wgrad(x.to(dev), dy.to(dev), dw0.clone().to(dev), stride, pad, split_k=split)
        assert rel_err(dw.cpu() - dw0, wr.grad.permute(0, 2, 3, 1)) < 2 * e


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,H,W,C,relu,res", [(2, 7, 7, 64, True, False), (3, 5, 6, 256, True, True),
                                              (2, 4, 4, 2048, False, False), (4, 9, 9, 16, True, True)])
def test_batchnorm(backend, dtype, N, H, W, C, relu, res):
    dev = select(backend)
    g = torch.Generator().manual_seed(N * C + H)
    x = (1.5 * torch.randn(N, H, W, C, generator=g) + 0.3).to(dtype)
    r = torch.randn(N, H, W, C, generator=g).to(dtype) if res else None
    gamma = 0.5 + torch.rand(C, generator=g); beta = 0.1 * torch.randn(C, generator=g)
    rm0 = 0.1 * torch.randn(C, generator=g); rv0 = 0.5 + torch.rand(C, generator=g)
    dy = torch.randn(N, H, W, C, generator=g).to(dtype)

    xr = x.float().permute(0, 3, 1, 2).requires_grad_()
    rr = r.float().permute(0, 3, 1, 2).requires_grad_() if res else None
    gr, br = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    rm, rv = rm0.clone(), rv0.clone()
    yr = F.batch_norm(xr, rm, rv, gr, br, True, 0.1, 1e-5)
    if res:
        yr = yr + rr
    if relu:
        yr = F.relu(yr)
    yr.backward(dy.float().permute(0, 3, 1, 2))

    rmd, rvd = rm0.clone().to(dev), rv0.clone().to(dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    if relu and dtype == torch.bfloat16:
        y, mean, rstd, bits = ops.bn_fwd(x.to(dev), gamma.to(dev), beta.to(dev), rmd, rvd, nbt, relu=relu,
                                         residual=r.to(dev) if res else None, want_bits=True)
        assert torch.equal(bits.cpu(), _pack_mask_bits(y.float().cpu()))     # the bit mask IS (y > 0)
    else:
        y, mean, rstd = ops.bn_fwd(x.to(dev), gamma.to(dev), beta.to(dev), rmd, rvd, nbt, relu=relu,
                                   residual=r.to(dev) if res else None)
    e = 3e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(y.float().cpu(), yr.detach().permute(0, 2, 3, 1)) < e
    assert rel_err(rmd.cpu(), rm) < 1e-4 and rel_err(rvd.cpu(), rv) < 1e-4 and int(nbt) == 1
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dx, dz = ops.bn_bwd(x.to(dev), dy.to(dev), y if relu else None, gamma.to(dev), mean, rstd, dg, db,
                        want_dz=True)
    if relu and not res:   # same result with the mask recomputed from x instead of read from y
        dg2, db2 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        dx2 = ops.bn_bwd(x.to(dev), dy.to(dev), None, gamma.to(dev), mean, rstd, dg2, db2, relu_beta=beta.to(dev))
        assert rel_err(dx2.float().cpu(), dx.float().cpu()) < (1e-6 if dtype == torch.float32 else 2e-2)
        assert rel_err(dg2.cpu(), dg.cpu()) < (1e-6 if dtype == torch.float32 else 2e-2)
    assert rel_err(dx.float().cpu(), xr.grad.permute(0, 2, 3, 1)) < (2e-4 if dtype == torch.float32 else 2e-2)
    assert rel_err(dg.cpu(), gr.grad) < (2e-4 if dtype == torch.float32 else 2e-2)
    assert rel_err(db.cpu(), br.grad) < (2e-4 if dtype == torch.float32 else 2e-2)
    if res:
        assert rel_err(dz.float().cpu(), rr.grad.permute(0, 2, 3, 1)) < e


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,H,W,C", [(2, 8, 8, 64), (1, 7, 9, 16), (2, 14, 14, 8)])
def test_maxpool(backend, dtype, N, H, W, C):
    dev = select(backend)
    g = torch.Generator().manual_seed(H * W)
    x = F.relu(torch.randn(N, H, W, C, generator=g)).to(dtype)  # ties at 0, like post-ReLU activations
    xr = x.float().permute(0, 3, 1, 2).requires_grad_()
    yr = F.max_pool2d(xr, 3, 2, 1)
    dy = torch.randn(yr.shape, generator=g).permute(0, 2, 3, 1).contiguous().to(dtype)
    yr.backward(dy.float().permute(0, 3, 1, 2))
    y, arg = ops.maxpool_fwd(x.to(dev))
    assert torch.equal(y.float().cpu(), yr.detach().permute(0, 2, 3, 1))
    dx = ops.maxpool_bwd(dy.to(dev), arg, x.shape)
    assert rel_err(dx.float().cpu(), xr.grad.permute(0, 2, 3, 1)) < (1e-6 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", DTYPES)
def test_prep_kernels(backend, dtype):
    dev = select(backend)
    g = torch.Generator().manual_seed(3)
    img = torch.randn(2, 3, 6, 5, generator=g)
    out = ops.image_to_nhwc(img.to(dev), dtype, 8).float().cpu()
    assert torch.equal(out[..., :3], img.permute(0, 2, 3, 1).to(dtype).float()) and out[..., 3:].abs().max() == 0
    w = torch.randn(40, 9, 3, generator=g)
    wp, wt = ops.weight_prep(w.to(dev), dtype, cpad=8)
    ref = torch.zeros(40, 9, 8); ref[..., :3] = w
    assert torch.equal(wp.float().cpu(), ref.to(dtype).float())
    assert torch.equal(wt.float().cpu(), ref.permute(2, 1, 0).to(dtype).float())
    w2 = torch.randn(100, 72, generator=g)
    wp, wt = ops.weight_prep(w2.to(dev), dtype)
    assert torch.equal(wp.float().cpu().squeeze(1), w2.to(dtype).float())
    assert torch.equal(wt.float().cpu().squeeze(1), w2.t().to(dtype).float())
    assert torch.equal(ops.cast_from_f32(w2.to(dev), dtype).float().cpu(), w2.to(dtype).float())


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,T,H,V", [(3, 12, 128, 1000), (5, 30, 1024, 300)])
def test_embedding(backend, dtype, B, T, H, V):
    dev = select(backend)
    g = torch.Generator().manual_seed(B + T)
    tokens = torch.randint(1, V, (B, T), generator=g)
    tokens[0, T // 2:] = 0  # padding tail
    tokens[1, 3] = 0
    words = 0.02 * torch.randn(V, H, generator=g); words[0] = 0.5  # non-zero pad row: must not matter
    pos = 0.02 * torch.randn(30, H, generator=g)
    gamma = 0.5 + torch.rand(H, generator=g); beta = 0.1 * torch.randn(H, generator=g)
    dout = torch.randn(B, T, H, generator=g).to(dtype)

    wr, pr, gr, br = (t.clone().requires_grad_() for t in (words, pos, gamma, beta))
    e = F.embedding(tokens, wr, padding_idx=0) + pr[:T].unsqueeze(0)
    ref = F.layer_norm(e, (H,), gr, br, 1e-8) * (tokens != 0).unsqueeze(-1).float()
    ref.backward(dout.float())

    td = tokens.to(dev)
    out, mean, rstd = ops.embedding_fwd(td, words.to(dev), pos.to(dev), gamma.to(dev), beta.to(dev), dtype)
    assert torch.allclose(out.float().cpu(), ref.detach(), **tol(dtype))
    dw, dp = torch.zeros(V, H, device=dev), torch.zeros(30, H, device=dev)
    dg, db = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    ops.embedding_bwd(td, words.to(dev), pos.to(dev), gamma.to(dev), mean, rstd, dout.to(dev), dw, dp, dg, db)
    tl = 5e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(dw.cpu(), wr.grad) < tl and dw[0].abs().max() == 0
    assert rel_err(dp.cpu(), pr.grad) < tl
    assert rel_err(dg.cpu(), gr.grad) < tl and rel_err(db.cpu(), br.grad) < tl


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,A,T,S,causal", [(2, 4, 12, 12, True), (3, 2, 30, 49, False), (2, 16, 30, 30, True),
                                            # beyond the tuned envelope (T <= 32, S <= 56): the general kernels -- the 8 x 8 grid of
                                            # 256 x 256 images, the 12 x 12 grid of 384 x 384, captions of 48 tokens, an odd grid
                                            (2, 2, 30, 64, False), (1, 2, 30, 144, False), (2, 2, 48, 48, True), (1, 1, 40, 99, False)])
def test_attention(backend, dtype, B, A, T, S, causal):
    dev = select(backend)
    g = torch.Generator().manual_seed(T + S)
    Hd = A * 64
    if causal:   # self-attention: packed qkv projection output, ragged key lengths
        qkv = torch.randn(B * T, 3 * Hd, generator=g).to(dtype)
        q, k, v = qkv[:, :Hd], qkv[:, Hd:2 * Hd], qkv[:, 2 * Hd:]
        lengths = torch.randint(2, T + 1, (B,), generator=g); lengths[0] = T
    else:
        q = torch.randn(B * T, Hd, generator=g).to(dtype)
        kv = torch.randn(B * S, 2 * Hd, generator=g).to(dtype)
        k, v = kv[:, :Hd], kv[:, Hd:]
        lengths = None
    dout = torch.randn(B * T, Hd, generator=g).to(dtype)

    def heads(t, L):
        return t.float().reshape(B, L, A, 64).transpose(1, 2)
    qr, kr, vr = (heads(q, T).requires_grad_(), heads(k, S).requires_grad_(), heads(v, S).requires_grad_())
    mask = torch.zeros(B, 1, T, S)
    if causal:
        mask = mask + torch.triu(torch.full((T, S), float("-inf")), 1)
        mask = mask.masked_fill((torch.arange(S)[None, :] >= lengths[:, None])[:, None, None, :], float("-inf"))
    p = torch.softmax(qr @ kr.transpose(-1, -2) / 8.0 + mask, dim=-1)
    oref = (p @ vr).transpose(1, 2).reshape(B * T, Hd)
    oref.backward(dout.float())

    ld = lengths.to(dev) if lengths is not None else None
    if causal:
        qkvd = qkv.to(dev); qd, kd, vd = qkvd[:, :Hd], qkvd[:, Hd:2 * Hd], qkvd[:, 2 * Hd:]
        dqkv = torch.empty_like(qkvd); dq, dk, dv = dqkv[:, :Hd], dqkv[:, Hd:2 * Hd], dqkv[:, 2 * Hd:]
    else:
        qd = q.to(dev); kvd = kv.to(dev); kd, vd = kvd[:, :Hd], kvd[:, Hd:]
        dq = torch.empty_like(qd); dkv = torch.empty_like(kvd); dk, dv = dkv[:, :Hd], dkv[:, Hd:]
    o = ops.attention_fwd(qd, kd, vd, B, A, T, S, causal, ld)
    e = 2e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(o.float().cpu(), oref.detach()) < e
    ops.attention_bwd(qd, kd, vd, dout.to(dev), dq, dk, dv, B, A, T, S, causal, ld)

    def unheads(t, L):
        return t.transpose(1, 2).reshape(B * L, Hd)
    assert rel_err(dq.float().cpu(), unheads(qr.grad, T)) < 2 * e
    assert rel_err(dk.float().cpu(), unheads(kr.grad, S)) < 2 * e
    assert rel_err(dv.float().cpu(), unheads(vr.grad, S)) < 2 * e


@pytest.mark.parametrize("backend", BACKENDS)
def test_attention_refuses_tiles_that_do_not_fit_lds(backend):
    """beyond the tuned envelope one (batch, head) must still fit one workgroup's 160 KiB: refused with the sizes in the message"""
    dev = select(backend)
    q = torch.zeros(30, 64, device=dev)
    kv = torch.zeros(400, 128, device=dev)
    with pytest.raises(_lib.VtxError, match="LDS"):
        ops.attention_fwd(q, kv[:, :64], kv[:, 64:], 1, 1, 30, 400, False, None)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("R,V", [(20, 1000), (58, 10000)])
def test_cross_entropy(backend, dtype, R, V):
    dev = select(backend)
    g = torch.Generator().manual_seed(R)
    logits = 3 * torch.randn(R, V, generator=g)
    targets = torch.randint(1, V, (R,), generator=g)
    targets[::5] = 0
    lr = logits.clone().requires_grad_()
    ref = F.cross_entropy(lr, targets, ignore_index=0)
    (ref * 1.7).backward()
    lc, lse = ops.cross_entropy_fwd(logits.to(dev), targets.to(dev), 0)
    assert abs(lc[0].item() - ref.item()) < 1e-5 * abs(ref.item()) + 1e-6
    assert int(lc[1].item()) == int((targets != 0).sum())
    d = ops.cross_entropy_bwd(logits.to(dev), targets.to(dev), lse, lc, torch.tensor([1.7], device=dev), dtype, 0)
    assert rel_err(d.float().cpu(), lr.grad) < (1e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", DTYPES)
def test_small_helpers(backend, dtype):
    dev = select(backend)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(333, 1000, generator=g).to(dtype)
    out0 = torch.randn(1000, generator=g)
    out = ops.colsum_acc(x.to(dev), out0.clone().to(dev))
    assert rel_err(out.cpu(), out0 + x.float().sum(0)) < (1e-5 if dtype == torch.float32 else 1e-5)
    a = torch.randn(64, 40, generator=g).to(dtype); b = torch.randn(64, 40, generator=g).to(dtype)
    assert torch.allclose(ops.add(a.to(dev), b.to(dev)).float().cpu(), (a.float() + b.float()).to(dtype).float())
    h = torch.randn(64, 40, generator=g).to(dtype); da = torch.randn(64, 40, generator=g).to(dtype)
    hr = h.float().requires_grad_()
    F.gelu(hr).backward(da.float())
    assert rel_err(ops.gelu_bwd(h.to(dev), da.to(dev)).float().cpu(), hr.grad) < (1e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("cand", [0, 1, 2, 3, 4, 5, 6, 11, 12])
def test_contraction_tile_variants_bf16(backend, cand):
    """Every block-tile configuration of the generation-2 bf16 kernel (256x256 ... 64x64; 11, 12: the 64-deep
    K-step variants for row-major operands), on all loader kinds: row-major, k-major (transpose-read) and the
    three conv gathers."""
    import ctypes
    from virtex_amd import _lib
    dev = select(backend)
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(cand)
    try:
        _lib.lib().vtx_set_tile_override(ctypes.c_int(cand))
        M, N, K = 300, 520, (96 if cand < 10 else 200)      # 200: a 64-deep K step with a ragged tail
        a = torch.randn(M, K, generator=g).to(dt); b = torch.randn(N, K, generator=g).to(dt)
        bias = torch.randn(N, generator=g); res = torch.randn(M, N, generator=g).to(dt)
        out = ops.gemm_nt(a.to(dev), b.to(dev), bias.to(dev), res.to(dev), act=ops.ACT_GELU)
        ref = F.gelu(a.float() @ b.float().t() + bias) + res.float()
        assert rel_err(out.float().cpu(), ref) < 1e-2
        out32 = ops.gemm_nt(a.to(dev), b.to(dev), bias.to(dev), out_f32=True)
        assert rel_err(out32.cpu(), a.float() @ b.float().t() + bias) < 1e-2
        at = torch.randn(200, 264, generator=g).to(dt); bt = torch.randn(200, 520, generator=g).to(dt)
        c0 = torch.randn(264, 520, generator=g)
        acc = ops.gemm_tn_acc(at.to(dev), bt.to(dev), c0.clone().to(dev), split_k=2)
        assert rel_err(acc.cpu(), c0 + at.float().t() @ bt.float()) < 1e-2
        # convs: 3x3 stride 1 and stride 2
        for (stride, H) in ((1, 9), (2, 10)):
            x = torch.randn(3, H, H, 32, generator=g).to(dt)
            w = (torch.randn(64, 3, 3, 32, generator=g) / 17).to(dt)
            xr = x.float().permute(0, 3, 1, 2).requires_grad_(); wr = w.float().permute(0, 3, 1, 2).requires_grad_()
            yr = F.conv2d(xr, wr, stride=stride, padding=1)
            dy = torch.randn(yr.shape, generator=g).permute(0, 2, 3, 1).contiguous().to(dt)
            yr.backward(dy.float().permute(0, 3, 1, 2))
            y = ops.conv2d_fwd(x.to(dev), w.to(dev), stride, 1)
            assert rel_err(y.float().cpu(), yr.detach().permute(0, 2, 3, 1)) < 1e-2
            dx = ops.conv2d_dgrad(dy.to(dev), w.permute(3, 1, 2, 0).contiguous().to(dev), x.shape, stride, 1)
            assert rel_err(dx.float().cpu(), xr.grad.permute(0, 2, 3, 1)) < 1e-2
            dw = ops.conv2d_wgrad(x.to(dev), dy.to(dev), torch.zeros(64, 3, 3, 32, device=dev), stride, 1)
            assert rel_err(dw.cpu(), wr.grad.permute(0, 2, 3, 1)) < 2e-2
    finally:
        _lib.lib().vtx_set_tile_override(ctypes.c_int(-1))


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("P,C", [(1000, 64), (777, 256), (130, 2048), (4097, 8)])
def test_batchnorm_apply_adjacent_form_equals_strided_form(backend, dtype, P, C):
    """The flat apply kernels of bn.hip walk the tensor with a thread's vectors adjacent (round 3) or one grid stride apart
    (rounds 1-2; still the form for more than 256 channel vectors per row): outputs, mask bits, dx, dgamma, dbeta bit for
    bit the same for every unroll factor, ragged sizes included (the default sizes of this suite only reach one vector
    per thread)."""
    import ctypes
    from virtex_amd import _lib
    dev = select(backend)
    if dtype == torch.float32 and C % 4:
        pytest.skip("channel vectors")

    def run(adj, unr, residual):
        _lib.call("vtx_set_switch", b"bn_adj", ctypes.c_int(adj)); _lib.call("vtx_set_bn_apply_unroll", ctypes.c_int(unr))
        try:
            g = torch.Generator().manual_seed(1)
            x = torch.randn(P, C, generator=g).to(dtype).to(dev)
            res = torch.randn(P, C, generator=g).to(dtype).to(dev) if residual else None
            gamma = torch.rand(C, generator=g) + 0.5; beta = torch.randn(C, generator=g) * 0.1
            rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
            bits = dtype == torch.bfloat16 and residual
            out = ops.bn_fwd(x.view(1, P, 1, C), gamma.to(dev), beta.to(dev), rm, rv, None, relu=True,
                             residual=res.view(1, P, 1, C) if residual else None, want_bits=bits)
            dz = torch.randn(P, C, generator=g).to(dtype).to(dev)
            dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
            dx = ops.bn_bwd(x.view(1, P, 1, C), dz.view(1, P, 1, C), None, gamma.to(dev), out[1], out[2], dg, db)
            return [out[0].cpu(), out[3].cpu() if bits else None, dx.cpu(), dg.cpu(), db.cpu()]
        finally:
            _lib.call("vtx_set_switch", b"bn_adj", ctypes.c_int(1)); _lib.call("vtx_set_bn_apply_unroll", ctypes.c_int(0))

    for residual in (False, True):
        ref = run(0, 1, residual)
        for unr in (1, 2, 4):
            for a, b in zip(ref, run(1, unr, residual)):
                assert (a is None and b is None) or torch.equal(a, b), (residual, unr)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("case", [(2, 9, 9, 32, 32, 3, 1, 1), (3, 8, 8, 64, 256, 1, 1, 0), (2, 10, 10, 32, 64, 3, 2, 1)])
def test_batchnorm_statistics_fused_in_conv_epilogue(backend, case):
    """bf16: the convolution epilogue emits per-strip sums of (y - shift), (y - shift)^2; BN forward fed
    with them must agree with BN forward that reduces the stored tensor itself."""
    dev = select(backend)
    N, H, W, C, KO, k, stride, pad = case
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, H, W, C, generator=g).to(dt).to(dev)
    w = (torch.randn(KO, k, k, C, generator=g) / (k * k * C) ** 0.5).to(dt).to(dev)
    gamma = (0.5 + torch.rand(KO, generator=g)).to(dev); beta = (0.1 * torch.randn(KO, generator=g)).to(dev)
    shift = (0.3 * torch.randn(KO, generator=g)).to(dev)        # e.g. the running mean
    if k == 1 and stride == 1:
        y, st = ops.gemm_nt(x.view(-1, C), w.view(KO, C), bn_shift=shift)
        y = y.view(N, H, W, KO)
    else:
        y, st = ops.conv2d_fwd(x, w, stride, pad, bn_shift=shift)
    assert st is not None and st.strips > 0
    rm1, rv1 = shift.clone(), torch.ones(KO, device=dev)
    out1, mean1, rstd1 = ops.bn_fwd(y, gamma, beta, rm1, rv1, None, stats=st)
    rm2, rv2 = shift.clone(), torch.ones(KO, device=dev)
    out2, mean2, rstd2 = ops.bn_fwd(y, gamma, beta, rm2, rv2, None)
    # fused statistics see the fp32 accumulators, the stand-alone pass the bf16-rounded tensor
    assert torch.allclose(mean1.cpu(), mean2.cpu(), atol=5e-3, rtol=1e-2)
    assert torch.allclose(rstd1.cpu(), rstd2.cpu(), rtol=1e-2)
    assert rel_err(out1.float().cpu(), out2.float().cpu()) < 2e-2
    assert torch.allclose(rm1.cpu(), rm2.cpu(), atol=1e-3) and torch.allclose(rv1.cpu(), rv2.cpu(), rtol=1e-2)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case,relu,res", [((2, 9, 9, 16, 32, 3, 3, 1, 1), True, False),
                                           ((3, 8, 8, 64, 256, 1, 1, 1, 0), True, True),
                                           ((2, 10, 10, 16, 64, 3, 3, 2, 1), True, False),
                                           ((2, 8, 8, 32, 64, 1, 1, 2, 0), False, False),
                                           ((1, 20, 20, 8, 64, 7, 7, 2, 3), True, False)])
def test_conv2d_infer_folded_batchnorm(backend, dtype, case, relu, res):
    """Eval-mode unit: conv -> BatchNorm(running statistics) -> (+identity) -> ReLU, with the BN folded into the
    weights/bias (vtx_bn_fold) and the tail fused into the convolution epilogue (vtx_conv2d_infer).
    Reference semantics: torchvision Bottleneck in eval mode (visual_backbones.py:68-74)."""
    dev = select(backend)
    N, H, W, C, KO, R, S, stride, pad = case
    g = torch.Generator().manual_seed(sum(case) + 7)
    cin = 3 if R == 7 else C                      # the stem: 3 input channels zero-padded to C=8
    x = torch.randn(N, H, W, C, generator=g)
    if cin != C:
        x[..., cin:] = 0
    x = x.to(dtype)
    w32 = torch.randn(KO, R * S, cin, generator=g) / (R * S * cin) ** 0.5
    gamma = 0.5 + torch.rand(KO, generator=g); beta = 0.2 * torch.randn(KO, generator=g)
    rm = 0.3 * torch.randn(KO, generator=g); rv = 0.5 + torch.rand(KO, generator=g)
    eps = 1e-5
    conv = F.conv2d(x.float()[..., :cin].permute(0, 3, 1, 2), w32.view(KO, R, S, cin).permute(0, 3, 1, 2), stride=stride, padding=pad)
    ref = F.batch_norm(conv, rm.clone(), rv.clone(), gamma, beta, training=False, eps=eps)
    r = None
    if res:
        r = torch.randn(ref.shape, generator=g).permute(0, 2, 3, 1).contiguous().to(dtype)
        ref = ref + r.float().permute(0, 3, 1, 2)
    if relu:
        ref = ref.relu()
    w, bias = ops.bn_fold(w32.to(dev), gamma.to(dev), beta.to(dev), rm.to(dev), rv.to(dev), eps, dtype, cpad=C)
    assert w.shape == (KO, R * S, C) and bias.shape == (KO,)
    sc = gamma / torch.sqrt(rv + eps)
    assert rel_err(bias.cpu(), beta - rm * sc) < 1e-5
    y = ops.conv2d_infer(x.to(dev), w.view(KO, R, S, C), bias, stride, pad, relu=relu,
                         residual=r.to(dev) if res else None)
    e = 2e-5 if dtype == torch.float32 else 1.5e-2
    assert rel_err(y.float().cpu(), ref.permute(0, 2, 3, 1)) < e
    if relu:
        assert (y.float() >= 0).all()


@pytest.mark.parametrize("backend", BACKENDS)
def test_contraction_profiler_counts_launches_flops_bytes(backend):
    """vtx_profile_start/stop: one class per kernel instantiation; algorithmic FLOPs 2*M*N*K and bytes
    (operands + output once) summed over the launches made while profiling is on."""
    dev = select(backend)
    g = torch.Generator().manual_seed(3)
    a = torch.randn(300, 64, generator=g).to(torch.bfloat16).to(dev)
    b = torch.randn(128, 64, generator=g).to(torch.bfloat16).to(dev)
    ops.gemm_nt(a, b)                         # not profiled
    ops.profile_start()
    ops.gemm_nt(a, b)
    ops.gemm_nt(a, b)
    x = torch.randn(2, 9, 9, 32, generator=g).to(torch.bfloat16).to(dev)
    w = torch.randn(32, 3, 3, 32, generator=g).to(torch.bfloat16).to(dev)
    ops.conv2d_fwd(x, w, 1, 1)
    rec = ops.profile_stop()
    ops.gemm_nt(a, b)                         # not profiled either
    assert len(rec) == 2 and sum(r["launches"] for r in rec) == 3
    gm = [r for r in rec if "ConvFwdA" not in r["name"]][0]
    cv = [r for r in rec if "ConvFwdA" in r["name"]][0]
    assert "PlainKC" in gm["name"] and gm["launches"] == 2
    assert gm["flops"] == 2 * (2.0 * 300 * 128 * 64)
    assert gm["bytes"] == 2 * 2.0 * (300 * 64 + 128 * 64 + 300 * 128)
    assert cv["flops"] == 2.0 * (2 * 9 * 9) * 32 * (9 * 32)
    assert cv["bytes"] == 2.0 * (2 * 9 * 9 * 32 + 32 * 288 + 2 * 9 * 9 * 32)
    assert gm["seconds"] > 0 and cv["seconds"] > 0
    ops.profile_start()
    assert ops.profile_stop() == []           # start resets the counters


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,A,T,S,causal", [(2, 3, 30, 49, False), (3, 2, 30, 30, True), (2, 2, 30, 64, False), (2, 2, 40, 40, True)])
def test_attention_dropout_mask_is_shared_by_forward_and_backward(backend, dtype, B, A, T, S, causal):
    """Attention dropout is a counter-based hash re-evaluated in backward (no mask tensor).  Recover the mask the
    forward pass used by running it with V = identity (then O[i][j] = dropout(P)[i][j]), rebuild the op in torch
    with that mask and check the output and all three gradients -- fp32 scalar kernel and bf16 MFMA kernel."""
    dev = select(backend)
    g = torch.Generator().manual_seed(17 * T + S)
    p_drop, seed, Hd = 0.25, 1234, A * 64
    q = torch.randn(B * T, Hd, generator=g).to(dtype)
    k = torch.randn(B * S, Hd, generator=g).to(dtype)
    v = torch.randn(B * S, Hd, generator=g).to(dtype)
    dout = torch.randn(B * T, Hd, generator=g).to(dtype)
    eye = torch.eye(S, 64).repeat(B, A).to(dtype)                       # [B*S, A*64]: V_h = I for every head
    pd = ops.attention_fwd(q.to(dev), k.to(dev), eye.to(dev), B, A, T, S, causal, None, p_drop, seed)
    pd = pd.float().cpu().reshape(B, T, A, 64)[..., :S].permute(0, 2, 1, 3)          # (B, A, T, S) dropped probabilities

    def heads(t, L):
        return t.float().reshape(B, L, A, 64).transpose(1, 2)
    qr, kr, vr = (heads(q, T).requires_grad_(), heads(k, S).requires_grad_(), heads(v, S).requires_grad_())
    mask = torch.triu(torch.full((T, S), float("-inf")), 1) if causal else torch.zeros(T, S)
    p = torch.softmax(qr @ kr.transpose(-1, -2) / 8.0 + mask, dim=-1)
    keep = (pd != 0) | (p.detach() < 1e-6)                 # a kept probability that underflowed to 0 is harmless
    frac = keep[p.detach() > 1e-6].float().mean().item()
    assert 0.65 < frac < 0.85                              # ~ 1 - p_drop
    e = 2e-5 if dtype == torch.float32 else 1.5e-2
    assert rel_err(pd, (p * keep / (1 - p_drop)).detach()) < e
    oref = ((p * keep / (1 - p_drop)) @ vr).transpose(1, 2).reshape(B * T, Hd)
    oref.backward(dout.float())
    o = ops.attention_fwd(q.to(dev), k.to(dev), v.to(dev), B, A, T, S, causal, None, p_drop, seed)
    assert rel_err(o.float().cpu(), oref.detach()) < e
    dq, dk, dv = torch.empty_like(q, device=dev), torch.empty_like(k, device=dev), torch.empty_like(v, device=dev)
    ops.attention_bwd(q.to(dev), k.to(dev), v.to(dev), dout.to(dev), dq, dk, dv, B, A, T, S, causal, None, p_drop, seed)

    def unheads(t, L):
        return t.transpose(1, 2).reshape(B * L, Hd)
    assert rel_err(dq.float().cpu(), unheads(qr.grad, T)) < 2 * e
    assert rel_err(dk.float().cpu(), unheads(kr.grad, S)) < 2 * e
    assert rel_err(dv.float().cpu(), unheads(vr.grad, S)) < 2 * e


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", DTYPES)
def test_batched_weight_preparation_and_cache(backend, dtype):
    """ops.prep_many: one launch for many weights == the per-weight kernel; copies are cached on the parameter and
    refreshed when its version counter or storage moves."""
    dev = select(backend)
    g = torch.Generator().manual_seed(5)
    conv = torch.nn.Parameter(torch.randn(40, 24, 3, 3, generator=g).contiguous(memory_format=torch.channels_last).to(dev))
    stem = torch.nn.Parameter(torch.randn(16, 3, 7, 7, generator=g).contiguous(memory_format=torch.channels_last).to(dev))
    lin = torch.nn.Parameter(torch.randn(70, 50, generator=g).to(dev))
    items = [(conv, None, True), (stem, 8, False), (lin, None, True)]
    assert ops.prep_many(items, dtype) == 3
    assert ops.prep_many(items, dtype) == 0                       # everything current: no launch
    for (p, cpad, want_wt) in items:
        w, wt = ops.prepped(p, dtype, cpad=cpad, want_wt=want_wt)
        w32 = ops._w32_view(p)
        rw, rwt = ops.weight_prep(w32, dtype, cpad=cpad, want_wt=want_wt)
        assert torch.equal(w.float().cpu(), rw.float().cpu())
        if want_wt:
            assert torch.equal(wt.float().cpu(), rwt.float().cpu())
        else:
            assert wt is None
    before = ops.prepped(lin, dtype)[0].float().cpu().clone()
    with torch.no_grad():
        lin.mul_(2.0)                                              # in-place edit bumps the version
    assert ops.prep_many(items, dtype) == 1
    assert torch.allclose(ops.prepped(lin, dtype)[0].float().cpu(), 2 * before, rtol=1e-2)
    # the transposed copy of the stem was not asked for at first: asking later computes just that
    w, wt = ops.prepped(stem, dtype, cpad=8, want_wt=True)
    assert wt is not None and wt.shape == (8, 49, 16)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", DTYPES)
def test_image_u8_preprocessing(backend, dtype):
    """uint8 HWC -> normalised NHWC: albumentations.Normalize's closed form ((x - 255*mean) * (1/(255*std)), fp32) +
    crop window + horizontal flip + channel padding (reference pipeline: transforms.py:85-97, captioning.py:61-64)."""
    dev = select(backend)
    g = torch.Generator().manual_seed(11)
    N, Hs, Ws, size, cpad = 3, 20, 26, 16, 8
    img = torch.randint(0, 256, (N, Hs, Ws, 3), generator=g, dtype=torch.uint8)
    crop = torch.tensor([[0, 0], [10, 4], [3, 2]], dtype=torch.int32)
    flip = torch.tensor([0, 1, 1], dtype=torch.uint8)
    mean = torch.tensor(ops.IMAGENET_COLOR_MEAN); std = torch.tensor(ops.IMAGENET_COLOR_STD)
    out = ops.image_u8_to_nhwc(img.to(dev), dtype, cpad, size=size, crop_xy=crop.to(dev), flip=flip.to(dev))
    ref = torch.zeros(N, size, size, cpad)
    for n in range(N):
        x0, y0 = crop[n].tolist()
        win = img[n, y0:y0 + size, x0:x0 + size].float()
        if flip[n]:
            win = win.flip(1)
        ref[n, :, :, :3] = (win - 255.0 * mean) * (1.0 / (255.0 * std))
    assert out.shape == (N, size, size, cpad) and out.dtype == dtype
    if dtype == torch.float32:
        assert torch.equal(out.cpu(), ref)                       # same fp32 expression: exact
    else:
        assert torch.equal(out.cpu(), ref.to(torch.bfloat16))    # one rounding to bf16
    full = ops.image_u8_to_nhwc(img.to(dev), dtype, cpad)        # no crop / flip: whole image
    assert full.shape == (N, Hs, Ws, cpad) and torch.equal(full[..., 3:].cpu(), torch.zeros(N, Hs, Ws, cpad - 3, dtype=dtype))


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", DTYPES)
def test_packed_stem_geometry(backend, dtype):
    """The 7x7/s2/p3 stem as a "valid" 7x8/s2 convolution on 4-channel pixels with a 3-pixel zero frame: in bf16 a
    16-byte chunk carries two adjacent pixels (conv_common.h).  Forward and weight gradient against torch's 7x7
    convolution of the original 3-channel image."""
    dev = select(backend)
    g = torch.Generator().manual_seed(21)
    N, H, KO = 2, 20, 64
    img = torch.randn(N, 3, H, H, generator=g)
    w7 = torch.randn(KO, 3, 7, 7, generator=g) / 12
    xr, wr = img.clone().requires_grad_(), w7.clone().requires_grad_()
    yr = F.conv2d(xr, wr, stride=2, padding=3)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    a0 = ops.image_to_nhwc(img.to(dev), dtype, 4, halo=3)
    assert a0.shape == (N, H + 6, H + 6, 4)
    wp = torch.zeros(KO, 7, 8, 4)
    wp[:, :, :7, :3] = w7.permute(0, 2, 3, 1)
    w = wp.to(dtype).to(dev)
    y = ops.conv2d_fwd(a0, w, 2, 0)
    assert _generation() == (2 if dtype == torch.bfloat16 else 1)    # bf16: the DMA kernel's filter-row K steps
    e = 2e-5 if dtype == torch.float32 else 1e-2
    assert y.shape == (N, H // 2, H // 2, KO)
    assert rel_err(y.float().cpu(), yr.detach().permute(0, 2, 3, 1)) < e
    dwp = torch.zeros(KO, 7, 8, 4, device=dev)
    ops.conv2d_wgrad(a0, dy.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev), dwp, 2, 0)
    assert rel_err(dwp[:, :, :7, :3].cpu(), wr.grad.permute(0, 2, 3, 1)) < 2 * e
    assert dwp[:, :, 7, :].abs().max().item() >= 0            # the padded tap column exists; its weights are zero


def _bn_bwd_reference(z, x, mean, rstd, gamma, beta, ymask, mode):
    """dz (masked gradient), s1 = sum dz, s2 = sum dz*xhat, dx -- what the fused epilogue + vtx_bn_bwd_fused compute."""
    xh = (x - mean) * rstd
    if mode == "ymask":
        keep = ymask > 0
    elif mode == "remask":
        keep = xh * gamma + beta > 0
    else:
        keep = torch.ones_like(z, dtype=torch.bool)
    dz = torch.where(keep, z, torch.zeros_like(z))
    P = z.shape[0]
    s1, s2 = dz.sum(0), (dz * xh).sum(0)
    dx = gamma * rstd * (dz - s1 / P - xh * s2 / P)
    return dz, s1, s2, dx


def _pack_mask_bits(y):
    """one bit per element, bit e of byte i = (y.flatten()[8 i + e] > 0): the layout vtx_bn_fwd(relu_bits) writes"""
    m = (y.flatten() > 0).to(torch.int32).view(-1, 8)
    return (m * (2 ** torch.arange(8, dtype=torch.int32))).sum(1).to(torch.uint8)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("mode", ["ymask", "ybits", "remask", "none"])
@pytest.mark.parametrize("cand", [-1, 0, 1, 2, 3, 4, 5, 6])
def test_batchnorm_backward_fused_in_gemm_epilogue(backend, mode, cand):
    """bf16: the input-gradient GEMM's epilogue masks its output with the ReLU of the BatchNorm that fed the
    convolution and emits sum dz / sum dz*xhat; vtx_bn_bwd_fused turns them into dx, dgamma, dbeta.  Against a torch
    fp32 evaluation of the same formulas, every block tile (statistics strips = block rows), ragged M, residual."""
    import ctypes
    from virtex_amd import _lib
    dev = select(backend)
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(7 + cand)
    M, N, K = 300, 256, 64
    a = torch.randn(M, K, generator=g).to(dt); b = (torch.randn(N, K, generator=g) / 8).to(dt)
    res = torch.randn(M, N, generator=g).to(dt)
    x = (0.7 * torch.randn(M, N, generator=g) + 0.3).to(dt)
    mean = x.float().mean(0); rstd = (x.float().var(0, unbiased=False) + 1e-5).rsqrt()
    gamma = 0.5 + torch.rand(N, generator=g); beta = 0.2 * torch.randn(N, generator=g)
    ymask = torch.relu(torch.randn(M, N, generator=g)).to(dt)
    z = a.float() @ b.float().t() + res.float()
    bits_mode = mode == "ybits"          # the mask as one bit per element instead of the tensor: identical results
    if bits_mode:
        mode = "ymask"
    dz_r, s1_r, s2_r, dx_r = _bn_bwd_reference(z, x.float(), mean, rstd, gamma, beta, ymask.float(), mode)
    bn = ops.BnBwd(x.to(dev), mean.to(dev), rstd.to(dev), ymask=ymask.to(dev) if (mode == "ymask" and not bits_mode) else None,
                   gamma=gamma.to(dev) if mode == "remask" else None, beta=beta.to(dev) if mode == "remask" else None,
                   ybits=_pack_mask_bits(ymask).to(dev) if bits_mode else None)
    try:
        _lib.lib().vtx_set_tile_override(ctypes.c_int(cand))
        dz, st = ops.gemm_nt_bnbwd(a.to(dev), b.to(dev), bn, residual=res.to(dev))
    finally:
        _lib.lib().vtx_set_tile_override(ctypes.c_int(-1))
    assert st is not None and st.strips > 0
    assert rel_err(dz.float().cpu(), dz_r) < 1e-2
    parts = st.parts[: st.strips * 2 * N].view(st.strips, 2, N).cpu()
    assert rel_err(parts[:, 0].sum(0), s1_r) < 1e-2 and rel_err(parts[:, 1].sum(0), s2_r) < 1e-2
    dgamma = torch.zeros(N, device=dev); dbeta = torch.zeros(N, device=dev)
    dx = ops.bn_bwd_fused(x.to(dev), dz, gamma.to(dev), mean.to(dev), rstd.to(dev), dgamma, dbeta, st)
    assert rel_err(dx.float().cpu(), dx_r) < 2e-2
    assert rel_err(dgamma.cpu(), s2_r) < 1e-2 and rel_err(dbeta.cpu(), s1_r) < 1e-2
    # and it equals the stand-alone path (reduce + mask + apply) on the plain gradient
    plain = ops.gemm_nt(a.to(dev), b.to(dev), residual=res.to(dev))
    dg2 = torch.zeros(N, device=dev); db2 = torch.zeros(N, device=dev)
    dx2 = ops.bn_bwd(x.to(dev), plain, ymask.to(dev) if mode == "ymask" else None, gamma.to(dev), mean.to(dev), rstd.to(dev),
                     dg2, db2, relu_beta=beta.to(dev) if mode == "remask" else None)
    assert rel_err(dx.float().cpu(), dx2.float().cpu()) < 2e-2 and rel_err(dgamma.cpu(), dg2.cpu()) < 1e-2


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("stride,H", [(1, 9), (2, 10)])
def test_batchnorm_backward_fused_in_conv_dgrad_epilogue(backend, stride, H):
    """The 3x3 input-gradient convolution (stride 1, and stride 2 = four parity-class launches writing consecutive
    strips through the row scatter) with the same fusion."""
    dev = select(backend)
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(11 + stride)
    Nb, C, KO = 3, 32, 64
    x_in = (0.7 * torch.randn(Nb, H, H, C, generator=g) + 0.3).to(dt)           # the BatchNorm input = conv input before BN/ReLU
    w = (torch.randn(KO, 3, 3, C, generator=g) / 17).to(dt)
    OH = (H + 2 - 3) // stride + 1
    dy = torch.randn(Nb, OH, OH, KO, generator=g).to(dt)
    xr = torch.zeros(Nb, C, H, H, requires_grad=True)
    F.conv2d(xr, w.float().permute(0, 3, 1, 2), stride=stride, padding=1).backward(dy.float().permute(0, 3, 1, 2))
    z = xr.grad.permute(0, 2, 3, 1).reshape(-1, C)
    xf = x_in.float().view(-1, C)
    mean = xf.mean(0); rstd = (xf.var(0, unbiased=False) + 1e-5).rsqrt()
    gamma = 0.5 + torch.rand(C, generator=g); beta = 0.2 * torch.randn(C, generator=g)
    dz_r, s1_r, s2_r, dx_r = _bn_bwd_reference(z, xf, mean, rstd, gamma, beta, None, "remask")
    bn = ops.BnBwd(x_in.to(dev), mean.to(dev), rstd.to(dev), gamma=gamma.to(dev), beta=beta.to(dev))
    dz, st = ops.conv2d_dgrad(dy.to(dev), w.permute(3, 1, 2, 0).contiguous().to(dev), x_in.shape, stride, 1, bn=bn)
    assert st is not None and st.strips >= (4 if stride == 2 else 1)
    assert rel_err(dz.float().cpu().view(-1, C), dz_r) < 1e-2
    dgamma = torch.zeros(C, device=dev); dbeta = torch.zeros(C, device=dev)
    dx = ops.bn_bwd_fused(x_in.to(dev), dz, gamma.to(dev), mean.to(dev), rstd.to(dev), dgamma, dbeta, st)
    assert rel_err(dx.float().cpu().view(-1, C), dx_r) < 2e-2
    assert rel_err(dgamma.cpu(), s2_r) < 1e-2 and rel_err(dbeta.cpu(), s1_r) < 1e-2


@pytest.mark.parametrize("backend", BACKENDS)
def test_fused_batchnorm_backward_is_refused_cleanly_in_fp32(backend):
    """fp32 (parity mode) does not fuse: strips == 0 and the output is the plain gradient."""
    dev = select(backend)
    g = torch.Generator().manual_seed(5)
    a = torch.randn(70, 32, generator=g); b = torch.randn(64, 32, generator=g); x = torch.randn(70, 64, generator=g)
    bn = ops.BnBwd(x.to(dev), x.mean(0).to(dev), torch.ones(64, device=dev), gamma=torch.ones(64, device=dev), beta=torch.zeros(64, device=dev))
    out, st = ops.gemm_nt_bnbwd(a.to(dev), b.to(dev), bn)
    assert st is None and rel_err(out.cpu(), a @ b.t()) < 1e-4


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("R,V,H", [(150, 1000, 64), (70, 304, 128)])
def test_tied_projection_cross_entropy_without_logits(backend, dtype, R, V, H):
    """vtx_tied_ce_fwd / _bwd against F.cross_entropy(h @ W^T + b) evaluated in fp32 on the same (rounded) operands:
    loss, log-sum-exp per row, the gradient wrt the logits incl. ignored rows and the row-0 (padding index) COLUMN,
    which the tied matrix does receive.  V is not a multiple of any tile."""
    dev = select(backend)
    g = torch.Generator().manual_seed(R + V)
    h = torch.randn(R, H, generator=g).to(dtype); w = (0.3 * torch.randn(V, H, generator=g)).to(dtype)
    bias = 0.2 * torch.randn(V, generator=g)
    tgt = torch.randint(1, V, (R,), generator=g)
    tgt[::7] = 0                                               # ignored rows
    logits = (h.float() @ w.float().t() + bias).requires_grad_()
    ref = F.cross_entropy(logits, tgt, ignore_index=0)
    ref.backward()
    lc, lse = ops.tied_ce_fwd(h.to(dev), w.to(dev), bias.to(dev), tgt.to(dev), 0)
    tol = 2e-5 if dtype == torch.float32 else 2e-4
    assert abs(lc[0].item() - ref.item()) < tol * abs(ref.item())
    assert lc[1].item() == (tgt != 0).sum().item()
    assert torch.allclose(lse.cpu(), torch.logsumexp(logits.detach(), 1), rtol=tol, atol=tol)
    gout = torch.tensor([1.7])
    d = ops.tied_ce_bwd(h.to(dev), w.to(dev), bias.to(dev), tgt.to(dev), lse, lc, gout.to(dev), 0)
    dref = 1.7 * logits.grad
    assert rel_err(d.float().cpu(), dref) < (2e-5 if dtype == torch.float32 else 1e-2)
    assert d.float().cpu()[::7].abs().max().item() == 0.0       # ignored rows contribute nothing
    assert d.float().cpu()[:, 0].abs().sum().item() > 0         # ... but column 0 of the others does


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("cand", [20, 21])
def test_feed_forward_epilogue_on_generation3(backend, cand):
    """bias -> pre-activation copy -> GELU -> dropout (the first GEMM of the decoder's feed-forward layer in training) on the lean
    strip path of the generation-3 kernels: output AND pre-activation equal generation 2 bit for bit (same arithmetic, same
    dropout hash of the element index), interior tiles and a ragged edge."""
    import ctypes
    from virtex_amd import _lib
    dev = select(backend)
    g = torch.Generator().manual_seed(40 + cand)
    M, N, K = 256 * 2 + 24, 256 * 2 + 8, 128
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev); b = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = {}
    try:
        for c in (2, cand):
            _lib.lib().vtx_set_tile_override(ctypes.c_int(c))
            out, pre = ops.gemm_nt(a, b, bias=bias, act=ops.ACT_GELU, want_preact=True, p_drop=0.25, seed=1234)
            assert _generation() == (3 if c >= 20 else 2)
            res[c] = (out.float().cpu(), pre.float().cpu())
    finally:
        _lib.lib().vtx_set_tile_override(ctypes.c_int(-1))
    assert torch.equal(res[cand][1], res[2][1]) and torch.equal(res[cand][0], res[2][0])
    ref_pre = a.float().cpu() @ b.float().cpu().t() + bias.cpu()
    assert rel_err(res[cand][1], ref_pre) < 1e-2
    kept = res[cand][0] != 0
    assert 0.70 < kept.float().mean().item() < 0.80                       # p = 0.25
    assert rel_err(res[cand][0][kept], (F.gelu(ref_pre) / 0.75)[kept]) < 2e-2


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("cand", [20, 21])
def test_tied_projection_cross_entropy_on_generation3(backend, cand):
    """The same two entry points on the generation-3 kernels (forced: 256x256 / 256x128 blocks), interior tiles AND a ragged last
    column / row of tiles: the row log-sum-exp epilogue with its bias values and targets hoisted, and the cross-entropy
    gradient epilogue (g * (softmax - onehot)) on the lean strip path -- equal to the generation-2 result element for element
    (same arithmetic), and to torch within bf16."""
    import ctypes
    from virtex_amd import _lib
    dev = select(backend)
    dtype = torch.bfloat16
    R, V, H = 256 * 2 + 24, 256 * 3 + 40, 128
    g = torch.Generator().manual_seed(cand)
    h = torch.randn(R, H, generator=g).to(dtype); w = (0.3 * torch.randn(V, H, generator=g)).to(dtype)
    bias = 0.2 * torch.randn(V, generator=g)
    tgt = torch.randint(1, V, (R,), generator=g)
    tgt[::7] = 0
    logits = (h.float() @ w.float().t() + bias).requires_grad_()
    ref = F.cross_entropy(logits, tgt, ignore_index=0)
    ref.backward()
    gout = torch.tensor([1.7])
    res = {}
    try:
        for c in (2, cand):                                   # 2: generation-2 128x128 tiles
            _lib.lib().vtx_set_tile_override(ctypes.c_int(c))
            lc, lse = ops.tied_ce_fwd(h.to(dev), w.to(dev), bias.to(dev), tgt.to(dev), 0)
            assert _generation() == (3 if c >= 20 else 2)
            d = ops.tied_ce_bwd(h.to(dev), w.to(dev), bias.to(dev), tgt.to(dev), lse, lc, gout.to(dev), 0)
            assert _generation() == (3 if c >= 20 else 2)
            res[c] = (lc.cpu().clone(), lse.cpu().clone(), d.float().cpu().clone())
    finally:
        _lib.lib().vtx_set_tile_override(ctypes.c_int(-1))
    assert abs(res[cand][0][0].item() - ref.item()) < 2e-4 * abs(ref.item())
    assert torch.allclose(res[cand][1], torch.logsumexp(logits.detach(), 1), rtol=2e-4, atol=2e-4)
    assert rel_err(res[cand][2], 1.7 * logits.grad) < 1e-2
    assert torch.allclose(res[cand][1], res[2][1], rtol=1e-6, atol=1e-6)      # (column groups differ: the fold order of the partials does)
    assert torch.equal(res[cand][2] != 0, res[2][2] != 0) and rel_err(res[cand][2], res[2][2]) < 1e-5


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,H,W,C", [(2, 12, 12, 64), (3, 9, 11, 16)])
def test_stem_tail_maxpool_backward_inside_batchnorm_backward(backend, dtype, N, H, W, C):
    """vtx_bn_bwd_maxpool == maxpool backward -> ReLU mask -> BatchNorm backward of torch, on the stem's layout."""
    dev = select(backend)
    g = torch.Generator().manual_seed(N * H + C)
    x = (torch.randn(N, H, W, C, generator=g) * 0.8 + 0.1).to(dtype)
    gamma = 0.5 + torch.rand(C, generator=g); beta = 0.2 * torch.randn(C, generator=g)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_()
    gr, br = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    y = F.relu(F.batch_norm(xr, None, None, gr, br, training=True, eps=1e-5))
    pooled = F.max_pool2d(y, 3, 2, 1)
    dpool = torch.randn(pooled.shape, generator=g).permute(0, 2, 3, 1).contiguous().to(dtype)
    pooled.backward(dpool.float().permute(0, 3, 1, 2))
    # our forward pieces give the saved statistics, the activations and the argmax
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    yk, mean, rstd = ops.bn_fwd(x.to(dev), gamma.to(dev), beta.to(dev), rm, rv, None, relu=True)
    pk, arg = ops.maxpool_fwd(yk)
    dgamma = torch.zeros(C, device=dev); dbeta = torch.zeros(C, device=dev)
    dx = ops.bn_bwd_maxpool(x.to(dev), dpool.to(dev), arg, gamma.to(dev), beta.to(dev), mean, rstd, dgamma, dbeta)
    if dtype == torch.float32:          # (in bf16 the rounded activations tie differently in the pooling windows than torch's fp32 ones)
        assert rel_err(dx.cpu(), xr.grad.permute(0, 2, 3, 1)) < 2e-4
        assert rel_err(dgamma.cpu(), gr.grad) < 2e-4 and rel_err(dbeta.cpu(), br.grad) < 2e-4
    # and it equals the three-kernel path (max-pool backward, then BatchNorm backward with the mask recomputed from x)
    dstem = ops.maxpool_bwd(dpool.to(dev), arg, tuple(x.shape))
    dg2 = torch.zeros(C, device=dev); db2 = torch.zeros(C, device=dev)
    dx2 = ops.bn_bwd(x.to(dev), dstem, None, gamma.to(dev), mean, rstd, dg2, db2, relu_beta=beta.to(dev))
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert rel_err(dx.float().cpu(), dx2.float().cpu()) < tol
    assert rel_err(dgamma.cpu(), dg2.cpu()) < tol and rel_err(dbeta.cpu(), db2.cpu()) < tol


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("M,K,N", [(4200, 64, 256), (4111, 128, 512), (8192, 64, 512)])
def test_expand1x1_streaming_kernel(backend, M, K, N):
    """The streaming kernel of the write-heavy 1x1 'expand' convolutions (expand1x1.hip; gemm_nt + statistics is routed
    to it unless VIRTEX_AMD_EXPAND1X1=0): output and BatchNorm sums against the tiled kernel's contract -- y = a @ w^T in bf16,
    per-strip sums of (y - shift) and (y - shift)^2 over the STORED values.  Ragged M (rows of the last strip masked),
    both K, one and two 256-column blocks."""
    import os
    if os.environ.get("VIRTEX_AMD_EXPAND1X1", "1") == "0":
        pytest.skip("the streaming kernel is switched off (VIRTEX_AMD_EXPAND1X1=0; read once per process)")
    dev = select(backend)
    g = torch.Generator().manual_seed(M + K + N)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16); w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    shift = 0.2 * torch.randn(N, generator=g)
    ops.profile_start()
    y, st = ops.gemm_nt(a.to(dev), w.to(dev), bn_shift=shift.to(dev))
    rec = ops.profile_stop()
    assert any("expand1x1" in r["name"] and r["launches"] == 1 for r in rec), [r["name"] for r in rec]   # the streaming kernel ran
    assert st is not None and 0 < st.strips <= min(512, M // 64)
    ref = a.float() @ w.float().t()
    assert rel_err(y.float().cpu(), ref) < 5e-3
    parts = st.parts[: st.strips * 2 * N].view(st.strips, 2, N).double().cpu()
    d = y.float().cpu().double() - shift.double()
    assert rel_err(parts[:, 0].sum(0), d.sum(0)) < 1e-5 and rel_err(parts[:, 1].sum(0), (d * d).sum(0)) < 1e-5
    # and the BatchNorm that consumes them
    gamma = 0.5 + torch.rand(N, generator=g); beta = 0.1 * torch.randn(N, generator=g)
    rm1, rv1 = shift.clone().to(dev), torch.ones(N, device=dev)
    out1, mean1, rstd1 = ops.bn_fwd(y.view(1, 1, M, N), gamma.to(dev), beta.to(dev), rm1, rv1, None, stats=st)
    rm2, rv2 = shift.clone().to(dev), torch.ones(N, device=dev)
    out2, mean2, rstd2 = ops.bn_fwd(y.view(1, 1, M, N), gamma.to(dev), beta.to(dev), rm2, rv2, None)
    assert torch.allclose(mean1.cpu(), mean2.cpu(), atol=1e-4, rtol=1e-4) and torch.allclose(rstd1.cpu(), rstd2.cpu(), rtol=1e-3)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,H,W,C", [(2, 12, 12, 64), (3, 9, 11, 16), (1, 7, 7, 64)])
def test_stem_tail_forward_batchnorm_relu_maxpool_in_one_pass(backend, dtype, N, H, W, C):
    """vtx_bn_fwd_maxpool against vtx_bn_fwd followed by vtx_maxpool3x3s2_fwd: pooled values, argmax, saved statistics
    and running statistics BIT-identical (every tap is rounded to the storage type before the comparison)."""
    dev = select(backend)
    g = torch.Generator().manual_seed(N + H + W + C)
    x = (0.8 * torch.randn(N, H, W, C, generator=g) + 0.1).to(dtype).to(dev)
    gamma = (0.5 + torch.rand(C, generator=g)).to(dev); beta = (0.2 * torch.randn(C, generator=g)).to(dev)
    rm1, rv1 = torch.zeros(C, device=dev), torch.ones(C, device=dev); nbt1 = torch.zeros((), dtype=torch.long, device=dev)
    rm2, rv2 = torch.zeros(C, device=dev), torch.ones(C, device=dev); nbt2 = torch.zeros((), dtype=torch.long, device=dev)
    y, mean_a, rstd_a = ops.bn_fwd(x, gamma, beta, rm1, rv1, nbt1, relu=True)
    pool_a, arg_a = ops.maxpool_fwd(y)
    pool_b, arg_b, mean_b, rstd_b = ops.bn_fwd_maxpool(x, gamma, beta, rm2, rv2, nbt2)
    if backend == "emu":
        assert torch.equal(pool_a.cpu(), pool_b.cpu()) and torch.equal(arg_a.cpu(), arg_b.cpu())
    else:       # two separately compiled kernels: allow the odd element whose fp32 pre-rounding value differs in the last bit
        assert (pool_a == pool_b).float().mean().item() > 0.9999 and (arg_a == arg_b).float().mean().item() > 0.999
        assert rel_err(pool_b.float().cpu(), pool_a.float().cpu()) < 1e-3
    assert torch.equal(mean_a.cpu(), mean_b.cpu()) and torch.equal(rstd_a.cpu(), rstd_b.cpu())
    assert torch.equal(rm1.cpu(), rm2.cpu()) and torch.equal(rv1.cpu(), rv2.cpu()) and int(nbt1) == int(nbt2) == 1


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("N,H", [(2, 40), (3, 38), (1, 46)])
def test_stem_streaming_kernel_with_statistics(backend, N, H):
    """stem.hip: the packed 7x8 / stride-2 stem convolution as a streaming kernel (filter resident in LDS, A fragments
    straight from global memory, no K tiling) with the BatchNorm statistics of the STORED output -- against torch's 7x7
    convolution of the 3-channel image, against the tiled kernel on the same operands, and the statistics against the
    stored tensor; ragged strip counts (M not a multiple of 16 x waves), the fused stem tail fed with them."""
    dev = select(backend)
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(5 * N + H)
    KO = 64
    img = torch.randn(N, 3, H, H, generator=g)
    w7 = torch.randn(KO, 3, 7, 7, generator=g) / 12
    a0 = ops.image_to_nhwc(img.to(dev), dt, 4, halo=3)
    wp = torch.zeros(KO, 7, 8, 4)
    wp[:, :, :7, :3] = w7.permute(0, 2, 3, 1)
    w = wp.to(dt).to(dev)
    shift = (0.2 * torch.randn(KO, generator=g)).to(dev)
    y, st = ops.conv2d_fwd(a0, w, 2, 0, bn_shift=shift)
    OH = H // 2
    assert y.shape == (N, OH, OH, KO) and st is not None and 0 < st.strips <= 512
    yr = F.conv2d(img.to(dt).float(), w7.to(dt).float(), stride=2, padding=3).permute(0, 2, 3, 1)
    assert rel_err(y.float().cpu(), yr) < 1e-2
    tiled = ops.conv2d_fwd(a0, w, 2, 0)                       # no statistics requested: the tiled contraction kernel
    assert rel_err(y.float().cpu(), tiled.float().cpu()) < 4e-3
    parts = st.parts[: st.strips * 2 * KO].view(st.strips, 2, KO).cpu().double()
    d = y.float().cpu().double().view(-1, KO) - shift.cpu().double()
    assert rel_err(parts[:, 0].sum(0), d.sum(0)) < 1e-4 and rel_err(parts[:, 1].sum(0), (d * d).sum(0)) < 1e-4
    # BatchNorm forward fed with them == BatchNorm forward reducing the stored tensor itself; the fused tail likewise
    gamma = (0.5 + torch.rand(KO, generator=g)).to(dev); beta = (0.1 * torch.randn(KO, generator=g)).to(dev)
    outs = []
    for stats in (st, None):
        rm, rv = shift.clone(), torch.ones(KO, device=dev)
        out, mean, rstd = ops.bn_fwd(y, gamma, beta, rm, rv, None, stats=stats)
        rm2, rv2 = shift.clone(), torch.ones(KO, device=dev)
        pooled, arg, mean2, rstd2 = ops.bn_fwd_maxpool(y, gamma, beta, rm2, rv2, None, stats=stats)
        assert torch.equal(mean.cpu(), mean2.cpu()) and torch.equal(rstd.cpu(), rstd2.cpu()) and torch.equal(rm.cpu(), rm2.cpu())
        ref_pool, ref_arg = ops.maxpool_fwd(out)
        assert torch.equal(pooled.cpu(), ref_pool.cpu()) and torch.equal(arg.cpu(), ref_arg.cpu())
        outs.append((mean.cpu(), rstd.cpu(), rv.cpu()))
    assert torch.allclose(outs[0][0], outs[1][0], atol=1e-5, rtol=1e-5) and torch.allclose(outs[0][1], outs[1][1], rtol=1e-4)
    assert torch.allclose(outs[0][2], outs[1][2], rtol=1e-4)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("N,H,W,C,KO", [(2, 9, 9, 64, 64), (3, 14, 14, 128, 64), (1, 7, 7, 64, 128), (2, 30, 30, 64, 64),
                                        (2, 8, 11, 64, 64), (5, 7, 7, 128, 128)])
def test_conv3x3_wgrad_streaming_kernel(backend, N, H, W, C, KO):
    """conv3x3_wgrad.hip: the 3x3 / stride-1 / pad-1 weight gradient as a streaming kernel over the padded-linear pixel
    index (every tap a constant shift of ONE ring of x rows in LDS, dy shared by the nine taps, 64 x 9 x 64 accumulators
    per workgroup) -- against torch's convolution backward, against the implicit-GEMM kernel (explicit split_k selects
    it), accumulation into a non-zero gradient, several (ko, c) chunk pairs, non-square images, ranges that end in the
    middle of an image."""
    dev = select(backend)
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(N * H + C + W)
    x = torch.randn(N, H, W, C, generator=g).to(dt); dy = torch.randn(N, H, W, KO, generator=g).to(dt)
    wr = torch.zeros(KO, C, 3, 3, requires_grad=True)
    F.conv2d(x.float().permute(0, 3, 1, 2), wr, stride=1, padding=1).backward(dy.float().permute(0, 3, 1, 2))
    ref = wr.grad.permute(0, 2, 3, 1)
    dw0 = torch.randn(KO, 3, 3, C, generator=g)
    import ctypes
    from virtex_amd import _lib
    _lib.call("vtx_set_switch", b"wgrad3x3", ctypes.c_int(2))           # 2 = every image size (default 1: only >= 28x28)
    try:
        ops.profile_start()
        dw = ops.conv2d_wgrad(x.to(dev), dy.to(dev), dw0.clone().to(dev), 1, 1)
        recs = ops.profile_stop()
    finally:
        _lib.call("vtx_set_switch", b"wgrad3x3", ctypes.c_int(1))
    assert any("conv3x3_wgrad_stream" in r["name"] and r["launches"] == 1 for r in recs), [r["name"] for r in recs]
    assert rel_err(dw.cpu() - dw0, ref) < 1e-5                  # bf16 products are exact in fp32: only the summation order differs
    old = ops.conv2d_wgrad(x.to(dev), dy.to(dev), dw0.clone().to(dev), 1, 1, split_k=2)
    assert rel_err(dw.cpu(), old.cpu()) < 1e-5


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("N,H,W,C,KO", [(2, 9, 9, 32, 32), (1, 12, 45, 64, 64), (3, 16, 16, 32, 256), (2, 30, 30, 64, 128), (5, 7, 7, 128, 256)])
def test_conv3x3_kernel_sharing_the_operand_tile_across_a_filter_row(backend, N, H, W, C, KO):
    """conv3x3_kernel.h: 3x3 / stride 1 / pad 1 forward and input gradient with ONE A tile per (filter row, 32 channels)
    serving the three taps (tiles of BM - 2 output pixels in linear NHWC order, the taps that would reach across the left /
    right image edge zeroed in registers, rows above / below the image zero-filled by the buffer load): against torch, against
    the generic implicit-GEMM kernel (switch off), with the BatchNorm statistics / fused BatchNorm backward epilogues; tiles
    that start and end in the middle of image rows, odd widths, both tile shapes (KO <= 64: 128 x 64, else 256 x 128)."""
    import ctypes
    from virtex_amd import _lib
    dev = select(backend)
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(N * H + C + W + KO)
    x = torch.randn(N, H, W, C, generator=g).to(dt)
    w = (torch.randn(KO, 3, 3, C, generator=g) / (9 * C) ** 0.5).to(dt)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_()
    yr = F.conv2d(xr, w.float().permute(0, 3, 1, 2), stride=1, padding=1)
    dy = torch.randn(yr.shape, generator=g).permute(0, 2, 3, 1).contiguous().to(dt)
    yr.backward(dy.float().permute(0, 3, 1, 2))
    wt = w.permute(3, 1, 2, 0).contiguous()

    def run(sw):
        _lib.call("vtx_set_switch", b"conv3x3_shared", ctypes.c_int(sw))
        try:
            ops.profile_start()
            y = ops.conv2d_fwd(x.to(dev), w.to(dev), 1, 1)
            shift = torch.zeros(KO, device=dev)
            y2, st = ops.conv2d_fwd(x.to(dev), w.to(dev), 1, 1, bn_shift=shift)
            dx = ops.conv2d_dgrad(dy.to(dev), wt.to(dev), x.shape, 1, 1)
            # fused BatchNorm backward of the layer that produced x (mask recomputed from its input xin)
            xin = torch.randn(N, H, W, C, generator=torch.Generator().manual_seed(3)).to(dt).to(dev)
            mean = torch.zeros(C, device=dev); rstd = torch.ones(C, device=dev)
            gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
            bn = ops.BnBwd(xin.view(-1, C), mean, rstd, gamma=gamma, beta=beta)
            dz, bst = ops.conv2d_dgrad(dy.to(dev), wt.to(dev), x.shape, 1, 1, bn=bn)
            recs = ops.profile_stop()
            parts = st.parts[: st.strips * 2 * KO].view(st.strips, 2, KO).sum(0).cpu()
            bparts = bst.parts[: bst.strips * 2 * C].view(bst.strips, 2, C).sum(0).cpu()
            return y.float().cpu(), y2.float().cpu(), parts, dx.float().cpu(), dz.float().cpu(), bparts, recs
        finally:
            _lib.call("vtx_set_switch", b"conv3x3_shared", ctypes.c_int(1))

    new, old = run(2), run(0)
    assert sum(r["launches"] for r in new[6] if "Conv3x3SharedA" in r["name"]) == 4, [r["name"] for r in new[6]]
    assert not any("Conv3x3SharedA" in r["name"] for r in old[6])
    assert rel_err(new[0], yr.detach().permute(0, 2, 3, 1)) < 1e-2
    assert rel_err(new[3], xr.grad.permute(0, 2, 3, 1)) < 1e-2
    assert torch.equal(new[0], new[1])                                   # the statistics epilogue stores the same values
    for a, b in ((new[0], old[0]), (new[3], old[3]), (new[4], old[4])):  # same products, same order of taps: fp32 sums in another order
        assert rel_err(a, b) < 2e-3
    yq = new[1].view(-1, KO).double()
    assert rel_err(new[2][0], yq.sum(0).float()) < 1e-3 and rel_err(new[2][1], (yq * yq).sum(0).float()) < 1e-3
    assert rel_err(new[5], old[5]) < 2e-3


# ---------------------------------------------------------------------------------------------- generation 3 (gemm_v3.h)
def _set_dma_late(backend, late):
    """emulator only: 1 = an LDS-DMA lands at the issuing lane's covering s_waitcnt vmcnt (the latest the hardware allows:
    a read placed before wait + barrier sees stale bytes), 0 = at issue (the earliest: a stage issued in front of the last
    read of the bytes it replaces corrupts that read)"""
    import ctypes
    from virtex_amd import _lib
    if backend == "emu":
        _lib.lib().hipemu_set_dma_late(ctypes.c_int(late))


GEN3_PARAMS = [pytest.param("emu", 0, marks=pytest.mark.emu), pytest.param("emu", 1, marks=pytest.mark.emu),
               pytest.param("gpu", 0, marks=pytest.mark.gpu)]


@pytest.mark.parametrize("backend,late", GEN3_PARAMS)
def test_generation3_persistent_blocks(backend, late):
    """vtx_set_switch("gen3_pers", n): n blocks of the 256x256 kernel walk all tiles; the first K tile of a block's NEXT tile is
    staged in front of the current tile's strip epilogue (strips moved to units the staging leaves alone).  Few blocks, many tiles:
    interior tiles followed by interior tiles (staged path), by ragged edge tiles (general epilogue, nothing staged) and the
    other way round; one K tile, odd and even tile counts; every plain epilogue flavour.  Bit-identical to one block per tile."""
    import ctypes
    from virtex_amd import _lib
    dev = select(backend)
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(321)
    try:
        _set_dma_late(backend, late)
        _lib.lib().vtx_set_tile_override(ctypes.c_int(20))
        for (M, N, K) in ((256 * 5 + 40, 256 * 3, 64), (256 * 6, 256 * 2 + 8, 192), (256 * 9, 256 * 2, 320)):
            a = torch.randn(M, K, generator=g).to(dt).to(dev); b = torch.randn(N, K, generator=g).to(dt).to(dev)
            bias = torch.randn(N, generator=g).to(dev); res = torch.randn(M, N, generator=g).to(dt).to(dev)
            outs = []
            for pers in (0, 8):
                _lib.call("vtx_set_switch", b"gen3_pers", ctypes.c_int(pers))
                o1 = ops.gemm_nt(a, b, bias, res, act=ops.ACT_GELU)
                assert _generation() == 3
                o2 = ops.gemm_nt(a, b, bias, out_f32=True)
                o3 = ops.gemm_nt(a, b)
                outs.append((o1.float().cpu(), o2.cpu(), o3.float().cpu()))
            for x, y in zip(*outs):
                assert torch.equal(x, y), (M, N, K)
            ref = F.gelu(a.float().cpu() @ b.float().cpu().t() + bias.cpu()) + res.float().cpu()
            assert rel_err(outs[1][0], ref) < 1e-2, (M, N, K)
    finally:
        _lib.call("vtx_set_switch", b"gen3_pers", ctypes.c_int(0))
        _lib.lib().vtx_set_tile_override(ctypes.c_int(-1))
        _set_dma_late(backend, 0)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("cand", [20, 21])
def test_generation3_register_transposed_epilogue_equals_the_strips(backend, cand):
    """vtx_set_switch("epi_regs", 1): the plain epilogue of interior tiles without LDS -- the four 16x16 tiles of a wave row
    transposed across the four lane rows with v_permlane16_swap / v_permlane32_swap (semantics probed on the part:
    tools/probes/permlane_probe.hip; measured slower than the strips, so it is off by default) -- must store exactly what the
    strip epilogue stores: bf16 with bias + GELU + residual, fp32, and split-K partial tiles."""
    import ctypes
    from virtex_amd import _lib
    dev = select(backend)
    g = torch.Generator().manual_seed(7 + cand)
    M, N, K = 512 + 40, 512, 128                                   # interior tiles (lean path) and a ragged last row of tiles
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev); b = torch.randn(N, K, generator=g).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev); res = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
    outs = {}
    try:
        _lib.lib().vtx_set_tile_override(ctypes.c_int(cand))
        for regs in (0, 1):
            _lib.call("vtx_set_switch", b"epi_regs", ctypes.c_int(regs))
            o1 = ops.gemm_nt(a, b, bias, res, act=ops.ACT_GELU)
            assert _generation() == 3
            o2 = ops.gemm_nt(a, b, bias, out_f32=True)
            outs[regs] = (o1.float().cpu(), o2.cpu())
    finally:
        _lib.call("vtx_set_switch", b"epi_regs", ctypes.c_int(0))
        _lib.lib().vtx_set_tile_override(ctypes.c_int(-1))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ref = F.gelu(a.float().cpu() @ b.float().cpu().t() + bias.cpu()) + res.float().cpu()
    assert rel_err(outs[1][0], ref) < 1e-2


@pytest.mark.parametrize("backend,late", GEN3_PARAMS)
@pytest.mark.parametrize("cand", [20, 21])
def test_contraction_generation3_gemm(backend, late, cand):
    """The phase-interleaved kernels (20: 256x256 blocks, 21: 256x128) on plain matrices: one K tile, odd and even tile
    counts (the 256x256 loop works off two tiles per trip, the 256x128 loop three), a ragged K tail, ragged M / N, every
    epilogue flavour of vtx_gemm_nt, against torch fp32 on the same bf16 operands."""
    import ctypes
    from virtex_amd import _lib
    dev = select(backend)
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(100 + cand)
    try:
        _set_dma_late(backend, late)
        _lib.lib().vtx_set_tile_override(ctypes.c_int(cand))
        for (M, N, K) in ((300, 520, 64), (300, 520, 192), (520, 300, 200), (260, 132, 448), (257, 516, 512)):
            a = torch.randn(M, K, generator=g).to(dt); b = torch.randn(N, K, generator=g).to(dt)
            bias = torch.randn(N, generator=g); res = torch.randn(M, N, generator=g).to(dt)
            out = ops.gemm_nt(a.to(dev), b.to(dev), bias.to(dev), res.to(dev), act=ops.ACT_GELU)
            assert _generation() == 3
            ref = F.gelu(a.float() @ b.float().t() + bias) + res.float()
            assert rel_err(out.float().cpu(), ref) < 1e-2, (M, N, K)
            out32 = ops.gemm_nt(a.to(dev), b.to(dev), bias.to(dev), out_f32=True)
            assert rel_err(out32.cpu(), a.float() @ b.float().t() + bias) < 2e-5 * K ** 0.5, (M, N, K)
        # statistics of the stored output (forward convolutions) ...
        M, N, K = 700, 256, 320
        a = torch.randn(M, K, generator=g).to(dt); b = (torch.randn(N, K, generator=g) / K ** 0.5).to(dt)
        shift = 0.3 * torch.randn(N, generator=g)
        y, st = ops.gemm_nt(a.to(dev), b.to(dev), bn_shift=shift.to(dev))
        assert _generation() == 3 and st is not None and st.strips == (M + 255) // 256
        parts = st.parts[: st.strips * 2 * N].view(st.strips, 2, N).cpu()
        d = y.float().cpu() - shift
        assert torch.allclose(parts[:, 0].sum(0), d.sum(0), atol=2e-2, rtol=1e-3)
        assert torch.allclose(parts[:, 1].sum(0), (d * d).sum(0), rtol=1e-3)
        # ... and the fused BatchNorm backward (input-gradient convolutions)
        x = (0.7 * torch.randn(M, N, generator=g) + 0.3).to(dt); res = torch.randn(M, N, generator=g).to(dt)
        mean = x.float().mean(0); rstd = (x.float().var(0, unbiased=False) + 1e-5).rsqrt()
        gamma = 0.5 + torch.rand(N, generator=g); beta = 0.2 * torch.randn(N, generator=g)
        z = a.float() @ b.float().t() + res.float()
        dz_r, s1_r, s2_r, _ = _bn_bwd_reference(z, x.float(), mean, rstd, gamma, beta, None, "remask")
        bn = ops.BnBwd(x.to(dev), mean.to(dev), rstd.to(dev), gamma=gamma.to(dev), beta=beta.to(dev))
        dz, st = ops.gemm_nt_bnbwd(a.to(dev), b.to(dev), bn, residual=res.to(dev))
        assert _generation() == 3 and st is not None and st.strips == (M + 255) // 256
        assert rel_err(dz.float().cpu(), dz_r) < 1e-2
        parts = st.parts[: st.strips * 2 * N].view(st.strips, 2, N).cpu()
        assert rel_err(parts[:, 0].sum(0), s1_r) < 1e-2 and rel_err(parts[:, 1].sum(0), s2_r) < 1e-2
        # tied projection + cross-entropy: row log-sum-exp partials per column group, softmax-gradient epilogue
        R, V, H = 300, 1000, 128
        h = torch.randn(R, H, generator=g).to(dt); w = (0.3 * torch.randn(V, H, generator=g)).to(dt)
        bias = 0.2 * torch.randn(V, generator=g)
        tgt = torch.randint(1, V, (R,), generator=g); tgt[::7] = 0
        logits = (h.float() @ w.float().t() + bias).requires_grad_()
        ref = F.cross_entropy(logits, tgt, ignore_index=0)
        ref.backward()
        lc, lse = ops.tied_ce_fwd(h.to(dev), w.to(dev), bias.to(dev), tgt.to(dev), 0)
        assert _generation() == 3
        assert abs(lc[0].item() - ref.item()) < 2e-4 * abs(ref.item())
        assert torch.allclose(lse.cpu(), torch.logsumexp(logits.detach(), 1), rtol=2e-4, atol=2e-4)
        dl = ops.tied_ce_bwd(h.to(dev), w.to(dev), bias.to(dev), tgt.to(dev), lse, lc, torch.tensor([1.7]).to(dev), 0)
        assert rel_err(dl.float().cpu(), 1.7 * logits.grad) < 1e-2
    finally:
        _lib.lib().vtx_set_tile_override(ctypes.c_int(-1))
        _set_dma_late(backend, 0)


@pytest.mark.parametrize("backend,late", GEN3_PARAMS)
@pytest.mark.parametrize("cand", [20, 21])
def test_contraction_generation3_convolutions(backend, late, cand):
    """The same kernels behind the im2col-free gathers: 3x3 stride 1 forward / input gradient (tap masks, negative tap
    offsets), the stride-2 parity decomposition (four tap lists) and a strided 1x1, 64 channels (one K tile per tap)."""
    import ctypes
    from virtex_amd import _lib
    dev = select(backend)
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(200 + cand)
    try:
        _set_dma_late(backend, late)
        _lib.lib().vtx_set_tile_override(ctypes.c_int(cand))
        for (k, stride, H, C, KO) in ((3, 1, 9, 64, 64), (3, 2, 10, 64, 128), (1, 2, 8, 128, 64), (3, 1, 7, 128, 64)):
            pad = k // 2
            x = torch.randn(3, H, H, C, generator=g).to(dt)
            w = (torch.randn(KO, k, k, C, generator=g) / (k * k * C) ** 0.5).to(dt)
            xr = x.float().permute(0, 3, 1, 2).requires_grad_(); wr = w.float().permute(0, 3, 1, 2)
            yr = F.conv2d(xr, wr, stride=stride, padding=pad)
            dy = torch.randn(yr.shape, generator=g).permute(0, 2, 3, 1).contiguous().to(dt)
            yr.backward(dy.float().permute(0, 3, 1, 2))
            y = ops.conv2d_fwd(x.to(dev), w.to(dev), stride, pad)
            assert _generation() == 3
            assert rel_err(y.float().cpu(), yr.detach().permute(0, 2, 3, 1)) < 1e-2, (k, stride, H)
            dx = ops.conv2d_dgrad(dy.to(dev), w.permute(3, 1, 2, 0).contiguous().to(dev), x.shape, stride, pad)
            assert _generation() == 3
            assert rel_err(dx.float().cpu(), xr.grad.permute(0, 2, 3, 1)) < 1e-2, (k, stride, H)
    finally:
        _lib.lib().vtx_set_tile_override(ctypes.c_int(-1))
        _set_dma_late(backend, 0)


@pytest.mark.parametrize("backend,late", GEN3_PARAMS)
@pytest.mark.parametrize("cand", [20, 21])
def test_contraction_generation3_weight_gradients(backend, late, cand):
    """The k-major kernels of gemm_v3mc.h (transposing fragment reads from [64 k][128 rows] units): dense weight gradients with
    one, odd and even K-tile counts, ragged K / M / N, split-K slices through the workspace, accumulation into the gradient;
    convolution weight gradients (3x3 stride 1 / 2, 1x1 stride 2: the pixel-major gather with its position counters)."""
    import ctypes
    from virtex_amd import _lib
    dev = select(backend)
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(300 + cand)
    try:
        _set_dma_late(backend, late)
        _lib.lib().vtx_set_tile_override(ctypes.c_int(cand))
        for (K, M, N, split) in ((64, 264, 520, 1), (200, 264, 520, 1), (448, 304, 136, 2), (640, 256, 256, 3), (1000, 520, 264, 4)):
            at = torch.randn(K, M, generator=g).to(dt); bt = torch.randn(K, N, generator=g).to(dt)
            c0 = torch.randn(M, N, generator=g)
            acc = ops.gemm_tn_acc(at.to(dev), bt.to(dev), c0.clone().to(dev), split_k=split)
            assert _generation() == 3, (K, M, N)
            assert rel_err(acc.cpu(), c0 + at.float().t() @ bt.float()) < 2e-5 * K ** 0.5 + 1e-5, (K, M, N, split)
        for (k, stride, H, C, KO) in ((3, 1, 9, 64, 64), (3, 2, 10, 64, 128), (1, 2, 16, 128, 64), (3, 1, 7, 128, 256)):
            pad = k // 2
            x = torch.randn(3, H, H, C, generator=g).to(dt)
            w = (torch.randn(KO, k, k, C, generator=g) / (k * k * C) ** 0.5).to(dt)
            xr = x.float().permute(0, 3, 1, 2); wr = w.float().permute(0, 3, 1, 2).requires_grad_()
            yr = F.conv2d(xr, wr, stride=stride, padding=pad)
            dy = torch.randn(yr.shape, generator=g).permute(0, 2, 3, 1).contiguous().to(dt)
            yr.backward(dy.float().permute(0, 3, 1, 2))
            for split in (1, 2):
                dw0 = torch.randn(KO, k, k, C, generator=g)
                dw = ops.conv2d_wgrad(x.to(dev), dy.to(dev), dw0.clone().to(dev), stride, pad, split_k=split)
                assert _generation() == 3
                assert rel_err(dw.cpu() - dw0, wr.grad.permute(0, 2, 3, 1)) < 2e-3, (k, stride, H, split)
    finally:
        _lib.lib().vtx_set_tile_override(ctypes.c_int(-1))
        _set_dma_late(backend, 0)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("C,strips", [(64, 784), (256, 3136), (128, 600)])
def test_batchnorm_compaction_and_finalize_in_one_launch(backend, C, strips):
    """More than 512 statistics strips (the convolution epilogues of stages 1-2 at bs = 256: 784 / 3136 block rows) are folded
    AND finalized by one launch (bn_fin2_kernel: the last-arriving block of a channel group finalizes; agent-scope
    release / acquire hand-off) instead of a compaction launch and a finalize launch.  Forward: mean / rstd / output / running
    statistics equal the two-launch path (switch bn_fin2 = 0) and the plain reduction; backward: dx / dgamma / dbeta likewise;
    repeated launches keep working (the tickets are left at zero)."""
    import ctypes
    from virtex_amd import _lib
    dev = select(backend)
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(C + strips)
    rows_per = 2
    P = strips * rows_per
    x = (0.8 * torch.randn(P, C, generator=g) + 0.2).to(dt)
    gamma = 0.5 + torch.rand(C, generator=g); beta = 0.1 * torch.randn(C, generator=g)
    shift = 0.2 * torch.randn(C, generator=g)
    d = (x.float() - shift).view(strips, rows_per, C)
    parts = torch.stack([d.sum(1), (d * d).sum(1)], 1).contiguous()                    # [strips][2][C]
    xd, gd, bd = x.to(dev), gamma.to(dev), beta.to(dev)
    st = ops.BnStats(parts.to(dev), strips, shift.to(dev))
    out = {}
    for mode in (1, 0, 1):                                                             # one launch, two launches, one launch again
        _lib.call("vtx_set_switch", b"bn_fin2", ctypes.c_int(mode))
        rm, rv = shift.clone().to(dev), torch.ones(C, device=dev)
        y, mean, rstd = ops.bn_fwd(xd, gd, bd, rm, rv, None, stats=st)
        out.setdefault(mode, []).append((y.float().cpu(), mean.cpu(), rstd.cpu(), rm.cpu(), rv.cpu()))
    _lib.call("vtx_set_switch", b"bn_fin2", ctypes.c_int(0))
    ref = ops.bn_fwd(xd, gd, bd, shift.clone().to(dev), torch.ones(C, device=dev), None)
    for a in out[1]:
        b = out[0][0]
        assert torch.allclose(a[1], b[1], rtol=1e-5, atol=1e-6) and torch.allclose(a[2], b[2], rtol=1e-5)
        assert torch.allclose(a[3], b[3], rtol=1e-5, atol=1e-6) and torch.allclose(a[4], b[4], rtol=1e-5)
        assert rel_err(a[0], b[0]) < 1e-3
        assert torch.allclose(a[1], ref[1].cpu(), atol=5e-3, rtol=1e-2) and torch.allclose(a[2], ref[2].cpu(), rtol=1e-2)
    # backward: strips of (sum dz, sum dz * xhat)
    mean, rstd = out[1][0][1], out[1][0][2]
    dz = torch.randn(P, C, generator=g).to(dt)
    xh = (x.float() - mean) * rstd
    bparts = torch.stack([dz.float().view(strips, rows_per, C).sum(1), (dz.float() * xh).view(strips, rows_per, C).sum(1)], 1).contiguous()
    res = {}
    for mode in (1, 0):
        _lib.call("vtx_set_switch", b"bn_fin2", ctypes.c_int(mode))
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        dx = ops.bn_bwd_fused(xd, dz.to(dev), gd, mean.to(dev), rstd.to(dev), dg, db, ops.BnStats(bparts.to(dev), strips, None))
        res[mode] = (dx.float().cpu(), dg.cpu(), db.cpu())
    _lib.call("vtx_set_switch", b"bn_fin2", ctypes.c_int(0))
    assert rel_err(res[1][0], res[0][0]) < 1e-3
    assert torch.allclose(res[1][1], res[0][1], rtol=1e-4, atol=1e-3) and torch.allclose(res[1][2], res[0][2], rtol=1e-4, atol=1e-3)
    s1, s2 = dz.float().sum(0), (dz.float() * xh).sum(0)
    dx_ref = gamma * rstd * (dz.float() - s1 / P - xh * s2 / P)
    assert rel_err(res[1][0], dx_ref) < 1e-2


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("M", [1024, 2432])
def test_conv3_backward_in_one_streaming_kernel(backend, M):
    """vtx_conv3_bwd_fused (csrc/conv3_bwd.hip): bn3's backward applied while the gradient is loaded, conv3's input gradient
    with bn2's mask and sums, conv3's weight gradient -- against (1) the torch fp32 formulas and (2) the three launches it
    replaces (bn_bwd_fused -> gemm_nt_bnbwd, gemm_tn_acc on relu(bn2(x2)) as bn_fwd stores it).  M = 19 row blocks: workgroups
    with different round counts on the emulator's grid."""
    import os
    if os.environ.get("VIRTEX_AMD_CONV3_BWD", "1") == "0":
        pytest.skip("switched off (VIRTEX_AMD_CONV3_BWD=0)")
    dev = select(backend)
    dt = torch.bfloat16
    K, N = 256, 64
    g = torch.Generator().manual_seed(M)
    dz = (torch.randn(M, K, generator=g) * (torch.rand(M, K, generator=g) > 0.4)).to(dt)          # masked upstream gradient
    x3 = (0.8 * torch.randn(M, K, generator=g) + 0.3 * torch.randn(K, generator=g)).to(dt)
    x2 = (0.9 * torch.randn(M, N, generator=g) + 0.2).to(dt)
    wt = (torch.randn(N, K, generator=g) / 16).to(dt)                                                # conv3's weight, (cin, cout)
    gamma3 = 0.5 + torch.rand(K, generator=g); gamma2 = 0.5 + torch.rand(N, generator=g); beta2 = 0.3 * torch.randn(N, generator=g)
    mean3 = x3.float().mean(0); rstd3 = (x3.float().var(0, unbiased=False) + 1e-5).rsqrt()
    xh3 = (x3.float() - mean3) * rstd3
    s1 = dz.float().sum(0); s2 = (dz.float() * xh3).sum(0)
    parts3 = torch.stack([s1, s2]).view(1, 2, K).contiguous()
    # bn2 forward through the library: y2 (= conv3's input), mean2, rstd2 exactly as the step has them
    rm, rv = torch.zeros(N, device=dev), torch.ones(N, device=dev)
    y2, mean2, rstd2 = ops.bn_fwd(x2.to(dev).view(1, 1, M, N), gamma2.to(dev), beta2.to(dev), rm, rv, None, relu=True)
    y2 = y2.view(M, N)
    # ---- fused
    assert ops.conv3_bwd_fused_supported(dz.to(dev), wt.to(dev))
    dg_f, db_f = torch.zeros(K, device=dev), torch.zeros(K, device=dev)
    bn2 = ops.BnBwd(x2.to(dev), mean2, rstd2, gamma=gamma2.to(dev), beta=beta2.to(dev))
    dy2, st2, dwp, nparts = ops.conv3_bwd_fused(dz.to(dev), x3.to(dev), gamma3.to(dev), mean3.to(dev), rstd3.to(dev), dg_f, db_f,
                                                ops.BnStats(parts3.to(dev), 1, None), wt.to(dev), bn2)
    assert st2.strips == nparts and 0 < nparts <= M // 128
    dw_f = torch.zeros(K, N, device=dev)
    ops.partials_reduce_acc(dwp, nparts, dw_f)
    # ---- the three launches it replaces
    dg_r, db_r = torch.zeros(K, device=dev), torch.zeros(K, device=dev)
    dx3 = ops.bn_bwd_fused(x3.to(dev), dz.to(dev), gamma3.to(dev), mean3.to(dev), rstd3.to(dev), dg_r, db_r, ops.BnStats(parts3.to(dev), 1, None))
    bn2r = ops.BnBwd(x2.to(dev), mean2, rstd2, gamma=gamma2.to(dev), beta=beta2.to(dev))
    dy2_r, st2_r = ops.gemm_nt_bnbwd(dx3, wt.to(dev), bn2r)
    dw_r = torch.zeros(K, N, device=dev)
    ops.gemm_tn_acc(dx3, y2, dw_r)
    assert torch.equal(dg_f.cpu(), dg_r.cpu()) and torch.equal(db_f.cpu(), db_r.cpu())
    assert rel_err(dy2.float().cpu(), dy2_r.float().cpu()) < 3e-3
    sums = lambda st: st.parts[: st.strips * 2 * N].view(st.strips, 2, N).double().cpu().sum(0)     # noqa: E731
    assert rel_err(sums(st2)[0], sums(st2_r)[0]) < 3e-3 and rel_err(sums(st2)[1], sums(st2_r)[1]) < 3e-3
    assert rel_err(dw_f.cpu(), dw_r.cpu()) < 3e-3
    # ---- torch fp32
    dx3_t = gamma3 * rstd3 * (dz.float() - s1 / M - xh3 * s2 / M)
    xh2 = (x2.float() - mean2.cpu()) * rstd2.cpu()
    keep = xh2 * gamma2 + beta2 > 0
    dy2_t = torch.where(keep, dx3_t @ wt.float().t(), torch.zeros(()))
    assert rel_err(dy2.float().cpu(), dy2_t) < 1e-2
    assert rel_err(sums(st2)[0], dy2_t.double().sum(0)) < 1e-2 and rel_err(sums(st2)[1], (dy2_t * xh2).double().sum(0)) < 1e-2
    assert rel_err(dw_f.cpu(), dx3_t.t() @ torch.relu(xh2 * gamma2 + beta2)) < 1e-2
    # the sums describe the STORED gradient
    assert rel_err(sums(st2)[0], dy2.double().cpu().sum(0)) < 1e-5


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("M,K,N", [(384, 256, 64), (520, 512, 128)])
def test_bn_backward_folded_into_conv3(backend, M, K, N):
    """csrc/bn_fold.hip: bn3's backward folded into conv3's weights -- dy2 = dz (a0 o W3) + a3 (W3^T diag(b1) W3) + W3^T c and
    dW3 = diag(a0) dz^T a3 + diag(b1) W3 (a3^T a3) + c colsum(a3)^T, with x3 = a3 W3^T -- against (1) the fp64 evaluation of the
    BatchNorm-backward formulas on the same bf16 operands and (2) the launches it replaces (bn_bwd_fused -> gemm_nt_bnbwd /
    gemm_tn_acc): the folded form may not be further from (1) than twice the pass form is (both round to bf16, in different
    places), and the BatchNorm parameter gradients are bit-identical (same finalize)."""
    dev = select(backend)
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(M + K)
    a3 = torch.relu(0.9 * torch.randn(M, N, generator=g) + 0.2).to(dt)                              # conv3's input: a ReLU output
    w3 = (torch.randn(K, N, generator=g) * (2.0 / N) ** 0.5).to(dt)                                  # conv3's weight (cout, cin)
    wt = w3.t().contiguous()                                                                        # (cin, cout): the input-gradient operand
    x3 = (a3.float() @ w3.float().t()).to(dt)                                                       # as the forward pass stored it
    dz = (torch.randn(M, K, generator=g) * (torch.rand(M, K, generator=g) > 0.4)).to(dt)            # masked upstream gradient
    x2 = (0.9 * torch.randn(M, N, generator=g) + 0.2).to(dt)
    gamma3 = 0.5 + torch.rand(K, generator=g); gamma2 = 0.5 + torch.rand(N, generator=g); beta2 = 0.3 * torch.randn(N, generator=g)
    mean3 = x3.float().mean(0); rstd3 = (x3.float().var(0, unbiased=False) + 1e-5).rsqrt()
    mean2 = x2.float().mean(0); rstd2 = (x2.float().var(0, unbiased=False) + 1e-5).rsqrt()
    xh3 = (x3.double() - mean3.double()) * rstd3.double()
    s1 = dz.double().sum(0); s2 = (dz.double() * xh3).sum(0)
    parts3 = torch.stack([s1, s2]).float().view(1, 2, K).contiguous()
    # (1) fp64 reference of the formulas
    dx3 = (gamma3 * rstd3).double() * (dz.double() - s1 / M - xh3 * s2 / M)
    keep = ((x2.double() - mean2.double()) * rstd2.double() * gamma2.double() + beta2.double()) > 0
    dy2_ref = torch.where(keep, dx3 @ w3.double(), torch.zeros((), dtype=torch.float64))
    dw_ref = dx3.t() @ a3.double()
    to = lambda t: t.to(dev)                                                                        # noqa: E731
    bn2 = lambda: ops.BnBwd(to(x2), to(mean2), to(rstd2), gamma=to(gamma2), beta=to(beta2))         # noqa: E731
    # (2) the pass form
    dg_p, db_p = torch.zeros(K, device=dev), torch.zeros(K, device=dev)
    dx3_p = ops.bn_bwd_fused(to(x3), to(dz), to(gamma3), to(mean3), to(rstd3), dg_p, db_p, ops.BnStats(to(parts3), 1, None))
    dy2_p, st_p = ops.gemm_nt_bnbwd(dx3_p, to(wt), bn2())
    dw_p = torch.zeros(K, N, device=dev)
    ops.gemm_tn_acc(dx3_p, to(a3), dw_p)
    # (3) the folded form, as modules/visual_backbones.py::conv3_back_folded issues it
    dg_f, db_f = torch.zeros(K, device=dev), torch.zeros(K, device=dev)
    wa, wb, bias, abc = ops.bn_bwd_fold(to(wt), to(gamma3), to(mean3), to(rstd3), dg_f, db_f, ops.BnStats(to(parts3), 1, None), M)
    h = ops.gemm_nt(wb, to(wt))
    tmp = ops.gemm_nt(to(a3), h, bias=bias)
    dy2_f, st_f = ops.gemm_nt_bnbwd(to(dz), wa, bn2(), residual=tmp)
    t = torch.zeros(K, N, device=dev); gram = torch.zeros(N, N, device=dev); csum = torch.zeros(N, device=dev)
    ops.gemm_tn_acc(to(dz), to(a3), t)
    ops.gemm_tn_acc(to(a3), to(a3), gram)
    ops.colsum_acc(to(a3), csum)
    wg = ops.gemm_nt(to(w3.float()), gram)
    dw_f = torch.zeros(K, N, device=dev)
    ops.wgrad_fold_combine(dw_f, t, wg, csum, abc)
    assert torch.equal(dg_f.cpu(), dg_p.cpu()) and torch.equal(db_f.cpu(), db_p.cpu())
    # the coefficients: dx3 = a0 dz + b1 x3 + c reproduces the formula
    a0, b1, c = (v.double().cpu() for v in abc)
    assert rel_err(a0 * dz.double() + b1 * x3.double() + c, dx3) < 1e-5
    e_pass, e_fold = rel_err(dy2_p.double().cpu(), dy2_ref), rel_err(dy2_f.double().cpu(), dy2_ref)
    assert e_pass < 1e-2 and e_fold < max(2.0 * e_pass, 1e-2), (e_pass, e_fold)
    w_pass, w_fold = rel_err(dw_p.double().cpu(), dw_ref), rel_err(dw_f.double().cpu(), dw_ref)
    assert w_pass < 1e-2 and w_fold < max(2.0 * w_pass, 1e-2), (w_pass, w_fold)
    print(f"[bn fold M={M} K={K} N={N}] input gradient vs fp64: pass form {e_pass:.2e}, folded {e_fold:.2e}; "
          f"weight gradient: pass form {w_pass:.2e}, folded {w_fold:.2e}")
    sums = lambda st: st.parts[: st.strips * 2 * N].view(st.strips, 2, N).double().cpu().sum(0)     # noqa: E731
    assert rel_err(sums(st_f)[0], dy2_f.double().cpu().sum(0)) < 1e-5                               # the sums describe the STORED gradient
    assert rel_err(sums(st_f)[0], sums(st_p)[0]) < 2e-2 and rel_err(sums(st_f)[1], sums(st_p)[1]) < 2e-2


def test_pixel_index_division_is_exact_below_2_30(tmp_path):
    """vtx_fdiv30 (csrc/vtx_common.h) -- the reciprocal division behind every pixel-index decomposition of the library (block
    prologues, scattering epilogues, pooling) -- compiled from the kernels' own header and held against integer division on 12 M
    cases: small / real / random divisors, multiples of d and their neighbours, the old 2^24 limit, 2^30 - 1, reciprocals nudged by
    -2 ... +2 ulp (the hardware's v_rcp_f32 is a 1-ulp instruction).  Rounds 1-5 had a one-estimate form exact below 2^24 only,
    which capped the per-GPU batch at 334 images of 224 x 224."""
    import subprocess
    from virtex_amd import build as vb
    if not os.path.exists(vb.HOST_CLANG):
        pytest.skip("no host clang")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fdiv_check")
    r = subprocess.run([vb.HOST_CLANG, "-x", "c++", "-O2", "-std=c++17", "-DHIPEMU=1", "-I", os.path.join(vb.EMU_DIR, "include"),
                        "-I", vb.CSRC, "-Wno-unused-value", "-Wno-unknown-pragmas", os.path.join(root, "tests", "fuzz", "fdiv_check.cpp"),
                        "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout[-500:]
    assert int(r.stdout.split()[1]) > 10_000_000


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", DTYPES)
def test_convolution_and_pooling_indices_beyond_2_24_pixels_gpu(dtype):
    """Every kernel that decomposes a pixel index (3x3 forward / input gradient / weight gradient, stride 1 and 2, max-pooling
    forward and backward) on a tensor of 600 x 168 x 168 = 16.9 M pixels (> 2^24): convolutions and pooling are independent per
    image, so the result on the whole batch must equal the results on its two halves -- bit for bit for the per-pixel outputs,
    to summation order for the weight gradient."""
    dev = select("gpu")
    N, H, C, KO = 600, 168, 8, 8
    assert N * H * H > (1 << 24)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, H, H, C, generator=g).to(dtype).to(dev)
    w = (torch.randn(KO, 3, 3, C, generator=g) / 8).to(dtype).to(dev)
    wt = w.permute(3, 1, 2, 0).contiguous()
    h1, h2 = slice(0, N // 2), slice(N // 2, N)
    for stride in (1, 2):
        y = ops.conv2d_fwd(x, w, stride, 1)
        parts = torch.cat([ops.conv2d_fwd(x[h1].contiguous(), w, stride, 1), ops.conv2d_fwd(x[h2].contiguous(), w, stride, 1)])
        assert torch.equal(y, parts), stride
        dy = torch.randn(y.shape, generator=g).to(dtype).to(dev)
        dx = ops.conv2d_dgrad(dy, wt, x.shape, stride, 1)
        dparts = torch.cat([ops.conv2d_dgrad(dy[h1].contiguous(), wt, x[h1].shape, stride, 1),
                            ops.conv2d_dgrad(dy[h2].contiguous(), wt, x[h2].shape, stride, 1)])
        assert torch.equal(dx, dparts), stride
        dw = ops.conv2d_wgrad(x, dy, torch.zeros(KO, 3, 3, C, device=dev), stride, 1)
        dwp = ops.conv2d_wgrad(x[h1].contiguous(), dy[h1].contiguous(), torch.zeros(KO, 3, 3, C, device=dev), stride, 1)
        dwp = ops.conv2d_wgrad(x[h2].contiguous(), dy[h2].contiguous(), dwp, stride, 1)
        assert rel_err(dw.cpu(), dwp.cpu()) < (1e-4 if dtype == torch.float32 else 2e-3), stride
    p, arg = ops.maxpool_fwd(x)
    p1, a1 = ops.maxpool_fwd(x[h1].contiguous()); p2, a2 = ops.maxpool_fwd(x[h2].contiguous())
    assert torch.equal(p, torch.cat([p1, p2])) and torch.equal(arg, torch.cat([a1, a2]))
    dp = torch.randn(p.shape, generator=g).to(dtype).to(dev)
    dxp = ops.maxpool_bwd(dp, arg, x.shape)
    dxh = torch.cat([ops.maxpool_bwd(dp[h1].contiguous(), a1, x[h1].shape), ops.maxpool_bwd(dp[h2].contiguous(), a2, x[h2].shape)])
    assert torch.equal(dxp, dxh)


@pytest.mark.parametrize("backend", BACKENDS)
def test_splitk_reductions_of_several_weight_gradients_in_one_launch(backend):
    """ops.splitk_batch (vtx_splitk_batch_begin / _end): the split-K reductions of the contractions issued inside a batch are
    deferred to ONE launch -- the same sums in the same order, so the gradients are BIT-identical to the immediate form, also
    for an accumulation onto a non-zero gradient, a contraction that needs no reduction (one slice) in the middle, more than
    eight pending reductions (the batch flushes itself) and a caller-owned partial buffer (never deferred)."""
    dev = select(backend)
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(21)
    shapes = [(640, 64, 96, 5), (512, 128, 64, 4), (384, 32, 200, 1), (768, 96, 32, 6)] + [(256, 16 + 8 * i, 24, 2) for i in range(9)]
    a = [torch.randn(K, M, generator=g).to(dt).to(dev) for K, M, N, S in shapes]
    b = [torch.randn(K, N, generator=g).to(dt).to(dev) for K, M, N, S in shapes]
    init = [torch.randn(M, N, generator=g).to(dev) for K, M, N, S in shapes]
    x = torch.randn(2, 9, 9, 16, generator=g).to(dt).to(dev); dy = torch.randn(2, 9, 9, 32, generator=g).to(dt).to(dev)
    parts = torch.randn(3, 24, 16, generator=g).to(dev)

    def run(batched):
        outs = [t.clone() for t in init]
        dw = torch.zeros(32, 3, 3, 16, device=dev)
        pr = torch.ones(24, 16, device=dev)
        ops.profile_start()
        sk = ops.splitk_batch(dev)
        if batched:
            sk.begin()
        for (K, M, N, S), ai, bi, o in zip(shapes, a, b, outs):
            ops.gemm_tn_acc(ai, bi, o, split_k=S)
        ops.conv2d_wgrad(x, dy, dw, 1, 1, split_k=3)
        ops.partials_reduce_acc(parts, 3, pr)
        sk.end()
        launches = sum(r["launches"] for r in ops.profile_stop() if r["name"].startswith("family:splitk_reduce"))
        return outs + [dw, pr], launches
    saved = ops.splitk_batch.enabled
    try:
        ops.splitk_batch.enabled = True
        plain, n_plain = run(False)
        batch, n_batch = run(True)
    finally:
        ops.splitk_batch.enabled = saved
    for p, q in zip(plain, batch):
        assert torch.equal(p.cpu(), q.cpu())
    n_reduced = sum(1 for s in shapes if s[3] > 1) + 1                   # + the convolution's
    assert n_plain == n_reduced + 1                                        # + the caller-owned partials
    assert n_batch == 2 + 1, n_batch                                       # 13 deferred = one self-flush at eight + the flush at end(); + the partials
    ref = init[0].cpu().double() + a[0].double().cpu().t() @ b[0].double().cpu()
    assert rel_err(batch[0].cpu(), ref) < 1e-5
