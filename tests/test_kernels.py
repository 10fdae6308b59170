"""Per-kernel parity: HIP kernel (through the C ABI) vs the torch fp32 op it replaces.
Runs on the CPU fiber emulator (`emu`) and on a real MI355X (`gpu`)."""
import pytest
import torch
import torch.nn.functional as F

from backends import BACKENDS, rel_err, select, tol
from virtex_amd import ops

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
