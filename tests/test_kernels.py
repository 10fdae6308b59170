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
