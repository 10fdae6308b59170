"""Thin, autograd-free Python bindings of the C ABI (include/virtex_amd.h).

Tensors are validated here (device, dtype, contiguity) and handed to the native library as
raw pointers plus the caller's current HIP stream.  Every function allocates its outputs
with torch (device memory plumbing) and returns them.
"""
import torch

from . import _lib
from ._lib import c_float, c_int, c_u64, call, dtype_code, ptr, stream_ptr


def _chk(t, name, dtype=None):
    if t is None:
        return
    if not t.is_contiguous():
        raise _lib.VtxError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise _lib.VtxError(f"{name} must be {dtype}, got {t.dtype}")


# ---------------------------------------------------------------------------------------
def layernorm_residual_fwd(x, y, gamma, beta, eps, p_drop=0.0, seed=0):
    """out = LN(x + dropout(y)); returns (out, mean, rstd). x,y: (..., H)."""
    H = x.shape[-1]
    rows = x.numel() // H
    _chk(x, "x"); _chk(y, "y", x.dtype); _chk(gamma, "gamma", torch.float32); _chk(beta, "beta", torch.float32)
    out = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    call("vtx_layernorm_residual_fwd", c_int(dtype_code(x.dtype)), ptr(x), ptr(y), ptr(gamma),
         ptr(beta), ptr(out), ptr(mean), ptr(rstd), c_int(rows), c_int(H), c_float(eps),
         c_float(p_drop), c_u64(seed), stream_ptr(x))
    return out, mean, rstd


def layernorm_residual_bwd(x, y, gamma, mean, rstd, dout, dgamma, dbeta, p_drop=0.0, seed=0):
    """Returns (dz, dy); dgamma/dbeta (fp32) are accumulated in place."""
    H = x.shape[-1]
    rows = x.numel() // H
    _chk(x, "x"); _chk(y, "y", x.dtype); _chk(dout, "dout", x.dtype)
    dz = torch.empty_like(x)
    dy = torch.empty_like(x) if (y is not None and p_drop > 0.0) else None
    call("vtx_layernorm_residual_bwd", c_int(dtype_code(x.dtype)), ptr(x), ptr(y), ptr(gamma),
         ptr(mean), ptr(rstd), ptr(dout), ptr(dz), ptr(dy), ptr(dgamma), ptr(dbeta), c_int(rows),
         c_int(H), c_float(p_drop), c_u64(seed), stream_ptr(x))
    return dz, (dy if dy is not None else dz)
