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


# ---------------------------------------------------------------------------------------
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
c_long = _lib.ctypes.c_long


def gemm_nt(a, b, bias=None, residual=None, act=ACT_NONE, want_preact=False, alpha=1.0,
            p_drop=0.0, seed=0, out=None):
    """C[M,N] = dropout(act(alpha * a[M,K] @ b[N,K]^T + bias)) + residual.
    a, b: 2-D (row stride may exceed the row length).  Returns C (and preact if asked)."""
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K and b.dtype == a.dtype
    if out is None:
        out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    pre = torch.empty(M, N, dtype=a.dtype, device=a.device) if want_preact else None
    _chk(bias, "bias", torch.float32)
    if residual is not None:
        assert residual.dim() == 2 and residual.stride(1) == 1 and residual.dtype == a.dtype
    call("vtx_gemm_nt", c_int(dtype_code(a.dtype)), c_int(M), c_int(N), c_int(K), ptr(a),
         c_long(a.stride(0)), ptr(b), c_long(b.stride(0)), ptr(out), c_long(out.stride(0)), ptr(bias),
         ptr(residual), c_long(residual.stride(0) if residual is not None else 0), ptr(pre),
         c_int(act), c_float(alpha), c_float(p_drop), c_u64(seed), stream_ptr(a))
    return (out, pre) if want_preact else out


def gemm_tn_acc(a, b, out, alpha=1.0, split_k=0):
    """out[M,N] (fp32) += alpha * a[K,M]^T @ b[K,N]."""
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    K, M = a.shape
    N = b.shape[1]
    assert b.shape[0] == K and b.dtype == a.dtype and out.dtype == torch.float32
    assert out.dim() == 2 and out.stride(1) == 1 and out.shape == (M, N)
    call("vtx_gemm_tn_acc", c_int(dtype_code(a.dtype)), c_int(M), c_int(N), c_int(K), ptr(a),
         c_long(a.stride(0)), ptr(b), c_long(b.stride(0)), ptr(out), c_long(out.stride(0)),
         c_float(alpha), c_int(split_k), stream_ptr(a))
    return out
