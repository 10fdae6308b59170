"""Thin, autograd-free Python bindings of the C ABI (include/virtex_amd.h).

Tensors are validated here (device, dtype, contiguity) and handed to the native library as
raw pointers plus the caller's current HIP stream.  Every function allocates its outputs
with torch (device memory plumbing) and returns them.
"""
import os

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


def _ws_key(device):
    """Scratch buffers are per (device, stream): the same kernels run concurrently on the compute stream, the
    weight-gradient side stream and the shortcut-branch stream (virtex_amd/streams.py)."""
    return (device, _lib.current_stream_handle(device) if device.type == "cuda" else 0)


_ln_ws = {}


def layernorm_residual_bwd(x, y, gamma, mean, rstd, dout, dgamma, dbeta, p_drop=0.0, seed=0):
    """Returns (dz, dy); dgamma/dbeta (fp32) are accumulated in place."""
    H = x.shape[-1]
    rows = x.numel() // H
    _chk(x, "x"); _chk(y, "y", x.dtype); _chk(dout, "dout", x.dtype)
    dz = torch.empty_like(x)
    dy = torch.empty_like(x) if (y is not None and p_drop > 0.0) else None
    ws = _ln_ws.get(_ws_key(x.device))
    lib = _lib.lib()
    lib.vtx_layernorm_workspace_floats.restype = _lib.ctypes.c_long
    need = int(lib.vtx_layernorm_workspace_floats(c_int(H)))          # per-block dgamma / dbeta partials (the library's block cap x 2 x H)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, int(lib.vtx_layernorm_workspace_floats(c_int(2048)))), dtype=torch.float32, device=x.device)
        _ln_ws[_ws_key(x.device)] = ws
    call("vtx_layernorm_residual_bwd", c_int(dtype_code(x.dtype)), ptr(x), ptr(y), ptr(gamma),
         ptr(mean), ptr(rstd), ptr(dout), ptr(dz), ptr(dy), ptr(dgamma), ptr(dbeta), ptr(ws), c_int(rows),
         c_int(H), c_float(p_drop), c_u64(seed), stream_ptr(x))
    return dz, (dy if dy is not None else dz)


# ---------------------------------------------------------------------------------------
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
c_long = _lib.ctypes.c_long


_stat_ws = {}
STAT_WS_FLOATS = 32 * 1024 * 1024      # 128 MB: the epilogue partials [rows / 64 + 4][2][N] of a 1.6 M-row x 256-channel tensor (56 x 56 at 512 images per GPU)


def stat_workspace(device):
    ws = _stat_ws.get(_ws_key(device))
    if ws is None:
        ws = torch.empty(STAT_WS_FLOATS if device.type == "cuda" else 1 << 20, dtype=torch.float32, device=device)
        _stat_ws[_ws_key(device)] = ws
    return ws


class BnStats:
    """Handle to BatchNorm statistics emitted by a convolution epilogue (valid until the next one)."""

    def __init__(self, parts, strips, shift):
        self.parts, self.strips, self.shift = parts, strips, shift


def _stat_args(out_rows, N, bn_shift, device):
    """(parts ptr, shift ptr, strips int*, parts tensor) -- NULLs unless statistics are requested and fit."""
    if bn_shift is None:
        return ptr(None), ptr(None), None, None
    ws = stat_workspace(device)
    if ((out_rows + 63) // 64 + 4) * 2 * N > ws.numel():
        return ptr(None), ptr(None), None, None
    return ptr(ws), ptr(bn_shift), c_int(0), ws


def gemm_nt(a, b, bias=None, residual=None, act=ACT_NONE, want_preact=False, alpha=1.0,
            p_drop=0.0, seed=0, out=None, out_f32=False, bn_shift=None):
    """C[M,N] = dropout(act(alpha * a[M,K] @ b[N,K]^T + bias)) + residual.
    a, b: 2-D (row stride may exceed the row length).  Returns C (and preact if asked)."""
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K and b.dtype == a.dtype
    odt = torch.float32 if out_f32 else a.dtype
    if out is None:
        out = torch.empty(M, N, dtype=odt, device=a.device)
    assert out.dtype == odt
    pre = torch.empty(M, N, dtype=odt, device=a.device) if want_preact else None
    _chk(bias, "bias", torch.float32)
    if residual is not None:
        assert residual.dim() == 2 and residual.stride(1) == 1 and residual.dtype == odt
    sp, ss, strips, ws = _stat_args(M, N, bn_shift, a.device)
    call("vtx_gemm_nt", c_int(dtype_code(a.dtype)), c_int(M), c_int(N), c_int(K), ptr(a),
         c_long(a.stride(0)), ptr(b), c_long(b.stride(0)), ptr(out), c_long(out.stride(0)), ptr(bias),
         ptr(residual), c_long(residual.stride(0) if residual is not None else 0), ptr(pre),
         c_int(act), c_float(alpha), c_float(p_drop), c_u64(seed), c_int(1 if out_f32 else 0),
         sp, ss, (_lib.ctypes.byref(strips) if strips is not None else ptr(None)), stream_ptr(a))
    if bn_shift is not None:
        st = BnStats(ws, strips.value, bn_shift) if (strips is not None and strips.value > 0) else None
        return out, st
    return (out, pre) if want_preact else out


class _BnBwdFusion(_lib.ctypes.Structure):
    """VtxBnBwdFusion of include/virtex_amd.h."""
    _fields_ = [("x", _lib.ctypes.c_void_p), ("ymask", _lib.ctypes.c_void_p), ("mean", _lib.ctypes.c_void_p),
                ("rstd", _lib.ctypes.c_void_p), ("gamma", _lib.ctypes.c_void_p), ("beta", _lib.ctypes.c_void_p),
                ("parts", _lib.ctypes.c_void_p), ("parts_cap", _lib.ctypes.c_long), ("strips", _lib.ctypes.c_int),
                ("ybits", _lib.ctypes.c_void_p)]


class BnBwd:
    """What an input-gradient kernel needs to fuse the backward of the BatchNorm(+ReLU) that produced its input's
    gradient: that BatchNorm's input `x`, saved statistics, and the ReLU mask source -- `ymask` (post-ReLU block
    output), `ybits` (the same mask, one bit per element, from bn_fwd(want_bits=True): 1/16 of the bytes) or
    `gamma`/`beta` (mask recomputed from x) or none of them (no ReLU)."""

    def __init__(self, x, mean, rstd, ymask=None, gamma=None, beta=None, ybits=None):
        self.x, self.mean, self.rstd, self.ymask, self.gamma, self.beta = x, mean, rstd, ymask, gamma, beta
        self.ybits = ybits
        if ybits is not None:
            assert ybits.dtype == torch.uint8 and ybits.is_contiguous() and ybits.numel() * 8 == x.numel()
            self.ymask = None

    def descriptor(self, rows, N, device):
        cap = ((rows + 63) // 64 + 4) * 2 * N
        parts = torch.empty(cap, dtype=torch.float32, device=device)
        # addresses through ptr(): a launch recording (virtex_amd.replay) keeps every tensor a recorded argument points into
        a = lambda t: ptr(t).value      # noqa: E731  (None -> NULL)
        d = _BnBwdFusion(a(self.x), a(self.ymask), a(self.mean), a(self.rstd), a(self.gamma), a(self.beta), a(parts), cap, 0,
                         a(self.ybits))
        return d, parts


def gemm_nt_bnbwd(a, b, bn: BnBwd, residual=None):
    """dz[M,N] = mask(a[M,K] @ b[N,K]^T + residual) plus the BatchNorm-backward sums (see BnBwd).  Returns
    (out, stats): stats = BnStats(parts, strips, None), or None when the build / dtype did not fuse -- `out` is then
    the PLAIN gradient and the stand-alone bn_bwd (with its masks) must follow."""
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K and b.dtype == a.dtype
    out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    if residual is not None:
        assert residual.shape == out.shape and residual.is_contiguous() and residual.dtype == a.dtype
    assert bn.x.numel() == M * N and bn.x.is_contiguous() and (bn.ymask is None or bn.ymask.is_contiguous())
    d, parts = bn.descriptor(M, N, a.device)
    call("vtx_gemm_nt_bnbwd", c_int(dtype_code(a.dtype)), c_int(M), c_int(N), c_int(K), ptr(a), c_long(a.stride(0)),
         ptr(b), c_long(b.stride(0)), ptr(out), c_long(N), ptr(residual), c_long(N), _lib.ctypes.byref(d), stream_ptr(a))
    return out, (BnStats(parts, d.strips, None) if d.strips > 0 else None)


_splitk_ws = {}
SPLITK_WS_FLOATS = 64 * 1024 * 1024      # 256 MB of fp32 partial sums per (device, stream): room for the three or four contractions of a reduction batch


def splitk_workspace(device):
    """One scratch buffer per (device, stream): weight-gradient GEMMs may run on a side stream
    concurrently with the main stream's (virtex_amd/modules: `wgrad_stream`)."""
    key = _ws_key(device)
    ws = _splitk_ws.get(key)
    if ws is None:
        n = SPLITK_WS_FLOATS if device.type == "cuda" else 4 * 1024 * 1024
        ws = torch.empty(n, dtype=torch.float32, device=device)
        _splitk_ws[key] = ws
    return ws


class splitk_batch:
    """The split-K reductions of the weight-gradient contractions issued between begin() and end() -- on ONE stream, the one
    end() is called on -- are folded by a single launch at end() (vtx_splitk_batch_begin / _end; csrc/gemm.hip).  The gradients
    are complete only after end() (ops.splitk_flush in between for a result that is read inside the batch).
    Round 6: built, bit-identical, 68 -> 23 reduce launches per step -- and measured SLOWER (23.20 vs 23.08 ms per step, serial
    25.54 vs 25.41; profiles/r06_splitk_batch_rejected.txt): a reduction issued right behind its contraction reads partial tiles
    that are still in the 256-MB Infinity Cache and re-uses ONE workspace region for every contraction of the step; deferred to
    the end of the block it reads them from HBM, out of regions that together no longer fit the cache.  OFF by default
    (VIRTEX_AMD_SPLITK_BATCH=1 turns it on); kept with its test as the measured answer to "get rid of the reduce launches"."""
    enabled = os.environ.get("VIRTEX_AMD_SPLITK_BATCH", "0") != "0"

    def __init__(self, device):
        self.device = device
        self.open = False

    def begin(self):
        if splitk_batch.enabled and not self.open:
            call("vtx_splitk_batch_begin")
            self.open = True
        return self

    def end(self):
        if self.open:
            self.open = False
            call("vtx_splitk_batch_end", _lib.c_void_p(_lib.current_stream_handle(self.device) if self.device.type == "cuda" else 0))


def splitk_flush(device):
    """Inside a splitk_batch: run the pending reductions now (the results of the split-K contractions issued so far are read
    by what follows); a no-op outside a batch."""
    call("vtx_splitk_batch_flush", _lib.c_void_p(_lib.current_stream_handle(device) if device.type == "cuda" else 0))


def gemm_tn_acc(a, b, out, alpha=1.0, split_k=0):
    """out[M,N] (fp32) += alpha * a[K,M]^T @ b[K,N]."""
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    K, M = a.shape
    N = b.shape[1]
    assert b.shape[0] == K and b.dtype == a.dtype and out.dtype == torch.float32
    assert out.dim() == 2 and out.stride(1) == 1 and out.shape == (M, N)
    ws = splitk_workspace(a.device)
    call("vtx_gemm_tn_acc", c_int(dtype_code(a.dtype)), c_int(M), c_int(N), c_int(K), ptr(a),
         c_long(a.stride(0)), ptr(b), c_long(b.stride(0)), ptr(out), c_long(out.stride(0)),
         c_float(alpha), c_int(split_k), ptr(ws), c_long(ws.numel()), stream_ptr(a))
    return out


# ---------------------------------------------------------------------------------------
# NHWC convolutions.  x: (N,H,W,C) contiguous; w: (KO,R,S,C); wt: (C,R,S,KO).
def _conv_out(H, W, R, S, stride, pad):
    return (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1


def conv2d_fwd(x, w, stride, pad, bn_shift=None):
    N, H, W, C = x.shape
    KO, R, S, C2 = w.shape
    assert C2 == C and w.dtype == x.dtype
    _chk(x, "x"); _chk(w, "w")
    OH, OW = _conv_out(H, W, R, S, stride, pad)
    y = torch.empty(N, OH, OW, KO, dtype=x.dtype, device=x.device)
    sp, ss, strips, ws = _stat_args(N * OH * OW, KO, bn_shift, x.device)
    call("vtx_conv2d_fwd", c_int(dtype_code(x.dtype)), c_int(N), c_int(H), c_int(W), c_int(C), c_int(KO),
         c_int(R), c_int(S), c_int(stride), c_int(pad), ptr(x), ptr(w), ptr(y), sp, ss,
         (_lib.ctypes.byref(strips) if strips is not None else ptr(None)), stream_ptr(x))
    if bn_shift is not None:
        return y, (BnStats(ws, strips.value, bn_shift) if (strips is not None and strips.value > 0) else None)
    return y


def conv2d_dgrad(dy, wt, x_shape, stride, pad, residual=None, bn: "BnBwd" = None):
    """bn given: returns (dx, stats) like gemm_nt_bnbwd -- dx is the masked gradient when stats is not None."""
    N, H, W, C = x_shape
    C2, R, S, KO = wt.shape
    assert C2 == C and dy.shape[-1] == KO and wt.dtype == dy.dtype
    _chk(dy, "dy"); _chk(wt, "wt"); _chk(residual, "residual", dy.dtype)
    dx = torch.empty(N, H, W, C, dtype=dy.dtype, device=dy.device)
    if bn is not None:
        assert bn.x.numel() == dx.numel() and bn.x.is_contiguous() and (bn.ymask is None or bn.ymask.is_contiguous())
        d, parts = bn.descriptor(N * H * W, C, dy.device)
        call("vtx_conv2d_dgrad_bnbwd", c_int(dtype_code(dy.dtype)), c_int(N), c_int(H), c_int(W), c_int(C), c_int(KO),
             c_int(R), c_int(S), c_int(stride), c_int(pad), ptr(dy), ptr(wt), ptr(dx), ptr(residual),
             _lib.ctypes.byref(d), stream_ptr(dy))
        return dx, (BnStats(parts, d.strips, None) if d.strips > 0 else None)
    call("vtx_conv2d_dgrad", c_int(dtype_code(dy.dtype)), c_int(N), c_int(H), c_int(W), c_int(C), c_int(KO),
         c_int(R), c_int(S), c_int(stride), c_int(pad), ptr(dy), ptr(wt), ptr(dx), ptr(residual), stream_ptr(dy))
    return dx


def conv2d_wgrad(x, dy, dw, stride, pad, split_k=0):
    """dw (KO,R,S,C) fp32 += weight gradient."""
    N, H, W, C = x.shape
    KO, R, S, C2 = dw.shape
    assert C2 == C and dy.shape[-1] == KO and dw.dtype == torch.float32
    _chk(x, "x"); _chk(dy, "dy", x.dtype); _chk(dw, "dw")
    ws = splitk_workspace(x.device)
    call("vtx_conv2d_wgrad", c_int(dtype_code(x.dtype)), c_int(N), c_int(H), c_int(W), c_int(C), c_int(KO),
         c_int(R), c_int(S), c_int(stride), c_int(pad), ptr(x), ptr(dy), ptr(dw), c_int(split_k), ptr(ws),
         c_long(ws.numel()), stream_ptr(x))
    return dw


# ---------------------------------------------------------------------------------------
# BatchNorm (training) on NHWC, fused ReLU / residual.
_bn_ws = {}


def bn_workspace(device, C):
    """fp32 scratch for the per-strip partial sums; one buffer per (device, stream), shared by every BN call
    issued on that stream (they are ordered there)."""
    lib = _lib.lib()
    lib.vtx_bn_workspace_floats.restype = _lib.ctypes.c_long
    need = lib.vtx_bn_workspace_floats(c_int(C))
    ws = _bn_ws.get(_ws_key(device))
    if ws is None or ws.numel() < need:
        # zero-initialised ONCE: the first 64 words are the tickets of bn_fin2_kernel (each launch leaves them at zero again)
        ws = torch.zeros(max(need, lib.vtx_bn_workspace_floats(c_int(2048))), dtype=torch.float32, device=device)
        _bn_ws[_ws_key(device)] = ws
    return ws


def bn_fwd(x, gamma, beta, running_mean, running_var, nbt, eps=1e-5, momentum=0.1, relu=True,
           residual=None, stats=None, want_bits=False):
    """want_bits (bf16 + relu): also returns the ReLU mask as one bit per element (uint8 [numel/8]) -- what the
    input-gradient epilogue of the NEXT block reads instead of the whole output tensor (BnBwd(ybits=...))."""
    C = x.shape[-1]
    P = x.numel() // C
    _chk(x, "x"); _chk(residual, "residual", x.dtype)
    ws = bn_workspace(x.device, C)
    y = torch.empty_like(x)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    rstd = torch.empty(C, dtype=torch.float32, device=x.device)
    bits = None
    if want_bits:
        assert relu and x.dtype == torch.bfloat16 and x.numel() % 8 == 0
        bits = torch.empty(x.numel() // 8, dtype=torch.uint8, device=x.device)
    call("vtx_bn_fwd", c_int(dtype_code(x.dtype)), ptr(x), ptr(residual), ptr(gamma), ptr(beta),
         ptr(running_mean), ptr(running_var), ptr(nbt), ptr(y), ptr(mean), ptr(rstd), ptr(ws), c_int(P),
         c_int(C), c_float(eps), c_float(momentum), c_int(1 if relu else 0),
         ptr(stats.parts if stats else None), c_int(stats.strips if stats else 0),
         ptr(stats.shift if stats else None), ptr(bits), stream_ptr(x))
    return (y, mean, rstd, bits) if want_bits else (y, mean, rstd)


def bn_bwd(x, dy, ymask, gamma, mean, rstd, dgamma, dbeta, want_dz=False, relu_beta=None):
    """ymask: the post-ReLU output (needed when a residual was added before the ReLU); relu_beta: instead,
    recompute the mask from x (BN directly followed by ReLU) and skip reading the output tensor."""
    C = x.shape[-1]
    P = x.numel() // C
    _chk(x, "x"); _chk(dy, "dy", x.dtype); _chk(ymask, "ymask", x.dtype)
    ws = bn_workspace(x.device, C)
    dx = torch.empty_like(x)
    dz = torch.empty_like(x) if want_dz else None
    call("vtx_bn_bwd", c_int(dtype_code(x.dtype)), ptr(x), ptr(dy), ptr(ymask), ptr(gamma), ptr(relu_beta), ptr(mean),
         ptr(rstd), ptr(dx), ptr(dz), ptr(dgamma), ptr(dbeta), ptr(ws), c_int(P), c_int(C), stream_ptr(x))
    return (dx, dz) if want_dz else dx


def bn_bwd_fused(x, dz, gamma, mean, rstd, dgamma, dbeta, stats: "BnStats"):
    """BatchNorm backward when the kernel that produced `dz` already masked it and emitted the sums (`stats` from
    gemm_nt_bnbwd / conv2d_dgrad(bn=...)): finalize + one pass."""
    C = x.shape[-1]
    P = x.numel() // C
    _chk(x, "x"); _chk(dz, "dz", x.dtype)
    ws = bn_workspace(x.device, C)
    dx = torch.empty_like(x)
    call("vtx_bn_bwd_fused", c_int(dtype_code(x.dtype)), ptr(x), ptr(dz), ptr(gamma), ptr(mean), ptr(rstd),
         ptr(stats.parts), c_int(stats.strips), ptr(dx), ptr(dgamma), ptr(dbeta), ptr(ws), c_int(P), c_int(C), stream_ptr(x))
    return dx


def conv3_bwd_fused_supported(dz, wt):
    """May vtx_conv3_bwd_fused take this Bottleneck?  dz: (..., K) gradient wrt bn3's output, wt: (N, K)."""
    if dz.dtype != torch.bfloat16:
        return False
    K = dz.shape[-1]
    return bool(_lib.lib().vtx_conv3_bwd_fused_supported(c_int(dtype_code(dz.dtype)), c_int(dz.numel() // K), c_int(K), c_int(wt.shape[0])))


def conv3_bwd_fused(dz, x3, gamma3, mean3, rstd3, dgamma3, dbeta3, stats3: "BnStats", wt, bn2: "BnBwd"):
    """The backward of a Bottleneck's conv3 in one streaming kernel (csrc/conv3_bwd.hip): bn3's backward applied to `dz` on
    the fly (its sums are `stats3`; dgamma3 / dbeta3 accumulated), conv3's input gradient with bn2's mask and backward sums,
    and conv3's weight gradient as per-workgroup partials.  Returns (dy2, stats2, dw_parts, nparts): fold the partials into
    the fp32 weight gradient (K, N) with partials_reduce_acc -- normally on the weight-gradient stream."""
    K, N = dz.shape[-1], wt.shape[0]
    M = dz.numel() // K
    _chk(dz, "dz", torch.bfloat16); _chk(x3, "x3", torch.bfloat16); _chk(wt, "wt", torch.bfloat16)
    assert x3.numel() == dz.numel() and wt.shape == (N, K) and bn2.x.numel() == M * N and bn2.x.is_contiguous()
    assert bn2.gamma is not None and bn2.beta is not None and bn2.ymask is None and bn2.ybits is None
    lib = _lib.lib()
    nparts = lib.vtx_conv3_bwd_fused_parts(c_int(M))
    dw_parts = torch.empty(nparts, K, N, dtype=torch.float32, device=dz.device)
    dy2 = torch.empty(*dz.shape[:-1], N, dtype=dz.dtype, device=dz.device)
    d, parts = bn2.descriptor(M, N, dz.device)
    ws = bn_workspace(dz.device, K)
    got = c_int(0)
    call("vtx_conv3_bwd_fused", c_int(dtype_code(dz.dtype)), c_int(M), c_int(K), c_int(N), ptr(dz), ptr(x3), ptr(gamma3),
         ptr(mean3), ptr(rstd3), ptr(stats3.parts), c_int(stats3.strips), ptr(dgamma3), ptr(dbeta3), ptr(ws), ptr(wt),
         c_long(wt.stride(0)), _lib.ctypes.byref(d), ptr(dy2), ptr(dw_parts), c_long(dw_parts.numel()),
         _lib.ctypes.byref(got), stream_ptr(dz))
    return dy2, BnStats(parts, d.strips, None), dw_parts, nparts


def bn_bwd_fold(wt, gamma, mean, rstd, dgamma, dbeta, stats: "BnStats", P):
    """BatchNorm backward of a 1x1 convolution's output folded into the convolution's weights (csrc/bn_fold.hip): the finalize of
    `stats` (dgamma / dbeta accumulated) and, from wt [N][K] bf16: (wa = a0 o wt, wb = b1 o wt, bias [N], abc [3][K])."""
    N, K = wt.shape
    _chk(wt, "wt", torch.bfloat16)
    ws = bn_workspace(wt.device, K)
    wa, wb = torch.empty(N, K, dtype=wt.dtype, device=wt.device), torch.empty(N, K, dtype=wt.dtype, device=wt.device)
    bias = torch.empty(N, dtype=torch.float32, device=wt.device)
    abc = torch.empty(3, K, dtype=torch.float32, device=wt.device)
    call("vtx_bn_bwd_fold", ptr(gamma), ptr(mean), ptr(rstd), ptr(stats.parts), c_int(stats.strips), ptr(dgamma), ptr(dbeta),
         ptr(ws), c_int(P), c_int(K), ptr(wt), c_long(wt.stride(0)), c_int(N), ptr(wa), ptr(wb), ptr(bias), ptr(abc), stream_ptr(wt))
    return wa, wb, bias, abc


def wgrad_fold_combine(dw, T, WG, s, abc):
    """dw[k][n] += a0[k] T[k][n] + b1[k] WG[k][n] + c[k] s[n]  (abc from bn_bwd_fold)."""
    K, N = dw.shape
    assert dw.dtype == torch.float32 and dw.stride(1) == 1 and T.shape == (K, N) and WG.shape == (K, N) and s.shape == (N,)
    _chk(T, "T", torch.float32); _chk(WG, "WG", torch.float32); _chk(s, "s", torch.float32); _chk(abc, "abc", torch.float32)
    call("vtx_wgrad_fold_combine", ptr(dw), c_long(dw.stride(0)), ptr(T), ptr(WG), ptr(s), ptr(abc), c_int(K), c_int(N), stream_ptr(dw))
    return dw


def partials_reduce_acc(parts, nparts, out):
    """out (M, N) fp32 += sum of parts[:nparts] (each (M, N) fp32)."""
    M, N = out.shape
    assert out.dtype == torch.float32 and out.stride(1) == 1 and parts.dtype == torch.float32 and parts.is_contiguous()
    assert parts.numel() >= nparts * M * N
    call("vtx_partials_reduce_acc", ptr(parts), c_int(nparts), c_int(M), c_int(N), ptr(out), c_long(out.stride(0)), stream_ptr(out))
    return out


def bn_fwd_maxpool(x, gamma, beta, running_mean, running_var, nbt, eps=1e-5, momentum=0.1, stats=None):
    """Stem tail forward: BatchNorm (training) + ReLU + MaxPool2d(3,2,1) in one pass; the normalised tensor is never
    written.  Returns (pooled, argmax, mean, rstd)."""
    N, H, W, C = x.shape
    _chk(x, "x")
    ws = bn_workspace(x.device, C)
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(N, OH, OW, C, dtype=x.dtype, device=x.device)
    arg = torch.empty(N, OH, OW, C, dtype=torch.uint8, device=x.device)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    rstd = torch.empty(C, dtype=torch.float32, device=x.device)
    call("vtx_bn_fwd_maxpool", c_int(dtype_code(x.dtype)), ptr(x), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
         ptr(nbt), ptr(y), ptr(arg), ptr(mean), ptr(rstd), ptr(ws), c_int(N), c_int(H), c_int(W), c_int(C), c_float(eps),
         c_float(momentum), ptr(stats.parts if stats else None), c_int(stats.strips if stats else 0),
         ptr(stats.shift if stats else None), stream_ptr(x))
    return y, arg, mean, rstd


def bn_bwd_maxpool(x, dpool, argmax, gamma, beta, mean, rstd, dgamma, dbeta):
    """Stem tail: BatchNorm+ReLU backward with the max-pool backward gathered on the fly (no pre-pool gradient tensor)."""
    N, H, W, C = x.shape
    _chk(x, "x"); _chk(dpool, "dpool", x.dtype); _chk(argmax, "argmax", torch.uint8)
    ws = bn_workspace(x.device, C)
    dx = torch.empty_like(x)
    call("vtx_bn_bwd_maxpool", c_int(dtype_code(x.dtype)), ptr(x), ptr(dpool), ptr(argmax), ptr(gamma), ptr(beta), ptr(mean),
         ptr(rstd), ptr(dx), ptr(dgamma), ptr(dbeta), ptr(ws), c_int(N), c_int(H), c_int(W), c_int(C), stream_ptr(x))
    return dx


def maxpool_fwd(x):
    N, H, W, C = x.shape
    _chk(x, "x")
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(N, OH, OW, C, dtype=x.dtype, device=x.device)
    arg = torch.empty(N, OH, OW, C, dtype=torch.uint8, device=x.device)
    call("vtx_maxpool3x3s2_fwd", c_int(dtype_code(x.dtype)), ptr(x), ptr(y), ptr(arg), c_int(N), c_int(H),
         c_int(W), c_int(C), stream_ptr(x))
    return y, arg


def maxpool_bwd(dy, arg, x_shape):
    N, H, W, C = x_shape
    _chk(dy, "dy")
    dx = torch.empty(N, H, W, C, dtype=dy.dtype, device=dy.device)
    call("vtx_maxpool3x3s2_bwd", c_int(dtype_code(dy.dtype)), ptr(dy), ptr(arg), ptr(dx), c_int(N), c_int(H),
         c_int(W), c_int(C), stream_ptr(dy))
    return dx


def image_to_nhwc(image, dtype, cpad=8, halo=0):
    """fp32 NCHW (B,3,H,W) -> NHWC `dtype` with channels zero-padded to `cpad` (and a zero frame of `halo` pixels)."""
    N, Cin, H, W = image.shape
    _chk(image, "image", torch.float32)
    out = torch.empty(N, H + 2 * halo, W + 2 * halo, cpad, dtype=dtype, device=image.device)
    call("vtx_image_to_nhwc_halo", c_int(dtype_code(dtype)), ptr(image), ptr(out), c_int(N), c_int(Cin), c_int(H),
         c_int(W), c_int(cpad), c_int(halo), stream_ptr(image))
    return out


def weight_prep(w32, dtype, cpad=None, want_w=True, want_wt=True):
    """fp32 [KO, T, C] (or [KO, C]) -> (w [KO,T,Cp], wt [Cp,T,KO]) in `dtype`."""
    if w32.dim() == 2:
        w32 = w32.unsqueeze(1)
    KO, T, C = w32.shape
    _chk(w32, "w32", torch.float32)
    Cp = cpad or C
    w = torch.empty(KO, T, Cp, dtype=dtype, device=w32.device) if want_w else None
    wt = torch.empty(Cp, T, KO, dtype=dtype, device=w32.device) if want_wt else None
    call("vtx_weight_prep", c_int(dtype_code(dtype)), ptr(w32), ptr(w), ptr(wt), c_int(KO), c_int(T), c_int(C),
         c_int(Cp), stream_ptr(w32))
    return w, wt


IMAGENET_COLOR_MEAN = (0.485, 0.456, 0.406)      # reference: virtex/data/transforms.py:85-89
IMAGENET_COLOR_STD = (0.229, 0.224, 0.225)


def image_u8_to_nhwc(images, dtype, cpad, size=None, crop_xy=None, flip=None, mean=IMAGENET_COLOR_MEAN,
                     std=IMAGENET_COLOR_STD, halo=0):
    """uint8 (N, Hs, Ws, 3) -> normalised (N, H, W, cpad) in `dtype`; H = W = size (default: the whole image).
    crop_xy: int32 (N, 2) window origins {x0, y0}; flip: uint8 (N,) horizontal-flip flags."""
    assert images.dtype == torch.uint8 and images.dim() == 4 and images.shape[-1] == 3 and images.is_contiguous()
    N, Hs, Ws, _ = images.shape
    H = W = size if size is not None else None
    if size is None:
        H, W = Hs, Ws
    if crop_xy is not None:
        assert crop_xy.dtype == torch.int32 and tuple(crop_xy.shape) == (N, 2) and crop_xy.is_contiguous()
    if flip is not None:
        assert flip.dtype == torch.uint8 and tuple(flip.shape) == (N,)
    out = torch.empty(N, H + 2 * halo, W + 2 * halo, cpad, dtype=dtype, device=images.device)
    ctypes = _lib.ctypes
    m = (ctypes.c_float * 3)(*mean)
    sd = (ctypes.c_float * 3)(*std)
    call("vtx_image_u8_to_nhwc", c_int(dtype_code(dtype)), ptr(images), ptr(out), c_int(N), c_int(Hs), c_int(Ws), c_int(H),
         c_int(W), c_int(cpad), c_int(halo), ptr(crop_xy), ptr(flip), m, sd, stream_ptr(images))
    return out


def image_augment_u8(images, params, size, dtype, cpad, halo=0, mean=IMAGENET_COLOR_MEAN, std=IMAGENET_COLOR_STD):
    """uint8 (N, Hs, Ws, 3) + per-image VtxAugParams records (uint8 tensor of N*40 bytes on the device) -> augmented,
    normalised (N, size+2*halo, size+2*halo, cpad) in `dtype` (vtx_image_augment_u8)."""
    assert images.dtype == torch.uint8 and images.dim() == 4 and images.shape[-1] == 3 and images.is_contiguous()
    N, Hs, Ws, _ = images.shape
    assert params.dtype == torch.uint8 and params.numel() == N * 40 and params.is_contiguous()
    out = torch.empty(N, size + 2 * halo, size + 2 * halo, cpad, dtype=dtype, device=images.device)
    scratch = torch.empty(N, dtype=torch.float32, device=images.device)
    ctypes = _lib.ctypes
    m = (ctypes.c_float * 3)(*mean)
    sd = (ctypes.c_float * 3)(*std)
    call("vtx_image_augment_u8", c_int(dtype_code(dtype)), ptr(images), ptr(out), ptr(params), ptr(scratch), c_int(N), c_int(Hs),
         c_int(Ws), c_int(size), c_int(cpad), c_int(halo), m, sd, stream_ptr(images))
    return out


# ---- compute copies of the master weights, cached on the parameter until its version changes -----------
def _w32_view(param):
    """fp32 master weight as [KO, T, C]: conv weights are stored (KO,R,S,C) physically (channels_last)."""
    t = param.detach()
    if t.dim() == 4:
        KO, C, R, S = t.shape
        return t.permute(0, 2, 3, 1).contiguous().view(KO, R * S, C)
    return t.contiguous().view(t.shape[0], 1, t.shape[1])


def _prep_entry(param, dtype, cpad):
    store = param.__dict__.setdefault("_vtx_prep", {})
    return store, (dtype, cpad)


def prepped(param, dtype, cpad=None, want_w=True, want_wt=True):
    """(w [KO,T,Cp], wt [Cp,T,KO]) compute copies of `param` in `dtype`; cached on the parameter object and
    recomputed when its version counter moves (optimizer step, load_state_dict, in-place edits)."""
    w32 = None
    store, key = _prep_entry(param, dtype, cpad)
    e = store.get(key)
    stamp = (param._version, param.data_ptr())
    if e is not None and e[0] == stamp and (e[1] is not None or not want_w) and (e[2] is not None or not want_wt):
        return e[1], e[2]
    w32 = _w32_view(param)
    _chk(w32, "weight", torch.float32)
    C = w32.shape[2]
    if dtype == torch.float32 and (cpad or C) == C:
        w = w32                                        # the master weight is its own compute copy
        wt = weight_prep(w32, dtype, want_w=False)[1] if want_wt else None
    else:
        w, wt = weight_prep(w32, dtype, cpad=cpad, want_w=want_w, want_wt=want_wt)
    if e is not None and e[0] == stamp:                # keep what was already there (e.g. w) when only wt was missing
        w = w if w is not None else e[1]
        wt = wt if wt is not None else e[2]
    store[key] = (stamp, w, wt)
    return w, wt


_prep_tables = {}


def prep_many(items, dtype):
    """Refresh the compute copies of many weights with ONE kernel launch.  items: (param, cpad, want_wt).
    Weights whose cached copies are current are skipped; output buffers are reused from step to step, so the
    device-side descriptor table is built once per (set of weights)."""
    import numpy as np
    stale = []
    for (param, cpad, want_wt) in items:
        store, key = _prep_entry(param, dtype, cpad)
        e = store.get(key)
        if e is not None and e[0] == (param._version, param.data_ptr()) and (e[2] is not None or not want_wt):
            continue
        if e is not None and e[1] is not None and e[1].device != param.device:
            e = None                                   # the module moved: old copies are on another device
        stale.append((param, cpad, want_wt, store, key, e))
    if not stale:
        return 0
    if len(stale) == 1:
        prepped(stale[0][0], dtype, stale[0][1], True, stale[0][2])
        return 1
    dev = stale[0][0].device
    descs, starts, keep, total = [], [], [], 0
    for (param, cpad, want_wt, store, key, e) in stale:
        w32 = _w32_view(param)
        KO, T, C = w32.shape
        Cp = cpad or C
        alias = dtype == torch.float32 and Cp == C
        w = w32 if alias else (e[1] if (e is not None and e[1] is not None and e[1].data_ptr() != w32.data_ptr()) else
                               torch.empty(KO, T, Cp, dtype=dtype, device=dev))
        wt = None
        if want_wt:
            wt = e[2] if (e is not None and e[2] is not None) else torch.empty(Cp, T, KO, dtype=dtype, device=dev)
        if alias and not want_wt:
            store[key] = ((param._version, param.data_ptr()), w, None)
            continue
        descs.append((w32.data_ptr(), 0 if alias else w.data_ptr(), wt.data_ptr() if wt is not None else 0, KO, T, C, Cp))
        starts.append(total)
        total += ((Cp + 31) // 32) * ((KO + 31) // 32) * T
        keep.append((store, key, param, w32, w, wt))
    if descs:
        sig = (dtype, tuple(descs))
        tab = _prep_tables.get(sig)
        if tab is None:
            if len(_prep_tables) > 16:
                _prep_tables.clear()
            arr = np.zeros(len(descs), dtype=np.dtype([("w32", "<u8"), ("w", "<u8"), ("wt", "<u8"), ("KO", "<i4"), ("T", "<i4"),
                                                       ("C", "<i4"), ("Cp", "<i4")]))
            for i, d in enumerate(descs):
                arr[i] = d
            tab = (torch.from_numpy(arr.view(np.uint8).copy()).to(dev), torch.tensor(starts, dtype=torch.int32, device=dev))
            _prep_tables[sig] = tab
        call("vtx_weight_prep_batched", c_int(dtype_code(dtype)), ptr(tab[0]), ptr(tab[1]), c_int(len(descs)), c_int(total),
             stream_ptr(keep[0][3]))
    for (store, key, param, w32, w, wt) in keep:
        store[key] = ((param._version, param.data_ptr()), w, wt)
    return len(stale)


def bn_fold(w32, gamma, beta, running_mean, running_var, eps, dtype, cpad=None):
    """Eval-mode BatchNorm folded into its convolution: fp32 [KO,T,C] -> (w [KO,T,Cp] in `dtype`, bias [KO] fp32)."""
    if w32.dim() == 2:
        w32 = w32.unsqueeze(1)
    KO, T, C = w32.shape
    for t, n in ((w32, "w32"), (gamma, "gamma"), (beta, "beta"), (running_mean, "running_mean"), (running_var, "running_var")):
        _chk(t, n, torch.float32)
    Cp = cpad or C
    w = torch.empty(KO, T, Cp, dtype=dtype, device=w32.device)
    bias = torch.empty(KO, dtype=torch.float32, device=w32.device)
    call("vtx_bn_fold", c_int(dtype_code(dtype)), ptr(w32), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
         c_float(eps), ptr(w), ptr(bias), c_int(KO), c_int(T), c_int(C), c_int(Cp), stream_ptr(w32))
    return w, bias


def conv2d_infer(x, w, bias, stride, pad, relu=False, residual=None):
    """y = act(conv(x, w) + bias (+ residual)) on NHWC; ReLU (if any) comes after the residual add."""
    N, H, W, C = x.shape
    KO, R, S, C2 = w.shape
    assert C2 == C and w.dtype == x.dtype
    _chk(x, "x"); _chk(w, "w"); _chk(bias, "bias", torch.float32)
    OH, OW = _conv_out(H, W, R, S, stride, pad)
    y = torch.empty(N, OH, OW, KO, dtype=x.dtype, device=x.device)
    if residual is not None:
        _chk(residual, "residual", x.dtype)
        assert residual.shape == y.shape
    call("vtx_conv2d_infer", c_int(dtype_code(x.dtype)), c_int(N), c_int(H), c_int(W), c_int(C), c_int(KO),
         c_int(R), c_int(S), c_int(stride), c_int(pad), ptr(x), ptr(w), ptr(bias), ptr(residual),
         c_int(1 if relu else 0), ptr(y), stream_ptr(x))
    return y


def cast_from_f32(src, dtype):
    _chk(src, "src", torch.float32)
    out = torch.empty(src.shape, dtype=dtype, device=src.device)
    call("vtx_cast_from_f32", c_int(dtype_code(dtype)), ptr(src), ptr(out), c_long(src.numel()), stream_ptr(src))
    return out


# ---------------------------------------------------------------------------------------
def embedding_fwd(tokens, words, positions, gamma, beta, dtype, padding_idx=0, eps=1e-8, p_drop=0.0, seed=0):
    B, T = tokens.shape
    V, H = words.shape
    _chk(tokens, "tokens", torch.int64); _chk(words, "words", torch.float32); _chk(positions, "positions", torch.float32)
    assert positions.shape[0] >= T
    out = torch.empty(B, T, H, dtype=dtype, device=tokens.device)
    mean = torch.empty(B * T, dtype=torch.float32, device=tokens.device)
    rstd = torch.empty(B * T, dtype=torch.float32, device=tokens.device)
    call("vtx_embedding_fwd", c_int(dtype_code(dtype)), ptr(tokens), ptr(words), ptr(positions), ptr(gamma),
         ptr(beta), ptr(out), ptr(mean), ptr(rstd), c_int(B), c_int(T), c_int(H), c_int(V), c_int(padding_idx),
         c_float(eps), c_float(p_drop), c_u64(seed), stream_ptr(tokens))
    return out, mean, rstd


def embedding_bwd(tokens, words, positions, gamma, mean, rstd, dout, dwords, dpositions, dgamma, dbeta,
                  padding_idx=0, p_drop=0.0, seed=0):
    B, T = tokens.shape
    V, H = words.shape
    _chk(dout, "dout")
    call("vtx_embedding_bwd", c_int(dtype_code(dout.dtype)), ptr(tokens), ptr(words), ptr(positions), ptr(gamma),
         ptr(mean), ptr(rstd), ptr(dout), ptr(dwords), ptr(dpositions), ptr(dgamma), ptr(dbeta), c_int(B),
         c_int(T), c_int(H), c_int(V), c_int(padding_idx), c_float(p_drop), c_u64(seed), stream_ptr(tokens))


# ---------------------------------------------------------------------------------------
def attention_fwd(q, k, v, B, heads, T, S, causal, key_lengths=None, p_drop=0.0, seed=0):
    """q: [B*T, >=heads*64] view, k/v: [B*S, ...] views (unit column stride). Returns o [B*T, heads*64]."""
    for t in (q, k, v):
        assert t.dim() == 2 and t.stride(1) == 1
    o = torch.empty(B * T, heads * 64, dtype=q.dtype, device=q.device)
    call("vtx_attention_fwd", c_int(dtype_code(q.dtype)), ptr(q), c_long(q.stride(0)), ptr(k), c_long(k.stride(0)),
         ptr(v), c_long(v.stride(0)), ptr(o), c_long(o.stride(0)), c_int(B), c_int(heads), c_int(T), c_int(S),
         c_int(64), c_int(1 if causal else 0), ptr(key_lengths), c_float(p_drop), c_u64(seed), stream_ptr(q))
    return o


def attention_bwd(q, k, v, dout, dq, dk, dv, B, heads, T, S, causal, key_lengths=None, p_drop=0.0, seed=0):
    for t in (q, k, v, dout, dq, dk, dv):
        assert t.dim() == 2 and t.stride(1) == 1
    call("vtx_attention_bwd", c_int(dtype_code(q.dtype)), ptr(q), c_long(q.stride(0)), ptr(k), c_long(k.stride(0)),
         ptr(v), c_long(v.stride(0)), ptr(dout), c_long(dout.stride(0)), ptr(dq), c_long(dq.stride(0)), ptr(dk),
         c_long(dk.stride(0)), ptr(dv), c_long(dv.stride(0)), c_int(B), c_int(heads), c_int(T), c_int(S), c_int(64),
         c_int(1 if causal else 0), ptr(key_lengths), c_float(p_drop), c_u64(seed), stream_ptr(q))


# ---------------------------------------------------------------------------------------
def cross_entropy_fwd(logits, targets, ignore_index=0):
    """logits fp32 [R, V]; targets int64 [R]. Returns (loss_and_count[2], lse[R])."""
    R, V = logits.shape
    assert logits.dtype == torch.float32 and logits.stride(1) == 1
    _chk(targets, "targets", torch.int64)
    lse = torch.empty(R, dtype=torch.float32, device=logits.device)
    row_loss = torch.empty(R, dtype=torch.float32, device=logits.device)
    lc = torch.empty(2, dtype=torch.float32, device=logits.device)
    call("vtx_cross_entropy_fwd", ptr(logits), c_long(logits.stride(0)), ptr(targets), ptr(lse), ptr(row_loss),
         ptr(lc), c_int(R), c_int(V), c_int(ignore_index), stream_ptr(logits))
    return lc, lse


def cross_entropy_bwd(logits, targets, lse, lc, grad_out, dtype, ignore_index=0):
    R, V = logits.shape
    _chk(grad_out, "grad_out", torch.float32)
    d = torch.empty(R, V, dtype=dtype, device=logits.device)
    call("vtx_cross_entropy_bwd", c_int(dtype_code(dtype)), ptr(logits), c_long(logits.stride(0)), ptr(targets),
         ptr(lse), ptr(lc), ptr(grad_out), ptr(d), c_long(V), c_int(R), c_int(V), c_int(ignore_index),
         stream_ptr(logits))
    return d


_tied_ce_ws = {}


def tied_ce_fwd(hidden, weight, bias, targets, ignore_index=0):
    """Tied projection + cross-entropy forward without logits: hidden [R,H], weight [V,H] (same dtype), bias fp32 [V],
    targets int64 [R].  Returns (loss_and_count[2], lse[R])."""
    assert hidden.dim() == 2 and weight.dim() == 2 and hidden.stride(1) == 1 and weight.stride(1) == 1
    R, H = hidden.shape
    V = weight.shape[0]
    assert weight.shape[1] == H and weight.dtype == hidden.dtype
    _chk(targets, "targets", torch.int64); _chk(bias, "bias", torch.float32)
    lib = _lib.lib()
    lib.vtx_tied_ce_partial_floats.restype = _lib.ctypes.c_long
    need = lib.vtx_tied_ce_partial_floats(c_int(R), c_int(V))
    key = _ws_key(hidden.device)
    ws = _tied_ce_ws.get(key)
    if ws is None or ws.numel() < 2 * need + 2 * R:
        ws = torch.empty(2 * need + 2 * R, dtype=torch.float32, device=hidden.device)
        _tied_ce_ws[key] = ws
    pmax, psum, tgt, row_loss = ws[:need], ws[need: 2 * need], ws[2 * need: 2 * need + R], ws[2 * need + R: 2 * need + 2 * R]
    lse = torch.empty(R, dtype=torch.float32, device=hidden.device)
    lc = torch.empty(2, dtype=torch.float32, device=hidden.device)
    call("vtx_tied_ce_fwd", c_int(dtype_code(hidden.dtype)), c_int(R), c_int(V), c_int(H), ptr(hidden), c_long(hidden.stride(0)),
         ptr(weight), c_long(weight.stride(0)), ptr(bias), ptr(targets), c_int(ignore_index), ptr(pmax), ptr(psum), c_long(need),
         ptr(tgt), ptr(lse), ptr(row_loss), ptr(lc), stream_ptr(hidden))
    return lc, lse


def tied_ce_bwd(hidden, weight, bias, targets, lse, lc, grad_out, ignore_index=0):
    """d(logits) [R,V] in the compute dtype, recomputing the projection (the logits are never stored)."""
    R, H = hidden.shape
    V = weight.shape[0]
    _chk(grad_out, "grad_out", torch.float32)
    d = torch.empty(R, V, dtype=hidden.dtype, device=hidden.device)
    call("vtx_tied_ce_bwd", c_int(dtype_code(hidden.dtype)), c_int(R), c_int(V), c_int(H), ptr(hidden), c_long(hidden.stride(0)),
         ptr(weight), c_long(weight.stride(0)), ptr(bias), ptr(targets), c_int(ignore_index), ptr(lse), ptr(lc), ptr(grad_out),
         ptr(d), stream_ptr(hidden))
    return d


def beam_step(logits, last, score_in, images, per_node, beam, eos):
    """One beam-search step on the device (vtx_beam_step).  logits fp32 [rows, V]; last int64 [rows] or None (first
    step); score_in fp32 [rows] or None.  Returns (score [images, beam], parent [images, beam], token [images, beam])."""
    rows, V = logits.shape
    assert logits.dtype == torch.float32 and logits.stride(1) == 1 and rows % images == 0
    _chk(last, "last", torch.int64); _chk(score_in, "score_in", torch.float32)
    dev = logits.device
    cand_lp = torch.empty(rows, per_node, dtype=torch.float32, device=dev)
    cand_tok = torch.empty(rows, per_node, dtype=torch.int64, device=dev)
    score = torch.empty(images, beam, dtype=torch.float32, device=dev)
    parent = torch.empty(images, beam, dtype=torch.int64, device=dev)
    token = torch.empty(images, beam, dtype=torch.int64, device=dev)
    call("vtx_beam_step", ptr(logits), c_long(logits.stride(0)), ptr(last), ptr(score_in), c_int(images), c_int(rows // images),
         c_int(V), c_int(eos), c_int(per_node), c_int(beam), ptr(cand_lp), ptr(cand_tok), ptr(score), ptr(parent), ptr(token),
         stream_ptr(logits))
    return score, parent, token


_colsum_ws = {}


def colsum_acc(x, out):
    assert x.dim() == 2 and x.stride(1) == 1 and out.dtype == torch.float32
    C = x.shape[1]
    ws = _colsum_ws.get(_ws_key(x.device))
    if ws is None or ws.numel() < 256 * C:
        ws = torch.empty(256 * max(C, 10000), dtype=torch.float32, device=x.device)
        _colsum_ws[_ws_key(x.device)] = ws
    call("vtx_colsum_acc", c_int(dtype_code(x.dtype)), ptr(x), c_long(x.stride(0)), ptr(out), ptr(ws),
         c_int(x.shape[0]), c_int(C), stream_ptr(x))
    return out


def add(a, b, out=None):
    _chk(a, "a"); _chk(b, "b", a.dtype)
    if out is None:
        out = torch.empty_like(a)
    call("vtx_add", c_int(dtype_code(a.dtype)), ptr(a), ptr(b), ptr(out), c_long(a.numel()), stream_ptr(a))
    return out


def gelu_bwd(h, da, p_drop=0.0, seed=0):
    _chk(h, "h"); _chk(da, "da", h.dtype)
    dh = torch.empty_like(h)
    call("vtx_gelu_bwd", c_int(dtype_code(h.dtype)), ptr(h), ptr(da), ptr(dh), c_long(h.numel()), c_float(p_drop),
         c_u64(seed), stream_ptr(h))
    return dh


def dropout_bwd(dx, p_drop, seed):
    """dy = mask(seed) * dx / (1 - p): the gradient through x + dropout(y) (pre-norm decoder sub-layers); p = 0: dx itself."""
    if p_drop <= 0.0:
        return dx
    _chk(dx, "dx")
    dy = torch.empty_like(dx)
    call("vtx_dropout_bwd", c_int(dtype_code(dx.dtype)), ptr(dx), ptr(dy), c_long(dx.numel()), c_float(p_drop), c_u64(seed),
         stream_ptr(dx))
    return dy


# ---------------------------------------------------------------------------------------
def sumsq(x, partials, out):
    call("vtx_sumsq", ptr(x), c_long(x.numel()), ptr(partials), ptr(out), stream_ptr(x))
    return out


def sgd_lookahead_step(p, g, m, slow, chunk_off, chunk_len, chunk_seg, seg_lr, seg_wd, lr_mult, momentum,
                       grad_scale, sumsq_buf, max_norm, do_lookahead, alpha, n_elems=0):
    """n_elems: sum of chunk_len (the profiler's algorithmic byte count; 0 = unknown)"""
    call("vtx_sgd_lookahead_step", ptr(p), ptr(g), ptr(m), ptr(slow), ptr(chunk_off), ptr(chunk_len),
         ptr(chunk_seg), c_int(chunk_off.numel()), c_long(int(n_elems)), ptr(seg_lr), ptr(seg_wd), c_float(lr_mult), c_float(momentum),
         c_float(grad_scale), ptr(sumsq_buf), c_float(max_norm if max_norm else 0.0),
         c_int(1 if do_lookahead else 0), c_float(alpha), stream_ptr(p))


def sgd_lookahead_step_dev(p, g, m, slow, chunk_off, chunk_len, chunk_seg, seg_lr, seg_wd, sched, momentum, grad_scale,
                           sumsq_buf, max_norm, alpha, n_elems=0):
    """sched: fp32[2] on the device = {LR multiplier, Lookahead-sync flag} of this step (graph-capturable form)"""
    _chk(sched, "sched", torch.float32)
    call("vtx_sgd_lookahead_step_dev", ptr(p), ptr(g), ptr(m), ptr(slow), ptr(chunk_off), ptr(chunk_len),
         ptr(chunk_seg), c_int(chunk_off.numel()), c_long(int(n_elems)), ptr(seg_lr), ptr(seg_wd), ptr(sched), c_float(momentum),
         c_float(grad_scale), ptr(sumsq_buf), c_float(max_norm if max_norm else 0.0), c_float(alpha), stream_ptr(p))


def set_dropout_epoch(t):
    """t: int32[1] device tensor (or None) mixed into every dropout seed on the device; the caller increments it per step"""
    call("vtx_set_dropout_epoch", ptr(t))


# ---------------------------------------------------------------------------------------
# per-launch timing of the contraction kernels (bench.py's roofline leg)
def profile_start(only_class=-1):
    """only_class: index (the "cls" field of profile_stop()'s records) of the one kernel class to time, -1 = all."""
    call("vtx_profile_select", c_int(only_class))
    call("vtx_profile_start")


def profile_stop():
    """-> list of {"name", "launches", "seconds", "flops", "bytes"} for every kernel class that was launched."""
    ctypes = _lib.ctypes
    lib = _lib.lib()
    n = lib.vtx_profile_stop()
    if n < 0:
        raise _lib.VtxError(lib.vtx_last_error().decode())
    out = []
    for i in range(n):
        name = ctypes.create_string_buffer(512)
        launches = ctypes.c_long(0)
        sec, fl, by = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
        call("vtx_profile_get", c_int(i), name, c_int(512), ctypes.byref(launches), ctypes.byref(sec),
             ctypes.byref(fl), ctypes.byref(by))
        if launches.value:
            out.append({"cls": i, "name": name.value.decode(), "launches": launches.value, "seconds": sec.value,
                        "flops": fl.value, "bytes": by.value})
    return out
