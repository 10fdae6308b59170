"""MI355X-native visual backbone: a drop-in for the reference's ``TorchvisionVisualBackbone``
(/root/reference/virtex/modules/visual_backbones.py:20-74).

Same constructor, same ``visual_feature_size`` attribute, same ``.cnn`` sub-module tree with
torchvision's parameter/buffer names (``conv1.weight``, ``bn1.*``, ``layer{1-4}.{i}.*``,
``layer*.0.downsample.{0,1}.*``) so reference checkpoints load unchanged and downstream code
that touches ``model.visual.cnn`` keeps working.  ``forward(image)`` takes the reference's
(B,3,H,W) fp32 batch and returns logical (B,C,h,w) features -- physically NHWC, which the
reference's own ``.view(B,C,-1).permute(0,2,1)`` consumes without a copy.

Nothing here computes with torch operators: the whole ResNet forward and backward is a
hand-scheduled sequence of C-ABI kernel launches (virtex_amd/ops.py) on the caller's HIP
stream -- NHWC implicit-GEMM convolutions on MFMA, fused BatchNorm+ReLU(+residual), maxpool.
torch supplies device memory and the autograd edge only.
"""
import os


import torch
from ..replay import traced_backward
from torch import nn

from .. import gradsink, ops

RESNET_BLOCKS = {"resnet50": ((3, 4, 6, 3), 64), "resnet101": ((3, 4, 23, 3), 64),
                 "wide_resnet50_2": ((3, 4, 6, 3), 128)}
STEM_CPAD = 8  # generic stem layout: the 3 input channels zero-padded to 8 (one 16-byte bf16 vector)
# Packed stem layout (even image widths): pixels padded to 4 channels, a zero frame of STEM_HALO pixels written by the
# input conversion, filter padded 7x7 -> 7x8.  The 7x7/s2/p3 convolution becomes a "valid" 7x8/s2 one whose 16-byte
# chunks hold two adjacent pixels: K = 7*8*4 = 224 instead of 7*7*8 = 392, no bounds logic, half the input bytes.
STEM_PACK_C, STEM_PACK_S, STEM_HALO = 4, 8, 3
# BatchNorm statistics from the convolution epilogue (ops.conv2d_fwd(bn_shift=...)).  Round 1 accumulated them in
# the accumulator layout (~48 extra VGPRs: the GEMMs lost more than the saved 5.7 GB read, 43.8 vs 42.1 ms); the
# round-2 epilogue takes them while the wave-private strip is drained (16 accumulators, parameters in LDS, same
# VGPR count as the plain kernel): 35.7 -> 35.0 ms alone, 33.4 -> 32.8 ms on top of the backward fusion.
FUSE_BN_STATS = os.environ.get("VIRTEX_AMD_FUSE_BN_STATS", "1") != "0"
# BatchNorm BACKWARD fused into the input-gradient kernels (bf16): the epilogue of the kernel that produces the
# gradient wrt a BatchNorm(+ReLU) output applies the ReLU mask, stores the masked gradient and emits the two sums
# (ops.BnBwd); the stand-alone reduction over (dy, x [, y]) and the mask / dz passes disappear.  Applies to bn1, bn2
# (mask recomputed from x) and bn3 (mask = block output) of every Bottleneck; shortcut BatchNorms, the stem and the
# last block (whose gradient comes from the text heads) keep the stand-alone kernels.
FUSE_BN_BWD = os.environ.get("VIRTEX_AMD_FUSE_BN_BWD", "1") != "0"
# the stem's tail (max-pool backward -> ReLU mask -> BatchNorm backward) in two passes without the pre-pool gradient
# tensor (vtx_bn_bwd_maxpool).  Rounds 1-2: slower than the three-kernel path, because the gather was a chain of memory
# latencies (a wait inside the branch around every candidate window).  Round 3: a thread owns a 2 x 2 pixel quad = four
# windows, everything requested before anything is used (pool_windows.h): 312 us of kernels instead of 618 at the very
# end of the backward pass, 26.00 -> 25.69 ms/step (profiles/r03_ab_session10.txt) -> on by default.
FUSE_STEM_TAIL = os.environ.get("VIRTEX_AMD_FUSE_STEM_TAIL", "1") != "0"
# forward counterpart: BatchNorm + ReLU + max-pool of the stem in one pass, the 411 MB tensor between them never written
# (pooled values and argmax bit-identical to the three-kernel path).  Round 3, interleaved A/B in one process:
# 27.72 -> 27.56 ms/step (profiles/r03_ab_session1.txt) -> on by default.
FUSE_STEM_FWD = os.environ.get("VIRTEX_AMD_FUSE_STEM_FWD", "1") != "0"
# The ReLU mask of a Bottleneck's output as one BIT per element, written by the BatchNorm + residual + ReLU pass that
# produces the output: the fused BatchNorm backward in the next block's input-gradient epilogue reads 1/16 of the bytes
# instead of the whole output tensor (2.8 GB of reads per step at bs 256; the output itself stays: it is the next
# block's input).
RELU_BITS = os.environ.get("VIRTEX_AMD_RELU_BITS", "1") != "0"
# The backward of conv3 of the stage-1 Bottlenecks as one streaming kernel (csrc/conv3_bwd.hip): bn3's backward applied while
# the gradient is loaded, conv3's input gradient with bn2's fused backward epilogue, conv3's weight gradient as per-workgroup
# partials -- the 411-MB gradient wrt conv3's output (written by one pass, re-read by two kernels) never exists.
FUSE_CONV3_BWD = os.environ.get("VIRTEX_AMD_FUSE_CONV3_BWD", "1") != "0"
# bn3's backward FOLDED INTO conv3's WEIGHTS for the Bottlenecks the streaming kernel does not take (stages 2-4; csrc/bn_fold.hip):
# BatchNorm backward is affine per channel and conv3's output is a linear image of conv3's input, so conv3's input gradient and
# weight gradient can be written with the gradient wrt bn3's OUTPUT, conv3's input and two small matrices -- the pass "read x3,
# read dz, write dx3" over the block's largest tensors disappears and x3 is not read in backward at all.  Round 6: built, parity
# green (the folded weight gradient is 15x closer to fp64 than the pass form's), and measured SLOWER: the 12 apply passes it
# removes cost 0.74 ms per step, the launches it adds 1.59 ms -- on the compute stream the [N][N] matrix H (one or four tiles: a
# latency-bound 26-us launch) and the [P][N] x [N][N] product (31 us) cost what the passes cost, and the Gram matrix / column sums /
# fp32 W3 G / combine on the weight-gradient stream are pure additions: 23.03 vs 22.61 ms per step, serial 25.55 vs 24.80
# (profiles/r06_bn3_fold_rejected.txt).  OFF; kept with its tests as the measured answer.
FUSE_BN3_FOLD = os.environ.get("VIRTEX_AMD_BN3_FOLD", "0") != "0"
# the stem convolution's epilogue emits the BatchNorm statistics (streaming kernel, stem.hip)
STEM_STATS = os.environ.get("VIRTEX_AMD_STEM_STATS", "1") != "0"


# ----------------------------------------------------------------------------------------
# Parameter containers (torchvision's module tree; they only HOLD parameters/buffers).
# ----------------------------------------------------------------------------------------
class _Bottleneck(nn.Module):
    def __init__(self, cin, planes, stride, base_width, project):
        super().__init__()
        mid, cout = planes * base_width // 64, planes * 4
        self.conv1 = nn.Conv2d(cin, mid, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(mid)
        self.conv2 = nn.Conv2d(mid, mid, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(mid)
        self.conv3 = nn.Conv2d(mid, cout, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(cout))
        self.stride = stride


class _ResNetParams(nn.Module):
    """torchvision.models.ResNet's attribute tree (conv1 bn1 relu maxpool layer1-4 avgpool fc)."""

    def __init__(self, name: str, zero_init_residual: bool = True):
        super().__init__()
        blocks, base_width = RESNET_BLOCKS[name]
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        cin = 64
        for s, (planes, nblk) in enumerate(zip((64, 128, 256, 512), blocks)):
            layer = []
            for b in range(nblk):
                layer.append(_Bottleneck(cin, planes, 2 if (b == 0 and s > 0) else 1, base_width, b == 0))
                cin = planes * 4
            setattr(self, f"layer{s + 1}", nn.Sequential(*layer))
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(cin, 1000)
        self.out_channels = cin
        for m in self.modules():  # torchvision's init
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, _Bottleneck):
                    nn.init.zeros_(m.bn3.weight)

    def forward(self, x):
        raise RuntimeError("`.cnn` only holds parameters; call the VisualBackbone (HIP path) instead")


# ----------------------------------------------------------------------------------------
# The hand-scheduled forward / backward.
# ----------------------------------------------------------------------------------------
class _Unit:
    """One conv + BatchNorm (+ReLU) pair: static description and its parameter slots."""

    def __init__(self, conv: nn.Conv2d, bn: nn.BatchNorm2d, relu: bool, cin_pad: int = 0):
        self.conv, self.bn, self.relu = conv, bn, relu
        self.k, self.stride, self.pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        self.cin, self.cout = conv.in_channels, conv.out_channels
        self.cin_pad = cin_pad or self.cin
        self.is_gemm = self.k == 1 and self.stride == 1


class VisualBackbone(nn.Module):
    """Base class (reference: virtex/modules/visual_backbones.py:8-17)."""

    def __init__(self, visual_feature_size: int):
        super().__init__()
        self.visual_feature_size = visual_feature_size


class TorchvisionVisualBackbone(VisualBackbone):
    def __init__(self, name: str = "resnet50", visual_feature_size: int = 2048,
                 pretrained: bool = False, frozen: bool = False,
                 compute_dtype: torch.dtype = torch.bfloat16):
        super().__init__(visual_feature_size)
        if pretrained:
            raise RuntimeError("ImageNet-pretrained weights need network access; load a state dict instead")
        if name not in RESNET_BLOCKS:
            raise KeyError(f"{name} is not a supported torchvision backbone ({sorted(RESNET_BLOCKS)})")
        self.cnn = _ResNetParams(name, zero_init_residual=True)
        self.cnn.fc = nn.Identity()
        # master conv weights live physically as (KO,R,S,C) -- the kernels' layout -- while keeping
        # torchvision's logical (KO,C,R,S) shape for state-dict compatibility
        for m in self.cnn.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
        self.compute_dtype = compute_dtype
        self.frozen = frozen
        self._stats_epoch = 0          # training-mode forwards so far (they rewrite the BatchNorm running statistics)
        if frozen:
            for p in self.cnn.parameters():
                p.requires_grad = False
            self.cnn.eval()

    def weight_plan(self):
        """(parameter, channel padding, transposed copy wanted) for every convolution: what ops.prep_many refreshes."""
        if not self.cnn.training:
            return []
        stem, blocks = self._units()
        units = [stem] + [u for blk in blocks for u in blk if u is not None]
        need_grad = any(u.conv.weight.requires_grad or u.bn.weight.requires_grad or u.bn.bias.requires_grad for u in units)
        return [(u.conv.weight, u.cin_pad, need_grad) for u in units if u is not stem]   # the stem has its own cache

    # -- export ---------------------------------------------------------------------------
    _D2_STAGE = {"layer1": "res2", "layer2": "res3", "layer3": "res4", "layer4": "res5"}

    def detectron2_backbone_state_dict(self):
        """`.cnn` state dict under Detectron2's ResNet naming (reference: visual_backbones.py:76-120):
        layerN -> res(N+1), bnK -> convK.norm, downsample.0/.1 -> shortcut / shortcut.norm, and the stem's
        conv1/bn1 (plus any other top-level entry) prefixed with ``stem.``.  Tensors are the live ones."""
        out = {}
        for key, tensor in self.cnn.state_dict().items():
            parts = key.split(".")
            if parts[0] in self._D2_STAGE:
                parts[0] = self._D2_STAGE[parts[0]]
            renamed = []
            i = 0
            while i < len(parts):
                tok = parts[i]
                if tok == "downsample" and i + 1 < len(parts) and parts[i + 1] in ("0", "1"):
                    renamed.append("shortcut" if parts[i + 1] == "0" else "shortcut.norm")
                    i += 2
                    continue
                if tok in ("bn1", "bn2", "bn3"):
                    tok = f"conv{tok[2]}.norm"
                renamed.append(tok)
                i += 1
            name = ".".join(renamed)
            if not name.startswith("res"):
                name = "stem." + name
            out[name] = tensor
        return {"model": out, "__author__": "Karan Desai", "matching_heuristics": True}

    # -- eval mode: running-statistics BatchNorm folded into the convolutions ----
    def _forward_eval(self, image, stem, blocks):
        dt = self.compute_dtype

        def run(u, a, relu, residual=None, packed=False):
            w, bias = _folded(u, dt, self._stats_epoch, packed)
            return ops.conv2d_infer(a, w, bias, u.stride, 0 if packed else u.pad, relu=relu, residual=residual)

        with torch.no_grad():
            a0, packed = _stem_input(image, dt)
            cur, _ = ops.maxpool_fwd(run(stem, a0, True, packed=packed))
            for (u1, u2, u3, ud) in blocks:
                t = run(u2, run(u1, cur, True), True)
                skip = run(ud, cur, False) if ud is not None else cur
                cur = run(u3, t, True, residual=skip)
        return cur.permute(0, 3, 1, 2)  # logical NCHW, physical NHWC

    # -- schedule ---------------------------------------------------------------------
    def _units(self):
        c = self.cnn
        stem = _Unit(c.conv1, c.bn1, True, cin_pad=STEM_CPAD)
        blocks = []
        for s in range(1, 5):
            for blk in getattr(c, f"layer{s}"):
                u1 = _Unit(blk.conv1, blk.bn1, True)
                u2 = _Unit(blk.conv2, blk.bn2, True)
                u3 = _Unit(blk.conv3, blk.bn3, True)  # ReLU after the residual add
                ud = _Unit(blk.downsample[0], blk.downsample[1], False) if blk.downsample is not None else None
                blocks.append((u1, u2, u3, ud))
        return stem, blocks

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        stem, blocks = self._units()
        units = [stem] + [u for blk in blocks for u in blk if u is not None]
        # BatchNorm mode follows the BN modules' own flag, exactly like the reference's torchvision tree
        # (`frozen=True` calls cnn.eval() once in the constructor, visual_backbones.py:49-53; a later
        # model.train() flips it back -- that quirk is the reference's and is kept).
        if not self.cnn.training:
            if torch.is_grad_enabled() and any(p.requires_grad for u in units for p in (u.conv.weight, u.bn.weight, u.bn.bias)):
                raise RuntimeError("eval-mode backbone with trainable parameters under autograd is not supported: "
                                   "wrap the call in torch.no_grad() or build the backbone with frozen=True")
            return self._forward_eval(image, stem, blocks)
        params = []
        for u in units:
            params += [u.conv.weight, u.bn.weight, u.bn.bias]
        return _ResNetFn.apply(image, self, *params)

    def forward_blocks(self, x: torch.Tensor, stage: int, first: int = 0, count: int = None) -> torch.Tensor:
        """Bottlenecks [first, first + count) of `cnn.layer{stage}` on a (B,C,h,w) activation (training mode): what
        `cnn.layer{stage}[first:first+count](x)` is on the reference's torchvision tree.  Differentiable wrt `x` and the
        blocks' parameters; logical NCHW in and out, NHWC in memory."""
        if not self.cnn.training:
            raise RuntimeError("forward_blocks runs the training-mode schedule (batch statistics)")
        depth = [len(getattr(self.cnn, f"layer{s}")) for s in range(1, 5)]
        if not 1 <= stage <= 4:
            raise ValueError(f"stage must be 1..4, got {stage}")
        count = depth[stage - 1] - first if count is None else count
        if first < 0 or count < 1 or first + count > depth[stage - 1]:
            raise ValueError(f"layer{stage} has {depth[stage - 1]} Bottlenecks; asked for [{first}, {first + count})")
        lo = sum(depth[:stage - 1]) + first
        blocks = self._units()[1][lo:lo + count]
        params = []
        for u in (u for blk in blocks for u in blk if u is not None):
            params += [u.conv.weight, u.bn.weight, u.bn.bias]
        return _BlocksFn.apply(x, self, lo, lo + count, *params)


class _BlocksFn(torch.autograd.Function):
    """Bottlenecks [lo, hi) of the backbone on a (B,C,h,w) activation: the same _forward_blocks / _backward_blocks the whole
    ResNet runs, with an autograd edge on the INPUT as well (the image has none), so that a block's forward and backward can
    be held against the oracle in isolation (tests/test_model_parity.py::test_single_bottleneck_*)."""

    @staticmethod
    def forward(ctx, x, module, lo, hi, *params):
        dt = module.compute_dtype
        blocks = module._units()[1][lo:hi]
        module._stats_epoch += 1
        need_grad = x.requires_grad or any(p.requires_grad for p in params)
        cur = x.permute(0, 2, 3, 1)
        if cur.dtype != dt or not cur.is_contiguous():
            cur = cur.to(dt).contiguous()
        rec = {}
        out = _forward_blocks(rec, blocks, cur, dt, need_grad, x.device)
        ctx.module, ctx.rec, ctx.blocks, ctx.in_dtype = module, rec, blocks, x.dtype
        return out.permute(0, 3, 1, 2)

    @staticmethod
    @traced_backward
    def backward(ctx, dout):
        dt = ctx.module.compute_dtype
        dcur = dout.permute(0, 2, 3, 1)
        if dcur.dtype != dt or not dcur.is_contiguous():
            dcur = dcur.to(dt).contiguous()
        dev = dcur.device
        grads = {}
        dx = _backward_blocks(ctx.rec, ctx.blocks, dcur, dt, dev, grads)
        branch_stream.join(dev)
        wgrad_stream.join(dev)
        out = [dx.permute(0, 3, 1, 2).to(ctx.in_dtype), None, None, None]
        for u in (u for blk in ctx.blocks for u in blk if u is not None):
            out += grads[u]
        return tuple(out)


def _stem_packed(image) -> bool:
    w = image.shape[2] if image.dtype == torch.uint8 else image.shape[-1]
    return w % 2 == 0


def _stem_input(image, dt):
    """(B,3,H,W) float batches are the reference's wire format (already normalised on the CPU).  uint8 (B,H,W,3)
    batches -- decoder output -- are normalised here, on the device, in the same kernel that changes the layout
    (ImageNet mean/std as in virtex/data/transforms.py:85-97): a quarter of the PCIe bytes, no CPU float pass."""
    packed = _stem_packed(image)
    cpad, halo = (STEM_PACK_C, STEM_HALO) if packed else (STEM_CPAD, 0)
    if image.dtype == torch.uint8:
        if image.dim() != 4 or image.shape[-1] != 3:
            raise ValueError("uint8 image batches must be (B, H, W, 3)")
        return ops.image_u8_to_nhwc(image.contiguous(), dt, cpad, halo=halo), packed
    return ops.image_to_nhwc(image.float().contiguous(), dt, cpad, halo=halo), packed


def _stem_weight32(u: "_Unit"):
    """fp32 (KO, 7*8, 4) zero-padded copy of the stem filter for the packed layout."""
    w32 = u.conv.weight.detach().permute(0, 2, 3, 1)                       # (KO, 7, 7, 3)
    wp = torch.zeros(u.cout, u.k, STEM_PACK_S, STEM_PACK_C, dtype=torch.float32, device=w32.device)
    wp[:, :, : u.k, : u.cin] = w32
    return wp.view(u.cout, u.k * STEM_PACK_S, STEM_PACK_C)


def _stem_weight(u: "_Unit", dtype):
    """Compute copy (KO, 7, 8, 4) of the stem filter, cached on the parameter until it changes."""
    p = u.conv.weight
    stamp = (p._version, p.data_ptr(), dtype)
    e = p.__dict__.get("_vtx_stem")
    if e is not None and e[0] == stamp:
        return e[1]
    wp = _stem_weight32(u)
    w = wp if dtype == torch.float32 else ops.weight_prep(wp, dtype, want_wt=False)[0]
    w = w.view(u.cout, u.k, STEM_PACK_S, STEM_PACK_C)
    p.__dict__["_vtx_stem"] = (stamp, w)
    return w


def _folded(u: _Unit, dtype, epoch: int = 0, packed: bool = False):
    """Eval mode: (w (KO,R,S,Cp), bias (KO,)) with the running-statistics BatchNorm folded into the
    convolution; cached on the unit's tensors until any of them is modified in place."""
    bn = u.bn
    src = (u.conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)
    # running statistics are updated by the BatchNorm kernel itself (no torch version bump): the module counts its
    # training-mode forwards and the key carries that count
    key = (dtype, epoch, packed) + tuple((t.data_ptr(), t._version) for t in src)
    cache = getattr(u.conv, "_vtx_folded", None)
    if cache is not None and cache[0] == key:
        return cache[1], cache[2]
    if packed:
        w, bias = ops.bn_fold(_stem_weight32(u), bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                              bn.eps, dtype)
        w = w.view(u.cout, u.k, STEM_PACK_S, STEM_PACK_C)
    else:
        w32 = u.conv.weight.detach().permute(0, 2, 3, 1).contiguous().view(u.cout, u.k * u.k, u.cin)
        w, bias = ops.bn_fold(w32, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps, dtype,
                              cpad=u.cin_pad)
        w = w.view(u.cout, u.k, u.k, u.cin_pad)
    u.conv._vtx_folded = (key, w, bias)
    return w, bias


def _prep_weight(u: _Unit, dtype, need_wt: bool):
    """fp32 master (KO,C,R,S logical) -> compute copies w (KO,R,S,Cp) and wt (Cp,R,S,KO); cached on the
    parameter until it changes (ops.prepped), normally already refreshed for the whole model by ops.prep_many."""
    w, wt = ops.prepped(u.conv.weight, dtype, cpad=u.cin_pad, want_w=True, want_wt=need_wt)
    w = w.view(u.cout, u.k, u.k, u.cin_pad)
    if wt is not None:
        wt = wt.view(u.cin_pad, u.k, u.k, u.cout)
    return w, wt


def _stem_conv(u: _Unit, a, w, dt):
    """The packed stem convolution; bf16: the streaming kernel also emits the BatchNorm statistics of its output
    (stem.hip) -- `stats` is None when the tiled kernel ran instead (fp32, VIRTEX_AMD_STEM_STREAM=0)."""
    if STEM_STATS and dt == torch.bfloat16:
        x, stats = ops.conv2d_fwd(a, w, u.stride, 0, bn_shift=u.bn.running_mean)
        return x, (stats if (stats is not None and stats.strips <= 512) else None)
    return ops.conv2d_fwd(a, w, u.stride, 0), None


def _conv_fwd(u: _Unit, x, w, bn_shift=None):
    """Returns (y, stats): `stats` are the BatchNorm statistics of y emitted by the convolution's own
    epilogue (None if not requested or the kernel that ran does not produce them -> stand-alone reduction)."""
    if u.is_gemm:
        N, H, W, C = x.shape
        if bn_shift is None:
            return ops.gemm_nt(x.view(-1, C), w.view(u.cout, C)).view(N, H, W, u.cout), None
        y, st = ops.gemm_nt(x.view(-1, C), w.view(u.cout, C), bn_shift=bn_shift)
        return y.view(N, H, W, u.cout), st
    if bn_shift is None:
        return ops.conv2d_fwd(x, w, u.stride, u.pad), None
    return ops.conv2d_fwd(x, w, u.stride, u.pad, bn_shift=bn_shift)


def _conv_dgrad(u: _Unit, dy, wt, x_shape, residual=None, bn=None):
    """bn (ops.BnBwd) given: returns (gradient, stats); stats is None when the kernel did not fuse (fp32 mode) and
    the gradient is then the plain one."""
    if u.is_gemm:
        N, H, W, C = x_shape
        r = residual.view(-1, C) if residual is not None else None
        if bn is not None:
            out, st = ops.gemm_nt_bnbwd(dy.view(-1, u.cout), wt.view(C, u.cout), bn, residual=r)
            return out.view(N, H, W, C), st
        return ops.gemm_nt(dy.view(-1, u.cout), wt.view(C, u.cout), residual=r).view(N, H, W, C)
    return ops.conv2d_dgrad(dy, wt, x_shape, u.stride, u.pad, residual=residual, bn=bn)


def _conv_wgrad(u: _Unit, x, dy):
    """Accumulates into the parameter's own gradient buffer when it has a suitable one (returns None),
    otherwise returns the gradient in the master weight's logical (KO,C,R,S) shape."""
    sink = gradsink.target(u.conv.weight, (u.cout, u.k, u.k, u.cin_pad)) if u.cin_pad == u.cin else None
    dw = sink if sink is not None else torch.zeros(u.cout, u.k, u.k, u.cin_pad, dtype=torch.float32, device=x.device)
    if u.is_gemm:
        ops.gemm_tn_acc(dy.view(-1, u.cout), x.view(-1, u.cin_pad), dw.view(u.cout, u.cin_pad))
    else:
        ops.conv2d_wgrad(x, dy, dw, u.stride, u.pad)
    if sink is not None:
        return None
    if u.cin_pad != u.cin:
        dw = dw[..., :u.cin].contiguous()
    return dw.permute(0, 3, 1, 2)


from ..streams import branch_stream, wgrad_stream  # noqa: E402  (shared with the text heads and the data-parallel engine)


class _Saved:
    __slots__ = ("a", "x", "y", "mean", "rstd", "wt", "bits")


def _run_unit(rec, u: _Unit, a, relu, dt, need_grad, residual=None):
    """conv -> train-mode BatchNorm (-> ReLU / residual join) of one unit; what backward needs goes into rec[u]."""
    bn = u.bn
    w, wt = _prep_weight(u, dt, need_wt=need_grad)
    # the conv epilogue also produces the batch statistics (taken against the running mean)
    x, stats = _conv_fwd(u, a, w, bn_shift=bn.running_mean if FUSE_BN_STATS else None)
    bits = None
    if residual is not None and relu and need_grad and RELU_BITS and FUSE_BN_BWD and dt == torch.bfloat16:
        y, mean, rstd, bits = ops.bn_fwd(x, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                                         bn.num_batches_tracked, eps=bn.eps,
                                         momentum=bn.momentum if bn.momentum is not None else 0.1, relu=relu,
                                         residual=residual, stats=stats, want_bits=True)
    else:
        y, mean, rstd = ops.bn_fwd(x, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                                   bn.num_batches_tracked, eps=bn.eps,
                                   momentum=bn.momentum if bn.momentum is not None else 0.1, relu=relu,
                                   residual=residual, stats=stats)
    s = _Saved()
    s.a, s.x, s.y, s.mean, s.rstd, s.wt, s.bits = a, x, y, mean, rstd, wt, bits
    rec[u] = s
    return y


def _forward_blocks(rec, blocks, cur, dt, need_grad, dev):
    """The Bottlenecks `blocks` (torchvision Bottleneck v1.5, reached from the reference's forward walk
    virtex/modules/visual_backbones.py:68-74) on the NHWC activation `cur`."""
    for bi, (u1, u2, u3, ud) in enumerate(blocks):
        inp = cur
        if ud is not None:
            # projection shortcut (1x1 conv + BN) on the branch stream, under the main branch's convolutions
            br = branch_stream(dev, inp)
            with br:
                skip = _run_unit(rec, ud, inp, False, dt, need_grad)
            t = _run_unit(rec, u1, inp, True, dt, need_grad)
            t = _run_unit(rec, u2, t, True, dt, need_grad)
            br.wait(skip)
        else:
            t = _run_unit(rec, u1, inp, True, dt, need_grad)
            t = _run_unit(rec, u2, t, True, dt, need_grad)
            skip = inp
        cur = _run_unit(rec, u3, t, True, dt, need_grad, residual=skip)
        # A block whose conv3 backward will run as the fused streaming kernel (conv3_bwd.hip) recomputes conv3's input
        # a3 = relu(bn2(x2)) from x2 inside that kernel: a3 is not kept for backward (103 MB per stage-1 block at bs 256).
        # The decision is the one _backward_blocks takes (same flags, same shapes); it then finds rec[u3].a is None.
        s3 = rec[u3]
        if (need_grad and bi + 1 < len(blocks) and FUSE_BN_BWD and FUSE_CONV3_BWD and dt == torch.bfloat16 and u3.is_gemm
                and u3.cin_pad == u3.cin and s3.wt is not None and ops.conv3_bwd_fused_supported(cur, s3.wt.view(u3.cin, u3.cout))):
            s3.a = None
            rec[u2].y = None
    return cur


def _backward_blocks(rec, blocks, dcur, dt, dev, grads):
    """Backward of the Bottlenecks `blocks` (last to first) from the gradient `dcur` (NHWC, compute dtype) wrt the last block's
    output; fills grads[unit] = [dW, dgamma, dbeta] (None where the kernel accumulated into the parameter's own gradient
    buffer) and returns the gradient wrt the first block's input."""
    def bn_back(u: _Unit, s: _Saved, dy, masked, want_dz=False, residual=False):
        sg, sb = gradsink.target(u.bn.weight), gradsink.target(u.bn.bias)
        dg = sg if sg is not None else torch.zeros(u.cout, dtype=torch.float32, device=dev)
        db = sb if sb is not None else torch.zeros(u.cout, dtype=torch.float32, device=dev)
        # BN directly followed by ReLU: the mask is recomputed from x (one tensor less to read);
        # BN + residual + ReLU: the mask is the saved block output
        out = ops.bn_bwd(s.x, dy, s.y if (masked and residual) else None, u.bn.weight.detach(), s.mean, s.rstd,
                         dg, db, want_dz=want_dz,
                         relu_beta=u.bn.bias.detach() if (masked and not residual) else None)
        grads[u] = [None, None if sg is not None else dg, None if sb is not None else db]
        return out

    def bn_back_fused(u: _Unit, s: _Saved, dz, st):
        """dz is already masked and its sums are in `st` (emitted by the kernel that produced it)."""
        sg, sb = gradsink.target(u.bn.weight), gradsink.target(u.bn.bias)
        dg = sg if sg is not None else torch.zeros(u.cout, dtype=torch.float32, device=dev)
        db = sb if sb is not None else torch.zeros(u.cout, dtype=torch.float32, device=dev)
        out = ops.bn_bwd_fused(s.x, dz, u.bn.weight.detach(), s.mean, s.rstd, dg, db, st)
        grads[u] = [None, None if sg is not None else dg, None if sb is not None else db]
        return out

    def relu_bn(u: _Unit, s: _Saved):
        """The fusion descriptor of an interior BatchNorm+ReLU: mask recomputed from its input."""
        return ops.BnBwd(s.x, s.mean, s.rstd, gamma=u.bn.weight.detach(), beta=u.bn.bias.detach()) if fuse else None

    def conv3_back_fused(u3: _Unit, u2: _Unit, s3: _Saved, s2: _Saved, dz, st):
        """bn3 backward + conv3 input gradient (+ bn2 mask / sums) + conv3 weight gradient in ONE launch; the partial
        weight gradients are folded into the parameter's gradient on the weight-gradient stream."""
        sg, sb = gradsink.target(u3.bn.weight), gradsink.target(u3.bn.bias)
        dg = sg if sg is not None else torch.zeros(u3.cout, dtype=torch.float32, device=dev)
        db = sb if sb is not None else torch.zeros(u3.cout, dtype=torch.float32, device=dev)
        dy2, st2, parts, nparts = ops.conv3_bwd_fused(dz, s3.x, u3.bn.weight.detach(), s3.mean, s3.rstd, dg, db, st,
                                                      s3.wt.view(u3.cin, u3.cout), relu_bn(u2, s2))
        sink = gradsink.target(u3.conv.weight, (u3.cout, 1, 1, u3.cin))
        dw = sink if sink is not None else torch.zeros(u3.cout, 1, 1, u3.cin, dtype=torch.float32, device=dev)
        with wgrad_stream(dev, parts):
            ops.partials_reduce_acc(parts, nparts, dw.view(u3.cout, u3.cin))
        grads[u3] = [None if sink is not None else dw.permute(0, 3, 1, 2), None if sg is not None else dg,
                     None if sb is not None else db]
        return dy2, st2

    def conv3_back_folded(u3: _Unit, u2: _Unit, s3: _Saved, s2: _Saved, dz, st):
        """bn3's backward folded into conv3's weights (csrc/bn_fold.hip): with dx3 = a0 dz + b1 x3 + c per channel and
        x3 = a3 . W3^T,   dy2 = dz . (a0 o W3) + a3 . H + bias  (H = W3^T diag(b1) W3, bias = W3^T c)   on the compute stream and
        dW3 = diag(a0) dz^T a3 + diag(b1) W3 (a3^T a3) + c colsum(a3)^T   on the weight-gradient stream.  No dx3, no read of x3."""
        sg, sb = gradsink.target(u3.bn.weight), gradsink.target(u3.bn.bias)
        dg = sg if sg is not None else torch.zeros(u3.cout, dtype=torch.float32, device=dev)
        db = sb if sb is not None else torch.zeros(u3.cout, dtype=torch.float32, device=dev)
        K, N = u3.cout, u3.cin
        wt = s3.wt.view(N, K)
        dz2, a3 = dz.view(-1, K), s3.a.view(-1, N)
        P = dz2.shape[0]
        wa, wb, bias, abc = ops.bn_bwd_fold(wt, u3.bn.weight.detach(), s3.mean, s3.rstd, dg, db, st, P)
        h = ops.gemm_nt(wb, wt)                                        # [N][N]
        tmp = ops.gemm_nt(a3, h, bias=bias)                            # [P][N]: the b1 x3 + c part of dx3, through W3
        dy2, st2 = ops.gemm_nt_bnbwd(dz2, wa, relu_bn(u2, s2), residual=tmp)
        sink = gradsink.target(u3.conv.weight, (K, 1, 1, N))
        dw = sink if sink is not None else torch.zeros(K, 1, 1, N, dtype=torch.float32, device=dev)
        with wgrad_stream(dev, dz2, a3, abc):
            scratch = torch.zeros(K * N + N * N + N, dtype=torch.float32, device=dev)
            t, gram, csum = scratch[:K * N].view(K, N), scratch[K * N:K * N + N * N].view(N, N), scratch[K * N + N * N:]
            ops.gemm_tn_acc(dz2, a3, t)                                # dz^T a3
            ops.gemm_tn_acc(a3, a3, gram)                              # the Gram matrix of conv3's input
            ops.colsum_acc(a3, csum)
            ops.splitk_flush(dev)                                      # t and gram are READ below: inside a reduction batch their folds run now
            w32 = u3.conv.weight.detach().permute(0, 2, 3, 1).reshape(K, N)          # the fp32 master (stored (KO,1,1,C): a view)
            wg = ops.gemm_nt(w32, gram)                                # W3 (a3^T a3), fp32 (gram is symmetric)
            ops.wgrad_fold_combine(dw.view(K, N), t, wg, csum, abc)
        grads[u3] = [None if sink is not None else dw.permute(0, 3, 1, 2), None if sg is not None else dg,
                     None if sb is not None else db]
        return dy2.view(*dz.shape[:-1], N), st2

    fuse = FUSE_BN_BWD and dt == torch.bfloat16
    st3 = None                  # sums for this block's bn3, when the next block's conv1 input gradient emitted them
    for bi in reversed(range(len(blocks))):
        (u1, u2, u3, ud) = blocks[bi]
        s1, s2, s3 = rec[u1], rec[u2], rec[u3]
        # the split-K reductions of this block's three or four weight gradients in ONE launch, issued below on the weight-gradient
        # stream right before the block's gradients are announced (ops.splitk_batch: measured slower, off by default)
        sk = ops.splitk_batch(dev).begin()
        fused3 = (fuse and FUSE_CONV3_BWD and st3 is not None and u3.is_gemm and u3.cin_pad == u3.cin
                  and s3.wt is not None and ops.conv3_bwd_fused_supported(dcur, s3.wt.view(u3.cin, u3.cout)))
        if s3.a is None and not fused3:
            raise RuntimeError("the forward pass dropped conv3's input for the fused conv3 backward, which this backward pass "
                               "does not take: FUSE_BN_BWD / FUSE_CONV3_BWD / vtx_set_switch('conv3_bwd') were changed between "
                               "the forward and the backward of one step")
        folded3 = (fuse and FUSE_BN3_FOLD and not fused3 and st3 is not None and u3.is_gemm and u3.cin_pad == u3.cin
                   and s3.wt is not None and u3.cout % 8 == 0 and u3.cin % 8 == 0)
        if fused3 or folded3:   # bn3's backward happens inside conv3's backward: no dx3
            dz, dx3 = dcur, None
        elif st3 is not None:   # dcur IS dz: masked by (block output > 0) in the producing epilogue
            dz = dcur
            dx3 = bn_back_fused(u3, s3, dz, st3)
        else:
            dx3, dz = bn_back(u3, s3, dcur, True, want_dz=True, residual=True)      # dz: gradient of the identity path
        br = None
        if ud is not None:
            # the shortcut's backward (BN backward, weight gradient, input gradient) on the branch stream,
            # under the main branch's three convolutions
            sd = rec[ud]
            br = branch_stream(dev, dz, sd.x, sd.a, sd.mean, sd.rstd)
            with br:
                dxd = bn_back(ud, sd, dz, False)
                with wgrad_stream(dev, sd.a, dxd):
                    grads[ud][0] = _conv_wgrad(ud, sd.a, dxd)
                dskip = _conv_dgrad(ud, dxd, sd.wt, sd.a.shape)
        if fused3:
            dy2, st2 = conv3_back_fused(u3, u2, s3, s2, dz, st3)
        elif folded3:
            dy2, st2 = conv3_back_folded(u3, u2, s3, s2, dz, st3)
        else:
            with wgrad_stream(dev, s3.a, dx3):
                grads[u3][0] = _conv_wgrad(u3, s3.a, dx3)
            if fuse:
                dy2, st2 = _conv_dgrad(u3, dx3, s3.wt, s3.a.shape, bn=relu_bn(u2, s2))
            else:
                dy2, st2 = _conv_dgrad(u3, dx3, s3.wt, s3.a.shape), None
        dx2 = bn_back_fused(u2, s2, dy2, st2) if st2 is not None else bn_back(u2, s2, dy2, True)
        with wgrad_stream(dev, s2.a, dx2):
            grads[u2][0] = _conv_wgrad(u2, s2.a, dx2)
        if fuse:
            dy1, st1 = _conv_dgrad(u2, dx2, s2.wt, s2.a.shape, bn=relu_bn(u1, s1))
        else:
            dy1, st1 = _conv_dgrad(u2, dx2, s2.wt, s2.a.shape), None
        dx1 = bn_back_fused(u1, s1, dy1, st1) if st1 is not None else bn_back(u1, s1, dy1, True)
        with wgrad_stream(dev, s1.a, dx1):
            grads[u1][0] = _conv_wgrad(u1, s1.a, dx1)
        if br is not None:
            br.wait(dskip)
            join = dskip
        else:
            join = dz
        # The block's input gradient (both branches joined in the epilogue).  The block input is the previous
        # block's output relu(bn3(x3) + skip): fuse THAT bn3's backward (mask = the saved output) into this kernel.
        if fuse and bi > 0:
            p3 = rec[blocks[bi - 1][2]]
            dcur, st3 = _conv_dgrad(u1, dx1, s1.wt, s1.a.shape, residual=join,
                                    bn=ops.BnBwd(p3.x, p3.mean, p3.rstd, ymask=p3.y, ybits=p3.bits))
        else:
            dcur, st3 = _conv_dgrad(u1, dx1, s1.wt, s1.a.shape, residual=join), None
        with wgrad_stream(dev):
            sk.end()
        # this block's gradient kernels are all enqueued: let the data-parallel engine start exchanging the
        # buckets they complete while the rest of the backbone's backward runs
        for u in (u3, u2, u1, ud):
            if u is not None:
                gradsink.mark_ready([p for p, g in zip((u.conv.weight, u.bn.weight, u.bn.bias), grads[u]) if g is None])
    return dcur


class _ResNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, module, *params):
        dt = module.compute_dtype
        dev = image.device
        stem, blocks = module._units()
        module._stats_epoch += 1
        need_grad = any(p.requires_grad for p in params)
        rec = {}

        def run_stem(u: _Unit, a):
            """The stem on the generic 8-channel layout (odd image widths) or with the stem tail unfused."""
            bn = u.bn
            if packed:
                w, wt = _stem_weight(u, dt), None
                x, stats = _stem_conv(u, a, w, dt)
            else:
                w, wt = _prep_weight(u, dt, need_wt=False)
                x, stats = _conv_fwd(u, a, w, bn_shift=bn.running_mean if FUSE_BN_STATS else None)
            y, mean, rstd = ops.bn_fwd(x, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                                       bn.num_batches_tracked, eps=bn.eps,
                                       momentum=bn.momentum if bn.momentum is not None else 0.1, relu=True,
                                       residual=None, stats=stats)
            s = _Saved()
            s.a, s.x, s.y, s.mean, s.rstd, s.wt, s.bits = a, x, y, mean, rstd, wt, None
            rec[u] = s
            return y

        a0, packed = _stem_input(image, dt)
        if FUSE_STEM_FWD and packed:
            bn = stem.bn
            x0, stats0 = _stem_conv(stem, a0, _stem_weight(stem, dt), dt)
            pooled, argmax, mean0, rstd0 = ops.bn_fwd_maxpool(x0, bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                                                             bn.running_var, bn.num_batches_tracked, eps=bn.eps,
                                                             momentum=bn.momentum if bn.momentum is not None else 0.1,
                                                             stats=stats0)
            s0 = _Saved()
            s0.a, s0.x, s0.y, s0.mean, s0.rstd, s0.wt, s0.bits = a0, x0, None, mean0, rstd0, None, None     # y: never materialised
            rec[stem] = s0
            y = x0                                                          # only its shape is used below
        else:
            y = run_stem(stem, a0)
            pooled, argmax = ops.maxpool_fwd(y)
        cur = _forward_blocks(rec, blocks, pooled, dt, need_grad, dev)
        ctx.module, ctx.rec, ctx.argmax, ctx.stem_out_shape = module, rec, argmax, y.shape
        ctx.units = (stem, blocks)
        ctx.stem_packed = packed
        ctx.nparams = len(params)
        return cur.permute(0, 3, 1, 2)  # logical NCHW, physical NHWC

    @staticmethod
    @traced_backward
    def backward(ctx, dfeat):
        module, rec = ctx.module, ctx.rec
        dt = module.compute_dtype
        stem, blocks = ctx.units
        dcur = dfeat.permute(0, 2, 3, 1)
        if dcur.dtype != dt or not dcur.is_contiguous():
            dcur = dcur.to(dt).contiguous()
        dev = dcur.device
        units = [stem] + [u for blk in blocks for u in blk if u is not None]
        grads = {}

        dcur = _backward_blocks(rec, blocks, dcur, dt, dev, grads)

        def bn_back(u: _Unit, s: _Saved, dy, masked):
            sg, sb = gradsink.target(u.bn.weight), gradsink.target(u.bn.bias)
            dg = sg if sg is not None else torch.zeros(u.cout, dtype=torch.float32, device=dev)
            db = sb if sb is not None else torch.zeros(u.cout, dtype=torch.float32, device=dev)
            out = ops.bn_bwd(s.x, dy, None, u.bn.weight.detach(), s.mean, s.rstd, dg, db,
                             relu_beta=u.bn.bias.detach() if masked else None)
            grads[u] = [None, None if sg is not None else dg, None if sb is not None else db]
            return out

        s0 = rec[stem]
        if FUSE_STEM_TAIL and dcur.is_contiguous():
            # max-pool backward gathered inside the stem's BatchNorm backward: this chain is the exposed end of the step
            sg, sb = gradsink.target(stem.bn.weight), gradsink.target(stem.bn.bias)
            dg = sg if sg is not None else torch.zeros(stem.cout, dtype=torch.float32, device=dev)
            db = sb if sb is not None else torch.zeros(stem.cout, dtype=torch.float32, device=dev)
            dx0 = ops.bn_bwd_maxpool(s0.x, dcur, ctx.argmax, stem.bn.weight.detach(), stem.bn.bias.detach(), s0.mean, s0.rstd, dg, db)
            grads[stem] = [None, None if sg is not None else dg, None if sb is not None else db]
        else:
            dstem = ops.maxpool_bwd(dcur, ctx.argmax, ctx.stem_out_shape)
            dx0 = bn_back(stem, s0, dstem, True)
        with wgrad_stream(dev, s0.a, dx0):               # no input gradient for the image
            if ctx.stem_packed:
                dwp = torch.zeros(stem.cout, stem.k, STEM_PACK_S, STEM_PACK_C, dtype=torch.float32, device=dev)
                ops.conv2d_wgrad(s0.a, dx0, dwp, stem.stride, 0)
                grads[stem][0] = dwp[:, :, : stem.k, : stem.cin].permute(0, 3, 1, 2)     # logical (KO,C,R,S)
            else:
                grads[stem][0] = _conv_wgrad(stem, s0.a, dx0)
        branch_stream.join(dev)           # fallback BN gradients of the shortcuts may have been allocated there
        wgrad_stream.join(dev)

        out = [None, None]
        for u in units:
            out += grads[u]
        return tuple(out)
