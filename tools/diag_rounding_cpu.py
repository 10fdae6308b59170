"""Where does the bf16 step's backbone-gradient deviation come from?  CPU experiment on the ORACLE (no HIP):

  A  the oracle under torch.autocast("cpu", bfloat16)  -- what stock PyTorch AMP does with the reference model
  B  fp32 oracle, every backbone conv / BatchNorm(+ReLU) OUTPUT rounded to bf16 in the forward only
     (straight-through: the backward is exact fp32 arithmetic on the perturbed activations)
  C  fp32 oracle, forward exact, every activation GRADIENT leaving a backbone conv / BN rounded to bf16
  D  like B but rounded to fp16 (11-bit significand: the reference's own AMP dtype)
  E  like B, but every ReLU takes its MASK from the exact fp32 forward of the same batch (round 4: what "fp32-exact ReLU
     masks" could buy at best -- an upper bound: in training mode the mask depends on the batch statistics, which no
     convolution epilogue knows, so a kernel could only approximate it)

each against the plain fp32 oracle on the same batch: per-tensor relative L2 and cosine of the 159 backbone
gradients.  B ~ A and C << B  =>  the deviation is the forward's 16-bit activations flipping ReLU masks
(a discontinuity of the gradient), not the precision of the backward arithmetic.

    python tools/diag_rounding_cpu.py [B] [image_size] [state]
"""
import copy
import sys

import torch
from torch import nn

sys.path.insert(0, ".")
from oracle import bicaptioning as port, synth  # noqa: E402


class _RoundST(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        return x.to(dtype).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g, None


class _RoundGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.dtype = dtype
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).to(g.dtype), None


def hook_backbone(model, fwd_dtype=None, bwd_dtype=None):
    hs = []
    for m in model.visual.cnn.modules():
        if isinstance(m, (nn.Conv2d, nn.BatchNorm2d, nn.ReLU)):
            def h(mod, inp, out, fd=fwd_dtype, bd=bwd_dtype):
                if fd is not None:
                    out = _RoundST.apply(out, fd)
                if bd is not None:
                    out = _RoundGrad.apply(out, bd)
                return out
            hs.append(m.register_forward_hook(h))
    return hs


def record_relu_masks(model, batch):
    """masks (output > 0) of every ReLU call of the backbone, in call order, from a plain forward"""
    masks, hs = [], []
    for m in model.visual.cnn.modules():
        if isinstance(m, nn.ReLU):
            hs.append(m.register_forward_hook(lambda mod, inp, out: masks.append((out > 0).clone())))
    model.train()
    with torch.no_grad():
        model(batch)
    for h in hs:
        h.remove()
    return masks


def hook_relu_masks(model, masks):
    """every ReLU call of the backbone returns input * recorded mask (exact gradient of that: the mask)"""
    state = {"k": 0}
    for m in model.visual.cnn.modules():
        if isinstance(m, nn.ReLU):
            m.inplace = False
            def h(mod, inp, out):
                k = state["k"]; state["k"] += 1
                return inp[0] * masks[k].to(inp[0].dtype)
            m.register_forward_hook(h)
    return state


def grads_of(model, batch, autocast=None):
    model.zero_grad(set_to_none=True)
    model.train()
    if autocast is not None:
        with torch.autocast("cpu", dtype=autocast):
            out = model(batch)
    else:
        out = model(batch)
    out["loss"].backward()
    return out["loss"].item(), {n: p.grad.detach().double().clone() for n, p in model.named_parameters()}


def compare(tag, ref, got, lref, lgot):
    rows = []
    for n, r in ref.items():
        if r.norm() == 0:
            continue
        g = got[n]
        rel = ((g - r).norm() / r.norm()).item()
        cos = (g.flatten() @ r.flatten() / (g.norm() * r.norm() + 1e-300)).item()
        rows.append((rel, cos, n))
    cnn = [r for r in rows if "cnn" in r[2]]
    txt = [r for r in rows if "cnn" not in r[2]]
    rel = sorted(r[0] for r in cnn)
    print(f"{tag:34s} loss {lgot:.5f} (fp32 {lref:.5f}) | text max rel {max(r[0] for r in txt):.2e} | "
          f"cnn({len(cnn)}): median rel {rel[len(rel) // 2]:.2e} p90 {rel[int(len(rel) * .9)]:.2e} max {rel[-1]:.2e} "
          f"min cos {min(r[1] for r in cnn):.4f}", flush=True)
    return rows


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 224
    state = sys.argv[3] if len(sys.argv) > 3 else "reference_init"
    om = synth.seeded_model(port.build_model, seed=0, dropout=0.0, randomize=(state != "reference_init"))
    if state == "random_bn3x0.2":
        with torch.no_grad():
            for n, p in om.named_parameters():
                if n.endswith("bn3.weight"):
                    p.mul_(0.2)
    batch = synth.synthetic_batch(B, image_size=S, seed=3, ragged=False)
    print(f"B={B} S={S} state={state}")
    lref, ref = grads_of(om, batch)
    la, ga = grads_of(copy.deepcopy(om), batch, autocast=torch.bfloat16)
    compare("A autocast(cpu, bf16)", ref, ga, lref, la)
    for tag, fd, bd in (("B fwd activations -> bf16 (ST)", torch.bfloat16, None),
                        ("C bwd activation grads -> bf16", None, torch.bfloat16),
                        ("D fwd activations -> fp16 (ST)", torch.float16, None)):
        m = copy.deepcopy(om)
        hook_backbone(m, fd, bd)
        l, g = grads_of(m, batch)
        compare(tag, ref, g, lref, l)
    # E: bf16 forward activations, ReLU masks of the exact forward.  (BatchNorm in train mode leaves the running statistics
    # of a deep copy changed by the recording pass: irrelevant for the gradients, which use batch statistics.)
    masks = record_relu_masks(copy.deepcopy(om), batch)
    m = copy.deepcopy(om)
    hook_backbone(m, torch.bfloat16, None)
    hook_relu_masks(m, masks)
    l, g = grads_of(m, batch)
    compare("E = B with fp32-exact ReLU masks", ref, g, lref, l)


if __name__ == "__main__":
    main()
