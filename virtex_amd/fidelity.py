"""How far is the bf16 step (the throughput mode) from the fp32 step (the mode pinned to the oracle)?

One forward/backward of the SAME weights on the SAME batch in both compute dtypes, dropout off (the two
attention kernels do not index their dropout masks identically), per-tensor relative L2 distance and cosine of
all 202 gradients.  Used by bench.py (`fidelity` object of the JSON line) and tests/test_fidelity.py.

What to expect, and why (profiles/r02_bf16_rounding_mechanism.txt, tools/diag_rounding_cpu.py): a 16-bit FORWARD
perturbs the ResNet's activations by ~1.5 % after 53 layers, which flips the ReLU mask of the ~1 % of elements that
sit within that distance of zero; a flipped mask element changes its activation gradient by 100 %, so backbone
gradients move by ~sqrt(flipped fraction) ~ 0.1-0.2 relative.  Stock `torch.autocast(bfloat16)` of the reference model
shows the same numbers (median 0.195 at the reference initialisation); rounding only the backward's activation
gradients to bf16 moves them by 0.4 %.  The reference's own fp16 AMP sits at 0.064 for the same reason.
"""
import contextlib
from typing import Dict

import torch


@contextlib.contextmanager
def _no_dropout(model):
    saved = []
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            saved.append((m, "p", m.p)); m.p = 0.0
        elif isinstance(getattr(m, "dropout", None), float):
            saved.append((m, "dropout", m.dropout)); m.dropout = 0.0
    try:
        yield
    finally:
        for m, k, v in saved:
            setattr(m, k, v)


def gradient_distance(grads_a: Dict[str, torch.Tensor], grads_b: Dict[str, torch.Tensor]) -> Dict[str, dict]:
    """Per tensor {rel, cos} of a against b (fp64 on the device the tensors live on); tensors whose reference
    gradient is exactly zero (e.g. everything behind a zero-initialised bn3.weight) are skipped."""
    out = {}
    for n, b in grads_b.items():
        b = b.detach().double().flatten()
        nb = b.norm()
        if nb.item() == 0.0:
            continue
        a = grads_a[n].detach().double().flatten().to(b.device)
        out[n] = {"rel": ((a - b).norm() / nb).item(), "cos": (a @ b / (a.norm() * nb + 1e-300)).item()}
    return out


def summarize(dist: Dict[str, dict]) -> dict:
    def stats(rows):
        if not rows:
            return None
        rel = sorted(r["rel"] for r in rows)
        cos = sorted(r["cos"] for r in rows)
        # min_cos is the single worst tensor: on a chaotic state (randomised BatchNorm, 100+ layers) it moves by several hundredths
        # under a rounding-level perturbation; p10_cos (the 10th percentile) is the robust low end of the same distribution
        return {"tensors": len(rows), "median_rel": round(rel[len(rel) // 2], 5), "p90_rel": round(rel[int(len(rel) * 0.9)], 5),
                "max_rel": round(rel[-1], 5), "min_cos": round(cos[0], 5), "p10_cos": round(cos[int(len(cos) * 0.1)], 5),
                "median_cos": round(cos[len(cos) // 2], 5)}
    return {"backbone": stats([v for k, v in dist.items() if "cnn" in k]),
            "text": stats([v for k, v in dist.items() if "cnn" not in k])}


def run_grads(model, batch, buckets=None):
    """(loss, {name: gradient}) of one training-mode forward/backward; with `buckets` the gradients are read from
    the data-parallel engine's flat buffer (they are views of it) after its finish()."""
    model.train()
    if buckets is not None:
        buckets.zero()
        buckets.begin()
    else:
        model.zero_grad(set_to_none=True)
    out = model(batch)
    out["loss"].backward()
    if buckets is not None:
        buckets.finish()
    return out["loss"].detach().float().item(), {n: p.grad.detach().clone() for n, p in model.named_parameters()
                                                 if p.grad is not None}


def clone_in_dtype(model, dtype):
    """A second model with the same weights / buffers computing in `dtype` (same device)."""
    from . import factories as vf
    th = model.textual
    arch = f"L{th.num_layers}_H{th.hidden_size}_A{th.attention_heads}_F{th.feedforward_size}"
    cnn = next(k for k, (blocks, width) in vf.visual_backbones.RESNET_BLOCKS.items()
               if len(model.visual.cnn.layer3) == blocks[2] and model.visual.cnn.layer1[0].conv1.out_channels == width)
    other = vf.build_bicaptioning_model(visual=f"torchvision::{cnn}", textual=f"transdec_postnorm::{arch}",
                                        vocab_size=th.vocab_size, dropout=th.dropout,
                                        max_caption_length=th.embedding.positions.num_embeddings, compute_dtype=dtype)
    other.load_state_dict(model.state_dict())
    return other.to(next(model.parameters()).device)


def bf16_vs_fp32(model, batch, buckets=None) -> dict:
    """`model` computes in bf16.  Returns the summary the bench line carries."""
    ref = clone_in_dtype(model, torch.float32)
    # BatchNorm running statistics are updated by every training forward: give both runs the same starting buffers
    # and restore the caller's afterwards
    buf = {n: b.detach().clone() for n, b in model.named_buffers()}
    with _no_dropout(model), _no_dropout(ref):
        l16, g16 = run_grads(model, batch, buckets)
        l32, g32 = run_grads(ref, batch)
    with torch.no_grad():
        for n, b in model.named_buffers():
            b.copy_(buf[n])
    s = summarize(gradient_distance(g16, g32))
    s.update(loss_bf16=round(l16, 6), loss_fp32=round(l32, 6), loss_rel=round(abs(l16 - l32) / abs(l32), 7),
             note="bf16 step vs the fp32 HIP step (the mode pinned to the CPU oracle), same weights and batch, dropout "
                  "off; backbone deviation = ReLU-mask flips of a 16-bit forward, equal to stock torch.autocast(bf16) of "
                  "the reference (profiles/r02_bf16_rounding_mechanism.txt)")
    del ref
    return s
