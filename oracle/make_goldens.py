"""Generate tests/golden/*.json from the VERBATIM reference (build container only).

Usage:  python -m oracle.make_goldens            (needs /root/reference)

For each case: build the oracle port with a fixed seed (that defines the weights), load
its state dict into the reference's own ``VirTexModel`` (imported from /root/reference,
see oracle/reference_import.py), run one train-mode forward/backward of the reference on
the seeded synthetic batch and record

* loss and both loss components,
* a strided sample of the forward/backward logits,
* for every named parameter: gradient L2 norm, sum, and 4 sampled entries,
* for every BatchNorm buffer after the step: L2 norm (running stats) / value (counter),
* (``<case>_eval.json``) the eval-mode forward of the same seeded state: loss, components, argmax
  predictions and a sample of the backbone features.

The fixtures are what pins the oracle on machines where /root/reference is absent
(the GPU box): tests/test_oracle.py replays the same seeds through the port.
"""
import json
import os
import sys

import torch

from oracle import bicaptioning as port
from oracle import reference_import, synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                          "tests", "golden")

CASES = {
    # name: (model kwargs, batch kwargs)
    "r50_l1_h1024_b2_full": (
        dict(textual="transdec_postnorm::L1_H1024_A16_F4096", vocab_size=10000),
        dict(batch_size=2, image_size=224, max_len=30, vocab_size=10000, seed=0, ragged=False)),
    "r50_l1_h1024_b2_ragged": (
        dict(textual="transdec_postnorm::L1_H1024_A16_F4096", vocab_size=10000),
        dict(batch_size=2, image_size=224, max_len=30, vocab_size=10000, seed=1, ragged=True)),
    "r50_l2_h128_b3_small": (
        dict(textual="transdec_postnorm::L2_H128_A2_F256", vocab_size=1000),
        dict(batch_size=3, image_size=64, max_len=12, vocab_size=1000, seed=2, ragged=True)),
    # the pre-norm decoder product (reference: factories.py "transdec_prenorm")
    "r50_l2_h128_b3_prenorm": (
        dict(textual="transdec_prenorm::L2_H128_A2_F256", vocab_size=1000),
        dict(batch_size=3, image_size=64, max_len=12, vocab_size=1000, seed=3, ragged=True)),
}


def summarize(model, out):
    g = {}
    for name, p in model.named_parameters():
        flat = p.grad.detach().flatten().double()
        idx = torch.linspace(0, flat.numel() - 1, 4).long()
        g[name] = {"norm": flat.norm().item(), "sum": flat.sum().item(),
                   "samples": flat[idx].tolist()}
    bufs = {}
    for name, b in model.named_buffers():
        bufs[name] = b.double().norm().item() if b.dtype.is_floating_point else int(b)
    return {
        "loss": out["loss"].item(),
        "loss_components": {k: v.item() for k, v in out["loss_components"].items()},
        "grads": g, "buffers": bufs,
    }


def logits_sample(logits):
    return logits.detach()[:, ::3, ::97].double().flatten().tolist()


def run_case(name, model_kw, batch_kw, use_reference=True):
    oracle_model = synth.seeded_model(port.build_model, seed=0, dropout=0.0, **model_kw)
    batch = synth.synthetic_batch(**batch_kw)
    if use_reference:
        model = reference_import.build_reference_model(dropout=0.0, **model_kw)
        model.load_state_dict(oracle_model.state_dict())
    else:
        model = oracle_model
    model.train()
    captured = {}
    hooks = [model.textual.register_forward_hook(
                 lambda m, i, o: captured.__setitem__("logits", o)),
             model.backward_textual.register_forward_hook(
                 lambda m, i, o: captured.__setitem__("backward_logits", o))]
    out = model(batch)
    for h in hooks:
        h.remove()
    out["loss"].backward()
    rec = summarize(model, out)
    rec["logits_sample"] = logits_sample(captured["logits"])
    rec["backward_logits_sample"] = logits_sample(captured["backward_logits"])
    return rec


def run_eval_case(model_kw, batch_kw, use_reference=True):
    """Eval-mode forward (running-statistics BatchNorm, no dropout) of the seeded state: what validation
    (scripts/pretrain_virtex.py validation loop) and the downstream feature extractors see."""
    oracle_model = synth.seeded_model(port.build_model, seed=0, dropout=0.0, **model_kw)
    batch = synth.synthetic_batch(**batch_kw)
    if use_reference:
        model = reference_import.build_reference_model(dropout=0.0, **model_kw)
        model.load_state_dict(oracle_model.state_dict())
    else:
        model = oracle_model
    model.eval()
    captured = {}
    hook = model.visual.register_forward_hook(lambda m, i, o: captured.__setitem__("features", o))
    with torch.no_grad():
        out = model(batch)
    hook.remove()
    feats = captured["features"].detach().double()
    return {
        "loss": out["loss"].item(),
        "loss_components": {k: v.item() for k, v in out["loss_components"].items()},
        "predictions": out["predictions"].tolist(),
        "features_norm": feats.norm().item(), "features_sum": feats.sum().item(),
        "features_sample": feats.flatten()[::max(1, feats.numel() // 64)][:64].tolist(),
    }


def main():
    if not reference_import.available():
        sys.exit("needs /root/reference")
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    only = sys.argv[1:]                      # optional: the cases to (re)generate; default all
    for name, (mkw, bkw) in CASES.items():
        if only and name not in only:
            continue
        rec = run_case(name, mkw, bkw, use_reference=True)
        rec["meta"] = {"model": mkw, "batch": bkw, "torch": torch.__version__,
                       "generator": "oracle/make_goldens.py (verbatim /root/reference classes)"}
        path = os.path.join(GOLDEN_DIR, name + ".json")
        with open(path, "w") as f:
            json.dump(rec, f, indent=0)
        print(name, "loss", rec["loss"], "->", path, os.path.getsize(path), "bytes")
        ev = run_eval_case(mkw, bkw, use_reference=True)
        ev["meta"] = rec["meta"]
        path = os.path.join(GOLDEN_DIR, name + "_eval.json")
        with open(path, "w") as f:
            json.dump(ev, f, indent=0)
        print(name, "eval loss", ev["loss"], "->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
