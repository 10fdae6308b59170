"""Seeded synthetic inputs and parameter perturbations (TEST INFRASTRUCTURE).

Batch format = the 5-key dict produced by the reference collate function
(virtex/data/datasets/captioning.py:79-100) and consumed by
virtex/models/captioning.py:71-138; synthetic recipe = SURVEY.md section 8(d).
"""
from typing import Dict, Optional

import torch


def synthetic_batch(batch_size: int, image_size: int = 224, max_len: int = 30,
                    vocab_size: int = 10000, seed: int = 0, ragged: bool = False,
                    device: str = "cpu") -> Dict[str, torch.Tensor]:
    """image ~ N(0,1) fp32 (B,3,S,S); captions: [SOS]=1 ... [EOS]=2, interior uniform in
    [4, V); ``noitpac_tokens`` = the unpadded caption reversed, right-padded with 0
    (virtex/data/datasets/captioning.py:75,89-93).  ragged=True draws lengths in
    [3, max_len] (at least one row keeps max_len, as the collate pads to the batch max)."""
    g = torch.Generator().manual_seed(seed)
    image = torch.randn(batch_size, 3, image_size, image_size, generator=g)
    tokens = torch.randint(4, vocab_size, (batch_size, max_len), generator=g)
    if ragged:
        lengths = torch.randint(3, max_len + 1, (batch_size,), generator=g)
        lengths[0] = max_len
    else:
        lengths = torch.full((batch_size,), max_len, dtype=torch.int64)
    cap = torch.zeros(batch_size, max_len, dtype=torch.int64)
    rev = torch.zeros(batch_size, max_len, dtype=torch.int64)
    for b in range(batch_size):
        n = int(lengths[b])
        row = tokens[b, :n].clone()
        row[0], row[n - 1] = 1, 2
        cap[b, :n] = row
        rev[b, :n] = row.flip(0)
    batch = {"image_id": torch.arange(batch_size), "image": image, "caption_tokens": cap,
             "noitpac_tokens": rev, "caption_lengths": lengths}
    return {k: v.to(device) for k, v in batch.items()}


@torch.no_grad()
def randomize_state(model: torch.nn.Module, seed: int = 1234) -> None:
    """Make parity tests non-degenerate: zero_init_residual leaves every
    bn3.weight == 0 and all biases/BN stats trivial.  Draw BN affine params and running
    stats, LayerNorm affine params and all biases from a seeded generator, walking
    parameters in name order so the reference and every re-implementation get identical
    values.  Weights of convs / linears / embeddings keep their reference init."""
    g = torch.Generator().manual_seed(seed)
    named = dict(model.named_parameters())
    named.update(dict(model.named_buffers()))
    for name in sorted(named):
        t = named[name]
        leaf = name.rsplit(".", 1)[-1]
        is_norm = (".bn" in name or "downsample.1" in name or "norm" in name)
        if leaf == "num_batches_tracked":
            continue
        if leaf == "running_mean":
            t.copy_(0.1 * torch.randn(t.shape, generator=g))
        elif leaf == "running_var":
            t.copy_(0.5 + torch.rand(t.shape, generator=g))
        elif is_norm and leaf == "weight":
            t.copy_(0.5 + torch.rand(t.shape, generator=g))
        elif leaf in ("bias", "in_proj_bias"):
            t.copy_(0.1 * torch.randn(t.shape, generator=g))


def seeded_model(builder, seed: int = 0, randomize: bool = True, **kw):
    """torch.manual_seed(seed) -> builder(**kw) -> randomize_state().  Weights are defined
    by the ORACLE's construction order; the reference model (and the HIP modules) receive
    them through load_state_dict, never through their own RNG consumption."""
    torch.manual_seed(seed)
    model = builder(**kw)
    if randomize:
        randomize_state(model, seed + 1234)
    return model
