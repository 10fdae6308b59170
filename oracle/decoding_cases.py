"""Deterministic stand-in for ``CaptioningModel.decoding_step`` plus the decoder configurations used to pin
virtex_amd/decoding.py against the reference's decoders (test infrastructure; see oracle/make_goldens.py)."""
import torch

V, EOS, SOS = 97, 2, 1

# name -> (kind, constructor kwargs, batch size, step seed)
CASES = {
    "beam5_node2": ("beam", dict(eos_index=EOS, max_steps=12, beam_size=5, per_node_beam_size=2), 4, 0),
    "beam3_node3": ("beam", dict(eos_index=EOS, max_steps=20, beam_size=3, per_node_beam_size=3), 3, 1),
    "beam1_greedy": ("beam", dict(eos_index=EOS, max_steps=10, beam_size=1, per_node_beam_size=1), 5, 2),
    "beam4_all": ("beam_all", dict(eos_index=EOS, max_steps=9, beam_size=4, per_node_beam_size=2), 2, 3),
    "nucleus09": ("nucleus", dict(eos_index=EOS, max_steps=15, nucleus_size=0.9), 6, 4),
    "nucleus05": ("nucleus", dict(eos_index=EOS, max_steps=15, nucleus_size=0.5), 3, 5),
}


def make_step(seed: int):
    """Next-token logits as a fixed function of the prefix (last token, first token, length): both
    implementations see exactly the same numbers.  EOS grows more likely with length so captions end."""
    g = torch.Generator().manual_seed(1234 + seed)
    by_last = 2.0 * torch.randn(V, V, generator=g)
    by_first = 0.3 * torch.randn(V, V, generator=g)
    by_len = 0.5 * torch.randn(64, V, generator=g)

    def step(partial: torch.Tensor) -> torch.Tensor:
        if partial.dim() == 1:
            partial = partial.unsqueeze(1)
        n = partial.size(1)
        logits = by_last[partial[:, -1]] + by_first[partial[:, 0]] + by_len[n]
        logits[:, EOS] += 0.6 * n - 2.0
        return logits

    return step


def run(decoder_cls_beam, decoder_cls_nucleus, name):
    kind, kw, B, seed = CASES[name]
    start = torch.full((B,), SOS, dtype=torch.long)
    step = make_step(seed)
    torch.manual_seed(99 + seed)                      # nucleus sampling draws from the default generator
    if kind == "nucleus":
        tokens, lp = decoder_cls_nucleus(**kw).search(start, step)
        return tokens, None
    dec = decoder_cls_beam(**kw)
    return dec.search(start, step, only_return_best=(kind == "beam"))
