"""Caption decoders for the inference branch of the captioning models.

Drop-ins for the reference's ``AutoRegressiveBeamSearch`` (/root/reference/virtex/utils/beam_search.py:24-238)
and ``AutoRegressiveNucleusSampling`` (/root/reference/virtex/utils/nucleus_sampling.py:25-123): same
constructors, same ``search(start_predictions, step)`` contract, same outputs for the same ``step`` function
(tests/test_decoding.py checks token-for-token equality against the reference classes and committed goldens).

``step(partial_captions)`` is ``CaptioningModel.decoding_step`` with the image features bound: it re-runs the
text head on the whole prefix (the reference has no KV cache, captioning.py:165-213) through the HIP kernels.
Everything here is batched tensor bookkeeping on the device the logits live on -- no per-row Python loops
(the reference walks the batch x beam rows in Python at beam_search.py:156-157 and nucleus_sampling.py:95-101).
"""
import warnings
from typing import Callable, Optional, Tuple

import torch

_REPEAT_PENALTY = -10000.0      # beam search: log-prob given to "same token again" (beam_search.py:157)
_REMOVED_LOGIT = -1e12          # nucleus sampling: logit of tokens outside the nucleus (nucleus_sampling.py:97-101)


class AutoRegressiveBeamSearch:
    def __init__(self, eos_index: int, max_steps: int = 50, beam_size: int = 5, per_node_beam_size: int = 2) -> None:
        self._eos_index = eos_index
        self.max_steps = max_steps
        self.beam_size = beam_size
        self.per_node_beam_size = per_node_beam_size or beam_size

    @torch.no_grad()
    def search(self, start_predictions: torch.Tensor, step: Callable[..., torch.Tensor],
               only_return_best: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (predictions (B, len) or (B, beam, len), log-probabilities (B,) or (B, beam)); the start tokens
        are implicit (not part of the returned sequences)."""
        B, W, P, eos = start_predictions.size(0), self.beam_size, self.per_node_beam_size, self._eos_index
        logp = torch.log_softmax(step(start_predictions), dim=1)               # (B, V)
        V = logp.size(1)
        score, first = logp.topk(W)                                            # (B, W) each
        if W == 1 and bool((first == eos).all()):
            warnings.warn("Empty captions predicted. You may want to increase beam size or ensure your step "
                          "function is working properly.", RuntimeWarning)
            return first.unsqueeze(-1), score
        beams = first.unsqueeze(-1)                                            # (B, W, 1)
        finished_row = logp.new_full((V,), float("-inf"))                      # a finished beam may only emit EOS again
        finished_row[eos] = 0.0
        for _ in range(self.max_steps - 1):
            last = beams[:, :, -1].reshape(B * W)
            if bool((last == eos).all()):
                break
            logp = torch.log_softmax(step(beams.view(B * W, -1)), dim=1)      # (B*W, V)
            logp.scatter_(1, last.unsqueeze(1), _REPEAT_PENALTY)              # no immediate repetition
            logp = torch.where((last == eos).unsqueeze(1), finished_row.unsqueeze(0), logp)
            cand_lp, cand_tok = logp.topk(P)                                   # (B*W, P)
            total = (cand_lp.view(B, W, P) + score.unsqueeze(2)).view(B, W * P)
            score, pick = total.topk(W)                                        # (B, W) indices into W*P
            parent = torch.div(pick, P, rounding_mode="floor")
            beams = torch.cat([beams.gather(1, parent.unsqueeze(-1).expand(B, W, beams.size(-1))),
                               cand_tok.view(B, W * P).gather(1, pick).unsqueeze(-1)], dim=-1)
        if not bool(torch.isfinite(score).all()):
            warnings.warn("Infinite log probs encountered. Some final captions may not make sense. This can happen "
                          "when the beam size is larger than the number of valid (non-zero probability) "
                          "transitions that the step function produces.", RuntimeWarning)
        if only_return_best:
            return beams[:, 0, :], score[:, 0]
        return beams, score


class AutoRegressiveNucleusSampling:
    def __init__(self, eos_index: int, max_steps: int = 50, nucleus_size: float = 0.9):
        self._eos_index = eos_index
        self.max_steps = max_steps
        self.nucleus_size = nucleus_size

    @torch.no_grad()
    def search(self, start_predictions: torch.Tensor, step: Callable[..., torch.Tensor]) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """-> (sampled tokens (B, <= max_steps), None).  Draws come from ``torch.multinomial`` on the device's
        default generator, one call per step with the same probabilities as the reference computes."""
        B, eos = start_predictions.size(0), self._eos_index
        seq = start_predictions.view(B, 1)
        for _ in range(self.max_steps):
            last = seq[:, -1]
            if bool((last == eos).all()):
                break
            logits = step(seq)                                                 # (B, V); filtered in place like the reference
            ordered, order = torch.sort(logits, descending=True)
            mass = torch.cumsum(torch.softmax(ordered, dim=-1), dim=-1)
            outside = mass > self.nucleus_size
            outside = torch.cat([torch.zeros_like(outside[:, :1]), outside[:, :-1]], dim=1)   # keep the token that crosses the threshold
            drop = torch.zeros_like(outside).scatter_(1, order, outside)      # back to vocabulary order
            drop.scatter_(1, last.unsqueeze(1), True)                          # and never the previous token again
            logits.masked_fill_(drop, _REMOVED_LOGIT)
            nxt = torch.multinomial(torch.softmax(logits, dim=-1), 1).view(B)
            nxt = torch.where(last == eos, torch.full_like(nxt, eos), nxt)
            seq = torch.cat([seq, nxt.unsqueeze(1)], dim=1)
        return seq[:, 1:], None
