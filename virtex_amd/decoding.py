"""Caption decoders for the inference branch of the captioning models.

Drop-ins for the reference's ``AutoRegressiveBeamSearch`` (/root/reference/virtex/utils/beam_search.py:24-238)
and ``AutoRegressiveNucleusSampling`` (/root/reference/virtex/utils/nucleus_sampling.py:25-123): same
constructors, same ``search(start_predictions, step)`` contract, same outputs for the same ``step`` function
(tests/test_decoding.py checks token-for-token equality against the reference classes and committed goldens).

``step(partial_captions)`` is ``CaptioningModel.decoding_step`` with the image features bound: it re-runs the
text head on the whole prefix (the reference has no KV cache, captioning.py:165-213) through the HIP kernels.
Everything here is batched -- no per-row Python loops (the reference walks the batch x beam rows in Python at
beam_search.py:156-157 and nucleus_sampling.py:95-101).  When the logits live where the HIP library runs, a beam-search
step is ONE C-ABI call (`vtx_beam_step`: log-softmax, repetition penalty, finished-beam rule, per-beam and per-image
top-k on the device); on plain CPU tensors the same arithmetic runs as torch tensor operations.  `IncrementalDecodingStep`
(below) is the KV-cached step function the models hand to these decoders.
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
        first_logits = step(start_predictions)
        if _hip_available(first_logits) and W <= 16 and P <= 16 and W * P <= 64 and W <= first_logits.size(1):
            return self._search_on_device(first_logits, step, only_return_best)
        logp = torch.log_softmax(first_logits, dim=1)                          # (B, V)
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


    def _search_on_device(self, first_logits, step, only_return_best):
        """The same search with every step's selection done by one `vtx_beam_step` call (csrc/beam.hip)."""
        from . import ops
        B, W, P, eos = first_logits.size(0), self.beam_size, self.per_node_beam_size, self._eos_index
        score, _, first = ops.beam_step(first_logits.float().contiguous(), None, None, B, W, W, eos)     # (B, W)
        if W == 1 and bool((first == eos).all()):
            warnings.warn("Empty captions predicted. You may want to increase beam size or ensure your step "
                          "function is working properly.", RuntimeWarning)
            return first.unsqueeze(-1), score
        beams = first.unsqueeze(-1)                                            # (B, W, 1)
        rows = torch.arange(B, device=first.device).view(B, 1)
        parent_rows = rows.expand(B, W).reshape(-1)                            # the W beams of an image continue its one row
        for _ in range(self.max_steps - 1):
            last = beams[:, :, -1].reshape(B * W).contiguous()
            if bool((last == eos).all()):
                break
            if hasattr(step, "reorder"):
                step.reorder(parent_rows)                                      # KV-cached step functions follow the beams
            logits = step(beams.view(B * W, -1)).float().contiguous()          # (B*W, V)
            score, parent, tok = ops.beam_step(logits, last, score.reshape(-1).contiguous(), B, P, W, eos)
            parent_rows = (parent + rows * W).reshape(-1)
            beams = torch.cat([beams.gather(1, parent.unsqueeze(-1).expand(B, W, beams.size(-1))), tok.unsqueeze(-1)], dim=-1)
        if not bool(torch.isfinite(score).all()):
            warnings.warn("Infinite log probs encountered. Some final captions may not make sense. This can happen "
                          "when the beam size is larger than the number of valid (non-zero probability) "
                          "transitions that the step function produces.", RuntimeWarning)
        if only_return_best:
            return beams[:, 0, :], score[:, 0]
        return beams, score


def _hip_available(t: torch.Tensor) -> bool:
    """Can the C-ABI kernels run on this tensor?  GPU tensors: always (the library is mandatory there); CPU tensors: only
    when a test has pointed the bindings at the kernel emulator -- plain CPU use keeps the torch arithmetic."""
    if t.is_cuda:
        return True
    from . import _lib
    return _lib._lib is not None and _lib.is_emulator()


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


# ------------------------------------------------------------------------------------------------------------------
# KV-cached incremental decoding.  The reference's `decoding_step` re-runs the text head on the
# whole prefix at every step (captioning.py:165-213: O(T^2) projections and attention, and the visual projection plus
# the cross-attention K/V projection of all 49 grid cells once per step and BEAM).  Here a step costs one token:
#   * visual projection and the cross-attention key/value projection once per image (shared by its beams and steps),
#   * per layer a self-attention K/V cache (rows x max_len x H), appended to at every step and re-ordered when the
#     decoder re-orders its rows (a beam is continued from another beam's prefix),
#   * the single new position goes through the same HIP kernels as training (embedding+LN, projections, one-query
#     attention with the cached keys masked by length, post-norm LayerNorms, FFN, tied output projection).
# It is a drop-in `step` for the reference's decoders too: which cached row a new row continues is recovered from the
# prefixes themselves, no cooperation of the decoder needed (ours passes the parents explicitly: `reorder`).
# ------------------------------------------------------------------------------------------------------------------
class IncrementalDecodingStep:
    def __init__(self, head, visual_features: torch.Tensor):
        from . import ops
        from .modules.textual_heads import _nhwc_rows
        self.ops, self.head = ops, head
        self.dt = head.compute_dtype
        H = head.hidden_size
        self.H, self.A = H, head.attention_heads
        self.Tmax = head.embedding.positions.num_embeddings
        mem_in, self.B, self.S = _nhwc_rows(visual_features, self.dt)
        self.dev = mem_in.device
        Wv, _ = ops.prepped(head.visual_projection.weight, self.dt, want_wt=False)
        mem = ops.gemm_nt(mem_in, Wv.view(H, -1), bias=head.visual_projection.bias.detach())            # (B*S, H)
        self.layers = []
        for layer in head.transformer.layers:
            w = {}
            for name, p in (("Win", layer.self_attn.in_proj_weight), ("Wo", layer.self_attn.out_proj.weight),
                            ("Win2", layer.multihead_attn.in_proj_weight), ("Wo2", layer.multihead_attn.out_proj.weight),
                            ("W1", layer.linear1.weight), ("W2", layer.linear2.weight)):
                w[name] = ops.prepped(p, self.dt, want_wt=False)[0].view(p.shape)
            kv_img = ops.gemm_nt(mem, w["Win2"][H:], bias=layer.multihead_attn.in_proj_bias.detach()[H:])   # (B*S, 2H) once per image
            self.layers.append(dict(layer=layer, w=w, kv_img=kv_img, kv=None, K=None, V=None))
        self.Wout = ops.prepped(head.output.weight, self.dt, want_wt=False)[0].view(head.output.weight.shape)
        self.rows = 0                # rows of the previous call
        self.prefix = None           # their prefixes (N, t)
        self.pending = None          # parents announced by the decoder for the next call

    # -- row bookkeeping ----------------------------------------------------------------------------------------
    def reorder(self, parent: torch.Tensor):
        """parent[i] = row of the PREVIOUS call that row i of the next call continues."""
        self.pending = parent.reshape(-1)

    def _parents(self, partial):
        """Which cached row does each row of `partial` (N, t) continue?  Rows of one image are contiguous; inside an
        image's group the parent is the cached row with the identical prefix (identical prefixes have identical caches)."""
        N, t = partial.shape
        if self.pending is not None and self.pending.numel() == N:
            parent, self.pending = self.pending, None
            return parent
        gp, gn = self.rows // self.B, N // self.B
        old = self.prefix.view(self.B, 1, gp, t - 1)
        new = partial[:, : t - 1].reshape(self.B, gn, 1, t - 1)
        eq = (old == new).all(-1)                                               # (B, gn, gp)
        idx = eq.to(torch.int8).argmax(-1)                                      # first identical prefix
        if not bool(eq.any(-1).all()):
            raise RuntimeError("incremental decoding: a prefix does not continue any cached prefix")
        return (idx + torch.arange(self.B, device=partial.device).view(self.B, 1) * gp).reshape(-1)

    def _regroup(self, parent, N, t):
        """Re-order / expand the per-row state: self-attention caches by parent, cross-attention K/V by image."""
        identity = N == self.rows and bool((parent == torch.arange(N, device=parent.device)).all())
        for L in self.layers:
            if L["K"] is not None and not identity:
                L["K"] = L["K"].index_select(0, parent)
                L["V"] = L["V"].index_select(0, parent)
            if L["kv"] is None or L["kv"].shape[0] != N * self.S:
                g = N // self.B
                kv = L["kv_img"].view(self.B, 1, self.S, 2 * self.H)
                L["kv"] = (kv.expand(self.B, g, self.S, 2 * self.H).reshape(N * self.S, 2 * self.H) if g > 1
                           else L["kv_img"])
        self.rows = N

    # -- one step -----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, partial_captions: torch.Tensor) -> torch.Tensor:
        ops, head, H, A, dt = self.ops, self.head, self.H, self.A, self.dt
        if partial_captions.dim() == 1:
            partial_captions = partial_captions.unsqueeze(1)
        partial = partial_captions.contiguous()
        N, t = partial.shape
        if t > self.Tmax:
            raise IndexError(f"prefix of {t} tokens exceeds max_caption_length = {self.Tmax}")
        if t == 1:
            if N % self.B:
                raise ValueError("rows must be a multiple of the number of images")
            for L in self.layers:
                L["K"] = L["V"] = None
            self.rows = N
            self._regroup(torch.arange(N, device=partial.device), N, t)
        else:
            self._regroup(self._parents(partial), N, t)
        self.prefix = partial
        pos = t - 1
        emb = head.embedding
        tok = partial[:, pos:].contiguous()                                     # (N, 1)
        x, _, _ = ops.embedding_fwd(tok, emb.words.weight.detach(), emb.positions.weight.detach()[pos:], emb.layer_norm.weight.detach(),
                                    emb.layer_norm.bias.detach(), dt, emb.padding_idx, emb.layer_norm.eps, 0.0, 0)
        x = x.view(N, H)
        lens = torch.full((N,), t, dtype=torch.int64, device=partial.device)
        for L in self.layers:
            layer, w = L["layer"], L["w"]
            if L["K"] is None:
                L["K"] = torch.zeros(N, self.Tmax, H, dtype=dt, device=partial.device)
                L["V"] = torch.zeros(N, self.Tmax, H, dtype=dt, device=partial.device)
            pre = getattr(head, "norm_first", False)

            def norm(v, ln):                 # pre-norm layers normalise the sub-layer INPUT, the join is then a plain sum
                return ops.layernorm_residual_fwd(v, None, ln.weight.detach(), ln.bias.detach(), ln.eps)[0]
            h1 = norm(x, layer.norm1) if pre else x
            qkv = ops.gemm_nt(h1, w["Win"], bias=layer.self_attn.in_proj_bias.detach())                 # (N, 3H)
            L["K"][:, pos] = qkv[:, H:2 * H]
            L["V"][:, pos] = qkv[:, 2 * H:]
            o1 = ops.attention_fwd(qkv[:, :H], L["K"].view(N * self.Tmax, H), L["V"].view(N * self.Tmax, H), N, A, 1, self.Tmax,
                                   False, lens)
            if pre:
                x1 = ops.gemm_nt(o1, w["Wo"], bias=layer.self_attn.out_proj.bias.detach(), residual=x)
                q2 = ops.gemm_nt(norm(x1, layer.norm2), w["Win2"][:H], bias=layer.multihead_attn.in_proj_bias.detach()[:H])
                o2 = ops.attention_fwd(q2, L["kv"][:, :H], L["kv"][:, H:], N, A, 1, self.S, False, None)
                x2 = ops.gemm_nt(o2, w["Wo2"], bias=layer.multihead_attn.out_proj.bias.detach(), residual=x1)
                a = ops.gemm_nt(norm(x2, layer.norm3), w["W1"], bias=layer.linear1.bias.detach(), act=ops.ACT_GELU)
                x = ops.gemm_nt(a, w["W2"], bias=layer.linear2.bias.detach(), residual=x2)
                continue
            y1 = ops.gemm_nt(o1, w["Wo"], bias=layer.self_attn.out_proj.bias.detach())
            x1, _, _ = ops.layernorm_residual_fwd(x, y1, layer.norm1.weight.detach(), layer.norm1.bias.detach(), layer.norm1.eps)
            q2 = ops.gemm_nt(x1, w["Win2"][:H], bias=layer.multihead_attn.in_proj_bias.detach()[:H])
            o2 = ops.attention_fwd(q2, L["kv"][:, :H], L["kv"][:, H:], N, A, 1, self.S, False, None)
            y2 = ops.gemm_nt(o2, w["Wo2"], bias=layer.multihead_attn.out_proj.bias.detach())
            x2, _, _ = ops.layernorm_residual_fwd(x1, y2, layer.norm2.weight.detach(), layer.norm2.bias.detach(), layer.norm2.eps)
            a = ops.gemm_nt(x2, w["W1"], bias=layer.linear1.bias.detach(), act=ops.ACT_GELU)
            y3 = ops.gemm_nt(a, w["W2"], bias=layer.linear2.bias.detach())
            x, _, _ = ops.layernorm_residual_fwd(x2, y3, layer.norm3.weight.detach(), layer.norm3.bias.detach(), layer.norm3.eps)
        if getattr(head, "norm_first", False):                                  # the stack's closing LayerNorm (textual_heads.py:192-193)
            fn = head.transformer.norm
            x = ops.layernorm_residual_fwd(x, None, fn.weight.detach(), fn.bias.detach(), fn.eps)[0]
        return ops.gemm_nt(x, self.Wout, bias=head.output.bias.detach(), out_f32=True)                  # (N, V) fp32 logits
