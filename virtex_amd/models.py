"""MI355X-native bicaptioning model: drop-in for the reference's ``CaptioningModel`` family
(/root/reference/virtex/models/captioning.py:13-283).

Same constructor, attribute tree (``visual``, ``textual``, ``backward_textual``, ``loss``,
``padding_idx``, ``sos_index``, ``eos_index``, ``decoder``), weight tying and output dict:
``{"loss", "loss_components": {"captioning_forward", "captioning_backward"}, ["predictions"]}``.

Training takes a fused path the reference does not have: decoder hidden states go straight
into the tied-projection + cross-entropy op (`_FusedTiedCrossEntropyFn`, csrc/tied_ce.hip): the
(B,T,V) logits exist only as MFMA accumulators -- the projection's epilogue emits log-sum-exp
partials, the backward recomputes the projection and emits the logit gradient in the compute
dtype.  The eval branch and ``decoding_step`` keep the reference semantics through
``textual(...)`` which returns logits.
"""
import copy
import functools
from typing import Any, Dict

import torch
from .replay import traced_backward
from torch import nn

import os

from . import ops
from .streams import branch_stream
from .modules.textual_heads import TextualHead, tied_projection_grads
from .modules.visual_backbones import VisualBackbone


# Training path: tied projection + cross-entropy with the logits living only in MFMA accumulators (csrc/tied_ce.hip).
# "0" selects the round-1 path (fp32 logits written by the GEMM, read twice by the loss kernels) for A/B runs.
FUSED_TIED_CE = os.environ.get("VIRTEX_AMD_FUSED_CE", "1") != "0"


class _FanoutFn(torch.autograd.Function):
    """y1, y2 = x, x -- with the SUM of the two gradients computed by the library (vtx_add) instead of by the autograd engine,
    and the stream dependency of that sum made explicit (the second consumer's backward runs on the branch stream).  Same
    arithmetic as the engine's accumulation; it exists so that every launch of the step is visible to virtex_amd.replay."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x)

    @staticmethod
    @traced_backward
    def backward(ctx, d1, d2):
        if d1 is None or d2 is None:
            return d1 if d2 is None else d2
        from . import ops
        from .streams import branch_stream
        branch_stream.join(d2.device)
        return ops.add(d1.contiguous(), d2.contiguous())


class _LossSumFn(torch.autograd.Function):
    """loss = forward-captioning loss (compute stream) + backward-captioning loss (branch stream).  The gradient of the sum is
    produced on the compute stream and consumed by the branch head's backward on the branch stream: the autograd engine orders
    the two streams by itself, in C++ -- invisible to virtex_amd.replay, whose recording would lose that edge.  Here the branch
    stream waits for the compute stream explicitly (a recorded stream operation; harmless next to the engine's own)."""

    @staticmethod
    def forward(ctx, a, b, branch):
        ctx.branch = branch
        return a + b

    @staticmethod
    @traced_backward
    def backward(ctx, g):
        if ctx.branch is not None and g.is_cuda:
            ctx.branch.wait_stream(torch.cuda.current_stream(g.device))
            g.record_stream(ctx.branch)
        return g, g, None


class _FusedTiedCrossEntropyFn(torch.autograd.Function):
    """mean_{tok[b,t+1] != pad} CE(hidden[b,t] @ words^T + bias, tok[b,t+1]) without (B,T,V) logits in HBM
    (reference: captioning.py:111-114 on textual_heads.py:277 logits)."""

    @staticmethod
    def forward(ctx, hidden, weight, bias, tokens, padding_idx):
        B, T, H = hidden.shape
        dt = hidden.dtype
        targets = torch.full((B, T), padding_idx, dtype=torch.int64, device=tokens.device)
        targets[:, :-1] = tokens[:, 1:]                    # row (b,t) predicts token t+1; the last step has no target
        targets = targets.view(-1)
        w, _ = ops.prepped(weight, dt, want_wt=False)
        h2 = hidden.reshape(B * T, H)
        lc, lse = ops.tied_ce_fwd(h2, w.view(weight.shape), bias.detach(), targets, padding_idx)
        ctx.weight_param, ctx.bias_param = weight, bias
        ctx.save_for_backward(h2, targets, lse, lc)
        ctx.cfg = (B, T, H, weight.shape[0], padding_idx, dt)
        return lc[0].clone()

    @staticmethod
    @traced_backward
    def backward(ctx, gout):
        h2, targets, lse, lc = ctx.saved_tensors
        B, T, H, V, padding_idx, dt = ctx.cfg
        g = gout.reshape(1).to(torch.float32).contiguous()
        w, wt = ops.prepped(ctx.weight_param, dt)
        d = ops.tied_ce_bwd(h2, w.view(V, H), ctx.bias_param.detach(), targets, lse, lc, g, padding_idx)     # (B*T, V) compute dtype
        dh = ops.gemm_nt(d, wt.view(H, V)).view(B, T, H)
        rW, rb = tied_projection_grads(d, h2, ctx.weight_param, ctx.bias_param)
        return dh, rW, rb, None, None


class _TiedCrossEntropyFn(torch.autograd.Function):
    """The same with materialised fp32 logits (round-1 path; A/B reference for the fused one)."""

    @staticmethod
    def forward(ctx, hidden, weight, bias, tokens, padding_idx):
        B, T, H = hidden.shape
        dt = hidden.dtype
        V = weight.shape[0]
        # row (b,t) predicts token t+1; the last step has no target -> ignore
        targets = torch.full((B, T), padding_idx, dtype=torch.int64, device=tokens.device)
        targets[:, :-1] = tokens[:, 1:]
        targets = targets.view(-1)
        w, _ = ops.prepped(weight, dt, want_wt=False)
        h2 = hidden.reshape(B * T, H)
        logits = ops.gemm_nt(h2, w.view(weight.shape), bias=bias.detach(), out_f32=True)
        ctx.weight_param, ctx.bias_param = weight, bias
        lc, lse = ops.cross_entropy_fwd(logits, targets, padding_idx)
        ctx.save_for_backward(h2, weight, targets, logits, lse, lc)
        ctx.cfg = (B, T, H, V, padding_idx, dt)
        return lc[0].clone()

    @staticmethod
    @traced_backward
    def backward(ctx, gout):
        h2, weight, targets, logits, lse, lc = ctx.saved_tensors
        B, T, H, V, padding_idx, dt = ctx.cfg
        g = gout.reshape(1).to(torch.float32).contiguous()
        d = ops.cross_entropy_bwd(logits, targets, lse, lc, g, dt, padding_idx)      # (B*T, V) compute dtype
        _, wt = ops.prepped(ctx.weight_param, dt, want_w=False)
        dh = ops.gemm_nt(d, wt.view(H, V)).view(B, T, H)
        rW, rb = tied_projection_grads(d, h2, ctx.weight_param, ctx.bias_param)
        return dh, rW, rb, None, None


# The tied `visual_projection` evaluated once per step for both heads instead of once per head as the reference does
# (textual_heads.py:245 sits inside forward): saves one 12544x1024x2048 GEMM forward, input gradient and weight gradient
# (0.62 GFLOP per image).  "0" = the reference's evaluation order (faithful mode for bit-level comparisons).
SHARE_VISUAL_PROJECTION = os.environ.get("VIRTEX_AMD_SHARE_VISUAL_PROJECTION", "1") != "0"

# Inference: KV-cached incremental decoding (decoding.IncrementalDecodingStep) instead of re-running the whole prefix at
# every step like the reference's decoding_step (kept: `model.decoding_step`, and "0" here, for comparisons).
INCREMENTAL_DECODING = os.environ.get("VIRTEX_AMD_INCREMENTAL_DECODING", "1") != "0"

# compute copies of the text heads' weights prepared on the branch stream under the backbone's first kernels instead of
# in front of the whole forward pass (the one preparation launch is 0.18 ms at the head of every step)
SPLIT_WEIGHT_PREP = os.environ.get("VIRTEX_AMD_SPLIT_WEIGHT_PREP", "0") != "0"   # measured neutral (profiles/r03_ab_session5.txt): off

# the two caption directions on two streams (A/B switch; see DESIGN.md, Streams)
HEAD_STREAMS = os.environ.get("VIRTEX_AMD_HEAD_STREAMS", "1") != "0"
# host order of the two heads' forward passes when they run on two streams ("0" = round-1 order, for A/B runs)
HEAD_ORDER_BRANCH_FIRST = os.environ.get("VIRTEX_AMD_HEAD_ORDER_BRANCH_FIRST", "1") != "0"
if HEAD_STREAMS and hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
    # gradients of the backward-captioning head are produced on the branch stream on purpose; autograd's
    # AccumulateGrad nodes synchronise with it correctly, the warning is only about that extra synchronisation
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)


class CaptioningModel(nn.Module):
    def __init__(self, visual: VisualBackbone, textual: TextualHead, caption_backward: bool = False,
                 sos_index: int = 1, eos_index: int = 2, decoder: Any = None):
        super().__init__()
        self.visual = visual
        self.textual = textual
        self.padding_idx = self.textual.padding_idx
        self.caption_backward = caption_backward
        if self.caption_backward:
            self.backward_textual = copy.deepcopy(self.textual)
            self.backward_textual.visual_projection = self.textual.visual_projection
            self.backward_textual.embedding = self.textual.embedding
            self.backward_textual.output = self.textual.output
        self.sos_index = sos_index
        self.eos_index = eos_index
        self.decoder = decoder
        self.loss = nn.CrossEntropyLoss(ignore_index=self.padding_idx)  # kept for API parity

    def _head_loss(self, head, visual_features, tokens, lengths, memory=None):
        hidden = head.features(visual_features, tokens, lengths, memory=memory) if memory is not None else \
            head.features(visual_features, tokens, lengths)
        fn = _FusedTiedCrossEntropyFn if FUSED_TIED_CE else _TiedCrossEntropyFn
        return fn.apply(hidden, head.output.weight, head.output.bias, tokens, self.padding_idx)

    def _refresh_compute_weights(self):
        """One launch for all the bf16/fp32 compute copies (and their transposes) the step will use; weights whose
        copies are current (no optimizer step / in-place edit since) cost nothing."""
        seen, vis_items, text_items = set(), [], []
        heads = [self.textual] + ([self.backward_textual] if self.caption_backward else [])
        for it in (self.visual.weight_plan() if hasattr(self.visual, "weight_plan") else []):
            if id(it[0]) not in seen:
                seen.add(id(it[0]))
                vis_items.append(it)
        for h in heads:
            for it in (h.weight_plan() if hasattr(h, "weight_plan") else []):
                if id(it[0]) not in seen:
                    seen.add(id(it[0]))
                    text_items.append(it)
        dt = getattr(self.textual, "compute_dtype", None)
        if dt is None:
            return None
        dev = next(self.parameters()).device
        if SPLIT_WEIGHT_PREP and self.training and vis_items and text_items and dev.type == "cuda" and branch_stream.enabled:
            # the text heads' copies (2/3 of the bytes) are not needed before the backbone is through: prepared on the
            # branch stream under the stem / first stage; the caller makes the compute stream wait before the heads start
            br = branch_stream(dev)
            with br:
                ops.prep_many(text_items, dt)
            ops.prep_many(vis_items, dt)
            return br
        if vis_items or text_items:
            ops.prep_many(vis_items + text_items, dt)
        return None

    def forward(self, batch: Dict[str, torch.Tensor]) -> Dict[str, Any]:
        text_prep = self._refresh_compute_weights()
        emb = getattr(self.textual, "embedding", None)
        if emb is not None and hasattr(emb, "defer_join"):
            # the backbone's backward runs after both heads' and joins the weight-gradient side stream itself
            emb.defer_join = bool(self.training and torch.is_grad_enabled()
                                  and any(p.requires_grad for p in self.visual.parameters()))
        visual_features = self.visual(batch["image"])
        if text_prep is not None:
            text_prep.wait()               # the text heads' compute copies are ready (they were prepared beside the backbone)
        batch_size = visual_features.size(0)
        if "caption_tokens" in batch:
            caption_tokens = batch["caption_tokens"]
            caption_lengths = batch["caption_lengths"]
            br = None
            memory = None
            if (self.training and self.caption_backward and SHARE_VISUAL_PROJECTION
                    and hasattr(self.textual, "project_visual_features")
                    and self.backward_textual.visual_projection is self.textual.visual_projection):
                # both heads read the same projected grid: evaluate the shared projection once
                memory = self.textual.project_visual_features(visual_features)
            memory_b = memory
            if memory is not None and self.training and torch.is_grad_enabled() and memory.requires_grad:
                memory, memory_b = _FanoutFn.apply(memory)          # the fan-in of the two heads' gradients stays in the library
            backward_loss = None
            if self.training:
                if self.caption_backward and HEAD_STREAMS:
                    # the two heads are independent until their losses are added: the backward-captioning head runs
                    # on the branch stream (autograd replays each node on the stream its forward ran on).  It is
                    # enqueued first, so that in backward (autograd visits the nodes created last first) the compute
                    # stream's head is enqueued before autograd makes the compute stream wait for this head's gradient
                    # of the shared features.  Measured neutral (tools/head_overlap.py: with HIP events and no profiler
                    # attached both chains run 11.85 -> 14.8 ms side by side in either order; the one-after-the-other
                    # picture in rocprofv3 kernel traces is an artefact of the tracer).
                    br = branch_stream(visual_features.device, visual_features, batch["noitpac_tokens"], caption_lengths,
                                       *([memory_b] if memory_b is not None else []))
                    if HEAD_ORDER_BRANCH_FIRST:
                        with br:
                            backward_loss = self._head_loss(self.backward_textual, visual_features,
                                                            batch["noitpac_tokens"], caption_lengths, memory_b)
                    else:
                        br.mark()
                loss = self._head_loss(self.textual, visual_features, caption_tokens, caption_lengths, memory)
            else:
                output_logits = self.textual(visual_features, caption_tokens, caption_lengths)
                loss = _logits_loss(output_logits, caption_tokens, self.padding_idx)
            output_dict: Dict[str, Any] = {
                "loss": loss, "loss_components": {"captioning_forward": loss.clone().detach()}}
            if self.caption_backward:
                backward_caption_tokens = batch["noitpac_tokens"]
                if self.training and br is not None:
                    if backward_loss is None:
                        with br:
                            backward_loss = self._head_loss(self.backward_textual, visual_features,
                                                            backward_caption_tokens, caption_lengths, memory_b)
                    br.wait(backward_loss)
                elif self.training:
                    backward_loss = self._head_loss(self.backward_textual, visual_features,
                                                    backward_caption_tokens, caption_lengths, memory_b)
                else:
                    backward_loss = _logits_loss(
                        self.backward_textual(visual_features, backward_caption_tokens, caption_lengths),
                        backward_caption_tokens, self.padding_idx)
                if self.training and br is not None and br.active:
                    output_dict["loss"] = _LossSumFn.apply(output_dict["loss"], backward_loss, branch_stream.side(backward_loss.device))
                else:
                    output_dict["loss"] = output_dict["loss"] + backward_loss
                output_dict["loss_components"].update(captioning_backward=backward_loss.clone().detach())
            if not self.training:
                output_dict["predictions"] = torch.argmax(output_logits, dim=-1)
        else:
            if self.decoder is None:
                raise ValueError("Decoder for predicting captions is missing!")
            start_predictions = visual_features.new_full((batch_size,), self.sos_index).long()
            if INCREMENTAL_DECODING and not self.training and hasattr(self.textual, "transformer") and hasattr(self.textual, "compute_dtype"):
                from .decoding import IncrementalDecodingStep
                decoding_step = IncrementalDecodingStep(self.textual, visual_features)      # one token per step, K/V cached
            else:
                decoding_step = functools.partial(self.decoding_step, visual_features)
            predicted_caption, _ = self.decoder.search(start_predictions, decoding_step)
            output_dict = {"predictions": predicted_caption}
        return output_dict

    def decoding_step(self, visual_features: torch.Tensor, partial_captions: torch.Tensor) -> torch.Tensor:
        """Next-token logits (N*beam, V) for the prefixes decoded so far -- the step function the caption decoders
        call (reference semantics: captioning.py:165-213).  The whole prefix is re-run through the text head; the
        lengths passed are the prefix length, whatever [EOS] / padding it already contains; a 1-D argument is the
        first step (one token per image)."""
        n_img = visual_features.size(0)
        if partial_captions.dim() == 1:
            prefix, lengths = partial_captions.unsqueeze(1), torch.ones_like(partial_captions)
        else:
            prefix = partial_captions
            lengths = torch.full((prefix.size(0),), prefix.size(1), dtype=prefix.dtype, device=prefix.device)
        beams = prefix.size(0) // n_img
        if beams > 1:       # every beam of an image attends to that image's grid
            visual_features = visual_features.unsqueeze(1).expand(-1, beams, -1, -1, -1).reshape(
                n_img * beams, *visual_features.shape[1:])
        return self.textual(visual_features, prefix, lengths)[:, -1, :]


def _logits_loss(logits, tokens, padding_idx):
    B, T, V = logits.shape
    targets = tokens[:, 1:].contiguous().view(-1)
    lc, _ = ops.cross_entropy_fwd(logits[:, :-1].contiguous().view(-1, V), targets, padding_idx)
    return lc[0].clone()


class ForwardCaptioningModel(CaptioningModel):
    def __init__(self, visual, textual, sos_index=1, eos_index=2, decoder=None):
        super().__init__(visual, textual, sos_index=sos_index, eos_index=eos_index,
                         caption_backward=False, decoder=decoder)


class BidirectionalCaptioningModel(CaptioningModel):
    def __init__(self, visual, textual, sos_index=1, eos_index=2, decoder=None):
        super().__init__(visual, textual, sos_index=sos_index, eos_index=eos_index,
                         caption_backward=True, decoder=decoder)


# Convenient handle, as in the reference (captioning.py:282-283).
VirTexModel = BidirectionalCaptioningModel
