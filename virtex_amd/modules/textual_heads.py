"""MI355X-native textual head: drop-ins for the reference's ``WordAndPositionalEmbedding``
(/root/reference/virtex/modules/embedding.py:8-86) and ``TransformerDecoderTextualHead``
(/root/reference/virtex/modules/textual_heads.py:98-292).

Same constructor arguments, same attribute tree and parameter names
(``visual_projection.*``, ``embedding.{words,positions,layer_norm}.*``,
``transformer.layers.{l}.{self_attn,multihead_attn,linear1,linear2,norm1,norm2,norm3}.*``,
``output.*`` with ``output.weight is embedding.words.weight``), re-assignable
``visual_projection`` / ``embedding`` / ``output`` and ``copy.deepcopy``-able, as
``CaptioningModel.__init__`` requires (/root/reference/virtex/models/captioning.py:57-63).

``nn.Linear`` / ``nn.Embedding`` / ``nn.TransformerDecoder`` objects appear below ONLY as
parameter containers (they give the reference's state-dict keys for free); their forward
methods are never called.  All arithmetic is a hand-scheduled sequence of C-ABI kernel launches:
MFMA GEMMs with fused bias/GELU/dropout epilogues, one fused attention kernel per
(batch, head), fused residual+dropout+LayerNorm, fused embedding gather+LayerNorm.
"""
import itertools

import torch
from ..replay import traced_backward
from torch import nn

from .. import gradsink, ops
from ..streams import wgrad_stream

_seed_counter = [0]


def next_dropout_seed() -> int:
    """Host-side counter mixed with torch's seed: deterministic given torch.manual_seed."""
    _seed_counter[0] += 1
    return (torch.initial_seed() * 1000003 + _seed_counter[0] * 7919) & 0x7FFFFFFFFFFFFFFF


def dropout_seed_state(value=None) -> int:
    """Read (and, with an argument, set) the host counter the dropout seeds are drawn from: two runs started from the same value
    draw the same seeds (virtex_amd.replay validates a recording against an eager step that way)."""
    if value is not None:
        _seed_counter[0] = int(value)
    return _seed_counter[0]


class WordAndPositionalEmbedding(nn.Module):
    def __init__(self, vocab_size: int, hidden_size: int, dropout: float = 0.0,
                 max_caption_length: int = 30, padding_idx: int = 0):
        super().__init__()
        self.vocab_size, self.padding_idx = vocab_size, padding_idx
        self.words = nn.Embedding(vocab_size, hidden_size, padding_idx=padding_idx)
        self.positions = nn.Embedding(max_caption_length, hidden_size)
        self.layer_norm = nn.LayerNorm(hidden_size, eps=1e-8, elementwise_affine=True)
        self.dropout = nn.Dropout(p=dropout)
        self.compute_dtype = torch.bfloat16
        # set by the model when another backward node (the backbone's) is guaranteed to run after this one and
        # joins the weight-gradient side stream itself (virtex_amd/streams.py)
        self.defer_join = False

    def forward(self, tokens: torch.Tensor) -> torch.Tensor:
        p = self.dropout.p if self.training else 0.0
        return _EmbeddingFn.apply(tokens, self.words.weight, self.positions.weight, self.layer_norm.weight,
                                  self.layer_norm.bias, self.padding_idx, self.layer_norm.eps, p,
                                  self.compute_dtype, self.defer_join)


class _EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tokens, words, positions, gamma, beta, padding_idx, eps, p, dtype, defer_join=False):
        tokens = tokens.contiguous()
        seed = next_dropout_seed()
        out, mean, rstd = ops.embedding_fwd(tokens, words.detach(), positions.detach(), gamma.detach(),
                                            beta.detach(), dtype, padding_idx, eps, p, seed)
        ctx.save_for_backward(tokens, words, positions, gamma, mean, rstd)
        ctx.cfg = (padding_idx, p, seed)
        ctx.owners = (words, positions, gamma, beta)
        ctx.defer_join = defer_join
        return out

    @staticmethod
    @traced_backward
    def backward(ctx, dout):
        tokens, words, positions, gamma, mean, rstd = ctx.saved_tensors
        padding_idx, p, seed = ctx.cfg
        # every kernel accumulates (+=): write straight into the parameters' gradient buffers when they have
        # one (also for the tied word matrix, which the output projection accumulates into as well)
        bufs, rets = [], []
        for owner in ctx.owners:
            t = gradsink.target(owner)
            if t is None:
                t = torch.zeros_like(owner, dtype=torch.float32)
                rets.append(t)
            else:
                rets.append(None)
            bufs.append(t)
        dev = dout.device
        dout = dout.contiguous()
        # Nothing on the compute stream reads these gradients.  When the word matrix accumulates in place, ALL of
        # its writers (this scatter and the tied output projection's weight gradient) use the one side stream, so
        # they stay ordered among themselves.
        on_side = rets[0] is None
        if on_side:
            with wgrad_stream(dev, dout, mean, rstd):
                ops.embedding_bwd(tokens, words.detach(), positions.detach(), gamma.detach(), mean, rstd,
                                  dout, bufs[0], bufs[1], bufs[2], bufs[3], padding_idx, p, seed)
        else:
            ops.embedding_bwd(tokens, words.detach(), positions.detach(), gamma.detach(), mean, rstd,
                              dout, bufs[0], bufs[1], bufs[2], bufs[3], padding_idx, p, seed)
        # The embedding is the first node of a head's forward, hence the last of its backward: unless a later
        # node is known to join the side stream (the backbone's backward), do it here; and always when a freshly
        # allocated gradient is handed back to autograd, which will read it on the compute stream.
        if not ctx.defer_join or (on_side and any(r is not None for r in rets)):
            wgrad_stream.join(dev)
        return None, rets[0], rets[1], rets[2], rets[3], None, None, None, None, None


class TextualHead(nn.Module):
    """Base class (reference: virtex/modules/textual_heads.py:15-43)."""

    def __init__(self, visual_feature_size: int, vocab_size: int, hidden_size: int):
        super().__init__()
        self.visual_feature_size = visual_feature_size
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size

    @property
    def textual_feature_size(self):
        return self.hidden_size


class TransformerDecoderTextualHead(TextualHead):
    def __init__(self, visual_feature_size: int, vocab_size: int, hidden_size: int, num_layers: int,
                 attention_heads: int, feedforward_size: int, dropout: float = 0.1,
                 norm_first: bool = False, mask_future_positions: bool = True,
                 max_caption_length: int = 30, padding_idx: int = 0,
                 compute_dtype: torch.dtype = torch.bfloat16):
        super().__init__(visual_feature_size, vocab_size, hidden_size)
        self.norm_first = bool(norm_first)
        if hidden_size % attention_heads != 0 or hidden_size // attention_heads != 64:
            raise ValueError("the fused attention kernel needs head_dim == 64 (H/A)")
        self.num_layers, self.attention_heads = num_layers, attention_heads
        self.feedforward_size, self.dropout = feedforward_size, dropout
        self.mask_future_positions, self.padding_idx = mask_future_positions, padding_idx
        self.compute_dtype = compute_dtype

        self.visual_projection = nn.Linear(visual_feature_size, self.textual_feature_size)
        self.embedding = WordAndPositionalEmbedding(vocab_size, self.textual_feature_size, dropout=dropout,
                                                    max_caption_length=max_caption_length,
                                                    padding_idx=padding_idx)
        self.embedding.compute_dtype = compute_dtype
        self.transformer = nn.TransformerDecoder(
            nn.TransformerDecoderLayer(self.textual_feature_size, attention_heads,
                                       dim_feedforward=feedforward_size, dropout=dropout, activation="gelu",
                                       batch_first=True, norm_first=self.norm_first),
            # a final LayerNorm closes a pre-norm stack (reference: textual_heads.py:192-193)
            num_layers=num_layers, norm=nn.LayerNorm(self.hidden_size) if self.norm_first else None)
        self.apply(self._init_weights)
        self.output = nn.Linear(self.textual_feature_size, vocab_size)
        self.output.weight = self.embedding.words.weight

    @staticmethod
    def _init_weights(module):
        # N(0, 0.02) for Linear / MHA / Embedding weights; biases untouched
        # (reference: textual_heads.py:202-214)
        if isinstance(module, nn.Linear):
            module.weight.data.normal_(mean=0.0, std=0.02)
        elif isinstance(module, nn.MultiheadAttention):
            module.in_proj_weight.data.normal_(mean=0.0, std=0.02)
            module.out_proj.weight.data.normal_(mean=0.0, std=0.02)
        elif isinstance(module, nn.Embedding):
            module.weight.data.normal_(mean=0.0, std=0.02)
            if module.padding_idx is not None:
                module.weight.data[module.padding_idx].zero_()

    # -- parameter order shared by forward/backward -------------------------------------
    def _layer_params(self, layer):
        return [layer.self_attn.in_proj_weight, layer.self_attn.in_proj_bias,
                layer.self_attn.out_proj.weight, layer.self_attn.out_proj.bias,
                layer.norm1.weight, layer.norm1.bias,
                layer.multihead_attn.in_proj_weight, layer.multihead_attn.in_proj_bias,
                layer.multihead_attn.out_proj.weight, layer.multihead_attn.out_proj.bias,
                layer.norm2.weight, layer.norm2.bias,
                layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias,
                layer.norm3.weight, layer.norm3.bias]

    def project_visual_features(self, visual_features):
        """The projected visual grid (B*S, H) of `visual_projection` as its own autograd node, so that a model whose
        heads SHARE the projection (CaptioningModel ties it, captioning.py:57-63) evaluates it -- forward, input
        gradient and weight gradient -- once per step instead of once per head (legal because the
        gradients add linearly; the reference's order, one evaluation per head, stays available as the faithful
        mode: `features(...)` without `memory`)."""
        return _VisualProjectionFn.apply(visual_features, self.visual_projection.weight, self.visual_projection.bias,
                                         self.compute_dtype)

    def features(self, visual_features, caption_tokens, caption_lengths, memory=None):
        """(B,C,h,w) visual grid + (B,T) tokens -> decoder hidden states (B,T,H) in compute dtype.
        memory: the output of `project_visual_features` when the caller shares it between heads."""
        emb = self.embedding
        emb.compute_dtype = self.compute_dtype
        x0 = emb(caption_tokens)
        params = [self.visual_projection.weight, self.visual_projection.bias]
        for layer in self.transformer.layers:
            params += self._layer_params(layer)
        p = self.dropout if self.training else 0.0
        fn = _DecoderFn
        if self.norm_first:                       # "transdec_prenorm" (virtex/factories.py:358-366): the final LayerNorm's parameters last
            fn = _PreNormDecoderFn
            params += [self.transformer.norm.weight, self.transformer.norm.bias]
        if memory is not None:
            B, _, h, w = visual_features.shape
            return fn.apply(memory, x0, caption_lengths, self, p, (B, h * w), *params)
        return fn.apply(visual_features, x0, caption_lengths, self, p, None, *params)

    def forward(self, visual_features, caption_tokens, caption_lengths):
        """Returns (B,T,V) fp32 logits, like the reference (textual_heads.py:216-278)."""
        h = self.features(visual_features, caption_tokens, caption_lengths)
        return _OutputProjectionFn.apply(h, self.output.weight, self.output.bias)

    def weight_plan(self):
        """(parameter, channel padding, transposed copy wanted) for every GEMM weight of this head."""
        items = [(self.visual_projection.weight, None, True)]
        for layer in self.transformer.layers:
            for w in (layer.self_attn.in_proj_weight, layer.self_attn.out_proj.weight, layer.multihead_attn.in_proj_weight,
                      layer.multihead_attn.out_proj.weight, layer.linear1.weight, layer.linear2.weight):
                items.append((w, None, True))
        items.append((self.output.weight, None, True))
        return items

    @staticmethod
    def make_future_mask(size, dtype, device):
        """Kept for API parity (reference :280-292); the fused attention kernel applies the
        causal mask arithmetically and never materialises it."""
        return torch.triu(torch.full((size, size), float("-inf"), dtype=dtype, device=device), diagonal=1)


def _nhwc_rows(visual_features, dtype):
    """(B,C,h,w) logical -> ([B*h*w, C] row-major view, B, S).  Zero-copy for NHWC-physical input."""
    B, C, h, w = visual_features.shape
    m = visual_features.permute(0, 2, 3, 1)
    if m.dtype != dtype or not m.is_contiguous():
        m = m.to(dtype).contiguous()
    return m.view(B * h * w, C), B, h * w


class _VisualProjectionFn(torch.autograd.Function):
    """mem = visual_grid @ Wv^T + bv as (B*S, H) rows (reference: textual_heads.py:240-245), shared by both heads."""

    @staticmethod
    def forward(ctx, visual_features, weight, bias, dt):
        mem_in, B, S = _nhwc_rows(visual_features, dt)
        Wv, _ = ops.prepped(weight, dt)
        mem = ops.gemm_nt(mem_in, Wv.view(weight.shape), bias=bias.detach())
        ctx.mem_in, ctx.owner, ctx.dt = mem_in, (weight, bias), dt
        ctx.vshape, ctx.needs_vis_grad = visual_features.shape, visual_features.requires_grad
        return mem

    @staticmethod
    @traced_backward
    def backward(ctx, dmem):
        weight, bias = ctx.owner
        dt = ctx.dt
        if dmem.dtype != dt or not dmem.is_contiguous():
            dmem = dmem.to(dt).contiguous()
        dW, rW = _sink_or_zeros(weight)
        db, rb = _sink_or_zeros(bias)
        if rW is None and rb is None:
            with wgrad_stream(dmem.device, dmem, ctx.mem_in):
                ops.gemm_tn_acc(dmem, ctx.mem_in, dW)
                ops.colsum_acc(dmem, db)
        else:
            ops.gemm_tn_acc(dmem, ctx.mem_in, dW)
            ops.colsum_acc(dmem, db)
        dvis = None
        if ctx.needs_vis_grad:
            _, Wv_t = ops.prepped(weight, dt)
            Bv, C, h, w = ctx.vshape
            dvis = ops.gemm_nt(dmem, Wv_t.view(C, -1)).view(Bv, h, w, C).permute(0, 3, 1, 2)
        return dvis, rW, rb, None


class _DecoderFn(torch.autograd.Function):
    """visual_projection + L post-norm decoder layers (reference: textual_heads.py:240-275 and
    torch/nn/modules/transformer.py:1143-1199), forward and hand-written backward."""

    @staticmethod
    def forward(ctx, visual_features, x0, lengths, head, p, shared_memory, *params):
        """shared_memory = (B, S): `visual_features` already IS the projected memory (B*S, H) of
        `_VisualProjectionFn` (the projection's own parameters are then not touched here)."""
        dt = head.compute_dtype
        A = head.attention_heads
        H = head.hidden_size
        ctx.shared = shared_memory is not None
        if ctx.shared:
            B, S = shared_memory
            mem, mem_in, Wv_t = visual_features, None, None
            if mem.dtype != dt or not mem.is_contiguous():
                mem = mem.to(dt).contiguous()
        else:
            mem_in, B, S = _nhwc_rows(visual_features, dt)
            bv = params[1].detach()
            Wv, Wv_t = ops.prepped(params[0], dt)
            Wv, Wv_t = Wv.view(H, -1), Wv_t.view(-1, H)
            mem = ops.gemm_nt(mem_in, Wv, bias=bv)                   # (B*S, H)
        T = x0.shape[1]
        x = x0.reshape(B * T, H)
        lengths = lengths.contiguous()
        saved_layers = []
        for li in range(head.num_layers):
            raw = params[2 + 18 * li: 2 + 18 * (li + 1)]
            P = [t.detach() for t in raw]
            (Win, bin_, Wo, bo, g1, b1, Win2, bin2, Wo2, bo2, g2, b2, W1, bf1, W2, bf2, g3, b3) = P
            seeds = [next_dropout_seed() for _ in range(7)]
            cw = {}
            for name, idx in (("Win", 0), ("Wo", 2), ("Win2", 6), ("Wo2", 8), ("W1", 12), ("W2", 14)):
                w32 = P[idx]
                w, wt = ops.prepped(raw[idx], dt)
                cw[name] = (w.view(w32.shape), wt.view(w32.shape[1], w32.shape[0]))
            # ---- masked self-attention
            qkv = ops.gemm_nt(x, cw["Win"][0], bias=bin_)                           # (B*T, 3H)
            o1 = ops.attention_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, A, T, T,
                                   head.mask_future_positions, lengths, p, seeds[0])
            y1 = ops.gemm_nt(o1, cw["Wo"][0], bias=bo)
            x1, m1, r1 = ops.layernorm_residual_fwd(x, y1, g1, b1, 1e-5, p, seeds[1])
            # ---- cross-attention over the visual grid
            q2 = ops.gemm_nt(x1, cw["Win2"][0][:H], bias=bin2[:H])                  # (B*T, H)
            kv2 = ops.gemm_nt(mem, cw["Win2"][0][H:], bias=bin2[H:])                # (B*S, 2H)
            o2 = ops.attention_fwd(q2, kv2[:, :H], kv2[:, H:], B, A, T, S, False, None, p, seeds[2])
            y2 = ops.gemm_nt(o2, cw["Wo2"][0], bias=bo2)
            x2, m2, r2 = ops.layernorm_residual_fwd(x1, y2, g2, b2, 1e-5, p, seeds[3])
            # ---- feed-forward
            a, hpre = ops.gemm_nt(x2, cw["W1"][0], bias=bf1, act=ops.ACT_GELU, want_preact=True,
                                  p_drop=p, seed=seeds[4])
            y3 = ops.gemm_nt(a, cw["W2"][0], bias=bf2)
            x3, m3, r3 = ops.layernorm_residual_fwd(x2, y3, g3, b3, 1e-5, p, seeds[5])
            saved_layers.append(dict(x=x, qkv=qkv, o1=o1, y1=y1, m1=m1, r1=r1, x1=x1, q2=q2, kv2=kv2, o2=o2,
                                     y2=y2, m2=m2, r2=r2, x2=x2, a=a, hpre=hpre, y3=y3, m3=m3, r3=r3,
                                     seeds=seeds, cw=cw, P=P))
            x = x3
        ctx.head, ctx.p, ctx.dims = head, p, (B, T, S, H, A)
        ctx.layer_params = [list(params[2 + 18 * li: 2 + 18 * (li + 1)]) for li in range(head.num_layers)]
        ctx.saved_layers, ctx.mem, ctx.mem_in, ctx.Wv_t, ctx.lengths = saved_layers, mem, mem_in, Wv_t, lengths
        ctx.vis_owner = (params[0], params[1])
        ctx.vshape = visual_features.shape
        ctx.needs_vis_grad = visual_features.requires_grad
        return x.view(B, T, H)

    @staticmethod
    @traced_backward
    def backward(ctx, dhid):
        head, p = ctx.head, ctx.p
        B, T, S, H, A = ctx.dims
        dt = head.compute_dtype
        dev = dhid.device
        dx = dhid.reshape(B * T, H)
        if dx.dtype != dt or not dx.is_contiguous():
            dx = dx.to(dt).contiguous()
        dmem = None
        pgrads = []

        def zeros(*shape):
            return torch.zeros(*shape, dtype=torch.float32, device=dev)

        def sink(p):
            """(buffer to accumulate into, value to return to autograd)"""
            t = gradsink.target(p)
            if t is not None:
                return t, None
            z = torch.zeros_like(p, dtype=torch.float32)
            return z, z

        def linear_grads(inp, dout, wp, bp):
            """dW ([out,in]) and dbias of y = inp @ W^T + b, accumulated into sinks.  With direct sinks (the
            parameters own gradient buffers) both kernels go to the weight-gradient side stream: nothing on the
            compute stream depends on them (virtex_amd/streams.py)."""
            dW, rW = sink(wp)
            db, rb = sink(bp)
            if rW is None and rb is None:
                with wgrad_stream(dev, inp, dout):
                    ops.gemm_tn_acc(dout, inp, dW)
                    ops.colsum_acc(dout, db)
            else:
                ops.gemm_tn_acc(dout, inp, dW)
                ops.colsum_acc(dout, db)
            return rW, rb

        for li in reversed(range(head.num_layers)):
            L = ctx.saved_layers[li]
            (Win, bin_, Wo, bo, g1, b1, Win2, bin2, Wo2, bo2, g2, b2, W1, bf1, W2, bf2, g3, b3) = L["P"]
            PP = ctx.layer_params[li]     # the nn.Parameters, same order
            cw, seeds = L["cw"], L["seeds"]
            # the split-K reductions of this layer's seven weight gradients in one launch (ops.splitk_batch: measured slower, off by
            # default), issued on the weight-gradient stream at the end of the layer -- before autograd's hooks announce the gradients
            sk = ops.splitk_batch(dev).begin()
            # ---- FFN block: x3 = LN3(x2 + drop(y3))
            (dg3, rdg3), (db3, rdb3) = sink(PP[16]), sink(PP[17])
            dz3, dy3 = ops.layernorm_residual_bwd(L["x2"], L["y3"], g3, L["m3"], L["r3"], dx, dg3, db3, p, seeds[5])
            dW2, dbf2 = linear_grads(L["a"], dy3, PP[14], PP[15])
            da = ops.gemm_nt(dy3, cw["W2"][1])                                  # (B*T, F)
            dh = ops.gelu_bwd(L["hpre"], da, p, seeds[4])
            dW1, dbf1 = linear_grads(L["x2"], dh, PP[12], PP[13])
            dx2 = ops.gemm_nt(dh, cw["W1"][1], residual=dz3)                    # + residual path
            # ---- cross-attention block: x2 = LN2(x1 + drop(y2))
            (dg2, rdg2), (db2, rdb2) = sink(PP[10]), sink(PP[11])
            dz2, dy2 = ops.layernorm_residual_bwd(L["x1"], L["y2"], g2, L["m2"], L["r2"], dx2, dg2, db2, p, seeds[3])
            dWo2, dbo2 = linear_grads(L["o2"], dy2, PP[8], PP[9])
            do2 = ops.gemm_nt(dy2, cw["Wo2"][1])
            dq2 = torch.empty_like(L["q2"])
            dkv2 = torch.empty_like(L["kv2"])
            ops.attention_bwd(L["q2"], L["kv2"][:, :H], L["kv2"][:, H:], do2, dq2, dkv2[:, :H], dkv2[:, H:],
                              B, A, T, S, False, None, p, seeds[2])
            dWin2, rWin2 = sink(PP[6])
            dbin2, rbin2 = sink(PP[7])

            def cross_in_proj_grads():
                ops.gemm_tn_acc(dq2, L["x1"], dWin2[:H])
                ops.gemm_tn_acc(dkv2, ctx.mem, dWin2[H:])
                ops.colsum_acc(dq2, dbin2[:H])
                ops.colsum_acc(dkv2, dbin2[H:])
            if rWin2 is None and rbin2 is None:
                with wgrad_stream(dev, dq2, dkv2, L["x1"], ctx.mem):
                    cross_in_proj_grads()
            else:
                cross_in_proj_grads()
            dx1 = ops.gemm_nt(dq2, _wt_cols(cw["Win2"][1], 0, H), residual=dz2)
            dmem = ops.gemm_nt(dkv2, _wt_cols(cw["Win2"][1], H, 3 * H), residual=dmem)
            # ---- self-attention block: x1 = LN1(x + drop(y1))
            (dg1, rdg1), (db1, rdb1) = sink(PP[4]), sink(PP[5])
            dz1, dy1 = ops.layernorm_residual_bwd(L["x"], L["y1"], g1, L["m1"], L["r1"], dx1, dg1, db1, p, seeds[1])
            dWo, dbo = linear_grads(L["o1"], dy1, PP[2], PP[3])
            do1 = ops.gemm_nt(dy1, cw["Wo"][1])
            qkv = L["qkv"]
            dqkv = torch.empty_like(qkv)
            ops.attention_bwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], do1, dqkv[:, :H], dqkv[:, H:2 * H],
                              dqkv[:, 2 * H:], B, A, T, T, head.mask_future_positions, ctx.lengths, p, seeds[0])
            dWin, dbin = linear_grads(L["x"], dqkv, PP[0], PP[1])
            dx = ops.gemm_nt(dqkv, cw["Win"][1], residual=dz1)
            with wgrad_stream(dev):
                sk.end()
            pgrads = [dWin, dbin, dWo, dbo, rdg1, rdb1, rWin2, rbin2, dWo2, dbo2, rdg2, rdb2, dW1, dbf1, dW2, dbf2,
                      rdg3, rdb3] + pgrads
        if ctx.shared:          # the projection is its own node: hand the memory's gradient back to it
            return (dmem, dx.view(B, T, H), None, None, None, None, None, None, *pgrads)
        # ---- visual projection
        dWv, rWv = sink(ctx.vis_owner[0])
        dbv, rbv = sink(ctx.vis_owner[1])
        if rWv is None and rbv is None:          # shared by both heads: both go through the one side stream, in order
            with wgrad_stream(dev, dmem, ctx.mem_in):
                ops.gemm_tn_acc(dmem, ctx.mem_in, dWv)
                ops.colsum_acc(dmem, dbv)
        else:
            ops.gemm_tn_acc(dmem, ctx.mem_in, dWv)
            ops.colsum_acc(dmem, dbv)
        dvis = None
        if ctx.needs_vis_grad:
            Bv, C, h, w = ctx.vshape
            dvis = ops.gemm_nt(dmem, ctx.Wv_t).view(Bv, h, w, C).permute(0, 3, 1, 2)
        return (dvis, dx.view(B, T, H), None, None, None, None, rWv, rbv, *pgrads)


class _PreNormDecoderFn(torch.autograd.Function):
    """visual_projection + L PRE-norm decoder layers + the final LayerNorm (reference: textual_heads.py:181-194 with
    norm_first=True; torch/nn/modules/transformer.py: x = x + sa(norm1(x)); x = x + mha(norm2(x), mem); x = x + ff(norm3(x))),
    on the kernels of the post-norm path: plain LayerNorm = vtx_layernorm_residual_fwd without a sub-layer operand, the
    residual join x + dropout(y) = the epilogue of the GEMM that produces y, its backward mask = vtx_dropout_bwd."""

    @staticmethod
    def forward(ctx, visual_features, x0, lengths, head, p, shared_memory, *params):
        dt = head.compute_dtype
        A, H = head.attention_heads, head.hidden_size
        ctx.shared = shared_memory is not None
        if ctx.shared:
            B, S = shared_memory
            mem, mem_in, Wv_t = visual_features, None, None
            if mem.dtype != dt or not mem.is_contiguous():
                mem = mem.to(dt).contiguous()
        else:
            mem_in, B, S = _nhwc_rows(visual_features, dt)
            Wv, Wv_t = ops.prepped(params[0], dt)
            Wv, Wv_t = Wv.view(H, -1), Wv_t.view(-1, H)
            mem = ops.gemm_nt(mem_in, Wv, bias=params[1].detach())
        T = x0.shape[1]
        x = x0.reshape(B * T, H)
        if x.dtype != dt or not x.is_contiguous():
            x = x.to(dt).contiguous()
        lengths = lengths.contiguous()
        saved_layers = []
        for li in range(head.num_layers):
            raw = params[2 + 18 * li: 2 + 18 * (li + 1)]
            P = [t.detach() for t in raw]
            (Win, bin_, Wo, bo, g1, b1, Win2, bin2, Wo2, bo2, g2, b2, W1, bf1, W2, bf2, g3, b3) = P
            seeds = [next_dropout_seed() for _ in range(7)]
            cw = {}
            for name, idx in (("Win", 0), ("Wo", 2), ("Win2", 6), ("Wo2", 8), ("W1", 12), ("W2", 14)):
                w, wt = ops.prepped(raw[idx], dt)
                cw[name] = (w.view(P[idx].shape), wt.view(P[idx].shape[1], P[idx].shape[0]))
            # ---- masked self-attention on norm1(x), joined to x
            h1, m1, r1 = ops.layernorm_residual_fwd(x, None, g1, b1, 1e-5)
            qkv = ops.gemm_nt(h1, cw["Win"][0], bias=bin_)
            o1 = ops.attention_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, A, T, T,
                                   head.mask_future_positions, lengths, p, seeds[0])
            x1 = ops.gemm_nt(o1, cw["Wo"][0], bias=bo, residual=x, p_drop=p, seed=seeds[1])
            # ---- cross-attention over the visual grid on norm2(x1)
            h2, m2, r2 = ops.layernorm_residual_fwd(x1, None, g2, b2, 1e-5)
            q2 = ops.gemm_nt(h2, cw["Win2"][0][:H], bias=bin2[:H])
            kv2 = ops.gemm_nt(mem, cw["Win2"][0][H:], bias=bin2[H:])
            o2 = ops.attention_fwd(q2, kv2[:, :H], kv2[:, H:], B, A, T, S, False, None, p, seeds[2])
            x2 = ops.gemm_nt(o2, cw["Wo2"][0], bias=bo2, residual=x1, p_drop=p, seed=seeds[3])
            # ---- feed-forward on norm3(x2)
            h3, m3, r3 = ops.layernorm_residual_fwd(x2, None, g3, b3, 1e-5)
            a, hpre = ops.gemm_nt(h3, cw["W1"][0], bias=bf1, act=ops.ACT_GELU, want_preact=True, p_drop=p, seed=seeds[4])
            x3 = ops.gemm_nt(a, cw["W2"][0], bias=bf2, residual=x2, p_drop=p, seed=seeds[5])
            saved_layers.append(dict(x=x, h1=h1, m1=m1, r1=r1, qkv=qkv, o1=o1, x1=x1, h2=h2, m2=m2, r2=r2, q2=q2, kv2=kv2,
                                     o2=o2, x2=x2, h3=h3, m3=m3, r3=r3, a=a, hpre=hpre, seeds=seeds, cw=cw, P=P))
            x = x3
        gf, bf = params[-2].detach(), params[-1].detach()
        out, mf, rf = ops.layernorm_residual_fwd(x, None, gf, bf, 1e-5)
        ctx.head, ctx.p, ctx.dims = head, p, (B, T, S, H, A)
        ctx.layer_params = [list(params[2 + 18 * li: 2 + 18 * (li + 1)]) for li in range(head.num_layers)]
        ctx.final = (x, mf, rf, gf, params[-2], params[-1])
        ctx.saved_layers, ctx.mem, ctx.mem_in, ctx.Wv_t, ctx.lengths = saved_layers, mem, mem_in, Wv_t, lengths
        ctx.vis_owner = (params[0], params[1])
        ctx.vshape = visual_features.shape
        ctx.needs_vis_grad = visual_features.requires_grad
        return out.view(B, T, H)

    @staticmethod
    @traced_backward
    def backward(ctx, dhid):
        head, p = ctx.head, ctx.p
        B, T, S, H, A = ctx.dims
        dt = head.compute_dtype
        dev = dhid.device
        dout = dhid.reshape(B * T, H)
        if dout.dtype != dt or not dout.is_contiguous():
            dout = dout.to(dt).contiguous()

        def linear_grads(inp, dy, wp, bp):
            dW, rW = _sink_or_zeros(wp)
            db, rb = _sink_or_zeros(bp)
            if rW is None and rb is None:
                with wgrad_stream(dev, inp, dy):
                    ops.gemm_tn_acc(dy, inp, dW)
                    ops.colsum_acc(dy, db)
            else:
                ops.gemm_tn_acc(dy, inp, dW)
                ops.colsum_acc(dy, db)
            return rW, rb

        def norm_back(x_in, gamma, mean, rstd, dh, gp, bp):
            """gradient of h = LayerNorm(x_in) wrt x_in; the norm's own parameter gradients accumulated"""
            dg, rdg = _sink_or_zeros(gp)
            db, rdb = _sink_or_zeros(bp)
            dz, _ = ops.layernorm_residual_bwd(x_in, None, gamma, mean, rstd, dh, dg, db)
            return dz, rdg, rdb

        xL, mf, rf, gf, gfp, bfp = ctx.final
        dx, rdgf, rdbf = norm_back(xL, gf, mf, rf, dout, gfp, bfp)
        dmem = None
        pgrads = []
        for li in reversed(range(head.num_layers)):
            L = ctx.saved_layers[li]
            (Win, bin_, Wo, bo, g1, b1, Win2, bin2, Wo2, bo2, g2, b2, W1, bf1, W2, bf2, g3, b3) = L["P"]
            PP = ctx.layer_params[li]
            cw, seeds = L["cw"], L["seeds"]
            # ---- x3 = x2 + dropout(linear2(dropout(gelu(linear1(norm3(x2))))))
            dy3 = ops.dropout_bwd(dx, p, seeds[5])
            dW2, dbf2 = linear_grads(L["a"], dy3, PP[14], PP[15])
            da = ops.gemm_nt(dy3, cw["W2"][1])
            dh = ops.gelu_bwd(L["hpre"], da, p, seeds[4])
            dW1, dbf1 = linear_grads(L["h3"], dh, PP[12], PP[13])
            dh3 = ops.gemm_nt(dh, cw["W1"][1])
            dz3, rdg3, rdb3 = norm_back(L["x2"], g3, L["m3"], L["r3"], dh3, PP[16], PP[17])
            dx2 = ops.add(dx, dz3)
            # ---- x2 = x1 + dropout(out_proj(cross-attention(norm2(x1), mem)))
            dy2 = ops.dropout_bwd(dx2, p, seeds[3])
            dWo2, dbo2 = linear_grads(L["o2"], dy2, PP[8], PP[9])
            do2 = ops.gemm_nt(dy2, cw["Wo2"][1])
            dq2 = torch.empty_like(L["q2"])
            dkv2 = torch.empty_like(L["kv2"])
            ops.attention_bwd(L["q2"], L["kv2"][:, :H], L["kv2"][:, H:], do2, dq2, dkv2[:, :H], dkv2[:, H:],
                              B, A, T, S, False, None, p, seeds[2])
            dWin2, rWin2 = _sink_or_zeros(PP[6])
            dbin2, rbin2 = _sink_or_zeros(PP[7])

            def cross_in_proj_grads():
                ops.gemm_tn_acc(dq2, L["h2"], dWin2[:H])
                ops.gemm_tn_acc(dkv2, ctx.mem, dWin2[H:])
                ops.colsum_acc(dq2, dbin2[:H])
                ops.colsum_acc(dkv2, dbin2[H:])
            if rWin2 is None and rbin2 is None:
                with wgrad_stream(dev, dq2, dkv2, L["h2"], ctx.mem):
                    cross_in_proj_grads()
            else:
                cross_in_proj_grads()
            dh2 = ops.gemm_nt(dq2, _wt_cols(cw["Win2"][1], 0, H))
            dmem = ops.gemm_nt(dkv2, _wt_cols(cw["Win2"][1], H, 3 * H), residual=dmem)
            dz2, rdg2, rdb2 = norm_back(L["x1"], g2, L["m2"], L["r2"], dh2, PP[10], PP[11])
            dx1 = ops.add(dx2, dz2)
            # ---- x1 = x + dropout(out_proj(self-attention(norm1(x))))
            dy1 = ops.dropout_bwd(dx1, p, seeds[1])
            dWo, dbo = linear_grads(L["o1"], dy1, PP[2], PP[3])
            do1 = ops.gemm_nt(dy1, cw["Wo"][1])
            qkv = L["qkv"]
            dqkv = torch.empty_like(qkv)
            ops.attention_bwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], do1, dqkv[:, :H], dqkv[:, H:2 * H],
                              dqkv[:, 2 * H:], B, A, T, T, head.mask_future_positions, ctx.lengths, p, seeds[0])
            dWin, dbin = linear_grads(L["h1"], dqkv, PP[0], PP[1])
            dh1 = ops.gemm_nt(dqkv, cw["Win"][1])
            dz1, rdg1, rdb1 = norm_back(L["x"], g1, L["m1"], L["r1"], dh1, PP[4], PP[5])
            dx = ops.add(dx1, dz1)
            pgrads = [dWin, dbin, dWo, dbo, rdg1, rdb1, rWin2, rbin2, dWo2, dbo2, rdg2, rdb2, dW1, dbf1, dW2, dbf2,
                      rdg3, rdb3] + pgrads
        pgrads += [rdgf, rdbf]
        if ctx.shared:
            return (dmem, dx.view(B, T, H), None, None, None, None, None, None, *pgrads)
        dWv, rWv = _sink_or_zeros(ctx.vis_owner[0])
        dbv, rbv = _sink_or_zeros(ctx.vis_owner[1])
        if rWv is None and rbv is None:
            with wgrad_stream(dev, dmem, ctx.mem_in):
                ops.gemm_tn_acc(dmem, ctx.mem_in, dWv)
                ops.colsum_acc(dmem, dbv)
        else:
            ops.gemm_tn_acc(dmem, ctx.mem_in, dWv)
            ops.colsum_acc(dmem, dbv)
        dvis = None
        if ctx.needs_vis_grad:
            Bv, C, h, w = ctx.vshape
            dvis = ops.gemm_nt(dmem, ctx.Wv_t).view(Bv, h, w, C).permute(0, 3, 1, 2)
        return (dvis, dx.view(B, T, H), None, None, None, None, rWv, rbv, *pgrads)


def _sink_or_zeros(p):
    """(fp32 buffer the gradient kernels accumulate into, value to hand back to autograd): the parameter's own
    gradient buffer and None when it has a suitable one (virtex_amd/gradsink.py), else a fresh zero tensor twice."""
    t = gradsink.target(p)
    if t is not None:
        return t, None
    z = torch.zeros_like(p, dtype=torch.float32)
    return z, z


def tied_projection_grads(d, h2, weight_param, bias_param):
    """Weight / bias gradient of logits = h2 @ W^T + b for the TIED matrix.  In-place accumulation into the word
    matrix' gradient buffer happens on the weight-gradient side stream, like the embedding scatter into the same
    buffer (one stream => ordered).  Returns what autograd should get (None where accumulated in place)."""
    dW, rW = _sink_or_zeros(weight_param)
    db, rb = _sink_or_zeros(bias_param)
    if rW is None:
        with wgrad_stream(d.device, d, h2):
            ops.gemm_tn_acc(d, h2, dW)
            ops.colsum_acc(d, db)
        if rb is not None:
            wgrad_stream.join(d.device)          # a fresh bias-gradient tensor goes back to autograd
    else:
        ops.gemm_tn_acc(d, h2, dW)
        ops.colsum_acc(d, db)
    return rW, rb


def _wt_cols(wt, c0, c1):
    """Columns [c0, c1) of a transposed weight wt = W^T (in_features, out_features) as a strided
    (in_features, c1-c0) view: the B operand of the input-gradient GEMM of the sub-projection
    W[c0:c1] (rows = in_features, k = the selected output slice, row stride = out_features)."""
    return wt[:, c0:c1]


class _OutputProjectionFn(torch.autograd.Function):
    """logits = hidden @ words^T + bias, fp32 (reference: textual_heads.py:199-200,277)."""

    @staticmethod
    def forward(ctx, hidden, weight, bias):
        B, T, H = hidden.shape
        dt = hidden.dtype
        w, _ = ops.prepped(weight, dt, want_wt=False)
        logits = ops.gemm_nt(hidden.reshape(B * T, H), w.view(weight.shape), bias=bias.detach(), out_f32=True)
        ctx.save_for_backward(hidden, weight)
        ctx.weight_param, ctx.bias_param = weight, bias
        return logits.view(B, T, -1)

    @staticmethod
    @traced_backward
    def backward(ctx, dlogits):
        hidden, weight = ctx.saved_tensors
        B, T, H = hidden.shape
        dt = hidden.dtype
        V = weight.shape[0]
        d = dlogits.reshape(B * T, V)
        d = d.contiguous() if d.dtype == dt else d.to(dt).contiguous()
        _, wt = ops.prepped(ctx.weight_param, dt, want_w=False)
        dh = ops.gemm_nt(d, wt.view(H, V)).view(B, T, H)
        rW, rb = tied_projection_grads(d, hidden.reshape(B * T, H), ctx.weight_param, ctx.bias_param)
        return dh, rW, rb
