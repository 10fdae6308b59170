"""Torch-CPU restatement of the reference bicaptioning model (TEST INFRASTRUCTURE).

Every class cites the reference lines it follows (paths relative to /root/reference).
Sub-module attribute names are the reference's, so ``state_dict()`` keys match the
reference key for key (370 keys / 202 unique parameter tensors for R_50_L1_H1024).

The arithmetic is delegated to the same ``torch.nn`` primitives the reference uses
(Conv2d, BatchNorm2d, TransformerDecoder, Embedding, LayerNorm, CrossEntropyLoss), so on
the same torch build this port agrees with the verbatim reference bit for bit; that is
asserted by ``tests/test_oracle.py`` whenever /root/reference is importable.
"""
import copy
import re
from typing import Dict, List, Optional

import torch
from torch import nn

# ---------------------------------------------------------------------------------
# ResNet v1.5 (torchvision graph; spec: SURVEY.md Appendix A.1).
# Reference call site: virtex/modules/visual_backbones.py:43-47 builds
# torchvision.models.<name>(pretrained, zero_init_residual=True) and sets fc=Identity;
# forward (:68-74) walks named_children() and returns right after "layer4".
# ---------------------------------------------------------------------------------
RESNET_SPECS = {
    # name: (blocks per stage, base_width)
    "resnet50": ((3, 4, 6, 3), 64),
    "resnet101": ((3, 4, 23, 3), 64),
    "wide_resnet50_2": ((3, 4, 6, 3), 128),
}


class Bottleneck(nn.Module):
    """1x1 -> 3x3 (carries the stride: 'v1.5') -> 1x1 (x4 channels) + identity."""

    def __init__(self, cin: int, planes: int, stride: int, base_width: int, project: bool):
        super().__init__()
        mid = planes * base_width // 64
        cout = planes * 4
        self.conv1 = nn.Conv2d(cin, mid, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(mid)
        self.conv2 = nn.Conv2d(mid, mid, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(mid)
        self.conv3 = nn.Conv2d(mid, cout, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(
                nn.Conv2d(cin, cout, 1, stride=stride, bias=False), nn.BatchNorm2d(cout)
            )

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        y += skip
        return self.relu(y)


class ResNet(nn.Module):
    """Children in torchvision order: conv1 bn1 relu maxpool layer1..4 avgpool fc."""

    def __init__(self, name: str = "resnet50", zero_init_residual: bool = True):
        super().__init__()
        blocks, base_width = RESNET_SPECS[name]
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        cin = 64
        for stage, (planes, nblk) in enumerate(zip((64, 128, 256, 512), blocks)):
            layers = []
            for b in range(nblk):
                stride = 2 if (b == 0 and stage > 0) else 1
                layers.append(Bottleneck(cin, planes, stride, base_width, project=(b == 0)))
                cin = planes * 4
            setattr(self, f"layer{stage + 1}", nn.Sequential(*layers))
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(cin, 1000)
        self.out_channels = cin

        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.zeros_(m.bn3.weight)

    def forward(self, x):  # full torchvision forward (unused by the hot path)
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


class VisualBackbone(nn.Module):
    """virtex/modules/visual_backbones.py:20-74 (TorchvisionVisualBackbone)."""

    def __init__(self, name: str = "resnet50", visual_feature_size: int = 2048,
                 pretrained: bool = False, frozen: bool = False):
        super().__init__()
        assert not pretrained, "no network: ImageNet weights unavailable"
        self.visual_feature_size = visual_feature_size
        self.cnn = ResNet(name, zero_init_residual=True)
        self.cnn.fc = nn.Identity()
        if frozen:
            for p in self.cnn.parameters():
                p.requires_grad = False
            self.cnn.eval()

    def forward(self, image):
        c = self.cnn
        x = c.maxpool(c.relu(c.bn1(c.conv1(image))))
        return c.layer4(c.layer3(c.layer2(c.layer1(x))))  # (B, C, h, w)


# ---------------------------------------------------------------------------------
# Text side.
# ---------------------------------------------------------------------------------
class WordAndPositionalEmbedding(nn.Module):
    """virtex/modules/embedding.py:24-74: LN_{1e-8}(words[tok]+positions[t]) -> dropout
    -> zero rows whose token is padding_idx."""

    def __init__(self, vocab_size, hidden_size, dropout=0.0, max_caption_length=30, padding_idx=0):
        super().__init__()
        self.vocab_size, self.padding_idx = vocab_size, padding_idx
        self.words = nn.Embedding(vocab_size, hidden_size, padding_idx=padding_idx)
        self.positions = nn.Embedding(max_caption_length, hidden_size)
        self.layer_norm = nn.LayerNorm(hidden_size, eps=1e-8, elementwise_affine=True)
        self.dropout = nn.Dropout(p=dropout)

    def forward(self, tokens):
        t = torch.arange(tokens.size(1), dtype=tokens.dtype, device=tokens.device)
        x = self.words(tokens) + self.positions(t.unsqueeze(0).expand_as(tokens))
        x = self.dropout(self.layer_norm(x))
        return x * (tokens != self.padding_idx).unsqueeze(-1).type(x.dtype)


class TextualHead(nn.Module):
    """virtex/modules/textual_heads.py:146-278 (TransformerDecoderTextualHead)."""

    def __init__(self, visual_feature_size, vocab_size, hidden_size, num_layers,
                 attention_heads, feedforward_size, dropout=0.1, norm_first=False,
                 mask_future_positions=True, max_caption_length=30, padding_idx=0):
        super().__init__()
        self.visual_feature_size, self.vocab_size, self.hidden_size = (
            visual_feature_size, vocab_size, hidden_size)
        self.num_layers, self.attention_heads = num_layers, attention_heads
        self.feedforward_size, self.dropout = feedforward_size, dropout
        self.mask_future_positions, self.padding_idx = mask_future_positions, padding_idx

        self.visual_projection = nn.Linear(visual_feature_size, hidden_size)
        self.embedding = WordAndPositionalEmbedding(
            vocab_size, hidden_size, dropout=dropout,
            max_caption_length=max_caption_length, padding_idx=padding_idx)
        layer = nn.TransformerDecoderLayer(
            hidden_size, attention_heads, dim_feedforward=feedforward_size, dropout=dropout,
            activation="gelu", batch_first=True, norm_first=norm_first)
        self.transformer = nn.TransformerDecoder(
            layer, num_layers=num_layers,
            norm=nn.LayerNorm(hidden_size) if norm_first else None)
        # "BERT" init N(0, .02) of Linear / MHA / Embedding weights, biases untouched
        # (textual_heads.py:202-214); applied BEFORE `output` exists (:195-200).
        for m in self.modules():
            if isinstance(m, nn.Linear):
                m.weight.data.normal_(0.0, 0.02)
            elif isinstance(m, nn.MultiheadAttention):
                m.in_proj_weight.data.normal_(0.0, 0.02)
                m.out_proj.weight.data.normal_(0.0, 0.02)
            elif isinstance(m, nn.Embedding):
                m.weight.data.normal_(0.0, 0.02)
                if m.padding_idx is not None:
                    m.weight.data[m.padding_idx].zero_()
        self.output = nn.Linear(hidden_size, vocab_size)
        self.output.weight = self.embedding.words.weight  # tied (:199-200)

    @property
    def textual_feature_size(self):
        return self.hidden_size

    def forward(self, visual_features, caption_tokens, caption_lengths):
        b, c, h, w = visual_features.size()
        memory = self.visual_projection(visual_features.view(b, c, -1).permute(0, 2, 1))
        T = caption_tokens.size(1)
        steps = torch.ones_like(caption_tokens).cumsum(dim=1)
        pad_mask = caption_lengths.unsqueeze(1) < steps  # True == padding (:255-256)
        x = self.embedding(caption_tokens)
        causal = None
        if self.mask_future_positions:
            causal = torch.triu(
                torch.full((T, T), float("-inf"), dtype=x.dtype, device=x.device), diagonal=1)
        x = self.transformer(x, memory, tgt_mask=causal, tgt_key_padding_mask=pad_mask)
        return self.output(x)


class BicaptioningModel(nn.Module):
    """virtex/models/captioning.py:40-138 with caption_backward=True (:258-283)."""

    def __init__(self, visual: VisualBackbone, textual: TextualHead,
                 sos_index: int = 1, eos_index: int = 2, decoder=None):
        super().__init__()
        self.visual, self.textual = visual, textual
        self.padding_idx = textual.padding_idx
        self.caption_backward = True
        self.backward_textual = copy.deepcopy(textual)
        self.backward_textual.visual_projection = textual.visual_projection
        self.backward_textual.embedding = textual.embedding
        self.backward_textual.output = textual.output
        self.sos_index, self.eos_index, self.decoder = sos_index, eos_index, decoder
        self.loss = nn.CrossEntropyLoss(ignore_index=self.padding_idx)

    def forward(self, batch: Dict[str, torch.Tensor]):
        feats = self.visual(batch["image"])
        V = self.textual.vocab_size
        out = {}
        if "caption_tokens" not in batch:
            # inference branch (captioning.py:144-162): decode with the forward head
            if self.decoder is None:
                raise ValueError("Decoder for predicting captions is missing!")
            start = feats.new_full((feats.size(0),), self.sos_index).long()
            tokens, _ = self.decoder.search(start, lambda partial: self.decoding_step(feats, partial))
            return {"predictions": tokens}
        losses = {}
        for key, head, toks in (("captioning_forward", self.textual, batch["caption_tokens"]),
                                ("captioning_backward", self.backward_textual, batch["noitpac_tokens"])):
            logits = head(feats, toks, batch["caption_lengths"])
            losses[key] = self.loss(logits[:, :-1].contiguous().view(-1, V),
                                    toks[:, 1:].contiguous().view(-1))
            if key == "captioning_forward":
                out["logits"] = logits
            else:
                out["backward_logits"] = logits
        out["loss"] = losses["captioning_forward"] + losses["captioning_backward"]
        out["loss_components"] = {k: v.clone().detach() for k, v in losses.items()}
        if not self.training:
            out["predictions"] = torch.argmax(out["logits"], dim=-1)
        return out


    def decoding_step(self, feats, partial):
        """Next-token logits for (beam-expanded) prefixes: the whole prefix is re-run, lengths = current
        prefix length regardless of EOS/padding inside it (captioning.py:165-213)."""
        n, c, h, w = feats.size()
        beams = partial.size(0) // n
        if beams > 1:
            feats = feats.unsqueeze(1).expand(n, beams, c, h, w).reshape(n * beams, c, h, w)
        if partial.dim() == 1:
            lengths = torch.ones_like(partial)
            partial = partial.unsqueeze(1)
        else:
            lengths = torch.full((partial.size(0),), partial.size(1), dtype=partial.dtype, device=partial.device)
        return self.textual(feats, partial, lengths)[:, -1, :]


# ---------------------------------------------------------------------------------
# Builders / config restatement (virtex/config.py defaults).
# ---------------------------------------------------------------------------------
DEFAULTS = dict(vocab_size=10000, padding_idx=0, sos_index=1, eos_index=2,
                max_caption_length=30, dropout=0.1, visual_feature_size=2048)


def parse_textual_name(name: str):
    """'transdec_postnorm::L1_H1024_A16_F4096' -> dict (virtex/factories.py:384-392)."""
    kind, arch = name.split("::")
    m = re.match(r"L(\d+)_H(\d+)_A(\d+)_F(\d+)", arch)
    L, H, A, F = (int(g) for g in m.groups())
    return dict(norm_first=("prenorm" in kind), num_layers=L, hidden_size=H,
                attention_heads=A, feedforward_size=F)


def build_model(visual: str = "torchvision::resnet50",
                textual: str = "transdec_postnorm::L1_H1024_A16_F4096",
                vocab_size: int = 10000, dropout: float = 0.1,
                max_caption_length: int = 30) -> BicaptioningModel:
    cnn = visual.split("::")[-1]
    vb = VisualBackbone(cnn, visual_feature_size=2048)
    vb.visual_feature_size = vb.cnn.out_channels
    th = TextualHead(vb.cnn.out_channels, vocab_size, dropout=dropout,
                     mask_future_positions=True, max_caption_length=max_caption_length,
                     padding_idx=0, **parse_textual_name(textual))
    return BicaptioningModel(vb, th, sos_index=1, eos_index=2)


# ---------------------------------------------------------------------------------
# Training step (scripts/pretrain_virtex.py:145-163), optimizer grouping
# (virtex/factories.py:529-545), Lookahead (virtex/optim/lookahead.py:82-102), cosine
# schedule with linear warm-up (virtex/optim/lr_scheduler.py:174-183).
# On a CUDA-less host autocast/GradScaler disable themselves => fp32.
# ---------------------------------------------------------------------------------
NO_DECAY = r".*textual.(embedding|transformer).*(norm.*|bias)"


def param_groups(named_parameters, cnn_lr=0.2, lr=0.001, weight_decay=1e-4) -> List[dict]:
    groups = []
    for name, p in named_parameters:
        groups.append({"params": [p],
                       "lr": cnn_lr if "cnn" in name else lr,
                       "weight_decay": 0.0 if re.match(NO_DECAY, name) else weight_decay})
    return groups


def lr_multiplier(step: int, total_steps: int = 500000, warmup_steps: int = 10000) -> float:
    import math
    if step < warmup_steps:
        return step / float(max(1, warmup_steps))
    frac = (step - warmup_steps) / (total_steps - warmup_steps)
    return max(0.0, math.cos(frac * math.pi / 2) ** 2)


class TrainStep:
    """zero_grad -> fwd -> bwd -> clip(10) -> SGD(m=.9) -> Lookahead(k=5, a=.5) -> LR."""

    def __init__(self, model: nn.Module, clip: float = 10.0, k: int = 5, alpha: float = 0.5,
                 momentum: float = 0.9, total_steps: int = 500000, warmup_steps: int = 10000,
                 start_step: int = 0):
        self.model, self.clip, self.k, self.alpha = model, clip, k, alpha
        self.groups = param_groups(model.named_parameters())
        self.base_lrs = [g["lr"] for g in self.groups]
        self.opt = torch.optim.SGD(self.groups, momentum=momentum)
        self.slow = [g["params"][0].detach().clone() for g in self.groups]
        self.kc, self.step_idx = 0, start_step
        self.total_steps, self.warmup_steps = total_steps, warmup_steps
        self._set_lr()

    def _set_lr(self):
        mult = lr_multiplier(self.step_idx, self.total_steps, self.warmup_steps)
        for g, base in zip(self.opt.param_groups, self.base_lrs):
            g["lr"] = base * mult

    def __call__(self, batch) -> torch.Tensor:
        self.opt.zero_grad()
        loss = self.model(batch)["loss"]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip)
        self.opt.step()
        self.kc += 1
        if self.kc >= self.k:
            self.kc = 0
            with torch.no_grad():
                for g, slow in zip(self.opt.param_groups, self.slow):
                    p = g["params"][0]
                    p.mul_(self.alpha).add_(slow, alpha=1.0 - self.alpha)
                    slow.copy_(p)
        self.step_idx += 1
        self._set_lr()
        return loss.detach()
