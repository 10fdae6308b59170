"""Batch assembly on the way into the step (first part).

The reference's wire format is the five-key dict produced by ``CaptioningDataset.collate_fn``
(/root/reference/virtex/data/datasets/captioning.py:79-100) and moved to the device key by key by ``cycle``
(/root/reference/virtex/utils/common.py:14-37).  These helpers produce the same dict; the one extension is that
``image`` may stay uint8 HWC (decoder output): the backbone then normalises it on the device
(``vtx_image_u8_to_nhwc``), so a 256-image batch crosses PCIe as 38.5 MB instead of 154 MB and the CPU never touches
the pixels as floats.
"""
from typing import Dict, Iterable, Iterator, List

import torch


def collate_captions(instances: List[Dict[str, torch.Tensor]], padding_idx: int = 0, pad_to: int = 0) -> Dict[str, torch.Tensor]:
    """Right-pad ``caption_tokens`` / ``noitpac_tokens`` with ``padding_idx`` to the longest caption of the batch and
    stack everything else -- key for key what the reference's collate_fn returns.  ``pad_to`` > 0 pads to at least that
    length instead (the maximum caption length): every batch then has ONE shape, which launch replay and hipGraphs need
    (virtex_amd.replay.StepReplay); the loss and every gradient are those of the shorter batch -- padded positions are masked
    in attention, zeroed by the embedding and ignored by the loss (tests/test_data.py)."""
    longest = max(max(int(d["caption_tokens"].numel()) for d in instances), int(pad_to))
    n = len(instances)
    caption_tokens = torch.full((n, longest), padding_idx, dtype=torch.long)
    noitpac_tokens = torch.full((n, longest), padding_idx, dtype=torch.long)
    for i, d in enumerate(instances):
        t, r = d["caption_tokens"], d["noitpac_tokens"]
        caption_tokens[i, : t.numel()] = t
        noitpac_tokens[i, : r.numel()] = r
    return {
        "image_id": torch.stack([d["image_id"] for d in instances], dim=0),
        "image": torch.stack([d["image"] for d in instances], dim=0),
        "caption_tokens": caption_tokens,
        "noitpac_tokens": noitpac_tokens,
        "caption_lengths": torch.stack([d["caption_lengths"] for d in instances]),
    }


def caption_instance(image_id: int, image, token_ids: Iterable[int], sos_id: int = 1, eos_id: int = 2,
                     max_caption_length: int = 30) -> Dict[str, torch.Tensor]:
    """One dataset item in the reference's format (captioning.py:66-77): [SOS] + tokens + [EOS], truncated to
    ``max_caption_length``; ``noitpac_tokens`` is the flipped sequence.  ``image`` is kept as given (float CHW or
    uint8 HWC)."""
    tokens = [sos_id, *token_ids, eos_id][:max_caption_length]
    t = torch.tensor(tokens, dtype=torch.long)
    return {"image_id": torch.tensor(image_id, dtype=torch.long), "image": torch.as_tensor(image),
            "caption_tokens": t, "noitpac_tokens": t.flip(0), "caption_lengths": torch.tensor(len(tokens), dtype=torch.long)}


def cycle(dataloader, device, start_iteration: int = 0) -> Iterator[Dict[str, torch.Tensor]]:
    """Endless batches on ``device``; a DistributedSampler gets the running iteration as its shuffling epoch, as in the
    reference.  Copies are issued non-blocking (pinned loaders overlap them with the previous step)."""
    iteration = start_iteration
    while True:
        sampler = getattr(dataloader, "sampler", None)
        if isinstance(sampler, torch.utils.data.DistributedSampler):
            sampler.set_epoch(iteration)
        for batch in dataloader:
            yield {k: v.to(device, non_blocking=True) for k, v in batch.items()}
            iteration += 1


# ----------------------------------------------------------------------------------------------------------------------
# The image / caption side of the dataset item (second part): what the reference does on the CPU per
# item -- alb.RandomResizedCrop, T.HorizontalFlip (with the caption's left<->right swap), alb.ColorJitter, alb.Normalize,
# np.transpose (virtex/data/transforms.py:5-97, virtex/factories.py:132-154), SentencePiece tokenisation
# (virtex/data/tokenizers.py:52-54) -- split into HOST parameter sampling / string work and ONE device kernel per batch
# (vtx_image_augment_u8) that reads the decoder's uint8 pixels and writes the stem's NHWC compute-dtype input.
# ----------------------------------------------------------------------------------------------------------------------
import math
import random as _random
from typing import Optional, Sequence, Tuple

COLOR_OPS = ("brightness", "contrast", "saturation", "hue")


def sample_random_resized_crop(height: int, width: int, rng: _random.Random, scale=(0.2, 1.0), ratio=(0.75, 1.333)):
    """(x0, y0, cw, ch): albumentations' RandomResizedCrop window (torchvision's algorithm: 10 attempts at a random
    area fraction and log-uniform aspect ratio, then the central crop with the ratio clamped), with the reference's
    defaults scale (0.2, 1.0), ratio (0.75, 1.333) (virtex/factories.py:138-140)."""
    area = height * width
    for _ in range(10):
        target = rng.uniform(*scale) * area
        aspect = math.exp(rng.uniform(math.log(ratio[0]), math.log(ratio[1])))
        w, h = int(round(math.sqrt(target * aspect))), int(round(math.sqrt(target / aspect)))
        if 0 < w <= width and 0 < h <= height:
            return rng.randint(0, width - w), rng.randint(0, height - h), w, h
    in_ratio = width / height
    if in_ratio < min(ratio):
        w, h = width, int(round(width / min(ratio)))
    elif in_ratio > max(ratio):
        h, w = height, int(round(height * max(ratio)))
    else:
        w, h = width, height
    return (width - w) // 2, (height - h) // 2, w, h


def center_crop_window(height: int, width: int, resize: int = 256, crop: int = 224):
    """The validation pipeline alb.SmallestMaxSize(256) + CenterCrop(224) (transforms.py:91-97) as ONE source window:
    the centre square that the 224-crop of the 256-resized image covers."""
    side = min(height, width) * crop / resize
    cw = ch = int(round(side))
    return (width - cw) // 2, (height - ch) // 2, cw, ch


def sample_color_jitter(rng: _random.Random, brightness=0.4, contrast=0.4, saturation=0.4, hue=0.1, p=0.8):
    """(brightness, contrast, saturation, hue, order code) of alb.ColorJitter with the reference's defaults
    (virtex/factories.py:146-148): factors uniform in [1-x, 1+x] (hue in [-x, x]), applied in a random order with
    probability p; identity otherwise."""
    if rng.random() >= p:
        return 1.0, 1.0, 1.0, 0.0, 0xE4
    b = rng.uniform(max(0.0, 1 - brightness), 1 + brightness)
    c = rng.uniform(max(0.0, 1 - contrast), 1 + contrast)
    s = rng.uniform(max(0.0, 1 - saturation), 1 + saturation)
    h = rng.uniform(-hue, hue)
    order = list(range(4))
    rng.shuffle(order)
    code = sum(op << (2 * k) for k, op in enumerate(order))
    return b, c, s, h, code


def flip_caption(caption: str) -> str:
    """The caption half of the reference's HorizontalFlip (transforms.py:29-35): swap the words left and right."""
    return caption.replace("left", "[TMP]").replace("right", "left").replace("[TMP]", "right")


def augment_batch(images_u8: torch.Tensor, windows: Sequence[Tuple[int, int, int, int]], flips: Sequence[bool], jitters=None,
                  size: int = 224, dtype: torch.dtype = torch.bfloat16, packed: bool = True) -> torch.Tensor:
    """uint8 (N, Hs, Ws, 3) on the device -> the backbone's stem input: (N, size+2*halo, size+2*halo, Cpad) NHWC in
    `dtype` (the packed 4-channel layout with its zero frame by default).  windows / flips / jitters: per-image
    parameters from the samplers above (jitters None = identity: validation)."""
    import numpy as np
    from . import ops
    from .modules.visual_backbones import STEM_CPAD, STEM_HALO, STEM_PACK_C
    n = images_u8.shape[0]
    rec = np.zeros(n, dtype=np.dtype([("x0", "<i4"), ("y0", "<i4"), ("cw", "<i4"), ("ch", "<i4"), ("flip", "<i4"),
                                      ("b", "<f4"), ("c", "<f4"), ("s", "<f4"), ("h", "<f4"), ("order", "<i4")]))
    for i in range(n):
        j = jitters[i] if jitters is not None else (1.0, 1.0, 1.0, 0.0, 0xE4)
        rec[i] = (*windows[i], int(bool(flips[i])), *j)
    params = torch.from_numpy(rec.view(np.uint8).copy()).to(images_u8.device)
    cpad, halo = (STEM_PACK_C, STEM_HALO) if packed else (STEM_CPAD, 0)
    return ops.image_augment_u8(images_u8, params, size, dtype, cpad, halo)


class SentencePieceBPETokenizer:
    """Same interface as the reference's tokenizer (virtex/data/tokenizers.py:6-62) over a trained SentencePiece model."""
    SP_SPACE = u"\u2581"

    def __init__(self, model_path: str):
        import sentencepiece as sp
        self.model_path = model_path
        self.model = sp.SentencePieceProcessor()
        self.model.Load(model_path)

    def __getstate__(self):
        state = self.__dict__.copy()
        state["model"] = None
        return state

    def __setstate__(self, state):
        import sentencepiece as sp
        self.__dict__ = state
        self.model = sp.SentencePieceProcessor()
        self.model.Load(self.model_path)

    def get_vocab_size(self) -> int:
        return len(self.model)

    def token_to_id(self, token: str) -> int:
        return self.model.piece_to_id(token)

    def id_to_token(self, token_id: int) -> str:
        return self.model.id_to_piece(token_id)

    def encode(self, text: str) -> List[int]:
        return self.model.EncodeAsIds(text)

    def decode(self, token_ids: List[int]) -> str:
        return self.model.DecodeIds(token_ids)


class CaptionTokenCache:
    """Pre-tokenised captions as int16 arrays (vocabularies here are <= 32 K): both orientations of every caption --
    as written and with left/right swapped for the flipped image -- are tokenised ONCE, so that a training item is a
    table lookup instead of a SentencePiece call per item and epoch (the reference tokenises in __getitem__,
    captioning.py:66)."""

    def __init__(self, tokenizer, captions: Sequence[str]):
        import numpy as np
        vocab = tokenizer.get_vocab_size() if hasattr(tokenizer, "get_vocab_size") else None
        dt = np.int16 if (vocab is not None and vocab <= 32767) else np.int32       # unknown or large vocabulary: no silent wrap
        self.plain = [np.asarray(tokenizer.encode(c), dtype=dt) for c in captions]
        self.flipped = [np.asarray(tokenizer.encode(flip_caption(c)), dtype=dt) for c in captions]

    def tokens(self, index: int, flipped: bool = False) -> List[int]:
        return (self.flipped if flipped else self.plain)[index].astype("int64").tolist()
