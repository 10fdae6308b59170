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


def collate_captions(instances: List[Dict[str, torch.Tensor]], padding_idx: int = 0) -> Dict[str, torch.Tensor]:
    """Right-pad ``caption_tokens`` / ``noitpac_tokens`` with ``padding_idx`` to the longest caption of the batch and
    stack everything else -- key for key what the reference's collate_fn returns."""
    longest = max(int(d["caption_tokens"].numel()) for d in instances)
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
