"""On-device synthetic COCO-shaped batches for the benchmark.

Same 5-key dict the reference's collate function emits
(/root/reference/virtex/data/datasets/captioning.py:79-100): fp32 N(0,1) images (B,3,S,S),
int64 captions [SOS]=1 ... [EOS]=2 with interior tokens uniform in [4, V), `noitpac_tokens` =
the reversed caption, all lengths = max_len (no padding: the benchmark configuration)."""
from typing import Dict

import torch


def synthetic_batch(batch_size: int, device, image_size: int = 224, max_len: int = 30,
                    vocab_size: int = 10000, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator(device="cpu").manual_seed(seed)
    image = torch.randn(batch_size, 3, image_size, image_size, generator=g)
    tokens = torch.randint(4, vocab_size, (batch_size, max_len), generator=g)
    tokens[:, 0], tokens[:, -1] = 1, 2
    lengths = torch.full((batch_size,), max_len, dtype=torch.int64)
    batch = {"image_id": torch.arange(batch_size), "image": image, "caption_tokens": tokens,
             "noitpac_tokens": tokens.flip(1).contiguous(), "caption_lengths": lengths}
    return {k: v.to(device) for k, v in batch.items()}
