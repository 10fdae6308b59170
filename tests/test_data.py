"""Batch assembly (first part): virtex_amd.data against the reference's collate_fn / item format."""
import pytest
import torch

from oracle import reference_import
from virtex_amd import data as vdata


def _items(n=4, seed=0, uint8=False):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        L = int(torch.randint(1, 40, (1,), generator=g))
        toks = torch.randint(4, 1000, (L,), generator=g).tolist()
        img = (torch.randint(0, 256, (8, 8, 3), generator=g, dtype=torch.uint8) if uint8
               else torch.randn(3, 8, 8, generator=g))
        out.append(vdata.caption_instance(100 + i, img, toks))
    return out


def test_collate_matches_the_reference_semantics():
    items = _items()
    b = vdata.collate_captions(items, padding_idx=0)
    assert set(b) == {"image_id", "image", "caption_tokens", "noitpac_tokens", "caption_lengths"}
    ref_tok = torch.nn.utils.rnn.pad_sequence([d["caption_tokens"] for d in items], batch_first=True, padding_value=0)
    ref_rev = torch.nn.utils.rnn.pad_sequence([d["noitpac_tokens"] for d in items], batch_first=True, padding_value=0)
    assert torch.equal(b["caption_tokens"], ref_tok) and torch.equal(b["noitpac_tokens"], ref_rev)
    assert b["caption_tokens"].dtype == torch.long and b["caption_lengths"].tolist() == [min(d["caption_tokens"].numel(), 30) for d in items]
    for d in items:                                      # item format: [SOS] ... [EOS], truncated to 30, reversed copy
        t = d["caption_tokens"]
        assert t[0] == 1 and t.numel() <= 30 and torch.equal(d["noitpac_tokens"], t.flip(0))
        assert (t[-1] == 2) or t.numel() == 30
    u8 = vdata.collate_captions(_items(uint8=True))
    assert u8["image"].dtype == torch.uint8 and u8["image"].shape == (4, 8, 8, 3)


@pytest.mark.reference
def test_collate_equals_live_reference():
    reference_import.import_reference()
    from virtex.data.datasets.captioning import CaptioningDataset
    items = _items(5, seed=3)
    fake_self = type("S", (), {"padding_idx": 0})()
    ref = CaptioningDataset.collate_fn(fake_self, items)
    ours = vdata.collate_captions(items, 0)
    assert list(ref) == list(ours)
    for k in ref:
        assert torch.equal(ref[k], ours[k]), k


def test_cycle_moves_batches_and_restarts():
    items = _items(6)
    loader = torch.utils.data.DataLoader(items, batch_size=4, collate_fn=vdata.collate_captions)
    it = vdata.cycle(loader, torch.device("cpu"))
    sizes = [next(it)["image"].shape[0] for _ in range(5)]
    assert sizes == [4, 2, 4, 2, 4]
