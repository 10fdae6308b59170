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


def test_fixed_length_collation_leaves_the_loss_and_the_gradients_alone():
    """collate_captions(pad_to=T) gives every batch one shape (what launch replay needs); the extra padding columns are masked
    in attention, zeroed by the embedding and ignored by the loss: same loss, same gradients as the reference collation
    (the ORACLE model on CPU, fp32 -- the property is the reference's, not a kernel's)."""
    from oracle import bicaptioning as port
    g = torch.Generator().manual_seed(5)
    items = []
    for i, L in enumerate((3, 6, 4)):
        toks = torch.randint(4, 300, (L,), generator=g).tolist()
        items.append(vdata.caption_instance(i, torch.randn(3, 64, 64, generator=g), toks, max_caption_length=12))
    short = vdata.collate_captions(items)
    fixed = vdata.collate_captions(items, pad_to=12)
    assert short["caption_tokens"].shape == (3, 8) and fixed["caption_tokens"].shape == (3, 12)
    assert torch.equal(fixed["caption_tokens"][:, :8], short["caption_tokens"]) and (fixed["caption_tokens"][:, 8:] == 0).all()
    assert torch.equal(fixed["caption_lengths"], short["caption_lengths"])
    torch.manual_seed(0)
    model = port.build_model(textual="transdec_postnorm::L1_H128_A2_F256", vocab_size=304, max_caption_length=12, dropout=0.0).train()
    grads = []
    for b in (short, fixed):
        model.zero_grad(set_to_none=True)
        out = model(b)
        out["loss"].backward()
        grads.append((out["loss"].item(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
    assert abs(grads[0][0] - grads[1][0]) <= 1e-6 * abs(grads[0][0])
    for n, ga in grads[0][1].items():
        gb = grads[1][1][n]
        assert torch.allclose(ga, gb, rtol=1e-4, atol=1e-7), n


def test_cycle_moves_batches_and_restarts():
    items = _items(6)
    loader = torch.utils.data.DataLoader(items, batch_size=4, collate_fn=vdata.collate_captions)
    it = vdata.cycle(loader, torch.device("cpu"))
    sizes = [next(it)["image"].shape[0] for _ in range(5)]
    assert sizes == [4, 2, 4, 2, 4]


# ---- SURVEY.md 8f f2, second part: augmentation on the device, tokenizer, caption cache -------------------------------
import random

import numpy as np

from backends import BACKENDS, select
from oracle import augment as aug_oracle


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_device_augmentation_matches_the_pipeline_restatement(backend, dtype):
    """vtx_image_augment_u8 (crop window + bilinear resize + flip + colour jitter in the sampled order + Normalize +
    layout) against oracle/augment.py, the numpy restatement of the reference's albumentations pipeline, on the same
    sampled parameters: identical up to single uint8 counts on a handful of pixels (float summation order of the
    contrast mean), for training (random windows) and validation (centre window, identity jitter) parameters."""
    dev = select(backend)
    rng = random.Random(7)
    g = torch.Generator().manual_seed(7)
    N, Hs, Ws, size = 5, 57, 75, 32
    imgs = torch.randint(0, 256, (N, Hs, Ws, 3), generator=g, dtype=torch.uint8)
    windows = [vdata.sample_random_resized_crop(Hs, Ws, rng) for _ in range(N - 1)] + [vdata.center_crop_window(Hs, Ws, 256, 224)]
    flips = [rng.random() < 0.5 for _ in range(N - 1)] + [False]
    jitters = [vdata.sample_color_jitter(rng, p=1.0) for _ in range(N - 1)] + [(1.0, 1.0, 1.0, 0.0, 0xE4)]
    for (x0, y0, cw, ch) in windows:
        assert 0 <= x0 and 0 <= y0 and x0 + cw <= Ws and y0 + ch <= Hs and cw > 0 and ch > 0
    out = vdata.augment_batch(imgs.to(dev), windows, flips, jitters, size=size, dtype=dtype, packed=False)
    assert out.shape == (N, size, size, 8) and out.dtype == dtype
    assert out[..., 3:].abs().max().item() == 0
    tol_count = 1.02 / (255 * 0.224)                                      # one uint8 count after Normalize
    for n in range(N):
        ref = aug_oracle.pipeline(imgs[n].numpy(), windows[n], flips[n], jitters[n], size)        # (3, size, size)
        got = out[n, :, :, :3].float().cpu().permute(2, 0, 1).numpy()
        diff = np.abs(got - ref)
        bf = 0.02 if dtype == torch.bfloat16 else 1e-5                     # bf16 storage of values up to ~2.7
        assert (diff > tol_count + bf).sum() == 0, n
        assert (diff > bf).mean() < 0.01, (n, (diff > bf).mean())
    packed = vdata.augment_batch(imgs.to(dev), windows, flips, jitters, size=size, dtype=dtype)   # the stem's packed layout
    assert packed.shape == (N, size + 6, size + 6, 4)
    assert torch.equal(packed[:, 3:-3, 3:-3, :3], out[..., :3]) and packed[:, :3].abs().max().item() == 0


@pytest.mark.parametrize("backend", BACKENDS)
def test_crop_window_flip_and_normalize_are_pinned_exactly(backend):
    """The part of the image pipeline that needs NO third-party arithmetic to define: with a crop window
    of the output size the bilinear resize is the identity (source coordinate = destination, weight 0), so what remains of
    albumentations' RandomResizedCrop / CenterCrop -> T.HorizontalFlip (cv2.flip(img, 1): column reversal) -> alb.Normalize
    (img.astype(float32) - 255 mean) * (1 / (255 std)) (virtex/data/transforms.py:5-35, 85-97, virtex/factories.py:132-154) is
    exact array arithmetic: window placement to the pixel, the flip, and Normalize to fp32 rounding.  (What stays unpinned until
    cv2 / albumentations are installable: the fixed-point weights of cv2.resize when the window is NOT the output size, and
    ColorJitter's uint8 tables -- oracle/augment.py restates their published formulas)"""
    dev = select(backend)
    g = torch.Generator().manual_seed(11)
    N, Hs, Ws, size = 6, 41, 53, 24
    imgs = torch.randint(0, 256, (N, Hs, Ws, 3), generator=g, dtype=torch.uint8)
    windows = [(0, 0, size, size), (Ws - size, Hs - size, size, size), (7, 3, size, size), (13, 17, size, size), (29, 0, size, size),
               vdata.center_crop_window(Hs, Ws, min(Hs, Ws), size)]                  # SmallestMaxSize(short side) = no resize, then CenterCrop
    assert windows[-1] == ((Ws - size) // 2, (Hs - size) // 2, size, size)          # albumentations' CenterCrop: floor of the half margin
    flips = [False, True, True, False, True, False]
    out = vdata.augment_batch(imgs.to(dev), windows, flips, None, size=size, dtype=torch.float32, packed=False)
    mean = np.array(aug_oracle.MEAN, dtype=np.float32) * np.float32(255.0)
    inv = np.float32(1.0) / (np.array(aug_oracle.STD, dtype=np.float32) * np.float32(255.0))
    for n, ((x0, y0, cw, ch), flip) in enumerate(zip(windows, flips)):
        crop = imgs[n, y0:y0 + ch, x0:x0 + cw].numpy()
        if flip:
            crop = crop[:, ::-1]
        want = (crop.astype(np.float32) - mean) * inv                               # albumentations.augmentations.functional.normalize
        got = out[n, :, :, :3].cpu().numpy()
        # the pixel values themselves, exactly (a wrong window or flip is off by whole counts)
        assert np.array_equal(np.rint(got / inv + mean).astype(np.int64), crop.astype(np.int64)), n
        assert np.abs(got - want).max() <= 4e-7 * np.abs(want).max() + 1e-7, (n, np.abs(got - want).max())
    # bf16 output = the fp32 value rounded once
    out16 = vdata.augment_batch(imgs.to(dev), windows, flips, None, size=size, dtype=torch.bfloat16, packed=False)
    assert torch.equal(out16[..., :3].cpu(), out[..., :3].cpu().to(torch.bfloat16))


def test_crop_and_jitter_samplers_follow_the_reference_defaults():
    rng = random.Random(3)
    areas, ratios = [], []
    for _ in range(2000):
        x0, y0, cw, ch = vdata.sample_random_resized_crop(300, 400, rng)
        assert 0 <= x0 <= 400 - cw and 0 <= y0 <= 300 - ch
        areas.append(cw * ch / 120000.0); ratios.append(cw / ch)
    assert 0.19 < min(areas) and max(areas) <= 1.0 and 0.4 < np.mean(areas) < 0.65      # scale (0.2, 1.0), large windows of extreme ratio are rejected
    assert 0.74 < min(ratios) and max(ratios) < 1.35                                    # ratio (3/4, 4/3)
    js = [vdata.sample_color_jitter(rng) for _ in range(2000)]
    ident = sum(1 for j in js if j[:4] == (1.0, 1.0, 1.0, 0.0))
    assert 300 < ident < 500                                                            # p = 0.8
    active = [j for j in js if j[:4] != (1.0, 1.0, 1.0, 0.0)]
    assert all(0.6 <= j[0] <= 1.4 and 0.6 <= j[1] <= 1.4 and 0.6 <= j[2] <= 1.4 and -0.1 <= j[3] <= 0.1 for j in active)
    assert all(sorted((j[4] >> (2 * k)) & 3 for k in range(4)) == [0, 1, 2, 3] for j in js)
    assert len({j[4] for j in active}) == 24                                            # every order of the four operations
    assert vdata.center_crop_window(480, 640) == (110, 30, 420, 420)
    assert vdata.flip_caption("a dog left of the right door") == "a dog right of the left door"


def test_sentencepiece_tokenizer_and_caption_cache(tmp_path):
    """Same interface as virtex/data/tokenizers.py on a model trained here (no network: a toy BPE model with the
    reference's special tokens, virtex/config.py:72-82), item format through caption_instance, int16 caption cache."""
    import sentencepiece as sp
    corpus = tmp_path / "captions.txt"
    words = ["a", "dog", "cat", "sits", "left", "right", "of", "the", "red", "door", "on", "grass", "two", "people", "walk"]
    rng = random.Random(0)
    corpus.write_text("\n".join(" ".join(rng.choice(words) for _ in range(rng.randint(4, 9))) for _ in range(400)))
    prefix = str(tmp_path / "toy")
    sp.SentencePieceTrainer.train(input=str(corpus), model_prefix=prefix, vocab_size=60, model_type="bpe", character_coverage=1.0,
                                  bos_id=-1, eos_id=-1, control_symbols="[SOS],[EOS],[MASK]", minloglevel=2)
    tok = vdata.SentencePieceBPETokenizer(prefix + ".model")
    assert tok.get_vocab_size() == 60
    assert (tok.token_to_id("<unk>"), tok.token_to_id("[SOS]"), tok.token_to_id("[EOS]"), tok.token_to_id("[MASK]")) == (0, 1, 2, 3)
    ids = tok.encode("a dog sits left of the door")
    assert all(isinstance(i, int) and 3 < i < 60 for i in ids) and tok.decode(ids) == "a dog sits left of the door"
    assert tok.id_to_token(1) == "[SOS]"
    import pickle
    assert pickle.loads(pickle.dumps(tok)).encode("two people walk") == tok.encode("two people walk")     # DataLoader workers
    caps = ["a dog sits left of the door", "two people walk on the grass"]
    cache = vdata.CaptionTokenCache(tok, caps)
    assert cache.tokens(0) == ids and cache.tokens(0, flipped=True) == tok.encode("a dog sits right of the door")
    assert cache.tokens(1, flipped=True) == cache.tokens(1) and cache.plain[0].dtype == np.int16
    item = vdata.caption_instance(7, torch.zeros(8, 8, 3, dtype=torch.uint8), cache.tokens(0))
    assert item["caption_tokens"][0] == 1 and item["caption_tokens"][-1] == 2 and int(item["caption_lengths"]) == len(ids) + 2


def test_augmentation_oracle_building_blocks_agree_with_independent_implementations():
    """cv2 / albumentations are not installable here, so `oracle/augment.py` restates their published formulas (parity unpinned,
    DESIGN.md 8 f2).  What CAN be checked is that its building blocks are the textbook operations, against implementations that
    are installed and share no code with it: the bilinear geometry of cv2.INTER_LINEAR (half-pixel centres, no antialiasing,
    edge clamping) = torch's `interpolate(mode="bilinear", align_corners=False)`; the ITU-R 601 luma of the grayscale used by
    contrast / saturation = Pillow's `convert("L")`; the hue rotation = Python's `colorsys` round trip.  Each within ONE count of
    the uint8 grid (the same margin cv2's fixed-point weights / integer HSV tables have against the formulas)."""
    import colorsys
    import torch.nn.functional as F
    from PIL import Image
    from oracle import augment as oa
    rng = np.random.default_rng(11)
    # -- crop + bilinear resize, up- and down-scaling, odd windows
    for (H, W, x0, y0, cw, ch, size) in [(97, 131, 5, 9, 100, 71, 64), (60, 80, 0, 0, 80, 60, 224), (300, 200, 17, 33, 150, 240, 224),
                                          (40, 40, 3, 2, 31, 37, 32)]:
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        mine = oa.resize_crop(img, x0, y0, cw, ch, size)
        crop = torch.from_numpy(img[y0:y0 + ch, x0:x0 + cw].astype(np.float32)).permute(2, 0, 1)[None]
        ref = F.interpolate(crop, size=(size, size), mode="bilinear", align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
        ref = np.clip(np.rint(ref), 0, 255)
        diff = np.abs(mine - ref)
        assert diff.max() <= 1 and (diff == 0).mean() > 0.99, (H, W, size, diff.max(), (diff == 0).mean())
    # -- luma
    img = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    pil = np.asarray(Image.fromarray(img).convert("L")).astype(np.float32)
    diff = np.abs(oa.gray(img.astype(np.float32)) - pil)
    assert diff.max() <= 1 and (diff == 0).mean() > 0.95
    # -- hue rotation: colorsys round trip per pixel (float), back on the uint8 grid
    img = rng.integers(0, 256, (24, 24, 3), dtype=np.uint8)
    for h in (0.1, -0.07, 0.5):
        mine = oa.hue_shift(img.astype(np.float32), h)
        ref = np.empty_like(mine)
        for y in range(img.shape[0]):
            for x in range(img.shape[1]):
                r, g, b = (float(v) / 255.0 for v in img[y, x])
                hh, ss, vv = colorsys.rgb_to_hsv(r, g, b)
                ref[y, x] = [c * 255.0 for c in colorsys.hsv_to_rgb((hh + h) % 1.0, ss, vv)]
        ref = np.clip(np.rint(ref), 0, 255)
        diff = np.abs(mine - ref)
        assert diff.max() <= 1 and (diff == 0).mean() > 0.97, (h, diff.max(), (diff == 0).mean())
