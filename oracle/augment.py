"""numpy restatement of the reference's per-item image pipeline (TEST INFRASTRUCTURE; parity UNPINNED: albumentations and
cv2 are not installed in this environment, so this follows their published formulas, not their binaries).

  alb.RandomResizedCrop -> crop + cv2.resize(INTER_LINEAR)         virtex/factories.py:138-140, transforms.py:38-48
  T.HorizontalFlip      -> cv2.flip(img, 1)                         transforms.py:5-35
  alb.ColorJitter       -> torchvision-style brightness / contrast / saturation / hue functionals on uint8 images,
                           in the sampled order                     virtex/factories.py:146-148
  alb.Normalize         -> (img - 255*mean) / (255*std)             transforms.py:85-89
  np.transpose(HWC->CHW)                                            virtex/data/datasets/captioning.py:64

cv2's INTER_LINEAR convention: source coordinate = (dst + 0.5) * scale - 0.5, taps clamped to the window.  Every stage
returns to the uint8 grid (round half to even, clip) like uint8 images do on the CPU path; cv2's 11-bit fixed-point
interpolation weights and its uint8 HSV tables can differ from this by one count.
Cross-checked (not pinned): resize geometry vs torch F.interpolate, luma vs Pillow convert("L"), hue rotation vs colorsys, each
within one count -- tests/test_data.py::test_augmentation_oracle_building_blocks_agree_with_independent_implementations.
"""
import numpy as np

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def q8(v):
    return np.clip(np.rint(v), 0, 255)


def t8(v):
    """brightness / contrast on uint8 images: a lookup table built as clip(...).astype(uint8) -- the cast truncates."""
    return np.floor(np.clip(v, 0, 255))


def resize_crop(img, x0, y0, cw, ch, size):
    crop = img[y0:y0 + ch, x0:x0 + cw].astype(np.float32)
    fy = (np.arange(size, dtype=np.float32) + 0.5) * np.float32(ch) / np.float32(size) - 0.5
    fx = (np.arange(size, dtype=np.float32) + 0.5) * np.float32(cw) / np.float32(size) - 0.5

    def taps(f, n):
        i0 = np.floor(f).astype(np.int64)
        w = (f - i0).astype(np.float32)
        w[i0 < 0] = 0; i0 = np.maximum(i0, 0)
        i1 = np.minimum(i0 + 1, n - 1)
        w[i0 > n - 1] = 0; i0 = np.minimum(i0, n - 1)
        return i0, i1, w
    y0i, y1i, wy = taps(fy, ch)
    x0i, x1i, wx = taps(fx, cw)
    top = crop[y0i][:, x0i] + wx[None, :, None] * (crop[y0i][:, x1i] - crop[y0i][:, x0i])
    bot = crop[y1i][:, x0i] + wx[None, :, None] * (crop[y1i][:, x1i] - crop[y1i][:, x0i])
    return q8(top + wy[:, None, None] * (bot - top)).astype(np.float32)


def gray(img):
    return q8(np.float32(0.299) * img[..., 0] + np.float32(0.587) * img[..., 1] + np.float32(0.114) * img[..., 2])


def hue_shift(img, h):
    r, g, b = img[..., 0], img[..., 1], img[..., 2]
    mx, mn = img.max(-1), img.min(-1)
    d = mx - mn
    safe = np.where(d > 0, d, 1)
    hh = np.where(mx == r, (g - b) / safe, np.where(mx == g, 2 + (b - r) / safe, 4 + (r - g) / safe)) / 6 + np.float32(h)
    hh = hh - np.floor(hh)
    s = d / np.where(mx > 0, mx, 1)
    i = np.floor(hh * 6)
    f = hh * 6 - i
    v = mx
    a, bb, c = v * (1 - s), v * (1 - s * f), v * (1 - s * (1 - f))
    k = (i.astype(np.int64) % 6)[..., None]
    out = np.select([k == 0, k == 1, k == 2, k == 3, k == 4, k == 5],
                    [np.stack([v, c, a], -1), np.stack([bb, v, a], -1), np.stack([a, v, c], -1),
                     np.stack([a, bb, v], -1), np.stack([c, a, v], -1), np.stack([v, a, bb], -1)])
    return np.where((d > 0)[..., None], q8(out), img).astype(np.float32)


def jitter(img, b, c, s, h, order):
    for k in range(4):
        op = (order >> (2 * k)) & 3
        if op == 0:
            img = t8(img * np.float32(b))
        elif op == 1:
            img = t8(img * np.float32(c) + np.float32(gray(img).mean(dtype=np.float64)) * np.float32(1 - c))
        elif op == 2:
            img = q8(img * np.float32(s) + gray(img)[..., None] * np.float32(1 - s))
        elif h != 0:
            img = hue_shift(img, h)
        img = img.astype(np.float32)
    return img


def pipeline(img_u8, window, flip, jit, size=224):
    """-> float32 (3, size, size), the reference's per-item `image` tensor."""
    img = resize_crop(img_u8, *window, size)
    if flip:
        img = img[:, ::-1]
    img = jitter(img, *jit)
    out = (img - 255 * np.asarray(MEAN, np.float32)) / (255 * np.asarray(STD, np.float32))
    return np.transpose(out, (2, 0, 1)).astype(np.float32)
