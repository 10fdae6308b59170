"""tests/golden/jpeg_cases.npz: small baseline JPEG streams of every supported kind with the pixels Pillow's libjpeg-turbo
decodes from them (EXIF orientation applied, as cv2.imread of the reference does).  python tests/golden/make_jpeg_goldens.py"""
import io
import os
import sys

import numpy as np
from PIL import Image, ImageOps

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from test_jpeg import _image, _with_orientation  # noqa: E402

rng = np.random.default_rng(7)
streams = []


def enc(arr, **kw):
    buf = io.BytesIO(); Image.fromarray(arr).save(buf, "JPEG", **kw); return buf.getvalue()


for (h, w, sub, q) in [(24, 40, 0, 85), (37, 53, 2, 75), (48, 32, 1, 60), (5, 3, 2, 50), (1, 1, 2, 75), (17, 4, 1, 90), (40, 56, 2, 30)]:
    streams.append(enc(_image(h, w, rng), quality=q, subsampling=sub))
streams.append(enc(_image(33, 47, rng)[..., 0], quality=80))                                        # greyscale
streams.append(enc(_image(48, 64, rng), quality=80, optimize=True, restart_marker_blocks=2))       # restart intervals, optimised tables
streams.append(enc(_image(32, 32, rng, smooth=False), quality=95, subsampling=2))                  # noise: long Huffman codes
for o in (3, 6, 8):
    streams.append(_with_orientation(enc(_image(24, 40, rng), quality=85, subsampling=2), o))
out = {"count": np.array(len(streams))}
for i, s in enumerate(streams):
    out[f"jpeg_{i}"] = np.frombuffer(s, dtype=np.uint8)
    out[f"rgb_{i}"] = np.asarray(ImageOps.exif_transpose(Image.open(io.BytesIO(s))).convert("RGB"))
np.savez_compressed(os.path.join(HERE, "jpeg_cases.npz"), **out)
print(len(streams), "cases", sum(len(s) for s in streams), "bytes of JPEG")
