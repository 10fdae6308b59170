"""JPEG decode + the COCO Captions reader: the part of the reference's dataset item in front of the
image transform -- `cv2.imread` + `cvtColor(BGR2RGB)` (/root/reference/virtex/data/datasets/coco_captions.py:56-63) and the
annotation bookkeeping of `CocoCaptionsDataset.__init__` (:22-52).

`decode_jpeg(data, device)` returns the decoder's uint8 (H, W, 3) RGB pixels ON THE DEVICE -- what
`virtex_amd.data.augment_batch` / `vtx_image_augment_u8` read; the file's bytes are the only thing that crosses PCIe besides
the int16 coefficients.  Host: marker parsing + Huffman decoding (C++, thread-safe, releases the GIL); device: dequantise,
inverse DCT, chroma upsampling, colour conversion -- bit-exact with libjpeg-turbo's defaults (csrc/jpeg.hip; pinned against
Pillow's libjpeg-turbo in tests/test_jpeg.py, CPU restatement in oracle/jpeg.py).  There is no fallback decoder: a stream the
kernels do not take (progressive, CMYK, ...) raises VtxError -- convert such files once, offline."""
import ctypes
import json
import os
import threading
import unicodedata
from collections import defaultdict
from typing import Dict, List, Tuple

import torch

from . import _lib
from ._lib import c_int, ptr, stream_ptr

c_long = ctypes.c_long


def jpeg_info(data: bytes) -> Dict[str, int]:
    info = (ctypes.c_int * 8)()
    _lib.call("vtx_jpeg_info", ctypes.c_char_p(data), c_long(len(data)), info)
    keys = ("width", "height", "components", "luma_h", "luma_v", "orientation", "blocks", "plane_bytes")
    return dict(zip(keys, [int(v) for v in info]))


def decode_jpeg(data: bytes, device, apply_orientation: bool = True) -> torch.Tensor:
    """bytes of one baseline JPEG -> uint8 (H, W, 3) RGB on `device` (EXIF orientation applied like cv2.imread)."""
    device = torch.device(device)
    info = jpeg_info(data)
    pin = device.type == "cuda"
    coef = torch.empty(info["blocks"] * 64, dtype=torch.int16, pin_memory=pin)
    qt = torch.empty(4 * 64, dtype=torch.int16, pin_memory=pin)                 # uint16 values: same bits
    buf = ctypes.c_char_p(data)
    _lib.call("vtx_jpeg_entropy_decode", buf, c_long(len(data)), ctypes.c_void_p(coef.data_ptr()), c_long(coef.numel()),
              ctypes.c_void_p(qt.data_ptr()))
    coef_d, qt_d = coef.to(device, non_blocking=True), qt.to(device, non_blocking=True)
    planes = torch.empty(info["plane_bytes"], dtype=torch.uint8, device=device)
    swap = apply_orientation and info["orientation"] >= 5
    H, W = (info["width"], info["height"]) if swap else (info["height"], info["width"])
    rgb = torch.empty(H, W, 3, dtype=torch.uint8, device=device)
    _lib.call("vtx_jpeg_reconstruct", buf, c_long(len(data)), ptr(coef_d), ptr(qt_d), ptr(planes), ptr(rgb),
              c_int(1 if apply_orientation else 0), stream_ptr(rgb))
    return rgb


_staging = {}        # device -> [pinned int16 buffer, event recorded after the last copy out of it]: ONE staging area per device, grown on demand
_staging_lock = threading.Lock()      # ... used by one decode_jpeg_batch call at a time (calls from several threads take turns)


def _staging_area(device, n16: int):
    """A pinned int16 buffer of at least n16 elements, reused from call to call (allocating page-locked memory costs
    milliseconds).  The previous call's copies out of it must have finished before a new decode writes into it."""
    ent = _staging.get(device)
    if ent is not None and ent[1] is not None:
        ent[1].synchronize()
    if ent is None or ent[0].numel() < n16:
        ent = _staging[device] = [torch.empty(max(n16, 1 << 20), dtype=torch.int16, pin_memory=(device.type == "cuda")), None]
    return ent


def decode_jpeg_batch(blobs, device, threads: int = 8, apply_orientation: bool = True, chunk: int = 16) -> List[torch.Tensor]:
    """Many files at once.  The serial Huffman decoding of the images runs on `threads` host threads (the C entry points hold no
    Python state and ctypes releases the GIL), each writing its int16 coefficients into ITS slice of one page-locked staging
    buffer; the main thread follows in input order: as soon as the next `chunk` images are decoded, their slice crosses PCIe in ONE
    asynchronous copy and the per-image device work (inverse DCT, upsampling, colour conversion: two launches per image, all
    images sharing one plane scratch in stream order) is enqueued on the caller's stream.  Per image the main thread issues one C
    call and one small allocation (round 5 allocated page-locked memory and issued two copies per image: 7 400-9 900 images/s
    whatever the thread count; this form scales with the host threads).  Returns the uint8 (H, W, 3) tensors in input order."""
    device = torch.device(device)
    if len(blobs) == 0:
        return []
    with _staging_lock:
        return _decode_jpeg_batch(blobs, device, threads, apply_orientation, chunk)


def _decode_jpeg_batch(blobs, device, threads, apply_orientation, chunk):
    from concurrent.futures import ThreadPoolExecutor
    n = len(blobs)
    infos = [jpeg_info(b) for b in blobs]
    QT = 4 * 64
    sizes = [i["blocks"] * 64 + QT for i in infos]                       # coefficients, then the four quantisation tables
    offs = [0]
    for sz in sizes:
        offs.append(offs[-1] + ((sz + 7) // 8) * 8)                       # 16-byte aligned slices
    stage = _staging_area(device, offs[-1])
    host = stage[0]
    base = host.data_ptr()

    def work(i):
        data = blobs[i]
        _lib.call("vtx_jpeg_entropy_decode", ctypes.c_char_p(data), c_long(len(data)), ctypes.c_void_p(base + 2 * offs[i]),
                  c_long(infos[i]["blocks"] * 64), ctypes.c_void_p(base + 2 * (offs[i] + infos[i]["blocks"] * 64)))
        return i

    dev_all = torch.empty(offs[-1], dtype=torch.int16, device=device)
    planes = torch.empty(max(i["plane_bytes"] for i in infos), dtype=torch.uint8, device=device)   # shared: the images' kernels run in stream order
    out = []
    with ThreadPoolExecutor(max_workers=max(1, threads)) as pool:
        done = pool.map(work, range(n))
        for c0 in range(0, n, chunk):
            c1 = min(n, c0 + chunk)
            for _ in range(c0, c1):
                next(done)                                                # (raises here what a worker raised)
            dev_all[offs[c0]:offs[c1]].copy_(host[offs[c0]:offs[c1]], non_blocking=True)
            for i in range(c0, c1):
                info, data = infos[i], blobs[i]
                swap = apply_orientation and info["orientation"] >= 5
                H, W = (info["width"], info["height"]) if swap else (info["height"], info["width"])
                rgb = torch.empty(H, W, 3, dtype=torch.uint8, device=device)
                nb = info["blocks"] * 64
                _lib.call("vtx_jpeg_reconstruct", ctypes.c_char_p(data), c_long(len(data)), ptr(dev_all[offs[i]:offs[i] + nb]),
                          ptr(dev_all[offs[i] + nb:offs[i] + nb + QT]), ptr(planes), ptr(rgb), c_int(1 if apply_orientation else 0),
                          stream_ptr(rgb))
                out.append(rgb)
    if device.type == "cuda":
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        stage[1] = ev
    return out


def normalize_caption(caption: str) -> str:
    """The reference's caption normalisation (coco_captions.py:33-38): lowercase, NFKD, combining marks stripped."""
    caption = unicodedata.normalize("NFKD", caption.lower())
    return "".join(ch for ch in caption if not unicodedata.combining(ch))


class CocoCaptionsReader:
    """`CocoCaptionsDataset` of the reference (coco_captions.py:11-63) with the image decoded on the device:
    item = {"image_id": int, "image": uint8 (H, W, 3) RGB tensor on `device`, "captions": [str]} -- same keys, same instance
    order (annotation order of first appearance), same caption normalisation.  `read_bytes(idx)` / `instances` give the raw
    pieces to loaders that decode a whole batch from a thread pool."""

    def __init__(self, data_root: str, split: str, device="cuda"):
        image_dir = os.path.join(data_root, f"{split}2017")
        with open(os.path.join(data_root, "annotations", f"captions_{split}2017.json")) as fh:
            captions = json.load(fh)
        per_image: Dict[int, List[str]] = defaultdict(list)
        for ann in captions["annotations"]:
            per_image[ann["image_id"]].append(normalize_caption(ann["caption"]))
        paths = {im["id"]: os.path.join(image_dir, im["file_name"]) for im in captions["images"]}
        self.instances: List[Tuple[int, str, List[str]]] = [(i, paths[i], per_image[i]) for i in per_image.keys()]
        self.device = torch.device(device)

    def __len__(self) -> int:
        return len(self.instances)

    def read_bytes(self, idx: int) -> bytes:
        with open(self.instances[idx][1], "rb") as fh:
            return fh.read()

    def __getitem__(self, idx: int):
        image_id, _, captions = self.instances[idx]
        return {"image_id": image_id, "image": decode_jpeg(self.read_bytes(idx), self.device), "captions": captions}
