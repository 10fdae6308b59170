"""JPEG decode throughput on the box: COCO-sized images (640 x 480, 4:2:0, quality 90) encoded with Pillow, decoded by
virtex_amd.jpeg.decode_jpeg_batch with 1 / 8 / 32 host threads, against Pillow's own decoder on one thread and on the same pool.
    python tools/bench_jpeg.py [--images 256]"""
import argparse
import io
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from virtex_amd import jpeg as vj  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=256)
    a = ap.parse_args()
    from PIL import Image
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:480, 0:640]
    blobs = []
    for i in range(16):
        img = np.stack([127 + 90 * np.sin(xx / (9.0 + i) + c) * np.cos(yy / (6.0 + c) - i) for c in range(3)], -1) + rng.normal(0, 10, (480, 640, 3))
        buf = io.BytesIO(); Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(buf, "JPEG", quality=90, subsampling=2)
        blobs.append(buf.getvalue())
    blobs = [blobs[i % 16] for i in range(a.images)]
    dev = torch.device("cuda", 0)
    ref = np.asarray(Image.open(io.BytesIO(blobs[0])).convert("RGB"))
    got = vj.decode_jpeg(blobs[0], dev).cpu().numpy()
    print(f"{a.images} images of 640x480 4:2:0 q90, {sum(map(len, blobs)) / a.images / 1e3:.0f} KB each; bit-exact with Pillow: {np.array_equal(ref, got)}; "
          f"host CPUs {os.cpu_count()}")
    for threads in (1, 2, 8, 16, 32):
        vj.decode_jpeg_batch(blobs[:16], dev, threads=threads)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = vj.decode_jpeg_batch(blobs, dev, threads=threads)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"virtex_amd decode_jpeg_batch, {threads:2d} host threads: {a.images / dt:8.0f} images/s")
        del out
    t0 = time.perf_counter()
    for b in blobs[:64]:
        np.asarray(Image.open(io.BytesIO(b)).convert("RGB"))
    print(f"Pillow (libjpeg-turbo), one thread, host memory:      {64 / (time.perf_counter() - t0):8.0f} images/s")
    # device part alone: coefficients already on the device
    info = vj.jpeg_info(blobs[0])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    outs = vj.decode_jpeg_batch(blobs[:32], dev, threads=8)
    torch.cuda.synchronize()
    print(f"(device kernels: idct over {info['blocks']} blocks + colour conversion per image; see rocprofv3 for their time)")


if __name__ == "__main__":
    main()
