"""JPEG decode (csrc/jpeg.hip, virtex_amd/jpeg.py) and the COCO Captions reader: SURVEY.md 8f row f2.

Pinning chain: Pillow's libjpeg-turbo (the same library family cv2.imread of the reference uses, default settings) ==
oracle/jpeg.py (bit for bit, test_oracle_*) == the HIP kernels through the C ABI (bit for bit, emulator and GPU).  The committed
fixture tests/golden/jpeg_cases.npz (made by tests/golden/make_jpeg_goldens.py from Pillow) keeps the first link checkable
where Pillow is absent."""
import io
import json
import os

import numpy as np
import pytest
import torch

from backends import BACKENDS, select
from oracle import jpeg as oj

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_cases.npz")


def _image(h, w, rng, smooth=True):
    if smooth:
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([127 + 100 * np.sin(xx / 7.0 + c) * np.cos(yy / 5.0 - c) for c in range(3)], -1) + rng.normal(0, 12, (h, w, 3))
    else:
        img = rng.integers(0, 256, (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def _encode(arr, **kw):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(arr).save(buf, "JPEG", **kw)
    return buf.getvalue()


def _pil_decode(data, orient=True):
    from PIL import Image, ImageOps
    im = Image.open(io.BytesIO(data))
    if orient:
        im = ImageOps.exif_transpose(im)
    return np.asarray(im.convert("RGB"))


def _with_orientation(data, orientation):
    """insert an EXIF APP1 segment carrying only the orientation tag behind SOI"""
    tiff = b"MM\x00\x2a\x00\x00\x00\x08" + b"\x00\x01" + b"\x01\x12\x00\x03\x00\x00\x00\x01" + bytes([0, orientation, 0, 0]) + b"\x00\x00\x00\x00"
    payload = b"Exif\x00\x00" + tiff
    return data[:2] + b"\xff\xe1" + (len(payload) + 2).to_bytes(2, "big") + payload + data[2:]


CASES = [(16, 16, 0, 75), (37, 53, 2, 75), (64, 48, 1, 90), (5, 3, 2, 50), (1, 1, 2, 75), (8, 9, 1, 30), (100, 131, 2, 95),
         (17, 4, 2, 75), (3, 5, 1, 75), (56, 72, 0, 20)]


def test_oracle_is_bitwise_equal_to_pillows_libjpeg():
    PIL = pytest.importorskip("PIL")  # noqa: F841
    rng = np.random.default_rng(0)
    for (h, w, sub, q) in CASES:
        for smooth in (True, False):
            data = _encode(_image(h, w, rng, smooth), quality=q, subsampling=sub)
            assert np.array_equal(oj.decode(data), _pil_decode(data)), (h, w, sub, q, smooth)
    grey = _encode(_image(33, 47, rng)[..., 0], quality=80)
    assert np.array_equal(oj.decode(grey), _pil_decode(grey))
    rst = _encode(_image(64, 80, rng), quality=80, optimize=True, restart_marker_blocks=2)
    assert oj.parse(rst)["restart"] == 2 and np.array_equal(oj.decode(rst), _pil_decode(rst))
    base = _encode(_image(24, 40, rng), quality=85, subsampling=2)
    for o in range(1, 9):                                  # EXIF orientation as cv2.imread / ImageOps.exif_transpose apply it
        data = _with_orientation(base, o)
        assert np.array_equal(oj.decode(data), _pil_decode(data)), o


def test_oracle_reproduces_the_committed_fixture():
    z = np.load(GOLDEN)
    n = int(z["count"])
    assert n >= 8
    for i in range(n):
        assert np.array_equal(oj.decode(z[f"jpeg_{i}"].tobytes()), z[f"rgb_{i}"]), i


@pytest.mark.parametrize("backend", BACKENDS)
def test_device_decoder_is_bitwise_equal_to_the_oracle(backend):
    from virtex_amd import jpeg as vj
    dev = select(backend)
    z = np.load(GOLDEN)
    for i in range(int(z["count"])):
        data = z[f"jpeg_{i}"].tobytes()
        got = vj.decode_jpeg(data, dev).cpu().numpy()
        assert got.shape == z[f"rgb_{i}"].shape and np.array_equal(got, z[f"rgb_{i}"]), i
        info = vj.jpeg_info(data)
        assert (info["width"], info["height"]) in ((got.shape[1], got.shape[0]), (got.shape[0], got.shape[1]))


@pytest.mark.parametrize("backend", BACKENDS)
def test_device_decoder_against_pillow_on_fresh_images(backend):
    pytest.importorskip("PIL")
    from virtex_amd import jpeg as vj
    dev = select(backend)
    rng = np.random.default_rng(5)
    sizes = [(224, 224), (333, 500), (61, 47)] if backend == "gpu" else [(40, 56), (23, 31)]
    for (h, w) in sizes:
        for sub in (0, 1, 2):
            data = _encode(_image(h, w, rng), quality=int(rng.integers(40, 96)), subsampling=sub,
                           restart_marker_blocks=int(rng.integers(0, 3)))
            assert np.array_equal(vj.decode_jpeg(data, dev).cpu().numpy(), _pil_decode(data)), (h, w, sub)
    data = _with_orientation(_encode(_image(24, 40, rng), quality=85), 6)
    assert np.array_equal(vj.decode_jpeg(data, dev).cpu().numpy(), _pil_decode(data))
    assert vj.decode_jpeg(data, dev, apply_orientation=False).shape == (24, 40, 3)


@pytest.mark.parametrize("backend", BACKENDS)
def test_streams_the_decoder_does_not_take_are_refused(backend):
    pytest.importorskip("PIL")
    from virtex_amd import _lib, jpeg as vj
    dev = select(backend)
    rng = np.random.default_rng(1)
    prog = _encode(_image(32, 32, rng), quality=80, progressive=True)
    with pytest.raises(_lib.VtxError):
        vj.decode_jpeg(prog, dev)
    with pytest.raises(_lib.VtxError):
        vj.decode_jpeg(b"not a jpeg at all", dev)
    good = _encode(_image(32, 32, rng), quality=80)
    with pytest.raises(_lib.VtxError):
        vj.decode_jpeg(good[: len(good) // 3], dev)          # truncated in the headers / tables


@pytest.mark.parametrize("backend", BACKENDS)
def test_coco_captions_reader_matches_the_reference_dataset(backend, tmp_path):
    """Annotation bookkeeping, caption normalisation and instance order of the reference's CocoCaptionsDataset
    (coco_captions.py:22-52), pixels from the device decoder."""
    pytest.importorskip("PIL")
    from virtex_amd import jpeg as vj
    dev = select(backend)
    rng = np.random.default_rng(2)
    root = tmp_path
    (root / "val2017").mkdir(); (root / "annotations").mkdir()
    images, anns, pixels = [], [], {}
    for k, (h, w) in enumerate([(32, 48), (40, 24), (17, 33)]):
        name = f"{k:012d}.jpg"
        data = _encode(_image(h, w, rng), quality=85)
        (root / "val2017" / name).write_bytes(data)
        pixels[100 + k] = _pil_decode(data)
        images.append({"id": 100 + k, "file_name": name})
    caps = [(101, "A Café on the LEFT side."), (100, "Two dogs  run"), (101, "Ünïcode açcents"), (102, "plain")]
    for j, (iid, c) in enumerate(caps):
        anns.append({"image_id": iid, "id": j, "caption": c})
    (root / "annotations" / "captions_val2017.json").write_text(json.dumps({"images": images, "annotations": anns}))
    ds = vj.CocoCaptionsReader(str(root), "val", device=dev)
    assert len(ds) == 3 and [i[0] for i in ds.instances] == [101, 100, 102]           # order of first appearance
    item = ds[0]
    assert item["image_id"] == 101 and item["captions"] == ["a cafe on the left side.", "unicode accents"]
    assert np.array_equal(item["image"].cpu().numpy(), pixels[101])
    if os.path.isdir("/root/reference"):                                               # the live reference class, where it exists
        import sys
        import types
        sys.modules.setdefault("cv2", types.ModuleType("cv2"))
        import importlib.util
        spec = importlib.util.spec_from_file_location("ref_coco", "/root/reference/virtex/data/datasets/coco_captions.py")
        mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
        ref = mod.CocoCaptionsDataset(str(root), "val")
        assert [(a, c) for a, _, c in ref.instances] == [(a, c) for a, _, c in ds.instances]


@pytest.mark.parametrize("backend", BACKENDS)
def test_batch_decode_on_host_threads_equals_single_decodes(backend):
    from virtex_amd import jpeg as vj
    dev = select(backend)
    z = np.load(GOLDEN)
    blobs = [z[f"jpeg_{i}"].tobytes() for i in range(int(z["count"]))]
    outs = vj.decode_jpeg_batch(blobs, dev, threads=4)
    assert len(outs) == len(blobs)
    for i, o in enumerate(outs):
        assert np.array_equal(o.cpu().numpy(), z[f"rgb_{i}"]), i


@pytest.mark.parametrize("backend", BACKENDS)
def test_oversubscribed_huffman_table_is_refused(backend):
    """A DHT whose code lengths over-subscribe the code space (three codes of length 1) used to index past the 9-bit lookahead
    table while it was being built (found by tests/fuzz/jpeg_host_fuzz.cpp under AddressSanitizer): refused now."""
    pytest.importorskip("PIL")
    from virtex_amd import _lib, jpeg as vj
    dev = select(backend)
    good = bytearray(_encode(_image(16, 16, np.random.default_rng(5)), quality=75))
    i = good.find(b"\xff\xc4")
    assert i > 0
    good[i + 5] = 3                                       # bits[0]: three codes of length 1
    with pytest.raises(_lib.VtxError, match="Huffman|DHT"):
        vj.decode_jpeg(bytes(good), dev)
    # a frame header asking for more than 2^28 pixels is refused before anything is allocated
    big = bytearray(_encode(_image(16, 16, np.random.default_rng(5)), quality=75))
    j = big.find(b"\xff\xc0")
    big[j + 5:j + 9] = bytes([0xFF, 0xFF, 0xFF, 0xFF])   # 65535 x 65535
    with pytest.raises(_lib.VtxError, match="pixel"):
        vj.jpeg_info(bytes(big))
    # a stream that ENDS with an empty SOS segment (FF DA 00 02) after a valid frame header: the component count would be read
    # one byte past the buffer (Python bytes hide it behind their trailing NUL, an mmap / numpy buffer does not)
    ok = _encode(_image(16, 16, np.random.default_rng(5)), quality=75)
    k = ok.find(b"\xff\xda")
    with pytest.raises(_lib.VtxError, match="empty SOS"):
        vj.jpeg_info(ok[:k] + b"\xff\xda\x00\x02")


def test_host_decoder_survives_mutated_streams_under_address_sanitizer(tmp_path):
    """The host half of the decoder (markers, tables, Huffman scan: the part that reads bytes from files) compiled with
    -fsanitize=address,undefined and driven with 40 000 mutated streams (byte flips, truncations, planted markers, insertions;
    baseline, optimised-table, grey and restart-interval seeds): every stream is decoded or refused, nothing is read or written
    out of bounds."""
    pytest.importorskip("PIL")
    import shutil
    import subprocess
    from PIL import Image
    from virtex_amd import build as vb
    select("emu")                                         # the emulator objects the harness links against are built
    clang = vb.HOST_CLANG
    if not os.path.exists(clang):
        pytest.skip("no host clang")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    obj = os.path.join(vb.EMU_DIR, "obj")
    flags = [f for f in vb.EMU_FLAGS if f != "-O2"] + ["-O1", "-I", os.path.join(root, "include"), "-fsanitize=address,undefined",
                                                        "-fno-omit-frame-pointer"]
    jo, ho, exe = str(tmp_path / "jpeg_asan.o"), str(tmp_path / "harness.o"), str(tmp_path / "harness")
    r = subprocess.run([clang] + flags + ["-c", os.path.join(root, "virtex_amd", "csrc", "jpeg.hip"), "-o", jo], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not available")
    assert r.returncode == 0, r.stderr[-2000:]
    subprocess.run([clang, "-x", "c++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-c",
                    os.path.join(root, "tests", "fuzz", "jpeg_host_fuzz.cpp"), "-o", ho], check=True)
    r = subprocess.run([clang, "-fsanitize=address,undefined", "-pthread", ho, jo, os.path.join(obj, "core.o"),
                        os.path.join(obj, "hipemu_rt.o"), "-o", exe], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("cannot link the sanitizer runtime: " + r.stderr[-300:])
    rng = np.random.default_rng(0)
    seeds = []
    for name, (h, w, kw) in {"a": (16, 16, dict(quality=75, subsampling=0)), "b": (37, 53, dict(quality=75, subsampling=2)),
                             "c": (64, 48, dict(quality=90, subsampling=1)), "d": (24, 40, dict(quality=60, subsampling=2, optimize=True)),
                             "f": (40, 56, dict(quality=85, subsampling=2, restart_marker_blocks=2))}.items():
        path = str(tmp_path / (name + ".jpg"))
        Image.fromarray(_image(h, w, rng)).save(path, "JPEG", **kw)
        seeds.append(path)
    grey = str(tmp_path / "e.jpg")
    Image.fromarray(_image(20, 28, rng)[:, :, 0].copy()).save(grey, "JPEG", quality=80)
    seeds.append(grey)
    sos = str(tmp_path / "g.jpg")                         # ends in an empty SOS segment: exact-size buffer, one byte short of s[0]
    blob = open(seeds[0], "rb").read()
    open(sos, "wb").write(blob[:blob.find(b"\xff\xda")] + b"\xff\xda\x00\x02")
    seeds.append(sos)
    probe = subprocess.run([exe, "1"] + seeds[:1], capture_output=True, text=True, timeout=120)
    if probe.returncode != 0 and "ERROR: AddressSanitizer" not in probe.stderr and "runtime error" not in probe.stderr:
        pytest.skip("the sanitizer runtime does not start in this environment: " + probe.stderr[-200:])
    r = subprocess.run([exe, "40000"] + seeds, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, (r.stderr[-3000:], r.stdout[-300:])
    decoded, refused = (int(x) for x in r.stdout.split()[1::2])
    assert decoded > 5000 and refused > 5000              # the mutations reach both the scan and the refusal paths


@pytest.mark.gpu
def test_large_frames_decode_bitwise_gpu():
    """camera-sized frames (the fixtures are thumbnails): 2048 x 1536 4:2:0 with restart intervals and an odd-sized 1999 x 1333
    4:2:2 frame -- 12 288 / 10 500 MCUs, plane offsets in the millions -- against Pillow, bit for bit"""
    pytest.importorskip("PIL")
    from virtex_amd import jpeg as vj
    dev = select("gpu")
    rng = np.random.default_rng(21)
    for (h, w, kw) in [(1536, 2048, dict(quality=88, subsampling=2, restart_marker_rows=4)), (1333, 1999, dict(quality=75, subsampling=1))]:
        data = _encode(_image(h, w, rng), **kw)
        out = vj.decode_jpeg(data, dev).cpu().numpy()
        assert out.shape == (h, w, 3) and np.array_equal(out, _pil_decode(data)), (h, w)
