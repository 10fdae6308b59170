"""TEST INFRASTRUCTURE (never imported by the product path): CPU restatement of the baseline JPEG decoder the reference reaches
through `cv2.imread` (/root/reference/virtex/data/datasets/coco_captions.py:59) -- libjpeg / libjpeg-turbo with its default
settings.  The library is a third-party dependency absent from /root/reference (opencv-python, requirements.txt; OpenCV links
libjpeg-turbo); what is restated is its published algorithm:

  * ITU T.81 baseline entropy decoding (Huffman, DC prediction, restart intervals)         -- jdhuff.c
  * dequantisation + the accurate integer inverse DCT "islow" (13-bit constants,
    columns then rows, DESCALE roundings)                                                   -- jidctint.c jpeg_idct_islow
  * "fancy" triangle-filter chroma upsampling h2v1 / h2v2, replication when the chroma
    plane is at most two samples wide                                                        -- jdsample.c
  * YCbCr -> RGB with 16-bit fixed-point tables                                              -- jdcolor.c build_ycc_rgb_table
  * EXIF orientation (cv2.imread applies it; OpenCV >= 3.1)                                  -- modules/imgcodecs/src/exif.cpp

PINNED: tests/test_jpeg.py checks this file bit for bit against Pillow's libjpeg-turbo (`PIL.Image.open`, plus
`ImageOps.exif_transpose` for the orientation) on encoded images of every supported sampling / quality / restart setting, and
commits small golden fixtures (tests/golden/jpeg_*.npz) for machines without Pillow.  numpy for the block arithmetic, pure
Python for the bit stream (small images only)."""
import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14,
                   21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60,
                   61, 54, 47, 55, 62, 63])


class _Bits:
    def __init__(self, data, pos):
        self.d, self.p, self.acc, self.n = data, pos, 0, 0

    def _fill(self):
        b = 0
        if self.p < len(self.d):
            b = self.d[self.p]
            if b == 0xFF:
                if self.p + 1 < len(self.d) and self.d[self.p + 1] == 0:
                    self.p += 2
                else:
                    b = 0                                   # a marker: zeros until the caller resynchronises
            else:
                self.p += 1
        self.acc = (self.acc << 8) | b
        self.n += 8

    def bit(self):
        if self.n == 0:
            self._fill()
        self.n -= 1
        return (self.acc >> self.n) & 1

    def bits(self, k):
        v = 0
        for _ in range(k):
            v = (v << 1) | self.bit()
        return v


def _huff_table(counts, symbols):
    table, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(counts[length - 1]):
            table[(length, code)] = symbols[k]
            code += 1
            k += 1
        code <<= 1
    return table


def _decode_symbol(br, table):
    code = 0
    for length in range(1, 17):
        code = (code << 1) | br.bit()
        s = table.get((length, code))
        if s is not None:
            return s
    raise ValueError("corrupt Huffman code")


def _extend(v, t):
    return v - (1 << t) + 1 if v < (1 << (t - 1)) else v


def parse(data: bytes):
    d = data
    assert d[0] == 0xFF and d[1] == 0xD8
    i, qt, dc, ac, frame, restart, orientation = 2, {}, {}, {}, None, 0, 1
    while i + 4 <= len(d):
        if d[i] != 0xFF:
            i += 1
            continue
        m = d[i + 1]
        if m in (0xD8, 0x01) or 0xD0 <= m <= 0xD7 or m == 0xFF:
            i += 2 if m != 0xFF else 1
            continue
        length = (d[i + 2] << 8) | d[i + 3]
        s = d[i + 4: i + 2 + length]
        if m == 0xDB:
            o = 0
            while o < len(s):
                pq, tq = s[o] >> 4, s[o] & 15
                o += 1
                vals = [(s[o + 2 * k] << 8) | s[o + 2 * k + 1] for k in range(64)] if pq else list(s[o: o + 64])
                t = np.zeros(64, dtype=np.int64)
                t[ZIGZAG] = vals
                qt[tq] = t
                o += 128 if pq else 64
        elif m == 0xC4:
            o = 0
            while o + 17 <= len(s):
                tc, th = s[o] >> 4, s[o] & 15
                counts = list(s[o + 1: o + 17])
                n = sum(counts)
                (ac if tc else dc)[th] = _huff_table(counts, list(s[o + 17: o + 17 + n]))
                o += 17 + n
        elif m in (0xC0, 0xC1):
            H, W, nc = (s[1] << 8) | s[2], (s[3] << 8) | s[4], s[5]
            comps = [dict(id=s[6 + 3 * k], h=s[7 + 3 * k] >> 4, v=s[7 + 3 * k] & 15, tq=s[8 + 3 * k]) for k in range(nc)]
            frame = dict(W=W, H=H, comps=comps)
        elif 0xC2 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
            raise ValueError("not a baseline stream")
        elif m == 0xDD:
            restart = (s[0] << 8) | s[1]
        elif m == 0xE1:
            orientation = _exif_orientation(bytes(s))
        elif m == 0xDA:
            ns = s[0]
            for k in range(ns):
                frame["comps"][k]["td"], frame["comps"][k]["ta"] = s[2 + 2 * k] >> 4, s[2 + 2 * k] & 15
            return dict(frame=frame, qt=qt, dc=dc, ac=ac, restart=restart, orientation=orientation, scan=i + 2 + length)
        i += 2 + length
    raise ValueError("no scan")


def _exif_orientation(s: bytes) -> int:
    if len(s) < 14 or s[:6] != b"Exif\0\0":
        return 1
    t = s[6:]
    le = t[:2] == b"II"
    u16 = lambda o: int.from_bytes(t[o: o + 2], "little" if le else "big")      # noqa: E731
    u32 = lambda o: int.from_bytes(t[o: o + 4], "little" if le else "big")      # noqa: E731
    if u16(2) != 42:
        return 1
    ifd = u32(4)
    for k in range(u16(ifd)):
        e = ifd + 2 + 12 * k
        if u16(e) == 0x0112:
            v = u16(e + 8)
            return v if 1 <= v <= 8 else 1
    return 1


def coefficients(data: bytes, P):
    """quantised coefficients per component: int64 [block rows][block cols][64] in natural order"""
    fr = P["frame"]
    comps = fr["comps"]
    if len(comps) == 1:
        comps[0]["h"] = comps[0]["v"] = 1
    hmax, vmax = max(c["h"] for c in comps), max(c["v"] for c in comps)
    mcux, mcuy = -(-fr["W"] // (8 * hmax)), -(-fr["H"] // (8 * vmax))
    out = [np.zeros((mcuy * c["v"], mcux * c["h"], 64), dtype=np.int64) for c in comps]
    br = _Bits(data, P["scan"])
    pred = [0] * len(comps)
    left = P["restart"]
    for my in range(mcuy):
        for mx in range(mcux):
            if P["restart"] and left == 0:
                p = br.p
                while not (data[p] == 0xFF and 0xD0 <= data[p + 1] <= 0xD7):
                    p += 1
                br = _Bits(data, p + 2)
                pred = [0] * len(comps)
                left = P["restart"]
            for k, c in enumerate(comps):
                for v in range(c["v"]):
                    for h in range(c["h"]):
                        blk = out[k][my * c["v"] + v, mx * c["h"] + h]
                        t = _decode_symbol(br, P["dc"][c["td"]])
                        pred[k] += _extend(br.bits(t), t) if t else 0
                        blk[0] = pred[k]
                        kk = 1
                        while kk < 64:
                            rs = _decode_symbol(br, P["ac"][c["ta"]])
                            r, s = rs >> 4, rs & 15
                            if s == 0:
                                if r != 15:
                                    break
                                kk += 16
                                continue
                            kk += r
                            blk[ZIGZAG[kk]] = _extend(br.bits(s), s)
                            kk += 1
            if P["restart"]:
                left -= 1
    return out, hmax, vmax


CB, P1 = 13, 2
F = dict(a=2446, b=3196, c=4433, d=6270, e=7373, f=9633, g=12299, h=15137, i=16069, j=16819, k=20995, l=25172)


def _idct8(x):
    """x: (..., 8) int64 along the last axis -> the eight outputs before DESCALE (jidctint.c)"""
    z2, z3 = x[..., 2], x[..., 6]
    z1 = (z2 + z3) * F["c"]
    tmp2 = z1 + z3 * (-F["h"])
    tmp3 = z1 + z2 * F["d"]
    z2, z3 = x[..., 0], x[..., 4]
    tmp0, tmp1 = (z2 + z3) << CB, (z2 - z3) << CB
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = x[..., 7], x[..., 5], x[..., 3], x[..., 1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * F["f"]
    tmp0, tmp1, tmp2, tmp3 = tmp0 * F["a"], tmp1 * F["j"], tmp2 * F["l"], tmp3 * F["g"]
    z1, z2, z3, z4 = z1 * -F["e"], z2 * -F["k"], z3 * -F["i"] + z5, z4 * -F["b"] + z5
    tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
    return np.stack([tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3], axis=-1)


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def idct_plane(coef, qt):
    """coef: (bh, bw, 64) quantised -> uint8 plane (bh*8, bw*8)"""
    bh, bw, _ = coef.shape
    blk = (coef * qt).reshape(bh, bw, 8, 8)                      # [row][col]
    ws = _descale(_idct8(np.swapaxes(blk, -1, -2)), CB - P1)     # columns pass: last axis = rows of a column -> ws[col][row]
    ws = np.swapaxes(ws, -1, -2)                                 # [row][col]
    out = _descale(_idct8(ws), CB + P1 + 3) + 128                # rows pass
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out.transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8)


def _upsample(pl, W, H, hs, vs):
    """pl: the chroma plane cut to its REAL size (ceil(H/vs), ceil(W/hs)) -> (H, W) int64"""
    p = pl.astype(np.int64)
    ch, cw = p.shape
    if hs == 1 and vs == 1:
        return p[:H, :W]
    fancy = cw > 2
    if vs == 2:
        if not fancy:
            return np.repeat(np.repeat(p, 2, 0), 2, 1)[:H, :W]
        up = np.concatenate([p[:1], p[:-1]], 0)
        dn = np.concatenate([p[1:], p[-1:]], 0)
        rows = np.empty((2 * ch, cw), dtype=np.int64)
        rows[0::2] = 3 * p + up
        rows[1::2] = 3 * p + dn
        lf = np.concatenate([rows[:, :1], rows[:, :-1]], 1)
        rt = np.concatenate([rows[:, 1:], rows[:, -1:]], 1)
        out = np.empty((2 * ch, 2 * cw), dtype=np.int64)
        out[:, 0::2] = (3 * rows + lf + 8) >> 4
        out[:, 1::2] = (3 * rows + rt + 7) >> 4
        return out[:H, :W]
    if not fancy:
        return np.repeat(p, 2, 1)[:H, :W]
    lf = np.concatenate([p[:, :1], p[:, :-1]], 1)
    rt = np.concatenate([p[:, 1:], p[:, -1:]], 1)
    out = np.empty((ch, 2 * cw), dtype=np.int64)
    out[:, 0::2] = (3 * p + lf + 1) >> 2
    out[:, 1::2] = (3 * p + rt + 2) >> 2
    return out[:H, :W]


def decode(data: bytes, apply_orientation: bool = True) -> np.ndarray:
    """bytes of a baseline JPEG -> uint8 (H, W, 3) RGB, as cv2.imread + BGR2RGB of the reference returns it"""
    P = parse(data)
    coefs, hmax, vmax = coefficients(data, P)
    fr = P["frame"]
    W, H = fr["W"], fr["H"]
    planes = [idct_plane(c, P["qt"][comp["tq"]]) for c, comp in zip(coefs, fr["comps"])]
    Y = planes[0][:H, :W].astype(np.int64)
    if len(planes) == 1:
        rgb = np.stack([Y, Y, Y], -1)
    else:
        cw, ch = -(-W // hmax), -(-H // vmax)
        cb = _upsample(planes[1][:ch, :cw], W, H, hmax, vmax) - 128
        cr = _upsample(planes[2][:ch, :cw], W, H, hmax, vmax) - 128
        R = Y + ((91881 * cr + 32768) >> 16)
        B = Y + ((116130 * cb + 32768) >> 16)
        G = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
        rgb = np.stack([R, G, B], -1)
    rgb = np.clip(rgb, 0, 255).astype(np.uint8)
    o = P["orientation"] if apply_orientation else 1
    if o == 2: rgb = rgb[:, ::-1]
    elif o == 3: rgb = rgb[::-1, ::-1]
    elif o == 4: rgb = rgb[::-1]
    elif o == 5: rgb = rgb.transpose(1, 0, 2)
    elif o == 6: rgb = rgb.transpose(1, 0, 2)[:, ::-1]
    elif o == 7: rgb = rgb.transpose(1, 0, 2)[::-1, ::-1]
    elif o == 8: rgb = rgb.transpose(1, 0, 2)[::-1]
    return np.ascontiguousarray(rgb)
