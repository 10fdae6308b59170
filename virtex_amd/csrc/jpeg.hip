// JPEG decode for the input pipeline: what `cv2.imread(path)` + `cv2.cvtColor(BGR2RGB)` does per
// dataset item in the reference (/root/reference/virtex/data/datasets/coco_captions.py:59-60) -- libjpeg(-turbo)'s baseline
// decoder with its default settings (JDCT_ISLOW, fancy upsampling, YCbCr -> RGB) -- split the way the hardware wants it:
//
//   host    the entropy-coded segment is a serial bit stream: marker parsing + Huffman decoding into quantised DCT
//           coefficients (int16 [component][block row][block column][64], natural order), one call per image, thread-safe
//           (callers decode a batch from a thread pool; ctypes releases the GIL);
//   device  everything that is per-block / per-pixel arithmetic, bit-exact with libjpeg's integer code:
//           jpeg_idct_kernel     dequantisation + the "islow" inverse DCT (jidctint.c: 13-bit constants, columns then rows,
//                                the same DESCALE roundings) -> uint8 component planes;
//           jpeg_color_kernel    "fancy" (triangle-filter) chroma upsampling for 4:2:0 / 4:2:2 (jdsample.c), 4:4:4 and grey
//                                passthrough, YCbCr -> RGB with jdcolor.c's 16-bit fixed-point tables, EXIF orientation
//                                (cv2.imread applies it) -> uint8 [H][W][3] RGB: the layout vtx_image_augment_u8 reads.
//
// Third-party algorithm restated (absent from /root/reference: it arrives through opencv-python, requirements.txt): the
// Independent JPEG Group's libjpeg 6b / libjpeg-turbo decoder; parity is pinned against Pillow's bundled libjpeg-turbo on
// encoded test images (tests/test_jpeg.py: bit-exact), the CPU restatement lives in oracle/jpeg.py.
// Not taken (VTX_ERR_SHAPE, the caller decides): progressive / arithmetic / lossless / 12-bit streams, CMYK / YCCK,
// sampling factors other than luma {1,2}x{1,2} over 1x1 chroma.
#include <stdlib.h>
#include <string.h>

#include "vtx_common.h"

namespace {

// ------------------------------------------------------------------------------------------------ host: the bit stream
const unsigned char kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    // canonical decoding (ITU T.81 F.2.2.3): per code length the largest code and the offset of its first symbol, plus a
    // 9-bit lookahead table (symbol << 4 | length, 0 = longer code)
    int maxcode[18];
    int valptr[17];
    unsigned char vals[256];
    unsigned short look[512];
    bool present = false;
};

bool build_huff(Huff& h, const unsigned char* bits /*[16]*/, const unsigned char* vals, int nvals) {
    int code = 0, k = 0;
    memset(h.look, 0, sizeof(h.look));
    for (int l = 1; l <= 16; ++l) {
        h.valptr[l] = k - code;
        for (int i = 0; i < bits[l - 1]; ++i, ++k, ++code) {
            if (k >= nvals || k >= 256) return false;
            if (code >= (1 << l)) return false;          // over-subscribed code lengths (a corrupt table): no such code of length l
            if (l <= 9) {
                const int first = code << (9 - l);
                for (int f = 0; f < (1 << (9 - l)); ++f) h.look[first + f] = (unsigned short)((vals[k] << 4) | l);
            }
        }
        h.maxcode[l] = bits[l - 1] ? code - 1 : -1;
        if (code > (1 << l)) return false;
        code <<= 1;
    }
    h.maxcode[17] = 0x7fffffff;
    memcpy(h.vals, vals, nvals < 256 ? nvals : 256);
    h.present = true;
    return true;
}

struct BitReader {
    const unsigned char* p; const unsigned char* end;
    unsigned long long acc = 0; int n = 0;
    bool marker_hit = false;
    void fill() {
        while (n <= 56) {
            int b = 0;
            if (!marker_hit && p < end) {
                b = *p;
                if (b == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) p += 2;                       // stuffed byte
                    else { marker_hit = true; b = 0; }                              // a marker: feed zeros, the MCU loop resynchronises
                } else ++p;
            }
            acc |= (unsigned long long)b << (56 - n);
            n += 8;
        }
    }
    inline int peek(int k) { if (n < k) fill(); return (int)(acc >> (64 - k)); }
    inline void skip(int k) { acc <<= k; n -= k; }
    inline int get(int k) { if (k == 0) return 0; const int v = peek(k); skip(k); return v; }
    void reset() { acc = 0; n = 0; marker_hit = false; }
};

inline int decode_symbol(BitReader& br, const Huff& h) {
    const int look = h.look[br.peek(9)];
    if (look) { br.skip(look & 15); return look >> 4; }
    int code = br.peek(16), l = 10;
    for (; l <= 16; ++l)
        if ((code >> (16 - l)) <= h.maxcode[l]) break;
    if (l > 16) return -1;
    br.skip(l);
    return h.vals[(code >> (16 - l)) + h.valptr[l]];
}
inline int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

struct Component { int id, h, v, tq, td, ta; int bw, bh; long off; int pred; };

struct Parsed {
    int W = 0, H = 0, ncomp = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0, restart = 0, orientation = 1;
    Component c[3];
    unsigned short qt[4][64];       // natural order
    bool qt_present[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    const unsigned char* scan = nullptr; const unsigned char* end = nullptr;
    long total_blocks = 0;
};

inline int rd16(const unsigned char* p) { return (p[0] << 8) | p[1]; }

// EXIF orientation (APP1 "Exif\0\0" + TIFF header + IFD0 tag 0x0112): cv2.imread rotates / mirrors by it
int exif_orientation(const unsigned char* p, int len) {
    if (len < 14 || memcmp(p, "Exif\0\0", 6) != 0) return 1;
    const unsigned char* t = p + 6; const int n = len - 6;
    const bool le = t[0] == 'I' && t[1] == 'I';
    if (!le && !(t[0] == 'M' && t[1] == 'M')) return 1;
    auto u16 = [&](int o) { return le ? (t[o] | (t[o + 1] << 8)) : ((t[o] << 8) | t[o + 1]); };
    auto u32 = [&](int o) { return le ? (unsigned)(t[o] | (t[o + 1] << 8) | (t[o + 2] << 16) | ((unsigned)t[o + 3] << 24))
                                      : (unsigned)(((unsigned)t[o] << 24) | (t[o + 1] << 16) | (t[o + 2] << 8) | t[o + 3]); };
    if (n < 8 || u16(2) != 42) return 1;
    const unsigned ifd = u32(4);
    if (ifd + 2 > (unsigned)n) return 1;
    const int cnt = u16((int)ifd);
    for (int i = 0; i < cnt; ++i) {
        const unsigned e = ifd + 2 + 12u * i;
        if (e + 12 > (unsigned)n) break;
        if (u16((int)e) == 0x0112) {
            const int v = u16((int)e + 8);
            return (v >= 1 && v <= 8) ? v : 1;
        }
    }
    return 1;
}

int parse(const unsigned char* d, long n, Parsed& P) {
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) { vtx_set_error("jpeg: no SOI marker"); return VTX_ERR_ARG; }
    long i = 2;
    bool have_sof = false;
    while (i + 4 <= n) {
        if (d[i] != 0xFF) { ++i; continue; }
        const int m = d[i + 1];
        if (m == 0xFF) { ++i; continue; }
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) { i += 2; continue; }
        if (m == 0xD9) break;
        const int len = rd16(d + i + 2);
        if (len < 2 || i + 2 + len > n) { vtx_set_error("jpeg: truncated segment %02x", m); return VTX_ERR_ARG; }
        const unsigned char* s = d + i + 4; const int sl = len - 2;
        if (m == 0xDB) {                                                            // DQT
            int o = 0;
            while (o < sl) {
                const int pq = s[o] >> 4, tq = s[o] & 15; ++o;
                if (tq > 3 || o + (pq ? 128 : 64) > sl) { vtx_set_error("jpeg: bad DQT"); return VTX_ERR_ARG; }
                for (int k = 0; k < 64; ++k) { P.qt[tq][kZigzag[k]] = (unsigned short)(pq ? rd16(s + o + 2 * k) : s[o + k]); }
                o += pq ? 128 : 64;
                P.qt_present[tq] = true;
            }
        } else if (m == 0xC4) {                                                     // DHT
            int o = 0;
            while (o + 17 <= sl) {
                const int tc = s[o] >> 4, th = s[o] & 15;
                int cnt = 0;
                for (int k = 0; k < 16; ++k) cnt += s[o + 1 + k];
                if (th > 3 || tc > 1 || o + 17 + cnt > sl || cnt > 256) { vtx_set_error("jpeg: bad DHT"); return VTX_ERR_ARG; }
                if (!build_huff(tc ? P.ac[th] : P.dc[th], s + o + 1, s + o + 17, cnt)) { vtx_set_error("jpeg: bad Huffman table"); return VTX_ERR_ARG; }
                o += 17 + cnt;
            }
        } else if (m == 0xC0 || m == 0xC1) {                                        // baseline / extended sequential, Huffman
            if (sl < 6 || s[0] != 8) { vtx_set_error("jpeg: only 8-bit samples"); return VTX_ERR_SHAPE; }
            P.H = rd16(s + 1); P.W = rd16(s + 3); P.ncomp = s[5];
            if ((P.ncomp != 1 && P.ncomp != 3) || sl < 6 + 3 * P.ncomp || P.W <= 0 || P.H <= 0) {
                vtx_set_error("jpeg: %d components (grey or YCbCr only)", P.ncomp); return VTX_ERR_SHAPE; }
            for (int k = 0; k < P.ncomp; ++k) {
                P.c[k].id = s[6 + 3 * k]; P.c[k].h = s[7 + 3 * k] >> 4; P.c[k].v = s[7 + 3 * k] & 15; P.c[k].tq = s[8 + 3 * k];
                if (P.c[k].tq > 3) { vtx_set_error("jpeg: bad quantisation table index"); return VTX_ERR_ARG; }
            }
            have_sof = true;
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            vtx_set_error("jpeg: SOF%d streams (progressive / lossless / arithmetic) are not taken", m - 0xC0); return VTX_ERR_SHAPE;
        } else if (m == 0xDD) {
            if (sl >= 2) P.restart = rd16(s);
        } else if (m == 0xE1) {
            P.orientation = exif_orientation(s, sl);
        } else if (m == 0xEE) {                                                     // Adobe: transform 0 with 3 components = RGB, not YCbCr
            if (sl >= 12 && memcmp(s, "Adobe", 5) == 0 && s[11] == 0 && P.ncomp == 3) { vtx_set_error("jpeg: Adobe RGB streams are not taken"); return VTX_ERR_SHAPE; }
        } else if (m == 0xDA) {                                                     // SOS
            if (!have_sof) { vtx_set_error("jpeg: SOS before SOF"); return VTX_ERR_ARG; }
            if (sl < 1) { vtx_set_error("jpeg: empty SOS segment"); return VTX_ERR_ARG; }   // (a stream ending in FF DA 00 02: s == d + n, s[0] is past the buffer)
            const int ns = s[0];
            if (ns != P.ncomp || sl < 1 + 2 * ns + 3) { vtx_set_error("jpeg: non-interleaved scans are not taken"); return VTX_ERR_SHAPE; }
            for (int k = 0; k < ns; ++k) {
                int ci = -1;
                for (int q = 0; q < P.ncomp; ++q) if (P.c[q].id == s[1 + 2 * k]) ci = q;
                if (ci != k) { vtx_set_error("jpeg: scan component order"); return VTX_ERR_SHAPE; }
                P.c[k].td = s[2 + 2 * k] >> 4; P.c[k].ta = s[2 + 2 * k] & 15;
                if (P.c[k].td > 3 || P.c[k].ta > 3 || !P.dc[P.c[k].td].present || !P.ac[P.c[k].ta].present || !P.qt_present[P.c[k].tq]) {
                    vtx_set_error("jpeg: scan refers to a missing table"); return VTX_ERR_ARG; }
            }
            P.scan = d + i + 2 + len; P.end = d + n;
            break;
        }
        i += 2 + len;
    }
    if (!have_sof || !P.scan) { vtx_set_error("jpeg: no frame / scan found"); return VTX_ERR_ARG; }
    if (P.ncomp == 1) { P.c[0].h = P.c[0].v = 1; }
    P.hmax = P.vmax = 1;
    for (int k = 0; k < P.ncomp; ++k) { if (P.c[k].h > P.hmax) P.hmax = P.c[k].h; if (P.c[k].v > P.vmax) P.vmax = P.c[k].v; }
    if (P.ncomp == 3) {
        const bool ok = (P.c[0].h == 1 || P.c[0].h == 2) && (P.c[0].v == 1 || P.c[0].v == 2) && P.c[1].h == 1 && P.c[1].v == 1 &&
                        P.c[2].h == 1 && P.c[2].v == 1 && !(P.c[0].h == 1 && P.c[0].v == 2);
        if (!ok) { vtx_set_error("jpeg: sampling %dx%d,%dx%d,%dx%d is not taken (4:4:4, 4:2:2, 4:2:0)", P.c[0].h, P.c[0].v, P.c[1].h, P.c[1].v, P.c[2].h, P.c[2].v); return VTX_ERR_SHAPE; }
    }
    // info[] carries block and byte counts as int: 2^28 pixels (16 384 x 16 384) is the largest frame taken (cv2.imread's own
    // limit, CV_IO_MAX_IMAGE_PIXELS, is 2^30); a corrupt header must not turn into a multi-gigabyte allocation request either
    if ((long)P.W * P.H > (1l << 28)) { vtx_set_error("jpeg: %d x %d pixels exceed the 2^28-pixel limit", P.W, P.H); return VTX_ERR_SHAPE; }
    P.mcux = vtx_cdiv(P.W, 8 * P.hmax); P.mcuy = vtx_cdiv(P.H, 8 * P.vmax);
    long off = 0;
    for (int k = 0; k < P.ncomp; ++k) {
        P.c[k].bw = P.mcux * P.c[k].h; P.c[k].bh = P.mcuy * P.c[k].v; P.c[k].off = off;
        off += (long)P.c[k].bw * P.c[k].bh;
    }
    P.total_blocks = off;
    return VTX_OK;
}

int decode_scan(Parsed& P, short* coef /*[total_blocks][64], zeroed*/) {
    BitReader br; br.p = P.scan; br.end = P.end;
    for (int k = 0; k < P.ncomp; ++k) P.c[k].pred = 0;
    int until_restart = P.restart, next_rst = 0;
    for (int my = 0; my < P.mcuy; ++my)
        for (int mx = 0; mx < P.mcux; ++mx) {
            if (P.restart && until_restart == 0) {
                // byte-align, expect RSTn
                br.reset();
                const unsigned char* q = br.p;
                while (q + 1 < P.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
                if (q + 1 >= P.end) { vtx_set_error("jpeg: restart marker missing"); return VTX_ERR_ARG; }
                (void)next_rst;
                br.p = q + 2;
                for (int k = 0; k < P.ncomp; ++k) P.c[k].pred = 0;
                until_restart = P.restart;
            }
            for (int k = 0; k < P.ncomp; ++k) {
                Component& c = P.c[k];
                const Huff& hd = P.dc[c.td]; const Huff& ha = P.ac[c.ta];
                for (int v = 0; v < c.v; ++v)
                    for (int h = 0; h < c.h; ++h) {
                        short* blk = coef + (c.off + (long)(my * c.v + v) * c.bw + (mx * c.h + h)) * 64;
                        int t = decode_symbol(br, hd);
                        if (t < 0 || t > 11) { vtx_set_error("jpeg: corrupt DC code"); return VTX_ERR_ARG; }
                        const int diff = t ? extend(br.get(t), t) : 0;
                        c.pred += diff;
                        blk[0] = (short)c.pred;
                        for (int kk = 1; kk < 64;) {
                            const int rs = decode_symbol(br, ha);
                            if (rs < 0) { vtx_set_error("jpeg: corrupt AC code"); return VTX_ERR_ARG; }
                            const int r = rs >> 4, s2 = rs & 15;
                            if (s2 == 0) {
                                if (r != 15) break;
                                kk += 16;
                                continue;
                            }
                            kk += r;
                            if (kk > 63) { vtx_set_error("jpeg: corrupt AC run"); return VTX_ERR_ARG; }
                            blk[kZigzag[kk]] = (short)extend(br.get(s2), s2);
                            ++kk;
                        }
                    }
            }
            if (P.restart) --until_restart;
        }
    return VTX_OK;
}

// ------------------------------------------------------------------------------------------------ device: IDCT
// jidctint.c (jpeg_idct_islow): CONST_BITS = 13, PASS1_BITS = 2
constexpr int CB = 13, P1 = 2;
constexpr int F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373,
              F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069, F_2_053119869 = 16819,
              F_2_562915447 = 20995, F_3_072711026 = 25172;

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// one 8-point pass: in[0..7] (already dequantised / workspace values), out[0..7] before the final DESCALE
__device__ __forceinline__ void idct8(const int* in, int* o) {
    int z2 = in[2], z3 = in[6];
    int z1 = (z2 + z3) * F_0_541196100;
    int tmp2 = z1 + z3 * (-F_1_847759065);
    int tmp3 = z1 + z2 * F_0_765366865;
    z2 = in[0]; z3 = in[4];
    int tmp0 = (z2 + z3) << CB;
    int tmp1 = (z2 - z3) << CB;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * F_1_175875602;
    tmp0 *= F_0_298631336; tmp1 *= F_2_053119869; tmp2 *= F_3_072711026; tmp3 *= F_1_501321110;
    z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    o[0] = tmp10 + tmp3; o[7] = tmp10 - tmp3; o[1] = tmp11 + tmp2; o[6] = tmp11 - tmp2;
    o[2] = tmp12 + tmp1; o[5] = tmp12 - tmp1; o[3] = tmp13 + tmp0; o[4] = tmp13 - tmp0;
}

struct IdctPlane { long coef_off; int bw, bh, qt, stride; long plane_off; };     // blocks of one component -> its padded plane
struct IdctArgs { IdctPlane pl[3]; int ncomp; long total_blocks; };

// 64 threads = 8 blocks x 8 lanes: lane (b, r) loads ROW r of block b (16 contiguous bytes), the columns pass reads the
// transposed view from LDS, the rows pass the transposed workspace -- libjpeg's order (columns first) and roundings
__global__ __launch_bounds__(64) void jpeg_idct_kernel(const short* __restrict__ coef, const unsigned short* __restrict__ qts /*[4][64]*/,
                                                         unsigned char* __restrict__ planes, IdctArgs a) {
    __shared__ int ws[8][8][9];
    const int lane = threadIdx.x, b = lane >> 3, r = lane & 7;
    const long blk = (long)blockIdx.x * 8 + b;
    int comp = 0;
    for (int k = 1; k < a.ncomp; ++k) if (blk >= a.pl[k].coef_off) comp = k;
    const IdctPlane& p = a.pl[comp];
    const bool live = blk < a.total_blocks;
    int v[8];
    if (live) {
        const uint4 raw = *reinterpret_cast<const uint4*>(coef + blk * 64 + r * 8);
        const unsigned short* q = qts + p.qt * 64 + r * 8;
        const unsigned u[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = (int)(short)(u[e] & 0xffffu) * (int)q[2 * e];
            v[2 * e + 1] = (int)(short)(u[e] >> 16) * (int)q[2 * e + 1];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) ws[b][r][e] = v[e];                  // dequantised block, [row][col]
    __syncthreads();
    int col[8], o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) col[e] = ws[b][e][r];                // this lane's COLUMN r
    idct8(col, o);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) ws[b][e][r] = descale(o[e], CB - P1);   // workspace [row][col]
    __syncthreads();
    int row[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) row[e] = ws[b][r][e];                // this lane's ROW r
    idct8(row, o);
    if (!live) return;
    unsigned char px[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int s = descale(o[e], CB + P1 + 3) + 128;                    // range_limit: centre + clamp
        px[e] = (unsigned char)(s < 0 ? 0 : (s > 255 ? 255 : s));
    }
    const long local = blk - p.coef_off;
    const int by = (int)(local / p.bw), bx = (int)(local % p.bw);
    unsigned char* dst = planes + p.plane_off + (long)(by * 8 + r) * p.stride + bx * 8;
    *reinterpret_cast<uint2*>(dst) = make_uint2(px[0] | (px[1] << 8) | (px[2] << 16) | ((unsigned)px[3] << 24),
                                                px[4] | (px[5] << 8) | (px[6] << 16) | ((unsigned)px[7] << 24));
}

// ------------------------------------------------------------------------------------------------ device: upsample + colour
struct ColorArgs {
    int W, H, ncomp, hs, vs;                 // luma sampling over chroma (1 or 2 each)
    int ystride, cstride, cw, ch;            // plane strides; REAL chroma size (downsampled_width / height)
    long yoff, cboff, croff;
    int fancy_h, fancy_v;                    // libjpeg takes the triangle filter only when the chroma plane is wider than 2 samples
    int orientation, outW, outH;
};

__device__ __forceinline__ int chroma_at(const unsigned char* __restrict__ pl, const ColorArgs& a, int x, int y) {
    if (a.hs == 1 && a.vs == 1) return pl[(long)y * a.cstride + x];
    const int j = x >> 1, cwm = a.cw - 1;
    if (a.vs == 1) {                                                  // h2v1
        const unsigned char* row = pl + (long)y * a.cstride;
        if (!a.fancy_h) return row[j];
        const int cur = row[j];
        if (x & 1) return (3 * cur + row[j < cwm ? j + 1 : j] + 2) >> 2;
        return (3 * cur + row[j > 0 ? j - 1 : 0] + 1) >> 2;
    }
    const int i = y >> 1;                                             // h2v2
    if (!a.fancy_h) return pl[(long)i * a.cstride + j];
    const int nb = (y & 1) ? (i < a.ch - 1 ? i + 1 : i) : (i > 0 ? i - 1 : 0);
    const unsigned char* r0 = pl + (long)i * a.cstride;
    const unsigned char* r1 = pl + (long)nb * a.cstride;
    const int cur = 3 * r0[j] + r1[j];
    if (x & 1) { const int jn = j < cwm ? j + 1 : j; return (3 * cur + (3 * r0[jn] + r1[jn]) + 7) >> 4; }
    const int jp = j > 0 ? j - 1 : 0;
    return (3 * cur + (3 * r0[jp] + r1[jp]) + 8) >> 4;
}

__global__ __launch_bounds__(256) void jpeg_color_kernel(const unsigned char* __restrict__ planes, unsigned char* __restrict__ rgb, ColorArgs a) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)a.W * a.H) return;
    const int y = (int)(idx / a.W), x = (int)(idx % a.W);
    const int Y = planes[a.yoff + (long)y * a.ystride + x];
    int R = Y, G = Y, B = Y;
    if (a.ncomp == 3) {
        const int cb = chroma_at(planes + a.cboff, a, x, y) - 128, cr = chroma_at(planes + a.croff, a, x, y) - 128;
        // jdcolor.c build_ycc_rgb_table: SCALEBITS = 16, FIX(1.40200) = 91881, FIX(1.77200) = 116130, FIX(0.71414) = 46802,
        // FIX(0.34414) = 22554, ONE_HALF = 32768; arithmetic right shifts
        R = Y + ((91881 * cr + 32768) >> 16);
        B = Y + ((116130 * cb + 32768) >> 16);
        G = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
        R = R < 0 ? 0 : (R > 255 ? 255 : R); G = G < 0 ? 0 : (G > 255 ? 255 : G); B = B < 0 ? 0 : (B > 255 ? 255 : B);
    }
    // EXIF orientation 1..8 -> destination (ox, oy) in the outW x outH image
    int ox = x, oy = y;
    switch (a.orientation) {
        case 2: ox = a.W - 1 - x; break;
        case 3: ox = a.W - 1 - x; oy = a.H - 1 - y; break;
        case 4: oy = a.H - 1 - y; break;
        case 5: ox = y; oy = x; break;
        case 6: ox = a.H - 1 - y; oy = x; break;
        case 7: ox = a.H - 1 - y; oy = a.W - 1 - x; break;
        case 8: ox = y; oy = a.W - 1 - x; break;
        default: break;
    }
    unsigned char* dst = rgb + ((long)oy * a.outW + ox) * 3;
    dst[0] = (unsigned char)R; dst[1] = (unsigned char)G; dst[2] = (unsigned char)B;
}

}  // namespace

// ---- C ABI ------------------------------------------------------------------------------------------------------------
// Step 1 (host): header only.  info[0..7] = {width, height, components, luma h sampling, luma v sampling, orientation,
// coefficient blocks (64 int16 each), plane bytes}; width / height are those of the STORED frame (before orientation).
extern "C" int vtx_jpeg_info(const void* data, long nbytes, int* info) {
    VTX_CHECK(data && info && nbytes > 0, VTX_ERR_ARG, "jpeg_info: null pointer");
    Parsed P;
    const int rc = parse((const unsigned char*)data, nbytes, P);
    if (rc) return rc;
    long plane = 0;
    for (int k = 0; k < P.ncomp; ++k) plane += (long)P.c[k].bw * 8 * P.c[k].bh * 8;
    info[0] = P.W; info[1] = P.H; info[2] = P.ncomp; info[3] = P.c[0].h; info[4] = P.c[0].v; info[5] = P.orientation;
    info[6] = (int)P.total_blocks; info[7] = (int)plane;
    return VTX_OK;
}

// Step 2 (host): entropy decoding.  coef: host buffer of info[6] * 64 int16 (pinned memory makes the upload asynchronous),
// zero-filled here; qt: host buffer of 4 * 64 uint16 (natural order).
extern "C" int vtx_jpeg_entropy_decode(const void* data, long nbytes, short* coef, long coef_elems, unsigned short* qt) {
    VTX_CHECK(data && coef && qt && nbytes > 0, VTX_ERR_ARG, "jpeg_entropy_decode: null pointer");
    Parsed P;
    int rc = parse((const unsigned char*)data, nbytes, P);
    if (rc) return rc;
    VTX_CHECK(coef_elems >= P.total_blocks * 64, VTX_ERR_WORKSPACE, "jpeg_entropy_decode: coefficient buffer holds %ld, %ld needed", coef_elems, P.total_blocks * 64);
    memset(coef, 0, (size_t)P.total_blocks * 64 * sizeof(short));
    memcpy(qt, P.qt, sizeof(P.qt));
    return decode_scan(P, coef);
}

// Step 3 (device): coefficients -> RGB.  coef_dev / qt_dev: the buffers of step 2 on the device; planes_dev: info[7] bytes
// of scratch; rgb_dev: [outH][outW][3] uint8 with (outW, outH) = (W, H), or (H, W) for the transposing orientations 5-8 when
// apply_orientation != 0.  Geometry is re-derived from the stream's header (`data`: only the markers are read).
extern "C" int vtx_jpeg_reconstruct(const void* data, long nbytes, const short* coef_dev, const unsigned short* qt_dev,
                                    unsigned char* planes_dev, unsigned char* rgb_dev, int apply_orientation, void* stream) {
    VTX_CHECK(data && coef_dev && qt_dev && planes_dev && rgb_dev, VTX_ERR_ARG, "jpeg_reconstruct: null pointer");
    Parsed P;
    const int rc = parse((const unsigned char*)data, nbytes, P);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    IdctArgs ia{};
    ia.ncomp = P.ncomp; ia.total_blocks = P.total_blocks;
    long poff = 0;
    for (int k = 0; k < P.ncomp; ++k) {
        ia.pl[k] = IdctPlane{P.c[k].off, P.c[k].bw, P.c[k].bh, P.c[k].tq, P.c[k].bw * 8, poff};
        poff += (long)P.c[k].bw * 8 * P.c[k].bh * 8;
    }
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3(vtx_cdiv(P.total_blocks, 8)), dim3(64), 0, st, coef_dev, qt_dev, planes_dev, ia);
    ColorArgs ca{};
    ca.W = P.W; ca.H = P.H; ca.ncomp = P.ncomp; ca.hs = P.hmax; ca.vs = P.vmax;
    ca.ystride = ia.pl[0].stride; ca.yoff = ia.pl[0].plane_off;
    if (P.ncomp == 3) {
        ca.cstride = ia.pl[1].stride; ca.cboff = ia.pl[1].plane_off; ca.croff = ia.pl[2].plane_off;
        ca.cw = vtx_cdiv((long)P.W * P.c[1].h, P.hmax); ca.ch = vtx_cdiv((long)P.H * P.c[1].v, P.vmax);
        ca.fancy_h = ca.cw > 2; ca.fancy_v = 1;
    }
    ca.orientation = apply_orientation ? P.orientation : 1;
    const bool swap = ca.orientation >= 5;
    ca.outW = swap ? P.H : P.W; ca.outH = swap ? P.W : P.H;
    hipLaunchKernelGGL(jpeg_color_kernel, dim3(vtx_cdiv((long)P.W * P.H, 256)), dim3(256), 0, st, planes_dev, rgb_dev, ca);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}
