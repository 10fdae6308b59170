// Image augmentation on the device: uint8 HWC decoder output -> random-resized crop (bilinear) -> horizontal flip ->
// colour jitter (brightness / contrast / saturation / hue in a per-image order) -> ImageNet normalisation -> the stem's
// NHWC layout in the compute dtype (channels zero-padded, optional zero frame).  Replaces the CPU pipeline
//   alb.RandomResizedCrop / T.HorizontalFlip / alb.ColorJitter / alb.Normalize / np.transpose
// of /root/reference/virtex/data/transforms.py:5-97 and virtex/factories.py:132-154 (ImageTransformsFactory), and the
// validation pipeline alb.SmallestMaxSize + CenterCrop + Normalize (transforms.py:91-97) as the special case
// crop window = the centre square, no flip, identity jitter.
// Arithmetic: float throughout, rounded to the uint8 grid after the resize and after every colour operation like the
// uint8 images of the CPU pipeline; formulas = albumentations 1.x' torchvision-style functionals (gray = .299R+.587G+.114B).
// cv2's fixed-point resize / HSV tables are NOT reproduced bit for bit (cv2 is absent here: parity unpinned).
#include "vtx_common.h"

namespace {

struct Rgb { float r, g, b; };
__device__ __forceinline__ float q8(float v) { return fminf(fmaxf(rintf(v), 0.f), 255.f); }
// brightness / contrast on uint8 images go through a lookup table built as clip(...).astype(uint8): the cast TRUNCATES
// (albumentations' uint8 functionals); only saturation (cv2.addWeighted) and the resize round
__device__ __forceinline__ float t8(float v) { return floorf(fminf(fmaxf(v, 0.f), 255.f)); }
__device__ __forceinline__ float gray_of(Rgb p) { return q8(0.299f * p.r + 0.587f * p.g + 0.114f * p.b); }

__device__ __forceinline__ Rgb hue_shift(Rgb p, float h) {      // HSV with H in [0,1), shift by h (fraction of a turn)
    const float mx = fmaxf(p.r, fmaxf(p.g, p.b)), mn = fminf(p.r, fminf(p.g, p.b)), d = mx - mn;
    if (d <= 0.f) return p;
    float hh = mx == p.r ? (p.g - p.b) / d : (mx == p.g ? 2.f + (p.b - p.r) / d : 4.f + (p.r - p.g) / d);
    hh = hh / 6.f + h;
    hh -= floorf(hh);
    const float s = d / mx, v = mx, i = floorf(hh * 6.f), f = hh * 6.f - i;
    const float a = v * (1.f - s), b = v * (1.f - s * f), c = v * (1.f - s * (1.f - f));
    Rgb o;
    switch ((int)i % 6) {
        case 0: o = {v, c, a}; break; case 1: o = {b, v, a}; break; case 2: o = {a, v, c}; break;
        case 3: o = {a, b, v}; break; case 4: o = {c, a, v}; break; default: o = {v, a, b}; break;
    }
    return {q8(o.r), q8(o.g), q8(o.b)};
}

// resized + flipped pixel (y, x) of image n, on the uint8 grid
__device__ __forceinline__ Rgb sample(const uint8_t* __restrict__ src, int n, int Hs, int Ws, const VtxAugParams& a, int size, int y, int x) {
    const int xd = a.flip ? size - 1 - x : x;
    const float fx = ((float)xd + 0.5f) * (float)a.cw / (float)size - 0.5f, fy = ((float)y + 0.5f) * (float)a.ch / (float)size - 0.5f;
    int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
    float wx = fx - (float)x0, wy = fy - (float)y0;
    if (x0 < 0) { x0 = 0; wx = 0.f; }
    if (y0 < 0) { y0 = 0; wy = 0.f; }
    int x1 = x0 + 1, y1 = y0 + 1;
    if (x1 > a.cw - 1) { x1 = a.cw - 1; if (x0 > a.cw - 1) { x0 = a.cw - 1; wx = 0.f; } }
    if (y1 > a.ch - 1) { y1 = a.ch - 1; if (y0 > a.ch - 1) { y0 = a.ch - 1; wy = 0.f; } }
    const uint8_t* base = src + (long)n * Hs * Ws * 3;
    const uint8_t* p00 = base + ((long)(a.y0 + y0) * Ws + a.x0 + x0) * 3;
    const uint8_t* p01 = base + ((long)(a.y0 + y0) * Ws + a.x0 + x1) * 3;
    const uint8_t* p10 = base + ((long)(a.y0 + y1) * Ws + a.x0 + x0) * 3;
    const uint8_t* p11 = base + ((long)(a.y0 + y1) * Ws + a.x0 + x1) * 3;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float top = (float)p00[c] + wx * ((float)p01[c] - (float)p00[c]);
        const float bot = (float)p10[c] + wx * ((float)p11[c] - (float)p10[c]);
        v[c] = q8(top + wy * (bot - top));
    }
    return {v[0], v[1], v[2]};
}

// colour operations `first`..`last` (positions in the image's order); op codes: 0 brightness, 1 contrast, 2 saturation, 3 hue
__device__ __forceinline__ Rgb jitter(Rgb p, const VtxAugParams& a, int first, int last, float gray_mean) {
    for (int k = first; k < last; ++k) {
        const int op = (a.order >> (2 * k)) & 3;
        if (op == 0) { p = {t8(p.r * a.brightness), t8(p.g * a.brightness), t8(p.b * a.brightness)}; }
        else if (op == 1) { const float m = gray_mean * (1.f - a.contrast); p = {t8(p.r * a.contrast + m), t8(p.g * a.contrast + m), t8(p.b * a.contrast + m)}; }
        else if (op == 2) { const float g = gray_of(p) * (1.f - a.saturation); p = {q8(p.r * a.saturation + g), q8(p.g * a.saturation + g), q8(p.b * a.saturation + g)}; }
        else if (a.hue != 0.f) p = hue_shift(p, a.hue);
    }
    return p;
}
__device__ __forceinline__ int contrast_pos(const VtxAugParams& a) {
    for (int k = 0; k < 4; ++k) if (((a.order >> (2 * k)) & 3) == 1) return k;
    return 4;
}

// pass 1 (only images whose contrast factor != 1): mean of the gray image at the point the contrast operation sees it
__global__ __launch_bounds__(256) void augment_mean_kernel(const uint8_t* __restrict__ src, const VtxAugParams* __restrict__ params,
                                                           float* __restrict__ gray_sum, int Hs, int Ws, int size) {
    __shared__ float red[4];
    const int n = blockIdx.y;
    const VtxAugParams a = params[n];
    if (a.contrast == 1.f) return;
    const int cp = contrast_pos(a);
    float s = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < size * size; i += gridDim.x * 256)
        s += gray_of(jitter(sample(src, n, Hs, Ws, a, size, i / size, i % size), a, 0, cp, 0.f));
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) gray_sum[n] = s;            // one block per image: deterministic
}

template <class T>
__global__ __launch_bounds__(256) void augment_kernel(const uint8_t* __restrict__ src, const VtxAugParams* __restrict__ params,
                                                      const float* __restrict__ gray_sum, T* __restrict__ dst, int N, int Hs, int Ws,
                                                      int size, int Cp, int halo, float m0, float m1, float m2, float r0, float r1, float r2) {
    const int So = size + 2 * halo;
    const long total = (long)N * So * So;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i / ((long)So * So));
        const int rem = (int)(i - (long)n * So * So);
        const int y = rem / So - halo, x = rem % So - halo;
        T* o = dst + i * Cp;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if ((unsigned)y < (unsigned)size && (unsigned)x < (unsigned)size) {
            const VtxAugParams a = params[n];
            const float gm = gray_sum[n] / (float)(size * size);
            const Rgb p = jitter(sample(src, n, Hs, Ws, a, size, y, x), a, 0, 4, gm);
            v0 = (p.r - m0) * r0; v1 = (p.g - m1) * r1; v2 = (p.b - m2) * r2;
        }
        Elem<T>::st(o, v0); Elem<T>::st(o + 1, v1); Elem<T>::st(o + 2, v2);
        for (int c = 3; c < Cp; ++c) Elem<T>::st(o + c, 0.f);
    }
}

}  // namespace

extern "C" int vtx_image_augment_u8(int dtype, const uint8_t* src, void* dst, const VtxAugParams* params, float* gray_sum, int N,
                                    int Hs, int Ws, int size, int Cpad, int halo, const float* mean, const float* std, void* stream) {
    VTX_CHECK(src && dst && params && gray_sum && mean && std, VTX_ERR_ARG, "image_augment_u8: null pointer");
    VTX_CHECK(N > 0 && Hs > 0 && Ws > 0 && size > 0 && Cpad >= 3 && halo >= 0, VTX_ERR_SHAPE, "image_augment_u8: bad shape");
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "image_augment_u8: bad dtype");
    hipStream_t st = (hipStream_t)stream;
    const float m0 = 255.f * mean[0], m1 = 255.f * mean[1], m2 = 255.f * mean[2];
    const float r0 = 1.f / (255.f * std[0]), r1 = 1.f / (255.f * std[1]), r2 = 1.f / (255.f * std[2]);
    if (hipMemsetAsync(gray_sum, 0, sizeof(float) * N, st) != hipSuccess) { vtx_set_error("image_augment_u8: memset failed"); return VTX_ERR_LAUNCH; }
    VTX_KLAUNCH("image_augment", 0, 3.0 * N * size * size, augment_mean_kernel, dim3(1, N), dim3(256), 0, st, src, params, gray_sum, Hs, Ws, size);
    const long total = (long)N * (size + 2 * halo) * (size + 2 * halo);
    long g = (total + 255) / 256; if (g > 8192) g = 8192;
    if (dtype == VTX_BF16)
        VTX_KLAUNCH("image_augment", 0, 3.0 * N * size * size + 2.0 * total * Cpad, (augment_kernel<bf16_t>), dim3((int)g), dim3(256), 0, st, src, params, gray_sum,
                    (bf16_t*)dst, N, Hs, Ws, size, Cpad, halo, m0, m1, m2, r0, r1, r2);
    else
        VTX_KLAUNCH("image_augment", 0, 3.0 * N * size * size + 4.0 * total * Cpad, (augment_kernel<float>), dim3((int)g), dim3(256), 0, st, src, params, gray_sum,
                    (float*)dst, N, Hs, Ws, size, Cpad, halo, m0, m1, m2, r0, r1, r2);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}
