// Layout / precision preparation kernels (HBM-bound plumbing of the step):
//  * vtx_image_to_nhwc : fp32 NCHW image (the batch format of the reference collate function,
//    /root/reference/virtex/data/datasets/captioning.py:79-100) -> NHWC dtype with the 3 input
//    channels zero-padded to Cp (8), so the stem conv runs on the same implicit-GEMM kernel.
//  * vtx_weight_prep   : fp32 master weight [KO][T][C] (T = R*S taps; 1 for linears) ->
//    compute copy w[KO][T][Cp] and transposed copy wt[Cp][T][KO] (both `dtype`); the
//    transposed copy is the B operand of every input-gradient GEMM.
//  * vtx_cast          : fp32 -> dtype elementwise.
#include "vtx_common.h"

namespace {

// output pixel grid (H + 2*halo) x (W + 2*halo): a zero frame of `halo` pixels lets the stem run as a "valid"
// convolution without any bounds logic (and with two 4-channel pixels per 16-byte chunk, see conv_common.h)
template <class T>
__global__ __launch_bounds__(256) void image_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                            int N, int Cin, int H, int W, int Cp, int halo) {
    const int Ho = H + 2 * halo, Wo = W + 2 * halo;
    const long total = (long)N * Ho * Wo;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / ((long)Ho * Wo);
        const int rem = (int)(i - n * Ho * Wo);
        const int y = rem / Wo - halo, x = rem % Wo - halo;
        const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        if constexpr (sizeof(T) == 2) {
            if (Cp == 4) {                       // the packed stem layout: one 8-byte store per pixel, not four 2-byte ones
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = (in && c < Cin) ? src[((n * Cin + c) * H + y) * W + x] : 0.f;
                *reinterpret_cast<uint2*>(dst + i * 4) = make_uint2(f2bf2(v[0], v[1]), f2bf2(v[2], v[3]));
                continue;
            }
        }
        for (int c = 0; c < Cp; ++c) {
            const float v = (in && c < Cin) ? src[((n * Cin + c) * H + y) * W + x] : 0.f;
            Elem<T>::st(dst + i * Cp + c, v);
        }
    }
}

// one 32x32 (ko x c) tile per block, for tap blockIdx.z
template <class T>
__global__ __launch_bounds__(256) void weight_prep_kernel(const float* __restrict__ w32, T* __restrict__ w,
                                                          T* __restrict__ wt, int KO, int Tn, int C, int Cp) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int c0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int ko = k0 + r, c = c0 + tx;
        float v = 0.f;
        if (ko < KO && c < C) v = w32[((long)ko * Tn + t) * C + c];
        tile[r][tx] = v;
        if (w && ko < KO && c < Cp) Elem<T>::st(w + ((long)ko * Tn + t) * Cp + c, v);
    }
    __syncthreads();
    if (wt) {
        for (int r = ty; r < 32; r += 8) {
            const int c = c0 + r, ko = k0 + tx;
            if (c < Cp && ko < KO) Elem<T>::st(wt + ((long)c * Tn + t) * KO + ko, tile[tx][r]);
        }
    }
}

// All the step's weight preparations in ONE launch: block b finds its descriptor by binary search in the
// prefix sums of the per-weight tile counts, then does the same 32x32 (ko x c) tile as weight_prep_kernel.
// (69 separate launches of 3-12 us each were 0.45 ms of a 39 ms step.)
template <class T>
__global__ __launch_bounds__(256) void weight_prep_batched_kernel(const VtxPrepDesc* __restrict__ descs,
                                                                  const int* __restrict__ tile_start, int ndesc) {
    int lo = 0, hi = ndesc - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tile_start[mid] <= b) lo = mid; else hi = mid - 1;
    }
    const VtxPrepDesc d = descs[lo];
    const int local = b - tile_start[lo];
    const int tc = (d.Cp + 31) >> 5, tk = (d.KO + 31) >> 5;
    const int t = local / (tc * tk), rem = local - t * (tc * tk);
    const int k0 = (rem / tc) * 32, c0 = (rem % tc) * 32;
    const float* w32 = (const float*)d.w32;
    T* w = (T*)d.w;
    T* wt = (T*)d.wt;
    __shared__ float tile[32][33];
    // Interior tiles of unpadded, 4-aligned matrices (all but the stem's and the last tile row of the 10 000-row tied matrix): ONE
    // 16-byte load and one or two 8-byte (bf16) / 16-byte (fp32) stores per thread -- the scalar form below moved the same tile
    // with 4-byte loads and 2-byte stores, four of each per thread, and ran at 3.0 TB/s (round 6: the launch sits at the very
    // start of the step, on the critical path).
    const bool vec = k0 + 32 <= d.KO && c0 + 32 <= d.C && d.C == d.Cp && (d.C & 3) == 0 && (d.KO & 3) == 0 &&
                     (((uintptr_t)w32 | (uintptr_t)w | (uintptr_t)wt) & 15) == 0;
    if (vec) {
        const int r = threadIdx.x >> 3, q = (threadIdx.x & 7) * 4;           // row of the tile, first of this thread's four columns
        const float4 v = *reinterpret_cast<const float4*>(w32 + ((long)(k0 + r) * d.T + t) * d.C + c0 + q);
        tile[r][q] = v.x; tile[r][q + 1] = v.y; tile[r][q + 2] = v.z; tile[r][q + 3] = v.w;
        if (w) {
            T* dst = w + ((long)(k0 + r) * d.T + t) * d.Cp + c0 + q;
            if constexpr (sizeof(T) == 2) *reinterpret_cast<uint2*>(dst) = make_uint2(f2bf2(v.x, v.y), f2bf2(v.z, v.w));
            else *reinterpret_cast<float4*>(dst) = v;
        }
        __syncthreads();
        if (wt) {                                                           // row r of the transposed tile = channel c0 + r, four ko
            T* dst = wt + ((long)(c0 + r) * d.T + t) * d.KO + k0 + q;
            const float a0 = tile[q][r], a1 = tile[q + 1][r], a2 = tile[q + 2][r], a3 = tile[q + 3][r];
            if constexpr (sizeof(T) == 2) *reinterpret_cast<uint2*>(dst) = make_uint2(f2bf2(a0, a1), f2bf2(a2, a3));
            else *reinterpret_cast<float4*>(dst) = make_float4(a0, a1, a2, a3);
        }
        return;
    }
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int ko = k0 + r, c = c0 + tx;
        float v = 0.f;
        if (ko < d.KO && c < d.C) v = w32[((long)ko * d.T + t) * d.C + c];
        tile[r][tx] = v;
        if (w && ko < d.KO && c < d.Cp) Elem<T>::st(w + ((long)ko * d.T + t) * d.Cp + c, v);
    }
    __syncthreads();
    if (wt) {
        for (int r = ty; r < 32; r += 8) {
            const int c = c0 + r, ko = k0 + tx;
            if (c < d.Cp && ko < d.KO) Elem<T>::st(wt + ((long)c * d.T + t) * d.KO + ko, tile[tx][r]);
        }
    }
}

template <class T>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        Elem<T>::st(dst + i, src[i]);
}

// Inference folding of BatchNorm into the preceding bias-free convolution:
//   scale[ko] = gamma[ko] * rsqrt(running_var[ko] + eps)
//   w'[ko][t][c] = w[ko][t][c] * scale[ko]   (cast to dtype, C zero-padded to Cp)
//   bias[ko]     = beta[ko] - running_mean[ko] * scale[ko]
template <class T>
__global__ __launch_bounds__(256) void bn_fold_kernel(const float* __restrict__ w32, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ rmean,
                                                      const float* __restrict__ rvar, float eps, T* __restrict__ w,
                                                      float* __restrict__ bias, int KO, int Tn, int C, int Cp) {
    const int ko = blockIdx.x;
    const float sc = gamma[ko] * rsqrtf(rvar[ko] + eps);
    if (threadIdx.x == 0) bias[ko] = beta[ko] - rmean[ko] * sc;
    const int per = Tn * Cp;
    for (int i = threadIdx.x; i < per; i += 256) {
        const int t = i / Cp, c = i - t * Cp;
        const float v = c < C ? w32[((long)ko * Tn + t) * C + c] * sc : 0.f;
        Elem<T>::st(w + (long)ko * per + i, v);
    }
}

// uint8 HWC pixels (what an image decoder produces) -> normalised NHWC in the compute dtype, channels padded:
//   dst[n][y][x][c] = (src[n][y0+y][xs][c] - 255*mean[c]) * (1 / (255*std[c])),  xs = flip ? x0+W-1-x : x0+x
// i.e. albumentations.Normalize(mean, std, max_pixel_value=255) (reference: virtex/data/transforms.py:85-97) fused
// with the crop window, the horizontal flip and the layout change; the batch crosses PCIe as 1 byte per value.
template <class T>
__global__ __launch_bounds__(256) void image_u8_kernel(const uint8_t* __restrict__ src, T* __restrict__ dst, int N, int Hs,
                                                       int Ws, int H, int W, int Cp, int halo, const int* __restrict__ crop_xy,
                                                       const uint8_t* __restrict__ flip, float m0, float m1, float m2,
                                                       float r0, float r1, float r2) {
    const int Ho = H + 2 * halo, Wo = W + 2 * halo;
    const long total = (long)N * Ho * Wo;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i / ((long)Ho * Wo));
        const int rem = (int)(i - (long)n * Ho * Wo);
        const int y = rem / Wo - halo, x = rem % Wo - halo;
        T* o = dst + i * Cp;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
            const int x0 = crop_xy ? crop_xy[2 * n] : 0, y0 = crop_xy ? crop_xy[2 * n + 1] : 0;
            const int xs = (flip && flip[n]) ? x0 + W - 1 - x : x0 + x;
            const uint8_t* p = src + (((long)n * Hs + (y0 + y)) * Ws + xs) * 3;
            v0 = ((float)p[0] - m0) * r0; v1 = ((float)p[1] - m1) * r1; v2 = ((float)p[2] - m2) * r2;
        }
        Elem<T>::st(o, v0); Elem<T>::st(o + 1, v1); Elem<T>::st(o + 2, v2);
        for (int c = 3; c < Cp; ++c) Elem<T>::st(o + c, 0.f);
    }
}

static int grid_for(long total) {
    long g = (total + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int vtx_image_to_nhwc_halo(int dtype, const float* src, void* dst, int N, int Cin, int H, int W, int Cp,
                                      int halo, void* stream) {
    VTX_CHECK(src && dst, VTX_ERR_ARG, "image_to_nhwc: null pointer");
    VTX_CHECK(N > 0 && Cin > 0 && Cp >= Cin && H > 0 && W > 0 && halo >= 0, VTX_ERR_SHAPE, "image_to_nhwc: bad shape");
    const long total = (long)N * (H + 2 * halo) * (W + 2 * halo);
    if (dtype == VTX_BF16)
        VTX_KLAUNCH("image_to_nhwc", 0, 4.0 * N * Cin * H * W + 2.0 * total * Cp, (image_to_nhwc_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src,
                           (bf16_t*)dst, N, Cin, H, W, Cp, halo);
    else if (dtype == VTX_F32)
        VTX_KLAUNCH("image_to_nhwc", 0, 4.0 * N * Cin * H * W + 4.0 * total * Cp, (image_to_nhwc_kernel<float>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src,
                           (float*)dst, N, Cin, H, W, Cp, halo);
    else VTX_CHECK(false, VTX_ERR_DTYPE, "image_to_nhwc: bad dtype");
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}
extern "C" int vtx_image_to_nhwc(int dtype, const float* src, void* dst, int N, int Cin, int H, int W, int Cp,
                                 void* stream) {
    return vtx_image_to_nhwc_halo(dtype, src, dst, N, Cin, H, W, Cp, 0, stream);
}

extern "C" int vtx_weight_prep(int dtype, const float* w32, void* w, void* wt, int KO, int T, int C, int Cp,
                               void* stream) {
    VTX_CHECK(w32 && (w || wt), VTX_ERR_ARG, "weight_prep: null pointer");
    VTX_CHECK(KO > 0 && T > 0 && C > 0 && Cp >= C && T < 65536, VTX_ERR_SHAPE, "weight_prep: bad shape");
    dim3 grid(vtx_cdiv(Cp, 32), vtx_cdiv(KO, 32), T), block(256);
    if (dtype == VTX_BF16)
        hipLaunchKernelGGL((weight_prep_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, w32, (bf16_t*)w,
                           (bf16_t*)wt, KO, T, C, Cp);
    else if (dtype == VTX_F32)
        hipLaunchKernelGGL((weight_prep_kernel<float>), grid, block, 0, (hipStream_t)stream, w32, (float*)w,
                           (float*)wt, KO, T, C, Cp);
    else VTX_CHECK(false, VTX_ERR_DTYPE, "weight_prep: bad dtype");
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

extern "C" int vtx_cast_from_f32(int dtype, const float* src, void* dst, long n, void* stream) {
    VTX_CHECK(src && dst && n >= 0, VTX_ERR_ARG, "cast: bad args");
    if (n == 0) return VTX_OK;
    if (dtype == VTX_BF16)
        hipLaunchKernelGGL((cast_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
    else if (dtype == VTX_F32)
        hipLaunchKernelGGL((cast_kernel<float>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, src, (float*)dst, n);
    else VTX_CHECK(false, VTX_ERR_DTYPE, "cast: bad dtype");
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

extern "C" int vtx_bn_fold(int dtype, const float* w32, const float* gamma, const float* beta, const float* running_mean,
                           const float* running_var, float eps, void* w, float* bias, int KO, int T, int C, int Cp,
                           void* stream) {
    VTX_CHECK(w32 && gamma && beta && running_mean && running_var && w && bias, VTX_ERR_ARG, "bn_fold: null pointer");
    VTX_CHECK(KO > 0 && T > 0 && C > 0 && Cp >= C, VTX_ERR_SHAPE, "bn_fold: bad shape");
    if (dtype == VTX_BF16)
        hipLaunchKernelGGL((bn_fold_kernel<bf16_t>), dim3(KO), dim3(256), 0, (hipStream_t)stream, w32, gamma, beta,
                           running_mean, running_var, eps, (bf16_t*)w, bias, KO, T, C, Cp);
    else if (dtype == VTX_F32)
        hipLaunchKernelGGL((bn_fold_kernel<float>), dim3(KO), dim3(256), 0, (hipStream_t)stream, w32, gamma, beta,
                           running_mean, running_var, eps, (float*)w, bias, KO, T, C, Cp);
    else VTX_CHECK(false, VTX_ERR_DTYPE, "bn_fold: bad dtype");
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

extern "C" int vtx_weight_prep_batched(int dtype, const VtxPrepDesc* descs, const int* tile_start, int ndesc,
                                       int total_tiles, void* stream) {
    VTX_CHECK(descs && tile_start && ndesc > 0 && total_tiles > 0, VTX_ERR_ARG, "weight_prep_batched: bad arguments");
    if (dtype == VTX_BF16)
        VTX_KLAUNCH("weight_prep", 0, 1024.0 * total_tiles * (4 + 2 + 2), (weight_prep_batched_kernel<bf16_t>), dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, descs, tile_start, ndesc);
    else if (dtype == VTX_F32)
        VTX_KLAUNCH("weight_prep", 0, 1024.0 * total_tiles * (4 + 4 + 4), (weight_prep_batched_kernel<float>), dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, descs, tile_start, ndesc);
    else VTX_CHECK(false, VTX_ERR_DTYPE, "weight_prep_batched: bad dtype");
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

extern "C" int vtx_image_u8_to_nhwc(int dtype, const uint8_t* src, void* dst, int N, int Hs, int Ws, int H, int W, int Cpad,
                                    int halo, const int* crop_xy, const uint8_t* flip, const float* mean, const float* std,
                                    void* stream) {
    VTX_CHECK(src && dst && mean && std, VTX_ERR_ARG, "image_u8_to_nhwc: null pointer");
    VTX_CHECK(N > 0 && H > 0 && W > 0 && Hs >= H && Ws >= W && Cpad >= 3 && halo >= 0, VTX_ERR_SHAPE, "image_u8_to_nhwc: bad shape");
    const float m0 = 255.f * mean[0], m1 = 255.f * mean[1], m2 = 255.f * mean[2];
    const float r0 = 1.f / (255.f * std[0]), r1 = 1.f / (255.f * std[1]), r2 = 1.f / (255.f * std[2]);
    const long total = (long)N * (H + 2 * halo) * (W + 2 * halo);
    if (dtype == VTX_BF16)
        VTX_KLAUNCH("image_to_nhwc", 0, 3.0 * N * H * W + 2.0 * total * Cpad, (image_u8_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst,
                           N, Hs, Ws, H, W, Cpad, halo, crop_xy, flip, m0, m1, m2, r0, r1, r2);
    else if (dtype == VTX_F32)
        VTX_KLAUNCH("image_to_nhwc", 0, 3.0 * N * H * W + 4.0 * total * Cpad, (image_u8_kernel<float>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, (float*)dst,
                           N, Hs, Ws, H, W, Cpad, halo, crop_xy, flip, m0, m1, m2, r0, r1, r2);
    else VTX_CHECK(false, VTX_ERR_DTYPE, "image_u8_to_nhwc: bad dtype");
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}
