// MaxPool2d(kernel 3, stride 2, padding 1) on NHWC activations, forward (+argmax) / backward.
// Replaces aten::max_pool2d_with_indices(+backward) of the ResNet stem
// (/root/reference/virtex/modules/visual_backbones.py:68-74; -inf padding, first maximum
// wins on ties -- post-ReLU inputs tie at 0 all the time, so the tie rule is observable).
#include "vtx_common.h"

namespace {

template <class T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          uint8_t* __restrict__ argmax, int N, int H, int W,
                                                          int C, int OH, int OW) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = C / VEC;
    const long total = (long)N * OH * OW * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c0 = (int)(i % cv) * VEC;
        long p = i / cv;
        const int ow = (int)(p % OW); p /= OW;
        const int oh = (int)(p % OH);
        const int n = (int)(p / OH);
        float best[VEC]; int idx[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) { best[j] = -INFINITY; idx[j] = 0; }
        bool first = true;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ih = oh * 2 - 1 + kh, iw = ow * 2 - 1 + kw;
                if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
                    Vec16<T> v; v.load(x + (((long)n * H + ih) * W + iw) * C + c0);
#pragma unroll
                    for (int j = 0; j < VEC; ++j)
                        if (first || v.v[j] > best[j]) { best[j] = v.v[j]; idx[j] = kh * 3 + kw; }
                    first = false;
                }
            }
        Vec16<T> o;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o.v[j] = best[j];
        const long off = (((long)n * OH + oh) * OW + ow) * C + c0;
        o.store(y + off);
#pragma unroll
        for (int j = 0; j < VEC; ++j) argmax[off + j] = (uint8_t)idx[j];
    }
}

// gather form: each input pixel sums the dy of the (<= 2x2) windows whose argmax points at it
template <class T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ argmax,
                                                          T* __restrict__ dx, int N, int H, int W, int C, int OH,
                                                          int OW) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = C / VEC;
    const long total = (long)N * H * W * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c0 = (int)(i % cv) * VEC;
        long p = i / cv;
        const int iw = (int)(p % W); p /= W;
        const int ih = (int)(p % H);
        const int n = (int)(p / H);
        Vec16<T> acc;
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc.v[j] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int th = ih + 1 - kh;
            if (th < 0 || (th & 1) || (th >> 1) >= OH) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int tw = iw + 1 - kw;
                if (tw < 0 || (tw & 1) || (tw >> 1) >= OW) continue;
                const long off = (((long)n * OH + (th >> 1)) * OW + (tw >> 1)) * C + c0;
                Vec16<T> g; g.load(dy + off);
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    if (argmax[off + j] == (uint8_t)(kh * 3 + kw)) acc.v[j] += g.v[j];
            }
        }
        acc.store(dx + (((long)n * H + ih) * W + iw) * C + c0);
    }
}

static int grid_for(long total) {
    long g = (total + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int vtx_maxpool3x3s2_fwd(int dtype, const void* x, void* y, uint8_t* argmax, int N, int H, int W,
                                    int C, void* stream) {
    VTX_CHECK(x && y && argmax, VTX_ERR_ARG, "maxpool_fwd: null pointer");
    const int vec = dtype == VTX_BF16 ? 8 : 4;
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "maxpool_fwd: bad dtype");
    VTX_CHECK(N > 0 && H > 0 && W > 0 && C % vec == 0, VTX_ERR_SHAPE, "maxpool_fwd: bad shape");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long total = (long)N * OH * OW * (C / vec);
    if (dtype == VTX_BF16)
        VTX_KLAUNCH("maxpool_fwd", 0, 2.0 * N * H * W * C + 3.0 * N * OH * OW * C, (maxpool_fwd_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)x, (bf16_t*)y, argmax, N, H, W, C, OH, OW);
    else
        VTX_KLAUNCH("maxpool_fwd", 0, 4.0 * N * H * W * C + 5.0 * N * OH * OW * C, (maxpool_fwd_kernel<float>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, (float*)y, argmax, N, H, W, C, OH, OW);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

extern "C" int vtx_maxpool3x3s2_bwd(int dtype, const void* dy, const uint8_t* argmax, void* dx, int N, int H,
                                    int W, int C, void* stream) {
    VTX_CHECK(dy && dx && argmax, VTX_ERR_ARG, "maxpool_bwd: null pointer");
    const int vec = dtype == VTX_BF16 ? 8 : 4;
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "maxpool_bwd: bad dtype");
    VTX_CHECK(N > 0 && H > 0 && W > 0 && C % vec == 0, VTX_ERR_SHAPE, "maxpool_bwd: bad shape");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long total = (long)N * H * W * (C / vec);
    if (dtype == VTX_BF16)
        VTX_KLAUNCH("maxpool_bwd", 0, 2.0 * N * H * W * C + 3.0 * N * OH * OW * C, (maxpool_bwd_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)dy, argmax, (bf16_t*)dx, N, H, W, C, OH, OW);
    else
        VTX_KLAUNCH("maxpool_bwd", 0, 4.0 * N * H * W * C + 5.0 * N * OH * OW * C, (maxpool_bwd_kernel<float>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)dy, argmax, (float*)dx, N, H, W, C, OH, OW);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}
