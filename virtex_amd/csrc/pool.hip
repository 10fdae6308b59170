// MaxPool2d(kernel 3, stride 2, padding 1) on NHWC activations, forward (+argmax) / backward.
// Replaces aten::max_pool2d_with_indices(+backward) of the ResNet stem
// (/root/reference/virtex/modules/visual_backbones.py:68-74; -inf padding, first maximum
// wins on ties -- post-ReLU inputs tie at 0 all the time, so the tie rule is observable).
#include "vtx_common.h"
#include "pool_windows.h"

namespace {

// Thread layout of both kernels: tx = channel vector (fixed for the thread's life), ty = pixel; a block covers
// TY = 256 / TX pixels per trip.  Pixel indices stay below 2^30 (reciprocal division, two per pixel); the eight
// argmax bytes of a vector travel as ONE 8-byte access.
__device__ __forceinline__ int pool_qdiv(int n, int d) { return vtx_fdiv30(n, d, __builtin_amdgcn_rcpf((float)d)); }   // exact for 0 <= n < 2^30
template <int VEC> struct ArgPack;
template <> struct ArgPack<8> {
    typedef uint2 W;
    __device__ static __forceinline__ uint32_t get(W w, int j) { return ((j < 4 ? w.x : w.y) >> (8 * (j & 3))) & 0xffu; }
    __device__ static __forceinline__ W make(const int* idx) {
        return make_uint2((uint32_t)idx[0] | ((uint32_t)idx[1] << 8) | ((uint32_t)idx[2] << 16) | ((uint32_t)idx[3] << 24),
                          (uint32_t)idx[4] | ((uint32_t)idx[5] << 8) | ((uint32_t)idx[6] << 16) | ((uint32_t)idx[7] << 24));
    }
};
template <> struct ArgPack<4> {
    typedef uint32_t W;
    __device__ static __forceinline__ uint32_t get(W w, int j) { return (w >> (8 * j)) & 0xffu; }
    __device__ static __forceinline__ W make(const int* idx) {
        return (uint32_t)idx[0] | ((uint32_t)idx[1] << 8) | ((uint32_t)idx[2] << 16) | ((uint32_t)idx[3] << 24);
    }
};

template <class T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          uint8_t* __restrict__ argmax, int N, int H, int W,
                                                          int C, int OH, int OW, int TX) {
    constexpr int VEC = Elem<T>::VEC;
    typedef ArgPack<VEC> AP;
    const int cv = C / VEC, TY = 256 / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int cvi = blockIdx.y * TX + tx;
    if (cvi >= cv || ty >= TY) return;
    const int c0 = cvi * VEC, P = N * OH * OW;
    for (int p = blockIdx.x * TY + ty; p < P; p += gridDim.x * TY) {
        const int n = pool_qdiv(p, OH * OW), rem = p - n * OH * OW;
        const int oh = pool_qdiv(rem, OW), ow = rem - oh * OW;
        float best[VEC]; int idx[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) { best[j] = -INFINITY; idx[j] = 0; }
        PoolTaps<T> taps;
        taps.request(x, n, oh, ow, c0, H, W, C);             // all nine taps in flight, then consumed in (kh, kw) order
        bool first = true;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const bool ok = (taps.valid >> k) & 1u;
            float v[VEC];
            vtx_unpack_raw16<T>(taps.v[k], v);
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                if (ok && (first || v[j] > best[j])) { best[j] = v[j]; idx[j] = k; }
            first = first && !ok;
        }
        Vec16<T> o;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o.v[j] = best[j];
        const long off = (long)p * C + c0;
        o.store(y + off);
        *reinterpret_cast<typename AP::W*>(argmax + off) = AP::make(idx);
    }
}

// gather form: each input pixel sums the dy of the (<= 2x2) windows whose argmax points at it
template <class T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ argmax,
                                                          T* __restrict__ dx, int N, int H, int W, int C, int OH,
                                                          int OW, int TX) {
    constexpr int VEC = Elem<T>::VEC;
    typedef ArgPack<VEC> AP;
    const int cv = C / VEC, TY = 256 / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int cvi = blockIdx.y * TX + tx;
    if (cvi >= cv || ty >= TY) return;
    const int c0 = cvi * VEC, P = N * H * W;
    for (int p = blockIdx.x * TY + ty; p < P; p += gridDim.x * TY) {
        const int n = pool_qdiv(p, H * W), rem = p - n * H * W;
        const int ih = pool_qdiv(rem, W), iw = rem - ih * W;
        PoolWindows<T> win;
        win.request(dy, argmax, n, ih, iw, c0, C, OH, OW);   // the <= 2 x 2 windows containing this pixel, all in flight
        Vec16<T> acc;
        win.gather(acc.v);
        acc.store(dx + (long)p * C + c0);
    }
}

struct PoolGrid { int TX, gy, gx; };
static PoolGrid grid_for(int pixels, int cv) {
    PoolGrid g;
    g.TX = 1; while (g.TX < cv && g.TX < 256) g.TX <<= 1;      // power of two >= cv (threads past cv idle), at most 256
    g.gy = (cv + g.TX - 1) / g.TX;
    const int TY = 256 / g.TX;
    long gx = ((long)pixels + TY - 1) / TY;
    g.gx = (int)(gx > 8192 ? 8192 : (gx < 1 ? 1 : gx));
    return g;
}

}  // namespace

extern "C" int vtx_maxpool3x3s2_fwd(int dtype, const void* x, void* y, uint8_t* argmax, int N, int H, int W,
                                    int C, void* stream) {
    VTX_CHECK(x && y && argmax, VTX_ERR_ARG, "maxpool_fwd: null pointer");
    const int vec = dtype == VTX_BF16 ? 8 : 4;
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "maxpool_fwd: bad dtype");
    VTX_CHECK(N > 0 && H > 0 && W > 0 && C % vec == 0, VTX_ERR_SHAPE, "maxpool_fwd: bad shape");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    VTX_CHECK((long)N * H * W < VTX_PIXEL_LIMIT, VTX_ERR_SHAPE, "maxpool_fwd: more than 2^30 pixels is not supported");
    const PoolGrid pg = grid_for(N * OH * OW, C / vec);
    if (dtype == VTX_BF16)
        VTX_KLAUNCH("maxpool_fwd", 0, 2.0 * N * H * W * C + 3.0 * N * OH * OW * C, (maxpool_fwd_kernel<bf16_t>), dim3(pg.gx, pg.gy), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)x, (bf16_t*)y, argmax, N, H, W, C, OH, OW, pg.TX);
    else
        VTX_KLAUNCH("maxpool_fwd", 0, 4.0 * N * H * W * C + 5.0 * N * OH * OW * C, (maxpool_fwd_kernel<float>), dim3(pg.gx, pg.gy), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, (float*)y, argmax, N, H, W, C, OH, OW, pg.TX);
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
    VTX_CHECK((long)N * H * W < VTX_PIXEL_LIMIT, VTX_ERR_SHAPE, "maxpool_bwd: more than 2^30 pixels is not supported");
    const PoolGrid pg = grid_for(N * H * W, C / vec);
    if (dtype == VTX_BF16)
        VTX_KLAUNCH("maxpool_bwd", 0, 2.0 * N * H * W * C + 3.0 * N * OH * OW * C, (maxpool_bwd_kernel<bf16_t>), dim3(pg.gx, pg.gy), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)dy, argmax, (bf16_t*)dx, N, H, W, C, OH, OW, pg.TX);
    else
        VTX_KLAUNCH("maxpool_bwd", 0, 4.0 * N * H * W * C + 5.0 * N * OH * OW * C, (maxpool_bwd_kernel<float>), dim3(pg.gx, pg.gy), dim3(256), 0, (hipStream_t)stream,
                           (const float*)dy, argmax, (float*)dx, N, H, W, C, OH, OW, pg.TX);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}
