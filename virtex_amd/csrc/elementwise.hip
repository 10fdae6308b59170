// Small HBM-bound helpers: column sums (bias gradients), elementwise add, GELU backward.
#include "vtx_common.h"

namespace {

// out[c] += sum_r x[r][c]   (x: [R][ld] dtype, C columns).  Same geometry as the BN reductions.
template <class T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, long ld, float* __restrict__ out,
                                                     int R, int C, int TX, int rows_per_block) {
    constexpr int VEC = Elem<T>::VEC;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX, TY = 256 / TX;
    const int c0 = (blockIdx.y * TX + tx) * VEC;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
    float a[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) a[j] = 0.f;
    if (c0 < C)
        for (int r = r0 + ty; r < r1; r += TY) {
            Vec16<T> v; v.load(x + (long)r * ld + c0);
#pragma unroll
            for (int j = 0; j < VEC; ++j) a[j] += v.v[j];
        }
    __shared__ float red[256 * VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) red[(ty * TX + tx) * VEC + j] = a[j];
    __syncthreads();
    if (ty == 0 && c0 < C) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float s = 0.f;
            for (int r = 0; r < TY; ++r) s += red[(r * TX + tx) * VEC + j];
            out[(size_t)blockIdx.x * C + c0 + j] = s;   // per-strip partial, summed by colsum_finalize
        }
    }
}

// block = 32 columns x 8 groups of partials
__global__ void colsum_finalize_kernel(const float* __restrict__ parts, float* __restrict__ out, int C, int nparts) {
    __shared__ float red[8][32];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5;
    float s = 0.f;
    if (c < C) {
        int b = grp;
        for (; b + 56 < nparts; b += 64) {       // eight partials in flight, added in the same order as one by one
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = parts[(size_t)(b + 8 * u) * C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u];
        }
        for (; b < nparts; b += 8) s += parts[(size_t)b * C + c];
    }
    red[grp][threadIdx.x & 31] = s;
    __syncthreads();
    if (grp == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += red[g][threadIdx.x & 31];
        out[c] += t;
    }
}

template <class T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out,
                                                  long nvec) {
    constexpr int VEC = Elem<T>::VEC;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        Vec16<T> x, y; x.load(a + i * VEC); y.load(b + i * VEC);
#pragma unroll
        for (int j = 0; j < VEC; ++j) x.v[j] += y.v[j];
        x.store(out + i * VEC);
    }
}

// dh = dropout'(da) * gelu'(h)   (FFN: a = dropout(gelu(h)))
template <class T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ h, const T* __restrict__ da,
                                                       T* __restrict__ dh, long nvec, Dropout drop) {
    constexpr int VEC = Elem<T>::VEC;
    drop = drop.resolved();
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        Vec16<T> x, g; x.load(h + i * VEC); g.load(da + i * VEC);
#pragma unroll
        for (int j = 0; j < VEC; ++j) g.v[j] = drop.apply(g.v[j], (uint64_t)(i * VEC + j)) * gelu_erf_grad(x.v[j]);
        g.store(dh + i * VEC);
    }
}

// dy = dropout'(dx): the gradient through y -> x + dropout(y) of a pre-norm decoder sub-layer (the mask is the one the GEMM
// epilogue that produced x + dropout(y) drew: same seed, element index = position in the dense [rows][H] output)
template <class T>
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const T* __restrict__ dx, T* __restrict__ dy, long nvec, Dropout drop) {
    constexpr int VEC = Elem<T>::VEC;
    drop = drop.resolved();
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        Vec16<T> g; g.load(dx + i * VEC);
#pragma unroll
        for (int j = 0; j < VEC; ++j) g.v[j] = drop.apply(g.v[j], (uint64_t)(i * VEC + j));
        g.store(dy + i * VEC);
    }
}

static int grid_for(long total) {
    long g = (total + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace

constexpr int VTX_COLSUM_MAX_PARTS = 256;
extern "C" long vtx_colsum_workspace_floats(int C) { return (long)VTX_COLSUM_MAX_PARTS * C; }

extern "C" int vtx_colsum_acc(int dtype, const void* x, long ld, float* out, float* workspace, int R, int C,
                              void* stream) {
    VTX_CHECK(x && out && workspace, VTX_ERR_ARG, "colsum_acc: null pointer");
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "colsum_acc: bad dtype");
    const int vec = dtype == VTX_BF16 ? 8 : 4;
    VTX_CHECK(R >= 0 && C > 0 && C % vec == 0 && ld % vec == 0, VTX_ERR_SHAPE, "colsum_acc: C and ld must be multiples of %d", vec);
    if (R == 0) return VTX_OK;
    const int cv = C / vec;
    int TX = 1;
    while (TX * 2 <= cv && TX < 256) TX *= 2;      // power of two <= min(cv, 256)
    const int gy = vtx_cdiv(cv, TX), TY = 256 / TX;
    int gx = VTX_COLSUM_MAX_PARTS;
    const int max_gx = vtx_cdiv(R, TY * 4);
    if (gx > max_gx) gx = max_gx;
    if (gx < 1) gx = 1;
    const int rows = vtx_cdiv(R, gx);
    gx = vtx_cdiv(R, rows);
    if (dtype == VTX_BF16)
        VTX_KLAUNCH("colsum", 0, 2.0 * R * C, (colsum_kernel<bf16_t>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ld, workspace, R, C, TX, rows);
    else
        VTX_KLAUNCH("colsum", 0, 4.0 * R * C, (colsum_kernel<float>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const float*)x, ld, workspace, R, C, TX, rows);
    VTX_KLAUNCH("colsum_finalize", 0, 8.0 * gx * C, colsum_finalize_kernel, dim3(vtx_cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, workspace, out, C, gx);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

extern "C" int vtx_add(int dtype, const void* a, const void* b, void* out, long n, void* stream) {
    VTX_CHECK(a && b && out, VTX_ERR_ARG, "add: null pointer");
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "add: bad dtype");
    const int vec = dtype == VTX_BF16 ? 8 : 4;
    VTX_CHECK(n >= 0 && n % vec == 0, VTX_ERR_SHAPE, "add: n must be a multiple of %d", vec);
    if (n == 0) return VTX_OK;
    if (dtype == VTX_BF16)
        VTX_KLAUNCH("add", 0, 6.0 * n, (add_kernel<bf16_t>), dim3(grid_for(n / vec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n / vec);
    else
        VTX_KLAUNCH("add", 0, 12.0 * n, (add_kernel<float>), dim3(grid_for(n / vec)), dim3(256), 0, (hipStream_t)stream, (const float*)a, (const float*)b, (float*)out, n / vec);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

extern "C" int vtx_gelu_bwd(int dtype, const void* h, const void* da, void* dh, long n, float p_drop, uint64_t seed,
                            void* stream) {
    VTX_CHECK(h && da && dh, VTX_ERR_ARG, "gelu_bwd: null pointer");
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "gelu_bwd: bad dtype");
    const int vec = dtype == VTX_BF16 ? 8 : 4;
    VTX_CHECK(n >= 0 && n % vec == 0, VTX_ERR_SHAPE, "gelu_bwd: n must be a multiple of %d", vec);
    if (n == 0) return VTX_OK;
    Dropout d = make_dropout(p_drop, seed);
    if (dtype == VTX_BF16)
        VTX_KLAUNCH("gelu_bwd", 0, 6.0 * n, (gelu_bwd_kernel<bf16_t>), dim3(grid_for(n / vec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h, (const bf16_t*)da, (bf16_t*)dh, n / vec, d);
    else
        VTX_KLAUNCH("gelu_bwd", 0, 12.0 * n, (gelu_bwd_kernel<float>), dim3(grid_for(n / vec)), dim3(256), 0, (hipStream_t)stream, (const float*)h, (const float*)da, (float*)dh, n / vec, d);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

// Gradient through a residual sub-layer's dropout, x_out = x + dropout(y)  ->  dy = mask(seed) * dx / (1 - p): the pre-norm
// decoder layers (nn.TransformerDecoderLayer(norm_first=True), /root/reference/virtex/modules/textual_heads.py:181-194;
// aten::native_dropout_backward).  The post-norm layers get this from vtx_layernorm_residual_bwd.
extern "C" int vtx_dropout_bwd(int dtype, const void* dx, void* dy, long n, float p_drop, uint64_t seed, void* stream) {
    VTX_CHECK(dx && dy, VTX_ERR_ARG, "dropout_bwd: null pointer");
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "dropout_bwd: bad dtype");
    const int vec = dtype == VTX_BF16 ? 8 : 4;
    VTX_CHECK(n >= 0 && n % vec == 0, VTX_ERR_SHAPE, "dropout_bwd: n must be a multiple of %d", vec);
    if (n == 0) return VTX_OK;
    Dropout d = make_dropout(p_drop, seed);
    if (dtype == VTX_BF16)
        VTX_KLAUNCH("dropout_bwd", 0, 4.0 * n, (dropout_bwd_kernel<bf16_t>), dim3(grid_for(n / vec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dx, (bf16_t*)dy, n / vec, d);
    else
        VTX_KLAUNCH("dropout_bwd", 0, 8.0 * n, (dropout_bwd_kernel<float>), dim3(grid_for(n / vec)), dim3(256), 0, (hipStream_t)stream, (const float*)dx, (float*)dy, n / vec, d);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}
