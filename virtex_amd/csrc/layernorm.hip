// Fused residual + dropout + LayerNorm, forward and backward (HBM-bound; one wave64 per row).
//
//   z   = x + dropout_p(y)            (y optional: plain LayerNorm when y == nullptr)
//   out = (z - mean) * rstd * gamma + beta
//
// Reference: torch.nn.TransformerDecoderLayer post-norm blocks
// `x = norm(x + dropout(sublayer(x)))` reached from
// /root/reference/virtex/modules/textual_heads.py:181-194,270-275 (eps 1e-5).
// Rows are kept in registers (NV 16-byte vectors per lane), statistics in fp32, two-pass
// variance like ATen's CPU kernel.
#include "vtx_common.h"

namespace {

template <class T, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(
    const T* __restrict__ x, const T* __restrict__ y, const float* __restrict__ gamma,
    const float* __restrict__ beta, T* __restrict__ out, float* __restrict__ mean_out,
    float* __restrict__ rstd_out, int rows, int H, float eps, Dropout drop) {
    constexpr int VEC = Elem<T>::VEC;
    drop = drop.resolved();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;  // whole wave exits together (row is wave-uniform)
    const size_t base = (size_t)row * H;
    // every vector of the row (and of the residual branch) is requested before any is used (vtx_loads_issued, vtx_common.h)
    uint4 xr[NV], yr[NV];
    const T* yp = y ? y : x;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * VEC, cc = col < H ? col : 0;
        xr[i] = *reinterpret_cast<const uint4*>(x + base + cc);
        yr[i] = *reinterpret_cast<const uint4*>(yp + base + cc);
    }
    // gamma / beta too: loaded where they are used (behind the two wave reductions) they were one more memory latency in a
    // kernel that IS a chain of latencies (one round of 7 680 one-row waves)
    float4 gq[NV][VEC / 4], bq[NV][VEC / 4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * VEC, cc = col < H ? col : 0;
#pragma unroll
        for (int j = 0; j < VEC / 4; ++j) {
            gq[i][j] = *reinterpret_cast<const float4*>(gamma + cc + 4 * j);
            bq[i][j] = *reinterpret_cast<const float4*>(beta + cc + 4 * j);
        }
    }
    vtx_loads_issued();
    Vec16<T> z[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * VEC;
        if (col < H) {
            vtx_unpack_raw16<T>(xr[i], z[i].v);
            if (y) {
                float t[VEC];
                vtx_unpack_raw16<T>(yr[i], t);
#pragma unroll
                for (int j = 0; j < VEC; ++j) z[i].v[j] += drop.apply(t[j], base + col + j);
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) s += z[i].v[j];
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) z[i].v[j] = 0.f;
        }
    }
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * VEC;
        if (col < H) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) { const float d = z[i].v[j] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * VEC;
        if (col < H) {
            float ga[VEC], be[VEC];
#pragma unroll
            for (int j = 0; j < VEC; j += 4) {
                const float4 g4 = gq[i][j / 4], b4 = bq[i][j / 4];
                ga[j] = g4.x; ga[j + 1] = g4.y; ga[j + 2] = g4.z; ga[j + 3] = g4.w;
                be[j] = b4.x; be[j + 1] = b4.y; be[j + 2] = b4.z; be[j + 3] = b4.w;
            }
            Vec16<T> o;
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                o.v[j] = (z[i].v[j] - mean) * rstd * ga[j] + be[j];
            o.store(out + base + col);
        }
    }
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// dz = rstd * (g - mean_H(g) - xhat * mean_H(g * xhat)),  g = dout * gamma
// dy = dropout-mask(dz)  (written only when dy != nullptr)
// dgamma += sum_rows dout * xhat ; dbeta += sum_rows dout      (fp32 atomics, once per block)
template <class T, int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(
    const T* __restrict__ x, const T* __restrict__ y, const float* __restrict__ gamma,
    const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
    const T* __restrict__ dout, T* __restrict__ dz, T* __restrict__ dy,
    float* __restrict__ partials, int rows, int H, Dropout drop) {
    constexpr int VEC = Elem<T>::VEC;
    drop = drop.resolved();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float ag[NV][VEC], ab[NV][VEC];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < VEC; ++j) ag[i][j] = ab[i][j] = 0.f;

    // gamma is the same for every row: once per wave, in front of the loop (it was re-requested per row, behind the row's loads)
    float gam[NV][VEC];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * VEC, cc = col < H ? col : 0;
#pragma unroll
        for (int j = 0; j < VEC; j += 4) {
            const float4 g4 = *reinterpret_cast<const float4*>(gamma + cc + j);
            gam[i][j] = g4.x; gam[i][j + 1] = g4.y; gam[i][j + 2] = g4.z; gam[i][j + 3] = g4.w;
        }
    }
    // A wave walks its rows with the NEXT row's vectors (and statistics) requested before the current row is worked on:
    // a row is load -> two wave reductions -> store, and without the prefetch a wave's rows were that chain back to back
    // (2.7 TB/s on 63 MB: profiles/r04_bench_default.json).
    const T* yp = y ? y : x;
    const int row0 = blockIdx.x * 4 + wv, rstep = gridDim.x * 4;
    uint4 nx[NV], ny[NV], ng[NV];
    float nmean = 0.f, nrstd = 0.f;
    if (row0 < rows) {
        const size_t b0 = (size_t)row0 * H;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (i * 64 + lane) * VEC, cc = col < H ? col : 0;
            nx[i] = *reinterpret_cast<const uint4*>(x + b0 + cc);
            ny[i] = *reinterpret_cast<const uint4*>(yp + b0 + cc);
            ng[i] = *reinterpret_cast<const uint4*>(dout + b0 + cc);
        }
        nmean = mean_in[row0]; nrstd = rstd_in[row0];
    }
    for (int row = row0; row < rows; row += rstep) {
        const size_t base = (size_t)row * H;
        const float mean = nmean, rstd = nrstd;
        uint4 xr[NV], yr[NV], gr[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) { xr[i] = nx[i]; yr[i] = ny[i]; gr[i] = ng[i]; }
        const int nrow = row + rstep;
        if (nrow < rows) {                                   // (wave-uniform)
            const size_t nb = (size_t)nrow * H;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int col = (i * 64 + lane) * VEC, cc = col < H ? col : 0;
                nx[i] = *reinterpret_cast<const uint4*>(x + nb + cc);
                ny[i] = *reinterpret_cast<const uint4*>(yp + nb + cc);
                ng[i] = *reinterpret_cast<const uint4*>(dout + nb + cc);
            }
            nmean = mean_in[nrow]; nrstd = rstd_in[nrow];
        }
        vtx_loads_issued();
        Vec16<T> xh[NV], g[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (i * 64 + lane) * VEC;
            if (col < H) {
                vtx_unpack_raw16<T>(xr[i], xh[i].v);
                if (y) {
                    float t[VEC];
                    vtx_unpack_raw16<T>(yr[i], t);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) xh[i].v[j] += drop.apply(t[j], base + col + j);
                }
                vtx_unpack_raw16<T>(gr[i], g[i].v);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float h = (xh[i].v[j] - mean) * rstd;
                    const float d = g[i].v[j];
                    ag[i][j] += d * h; ab[i][j] += d;
                    const float gg = d * gam[i][j];
                    xh[i].v[j] = h; g[i].v[j] = gg;
                    s1 += gg; s2 += gg * h;
                }
            }
        }
        s1 = wave_sum(s1) / (float)H;
        s2 = wave_sum(s2) / (float)H;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (i * 64 + lane) * VEC;
            if (col < H) {
                Vec16<T> o;
#pragma unroll
                for (int j = 0; j < VEC; ++j) o.v[j] = rstd * (g[i].v[j] - s1 - xh[i].v[j] * s2);
                o.store(dz + base + col);
                if (dy) {
                    Vec16<T> m;
#pragma unroll
                    for (int j = 0; j < VEC; ++j) m.v[j] = drop.apply(o.v[j], base + col + j);
                    m.store(dy + base + col);
                }
            }
        }
    }
    // block-level combine of the 4 waves' column partials through LDS, then one atomic/column
    __shared__ float red[4][64 * VEC + 1];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * VEC;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < VEC; ++j) red[wv][lane * VEC + j] = pass ? ab[i][j] : ag[i][j];
            __syncthreads();
            if (wv == 0 && col < H) {
                // per-block partial (no atomics: a thousand blocks adding into the same 2*H addresses
                // serialise in L2); summed by ln_bwd_finalize_kernel
                float* dst = partials + ((size_t)blockIdx.x * 2 + pass) * H;
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    dst[col + j] = red[0][lane * VEC + j] + red[1][lane * VEC + j] + red[2][lane * VEC + j] +
                                   red[3][lane * VEC + j];
            }
        }
    }
}

// dgamma[c] += sum_b partials[b][0][c] ; dbeta[c] += sum_b partials[b][1][c]   (32 columns x 8 groups)
__global__ void ln_bwd_finalize_kernel(const float* __restrict__ partials, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, int H, int nparts) {
    __shared__ float red[2][8][32];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5;
    float a = 0.f, b = 0.f;
    if (c < H) {
        int p = grp;
        for (; p + 24 < nparts; p += 32) {       // four strips (eight loads) in flight, added in the same order as one by one
            float ta[4], tb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { ta[u] = partials[((size_t)(p + 8 * u) * 2) * H + c]; tb[u] = partials[((size_t)(p + 8 * u) * 2 + 1) * H + c]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { a += ta[u]; b += tb[u]; }
        }
        for (; p < nparts; p += 8) { a += partials[((size_t)p * 2) * H + c]; b += partials[((size_t)p * 2 + 1) * H + c]; }
    }
    red[0][grp][threadIdx.x & 31] = a; red[1][grp][threadIdx.x & 31] = b;
    __syncthreads();
    if (grp == 0 && c < H) {
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) { t0 += red[0][g][threadIdx.x & 31]; t1 += red[1][g][threadIdx.x & 31]; }
        dgamma[c] += t0; dbeta[c] += t1;
    }
}

constexpr int VTX_LN_MAX_PARTS = 512;       // blocks (= dgamma / dbeta partial rows) of the backward kernel; 1 024 measured: the kernel 28.7 -> 27 us, its finalize 7 -> 13 us

template <class T>
int ln_fwd_t(const void* x, const void* y, const float* gamma, const float* beta, void* out,
             float* mean, float* rstd, int rows, int H, float eps, Dropout d, hipStream_t st) {
    constexpr int VEC = Elem<T>::VEC;
    const int nv = vtx_cdiv(H, 64 * VEC);
    dim3 grid(vtx_cdiv(rows, 4)), block(256);
#define VTX_LN_FWD(NV)                                                                          \
    VTX_KLAUNCH("layernorm_fwd", 0, (double)rows * H * sizeof(T) * (y ? 3 : 2), (ln_fwd_kernel<T, NV>), grid, block, 0, st, (const T*)x, (const T*)y,    \
                       gamma, beta, (T*)out, mean, rstd, rows, H, eps, d)
    if (nv <= 1) VTX_LN_FWD(1);
    else if (nv <= 2) VTX_LN_FWD(2);
    else if (nv <= 4) VTX_LN_FWD(4);
    else VTX_LN_FWD(8);
#undef VTX_LN_FWD
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

template <class T>
int ln_bwd_t(const void* x, const void* y, const float* gamma, const float* mean, const float* rstd,
             const void* dout, void* dz, void* dy, float* dgamma, float* dbeta, float* partials, int rows,
             int H, Dropout d, hipStream_t st) {
    constexpr int VEC = Elem<T>::VEC;
    const int nv = vtx_cdiv(H, 64 * VEC);
    int nblk = vtx_cdiv(rows, 4);
    if (nblk > VTX_LN_MAX_PARTS) nblk = VTX_LN_MAX_PARTS;
    dim3 grid(nblk), block(256);
#define VTX_LN_BWD(NV)                                                                          \
    VTX_KLAUNCH("layernorm_bwd", 0, (double)rows * H * sizeof(T) * ((y ? 3 : 2) + 1 + (dy ? 1 : 0)), (ln_bwd_kernel<T, NV>), grid, block, 0, st, (const T*)x, (const T*)y,    \
                       gamma, mean, rstd, (const T*)dout, (T*)dz, (T*)dy, partials, rows, H, d)
    if (nv <= 1) VTX_LN_BWD(1);
    else if (nv <= 2) VTX_LN_BWD(2);
    else if (nv <= 4) VTX_LN_BWD(4);
    else VTX_LN_BWD(8);
#undef VTX_LN_BWD
    VTX_KLAUNCH("layernorm_bwd_finalize", 0, 8.0 * nblk * H, ln_bwd_finalize_kernel, dim3(vtx_cdiv(H, 32)), dim3(256), 0, st, partials, dgamma, dbeta, H, nblk);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

}  // namespace

extern "C" int vtx_layernorm_residual_fwd(int dtype, const void* x, const void* y,
                                          const float* gamma, const float* beta, void* out,
                                          float* mean, float* rstd, int rows, int H, float eps,
                                          float p_drop, uint64_t seed, void* stream) {
    VTX_CHECK(x && gamma && beta && out && mean && rstd, VTX_ERR_ARG, "layernorm_fwd: null pointer");
    VTX_CHECK(rows >= 0 && H > 0, VTX_ERR_ARG, "layernorm_fwd: bad shape rows=%d H=%d", rows, H);
    const int vec = dtype == VTX_BF16 ? 8 : 4;
    VTX_CHECK(H % vec == 0 && H <= 64 * vec * 8, VTX_ERR_SHAPE,
              "layernorm_fwd: H=%d must be a multiple of %d and <= %d", H, vec, 64 * vec * 8);
    if (rows == 0) return VTX_OK;
    Dropout d = make_dropout(y ? p_drop : 0.f, seed);
    if (dtype == VTX_BF16)
        return ln_fwd_t<bf16_t>(x, y, gamma, beta, out, mean, rstd, rows, H, eps, d, (hipStream_t)stream);
    if (dtype == VTX_F32)
        return ln_fwd_t<float>(x, y, gamma, beta, out, mean, rstd, rows, H, eps, d, (hipStream_t)stream);
    VTX_CHECK(false, VTX_ERR_DTYPE, "layernorm_fwd: bad dtype %d", dtype);
}

extern "C" long vtx_layernorm_workspace_floats(int H) { return (long)VTX_LN_MAX_PARTS * 2 * H; }

extern "C" int vtx_layernorm_residual_bwd(int dtype, const void* x, const void* y,
                                          const float* gamma, const float* mean, const float* rstd,
                                          const void* dout, void* dz, void* dy, float* dgamma,
                                          float* dbeta, float* workspace, int rows, int H, float p_drop,
                                          uint64_t seed, void* stream) {
    VTX_CHECK(x && gamma && mean && rstd && dout && dz && dgamma && dbeta && workspace, VTX_ERR_ARG,
              "layernorm_bwd: null pointer");
    const int vec = dtype == VTX_BF16 ? 8 : 4;
    VTX_CHECK(H > 0 && H % vec == 0 && H <= 64 * vec * 8, VTX_ERR_SHAPE, "layernorm_bwd: bad H=%d", H);
    if (rows == 0) return VTX_OK;
    Dropout d = make_dropout(y ? p_drop : 0.f, seed);
    if (dtype == VTX_BF16)
        return ln_bwd_t<bf16_t>(x, y, gamma, mean, rstd, dout, dz, dy, dgamma, dbeta, workspace, rows, H, d, (hipStream_t)stream);
    if (dtype == VTX_F32)
        return ln_bwd_t<float>(x, y, gamma, mean, rstd, dout, dz, dy, dgamma, dbeta, workspace, rows, H, d, (hipStream_t)stream);
    VTX_CHECK(false, VTX_ERR_DTYPE, "layernorm_bwd: bad dtype %d", dtype);
}
