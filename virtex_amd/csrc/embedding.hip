// WordAndPositionalEmbedding, forward and backward, one wave64 per token (HBM-bound gathers).
//
//   e   = words[tok] + positions[t]
//   out = (tok != padding_idx) * dropout_p( LayerNorm_eps(e) * gamma + beta )
//
// Reference: /root/reference/virtex/modules/embedding.py:46-74 (eps = 1e-8, dropout after the
// norm, padded positions zeroed last).  Backward scatters into the TIED word matrix with fp32
// atomics, skipping padding_idx rows exactly like aten::embedding_dense_backward does
// (nn.Embedding(padding_idx=...), embedding.py:36); positions have no padding index.
#include "vtx_common.h"

namespace {

template <int NV> struct RowF32 {  // H fp32 values of one row, 4 per lane per chunk
    float v[NV][4];
};

__device__ __forceinline__ void emb_ld4(const float* p, float* v) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
__device__ __forceinline__ void emb_ld4(const bf16_t* p, float* v) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
__device__ __forceinline__ void emb_st4(float* p, const float* v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void emb_st4(bf16_t* p, const float* v) { *reinterpret_cast<uint2*>(p) = make_uint2(f2bf2(v[0], v[1]), f2bf2(v[2], v[3])); }

template <class T, int NV>
__global__ __launch_bounds__(256) void embed_fwd_kernel(
    const long long* __restrict__ tokens, const float* __restrict__ words, const float* __restrict__ pos,
    const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ out,
    float* __restrict__ mean_out, float* __restrict__ rstd_out, int rows, int Tlen, int H, int V,
    int padding_idx, float eps, Dropout drop) {
    drop = drop.resolved();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    long long tok = tokens[row];
    if (tok < 0 || tok >= V) tok = padding_idx;  // defensive: never read out of the table
    const int t = row % Tlen;
    const float* wrow = words + (size_t)tok * H;
    const float* prow = pos + (size_t)t * H;
    // every vector of the two table rows is requested before any is used (vtx_loads_issued, vtx_common.h)
    float4 wa[NV], pa[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * 4, cc = col < H ? col : 0;
        wa[i] = *reinterpret_cast<const float4*>(wrow + cc);
        pa[i] = *reinterpret_cast<const float4*>(prow + cc);
    }
    vtx_loads_issued();
    RowF32<NV> e;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * 4;
        if (col < H) {
            const float4 a = wa[i], b = pa[i];
            e.v[i][0] = a.x + b.x; e.v[i][1] = a.y + b.y; e.v[i][2] = a.z + b.z; e.v[i][3] = a.w + b.w;
            s += e.v[i][0] + e.v[i][1] + e.v[i][2] + e.v[i][3];
        } else {
            e.v[i][0] = e.v[i][1] = e.v[i][2] = e.v[i][3] = 0.f;
        }
    }
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * 4;
        if (col < H) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = e.v[i][j] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
    const float keep = tok != padding_idx ? 1.f : 0.f;
    const size_t base = (size_t)row * H;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * 4;
        if (col < H) {
            const float4 g4 = *reinterpret_cast<const float4*>(gamma + col), b4 = *reinterpret_cast<const float4*>(beta + col);
            const float ga[4] = {g4.x, g4.y, g4.z, g4.w}, be[4] = {b4.x, b4.y, b4.z, b4.w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float y = (e.v[i][j] - mean) * rstd * ga[j] + be[j];
                o[j] = keep * drop.apply(y, base + col + j);
            }
            emb_st4(out + base + col, o);
        }
    }
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// grid = (Tlen, GY): blocks with the same blockIdx.x own position t and stride over the batch,
// so the positional / gamma / beta gradients are reduced in registers before any atomic.
template <class T, int NV>
__global__ __launch_bounds__(256) void embed_bwd_kernel(
    const long long* __restrict__ tokens, const float* __restrict__ words, const float* __restrict__ pos,
    const float* __restrict__ gamma, const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
    const T* __restrict__ dout, float* __restrict__ dwords, float* __restrict__ dpos,
    float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int Tlen, int H, int V, int padding_idx,
    Dropout drop) {
    drop = drop.resolved();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int t = blockIdx.x;
    const float* prow = pos + (size_t)t * H;
    __shared__ __attribute__((aligned(16))) float stage[4][256];
    RowF32<NV> ap, ag, ab;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) ap.v[i][j] = ag.v[i][j] = ab.v[i][j] = 0.f;

    for (int b = blockIdx.y * 4 + wv; b < B; b += gridDim.y * 4) {
        const int row = b * Tlen + t;
        long long tok = tokens[row];
        if (tok < 0 || tok >= V) tok = padding_idx;
        if (tok == padding_idx) continue;  // output was masked to zero: no gradient anywhere (wave-uniform)
        const float* wrow = words + (size_t)tok * H;
        const float mean = mean_in[row], rstd = rstd_in[row];
        const size_t base = (size_t)row * H;
        RowF32<NV> xh, g;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (i * 64 + lane) * 4;
            if (col < H) {
                const float4 a = *reinterpret_cast<const float4*>(wrow + col);
                const float4 p = *reinterpret_cast<const float4*>(prow + col);
                const float ev[4] = {a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w};
                float dv[4];
                emb_ld4(dout + base + col, dv);                 // one 8- / 16-byte load, not four element loads
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float h = (ev[j] - mean) * rstd;
                    const float d = drop.apply(dv[j], base + col + j);
                    ag.v[i][j] += d * h; ab.v[i][j] += d;
                    const float gg = d * gamma[col + j];
                    xh.v[i][j] = h; g.v[i][j] = gg;
                    s1 += gg; s2 += gg * h;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) xh.v[i][j] = g.v[i][j] = 0.f;
            }
        }
        s1 = wave_sum(s1) / (float)H;
        s2 = wave_sum(s2) / (float)H;
        float* dwrow = dwords + (size_t)tok * H;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = (i * 64 + lane) * 4;
            float de[4] = {0.f, 0.f, 0.f, 0.f};
            if (col < H) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    de[j] = rstd * (g.v[i][j] - s1 - xh.v[i][j] * s2);
                    ap.v[i][j] += de[j];
                }
            }
            // The scatter into the tied matrix: a lane holds 4 consecutive columns, so `atomicAdd(dwrow + col + j)` makes
            // every wave-instruction touch 64 floats 16 bytes apart (eight cache lines, eight floats each).  Through a
            // wave-private LDS row the same values go out with lane-contiguous addresses: 256 contiguous bytes per
            // instruction, a quarter of the requests the L2 atomic units see.
            *reinterpret_cast<float4*>(&stage[wv][lane * 4]) = make_float4(de[0], de[1], de[2], de[3]);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = i * 256 + j * 64 + lane;
                if (c < H) atomicAdd(dwrow + c, stage[wv][j * 64 + lane]);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __shared__ float red[4][256 + 1];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * 4;
#pragma unroll
        for (int pass = 0; pass < 3; ++pass) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j)
                red[wv][lane * 4 + j] = pass == 0 ? ap.v[i][j] : (pass == 1 ? ag.v[i][j] : ab.v[i][j]);
            __syncthreads();
            if (wv == 0 && col < H) {
                float* dst = pass == 0 ? dpos + (size_t)t * H : (pass == 1 ? dgamma : dbeta);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    atomicAdd(dst + col + j, red[0][lane * 4 + j] + red[1][lane * 4 + j] + red[2][lane * 4 + j] +
                                                 red[3][lane * 4 + j]);
            }
        }
    }
}

}  // namespace

#define VTX_EMB_DISPATCH(FAM, BYTES, KERNEL, T, ...)                                                        \
    do {                                                                                                    \
        const int nv_ = vtx_cdiv(H, 256);                                                                   \
        if (nv_ <= 1) VTX_KLAUNCH(FAM, 0, BYTES, (KERNEL<T, 1>), grid, block, 0, st, __VA_ARGS__);          \
        else if (nv_ <= 2) VTX_KLAUNCH(FAM, 0, BYTES, (KERNEL<T, 2>), grid, block, 0, st, __VA_ARGS__);     \
        else if (nv_ <= 4) VTX_KLAUNCH(FAM, 0, BYTES, (KERNEL<T, 4>), grid, block, 0, st, __VA_ARGS__);     \
        else VTX_KLAUNCH(FAM, 0, BYTES, (KERNEL<T, 8>), grid, block, 0, st, __VA_ARGS__);                   \
    } while (0)

extern "C" int vtx_embedding_fwd(int dtype, const long long* tokens, const float* words, const float* positions,
                                 const float* gamma, const float* beta, void* out, float* mean, float* rstd,
                                 int B, int T, int H, int V, int padding_idx, float eps, float p_drop,
                                 uint64_t seed, void* stream) {
    VTX_CHECK(tokens && words && positions && gamma && beta && out && mean && rstd, VTX_ERR_ARG, "embedding_fwd: null pointer");
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "embedding_fwd: bad dtype");
    VTX_CHECK(B >= 0 && T > 0 && H > 0 && H % 8 == 0 && H <= 2048, VTX_ERR_SHAPE, "embedding_fwd: H=%d must be a multiple of 8, <= 2048", H);
    if (B == 0) return VTX_OK;
    hipStream_t st = (hipStream_t)stream;
    const int rows = B * T;
    dim3 grid(vtx_cdiv(rows, 4)), block(256);
    Dropout d = make_dropout(p_drop, seed);
    if (dtype == VTX_BF16)
        VTX_EMB_DISPATCH("embedding_fwd", (double)rows * H * (4.0 + 2.0), embed_fwd_kernel, bf16_t, tokens, words, positions, gamma, beta, (bf16_t*)out, mean, rstd, rows, T, H, V, padding_idx, eps, d);
    else
        VTX_EMB_DISPATCH("embedding_fwd", (double)rows * H * (4.0 + 4.0), embed_fwd_kernel, float, tokens, words, positions, gamma, beta, (float*)out, mean, rstd, rows, T, H, V, padding_idx, eps, d);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

extern "C" int vtx_embedding_bwd(int dtype, const long long* tokens, const float* words, const float* positions,
                                 const float* gamma, const float* mean, const float* rstd, const void* dout,
                                 float* dwords, float* dpositions, float* dgamma, float* dbeta, int B, int T,
                                 int H, int V, int padding_idx, float p_drop, uint64_t seed, void* stream) {
    VTX_CHECK(tokens && words && positions && gamma && mean && rstd && dout && dwords && dpositions && dgamma && dbeta,
              VTX_ERR_ARG, "embedding_bwd: null pointer");
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "embedding_bwd: bad dtype");
    VTX_CHECK(B >= 0 && T > 0 && H > 0 && H % 8 == 0 && H <= 2048, VTX_ERR_SHAPE, "embedding_bwd: bad H=%d", H);
    if (B == 0) return VTX_OK;
    hipStream_t st = (hipStream_t)stream;
    int gy = vtx_cdiv(B, 4);
    if (gy > 8) gy = 8;      // T*gy blocks add into dgamma/dbeta: keep the atomic fan-in small
    dim3 grid(T, gy), block(256);
    Dropout d = make_dropout(p_drop, seed);
    if (dtype == VTX_BF16)
        VTX_EMB_DISPATCH("embedding_bwd", (double)B * T * H * (2.0 + 4.0 + 8.0), embed_bwd_kernel, bf16_t, tokens, words, positions, gamma, mean, rstd, (const bf16_t*)dout, dwords, dpositions, dgamma, dbeta, B, T, H, V, padding_idx, d);
    else
        VTX_EMB_DISPATCH("embedding_bwd", (double)B * T * H * (4.0 + 4.0 + 8.0), embed_bwd_kernel, float, tokens, words, positions, gamma, mean, rstd, (const float*)dout, dwords, dpositions, dgamma, dbeta, B, T, H, V, padding_idx, d);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}
