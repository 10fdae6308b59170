// Fused multi-head attention for the tiny tiles of the caption decoder (T <= 32 queries,
// S <= 56 keys, head_dim = 64): one workgroup per (batch, head); Q/K/V tiles live in LDS, the
// whole score matrix in LDS, softmax in fp32 with wave64 shuffles.  The backward kernel
// recomputes the probabilities instead of reading them from HBM.
//
//   P = dropout_p( softmax( Q K^T / sqrt(d) + causal + key_padding ) ),  O = P V
//
// Replaces aten::scaled_dot_product_attention (+backward) inside
// torch.nn.functional.multi_head_attention_forward, called by nn.TransformerDecoderLayer's
// _sa_block / _mha_block from /root/reference/virtex/modules/textual_heads.py:270-275:
// self-attention gets the causal mask (:261-265) merged with the key-padding mask
// (:255-256, -inf at j >= caption_length[b]); cross-attention over the 7x7 grid gets none.
// Q/K/V/O are addressed in place inside the packed projection outputs ([rows][ld] with a
// per-head column offset), so no head split / merge copies exist.
#include "vtx_common.h"

namespace {

constexpr int TMAX = 32, SMAX = 56, D = 64, KP = D + 1;

struct AttnArgs {
    const void *q, *k, *v;
    long ldq, ldk, ldv, ldo;  // row strides (elements)
    int T, S, heads;
    float scale;
    int causal;
    const long long* lengths;  // nullable: keys j >= lengths[b] are masked
    Dropout drop;
};

// rows x 64 tile, global (dtype, 16-byte vector loads: head columns start at multiples of 64 elements and
// row strides are multiples of 8, so every chunk is aligned) -> LDS fp32
template <class T> __device__ __forceinline__ void load_tile(float* dst, int dstride, const T* src, long ld,
                                                             int rows, int tid) {
    constexpr int VEC = Elem<T>::VEC, CPR = D / VEC;
    for (int i = tid; i < rows * CPR; i += 256) {
        const int r = i / CPR, c = (i % CPR) * VEC;
        Vec16<T> v; v.load(src + (long)r * ld + c);
#pragma unroll
        for (int j = 0; j < VEC; ++j) dst[r * dstride + c + j] = v.v[j];
    }
}
// out[r][c..c+VEC) = sum_{k<n} coef(r,k) * mat[k*mstride + c..]  -> global dtype with 16-byte stores.
// (k outer, VEC accumulators inner: every coefficient is read from LDS once per chunk)
template <class T, class F> __device__ __forceinline__ void matmul_store(T* dst, long ld, int rows, int tid, int n,
                                                                         const float* mat, int mstride, F coef) {
    constexpr int VEC = Elem<T>::VEC, CPR = D / VEC;
    for (int i = tid; i < rows * CPR; i += 256) {
        const int r = i / CPR, c = (i % CPR) * VEC;
        Vec16<T> v;
#pragma unroll
        for (int j = 0; j < VEC; ++j) v.v[j] = 0.f;
        for (int k = 0; k < n; ++k) {
            const float w = coef(r, k);
            const float* m = mat + k * mstride + c;
#pragma unroll
            for (int j = 0; j < VEC; ++j) v.v[j] += w * m[j];
        }
        v.store(dst + (long)r * ld + c);
    }
}

// scores -> P (pre-dropout softmax) in sP[T][SMAX]
__device__ __forceinline__ void scores_softmax(const float* sQ, const float* sK, float* sP, const AttnArgs& a,
                                               int len, int tid) {
    // one query x four keys per thread: each Q value is read from LDS once per four products
    const int S4 = (a.S + 3) / 4;
    for (int idx = tid; idx < a.T * S4; idx += 256) {
        const int i = idx / S4, j0 = (idx % S4) * 4;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        const float* q = sQ + i * D;
        const float* k0 = sK + j0 * KP;     // rows beyond S are inside the SMAX-row buffer (never stored)
#pragma unroll 8
        for (int d = 0; d < D; ++d) {
            const float qv = q[d];
#pragma unroll
            for (int t = 0; t < 4; ++t) s[t] += qv * k0[t * KP + d];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = j0 + t;
            if (j < a.S) sP[i * SMAX + j] = ((a.causal && j > i) || j >= len) ? -INFINITY : s[t] * a.scale;
        }
    }
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6;
    for (int i = wv; i < a.T; i += 4) {
        const float s = lane < a.S ? sP[i * SMAX + lane] : -INFINITY;
        const float m = wave_max(s);
        const float e = (lane < a.S && m > -INFINITY) ? __expf(s - m) : 0.f;
        const float sum = wave_sum(e);
        if (lane < a.S) sP[i * SMAX + lane] = sum > 0.f ? e / sum : 0.f;
    }
    __syncthreads();
}

template <class T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a, T* __restrict__ o) {
    __shared__ float sQ[TMAX * D], sK[SMAX * KP], sV[SMAX * KP], sP[TMAX * SMAX];
    const int tid = threadIdx.x;
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    load_tile<T>(sQ, D, (const T*)a.q + (long)b * a.T * a.ldq + h * D, a.ldq, a.T, tid);
    load_tile<T>(sK, KP, (const T*)a.k + (long)b * a.S * a.ldk + h * D, a.ldk, a.S, tid);
    load_tile<T>(sV, KP, (const T*)a.v + (long)b * a.S * a.ldv + h * D, a.ldv, a.S, tid);
    __syncthreads();
    const int len = a.lengths ? (int)a.lengths[b] : a.S;
    scores_softmax(sQ, sK, sP, a, len, tid);
    const uint64_t pbase = (uint64_t)blockIdx.x * (TMAX * SMAX);
    // apply the attention dropout once, in place, then O = Pd V with 16-byte stores
    if (a.drop.thresh) {
        for (int idx = tid; idx < a.T * a.S; idx += 256) {
            const int i = idx / a.S, j = idx % a.S;
            sP[i * SMAX + j] = a.drop.apply(sP[i * SMAX + j], pbase + i * SMAX + j);
        }
        __syncthreads();
    }
    matmul_store<T>(o + (long)b * a.T * a.ldo + h * D, a.ldo, a.T, tid, a.S, sV, KP,
                    [&](int i, int j) { return sP[i * SMAX + j]; });
}

template <class T>
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnArgs a, const T* __restrict__ dout, T* __restrict__ dq,
                                                       T* __restrict__ dk, T* __restrict__ dv, long lddq,
                                                       long lddk, long lddv) {
    __shared__ float sQ[TMAX * D], sO[TMAX * D], sK[SMAX * KP], sV[SMAX * KP], sP[TMAX * SMAX], sG[TMAX * SMAX];
    const int tid = threadIdx.x;
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    load_tile<T>(sQ, D, (const T*)a.q + (long)b * a.T * a.ldq + h * D, a.ldq, a.T, tid);
    load_tile<T>(sO, D, dout + (long)b * a.T * a.ldo + h * D, a.ldo, a.T, tid);
    load_tile<T>(sK, KP, (const T*)a.k + (long)b * a.S * a.ldk + h * D, a.ldk, a.S, tid);
    load_tile<T>(sV, KP, (const T*)a.v + (long)b * a.S * a.ldv + h * D, a.ldv, a.S, tid);
    __syncthreads();
    const int len = a.lengths ? (int)a.lengths[b] : a.S;
    scores_softmax(sQ, sK, sP, a, len, tid);
    const uint64_t pbase = (uint64_t)blockIdx.x * (TMAX * SMAX);
    // dV[j][d] = sum_i Pd[i][j] dO[i][d]
    matmul_store<T>(dv + (long)b * a.S * lddv + h * D, lddv, a.S, tid, a.T, sO, D,
                    [&](int j, int i) { return a.drop.apply(sP[i * SMAX + j], pbase + i * SMAX + j); });
    // dP[i][j] = dropout'( sum_d dO[i][d] V[j][d] )
    const int S4 = (a.S + 3) / 4;
    for (int idx = tid; idx < a.T * S4; idx += 256) {
        const int i = idx / S4, j0 = (idx % S4) * 4;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float* go = sO + i * D;
        const float* v0 = sV + j0 * KP;
#pragma unroll 8
        for (int d = 0; d < D; ++d) {
            const float gv = go[d];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] += gv * v0[t * KP + d];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (j0 + t < a.S) sG[i * SMAX + j0 + t] = a.drop.apply(acc[t], pbase + i * SMAX + j0 + t);
    }
    __syncthreads();
    // dS = P * (dP - rowsum(dP * P)) * scale   (written over sG)
    const int lane = tid & 63, wv = tid >> 6;
    for (int i = wv; i < a.T; i += 4) {
        const float p = lane < a.S ? sP[i * SMAX + lane] : 0.f;
        const float g = lane < a.S ? sG[i * SMAX + lane] : 0.f;
        const float dot = wave_sum(p * g);
        if (lane < a.S) sG[i * SMAX + lane] = p * (g - dot) * a.scale;
    }
    __syncthreads();
    matmul_store<T>(dq + (long)b * a.T * lddq + h * D, lddq, a.T, tid, a.S, sK, KP,
                    [&](int i, int j) { return sG[i * SMAX + j]; });
    matmul_store<T>(dk + (long)b * a.S * lddk + h * D, lddk, a.S, tid, a.T, sQ, D,
                    [&](int j, int i) { return sG[i * SMAX + j]; });
}

static int check(const char* who, int dtype, int B, int heads, int T, int S, int head_dim) {
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "%s: bad dtype", who);
    VTX_CHECK(B >= 0 && heads > 0 && T > 0 && S > 0, VTX_ERR_ARG, "%s: bad shape", who);
    VTX_CHECK(head_dim == D && T <= TMAX && S <= SMAX, VTX_ERR_SHAPE,
              "%s: supports head_dim == 64, T <= %d, S <= %d (got d=%d T=%d S=%d)", who, TMAX, SMAX, head_dim, T, S);
    return VTX_OK;
}

}  // namespace

extern "C" int vtx_attention_fwd(int dtype, const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                                 void* o, long ldo, int B, int heads, int T, int S, int head_dim, int causal,
                                 const long long* key_lengths, float p_drop, uint64_t seed, void* stream) {
    VTX_CHECK(q && k && v && o, VTX_ERR_ARG, "attention_fwd: null pointer");
    int rc = check("attention_fwd", dtype, B, heads, T, S, head_dim);
    if (rc) return rc;
    if (B == 0) return VTX_OK;
    AttnArgs a{q, k, v, ldq, ldk, ldv, ldo, T, S, heads, 1.0f / sqrtf((float)head_dim), causal, key_lengths,
               make_dropout(p_drop, seed)};
    dim3 grid(B * heads), block(256);
    if (dtype == VTX_BF16) hipLaunchKernelGGL((attn_fwd_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, a, (bf16_t*)o);
    else hipLaunchKernelGGL((attn_fwd_kernel<float>), grid, block, 0, (hipStream_t)stream, a, (float*)o);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

extern "C" int vtx_attention_bwd(int dtype, const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                                 const void* dout, long ldo, void* dq, long lddq, void* dk, long lddk, void* dv,
                                 long lddv, int B, int heads, int T, int S, int head_dim, int causal,
                                 const long long* key_lengths, float p_drop, uint64_t seed, void* stream) {
    VTX_CHECK(q && k && v && dout && dq && dk && dv, VTX_ERR_ARG, "attention_bwd: null pointer");
    int rc = check("attention_bwd", dtype, B, heads, T, S, head_dim);
    if (rc) return rc;
    if (B == 0) return VTX_OK;
    AttnArgs a{q, k, v, ldq, ldk, ldv, ldo, T, S, heads, 1.0f / sqrtf((float)head_dim), causal, key_lengths,
               make_dropout(p_drop, seed)};
    dim3 grid(B * heads), block(256);
    if (dtype == VTX_BF16)
        hipLaunchKernelGGL((attn_bwd_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, a, (const bf16_t*)dout,
                           (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, lddq, lddk, lddv);
    else
        hipLaunchKernelGGL((attn_bwd_kernel<float>), grid, block, 0, (hipStream_t)stream, a, (const float*)dout,
                           (float*)dq, (float*)dk, (float*)dv, lddq, lddk, lddv);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}
