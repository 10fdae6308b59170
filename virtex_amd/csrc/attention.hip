// Fused multi-head attention for the tiny tiles of the caption decoder (tuned for T <= 32 queries,
// S <= 56 keys, head_dim = 64; larger tiles: the general kernels below): one workgroup per (batch, head); Q/K/V tiles live in LDS, the
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
#include <atomic>

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
    a.drop = a.drop.resolved();
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
    a.drop = a.drop.resolved();
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

// ---------------------------------------------------------------------------------------------------------
// Shapes outside the tuned envelope (T > 32 queries or S > 56 keys: another crop size -- 256 x 256 images give an 8 x 8 grid --
// or a longer caption limit): the same algorithm with run-time tile sizes in dynamic LDS and a softmax that walks a row in
// steps of 64 lanes.  A correctness path (the reference takes any shape: scaled_dot_product_attention), not a tuned one;
// both compute dtypes; bounded by LDS: 4 * (T*64 + 2*S'*65 + T*S') bytes forward, 4 * (2*T*64 + 2*S'*65 + 2*T*S') backward
// (S' = S rounded up to 4) must fit 160 KiB -- e.g. T = 32 with S <= 288 forward / 224 backward (448 x 448 images: S = 196).
// The dropout mask of element (i, j) of block `bh` is drawn at index bh*T*S + i*S + j in both directions.
struct BigLds { float *q, *o, *k, *v, *p, *g; };
__device__ __forceinline__ BigLds big_carve(float* base, int T, int S, bool bwd) {
    const int S4r = (S + 3) / 4 * 4;
    BigLds l;
    l.q = base; base += T * D;
    l.o = base; if (bwd) base += T * D;
    l.k = base; base += S4r * KP;
    l.v = base; base += S4r * KP;
    l.p = base; base += T * S;
    l.g = base;
    return l;
}
static size_t big_lds_bytes(int T, int S, bool bwd) {
    const size_t S4r = (size_t)(S + 3) / 4 * 4;
    return 4 * ((bwd ? 2 : 1) * (size_t)T * D + 2 * S4r * KP + (bwd ? 2 : 1) * (size_t)T * S);
}

__device__ __forceinline__ void big_scores_softmax(const BigLds& l, const AttnArgs& a, int len, int tid) {
    const int S = a.S, S4 = (S + 3) / 4;
    for (int idx = tid; idx < a.T * S4; idx += 256) {
        const int i = idx / S4, j0 = (idx % S4) * 4;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        const float* q = l.q + i * D;
        const float* k0 = l.k + j0 * KP;    // rows S .. S' exist in the buffer (whatever they hold is never stored)
#pragma unroll 8
        for (int d = 0; d < D; ++d) {
            const float qv = q[d];
#pragma unroll
            for (int t = 0; t < 4; ++t) s[t] += qv * k0[t * KP + d];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = j0 + t;
            if (j < S) l.p[i * S + j] = ((a.causal && j > i) || j >= len) ? -INFINITY : s[t] * a.scale;
        }
    }
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6;
    for (int i = wv; i < a.T; i += 4) {
        float* row = l.p + i * S;
        float m = -INFINITY;
        for (int j = lane; j < S; j += 64) m = fmaxf(m, row[j]);
        m = wave_max(m);
        float sum = 0.f;
        for (int j = lane; j < S; j += 64) {
            const float e = m > -INFINITY ? __expf(row[j] - m) : 0.f;
            row[j] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        const float inv = sum > 0.f ? 1.f / sum : 0.f;
        for (int j = lane; j < S; j += 64) row[j] *= inv;
    }
    __syncthreads();
}

template <class T>
__global__ __launch_bounds__(256) void attn_fwd_big_kernel(AttnArgs a, T* __restrict__ o) {
    a.drop = a.drop.resolved();
    HIP_DYNAMIC_SHARED(float, smem)
    const BigLds l = big_carve(smem, a.T, a.S, false);
    const int tid = threadIdx.x, S = a.S;
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    load_tile<T>(l.q, D, (const T*)a.q + (long)b * a.T * a.ldq + h * D, a.ldq, a.T, tid);
    load_tile<T>(l.k, KP, (const T*)a.k + (long)b * S * a.ldk + h * D, a.ldk, S, tid);
    load_tile<T>(l.v, KP, (const T*)a.v + (long)b * S * a.ldv + h * D, a.ldv, S, tid);
    __syncthreads();
    const int len = a.lengths ? (int)a.lengths[b] : S;
    big_scores_softmax(l, a, len, tid);
    const uint64_t pbase = (uint64_t)blockIdx.x * (uint64_t)(a.T * S);
    if (a.drop.thresh) {
        for (int idx = tid; idx < a.T * S; idx += 256) l.p[idx] = a.drop.apply(l.p[idx], pbase + idx);
        __syncthreads();
    }
    matmul_store<T>(o + (long)b * a.T * a.ldo + h * D, a.ldo, a.T, tid, S, l.v, KP,
                    [&](int i, int j) { return l.p[i * S + j]; });
}

template <class T>
__global__ __launch_bounds__(256) void attn_bwd_big_kernel(AttnArgs a, const T* __restrict__ dout, T* __restrict__ dq,
                                                           T* __restrict__ dk, T* __restrict__ dv, long lddq,
                                                           long lddk, long lddv) {
    a.drop = a.drop.resolved();
    HIP_DYNAMIC_SHARED(float, smem)
    const BigLds l = big_carve(smem, a.T, a.S, true);
    const int tid = threadIdx.x, S = a.S;
    const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    load_tile<T>(l.q, D, (const T*)a.q + (long)b * a.T * a.ldq + h * D, a.ldq, a.T, tid);
    load_tile<T>(l.o, D, dout + (long)b * a.T * a.ldo + h * D, a.ldo, a.T, tid);
    load_tile<T>(l.k, KP, (const T*)a.k + (long)b * S * a.ldk + h * D, a.ldk, S, tid);
    load_tile<T>(l.v, KP, (const T*)a.v + (long)b * S * a.ldv + h * D, a.ldv, S, tid);
    __syncthreads();
    const int len = a.lengths ? (int)a.lengths[b] : S;
    big_scores_softmax(l, a, len, tid);
    const uint64_t pbase = (uint64_t)blockIdx.x * (uint64_t)(a.T * S);
    // dV[j][d] = sum_i Pd[i][j] dO[i][d]
    matmul_store<T>(dv + (long)b * S * lddv + h * D, lddv, S, tid, a.T, l.o, D,
                    [&](int j, int i) { return a.drop.apply(l.p[i * S + j], pbase + i * S + j); });
    // dP[i][j] = dropout'( sum_d dO[i][d] V[j][d] )
    const int S4 = (S + 3) / 4;
    for (int idx = tid; idx < a.T * S4; idx += 256) {
        const int i = idx / S4, j0 = (idx % S4) * 4;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float* go = l.o + i * D;
        const float* v0 = l.v + j0 * KP;
#pragma unroll 8
        for (int d = 0; d < D; ++d) {
            const float gv = go[d];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] += gv * v0[t * KP + d];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (j0 + t < S) l.g[i * S + j0 + t] = a.drop.apply(acc[t], pbase + i * S + j0 + t);
    }
    __syncthreads();
    // dS = P * (dP - rowsum(dP * P)) * scale   (written over g)
    const int lane = tid & 63, wv = tid >> 6;
    for (int i = wv; i < a.T; i += 4) {
        float dot = 0.f;
        for (int j = lane; j < S; j += 64) dot += l.p[i * S + j] * l.g[i * S + j];
        dot = wave_sum(dot);
        for (int j = lane; j < S; j += 64) l.g[i * S + j] = l.p[i * S + j] * (l.g[i * S + j] - dot) * a.scale;
    }
    __syncthreads();
    matmul_store<T>(dq + (long)b * a.T * lddq + h * D, lddq, a.T, tid, S, l.k, KP,
                    [&](int i, int j) { return l.g[i * S + j]; });
    matmul_store<T>(dk + (long)b * S * lddk + h * D, lddk, S, tid, a.T, l.q, D,
                    [&](int j, int i) { return l.g[i * S + j]; });
}

// ---------------------------------------------------------------------------------------------------------
// bf16: the same attention on the matrix cores, ONE WAVE per (batch, head), 4 waves per workgroup.
//
// Notation: mfma(X, Y) = v_mfma_f32_16x16x32_bf16 with X as the first and Y as the second operand gives every
// lane out[y = lane&15][x = 4*(lane>>4) + r], r = 0..3, = sum_k X[x][k] * Y[y][k]; lane (., g = lane>>4) supplies
// eight k of row lane&15 of each operand -- WHICH eight is free as long as both operands agree.
//
//   scores[i][j] = mfma(K rows j, Q rows i), k = d: both fragments are 16-byte global loads (no LDS);
//                  lane (i, g) ends up with scores[i][16*jt + 4g + r] for the four key tiles jt: the four lanes
//                  sharing a query row hold all 64 keys -> masks, softmax and dropout stay in registers
//                  (row max / sum = 16 local values + two xor-shuffles).
//   O[i][d]      = mfma(V^T rows d, P rows i), k = j with the k-set of lane (., g) in step ks chosen as
//                  {32ks + 4g + r} U {32ks + 16 + 4g + r}: exactly the probabilities the lane already owns (tiles
//                  2ks and 2ks+1), so P never moves.  V^T fragments with that k-set are two ds_read_b64_tr_b16
//                  from the row-major V tile in LDS (the only LDS tile).
// The backward kernel adds dP = mfma(V rows j, dO rows i) in the scores layout, the softmax backward in
// registers, dQ = mfma(K^T rows d, dS rows i) like O, and dV / dK (contraction over the queries) from the
// probabilities / dS written once to LDS in bf16 and read back transposed with the same tr instruction.
constexpr int SP = 64;                     // keys padded to four 16-wide tiles
constexpr int VROW = D + 8;                // LDS row pitch (elements): 144 B, keeps the tr reads off one bank set

typedef short v4s_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_bf16(bf16_t* p, f32x4_t v) {
    *reinterpret_cast<uint2*>(p) = make_uint2(f2bf2(v[0], v[1]),
                                              f2bf2(v[2], v[3]));
}
__device__ __forceinline__ bf16x8_t ld_frag(const bf16_t* base, long ld, int row, int nrows, int col) {
    // 8 consecutive bf16 of row `row` (zeros past the end of the tile).  The load is unconditional (row 0 stands in for rows
    // past the end) and masked afterwards: as `if (row < nrows) load` every fragment waited for its own round trip
    const bool in = row < nrows;
    const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(base + (long)(in ? row : 0) * ld + col);
    return in ? v : bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
}
__device__ __forceinline__ bf16x8_t pack8(const float* lo, const float* hi) {
    bf16x8_t r;
#pragma unroll
    for (int t = 0; t < 4; ++t) { r[t] = (short)f2bf(lo[t]); r[4 + t] = (short)f2bf(hi[t]); }
    return r;
}
// rows x 64 bf16 tile -> LDS (pitch VROW), rows >= nrows zero-filled up to PAD rows; one wave.  Two phases so that a kernel
// can have the loads of all its tiles in flight before the first LDS store waits for any of them.
template <int PAD> struct TileStage {
    static constexpr int TRIPS = PAD * 8 / 64;
    uint4 v[TRIPS];
    int rows;
    __device__ __forceinline__ void request(const bf16_t* src, long ld, int nrows, int lane) {
        rows = nrows;
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            const int c = lane + 64 * t, r = c >> 3, col = (c & 7) * 8;
            v[t] = *reinterpret_cast<const uint4*>(src + (long)(r < nrows ? r : 0) * ld + col);
        }
    }
    __device__ __forceinline__ void put(bf16_t* dst, int lane) const {       // (the zero fill happens here: a select next to the
#pragma unroll                                                               //  load would wait for it)
        for (int t = 0; t < TRIPS; ++t) {
            const int c = lane + 64 * t, r = c >> 3, col = (c & 7) * 8;
            *reinterpret_cast<uint4*>(dst + r * VROW + col) = r < rows ? v[t] : make_uint4(0u, 0u, 0u, 0u);
        }
    }
};
// fragment of the TRANSPOSED tile: rows = columns c0..c0+15 of the LDS tile, k-set {kb + jj} U {kb + 16 + jj}
__device__ __forceinline__ bf16x8_t tr_frag(const bf16_t* tile, int c0, int kb, int lane) {
    const int w = lane & 15;
    const bf16_t* pa = tile + (kb + (w >> 2)) * VROW + c0 + 4 * (w & 3);
    const v4s_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)pa);
    const v4s_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(pa + 16 * VROW));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
// same, standard k-set {kb + 0..7}: two reads 4 rows apart
__device__ __forceinline__ bf16x8_t tr_frag_std(const bf16_t* tile, int c0, int kb, int lane) {
    const int w = lane & 15;
    const bf16_t* pa = tile + (kb + (w >> 2)) * VROW + c0 + 4 * (w & 3);
    const v4s_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)pa);
    const v4s_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(pa + 4 * VROW));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ float quad_max(float v) {   // over the four lanes sharing lane & 15
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// scores -> probabilities (pre-dropout) for the two query tiles; p[it][jt][r] belongs to (i, j) =
// (16 it + lane&15, 16 jt + 4 (lane>>4) + r)
__device__ __forceinline__ void mfma_probabilities(const AttnArgs& a, const bf16_t* q, const bf16_t* k, int len, int lane,
                                                   float (&p)[2][4][4]) {
    const int w = lane & 15, g = lane >> 4;
    f32x4_t sc[2][4];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) sc[it][jt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t fq[2], fk[4];
#pragma unroll
        for (int it = 0; it < 2; ++it) fq[it] = ld_frag(q, a.ldq, 16 * it + w, a.T, 32 * ks + 8 * g);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) fk[jt] = ld_frag(k, a.ldk, 16 * jt + w, a.S, 32 * ks + 8 * g);
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
                sc[it][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[jt], fq[it], sc[it][jt], 0, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = 16 * it + w;
        float m = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 16 * jt + 4 * g + r;
                const bool masked = j >= a.S || j >= len || (a.causal && j > i);
                p[it][jt][r] = masked ? -INFINITY : sc[it][jt][r] * a.scale;
                m = fmaxf(m, p[it][jt][r]);
            }
        m = quad_max(m);
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = (m > -INFINITY && p[it][jt][r] > -INFINITY) ? __expf(p[it][jt][r] - m) : 0.f;
                p[it][jt][r] = e;
                sum += e;
            }
        sum = quad_sum(sum);
        const float inv = sum > 0.f ? 1.f / sum : 0.f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) p[it][jt][r] *= inv;
    }
}

// out[i][16 dt + 4g + r] (+)= sum_j coef[i][j] * tile[j][d]: the "O = P V" product; coef in the scores layout
__device__ __forceinline__ void mfma_rows_times_tile(const float (&c)[2][4][4], const bf16_t* tile, int lane,
                                                     f32x4_t (&out)[2][4]) {
    const int g = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t fc[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) fc[it] = pack8(c[it][2 * ks], c[it][2 * ks + 1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x8_t ft = tr_frag(tile, 16 * dt, 32 * ks + 4 * g, lane);
#pragma unroll
            for (int it = 0; it < 2; ++it)
                out[it][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ft, fc[it], out[it][dt], 0, 0, 0);
        }
    }
}
__device__ __forceinline__ void store_rows(bf16_t* dst, long ld, int nrows, int lane, const f32x4_t (&v)[2][4]) {
    const int w = lane & 15, g = lane >> 4;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = 16 * it + w;
        if (i >= nrows) continue;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) st4_bf16(dst + (long)i * ld + 16 * dt + 4 * g, v[it][dt]);
    }
}

__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(AttnArgs a, bf16_t* __restrict__ o, int nbh) {
    a.drop = a.drop.resolved();
    __shared__ __attribute__((aligned(16))) bf16_t sV[4][SP * VROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.x * 4 + wave;
    if (bh >= nbh) return;
    const int b = bh / a.heads, h = bh % a.heads;
    const bf16_t* q = (const bf16_t*)a.q + (long)b * a.T * a.ldq + h * D;
    const bf16_t* k = (const bf16_t*)a.k + (long)b * a.S * a.ldk + h * D;
    const bf16_t* v = (const bf16_t*)a.v + (long)b * a.S * a.ldv + h * D;
    const int len = a.lengths ? (int)a.lengths[b] : a.S;      // requested with the tile, not behind it
    {
        TileStage<SP> tv;
        tv.request(v, a.ldv, a.S, lane);
        vtx_loads_issued();
        tv.put(sV[wave], lane);
    }
    float p[2][4][4];
    mfma_probabilities(a, q, k, len, lane, p);
    if (a.drop.thresh) {
        const uint64_t pbase = (uint64_t)bh * (TMAX * SMAX);
        const int w = lane & 15, g = lane >> 4;
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    p[it][jt][r] = a.drop.apply(p[it][jt][r], pbase + (16 * it + w) * SMAX + 16 * jt + 4 * g + r);
    }
    __builtin_amdgcn_wave_barrier();          // sV written by this wave's own lanes
    f32x4_t acc[2][4];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[it][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    mfma_rows_times_tile(p, sV[wave], lane, acc);
    store_rows(o + (long)b * a.T * a.ldo + h * D, a.ldo, a.T, lane, acc);
}

// scores-layout coefficients c[it][jt][r] -> LDS tile [i][j] (bf16), so that products contracting over the
// queries can read them transposed
__device__ __forceinline__ void spill_coef(bf16_t* tile, const float (&c)[2][4][4], int lane) {
    const int w = lane & 15, g = lane >> 4;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
            { const f32x4_t t4 = {c[it][jt][0], c[it][jt][1], c[it][jt][2], c[it][jt][3]}; st4_bf16(tile + (16 * it + w) * VROW + 16 * jt + 4 * g, t4); }
}
// out[j][16 dt + 4g + r] = sum_i coef[i][j] * tile[i][d]   (coef and tile both in LDS, 32 rows = one MFMA K step)
__device__ __forceinline__ void mfma_coefT_times_tile(const bf16_t* coef, const bf16_t* tile, bf16_t* dst, long ld,
                                                      int nrows, int lane) {
    const int w = lane & 15, g = lane >> 4;
    bf16x8_t ft[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) ft[dt] = tr_frag_std(tile, 16 * dt, 8 * g, lane);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
        const bf16x8_t fc = tr_frag_std(coef, 16 * jt, 8 * g, lane);
        const int j = 16 * jt + w;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4_t r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ft[dt], fc, zero, 0, 0, 0);
            if (j < nrows) st4_bf16(dst + (long)j * ld + 16 * dt + 4 * g, r);
        }
    }
}

__global__ __launch_bounds__(128) void attn_bwd_mfma_kernel(AttnArgs a, const bf16_t* __restrict__ dout,
                                                            bf16_t* __restrict__ dq, bf16_t* __restrict__ dk,
                                                            bf16_t* __restrict__ dv, long lddq, long lddk, long lddv,
                                                            int nbh) {
    a.drop = a.drop.resolved();
    __shared__ __attribute__((aligned(16))) bf16_t sK[2][SP * VROW], sQ[2][TMAX * VROW], sO[2][TMAX * VROW], sC[2][TMAX * VROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.x * 2 + wave;
    if (bh >= nbh) return;
    const int b = bh / a.heads, h = bh % a.heads;
    const int w = lane & 15, g = lane >> 4;
    const bf16_t* q = (const bf16_t*)a.q + (long)b * a.T * a.ldq + h * D;
    const bf16_t* k = (const bf16_t*)a.k + (long)b * a.S * a.ldk + h * D;
    const bf16_t* v = (const bf16_t*)a.v + (long)b * a.S * a.ldv + h * D;
    const bf16_t* go = dout + (long)b * a.T * a.ldo + h * D;
    const int len = a.lengths ? (int)a.lengths[b] : a.S;      // requested with the tiles, not behind them
    {
        TileStage<SP> tk; TileStage<TMAX> tq, to;
        tk.request(k, a.ldk, a.S, lane); tq.request(q, a.ldq, a.T, lane); to.request(go, a.ldo, a.T, lane);
        vtx_loads_issued();
        tk.put(sK[wave], lane); tq.put(sQ[wave], lane); to.put(sO[wave], lane);
    }
    const uint64_t pbase = (uint64_t)bh * (TMAX * SMAX);
    float p[2][4][4];
    mfma_probabilities(a, q, k, len, lane, p);
    // dV[j][d] = sum_i dropout(P)[i][j] dO[i][d]
    {
        float pd[2][4][4];
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pd[it][jt][r] = a.drop.apply(p[it][jt][r], pbase + (16 * it + w) * SMAX + 16 * jt + 4 * g + r);
        spill_coef(sC[wave], pd, lane);
    }
    __builtin_amdgcn_wave_barrier();
    mfma_coefT_times_tile(sC[wave], sO[wave], dv + (long)b * a.S * lddv + h * D, lddv, a.S, lane);
    // dP[i][j] = dropout'( sum_d dO[i][d] V[j][d] ), in the scores layout
    f32x4_t dp[2][4];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) dp[it][jt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t fo[2], fv[4];
#pragma unroll
        for (int it = 0; it < 2; ++it) fo[it] = ld_frag(go, a.ldo, 16 * it + w, a.T, 32 * ks + 8 * g);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) fv[jt] = ld_frag(v, a.ldv, 16 * jt + w, a.S, 32 * ks + 8 * g);
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
                dp[it][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv[jt], fo[it], dp[it][jt], 0, 0, 0);
    }
    // dS = P * (dP - rowsum(dP * P)) * scale
    float ds[2][4][4];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        float dot = 0.f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float gji = a.drop.apply(dp[it][jt][r], pbase + (16 * it + w) * SMAX + 16 * jt + 4 * g + r);
                ds[it][jt][r] = gji;
                dot += gji * p[it][jt][r];
            }
        dot = quad_sum(dot);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ds[it][jt][r] = p[it][jt][r] * (ds[it][jt][r] - dot) * a.scale;
    }
    // dQ[i][d] = sum_j dS[i][j] K[j][d]
    {
        f32x4_t acc[2][4];
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) acc[it][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        mfma_rows_times_tile(ds, sK[wave], lane, acc);
        store_rows(dq + (long)b * a.T * lddq + h * D, lddq, a.T, lane, acc);
    }
    // dK[j][d] = sum_i dS[i][j] Q[i][d]
    __builtin_amdgcn_wave_barrier();          // every lane is done reading the dropout(P) tile
    spill_coef(sC[wave], ds, lane);
    __builtin_amdgcn_wave_barrier();
    mfma_coefT_times_tile(sC[wave], sQ[wave], dk + (long)b * a.S * lddk + h * D, lddk, a.S, lane);
}

static int check(const char* who, int dtype, int B, int heads, int T, int S, int head_dim) {
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "%s: bad dtype", who);
    VTX_CHECK(B >= 0 && heads > 0 && T > 0 && S > 0, VTX_ERR_ARG, "%s: bad shape", who);
    VTX_CHECK(head_dim == D, VTX_ERR_SHAPE, "%s: supports head_dim == 64 (got %d)", who, head_dim);
    return VTX_OK;
}
// the tuned kernels take T <= 32, S <= 56; anything else runs on the general kernels while its tiles fit the 160 KiB of LDS
// (the limit is the DEVICE's: hipDeviceAttributeMaxSharedMemoryPerBlock of the current device, 160 KiB on gfx950 -- read once;
// a part with a smaller LDS refuses the shape here with the sizes in the message instead of failing the launch)
static size_t big_lds_max() {
    static std::atomic<long> cached{-1};
    long v = cached.load(std::memory_order_relaxed);
    if (v < 0) {
        int dev = 0, lim = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&lim, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || lim <= 0) {
            (void)hipGetLastError();
            lim = 64 * 1024;                  // what every gfx9 part has
        }
        v = lim;
        cached.store(v, std::memory_order_relaxed);
    }
    return (size_t)v;
}
static bool tuned_shape(int T, int S) { return T <= TMAX && S <= SMAX; }
// `reserved`: the largest dynamic-LDS size already granted to this kernel instantiation (one static per call site): the
// attribute is set only when a launch needs more than any launch before it, not on every call
template <class K> static int big_prepare(const char* who, K kern, int T, int S, bool bwd, size_t* lds, std::atomic<long>& reserved) {
    *lds = big_lds_bytes(T, S, bwd);
    const size_t lim = big_lds_max();
    VTX_CHECK(*lds <= lim, VTX_ERR_SHAPE, "%s: T=%d queries x S=%d keys need %zu bytes of LDS (limit %zu): beyond the "
              "tuned envelope (T <= %d, S <= %d) the tiles of one (batch, head) must fit one workgroup", who, T, S, *lds, lim, TMAX, SMAX);
    if ((long)*lds > reserved.load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)*lds) != hipSuccess) {
            (void)hipGetLastError();
            vtx_set_error("%s: cannot reserve %zu bytes of LDS", who, *lds);
            return VTX_ERR_LAUNCH;
        }
        long prev = reserved.load(std::memory_order_relaxed);
        while (prev < (long)*lds && !reserved.compare_exchange_weak(prev, (long)*lds)) {}
    }
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
    if (dtype == VTX_BF16)
        VTX_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, VTX_ERR_SHAPE, "attention_fwd: row strides must be multiples of 8");
    // algorithmic work of one call (the roofline accounting of bench.py): QK^T and PV = 2 x 2 T S 64 FLOP per (batch, head);
    // q, o (T rows) and k, v (S rows) of 64 elements each
    const double fl_fwd = 4.0 * B * heads * T * S * 64, by_fwd = (double)B * heads * 64 * (2.0 * T + 2.0 * S);
    if (!tuned_shape(T, S)) {
        size_t lds = 0;
        if (dtype == VTX_BF16) {
            VTX_CHECK(ldo % 8 == 0, VTX_ERR_SHAPE, "attention_fwd: row strides must be multiples of 8");
            static std::atomic<long> reserved{0};
            if ((rc = big_prepare("attention_fwd", attn_fwd_big_kernel<bf16_t>, T, S, false, &lds, reserved))) return rc;
            VTX_KLAUNCH("attention_fwd", fl_fwd, 2.0 * by_fwd, (attn_fwd_big_kernel<bf16_t>), grid, block, lds, (hipStream_t)stream, a, (bf16_t*)o);
        } else {
            static std::atomic<long> reserved{0};
            if ((rc = big_prepare("attention_fwd", attn_fwd_big_kernel<float>, T, S, false, &lds, reserved))) return rc;
            VTX_KLAUNCH("attention_fwd", fl_fwd, 4.0 * by_fwd, (attn_fwd_big_kernel<float>), grid, block, lds, (hipStream_t)stream, a, (float*)o);
        }
        VTX_LAUNCH_CHECK();
        return VTX_OK;
    }
    if (dtype == VTX_BF16) {
        VTX_KLAUNCH("attention_fwd", fl_fwd, 2.0 * by_fwd, attn_fwd_mfma_kernel, dim3(vtx_cdiv(B * heads, 4)), block, 0, (hipStream_t)stream, a, (bf16_t*)o, B * heads);
    }
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
        VTX_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0,
                  VTX_ERR_SHAPE, "attention_bwd: row strides must be multiples of 8 (inputs) / 4 (gradients)");
    // scores recomputed + dP, dQ, dK, dV = 5 contractions of 2 T S 64 FLOP; q, dout, dq (T rows), k, v, dk, dv (S rows)
    const double fl_bwd = 10.0 * B * heads * T * S * 64, by_bwd = (double)B * heads * 64 * (3.0 * T + 4.0 * S);
    if (!tuned_shape(T, S)) {
        size_t lds = 0;
        if (dtype == VTX_BF16) {
            VTX_CHECK(lddq % 8 == 0 && lddk % 8 == 0 && lddv % 8 == 0, VTX_ERR_SHAPE, "attention_bwd: gradient row strides must be multiples of 8");
            static std::atomic<long> reserved{0};
            if ((rc = big_prepare("attention_bwd", attn_bwd_big_kernel<bf16_t>, T, S, true, &lds, reserved))) return rc;
            VTX_KLAUNCH("attention_bwd", fl_bwd, 2.0 * by_bwd, (attn_bwd_big_kernel<bf16_t>), grid, block, lds, (hipStream_t)stream, a, (const bf16_t*)dout,
                               (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, lddq, lddk, lddv);
        } else {
            static std::atomic<long> reserved{0};
            if ((rc = big_prepare("attention_bwd", attn_bwd_big_kernel<float>, T, S, true, &lds, reserved))) return rc;
            VTX_KLAUNCH("attention_bwd", fl_bwd, 4.0 * by_bwd, (attn_bwd_big_kernel<float>), grid, block, lds, (hipStream_t)stream, a, (const float*)dout,
                               (float*)dq, (float*)dk, (float*)dv, lddq, lddk, lddv);
        }
        VTX_LAUNCH_CHECK();
        return VTX_OK;
    }
    if (dtype == VTX_BF16) {
        VTX_KLAUNCH("attention_bwd", fl_bwd, 2.0 * by_bwd, attn_bwd_mfma_kernel, dim3(vtx_cdiv(B * heads, 2)), dim3(128), 0, (hipStream_t)stream, a,
                           (const bf16_t*)dout, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, lddq, lddk, lddv, B * heads);
    }
    else
        hipLaunchKernelGGL((attn_bwd_kernel<float>), grid, block, 0, (hipStream_t)stream, a, (const float*)dout,
                           (float*)dq, (float*)dk, (float*)dv, lddq, lddk, lddv);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}
