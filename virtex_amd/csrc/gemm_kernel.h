// One MFMA contraction kernel for every GEMM-shaped op of the hot path (gfx950, wave64).
//
//   C[m][n] (+)= epilogue( sum_k A(m,k) * B(n,k) )
//
// Block = 256 threads = 4 waves in a 2x2 grid; block tile BM x BN (64|128), wave tile
// (BM/2)x(BN/2) as (MT x NT) 16x16 MFMA tiles, K advanced in steps of BK = 64 bytes per row.
// Operand tiles are staged global -> VGPR (16-byte loads, issued one K-step ahead) -> LDS
// ([row][BK+pad], double buffered, one barrier per K-step) -> MFMA fragments.
//   bf16: v_mfma_f32_16x16x32_bf16 (fragment = ds_read_b128 of 8 consecutive k)
//   fp32: v_mfma_f32_16x16x4_f32   (exact fp32; parity mode)
// The MFMA is issued as D = Btile x Atile so that each lane ends up with 4 CONSECUTIVE n of
// one m (lane l: m = l&15, n = 4*(l>>4)..+3) -> vector epilogue loads/stores.
//
// Operands are described by "loaders" (how a 16-byte chunk of the tile maps to global
// memory):
//   KC loaders: chunk = VEC consecutive k of one row   (row-major [rows][K] views, im2col-free
//               NHWC conv gathers for forward and input-gradient)
//   MC loaders: chunk = VEC consecutive rows at one k  (k-major [K][rows] views: weight
//               gradients; transposed into the LDS tile on store)
#pragma once
#include "vtx_common.h"

namespace vtxg {

constexpr int NTHREADS = 256;

template <class T> struct Mma;
template <> struct Mma<bf16_t> {
    static constexpr int KI = 32;
    typedef bf16x8_t Frag;
    __device__ static __forceinline__ Frag load(const bf16_t* frag, int kk) {
        return *reinterpret_cast<const bf16x8_t*>(frag + kk * 32);
    }
    __device__ static __forceinline__ f32x4_t mma(Frag a, Frag b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static constexpr int KI = 4;
    typedef float Frag;
    __device__ static __forceinline__ Frag load(const float* frag, int kk) { return frag[kk * 4]; }
    __device__ static __forceinline__ f32x4_t mma(Frag a, Frag b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
};

__device__ __forceinline__ uint4 ld16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ uint4 zero16() { return make_uint4(0u, 0u, 0u, 0u); }

// q = n / d for 0 <= n < 2^24 (exact after one correction step); inv = 1.0f / d
__device__ __forceinline__ int fdiv(int n, int d, float inv) {
    int q = (int)((float)n * inv);
    int r = n - q * d;
    if (r < 0) --q; else if (r >= d) ++q;
    return q;
}

// ------------------------------------------------------------------ plain matrix loaders
// rows x K view, k contiguous: element (r,k) at p[r*ld + k]
template <class T, int SL> struct PlainKC {
    static constexpr bool MC = false;
    const T* p; long ld; int rows; int K;
    struct State { const T* rp[SL]; int kc[SL]; };
    // slot i of this thread stages the chunk (global row r, k = k0 + kc)
    __device__ __forceinline__ void init_slot(State& s, int i, int r, int kc) const {
        s.kc[i] = kc;
        s.rp[i] = r < rows ? p + (long)r * ld + kc : nullptr;
    }
    __device__ __forceinline__ const T* ptr(const State& s, int i, int k0) const {
        return (s.rp[i] && k0 + s.kc[i] < K) ? s.rp[i] + k0 : nullptr;
    }
};
// K x rows view, rows contiguous: element (r,k) at p[k*ld + r]
template <class T, int SL> struct PlainMC {
    static constexpr bool MC = true;
    const T* p; long ld; int rows; int K;
    struct State { int r0[SL < 2 ? 2 : SL]; };
    // slot i stages VEC consecutive rows starting at global row r0 (r0 < 0: nothing), at global k
    __device__ __forceinline__ void init_slot(State& s, int i, int r0) const { s.r0[i] = r0 < rows ? r0 : -1; }
    __device__ __forceinline__ const T* ptr(const State& s, int i, int k) const {
        return (s.r0[i] >= 0 && k < K) ? p + (long)k * ld + s.r0[i] : nullptr;
    }
};

// ------------------------------------------------------------------ NHWC conv geometry
struct ConvGeo {
    int N, H, W, C, logC;      // input  NHWC (C power of two, >= VEC)
    int KO, logKO;             // output channels (power of two for dgrad)
    int R, S, rcpS;            // filter; rcpS = ceil(65536 / S)
    int stride, logStride, pad;
    int OH, OW;
    float inv_ow, inv_ohow;
};
static inline int vtx_ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// forward: A(m,k): m = (n,oh,ow), k = (kh,kw,ci) -> x[n][oh*s-p+kh][ow*s-p+kw][ci]
template <class T, int SL> struct ConvFwdA {
    static constexpr bool MC = false;
    const T* x; ConvGeo g; int rows; int K;
    struct State { long base[SL]; int ih0[SL], iw0[SL]; bool ok[SL]; int kc[SL]; };
    __device__ __forceinline__ void init_slot(State& s, int i, int m, int kc) const {
        s.kc[i] = kc;
        s.ok[i] = m < rows;
        const int mm = s.ok[i] ? m : 0;
        const int n = mm / (g.OH * g.OW), rem = mm - n * g.OH * g.OW;
        const int oh = rem / g.OW, ow = rem - oh * g.OW;
        s.base[i] = (long)n * g.H * g.W * g.C;
        s.ih0[i] = oh * g.stride - g.pad;
        s.iw0[i] = ow * g.stride - g.pad;
    }
    __device__ __forceinline__ const T* ptr(const State& s, int i, int k0) const {
        const int k = k0 + s.kc[i];
        const int tap = k >> g.logC, ci = k & (g.C - 1);
        const int kh = (tap * g.rcpS) >> 16, kw = tap - kh * g.S;
        const int ih = s.ih0[i] + kh, iw = s.iw0[i] + kw;
        const bool ok = s.ok[i] && k < K && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
        return ok ? x + s.base[i] + ((long)ih * g.W + iw) * g.C + ci : nullptr;
    }
};

// input gradient: A(m,k): m = (n,ih,iw), k = (kh,kw,co) -> dy[n][(ih+p-kh)/s][(iw+p-kw)/s][co]
template <class T, int SL> struct ConvDgradA {
    static constexpr bool MC = false;
    const T* dy; ConvGeo g; int rows; int K;
    struct State { long base[SL]; int ihp[SL], iwp[SL]; bool ok[SL]; int kc[SL]; };
    __device__ __forceinline__ void init_slot(State& s, int i, int m, int kc) const {
        s.kc[i] = kc;
        s.ok[i] = m < rows;
        const int mm = s.ok[i] ? m : 0;
        const int n = mm / (g.H * g.W), rem = mm - n * g.H * g.W;
        const int ih = rem / g.W, iw = rem - ih * g.W;
        s.base[i] = (long)n * g.OH * g.OW * g.KO;
        s.ihp[i] = ih + g.pad;
        s.iwp[i] = iw + g.pad;
    }
    __device__ __forceinline__ const T* ptr(const State& s, int i, int k0) const {
        const int k = k0 + s.kc[i];
        const int tap = k >> g.logKO, co = k & (g.KO - 1);
        const int kh = (tap * g.rcpS) >> 16, kw = tap - kh * g.S;
        const int th = s.ihp[i] - kh, tw = s.iwp[i] - kw;
        const int sm = g.stride - 1;
        const int oh = th >> g.logStride, ow = tw >> g.logStride;
        const bool ok = s.ok[i] && k < K && th >= 0 && tw >= 0 && ((th | tw) & sm) == 0 &&
                        oh < g.OH && ow < g.OW;
        return ok ? dy + s.base[i] + ((long)oh * g.OW + ow) * g.KO + co : nullptr;
    }
};

// weight gradient: B(r,k): r = (kh,kw,ci), k = (n,oh,ow) -> x[n][oh*s-p+kh][ow*s-p+kw][ci]
template <class T, int SL> struct ConvWgradB {
    static constexpr bool MC = true;
    const T* x; ConvGeo g; int rows; int K;  // rows = R*S*C, K = N*OH*OW
    struct State { int kh[SL < 2 ? 2 : SL], kw[SL < 2 ? 2 : SL], ci[SL < 2 ? 2 : SL]; };
    __device__ __forceinline__ void init_slot(State& s, int i, int r0) const {
        const int tap = r0 >> g.logC;
        s.ci[i] = r0 < rows ? (r0 & (g.C - 1)) : -1;
        s.kh[i] = (tap * g.rcpS) >> 16;
        s.kw[i] = tap - s.kh[i] * g.S;
    }
    __device__ __forceinline__ const T* ptr(const State& s, int i, int pix) const {
        const int ohow = g.OH * g.OW;
        const int n = fdiv(pix, ohow, g.inv_ohow), rem = pix - n * ohow;
        const int oh = fdiv(rem, g.OW, g.inv_ow), ow = rem - oh * g.OW;
        const int ih = oh * g.stride - g.pad + s.kh[i], iw = ow * g.stride - g.pad + s.kw[i];
        const bool ok = s.ci[i] >= 0 && pix < K && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
        return ok ? x + (((long)n * g.H + ih) * g.W + iw) * g.C + s.ci[i] : nullptr;
    }
};

// ------------------------------------------------------------------ epilogues
enum { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2 };

template <class T> __device__ __forceinline__ void st4(T* p, const float* v);
template <> __device__ __forceinline__ void st4<float>(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, const float* v) {
    *reinterpret_cast<uint2*>(p) = make_uint2((uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16),
                                              (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16));
}
template <class T> __device__ __forceinline__ void ld4(const T* p, float* v);
template <> __device__ __forceinline__ void ld4<float>(const float* p, float* v) {
    float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void ld4<bf16_t>(const bf16_t* p, float* v) {
    uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}

// out = dropout(act(acc*alpha + bias)) + residual ; optional copy of the pre-activation
template <class T> struct EpiStore {
    T* out; long ldc; const float* bias; const T* residual; long ldr; T* preact; int act;
    float alpha; Dropout drop; int M, N;
    __device__ __forceinline__ void operator()(int m, int n, f32x4_t acc) const {
        if (m >= M || n >= N) return;
        float v[4] = {acc[0] * alpha, acc[1] * alpha, acc[2] * alpha, acc[3] * alpha};
        if (bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += bias[n + j];
        }
        const long o = (long)m * ldc + n;
        if (preact) st4<T>(preact + o, v);
        if (act == ACT_GELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
        } else if (act == ACT_RELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (drop.thresh) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = drop.apply(v[j], (uint64_t)(o + j));
        }
        if (residual) {
            float r[4]; ld4<T>(residual + (long)m * ldr + n, r);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += r[j];
        }
        st4<T>(out + o, v);
    }
};
// out(fp32) += alpha * acc     (split-K partial sums and "+=" gradient accumulation)
struct EpiAtomic {
    float* out; long ldc; float alpha; int M, N;
    __device__ __forceinline__ void operator()(int m, int n, f32x4_t acc) const {
        if (m >= M || n >= N) return;
        float* o = out + (long)m * ldc + n;
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(o + j, alpha * acc[j]);
    }
};

// ------------------------------------------------------------------ the kernel
// LDS image of one operand tile.
//  * padded   (KC operands, and every fp32 operand): elem(row,k) at row*(BK+VEC) + k
//  * swizzled (bf16 MC operands): 64-byte rows, the four 16-byte k-slots of a row XOR-permuted
//    by SWZ(row>>2) so that (a) the transposing 4-byte stores of a half-wave hit 32 distinct banks
//    and (b) every 16-lane group of the ds_read_b128 fragment loads hits 64 distinct banks.
__device__ __forceinline__ int swz_slot(int slot, int row) { return slot ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3); }

template <class T, bool MC> struct TileImage {
    static constexpr int VEC = Elem<T>::VEC, BK = 4 * VEC;
    static constexpr bool SWZ = MC && (sizeof(T) == 2);
    static constexpr int RS = SWZ ? BK : BK + VEC;   // row stride (elements)
    // element offset of the fragment (8 consecutive k for bf16 / 1 element for fp32) of `row`
    __device__ static __forceinline__ int frag(int row, int lane) {
        if constexpr (SWZ) return row * RS + swz_slot(lane >> 4, row) * 8;
        else if constexpr (sizeof(T) == 2) return row * RS + (lane >> 4) * 8;
        else return row * RS + (lane >> 4);
    }
};

// Stages one operand tile (ROWS x BK) : global -> registers (load) -> LDS (store).
template <class T, int ROWS, class L> struct Stager {
    static constexpr int VEC = Elem<T>::VEC, BK = 4 * VEC, SL = ROWS / 64;
    static constexpr bool MC = L::MC, PACK = MC && (sizeof(T) == 2);
    static constexpr int NR = PACK ? 2 : SL;   // 16-byte registers held per thread
    static constexpr int CPR = ROWS / VEC;     // 16-byte chunks per k (MC)
    typedef TileImage<T, MC> Img;
    uint4 r[NR];
    typename L::State st;
    int rc, kq;                                // MC: row chunk / k index (pair index when PACK)
    bool active;

    __device__ __forceinline__ void init(const L& l, int row0, int tid) {
        if constexpr (!MC) {                   // chunk (row = tid/4 + 64 i, k chunk = tid%4)
            active = true; rc = kq = 0;
#pragma unroll
            for (int i = 0; i < SL; ++i) l.init_slot(st, i, row0 + (tid >> 2) + 64 * i, (tid & 3) * VEC);
        } else if constexpr (PACK) {           // thread = (k pair kq in 0..15, row chunk rc), kq fastest
            kq = tid & 15; rc = tid >> 4; active = rc < CPR;
            l.init_slot(st, 0, active ? row0 + rc * VEC : (1 << 30));
        } else {                               // fp32: chunk c = tid + 256*i -> (k = c / CPR, rc = c % CPR)
            rc = tid % CPR; kq = tid / CPR; active = true;
            l.init_slot(st, 0, row0 + rc * VEC);
        }
    }
    __device__ static __forceinline__ uint4 fetch(const T* p) { return p ? ld16(p) : zero16(); }
    __device__ __forceinline__ void load(const L& l, int k0) {
        if constexpr (!MC) {
#pragma unroll
            for (int i = 0; i < SL; ++i) r[i] = fetch(l.ptr(st, i, k0));
        } else if constexpr (PACK) {
            r[0] = fetch(l.ptr(st, 0, k0 + 2 * kq));
            r[1] = fetch(l.ptr(st, 0, k0 + 2 * kq + 1));
        } else {
#pragma unroll
            for (int i = 0; i < SL; ++i) r[i] = fetch(l.ptr(st, 0, k0 + kq + (NTHREADS / CPR) * i));
        }
    }
    __device__ __forceinline__ void store(T* tile, int tid) const {
        if constexpr (!MC) {
#pragma unroll
            for (int i = 0; i < SL; ++i)
                *reinterpret_cast<uint4*>(tile + ((tid >> 2) + 64 * i) * Img::RS + (tid & 3) * VEC) = r[i];
        } else if constexpr (PACK) {
            if (!active) return;
            // r[0] = rows rc*8..+7 at k = 2*kq, r[1] = same rows at k+1 -> 8 dwords {k, k+1} per row
            const uint32_t a[4] = {r[0].x, r[0].y, r[0].z, r[0].w}, b[4] = {r[1].x, r[1].y, r[1].z, r[1].w};
            uint32_t d[8];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                d[2 * m] = (a[m] & 0xffffu) | (b[m] << 16);
                d[2 * m + 1] = (a[m] >> 16) | (b[m] & 0xffff0000u);
            }
            const bool odd = rc & 1;           // odd row chunks walk their rows pairwise swapped, so
            uint32_t* t32 = reinterpret_cast<uint32_t*>(tile);   // the two half-waves use different bank halves
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t v = odd ? d[j ^ 1] : d[j];
                const int row = rc * 8 + (odd ? (j ^ 1) : j);
                t32[row * 16 + swz_slot(kq >> 2, row) * 4 + (kq & 3)] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < SL; ++i) {
                const int kl = kq + (NTHREADS / CPR) * i;
                const T* e = reinterpret_cast<const T*>(&r[i]);
#pragma unroll
                for (int j = 0; j < VEC; ++j) tile[(rc * VEC + j) * Img::RS + kl] = e[j];
            }
        }
    }
};

template <class T, int BM, int BN, class AL, class BL, class EP>
__global__ __launch_bounds__(NTHREADS) void contraction_kernel(AL al, BL bl, EP ep, int K,
                                                               int tiles_n, int kt_per_split) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int BK = 4 * VEC;        // 64 bytes of k per row
    constexpr int LDSK = BK + VEC;     // padded row (allocation uses the larger image)
    constexpr int MT = BM / 32, NT = BN / 32;
    constexpr int KSTEPS = BK / Mma<T>::KI;
    typedef TileImage<T, AL::MC> ImgA;
    typedef TileImage<T, BL::MC> ImgB;
    static_assert(!(ImgA::SWZ || ImgB::SWZ) || KSTEPS == 1, "swizzled image assumes one MFMA k-step per tile");
    __shared__ __attribute__((aligned(16))) T lds[2][(BM + BN) * LDSK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = blockIdx.x;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int nkt = (K + BK - 1) / BK;
    const int kt0 = blockIdx.y * kt_per_split;
    const int kt1 = kt0 + kt_per_split < nkt ? kt0 + kt_per_split : nkt;

    Stager<T, BM, AL> sa;
    Stager<T, BN, BL> sb;
    sa.init(al, m0, tid);
    sb.init(bl, n0, tid);

    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment offsets of this lane inside the A / B images (row = wave offset + lane&15)
    const int fa0 = ImgA::frag(wm * (BM / 2) + (lane & 15), lane);
    const int fb0 = ImgB::frag(wn * (BN / 2) + (lane & 15), lane);

    if (kt0 < kt1) {
        sa.load(al, kt0 * BK);
        sb.load(bl, kt0 * BK);
        sa.store(lds[0], tid);
        sb.store(lds[0] + BM * LDSK, tid);
        __syncthreads();
        for (int kt = kt0; kt < kt1; ++kt) {
            const int buf = (kt - kt0) & 1;
            if (kt + 1 < kt1) { sa.load(al, (kt + 1) * BK); sb.load(bl, (kt + 1) * BK); }
            const T* ta = lds[buf] + fa0;
            const T* tb = lds[buf] + BM * LDSK + fb0;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                typename Mma<T>::Frag fa[MT], fb[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[i] = Mma<T>::load(ta + i * 16 * ImgA::RS, kk);
#pragma unroll
                for (int j = 0; j < NT; ++j) fb[j] = Mma<T>::load(tb + j * 16 * ImgB::RS, kk);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = Mma<T>::mma(fb[j], fa[i], acc[i][j]);
            }
            if (kt + 1 < kt1) { sa.store(lds[buf ^ 1], tid); sb.store(lds[buf ^ 1] + BM * LDSK, tid); }
            __syncthreads();
        }
    }
    // D = Btile x Atile  =>  lane holds C[m = .. + (lane&15)][n = .. + 4*(lane>>4) + 0..3]
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
            ep(m0 + wm * (BM / 2) + i * 16 + (lane & 15), n0 + wn * (BN / 2) + j * 16 + 4 * (lane >> 4),
               acc[i][j]);
}

// ------------------------------------------------------------------ bf16 kernel, generation 2
// Same tiling and epilogues, different staging: both operands go HBM -> LDS by LDS-DMA
// (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB per wave-instruction, no VGPR round trip, no
// ds_write), the next tile's DMA is in flight under the current tile's MFMAs.
//  * KC operands: image [row][32 k] (64-byte rows); the DMA destination is lane-linear, so the
//    XOR swizzle of the four 16-byte k-slots is applied on the SOURCE address (lane (row, slot s')
//    fetches chunk s' ^ SWZ(row)); fragments = ds_read_b128.
//  * MC operands: image [k][rows] exactly as in HBM (k-major), row-chunk index XOR-swizzled by k
//    on the source side; fragments = 2 x ds_read_b64_tr_b16 (the gfx950 LDS transpose read: within a
//    16-lane group lane i receives element (i%4) of the 8 bytes addressed by lane 4j + i/4, j = 0..3),
//    so the weight-gradient GEMMs need no transposition work at all.
static __device__ const uint32_t vtx_zero_page[4] = {0u, 0u, 0u, 0u};
typedef short v4s_t __attribute__((ext_vector_type(4)));

template <int ROWS> __device__ __forceinline__ int swz_mc(int chunk, int k) {
    if constexpr (ROWS == 128) return chunk ^ (((k & 3) << 1) | (((k >> 3) & 1) << 3));
    else return chunk ^ ((((k >> 1) & 1) << 1) | (((k >> 3) & 1) << 2));
}

template <int ROWS, class L> struct DmaStager {
    static constexpr bool MC = L::MC;
    static constexpr int NI = ROWS / 64;        // wave-instructions (1 KiB each) per wave per tile
    static constexpr int CH = ROWS / 8;         // 16-byte row chunks per k (MC image)
    static constexpr int KPI = 64 / CH;         // k rows per wave-instruction (MC image)
    typename L::State st;
    int kl[NI];                                 // MC: local k of each slot

    __device__ __forceinline__ void init(const L& l, int row0, int wave, int lane) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int q = wave + 4 * i;         // which 1 KiB piece of the tile image
            if constexpr (!MC) {
                const int row = 16 * q + (lane >> 2);
                l.init_slot(st, i, row0 + row, 8 * swz_slot(lane & 3, row));
            } else {
                kl[i] = q * KPI + lane / CH;
                l.init_slot(st, i, row0 + 8 * swz_mc<ROWS>(lane % CH, kl[i]));
            }
        }
    }
    __device__ __forceinline__ void issue(const L& l, int k0, bf16_t* tile, int wave) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const bf16_t* src;
            if constexpr (!MC) src = l.ptr(st, i, k0);
            else src = l.ptr(st, i, k0 + kl[i]);
            if (!src) src = reinterpret_cast<const bf16_t*>(vtx_zero_page);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(tile + (wave + 4 * i) * 512),
                                             16, 0, 0);
        }
    }
    // element offset (inside the tile image) of this lane's fragment source for fragment rows r0..r0+15
    __device__ static __forceinline__ int frag_off(int r0, int lane) {
        if constexpr (!MC) { const int row = r0 + (lane & 15); return row * 32 + swz_slot(lane >> 4, row) * 8; }
        else return 0;
    }
    __device__ static __forceinline__ bf16x8_t frag(const bf16_t* tile, int r0, int lane) {
        if constexpr (!MC) {
            return *reinterpret_cast<const bf16x8_t*>(tile + frag_off(r0, lane));
        } else {
            const int w = lane & 15, ka = 8 * (lane >> 4) + (w >> 2), rr = r0 + 4 * (w & 3);
            const bf16_t* pa = tile + ka * ROWS + swz_mc<ROWS>(rr >> 3, ka) * 8 + (rr & 7);
            const bf16_t* pb = tile + (ka + 4) * ROWS + swz_mc<ROWS>(rr >> 3, ka + 4) * 8 + (rr & 7);
            const v4s_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)pa);
            const v4s_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)pb);
            return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }
};

template <int BM, int BN, class AL, class BL, class EP>
__global__ __launch_bounds__(NTHREADS) void contraction_v2_kernel(AL al, BL bl, EP ep, int K, int tiles_n,
                                                                  int kt_per_split, int abl) {
    constexpr int BK = 32, MT = BM / 32, NT = BN / 32;
    constexpr int TILE = (BM + BN) * BK;       // elements per stage
    constexpr int STAGES = 3;                  // 3 x 16 KiB (128x128): tile kt+2 is in flight under tile kt
    constexpr int NDMA = BM / 64 + BN / 64;    // DMA instructions per wave per tile
    __shared__ __attribute__((aligned(1024))) bf16_t lds[STAGES * TILE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = blockIdx.x;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int nkt = (K + BK - 1) / BK;
    const int kt0 = blockIdx.y * kt_per_split;
    const int kt1 = kt0 + kt_per_split < nkt ? kt0 + kt_per_split : nkt;

    DmaStager<BM, AL> sa;
    DmaStager<BN, BL> sb;
    sa.init(al, m0, wave, lane);
    sb.init(bl, n0, wave, lane);

    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // s_waitcnt immediates (gfx9 encoding: vmcnt[3:0]|[15:14], expcnt[6:4], lgkmcnt[11:8]); only vmcnt waits
    constexpr int WAIT_ONE_TILE_LEFT = (NDMA & 0xF) | ((NDMA >> 4) << 14) | (0x7 << 4) | (0xF << 8);
    constexpr int WAIT_ALL = 0 | (0x7 << 4) | (0xF << 8);

    // Pipeline: three LDS stages, one s_barrier per K step, the DMA of tile kt+2 is issued right after
    // the barrier of step kt (into the stage tile kt-1 occupied) and has two MFMA phases to land.
    // (Measured alternative: reading the fragments one step ahead into a second register set was SLOWER,
    //  540 vs 615 TF/s on 7680x4096x1024 -- see DESIGN.md section 6.)
    if (kt0 < kt1) {
        sa.issue(al, kt0 * BK, lds, wave);
        sb.issue(bl, kt0 * BK, lds + BM * BK, wave);
        if (kt0 + 1 < kt1) {
            sa.issue(al, (kt0 + 1) * BK, lds + TILE, wave);
            sb.issue(bl, (kt0 + 1) * BK, lds + TILE + BM * BK, wave);
        }
        int stage = 0;                          // stage holding tile kt
        for (int kt = kt0; kt < kt1; ++kt) {
            if (!(abl & 8)) {
            // this wave's pieces of tile kt have landed (tile kt+1 may still be in flight) ...
            if (kt + 1 < kt1) __builtin_amdgcn_s_waitcnt(WAIT_ONE_TILE_LEFT);
            else __builtin_amdgcn_s_waitcnt(WAIT_ALL);
            // ... and after the barrier so have everybody's; all waves are also done reading the stage
            // that held tile kt-1, which is the one tile kt+2 is DMA'd into next.
            __builtin_amdgcn_s_barrier();
            }
            const bf16_t* cur = lds + stage * TILE;
            if (kt + 2 < kt1 && !(abl & 4)) {
                const int s2 = stage + 2 >= STAGES ? stage + 2 - STAGES : stage + 2;
                sa.issue(al, (kt + 2) * BK, lds + s2 * TILE, wave);
                sb.issue(bl, (kt + 2) * BK, lds + s2 * TILE + BM * BK, wave);
            }
            bf16x8_t fa[MT], fb[NT];
            if (!(abl & 2) || kt == kt0) {
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = DmaStager<BM, AL>::frag(cur, wm * (BM / 2) + i * 16, lane);
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[j] = DmaStager<BN, BL>::frag(cur + BM * BK, wn * (BN / 2) + j * 16, lane);
            }
            if (!(abl & 1)) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < MT; ++i) asm volatile("" ::"v"(fa[i]));
#pragma unroll
                for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(fb[j]));
            }
            stage = stage + 1 >= STAGES ? 0 : stage + 1;
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
            ep(m0 + wm * (BM / 2) + i * 16 + (lane & 15), n0 + wn * (BN / 2) + j * 16 + 4 * (lane >> 4),
               acc[i][j]);
}

extern int g_vtx_contraction_generation;
extern int g_vtx_ablate;   // measurement only: bit0 no MFMA, bit1 no fragment reads, bit2 no DMA, bit3 no barrier   // 2 (default): DMA kernel for bf16; 1: register-staged kernel

// ------------------------------------------------------------------ host-side launch
template <class T, int BM, int BN, class AL, class BL, class EP>
inline void launch_tile(const AL& al, const BL& bl, const EP& ep, int M, int N, int K, int split_k,
                        hipStream_t st) {
    constexpr int BK = 4 * Elem<T>::VEC;
    const int tiles_m = vtx_cdiv(M, BM), tiles_n = vtx_cdiv(N, BN);
    const int nkt = vtx_cdiv(K, BK);
    if (split_k < 1) split_k = 1;
    if (split_k > nkt) split_k = nkt;
    const int per = vtx_cdiv(nkt, split_k);
    split_k = vtx_cdiv(nkt, per);
    dim3 grid(tiles_m * tiles_n, split_k), block(NTHREADS);
    if constexpr (sizeof(T) == 2) {
        if (g_vtx_contraction_generation >= 2) {
            hipLaunchKernelGGL((contraction_v2_kernel<BM, BN, AL, BL, EP>), grid, block, 0, st, al, bl, ep, K,
                               tiles_n, per, g_vtx_ablate);
            return;
        }
    }
    hipLaunchKernelGGL((contraction_kernel<T, BM, BN, AL, BL, EP>), grid, block, 0, st, al, bl, ep, K,
                       tiles_n, per);
}

// Tile choice: 128x128 by default; narrower N tile for N <= 64; smaller M tile when M is tiny
// or when the grid would not fill the 256 CUs.
template <class T, template <class, int> class ALT, template <class, int> class BLT, class EP, class FA, class FB>
inline void launch_auto(FA make_a, FB make_b, const EP& ep, int M, int N, int K, int split_k, hipStream_t st) {
    const bool n64 = N <= 64;
    const long blocks128 = (long)vtx_cdiv(M, 128) * vtx_cdiv(N, n64 ? 64 : 128) * (split_k < 1 ? 1 : split_k);
    const bool m64 = M <= 64 || blocks128 < 256;
    if (m64) {
        ALT<T, 1> a; make_a(a);
        if (n64 || blocks128 < 128) { BLT<T, 1> b; make_b(b); launch_tile<T, 64, 64>(a, b, ep, M, N, K, split_k, st); }
        else { BLT<T, 2> b; make_b(b); launch_tile<T, 64, 128>(a, b, ep, M, N, K, split_k, st); }
    } else {
        ALT<T, 2> a; make_a(a);
        if (n64) { BLT<T, 1> b; make_b(b); launch_tile<T, 128, 64>(a, b, ep, M, N, K, split_k, st); }
        else { BLT<T, 2> b; make_b(b); launch_tile<T, 128, 128>(a, b, ep, M, N, K, split_k, st); }
    }
}

}  // namespace vtxg
