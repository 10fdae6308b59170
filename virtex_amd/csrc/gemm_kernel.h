// One MFMA contraction kernel for every GEMM-shaped op of the hot path (gfx950, wave64).
//
//   C[m][n] (+)= epilogue( sum_k A(m,k) * B(n,k) )
//
// Two generations share the tiling, the loaders' index spaces and the epilogues:
//   * generation 2 (`contraction_v2_kernel`, bf16, the product path): operand tiles go HBM -> LDS by LDS-DMA through
//     buffer descriptors (no VGPR round trip, hardware zero fill), 3 stages, one barrier per K step; fragments are
//     ds_read_b128 (k-contiguous images) or 2 x ds_read_b64_tr_b16 (k-major images); blocks of 4 or 8 waves.
//   * generation 1 (`contraction_kernel`, fp32 parity mode and whatever generation 2 does not take): tiles staged
//     global -> VGPR (16-byte loads, one K step ahead) -> LDS ([row][BK+pad], double buffered), 4 waves in a 2x2 grid.
//   bf16: v_mfma_f32_16x16x32_bf16 (fragment = 8 consecutive k)     fp32: v_mfma_f32_16x16x4_f32 (exact; parity mode)
// The MFMA is issued as D = Btile x Atile so that each lane ends up with 4 CONSECUTIVE n of
// one m (lane l: m = l&15, n = 4*(l>>4)..+3) -> vector epilogue loads/stores.
//
// Operands are described by "loaders" (how a 16-byte chunk of the tile maps to global memory), each with a pointer
// interface (init_slot / ptr: generation 1) and a buffer interface (view / binit / voff / soff: generation 2):
//   KC loaders: chunk = VEC consecutive k of one row   (row-major [rows][K] views, im2col-free
//               NHWC conv gathers for forward and input-gradient)
//   MC loaders: chunk = VEC consecutive rows at one k  (k-major [K][rows] views: weight
//               gradients; no transposition work -- the LDS transpose read does it)
#pragma once
#include <stdlib.h>

#include <atomic>
#include <type_traits>

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

// q = n / d for 0 <= n < 2^30 (vtx_common.h: two float estimates + one correction step); inv = 1.0f / d
__device__ __forceinline__ int fdiv(int n, int d, float inv) { return vtx_fdiv30(n, d, inv); }

// ------------------------------------------------------------------ buffer addressing (generation-2 kernel)
// The DMA kernel addresses every operand through a BUFFER DESCRIPTOR (buffer_load_dwordx4 ... offen lds):
//   address = descriptor base (wave-uniform) + voffset (32-bit, per lane) + soffset (32-bit, wave-uniform SGPR).
// A loader splits its index arithmetic accordingly: everything that depends on the lane (which row / pixel / channel
// chunk it stages) is folded ONCE, at block start, into a byte offset `off` (B-state, one VGPR per staged chunk);
// everything that depends on the K step (k0, and for convolutions the filter tap k0 falls into -- a K step never
// straddles taps because C % BK == 0) is scalar arithmetic.  Lanes that must deliver zeros -- rows outside the
// matrix, padding pixels, the K tail -- pass an out-of-range voffset: the hardware range check returns 0 and the
// LDS-DMA writes it.  Per staged chunk the K loop therefore issues 0 (plain matrices), 3 (convolution gathers: test
// the lane's tap-validity bit) or ~18 (pixel-major weight-gradient gather: carry-propagating (n,oh,ow) counters)
// 32-bit VALU instructions where the pointer form needed 10 / 25 / 45 with 64-bit multiplies, two float divisions per
// pixel and a zero-page select.  Operands must be smaller than 2 GiB (host check `buf_ok`; larger or oddly shaped
// problems run on the register-staged generation-1 kernel through the pointer interface `init_slot` / `ptr`).
struct BufView { const void* base; uint32_t bytes; };
constexpr uint32_t VTX_OOB = 0x80000000u;          // voffset of a lane that must read zeros (>= every accepted size)
constexpr double VTX_BUF_LIMIT = 2.0e9;            // bytes

// q = n / d for 0 <= n < 2^30 through the hardware reciprocal (one v_rcp_f32 + fdiv): the block
// prologues and the scattering epilogue decompose a few row indices each; a generic 32-bit division is ~25 instructions
__device__ __forceinline__ int qdiv(int n, int d) { return fdiv(n, d, __builtin_amdgcn_rcpf((float)d)); }

// ------------------------------------------------------------------ plain matrix loaders
// rows x K view, k contiguous: element (r,k) at p[r*ld + k]
template <class T, int SL> struct PlainKC {
    static constexpr bool MC = false;
    static constexpr bool TAILS = true;            // K need not be a multiple of the K step
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
    // ---- buffer interface
    struct BState { uint32_t off[SL]; int kc[SL]; };
    bool buf_ok(int) const { return (double)rows * (double)ld * sizeof(T) < VTX_BUF_LIMIT; }
    __device__ __forceinline__ BufView view() const { return {p, (uint32_t)(((long)(rows - 1) * ld + K) * (long)sizeof(T))}; }
    template <int BK> __device__ __forceinline__ void binit(BState& s, int i, int r, int kc) const {
        s.kc[i] = kc;
        s.off[i] = r < rows ? (uint32_t)(((long)r * ld + kc) * (long)sizeof(T)) : VTX_OOB;
    }
    template <bool FULL, int BK> __device__ __forceinline__ uint32_t voff(const BState& s, int i, int k0) const {
        if constexpr (FULL) return s.off[i];
        else return k0 + s.kc[i] < K ? s.off[i] : VTX_OOB;
    }
    template <int BK> __device__ __forceinline__ uint32_t soff(int k0) const { return (uint32_t)k0 * (uint32_t)sizeof(T); }
};
// K x rows view, rows contiguous: element (r,k) at p[k*ld + r]
template <class T, int SL> struct PlainMC {
    static constexpr bool MC = true;
    static constexpr bool TAILS = true;
    const T* p; long ld; int rows; int K;
    struct State { int r0[SL < 2 ? 2 : SL]; };
    // slot i stages VEC consecutive rows starting at global row r0 (r0 < 0: nothing), at global k
    __device__ __forceinline__ void init_slot(State& s, int i, int r0) const { s.r0[i] = r0 < rows ? r0 : -1; }
    __device__ __forceinline__ const T* ptr(const State& s, int i, int k) const {
        return (s.r0[i] >= 0 && k < K) ? p + (long)k * ld + s.r0[i] : nullptr;
    }
    // ---- buffer interface: slot i stages rows r0..r0+7 at k = k0 + kl (kl = the slot's k inside the staged tile)
    struct BState { uint32_t off[SL < 2 ? 2 : SL]; };
    bool buf_ok(int) const { return (double)K * (double)ld * sizeof(T) < VTX_BUF_LIMIT; }
    __device__ __forceinline__ BufView view() const {
        return {p, (uint32_t)(((long)(K - 1) * ld + ((rows + 7) & ~7)) * (long)sizeof(T))};
    }
    template <int BK> __device__ __forceinline__ void binit(BState& s, int i, int r0, int kl, int /*k_first*/) const {
        s.off[i] = r0 < rows ? (uint32_t)(((long)kl * ld + r0) * (long)sizeof(T)) : VTX_OOB;
    }
    template <bool FULL, int BK> __device__ __forceinline__ uint32_t voff(BState& s, int i, int k0, int kl) const {
        if constexpr (FULL) return s.off[i];
        else return k0 + kl < K ? s.off[i] : VTX_OOB;
    }
    template <int BK> __device__ __forceinline__ uint32_t soff(int k0) const { return (uint32_t)((long)k0 * ld * (long)sizeof(T)); }
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
    // ---- buffer interface.  Two shapes of K step:
    //  * tap mode (C % BK == 0): the step lies inside ONE filter tap (kh,kw): soffset = that tap's pixel offset + the
    //    step's first channel, the lane tests bit `tap` of its validity mask (which taps of ITS output pixel are
    //    inside the image, computed once);
    //  * row mode (C < BK, S*C == BK, pad 0: the packed 4-channel stem): the step is one whole filter ROW, i.e. BK
    //    contiguous elements of the input row; soffset = kh input rows, every lane is valid (the image carries its
    //    own zero frame).
    static constexpr bool TAILS = false;
    struct BState { uint32_t off[SL]; uint32_t mask[SL]; };
    long bias() const { return ((long)g.pad * g.W + g.pad) * g.C; }     // elements in front of x a padding tap may address
    bool buf_ok(int BK) const {
        const bool tap = g.C % BK == 0 && g.R * g.S <= 32;
        const bool row = g.C < BK && g.S * g.C == BK && g.pad == 0;
        return (tap || row) && K % BK == 0 && ((double)g.N * g.H * g.W * g.C + (double)bias()) * sizeof(T) < VTX_BUF_LIMIT;
    }
    __device__ __forceinline__ BufView view() const {
        const long b = ((long)g.pad * g.W + g.pad) * g.C;
        return {x - b, (uint32_t)(((long)g.N * g.H * g.W * g.C + b) * (long)sizeof(T))};
    }
    template <int BK> __device__ __forceinline__ void binit(BState& s, int i, int m, int kc) const {
        const bool ok = m < rows;
        const int mm = ok ? m : 0;
        const int n = qdiv(mm, g.OH * g.OW), rem = mm - n * g.OH * g.OW;
        const int oh = qdiv(rem, g.OW), ow = rem - oh * g.OW;
        const int ih0 = oh * g.stride - g.pad, iw0 = ow * g.stride - g.pad;
        const long pix = ((long)n * g.H + ih0) * g.W + iw0;             // >= -(pad*W + pad)
        s.off[i] = (uint32_t)(((pix + (long)g.pad * g.W + g.pad) * g.C + kc) * (long)sizeof(T));
        uint32_t mk = 0;
        if (g.C >= BK) {
            for (int kh = 0, t = 0; kh < g.R; ++kh)
                for (int kw = 0; kw < g.S; ++kw, ++t)
                    if (ok && (unsigned)(ih0 + kh) < (unsigned)g.H && (unsigned)(iw0 + kw) < (unsigned)g.W) mk |= 1u << t;
        } else mk = 1u;                                                  // row mode: rows past M re-read pixel 0 (their results are discarded)
        s.mask[i] = mk;
    }
    template <bool FULL, int BK> __device__ __forceinline__ uint32_t voff(const BState& s, int i, int k0) const {
        const int tap = g.C >= BK ? (k0 >> g.logC) : 0;                   // wave-uniform
        return ((s.mask[i] >> tap) & 1u) ? s.off[i] : VTX_OOB;
    }
    template <int BK> __device__ __forceinline__ uint32_t soff(int k0) const {
        if (g.C >= BK) {
            const int tap = k0 >> g.logC, ci0 = k0 & (g.C - 1);
            const int kh = (tap * g.rcpS) >> 16, kw = tap - kh * g.S;
            return (uint32_t)(((kh * g.W + kw) * g.C + ci0) * (int)sizeof(T));
        }
        return (uint32_t)((k0 / BK) * g.W * g.C * (int)sizeof(T));
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
    // ---- buffer interface (stride 1; KO % BK == 0 so that a K step lies inside one tap).  Row m = input pixel
    // (n,ih,iw) reads dy pixel P - (kh*OW + kw) with P = (n*OH + ih+pad)*OW + iw+pad: the lane part is P, the tap part
    // is uniform and NEGATIVE, so the descriptor starts `maxoff` elements before dy and soffset = maxoff - tap offset.
    static constexpr bool TAILS = false;
    struct BState { uint32_t off[SL]; uint32_t mask[SL]; };
    long maxoff() const { return ((long)(g.R - 1) * g.OW + (g.S - 1)) * g.KO; }
    bool buf_ok(int BK) const {
        return g.stride == 1 && g.KO % BK == 0 && g.R * g.S <= 32 && K % BK == 0 &&
               ((double)g.N * g.OH * g.OW * g.KO + 2.0 * (double)maxoff() + g.KO) * sizeof(T) < VTX_BUF_LIMIT;
    }
    __device__ __forceinline__ BufView view() const {
        const long mo = ((long)(g.R - 1) * g.OW + (g.S - 1)) * g.KO;
        return {dy - mo, (uint32_t)(((long)g.N * g.OH * g.OW * g.KO + 2 * mo + g.KO) * (long)sizeof(T))};
    }
    template <int BK> __device__ __forceinline__ void binit(BState& s, int i, int m, int kc) const {
        const bool ok = m < rows;
        const int mm = ok ? m : 0;
        const int n = qdiv(mm, g.H * g.W), rem = mm - n * g.H * g.W;
        const int ih = qdiv(rem, g.W), iw = rem - ih * g.W;
        const long P = ((long)n * g.OH + ih + g.pad) * g.OW + iw + g.pad;
        s.off[i] = (uint32_t)((P * g.KO + kc) * (long)sizeof(T));
        uint32_t mk = 0;
        for (int kh = 0, t = 0; kh < g.R; ++kh)
            for (int kw = 0; kw < g.S; ++kw, ++t)
                if (ok && (unsigned)(ih + g.pad - kh) < (unsigned)g.OH && (unsigned)(iw + g.pad - kw) < (unsigned)g.OW) mk |= 1u << t;
        s.mask[i] = mk;
    }
    template <bool FULL, int BK> __device__ __forceinline__ uint32_t voff(const BState& s, int i, int k0) const {
        return ((s.mask[i] >> (k0 >> g.logKO)) & 1u) ? s.off[i] : VTX_OOB;
    }
    template <int BK> __device__ __forceinline__ uint32_t soff(int k0) const {
        const int tap = k0 >> g.logKO, co0 = k0 & (g.KO - 1);
        const int kh = (tap * g.rcpS) >> 16, kw = tap - kh * g.S;
        return (uint32_t)(((((g.R - 1) - kh) * g.OW + ((g.S - 1) - kw)) * g.KO + co0) * (int)sizeof(T));
    }
};

// stride-2 input gradient, one PARITY CLASS (pa,pb) of the input pixels at a time: pixel (2*ih2+pa,
// 2*iw2+pb) only receives taps with kh = (pa+p) mod 2 (+2, ...), so the class is a dense GEMM over its
// own tap list (1, 2 or 4 taps for a 3x3) instead of a 9-tap gather that is 3/4 zeros.
struct TapList { int n; int kh[4], kw[4]; };
template <class T, int SL> struct ConvDgradS2A {
    static constexpr bool MC = false;
    const T* dy; ConvGeo g; int rows; int K; int pa, pb; TapList taps;   // rows = N*(H/2)*(W/2), K = taps.n*KO
    struct State { long base[SL]; int ihp[SL], iwp[SL]; bool ok[SL]; int kc[SL]; };
    __device__ __forceinline__ void init_slot(State& s, int i, int m, int kc) const {
        s.kc[i] = kc;
        s.ok[i] = m < rows;
        const int mm = s.ok[i] ? m : 0;
        const int h2 = g.H >> 1, w2 = g.W >> 1;
        const int n = mm / (h2 * w2), rem = mm - n * h2 * w2;
        const int ih2 = rem / w2, iw2 = rem - ih2 * w2;
        s.base[i] = (long)n * g.OH * g.OW * g.KO;
        s.ihp[i] = 2 * ih2 + pa + g.pad;
        s.iwp[i] = 2 * iw2 + pb + g.pad;
    }
    __device__ __forceinline__ const T* ptr(const State& s, int i, int k0) const {
        const int k = k0 + s.kc[i];
        const int t = k >> g.logKO, co = k & (g.KO - 1);
        const int tt = t < 4 ? t : 3;
        const int oh = (s.ihp[i] - taps.kh[tt]) >> 1, ow = (s.iwp[i] - taps.kw[tt]) >> 1;
        const bool ok = s.ok[i] && k < K && (unsigned)oh < (unsigned)g.OH && (unsigned)ow < (unsigned)g.OW;
        return ok ? dy + s.base[i] + ((long)oh * g.OW + ow) * g.KO + co : nullptr;
    }
    // ---- buffer interface.  ihp and the class's kh have the same parity, so (ihp - kh) >> 1 = (ihp >> 1) - (kh >> 1):
    // lane part P = (n*OH + (ihp>>1))*OW + (iwp>>1), uniform tap part (kh>>1)*OW + (kw>>1) <= OW + 1 (filters <= 4x4).
    static constexpr bool TAILS = false;
    struct BState { uint32_t off[SL]; uint32_t mask[SL]; };
    bool buf_ok(int BK) const {
        return g.KO % BK == 0 && K % BK == 0 &&
               ((double)g.N * g.OH * g.OW * g.KO + 2.0 * (g.OW + 1.0) * g.KO + g.KO) * sizeof(T) < VTX_BUF_LIMIT;
    }
    __device__ __forceinline__ BufView view() const {
        const long mo = (long)(g.OW + 1) * g.KO;
        return {dy - mo, (uint32_t)(((long)g.N * g.OH * g.OW * g.KO + 2 * mo + g.KO) * (long)sizeof(T))};
    }
    template <int BK> __device__ __forceinline__ void binit(BState& s, int i, int m, int kc) const {
        const bool ok = m < rows;
        const int mm = ok ? m : 0;
        const int h2 = g.H >> 1, w2 = g.W >> 1;
        const int n = qdiv(mm, h2 * w2), rem = mm - n * h2 * w2;
        const int ih2 = qdiv(rem, w2), iw2 = rem - ih2 * w2;
        const int ihp = 2 * ih2 + pa + g.pad, iwp = 2 * iw2 + pb + g.pad;
        const long P = ((long)n * g.OH + (ihp >> 1)) * g.OW + (iwp >> 1);
        s.off[i] = (uint32_t)((P * g.KO + kc) * (long)sizeof(T));
        uint32_t mk = 0;
        for (int t = 0; t < 4; ++t)
            if (t < taps.n && ok && (unsigned)((ihp >> 1) - (taps.kh[t] >> 1)) < (unsigned)g.OH &&
                (unsigned)((iwp >> 1) - (taps.kw[t] >> 1)) < (unsigned)g.OW) mk |= 1u << t;
        s.mask[i] = mk;
    }
    template <bool FULL, int BK> __device__ __forceinline__ uint32_t voff(const BState& s, int i, int k0) const {
        return ((s.mask[i] >> (k0 >> g.logKO)) & 1u) ? s.off[i] : VTX_OOB;
    }
    template <int BK> __device__ __forceinline__ uint32_t soff(int k0) const {
        const int t = k0 >> g.logKO, co0 = k0 & (g.KO - 1);
        const int tt = t < 4 ? t : 3;
        const int tapoff = (taps.kh[tt] >> 1) * g.OW + (taps.kw[tt] >> 1);
        return (uint32_t)((((g.OW + 1) - tapoff) * g.KO + co0) * (int)sizeof(T));
    }
};
// B operand of the above: rows = input channels of wt[C][R][S][KO], k = (tap index, co)
template <class T, int SL> struct TapKC {
    static constexpr bool MC = false;
    const T* p; long ld; int rows; int K; int logKO; int S; TapList taps;
    struct State { const T* rp[SL]; int kc[SL]; };
    __device__ __forceinline__ void init_slot(State& s, int i, int r, int kc) const {
        s.kc[i] = kc;
        s.rp[i] = r < rows ? p + (long)r * ld : nullptr;
    }
    __device__ __forceinline__ const T* ptr(const State& s, int i, int k0) const {
        const int k = k0 + s.kc[i];
        if (!s.rp[i] || k >= K) return nullptr;
        const int t = k >> logKO, co = k & ((1 << logKO) - 1);
        const int tt = t < 4 ? t : 3;
        return s.rp[i] + ((long)(taps.kh[tt] * S + taps.kw[tt]) << logKO) + co;
    }
    // ---- buffer interface
    static constexpr bool TAILS = false;
    struct BState { uint32_t off[SL]; };
    bool buf_ok(int BK) const { return (1 << logKO) % BK == 0 && K % BK == 0 && (double)rows * (double)ld * sizeof(T) < VTX_BUF_LIMIT; }
    __device__ __forceinline__ BufView view() const { return {p, (uint32_t)((long)rows * ld * (long)sizeof(T))}; }
    template <int BK> __device__ __forceinline__ void binit(BState& s, int i, int r, int kc) const {
        s.off[i] = r < rows ? (uint32_t)(((long)r * ld + kc) * (long)sizeof(T)) : VTX_OOB;
    }
    template <bool FULL, int BK> __device__ __forceinline__ uint32_t voff(const BState& s, int i, int) const { return s.off[i]; }
    template <int BK> __device__ __forceinline__ uint32_t soff(int k0) const {
        const int t = k0 >> logKO, co0 = k0 & ((1 << logKO) - 1);
        const int tt = t < 4 ? t : 3;
        return (uint32_t)((((taps.kh[tt] * S + taps.kw[tt]) << logKO) + co0) * (int)sizeof(T));
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
    // ---- buffer interface.  A slot stages 8 consecutive rows (one tap, 8 channels -- or two pixels of the packed
    // stem) at ONE output pixel, and its pixel advances by exactly BK per K step, so the slot carries (oh, ow) and the
    // byte offset `pos` of its input element as counters: adding BK = (dn, doh, dow) in the mixed radix (N, OH, OW)
    // needs one conditional subtraction per digit (dow < OW, doh < OH), each carry moves `pos` by a constant.  No
    // division, no 64-bit arithmetic in the K loop.  dhw packs the lane's tap displacement (kh - pad, kw - pad).
    static constexpr bool TAILS = true;
    struct BState { int oh[SL < 2 ? 2 : SL], ow[SL < 2 ? 2 : SL]; uint32_t pos[SL < 2 ? 2 : SL]; int dh[SL < 2 ? 2 : SL], dw[SL < 2 ? 2 : SL]; };
    bool buf_ok(int) const { return (double)g.N * g.H * g.W * g.C * sizeof(T) < VTX_BUF_LIMIT; }
    __device__ __forceinline__ BufView view() const { return {x, (uint32_t)((long)g.N * g.H * g.W * g.C * (long)sizeof(T))}; }
    template <int BK> __device__ __forceinline__ void binit(BState& s, int i, int r0, int kl, int k_first) const {
        const int tap = r0 >> g.logC, ci = r0 & (g.C - 1);
        const int kh = (tap * g.rcpS) >> 16, kw = tap - kh * g.S;
        const int pix = k_first + kl;                                    // this slot's first output pixel
        const int ohow = g.OH * g.OW;
        const int n = qdiv(pix, ohow), rem = pix - n * ohow;
        const int oh = qdiv(rem, g.OW), ow = rem - oh * g.OW;
        s.oh[i] = oh; s.ow[i] = ow;
        s.dh[i] = r0 < rows ? kh - g.pad : (1 << 28);                    // invalid row chunk: never inside the image
        s.dw[i] = kw - g.pad;
        const long e = (((long)n * g.H + oh * g.stride + kh - g.pad) * g.W + ow * g.stride + kw - g.pad) * g.C + ci;
        s.pos[i] = (uint32_t)(e * (long)sizeof(T));                      // meaningful only where the lane is valid
    }
    template <bool FULL, int BK> __device__ __forceinline__ uint32_t voff(BState& s, int i, int k0, int kl) const {
        const int ih = (s.oh[i] << g.logStride) + s.dh[i], iw = (s.ow[i] << g.logStride) + s.dw[i];
        bool ok = (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
        if constexpr (!FULL) ok = ok && k0 + kl < K;
        const uint32_t r = ok ? s.pos[i] : VTX_OOB;
        // advance by BK output pixels (all step constants are wave-uniform)
        const int dn = BK / (g.OH * g.OW), r1 = BK - dn * g.OH * g.OW, doh = r1 / g.OW, dow = r1 - doh * g.OW;
        const int esz = g.C * (int)sizeof(T);
        const int step = ((dn * g.H + doh * g.stride) * g.W + dow * g.stride) * esz;
        const int c1 = (g.stride * g.W - g.OW * g.stride) * esz;        // ow wrapped: next output row
        const int c2 = (g.H - g.OH * g.stride) * g.W * esz;             // oh wrapped: next image
        int ow = s.ow[i] + dow, oh = s.oh[i] + doh;
        uint32_t pos = s.pos[i] + (uint32_t)step;
        if (ow >= g.OW) { ow -= g.OW; oh += 1; pos += (uint32_t)c1; }
        if (oh >= g.OH) { oh -= g.OH; pos += (uint32_t)c2; }
        s.ow[i] = ow; s.oh[i] = oh; s.pos[i] = pos;
        return r;
    }
    template <int BK> __device__ __forceinline__ uint32_t soff(int) const { return 0u; }
};

// ------------------------------------------------------------------ epilogues
enum { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2, ACT_RES_RELU = 3,      // 3: ReLU AFTER the residual add (inference Bottleneck tail)
       ACT_SOFTMAX_GRAD = 4 };   // out = g[m] * (exp(v - lse[m]) - [n == target[m]]): the cross-entropy gradient wrt the logits v

template <class T> __device__ __forceinline__ void st4(T* p, const float* v);
template <> __device__ __forceinline__ void st4<float>(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, const float* v) {
    *reinterpret_cast<uint2*>(p) = make_uint2(f2bf2(v[0], v[1]),
                                              f2bf2(v[2], v[3]));
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

template <class T> __device__ __forceinline__ void st4v(T* p, f32x4_t v);
template <> __device__ __forceinline__ void st4v<float>(float* p, f32x4_t v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void st4v<bf16_t>(bf16_t* p, f32x4_t v) {
    *reinterpret_cast<uint2*>(p) = make_uint2(f2bf2(v[0], v[1]),
                                              f2bf2(v[2], v[3]));
}
__device__ __forceinline__ uint32_t bf2_add(uint32_t a, uint32_t b) {   // two packed bf16 + two packed bf16
    const float lo = __uint_as_float(a << 16) + __uint_as_float(b << 16);
    const float hi = __uint_as_float(a & 0xffff0000u) + __uint_as_float(b & 0xffff0000u);
    return f2bf2(lo, hi);
}
template <class T> __device__ __forceinline__ uint4 add16(uint4 a, uint4 b);
template <> __device__ __forceinline__ uint4 add16<bf16_t>(uint4 a, uint4 b) {
    return make_uint4(bf2_add(a.x, b.x), bf2_add(a.y, b.y), bf2_add(a.z, b.z), bf2_add(a.w, b.w));
}
template <> __device__ __forceinline__ uint4 add16<float>(uint4 a, uint4 b) {
    return make_uint4(__float_as_uint(__uint_as_float(a.x) + __uint_as_float(b.x)),
                      __float_as_uint(__uint_as_float(a.y) + __uint_as_float(b.y)),
                      __float_as_uint(__uint_as_float(a.z) + __uint_as_float(b.z)),
                      __float_as_uint(__uint_as_float(a.w) + __uint_as_float(b.w)));
}

__device__ __forceinline__ uint32_t bf2_relu(uint32_t a) {   // two packed bf16: negative (sign bit) -> +0
    return ((a & 0x8000u) ? 0u : (a & 0xffffu)) | ((a & 0x80000000u) ? 0u : (a & 0xffff0000u));
}
template <class T> __device__ __forceinline__ uint4 relu16(uint4 a);
template <> __device__ __forceinline__ uint4 relu16<bf16_t>(uint4 a) {
    return make_uint4(bf2_relu(a.x), bf2_relu(a.y), bf2_relu(a.z), bf2_relu(a.w));
}
template <> __device__ __forceinline__ uint4 relu16<float>(uint4 a) {
    return make_uint4((a.x & 0x80000000u) ? 0u : a.x, (a.y & 0x80000000u) ? 0u : a.y,
                      (a.z & 0x80000000u) ? 0u : a.z, (a.w & 0x80000000u) ? 0u : a.w);
}

// out = dropout(act(acc*alpha + bias)) + residual ; optional copy of the pre-activation
// (act = ACT_RES_RELU: out = relu(acc*alpha + bias + residual))
// Statistics modes of the wide epilogue (generation-2 kernel; compile-time, the plain epilogue pays nothing):
//   STATS_FWD  the output is the INPUT of a training-mode BatchNorm: per (block row, channel) partial sums of
//              (value - stat_shift[n]) and its square, taken from the stored (rounded) values;
//   STATS_BWD  the output is the gradient wrt the OUTPUT of a BatchNorm(+ReLU) whose input was bn_x: the epilogue
//              applies the ReLU mask (bn_y > 0 when the block output is given, else xhat*gamma+beta > 0 recomputed
//              from bn_x, else none), stores the MASKED gradient dz and emits the two sums BatchNorm's backward
//              needs, sum dz and sum dz*xhat -- the stand-alone reduction pass over dz and x disappears.
// Both are accumulated per lane while the wave-private strip is drained (a lane always drains the same 16-byte
// column chunk, so 2 x 8 accumulators suffice); per-channel parameters sit in LDS, not in registers.
enum { STATS_NONE = 0, STATS_FWD = 1, STATS_BWD = 2 };

template <class T> __device__ __forceinline__ void unpack16(uint4 w, float* f);
template <> __device__ __forceinline__ void unpack16<bf16_t>(uint4 w, float* f) {
    const uint32_t u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(u[i] << 16); f[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u); }
}
template <> __device__ __forceinline__ void unpack16<float>(uint4 w, float* f) {
    f[0] = __uint_as_float(w.x); f[1] = __uint_as_float(w.y); f[2] = __uint_as_float(w.z); f[3] = __uint_as_float(w.w);
}
template <class T> __device__ __forceinline__ uint4 pack16(const float* f);
template <> __device__ __forceinline__ uint4 pack16<bf16_t>(const float* f) {
    return make_uint4(f2bf2(f[0], f[1]), f2bf2(f[2], f[3]),
                      f2bf2(f[4], f[5]), f2bf2(f[6], f[7]));
}
template <> __device__ __forceinline__ uint4 pack16<float>(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}

template <class T, int STATS_MODE = STATS_NONE> struct EpiStore {
    static constexpr bool STAGED = true;   // generation-2 kernel: 16-byte stores via a wave-private LDS strip
    static constexpr int SMODE = STATS_MODE;
    static constexpr bool STATS = STATS_MODE != STATS_NONE;
    static constexpr bool ROWLSE = false;
    typedef T Out;
    T* out; long ldc; const float* bias; const T* residual; long ldr; T* preact; int act;
    float alpha; Dropout drop; int M, N;
    long split_stride = 0;   // split-K: slice s writes its partial result at out + s*split_stride ...
    long split_off = 0;      // ... = this block's s*split_stride, set by the kernel (set_slice)
    // Fused BatchNorm statistics (generation-2 kernel only): stat_parts[strip][2][N], one strip per block row
    // (tile_m) of the grid, summed by the BatchNorm finalize kernels.  N must be a multiple of 16 bytes' worth.
    float* stat_parts = nullptr;
    const float* stat_shift = nullptr;           // STATS_FWD
    const T* bn_x = nullptr; long ldx = 0;       // STATS_BWD: the BatchNorm's input, same [M][N] coordinates as `out`
    const T* bn_y = nullptr; long ldy = 0;       //            the post-ReLU block output (mask), or nullptr
    const uint8_t* bn_ybits = nullptr;           //            the same mask as ONE BIT per element (byte (m*ldy + n)/8, bit e = column
                                                 //            n+e > 0; written by vtx_bn_fwd): 1/16 of the bytes of bn_y; wins over bn_y
    const float* bn_mean = nullptr; const float* bn_rstd = nullptr;
    const float* bn_gamma = nullptr; const float* bn_beta = nullptr;   // mask recomputed from bn_x when bn_y == nullptr
    // ACT_SOFTMAX_GRAD (tied projection + cross-entropy backward: the logits are recomputed, never stored)
    const float* ce_lse = nullptr; const long long* ce_targets = nullptr;
    const float* ce_gout = nullptr; const float* ce_lc = nullptr; int ce_ignore = 0;
    __device__ __forceinline__ void softmax_grad(int m, int n, float* v) const {
        const long long t = ce_targets[m];
        const bool ignored = t == ce_ignore || t < 0 || t >= N;
        const float g = ignored ? 0.f : ce_gout[0] / ce_lc[1];
        const float l = ce_lse[m];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = g * (__expf(v[j] - l) - ((long long)(n + j) == t ? 1.f : 0.f));
    }
    // Row scatter for the parity-decomposed stride-2 input gradient: GEMM row m = (n, ih2, iw2) is output
    // pixel (n, 2*ih2+map_pa, 2*iw2+map_pb) of an H x W image.  map_on = 0: identity.
    int map_on = 0, map_H = 0, map_W = 0, map_pa = 0, map_pb = 0;
    // Non-temporal output stores (vtx_nt_policy: outputs of >= 200 MB bypass the caches on their way out).
    int nt = 0;
    __device__ __forceinline__ long out_row(int m) const {
        if (!map_on) return m;
        const int h2 = map_H >> 1, w2 = map_W >> 1;
        const int n = qdiv(m, h2 * w2), rem = m - n * h2 * w2;
        const int ih2 = qdiv(rem, w2), iw2 = rem - ih2 * w2;
        return ((long)n * map_H + 2 * ih2 + map_pa) * map_W + 2 * iw2 + map_pb;
    }
    // per-lane part (4 consecutive n of one m): everything except the residual add and the store
    // What the per-lane part needs from memory besides the accumulators is loaded ONCE per wave tile (the bias of the
    // lane's 4 columns in each column tile) / once per 16-row step (the cross-entropy row data of the lane's row), each
    // group in one block of loads behind one wait -- not once per 16x16 tile behind its own wait: sixteen L2 latencies
    // in a row per wave in the epilogue of every biased GEMM.
    struct RowData { long long t; float g, l; };
    __device__ __forceinline__ float4 bias4(int n) const {
        return n < N ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __device__ __forceinline__ RowData row_data(int m) const {
        RowData r{-1, 0.f, 0.f};
        if constexpr (STATS_MODE != STATS_NONE) return r;        // the cross-entropy gradient never carries statistics
        if (act == ACT_SOFTMAX_GRAD && m < M) {
            r.t = ce_targets[m];
            const bool ignored = r.t == ce_ignore || r.t < 0 || r.t >= N;
            r.g = ignored ? 0.f : ce_gout[0] / ce_lc[1];
            r.l = ce_lse[m];
        }
        return r;
    }
    __device__ __forceinline__ f32x4_t transform(int m, int n, f32x4_t v, float4 b, const RowData& rd) const {
        if (m >= M || n >= N) return v;
        v *= alpha;
        if constexpr (STATS_MODE == STATS_NONE) {                // statistics epilogues belong to bias-free convolutions
            if (bias) { v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
        }
        // statistics epilogues belong to the bias-free, activation-free convolutions in front of / behind a BatchNorm: the
        // activation / dropout / pre-activation code is not even compiled into them (it was: GELU, the dropout hash and the
        // softmax gradient made up most of the instructions of the fused-BatchNorm-backward kernels, skipped at run time
        // behind a dozen branches per 16x16 tile)
        if constexpr (STATS_MODE == STATS_NONE) {
            const long o = (long)m * ldc + n;
            if (preact) st4v<T>(preact + o, v);
            if (act == ACT_GELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
            } else if (act == ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
            } else if (act == ACT_SOFTMAX_GRAD) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = rd.g * (__expf(v[j] - rd.l) - ((long long)(n + j) == rd.t ? 1.f : 0.f));
            }
            if (drop.thresh) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = drop.apply(v[j], (uint64_t)(o + j));
            }
        }
        return v;
    }
    // The global operands of ONE 16-byte output chunk (row m, columns n .. n+EPV-1).  They do not depend on the
    // accumulators, so the kernel issues them for every chunk of a 16-row step BEFORE the step's transform and its trip
    // through the LDS strip, and consumes them (finish / finish_stats) afterwards: the loads of a step are all in flight
    // together and land under the strip traffic.  (Loading inside the per-chunk store -- rounds 1-2 -- made every chunk
    // a chain of residual load -> wait -> x / mask load -> wait -> sixteen single LDS parameter reads with a wait each ->
    // store: one 16-byte load in flight per wave in kernels that are HBM-bound.)
    // Only the fused BatchNorm-backward epilogue (three operand tensors, HBM-bound kernels) hoists; the other modes load
    // their residual inside finish (hoisting there costs registers in every instantiation for an operand most launches
    // do not have).
    struct Ops { uint4 res, x; uint32_t mb; };
    __device__ __forceinline__ Ops load_ops(int m, int n) const {
        Ops o;
        o.res = o.x = make_uint4(0u, 0u, 0u, 0u);
        o.mb = 0u;
        if constexpr (STATS_MODE != STATS_BWD) return o;
        if (m >= M || n >= N) return o;
        constexpr int EPV = 16 / (int)sizeof(T);
        const long mr = out_row(m);
        if (residual) o.res = *reinterpret_cast<const uint4*>(residual + mr * ldr + n);
        if constexpr (STATS_MODE == STATS_BWD) {
            static_assert(STATS_MODE != STATS_BWD || EPV == 8 || EPV == 4, "chunk = 16 bytes");
            o.x = *reinterpret_cast<const uint4*>(bn_x + mr * ldx + n);
            // the mask byte of the chunk (bf16: one byte per 16-byte chunk); without a bit mask the load is aimed at one
            // fixed valid byte (a broadcast, no traffic) instead of being skipped: a skipped load leaves a default to be
            // written on the other path, and hipcc guards that write with s_waitcnt vmcnt(0) -- in front of the step
            const uint8_t* bp = reinterpret_cast<const uint8_t*>(bn_rstd);
            if constexpr (EPV == 8) bp = bn_ybits ? bn_ybits + ((mr * ldy + n) >> 3) : bp;
            o.mb = *bp;
        }
        return o;
    }
    // 16 bytes (8 bf16 / 4 fp32) of row m starting at column n: + residual, store.  N is a multiple of 4,
    // so a chunk is either entirely inside the row, or (bf16 only) its first half is.
    __device__ __forceinline__ void finish(int m, int n, uint4 w, const Ops&) const {
        if (m >= M || n >= N) return;
        constexpr int EPV = 16 / (int)sizeof(T);
        const long mr = out_row(m);
        const long o = mr * ldc + n + split_off;
        const bool full = n + EPV <= N;
        if (residual) {
            if (full) w = add16<T>(w, *reinterpret_cast<const uint4*>(residual + mr * ldr + n));
            else {
                const uint2 r = *reinterpret_cast<const uint2*>(residual + mr * ldr + n);
                w = add16<T>(w, make_uint4(r.x, r.y, 0u, 0u));
            }
        }
        if (act == ACT_RES_RELU) w = relu16<T>(w);
        // (non-temporal stores measured: -17 % on the write-bound 64->256 1x1 convolution alone, nothing on the
        //  step -- the BatchNorm statistics pass that follows loses its Infinity-Cache hits)
        if (full) {
            if (nt) st16_nt(out + o, u32x4_t{w.x, w.y, w.z, w.w});
            else *reinterpret_cast<uint4*>(out + o) = w;
        } else *reinterpret_cast<uint2*>(out + o) = make_uint2(w.x, w.y);   // bf16: 4 elements
    }
    // The same, accumulating the statistics of this chunk.  par = this chunk's columns inside the block's LDS parameter
    // table ([4][PBN] floats: FWD {shift}; BWD {rstd, -mean*rstd, gamma, beta}), read as 16-byte vectors; s1/s2 = the
    // lane's accumulators.
    template <int EPV> __device__ static __forceinline__ void ld_par(const float* p, float* v) {
#pragma unroll
        for (int q = 0; q < EPV / 4; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(p + 4 * q);
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    }
    // The statistics arithmetic of ONE 16-byte chunk: w = the chunk's stored-format accumulators, r = its residual (BWD: ops.res,
    // FWD: passed in), mb = its mask bits; returns the chunk to store and adds its sums to s1 / s2.  Shared by the general path
    // (finish_stats: bounds, row map, per-tensor strides) and the interior-tile path (finish_stats_at): identical roundings.
    __device__ __forceinline__ uint4 stats_math(uint4 w, bool has_res, uint4 r16, uint4 x16, uint32_t mb, bool remask,
                                                const float* par, int PBN, float* s1, float* s2) const {
        constexpr int EPV = 16 / (int)sizeof(T);
        float f[EPV];
        unpack16<T>(w, f);
        if (has_res) {
            float r[EPV];
            unpack16<T>(r16, r);
#pragma unroll
            for (int e = 0; e < EPV; ++e) f[e] += r[e];
        }
        if constexpr (STATS_MODE == STATS_FWD) {
            float sh[EPV];
            ld_par<EPV>(par, sh);
            w = pack16<T>(f);
            unpack16<T>(w, f);                   // statistics of what is stored (what the BatchNorm will read)
#pragma unroll
            for (int e = 0; e < EPV; ++e) { const float d = f[e] - sh[e]; s1[e] += d; s2[e] += d * d; }
            return w;
        } else {
            // four elements (two stored words) at a time: the 16 parameter values of a half are all the registers the
            // table costs (the kernel class lives on six waves per SIMD = 80 VGPRs)
            static_assert(STATS_MODE != STATS_BWD || sizeof(T) == 2, "the fused BatchNorm backward epilogue is bf16");
            const uint32_t xi[4] = {x16.x, x16.y, x16.z, x16.w};
            uint32_t wo[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float rs[4], sh[4], ga[4], be[4], x[4], xh[4];
                ld_par<4>(par + 4 * h, rs);
                ld_par<4>(par + PBN + 4 * h, sh);
                ld_par<4>(par + 2 * PBN + 4 * h, ga);
                ld_par<4>(par + 3 * PBN + 4 * h, be);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    x[2 * k] = __uint_as_float(xi[2 * h + k] << 16);
                    x[2 * k + 1] = __uint_as_float(xi[2 * h + k] & 0xffff0000u);
                }
                float* g = f + 4 * h;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    g[e] = (mb >> (4 * h + e)) & 1u ? g[e] : 0.f;
                    xh[e] = x[e] * rs[e] + sh[e];
                    const bool pos = xh[e] * ga[e] + be[e] > 0.f;
                    g[e] = (!remask || pos) ? g[e] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {    // sums of what is stored: bn_bwd_apply_fused combines them with the ROUNDED dz
                    const uint32_t u = f2bf2(g[2 * k], g[2 * k + 1]);
                    wo[2 * h + k] = u;
                    g[2 * k] = __uint_as_float(u << 16); g[2 * k + 1] = __uint_as_float(u & 0xffff0000u);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1[4 * h + e] += g[e]; s2[4 * h + e] += g[e] * xh[e]; }
            }
            return make_uint4(wo[0], wo[1], wo[2], wo[3]);
        }
    }
    __device__ __forceinline__ void finish_stats(int m, int n, uint4 w, const Ops& ops, const float* par, int PBN, float* s1,
                                                 float* s2) const {
        if (m >= M || n >= N) return;
        constexpr int EPV = 16 / (int)sizeof(T);
        const long mr = out_row(m);
        uint4 r16 = make_uint4(0u, 0u, 0u, 0u), x16 = r16;
        if constexpr (STATS_MODE == STATS_BWD) { r16 = ops.res; x16 = ops.x; }
        else if (residual) r16 = *reinterpret_cast<const uint4*>(residual + mr * ldr + n);
        uint32_t mb = 0xffu;
        bool remask = false;
        if constexpr (STATS_MODE == STATS_BWD) {
            remask = !bn_y && !bn_ybits && bn_beta;      // mask recomputed from x (interior BatchNorms)
            mb = bn_ybits ? ops.mb : 0xffu;
            if (!bn_ybits && bn_y) {                     // the mask as the whole post-ReLU tensor (VIRTEX_AMD_RELU_BITS=0)
                float y[EPV];
                unpack16<T>(*reinterpret_cast<const uint4*>(bn_y + mr * ldy + n), y);
                mb = 0u;
#pragma unroll
                for (int e = 0; e < EPV; ++e) mb |= y[e] > 0.f ? 1u << e : 0u;
            }
        }
        w = stats_math(w, residual != nullptr, r16, x16, mb, remask, par, PBN, s1, s2);
        if (nt) st16_nt(out + mr * ldc + n, u32x4_t{w.x, w.y, w.z, w.w});
        else *reinterpret_cast<uint4*>(out + mr * ldc + n) = w;
    }
    // ---- interior tiles (block-uniform: every row and column of the tile exists, no row map, every tensor dense with row
    // stride N, no whole-tensor mask): the chunk is addressed by ONE precomputed element offset e = m*N + n shared by out,
    // residual, bn_x and (>> 3) the mask bits -- no bounds logic, no per-chunk 64-bit multiplies, no row-map division
    __device__ __forceinline__ Ops load_ops_at(long e) const {
        Ops o;
        o.res = o.x = make_uint4(0u, 0u, 0u, 0u);
        o.mb = 0u;
        if constexpr (STATS_MODE == STATS_BWD) {
            if (residual) o.res = *reinterpret_cast<const uint4*>(residual + e);
            o.x = *reinterpret_cast<const uint4*>(bn_x + e);
            const uint8_t* bp = bn_ybits ? bn_ybits + (e >> 3) : reinterpret_cast<const uint8_t*>(bn_rstd);   // (see load_ops)
            o.mb = *bp;
        }
        return o;
    }
    __device__ __forceinline__ void finish_stats_at(long e, uint4 w, const Ops& ops, const float* par, int PBN, float* s1, float* s2) const {
        uint32_t mb = 0xffu;
        bool remask = false;
        if constexpr (STATS_MODE == STATS_BWD) { remask = !bn_ybits && bn_beta; mb = bn_ybits ? ops.mb : 0xffu; }
        uint4 r16 = make_uint4(0u, 0u, 0u, 0u), x16 = r16;
        if constexpr (STATS_MODE == STATS_BWD) { r16 = ops.res; x16 = ops.x; }
        w = stats_math(w, STATS_MODE == STATS_BWD && residual != nullptr, r16, x16, mb, remask, par, PBN, s1, s2);
        if (nt) st16_nt(out + e, u32x4_t{w.x, w.y, w.z, w.w});
        else *reinterpret_cast<uint4*>(out + e) = w;
    }
    __device__ __forceinline__ void operator()(int m, int n, f32x4_t acc) const {
        if (m >= M || n >= N) return;
        float v[4] = {acc[0] * alpha, acc[1] * alpha, acc[2] * alpha, acc[3] * alpha};
        if (bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += bias[n + j];
        }
        const long mr = out_row(m);
        const long o = mr * ldc + n + split_off;
        if (preact) st4<T>(preact + o, v);
        if (act == ACT_GELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
        } else if (act == ACT_RELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        } else if (act == ACT_SOFTMAX_GRAD) softmax_grad(m, n, v);
        if (drop.thresh) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = drop.apply(v[j], (uint64_t)(o + j));
        }
        if (residual) {
            float r[4]; ld4<T>(residual + mr * ldr + n, r);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += r[j];
        }
        if (act == ACT_RES_RELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        st4<T>(out + o, v);
    }
};
// Host side: may this launch take the kernel instantiation whose statistics epilogue is compiled for INTERIOR tiles only
// (tile_epilogue<..., LEAN = true>)?  Every tile of the grid must lie inside the matrix, every tensor must be dense with row
// stride N, no row map, no whole-tensor mask -- at bs = 256 every BatchNorm-carrying convolution of ResNet-50 qualifies
// (M = 256 * H * W is a multiple of every block height).
template <class EP> inline bool lean_host_ok(const EP&, int, int, int, int) { return false; }
template <class T, int S> inline bool lean_host_ok(const EpiStore<T, S>& ep, int M, int N, int BM, int BN) {
    if (sizeof(T) != 2 || S == STATS_NONE) return false;
    static const bool off = getenv("VIRTEX_AMD_LEAN_EPILOGUE") && atoi(getenv("VIRTEX_AMD_LEAN_EPILOGUE")) == 0;      // A/B
    if (off || ep.map_on || ep.ldc != N || (N & 7) || M % BM != 0 || N % BN != 0) return false;
    if (S == STATS_FWD) return ep.residual == nullptr;
    return !ep.bn_y && ep.ldx == N && (!ep.residual || ep.ldr == N) && (!ep.bn_ybits || ep.ldy == N);
}
// out(fp32) += alpha * acc     (split-K partial sums and "+=" gradient accumulation)
struct EpiAtomic {
    struct Ops {};
    static constexpr bool STAGED = false;
    static constexpr bool STATS = false;
    static constexpr int SMODE = STATS_NONE;
    static constexpr bool ROWLSE = false;
    typedef float Out;
    float* out; long ldc; float alpha; int M, N;
    float* stat_parts = nullptr;
    const float* stat_shift = nullptr;
    __device__ __forceinline__ f32x4_t transform(int, int, f32x4_t v) const { return v; }
    __device__ __forceinline__ void operator()(int m, int n, f32x4_t acc) const {
        if (m >= M || n >= N) return;
        float* o = out + (long)m * ldc + n;
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(o + j, alpha * acc[j]);
    }
};

// Row-wise log-sum-exp partials of v = alpha*acc + bias over the columns of a WAVE tile -- the forward of the tied
// output projection + cross-entropy without ever writing the [M][N] logits: every wave emits, per row, the maximum
// and the sum of exp(v - max) over its own columns into pmax/psum[column group][M]; a tiny kernel folds the groups
// into lse[m].  The lane that holds column target[m] also records that logit.  Nothing else is stored.
struct EpiRowLse {
    struct Ops {};
    static constexpr bool STAGED = false;
    static constexpr bool STATS = false;
    static constexpr int SMODE = STATS_NONE;
    static constexpr bool ROWLSE = true;
    typedef float Out;
    const float* bias; float alpha; int M, N;
    const long long* targets; float* tgt_logit; float* pmax; float* psum;
    float* stat_parts = nullptr; const float* stat_shift = nullptr;
    const void* residual = nullptr; const void* preact = nullptr; const void* bn_x = nullptr; const void* bn_y = nullptr;
    __device__ __forceinline__ f32x4_t transform(int, int, f32x4_t v) const { return v; }
    __device__ __forceinline__ void operator()(int, int, f32x4_t) const {}
};

// the K slice a block works on (split-K): only the storing epilogue cares (and resolves its dropout epoch, vtx_common.h)
template <class EP> __device__ __forceinline__ void set_slice(EP&, int) {}
template <class T, int S> __device__ __forceinline__ void set_slice(EpiStore<T, S>& ep, int slice) {
    ep.split_off = (long)slice * ep.split_stride;
    ep.drop = ep.drop.resolved();
}

// mw / nw: first row / column of the wave tile; group: index of this column range (tile_n * waves-per-row + wn)
// (round 4: the bias values and the column bounds of a lane do not depend on the row step -- they were re-loaded, one 4-byte
//  global load per element behind possibly-aliasing stores, in every one of the MT steps; the targets of all steps are requested
//  up front too)
template <int MT, int NT, class EP>
__device__ __forceinline__ void rowlse_epilogue(const EP& ep, f32x4_t (&acc)[MT][NT], int mw, int nw, int lane, int group) {
    float bz[NT][4];
    bool okc[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = nw + j * 16 + 4 * (lane >> 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            okc[j][q] = n + q < ep.N;
            bz[j][q] = (okc[j][q] && ep.bias) ? ep.bias[n + q] : 0.f;
        }
    }
    long long tg[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = mw + i * 16 + (lane & 15);
        tg[i] = m < ep.M ? ep.targets[m] : -1;
    }
    vtx_loads_issued();
    const float alpha = ep.alpha;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = mw + i * 16 + (lane & 15);
        const bool mok = m < ep.M;
        const int tc = (int)tg[i] - nw - 4 * (lane >> 4);          // the target's column relative to this lane's first column
        float v[NT][4];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float x = acc[i][j][q] * alpha + bz[j][q];
                v[j][q] = x;
                if (okc[j][q]) {
                    mx = fmaxf(mx, x);
                    if (mok && tc == j * 16 + q) ep.tgt_logit[m] = x;
                }
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (okc[j][q]) sm += __expf(v[j][q] - mx);
        }
        sm += __shfl_xor(sm, 16, 64);
        sm += __shfl_xor(sm, 32, 64);
        if ((lane >> 4) == 0 && mok) {
            ep.pmax[(size_t)group * ep.M + m] = mx;
            ep.psum[(size_t)group * ep.M + m] = sm;
        }
    }
}

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
    set_slice(ep, (int)blockIdx.y);

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
    if constexpr (EP::ROWLSE) {
        rowlse_epilogue<MT, NT>(ep, acc, m0 + wm * (BM / 2), n0 + wn * (BN / 2), lane, (tile % tiles_n) * 2 + wn);
    } else {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                ep(m0 + wm * (BM / 2) + i * 16 + (lane & 15), n0 + wn * (BN / 2) + j * 16 + 4 * (lane >> 4),
                   acc[i][j]);
    }
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
typedef short v4s_t __attribute__((ext_vector_type(4)));

template <int ROWS> __device__ __forceinline__ int swz_mc(int chunk, int k) {
    if constexpr (ROWS >= 128) return chunk ^ (((k & 3) << 1) | (((k >> 3) & 1) << 3));
    else return chunk ^ ((((k >> 1) & 1) << 1) | (((k >> 3) & 1) << 2));
}

// 128-byte rows (BK = 64): the eight 16-byte k-slots of a row are XOR-permuted by (row >> 1) & 7 -- every 16-lane
// group of a ds_read_b128 fragment load (rows r..r+15 at one slot, or two adjacent slots) then covers all 64 banks.
__device__ __forceinline__ int swz_slot64(int slot, int row) { return slot ^ ((row >> 1) & 7); }

template <int ROWS, int NW, class L, int BK = 32> struct DmaStager {
    static constexpr bool MC = L::MC;
    static_assert(!MC || BK == 32, "k-major operands are staged 32 k at a time");
    static constexpr int SLOTS = BK / 8;        // 16-byte k-slots per row (KC image)
    static constexpr int RPI = 64 / SLOTS;      // rows per wave-instruction (KC image): 16 x 64 B or 8 x 128 B
    static constexpr int NI = MC ? ROWS / (16 * NW) : ROWS / (RPI * NW);   // wave-instructions (1 KiB each) per wave per tile
    static constexpr int CH = ROWS / 8;         // 16-byte row chunks per k (MC image)
    static constexpr int KPI = 64 / CH > 0 ? 64 / CH : 1;   // k rows per wave-instruction (MC image)
    static_assert(NI >= 1, "tile too small for this many waves");
    typename L::BState st;
    int kl[NI];                                 // MC: local k of each slot
    __amdgpu_buffer_rsrc_t rsrc;                // the operand's buffer descriptor (4 SGPRs)

    // k_first = first k of this block's K range (split-K slices start in the middle)
    __device__ __forceinline__ void init(const L& l, int row0, int wave, int lane, int k_first) {
        const BufView v = l.view();
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(v.base), (short)0, (int)v.bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int q = wave + NW * i;        // which 1 KiB piece of the tile image
            if constexpr (!MC) {
                const int row = RPI * q + lane / SLOTS;
                const int phys = lane % SLOTS;
                kl[i] = 0;
                l.template binit<BK>(st, i, row0 + row, 8 * (BK == 32 ? swz_slot(phys, row) : swz_slot64(phys, row)));
            } else if constexpr (CH <= 64) {
                kl[i] = q * KPI + lane / CH;
                l.template binit<BK>(st, i, row0 + 8 * swz_mc<ROWS>(lane % CH, kl[i]), kl[i], k_first);
            }
        }
    }
    template <bool FULL> __device__ __forceinline__ void issue_t(const L& l, int k0, bf16_t* tile, int wave) {
        const uint32_t so = l.template soff<BK>(k0);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            uint32_t vo;
            if constexpr (!MC) vo = l.template voff<FULL, BK>(st, i, k0);
            else vo = l.template voff<FULL, BK>(st, i, k0, kl[i]);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(tile + (wave + NW * i) * 512),
                                                     16, (int)vo, (int)so, 0, 0);
        }
    }
    // Tiles are issued in increasing k0, each exactly once (position-tracking loaders rely on it).
    __device__ __forceinline__ void issue(const L& l, int k0, bf16_t* tile, int wave) {
        if constexpr (L::TAILS) {
            if (k0 + BK <= l.K) issue_t<true>(l, k0, tile, wave);        // wave-uniform branch
            else issue_t<false>(l, k0, tile, wave);
        } else issue_t<true>(l, k0, tile, wave);
    }
    // fragment of rows r0..r0+15 for the 32-deep MFMA step `h` of the staged tile (h = 0 when BK = 32)
    __device__ static __forceinline__ bf16x8_t frag(const bf16_t* tile, int r0, int lane, int h = 0) {
        if constexpr (!MC) {
            const int row = r0 + (lane & 15);
            if constexpr (BK == 32)
                return *reinterpret_cast<const bf16x8_t*>(tile + row * 32 + swz_slot(lane >> 4, row) * 8);
            else
                return *reinterpret_cast<const bf16x8_t*>(tile + row * 64 + swz_slot64(4 * h + (lane >> 4), row) * 8);
        } else {
            const int w = lane & 15, ka = 8 * (lane >> 4) + (w >> 2), rr = r0 + 4 * (w & 3);
            const bf16_t* pa = tile + ka * ROWS + swz_mc<ROWS>(rr >> 3, ka) * 8 + (rr & 7);
            const bf16_t* pb = tile + (ka + 4) * ROWS + swz_mc<ROWS>(rr >> 3, ka + 4) * 8 + (rr & 7);
            const v4s_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)pa);
            const v4s_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)pb);
            return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }
    // k-major images inside the DMA pipeline: the two transposing reads of a fragment as asm statements (vtx_ds_read_tr16:
    // the builtin makes hipcc drain every LDS-DMA in flight before the first fragment read of each K step).  The caller
    // issues all fragments of the step, then vtx_ds_tr_wait(), then combines the halves.
    __device__ static __forceinline__ void frag_tr(const bf16_t* tile, int r0, int lane, vtx_v4s_t (&out)[2]) {
        static_assert(MC, "transposing reads belong to k-major images");
        const int w = lane & 15, ka = 8 * (lane >> 4) + (w >> 2), rr = r0 + 4 * (w & 3);
        out[0] = vtx_ds_read_tr16(tile + ka * ROWS + swz_mc<ROWS>(rr >> 3, ka) * 8 + (rr & 7));
        out[1] = vtx_ds_read_tr16(tile + (ka + 4) * ROWS + swz_mc<ROWS>(rr >> 3, ka + 4) * 8 + (rr & 7));
    }
};

// rows I .. MT-1 of a k-major K step (see the kernel): wait for row I's A fragment, multiply it with every B fragment
template <int I, int MT, int NT>
__device__ __forceinline__ void mc_rows(vtx_v4s_t (&ra)[MT][2], vtx_v4s_t (&rb)[NT][2], bf16x8_t (&fb)[NT], f32x4_t (&acc)[MT][NT]) {
    if constexpr (I < MT) {
        vtx_ds_tr_wait_n<2 * (MT - 1 - I)>();
        if constexpr (I == 0) {
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[j] = __builtin_shufflevector(rb[j][0], rb[j][1], 0, 1, 2, 3, 4, 5, 6, 7);
        }
        const bf16x8_t fa = __builtin_shufflevector(ra[I][0], ra[I][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[I][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa, acc[I][j], 0, 0, 0);
        mc_rows<I + 1, MT, NT>(ra, rb, fb, acc);
    }
}

// ------------------------------------------------------------------ the epilogue of a block tile (shared by the kernels below)
template <int BN, int NW, class EP> struct EpiShape {
    static constexpr int PPT = EP::SMODE == STATS_NONE ? 1 : (BN + 64 * NW - 1) / (64 * NW);    // parameter-table columns per thread
};
// Statistics epilogues: the per-channel parameters of this block's columns are fetched in FRONT of the K loop, into
// registers, and written to their LDS table after it -- fetched there (rounds 1-2) they were three to four dependent L2
// round trips in front of the epilogue of blocks whose whole K loop is one or two steps.
template <int BN, int NW, class EP>
__device__ __forceinline__ void epi_prefetch(const EP& ep, float (&pre)[EpiShape<BN, NW, EP>::PPT][4], int tid, int n0) {
    constexpr int PPT = EpiShape<BN, NW, EP>::PPT;
    if constexpr (EP::SMODE != STATS_NONE) {
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int c = tid + q * 64 * NW, n = n0 + c;
            pre[q][0] = pre[q][1] = pre[q][2] = pre[q][3] = 0.f;
            if (c < BN && n < ep.N) {
                if constexpr (EP::SMODE == STATS_FWD) {
                    if (ep.stat_shift) pre[q][0] = ep.stat_shift[n];
                } else {
                    pre[q][0] = ep.bn_rstd[n]; pre[q][1] = ep.bn_mean[n];
                    if (ep.bn_gamma) pre[q][2] = ep.bn_gamma[n];
                    if (ep.bn_beta) pre[q][3] = ep.bn_beta[n];
                }
            }
        }
    } else {
        pre[0][0] = pre[0][1] = pre[0][2] = pre[0][3] = 0.f;
    }
}
#ifndef VTX_EPI_OPS_ALL
#define VTX_EPI_OPS_ALL 1      // 1: the epilogue operands of ALL 16-row steps of a wave tile are fetched up front (<= 4 chunks)
#endif
#ifndef VTX_EPI_ALL_MAX
#define VTX_EPI_ALL_MAX 4      // ... up to this many chunks per lane (measurement builds: 8 = the four-wave 128x128 tile too)
#endif
// acc: the wave's MT x NT accumulator tiles (D = Btile x Atile: lane holds C[m = .. + (lane&15)][n = .. + 4*(lane>>4) + 0..3]);
// lds: the block's stage memory (LDS_BYTES), free once every wave has left the K loop; tile_m / tile_n: the tile's place in
// the grid (statistics strip / column group)
template <int BM, int BN, int WM, int WN, int LDS_BYTES, class EP, bool LEAN = false>
__device__ __forceinline__ void tile_epilogue(EP& ep, f32x4_t (&acc)[BM / WM / 16][BN / WN / 16],
                                              float (&pre)[EpiShape<BN, WM * WN, EP>::PPT][4], bf16_t* lds, int m0, int n0,
                                              int tile_m, int tile_n, int tid, int lane, int wave) {
    constexpr int NW = WM * WN, WTM = BM / WM, WTN = BN / WN, MT = WTM / 16, NT = WTN / 16;
    constexpr int PPT = EpiShape<BN, NW, EP>::PPT;
    const int wm = wave / WN, wn = wave % WN;
    // D = Btile x Atile  =>  lane holds C[m = .. + (lane&15)][n = .. + 4*(lane>>4) + 0..3]
    if constexpr (EP::ROWLSE) {
        rowlse_epilogue<MT, NT>(ep, acc, m0 + wm * WTM, n0 + wn * WTN, lane, tile_n * WN + wn);
    } else if constexpr (!EP::STAGED) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                ep(m0 + wm * WTM + i * 16 + (lane & 15), n0 + wn * WTN + j * 16 + 4 * (lane >> 4), acc[i][j]);
    } else {
        // Wide epilogue: 16 rows at a time go through a wave-private LDS strip so that every global
        // access is 16 bytes per lane and 8 (bf16) / 4 (fp32) full 128/256-byte row segments per
        // wave-instruction instead of sixteen 32-byte ones.
        typedef typename EP::Out TO;
        constexpr int ROWB = WTN * (int)sizeof(TO) + 16;          // padded strip row (bytes)
        constexpr int CPR = WTN * (int)sizeof(TO) / 16;           // 16-byte chunks per row
        constexpr int EPV = 16 / (int)sizeof(TO);                 // elements per chunk
        constexpr int SM = EP::SMODE;
        constexpr int PAR_OFF = (NW * 16 * ROWB + 15) & ~15;      // statistics: parameter table [4][BN], then
        constexpr int RED_OFF = PAR_OFF + 4 * BN * 4;             // the cross-wave fold [NW][2][WTN]
        static_assert(SM == STATS_NONE || (64 % CPR == 0 && RED_OFF + NW * 2 * WTN * 4 <= LDS_BYTES),
                      "statistics epilogue: a lane must keep its column chunk, and the tables must fit the stage memory");
        __syncthreads();                                          // every wave is done with the stages
        char* strip = reinterpret_cast<char*>(lds) + wave * (16 * ROWB);
        float* par = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + PAR_OFF);
        float* red = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + RED_OFF);
        float s1[SM != STATS_NONE ? EPV : 1], s2[SM != STATS_NONE ? EPV : 1];
        if constexpr (SM != STATS_NONE) {
#pragma unroll
            for (int e = 0; e < EPV; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
            for (int q = 0; q < PPT; ++q) {                       // per-channel parameters of this block's columns
                const int c = tid + q * 64 * NW;
                if (c < BN) {
                    if constexpr (SM == STATS_FWD) par[c] = pre[q][0];
                    else {
                        par[c] = pre[q][0]; par[BN + c] = -pre[q][1] * pre[q][0];
                        par[2 * BN + c] = pre[q][2]; par[3 * BN + c] = pre[q][3];
                    }
                }
            }
            __syncthreads();
        }
        // Interior tiles of the bf16 statistics epilogues (block-uniform test; round 4): no bounds logic, no row map, one
        // element offset per chunk -- the general loop below spends most of its instructions on those (measured on the
        // 8-wave kernels of gemm_v3.h, nothing overlapping: 11 500 / 19 800 cycles per 64x64 wave tile forward / backward
        // against 4 000 for a plain store)
        constexpr bool lean_done = LEAN;          // the host picked the instantiation (lean_host_ok): no second code path, no
        static_assert(!LEAN || (SM != STATS_NONE && sizeof(TO) == 2 && (16 * CPR) % 64 == 0), "lean epilogue: bf16 statistics modes");   // run-time test
        if constexpr (LEAN) {
            {
                constexpr int NCHL = 16 * CPR / 64, RSTEP = 64 / CPR;      // chunks per lane and step; rows between them
                const int ch = lane % CPR, r0 = lane / CPR;
                const long e0 = (long)(m0 + wm * WTM + r0) * ep.N + (n0 + wn * WTN + ch * EPV);
                const float* parl = par + wn * WTN + ch * EPV;
                const char* rd_ptr = strip + r0 * ROWB + ch * 16;
                const float alpha = ep.alpha;
                // wave tiles of at most four chunks per lane: the operand chunks of ALL steps are requested up front
                constexpr bool ALL = VTX_EPI_OPS_ALL && SM == STATS_BWD && MT * NCHL <= VTX_EPI_ALL_MAX;
                typename EP::Ops o[ALL ? MT : 1][NCHL];
                if constexpr (ALL) {
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int q = 0; q < NCHL; ++q) o[i][q] = ep.load_ops_at(e0 + (long)(i * 16 + q * RSTEP) * ep.N);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i) {
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        st4v<TO>(reinterpret_cast<TO*>(strip + (lane & 15) * ROWB) + j * 16 + 4 * (lane >> 4), acc[i][j] * alpha);
                    if constexpr (!ALL) {
                        // the step's accumulators are on their way to the strip (their registers are free) BEFORE the operand
                        // chunks are requested: the 256x128 kernel lives on 128 VGPRs (two blocks per CU)
                        vtx_loads_issued();
#pragma unroll
                        for (int q = 0; q < NCHL; ++q) o[0][q] = ep.load_ops_at(e0 + (long)(i * 16 + q * RSTEP) * ep.N);
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int q = 0; q < NCHL; ++q) {
                        const uint4 w = *reinterpret_cast<const uint4*>(rd_ptr + q * RSTEP * ROWB);
                        ep.finish_stats_at(e0 + (long)(i * 16 + q * RSTEP) * ep.N, w, o[ALL ? i : 0][q], parl, BN, s1, s2);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        float4 bv[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) bv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (SM == STATS_NONE) {
            if (ep.bias) {
#pragma unroll
                for (int j = 0; j < NT; ++j) bv[j] = ep.bias4(n0 + wn * WTN + j * 16 + 4 * (lane >> 4));
            }
        }
        constexpr int NCH = (16 * CPR + 63) / 64;                 // chunks a lane drains per 16-row step
        // wave tiles of at most four output chunks per lane (8 waves on 128x128): the operand chunks of ALL steps up front
        constexpr bool OPS_ALL = VTX_EPI_OPS_ALL && SM == STATS_BWD && MT * NCH <= 4;
        typename EP::Ops ops[OPS_ALL ? MT : 1][NCH];
        if constexpr (OPS_ALL && !lean_done) {
            {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int q = 0; q < NCH; ++q) {
                    const int c = lane + 64 * q;
                    ops[i][q] = ep.load_ops(c < 16 * CPR ? m0 + wm * WTM + i * 16 + c / CPR : ep.M, n0 + wn * WTN + (c % CPR) * EPV);
                }
            }
        }
        if constexpr (!lean_done) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int mrow = m0 + wm * WTM + i * 16;
            // the step's global operands (residual, BatchNorm input, mask) first: independent of the accumulators
            if constexpr (!OPS_ALL) {
#pragma unroll
                for (int q = 0; q < NCH; ++q) {
                    const int c = lane + 64 * q;
                    ops[0][q] = ep.load_ops(c < 16 * CPR ? mrow + c / CPR : ep.M, n0 + wn * WTN + (c % CPR) * EPV);
                }
            }
            const typename EP::RowData rd = ep.row_data(mrow + (lane & 15));
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const f32x4_t v = ep.transform(mrow + (lane & 15), n0 + wn * WTN + j * 16 + 4 * (lane >> 4), acc[i][j], bv[j], rd);
                st4v<TO>(reinterpret_cast<TO*>(strip + (lane & 15) * ROWB) + j * 16 + 4 * (lane >> 4), v);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                const int c = lane + 64 * q;
                if (c < 16 * CPR) {
                    const int r = c / CPR, ch = c % CPR;
                    const uint4 w = *reinterpret_cast<const uint4*>(strip + r * ROWB + ch * 16);
                    if constexpr (SM == STATS_NONE) ep.finish(mrow + r, n0 + wn * WTN + ch * EPV, w, ops[OPS_ALL ? i : 0][q]);
                    else ep.finish_stats(mrow + r, n0 + wn * WTN + ch * EPV, w, ops[OPS_ALL ? i : 0][q], par + wn * WTN + ch * EPV, BN, s1, s2);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        }
        if constexpr (SM != STATS_NONE) {
            // fold the lanes that drained the same column chunk (they differ in the bits above log2(CPR)) ...
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
#pragma unroll
                for (int msk = CPR; msk < 64; msk <<= 1) { s1[e] += __shfl_xor(s1[e], msk, 64); s2[e] += __shfl_xor(s2[e], msk, 64); }
            }
            if (lane < CPR) {
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    red[(wave * 2 + 0) * WTN + lane * EPV + e] = s1[e];
                    red[(wave * 2 + 1) * WTN + lane * EPV + e] = s2[e];
                }
            }
            __syncthreads();
            // ... then the WM waves of each column range: one partial per (block row, channel), no atomics
            float* dst = ep.stat_parts + (size_t)tile_m * 2 * ep.N;
            for (int t = tid; t < 2 * BN; t += 64 * NW) {
                const int which = t / BN, c = t % BN, wcol = c / WTN, cc = c % WTN;
                float a = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < WM; ++w2) a += red[((w2 * WN + wcol) * 2 + which) * WTN + cc];
                if (n0 + c < ep.N) dst[(size_t)which * ep.N + n0 + c] = a;
            }
        }
    }
}

#ifndef VTX_EPI_BWD_OCC
#define VTX_EPI_BWD_OCC 4      // waves per SIMD the 8-wave 128x128 kernels with the fused BatchNorm-backward epilogue are built for
#endif
// Block = WM x WN waves; block tile BM x BN; wave tile (BM/WM) x (BN/WN).  Large tiles matter for the
// L2 -> LDS bandwidth, not only for LDS: a 128x128 tile needs 2*(128+128)*64 B per 2*128*128*32 flop
// = 64 flop/B, i.e. 39 TB/s of cache bandwidth at the MFMA peak (L2 delivers ~34); 256x256 needs half.
template <int BM, int BN, int WM, int WN, class AL, class BL, class EP, int BK = 32, int STAGES = 3, bool LEAN = false>
// Second launch bound = minimum waves per SIMD the register allocation must leave room for: 8-wave blocks on 128x128
// tiles are meant to run THREE per CU (six waves per SIMD = 80 VGPRs: the HBM-bound layers live on blocks in flight),
// 8-wave blocks on 256x128 tiles two (128 VGPRs), 4-wave blocks on 128x128 tiles three (168), on 128x64 / 64x128 four (128).
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 8 && BM * BN <= 128 * 128) ? (EP::SMODE == STATS_BWD ? VTX_EPI_BWD_OCC : 6) : (WM * WN == 8 && BM * BN <= 256 * 128) ? 4 :
                                           (WM * WN == 4 && BM * BN == 128 * 128) ? 3 : (WM * WN == 4 && BM * BN == 128 * 64) ? 4 : 1)
void contraction_v2_kernel(AL al, BL bl, EP ep, int K, int tiles_n,
                                                                      int kt_per_split, int abl) {
    constexpr int NW = WM * WN, WTM = BM / WM, WTN = BN / WN, MT = WTM / 16, NT = WTN / 16;
    constexpr int TILE = (BM + BN) * BK;       // elements per stage
    typedef DmaStager<BM, NW, AL, BK> SA;
    typedef DmaStager<BN, NW, BL, BK> SB;
    constexpr int NDMA = SA::NI + SB::NI;      // DMA instructions per wave per tile
    HIP_DYNAMIC_SHARED(bf16_t, lds)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch policy; a speed assumption
    // only).  Give every XCD a contiguous range of tiles so that the tiles sharing an A panel (same
    // tile_m, consecutive tile ids) hit the same private L2 instead of eight different ones.
    // Split-K launches (grid = tiles x slices, dispatched x-fastest): what the blocks of one XCD can share are the
    // operand panels of ONE K slice (every tile of a slice reads the same rows of k), so the XCD's contiguous range is
    // taken from the slice-major list of (slice, tile) pairs -- with the tile index alone deciding the XCD, the 8 tiles
    // of a small weight-gradient GEMM would each re-read their panels through a different L2.
    int tile, slice = 0;
    if (abl & 32) {                              // A/B switch (vtx_set_switch("tile_order", 1)): plain order, tile = block index
        tile = blockIdx.x; slice = blockIdx.y;
    } else if (gridDim.y == 1 || (abl & 16)) {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        slice = blockIdx.y;
    } else {
        const int T = gridDim.x, nwg = T * gridDim.y, L = blockIdx.y * T + blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = L & 7, idx = L >> 3;
        const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        slice = w / T; tile = w - slice * T;
    }
    set_slice(ep, slice);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int nkt = (K + BK - 1) / BK;
    const int kt0 = slice * kt_per_split;
    const int kt1 = kt0 + kt_per_split < nkt ? kt0 + kt_per_split : nkt;

    SA sa;
    SB sb;
    sa.init(al, m0, wave, lane, kt0 * BK);
    sb.init(bl, n0, wave, lane, kt0 * BK);

    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // (statistics epilogues: the per-channel parameters of this block's columns are requested now, see epi_prefetch)
    constexpr int PPT = EpiShape<BN, NW, EP>::PPT;
    float pre[PPT][4];
    epi_prefetch<BN, NW>(ep, pre, tid, n0);

#ifndef VTX_ABLATE          // measurement builds (tools/ablate_gemm.py) compile the ablation switches in
    abl = 0;                // (bits 4 and 5, the A/B switches of the block order, have been consumed above)
#endif
    // s_waitcnt immediates (gfx9 encoding: vmcnt[3:0]|[15:14], expcnt[6:4], lgkmcnt[11:8]); only vmcnt waits
    constexpr int INFLIGHT = NDMA * (STAGES - 2);       // DMA instructions that may still be pending at a K step
    static_assert(INFLIGHT < 64, "vmcnt is 6 bits");
    constexpr int WAIT_TILES_LEFT = (INFLIGHT & 0xF) | ((INFLIGHT >> 4) << 14) | (0x7 << 4) | (0xF << 8);
    constexpr int WAIT_ALL = 0 | (0x7 << 4) | (0xF << 8);

    // Pipeline: STAGES LDS stages, one s_barrier per K step (BK deep = BK/32 MFMA steps), the DMA of tile
    // kt+STAGES-1 is issued right after the barrier of step kt (into the stage tile kt-1 occupied) and has
    // STAGES-1 MFMA phases to land.
    if (kt0 < kt1) {
#pragma unroll
        for (int p = 0; p < STAGES - 1; ++p) {
            if (kt0 + p < kt1) {
                sa.issue(al, (kt0 + p) * BK, lds + p * TILE, wave);
                sb.issue(bl, (kt0 + p) * BK, lds + p * TILE + BM * BK, wave);
            }
        }
        int stage = 0;                          // stage holding tile kt
        for (int kt = kt0; kt < kt1; ++kt) {
            if (!(abl & 8)) {
                // this wave's pieces of tile kt have landed (later tiles may still be in flight) ...
                if (STAGES > 2 && kt + (STAGES - 2) < kt1) __builtin_amdgcn_s_waitcnt(WAIT_TILES_LEFT);
                else __builtin_amdgcn_s_waitcnt(WAIT_ALL);
                // ... and after the barrier so have everybody's; all waves are also done reading the
                // stage that held tile kt-1, which is the one the next DMA goes into.
                __builtin_amdgcn_s_barrier();
            }
            const bf16_t* cur = lds + stage * TILE;
            if (kt + (STAGES - 1) < kt1 && !(abl & 4)) {
                const int s2 = stage + (STAGES - 1) >= STAGES ? stage - 1 : stage + (STAGES - 1);
                sa.issue(al, (kt + (STAGES - 1)) * BK, lds + s2 * TILE, wave);
                sb.issue(bl, (kt + (STAGES - 1)) * BK, lds + s2 * TILE + BM * BK, wave);
            }
#pragma unroll
            for (int h = 0; h < BK / 32; ++h) {
                bf16x8_t fa[MT], fb[NT];
                static_assert(SA::MC == SB::MC, "operand pairs are both row-major or both k-major");
                if constexpr (SA::MC) {
                    // k-major operands: every transposing read of the step is issued first (B fragments, then the A
                    // fragments in the order the MFMA rows use them); row i of the MFMAs starts as soon as ITS A fragment
                    // has returned (LDS returns in order: all but the 2 (MT-1-i) youngest reads), the later rows' reads
                    // land under the earlier rows' MFMAs -- the ladder hipcc builds by itself for the row-major kernels
                    vtx_v4s_t ra[MT][2], rb[NT][2];
#pragma unroll
                    for (int j = 0; j < NT; ++j) SB::frag_tr(cur + BM * BK, wn * WTN + j * 16, lane, rb[j]);
#pragma unroll
                    for (int i = 0; i < MT; ++i) SA::frag_tr(cur, wm * WTM + i * 16, lane, ra[i]);
                    static_assert(2 * (MT - 1) <= 15, "lgkmcnt ladder");
                    mc_rows<0, MT, NT>(ra, rb, fb, acc);
                    continue;
                } else if (!(abl & 2) || kt == kt0) {
#pragma unroll
                    for (int i = 0; i < MT; ++i) fa[i] = SA::frag(cur, wm * WTM + i * 16, lane, h);
#pragma unroll
                    for (int j = 0; j < NT; ++j) fb[j] = SB::frag(cur + BM * BK, wn * WTN + j * 16, lane, h);
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
            }
            stage = stage + 1 >= STAGES ? 0 : stage + 1;
        }
    }
    tile_epilogue<BM, BN, WM, WN, STAGES * TILE * 2, EP, LEAN>(ep, acc, pre, lds, m0, n0, tile / tiles_n, tile % tiles_n, tid, lane, wave);
}

extern int g_vtx_contraction_generation;   // 2 (default): DMA kernel for bf16; 1: register-staged kernel
extern std::atomic<long> g_vtx_generation_count[4];   // launches per generation (process-wide; vtx_contraction_generation_counts)
extern thread_local int g_vtx_last_generation;  // 1 / 2: which kernel generation this thread's last launch_auto picked
extern thread_local int g_vtx_last_colgroups;   // column groups (tiles_n x waves per tile row) of this thread's last launch: EpiRowLse partials
extern int g_vtx_ablate;   // measurement only: bit0 no MFMA, bit1 no fragment reads, bit2 no DMA, bit3 no barrier

// ------------------------------------------------------------------ host-side launch
template <class T, int BM, int BN, class AL, class BL, class EP>
inline void launch_v1(const AL& al, const BL& bl, const EP& ep, int M, int N, int K, int split_k, hipStream_t st) {
    constexpr int BK = 4 * Elem<T>::VEC;
    const int tiles_m = vtx_cdiv(M, BM), tiles_n = vtx_cdiv(N, BN);
    const int nkt = vtx_cdiv(K, BK);
    if (split_k < 1) split_k = 1;
    if (split_k > nkt) split_k = nkt > 0 ? nkt : 1;
    const int per = vtx_cdiv(nkt, split_k);       // K == 0: no K steps, the epilogue alone runs
    split_k = per > 0 ? vtx_cdiv(nkt, per) : 1;
    dim3 grid(tiles_m * tiles_n, split_k), block(NTHREADS);
    g_vtx_last_colgroups = tiles_n * 2;
    hipLaunchKernelGGL((contraction_kernel<T, BM, BN, AL, BL, EP>), grid, block, 0, st, al, bl, ep, K, tiles_n, per);
}

// ---- optional per-launch timing (vtx_profile_start / vtx_profile_stop in core.hip): one class per kernel
// instantiation, begin/end HIP events attached to every launch, algorithmic FLOPs and bytes summed.
// elements of the operand tensor a loader reads (the tensor itself, not its im2col view)
template <class T, int SL> inline double algo_elems(const PlainKC<T, SL>& l) { return (double)l.rows * l.K; }
template <class T, int SL> inline double algo_elems(const PlainMC<T, SL>& l) { return (double)l.rows * l.K; }
template <class T, int SL> inline double algo_elems(const TapKC<T, SL>& l) { return (double)l.rows * l.K; }
template <class T, int SL> inline double algo_elems(const ConvFwdA<T, SL>& l) { return (double)l.g.N * l.g.H * l.g.W * l.g.C; }
template <class T, int SL> inline double algo_elems(const ConvWgradB<T, SL>& l) { return (double)l.g.N * l.g.H * l.g.W * l.g.C; }
template <class T, int SL> inline double algo_elems(const ConvDgradA<T, SL>& l) { return (double)l.g.N * l.g.OH * l.g.OW * l.g.KO; }
template <class T, int SL> inline double algo_elems(const ConvDgradS2A<T, SL>& l) { return (double)l.g.N * l.g.OH * l.g.OW * l.g.KO; }
template <class EP> inline double epi_bytes(const EP&, double mn, int) { return mn * 4; }
inline double epi_bytes(const EpiRowLse& e, double, int) { return 0.0; }     // the logits are never written
template <class T, int S> inline double epi_bytes(const EpiStore<T, S>& e, double mn, int split_k) {
    return mn * sizeof(T) * (split_k > 1 ? split_k : 1) + (e.residual ? mn * sizeof(T) : 0.0) + (e.preact ? mn * sizeof(T) : 0.0) +
           (e.bn_x ? mn * sizeof(T) : 0.0) + (e.bn_ybits ? mn / 8 : e.bn_y ? mn * sizeof(T) : 0.0);
}

// Outputs of at least VIRTEX_AMD_NT_STORE_MB megabytes (default 200; 0 = never) are stored non-temporally: they do not
// fit the 256 MB Infinity Cache next to their own inputs, so caching them only evicts what the next kernel could hit.
// A/B in one session: 31.37 -> 31.17 ms/step (profiles/r02_ab_buffer_addressing.txt).
inline int vtx_nt_policy(double out_bytes) {
    static const double thr = [] { const char* e = getenv("VIRTEX_AMD_NT_STORE_MB"); return e ? atof(e) * 1e6 : 200e6; }();
    return thr > 0.0 && out_bytes >= thr ? 1 : 0;
}

template <int BM, int BN, int WM, int WN, int BK = 32, int STAGES = 3, class AL, class BL, class EP>
inline int launch_v2(const AL& al, const BL& bl, const EP& ep_in, int M, int N, int K, int split_k, hipStream_t st) {
    const int tiles_m = vtx_cdiv(M, BM), tiles_n = vtx_cdiv(N, BN);
    const int nkt = vtx_cdiv(K, BK);
    if (split_k < 1) split_k = 1;
    if (split_k > nkt) split_k = nkt > 0 ? nkt : 1;
    const int per = vtx_cdiv(nkt, split_k);       // K == 0: no K steps, the epilogue alone runs
    split_k = per > 0 ? vtx_cdiv(nkt, per) : 1;
    constexpr size_t lds_bytes = STAGES * (size_t)(BM + BN) * BK * 2;
    auto kern = contraction_v2_kernel<BM, BN, WM, WN, AL, BL, EP, BK, STAGES, false>;
    if constexpr (EP::STATS && EP::STAGED) {
        // every tile interior, dense tensors: the instantiation whose statistics epilogue carries no bounds / row-map logic
        if (lean_host_ok(ep_in, M, N, BM, BN)) kern = contraction_v2_kernel<BM, BN, WM, WN, AL, BL, EP, BK, STAGES, true>;
    }
    static bool attr_set = false;
    if (lds_bytes > 65536 && !attr_set) {
        hipFuncSetAttribute((const void*)contraction_v2_kernel<BM, BN, WM, WN, AL, BL, EP, BK, STAGES, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if constexpr (EP::STATS && EP::STAGED)
            hipFuncSetAttribute((const void*)contraction_v2_kernel<BM, BN, WM, WN, AL, BL, EP, BK, STAGES, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_set = true;
    }
    dim3 grid(tiles_m * tiles_n, split_k), block(64 * WM * WN);
    g_vtx_last_colgroups = tiles_n * WN;
    EP ep = ep_in;
    if constexpr (EP::STAGED) ep.nt = vtx_nt_policy((double)M * N * sizeof(typename EP::Out));
    bool prof = g_vtx_prof_on != 0;
    if (prof) {
        static const int cls = vtx_prof_register(__PRETTY_FUNCTION__);
        prof = g_vtx_prof_only < 0 || g_vtx_prof_only == cls;     // optionally time one kernel class only
        if (prof) {
            // the launch itself carries the two events: they take the dispatch's own begin / end timestamps (what
            // rocprofv3 reports), not the stream-order times around it -- with three streams sharing the chip a
            // kernel often waits for CUs after its predecessor on the stream has finished
            hipEvent_t e0, e1;
            vtx_prof_events(cls, 2.0 * M * N * K, 2.0 * (algo_elems(al) + algo_elems(bl)) + epi_bytes(ep, (double)M * N, split_k), &e0, &e1);
            hipExtLaunchKernelGGL(kern, grid, block, (uint32_t)lds_bytes, st, e0, e1, 0, al, bl, ep, K, tiles_n, per, g_vtx_ablate);
            return tiles_m;
        }
    }
    hipLaunchKernelGGL(kern, grid, block, lds_bytes, st, al, bl, ep, K, tiles_n, per, g_vtx_ablate);
    return tiles_m;         // number of statistics strips (block rows) this launch produced
}

#include "gemm_v3.h"

// Tile choice.  Score = (tile efficiency) x (wave quantisation of the grid over 256 CUs) x (padding
// efficiency); candidates with BN > what N needs are skipped.
struct TileCand { int bm, bn, resident; float eff; };
extern int g_vtx_tile_override;   // tests: force a candidate (-1 = automatic)
extern int g_vtx_sw_tile64x256;   // vtx_set_switch("tile64x256"): one 64x256 tile for k-major problems with M <= 64, 128 < N <= 256
extern int g_vtx_sw_stats_tile;   // vtx_set_switch("stats_tile"): which statistics epilogues take 8-wave 128x128 tiles on large M
extern int g_vtx_sw_mc_eff128;    // percent: relative efficiency of 128x128 tiles for k-major operands (vtx_set_switch("mc_eff128"))
inline int pick_tile(int M, int N, int splits, bool allow256, bool mc = false) {
    if (g_vtx_tile_override >= 0 && (allow256 || g_vtx_tile_override >= 2)) return g_vtx_tile_override;
    // eff = measured relative throughput on a large NT GEMM (tools/ablate_gemm.py: 256x128 806 TF/s,
    // 128x128 680, 256x256 567 -- one resident block per CU cannot hide its own epilogue)
    static const TileCand cands[6] = {{256, 256, 1, 0.70f}, {256, 128, 2, 1.00f}, {128, 128, 3, 0.84f},
                                      {128, 64, 4, 0.55f}, {64, 128, 4, 0.55f}, {64, 64, 4, 0.35f}};
    // cost model: every CU works through ceil(blocks / 256) tiles (co-resident blocks share the CU, so
    // residency changes overlap, not the amount of work); a tile costs its area over the measured relative
    // throughput, and a CU that only ever holds one block loses the epilogue/main-loop overlap (~10 %).
    // Validated against tools/sweep_tiles.py on every GEMM-shaped op of the step.
    int best = 5; float best_cost = 3.4e38f;
    for (int c = 0; c < 6; ++c) {
        const TileCand& t = cands[c];
        if (!allow256 && t.bm == 256) continue;
        if (t.bn > 64 && N <= 64) continue;
        if (t.bn > 128 && N <= 128) continue;
        if (t.bm > 64 && M <= 64) continue;
        const long tm = vtx_cdiv(M, t.bm), tn = vtx_cdiv(N, t.bn);
        const long blocks = tm * tn * (splits < 1 ? 1 : splits);
        const long rounds = (blocks + 255) / 256;
        const float overlap = (rounds < 2 && t.resident > 1) ? 0.9f : 1.0f;
        // k-major operands (weight gradients): since their LDS-DMA pipeline works (round 3) the 4-wave 128x128 tile,
        // three blocks per CU, is the faster one per flop (tools/sweep_tiles.py on the repaired kernels)
        const float eff = (mc && c == 2) ? 0.01f * (float)g_vtx_sw_mc_eff128 : t.eff;
        const float cost = (float)rounds * (float)(t.bm * t.bn) / (eff * overlap);
        if (cost < best_cost) { best_cost = cost; best = c; }
    }
    return best;
}

// Generation 3 or not: 0 = no, 1 = 256x256 blocks, 2 = 256x128 blocks.  Tile override 20 / 21 forces them (tests, sweeps).
// Cost model from the per-wave time stamps of profiles/r04_gen3_stamps.txt (shader-clock cycles, one block per CU, so a
// launch takes ceil(tiles / 256) rounds of  prologue + K tiles x cycles per K tile + epilogue ):
//                         prologue   per 64-deep K tile            epilogue: plain / statistics / fused BatchNorm backward
//   256x256               4 400      2 812 (gather loaders 3 700)   6 600 / 11 300 / 24 300
//   256x128               4 400      1 819 (gather loaders 2 450)   4 000 /  7 100 / 13 400
// and taken when its predicted rate beats what generation 2 reaches on the class (850 TFLOP/s plain matrices, 950 gathers).
extern int g_vtx_sw_gen3;
extern int g_vtx_sw_gen3_s2;      // vtx_set_switch("gen3_s2"): 1 = the picker may take generation 3 for stride-2 input gradients (A/B; default 0)
inline int pick_gen3(int M, int N, int K, int splits, bool gather, int smode) {
    if (g_vtx_tile_override == 20) return 1;
    if (g_vtx_tile_override == 21) return 2;
    if (g_vtx_tile_override >= 0 || !g_vtx_sw_gen3) return 0;
    if (K < 256 || M < 1024 || N < 128 || splits > 1) return 0;
    const double nkt = (double)vtx_cdiv(K, 64);
    const double epi256 = smode == STATS_NONE ? 6600.0 : smode == STATS_FWD ? 11300.0 : 24300.0;
    const double epi128 = smode == STATS_NONE ? 4000.0 : smode == STATS_FWD ? 7100.0 : 13400.0;
    const long t256 = (long)vtx_cdiv(M, 256) * vtx_cdiv(N, 256), t128 = (long)vtx_cdiv(M, 256) * vtx_cdiv(N, 128);
    const double c256 = N > 128 ? (double)((t256 + 255) / 256) * (4400.0 + nkt * (gather ? 3700.0 : 2812.0) + epi256) : 1e30;
    const double c128 = (double)((t128 + 255) / 256) * (4400.0 + nkt * (gather ? 2450.0 : 1819.0) + epi128);
    const double cyc = c256 <= c128 ? c256 : c128;
    const double rate = 2.0 * M * N * (double)K / (cyc / 2.1e9);                  // FLOP/s at the ~2.1 GHz the stamps were taken at
    const double need = (gather ? 950e12 : 850e12) * (g_vtx_sw_gen3 >= 2 ? 0.01 * g_vtx_sw_gen3 : 1.0);   // gen3 = 2..: threshold in percent (sweeps)
    if (rate < need) return 0;
    return c256 <= c128 ? 1 : 2;
}

// Generation 3 for the weight gradients (k-major pairs, gemm_v3mc.h): tile (1 = 256x256, 2 = 256x128, 0 = generation 2) and
// the split-K slice count that goes with it -- enough slices to give every CU a block, at least 4 K tiles of 64 per slice.
// The callers' split policy (vtx_pick_split_k) asks first, so that launch_auto sees the slice count this plan assumes.
// Measured per shape at bs = 256 (profiles/r04_gen3_mc_per_shape.txt, reductions included): text-head gradients +6...23 %,
// 1x1 convolutions at 14x14 / 7x7 +12...19 % (256x128 blocks), 3x3 at 14x14 / 7x7 +25...34 % (256x256 blocks); the
// HBM-bound gradients of the 56x56 / 28x28 stages (M N / (M + N) below ~200 FLOP per operand byte pair) tie or lose
// and stay on generation 2.  Tile and slices by a cycle model fitted to the same table:
//   256x256: 4 400 + slices' K tiles x 3 000 (gather 3 600) + 13 000 (fp32 tile through the strip); 256x128: ... x 1 800 (2 450) + 6 000;
//   + the reduce launch: 6 000 + (slices + 2) x M N x 4 bytes at 4 TB/s.
extern int g_vtx_sw_gen3_mc;      // vtx_set_switch("gen3_mc"): 0 = only when forced (tile override 20 / 21), n = the intensity threshold (default 800)
inline int plan_gen3_mc(int M, int N, int K, bool gather, int* split) {
    *split = 1;
    if (g_vtx_tile_override >= 0 && g_vtx_tile_override != 20 && g_vtx_tile_override != 21) return 0;
    const bool forced = g_vtx_tile_override == 20 || g_vtx_tile_override == 21;
    if (!forced) {
        if (!g_vtx_sw_gen3_mc || K < 2048 || M < 256 || N < 256) return 0;
        if ((double)M * N / ((double)M + N) < (double)g_vtx_sw_gen3_mc) return 0;
    }
    const int nkt = vtx_cdiv(K, 64);
    int best = 0, best_s = 1; double best_c = 1e30;
    for (int c = 1; c <= 2; ++c) {
        if (forced && c != g_vtx_tile_override - 19) continue;
        const long t = (long)vtx_cdiv(M, 256) * vtx_cdiv(N, c == 1 ? 256 : 128);
        long s = t >= 256 ? 1 : 256 / t;
        if (s > nkt / 4) s = nkt / 4 > 0 ? nkt / 4 : 1;
        const int per = vtx_cdiv(nkt, (int)s);
        s = vtx_cdiv(nkt, per);
        const double ck = c == 1 ? (gather ? 3600.0 : 3000.0) : (gather ? 2450.0 : 1800.0);
        const double epi = c == 1 ? 13000.0 : 6000.0;
        double cyc = (double)((t * s + 255) / 256) * (4400.0 + per * ck + epi);
        if (s > 1) cyc += 6000.0 + (double)(s + 2) * M * N * 4.0 / 4.0e12 * 2.1e9;
        if (cyc < best_c) { best_c = cyc; best = c; best_s = (int)s; }
    }
    *split = best_s;
    return best;
}

// Returns the number of BatchNorm-statistics strips written (0 when the kernel generation that ran
// does not produce them: the caller then falls back to the stand-alone reduction).
template <class T, template <class, int> class ALT, template <class, int> class BLT, class EP, class FA, class FB>
inline int launch_auto(FA make_a, FB make_b, const EP& ep, int M, int N, int K, int split_k, hipStream_t st) {
    constexpr bool BF = sizeof(T) == 2;
    // the DMA kernel addresses its operands through buffer descriptors: both must qualify (size, channel multiples)
    auto buf_ok = [&](int bk) { ALT<T, 1> a; make_a(a); BLT<T, 1> b; make_b(b); return a.buf_ok(bk) && b.buf_ok(bk); };
    const bool v2 = BF && g_vtx_contraction_generation >= 2 && buf_ok(32);
    int c = pick_tile(M, N, split_k, v2, ALT<T, 1>::MC);
    if constexpr (EP::STATS) {
        // The HBM-bound convolutions of stages 1-2 with a BatchNorm epilogue (K = 64...512: a handful of K steps, then
        // an epilogue that reads up to three more [M][N] tensors) are chains of memory latencies, not MFMA work: on
        // 128x128 tiles with EIGHT waves (wave tile 32x64) the epilogue chain is half as long and three blocks fit a
        // CU.  Measured per layer at bs=256 (tools/bench_1x1.py: tensors of 0.25-2.2 GB, not cache-fed): input gradient
        // + fused BatchNorm backward 320 -> 275 us (64->256 @56), 157 -> 146 (128->512 @28); 14x14 layers lose 10 %.
        const int rule = g_vtx_sw_stats_tile;
        const bool stats_tile = rule == 1 || (rule == 2 && EP::SMODE == STATS_FWD) || (rule == 3 && EP::SMODE == STATS_BWD);
        if (stats_tile && v2 && c == 1 && g_vtx_tile_override < 0 && M >= 100000) c = 6;
        // Round 4, after the lean epilogues: the 1x1 input gradients with the fused BatchNorm backward are faster on FOUR-wave
        // 128x128 tiles (48 KiB of stages: three blocks per CU) at every image size the generation-3 picker leaves to this
        // kernel -- residual joins (mask bits + identity gradient, tools/bench_join_tiles.py) 301 -> 252 us (K 64, N 256 @56),
        // 173 -> 145 (128, 512 @28), 112 -> 94 (256, 1024 @14), 62 -> 55 (512, 2048 @7); mask recomputed from x 125 -> 87
        // (128 -> 512 @28), 86 -> 75 (512 -> 128 @28) (profiles/r04_join_tiles.txt).  vtx_set_switch("stats_tile", 4); 0 = off.
        // (4 + bit 0: the forward statistics class too; 4 + bit 1: every A loader, not only plain rows -- sweeps)
        if (rule >= 4 && v2 && c == 1 && g_vtx_tile_override < 0) {
            const bool plain = std::is_same<ALT<T, 1>, PlainKC<T, 1>>::value;
            const bool bwd = EP::SMODE == STATS_BWD, fwd = EP::SMODE == STATS_FWD && ((rule - 4) & 1);
            if ((bwd || fwd) && (plain || ((rule - 4) & 2))) c = 2;
        }
    }
    if constexpr (BF && ALT<T, 1>::MC) {
        // Weight gradients of layers with at most 64 output channels and 129...256 rows of taps x channels (the stem: 64 x 224
        // over K = 3.2 M pixels): ONE 64 x 256 tile instead of two 64 x 128 ones -- the gradient operand dy, 5x the size of the
        // image operand, is then read once per K slice instead of twice (the kernel is HBM-bound: 822 -> 411 MB of dy).
        if (v2 && g_vtx_sw_tile64x256 && g_vtx_tile_override < 0 && M <= 64 && N > 128 && N <= 256) c = 7;
    }
#define VTX_V1(BM_, BN_, SA_, SB_)                                                          \
    { ALT<T, SA_> a; make_a(a); BLT<T, SB_> b; make_b(b); launch_v1<T, BM_, BN_>(a, b, ep, M, N, K, split_k, st); }
#define VTX_V2(BM_, BN_, WM_, WN_, SA_, SB_)                                                \
    { ALT<T, SA_> a; make_a(a); BLT<T, SB_> b; make_b(b); strips = launch_v2<BM_, BN_, WM_, WN_>(a, b, ep, M, N, K, split_k, st); }
    int strips = 0;
    if constexpr (BF && ALT<T, 1>::MC && BLT<T, 1>::MC && !EP::STATS && EP::STAGED) {
        // generation 3, k-major pairs (the weight gradients): only with the slice count its plan assumes
        int s3 = 1;
        const int g3 = v2 ? plan_gen3_mc(M, N, K, !std::is_same<BLT<T, 1>, PlainMC<T, 1>>::value, &s3) : 0;
        const bool forced = g_vtx_tile_override == 20 || g_vtx_tile_override == 21;
        // (the caller sized its workspace epilogue / reduction for split_k slices: the 64-deep K tiles must split into exactly that many)
        const int sk = split_k < 1 ? 1 : split_k, nkt64 = vtx_cdiv(K, 64);
        const bool slices_ok = sk == 1 || (sk <= nkt64 && vtx_cdiv(nkt64, vtx_cdiv(nkt64, sk)) == sk);
        if (g3 && slices_ok && (forced || s3 == sk) && buf_ok(64)) {
            g_vtx_last_generation = 3;
            g_vtx_generation_count[3].fetch_add(1, std::memory_order_relaxed);
            if (g3 == 1) { ALT<T, 4> a; make_a(a); BLT<T, 4> b; make_b(b); launch_v3<256>(a, b, ep, M, N, K, split_k, st); }
            else { ALT<T, 4> a; make_a(a); BLT<T, 2> b; make_b(b); launch_v3<128>(a, b, ep, M, N, K, split_k, st); }
            return 0;
        }
    }
    if constexpr (BF && !ALT<T, 1>::MC && !BLT<T, 1>::MC) {
        // generation 3 (gemm_v3.h): 8-wave 256x256 / 256x128 blocks with the phase-interleaved K loop
        // (the parity classes of the stride-2 input gradients lose on generation 3 at every shape of the step -- 107 vs 142-148 us
        //  256->256 k3 @28, 142 vs 182-188 512->1024 k1 @28, 98 vs 116-119 1024->2048 k1 @14: profiles/r04_gen3_per_shape_lean_epilogue.txt
        //  -- while the cycle model, fitted to the stride-1 gathers, predicts a tie: they stay on generation 2 unless forced)
        const bool s2 = std::is_same<ALT<T, 1>, ConvDgradS2A<T, 1>>::value && g_vtx_tile_override < 0 && !g_vtx_sw_gen3_s2;
        const int g3 = v2 && !s2 ? pick_gen3(M, N, K, split_k, !std::is_same<ALT<T, 1>, PlainKC<T, 1>>::value, EP::SMODE) : 0;
        if (g3 && buf_ok(64)) {
            g_vtx_last_generation = 3;
            g_vtx_generation_count[3].fetch_add(1, std::memory_order_relaxed);
            if (g3 == 1) { ALT<T, 4> a; make_a(a); BLT<T, 4> b; make_b(b); strips = launch_v3<256>(a, b, ep, M, N, K, split_k, st); }
            else { ALT<T, 4> a; make_a(a); BLT<T, 2> b; make_b(b); strips = launch_v3<128>(a, b, ep, M, N, K, split_k, st); }
            return EP::STATS ? strips : 0;
        }
    }
#define VTX_V2X(BM_, BN_, WM_, WN_, BK_, ST_, SA_, SB_)                                    \
    { ALT<T, SA_> a; make_a(a); BLT<T, SB_> b; make_b(b); strips = launch_v2<BM_, BN_, WM_, WN_, BK_, ST_>(a, b, ep, M, N, K, split_k, st); }
    g_vtx_last_generation = v2 ? 2 : 1;
    g_vtx_generation_count[v2 ? 2 : 1].fetch_add(1, std::memory_order_relaxed);
    if constexpr (BF && !ALT<T, 1>::MC && !BLT<T, 1>::MC && !EP::STATS) {
        // Row-major operands, small grids: when 256x128 tiles would not even give every CU one block, LDS
        // capacity is free, so stage 64-deep K steps -- every LDS-DMA row is a whole 128-byte line (measured
        // 30 vs 22.6 TB/s L2->LDS, tools/probes/dma_probe.hip) and there are half as many barriers.
        // tools/sweep_tiles.py: +9...19 % on exactly these shapes (ffn2 fwd, ffn1/in_proj/vocab dgrad, the
        // 7x7 stage), -10...40 % on larger grids where two co-resident 256x128 blocks overlap instead.
        const long t256 = (long)vtx_cdiv(M, 256) * vtx_cdiv(N, 128) * (split_k < 1 ? 1 : split_k);
        const bool small_grid = g_vtx_tile_override < 0 && t256 <= 256 && K >= 256 && N > 64 && M > 128;
        if (v2 && (small_grid || g_vtx_tile_override >= 10) && buf_ok(64)) {
            // (256x256 with 4 stages / 64-deep steps and 256x128 with 64-deep x 3 stages were measured and removed:
            //  profiles/r02_tile_variants_256x256.txt)
            if (g_vtx_tile_override == 11) VTX_V2X(256, 128, 4, 2, 64, 2, 4, 2)     //  96 KiB LDS: one block per CU
            else VTX_V2X(128, 128, 2, 2, 64, 2, 4, 4)                               //  64 KiB: two blocks per CU
            return 0;
        }
    }
    if constexpr (BF) {
        if (v2) {
            switch (c) {
                case 0: VTX_V2(256, 256, 2, 4, 2, 2) break;
                case 1: VTX_V2(256, 128, 4, 2, 2, 1) break;
                case 2: VTX_V2(128, 128, 2, 2, 2, 2) break;
                case 3: VTX_V2(128, 64, 2, 2, 2, 1) break;
                case 4: VTX_V2(64, 128, 2, 2, 1, 2) break;
                case 6: VTX_V2(128, 128, 4, 2, 1, 1) break;     // 8 waves on 128x128: wave tile 32x64 (short epilogue chains)
                case 7:                                          // k-major operands only (see above)
                    if constexpr (ALT<T, 1>::MC) { VTX_V2(64, 256, 1, 4, 1, 4) break; }
                    else { VTX_V2(64, 64, 2, 2, 1, 1) break; }
                default: VTX_V2(64, 64, 2, 2, 1, 1) break;
            }
            return EP::STATS ? strips : 0;
        }
    }
    switch (c) {
        case 2: VTX_V1(128, 128, 2, 2) break;
        case 3: VTX_V1(128, 64, 2, 1) break;
        case 4: VTX_V1(64, 128, 1, 2) break;
        default: VTX_V1(64, 64, 1, 1) break;
    }
#undef VTX_V1
#undef VTX_V2
#undef VTX_V2X
    return 0;
}

}  // namespace vtxg
