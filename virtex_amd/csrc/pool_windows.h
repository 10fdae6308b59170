// MaxPool2d(kernel 3, stride 2, padding 1) on NHWC activations: the memory-access skeletons shared by pool.hip and the
// stem-tail kernels of bn.hip.
//
// Both directions visit a small set of candidate vectors per output: the forward its 3 x 3 taps, the backward (gather
// form) the <= 2 x 2 pooling windows that contain an input pixel.  Written as `if (candidate exists) { load; use; }`
// hipcc puts an s_waitcnt vmcnt(0) inside every branch: a chain of up to nine (forward) / eight (backward: gradient +
// argmax per window) memory latencies per thread and trip -- the stand-alone backward ran at 2.4 TB/s, the fused
// forward tail at 3.7.  Here EVERY candidate is requested first, from a clamped (always valid) address, and the ones
// that do not exist are masked when the values are used: one latency per trip.  The order in which candidates are
// consumed is the order of the 9-tap loops of rounds 1-2, so results (first maximum on ties, fp32 summation order)
// are bit-identical.
#pragma once
#include "vtx_common.h"

// ---- forward: the 3 x 3 taps of output pixel (oh, ow) of image n, channel vector c0
template <class T> struct PoolTaps {
    uint4 v[9];
    uint32_t valid;      // bit kh*3+kw
    __device__ __forceinline__ void request(const T* __restrict__ x, int n, int oh, int ow, int c0, int H, int W, int C) {
        valid = 0u;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ih = oh * 2 - 1 + kh, iw = ow * 2 - 1 + kw;
                const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                valid |= ok ? 1u << (kh * 3 + kw) : 0u;
                const int ihc = ih < 0 ? 0 : (ih >= H ? H - 1 : ih), iwc = iw < 0 ? 0 : (iw >= W ? W - 1 : iw);
                v[kh * 3 + kw] = *reinterpret_cast<const uint4*>(x + (((long)n * H + ihc) * W + iwc) * C + c0);
            }
        vtx_loads_issued();
    }
};

// ---- backward: the pooling windows (oh, ow) whose 3 x 3 footprint contains input pixel (ih, iw).  th = ih + 1 - kh must be
// even, so an odd ih lies in two window rows (taps kh = 0 and 2), an even one in one (kh = 1); columns alike.
template <class T> struct PoolWindows {
    static constexpr int VEC = 16 / (int)sizeof(T);
    uint4 d[4];
    uint32_t am[4][2];   // the VEC argmax bytes of each window
    uint32_t tap[4];     // kh*3+kw of this pixel inside window q; 0xffffffff: the window does not exist
    __device__ __forceinline__ void request(const T* __restrict__ dpool, const uint8_t* __restrict__ argmax, int n, int ih, int iw,
                                            int c0, int C, int OH, int OW) {
        const int kh[2] = {(ih & 1) ? 0 : 1, 2}, kw[2] = {(iw & 1) ? 0 : 1, 2};
        int rh[2], rw[2]; bool vh[2], vw[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int th = ih + 1 - kh[a], tw = iw + 1 - kw[a];
            vh[a] = (a == 0 || (ih & 1)) && th >= 0 && (th >> 1) < OH;
            vw[a] = (a == 0 || (iw & 1)) && tw >= 0 && (tw >> 1) < OW;
            rh[a] = vh[a] ? th >> 1 : 0;
            rw[a] = vw[a] ? tw >> 1 : 0;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int q = a * 2 + b;
                const long off = (((long)n * OH + rh[a]) * OW + rw[b]) * C + c0;
                d[q] = *reinterpret_cast<const uint4*>(dpool + off);
                if constexpr (VEC == 8) { const uint2 t = *reinterpret_cast<const uint2*>(argmax + off); am[q][0] = t.x; am[q][1] = t.y; }
                else { am[q][0] = *reinterpret_cast<const uint32_t*>(argmax + off); am[q][1] = 0u; }
                tap[q] = (vh[a] && vw[b]) ? (uint32_t)(kh[a] * 3 + kw[b]) : 0xffffffffu;
            }
        vtx_loads_issued();
    }
    // g[j] = sum over the windows whose argmax is this pixel (ascending (kh, kw): the order of the 9-tap loop)
    __device__ __forceinline__ void gather(float* g) const {
#pragma unroll
        for (int j = 0; j < VEC; ++j) g[j] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float f[VEC];
            vtx_unpack_raw16<T>(d[q], f);
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                if (((am[q][j >> 2] >> (8 * (j & 3))) & 0xffu) == tap[q]) g[j] += f[j];
        }
    }
};

// ---- backward, four input pixels at a time: the 2 x 2 quad (2a .. 2a+1, 2b .. 2b+1) lies in exactly the four windows
// (a .. a+1, b .. b+1), so a thread that owns a quad requests 4 window vectors (+ argmax) for 4 pixel vectors instead of
// 1 + 2 + 2 + 4 = 9, and has four pixel vectors in flight per trip.  Per pixel the windows are consumed in ascending tap
// order, as everywhere else: pixel 0 = (even, even): window 0 tap 4; pixel 1 = (even, odd): window 1 tap 3, window 0 tap 5;
// pixel 2 = (odd, even): window 2 tap 1, window 0 tap 7; pixel 3 = (odd, odd): windows 3, 2, 1, 0 at taps 0, 2, 6, 8.
template <class T> struct PoolQuad {
    static constexpr int VEC = 16 / (int)sizeof(T);
    uint4 x[4];          // pixels (2a, 2b), (2a, 2b+1), (2a+1, 2b), (2a+1, 2b+1)
    uint4 d[4];          // windows (a, b), (a, b+1), (a+1, b), (a+1, b+1)
    uint32_t am[4][2];
    uint32_t pvalid;     // bit k: pixel k is inside the image
    uint32_t wvalid;     // bit q: window q exists
    long xoff[4];        // element offsets of the four pixel vectors (clamped for pixels outside)
    __device__ __forceinline__ void request(const T* __restrict__ xin, const T* __restrict__ dpool, const uint8_t* __restrict__ argmax,
                                            int n, int a, int b, int c0, int H, int W, int C, int OH, int OW) {
        const bool h1 = 2 * a + 1 < H, w1 = 2 * b + 1 < W, oh1 = a + 1 < OH, ow1 = b + 1 < OW;
        pvalid = 1u | (w1 ? 2u : 0u) | (h1 ? 4u : 0u) | ((h1 && w1) ? 8u : 0u);
        wvalid = 1u | (ow1 ? 2u : 0u) | (oh1 ? 4u : 0u) | ((oh1 && ow1) ? 8u : 0u);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ih = 2 * a + ((k >> 1) && h1 ? 1 : 0), iw = 2 * b + ((k & 1) && w1 ? 1 : 0);
            xoff[k] = (((long)n * H + ih) * W + iw) * C + c0;
            x[k] = *reinterpret_cast<const uint4*>(xin + xoff[k]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int oh = a + ((q >> 1) && oh1 ? 1 : 0), ow = b + ((q & 1) && ow1 ? 1 : 0);
            const long off = (((long)n * OH + oh) * OW + ow) * C + c0;
            d[q] = *reinterpret_cast<const uint4*>(dpool + off);
            if constexpr (VEC == 8) { const uint2 t = *reinterpret_cast<const uint2*>(argmax + off); am[q][0] = t.x; am[q][1] = t.y; }
            else { am[q][0] = *reinterpret_cast<const uint32_t*>(argmax + off); am[q][1] = 0u; }
        }
        vtx_loads_issued();
    }
    __device__ __forceinline__ void add(int q, uint32_t tap, float* g) const {
        const uint32_t t = (wvalid >> q) & 1u ? tap : 0xffffffffu;
        float f[VEC];
        vtx_unpack_raw16<T>(d[q], f);
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            if (((am[q][j >> 2] >> (8 * (j & 3))) & 0xffu) == t) g[j] += f[j];
    }
    // gradient of pixel K gathered from its windows
    template <int K> __device__ __forceinline__ void gather(float* g) const {
#pragma unroll
        for (int j = 0; j < VEC; ++j) g[j] = 0.f;
        if constexpr (K == 0) add(0, 4u, g);
        else if constexpr (K == 1) { add(1, 3u, g); add(0, 5u, g); }
        else if constexpr (K == 2) { add(2, 1u, g); add(0, 7u, g); }
        else { add(3, 0u, g); add(2, 2u, g); add(1, 6u, g); add(0, 8u, g); }
    }
};
