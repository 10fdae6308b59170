// Training-mode BatchNorm2d on NHWC activations, fused with ReLU and the residual add
// (HBM-bound).  x is viewed as [P = N*H*W][C]; statistics are fp32.
//
//   forward : stats (sum, sum of squares per channel) -> finalize (mean, rstd, running stats,
//             per-channel scale/shift) -> apply  y = act(x*scale + shift (+ residual))
//   backward: dz = dy * (y > 0)   [y = the post-ReLU tensor, nullptr when no ReLU follows]
//             reduce  s1 = sum dz, s2 = sum dz*xhat  -> finalize (dgamma += s2, dbeta += s1)
//             apply   dx = gamma*rstd * (dz - s1/P - xhat*s2/P) ; optionally also emit dz
//
// Replaces aten::batch_norm / relu_ / add_ (+ their backward) of torchvision's Bottleneck,
// reached from /root/reference/virtex/modules/visual_backbones.py:68-74 (eps 1e-5,
// momentum 0.1, unbiased running variance).
#include <stdlib.h>
#include <type_traits>

#include "vtx_common.h"
#include "pool_windows.h"

extern int g_vtx_sw_bn_red_adj;      // vtx_set_switch("bn_red_adj"): stand-alone BatchNorm reductions walk the tensor in interleaved trips
extern int g_vtx_sw_pool_xcd;        // vtx_set_switch("pool_xcd"): XCD-major block order of the stem's pooling tails
extern int g_vtx_sw_bn_adj, g_vtx_sw_bn_grid;   // vtx_set_switch("bn_adj" / "bn_grid"): form and grid cap of the flat apply kernels
extern int g_vtx_sw_bn_fin_wide;     // vtx_set_switch("bn_fin_wide"): 1024-thread finalize / compaction blocks (default off)

namespace {

// x viewed as [P][cv] 16-byte vectors.  Block: TX threads across channel vectors, TY = 256/TX
// pixel rows; blockIdx.y selects the channel-vector group, blockIdx.x a strip of pixels.
template <class T, bool BWD>
__global__ __launch_bounds__(256) void bn_reduce_kernel(
    const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ ymask,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ sums, int P, int C, int TX, int rows_per_block) {
    constexpr int VEC = Elem<T>::VEC;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX, TY = 256 / TX;
    const int c0 = (blockIdx.y * TX + tx) * VEC;
    const int p0 = blockIdx.x * rows_per_block;
    const int p1 = p0 + rows_per_block < P ? p0 + rows_per_block : P;
    float a[VEC], b[VEC], mu[VEC], rs[VEC], ga[VEC], be[VEC];
    const bool remask = BWD && beta != nullptr;   // ReLU mask recomputed from x: y = xhat*gamma + beta > 0
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        ga[j] = remask ? gamma[c0 + j] : 0.f;
        be[j] = remask ? beta[c0 + j] : 0.f;
        a[j] = b[j] = 0.f;
        // forward: shift by the channel's first sample (shifted-data variance: no catastrophic
        // cancellation in E[x^2]-E[x]^2 when |mean| >> std); backward: the saved statistics
        mu[j] = BWD ? mean[c0 + j] : Elem<T>::ld(x + c0 + j);
        rs[j] = BWD ? rstd[c0 + j] : 0.f;
    }
    // UNR rows per trip, all loads issued before any use: one 16-byte load per wave in flight cannot cover
    // the HBM latency with <= 512 blocks on 256 CUs (measured 2.9 TB/s before)
    constexpr int UNR = 4;
    // Which rows a block sums.  rows_per_block > 0: one contiguous chunk per block (rounds 1-3).  rows_per_block == 0 (round 4,
    // vtx_set_switch("bn_red_adj", 1)): trips of UNR * TY rows INTERLEAVED over the blocks -- at any moment the blocks of the grid
    // read neighbouring addresses instead of gridDim.x places 0.4-1.6 MB apart, the access shape that took the flat apply kernels
    // from 4.4 to 5.7 TB/s (tools/probes/stream_probe.hip).  A thread's channel vector does not depend on the row, so only the
    // order of the fp32 additions changes.  Measured NEUTRAL here (the reductions read, they do not write: default off).
    const bool adj = rows_per_block == 0;
    const int pbeg = adj ? blockIdx.x * UNR * TY : p0, pend = adj ? P : p1, pstep = adj ? gridDim.x * UNR * TY : UNR * TY;
    for (int pb = pbeg + ty; pb < pend; pb += pstep) {
        Vec16<T> xv[UNR], g[UNR], m[UNR];
        bool ok[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int p = pb + u * TY;
            ok[u] = p < pend;
            const size_t off = (size_t)(ok[u] ? p : pbeg) * C + c0;
            xv[u].load(x + off);
            if (BWD) {
                g[u].load(dy + off);
                if (ymask) m[u].load(ymask + off);
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (!ok[u]) continue;
            if (BWD) {
                if (ymask) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) g[u].v[j] = m[u].v[j] > 0.f ? g[u].v[j] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float xh = (xv[u].v[j] - mu[j]) * rs[j];
                    if (remask) g[u].v[j] = xh * ga[j] + be[j] > 0.f ? g[u].v[j] : 0.f;
                    a[j] += g[u].v[j]; b[j] += g[u].v[j] * xh;
                }
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) { const float d = xv[u].v[j] - mu[j]; a[j] += d; b[j] += d * d; }
            }
        }
    }
    __shared__ float red[2][256 * VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        red[0][(ty * TX + tx) * VEC + j] = a[j];
        red[1][(ty * TX + tx) * VEC + j] = b[j];
    }
    __syncthreads();
    if (ty == 0) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float s0 = 0.f, s1 = 0.f;
            for (int r = 0; r < TY; ++r) { s0 += red[0][(r * TX + tx) * VEC + j]; s1 += red[1][(r * TX + tx) * VEC + j]; }
            // one partial per (pixel strip, channel): no atomics (thousands of blocks hitting the
            // same 2*C addresses serialised in L2), summed by the finalize kernel
            sums[(size_t)blockIdx.x * 2 * C + c0 + j] = s0;
            sums[(size_t)blockIdx.x * 2 * C + C + c0 + j] = s1;
        }
    }
}


// Sums the per-strip partials of FIN_CH channels with FIN_GR thread groups (block = 256 threads); returns the
// totals to the group-0 threads.  Latency-bound (the partials sit in the Infinity Cache, ~1 us away): many
// short chains with four independent loads per trip, not a few long ones (22 us -> see DESIGN.md 6).
constexpr int FIN_CH = 16, FIN_GR_MAX = 64;
// Thread groups per block = blockDim.x / FIN_CH: 16, or 64 with the measurement switch "bn_fin_wide".  Measured in round 3
// (profiles/r03_kernel_stats_*): the 106 finalize launches per step take 4.8-5.0 us each with either block size -- they
// are launch / first-load latency, not the chain of trips -- so the default stays at 256 threads (and at the summation
// order the calibrated fp32 gradient bounds of tests/test_fidelity.py were measured with).
static inline dim3 fin_block(int nparts) { return dim3((nparts > 64 && g_vtx_sw_bn_fin_wide) ? FIN_CH * FIN_GR_MAX : 256); }
__device__ __forceinline__ void sum_partials(const float* __restrict__ sums, int C, int nparts, int c, int grp,
                                             float& t0, float& t1) {
    __shared__ float red[2][FIN_GR_MAX][FIN_CH];
    const int NG = blockDim.x / FIN_CH;
    float a = 0.f, b = 0.f;
    if (c < C) {
        int p = grp;
        for (; p + 3 * NG < nparts; p += 4 * NG) {
            const float* q = sums + (size_t)p * 2 * C + c;
            const size_t st = (size_t)NG * 2 * C;
            const float a0 = q[0], a1 = q[st], a2 = q[2 * st], a3 = q[3 * st];
            const float b0 = q[C], b1 = q[st + C], b2 = q[2 * st + C], b3 = q[3 * st + C];
            a += (a0 + a1) + (a2 + a3); b += (b0 + b1) + (b2 + b3);
        }
        for (; p < nparts; p += NG) { a += sums[(size_t)p * 2 * C + c]; b += sums[(size_t)p * 2 * C + C + c]; }
    }
    red[0][grp][threadIdx.x % FIN_CH] = a; red[1][grp][threadIdx.x % FIN_CH] = b;
    __syncthreads();
    t0 = t1 = 0.f;
    if (grp == 0) {
        for (int g = 0; g < NG; ++g) { t0 += red[0][g][threadIdx.x % FIN_CH]; t1 += red[1][g][threadIdx.x % FIN_CH]; }
    }
}

// Folds many statistics strips into few: block (x = 32-channel group, y = chunk of `per` strips) writes
// one compact strip.  Used when the convolution epilogue produced thousands of strips (M = 800k rows).
// blockDim.x / 32 thread groups (32 with the 1024-thread blocks the host launches) share a chunk's strips.
__global__ void bn_compact_parts_kernel(const float* __restrict__ sums, float* __restrict__ out, int C, int nparts,
                                        int per) {
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5, NG = blockDim.x >> 5;
    const int p0 = blockIdx.y * per, p1 = p0 + per < nparts ? p0 + per : nparts;
    __shared__ float red[2][32][32];
    float a = 0.f, b = 0.f;
    if (c < C)
        for (int p = p0 + grp; p < p1; p += NG) { a += sums[(size_t)p * 2 * C + c]; b += sums[(size_t)p * 2 * C + C + c]; }
    red[0][grp][threadIdx.x & 31] = a; red[1][grp][threadIdx.x & 31] = b;
    __syncthreads();
    if (grp == 0 && c < C) {
        float t0 = 0.f, t1 = 0.f;
        for (int g = 0; g < NG; ++g) { t0 += red[0][g][threadIdx.x & 31]; t1 += red[1][g][threadIdx.x & 31]; }
        out[(size_t)blockIdx.y * 2 * C + c] = t0;
        out[(size_t)blockIdx.y * 2 * C + C + c] = t1;
    }
}

// Compaction AND finalize in ONE launch (round 4 experiment, OFF by default: measured 0.26 ms/step slower than the two
// dependent 5-us launches it replaces, 38 per step, on the critical path of every BatchNorm of stages 1-2): block (x = 32-channel group, y = chunk of `per` strips) folds its chunk into compact strip y as
// bn_compact_parts_kernel does, publishes it, and draws a ticket; the block that draws the last ticket of its channel group
// folds the ny compact strips and runs the finalize arithmetic of bn_fwd_finalize_kernel / bn_bwd_finalize_kernel for those
// 32 channels.  Hand-off = the write-through form of cdna_hip_programming.md (section 5, in-launch split-K reduction): sc1
// slab stores (relaxed agent-scope atomic stores), every wave vmcnt(0), block barrier, ONE lane: relaxed agent-scope
// fetch_add; the last arriver reads the slabs with sc1 loads.  Correct for any placement of the blocks on XCDs.  tickets[x] is zero when
// the launch starts (zero-initialised workspace header) and is put back to zero by the last arriver.
struct BnFin2Args {
    // forward
    const float* pre_shift; const float* gamma; const float* beta; float* mean; float* rstd; float* scale; float* shift;
    float* running_mean; float* running_var; long long* nbt; float eps, momentum;
    // backward
    const float* rstd_in; float* coef; float* dgamma; float* dbeta;
    int P;
};
template <bool BWD>
__global__ __launch_bounds__(256) void bn_fin2_kernel(const float* __restrict__ sums, float* __restrict__ compact, int* __restrict__ tickets,
                                                      int C, int nparts, int per, BnFin2Args a) {
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5, NG = blockDim.x >> 5;
    const int ny = gridDim.y;
    const int p0 = blockIdx.y * per, p1 = p0 + per < nparts ? p0 + per : nparts;
    __shared__ float red[2][8][32];
    __shared__ int s_last;
    float s0 = 0.f, s1 = 0.f;
    if (c < C)
        for (int p = p0 + grp; p < p1; p += NG) { s0 += sums[(size_t)p * 2 * C + c]; s1 += sums[(size_t)p * 2 * C + C + c]; }
    red[0][grp][threadIdx.x & 31] = s0; red[1][grp][threadIdx.x & 31] = s1;
    __syncthreads();
    if (grp == 0 && c < C) {
        float t0 = 0.f, t1 = 0.f;
        for (int g = 0; g < NG; ++g) { t0 += red[0][g][threadIdx.x & 31]; t1 += red[1][g][threadIdx.x & 31]; }
        // write-through (sc1) stores: relaxed agent-scope atomic stores of 4 bytes -- visible to every XCD without a release
        // fence (the fence form, buffer_wbl2 in every block, measured 0.26 ms/step slower than two launches)
#ifndef HIPEMU
        __hip_atomic_store(&compact[(size_t)blockIdx.y * 2 * C + c], t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&compact[(size_t)blockIdx.y * 2 * C + C + c], t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        compact[(size_t)blockIdx.y * 2 * C + c] = t0;
        compact[(size_t)blockIdx.y * 2 * C + C + c] = t1;
#endif
    }
    // ---- publish the compact strip (every wave: its stores have left), draw the ticket
#ifndef HIPEMU
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __syncthreads();
    if (threadIdx.x == 0) {
#ifndef HIPEMU
        const int t = __hip_atomic_fetch_add(&tickets[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        const int t = atomicAdd(&tickets[blockIdx.x], 1);
#endif
        s_last = t == ny - 1;
    }
    __syncthreads();
    if (!s_last) return;
    // ---- the last arriver of this channel group: fold the ny compact strips (sc1 loads: straight from the coherent level), finalize
    s0 = s1 = 0.f;
    if (c < C)
        for (int y = grp; y < ny; y += NG) {
#ifndef HIPEMU
            s0 += __hip_atomic_load(&compact[(size_t)y * 2 * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s1 += __hip_atomic_load(&compact[(size_t)y * 2 * C + C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
            s0 += compact[(size_t)y * 2 * C + c]; s1 += compact[(size_t)y * 2 * C + C + c];
#endif
        }
    red[0][grp][threadIdx.x & 31] = s0; red[1][grp][threadIdx.x & 31] = s1;
    __syncthreads();
    if (threadIdx.x == 0) tickets[blockIdx.x] = 0;              // re-armed for the next launch on this stream
    if (grp != 0 || c >= C) return;
    float t0 = 0.f, t1 = 0.f;
    for (int g = 0; g < NG; ++g) { t0 += red[0][g][threadIdx.x & 31]; t1 += red[1][g][threadIdx.x & 31]; }
    if constexpr (!BWD) {
        if (c == 0 && a.nbt) *a.nbt += 1;
        const float ms = t0 / (float)a.P;
        float var = t1 / (float)a.P - ms * ms;
        var = var > 0.f ? var : 0.f;
        const float m = a.pre_shift[c] + ms;
        const float r = rsqrtf(var + a.eps);
        a.mean[c] = m; a.rstd[c] = r;
        const float sc = a.gamma[c] * r;
        a.scale[c] = sc; a.shift[c] = a.beta[c] - m * sc;
        if (a.running_mean) {
            a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * m;
            const float unbiased = a.P > 1 ? var * ((float)a.P / (float)(a.P - 1)) : var;
            a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * unbiased;
        }
    } else {
        a.coef[c] = a.gamma[c] * a.rstd_in[c];
        a.coef[C + c] = t0 / (float)a.P;
        a.coef[2 * C + c] = t1 / (float)a.P;
        a.dgamma[c] += t1;
        a.dbeta[c] += t0;
    }
}

template <class T>
__global__ void bn_fwd_finalize_kernel(const T* __restrict__ x, const float* __restrict__ pre_shift,
                                       const float* __restrict__ sums, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float* __restrict__ mean,
                                       float* __restrict__ rstd, float* __restrict__ scale,
                                       float* __restrict__ shift, float* __restrict__ running_mean,
                                       float* __restrict__ running_var, long long* __restrict__ nbt,
                                       int P, int C, float eps, float momentum, int nparts) {
    const int c = blockIdx.x * FIN_CH + threadIdx.x % FIN_CH, grp = threadIdx.x / FIN_CH;
    float t0, t1;
    sum_partials(sums, C, nparts, c, grp, t0, t1);
    if (c == 0 && grp == 0 && nbt) *nbt += 1;
    if (c >= C || grp != 0) return;
    const float ms = t0 / (float)P;                   // mean of (x - x[0][c])
    float var = t1 / (float)P - ms * ms;
    var = var > 0.f ? var : 0.f;
    // the sums are of (x - shift): shift = the channel's first sample (stand-alone reduction) or the
    // caller's per-channel shift (statistics produced by the convolution epilogue)
    const float m = (pre_shift ? pre_shift[c] : Elem<T>::ld(x + c)) + ms;
    const float r = rsqrtf(var + eps);
    mean[c] = m; rstd[c] = r;
    const float sc = gamma[c] * r;
    scale[c] = sc; shift[c] = beta[c] - m * sc;
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
        const float unbiased = P > 1 ? var * ((float)P / (float)(P - 1)) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

// The three apply kernels below share one thread layout: x is a flat array of 16-byte vectors, thread t of the grid
// walks vectors t, t + S, t + 2S, ... with S = gridDim.x * 256.  The host makes S a multiple of cv = C / VEC (a power
// of two), so a thread stays on ONE channel vector for its whole life: the per-channel coefficients are loaded into
// registers once, and a trip is UNR payload loads, ~2 VALU per element, UNR stores -- the first version reloaded every
// coefficient table for every vector (six to ten cached loads per 16 bytes of payload) behind a 64-bit modulo.
// ADJ (round 3, tools/probes/stream_probe.hip): the UNR vectors of a trip are ADJACENT -- a block owns 256 * UNR consecutive
// vectors per trip (cv divides 256, so a thread still keeps its channel vector) -- instead of one grid stride (16.8 MB) apart.
// Measured on the 411 MB tensors of stage 1 (2 reads + 1 write): 282 us / 4.37 TB/s in the strided form, 218 us / 5.67 TB/s
// adjacent with four vectors per thread and 8192 blocks; hipMemcpy moves one such tensor at 5.34 TB/s.
template <class T, int UNR, bool ADJ>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ residual,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ beta, T* __restrict__ y,
                                                       uint8_t* __restrict__ bits, long nvec, int C, int relu) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = C / VEC;
    const long t0 = ADJ ? (long)blockIdx.x * 256 * UNR + threadIdx.x : (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = ADJ ? 256 : (long)gridDim.x * 256;                       // between the vectors of one trip
    const long trip = (long)gridDim.x * 256 * UNR;
    const int c0 = (int)(t0 & (long)(cv - 1)) * VEC;
    float mu[VEC], sc[VEC], be[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { mu[j] = mean[c0 + j]; sc[j] = scale[c0 + j]; be[j] = beta[c0 + j]; }
    for (long i0 = t0; i0 < nvec; i0 += trip) {
        Vec16<T> v[UNR], r[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long i = i0 + u * stride;
            if (i < nvec) { v[u].load(x + i * VEC); if (residual) r[u].load(residual + i * VEC); }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long i = i0 + u * stride;
            if (i >= nvec) break;
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[u].v[j] = (v[u].v[j] - mu[j]) * sc[j] + be[j];  // centred first: no cancellation
            if (residual) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) v[u].v[j] += r[u].v[j];
            }
            if (relu) {
                if (bits) {                      // the ReLU mask as one bit per element: what the backward reads instead of y
                    uint32_t mb = 0;
#pragma unroll
                    for (int j = 0; j < VEC; ++j) mb |= (v[u].v[j] > 0.f ? 1u : 0u) << j;
                    bits[i] = (uint8_t)mb;
                }
#pragma unroll
                for (int j = 0; j < VEC; ++j) v[u].v[j] = fmaxf(v[u].v[j], 0.f);
            }
            v[u].store(y + i * VEC);
        }
    }
}

// coef[0][c] = gamma*rstd, coef[1][c] = s1/P, coef[2][c] = s2/P ; dgamma += s2 ; dbeta += s1
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ gamma,
                                       const float* __restrict__ rstd, float* __restrict__ coef,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int P, int C,
                                       int nparts) {
    const int c = blockIdx.x * FIN_CH + threadIdx.x % FIN_CH, grp = threadIdx.x / FIN_CH;
    float s1, s2;
    sum_partials(sums, C, nparts, c, grp, s1, s2);
    if (c >= C || grp != 0) return;
    coef[c] = gamma[c] * rstd[c];
    coef[C + c] = s1 / (float)P;
    coef[2 * C + c] = s2 / (float)P;
    dgamma[c] += s2;
    dbeta[c] += s1;
}

template <class T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ ymask,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ coef,
    const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ dx, T* __restrict__ dz_out,
    long nvec, int C) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = C / VEC;
    const long t0 = (long)blockIdx.x * 256 + threadIdx.x, stride = (long)gridDim.x * 256;
    const int c0 = (int)(t0 & (long)(cv - 1)) * VEC;
    float mu[VEC], rs[VEC], k0[VEC], k1[VEC], k2[VEC], ga[VEC], be[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        mu[j] = mean[c0 + j]; rs[j] = rstd[c0 + j];
        k0[j] = coef[c0 + j]; k1[j] = coef[C + c0 + j]; k2[j] = coef[2 * C + c0 + j];
        ga[j] = beta ? gamma[c0 + j] : 0.f; be[j] = beta ? beta[c0 + j] : 0.f;
    }
    for (long i = t0; i < nvec; i += stride) {
        Vec16<T> xv, g; xv.load(x + i * VEC); g.load(dy + i * VEC);
        if (ymask) {
            Vec16<T> m; m.load(ymask + i * VEC);
#pragma unroll
            for (int j = 0; j < VEC; ++j) g.v[j] = m.v[j] > 0.f ? g.v[j] : 0.f;
        }
        Vec16<T> o;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float xh = (xv.v[j] - mu[j]) * rs[j];
            if (beta) g.v[j] = xh * ga[j] + be[j] > 0.f ? g.v[j] : 0.f;
            o.v[j] = k0[j] * (g.v[j] - k1[j] - xh * k2[j]);
        }
        if (dz_out) g.store(dz_out + i * VEC);
        o.store(dx + i * VEC);
    }
}

// backward apply when the producing kernel already masked the gradient and emitted the sums (VtxBnBwdFusion):
// dx = gamma*rstd * (dz - s1/P - xhat*s2/P); reads x and dz, writes dx -- nothing else
// UNR independent 16-byte vectors per thread and trip, all loads issued before the first use (memory-level parallelism)
template <class T, int UNR, bool ADJ>
__global__ __launch_bounds__(256) void bn_bwd_apply_fused_kernel(const T* __restrict__ x, const T* __restrict__ dz,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 const float* __restrict__ coef, T* __restrict__ dx, long nvec, int C) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = C / VEC;
    const long t0 = ADJ ? (long)blockIdx.x * 256 * UNR + threadIdx.x : (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = ADJ ? 256 : (long)gridDim.x * 256;                       // between the vectors of one trip (see bn_apply_kernel)
    const long trip = (long)gridDim.x * 256 * UNR;
    const int c0 = (int)(t0 & (long)(cv - 1)) * VEC;
    // dx = k0*(dz - k1 - xhat*k2), xhat = (x - mu)*rs  ==  k0*dz - (x - mu)*(k0*k2*rs) - k0*k1
    float mu[VEC], a0[VEC], a1[VEC], a2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const float k0 = coef[c0 + j], k1 = coef[C + c0 + j], k2 = coef[2 * C + c0 + j];
        mu[j] = mean[c0 + j]; a0[j] = k0; a1[j] = k0 * k2 * rstd[c0 + j]; a2[j] = k0 * k1;
    }
    for (long i0 = t0; i0 < nvec; i0 += trip) {
        Vec16<T> xv[UNR], g[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long i = i0 + u * stride;
            if (i < nvec) { xv[u].load(x + i * VEC); g[u].load(dz + i * VEC); }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long i = i0 + u * stride;
            if (i >= nvec) break;
            Vec16<T> o;
#pragma unroll
            for (int j = 0; j < VEC; ++j) o.v[j] = a0[j] * g[u].v[j] - (xv[u].v[j] - mu[j]) * a1[j] - a2[j];
            o.store(dx + i * VEC);
        }
    }
}

// ---- the stem's tail: MaxPool2d(3,2,1) backward gathered on the fly inside the BatchNorm backward -------------
// gradient wrt the pre-pool tensor at pixel (n, ih, iw): the dy of the (<= 2x2) pooling windows whose argmax is this
// pixel (same gather as maxpool_bwd_kernel in pool.hip; first-maximum tie rule lives in the forward's argmax)
__device__ __forceinline__ int bnpool_qdiv(int n, int d) { return vtx_fdiv30(n, d, __builtin_amdgcn_rcpf((float)d)); }   // exact for 0 <= n < 2^30
// (PoolQuad, pool_windows.h: a thread owns a 2 x 2 quad of input pixels = exactly four pooling windows; everything it needs is
// requested before anything is used)
// reduce: s1 = sum dz, s2 = sum dz*xhat with dz = pool-gathered gradient masked by relu(xhat*gamma+beta) > 0
template <class T>
__global__ __launch_bounds__(256) void pool_bn_bwd_reduce_kernel(
    const T* __restrict__ x, const T* __restrict__ dpool, const uint8_t* __restrict__ argmax, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ sums,
    int N, int H, int W, int C, int OH, int OW, int TX, int quads_per_block) {
    constexpr int VEC = Elem<T>::VEC;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX, TY = 256 / TX;
    const int c0 = (blockIdx.y * TX + tx) * VEC;
    const int QH = (H + 1) >> 1, QW = (W + 1) >> 1, NQ = N * QH * QW;
    const int q0 = blockIdx.x * quads_per_block;
    const int q1 = q0 + quads_per_block < NQ ? q0 + quads_per_block : NQ;
    float a[VEC], b[VEC], mu[VEC], rs[VEC], ga[VEC], be[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { a[j] = b[j] = 0.f; mu[j] = mean[c0 + j]; rs[j] = rstd[c0 + j]; ga[j] = gamma[c0 + j]; be[j] = beta[c0 + j]; }
    for (int q = q0 + ty; q < q1; q += TY) {
        const int n = bnpool_qdiv(q, QH * QW), rem = q - n * QH * QW;
        const int qa = bnpool_qdiv(rem, QW), qb = rem - qa * QW;
        PoolQuad<T> quad;
        quad.request(x, dpool, argmax, n, qa, qb, c0, H, W, C, OH, OW);
        auto pixel = [&](auto K) {
            constexpr int k = decltype(K)::value;
            float g[VEC], xv[VEC];
            quad.template gather<k>(g);
            vtx_unpack_raw16<T>(quad.x[k], xv);
            const bool in = (quad.pvalid >> k) & 1u;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float xh = (xv[j] - mu[j]) * rs[j];
                const float d = (in && xh * ga[j] + be[j] > 0.f) ? g[j] : 0.f;
                a[j] += d; b[j] += d * xh;
            }
        };
        pixel(std::integral_constant<int, 0>{}); pixel(std::integral_constant<int, 1>{});
        pixel(std::integral_constant<int, 2>{}); pixel(std::integral_constant<int, 3>{});
    }
    __shared__ float red[2][256 * VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { red[0][(ty * TX + tx) * VEC + j] = a[j]; red[1][(ty * TX + tx) * VEC + j] = b[j]; }
    __syncthreads();
    if (ty == 0) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float s0 = 0.f, s1 = 0.f;
            for (int r = 0; r < TY; ++r) { s0 += red[0][(r * TX + tx) * VEC + j]; s1 += red[1][(r * TX + tx) * VEC + j]; }
            sums[(size_t)blockIdx.x * 2 * C + c0 + j] = s0;
            sums[(size_t)blockIdx.x * 2 * C + C + c0 + j] = s1;
        }
    }
}

template <class T>
__global__ __launch_bounds__(256) void pool_bn_bwd_apply_kernel(
    const T* __restrict__ x, const T* __restrict__ dpool, const uint8_t* __restrict__ argmax, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ coef, const float* __restrict__ gamma, const float* __restrict__ beta,
    T* __restrict__ dx, int N, int H, int W, int C, int OH, int OW, int xcd_major) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = C / VEC;
    const int QH = (H + 1) >> 1, QW = (W + 1) >> 1;
    const long total = (long)N * QH * QW * cv;
    // XCD-major block order (vtx_common.h): the pixel quads of neighbouring blocks share pooling windows and argmax rows.  Only where
    // a block covers whole channel-vector groups (256 % cv == 0), so that a thread's channel vector does not depend on the block
    const int blk = (xcd_major && 256 % cv == 0) ? vtx_xcd_major_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    // gridDim.x * 256 is a multiple of cv (apply_grid): a thread keeps its channel vector, coefficients live in registers
    const int c0 = (int)(((long)blk * 256 + threadIdx.x) % cv) * VEC;
    float mu[VEC], rs[VEC], ga[VEC], be[VEC], k0[VEC], k1[VEC], k2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        mu[j] = mean[c0 + j]; rs[j] = rstd[c0 + j]; ga[j] = gamma[c0 + j]; be[j] = beta[c0 + j];
        k0[j] = coef[c0 + j]; k1[j] = coef[C + c0 + j]; k2[j] = coef[2 * C + c0 + j];
    }
    for (long i = (long)blk * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int q = (int)(i / cv);
        const int n = bnpool_qdiv(q, QH * QW), rem = q - n * QH * QW;
        const int qa = bnpool_qdiv(rem, QW), qb = rem - qa * QW;
        PoolQuad<T> quad;
        quad.request(x, dpool, argmax, n, qa, qb, c0, H, W, C, OH, OW);
        auto pixel = [&](auto K) {
            constexpr int k = decltype(K)::value;
            float g[VEC], xv[VEC];
            quad.template gather<k>(g);
            vtx_unpack_raw16<T>(quad.x[k], xv);
            Vec16<T> o;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float xh = (xv[j] - mu[j]) * rs[j];
                const float d = xh * ga[j] + be[j] > 0.f ? g[j] : 0.f;
                o.v[j] = k0[j] * (d - k1[j] - xh * k2[j]);
            }
            if ((quad.pvalid >> k) & 1u) o.store(dx + quad.xoff[k]);
        };
        pixel(std::integral_constant<int, 0>{}); pixel(std::integral_constant<int, 1>{});
        pixel(std::integral_constant<int, 2>{}); pixel(std::integral_constant<int, 3>{});
    }
}

// ---- the stem's forward tail in one pass: y = relu(BatchNorm(x)) is max-pooled (3x3, stride 2, pad 1) as it is
// computed and never written -- the 411 MB tensor between the stem's BatchNorm and its pooling disappears (its backward
// recomputes the ReLU mask from x).  Every tap value is rounded to the storage type before the comparison, so pooled
// values AND argmax are bit-identical to vtx_bn_fwd followed by vtx_maxpool3x3s2_fwd (first maximum wins on ties).
// Thread layout of the pooling kernels: tx = channel vector (coefficients in registers), ty = output pixel.
template <class T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                                  const float* __restrict__ scale, const float* __restrict__ beta,
                                                                  T* __restrict__ y, uint8_t* __restrict__ argmax, int N, int H, int W,
                                                                  int C, int OH, int OW, int TX, int xcd_major) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = C / VEC, TY = 256 / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int cvi = blockIdx.y * TX + tx;
    if (cvi >= cv || ty >= TY) return;
    const int c0 = cvi * VEC, P = N * OH * OW;
    float mu[VEC], sc[VEC], be[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { mu[j] = mean[c0 + j]; sc[j] = scale[c0 + j]; be[j] = beta[c0 + j]; }
    // (pooling windows of vertically / horizontally adjacent output pixels overlap: with consecutive blocks on consecutive XCDs the
    //  input was fetched 1.5 x -- 621 MB for 411 MB, PMC; in XCD-major order the overlaps meet in one L2)
    const int blk = (xcd_major && gridDim.y == 1) ? vtx_xcd_major_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    for (int p = blk * TY + ty; p < P; p += gridDim.x * TY) {
        const int n = bnpool_qdiv(p, OH * OW), rem = p - n * OH * OW;
        const int oh = bnpool_qdiv(rem, OW), ow = rem - oh * OW;
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
            for (int j = 0; j < VEC; ++j) {
                float a = fmaxf((v[j] - mu[j]) * sc[j] + be[j], 0.f);      // exactly bn_apply_kernel's arithmetic
                if constexpr (sizeof(T) == 2) a = bf2f(f2bf(a));              // ... and its storage rounding
                if (ok && (first || a > best[j])) { best[j] = a; idx[j] = k; }
            }
            first = first && !ok;
        }
        Vec16<T> o;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o.v[j] = best[j];
        const long off = (long)p * C + c0;
        o.store(y + off);
        if constexpr (VEC == 8)
            *reinterpret_cast<uint2*>(argmax + off) =
                make_uint2((uint32_t)idx[0] | ((uint32_t)idx[1] << 8) | ((uint32_t)idx[2] << 16) | ((uint32_t)idx[3] << 24),
                           (uint32_t)idx[4] | ((uint32_t)idx[5] << 8) | ((uint32_t)idx[6] << 16) | ((uint32_t)idx[7] << 24));
        else
            *reinterpret_cast<uint32_t*>(argmax + off) =
                (uint32_t)idx[0] | ((uint32_t)idx[1] << 8) | ((uint32_t)idx[2] << 16) | ((uint32_t)idx[3] << 24);
    }
}

// vectors in flight per thread of the apply kernels: 0 = by size, or 1, 2, 4 (A/B switch VIRTEX_AMD_BN_UNROLL)
int g_bn_apply_unroll = getenv("VIRTEX_AMD_BN_UNROLL") ? atoi(getenv("VIRTEX_AMD_BN_UNROLL")) : 0;
constexpr int VTX_BN_MAX_PARTS = 512;
struct ReducePlan { int TX, gy, gx, rows, krows; };   // krows: what bn_reduce_kernel gets (0 = interleaved trips)
static ReducePlan plan_reduce(int P, int C, int vec) {
    ReducePlan r;
    const int cv = C / vec;
    r.TX = cv < 256 ? cv : 256;
    r.gy = cv / r.TX;
    const int TY = 256 / r.TX;
    int gx = VTX_BN_MAX_PARTS;
    const int max_gx = vtx_cdiv(P, TY * 4);
    if (gx > max_gx) gx = max_gx;
    if (gx < 1) gx = 1;
    r.rows = vtx_cdiv(P, gx);
    r.gx = vtx_cdiv(P, r.rows);
    r.krows = g_vtx_sw_bn_red_adj ? 0 : r.rows;      // interleaved trips (see bn_reduce_kernel); same grid, same partial layout
    return r;
}
// grid of the flat apply kernels: at most 4096 blocks, and gridDim.x * 256 a multiple of cv (see bn_apply_kernel)
static int apply_grid(long nvec, int cv = 1) {
    long g = (nvec + 255) / 256;
    g = g > 4096 ? 4096 : (g < 1 ? 1 : g);
    const long q = cv > 256 ? cv / 256 : 1;
    g = (g + q - 1) / q * q;
    return (int)g;
}
// Form of the flat apply kernels for one launch: adjacent vectors (ADJ) when the channel vectors of a row divide the block
// (cv <= 256) and the switch allows; unroll by size; grid cap 8192 blocks for ADJ (4096 for the strided form, whose grid must
// be a multiple of cv / 256).  vtx_set_switch("bn_adj", 0 | 1), ("bn_grid", cap), vtx_set_bn_apply_unroll.
struct ApplyPlan { bool adj; int unr, grid; };
static ApplyPlan plan_apply(long nvec, int cv, int max_unr) {
    ApplyPlan p;
    p.adj = g_vtx_sw_bn_adj && cv <= 256;
    if (p.adj) {
        // tools/bench_bn_apply.py (profiles/r03_bn_apply_forms.txt): four vectors per thread pay on the 411 MB tensors only
        // (220 vs 229 us), two on everything from 25 MB up (51.6 vs 61.0 us at 103 MB with four)
        p.unr = g_bn_apply_unroll ? g_bn_apply_unroll : (nvec >= (16L << 20) ? 4 : nvec >= (1L << 18) ? 2 : 1);
        if (p.unr > max_unr) p.unr = max_unr;
        long g = (nvec + 256L * p.unr - 1) / (256L * p.unr);
        const long cap = g_vtx_sw_bn_grid > 0 ? g_vtx_sw_bn_grid : 8192;
        p.grid = (int)(g > cap ? cap : (g < 1 ? 1 : g));
    } else {
        p.unr = g_bn_apply_unroll ? g_bn_apply_unroll : (nvec >= (6L << 20) ? 2 : 1);
        if (p.unr > max_unr) p.unr = max_unr;
        if (p.unr == 3) p.unr = 2;
        p.grid = apply_grid(vtx_cdiv(nvec, p.unr), cv);
    }
    return p;
}
static bool bn_shape_ok(int C, int vec) { return C > 0 && C % vec == 0 && ((C / vec) & (C / vec - 1)) == 0; }

}  // namespace

constexpr int VTX_BN_WS_HEADER = 64;      // ints in front of the workspace: tickets of bn_fin2_kernel (the caller zero-initialises the buffer ONCE)
extern int g_vtx_sw_bn_fin2;              // vtx_set_switch("bn_fin2"): 1 = compaction + finalize in one launch, 0 (default: faster, see core.hip) two launches
extern "C" long vtx_bn_workspace_floats(int C) { return VTX_BN_WS_HEADER + (long)(2 * VTX_BN_MAX_PARTS + 4) * C; }

// workspace layout (fp32): scale[C] | coef[3*C] | partial sums [nparts][2][C]   (need not be zeroed)
extern "C" int vtx_bn_fwd(int dtype, const void* x, const void* residual, const float* gamma,
                          const float* beta, float* running_mean, float* running_var,
                          long long* num_batches_tracked, void* y, float* save_mean, float* save_rstd,
                          float* workspace, int P, int C, float eps, float momentum, int relu,
                          const float* pre_partials, int pre_nparts, const float* pre_shift, uint8_t* relu_bits,
                          void* stream) {
    VTX_CHECK(x && gamma && beta && y && save_mean && save_rstd && workspace, VTX_ERR_ARG, "bn_fwd: null pointer");
    VTX_CHECK(!relu_bits || (dtype == VTX_BF16 && relu), VTX_ERR_ARG, "bn_fwd: relu_bits needs bf16 and relu");
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "bn_fwd: bad dtype %d", dtype);
    const int vec = dtype == VTX_BF16 ? 8 : 4;
    VTX_CHECK(P > 0 && bn_shape_ok(C, vec), VTX_ERR_SHAPE, "bn_fwd: C=%d must be vec*2^k, P=%d > 0", C, P);
    hipStream_t st = (hipStream_t)stream;
    int* tickets = reinterpret_cast<int*>(workspace); (void)tickets;
    workspace += VTX_BN_WS_HEADER;
    float* scale = workspace; float* sums = workspace + 4 * C;
    ReducePlan rp = plan_reduce(P, C, vec);
    const long nvec = (long)P * C / vec;
    const bool fused = pre_partials != nullptr && pre_nparts > 0;   // statistics came with the conv epilogue
    bool fin_done = false;
    VTX_CHECK(!fused || pre_shift, VTX_ERR_ARG, "bn_fwd: fused statistics need the shift vector they were taken against");
    if (fused) {
        const float* parts = pre_partials;
        int np = pre_nparts;
        if (np > 512 && g_vtx_sw_bn_fin2 && C <= 32 * VTX_BN_WS_HEADER) {
            // thousands of strips: folded and finalized by ONE launch (bn_fin2_kernel: last-arriving block per channel group)
            const int per = vtx_cdiv(np, 64), ny = vtx_cdiv(np, per);
            BnFin2Args fa{};
            fa.pre_shift = pre_shift; fa.gamma = gamma; fa.beta = beta; fa.mean = save_mean; fa.rstd = save_rstd; fa.scale = scale; fa.shift = scale + C;
            fa.running_mean = running_mean; fa.running_var = running_var; fa.nbt = num_batches_tracked; fa.eps = eps; fa.momentum = momentum; fa.P = P;
            VTX_KLAUNCH("bn_finalize", 0, 8.0 * np * C, (bn_fin2_kernel<false>), dim3(vtx_cdiv(C, 32), ny), dim3(256), 0, st, parts, sums, tickets, C, np, per, fa);
            fin_done = true;
        } else if (np > 512) {      // the same in two launches (A/B: vtx_set_switch("bn_fin2", 0))
            const int per = vtx_cdiv(np, 64), ny = vtx_cdiv(np, per);
            VTX_KLAUNCH("bn_finalize", 0, 8.0 * np * C, bn_compact_parts_kernel, dim3(vtx_cdiv(C, 32), ny), dim3(g_vtx_sw_bn_fin_wide ? 1024 : 256), 0, st, parts, sums, C, np, per);
            parts = sums; np = ny;
        }
        sums = const_cast<float*>(parts); rp.gx = np;
    }
    else if (dtype == VTX_BF16)
        VTX_KLAUNCH("bn_fwd_reduce", 0, 2.0 * P * C, (bn_reduce_kernel<bf16_t, false>), dim3(rp.gx, rp.gy), dim3(256), 0, st, (const bf16_t*)x,
                           (const bf16_t*)nullptr, (const bf16_t*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, sums, P, C, rp.TX, rp.krows);
    else
        VTX_KLAUNCH("bn_fwd_reduce", 0, 4.0 * P * C, (bn_reduce_kernel<float, false>), dim3(rp.gx, rp.gy), dim3(256), 0, st, (const float*)x,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, sums, P, C, rp.TX, rp.krows);
    if (fin_done) {}
    else if (dtype == VTX_BF16)
        VTX_KLAUNCH("bn_finalize", 0, 8.0 * rp.gx * C, (bn_fwd_finalize_kernel<bf16_t>), dim3(vtx_cdiv(C, FIN_CH)), fin_block(rp.gx), 0, st, (const bf16_t*)x, fused ? pre_shift : (const float*)nullptr, sums, gamma, beta,
                           save_mean, save_rstd, scale, scale + C, running_mean, running_var, num_batches_tracked, P, C, eps, momentum, rp.gx);
    else
        VTX_KLAUNCH("bn_finalize", 0, 8.0 * rp.gx * C, (bn_fwd_finalize_kernel<float>), dim3(vtx_cdiv(C, FIN_CH)), fin_block(rp.gx), 0, st, (const float*)x, fused ? pre_shift : (const float*)nullptr, sums, gamma, beta,
                           save_mean, save_rstd, scale, scale + C, running_mean, running_var, num_batches_tracked, P, C, eps, momentum, rp.gx);
    const ApplyPlan ap = plan_apply(nvec, C / vec, dtype == VTX_BF16 ? 4 : 1);
#define VTX_FWD_APPLY(T, U, A, BYTES)                                                                                              \
    VTX_KLAUNCH("bn_fwd_apply", 0, BYTES, (bn_apply_kernel<T, U, A>), dim3(ap.grid), dim3(256), 0, st, (const T*)x, (const T*)residual, \
                save_mean, scale, beta, (T*)y, relu_bits, nvec, C, relu)
    const double fb = (dtype == VTX_BF16 ? 2.0 : 4.0) * P * C * (residual ? 3 : 2);
    if (dtype == VTX_BF16) {
        if (ap.adj) { if (ap.unr == 4) VTX_FWD_APPLY(bf16_t, 4, true, fb); else if (ap.unr == 2) VTX_FWD_APPLY(bf16_t, 2, true, fb); else VTX_FWD_APPLY(bf16_t, 1, true, fb); }
        else { if (ap.unr >= 2) VTX_FWD_APPLY(bf16_t, 2, false, fb); else VTX_FWD_APPLY(bf16_t, 1, false, fb); }
    } else {
        relu_bits = nullptr;
        if (ap.adj) VTX_FWD_APPLY(float, 1, true, fb); else VTX_FWD_APPLY(float, 1, false, fb);
    }
#undef VTX_FWD_APPLY
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

// dx = k0*(dz - k1 - xhat*k2): the flat apply of the fused backward (and of BatchNorms without a ReLU behind them)
static void launch_bwd_apply_fused(int dtype, const void* x, const void* dz, const float* save_mean, const float* save_rstd,
                                   const float* coef, void* dx, long nvec, int P, int C, int vec, hipStream_t st) {
    const ApplyPlan ap = plan_apply(nvec, C / vec, dtype == VTX_BF16 ? 4 : 1);
#define VTX_BWD_APPLY(T, U, A, BYTES)                                                                                              \
    VTX_KLAUNCH("bn_bwd_apply", 0, BYTES, (bn_bwd_apply_fused_kernel<T, U, A>), dim3(ap.grid), dim3(256), 0, st, (const T*)x, (const T*)dz, \
                save_mean, save_rstd, coef, (T*)dx, nvec, C)
    if (dtype == VTX_BF16) {
        if (ap.adj) { if (ap.unr == 4) VTX_BWD_APPLY(bf16_t, 4, true, 6.0 * P * C); else if (ap.unr == 2) VTX_BWD_APPLY(bf16_t, 2, true, 6.0 * P * C); else VTX_BWD_APPLY(bf16_t, 1, true, 6.0 * P * C); }
        else { if (ap.unr == 4) VTX_BWD_APPLY(bf16_t, 4, false, 6.0 * P * C); else if (ap.unr == 2) VTX_BWD_APPLY(bf16_t, 2, false, 6.0 * P * C); else VTX_BWD_APPLY(bf16_t, 1, false, 6.0 * P * C); }
    } else {
        if (ap.adj) VTX_BWD_APPLY(float, 1, true, 12.0 * P * C); else VTX_BWD_APPLY(float, 1, false, 12.0 * P * C);
    }
#undef VTX_BWD_APPLY
}

// workspace: same buffer / layout as vtx_bn_fwd (vtx_bn_workspace_floats(C) floats)
extern "C" int vtx_bn_bwd(int dtype, const void* x, const void* dy, const void* ymask, const float* gamma,
                          const float* relu_beta, const float* save_mean, const float* save_rstd, void* dx, void* dz_out,
                          float* dgamma, float* dbeta, float* workspace, int P, int C, void* stream) {
    VTX_CHECK(x && dy && gamma && save_mean && save_rstd && dx && dgamma && dbeta && workspace, VTX_ERR_ARG,
              "bn_bwd: null pointer");
    VTX_CHECK(!(ymask && relu_beta), VTX_ERR_ARG, "bn_bwd: pass either ymask or relu_beta, not both");
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "bn_bwd: bad dtype %d", dtype);
    const int vec = dtype == VTX_BF16 ? 8 : 4;
    VTX_CHECK(P > 0 && bn_shape_ok(C, vec), VTX_ERR_SHAPE, "bn_bwd: C=%d must be vec*2^k, P=%d > 0", C, P);
    hipStream_t st = (hipStream_t)stream;
    int* tickets = reinterpret_cast<int*>(workspace); (void)tickets;
    workspace += VTX_BN_WS_HEADER;
    float* coef = workspace + C; float* sums = workspace + 4 * C;
    ReducePlan rp = plan_reduce(P, C, vec);
    const long nvec = (long)P * C / vec;
    if (dtype == VTX_BF16)
        VTX_KLAUNCH("bn_bwd_reduce", 0, 2.0 * P * C * (ymask ? 3 : 2), (bn_reduce_kernel<bf16_t, true>), dim3(rp.gx, rp.gy), dim3(256), 0, st, (const bf16_t*)x,
                           (const bf16_t*)dy, (const bf16_t*)ymask, save_mean, save_rstd, gamma, relu_beta, sums, P, C, rp.TX, rp.krows);
    else
        VTX_KLAUNCH("bn_bwd_reduce", 0, 4.0 * P * C * (ymask ? 3 : 2), (bn_reduce_kernel<float, true>), dim3(rp.gx, rp.gy), dim3(256), 0, st, (const float*)x,
                           (const float*)dy, (const float*)ymask, save_mean, save_rstd, gamma, relu_beta, sums, P, C, rp.TX, rp.krows);
    VTX_KLAUNCH("bn_finalize", 0, 8.0 * rp.gx * C, bn_bwd_finalize_kernel, dim3(vtx_cdiv(C, FIN_CH)), fin_block(rp.gx), 0, st, sums, gamma, save_rstd, coef,
                dgamma, dbeta, P, C, rp.gx);
    if (!ymask && !relu_beta && !dz_out) {
        // no ReLU behind this BatchNorm (the projection shortcuts): dy is the gradient itself, so the apply is the one
        // of the fused path -- read x and dy, write dx, four coefficients per channel in registers, two vectors in flight
        launch_bwd_apply_fused(dtype, x, dy, save_mean, save_rstd, coef, dx, nvec, P, C, vec, st);
        VTX_LAUNCH_CHECK();
        return VTX_OK;
    }
    if (dtype == VTX_BF16)
        VTX_KLAUNCH("bn_bwd_apply", 0, 2.0 * P * C * (3 + (ymask ? 1 : 0) + (dz_out ? 1 : 0)), (bn_bwd_apply_kernel<bf16_t>), dim3(apply_grid(nvec, C / vec)), dim3(256), 0, st, (const bf16_t*)x,
                           (const bf16_t*)dy, (const bf16_t*)ymask, save_mean, save_rstd, coef, gamma, relu_beta, (bf16_t*)dx,
                           (bf16_t*)dz_out, nvec, C);
    else
        VTX_KLAUNCH("bn_bwd_apply", 0, 4.0 * P * C * (3 + (ymask ? 1 : 0) + (dz_out ? 1 : 0)), (bn_bwd_apply_kernel<float>), dim3(apply_grid(nvec, C / vec)), dim3(256), 0, st, (const float*)x,
                           (const float*)dy, (const float*)ymask, save_mean, save_rstd, coef, gamma, relu_beta, (float*)dx,
                           (float*)dz_out, nvec, C);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

// workspace: same buffer / layout as vtx_bn_fwd.  pre_partials: [pre_nparts][2][C] {sum dz, sum dz*xhat} from the
// epilogue of the kernel that produced dz (VtxBnBwdFusion in virtex_amd.h).
extern "C" int vtx_bn_bwd_fused(int dtype, const void* x, const void* dz, const float* gamma, const float* save_mean,
                                const float* save_rstd, const float* pre_partials, int pre_nparts, void* dx, float* dgamma,
                                float* dbeta, float* workspace, int P, int C, void* stream) {
    VTX_CHECK(x && dz && gamma && save_mean && save_rstd && pre_partials && dx && dgamma && dbeta && workspace, VTX_ERR_ARG,
              "bn_bwd_fused: null pointer");
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "bn_bwd_fused: bad dtype %d", dtype);
    const int vec = dtype == VTX_BF16 ? 8 : 4;
    VTX_CHECK(P > 0 && pre_nparts > 0 && bn_shape_ok(C, vec), VTX_ERR_SHAPE, "bn_bwd_fused: C=%d must be vec*2^k, P=%d > 0", C, P);
    hipStream_t st = (hipStream_t)stream;
    int* tickets = reinterpret_cast<int*>(workspace); (void)tickets;
    workspace += VTX_BN_WS_HEADER;
    float* coef = workspace + C; float* sums = workspace + 4 * C;
    const float* parts = pre_partials;
    int np = pre_nparts;
    if (np > 512 && g_vtx_sw_bn_fin2 && C <= 32 * VTX_BN_WS_HEADER) {
        const int per = vtx_cdiv(np, 64), ny = vtx_cdiv(np, per);
        BnFin2Args fa{};
        fa.gamma = gamma; fa.rstd_in = save_rstd; fa.coef = coef; fa.dgamma = dgamma; fa.dbeta = dbeta; fa.P = P;
        VTX_KLAUNCH("bn_finalize", 0, 8.0 * np * C, (bn_fin2_kernel<true>), dim3(vtx_cdiv(C, 32), ny), dim3(256), 0, st, parts, sums, tickets, C, np, per, fa);
    } else {
        if (np > 512) {
            const int per = vtx_cdiv(np, 64), ny = vtx_cdiv(np, per);
            VTX_KLAUNCH("bn_finalize", 0, 8.0 * np * C, bn_compact_parts_kernel, dim3(vtx_cdiv(C, 32), ny), dim3(g_vtx_sw_bn_fin_wide ? 1024 : 256), 0, st, parts, sums, C, np, per);
            parts = sums; np = ny;
        }
        VTX_KLAUNCH("bn_finalize", 0, 8.0 * np * C, bn_bwd_finalize_kernel, dim3(vtx_cdiv(C, FIN_CH)), fin_block(np), 0, st, parts, gamma, save_rstd, coef,
                    dgamma, dbeta, P, C, np);
    }
    const long nvec = (long)P * C / vec;
    launch_bwd_apply_fused(dtype, x, dz, save_mean, save_rstd, coef, dx, nvec, P, C, vec, st);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

// The finalize half of vtx_bn_bwd_fused alone: [compaction +] bn_bwd_finalize of the epilogue partials -> coef[3][C] inside
// `workspace` (valid until the next BatchNorm call on this stream), dgamma / dbeta accumulated.  For kernels that apply the
// BatchNorm backward themselves while they load the gradient (conv3_bwd.hip).
int vtx_bn_bwd_finalize_only(const float* gamma, const float* save_rstd, const float* pre_partials, int pre_nparts, float* dgamma,
                             float* dbeta, float* workspace, int P, int C, hipStream_t st, const float** coef_out) {
    VTX_CHECK(gamma && save_rstd && pre_partials && dgamma && dbeta && workspace && coef_out, VTX_ERR_ARG, "bn_bwd_finalize: null pointer");
    VTX_CHECK(P > 0 && pre_nparts > 0 && bn_shape_ok(C, 8), VTX_ERR_SHAPE, "bn_bwd_finalize: C=%d must be 8*2^k, P=%d > 0", C, P);
    workspace += VTX_BN_WS_HEADER;
    float* coef = workspace + C; float* sums = workspace + 4 * C;
    const float* parts = pre_partials;
    int np = pre_nparts;
    if (np > 512) {
        const int per = vtx_cdiv(np, 64), ny = vtx_cdiv(np, per);
        VTX_KLAUNCH("bn_finalize", 0, 8.0 * np * C, bn_compact_parts_kernel, dim3(vtx_cdiv(C, 32), ny), dim3(g_vtx_sw_bn_fin_wide ? 1024 : 256), 0, st, parts, sums, C, np, per);
        parts = sums; np = ny;
    }
    VTX_KLAUNCH("bn_finalize", 0, 8.0 * np * C, bn_bwd_finalize_kernel, dim3(vtx_cdiv(C, FIN_CH)), fin_block(np), 0, st, parts, gamma, save_rstd, coef,
                dgamma, dbeta, P, C, np);
    VTX_LAUNCH_CHECK();
    *coef_out = coef;
    return VTX_OK;
}

// The stem's backward tail in two passes instead of three kernels and an intermediate tensor:
//   dx = BatchNormBackward(ReLUBackward(MaxPoolBackward(dpool)))   with x = the stem convolution's output [N][H][W][C].
// Replaces aten::max_pool2d_with_indices_backward + threshold_backward + native_batch_norm_backward of the stem
// (/root/reference/virtex/modules/visual_backbones.py:68-74: conv1 -> bn1 -> relu -> maxpool).  The pre-pool gradient
// (411 MB at B = 256) is gathered on the fly by both passes and never written.
extern "C" int vtx_bn_bwd_maxpool(int dtype, const void* x, const void* dpool, const uint8_t* argmax, const float* gamma,
                                  const float* beta, const float* save_mean, const float* save_rstd, void* dx, float* dgamma,
                                  float* dbeta, float* workspace, int N, int H, int W, int C, void* stream) {
    VTX_CHECK(x && dpool && argmax && gamma && beta && save_mean && save_rstd && dx && dgamma && dbeta && workspace, VTX_ERR_ARG,
              "bn_bwd_maxpool: null pointer");
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "bn_bwd_maxpool: bad dtype %d", dtype);
    const int vec = dtype == VTX_BF16 ? 8 : 4;
    VTX_CHECK(N > 0 && H > 0 && W > 0 && bn_shape_ok(C, vec) && C % 8 == 0, VTX_ERR_SHAPE, "bn_bwd_maxpool: C=%d must be 8*2^k", C);
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long Pl = (long)N * H * W;
    VTX_CHECK(Pl < (1L << 31), VTX_ERR_SHAPE, "bn_bwd_maxpool: too many pixels");
    const int P = (int)Pl;
    const long NQl = (long)N * ((H + 1) / 2) * ((W + 1) / 2);            // 2 x 2 quads of input pixels: one per thread and trip
    VTX_CHECK(NQl < VTX_PIXEL_LIMIT, VTX_ERR_SHAPE, "bn_bwd_maxpool: more than 2^30 pixel quads is not supported");
    const int NQ = (int)NQl;
    hipStream_t st = (hipStream_t)stream;
    int* tickets = reinterpret_cast<int*>(workspace); (void)tickets;
    workspace += VTX_BN_WS_HEADER;
    float* coef = workspace + C; float* sums = workspace + 4 * C;
    ReducePlan rp = plan_reduce(NQ, C, vec);
    const double el = dtype == VTX_BF16 ? 2.0 : 4.0;
    const double pool_bytes = (double)N * OH * OW * C * (el + 1.0);
    if (dtype == VTX_BF16)
        VTX_KLAUNCH("bn_bwd_reduce", 0, el * P * C + pool_bytes, (pool_bn_bwd_reduce_kernel<bf16_t>), dim3(rp.gx, rp.gy), dim3(256), 0, st, (const bf16_t*)x,
                    (const bf16_t*)dpool, argmax, save_mean, save_rstd, gamma, beta, sums, N, H, W, C, OH, OW, rp.TX, rp.rows);
    else
        VTX_KLAUNCH("bn_bwd_reduce", 0, el * P * C + pool_bytes, (pool_bn_bwd_reduce_kernel<float>), dim3(rp.gx, rp.gy), dim3(256), 0, st, (const float*)x,
                    (const float*)dpool, argmax, save_mean, save_rstd, gamma, beta, sums, N, H, W, C, OH, OW, rp.TX, rp.rows);
    VTX_KLAUNCH("bn_finalize", 0, 8.0 * rp.gx * C, bn_bwd_finalize_kernel, dim3(vtx_cdiv(C, FIN_CH)), fin_block(rp.gx), 0, st, sums, gamma, save_rstd, coef,
                dgamma, dbeta, P, C, rp.gx);
    const long nvec = (long)NQ * C / vec;
    if (dtype == VTX_BF16)
        VTX_KLAUNCH("bn_bwd_apply", 0, 2.0 * el * P * C + pool_bytes, (pool_bn_bwd_apply_kernel<bf16_t>), dim3(apply_grid(nvec, C / vec)), dim3(256), 0, st, (const bf16_t*)x,
                    (const bf16_t*)dpool, argmax, save_mean, save_rstd, coef, gamma, beta, (bf16_t*)dx, N, H, W, C, OH, OW, g_vtx_sw_pool_xcd);
    else
        VTX_KLAUNCH("bn_bwd_apply", 0, 2.0 * el * P * C + pool_bytes, (pool_bn_bwd_apply_kernel<float>), dim3(apply_grid(nvec, C / vec)), dim3(256), 0, st, (const float*)x,
                    (const float*)dpool, argmax, save_mean, save_rstd, coef, gamma, beta, (float*)dx, N, H, W, C, OH, OW, g_vtx_sw_pool_xcd);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}

// measurement switch (tools/bench_bn_apply.py): vectors in flight per thread of the fused backward apply kernel
extern "C" int vtx_set_bn_apply_unroll(int n) {
    VTX_CHECK(n == 0 || n == 1 || n == 2 || n == 4, VTX_ERR_ARG, "bn apply unroll must be 0 (automatic), 1, 2 or 4");
    g_bn_apply_unroll = n;
    return VTX_OK;
}

// The stem's forward tail: BatchNorm (training statistics, running-statistics update) + ReLU + MaxPool2d(3,2,1) with the
// normalised tensor never written.  Replaces aten::batch_norm + relu_ + max_pool2d_with_indices of
// /root/reference/virtex/modules/visual_backbones.py:68-74 (torchvision ResNet: conv1 -> bn1 -> relu -> maxpool).
// x: [N][H][W][C] the stem convolution's output; pooled: [N][OH][OW][C]; argmax: uint8, the layout of vtx_maxpool3x3s2_fwd.
extern "C" int vtx_bn_fwd_maxpool(int dtype, const void* x, const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, long long* num_batches_tracked, void* pooled, uint8_t* argmax,
                                  float* save_mean, float* save_rstd, float* workspace, int N, int H, int W, int C, float eps,
                                  float momentum, const float* pre_partials, int pre_nparts, const float* pre_shift,
                                  void* stream) {
    VTX_CHECK(x && gamma && beta && pooled && argmax && save_mean && save_rstd && workspace, VTX_ERR_ARG, "bn_fwd_maxpool: null pointer");
    VTX_CHECK(dtype == VTX_BF16 || dtype == VTX_F32, VTX_ERR_DTYPE, "bn_fwd_maxpool: bad dtype %d", dtype);
    const int vec = dtype == VTX_BF16 ? 8 : 4;
    VTX_CHECK(N > 0 && H > 0 && W > 0 && bn_shape_ok(C, vec), VTX_ERR_SHAPE, "bn_fwd_maxpool: C=%d must be vec*2^k", C);
    VTX_CHECK((long)N * H * W < VTX_PIXEL_LIMIT, VTX_ERR_SHAPE, "bn_fwd_maxpool: more than 2^30 pixels is not supported");
    hipStream_t st = (hipStream_t)stream;
    int* tickets = reinterpret_cast<int*>(workspace); (void)tickets;
    workspace += VTX_BN_WS_HEADER;
    const int P = N * H * W, OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    float* scale = workspace; float* sums = workspace + 4 * C;
    ReducePlan rp = plan_reduce(P, C, vec);
    const bool fused = pre_partials != nullptr && pre_nparts > 0;   // statistics came with the convolution's epilogue
    VTX_CHECK(!fused || (pre_shift && pre_nparts <= 512), VTX_ERR_ARG, "bn_fwd_maxpool: fused statistics need their shift vector and <= 512 strips");
    if (fused) { sums = const_cast<float*>(pre_partials); rp.gx = pre_nparts; }
    else if (dtype == VTX_BF16)
        VTX_KLAUNCH("bn_fwd_reduce", 0, 2.0 * P * C, (bn_reduce_kernel<bf16_t, false>), dim3(rp.gx, rp.gy), dim3(256), 0, st, (const bf16_t*)x,
                    (const bf16_t*)nullptr, (const bf16_t*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, sums, P, C, rp.TX, rp.krows);
    else
        VTX_KLAUNCH("bn_fwd_reduce", 0, 4.0 * P * C, (bn_reduce_kernel<float, false>), dim3(rp.gx, rp.gy), dim3(256), 0, st, (const float*)x,
                    (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, sums, P, C, rp.TX, rp.krows);
    if (dtype == VTX_BF16)
        VTX_KLAUNCH("bn_finalize", 0, 8.0 * rp.gx * C, (bn_fwd_finalize_kernel<bf16_t>), dim3(vtx_cdiv(C, FIN_CH)), fin_block(rp.gx), 0, st, (const bf16_t*)x, fused ? pre_shift : (const float*)nullptr, sums, gamma, beta,
                    save_mean, save_rstd, scale, scale + C, running_mean, running_var, num_batches_tracked, P, C, eps, momentum, rp.gx);
    else
        VTX_KLAUNCH("bn_finalize", 0, 8.0 * rp.gx * C, (bn_fwd_finalize_kernel<float>), dim3(vtx_cdiv(C, FIN_CH)), fin_block(rp.gx), 0, st, (const float*)x, fused ? pre_shift : (const float*)nullptr, sums, gamma, beta,
                    save_mean, save_rstd, scale, scale + C, running_mean, running_var, num_batches_tracked, P, C, eps, momentum, rp.gx);
    const int cv = C / vec;
    int TX = 1; while (TX < cv && TX < 256) TX <<= 1;
    const int gy = (cv + TX - 1) / TX, TY = 256 / TX;
    long gx = ((long)N * OH * OW + TY - 1) / TY; gx = gx > 8192 ? 8192 : (gx < 1 ? 1 : gx);
    const double el = dtype == VTX_BF16 ? 2.0 : 4.0;
    if (dtype == VTX_BF16)
        VTX_KLAUNCH("bn_fwd_apply", 0, el * P * C + (el + 1.0) * N * OH * OW * C, (bn_relu_maxpool_fwd_kernel<bf16_t>), dim3((int)gx, gy), dim3(256), 0, st,
                    (const bf16_t*)x, save_mean, scale, beta, (bf16_t*)pooled, argmax, N, H, W, C, OH, OW, TX, g_vtx_sw_pool_xcd);
    else
        VTX_KLAUNCH("bn_fwd_apply", 0, el * P * C + (el + 1.0) * N * OH * OW * C, (bn_relu_maxpool_fwd_kernel<float>), dim3((int)gx, gy), dim3(256), 0, st,
                    (const float*)x, save_mean, scale, beta, (float*)pooled, argmax, N, H, W, C, OH, OW, TX, g_vtx_sw_pool_xcd);
    VTX_LAUNCH_CHECK();
    return VTX_OK;
}
