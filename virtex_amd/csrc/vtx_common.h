// Shared device/host helpers for the virtex_amd HIP kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/virtex_amd.h"

// ---------------------------------------------------------------------------------------
// Error plumbing for the C ABI: every entry point returns 0 or a negative vtx_status and
// records a thread-local message (backward arrives on autograd's thread -> no globals).
// ---------------------------------------------------------------------------------------
void vtx_set_error(const char* fmt, ...);
#define VTX_CHECK(cond, code, ...)            \
    do {                                      \
        if (!(cond)) {                        \
            vtx_set_error(__VA_ARGS__);       \
            return (code);                    \
        }                                     \
    } while (0)
#define VTX_LAUNCH_CHECK()                                                     \
    do {                                                                       \
        hipError_t e_ = hipGetLastError();                                     \
        if (e_ != hipSuccess) {                                                \
            vtx_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,       \
                          hipGetErrorString(e_));                              \
            return VTX_ERR_LAUNCH;                                             \
        }                                                                      \
    } while (0)

static inline int vtx_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------
// floor(n / d) for 0 <= n < 2^30, 1 <= d < 2^24, through a float reciprocal `inv` = 1 / d good to 2 ulp (v_rcp_f32, or a
// correctly rounded host value): what every pixel-index decomposition of the library uses (block prologues, scattering
// epilogues, pooling) -- a generic 32-bit division is ~25 instructions, this is 13.  (float)n keeps 24 bits, so the first
// estimate is off by up to n 2^-21 / d + 1 (<= 725 for n < 2^30); its remainder is small enough to be exact in float, the second
// estimate (on the remainder) lands within 1 of the quotient, one correction step finishes.  Rounds 1-5 had the one-estimate
// form, exact below 2^24 only: tensors of more than 16.7 M pixels (334 images of 224 x 224 per GPU) were refused.
// VTX_PIXEL_LIMIT is what the host entry points check.
// ---------------------------------------------------------------------------------------
constexpr long VTX_PIXEL_LIMIT = 1L << 30;
// Workgroups are dealt to the eight XCDs round-robin (XCD = block index & 7) and every XCD has an L2 of its own.  A kernel whose
// NEIGHBOURING blocks read overlapping data (pooling windows, convolution strips) should therefore give each XCD a CONTIGUOUS
// range of its work: logical block = (b & 7) * (g / 8) + (b >> 3) for a grid of g blocks (g % 8 == 0; otherwise the plain order).
// Measured on the stem's streaming kernel: 370 -> 118 MB fetched per launch for a 108-MB input (profiles/r06_stem_xcd_order.txt).
#if defined(__HIPCC__) || defined(HIPEMU)
__device__ __forceinline__ int vtx_xcd_major_block(int b, int g) { return (g & 7) ? b : (b & 7) * (g >> 3) + (b >> 3); }
#endif
#if defined(__HIPCC__) || defined(HIPEMU)
__device__ __forceinline__ int vtx_fdiv30(int n, int d, float inv) {
    int q = (int)((float)n * inv);
    int r = n - q * d;
    q += (int)((float)r * inv);
    r = n - q * d;
    if (r < 0) --q; else if (r >= d) ++q;
    return q;
}
#endif

// ---------------------------------------------------------------------------------------
// Optional per-launch timing (vtx_profile_start / vtx_profile_stop in core.hip): while profiling is on a launch
// carries a begin and an end HIP event (hipExtLaunchKernel: the dispatch's own timestamps, what rocprofv3 reports)
// and its algorithmic FLOPs / HBM bytes are summed per class.  Contraction kernels register one class per
// template instantiation (gemm_kernel.h); every other kernel goes through VTX_KLAUNCH with a family name -- one class per
// launch site AND instantiation of the enclosing launcher, registered as "family:<name>|<kernel expression>|[T = ..., ...]"
// so that bench.py can name the single kernel instantiation a class stands for (its headline is the largest one).
// ---------------------------------------------------------------------------------------
namespace vtxg {
extern int g_vtx_prof_on, g_vtx_prof_only;
int vtx_prof_register(const char* pretty_name);
int vtx_prof_register_family(const char* family, const char* kernel_expr, const char* enclosing_pretty);
void vtx_prof_events(int cls, double flops, double bytes, hipEvent_t* start, hipEvent_t* stop);
}
#define VTX_KLAUNCH(fam, flops_, bytes_, kern, grid, block, shmem, st, ...)                                      \
    do {                                                                                                         \
        bool vtx_done_ = false;                                                                                  \
        if (vtxg::g_vtx_prof_on) {                                                                               \
            static const int vtx_cls_ = vtxg::vtx_prof_register_family("family:" fam, #kern, __PRETTY_FUNCTION__);  \
            if (vtxg::g_vtx_prof_only < 0 || vtxg::g_vtx_prof_only == vtx_cls_) {                                \
                hipEvent_t vtx_e0_, vtx_e1_;                                                                     \
                vtxg::vtx_prof_events(vtx_cls_, (double)(flops_), (double)(bytes_), &vtx_e0_, &vtx_e1_);         \
                hipExtLaunchKernelGGL(kern, grid, block, shmem, st, vtx_e0_, vtx_e1_, 0, __VA_ARGS__);           \
                vtx_done_ = true;                                                                                \
            }                                                                                                    \
        }                                                                                                        \
        if (!vtx_done_) hipLaunchKernelGGL(kern, grid, block, shmem, st, __VA_ARGS__);                           \
    } while (0)

// ---------------------------------------------------------------------------------------
// bf16 stored as raw uint16 (round-to-nearest-even), fp32 math everywhere.
// ---------------------------------------------------------------------------------------
typedef uint16_t bf16_t;
typedef short bf16x8_t __attribute__((ext_vector_type(8)));   // one 16x16x32 MFMA operand
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// fp32 -> bf16, round to nearest even.  gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32: two values per
// instruction where the integer sequence below costs six each) -- every bf16-writing epilogue and the BatchNorm /
// LayerNorm apply passes go through these two functions.  The CPU emulator build keeps the integer form (same
// rounding; NaNs are quieted by both).
#ifdef HIPEMU
__device__ __forceinline__ bf16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }
#else
typedef __bf16 vtx_bf16x2_hw __attribute__((ext_vector_type(2)));
typedef float vtx_f32x2_hw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((vtx_f32x2_hw){lo, hi}, vtx_bf16x2_hw));
}
#endif

// ---------------------------------------------------------------------------------------
// ds_read_b64_tr_b16 (the transposing LDS read that feeds MFMA fragments from k-major images) in kernels that also have
// LDS-DMA (buffer_load ... lds) in flight.  hipcc (ROCm 7.2) treats the BUILTIN as a read of unknown LDS memory and puts
// `s_waitcnt vmcnt(0)` in front of the first one after every DMA issue: the whole prefetch pipeline is drained once per K
// step (found in round 3 in every k-major instantiation of the contraction kernel: the weight gradients).  An asm
// statement is invisible to that pass; the price is that hipcc does not count it either -- the caller issues all reads
// of a step into SEPARATE result variables, then vtx_ds_tr_wait() (s_waitcnt lgkmcnt(0) + a scheduling barrier), and only
// then touches the results (cdna_hip_programming.md 5.7 form (iii), rule 18).  The emulator build keeps the builtin.
// ---------------------------------------------------------------------------------------
typedef short vtx_v4s_t __attribute__((ext_vector_type(4)));
#if defined(HIPEMU) || defined(VTX_TR_BUILTIN)      // VTX_TR_BUILTIN: A/B builds only (the drained pipeline of rounds 1-2)
__device__ __forceinline__ vtx_v4s_t vtx_ds_read_tr16(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) vtx_v4s_t*)p);
}
// the same from a byte offset inside the kernel's dynamic LDS (which starts at LDS address 0 when the kernel has no static
// __shared__ object: the offset IS the address -- no per-read "+ base" instruction)
__device__ __forceinline__ vtx_v4s_t vtx_ds_read_tr16_at(const char* dyn_lds, uint32_t off) { return vtx_ds_read_tr16(dyn_lds + off); }
// ... and with a compile-time byte offset on top (the instruction's 16-bit offset field: no address arithmetic per read)
template <int IMM> __device__ __forceinline__ vtx_v4s_t vtx_ds_read_tr16_imm(const char* dyn_lds, uint32_t off) { return vtx_ds_read_tr16(dyn_lds + off + IMM); }
__device__ __forceinline__ void vtx_ds_tr_wait() {}
template <int N> __device__ __forceinline__ void vtx_ds_tr_wait_n() {}
#else
__device__ __forceinline__ vtx_v4s_t vtx_ds_read_tr16(const void* p) {
    vtx_v4s_t r;
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(a) : "memory");
    return r;
}
__device__ __forceinline__ vtx_v4s_t vtx_ds_read_tr16_at(const char*, uint32_t off) {
    vtx_v4s_t r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(off) : "memory");
    return r;
}
template <int IMM> __device__ __forceinline__ vtx_v4s_t vtx_ds_read_tr16_imm(const char*, uint32_t off) {
    static_assert(IMM >= 0 && IMM < 65536, "ds offset field is 16 bits");
    vtx_v4s_t r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(off), "n"(IMM) : "memory");
    return r;
}
__device__ __forceinline__ void vtx_ds_tr_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// all but the N most recently issued LDS operations of this wave have returned (LDS returns in order)
template <int N> __device__ __forceinline__ void vtx_ds_tr_wait_n() {
    static_assert(N >= 0 && N <= 15, "lgkmcnt is 4 bits");
#ifdef VTX_TR_NOLADDER                  // A/B builds: one full wait per K step
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
#endif
    __builtin_amdgcn_sched_barrier(0);
}
#endif

template <class T> struct Elem;
template <> struct Elem<float> {
    static constexpr int VEC = 4;  // elements per 16-byte access
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// 16-byte vector of T unpacked to / packed from fp32 registers.
template <class T> struct Vec16;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
// streaming (non-temporal) 16-byte accesses.  Measured on the BatchNorm apply kernels: SLOWER (43.9 vs 42.2
// ms/step) -- the producer's output is still in the Infinity Cache when the next kernel reads it, and nt
// bypasses it.  Kept for kernels whose inputs really are cold.
__device__ __forceinline__ u32x4_t ld16_nt(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p)); }
__device__ __forceinline__ void st16_nt(void* p, u32x4_t v) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(p)); }

template <> struct Vec16<float> {
    float v[4];
    __device__ __forceinline__ void load(const float* p) {
        float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    __device__ __forceinline__ void load_nt(const float* p) {
        const u32x4_t t = ld16_nt(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __uint_as_float(t[i]);
    }
    __device__ __forceinline__ void store(float* p) const {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __device__ __forceinline__ void store_nt(float* p) const {
        st16_nt(p, u32x4_t{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])});
    }
};
template <> struct Vec16<bf16_t> {
    float v[8];
    __device__ __forceinline__ void load(const bf16_t* p) {
        uint4 t = *reinterpret_cast<const uint4*>(p);
        uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    __device__ __forceinline__ void load_nt(const bf16_t* p) {
        const u32x4_t t = ld16_nt(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(t[i] << 16);
            v[2 * i + 1] = __uint_as_float(t[i] & 0xffff0000u);
        }
    }
    __device__ __forceinline__ void store(bf16_t* p) const {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = f2bf2(v[2 * i], v[2 * i + 1]);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __device__ __forceinline__ void store_nt(bf16_t* p) const {
        u32x4_t w;
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = f2bf2(v[2 * i], v[2 * i + 1]);
        st16_nt(p, w);
    }
};

// Raw 16-byte vectors: kernels that must have SEVERAL loads in flight request them all as uint4 (address clamped where the
// element does not exist), call vtx_loads_issued(), and only then unpack / mask.  Written as `if (exists) { load; use; }` hipcc
// (ROCm 7.2) puts an s_waitcnt vmcnt(0) inside every branch: a chain of memory latencies per thread (round 3: pooling,
// LayerNorm, embedding, the contraction epilogue).  vtx_loads_issued keeps the scheduler from sinking the requested loads back between their uses (it does, to save registers)
__device__ __forceinline__ void vtx_loads_issued() {
#ifndef HIPEMU
    __builtin_amdgcn_sched_barrier(0);
#endif
}
template <class T> __device__ __forceinline__ void vtx_unpack_raw16(uint4 w, float* f);
template <> __device__ __forceinline__ void vtx_unpack_raw16<bf16_t>(uint4 w, float* f) {
    const uint32_t u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(u[i] << 16); f[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u); }
}
template <> __device__ __forceinline__ void vtx_unpack_raw16<float>(uint4 w, float* f) {
    f[0] = __uint_as_float(w.x); f[1] = __uint_as_float(w.y); f[2] = __uint_as_float(w.z); f[3] = __uint_as_float(w.w);
}

// ---------------------------------------------------------------------------------------
// wave64 / block reductions
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}
// Sum over a block of NW waves; every thread gets the result.  `red` = NW floats of LDS.
template <int NW> __device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += red[i];
    return t;
}
template <int NW> __device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
    return t;
}

// ---------------------------------------------------------------------------------------
// Counter-based dropout RNG: keep-mask = f(seed, element index) recomputed in backward, so
// no mask tensor ever touches HBM.  (Bit parity with torch's Philox stream is a non-goal:
// SURVEY.md 7.3-8; parity runs use p = 0.)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t vtx_hash32(uint32_t seed, uint32_t idx_lo, uint32_t idx_hi) {
    uint32_t x = idx_lo * 0x9E3779B1u + (idx_hi ^ seed) * 0x85EBCA77u + seed;
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16; x += seed * 0xC2B2AE3Du;
    x ^= x >> 15; x *= 0x2c1b3c6du;
    x ^= x >> 12;
    return x;
}
// `epoch` (optional, device memory; vtx_set_dropout_epoch): a step counter mixed into the seed ON THE DEVICE, so that a
// captured hipGraph of the training step -- whose by-value seeds are frozen at capture time -- still draws new masks
// every replay (the forward and backward launches of one step read the same value; the optimizer step increments it).
// A kernel resolves it once at entry (`drop = drop.resolved()`), never per element.
struct Dropout {
    uint32_t seed;
    uint32_t thresh;  // keep iff hash >= thresh ; thresh = p * 2^32
    float scale;      // 1/(1-p)
    const uint32_t* epoch;
    __device__ __forceinline__ float apply(float v, uint64_t idx) const {
        if (thresh == 0u) return v;
        return vtx_hash32(seed, (uint32_t)idx, (uint32_t)(idx >> 32)) >= thresh ? v * scale : 0.f;
    }
    __device__ __forceinline__ Dropout resolved() const {
        Dropout d = *this;
        if (thresh != 0u && epoch) d.seed = seed + *epoch * 0x9E3779B1u;
        d.epoch = nullptr;
        return d;
    }
};
extern const uint32_t* g_vtx_dropout_epoch;      // core.hip
static inline Dropout make_dropout(float p, uint64_t seed) {
    Dropout d;
    d.epoch = g_vtx_dropout_epoch;
    d.seed = (uint32_t)(seed * 0x9E3779B97F4A7C15ull >> 32) ^ (uint32_t)seed;
    if (p <= 0.f) { d.thresh = 0u; d.scale = 1.f; }
    else {
        double t = (double)p * 4294967296.0;
        d.thresh = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
        d.scale = 1.f / (1.f - p);
    }
    return d;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
