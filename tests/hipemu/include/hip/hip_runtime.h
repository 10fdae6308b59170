// hipemu -- a tiny CPU emulator of the HIP device programming model (TEST INFRASTRUCTURE).
//
// The build container has hipcc but no GPU.  To exercise the *actual kernel sources* of
// virtex_amd/csrc on CPU, the emulator build puts this directory in front of the include
// path so that `#include <hip/hip_runtime.h>` resolves here, and compiles the unchanged
// .hip files as host C++ (clang, for ext_vector_type).  Every HIP thread becomes a fiber;
// __syncthreads(), wave shuffles and MFMA are implemented as rendezvous points of the
// block's / wave's fibers, using the gfx950 lane<->element maps documented in
// /opt/skills/guides/cdna_hip_programming.md section 3.  The product never loads this.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>

#define HIPEMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#ifdef HIPEMU_STATIC_LDS
// sanitizer builds run with ONE worker thread (HIPEMU_THREADS=1): static LDS arrays become plain statics, which AddressSanitizer
// fences with red zones (it does not instrument thread-local storage) -- tools/build_emu_asan.sh
#define __shared__ static
#else
#define __shared__ static thread_local
#endif
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)hipemu::dyn_smem();

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
enum { hipMemcpyDeviceToDevice = 3 };

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_ { unsigned x, y, z; };

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct ushort4 { unsigned short x, y, z, w; };
inline float2 make_float2(float a, float b) { return {a, b}; }
inline float4 make_float4(float a, float b, float c, float d) { return {a, b, c, d}; }
inline uint2 make_uint2(unsigned a, unsigned b) { return {a, b}; }
inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return {a, b, c, d}; }
inline int4 make_int4(int a, int b, int c, int d) { return {a, b, c, d}; }

namespace hipemu {
struct Fiber;
struct ThreadCtx {           // one per OS worker thread
    Fiber* cur = nullptr;
    uint3_ bid{}, bdim{}, gdim{};
    char* dyn = nullptr;
};
extern thread_local ThreadCtx tls;
// LDS-DMA issued but not yet "landed" (only in the late-landing mode, see hipemu_set_dma_late)
struct PendingDma { void* dst; int size; char data[16]; };
struct Fiber {
    uint3_ tid;
    int linear, wave, lane;
    // late-landing LDS-DMA: what this lane has issued and no s_waitcnt vmcnt has retired yet (oldest first)
    PendingDma pend[64];
    int npend = 0;
    // scheduler state lives in the runtime
};
// Late-landing mode (tests of the counted-vmcnt pipelines): an LDS-DMA writes its LDS bytes only when the ISSUING lane
// executes an s_waitcnt whose vmcnt leaves it out (or __syncthreads, which hipcc fences with vmcnt(0) while a DMA is in
// flight) -- the LATEST moment the hardware allows.  A kernel that reads a staged tile before wait + barrier then sees
// the old bytes and fails its test; the default (0) mode lands every DMA at issue, the EARLIEST moment, which exposes
// a stage issued before the last read of the bytes it overwrites.  Unretired DMAs of a finished lane are dropped.
extern int g_dma_late;
inline void dma_retire(int leave) {
    Fiber* f = tls.cur;
    if (f->npend <= leave) return;
    const int n = f->npend - leave;
    for (int i = 0; i < n; ++i) memcpy(f->pend[i].dst, f->pend[i].data, f->pend[i].size);
    for (int i = n; i < f->npend; ++i) f->pend[i - n] = f->pend[i];
    f->npend = leave;
}
inline void dma_write(void* dst, const void* src16, int size) {
    if (!g_dma_late) { memcpy(dst, src16, size); return; }
    Fiber* f = tls.cur;
    if (f->npend >= 64) { fprintf(stderr, "hipemu: more than 64 LDS-DMA in flight in one lane (vmcnt is 6 bits)\n"); abort(); }
    PendingDma& p = f->pend[f->npend++];
    p.dst = dst; p.size = size; memcpy(p.data, src16, size);
}
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void block_barrier();
void* wave_exchange(const void* mine, size_t bytes);  // returns base of 64 slots (stride 64 B)
inline char* dyn_smem() { return tls.dyn; }
const uint3_& cur_tid();
}  // namespace hipemu

#define threadIdx (hipemu::cur_tid())
#define blockIdx (hipemu::tls.bid)
#define blockDim (hipemu::tls.bdim)
#define gridDim (hipemu::tls.gdim)
#define warpSize 64

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch((grid), (block), (shmem), [=]() { kernel(__VA_ARGS__); })

inline void __syncthreads() { hipemu::dma_retire(0); hipemu::block_barrier(); }
inline int __lane_id() { return hipemu::tls.cur->lane; }

// ---- wave cross-lane ops -------------------------------------------------------------
template <class T> inline T __shfl(T v, int src, int width = 64) {
    static_assert(sizeof(T) <= 64, "");
    int lane = hipemu::tls.cur->lane;
    char* base = (char*)hipemu::wave_exchange(&v, sizeof(T));
    int s = (lane & ~(width - 1)) + (src & (width - 1));
    T r; memcpy(&r, base + 64 * s, sizeof(T)); return r;
}
template <class T> inline T __shfl_xor(T v, int mask, int width = 64) {
    int lane = hipemu::tls.cur->lane;
    char* base = (char*)hipemu::wave_exchange(&v, sizeof(T));
    int s = lane ^ mask;
    if ((s & ~(width - 1)) != (lane & ~(width - 1))) s = lane;
    T r; memcpy(&r, base + 64 * s, sizeof(T)); return r;
}
template <class T> inline T __shfl_down(T v, unsigned delta, int width = 64) {
    int lane = hipemu::tls.cur->lane;
    char* base = (char*)hipemu::wave_exchange(&v, sizeof(T));
    int s = lane + (int)delta;
    if ((s & ~(width - 1)) != (lane & ~(width - 1))) s = lane;
    T r; memcpy(&r, base + 64 * s, sizeof(T)); return r;
}
inline unsigned long long __ballot(int pred) {
    int p = pred != 0;
    char* base = (char*)hipemu::wave_exchange(&p, sizeof(int));
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) { int q; memcpy(&q, base + 64 * l, 4); if (q) m |= 1ull << l; }
    return m;
}

// ---- MFMA (gfx950 lane<->element maps, guide section 3) -------------------------------
typedef short hipemu_bf16x8 __attribute__((ext_vector_type(8)));
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));
inline float hipemu_bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

// D(16x16) = A(16x32) B(32x16) + C.  lane l holds A[l&15][8*(l>>4)+j], B[8*(l>>4)+j][l&15];
// C/D reg i of lane l = element [4*(l>>4)+i][l&15].
inline hipemu_f32x4 hipemu_mfma_16x16x32_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x4 c) {
    struct { hipemu_bf16x8 a, b; } mine{a, b};
    char* base = (char*)hipemu::wave_exchange(&mine, sizeof(mine));
    int lane = hipemu::tls.cur->lane, col = lane & 15;
    hipemu_f32x4 d = c;
    for (int i = 0; i < 4; ++i) {
        int row = 4 * (lane >> 4) + i;
        float acc = c[i];
        for (int k = 0; k < 32; ++k) {
            unsigned short av, bv;
            memcpy(&av, base + 64 * (row + 16 * (k >> 3)) + 2 * (k & 7), 2);
            memcpy(&bv, base + 64 * (col + 16 * (k >> 3)) + 16 + 2 * (k & 7), 2);
            acc += hipemu_bf2f(av) * hipemu_bf2f(bv);
        }
        d[i] = acc;
    }
    return d;
}
// D(32x32) = A(32x16) B(16x32) + C. lane l: A[l&31][8*(l>>5)+j], B[8*(l>>5)+j][l&31];
// C/D reg r of lane l = [(r&3)+8*(r>>2)+4*(l>>5)][l&31].
inline hipemu_f32x16 hipemu_mfma_32x32x16_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x16 c) {
    struct { hipemu_bf16x8 a, b; } mine{a, b};
    char* base = (char*)hipemu::wave_exchange(&mine, sizeof(mine));
    int lane = hipemu::tls.cur->lane, col = lane & 31;
    hipemu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            unsigned short av, bv;
            memcpy(&av, base + 64 * (row + 32 * (k >> 3)) + 2 * (k & 7), 2);
            memcpy(&bv, base + 64 * (col + 32 * (k >> 3)) + 16 + 2 * (k & 7), 2);
            acc += hipemu_bf2f(av) * hipemu_bf2f(bv);
        }
        d[r] = acc;
    }
    return d;
}
// f32 16x16x4: lane l holds A[l&15][l>>4], B[l>>4][l&15].
inline hipemu_f32x4 hipemu_mfma_16x16x4_f32(float a, float b, hipemu_f32x4 c) {
    struct { float a, b; } mine{a, b};
    char* base = (char*)hipemu::wave_exchange(&mine, sizeof(mine));
    int lane = hipemu::tls.cur->lane, col = lane & 15;
    hipemu_f32x4 d = c;
    for (int i = 0; i < 4; ++i) {
        int row = 4 * (lane >> 4) + i;
        float acc = c[i];
        for (int k = 0; k < 4; ++k) {
            float av, bv;
            memcpy(&av, base + 64 * (row + 16 * k), 4);
            memcpy(&bv, base + 64 * (col + 16 * k) + 4, 4);
            acc = fmaf(av, bv, acc);
        }
        d[i] = acc;
    }
    return d;
}
// f32 32x32x2: lane l holds A[l&31][l>>5], B[l>>5][l&31].
inline hipemu_f32x16 hipemu_mfma_32x32x2_f32(float a, float b, hipemu_f32x16 c) {
    struct { float a, b; } mine{a, b};
    char* base = (char*)hipemu::wave_exchange(&mine, sizeof(mine));
    int lane = hipemu::tls.cur->lane, col = lane & 31;
    hipemu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av, bv;
            memcpy(&av, base + 64 * (row + 32 * k), 4);
            memcpy(&bv, base + 64 * (col + 32 * k) + 4, 4);
            acc = fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) hipemu_mfma_16x16x32_bf16(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) hipemu_mfma_32x32x16_bf16(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hipemu_mfma_16x16x4_f32(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) hipemu_mfma_32x32x2_f32(a, b, c)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
/* gfx9 s_waitcnt immediate: vmcnt = bits [3:0] | [15:14] << 4.  Only LDS-DMA is modelled (late-landing mode). */
#define __builtin_amdgcn_s_waitcnt(x) hipemu::dma_retire((int)(((x) & 0xF) | ((((x) >> 14) & 3) << 4)))
#define __builtin_amdgcn_s_barrier() hipemu::block_barrier()
/* wave_barrier is only a scheduling fence on hardware (a wave runs in lockstep); the fibres of a wave do not */
#define __builtin_amdgcn_wave_barrier() do { int z_ = 0; (void)hipemu::wave_exchange(&z_, sizeof(z_)); } while (0)
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
enum { hipDeviceAttributeMaxSharedMemoryPerBlock = 1 };
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, int attr, int) { *v = attr == hipDeviceAttributeMaxSharedMemoryPerBlock ? 160 * 1024 : 0; return hipSuccess; }   /* the emulated part is gfx950 */
// events: the emulator runs launches synchronously, so an event is just a host timestamp
#include <chrono>
struct hipemuEvent { double t; };
typedef hipemuEvent* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemuEvent{0.0}; return hipSuccess; }
enum { hipEventReleaseToDevice = 0x40000000 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
    e->t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    return hipSuccess;
}
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)((b->t - a->t) * 1e3); return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
#define hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, e0, e1, flags, ...) \
    do { hipEventRecord((e0), (stream)); hipemu::launch((grid), (block), (shmem), [=]() { kernel(__VA_ARGS__); }); \
         hipEventRecord((e1), (stream)); } while (0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))

// ---- LDS-DMA and LDS transpose read (gfx950) ---------------------------------------------
// global_load_lds: lane l copies `size` bytes from its own global address to lds_base + l*size.
inline void hipemu_global_load_lds(const void* src, void* lds_base, int size, int offset) {
    hipemu::dma_write((char*)lds_base + offset + hipemu::tls.cur->lane * size, src, size);
}
#define __builtin_amdgcn_global_load_lds(src, dst, size, off, aux) \
    hipemu_global_load_lds((const void*)(uintptr_t)(src), (void*)(uintptr_t)(dst), (size), (off))
// buffer_load_dwordx4 ... offen lds through a raw buffer descriptor (stride 0): lane l copies `size` bytes from
// base + voffset + soffset + imm to lds_base + l*size; every dword whose (voffset + imm) offset lies outside
// [0, num_records) reads as zero (the range check ignores soffset, like the hardware's).  A lane that passes the range
// check but whose full address leaves the window is a kernel bug: abort with a diagnostic.
struct hipemu_rsrc { const char* base; uint32_t bytes; };
typedef hipemu_rsrc __amdgpu_buffer_rsrc_t;
inline hipemu_rsrc hipemu_make_rsrc(const void* p, int num) { return hipemu_rsrc{(const char*)p, (uint32_t)num}; }
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, num, flags) hipemu_make_rsrc((p), (num))
inline void hipemu_buffer_load_lds(hipemu_rsrc r, void* lds_base, int size, unsigned voff, unsigned soff, int imm) {
    char* const dst_lds = (char*)lds_base + hipemu::tls.cur->lane * size;
    char tmp[16];
    char* dst = tmp;
    if (size > 16) abort();
    for (int d = 0; d < size; d += 4) {
        const unsigned long long o = (unsigned long long)voff + (unsigned)imm + (unsigned)d;
        if (o + 4 > r.bytes) { memset(dst + d, 0, 4); continue; }
        if (o + soff + 4 > r.bytes) {
            fprintf(stderr, "hipemu: buffer load in range by voffset (%u) but voffset+soffset (%u) leaves the %u-byte window\n",
                    voff, soff, r.bytes);
            abort();
        }
        memcpy(dst + d, r.base + o + soff, 4);
    }
    hipemu::dma_write(dst_lds, tmp, size);
}
#define __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, size, voff, soff, imm, aux) \
    hipemu_buffer_load_lds((rsrc), (void*)(uintptr_t)(dst), (size), (unsigned)(voff), (unsigned)(soff), (imm))
// ds_read_b64_tr_b16 (semantics measured on MI355X, tools/probes/tr_probe.hip): within each 16-lane
// group, lane i receives for j = 0..3 element (i % 4) of the 8 bytes addressed by lane 4*j + i/4.
typedef short hipemu_v4s __attribute__((ext_vector_type(4)));
inline hipemu_v4s hipemu_ds_read_tr16_b64(const void* p) {
    unsigned short mine[4]; memcpy(mine, p, 8);
    char* base = (char*)hipemu::wave_exchange(mine, 8);
    const int lane = hipemu::tls.cur->lane, gb = lane & ~15, i = lane & 15;
    hipemu_v4s r;
    for (int j = 0; j < 4; ++j) { unsigned short v; memcpy(&v, base + 64 * (gb + 4 * j + i / 4) + 2 * (i % 4), 2); r[j] = (short)v; }
    return r;
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) hipemu_ds_read_tr16_b64((const void*)(uintptr_t)(p))

// v_permlane16_swap_b32 / v_permlane32_swap_b32 (gfx950; semantics measured on MI355X, tools/probes/permlane_probe.hip):
//   permlane16_swap(a, b): r[0] = rows {a0, b0, a2, b2}, r[1] = rows {a1, b1, a3, b3}   (rows of 16 lanes: a.row1 <-> b.row0, a.row3 <-> b.row2)
//   permlane32_swap(a, b): r[0] = {a.lanes 0-31, b.lanes 0-31}, r[1] = {a.lanes 32-63, b.lanes 32-63}   (a.upper <-> b.lower)
typedef unsigned hipemu_u32x2 __attribute__((ext_vector_type(2)));
inline hipemu_u32x2 hipemu_permlane_swap(unsigned a, unsigned b, int half) {
    unsigned mine[2] = {a, b};
    char* base = (char*)hipemu::wave_exchange(mine, 8);
    const int lane = hipemu::tls.cur->lane;
    const bool upper = (lane & half) != 0;              // odd row (half = 16) / upper half of the wave (half = 32)
    unsigned pa, pb;
    memcpy(&pa, base + 64 * (lane ^ half), 4);
    memcpy(&pb, base + 64 * (lane ^ half) + 4, 4);
    hipemu_u32x2 r;
    r[0] = upper ? pb : a;                              // r[0]: a where it stays, the partner's b in the swapped positions
    r[1] = upper ? b : pa;
    return r;
}
#define __builtin_amdgcn_permlane16_swap(a, b, fi, bc) hipemu_permlane_swap((unsigned)(a), (unsigned)(b), 16)
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) hipemu_permlane_swap((unsigned)(a), (unsigned)(b), 32)

// ---- atomics (blocks run on several OS threads) ---------------------------------------
inline float atomicAdd(float* p, float v) {
    unsigned* ip = (unsigned*)p; unsigned old = __atomic_load_n(ip, __ATOMIC_RELAXED), nw;
    float f;
    do { memcpy(&f, &old, 4); f += v; memcpy(&nw, &f, 4); }
    while (!__atomic_compare_exchange_n(ip, &old, nw, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    memcpy(&f, &old, 4); return f;
}
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline float atomicMax(float* p, float v) {  // not in HIP for float; helper for tests only
    unsigned* ip = (unsigned*)p; unsigned old = __atomic_load_n(ip, __ATOMIC_RELAXED), nw; float f;
    do { memcpy(&f, &old, 4); if (f >= v) break; memcpy(&nw, &v, 4); }
    while (!__atomic_compare_exchange_n(ip, &old, nw, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return f;
}
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

// ---- math -----------------------------------------------------------------------------
inline float __expf(float x) { return expf(x); }
inline float __logf(float x) { return logf(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
template <class T> inline T min(T a, T b) { return a < b ? a : b; }
template <class T> inline T max(T a, T b) { return a > b ? a : b; }
