// What does the shape of a streaming kernel cost on this part?  The BatchNorm apply passes (forward: read x, write y;
// fused backward: read x and dz, write dx) run at 5.3-5.4 TB/s in the step while the join epilogue of the contraction kernel
// reaches 6.4-6.5 and hipMemcpy 6.3.  Every variant moves the same tensors ([802816][256] bf16 = 411 MB each,
// 2 reads + 1 write = 1.23 GB, or 1 read + 1 write) with the arithmetic of the fused backward apply:
//   gs<U>     the library's form: grid of 4096 x 256 threads, grid-stride, U vectors per thread and trip one grid-stride apart
//   adj<U>    U vectors per thread and trip ADJACENT (256 vectors apart: the block owns a contiguous 256*U-vector chunk per trip)
//   chunk<U>  every block owns ONE contiguous range of the tensor (nvec / gridDim.x) and walks it adj<U>-wise
//   nt        the best of them with non-temporal stores
//   memcpy    hipMemcpyAsync device-to-device of one tensor (1 read + 1 write)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/stream_probe tools/probes/stream_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void unpack(uint4 w, float* f) {
    const uint32_t u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(u[i] << 16); f[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u); }
}
__device__ __forceinline__ uint32_t pk(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const bf2 r = __builtin_convertvector(f2{a, b}, bf2);
    return *reinterpret_cast<const uint32_t*>(&r);
}
template <bool NT> __device__ __forceinline__ void st(void* p, uint4 v) {
    if (NT) __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(p)); else *reinterpret_cast<uint4*>(p) = v;
}
// the arithmetic of bn_bwd_apply_fused_kernel: per-channel coefficients in registers (channel vector fixed per thread)
struct Coef { float mu[8], a0[8], a1[8], a2[8]; };
__device__ __forceinline__ Coef coef(const float* c, int c0) {
    Coef k;
#pragma unroll
    for (int j = 0; j < 8; ++j) { k.mu[j] = c[c0 + j]; k.a0[j] = c[256 + c0 + j]; k.a1[j] = c[512 + c0 + j]; k.a2[j] = c[768 + c0 + j]; }
    return k;
}
__device__ __forceinline__ uint4 apply(const Coef& k, uint4 xr, uint4 gr) {
    float x[8], g[8];
    unpack(xr, x); unpack(gr, g);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = k.a0[j] * g[j] - (x[j] - k.mu[j]) * k.a1[j] - k.a2[j];
    return make_uint4(pk(o[0], o[1]), pk(o[2], o[3]), pk(o[4], o[5]), pk(o[6], o[7]));
}
// READS = 2: x and dz; READS = 1: x only (forward apply)
template <int U, int READS, bool NT> __global__ __launch_bounds__(256) void gs(const uint4* x, const uint4* dz, uint4* dx, const float* c, long nvec) {
    const long t0 = (long)blockIdx.x * 256 + threadIdx.x, stride = (long)gridDim.x * 256;
    const Coef k = coef(c, (int)(t0 & 31) * 8);
    for (long i0 = t0; i0 < nvec; i0 += U * stride) {
        uint4 xv[U], gv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const long i = i0 + u * stride; if (i < nvec) { xv[u] = x[i]; gv[u] = READS == 2 ? dz[i] : xv[u]; } }
#pragma unroll
        for (int u = 0; u < U; ++u) { const long i = i0 + u * stride; if (i < nvec) st<NT>(dx + i, apply(k, xv[u], gv[u])); }
    }
}
template <int U, int READS, bool NT> __global__ __launch_bounds__(256) void adj(const uint4* x, const uint4* dz, uint4* dx, const float* c, long nvec) {
    const Coef k = coef(c, (int)(threadIdx.x & 31) * 8);
    const long stride = (long)gridDim.x * 256 * U;
    for (long b0 = (long)blockIdx.x * 256 * U; b0 < nvec; b0 += stride) {
        uint4 xv[U], gv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const long i = b0 + u * 256 + threadIdx.x; if (i < nvec) { xv[u] = x[i]; gv[u] = READS == 2 ? dz[i] : xv[u]; } }
#pragma unroll
        for (int u = 0; u < U; ++u) { const long i = b0 + u * 256 + threadIdx.x; if (i < nvec) st<NT>(dx + i, apply(k, xv[u], gv[u])); }
    }
}
template <int U, int READS, bool NT> __global__ __launch_bounds__(256) void chunk(const uint4* x, const uint4* dz, uint4* dx, const float* c, long nvec) {
    const Coef k = coef(c, (int)(threadIdx.x & 31) * 8);
    const long per = ((nvec + gridDim.x - 1) / gridDim.x + 256 * U - 1) / (256 * U) * (256 * U);
    const long lo = (long)blockIdx.x * per, hi = lo + per < nvec ? lo + per : nvec;
    for (long b0 = lo; b0 < hi; b0 += 256 * U) {
        uint4 xv[U], gv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const long i = b0 + u * 256 + threadIdx.x; if (i < hi) { xv[u] = x[i]; gv[u] = READS == 2 ? dz[i] : xv[u]; } }
#pragma unroll
        for (int u = 0; u < U; ++u) { const long i = b0 + u * 256 + threadIdx.x; if (i < hi) st<NT>(dx + i, apply(k, xv[u], gv[u])); }
    }
}
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)
template <class F> static float run(F f, int iters = 10) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / iters * 1e3f;
}
int main() {
    const long nvec = 802816L * 256 / 8;                       // 16-byte vectors per tensor (411 MB)
    uint4 *x, *dz, *dx; float* c;
    CK(hipMalloc(&x, nvec * 16)); CK(hipMalloc(&dz, nvec * 16)); CK(hipMalloc(&dx, nvec * 16)); CK(hipMalloc(&c, 4096));
    CK(hipMemset(x, 0x3c, nvec * 16)); CK(hipMemset(dz, 0x3d, nvec * 16)); CK(hipMemset(c, 0, 4096));
    const double b2 = 3.0 * nvec * 16, b1 = 2.0 * nvec * 16;
#define RUN(NAME, KERN, GRID, BYTES)                                                                   \
    { const float us = run([&] { hipLaunchKernelGGL(KERN, dim3(GRID), dim3(256), 0, 0, x, dz, dx, c, nvec); }); \
      printf("%-28s grid %5d  %7.1f us  %6.0f GB/s\n", NAME, (int)(GRID), us, BYTES / us * 1e-3); }
    printf("-- fused backward apply: 2 reads + 1 write (1.23 GB)\n");
    RUN("gs<1>", (gs<1, 2, false>), 4096, b2)  RUN("gs<2> (library)", (gs<2, 2, false>), 4096, b2)  RUN("gs<4>", (gs<4, 2, false>), 4096, b2)
    RUN("gs<2> grid 2048", (gs<2, 2, false>), 2048, b2)  RUN("gs<2> grid 8192", (gs<2, 2, false>), 8192, b2)  RUN("gs<2> grid 16384", (gs<2, 2, false>), 16384, b2)
    RUN("adj<2>", (adj<2, 2, false>), 4096, b2)  RUN("adj<4>", (adj<4, 2, false>), 4096, b2)  RUN("adj<4> grid 2048", (adj<4, 2, false>), 2048, b2)
    RUN("adj<4> grid 8192", (adj<4, 2, false>), 8192, b2)  RUN("adj<8> grid 2048", (adj<8, 2, false>), 2048, b2)
    RUN("chunk<2> grid 2048", (chunk<2, 2, false>), 2048, b2)  RUN("chunk<4> grid 2048", (chunk<4, 2, false>), 2048, b2)
    RUN("chunk<4> grid 4096", (chunk<4, 2, false>), 4096, b2)  RUN("chunk<4> grid 1024", (chunk<4, 2, false>), 1024, b2)
    RUN("gs<2> nt", (gs<2, 2, true>), 4096, b2)  RUN("adj<4> nt", (adj<4, 2, true>), 4096, b2)  RUN("chunk<4> nt grid 2048", (chunk<4, 2, true>), 2048, b2)
    printf("-- forward apply: 1 read + 1 write (0.82 GB)\n");
    RUN("gs<2> (library)", (gs<2, 1, false>), 4096, b1)  RUN("gs<4>", (gs<4, 1, false>), 4096, b1)  RUN("adj<4>", (adj<4, 1, false>), 4096, b1)
    RUN("adj<8> grid 2048", (adj<8, 1, false>), 2048, b1)  RUN("chunk<4> grid 2048", (chunk<4, 1, false>), 2048, b1)  RUN("adj<4> nt", (adj<4, 1, true>), 4096, b1)
    { const float us = run([&] { hipMemcpyAsync(dx, x, nvec * 16, hipMemcpyDeviceToDevice, 0); });
      printf("%-28s             %7.1f us  %6.0f GB/s\n", "hipMemcpyAsync D2D", us, b1 / us * 1e-3); }
    return 0;
}
