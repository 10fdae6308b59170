// What is the HBM WRITE ceiling of this part for the store patterns of the contraction epilogue?  (The write-heavy
// "expand" 1x1 convolutions -- 64->256 @56x56: 103 MB in, 411 MB out -- run at 3.3 TB/s of algorithmic traffic while
// read-heavy layers reach 4.5-6.5.)  Every variant writes the same 411 MB [802816][256] bf16 tensor once:
//   A  each wave-instruction writes 1 KiB contiguous (64 lanes x 16 B), waves walk the tensor linearly
//   B  the 8-wave 128x128 tile's pattern: a wave-instruction writes 8 rows x 128 B (row pitch 512 B); the block's two
//      wave columns cover 256 B of each row, the other half of the row belongs to another block
//   C  like B with four wave columns: a block covers whole 512-byte rows
//   each with ordinary and non-temporal stores; D = A with a 1:4 read:write mix (reads 103 MB)
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/write_probe tools/probes/write_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) {
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}
constexpr long ROWS = 802816, ROWB = 512;            // bytes per row
template <bool NT> __global__ __launch_bounds__(512) void pat_a(char* out, const char* in, int mix) {
    const long nchunk = ROWS * ROWB / 16, stride = (long)gridDim.x * 512;
    u32x4 v = {1u, 2u, 3u, (uint32_t)threadIdx.x};
    for (long i = (long)blockIdx.x * 512 + threadIdx.x; i < nchunk; i += stride) {
        if (mix && (i & 3) == 0) { const u32x4 r = *reinterpret_cast<const u32x4*>(in + (i >> 2) * 16); v[0] += r[0]; }
        st<NT>(reinterpret_cast<u32x4*>(out + i * 16), v);
    }
}
// block = 8 waves as 4 (rows) x WC (columns) ... generalised: WC wave columns of 128 B each, 8/WC wave rows of 16 rows
template <bool NT, int WC> __global__ __launch_bounds__(512) void pat_tile(char* out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wr = wave / WC, wc = wave % WC;
    constexpr int WR = 8 / WC, TROWS = 32 * WR;          // rows per block: each wave 32 rows (2 strips of 16)
    constexpr int CBLK = 512 / (128 * WC);               // column blocks per row
    u32x4 v = {1u, 2u, 3u, (uint32_t)threadIdx.x};
    const long ntile = ROWS / TROWS * CBLK;
    for (long t = blockIdx.x; t < ntile; t += gridDim.x) {
        const long tr = t / CBLK; const int tc = (int)(t % CBLK);
        for (int it = 0; it < 2; ++it)
            for (int q = 0; q < 2; ++q) {                 // 16 rows x 128 B = 2 wave-instructions of 8 rows
                const long row = tr * TROWS + wr * 32 + it * 16 + q * 8 + (lane >> 3);
                st<NT>(reinterpret_cast<u32x4*>(out + row * ROWB + (tc * WC + wc) * 128 + (lane & 7) * 16), v);
            }
    }
}
template <class F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
    char *out, *in; const size_t bytes = ROWS * ROWB;
    hipMalloc(&out, bytes); hipMalloc(&in, bytes / 4); hipMemset(in, 1, bytes / 4);
    const double gb = bytes / 1e9;
    for (int grid : {2048, 8192}) {
        printf("grid %d blocks of 512 threads\n", grid);
        float t;
        t = timeit([&] { pat_a<false><<<grid, 512>>>(out, in, 0); }); printf("  A linear 1 KiB/wave-instr          : %7.1f us  %5.2f TB/s\n", t * 1e3, gb / t);
        t = timeit([&] { pat_a<true><<<grid, 512>>>(out, in, 0); });  printf("  A non-temporal                      : %7.1f us  %5.2f TB/s\n", t * 1e3, gb / t);
        t = timeit([&] { pat_tile<false, 2><<<grid, 512>>>(out); });  printf("  B 8 rows x 128 B, half rows per block: %7.1f us  %5.2f TB/s\n", t * 1e3, gb / t);
        t = timeit([&] { pat_tile<true, 2><<<grid, 512>>>(out); });   printf("  B non-temporal                      : %7.1f us  %5.2f TB/s\n", t * 1e3, gb / t);
        t = timeit([&] { pat_tile<false, 4><<<grid, 512>>>(out); });  printf("  C 8 rows x 128 B, whole rows per block: %7.1f us  %5.2f TB/s\n", t * 1e3, gb / t);
        t = timeit([&] { pat_tile<true, 4><<<grid, 512>>>(out); });   printf("  C non-temporal                      : %7.1f us  %5.2f TB/s\n", t * 1e3, gb / t);
        t = timeit([&] { pat_a<false><<<grid, 512>>>(out, in, 1); }); printf("  D linear, 1:4 read:write (514 MB)   : %7.1f us  %5.2f TB/s\n", t * 1e3, 1.25 * gb / t);
        t = timeit([&] { pat_a<true><<<grid, 512>>>(out, in, 1); });  printf("  D non-temporal stores               : %7.1f us  %5.2f TB/s\n", t * 1e3, 1.25 * gb / t);
    }
    return 0;
}
