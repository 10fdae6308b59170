// Probe: how fast does LDS-DMA (global_load_lds_dwordx4) stream L2-resident data into LDS, as a function of
// how many bytes of each 128-byte line one wave-instruction asks for?
//   mode 0: 16 rows x 64 B per instruction  (the BK=32 bf16 image of the contraction kernel)
//   mode 1:  8 rows x 128 B per instruction (full lines: a BK=64 image)
//   mode 2:  4 rows x 256 B per instruction
// Every block walks a row-major matrix with a 2 KiB row pitch (K=1024 bf16) that is shared by the whole chip
// (region = 4 MiB -> L2 resident), like the A panel of a GEMM.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ __launch_bounds__(512) void probe(const char* __restrict__ src, int iters, int rows_total, unsigned* sink) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int BPR = MODE == 0 ? 64 : (MODE == 1 ? 128 : 256);   // bytes per row per instruction
    constexpr int RPI = 1024 / BPR;                                  // rows per instruction
    constexpr int LPR = BPR / 16;                                    // lanes per row
    const int r_in = lane / LPR, c_in = (lane % LPR) * 16;
    // block's first row: spread blocks over the region
    int row0 = (blockIdx.x * 97) % (rows_total - 4096);
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        // one "tile": 256 rows; each of the 8 waves issues 256/(8*RPI) ... keep bytes per iteration fixed: 3 instr / wave
        const int kbyte = (it * BPR) & 2047;                         // walk along the row (K direction)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int row = row0 + (wave * 3 + i) * RPI + r_in;
            const char* p = src + (size_t)row * 2048 + ((kbyte + c_in) & 2047);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                             (__attribute__((address_space(3))) void*)(lds + ((it & 1) * 24 + wave * 3 + i) * 1024),
                                             16, 0, 0);
        }
        if ((it & 7) == 7) { __builtin_amdgcn_s_waitcnt(0x0070 | (0xF << 8)); }   // vmcnt(0): keep <= 24 in flight
        if (kbyte + BPR >= 2048) row0 = (row0 + 256) % (rows_total - 4096);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    acc += lds[threadIdx.x * 4];
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE> static void run(const char* src, int rows_total, unsigned* sink, int blocks_per_cu) {
    const int iters = 4096;
    dim3 grid(256 * blocks_per_cu), block(512);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe<MODE>, grid, block, 49152, 0, src, iters, rows_total, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(probe<MODE>, grid, block, 49152, 0, src, iters, rows_total, sink);
    hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)grid.x * iters * 24 * 1024;
    printf("mode %d (%3d B/row/instr) blocks/CU %d: %.1f us, %.2f TB/s, %.1f B/clk/CU @2.4GHz\n", MODE,
           MODE == 0 ? 64 : (MODE == 1 ? 128 : 256), blocks_per_cu, ms * 1e3, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
    const int rows_total = 2048 + 4096;           // 12 MiB region, walks stay inside
    char* src; unsigned* sink;
    hipMalloc(&src, (size_t)rows_total * 2048); hipMemset(src, 1, (size_t)rows_total * 2048);
    hipMalloc(&sink, 64);
    for (int bpc = 1; bpc <= 2; ++bpc) {
        run<0>(src, rows_total, sink, bpc);
        run<1>(src, rows_total, sink, bpc);
        run<2>(src, rows_total, sink, bpc);
    }
    return 0;
}
