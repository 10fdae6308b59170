// Probe the gfx950 LDS transpose read: which source element lands in (lane, j)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint16_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    if (mode == 0) addr = l * 8;                                   // lane l -> elements 4l..4l+3
    else if (mode == 1) addr = ((l & 15) * 64 + (l >> 4) * 4) * 2; // 16 rows of 64 elems: lane -> row l&15, 4 elems at col (l>>4)*4
    else addr = ((l >> 2) * 128 + (l & 3) * 4) * 2;                // [k][128 rows]-like: lane -> k-row l>>2, 4 consecutive cols
    addr += (unsigned)(uintptr_t)lds;  // LDS base (address space 3 pointer value)
    uint64_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)(v >> (16 * j));
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
