// Probe of `buffer_load_dwordx4 ... offen lds` (raw buffer descriptor, stride 0) on gfx950: what the generation-2
// contraction kernel relies on.  Prints one line per question:
//   1. in-range lanes copy base + voffset + soffset, 16 bytes per lane, lane-linear into LDS
//   2. lanes with voffset = 0x80000000 write ZEROS into LDS (the hardware range check, not a skipped write)
//   3. is soffset part of the range check?  (voffset in range, voffset + soffset beyond num_records)
//   4. a descriptor base BEFORE the allocation with in-range final addresses is fine
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/buf_probe tools/probes/buf_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__global__ void probe(const uint32_t* src, int num_bytes, int base_shift_bytes, const int* voff, int soff, uint32_t* out) {
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 64 * 4; i += 64) lds[i] = 0xDEADBEEFu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)src - base_shift_bytes), (short)0,
                                                                  num_bytes + base_shift_bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff[threadIdx.x] + base_shift_bytes, soff, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 4; i += 64) out[i] = lds[i];
}

int main() {
    const int N = 4096;                                    // dwords
    std::vector<uint32_t> h(N);
    for (int i = 0; i < N; ++i) h[i] = 0x10000000u + i;
    uint32_t *d, *o; int* dv;
    hipMalloc(&d, 2 * N * 4); hipMalloc(&o, 256 * 4); hipMalloc(&dv, 64 * 4);
    hipMemcpy(d + N, h.data(), N * 4, hipMemcpyHostToDevice);       // the "tensor" lives in the second half
    hipMemset(d, 0x55, N * 4);
    const uint32_t* t = d + N;
    std::vector<int> v(64); std::vector<uint32_t> r(256);
    auto run = [&](int bytes, int shift, int soff) {
        hipMemcpy(dv, v.data(), 64 * 4, hipMemcpyHostToDevice);
        probe<<<1, 64, 64 * 16>>>(t, bytes, shift, dv, soff, o);
        hipDeviceSynchronize();
        hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
    };
    // 1 + 2: even lanes in range (lane*48 bytes), odd lanes out of range
    for (int l = 0; l < 64; ++l) v[l] = (l & 1) ? (int)0x80000000u : l * 48;
    run(N * 4, 0, 64);
    int ok1 = 1, ok2 = 1;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
            const uint32_t got = r[l * 4 + j];
            if (l & 1) { if (got != 0u) ok2 = 0; }
            else if (got != 0x10000000u + (l * 48 + 64) / 4 + j) ok1 = 0;
        }
    printf("1 in-range copy (voffset + soffset, lane-linear LDS): %s\n", ok1 ? "OK" : "WRONG");
    printf("2 voffset 0x80000000 writes zeros to LDS: %s (lane 1 dword 0 = 0x%08x)\n", ok2 ? "OK" : "NO", r[4]);
    // 3: voffset in range, voffset + soffset out of range
    for (int l = 0; l < 64; ++l) v[l] = l * 16;
    run(2048, 0, 2048);                                    // window 2048 bytes, final offsets 2048..3071
    printf("3 soffset outside the range check (data returned although voffset+soffset >= num_records): %s (lane 0 dword 0 = 0x%08x, expect 0x%08x if unchecked)\n",
           r[0] == 0x10000000u + 512 ? "YES" : "NO", r[0], 0x10000000u + 512);
    // 3b: voffset itself beyond num_records but negative soffset impossible; check boundary: last dwordx4 partially out of range
    for (int l = 0; l < 64; ++l) v[l] = 2048 - 8;          // dwords 0,1 in range, 2,3 out
    run(2048, 0, 0);
    printf("3b straddling chunk: dwords = %08x %08x %08x %08x (per-dword check if the last two are 0)\n", r[0], r[1], r[2], r[3]);
    // 4: descriptor base 1024 bytes before the tensor
    for (int l = 0; l < 64; ++l) v[l] = l * 16;
    run(N * 4, 1024, 0);
    int ok4 = 1;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (r[l * 4 + j] != 0x10000000u + l * 4 + j) ok4 = 0;
    printf("4 descriptor base in front of the tensor: %s\n", ok4 ? "OK" : "WRONG");
    return 0;
}
