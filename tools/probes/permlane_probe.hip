#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned lane = threadIdx.x;
    unsigned a = 1000 + lane, b = 2000 + lane;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[lane] = r[0]; out[64 + lane] = r[1];
    unsigned c = 1000 + lane, d = 2000 + lane;
    auto q = __builtin_amdgcn_permlane32_swap(c, d, false, false);
    out[128 + lane] = q[0]; out[192 + lane] = q[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"permlane16_swap r[0] (from a=1000+lane)", "permlane16_swap r[1] (from b=2000+lane)", "permlane32_swap r[0]", "permlane32_swap r[1]"};
    for (int s = 0; s < 4; ++s) { printf("%s:\n", names[s]); for (int i = 0; i < 64; ++i) printf("%u%s", h[s * 64 + i], (i % 16 == 15) ? "\n" : " "); }
    return 0;
}
