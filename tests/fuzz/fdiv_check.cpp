// Host check of vtx_fdiv30 (virtex_amd/csrc/vtx_common.h): floor(n / d) for 0 <= n < 2^30 through a float reciprocal that may be
// off by up to 2 ulp (the hardware's v_rcp_f32 is specified to 1 ulp; host-side reciprocals in the geometry structs are correctly
// rounded).  Compiled with the emulator's shim (tests/test_kernels.py::test_pixel_index_division_is_exact_below_2_30): the SAME
// function text the kernels compile.  Prints "ok <cases>" or the first failing case.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vtx_common.h"

static float nudge(float x, int ulps) {
    uint32_t u; memcpy(&u, &x, 4); u += ulps; float y; memcpy(&y, &u, 4); return y;
}

int main() {
    uint64_t state = 88172645463325252ull;
    auto rnd = [&]() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state; };
    long cases = 0;
    const int fixed[] = {1, 2, 3, 5, 7, 9, 14, 28, 29, 49, 56, 57, 58, 112, 113, 196, 224, 225, 230, 784, 3136, 3249, 12544, 12996,
                         50176, 52900, 65535, 65536, 65537, 1000003, (1 << 24) - 1};
    for (int pass = 0; pass < 2; ++pass) {
        const int nd = pass == 0 ? (int)(sizeof(fixed) / sizeof(fixed[0])) : 4000;
        for (int i = 0; i < nd; ++i) {
            const int d = pass == 0 ? fixed[i] : (int)(rnd() % ((1u << 24) - 1)) + 1;
            for (int ulps = -2; ulps <= 2; ++ulps) {
                const float inv = nudge(1.0f / (float)d, ulps);
                for (int j = 0; j < 600; ++j) {
                    long n;
                    const long top = (1L << 30) - 1;
                    if (j < 6) { const long e[] = {0, 1, (long)d - 1, d, top, top - 1}; n = e[j]; }
                    else if (j < 300) { const long k = (long)(rnd() % (top / d + 1)); n = k * d + (long)(j % 3) - 1; }   // multiples of d and their neighbours
                    else if (j < 400) n = (1L << 24) - 50 + (j - 300);                                                  // across the old limit
                    else n = (long)(rnd() % (unsigned long)(top + 1));
                    if (n < 0) n = 0;
                    if (n > top) n = top;
                    const int q = vtx_fdiv30((int)n, d, inv);
                    ++cases;
                    if (q != (int)(n / d)) { printf("FAIL n=%ld d=%d ulps=%d got %d want %ld\n", n, d, ulps, q, n / d); return 1; }
                }
            }
        }
    }
    printf("ok %ld\n", cases);
    return 0;
}
