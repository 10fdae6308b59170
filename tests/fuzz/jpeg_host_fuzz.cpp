// ASAN harness for the host half of the JPEG decoder (vtx_jpeg_info + vtx_jpeg_entropy_decode): mutated streams must be refused
// or decoded, with no out-of-bounds access.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include <string>
extern "C" int vtx_jpeg_info(const void* data, long n, int* info);
extern "C" int vtx_jpeg_entropy_decode(const void* data, long n, short* coef, long coef_elems, unsigned short* qt);
static uint64_t s = 88172645463325252ull;
static uint32_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); }
int main(int argc, char** argv) {
    std::vector<std::vector<uint8_t>> seeds;
    for (int i = 2; i < argc; ++i) {
        FILE* f = fopen(argv[i], "rb"); if (!f) continue;
        std::vector<uint8_t> b; uint8_t buf[4096]; size_t k;
        while ((k = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + k);
        fclose(f); seeds.push_back(b);
    }
    long N = atol(argv[1]), ok = 0, refused = 0;
    for (long it = 0; it < N; ++it) {
        std::vector<uint8_t> d = seeds[rnd() % seeds.size()];
        uint32_t mode = rnd() % 100;
        if (mode < 50) { int k = 1 + rnd() % 4; for (int j = 0; j < k; ++j) d[rnd() % d.size()] = (uint8_t)rnd(); }
        else if (mode < 70) d.resize(2 + rnd() % (d.size() - 2));
        else if (mode < 85) { size_t i = rnd() % (d.size() - 4); static const uint8_t mk[] = {0xC0, 0xC4, 0xDB, 0xDA, 0xDD, 0xE1, 0xD9, 0xC2};
                              d[i] = 0xFF; d[i + 1] = mk[rnd() % 8]; }
        else { size_t i = rnd() % d.size(); int k = 1 + rnd() % 40; std::vector<uint8_t> ins(k); for (auto& x : ins) x = (uint8_t)rnd();
               d.insert(d.begin() + i, ins.begin(), ins.end()); }
        // exact-size heap copy: any read past the end is an ASAN report
        uint8_t* p = (uint8_t*)malloc(d.size()); memcpy(p, d.data(), d.size());
        int info[8] = {0};
        if (vtx_jpeg_info(p, (long)d.size(), info) != 0) { ++refused; free(p); continue; }
        // info: whatever the ABI defines; the coefficient count is derived the way virtex_amd/jpeg.py does
        long coef_elems = (long)info[6] * 64;
        if (coef_elems <= 0 || coef_elems > (1l << 26)) { ++refused; free(p); continue; }
        int16_t* coef = (int16_t*)malloc(sizeof(int16_t) * coef_elems);
        uint16_t* qt = (uint16_t*)malloc(sizeof(uint16_t) * 4 * 64);
        if (vtx_jpeg_entropy_decode(p, (long)d.size(), coef, coef_elems, qt) == 0) ++ok; else ++refused;
        free(coef); free(qt); free(p);
    }
    printf("decoded %ld refused %ld\n", ok, refused);
    return 0;
}
