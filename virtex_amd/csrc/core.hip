// Version / backend / error reporting of the C ABI.
#include <stdarg.h>
#include <stdio.h>

#include "vtx_common.h"

static thread_local char g_err[512] = "";

void vtx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int vtx_version(void) { return 100; }
extern "C" const char* vtx_last_error(void) { return g_err; }
extern "C" const char* vtx_backend(void) {
#ifdef HIPEMU
    return "hipemu";
#else
    return "hip:gfx950";
#endif
}

// Runtime switch used for A/B measurements of the two contraction-kernel generations.
namespace vtxg { int g_vtx_contraction_generation = 2; int g_vtx_ablate = 0; int g_vtx_tile_override = -1; }
extern "C" int vtx_set_contraction_generation(int gen) {
    VTX_CHECK(gen == 1 || gen == 2, VTX_ERR_ARG, "contraction generation must be 1 or 2");
    vtxg::g_vtx_contraction_generation = gen;
    return VTX_OK;
}
extern "C" int vtx_set_ablation(int bits) { vtxg::g_vtx_ablate = bits; return VTX_OK; }
// tests: force tile candidate 0..5 = 256x256, 256x128, 128x128, 128x64, 64x128, 64x64 (-1: automatic)
extern "C" int vtx_set_tile_override(int c) { vtxg::g_vtx_tile_override = c; return VTX_OK; }
