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
