// fx_api.cpp — error plumbing and version entry points of libfxctr (host-only translation unit).
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/fxctr.h"

static thread_local char g_fx_error[512] = "";

void fx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_fx_error, sizeof(g_fx_error), fmt, ap);
    va_end(ap);
}

extern "C" int fx_abi_version(void) { return FX_ABI_VERSION; }

extern "C" const char* fx_last_error(void) { return g_fx_error; }
