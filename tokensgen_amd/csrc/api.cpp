// Error plumbing and version string of libtokensgen_hip.so (host only).
#include <stdarg.h>
#include <stdio.h>

#include "tokensgen_hip.h"

static thread_local char g_err[512] = "";

extern "C" int tg_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* tg_last_error_string(void) { return g_err; }
extern "C" const char* tg_version(void) { return "tokensgen_hip 0.1 (gfx950)"; }
