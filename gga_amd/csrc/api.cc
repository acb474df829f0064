// Error reporting + ABI version for libgga_hip.so (include/gga_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/gga_hip.h"

static thread_local char g_err[512] = "";

void gga_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* gga_last_error(void) { return g_err; }
extern "C" int gga_abi_version(void) { return 7; }
