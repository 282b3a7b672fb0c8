// Thread-local error channel of the C-ABI (no exceptions cross the boundary).
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void getmi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* getmi_last_error(void) { return g_err; }
extern "C" int getmi_version(void) { return 100; }
extern "C" const char* getmi_arch(void) { return "gfx950"; }
