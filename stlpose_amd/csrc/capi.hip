// Error reporting + version of the C ABI.
#include <stdarg.h>
#include "common.cuh"

static thread_local char g_err[512] = "";

int stl_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}

extern "C" const char* stl_last_error(void) { return g_err; }
extern "C" int stl_version(void) { return 1; }
