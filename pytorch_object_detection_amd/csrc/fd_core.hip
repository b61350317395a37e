// fd_core.hip — error state + version
#include "fd_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void fd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* fd_last_error(void) { return g_err; }
extern "C" int32_t fd_version(void) { return 100; }
