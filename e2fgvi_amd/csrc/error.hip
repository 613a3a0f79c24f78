// Thread-local error string of the C ABI (include/e2fgvi_hip.h).
#include "common.h"

static thread_local char g_err[512] = "";

void e2fgvi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* e2fgvi_last_error(void) { return g_err; }
extern "C" int e2fgvi_abi_version(void) { return 8; }
