// ABI version + thread-local error string of libfairygen_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/fairygen_hip.h"

static thread_local char g_err[512] = "";

void fg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int fg_version(void) { return 3; }
extern "C" const char* fg_last_error(void) { return g_err; }
