// Error reporting and version entry points of the C-ABI (include/interactron_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

static thread_local char g_last_error[512] = "";

void ix_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

extern "C" const char* ix_last_error(void) { return g_last_error; }
extern "C" int ix_version(void) { return 1; }
