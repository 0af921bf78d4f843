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

// Dropout masks are pure functions of (seed, element index); the seed is a launch ARGUMENT, so a captured HIP graph would
// replay the same masks for ever.  ix_set_dropout_salt(p): from now on every dropout-carrying kernel XORs the 64-bit word at
// device address p into its seed when it RUNS -- the caller bumps that word between replays (interactron_amd/graphs.py).
// NULL (default) switches it off.  One host thread per process; the pointer must stay valid while launches that saw it run.
const uint64_t* ix_g_salt = nullptr;
extern "C" int ix_set_dropout_salt(const void* device_word) {
    ix_g_salt = static_cast<const uint64_t*>(device_word);
    return IX_OK;
}
