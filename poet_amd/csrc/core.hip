// Library plumbing: version + thread-local last-error string (the only state the library keeps).
#include <stdarg.h>
#include <stdio.h>

#include "common.cuh"

namespace poet {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace poet

extern "C" int poet_hip_version(void) { return POET_ABI_VERSION; }
extern "C" const char* poet_hip_last_error(void) { return poet::g_err; }
