// Error plumbing and version of the C ABI (include/dun_hip.h).
#include <stdarg.h>

#include "common.h"

namespace dun {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace dun

extern "C" int dun_abi_version(void) { return 4; }
extern "C" const char* dun_last_error(void) { return dun::g_err; }
