// Error plumbing and version entry points of the C ABI (include/aicg.h).
#include "common.h"

namespace aicg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace aicg

extern "C" const char* aicg_last_error(void) { return aicg::g_err; }
extern "C" int aicg_abi_version(void) { return 1; }
