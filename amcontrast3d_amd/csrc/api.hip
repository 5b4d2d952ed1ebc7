// Library identification and per-thread error text for the C-ABI (include/amc3d.h).
#include <stdarg.h>

#include "common.h"

namespace amc {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace amc

AMC_API const char *amc3d_version(void) { return "amc3d-hip gfx950 1"; }
AMC_API const char *amc3d_last_error(void) { return amc::g_err; }
