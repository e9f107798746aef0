// capi.hip -- error reporting and versioning of the C ABI (include/stmask_hip.h).
#include <stdarg.h>

#include "stm_common.h"

static thread_local char g_err[512] = "";

void stm_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int stm_version(void) { return STM_ABI_VERSION; }
extern "C" const char* stm_last_error_string(void) { return g_err; }
