// capi.hip -- error reporting and versioning of the C ABI (include/stmask_hip.h).
#include <stdarg.h>
#include <stdlib.h>

#include <atomic>

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
extern "C" size_t stm_struct_bytes(int which)
{
    switch (which) {
        case 0: return sizeof(stm_deform_geom);
        case 1: return sizeof(stm_conv_geom);
        case 2: return sizeof(stm_conv_window);
        case 3: return sizeof(stm_head_layout);
        default: return 0;
    }
}

static std::atomic<int> g_env_gen{0};
int stm_env_generation() { return g_env_gen.load(std::memory_order_relaxed); }
int stm_env_int_uncached(const char* name, int dflt)
{
    const char* s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}
// test / A-B aid: re-read every STM_* switch at its next use
extern "C" void stm_debug_reload_tunables() { g_env_gen.fetch_add(1, std::memory_order_relaxed); }
