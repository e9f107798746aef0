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

// COCO compressed RLE strings (pycocotools maskApi.c rleToString) of n masks' run lengths, ON THE HOST: counts [n][ld] (row i holds n_runs[i] runs,
// as stm_mask_resize_rle_f32 leaves them after the copy to the host), out [n][out_ld] bytes, out_len [n].  Plain C: the only part of the output stage
// that is not a device kernel, a few hundred bytes per mask -- but a Python loop over them took longer than the GPU step it follows.
// Returns STM_EINVAL when a string does not fit out_ld (out_len[i] then holds the length it needs).
extern "C" int stm_rle_strings_host(const uint32_t* counts, int ld, const int* n_runs, int n, char* out, int out_ld, int* out_len)
{
    STM_REQUIRE(n >= 0 && ld >= 0 && out_ld >= 0, STM_EINVAL, "stm_rle_strings_host: negative size");
    if (n == 0) return STM_OK;
    STM_REQUIRE(counts && n_runs && out && out_len, STM_ENULL, "stm_rle_strings_host: NULL argument");
    int rc = STM_OK;
    for (int i = 0; i < n; ++i) {
        const uint32_t* c = counts + (size_t)i * ld;
        char* o = out + (size_t)i * out_ld;
        const int m = n_runs[i] < ld ? n_runs[i] : ld;
        int len = 0;
        for (int j = 0; j < m; ++j) {
            long long x = (long long)c[j];
            if (j > 2) x -= (long long)c[j - 2];
            bool more = true;
            while (more) {
                int ch = (int)(x & 0x1f);
                x >>= 5;
                more = (ch & 0x10) ? x != -1 : x != 0;
                if (more) ch |= 0x20;
                if (len < out_ld) o[len] = (char)(ch + 48);
                ++len;
            }
        }
        out_len[i] = len;
        if (len > out_ld) rc = STM_EINVAL;
    }
    if (rc != STM_OK) stm_set_error("stm_rle_strings_host: a string needs more than %d bytes", out_ld);
    return rc;
}
