// stm_common.h -- shared helpers for the gfx950 kernels behind include/stmask_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/stmask_hip.h"

#define STM_WAVE 64  // gfx950 wavefront width (hard-coded: warpSize folds to 64, no macro exists)

void stm_set_error(const char* fmt, ...);

#define STM_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            stm_set_error(__VA_ARGS__);   \
            return (code);                \
        }                                 \
    } while (0)

#define STM_CHECK_LAUNCH(name)                                                       \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            stm_set_error("%s: launch failed: %s", (name), hipGetErrorString(e_));   \
            return STM_ELAUNCH;                                                      \
        }                                                                            \
    } while (0)

// A/B switches (STM_* environment variables) are read ONCE per process, at first use -- never per launch.
// stm_debug_reload_tunables() (tests, A/B scripts) bumps the generation so they are read again.
int stm_env_generation();
int stm_env_int_uncached(const char* name, int dflt);
#define STM_ENV_INT(name, dflt)                                                          \
    ([]() -> int {                                                                       \
        static int v_ = 0, gen_ = -1;                                                    \
        const int g_ = stm_env_generation();                                             \
        if (gen_ != g_) { v_ = stm_env_int_uncached((name), (dflt)); gen_ = g_; }        \
        return v_;                                                                       \
    }())

int* stm_internal_range_flag();
long long stm_internal_fused_dcn_launches();   // dcn_fused.hip: launches of dcn_fused_kernel (stm_debug_launch_count(1))   // conv_bf16x.hip: the device flag registered with stm_planar_set_range_flag (or null)

static inline hipStream_t stm_hs(stm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int stm_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- device helpers -----------------------------------------------------------------------------

// Workgroup ids are dealt round-robin to the 8 XCDs of the chip, each with its own L2.  Kernels whose neighbouring
// workgroups re-read the same lines (gathers, windows, per-image tables) launch 8 * ceil(nblocks / 8) workgroups and map
// id -> (id & 7) * per_xcd + (id >> 3): each XCD then works on a contiguous run of logical blocks and the shared lines are
// fetched through one L2 instead of eight.  Returns -1 for the padding workgroups.
__device__ __forceinline__ int64_t stm_xcd_block(int64_t nblocks)
{
    const int64_t per_xcd = (nblocks + 7) >> 3;
    const int64_t b = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    return b < nblocks ? b : -1;
}
__device__ __forceinline__ int64_t stm_xcd_block_or_plain(int xcd, int64_t nblocks) { return xcd ? stm_xcd_block(nblocks) : (int64_t)blockIdx.x; }
static inline unsigned stm_xcd_grid(int64_t nblocks) { return (unsigned)(8 * ((nblocks + 7) / 8)); }

// Canonical exp (oracle/stm_oracle.c: stm_exp_f64): identical IEEE operation sequence in double, rounded
// once to fp32.  Compiled with -ffp-contract=off; every fused step is an explicit fma().
__device__ __forceinline__ double stm_exp_f64(double x)
{
    if (x > 709.0) return __longlong_as_double(0x7FF0000000000000LL);
    if (x < -745.0) return 0.0;
    const double LOG2E = 1.4426950408889634074;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double kf = rint(x * LOG2E);
    double r = fma(-kf, LN2_HI, x);
    r = fma(-kf, LN2_LO, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    int k = (int)kf;
    int k1 = k / 2, k2 = k - k1;
    double s1 = __longlong_as_double((long long)(k1 + 1023) << 52);
    double s2 = __longlong_as_double((long long)(k2 + 1023) << 52);
    return p * s1 * s2;
}

__device__ __forceinline__ float stm_expf_canon(float x) { return (float)stm_exp_f64((double)x); }

// box_utils.py:37-88 in fp32, reference operand order (no contraction: -ffp-contract=off).
__device__ __forceinline__ float stm_iou(const float4 a, const float4 b)
{
    float mx = fminf(a.z, b.z) - fmaxf(a.x, b.x);
    float my = fminf(a.w, b.w) - fmaxf(a.y, b.y);
    mx = mx < 0.0f ? 0.0f : mx;
    my = my < 0.0f ? 0.0f : my;
    float inter = mx * my;
    float area_a = (a.z - a.x) * (a.w - a.y);
    float area_b = (b.z - b.x) * (b.w - b.y);
    float uni = area_a + area_b - inter;
    return inter / uni;
}

// box_utils.py:298-316 (cast=False)
__device__ __forceinline__ void stm_sanitize(float x1, float x2, int size, int padding, float& lo, float& hi)
{
    float a = x1 * (float)size, b = x2 * (float)size;
    float l = fminf(a, b), h = fmaxf(a, b);
    l = l - (float)padding;
    h = h + (float)padding;
    lo = l < 0.0f ? 0.0f : l;
    hi = h > (float)size ? (float)size : h;
}

// Sum over the 16 lanes of a DPP row (lanes 16 q .. 16 q + 15): four data-parallel-primitive moves at VALU speed, every lane of the row ends with the
// row's sum -- quad_perm [1, 0, 3, 2], quad_perm [2, 3, 0, 1], row_half_mirror, row_mirror.  (__shfl_xor goes through ds_bpermute: an LDS round trip
// per step; the tail kernels that fold many small sums spent most of their time there.)  One fixed order: run-to-run identical.
// (The builtins exist only in the device pass; the host pass needs the names.)
__device__ __forceinline__ float stm_row16_sum(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));
#endif
    return v;
}
__device__ __forceinline__ unsigned stm_row16_sum(unsigned v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false);
#endif
    return v;
}
// ... and over the whole wave: the four row sums read from lanes 0, 16, 32, 48 and added in that order (wave-uniform result)
__device__ __forceinline__ float stm_wave_sum(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    v = stm_row16_sum(v);
    const int i = __builtin_bit_cast(int, v);
    v = ((__builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16))) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32))) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48));
#endif
    return v;
}
__device__ __forceinline__ int stm_wave_sum_rows(unsigned row_sums)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readlane((int)row_sums, 0) + __builtin_amdgcn_readlane((int)row_sums, 16) + __builtin_amdgcn_readlane((int)row_sums, 32) +
           __builtin_amdgcn_readlane((int)row_sums, 48);
#else
    return (int)row_sums;
#endif
}
