// planar_common.h -- device helpers shared by the planar convolution kernels (conv_bf16x.hip, conv_kxr.hip): vector types, the
// LDS chunk swizzle, the plane splits.  Everything is internal to the library (anonymous namespace per translation unit).
#pragma once
#include "stm_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int CV_BM = 128;            // output pixels per workgroup
constexpr int CV_BN = 128;            // output channels per workgroup
constexpr int CV_BK = 32;             // input channels per K-slab (one tap)

// 16-B chunk swizzle of the 64-B LDS rows: chunk ^ f((row >> 2) & 3) with f = (0, 2, 3, 1).  f being a permutation keeps the
// row-major fragment reads of the 32x32x16 MFMA operands conflict-free (16-lane groups of ds_read_b128 see four row
// blocks with four different f), and this particular f does the same for the 16x16x32 operands, whose 16-lane groups mix
// two chunk indices (chunk c of rows 0-3 / 12-15 with chunk c^1 of rows 4-11).
__device__ __forceinline__ int swz(int row) { return (0x1320 >> (((row >> 2) & 3) << 2)) & 3; }
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 64 + ((chunk ^ swz(row)) << 4); }

// two fp32 -> three packed bf16 pairs (round-to-nearest of the running residual; the subtractions are exact)
__device__ __forceinline__ void split2(f32x2 v, unsigned& p0, unsigned& p1, unsigned& p2)
{
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    const f32x2 r1 = v - __builtin_convertvector(h, f32x2);
    const bf16x2 m = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
    const bf16x2 l = __builtin_convertvector(r2, bf16x2);
    p0 = __builtin_bit_cast(unsigned, h);
    p1 = __builtin_bit_cast(unsigned, m);
    p2 = __builtin_bit_cast(unsigned, l);
}

// fp16 form of the split: x = h0 + h1 (11 + 11 significand bits, round to nearest).  With the products h0 g0 + h0 g1 + h1 g0
// the error is ~2^-21 |x g| -- at the level of fp32's own accumulation error for K in the hundreds -- for HALF the MFMAs of
// the bf16 three-plane form.  The price is fp16's range: |x| must stay below 65504 (beyond it the high plane becomes inf
// and the result is non-finite, never silently wrong); weights are brought to ~2^10 by a power-of-two scale per layer that
// the epilogue takes out again, small activations keep an absolute error of 2^-25.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// fp16 two-plane format: x = h + l / 2048 with h = RN16(x), l = RN16((x - h) * 2048).  The residual of a normal h is below
// 2^-11 |x|, so the scaled low plane has the magnitude of x itself and keeps its 11 bits wherever h is normal: 22 bits of
// x for 6.1e-5 <= |x| <= 65504, an absolute error below 1.5e-11 under that.  Both operands scale their low plane, the
// kernel keeps h*h in one accumulator and h*l + l*h in the other and adds them as acc + accl / 2048.
#define STM_F16_LOW_SCALE 2048.0f
__device__ __forceinline__ void split2_f16(f32x2 v, unsigned& p0, unsigned& p1)
{
    const f16x2 h = __builtin_convertvector(v, f16x2);
    const f32x2 r1 = (v - __builtin_convertvector(h, f32x2)) * STM_F16_LOW_SCALE;
    const f16x2 l = __builtin_convertvector(r1, f16x2);
    p0 = __builtin_bit_cast(unsigned, h);
    p1 = __builtin_bit_cast(unsigned, l);
}

// |x| > 65504, inf or nan in any of 8 values (the magnitude bits of those are the largest as integers): such a value has
// no fp16 plane representation.  A producer that meets one raises the caller's sticky flag (stm_planar_set_range_flag),
// because downstream the damage is silent: inf * w + (-inf) * w = nan, and a ReLU epilogue turns nan into 0.
__device__ __forceinline__ void f16_range_check8(const float (&v)[8], int* flag)
{
    unsigned m = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) m = max(m, __builtin_bit_cast(unsigned, v[e]) & 0x7fffffffu);
    if (m > 0x477fe000u && flag) *reinterpret_cast<volatile int*>(flag) = 1;
}

}  // namespace
