// output.hip -- output stage on the device: the mask leg of postprocess_ytbvis
// (layers/output_utils.py:85-106) -- un-pad, bilinear resize to the original frame size, threshold at 0.5 and
// COCO run-length encoding -- without ever materialising the full-resolution masks in host memory.
//
// The reference copies every [ori_h, ori_w] mask to the host (`.cpu()`, output_utils.py:103) and run-length encodes it
// with pycocotools: 921 KB per mask at 720p.  Here two integer/byte kernels leave only the run lengths (a few hundred
// 32-bit counts per mask) to be copied:
//   1. resize + threshold + bit-pack: lanes walk the COLUMN-major pixel order RLE needs (consecutive lanes = consecutive
//      rows of one column), a wavefront ballot turns 64 pixels into one 64-bit word;
//   2. run extraction, one workgroup per mask: transitions are the set bits of w ^ ((w << 1) | carry); per-word
//      popcounts are prefix-summed across the workgroup, every thread then emits its words' transition positions in
//      order and the counts are the first differences.
// Arithmetic follows ATen's bilinear kernel (align_corners=False) in fp32, operand order as in oracle/stm_oracle.c, so
// the bits agree with the oracle exactly (-ffp-contract=off).
#include "stm_common.h"

namespace {

__global__ __launch_bounds__(256) void resize_threshold_pack_kernel(const float* __restrict__ masks, int mh, int mw,
                                                                    int crop_h, int crop_w, int out_h, int out_w, float thr,
                                                                    unsigned long long* __restrict__ bits, int words)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int word = blockIdx.x * 4 + wave;
    const int i = blockIdx.y;
    if (word >= words) return;
    const int64_t p = (int64_t)word * 64 + lane;   // column-major pixel index: p = x * out_h + y
    bool b = false;
    if (p < (int64_t)out_h * out_w) {
        const int x = (int)(p / out_h), y = (int)(p - (int64_t)x * out_h);
        const float* m = masks + (int64_t)i * mh * mw;
        const float sh = (float)crop_h / (float)out_h, sw = (float)crop_w / (float)out_w;
        float fy = sh * ((float)y + 0.5f) - 0.5f, fx = sw * ((float)x + 0.5f) - 0.5f;
        if (fy < 0.0f) fy = 0.0f;
        if (fx < 0.0f) fx = 0.0f;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < crop_h - 1 ? 1 : 0), x1 = x0 + (x0 < crop_w - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
        const float v = hy * (hx * m[y0 * mw + x0] + lx * m[y0 * mw + x1]) + ly * (hx * m[y1 * mw + x0] + lx * m[y1 * mw + x1]);
        b = v > thr;
    }
    const unsigned long long bal = __ballot(b);
    if (lane == 0) bits[(int64_t)i * words + word] = bal;
}

// bits of word w that are real pixels (the last word is zero-padded: a 1 -> padding "transition" is not a run boundary)
__device__ __forceinline__ unsigned long long valid_bits(int w, int64_t n_px)
{
    const int64_t rem = n_px - (int64_t)w * 64;
    return rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
}

// one workgroup per mask; counts[i][0..n_runs[i]) (capacity max_runs; n_runs reports the true number)
__global__ __launch_bounds__(1024) void rle_runs_kernel(const unsigned long long* __restrict__ bits, int words, int64_t n_px,
                                                        unsigned int* __restrict__ counts, int max_runs, int* __restrict__ n_runs,
                                                        unsigned int* __restrict__ trans_ws)
{
    __shared__ int wave_tot[16];
    __shared__ int block_base;
    const int i = blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned long long* bw = bits + (int64_t)i * words;
    unsigned int* T = trans_ws + (int64_t)i * max_runs;   // transition positions
    unsigned int* cnt = counts + (int64_t)i * max_runs;
    const int per = (words + 1023) / 1024;                // consecutive words per thread
    const int w0 = tid * per, w1 = min(words, w0 + per);
    // pass 1: transitions in my words
    int mine = 0;
    for (int w = w0; w < w1; ++w) {
        const unsigned long long cur = bw[w];
        const unsigned long long prev = w ? (bw[w - 1] >> 63) : 0ull;
        mine += __popcll((cur ^ ((cur << 1) | prev)) & valid_bits(w, n_px));
    }
    // exclusive scan over the 1024 threads (wave scan + wave totals)
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wave) base += wave_tot[w];
        total += wave_tot[w];
    }
    int pos = base + incl - mine;
    // pass 2: emit transition positions in order
    for (int w = w0; w < w1; ++w) {
        const unsigned long long cur = bw[w];
        const unsigned long long prev = w ? (bw[w - 1] >> 63) : 0ull;
        unsigned long long d = (cur ^ ((cur << 1) | prev)) & valid_bits(w, n_px);
        while (d) {
            const int b = __ffsll((long long)d) - 1;
            if (pos < max_runs) T[pos] = (unsigned int)(w * 64 + b);
            ++pos;
            d &= d - 1;
        }
    }
    __threadfence_block();
    __syncthreads();
    // counts[j] = T[j] - T[j-1] (T[-1] = 0); last count = n_px - T[last]
    const int nr = total + 1;
    for (int j = tid; j < min(nr, max_runs); j += 1024) {
        const unsigned int hi = (j < total) ? T[j] : (unsigned int)n_px;
        const unsigned int lo = j ? T[j - 1] : 0u;
        cnt[j] = hi - lo;
    }
    if (tid == 0) n_runs[i] = nr;
    (void)block_base;
}

}  // namespace

extern "C" size_t stm_mask_rle_workspace_bytes(int n, int out_h, int out_w, int max_runs)
{
    size_t words = ((size_t)out_h * out_w + 63) / 64;
    return (size_t)n * words * 8 + (size_t)n * max_runs * 4 + 256;
}

extern "C" int stm_mask_resize_rle_f32(const float* masks, int n, int mh, int mw, int crop_h, int crop_w, int out_h, int out_w,
                                       float thr, uint32_t* counts, int max_runs, int* n_runs, void* workspace,
                                       size_t workspace_bytes, stm_stream_t stream)
{
    STM_REQUIRE(n >= 0, STM_EINVAL, "stm_mask_resize_rle_f32: n=%d", n);
    if (n == 0) return STM_OK;
    STM_REQUIRE(masks && counts && n_runs, STM_ENULL, "stm_mask_resize_rle_f32: masks/counts/n_runs must be non-NULL");
    STM_REQUIRE(mh > 0 && mw > 0 && crop_h > 0 && crop_h <= mh && crop_w > 0 && crop_w <= mw && out_h > 0 && out_w > 0,
                STM_EINVAL, "stm_mask_resize_rle_f32: bad sizes");
    STM_REQUIRE((int64_t)out_h * out_w < ((int64_t)1 << 31) && max_runs > 0 && n <= 65535, STM_EINVAL,
                "stm_mask_resize_rle_f32: output too large");
    STM_REQUIRE(workspace && workspace_bytes >= stm_mask_rle_workspace_bytes(n, out_h, out_w, max_runs), STM_EWORKSPACE,
                "stm_mask_resize_rle_f32: workspace too small");
    const int64_t n_px = (int64_t)out_h * out_w;
    const int words = (int)((n_px + 63) / 64);
    unsigned long long* bits = reinterpret_cast<unsigned long long*>(workspace);
    unsigned int* trans = reinterpret_cast<unsigned int*>(bits + (size_t)n * words);
    hipLaunchKernelGGL(resize_threshold_pack_kernel, dim3(stm_cdiv(words, 4), n), dim3(256), 0, stm_hs(stream), masks, mh, mw,
                       crop_h, crop_w, out_h, out_w, thr, bits, words);
    STM_CHECK_LAUNCH("resize_threshold_pack_kernel");
    hipLaunchKernelGGL(rle_runs_kernel, dim3(n), dim3(1024), 0, stm_hs(stream), bits, words, n_px, counts, max_runs, n_runs,
                       trans);
    STM_CHECK_LAUNCH("rle_runs_kernel");
    return STM_OK;
}


// ---- frame pre-processing: eval.py:703-717 = mmcv.imresize (cv2 INTER_LINEAR on uint8) ->
// (im - MEANS) / STD in float64 -> zero pad to a multiple of 32 -> CHW fp32, as ONE pass over the output.  The 8-bit
// bilinear arithmetic is OpenCV's fixed-point path, restated in oracle/stm_oracle.c (orc_resize_pixel_u8) with the
// derivation; this kernel performs the identical integer / float / double operation sequence, so the two agree bit
// for bit.  Thread = one output pixel, three channels; HBM-bound (2.8 MB in, 2.9 MB out per 720p frame).
namespace {

__device__ __forceinline__ void resize_coeffs(int src, int dst, int d, bool clamp_f, int& s, int& c0, int& c1)
{
    const double inv = (double)dst / (double)src;
    const double scale = 1.0 / inv;
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    s = (int)floorf(f);
    f -= (float)s;
    if (clamp_f) {
        if (s < 0) { f = 0.0f; s = 0; }
        if (s >= src - 1) { f = 0.0f; s = src - 1; }
    }
    int a0 = (int)rintf((1.0f - f) * 2048.0f), a1 = (int)rintf(f * 2048.0f);   // cvRound: round half to even
    c0 = min(max(a0, -32768), 32767);
    c1 = min(max(a1, -32768), 32767);
}

__global__ __launch_bounds__(256) void preprocess_u8_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, int H0, int W0,
                                                            int h, int w, int Hp, int Wp, double m0, double m1, double m2, double s0,
                                                            double s1, double s2, int mode)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int i = blockIdx.z;
    if (x >= Wp || y >= Hp) return;
    float o[3] = {0.0f, 0.0f, 0.0f};
    if (y < h && x < w) {
        int sx, sy, a0, a1, b0, b1;
        resize_coeffs(W0, w, x, true, sx, a0, a1);
        resize_coeffs(H0, h, y, false, sy, b0, b1);
        const int sx1 = min(sx + 1, W0 - 1);
        const int y0 = min(max(sy, 0), H0 - 1), y1 = min(max(sy + 1, 0), H0 - 1);
        const uint8_t* r0 = img + ((size_t)i * H0 + y0) * W0 * 3;
        const uint8_t* r1 = img + ((size_t)i * H0 + y1) * W0 * 3;
        const double mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int D0 = r0[sx * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
            const int D1 = r1[sx * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
            int v = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2;
            v = min(max(v, 0), 255);
            double d = (double)v;
            if (mode == 1) d = (d - mean[c]) / stdv[c];
            else if (mode == 2) d = d - mean[c];
            else if (mode == 3) d = d / 255.0;
            o[c] = (float)d;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) out[(((size_t)i * 3 + c) * Hp + y) * Wp + x] = o[c];
}

}  // namespace

extern "C" int stm_preprocess_u8_f32(const uint8_t* img, float* out, int n, int H0, int W0, int h, int w, int Hp, int Wp,
                                     const double* mean, const double* stdv, int mode, stm_stream_t stream)
{
    STM_REQUIRE(img && out, STM_ENULL, "stm_preprocess_u8_f32: img/out must be non-NULL");
    STM_REQUIRE(n > 0 && n <= 65535 && H0 > 0 && W0 > 0 && h > 0 && w > 0 && Hp >= h && Wp >= w, STM_EINVAL,
                "stm_preprocess_u8_f32: bad sizes n=%d src=%dx%d dst=%dx%d padded=%dx%d", n, H0, W0, h, w, Hp, Wp);
    STM_REQUIRE(mode >= 0 && mode <= 3, STM_EINVAL, "stm_preprocess_u8_f32: mode %d not in 0..3", mode);
    STM_REQUIRE(mode == 0 || mode == 3 || (mean && (mode == 2 || stdv)), STM_ENULL, "stm_preprocess_u8_f32: mean/std needed for mode %d", mode);
    const double m[3] = {mean ? mean[0] : 0.0, mean ? mean[1] : 0.0, mean ? mean[2] : 0.0};
    const double s[3] = {stdv ? stdv[0] : 1.0, stdv ? stdv[1] : 1.0, stdv ? stdv[2] : 1.0};
    const dim3 grid(stm_cdiv(Wp, 64), stm_cdiv(Hp, 4), n);
    hipLaunchKernelGGL(preprocess_u8_kernel, grid, dim3(256), 0, stm_hs(stream), img, out, H0, W0, h, w, Hp, Wp, m[0], m[1], m[2],
                       s[0], s[1], s[2], mode);
    STM_CHECK_LAUNCH("preprocess_u8_kernel");
    return STM_OK;
}
