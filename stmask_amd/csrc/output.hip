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
