// postproc.hip -- box decoding, candidate generation and Fast NMS for gfx950 (MI355X).
//
// Replaces the torch op chains of layers/box_utils.py:238-283 (decode), layers/functions/TF_utils.py:54-82
// (generate_candidate), layers/functions/detection_TF.py:85-204 (cc_fast_nms / fast_nms; == detection.py:139-263)
// and layers/box_utils.py:37-88 (jaccard).  These are latency-bound integer / compare kernels: the reference
// spends 6-8 tiny launches and several device->host syncs per frame here; we spend one or two launches and none.
//
// Bit-exactness contract (tests/test_gpu_postproc.py): all fp32 arithmetic uses the reference's operand order
// with IEEE basic operations (file compiled with -ffp-contract=off), exp is the canonical stm_exp_f64, sorting
// is by (score descending, row ascending) -- identical to oracle/stm_oracle.c.
//
// Fast NMS runs in ONE workgroup per frame: wave-per-row score reduction with wavefront shuffles, a bitonic
// sort of packed (score,row) keys in LDS (up to 16384 keys = 128 KiB of the 160 KiB LDS), the 200x200 IoU upper
// triangle as one column per lane, and a ballot-based ordered compaction.
#include "stm_common.h"

namespace {

constexpr int NMS_THREADS = 1024;
constexpr int NMS_WAVES = NMS_THREADS / STM_WAVE;
constexpr int NMS_MAX_KEYS = 16384;
#ifndef NMS_SELECT_FROM
#define NMS_SELECT_FROM 4096    // candidates from which cc_nms_kernel selects the top_k before sorting: the selection costs ~70 us flat at 32 frames, the sort 30 / 60 / 115 / 240 us for up to
                                // 2 048 / 4 096 / 6 000 / 12 000 candidates (profiles/r06_cc_nms_select.txt; timing builds: make variant ... VFLAGS=-DNMS_SELECT_FROM=n)
#endif
constexpr int NMS_MAX_TOPK = 512;

// ------------------------------------------------------------------------------------------ decode
__device__ __forceinline__ float4 decode_one(const float4 l, const float4 p)
{
    const float v0 = 0.1f, v1 = 0.2f;
    float t0 = l.x * v0, t1 = l.y * v0;
    float cx = p.x + t0 * p.z;
    float cy = p.y + t1 * p.w;
    float w = p.z * stm_expf_canon(l.z * v1);
    float h = p.w * stm_expf_canon(l.w * v1);
    float x1 = cx - w / 2.0f;
    float y1 = cy - h / 2.0f;
    return make_float4(x1, y1, w + x1, h + y1);
}

__global__ void decode_kernel(const float4* __restrict__ loc, const float4* __restrict__ priors, float4* __restrict__ boxes,
                              int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) boxes[i] = decode_one(loc[i], priors[i]);
}

// Featurealign.py:46-69
__global__ void fcb_ali_kernel(const float* __restrict__ loc, float* __restrict__ off, int B, int HW, int kh, int kw)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)B * HW) return;
    int b = (int)(t / HW), n = (int)(t - (int64_t)b * HW);
    const float* l = loc + (int64_t)b * 4 * HW + n;
    const int K = kh * kw;
    float dx = (l[0] * 0.1f) * (float)kw;
    float dy = (l[HW] * 0.1f) * (float)kh;
    float dw = stm_expf_canon(l[2 * (int64_t)HW] * 0.2f) - 1.0f;
    float dh = stm_expf_canon(l[3 * (int64_t)HW] * 0.2f) - 1.0f;
    // python: arange(-ks // 2 + 1, ks // 2 + 1) with floor division
    int h0 = -((kh + 1) / 2) + 1, w0 = -((kw + 1) / 2) + 1;
    float* o = off + (int64_t)b * 2 * K * HW + n;
    for (int k = 0; k < K; ++k) {
        int i = k / kw, j = k - i * kw;
        o[(int64_t)(2 * k) * HW] = dy + dh * (float)(h0 + i);
        o[(int64_t)(2 * k + 1) * HW] = dx + dw * (float)(w0 + j);
    }
}

// ---------------------------------------------------------------------------- per-row statistics (K1)
// 256 rows per workgroup; the [256][ncls] slab of conf is contiguous in memory, so it is copied coalesced into
// LDS and each thread then scans its own row (row stride ncls words: conflict-free for odd ncls such as 41).
// Writes, per row: decoded box (optional), flag = max_{c>=1} conf > thresh, score = maxconf * centerness.
// LOGITS: `conf` holds the raw class logits and the row's softmax (STMask.py:314: F.softmax(pred_outs['conf'], -1)) is taken here --
// max over all classes, sum of exp(x - max) in class order, p = exp(x_fg - max) / sum -- instead of by a separate pass that reads and
// writes the whole [B, N, ncls] tensor (0.13 ms per step at batch 32).  Only the best foreground probability is needed downstream.
template <bool LOGITS>
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ loc, const float* __restrict__ priors,
                                                        const float* __restrict__ conf,
                                                        const float* __restrict__ centerness, int N, int ncls,
                                                        float thresh, float4* __restrict__ box_out,
                                                        int64_t* __restrict__ flag_out, float* __restrict__ score_out)
{
    extern __shared__ float slab[];
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * 256;
    const int rows = min(256, N - r0);
    const float* cb = conf + ((int64_t)b * N + r0) * ncls;
    // (eight loads in flight per thread: as a plain loop the copy ran one dependent global round trip per element -- 41 of them per workgroup, most of the
    // kernel's 69 us at batch 32 for 80 MB of logits)
    const int total = rows * ncls;
    int idx = threadIdx.x;
    for (; idx + 7 * 256 < total; idx += 8 * 256) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = cb[idx + u * 256];
#pragma unroll
        for (int u = 0; u < 8; ++u) slab[idx + u * 256] = v[u];
    }
    for (; idx < total; idx += 256) slab[idx] = cb[idx];
    __syncthreads();
    const int r = threadIdx.x;
    if (r >= rows) return;
    const float* row = slab + r * ncls;
    float m = row[1];
    for (int c = 2; c < ncls; ++c) m = row[c] > m ? row[c] : m;
    if constexpr (LOGITS) {
        const float mx = row[0] > m ? row[0] : m;
        float sum = 0.0f;
        for (int c = 0; c < ncls; ++c) sum += expf(row[c] - mx);
        m = expf(m - mx) / sum;
    }
    const int64_t gi = (int64_t)b * N + r0 + r;
    if (box_out) {
        const float4 l = reinterpret_cast<const float4*>(loc)[gi];
        const float4 p = reinterpret_cast<const float4*>(priors)[r0 + r];
        box_out[gi] = decode_one(l, p);
    }
    const bool keep = m > thresh;
    if (flag_out) flag_out[gi] = keep ? 1 : 0;
    if (score_out) {
        float s = centerness ? m * centerness[gi] : m;
        // sentinel: rows that are not candidates never enter the sort
        score_out[gi] = keep ? s : __int_as_float(0x7FC00000);  // NaN
    }
}

// -------------------------------------------------------------------------- ordered compaction (K2)
// One workgroup per frame; in place (writes always land in chunks that were already read, see header).
__global__ __launch_bounds__(NMS_THREADS) void compact_kernel(int64_t* __restrict__ keep_idx, float4* __restrict__ box,
                                                              int* __restrict__ count, int N)
{
    __shared__ int wave_cnt[NMS_WAVES];
    __shared__ int running;
    const int b = blockIdx.x;
    int64_t* ki = keep_idx + (int64_t)b * N;
    float4* bx = box + (int64_t)b * N;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) running = 0;
    __syncthreads();
    for (int base = 0; base < N; base += NMS_THREADS) {
        const int i = base + tid;
        const bool f = (i < N) && (ki[i] != 0);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f) v = bx[i];
        const unsigned long long bal = __ballot(f);
        const int lane_prefix = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();  // all reads of this chunk done, wave counts visible
        int wp = 0, tot = 0;
        for (int w = 0; w < NMS_WAVES; ++w) {
            int c = wave_cnt[w];
            if (w < wave) wp += c;
            tot += c;
        }
        const int start = running;
        if (f) {
            const int pos = start + wp + lane_prefix;
            ki[pos] = i;
            bx[pos] = v;
        }
        __syncthreads();
        if (tid == 0) running = start + tot;
        __syncthreads();
    }
    // clear the tail so padded consumers see deterministic values
    const int K = running;
    for (int i = K + tid; i < N; i += NMS_THREADS) {
        ki[i] = 0;
        bx[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid == 0) count[b] = K;
}

// ------------------------------------------------------------------------------------- sort helpers
__device__ __forceinline__ unsigned int ord_desc(float s)
{
    s = s + 0.0f;  // -0 -> +0: they compare equal in the reference sort (x + 0.0 is not foldable under IEEE)
    unsigned int u = __float_as_uint(s);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending order as unsigned
    return ~u;                                        // descending
}
__device__ __forceinline__ float ord_desc_inv(unsigned int k)
{
    unsigned int u = ~k;
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    return __uint_as_float(u);
}

__device__ __forceinline__ void bitonic_sort_lds(unsigned long long* keys, int Kp, int tid, int nthreads)
{
    for (int k = 2; k <= Kp; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < Kp; i += nthreads) {
                int ixj = i ^ j;
                if (ixj > i) {
                    unsigned long long a = keys[i], b = keys[ixj];
                    bool asc = ((i & k) == 0);
                    if ((a > b) == asc) {
                        keys[i] = b;
                        keys[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// wave-wide max over foreground classes with "first maximum wins" (torch.max tie rule), via wavefront shuffles
__device__ __forceinline__ void wave_row_max(const float* __restrict__ row, int ncls, int lane, float& best, int& arg)
{
    float v = -INFINITY;
    int a = 0x7FFFFFFF;
    for (int c = 1 + lane; c < ncls; c += STM_WAVE) {
        float t = row[c];
        if (t > v) { v = t; a = c - 1; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        float ov = __shfl_xor(v, off, STM_WAVE);
        int oa = __shfl_xor(a, off, STM_WAVE);
        if (ov > v || (ov == v && oa < a)) { v = ov; a = oa; }
    }
    best = v;
    arg = a;
}

// IoU column max for sorted box j over rows i < j, NaN-propagating like torch.max (detection_TF.py:108-118)
__device__ __forceinline__ bool nms_keep_column(const float4* sbox, int j, float thr)
{
    const float4 bj = sbox[j];
    float mx = 0.0f;
    bool nan_ = false;
    for (int i = 0; i < j; ++i) {
        float v = stm_iou(sbox[i], bj);
        if (v != v) nan_ = true;
        else if (v > mx) mx = v;
    }
    return !nan_ && mx <= thr;
}

// ------------------------------------------------------------------------------ cross-class NMS (K3)
// MODE 0: scores from candidate conf rows [K, ncls] (reference-shaped call)
// MODE 1: scores precomputed per row by row_stats_kernel (NaN = not a candidate), rows = all N priors
template <int MODE>
__global__ __launch_bounds__(NMS_THREADS) void cc_nms_kernel(const float* __restrict__ conf,
                                                             const float* __restrict__ boxes,
                                                             const float* __restrict__ centerness,
                                                             const float* __restrict__ row_score, int K_cap, int ncls,
                                                             const int* __restrict__ k_dev, float iou_thr, int top_k,
                                                             int Kp, int64_t* __restrict__ idx_out,
                                                             int64_t* __restrict__ cls_out, float* __restrict__ score_out,
                                                             float* __restrict__ box_out, int* __restrict__ count_out)
{
    // all LDS is one dynamic array (no static __shared__ in front: keeps the base 16-byte aligned)
    extern __shared__ unsigned long long keys[];  // [Kp] | float4 sbox[top_k] | int srow[top_k] | int keepf[top_k] | misc
    float4* sbox = reinterpret_cast<float4*>(keys + Kp);
    int* srow = reinterpret_cast<int*>(sbox + top_k);
    int* keepf = srow + top_k;
    int* wave_cnt = keepf + top_k;   // [NMS_WAVES]
    int& n_cand = wave_cnt[NMS_WAVES];

    const int b = blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* conf_b = conf + (int64_t)b * K_cap * ncls;
    const float4* boxes_b = reinterpret_cast<const float4*>(boxes) + (int64_t)b * K_cap;
    const float* cen_b = centerness ? centerness + (int64_t)b * K_cap : nullptr;
    int K = K_cap;
    if (MODE == 0 && k_dev) K = min(K_cap, max(k_dev[b], 0));

    for (int i = tid; i < Kp; i += NMS_THREADS) keys[i] = ~0ull;  // padding sorts last
    if (tid == 0) n_cand = 0;
    __syncthreads();

    if (MODE == 0) {
        for (int row = wave; row < K; row += NMS_WAVES) {
            float best;
            int arg;
            wave_row_max(conf_b + (int64_t)row * ncls, ncls, lane, best, arg);
            if (lane == 0) {
                float s = cen_b ? best * cen_b[row] : best;
                keys[row] = ((unsigned long long)ord_desc(s) << 32) | (unsigned int)row;
            }
        }
        if (tid == 0) n_cand = K;
    } else {
        const float* rs = row_score + (int64_t)b * K_cap;
        for (int row = tid; row < K_cap; row += NMS_THREADS) {
            float s = rs[row];
            if (s == s) {  // candidate
                int pos = atomicAdd(&n_cand, 1);
                if (pos < Kp) keys[pos] = ((unsigned long long)ord_desc(s) << 32) | (unsigned int)row;
            }
        }
        __syncthreads();
        if (n_cand > Kp || n_cand > max(NMS_SELECT_FROM, top_k)) {
            // More candidates than the LDS sort holds (only possible with N > NMS_MAX_KEYS priors, e.g. 58 860 at 736x1280) -- or, since round 6, simply
            // MANY: the bitonic sort of 8 192-16 384 keys is 91-105 barrier-separated stages of one workgroup, and only the top_k best are ever read, so
            // from NMS_SELECT_FROM candidates on the same exact selection runs first and 256 keys are sorted (the benchmark's frames have ~40 candidates and
            // never get here; a frame with 12 000: 240 -> 70 us).
            // Only the top_k best (score descending, row ascending) ever matter, so select exactly those: a radix select
            // (four 8-bit passes over the rows) finds the score key of the top_k-th best candidate; all strictly better
            // candidates enter the sort, and of the candidates that tie with that key the first ones in row order.
            // Deterministic and identical to sorting everything (the reference sorts all candidates, detection_TF.py:93).
            int* hist = reinterpret_cast<int*>(keys);                 // the key array is rebuilt below
            int& sel_bin = wave_cnt[NMS_WAVES + 1];
            int& sel_acc = wave_cnt[NMS_WAVES + 2];
            int& running = wave_cnt[NMS_WAVES + 3];
            unsigned int prefix = 0, pmask = 0;
            int want = top_k;                                          // 1-based rank, among the rows matching `prefix`
            for (int shift = 24; shift >= 0; shift -= 8) {
                __syncthreads();
                for (int i = tid; i < 256; i += NMS_THREADS) hist[i] = 0;
                __syncthreads();
                for (int row = tid; row < K_cap; row += NMS_THREADS) {
                    const float s = rs[row];
                    if (s == s) {
                        const unsigned int k32 = ord_desc(s);
                        if ((k32 & pmask) == prefix) atomicAdd(&hist[(k32 >> shift) & 255u], 1);
                    }
                }
                __syncthreads();
                if (tid == 0) {
                    int acc = 0, bin = 0;
                    for (; bin < 255; ++bin) {
                        if (acc + hist[bin] >= want) break;
                        acc += hist[bin];
                    }
                    sel_bin = bin;
                    sel_acc = acc;
                }
                __syncthreads();
                prefix |= (unsigned int)sel_bin << shift;
                pmask |= 255u << shift;
                want -= sel_acc;
            }
            __syncthreads();
            const unsigned int tkey = prefix;                          // `want` (>= 1) candidates with this key are needed
            for (int i = tid; i < Kp; i += NMS_THREADS) keys[i] = ~0ull;
            if (tid == 0) { n_cand = 0; running = 0; }
            __syncthreads();
            for (int row = tid; row < K_cap; row += NMS_THREADS) {
                const float s = rs[row];
                if (s == s && ord_desc(s) < tkey) {
                    const int pos = atomicAdd(&n_cand, 1);             // < top_k of them
                    keys[pos] = ((unsigned long long)ord_desc(s) << 32) | (unsigned int)row;
                }
            }
            __syncthreads();
            const int n_better = n_cand;
            for (int base = 0; base < K_cap; base += NMS_THREADS) {    // ties, in row order, until `want` are in
                const int row = base + tid;
                bool f = false;
                if (row < K_cap) {
                    const float s = rs[row];
                    f = (s == s) && ord_desc(s) == tkey;
                }
                const unsigned long long bal = __ballot(f);
                const int lane_prefix = __popcll(bal & ((1ull << lane) - 1ull));
                if (lane == 0) wave_cnt[wave] = __popcll(bal);
                __syncthreads();
                int wp = 0, tot = 0;
                for (int w = 0; w < NMS_WAVES; ++w) {
                    const int c = wave_cnt[w];
                    if (w < wave) wp += c;
                    tot += c;
                }
                const int start = running;
                const int pos = start + wp + lane_prefix;
                if (f && pos < want) keys[n_better + pos] = ((unsigned long long)tkey << 32) | (unsigned int)row;
                __syncthreads();
                if (tid == 0) running = start + tot;
                __syncthreads();
                if (start + tot >= want) break;                        // uniform
            }
            if (tid == 0) n_cand = n_better + min(running, want);
        }
    }
    __syncthreads();
    const int nc = min(n_cand, Kp);
    // sort only as many keys as there are candidates (padding keys beyond nc are ~0 and already "sorted last")
    int Ks = 64;
    while (Ks < nc) Ks <<= 1;
    bitonic_sort_lds(keys, Ks, tid, NMS_THREADS);

    const int n = min(nc, top_k);
    if (tid < n) {
        int row = (int)(keys[tid] & 0xFFFFFFFFull);
        srow[tid] = row;
        sbox[tid] = boxes_b[row];
    }
    __syncthreads();
    bool keep = false;
    if (tid < n) keep = nms_keep_column(sbox, tid, iou_thr);
    // ordered compaction of the <= top_k flags (top_k <= NMS_MAX_TOPK <= NMS_THREADS)
    const unsigned long long bal = __ballot(keep);
    const int lane_prefix = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int wp = 0, tot = 0;
    for (int w = 0; w < NMS_WAVES; ++w) {
        int c = wave_cnt[w];
        if (w < wave) wp += c;
        tot += c;
    }
    int64_t* io = idx_out + (int64_t)b * top_k;
    int64_t* co = cls_out + (int64_t)b * top_k;
    float* so = score_out + (int64_t)b * top_k;
    float4* bo = box_out ? reinterpret_cast<float4*>(box_out) + (int64_t)b * top_k : nullptr;
    if (keep) {
        const int pos = wp + lane_prefix;
        io[pos] = srow[tid];
        so[pos] = ord_desc_inv((unsigned int)(keys[tid] >> 32));
        if (bo) bo[pos] = sbox[tid];
        keepf[pos] = srow[tid];
    }
    for (int i = tot + tid; i < top_k; i += NMS_THREADS) {  // deterministic padding
        io[i] = 0;
        co[i] = 0;
        so[i] = 0.0f;
        if (bo) bo[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid == 0) count_out[b] = tot;
    __syncthreads();
    // classes of the survivors: argmax over foreground classes + 1 (detection_TF.py:87,126)
    for (int t = wave; t < tot; t += NMS_WAVES) {
        float best;
        int arg;
        wave_row_max(conf_b + (int64_t)keepf[t] * ncls, ncls, lane, best, arg);
        if (lane == 0) co[t] = (int64_t)arg + 1;
    }
}

// ------------------------------------------------------------------------------- per-class NMS
// stage A: one workgroup per foreground class: sort conf[:,c]*centerness, top_k, IoU, keep & score > conf_thresh
// workspace per class: float kscore[top_k], int krow[top_k]; counts[ncls-1]
// Batched form (blockIdx.y = frame): frame b reads conf + b * conf_bs (rows of ncls), boxes + b * box_bs (float4 rows), centerness + b * cen_bs, its
// candidate count k_dev[b], and writes workspace block b.  row_index (optional, + b * K_cap): candidate row i of the frame is row row_index[i] of
// conf / centerness (the boxes are already compacted) -- the batched pipeline passes the candidate pass's keep list instead of gathering the
// [N, ncls] confidence rows.  The LDS sort covers next_pow2(K) keys of the frame's OWN count, not the capacity: same order, a fraction of the passes.
__global__ __launch_bounds__(NMS_THREADS) void pc_nms_class_kernel(const float* __restrict__ conf,
                                                                   const float* __restrict__ boxes,
                                                                   const float* __restrict__ centerness, int K_cap,
                                                                   int ncls, const int* __restrict__ k_dev, float iou_thr,
                                                                   int top_k, float conf_thresh, int Kp,
                                                                   float* __restrict__ ws_score, int* __restrict__ ws_row,
                                                                   int* __restrict__ ws_count, const int64_t* __restrict__ row_index,
                                                                   int64_t conf_bs, int64_t box_bs, int64_t cen_bs, int64_t ws_bs)
{
    extern __shared__ unsigned long long keys[];  // same carve-up as cc_nms_kernel
    float4* sbox = reinterpret_cast<float4*>(keys + Kp);
    int* srow = reinterpret_cast<int*>(sbox + top_k);
    int* wave_cnt = srow + 2 * top_k;
    const int c = blockIdx.x;  // foreground class index 0..ncls-2
    const int fb = blockIdx.y;  // frame
    conf += fb * conf_bs;
    boxes += fb * box_bs;
    if (centerness) centerness += fb * cen_bs;
    if (row_index) row_index += (int64_t)fb * K_cap;
    ws_score += fb * ws_bs; ws_row += fb * ws_bs; ws_count += (int64_t)fb * ncls;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int K = K_cap;
    if (k_dev) K = min(K_cap, max(k_dev[fb], 0));
    int Ks = 2;
    while (Ks < K) Ks <<= 1;                     // (<= Kp: K <= K_cap)
    for (int i = tid; i < Ks; i += NMS_THREADS) {
        unsigned long long key = ~0ull;
        if (i < K) {
            const int64_t r = row_index ? row_index[i] : (int64_t)i;
            float s = conf[r * ncls + c + 1];
            if (centerness) s = s * centerness[r];
            key = ((unsigned long long)ord_desc(s) << 32) | (unsigned int)i;
        }
        keys[i] = key;
    }
    __syncthreads();
    bitonic_sort_lds(keys, Ks, tid, NMS_THREADS);
    const int n = min(K, top_k);
    const float4* b4 = reinterpret_cast<const float4*>(boxes);
    float sc = 0.0f;
    if (tid < n) {
        int row = (int)(keys[tid] & 0xFFFFFFFFull);
        srow[tid] = row;
        sbox[tid] = b4[row];
        sc = ord_desc_inv((unsigned int)(keys[tid] >> 32));
    }
    __syncthreads();
    bool keep = false;
    if (tid < n) keep = nms_keep_column(sbox, tid, iou_thr) && (sc > conf_thresh);
    const unsigned long long bal = __ballot(keep);
    const int lane_prefix = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int wp = 0, tot = 0;
    for (int w = 0; w < NMS_WAVES; ++w) {
        int cc = wave_cnt[w];
        if (w < wave) wp += cc;
        tot += cc;
    }
    if (keep) {
        int pos = wp + lane_prefix;
        ws_score[(int64_t)c * top_k + pos] = sc;
        ws_row[(int64_t)c * top_k + pos] = srow[tid];
    }
    if (tid == 0) ws_count[c] = tot;
}

// stage B: merge the per-class survivors (class-major order), global stable sort desc, first max_det
__global__ __launch_bounds__(NMS_THREADS) void pc_nms_merge_kernel(const float* __restrict__ boxes, int ncls, int top_k,
                                                                   int max_det, int Kp, const float* __restrict__ ws_score,
                                                                   const int* __restrict__ ws_row,
                                                                   const int* __restrict__ ws_count,
                                                                   int64_t* __restrict__ idx_out,
                                                                   int64_t* __restrict__ cls_out,
                                                                   float* __restrict__ score_out,
                                                                   float* __restrict__ box_out, int* __restrict__ count_out,
                                                                   const int64_t* __restrict__ row_index, int K_cap, int64_t box_bs, int64_t ws_bs)
{
    extern __shared__ unsigned long long keys[];  // [Kp] (low word = flattened position p) | int cls_start[ncls]
    int* cls_start = reinterpret_cast<int*>(keys + Kp);
    const int tid = threadIdx.x;
    const int nc = ncls - 1;
    const int fb = blockIdx.x;                    // frame (batched form: see pc_nms_class_kernel)
    boxes += fb * box_bs;
    ws_score += fb * ws_bs; ws_row += fb * ws_bs; ws_count += (int64_t)fb * ncls;
    idx_out += (int64_t)fb * max_det; cls_out += (int64_t)fb * max_det; score_out += (int64_t)fb * max_det;
    if (box_out) box_out += (int64_t)fb * max_det * 4;
    count_out += fb;
    if (row_index) row_index += (int64_t)fb * K_cap;
    if (tid == 0) {
        int run = 0;
        for (int c = 0; c < nc; ++c) {
            cls_start[c] = run;
            run += ws_count[c];
        }
        cls_start[nc] = run;
    }
    for (int i = tid; i < Kp; i += NMS_THREADS) keys[i] = ~0ull;
    __syncthreads();
    const int total = cls_start[nc];
    for (int t = tid; t < nc * top_k; t += NMS_THREADS) {
        int c = t / top_k, r = t - c * top_k;
        if (r < ws_count[c]) {
            int p = cls_start[c] + r;
            // low word: flattened position (stable tie order) in the high 19 bits would overflow; pack (p, c, r)
            // via p only and recover (c, r) by search below
            keys[p] = ((unsigned long long)ord_desc(ws_score[t]) << 32) | (unsigned int)p;
        }
    }
    __syncthreads();
    bitonic_sort_lds(keys, Kp, tid, NMS_THREADS);
    const int m = min(total, max_det);
    const float4* b4 = reinterpret_cast<const float4*>(boxes);
    float4* bo = box_out ? reinterpret_cast<float4*>(box_out) : nullptr;
    for (int t = tid; t < max_det; t += NMS_THREADS) {
        if (t < m) {
            int p = (int)(keys[t] & 0xFFFFFFFFull);
            int c = 0;
            while (c + 1 < nc && cls_start[c + 1] <= p) ++c;
            int r = p - cls_start[c];
            int row = ws_row[(int64_t)c * top_k + r];
            idx_out[t] = row_index ? row_index[row] : (int64_t)row;       // (the caller's row numbering: prior index in the batched pipeline)
            cls_out[t] = c + 1;
            score_out[t] = ord_desc_inv((unsigned int)(keys[t] >> 32));
            if (bo) bo[t] = b4[row];
        } else {
            idx_out[t] = 0;
            cls_out[t] = 0;
            score_out[t] = 0.0f;
            if (bo) bo[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (tid == 0) count_out[0] = m;
}

__global__ void jaccard_kernel(const float4* __restrict__ a, int na, const float4* __restrict__ b, int nb,
                               float* __restrict__ out)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)na * nb) return;
    int i = (int)(t / nb), j = (int)(t - (int64_t)i * nb);
    out[t] = stm_iou(a[i], b[j]);
}

int next_pow2(int v)
{
    int p = 64;
    while (p < v) p <<= 1;
    return p;
}

size_t nms_lds_bytes(int Kp, int top_k) { return (size_t)Kp * 8 + (size_t)top_k * (16 + 4 + 4) + 4 * (NMS_WAVES + 16); }

template <typename F>
void allow_big_lds(F kernel, size_t bytes)
{
    if (bytes > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

extern "C" int stm_decode_boxes_f32(const float* loc, const float* priors, float* boxes, int64_t n, stm_stream_t stream)
{
    STM_REQUIRE(n >= 0, STM_EINVAL, "stm_decode_boxes_f32: n=%lld", (long long)n);
    if (n == 0) return STM_OK;
    STM_REQUIRE(loc && priors && boxes, STM_ENULL, "stm_decode_boxes_f32: loc/priors/boxes must be non-NULL");
    STM_REQUIRE(((uintptr_t)loc % 16 == 0) && ((uintptr_t)priors % 16 == 0) && ((uintptr_t)boxes % 16 == 0), STM_EINVAL,
                "stm_decode_boxes_f32: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(decode_kernel, dim3(stm_cdiv(n, 256)), dim3(256), 0, stm_hs(stream),
                       reinterpret_cast<const float4*>(loc), reinterpret_cast<const float4*>(priors),
                       reinterpret_cast<float4*>(boxes), n);
    STM_CHECK_LAUNCH("decode_kernel");
    return STM_OK;
}

extern "C" int stm_fcb_ali_offsets_f32(const float* loc, float* offset, int B, int H, int W, int kh, int kw,
                                       stm_stream_t stream)
{
    STM_REQUIRE(loc && offset, STM_ENULL, "stm_fcb_ali_offsets_f32: loc/offset must be non-NULL");
    STM_REQUIRE(B > 0 && H > 0 && W > 0 && kh > 0 && kw > 0, STM_EINVAL, "stm_fcb_ali_offsets_f32: bad sizes");
    int64_t total = (int64_t)B * H * W;
    hipLaunchKernelGGL(fcb_ali_kernel, dim3(stm_cdiv(total, 256)), dim3(256), 0, stm_hs(stream), loc, offset, B, H * W, kh,
                       kw);
    STM_CHECK_LAUNCH("fcb_ali_kernel");
    return STM_OK;
}

extern "C" int stm_generate_candidates_f32(const float* loc, const float* priors, const float* conf, int N, int ncls,
                                           float thresh, int batch, int64_t* keep_idx, float* cand_box, int* count,
                                           stm_stream_t stream)
{
    STM_REQUIRE(loc && priors && conf && keep_idx && cand_box && count, STM_ENULL,
                "stm_generate_candidates_f32: all pointers must be non-NULL");
    STM_REQUIRE(N > 0 && ncls >= 2 && ncls <= 160 && batch > 0 && batch <= 65535, STM_EINVAL,
                "stm_generate_candidates_f32: bad sizes N=%d ncls=%d batch=%d", N, ncls, batch);
    STM_REQUIRE(((uintptr_t)loc % 16 == 0) && ((uintptr_t)priors % 16 == 0) && ((uintptr_t)cand_box % 16 == 0), STM_EINVAL,
                "stm_generate_candidates_f32: loc/priors/cand_box must be 16-byte aligned");
    size_t lds = (size_t)256 * ncls * sizeof(float);
    allow_big_lds(row_stats_kernel<false>, lds);
    hipLaunchKernelGGL(row_stats_kernel<false>, dim3(stm_cdiv(N, 256), batch), dim3(256), lds, stm_hs(stream), loc, priors, conf,
                       (const float*)nullptr, N, ncls, thresh, reinterpret_cast<float4*>(cand_box), keep_idx,
                       (float*)nullptr);
    STM_CHECK_LAUNCH("row_stats_kernel");
    hipLaunchKernelGGL(compact_kernel, dim3(batch), dim3(NMS_THREADS), 0, stm_hs(stream), keep_idx,
                       reinterpret_cast<float4*>(cand_box), count, N);
    STM_CHECK_LAUNCH("compact_kernel");
    return STM_OK;
}

extern "C" int stm_cc_fast_nms_f32(const float* conf, const float* boxes, const float* centerness, int K, int ncls,
                                   const int* k_dev, float iou_thr, int top_k, int batch, int64_t* idx_out,
                                   int64_t* cls_out, float* score_out, float* box_out, int* count_out, stm_stream_t stream)
{
    STM_REQUIRE(idx_out && cls_out && score_out && count_out, STM_ENULL, "stm_cc_fast_nms_f32: outputs must be non-NULL");
    STM_REQUIRE(K >= 0 && ncls >= 2 && batch > 0, STM_EINVAL, "stm_cc_fast_nms_f32: bad sizes K=%d ncls=%d batch=%d", K,
                ncls, batch);
    STM_REQUIRE(top_k > 0 && top_k <= NMS_MAX_TOPK, STM_EINVAL, "stm_cc_fast_nms_f32: top_k=%d not in 1..%d", top_k,
                NMS_MAX_TOPK);
    STM_REQUIRE(K <= NMS_MAX_KEYS, STM_EUNSUPPORTED, "stm_cc_fast_nms_f32: K=%d > %d candidates (use stm_cc_fast_nms_ws_f32)", K, NMS_MAX_KEYS);
    if (K == 0) {
        (void)hipMemsetAsync(count_out, 0, sizeof(int) * batch, stm_hs(stream));
        return STM_OK;
    }
    STM_REQUIRE(conf && boxes, STM_ENULL, "stm_cc_fast_nms_f32: conf/boxes must be non-NULL");
    STM_REQUIRE((uintptr_t)boxes % 16 == 0 && (!box_out || (uintptr_t)box_out % 16 == 0), STM_EINVAL,
                "stm_cc_fast_nms_f32: boxes must be 16-byte aligned");
    const int Kp = next_pow2(K);
    const size_t lds = nms_lds_bytes(Kp, top_k);
    allow_big_lds(cc_nms_kernel<0>, lds);
    hipLaunchKernelGGL(cc_nms_kernel<0>, dim3(batch), dim3(NMS_THREADS), lds, stm_hs(stream), conf, boxes, centerness,
                       (const float*)nullptr, K, ncls, k_dev, iou_thr, top_k, Kp, idx_out, cls_out, score_out, box_out,
                       count_out);
    STM_CHECK_LAUNCH("cc_nms_kernel");
    return STM_OK;
}

extern "C" size_t stm_cc_fast_nms_workspace_bytes(int K, int batch) { return (size_t)batch * (size_t)(K > 0 ? K : 0) * 4 + 256; }

extern "C" int stm_cc_fast_nms_ws_f32(const float* conf, const float* boxes, const float* centerness, int K, int ncls, float iou_thr,
                                      int top_k, int batch, int64_t* idx_out, int64_t* cls_out, float* score_out, float* box_out,
                                      int* count_out, void* workspace, size_t workspace_bytes, stm_stream_t stream)
{
    STM_REQUIRE(idx_out && cls_out && score_out && count_out, STM_ENULL, "stm_cc_fast_nms_ws_f32: outputs must be non-NULL");
    STM_REQUIRE(K >= 0 && ncls >= 2 && ncls <= 160 && batch > 0 && batch <= 65535, STM_EINVAL,
                "stm_cc_fast_nms_ws_f32: bad sizes K=%d ncls=%d batch=%d", K, ncls, batch);
    STM_REQUIRE(top_k > 0 && top_k <= NMS_MAX_TOPK, STM_EINVAL, "stm_cc_fast_nms_ws_f32: top_k=%d not in 1..%d", top_k, NMS_MAX_TOPK);
    if (K == 0) {
        (void)hipMemsetAsync(count_out, 0, sizeof(int) * batch, stm_hs(stream));
        return STM_OK;
    }
    STM_REQUIRE(conf && boxes, STM_ENULL, "stm_cc_fast_nms_ws_f32: conf/boxes must be non-NULL");
    STM_REQUIRE(workspace && workspace_bytes >= stm_cc_fast_nms_workspace_bytes(K, batch), STM_EWORKSPACE,
                "stm_cc_fast_nms_ws_f32: workspace %zu < %zu", workspace_bytes, stm_cc_fast_nms_workspace_bytes(K, batch));
    STM_REQUIRE((uintptr_t)boxes % 16 == 0 && (uintptr_t)workspace % 16 == 0 && (!box_out || (uintptr_t)box_out % 16 == 0), STM_EINVAL,
                "stm_cc_fast_nms_ws_f32: boxes / workspace / box_out must be 16-byte aligned");
    float* score_all = reinterpret_cast<float*>(workspace);
    const size_t lds1 = (size_t)256 * ncls * sizeof(float);
    allow_big_lds(row_stats_kernel<false>, lds1);
    // every row is a candidate here (the caller filtered already): threshold -inf
    hipLaunchKernelGGL(row_stats_kernel<false>, dim3(stm_cdiv(K, 256), batch), dim3(256), lds1, stm_hs(stream), (const float*)nullptr,
                       (const float*)nullptr, conf, centerness, K, ncls, -INFINITY, (float4*)nullptr, (int64_t*)nullptr, score_all);
    STM_CHECK_LAUNCH("row_stats_kernel");
    const int Kp = next_pow2(min(K, NMS_MAX_KEYS));
    const size_t lds = nms_lds_bytes(Kp, top_k);
    allow_big_lds(cc_nms_kernel<1>, lds);
    hipLaunchKernelGGL(cc_nms_kernel<1>, dim3(batch), dim3(NMS_THREADS), lds, stm_hs(stream), conf, boxes, centerness, score_all, K, ncls,
                       (const int*)nullptr, iou_thr, top_k, Kp, idx_out, cls_out, score_out, box_out, count_out);
    STM_CHECK_LAUNCH("cc_nms_kernel");
    return STM_OK;
}

// Fused generate_candidate + cc_fast_nms without any host round trip (STMask.py:313-320 chain).
//   loc [batch,N,4], priors [N,4], conf [batch,N,ncls] soft-maxed, centerness [batch,N] or NULL
//   -> idx_out [batch,top_k] = PRIOR index of each detection, cls/score/box likewise, count [batch].
//   workspace: stm_detect_cc_workspace_bytes(N, batch) (decoded boxes + per-row scores).
extern "C" size_t stm_detect_cc_workspace_bytes(int N, int batch) { return (size_t)batch * N * (16 + 4) + 256; }

static int detect_cc_impl(const float* loc, const float* priors, const float* conf, const float* centerness, int N,
                          int ncls, float conf_thresh, float iou_thr, int top_k, int batch, int64_t* idx_out,
                          int64_t* cls_out, float* score_out, float* box_out, int* count_out, void* workspace,
                          size_t workspace_bytes, stm_stream_t stream, bool logits);

extern "C" int stm_detect_cc_f32(const float* loc, const float* priors, const float* conf, const float* centerness, int N,
                                 int ncls, float conf_thresh, float iou_thr, int top_k, int batch, int64_t* idx_out,
                                 int64_t* cls_out, float* score_out, float* box_out, int* count_out, void* workspace,
                                 size_t workspace_bytes, stm_stream_t stream)
{
    return detect_cc_impl(loc, priors, conf, centerness, N, ncls, conf_thresh, iou_thr, top_k, batch, idx_out, cls_out, score_out, box_out,
                          count_out, workspace, workspace_bytes, stream, false);
}

// the same with `conf` = raw class logits [batch, N, ncls]: the softmax of STMask.py:314 is folded into the per-row pass
extern "C" int stm_detect_cc_logits_f32(const float* loc, const float* priors, const float* conf_logits, const float* centerness, int N,
                                        int ncls, float conf_thresh, float iou_thr, int top_k, int batch, int64_t* idx_out,
                                        int64_t* cls_out, float* score_out, float* box_out, int* count_out, void* workspace,
                                        size_t workspace_bytes, stm_stream_t stream)
{
    return detect_cc_impl(loc, priors, conf_logits, centerness, N, ncls, conf_thresh, iou_thr, top_k, batch, idx_out, cls_out, score_out,
                          box_out, count_out, workspace, workspace_bytes, stream, true);
}

static int detect_cc_impl(const float* loc, const float* priors, const float* conf, const float* centerness, int N,
                          int ncls, float conf_thresh, float iou_thr, int top_k, int batch, int64_t* idx_out,
                          int64_t* cls_out, float* score_out, float* box_out, int* count_out, void* workspace,
                          size_t workspace_bytes, stm_stream_t stream, bool logits)
{
    STM_REQUIRE(loc && priors && conf && idx_out && cls_out && score_out && count_out, STM_ENULL,
                "stm_detect_cc_f32: required pointer is NULL");
    STM_REQUIRE(N > 0 && ncls >= 2 && ncls <= 160 && batch > 0 && batch <= 65535, STM_EINVAL,
                "stm_detect_cc_f32: bad sizes N=%d ncls=%d batch=%d", N, ncls, batch);
    STM_REQUIRE(top_k > 0 && top_k <= NMS_MAX_TOPK, STM_EINVAL, "stm_detect_cc_f32: top_k=%d not in 1..%d", top_k,
                NMS_MAX_TOPK);
    STM_REQUIRE(workspace && workspace_bytes >= stm_detect_cc_workspace_bytes(N, batch), STM_EWORKSPACE,
                "stm_detect_cc_f32: workspace %zu < %zu", workspace_bytes, stm_detect_cc_workspace_bytes(N, batch));
    STM_REQUIRE(((uintptr_t)loc % 16 == 0) && ((uintptr_t)priors % 16 == 0) && ((uintptr_t)workspace % 16 == 0) &&
                    (!box_out || (uintptr_t)box_out % 16 == 0),
                STM_EINVAL, "stm_detect_cc_f32: loc/priors/workspace/box_out must be 16-byte aligned");
    float4* box_all = reinterpret_cast<float4*>(workspace);
    float* score_all = reinterpret_cast<float*>(box_all + (size_t)batch * N);
    size_t lds1 = (size_t)256 * ncls * sizeof(float);
    if (logits) {
        allow_big_lds(row_stats_kernel<true>, lds1);
        hipLaunchKernelGGL(row_stats_kernel<true>, dim3(stm_cdiv(N, 256), batch), dim3(256), lds1, stm_hs(stream), loc, priors, conf,
                           centerness, N, ncls, conf_thresh, box_all, (int64_t*)nullptr, score_all);
    } else {
        allow_big_lds(row_stats_kernel<false>, lds1);
        hipLaunchKernelGGL(row_stats_kernel<false>, dim3(stm_cdiv(N, 256), batch), dim3(256), lds1, stm_hs(stream), loc, priors, conf,
                           centerness, N, ncls, conf_thresh, box_all, (int64_t*)nullptr, score_all);
    }
    STM_CHECK_LAUNCH("row_stats_kernel");
    const int Kp = next_pow2(min(N, NMS_MAX_KEYS));
    const size_t lds = nms_lds_bytes(Kp, top_k);
    allow_big_lds(cc_nms_kernel<1>, lds);
    hipLaunchKernelGGL(cc_nms_kernel<1>, dim3(batch), dim3(NMS_THREADS), lds, stm_hs(stream), conf,
                       reinterpret_cast<const float*>(box_all), centerness, score_all, N, ncls, (const int*)nullptr, iou_thr,
                       top_k, Kp, idx_out, cls_out, score_out, box_out, count_out);
    STM_CHECK_LAUNCH("cc_nms_kernel");
    return STM_OK;
}

extern "C" size_t stm_fast_nms_workspace_bytes(int K, int ncls, int top_k)
{
    (void)K;
    return (size_t)(ncls - 1) * top_k * 8 + (size_t)ncls * 4 + 256;
}

extern "C" int stm_fast_nms_batched_f32(const float* conf, int64_t conf_bstride, const int64_t* row_index, const float* boxes, const float* centerness,
                                        int64_t cen_bstride, int K, int ncls, const int* k_dev, float iou_thr, int top_k, float conf_thresh, int max_det,
                                        int B, int64_t* idx_out, int64_t* cls_out, float* score_out, float* box_out, int* count_out,
                                        void* workspace, size_t workspace_bytes, stm_stream_t stream);
extern "C" int stm_fast_nms_f32(const float* conf, const float* boxes, const float* centerness, int K, int ncls,
                                const int* k_dev, float iou_thr, int top_k, float conf_thresh, int max_det,
                                int64_t* idx_out, int64_t* cls_out, float* score_out, float* box_out, int* count_out,
                                void* workspace, size_t workspace_bytes, stm_stream_t stream)
{
    return stm_fast_nms_batched_f32(conf, 0, nullptr, boxes, centerness, 0, K, ncls, k_dev, iou_thr, top_k, conf_thresh, max_det, 1, idx_out, cls_out,
                                    score_out, box_out, count_out, workspace, workspace_bytes, stream);
}

extern "C" size_t stm_fast_nms_batched_workspace_bytes(int K, int ncls, int top_k, int B)
{
    if (B <= 0) return 0;
    return (size_t)B * stm_fast_nms_workspace_bytes(K, ncls, top_k);
}

extern "C" int stm_fast_nms_batched_f32(const float* conf, int64_t conf_bstride, const int64_t* row_index, const float* boxes, const float* centerness,
                                        int64_t cen_bstride, int K, int ncls, const int* k_dev, float iou_thr, int top_k, float conf_thresh, int max_det,
                                        int B, int64_t* idx_out, int64_t* cls_out, float* score_out, float* box_out, int* count_out,
                                        void* workspace, size_t workspace_bytes, stm_stream_t stream)
{
    STM_REQUIRE(B >= 1, STM_EINVAL, "stm_fast_nms_f32: B=%d", B);
    STM_REQUIRE(idx_out && cls_out && score_out && count_out, STM_ENULL, "stm_fast_nms_f32: outputs must be non-NULL");
    STM_REQUIRE(K >= 0 && ncls >= 2 && ncls <= 128, STM_EINVAL, "stm_fast_nms_f32: bad sizes K=%d ncls=%d", K, ncls);
    STM_REQUIRE(top_k > 0 && top_k <= NMS_MAX_TOPK && max_det > 0, STM_EINVAL, "stm_fast_nms_f32: top_k=%d max_det=%d",
                top_k, max_det);
    STM_REQUIRE(K <= NMS_MAX_KEYS, STM_EUNSUPPORTED, "stm_fast_nms_f32: K=%d > %d candidates", K, NMS_MAX_KEYS);
    STM_REQUIRE((int64_t)(ncls - 1) * top_k <= NMS_MAX_KEYS, STM_EUNSUPPORTED, "stm_fast_nms_f32: (ncls-1)*top_k too large");
    if (K == 0) {
        (void)hipMemsetAsync(count_out, 0, sizeof(int) * B, stm_hs(stream));
        return STM_OK;
    }
    STM_REQUIRE(conf && boxes, STM_ENULL, "stm_fast_nms_f32: conf/boxes must be non-NULL");
    STM_REQUIRE(B == 1 || k_dev, STM_ENULL, "stm_fast_nms_batched_f32: the batched form takes the frames' candidate counts on the device");
    STM_REQUIRE(workspace && workspace_bytes >= (size_t)B * stm_fast_nms_workspace_bytes(K, ncls, top_k), STM_EWORKSPACE,
                "stm_fast_nms_f32: workspace too small");
    // workspace: [B][(ncls - 1) * top_k] scores | the same of rows | [B][ncls] counts
    const int64_t ws_bs = (int64_t)(ncls - 1) * top_k;
    float* ws_score = reinterpret_cast<float*>(workspace);
    int* ws_row = reinterpret_cast<int*>(ws_score + (size_t)B * ws_bs);
    int* ws_count = ws_row + (size_t)B * ws_bs;
    const int Kp = next_pow2(K);
    size_t lds = nms_lds_bytes(Kp, top_k);
    allow_big_lds(pc_nms_class_kernel, lds);
    hipLaunchKernelGGL(pc_nms_class_kernel, dim3(ncls - 1, B), dim3(NMS_THREADS), lds, stm_hs(stream), conf, boxes, centerness,
                       K, ncls, k_dev, iou_thr, top_k, conf_thresh, Kp, ws_score, ws_row, ws_count, row_index, conf_bstride, (int64_t)K * 4, cen_bstride, ws_bs);
    STM_CHECK_LAUNCH("pc_nms_class_kernel");
    const int Kp2 = next_pow2((ncls - 1) * top_k);
    size_t lds2 = (size_t)Kp2 * 8 + (size_t)(ncls + 4) * 4;
    allow_big_lds(pc_nms_merge_kernel, lds2);
    hipLaunchKernelGGL(pc_nms_merge_kernel, dim3(B), dim3(NMS_THREADS), lds2, stm_hs(stream), boxes, ncls, top_k, max_det,
                       Kp2, ws_score, ws_row, ws_count, idx_out, cls_out, score_out, box_out, count_out, row_index, K, (int64_t)K * 4, ws_bs);
    STM_CHECK_LAUNCH("pc_nms_merge_kernel");
    return STM_OK;
}

extern "C" int stm_jaccard_f32(const float* a, int na, const float* b, int nb, float* out, stm_stream_t stream)
{
    STM_REQUIRE(na >= 0 && nb >= 0, STM_EINVAL, "stm_jaccard_f32: negative size");
    if (na == 0 || nb == 0) return STM_OK;
    STM_REQUIRE(a && b && out, STM_ENULL, "stm_jaccard_f32: a/b/out must be non-NULL");
    STM_REQUIRE((uintptr_t)a % 16 == 0 && (uintptr_t)b % 16 == 0, STM_EINVAL, "stm_jaccard_f32: boxes must be 16-byte aligned");
    int64_t total = (int64_t)na * nb;
    hipLaunchKernelGGL(jaccard_kernel, dim3(stm_cdiv(total, 256)), dim3(256), 0, stm_hs(stream),
                       reinterpret_cast<const float4*>(a), na, reinterpret_cast<const float4*>(b), nb, out);
    STM_CHECK_LAUNCH("jaccard_kernel");
    return STM_OK;
}
