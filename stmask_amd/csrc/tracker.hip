// tracker.hip -- the row bookkeeping of the batched clip pipeline (stmask_amd/pipeline.py) as a handful of kernels.
//
// Replaces, per step and for ALL clips of the batch at once, the torch op chains the reference spends on Track_TF.track
// (layers/functions/track_TF.py:50-181) and TF_utils.CandidateShift (TF_utils.py:12-51):
//   gather_detections   the per-key index_selects of the Fast-NMS survivors (detection_TF.py:120-134) + their concatenation
//   shift_rois          bbox_feat_extractor's box -> RoI conversion (track_to_segment_head.py:65-88 up to roi_align)
//   shift_apply         decode(loc_shift, center_size(box)) + coefficient shift + score decay (TF_utils.py:40-48)
//   match_scores        compute_comp_scores + argmax (TF_utils.py:99-120, track_TF.py:104-129)
//   gather_rows2        the tracker update: cat(prev, det).index_select(plan) for every row tensor (track_TF.py:132-156)
//   keep_flags / pack   the keep rule (track_TF.py:158-165) and the fixed-shape [clips, top_k, 40] output rows
// These are latency-bound integer / copy kernels over a few thousand rows: the win is launches (~140 -> ~15 per step), not
// bytes.  All fp32 arithmetic uses the reference's operand order (file compiled with -ffp-contract=off), so results equal the
// torch chains bit for bit (tests/test_gpu_tracker.py compares them).
#include "stm_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// grid (top_k, B), block 64: slot (b, j) is a detection iff j < cnt[b]; its row = sum_{b' < b} cnt[b'] + j
struct GatherDetArgs {
    const int64_t* idx;      // [B, top_k] prior index
    const int64_t* cls;      // [B, top_k]
    const float* score;      // [B, top_k]
    const float* box;        // [B, top_k, 4]
    const int* cnt;          // [B]
    const float* coeff;      // [B, N, mdim]
    const float* track;      // [B, N, edim]
    const float* cen;        // [B, N] or null
    float* o_box; int64_t* o_cls; float* o_score; float* o_coeff; float* o_track; float* o_cen; int* o_clip;
    int B, top_k, N, mdim, edim, D;
};

__global__ __launch_bounds__(64) void gather_detections_kernel(const GatherDetArgs a)
{
    const int b = blockIdx.y, j = blockIdx.x;
    if (j >= a.cnt[b]) return;
    int row = j;
    for (int q = 0; q < b; ++q) row += a.cnt[q];
    if (row >= a.D) return;                          // host and device counts disagree: never write out of bounds
    const int lane = threadIdx.x;
    const int64_t slot = (int64_t)b * a.top_k + j;
    const int64_t src = (int64_t)b * a.N + a.idx[slot];
    if (lane < 4) a.o_box[(int64_t)row * 4 + lane] = a.box[slot * 4 + lane];
    if (lane == 4) a.o_cls[row] = a.cls[slot];
    if (lane == 5) a.o_score[row] = a.score[slot];
    if (lane == 6) a.o_clip[row] = b;
    if (lane == 7 && a.o_cen) a.o_cen[row] = a.cen ? a.cen[src] : 0.0f;
    for (int c = lane; c < a.mdim; c += 64) a.o_coeff[(int64_t)row * a.mdim + c] = a.coeff[src * a.mdim + c];
    for (int c = lane; c < a.edim; c += 64) a.o_track[(int64_t)row * a.edim + c] = a.track[src * a.edim + c];
}

// ---------------------------------------------------------------------------------------------------------------------
// rois[r] = (clip, sanitize_coordinates_hw(box, fh, fw)) -- box_utils.py:298-337 with cast=False, padding 0
__global__ __launch_bounds__(256) void shift_rois_kernel(const float* __restrict__ box, const int* __restrict__ clip, float* __restrict__ rois,
                                                         int n, int fh, int fw)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const float4 b = reinterpret_cast<const float4*>(box)[r];
    float x1, x2, y1, y2;
    stm_sanitize(b.x, b.z, fw, 0, x1, x2);
    stm_sanitize(b.y, b.w, fh, 0, y1, y2);
    float* o = rois + (int64_t)r * 5;
    o[0] = (float)clip[r];
    o[1] = x1; o[2] = y1; o[3] = x2; o[4] = y2;
}

// box' = decode(loc_shift, center_size(box)) (box_utils.py:25-35,238-283); coeff' = coeff + coeff_shift; score' = score * 0.95
__global__ __launch_bounds__(256) void shift_apply_kernel(const float* __restrict__ loc, const float* __restrict__ coeff_shift,
                                                          float* __restrict__ box, float* __restrict__ coeff, float* __restrict__ score, int n,
                                                          int mdim, float decay)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n) {
        const float4 b = reinterpret_cast<const float4*>(box)[t];
        const float4 l = reinterpret_cast<const float4*>(loc)[t];
        // center_size: ((x2 + x1) / 2, (y2 + y1) / 2, x2 - x1, y2 - y1)
        const float cx = (b.z + b.x) / 2.0f, cy = (b.w + b.y) / 2.0f, pw = b.z - b.x, ph = b.w - b.y;
        const float v0 = 0.1f, v1 = 0.2f;
        const float t0 = l.x * v0, t1 = l.y * v0;
        const float ncx = cx + t0 * pw, ncy = cy + t1 * ph;
        const float w = pw * stm_expf_canon(l.z * v1), h = ph * stm_expf_canon(l.w * v1);
        const float x1 = ncx - w / 2.0f, y1 = ncy - h / 2.0f;
        reinterpret_cast<float4*>(box)[t] = make_float4(x1, y1, w + x1, h + y1);
        score[t] = score[t] * decay;
    }
    const int64_t total = (int64_t)n * mdim;
    for (int64_t i = t; i < total; i += (int64_t)gridDim.x * 256) coeff[i] = coeff[i] + coeff_shift[i];
}

// ---------------------------------------------------------------------------------------------------------------------
// One wave per detection row.  comp over the columns [dummy | prev rows of the same clip]:
//   col 0:      ((0 + 1) / 2 + c0 * score) + c1 * dummy + c2 * dummy + c3 * 1
//   col 1 + p:  (((cos + 1) / 2 + c0 * score) + c1 * mask_iou) + c2 * box_iou + c3 * (class equal)
// in exactly this association (TF_utils.py:117-120 evaluates left to right); argmax with the lowest column on ties.
// match[d] = 0 (new instance) or 1 + global prev row.
struct MatchArgs {
    const float* cos;        // [D, Pn] raw dot products of the track embeddings, or null: computed here from the two tables below
    const float* det_track;  // [D, E]
    const float* prev_track; // [Pn, E]
    int E;
    const float* miou;       // [D, Pn] (pairs of different clips are 0 and never read)
    const float* det_box;    // [D, 4]
    const float* prev_box;   // [Pn, 4]
    const float* det_score;  // [D]
    const int64_t* det_cls;  // [D]
    const int64_t* prev_cls; // [Pn]
    const int* det_clip;     // [D]
    const int* prev_off;     // [B + 1] row ranges of the clips in the prev table
    int* match;              // [D]
    int D, Pn;
    float c0, c1, c2, c3, dummy;
};

__global__ __launch_bounds__(256) void match_scores_kernel(const MatchArgs a)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int d = blockIdx.x * 4 + wave;
    if (d >= a.D) return;
    const int clip = a.det_clip[d];
    const int p0 = a.prev_off[clip], p1 = a.prev_off[clip + 1];
    const float s = a.det_score[d];
    const float4 db = reinterpret_cast<const float4*>(a.det_box)[d];
    const int64_t dc = a.det_cls[d];
    float best = -INFINITY;
    int arg = 0x7FFFFFFF;
    if (lane == 0) {
        float t = (0.0f + 1.0f) / 2.0f + a.c0 * s;
        t = t + a.c1 * a.dummy;
        t = t + a.c2 * a.dummy;
        t = t + a.c3 * 1.0f;
        best = t;
        arg = 0;
    }
    for (int p = p0 + lane; p < p1; p += 64) {
        float dot;
        if (a.cos) {
            dot = a.cos[(int64_t)d * a.Pn + p];
        } else {
            // the detection's embedding is uniform over the wave, the lane walks its own prev row (E is 128: the all-pairs matrix
            // product this replaces spent 120 us in a library GEMM on a [164 x 128] x [128 x 3584] problem of which one pair
            // in 32 -- the same-clip ones -- is ever read)
            const float* x = a.det_track + (int64_t)d * a.E;
            const float* y = a.prev_track + (int64_t)p * a.E;
            dot = 0.0f;
            if ((a.E & 3) == 0) {
                for (int e = 0; e < a.E; e += 4) {
                    const float4 xv = *reinterpret_cast<const float4*>(x + e), yv = *reinterpret_cast<const float4*>(y + e);
                    dot = __builtin_fmaf(xv.x, yv.x, dot); dot = __builtin_fmaf(xv.y, yv.y, dot);
                    dot = __builtin_fmaf(xv.z, yv.z, dot); dot = __builtin_fmaf(xv.w, yv.w, dot);
                }
            } else {
                for (int e = 0; e < a.E; ++e) dot = __builtin_fmaf(x[e], y[e], dot);
            }
        }
        const float cs = (dot + 1.0f) / 2.0f;
        const float bi = stm_iou(db, reinterpret_cast<const float4*>(a.prev_box)[p]);
        float t = cs + a.c0 * s;
        t = t + a.c1 * a.miou[(int64_t)d * a.Pn + p];
        t = t + a.c2 * bi;
        t = t + a.c3 * (a.prev_cls[p] == dc ? 1.0f : 0.0f);
        if (t > best || (t == best && p + 1 < arg)) { best = t; arg = p + 1; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, STM_WAVE);
        const int oa = __shfl_xor(arg, off, STM_WAVE);
        if (ov > best || (ov == best && oa < arg)) { best = ov; arg = oa; }
    }
    if (lane == 0) a.match[d] = arg == 0x7FFFFFFF ? 0 : arg;
}

// ---------------------------------------------------------------------------------------------------------------------
// out_t[r] = plan[r] < n_a ? a_t[plan[r]] : b_t[plan[r] - n_a]  for up to 8 row tensors t in one launch (rows of row_bytes[t])
struct GatherRowsArgs {
    const uint8_t* a[8];
    const uint8_t* b[8];
    uint8_t* out[8];
    int row_bytes[8];
    int vec[8];              // 1: rows and bases are 16-byte aligned
    int n_tensors, n_rows, n_a;
    const int* plan;
};

__global__ __launch_bounds__(256) void gather_rows2_kernel(const GatherRowsArgs g)
{
    const int r = blockIdx.x;
    if (r >= g.n_rows) return;
    const int64_t src = g.plan[r];
    const bool from_a = src < g.n_a;
    const int64_t sr = from_a ? src : src - g.n_a;
#pragma unroll 1
    for (int t = 0; t < g.n_tensors; ++t) {
        const int rb = g.row_bytes[t];
        const uint8_t* s = (from_a ? g.a[t] : g.b[t]) + sr * rb;
        uint8_t* o = g.out[t] + (int64_t)r * rb;
        if (g.vec[t]) {
            for (int i = threadIdx.x * 16; i < rb; i += 256 * 16) *reinterpret_cast<uint4*>(o + i) = *reinterpret_cast<const uint4*>(s + i);
        } else {
            for (int i = threadIdx.x * 4; i < rb; i += 256 * 4) *reinterpret_cast<uint32_t*>(o + i) = *reinterpret_cast<const uint32_t*>(s + i);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// keep[r] = tracked[r] <= max_age && #(mask[r] > 0.5) > 1 && score[r] > thr   (track_TF.py:158-165)
// one block per row; the pixel count stops as soon as it is decided
__global__ __launch_bounds__(256) void keep_flags_kernel(const float* __restrict__ mask, const float* __restrict__ score, const int* __restrict__ tracked,
                                                         int* __restrict__ keep, int hw, int max_age, float thr)
{
    __shared__ int cnt;
    const int r = blockIdx.x;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const bool other = tracked[r] <= max_age && score[r] > thr;   // block-uniform
    if (other) {
        const float* m = mask + (int64_t)r * hw;
        for (int base = 0; base < hw; base += 1024) {
            int c = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = base + u * 256 + threadIdx.x;
                if (i < hw && m[i] > 0.5f) ++c;
            }
            if (c) atomicAdd(&cnt, c);
            __syncthreads();
            const bool decided = cnt > 1;                         // every thread reads the same value ...
            __syncthreads();                                      // ... before anyone adds to it again
            if (decided) break;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) keep[r] = (other && cnt > 1) ? 1 : 0;
}

// the same rule from the masks' bit words (stm_lincomb_sigmoid_crop_bits_f32: bit = value > 0.5): one wave per row, popcounts
__global__ __launch_bounds__(256) void keep_flags_bits_kernel(const unsigned long long* __restrict__ bits, const float* __restrict__ score,
                                                              const int* __restrict__ tracked, int* __restrict__ keep, int n_rows, int words,
                                                              int max_age, float thr)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n_rows) return;
    const bool other = tracked[r] <= max_age && score[r] > thr;   // wave-uniform
    int c = 0;
    if (other) {
        const unsigned long long* w = bits + (int64_t)r * words;
        for (int i = lane; i < words; i += 64) c += __popcll(w[i]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, STM_WAVE);
    }
    if (lane == 0) keep[r] = (other && c > 1) ? 1 : 0;
}

// one block per clip: kept rows of the clip, in row order, become rows 0.. of out[clip]: (box 4, score, class, object id =
// row index within the clip, 1, mask coefficients); rows past the kept count (and past top_k) are zero
__global__ __launch_bounds__(256) void pack_tracked_kernel(const int* __restrict__ keep, const int* __restrict__ off, const float* __restrict__ box,
                                                           const float* __restrict__ score, const int64_t* __restrict__ cls,
                                                           const float* __restrict__ coeff, float* __restrict__ out, int top_k, int cols, int mdim)
{
    __shared__ int wave_cnt[4];
    __shared__ int running;
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int p0 = off[b], p1 = off[b + 1];
    float* ob = out + (int64_t)b * top_k * cols;
    if (tid == 0) running = 0;
    __syncthreads();
    for (int base = p0; base < p1; base += 256) {
        const int r = base + tid;
        const bool f = r < p1 && keep[r] != 0;
        const unsigned long long bal = __ballot(f);
        const int lane_prefix = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int wp = 0, tot = 0;
        for (int w = 0; w < 4; ++w) {
            if (w < wave) wp += wave_cnt[w];
            tot += wave_cnt[w];
        }
        const int start = running;
        const int pos = start + wp + lane_prefix;
        if (f && pos < top_k) {
            float* o = ob + (int64_t)pos * cols;
            const float4 bx = reinterpret_cast<const float4*>(box)[r];
            o[0] = bx.x; o[1] = bx.y; o[2] = bx.z; o[3] = bx.w;
            o[4] = score[r];
            o[5] = (float)cls[r];
            o[6] = (float)(r - p0);
            o[7] = 1.0f;
            for (int c = 0; c < mdim; ++c) o[8 + c] = coeff[(int64_t)r * mdim + c];
        }
        __syncthreads();
        if (tid == 0) running = start + tot;
        __syncthreads();
    }
    const int kept = min(running, top_k);
    for (int i = kept * cols + tid; i < top_k * cols; i += 256) ob[i] = 0.0f;
}

}  // namespace

extern "C" int stm_gather_detections_f32(const int64_t* idx, const int64_t* cls, const float* score, const float* box, const int* count,
                                         const float* mask_coeff, const float* track, const float* centerness, int B, int top_k, int N,
                                         int mask_dim, int embed_dim, int D, float* out_box, int64_t* out_cls, float* out_score,
                                         float* out_coeff, float* out_track, float* out_centerness, int* out_clip, stm_stream_t stream)
{
    STM_REQUIRE(B > 0 && top_k > 0 && N > 0 && mask_dim > 0 && embed_dim > 0 && D >= 0 && B <= 65535, STM_EINVAL,
                "stm_gather_detections_f32: bad sizes");
    if (D == 0) return STM_OK;
    STM_REQUIRE(idx && cls && score && box && count && mask_coeff && track && out_box && out_cls && out_score && out_coeff && out_track &&
                    out_clip, STM_ENULL, "stm_gather_detections_f32: NULL argument");
    GatherDetArgs a;
    a.idx = idx; a.cls = cls; a.score = score; a.box = box; a.cnt = count; a.coeff = mask_coeff; a.track = track; a.cen = centerness;
    a.o_box = out_box; a.o_cls = out_cls; a.o_score = out_score; a.o_coeff = out_coeff; a.o_track = out_track; a.o_cen = out_centerness;
    a.o_clip = out_clip; a.B = B; a.top_k = top_k; a.N = N; a.mdim = mask_dim; a.edim = embed_dim; a.D = D;
    hipLaunchKernelGGL(gather_detections_kernel, dim3(top_k, B), dim3(64), 0, stm_hs(stream), a);
    STM_CHECK_LAUNCH("gather_detections_kernel");
    return STM_OK;
}

extern "C" int stm_shift_rois_f32(const float* box, const int* clip, float* rois, int n, int feat_h, int feat_w, stm_stream_t stream)
{
    STM_REQUIRE(n >= 0 && feat_h > 0 && feat_w > 0, STM_EINVAL, "stm_shift_rois_f32: bad sizes");
    if (n == 0) return STM_OK;
    STM_REQUIRE(box && clip && rois, STM_ENULL, "stm_shift_rois_f32: NULL argument");
    STM_REQUIRE((uintptr_t)box % 16 == 0, STM_EINVAL, "stm_shift_rois_f32: boxes must be 16-byte aligned");
    hipLaunchKernelGGL(shift_rois_kernel, dim3(stm_cdiv(n, 256)), dim3(256), 0, stm_hs(stream), box, clip, rois, n, feat_h, feat_w);
    STM_CHECK_LAUNCH("shift_rois_kernel");
    return STM_OK;
}

extern "C" int stm_shift_apply_f32(const float* loc_shift, const float* coeff_shift, float* box, float* coeff, float* score, int n, int mask_dim,
                                   float score_decay, stm_stream_t stream)
{
    STM_REQUIRE(n >= 0 && mask_dim > 0, STM_EINVAL, "stm_shift_apply_f32: bad sizes");
    if (n == 0) return STM_OK;
    STM_REQUIRE(loc_shift && coeff_shift && box && coeff && score, STM_ENULL, "stm_shift_apply_f32: NULL argument");
    STM_REQUIRE((uintptr_t)box % 16 == 0 && (uintptr_t)loc_shift % 16 == 0, STM_EINVAL, "stm_shift_apply_f32: box / loc must be 16-byte aligned");
    const int64_t work = (int64_t)n * mask_dim;
    hipLaunchKernelGGL(shift_apply_kernel, dim3(stm_cdiv(work, 256)), dim3(256), 0, stm_hs(stream), loc_shift, coeff_shift, box, coeff, score, n,
                       mask_dim, score_decay);
    STM_CHECK_LAUNCH("shift_apply_kernel");
    return STM_OK;
}

namespace {
int launch_match_scores(const char* who, const float* cos, const float* det_track, const float* prev_track, int E, const float* mask_iou,
                        const float* det_box, const float* prev_box, const float* det_score, const int64_t* det_cls, const int64_t* prev_cls,
                        const int* det_clip, const int* prev_offsets, int D, int Pn, const float* coeff4, float dummy_iou, int* match,
                        stm_stream_t stream);
}

extern "C" int stm_match_scores_f32(const float* cos, const float* mask_iou, const float* det_box, const float* prev_box, const float* det_score,
                                    const int64_t* det_cls, const int64_t* prev_cls, const int* det_clip, const int* prev_offsets, int D, int Pn,
                                    const float* coeff4, float dummy_iou, int* match, stm_stream_t stream)
{
    STM_REQUIRE(D <= 0 || Pn <= 0 || cos, STM_ENULL, "stm_match_scores_f32: cos is NULL");
    return launch_match_scores("stm_match_scores_f32", cos, nullptr, nullptr, 0, mask_iou, det_box, prev_box, det_score, det_cls, prev_cls, det_clip,
                               prev_offsets, D, Pn, coeff4, dummy_iou, match, stream);
}

extern "C" int stm_match_scores_embed_f32(const float* det_track, const float* prev_track, int embed_dim, const float* mask_iou,
                                          const float* det_box, const float* prev_box, const float* det_score, const int64_t* det_cls,
                                          const int64_t* prev_cls, const int* det_clip, const int* prev_offsets, int D, int Pn,
                                          const float* coeff4, float dummy_iou, int* match, stm_stream_t stream)
{
    STM_REQUIRE(embed_dim > 0, STM_EINVAL, "stm_match_scores_embed_f32: embed_dim must be positive");
    STM_REQUIRE(D <= 0 || Pn <= 0 || (det_track && prev_track), STM_ENULL, "stm_match_scores_embed_f32: NULL embedding table");
    STM_REQUIRE(embed_dim % 4 != 0 || ((uintptr_t)det_track % 16 == 0 && (uintptr_t)prev_track % 16 == 0), STM_EINVAL,
                "stm_match_scores_embed_f32: embedding tables must be 16-byte aligned");
    return launch_match_scores("stm_match_scores_embed_f32", nullptr, det_track, prev_track, embed_dim, mask_iou, det_box, prev_box, det_score,
                               det_cls, prev_cls, det_clip, prev_offsets, D, Pn, coeff4, dummy_iou, match, stream);
}

namespace {
int launch_match_scores(const char* who, const float* cos, const float* det_track, const float* prev_track, int E, const float* mask_iou,
                        const float* det_box, const float* prev_box, const float* det_score, const int64_t* det_cls, const int64_t* prev_cls,
                        const int* det_clip, const int* prev_offsets, int D, int Pn, const float* coeff4, float dummy_iou, int* match,
                        stm_stream_t stream)
{
    STM_REQUIRE(D >= 0 && Pn >= 0, STM_EINVAL, "%s: bad sizes", who);
    if (D == 0) return STM_OK;
    STM_REQUIRE(det_box && det_score && det_cls && det_clip && prev_offsets && coeff4 && match, STM_ENULL, "%s: NULL argument", who);
    STM_REQUIRE(Pn == 0 || (mask_iou && prev_box && prev_cls), STM_ENULL, "%s: NULL prev argument", who);
    STM_REQUIRE((uintptr_t)det_box % 16 == 0 && (uintptr_t)prev_box % 16 == 0, STM_EINVAL, "%s: boxes must be 16-byte aligned", who);
    MatchArgs a;
    a.cos = cos; a.det_track = det_track; a.prev_track = prev_track; a.E = E; a.miou = mask_iou; a.det_box = det_box; a.prev_box = prev_box; a.det_score = det_score; a.det_cls = det_cls; a.prev_cls = prev_cls;
    a.det_clip = det_clip; a.prev_off = prev_offsets; a.match = match; a.D = D; a.Pn = Pn;
    a.c0 = coeff4[0]; a.c1 = coeff4[1]; a.c2 = coeff4[2]; a.c3 = coeff4[3]; a.dummy = dummy_iou;
    hipLaunchKernelGGL(match_scores_kernel, dim3(stm_cdiv(D, 4)), dim3(256), 0, stm_hs(stream), a);
    STM_CHECK_LAUNCH("match_scores_kernel");
    return STM_OK;
}
}  // namespace

extern "C" int stm_gather_rows2(const void* const* a_rows, const void* const* b_rows, void* const* out_rows, const int* row_bytes, int n_tensors,
                                const int* plan, int n_rows, int n_a, stm_stream_t stream)
{
    STM_REQUIRE(n_tensors > 0 && n_tensors <= 8 && n_rows >= 0 && n_a >= 0, STM_EINVAL, "stm_gather_rows2: bad sizes (1..8 tensors)");
    if (n_rows == 0) return STM_OK;
    STM_REQUIRE(a_rows && b_rows && out_rows && row_bytes && plan, STM_ENULL, "stm_gather_rows2: NULL argument");
    GatherRowsArgs g;
    g.n_tensors = n_tensors; g.n_rows = n_rows; g.n_a = n_a; g.plan = plan;
    for (int t = 0; t < 8; ++t) {
        g.a[t] = g.b[t] = nullptr; g.out[t] = nullptr; g.row_bytes[t] = 0; g.vec[t] = 0;
        if (t >= n_tensors) continue;
        STM_REQUIRE(row_bytes[t] > 0 && row_bytes[t] % 4 == 0 && out_rows[t], STM_EINVAL, "stm_gather_rows2: tensor %d: row bytes must be a positive multiple of 4", t);
        g.a[t] = static_cast<const uint8_t*>(a_rows[t]); g.b[t] = static_cast<const uint8_t*>(b_rows[t]); g.out[t] = static_cast<uint8_t*>(out_rows[t]);
        g.row_bytes[t] = row_bytes[t];
        g.vec[t] = row_bytes[t] % 16 == 0 && (uintptr_t)g.a[t] % 16 == 0 && (uintptr_t)g.b[t] % 16 == 0 && (uintptr_t)g.out[t] % 16 == 0;
    }
    hipLaunchKernelGGL(gather_rows2_kernel, dim3(n_rows), dim3(256), 0, stm_hs(stream), g);
    STM_CHECK_LAUNCH("gather_rows2_kernel");
    return STM_OK;
}

extern "C" int stm_pack_tracked_f32(const float* mask, const float* score, const int* tracked, const int* offsets, const float* box,
                                    const int64_t* cls, const float* mask_coeff, int n_rows, int hw, int B, int top_k, int cols, int mask_dim,
                                    int max_age, float score_thr, int* keep_ws, float* out, stm_stream_t stream)
{
    STM_REQUIRE(n_rows >= 0 && hw > 0 && B > 0 && top_k > 0 && mask_dim > 0 && cols >= 8 + mask_dim, STM_EINVAL, "stm_pack_tracked_f32: bad sizes");
    STM_REQUIRE(offsets && out, STM_ENULL, "stm_pack_tracked_f32: NULL argument");
    STM_REQUIRE(n_rows == 0 || (mask && score && tracked && box && cls && mask_coeff && keep_ws), STM_ENULL, "stm_pack_tracked_f32: NULL row argument");
    STM_REQUIRE((uintptr_t)box % 16 == 0, STM_EINVAL, "stm_pack_tracked_f32: boxes must be 16-byte aligned");
    if (n_rows > 0) {
        hipLaunchKernelGGL(keep_flags_kernel, dim3(n_rows), dim3(256), 0, stm_hs(stream), mask, score, tracked, keep_ws, hw, max_age, score_thr);
        STM_CHECK_LAUNCH("keep_flags_kernel");
    }
    hipLaunchKernelGGL(pack_tracked_kernel, dim3(B), dim3(256), 0, stm_hs(stream), keep_ws, offsets, box, score, cls, mask_coeff, out, top_k, cols,
                       mask_dim);
    STM_CHECK_LAUNCH("pack_tracked_kernel");
    return STM_OK;
}

extern "C" int stm_pack_tracked_bits_f32(const uint64_t* mask_bits, int words, const float* score, const int* tracked, const int* offsets,
                                         const float* box, const int64_t* cls, const float* mask_coeff, int n_rows, int B, int top_k, int cols,
                                         int mask_dim, int max_age, float score_thr, int* keep_ws, float* out, stm_stream_t stream)
{
    STM_REQUIRE(n_rows >= 0 && words > 0 && B > 0 && top_k > 0 && mask_dim > 0 && cols >= 8 + mask_dim, STM_EINVAL, "stm_pack_tracked_bits_f32: bad sizes");
    STM_REQUIRE(offsets && out, STM_ENULL, "stm_pack_tracked_bits_f32: NULL argument");
    STM_REQUIRE(n_rows == 0 || (mask_bits && score && tracked && box && cls && mask_coeff && keep_ws), STM_ENULL, "stm_pack_tracked_bits_f32: NULL row argument");
    STM_REQUIRE((uintptr_t)box % 16 == 0, STM_EINVAL, "stm_pack_tracked_bits_f32: boxes must be 16-byte aligned");
    if (n_rows > 0) {
        hipLaunchKernelGGL(keep_flags_bits_kernel, dim3(stm_cdiv(n_rows, 4)), dim3(256), 0, stm_hs(stream),
                           reinterpret_cast<const unsigned long long*>(mask_bits), score, tracked, keep_ws, n_rows, words, max_age, score_thr);
        STM_CHECK_LAUNCH("keep_flags_bits_kernel");
    }
    hipLaunchKernelGGL(pack_tracked_kernel, dim3(B), dim3(256), 0, stm_hs(stream), keep_ws, offsets, box, score, cls, mask_coeff, out, top_k, cols,
                       mask_dim);
    STM_CHECK_LAUNCH("pack_tracked_kernel");
    return STM_OK;
}
