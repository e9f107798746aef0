// mask_ops.hip -- prototype linear combination + sigmoid + crop, and binary mask IoU, for gfx950.
//
// lincomb replaces generate_mask (layers/mask_utils.py:111-128) + crop / sanitize_coordinates
// (layers/box_utils.py:298-364): the reference runs a matmul, tanh, sigmoid, 8 element-wise kernels and a
// permute copy; here it is one write-dominated pass (bytes = proto 1.97 MB + n * 61 KB).
//   One thread = one prototype pixel (its 32 prototype values stay in registers); a workgroup covers 256
//   consecutive pixels x DCHUNK detections; tanh(coeff) and the crop rectangles of the chunk sit in LDS and are
//   read as broadcasts; each output row [d][pixel] is written with coalesced stores.  Pixels outside the crop
//   rectangle skip the dot product and store 0.
//
// mask_iou replaces layers/box_utils.py:435-447 applied to m.gt(0.5).float() (track_TF.py:85,107): masks are
// bit-packed with wavefront ballots (64 pixels -> one 64-bit word), pairs are reduced with popcount.
#include "stm_common.h"

namespace {

constexpr int LC_DCHUNK = 32;  // most detections per workgroup (it holds the prototypes of its 256 pixels in registers and walks its rows:
                               // every chunk re-reads them, 128 B per pixel -- 880 MB per launch at 8 rows per chunk and 3 600 rows)

#ifndef LC_ABL
#define LC_ABL 0      // timing ablations (RESULTS WRONG): 1 no prototype loads, 2 no mask stores, 4 no bit words
#endif
template <int M>
__global__ __launch_bounds__(256) void lincomb_kernel(const float* __restrict__ proto, const float* __restrict__ coeff,
                                                      const float* __restrict__ boxes, float* __restrict__ out, int h,
                                                      int w, int n, int apply_tanh, const int* __restrict__ n_dev,
                                                      const int* __restrict__ row_proto, unsigned long long* __restrict__ bits,
                                                      float bits_thr, int dchunk)
{
    __shared__ float sc[LC_DCHUNK * M];
    __shared__ float sb[LC_DCHUNK * 4];  // x1, x2, y1, y2 (float bounds, padding 1)
    const int hw = h * w;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int d0 = blockIdx.y * dchunk;
    const int nd = min(dchunk, n - d0);
    const int n_valid = n_dev ? min(max(*n_dev, 0), n) : n;

    for (int idx = threadIdx.x; idx < nd * M; idx += 256) {
        float v = coeff[(int64_t)d0 * M + idx];
        sc[idx] = apply_tanh ? tanhf(v) : v;
    }
    if (threadIdx.x < nd) {
        float x1 = 0.f, x2 = (float)w, y1 = 0.f, y2 = (float)h;
        if (boxes) {
            const float* b = boxes + (int64_t)(d0 + threadIdx.x) * 4;
            stm_sanitize(b[0], b[2], w, 1, x1, x2);
            stm_sanitize(b[1], b[3], h, 1, y1, y2);
        }
        sb[threadIdx.x * 4 + 0] = x1;
        sb[threadIdx.x * 4 + 1] = x2;
        sb[threadIdx.x * 4 + 2] = y1;
        sb[threadIdx.x * 4 + 3] = y2;
    }
    // Which rows' crop rectangles touch this workgroup's 256 pixels at all.  The tracked sets of a 32-clip step are ~3 600 rows whose boxes cover ~12 %
    // of the frame: four waves walking every row pixel by pixel issued ~40 instructions per wave and row to store zeros (80 of the launch's 108 us with
    // every store removed).  Rows that miss the workgroup's pixel span are zero-filled by ONE 16-byte store per lane (a wave covers the whole 256-pixel
    // segment of a row; the waves take such rows in turn) and skipped by the pixel loop below.
    __shared__ int hit[LC_DCHUNK];
    const int p0 = blockIdx.x * 256;
    const bool vec_ok = (hw & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;       // row bases are then 16-byte aligned and a quad never straddles the end of a row
    if (threadIdx.x < nd) {
        const int d = threadIdx.x;
        const int pl = min(p0 + 255, hw - 1);
        const int ya = p0 / w, yb = pl / w;
        const float fya = (float)ya, fyb = (float)yb;
        bool touch = (d0 + d < n_valid) && fyb >= sb[d * 4 + 2] && fya < sb[d * 4 + 3];
        if (touch && ya == yb) {                             // one image row: the x span decides too
            const float fxa = (float)(p0 - ya * w), fxb = (float)(pl - ya * w);
            touch = fxb >= sb[d * 4] && fxa < sb[d * 4 + 1];
        }
        hit[d] = (touch || !vec_ok) ? 1 : 0;
    }
    __syncthreads();
    // bits != null: also the binarised mask (v > bits_thr) as 64-pixel words [n][ceil(hw / 64)] -- what mask IoU consumes
    // (box_utils.py:435-447 on m.gt(0.5)); a wave is 64 consecutive pixels, so the word is one ballot.  Waves that lie
    // entirely past the last pixel leave; lanes past it inside the last wave stay for the ballot and vote 0.
    const bool live = pix < hw;
    const int words = (hw + 63) >> 6;
    if (p0 < hw) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const int q = p0 + 4 * lane;                         // this lane's quad of the 256-pixel segment
        for (int d = wv; d < nd; d += 4) {
            if (hit[d]) continue;
            if (q < hw && !(LC_ABL & 2)) *reinterpret_cast<float4*>(out + (int64_t)(d0 + d) * hw + q) = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bits && lane < 4 && (p0 >> 6) + lane < words && !(LC_ABL & 4)) bits[(int64_t)(d0 + d) * words + (p0 >> 6) + lane] = 0ull;
        }
    }
    if (!bits && !live) return;
    if (bits && ((pix & ~63) >= hw)) return;

    float p[M];
    int cur = -1;  // prototype set currently held in registers (rows of several frames may share one launch)
    const int y = pix / w, x = pix - y * w;
    const float fx = (float)x, fy = (float)y;
    for (int d = 0; d < nd; ++d) {
        if (!hit[d]) continue;                               // zero-filled above
        const int want = row_proto ? row_proto[d0 + d] : 0;  // wave-uniform
        if (want != cur && live) {
            cur = want;
            const float4* pr = reinterpret_cast<const float4*>(proto + ((int64_t)cur * hw + pix) * M);
#pragma unroll
            for (int q = 0; q < M / 4; ++q) {
                float4 v = (LC_ABL & 1) ? make_float4(0.01f * q, 0.02f, 0.03f, 0.04f) : pr[q];
                p[4 * q] = v.x; p[4 * q + 1] = v.y; p[4 * q + 2] = v.z; p[4 * q + 3] = v.w;
            }
        }
        float v = 0.0f;
        const bool inside = live && (d0 + d < n_valid) && fx >= sb[d * 4] && fx < sb[d * 4 + 1] && fy >= sb[d * 4 + 2] &&
                            fy < sb[d * 4 + 3];
        if (inside) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < M; ++k) acc = fmaf(p[k], sc[d * M + k], acc);
            v = 1.0f / (1.0f + expf(-acc));
        }
        if (live && (!(LC_ABL & 2) || v == 12345.678f)) out[(int64_t)(d0 + d) * hw + pix] = v;
        if (bits && !(LC_ABL & 4)) {
            const unsigned long long bal = __ballot(v > bits_thr);
            if ((threadIdx.x & 63) == 0) bits[(int64_t)(d0 + d) * words + (pix >> 6)] = bal;
        }
    }
}

// 64 consecutive pixels -> one 64-bit word via ballot.  grid: (ceil(words/4), n), block 256 = 4 waves = 4 words
__global__ __launch_bounds__(256) void mask_pack_kernel(const float* __restrict__ m, unsigned long long* __restrict__ bits,
                                                        int hw, int words, float thr)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int word = blockIdx.x * 4 + wave;
    const int i = blockIdx.y;
    if (word >= words) return;
    const int pix = word * 64 + lane;
    bool b = false;
    if (pix < hw) b = m[(int64_t)i * hw + pix] > thr;
    unsigned long long bal = __ballot(b);
    if (lane == 0) bits[(int64_t)i * words + word] = bal;
}

// Small problems (fewer than MIOU_BIG_ROWS rows: single-stream and small-batch steps): grid (ceil(n2/256), n1), row i staged in LDS, one thread per
// column j -- many tiny workgroups that finish in ~20-40 us whatever the batch, where the row-block kernel below needs a workgroup's ~100 us
__global__ __launch_bounds__(256) void mask_iou_pairs_small_kernel(const unsigned long long* __restrict__ b1,
                                                             const unsigned long long* __restrict__ b2, int n2, int words,
                                                             float* __restrict__ out, const int* __restrict__ g1,
                                                             const int* __restrict__ g2)
{
    extern __shared__ unsigned long long arow[];
    const int i = blockIdx.y;
    // group ids (the clip a mask belongs to in the batched pipeline): pairs of different groups are never compared by the
    // caller -- their IoU is left at 0 and their words are not read.  A workgroup none of whose 256 columns belongs to row i's
    // group leaves at once (rows sorted by group, as the pipeline keeps them, make that the common case; any order is correct).
    const int gi = g1 ? g1[i] : 0;
    if (g2) {
        const int j = blockIdx.x * 256 + threadIdx.x;
        const bool mine = j < n2 && g2[j] == gi;
        if (!__syncthreads_or(mine ? 1 : 0)) {
            if (j < n2) out[(int64_t)i * n2 + j] = 0.0f;
            return;
        }
    }
    for (int t = threadIdx.x; t < words; t += 256) arow[t] = b1[(int64_t)i * words + t];
    __syncthreads();
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n2) return;
    if (g2 && g2[j] != gi) { out[(int64_t)i * n2 + j] = 0.0f; return; }
    int inter = 0, a1 = 0, a2 = 0;
    const unsigned long long* q = b2 + (int64_t)j * words;
    for (int t = 0; t < words; ++t) {
        unsigned long long a = arow[t], b = q[t];
        inter += __popcll(a & b);
        a1 += __popcll(a);
        a2 += __popcll(b);
    }
    float fi = (float)inter, uni = ((float)a1 + (float)a2) - fi;
    out[(int64_t)i * n2 + j] = (uni == 0.0f) ? 0.0f : fi / uni;
}

// grid: ceil(n1 / R) workgroups; R <= 8 rows of the first set staged in LDS, so a column's words are fetched once for eight pairs (one row per
// workgroup read 1.1 GB through L2 per step for 17 MB of masks).  A wave takes the columns j in chunks of 64: a lane checks one column's group against
// the rows' groups (the clip a mask belongs to in the batched pipeline: pairs of different groups are never compared by the caller, their IoU is 0 and
// their words are not read), writes the zeros of the unmatched pairs and lists the matched columns; those are then dealt to the waves FOUR at a time, one per DPP
// row of 16 lanes: lane s of a row reads words s, s + 16, ... of its column (128 contiguous bytes per row and step), counts the eight intersections
// against the staged rows (the same LDS word for all four rows: a broadcast) and the column's area, and the nine counts are summed over the row's 16
// lanes with DPP moves (no LDS round trips: with __shfl butterflies over 64 lanes the reductions were most of the kernel).  Integer counts: the same
// IoU bits as a one-thread-per-pair loop.
constexpr int MIOU_R = 8, MIOU_LIST = 2048, MIOU_BIG_ROWS = 3072;   // (crossover of the two kernels at 110 detections per clip: 124 vs 111 us at 3 520 rows, 92 vs 108 at 2 640; scripts/sweep_miou_rule.py)
// staged rows (dynamic LDS) beside the kernel's 16.5 KB of static LDS (matched-column list) inside the 64 KB a kernel gets without an attribute:
// long mask rows (config 5: 920 words) stage 6 rows per workgroup instead of 8.  STM_MIOU_BIG_ROWS (read once) moves the rule's threshold (tests).
constexpr int MIOU_DYN_LDS = 46 * 1024;
__global__ __launch_bounds__(256) void mask_iou_pairs_kernel(const unsigned long long* __restrict__ b1,
                                                             const unsigned long long* __restrict__ b2, int n1, int n2, int words,
                                                             float* __restrict__ out, const int* __restrict__ g1,
                                                             const int* __restrict__ g2, int R)
{
    extern __shared__ unsigned long long arow[];               // [R][words]
    __shared__ int a1_s[MIOU_R], gi_s[MIOU_R];
    const int i0 = blockIdx.x * R, nr = min(R, n1 - i0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int r = wave; r < MIOU_R; r += 4) {                   // a wave stages two rows and counts their bits
        unsigned pa = 0;
        for (int t = lane; t < words && r < nr; t += 64) {
            const unsigned long long a = b1[(int64_t)(i0 + r) * words + t];
            arow[r * words + t] = a;
            pa += (unsigned)__popcll(a);
        }
        pa = stm_row16_sum(pa);
        const int tot = stm_wave_sum_rows(pa);
        if (lane == 0) { a1_s[r] = tot; gi_s[r] = r < nr ? (g1 ? g1[i0 + r] : 0) : -0x7fffffff; }
    }
    __syncthreads();
    int gi[MIOU_R];
#pragma unroll
    for (int r = 0; r < MIOU_R; ++r) gi[r] = gi_s[r];
    const int q = lane >> 4, sl = lane & 15;                   // DPP row of the lane (= which of the four columns of a pass), lane within the row
    // pass 1: every column's group against the rows' groups; zeros for the unmatched pairs; the matched columns (with their match bits) are appended to
    // a list in LDS -- chunk by chunk in column order, each wave's chunk at the offset the workgroup's running count gives it: a deterministic order
    __shared__ int list_j[MIOU_LIST], list_m[MIOU_LIST], chunk_n[4], list_n;
    if (tid == 0) list_n = 0;
    __syncthreads();
    for (int j0 = 0; j0 < n2; j0 += 256) {
        const int j = j0 + tid;
        const int gj = j < n2 ? (g2 ? g2[j] : 0) : 0x7fffffff;
        unsigned match = 0;                                    // bit r: row r and column j are of one group
#pragma unroll
        for (int r = 0; r < MIOU_R; ++r) match |= (unsigned)(r < nr && gj == gi[r]) << r;
        if (j < n2) {
#pragma unroll
            for (int r = 0; r < MIOU_R; ++r)
                if (r < nr && !((match >> r) & 1u)) out[(int64_t)(i0 + r) * n2 + j] = 0.0f;
        }
        const unsigned long long mb = __ballot(match != 0);
        if (lane == 0) chunk_n[wave] = __popcll(mb);
        __syncthreads();
        int base = list_n;
        for (int w = 0; w < wave; ++w) base += chunk_n[w];
        if (match) {
            const int pos = base + __popcll(mb & ((1ull << lane) - 1ull));
            if (pos < MIOU_LIST) { list_j[pos] = j; list_m[pos] = (int)match; }
        }
        __syncthreads();
        if (tid == 0) list_n += chunk_n[0] + chunk_n[1] + chunk_n[2] + chunk_n[3];
        __syncthreads();
    }
    const int nlist = min(list_n, MIOU_LIST);
    // pass 2: the matched columns, four per wave and pass (one per DPP row), dealt round-robin to the waves; all of a column's words of a lane in flight
    for (int p0 = wave * 4; p0 < nlist; p0 += 16) {
        const int pi = min(p0 + q, nlist - 1);
        const int jq = list_j[pi];
        const unsigned mq = (unsigned)list_m[pi];
        const unsigned long long* col = b2 + (int64_t)jq * words;
        unsigned hi[MIOU_R], lo = 0;
#pragma unroll
        for (int r = 0; r < MIOU_R; ++r) hi[r] = 0;
        for (int t0 = sl; t0 < words; t0 += 16 * 16) {
            unsigned long long c[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) c[u] = t0 + 16 * u < words ? col[t0 + 16 * u] : 0ull;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int t = t0 + 16 * u;
                if (t < words) {
                    lo += (unsigned)__popcll(c[u]);
#pragma unroll
                    for (int r = 0; r < MIOU_R; ++r)
                        if (r < nr) hi[r] += (unsigned)__popcll(arow[r * words + t] & c[u]);
                }
            }
        }
        lo = stm_row16_sum(lo);
#pragma unroll
        for (int r = 0; r < MIOU_R; ++r) hi[r] = stm_row16_sum(hi[r]);
        // lane r of DPP row q writes pair (row r, this row's column) if they are of one group
        if (sl < MIOU_R && p0 + q < nlist) {
            unsigned h = 0;
#pragma unroll
            for (int r = 0; r < MIOU_R; ++r)
                if (sl == r) h = hi[r];
            if (sl < nr && ((mq >> sl) & 1u)) {
                const float fi = (float)h, uni = ((float)a1_s[sl] + (float)lo) - fi;
                out[(int64_t)(i0 + sl) * n2 + jq] = (uni == 0.0f) ? 0.0f : fi / uni;
            }
        }
    }
    // (more matched columns than the list holds -- over 2 048 masks of one group against a block of rows: the rest one column per pass, in column order)
    if (list_n > MIOU_LIST) {
        for (int j = 0, seen = 0; j < n2; ++j) {
            const int gj = g2 ? g2[j] : 0;
            unsigned match = 0;
#pragma unroll
            for (int r = 0; r < MIOU_R; ++r) match |= (unsigned)(r < nr && gj == gi[r]) << r;
            if (!match) continue;
            if (seen++ < MIOU_LIST) continue;
            if (wave != 0) continue;
            const unsigned long long* col = b2 + (int64_t)j * words;
            for (int r = 0; r < nr; ++r) {
                if (!((match >> r) & 1u)) continue;
                unsigned h = 0, o = 0;
                for (int t = lane; t < words; t += 64) { const unsigned long long c = col[t]; o += (unsigned)__popcll(c); h += (unsigned)__popcll(arow[r * words + t] & c); }
                const int hs = stm_wave_sum_rows(stm_row16_sum(h)), os = stm_wave_sum_rows(stm_row16_sum(o));
                if (lane == 0) {
                    const float fi = (float)hs, uni = ((float)a1_s[r] + (float)os) - fi;
                    out[(int64_t)(i0 + r) * n2 + j] = (uni == 0.0f) ? 0.0f : fi / uni;
                }
            }
        }
    }
}

}  // namespace

extern "C" int stm_lincomb_sigmoid_crop_bits_f32(const float* proto, const float* coeff, const float* boxes, float* out, int h, int w, int m,
                                                 int n, int apply_tanh, const int* n_dev, const int* row_proto, uint64_t* bits, float bits_thr,
                                                 stm_stream_t stream);
extern "C" int stm_lincomb_sigmoid_crop_f32(const float* proto, const float* coeff, const float* boxes, float* out, int h,
                                            int w, int m, int n, int apply_tanh, const int* n_dev, const int* row_proto,
                                            stm_stream_t stream)
{
    return stm_lincomb_sigmoid_crop_bits_f32(proto, coeff, boxes, out, h, w, m, n, apply_tanh, n_dev, row_proto, nullptr, 0.5f, stream);
}

extern "C" int stm_lincomb_sigmoid_crop_bits_f32(const float* proto, const float* coeff, const float* boxes, float* out, int h, int w, int m,
                                                 int n, int apply_tanh, const int* n_dev, const int* row_proto, uint64_t* bits_out, float bits_thr,
                                                 stm_stream_t stream)
{
    unsigned long long* bits = reinterpret_cast<unsigned long long*>(bits_out);
    STM_REQUIRE(n >= 0, STM_EINVAL, "stm_lincomb_sigmoid_crop_f32: n=%d", n);
    if (n == 0) return STM_OK;
    STM_REQUIRE(proto && coeff && out, STM_ENULL, "stm_lincomb_sigmoid_crop_f32: proto/coeff/out must be non-NULL");
    STM_REQUIRE(h > 0 && w > 0, STM_EINVAL, "stm_lincomb_sigmoid_crop_f32: bad mask size %dx%d", h, w);
    STM_REQUIRE((uintptr_t)proto % 16 == 0, STM_EINVAL, "stm_lincomb_sigmoid_crop_f32: proto must be 16-byte aligned");
    // rows per workgroup: 8 while that still gives every CU several workgroups, 32 on the big tracked sets (less prototype re-reading)
    const int dchunk = (int64_t)stm_cdiv((int64_t)h * w, 256) * stm_cdiv(n, 32) >= 2048 ? 32 : 8;
    dim3 grid(stm_cdiv((int64_t)h * w, 256), stm_cdiv(n, dchunk));
    STM_REQUIRE(grid.y <= 65535, STM_EINVAL, "stm_lincomb_sigmoid_crop_f32: n=%d too large", n);
    if (m == 32) {
        hipLaunchKernelGGL(lincomb_kernel<32>, grid, dim3(256), 0, stm_hs(stream), proto, coeff, boxes, out, h, w, n,
                           apply_tanh, n_dev, row_proto, bits, bits_thr, dchunk);
    } else if (m == 8) {
        hipLaunchKernelGGL(lincomb_kernel<8>, grid, dim3(256), 0, stm_hs(stream), proto, coeff, boxes, out, h, w, n,
                           apply_tanh, n_dev, row_proto, bits, bits_thr, dchunk);
    } else if (m == 64) {
        hipLaunchKernelGGL(lincomb_kernel<64>, grid, dim3(256), 0, stm_hs(stream), proto, coeff, boxes, out, h, w, n,
                           apply_tanh, n_dev, row_proto, bits, bits_thr, dchunk);
    } else {
        STM_REQUIRE(false, STM_EUNSUPPORTED, "stm_lincomb_sigmoid_crop_f32: mask_dim %d not in {8,32,64}", m);
    }
    STM_CHECK_LAUNCH("lincomb_kernel");
    return STM_OK;
}

extern "C" size_t stm_mask_iou_workspace_bytes(int n1, int n2, int hw)
{
    size_t words = (size_t)((hw + 63) / 64);
    return ((size_t)n1 + (size_t)n2) * words * 8 + 64;
}

extern "C" int stm_mask_iou_grouped_f32(const float* m1, int n1, const float* m2, int n2, int hw, float thr, float* out, const int* group1,
                                        const int* group2, void* workspace, size_t workspace_bytes, stm_stream_t stream);
extern "C" int stm_mask_iou_f32(const float* m1, int n1, const float* m2, int n2, int hw, float thr, float* out,
                                void* workspace, size_t workspace_bytes, stm_stream_t stream)
{
    return stm_mask_iou_grouped_f32(m1, n1, m2, n2, hw, thr, out, nullptr, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int stm_mask_iou_grouped_f32(const float* m1, int n1, const float* m2, int n2, int hw, float thr, float* out, const int* group1,
                                        const int* group2, void* workspace, size_t workspace_bytes, stm_stream_t stream)
{
    STM_REQUIRE((group1 == nullptr) == (group2 == nullptr), STM_EINVAL, "stm_mask_iou_grouped_f32: give both group arrays or neither");
    STM_REQUIRE(n1 >= 0 && n2 >= 0 && hw > 0, STM_EINVAL, "stm_mask_iou_f32: bad sizes");
    if (n1 == 0 || n2 == 0) return STM_OK;
    STM_REQUIRE(m1 && m2 && out, STM_ENULL, "stm_mask_iou_f32: m1/m2/out must be non-NULL");
    STM_REQUIRE(workspace && workspace_bytes >= stm_mask_iou_workspace_bytes(n1, n2, hw), STM_EWORKSPACE,
                "stm_mask_iou_f32: workspace too small");
    STM_REQUIRE(n1 <= 65535 && n2 <= 65535, STM_EINVAL, "stm_mask_iou_f32: too many masks");

    const int words = (hw + 63) / 64;
    unsigned long long* b1 = reinterpret_cast<unsigned long long*>(workspace);
    unsigned long long* b2 = b1 + (size_t)n1 * words;
    hipLaunchKernelGGL(mask_pack_kernel, dim3(stm_cdiv(words, 4), n1), dim3(256), 0, stm_hs(stream), m1, b1, hw, words, thr);
    hipLaunchKernelGGL(mask_pack_kernel, dim3(stm_cdiv(words, 4), n2), dim3(256), 0, stm_hs(stream), m2, b2, hw, words, thr);
    STM_CHECK_LAUNCH("mask_pack_kernel");
    const int rows_wg = std::min(MIOU_R, MIOU_DYN_LDS / (words * 8));
    STM_REQUIRE(rows_wg >= 1, STM_EUNSUPPORTED, "stm_mask_iou_f32: a mask row of %d words does not fit the 46 KB of LDS the kernel stages rows in", words);
    if (n1 < STM_ENV_INT("STM_MIOU_BIG_ROWS", MIOU_BIG_ROWS))
        hipLaunchKernelGGL(mask_iou_pairs_small_kernel, dim3(stm_cdiv(n2, 256), n1), dim3(256), (size_t)words * 8, stm_hs(stream), b1, b2, n2, words, out, group1, group2);
    else
        hipLaunchKernelGGL(mask_iou_pairs_kernel, dim3(stm_cdiv(n1, rows_wg)), dim3(256), (size_t)rows_wg * words * 8, stm_hs(stream), b1,
                           b2, n1, n2, words, out, group1, group2, rows_wg);
    STM_CHECK_LAUNCH("mask_iou_pairs_kernel");
    return STM_OK;
}

// mask IoU of masks that are already bit-packed (stm_lincomb_sigmoid_crop_bits_f32): the pairs kernel alone
extern "C" int stm_mask_iou_bits_f32(const uint64_t* bits1, int n1, const uint64_t* bits2, int n2, int hw, float* out, const int* group1,
                                     const int* group2, stm_stream_t stream)
{
    STM_REQUIRE((group1 == nullptr) == (group2 == nullptr), STM_EINVAL, "stm_mask_iou_bits_f32: give both group arrays or neither");
    STM_REQUIRE(n1 >= 0 && n2 >= 0 && hw > 0, STM_EINVAL, "stm_mask_iou_bits_f32: bad sizes");
    if (n1 == 0 || n2 == 0) return STM_OK;
    STM_REQUIRE(bits1 && bits2 && out, STM_ENULL, "stm_mask_iou_bits_f32: bits1/bits2/out must be non-NULL");
    STM_REQUIRE(n1 <= 65535 && n2 <= 65535, STM_EINVAL, "stm_mask_iou_bits_f32: too many masks");

    const int words = (hw + 63) / 64;
    const int rows_wg = std::min(MIOU_R, MIOU_DYN_LDS / (words * 8));
    STM_REQUIRE(rows_wg >= 1, STM_EUNSUPPORTED, "stm_mask_iou_bits_f32: a mask row of %d words does not fit the 46 KB of LDS the kernel stages rows in", words);
    if (n1 < STM_ENV_INT("STM_MIOU_BIG_ROWS", MIOU_BIG_ROWS))
        hipLaunchKernelGGL(mask_iou_pairs_small_kernel, dim3(stm_cdiv(n2, 256), n1), dim3(256), (size_t)words * 8, stm_hs(stream),
                           reinterpret_cast<const unsigned long long*>(bits1), reinterpret_cast<const unsigned long long*>(bits2), n2, words, out, group1, group2);
    else
        hipLaunchKernelGGL(mask_iou_pairs_kernel, dim3(stm_cdiv(n1, rows_wg)), dim3(256), (size_t)rows_wg * words * 8, stm_hs(stream),
                           reinterpret_cast<const unsigned long long*>(bits1), reinterpret_cast<const unsigned long long*>(bits2), n1, n2, words, out, group1,
                           group2, rows_wg);
    STM_CHECK_LAUNCH("mask_iou_pairs_kernel");
    return STM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Fused conv epilogue for the dense trunk: y = act(y + bias[c] (+ residual)), in place.
// The reference runs conv -> BatchNorm -> ReLU (-> += residual -> ReLU) as separate full passes over the activations
// (backbone.py:38-58); with BN folded into the conv weights (stmask_amd/fuse.py) what is left is one bias add, an
// optional residual add and a ReLU -- one HBM pass here instead of three to five.
// Layout: channel index = (i / inner) % C; inner = H*W for NCHW, 1 for NHWC (channels_last).
namespace {
__global__ __launch_bounds__(256) void bias_act_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                       const float* __restrict__ res, int64_t n4, int C, int64_t inner,
                                                       int relu)
{
    const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i4 >= n4) return;
    float4 v = reinterpret_cast<float4*>(y)[i4];
    float4 b;
    if (inner == 1) {  // NHWC: 4 consecutive channels
        b = *reinterpret_cast<const float4*>(bias + (int)((i4 * 4) % C));
    } else {           // NCHW: 4 consecutive pixels of one channel (inner % 4 == 0)
        const float s = bias[(int)(((i4 * 4) / inner) % C)];
        b = make_float4(s, s, s, s);
    }
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    if (res) {
        const float4 r = reinterpret_cast<const float4*>(res)[i4];
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    reinterpret_cast<float4*>(y)[i4] = v;
}
}  // namespace

extern "C" int stm_bias_act_f32(float* y, const float* bias, const float* residual, int64_t n, int C, int64_t inner,
                                int relu, stm_stream_t stream)
{
    STM_REQUIRE(n >= 0, STM_EINVAL, "stm_bias_act_f32: n < 0");
    if (n == 0) return STM_OK;
    STM_REQUIRE(y && bias, STM_ENULL, "stm_bias_act_f32: y/bias must be non-NULL");
    STM_REQUIRE(C > 0 && inner > 0 && n % ((int64_t)C * inner) == 0, STM_EINVAL, "stm_bias_act_f32: n not a multiple of C*inner");
    STM_REQUIRE((inner == 1 && C % 4 == 0) || (inner > 1 && inner % 4 == 0), STM_EUNSUPPORTED,
                "stm_bias_act_f32: needs C %% 4 == 0 (NHWC) or H*W %% 4 == 0 (NCHW)");
    STM_REQUIRE(((uintptr_t)y % 16 == 0) && ((uintptr_t)bias % 16 == 0) && (!residual || (uintptr_t)residual % 16 == 0),
                STM_EINVAL, "stm_bias_act_f32: pointers must be 16-byte aligned");
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(bias_act_kernel, dim3(stm_cdiv(n4, 256)), dim3(256), 0, stm_hs(stream), y, bias, residual, n4, C, inner,
                       relu);
    STM_CHECK_LAUNCH("bias_act_kernel");
    return STM_OK;
}


// ---- head output assembly (prediction_head_FC.py:168-195) -----------------------------------------------------------
// The planar head (stmask_amd/planar.py) leaves, per kernel shape k, one fp32 matrix [pixels, 3*64] (conf | centerness+bbox
// | mask groups, zero-padded to 64 channels each) and one [pixels, embed] track matrix, pixel axis = the FPN levels
// concatenated (level l: B images of hw_l pixels).  The reference concatenates the per-level, per-k tensors into
// conf [B, N, n_cls], loc [B, N, 4], mask_coeff [B, N, mask_dim], track [B, N, embed] (L2-normalised) with prior index
// n = off_l + p*K + k, and centerness [B, N, 1] = tanh(.) concatenated along H, i.e. prior index off_l + k*hw_l + p
// (prediction_head_FC.py:189 concatenates on dim 1).  One wave per (image, prior): ~60 torch launches become one.
namespace {

struct HeadArgs {
    const float* small[4];
    const float* trk[4];
    float *conf, *loc, *mask, *track, *cen;
    int B, K, n_levels, n_cls, mask_dim, embed, gpad, small_ld, trk_ld, N;
    int lvl_start[9], lvl_hw[8], lvl_off[8];
    int vec4, xcd;   // track rows as float4 (embed % 4 == 0 <= 256, 16-byte aligned strides); XCD-contiguous block order
};

// 16 lanes per (image, prior) row, four rows per wave: the rows are short (41 + 4 + 32 + 1 + 128 floats), so a whole wave per
// row left most lanes idle in every access (41, 4, 32, 1 of 64); here the track embedding moves as float4 per lane when the
// strides allow, and the L2 norm is a 16-lane reduction.
__global__ __launch_bounds__(256) void head_assemble_kernel(const HeadArgs a)
{
    const int sub = threadIdx.x & 15;
    const int64_t row = (int64_t)stm_xcd_block_or_plain(a.xcd, ((int64_t)a.B * a.N + 15) >> 4) * 16 + (threadIdx.x >> 4);      // (b, n)
    if (row < 0 || row >= (int64_t)a.B * a.N) return;
    const int b = (int)(row / a.N), n = (int)(row - (int64_t)b * a.N);
    int l = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i)
        if (i < a.n_levels && n >= a.lvl_off[i]) l = i;
    const int r = n - a.lvl_off[l];
    const int p = r / a.K, k = r - p * a.K;
    const int hw = a.lvl_hw[l];
    const int64_t src = (int64_t)a.lvl_start[l] + (int64_t)b * hw + p;
    const float* sm = a.small[k] + src * a.small_ld;
    const float* tk = a.trk[k] + src * a.trk_ld;
    const int64_t o = (int64_t)b * a.N + n;
    for (int c = sub; c < a.n_cls; c += 16) a.conf[o * a.n_cls + c] = sm[c];
    if (sub < 4) a.loc[o * 4 + sub] = sm[a.gpad + 1 + sub];
    for (int c = sub; c < a.mask_dim; c += 16) a.mask[o * a.mask_dim + c] = sm[2 * a.gpad + c];
    if (sub == 0) a.cen[(int64_t)b * a.N + a.lvl_off[l] + (int64_t)k * hw + p] = tanhf(sm[a.gpad]);
    // track: x / max(||x||_2, 1e-12) (F.normalize, prediction_head_FC.py:177)
    float ss = 0.0f;
    if (a.vec4) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 v[4];                                                   // embed <= 256: at most four float4 per lane
        const int nv = a.embed >> 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c4 = sub + 16 * q;
            if (c4 < nv) {
                v[q] = *reinterpret_cast<const f4*>(tk + 4 * c4);
                ss = fmaf(v[q].x, v[q].x, ss); ss = fmaf(v[q].y, v[q].y, ss); ss = fmaf(v[q].z, v[q].z, ss); ss = fmaf(v[q].w, v[q].w, ss);
            }
        }
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) ss += __shfl_xor(ss, d);
        const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c4 = sub + 16 * q;
            if (c4 < nv) *reinterpret_cast<f4*>(a.track + o * a.embed + 4 * c4) = v[q] * inv;
        }
    } else {
        for (int c = sub; c < a.embed; c += 16) { const float v = tk[c]; ss = fmaf(v, v, ss); }
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) ss += __shfl_xor(ss, d);
        const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        for (int c = sub; c < a.embed; c += 16) a.track[o * a.embed + c] = tk[c] * inv;
    }
}

}  // namespace

extern "C" int stm_head_assemble_f32(const float* const* small, const float* const* trk, const stm_head_layout* L, float* conf,
                                     float* loc, float* mask, float* track, float* centerness, stm_stream_t stream)
{
    STM_REQUIRE(small && trk && L && conf && loc && mask && track && centerness, STM_ENULL, "stm_head_assemble_f32: NULL argument");
    STM_REQUIRE(L->B > 0 && L->K > 0 && L->K <= 4 && L->n_levels > 0 && L->n_levels <= 8 && L->n_cls > 0 && L->n_cls <= L->group_pad &&
                L->mask_dim > 0 && L->mask_dim <= L->group_pad && L->embed_dim > 0 && L->group_pad >= 5 &&
                L->small_ld >= 3 * L->group_pad && L->trk_ld >= L->embed_dim, STM_EINVAL, "stm_head_assemble_f32: bad layout");
    HeadArgs a;
    for (int k = 0; k < 4; ++k) {
        a.small[k] = k < L->K ? small[k] : nullptr;
        a.trk[k] = k < L->K ? trk[k] : nullptr;
        STM_REQUIRE(k >= L->K || (a.small[k] && a.trk[k]), STM_ENULL, "stm_head_assemble_f32: input %d is NULL", k);
    }
    a.conf = conf; a.loc = loc; a.mask = mask; a.track = track; a.cen = centerness;
    a.B = L->B; a.K = L->K; a.n_levels = L->n_levels; a.n_cls = L->n_cls; a.mask_dim = L->mask_dim; a.embed = L->embed_dim;
    a.gpad = L->group_pad; a.small_ld = L->small_ld; a.trk_ld = L->trk_ld;
    int off = 0;
    for (int l = 0; l < 8; ++l) {
        a.lvl_start[l] = l < L->n_levels ? L->lvl_start[l] : 0;
        a.lvl_hw[l] = l < L->n_levels ? L->lvl_hw[l] : 1;
        a.lvl_off[l] = off;
        if (l < L->n_levels) {
            STM_REQUIRE(L->lvl_hw[l] > 0, STM_EINVAL, "stm_head_assemble_f32: level %d has no pixels", l);
            off += L->lvl_hw[l] * L->K;
        }
    }
    a.lvl_start[8] = 0;
    a.N = off;
    const int64_t rows = (int64_t)a.B * a.N;
    a.vec4 = a.embed % 4 == 0 && a.embed <= 256 && a.trk_ld % 4 == 0 && ((uintptr_t)track % 16) == 0;
    for (int k = 0; k < a.K; ++k) a.vec4 = a.vec4 && ((uintptr_t)a.trk[k] % 16) == 0;
    a.xcd = 1;
    hipLaunchKernelGGL(head_assemble_kernel, dim3(stm_xcd_grid(stm_cdiv(rows, 16))), dim3(256), 0, stm_hs(stream), a);
    STM_CHECK_LAUNCH("head_assemble_kernel");
    return STM_OK;
}
