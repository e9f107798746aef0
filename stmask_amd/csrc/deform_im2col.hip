// deform_im2col.hip -- modulated deformable im2col for gfx950 (MI355X).
//
// Replaces the im2col half of dcn_v2.DCN (backbone.py:21-26,45) and mmcv.ops.DeformConv2d
// (Featurealign.py:27-31,72).  HBM-bound: per image it reads x (C*H*W), offsets+mask (3K*Ho*Wo) and writes
// the column buffer (C*K*Ho*Wo floats) -- the write dominates.
//
// Two variants behind one entry point:
//   1  direct   : one thread per (position, tap), loops channels, 4 global gathers per output.  Simple,
//                 used as the cross-check and for shapes the tiled kernel does not like.
//   2  LDS tiled: a workgroup owns TH full output rows x CCH channels.  The input rows those outputs can
//                 touch (+/- HALO rows for the learned offsets, +1 zero column each side) are staged once
//                 into LDS, four channels interleaved per pixel, so one ds_read_b128 per bilinear corner
//                 serves four channels and the zero padding removes every border test from the inner loop.
//                 Each work item is 4 consecutive output positions of one tap: offsets / mask are read as
//                 coalesced 16-byte loads, results leave as coalesced 16-byte stores (one per channel).
//                 Samples whose corners leave the staged rows (large offsets) fall back to global gathers,
//                 so the result is exact for ANY offset.
#include "stm_common.h"

#include <atomic>

namespace {

struct ImcolArgs {
    const float* x;
    const float* off;
    const float* mask;
    float* cols;
    int64_t off_bs, mask_bs;
    int mask_logit;
    int B, C, H, W, kh, kw, sh, sw, ph, pw, dh, dw, dg, Ho, Wo;
    // tiled variant only
    int th, cch, R, LW, halo, tiles_y;
};

__device__ __forceinline__ float sigmoidf_dev(float v) { return 1.0f / (1.0f + expf(-v)); }

// The one bilinear blend every variant uses: an explicit 1-mul + 3-fma chain (the file is built with
// -ffp-contract=off and -fno-slp-vectorize: left to itself hipcc turned the mul/add form into v_pk_mul / v_pk_add pairs
// plus 48 v_mov shuffles per 16 outputs).  Identical in all variants, so they agree bit for bit.
__device__ __forceinline__ float bilerp(float w1, float w2, float w3, float w4, float v1, float v2, float v3, float v4)
{
    return fmaf(w4, v4, fmaf(w3, v3, fmaf(w2, v2, w1 * v1)));
}

// Bilinear sample with the DCNv2 border rule straight from global memory (fallback / direct variant).
__device__ __forceinline__ float sample_global(const float* __restrict__ im, int H, int W, float fy, float fx)
{
    if (!(fy > -1.0f && fx > -1.0f && fy < (float)H && fx < (float)W)) return 0.0f;
    float fl_y = floorf(fy), fl_x = floorf(fx);
    int h_low = (int)fl_y, w_low = (int)fl_x;
    int h_high = h_low + 1, w_high = w_low + 1;
    float lh = fy - fl_y, lw = fx - fl_x, hh = 1.0f - lh, hw = 1.0f - lw;
    float v1 = (h_low >= 0 && w_low >= 0) ? im[h_low * W + w_low] : 0.0f;
    float v2 = (h_low >= 0 && w_high <= W - 1) ? im[h_low * W + w_high] : 0.0f;
    float v3 = (h_high <= H - 1 && w_low >= 0) ? im[h_high * W + w_low] : 0.0f;
    float v4 = (h_high <= H - 1 && w_high <= W - 1) ? im[h_high * W + w_high] : 0.0f;
    return bilerp(hh * hw, hh * lw, lh * hw, lh * lw, v1, v2, v3, v4);
}

// ------------------------------------------------------------------------------------ variant 1
// grid: x = ceil(HWo/256), y = dg * chunks_per_group * K, z = B
__global__ __launch_bounds__(256) void deform_im2col_direct(ImcolArgs a, int cpb, int chunks_per_group)
{
    const int K = a.kh * a.kw;
    const int HWo = a.Ho * a.Wo;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= HWo) return;
    int by = blockIdx.y;
    const int k = by % K;
    by /= K;
    const int chunk = by % chunks_per_group;
    const int g = by / chunks_per_group;
    const int b = blockIdx.z;
    const int Cg = a.C / a.dg;
    const int c0 = g * Cg + chunk * cpb;
    const int c1 = min(g * Cg + Cg, c0 + cpb);

    const int ho = n / a.Wo, wo = n - ho * a.Wo;
    const int i = k / a.kw, j = k - i * a.kw;
    const float* ob = a.off + (int64_t)b * a.off_bs + (int64_t)g * 2 * K * HWo;
    const float dy = ob[(int64_t)(2 * k) * HWo + n];
    const float dx = ob[(int64_t)(2 * k + 1) * HWo + n];
    float m = 1.0f;
    if (a.mask) {
        m = a.mask[(int64_t)b * a.mask_bs + (int64_t)(g * K + k) * HWo + n];
        if (a.mask_logit) m = sigmoidf_dev(m);
    }
    const float fy = (float)(ho * a.sh - a.ph + i * a.dh) + dy;
    const float fx = (float)(wo * a.sw - a.pw + j * a.dw) + dx;

    // corner weights (mask folded in) and clamped addresses, computed once for all channels
    float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f;
    int a1 = 0, a2 = 0, a3 = 0, a4 = 0;
    if (fy > -1.0f && fx > -1.0f && fy < (float)a.H && fx < (float)a.W) {
        float fl_y = floorf(fy), fl_x = floorf(fx);
        int h_low = (int)fl_y, w_low = (int)fl_x, h_high = h_low + 1, w_high = w_low + 1;
        float lh = fy - fl_y, lw = fx - fl_x, hh = 1.0f - lh, hw = 1.0f - lw;
        bool t = h_low >= 0, l = w_low >= 0, bt = h_high <= a.H - 1, r = w_high <= a.W - 1;
        int hl = max(h_low, 0), wl = max(w_low, 0), hh_i = min(h_high, a.H - 1), wh_i = min(w_high, a.W - 1);
        w1 = (t && l) ? hh * hw * m : 0.f;
        w2 = (t && r) ? hh * lw * m : 0.f;
        w3 = (bt && l) ? lh * hw * m : 0.f;
        w4 = (bt && r) ? lh * lw * m : 0.f;
        a1 = hl * a.W + wl;
        a2 = hl * a.W + wh_i;
        a3 = hh_i * a.W + wl;
        a4 = hh_i * a.W + wh_i;
    }
    const int64_t HW = (int64_t)a.H * a.W;
    const float* xb = a.x + ((int64_t)b * a.C + c0) * HW;
    float* cb = a.cols + (((int64_t)b * a.C + c0) * K + k) * HWo + n;
#pragma unroll 4
    for (int c = c0; c < c1; ++c) {
        float v = bilerp(w1, w2, w3, w4, xb[a1], xb[a2], xb[a3], xb[a4]);
        *cb = v;
        xb += HW;
        cb += (int64_t)K * HWo;
    }
}

// rare path of the tiled kernel: 4 channels of one far-away sample, same expression order as the direct kernel
// (mask folded into the corner weights) so both variants agree bit for bit; kept out of line so it costs no
// registers in the main loop
__device__ __noinline__ float4 sample_global4(const float* __restrict__ xc, int64_t HW, int H, int W, float fy, float fx,
                                              float m)
{
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!(fy > -1.0f && fx > -1.0f && fy < (float)H && fx < (float)W)) return r;
    float fl_y = floorf(fy), fl_x = floorf(fx);
    int h_low = (int)fl_y, w_low = (int)fl_x, h_high = h_low + 1, w_high = w_low + 1;
    float lh = fy - fl_y, lw = fx - fl_x, hh = 1.0f - lh, hw = 1.0f - lw;
    bool t = h_low >= 0, l = w_low >= 0, bt = h_high <= H - 1, rt = w_high <= W - 1;
    int hl = max(h_low, 0), wl = max(w_low, 0), hh_i = min(h_high, H - 1), wh_i = min(w_high, W - 1);
    float w1 = (t && l) ? hh * hw * m : 0.f;
    float w2 = (t && rt) ? hh * lw * m : 0.f;
    float w3 = (bt && l) ? lh * hw * m : 0.f;
    float w4 = (bt && rt) ? lh * lw * m : 0.f;
    int a1 = hl * W + wl, a2 = hl * W + wh_i, a3 = hh_i * W + wl, a4 = hh_i * W + wh_i;
    r.x = bilerp(w1, w2, w3, w4, xc[a1], xc[a2], xc[a3], xc[a4]);
    xc += HW;
    r.y = bilerp(w1, w2, w3, w4, xc[a1], xc[a2], xc[a3], xc[a4]);
    xc += HW;
    r.z = bilerp(w1, w2, w3, w4, xc[a1], xc[a2], xc[a3], xc[a4]);
    xc += HW;
    r.w = bilerp(w1, w2, w3, w4, xc[a1], xc[a2], xc[a3], xc[a4]);
    return r;
}

__device__ __forceinline__ void unpack(const float4 v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
__device__ __forceinline__ void unpack(const float v, float (&o)[1]) { o[0] = v; }
// LDS column swizzle.  A work item is 4 CONSECUTIVE output positions, so the 16 lanes of a ds_read_b128 group read
// pixels 4*s apart (64-128 B apart): un-swizzled they fall on 4 (stride 1) or 2 (stride 2) of the 16 16-byte bank
// groups -> 4- / 8-way conflicts.  XOR-ing column bits [5:4] into bits [1:0] spreads columns {0,4,8,...,60} over all
// 16 groups (2-way left at stride 2).  Bijective inside every aligned 4-column block, so LW only has to be a multiple of 4.
__device__ __forceinline__ int swz(int col) { return col ^ ((col >> 4) & 3); }
template <int NP> struct PosVec { using type = float4; };
template <> struct PosVec<1> { using type = float; };

// ------------------------------------------------------------------------------------ variant 2
// grid: x = tiles_y * (C / cch), y = B.  Dynamic LDS: (cch/4) * R * LW float4.
template <int NP, int NTHR, int MINW = 1>
__global__ __launch_bounds__(NTHR, MINW) void deform_im2col_lds(ImcolArgs a)
{
    using VecT = typename PosVec<NP>::type;
    extern __shared__ float4 tile[];
    const int K = a.kh * a.kw;
    const int HWo = a.Ho * a.Wo;
    const int64_t HW = (int64_t)a.H * a.W;
    const int tid = threadIdx.x;
    const int ty = blockIdx.x % a.tiles_y;
    const int chunk = blockIdx.x / a.tiles_y;
    const int b = blockIdx.y;
    const int c0 = chunk * a.cch;
    const int g = c0 / (a.C / a.dg);
    const int nq = a.cch >> 2;
    const int ho0 = ty * a.th;
    const int rows_out = min(a.th, a.Ho - ho0);
    const int y0 = ho0 * a.sh - a.ph - a.halo;  // input row held by LDS row 0
    const int RL = a.R * a.LW;

    const int n0 = ho0 * a.Wo;
    const int NT = rows_out * a.Wo;
    const int items_per_k = (NT + NP - 1) / NP;
    const int n_items = K * items_per_k;
    const float* ob = a.off + (int64_t)b * a.off_bs + (int64_t)g * 2 * K * HWo;
    const float* mb = a.mask ? a.mask + (int64_t)b * a.mask_bs + (int64_t)g * K * HWo : nullptr;

    // ---- 1. issue the offset / mask loads of this thread's first PRE items NOW: their HBM/L2 latency is hidden
    //         behind the staging phase below (they do not depend on LDS)
    constexpr int PRE = 3;
    VecT pdy[PRE], pdx[PRE], pm[PRE];
#pragma unroll
    for (int it = 0; it < PRE; ++it) {
        const int item = tid + it * NTHR;
        if (item < n_items) {
            const int k = item / items_per_k;
            const int nb = n0 + (item - k * items_per_k) * NP;
            pdy[it] = *reinterpret_cast<const VecT*>(ob + (int64_t)(2 * k) * HWo + nb);
            pdx[it] = *reinterpret_cast<const VecT*>(ob + (int64_t)(2 * k + 1) * HWo + nb);
            if (mb) pm[it] = *reinterpret_cast<const VecT*>(mb + (int64_t)k * HWo + nb);
        }
    }

    // ---- 2. stage: thread <-> pixel, 4 channel loads (each coalesced across the wave) -> one ds_write_b128.
    //         Loads are issued in batches of SU pixels per thread before any LDS write, so SU*4 loads are in
    //         flight per lane instead of 4 (the unbatched loop waited vmcnt(0) once per pixel).
    {
        constexpr int SU = 4;
        const float* xb = a.x + ((int64_t)b * a.C + c0) * HW;
        const int total = nq * RL;
        // (q, r, col) advance incrementally with idx += 256: no per-element integer division
        const int step_r = NTHR / a.LW, step_c = NTHR - step_r * a.LW;
        int r = tid / a.LW, col = tid - r * a.LW, q = 0;
        while (r >= a.R) { r -= a.R; ++q; }
        for (int base = tid; base < total; base += NTHR * SU) {
            float4 v[SU];
            int dst[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                dst[u] = q * RL + r * a.LW + swz(col);
                const int yy = y0 + r, xx = col - 1;
                if (base + u * NTHR < total && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
                    const float* p = xb + (int64_t)(4 * q) * HW + (int64_t)yy * a.W + xx;
                    v[u].x = p[0];
                    v[u].y = p[HW];
                    v[u].z = p[2 * HW];
                    v[u].w = p[3 * HW];
                }
                col += step_c;
                r += step_r;
                if (col >= a.LW) { col -= a.LW; ++r; }
                while (r >= a.R) { r -= a.R; ++q; }
            }
#pragma unroll
            for (int u = 0; u < SU; ++u)
                if (base + u * NTHR < total) tile[dst[u]] = v[u];
        }
    }
    __syncthreads();

    // ---- 3. items: (4 consecutive positions) x (one tap), all channel quads of the block
    auto process = [&](const int item, const VecT dyq, const VecT dxq, const VecT mq) {
        const int k = item / items_per_k;
        const int pq = item - k * items_per_k;
        const int i = k / a.kw, j = k - i * a.kw;
        const int nb = n0 + pq * NP;
        float dyv[NP], dxv[NP], mv[NP];
        unpack(dyq, dyv);
        unpack(dxq, dxv);
        unpack(mq, mv);

        // per-position coefficients
        float w1[NP], w2[NP], w3[NP], w4[NP], fyv[NP], fxv[NP];
        int la[NP], lb[NP];    // LDS index of the (h_low, w_low) / (h_low, w_high) corners (swizzled columns)
        unsigned farmask = 0;  // positions whose corners leave the staged rows
        int ho = nb / a.Wo, wo = nb - ho * a.Wo;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float m = 1.0f;
            if (mb) m = a.mask_logit ? sigmoidf_dev(mv[p]) : mv[p];
            float fy = (float)(ho * a.sh - a.ph + i * a.dh) + dyv[p];
            float fx = (float)(wo * a.sw - a.pw + j * a.dw) + dxv[p];
            fyv[p] = fy;
            fxv[p] = fx;
            bool valid = fy > -1.0f && fx > -1.0f && fy < (float)a.H && fx < (float)a.W;
            float fl_y = floorf(fy), fl_x = floorf(fx);
            int h_low = (int)fl_y, w_low = (int)fl_x;
            float lh = fy - fl_y, lw = fx - fl_x, hh = 1.0f - lh, hw = 1.0f - lw;
            int r = h_low - y0;
            bool in_rows = (r >= 0) && (r + 1 < a.R);
            bool use = valid && in_rows;
            w1[p] = use ? hh * hw * m : 0.f;
            w2[p] = use ? hh * lw * m : 0.f;
            w3[p] = use ? lh * hw * m : 0.f;
            w4[p] = use ? lh * lw * m : 0.f;
            la[p] = use ? (r * a.LW + swz(w_low + 1)) : 0;
            lb[p] = use ? (r * a.LW + swz(w_low + 2)) : 0;
            if (valid && !in_rows) farmask |= (1u << p);
            mv[p] = m;
            if (++wo == a.Wo) { wo = 0; ++ho; }
        }

        // cols[b][c*K + k][HWo].  (A tile-blocked column buffer -- one contiguous region per workgroup -- was built and
        // measured: no gain, and the GEMM's B loader pays for it; removed.)
        const int64_t cs = (int64_t)K * HWo;
        float* cb = a.cols + (((int64_t)b * a.C + c0) * K + k) * HWo + nb;
        for (int q = 0; q < nq; ++q) {
            const float4* tq = tile + q * RL;
            float4 acc[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const float4 v1 = tq[la[p]];
                const float4 v2 = tq[lb[p]];
                const float4 v3 = tq[la[p] + a.LW];
                const float4 v4 = tq[lb[p] + a.LW];
                acc[p].x = bilerp(w1[p], w2[p], w3[p], w4[p], v1.x, v2.x, v3.x, v4.x);
                acc[p].y = bilerp(w1[p], w2[p], w3[p], w4[p], v1.y, v2.y, v3.y, v4.y);
                acc[p].z = bilerp(w1[p], w2[p], w3[p], w4[p], v1.z, v2.z, v3.z, v4.z);
                acc[p].w = bilerp(w1[p], w2[p], w3[p], w4[p], v1.w, v2.w, v3.w, v4.w);
            }
            if (farmask) {  // rare: offsets larger than the halo -> exact global gather
                const float* xc = a.x + ((int64_t)b * a.C + c0 + 4 * q) * HW;
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    if (farmask & (1u << p)) acc[p] = sample_global4(xc, HW, a.H, a.W, fyv[p], fxv[p], mv[p]);
            }
            float* c_ = cb + (int64_t)(4 * q) * cs;
            if constexpr (NP == 4) {
                typedef float f4nt __attribute__((ext_vector_type(4)));
                const f4nt o0 = {acc[0].x, acc[1].x, acc[2].x, acc[3].x}, o1 = {acc[0].y, acc[1].y, acc[2].y, acc[3].y};
                const f4nt o2 = {acc[0].z, acc[1].z, acc[2].z, acc[3].z}, o3 = {acc[0].w, acc[1].w, acc[2].w, acc[3].w};
                *reinterpret_cast<f4nt*>(c_) = o0;     // (nontemporal stores measured: no difference, 172 us either way at batch 8)
                *reinterpret_cast<f4nt*>(c_ + cs) = o1;
                *reinterpret_cast<f4nt*>(c_ + 2 * cs) = o2;
                *reinterpret_cast<f4nt*>(c_ + 3 * cs) = o3;
            } else {
                c_[0] = acc[0].x;
                c_[cs] = acc[0].y;
                c_[2 * cs] = acc[0].z;
                c_[3 * cs] = acc[0].w;
            }
        }
    };

#pragma unroll
    for (int it = 0; it < PRE; ++it) {
        const int item = tid + it * NTHR;
        if (item < n_items) process(item, pdy[it], pdx[it], pm[it]);
    }
    for (int item = tid + PRE * NTHR; item < n_items; item += NTHR) {  // tiles with more than PRE*256 items
        const int k = item / items_per_k;
        const int nb = n0 + (item - k * items_per_k) * NP;
        VecT dq = *reinterpret_cast<const VecT*>(ob + (int64_t)(2 * k) * HWo + nb);
        VecT xq = *reinterpret_cast<const VecT*>(ob + (int64_t)(2 * k + 1) * HWo + nb);
        VecT mq = dq;
        if (mb) mq = *reinterpret_cast<const VecT*>(mb + (int64_t)k * HWo + nb);
        process(item, dq, xq, mq);
    }
}

// ------------------------------------------------------------------------------------ variant 3
// Same tile / item geometry as variant 2 (NP = 4), but the channel loop is INSIDE the workgroup: the bilinear
// coefficients of a thread's (<= IT) items are computed once and stay in registers while the workgroup walks the
// channel quads of its chunk, re-staging ONE quad buffer in LDS per step.  PMC counters of variant 2 showed ~44 VALU
// instructions per output element (coefficient set-up and 64-bit staging addresses re-done for every 4 channels) and
// waves issuing 51 % of their lifetime: instruction-bound, not HBM-bound.  Here the set-up is amortised over cch
// (16-64) channels and staging uses 32-bit offsets from a wave-uniform per-quad base.
// A register-prefetch pipeline over the quads (loads of quad q+1 issued before the FMAs of quad q) was measured and
// REJECTED: the prefetch registers pushed the kernel to 256 VGPR + 172 AGPR (1 wave/SIMD) and it ran 1.4x slower --
// on this kernel occupancy hides latency better than software pipelining.
// grid: 1-D, tiles_y * (C / cch) * B, XCD-aware id -> tile map.  Dynamic LDS: R * LW float4 (one quad).
template <int IT>
__global__ __launch_bounds__(256) void deform_im2col_lds3(ImcolArgs a)
{
    extern __shared__ float4 tile[];
    const int K = a.kh * a.kw;
    const int HWo = a.Ho * a.Wo;
    const int HW = a.H * a.W;
    const int tid = threadIdx.x;
    // XCD-aware block -> tile map (1-D grid).  Workgroups are dealt round-robin over the 8 XCDs, each with its own L2.
    // All row tiles of one (image, channel chunk) share halo rows and the offsets, so they are given to ONE XCD:
    // ids congruent mod 8 walk the row tiles of a plane before moving to the next plane (speed only, any placement
    // is correct).  Planes = B * C/cch; the tail (planes % 8) falls back to the plain order.
    int ty, plane;
    {
        const int planes = a.B * (a.C / a.cch);
        const int id = blockIdx.x;
        const int full = (planes / 8) * 8 * a.tiles_y;   // ids covered by whole groups of 8 planes
        if (id < full) {
            const int xcd = id & 7, j = id >> 3;
            ty = j % a.tiles_y;
            plane = xcd + 8 * (j / a.tiles_y);
        } else {
            const int j = id - full;
            ty = j % a.tiles_y;
            plane = (planes / 8) * 8 + j / a.tiles_y;
        }
    }
    const int chunk = plane % (a.C / a.cch);
    const int b = plane / (a.C / a.cch);
    const int c0 = chunk * a.cch;
    const int g = c0 / (a.C / a.dg);
    const int nq = a.cch >> 2;
    const int ho0 = ty * a.th;
    const int rows_out = min(a.th, a.Ho - ho0);
    const int y0 = ho0 * a.sh - a.ph - a.halo;
    const int RL = a.R * a.LW;
    const int n0 = ho0 * a.Wo;
    const int NT = rows_out * a.Wo;
    const int items_per_k = NT >> 2;
    const int n_items = K * items_per_k;
    const float* ob = a.off + (int64_t)b * a.off_bs + (int64_t)g * 2 * K * HWo;
    const float* mb = a.mask ? a.mask + (int64_t)b * a.mask_bs + (int64_t)g * K * HWo : nullptr;

    // ---- 1. per-item coefficients, once per workgroup
    float4 wq[IT][4];        // corner weights (w1..w4) of the 4 positions, mask folded in
    int la[IT][4], lb[IT][4];
    int sbase[IT];           // k * HWo + first position: store offset inside one channel's K*HWo block
    unsigned far[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int item = tid + it * 256;
        far[it] = 0;
        sbase[it] = -1;
#pragma unroll
        for (int p = 0; p < 4; ++p) { wq[it][p] = make_float4(0.f, 0.f, 0.f, 0.f); la[it][p] = 0; lb[it][p] = 0; }
        if (item < n_items) {
            const int k = item / items_per_k;
            const int nb = n0 + ((item - k * items_per_k) << 2);
            const int i = k / a.kw, j = k - i * a.kw;
            sbase[it] = k * HWo + nb;
            float dyv[4], dxv[4], mv[4] = {1.f, 1.f, 1.f, 1.f};
            unpack(*reinterpret_cast<const float4*>(ob + (int64_t)(2 * k) * HWo + nb), dyv);
            unpack(*reinterpret_cast<const float4*>(ob + (int64_t)(2 * k + 1) * HWo + nb), dxv);
            if (mb) unpack(*reinterpret_cast<const float4*>(mb + (int64_t)k * HWo + nb), mv);
            int ho = nb / a.Wo, wo = nb - ho * a.Wo;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float m = mv[p];
                if (mb && a.mask_logit) m = sigmoidf_dev(m);
                float fy = (float)(ho * a.sh - a.ph + i * a.dh) + dyv[p];
                float fx = (float)(wo * a.sw - a.pw + j * a.dw) + dxv[p];
                bool valid = fy > -1.0f && fx > -1.0f && fy < (float)a.H && fx < (float)a.W;
                float fl_y = floorf(fy), fl_x = floorf(fx);
                int h_low = (int)fl_y, w_low = (int)fl_x;
                float lh = fy - fl_y, lw = fx - fl_x, hh = 1.0f - lh, hw = 1.0f - lw;
                int r = h_low - y0;
                bool in_rows = (r >= 0) && (r + 1 < a.R);
                bool use = valid && in_rows;
                wq[it][p].x = use ? hh * hw * m : 0.f;
                wq[it][p].y = use ? hh * lw * m : 0.f;
                wq[it][p].z = use ? lh * hw * m : 0.f;
                wq[it][p].w = use ? lh * lw * m : 0.f;
                la[it][p] = use ? (r * a.LW + swz(w_low + 1)) : 0;
                lb[it][p] = use ? (r * a.LW + swz(w_low + 2)) : 0;
                if (valid && !in_rows) far[it] |= (1u << p);
                if (++wo == a.Wo) { wo = 0; ++ho; }
            }
        }
    }

    // ---- 2. walk the channel quads of this chunk through one LDS buffer
    const int step_r = 256 / a.LW, step_c = 256 - step_r * a.LW;
    const int r_first = tid / a.LW, c_first = tid - r_first * a.LW;
    const int64_t cs = (int64_t)K * HWo;
    float* const cbase = a.cols + ((int64_t)b * a.C + c0) * cs;
    for (int q = 0; q < nq; ++q) {
        const float* xq = a.x + ((int64_t)b * a.C + c0 + 4 * q) * HW;   // wave-uniform base of this quad
        if (q) __syncthreads();                                          // previous quad fully consumed
        {
            constexpr int SU = 4;
            int r = r_first, col = c_first;
            for (int base = tid; base < RL; base += 256 * SU) {
                float4 v[SU];
                int dst[SU];
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    dst[u] = r * a.LW + swz(col);
                    const int yy = y0 + r, xx = col - 1;
                    if (base + u * 256 < RL && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
                        const int o = yy * a.W + xx;
                        v[u].x = xq[o];
                        v[u].y = xq[o + HW];
                        v[u].z = xq[o + 2 * HW];
                        v[u].w = xq[o + 3 * HW];
                    }
                    col += step_c;
                    r += step_r;
                    if (col >= a.LW) { col -= a.LW; ++r; }
                }
#pragma unroll
                for (int u = 0; u < SU; ++u)
                    if (base + u * 256 < RL) tile[dst[u]] = v[u];
            }
        }
        __syncthreads();
        float* cq = cbase + (int64_t)(4 * q) * cs;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            if (sbase[it] < 0) continue;
            float4 acc[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float4 v1 = tile[la[it][p]], v2 = tile[lb[it][p]], v3 = tile[la[it][p] + a.LW], v4 = tile[lb[it][p] + a.LW];
                const float4 w = wq[it][p];
                acc[p].x = bilerp(w.x, w.y, w.z, w.w, v1.x, v2.x, v3.x, v4.x);
                acc[p].y = bilerp(w.x, w.y, w.z, w.w, v1.y, v2.y, v3.y, v4.y);
                acc[p].z = bilerp(w.x, w.y, w.z, w.w, v1.z, v2.z, v3.z, v4.z);
                acc[p].w = bilerp(w.x, w.y, w.z, w.w, v1.w, v2.w, v3.w, v4.w);
            }
            if (far[it]) {  // rare: offsets larger than the halo -> exact global gather (coordinates re-derived)
                const int k = sbase[it] / HWo;
                const int nb = sbase[it] - k * HWo;
                const int i = k / a.kw, j = k - i * a.kw;
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if (far[it] & (1u << p)) {
                        const int n = nb + p;
                        const int ho = n / a.Wo, wo = n - ho * a.Wo;
                        float m = 1.0f;
                        if (mb) {
                            m = mb[(int64_t)k * HWo + n];
                            if (a.mask_logit) m = sigmoidf_dev(m);
                        }
                        const float fy = (float)(ho * a.sh - a.ph + i * a.dh) + ob[(int64_t)(2 * k) * HWo + n];
                        const float fx = (float)(wo * a.sw - a.pw + j * a.dw) + ob[(int64_t)(2 * k + 1) * HWo + n];
                        acc[p] = sample_global4(xq, HW, a.H, a.W, fy, fx, m);
                    }
            }
            float* c_ = cq + sbase[it];
            typedef float f4nt __attribute__((ext_vector_type(4)));
            const f4nt o0 = {acc[0].x, acc[1].x, acc[2].x, acc[3].x}, o1 = {acc[0].y, acc[1].y, acc[2].y, acc[3].y};
            const f4nt o2 = {acc[0].z, acc[1].z, acc[2].z, acc[3].z}, o3 = {acc[0].w, acc[1].w, acc[2].w, acc[3].w};
            *reinterpret_cast<f4nt*>(c_) = o0;
            *reinterpret_cast<f4nt*>(c_ + cs) = o1;
            *reinterpret_cast<f4nt*>(c_ + 2 * cs) = o2;
            *reinterpret_cast<f4nt*>(c_ + 3 * cs) = o3;
        }
    }
}


}  // namespace

static int validate_geom(const stm_deform_geom* g, const char* who)
{
    STM_REQUIRE(g, STM_ENULL, "%s: geometry is NULL", who);
    STM_REQUIRE(g->B > 0 && g->C > 0 && g->H > 0 && g->W > 0, STM_EINVAL, "%s: empty input %dx%dx%dx%d", who, g->B,
                g->C, g->H, g->W);
    STM_REQUIRE(g->kh > 0 && g->kw > 0 && g->sh > 0 && g->sw > 0 && g->dh > 0 && g->dw > 0 && g->ph >= 0 &&
                    g->pw >= 0,
                STM_EINVAL, "%s: bad kernel/stride/pad/dilation", who);
    STM_REQUIRE(g->dg > 0 && g->C % g->dg == 0, STM_EINVAL, "%s: C=%d not divisible by deform groups %d", who, g->C,
                g->dg);
    int Ho = (g->H + 2 * g->ph - (g->dh * (g->kh - 1) + 1)) / g->sh + 1;
    int Wo = (g->W + 2 * g->pw - (g->dw * (g->kw - 1) + 1)) / g->sw + 1;
    STM_REQUIRE(Ho == g->Ho && Wo == g->Wo, STM_EINVAL, "%s: output size %dx%d does not match conv arithmetic %dx%d",
                who, g->Ho, g->Wo, Ho, Wo);
    return STM_OK;
}

extern "C" int stm_deform_im2col_f32(const float* x, const float* offset, int64_t off_bstride, const float* mask,
                                     int64_t mask_bstride, int mask_is_logit, float* cols, const stm_deform_geom* g,
                                     int variant, stm_stream_t stream)
{
    int rc = validate_geom(g, "stm_deform_im2col_f32");
    if (rc) return rc;
    STM_REQUIRE(x && offset && cols, STM_ENULL, "stm_deform_im2col_f32: x/offset/cols must be non-NULL");
    const int K = g->kh * g->kw, HWo = g->Ho * g->Wo, Cg = g->C / g->dg;
    STM_REQUIRE(off_bstride >= (int64_t)g->dg * 2 * K * HWo, STM_EINVAL,
                "stm_deform_im2col_f32: offset batch stride %lld < %lld", (long long)off_bstride,
                (long long)g->dg * 2 * K * HWo);
    STM_REQUIRE(!mask || mask_bstride >= (int64_t)g->dg * K * HWo, STM_EINVAL,
                "stm_deform_im2col_f32: mask batch stride too small");
    STM_REQUIRE(variant >= 0 && variant <= 3, STM_EINVAL, "stm_deform_im2col_f32: variant %d not in 0..3", variant);

    ImcolArgs a;
    a.x = x; a.off = offset; a.mask = mask; a.cols = cols;
    a.off_bs = off_bstride; a.mask_bs = mask ? mask_bstride : 0; a.mask_logit = mask_is_logit;
    a.B = g->B; a.C = g->C; a.H = g->H; a.W = g->W; a.kh = g->kh; a.kw = g->kw; a.sh = g->sh; a.sw = g->sw;
    a.ph = g->ph; a.pw = g->pw; a.dh = g->dh; a.dw = g->dw; a.dg = g->dg; a.Ho = g->Ho; a.Wo = g->Wo;
    a.th = a.cch = a.R = a.LW = a.halo = a.tiles_y = 0;

    bool tiled_ok = (Cg % 4 == 0);
    const bool auto_variant = (variant == 0);
    int auto_items = 0;
    if (variant == 0) {
        // measured on MI355X (scripts/exp_im2col.sh, true kernel durations, batch 8): variant 3 (coefficients once per
        // workgroup, 32 channels streamed through LDS, XCD-aware tiles) wins on the large layers -- 2 items per thread
        // for >= 3840 output positions, 1 item per thread for the mid-size stride-2 layer -- variant 2 on the small ones.
        if (!tiled_ok) variant = 1;
        else if (Cg % 32 == 0 && (int64_t)HWo * K >= 3840 * 9) { variant = 3; auto_items = 512; }
        else if (Cg % 32 == 0 && g->sh == 2 && HWo >= 960) { variant = 3; auto_items = 256; }
        else variant = 2;
    }
    if (variant >= 2 && !tiled_ok) variant = 1;

    if (variant == 1) {
        // channels per block: keep >= ~1024 blocks in flight, but amortise the coefficient set-up
        int cpb = Cg;
        int64_t base_blocks = (int64_t)stm_cdiv(HWo, 256) * K * g->dg * g->B;
        while (cpb > 16 && base_blocks * (Cg / cpb) < 2048 && cpb % 2 == 0) cpb /= 2;
        int chunks = stm_cdiv(Cg, cpb);
        dim3 grid(stm_cdiv(HWo, 256), g->dg * chunks * K, g->B);
        hipLaunchKernelGGL(deform_im2col_direct, grid, dim3(256), 0, stm_hs(stream), a, cpb, chunks);
        STM_CHECK_LAUNCH("deform_im2col_direct");
        return STM_OK;
    }

    // ---- tiled: pick rows per tile / channels per block --------------------------------------------
    const int halo = 3;                               // learned offsets of trained DCNs rarely exceed +-3 rows (beyond: exact global gather)
    const int LW = ((g->W + 2 + 3) / 4) * 4;          // +1 zero column each side, rounded up for the column swizzle
    bool vec = (HWo % 4 == 0) && (off_bstride % 4 == 0) && (!mask || mask_bstride % 4 == 0) &&
               ((uintptr_t)offset % 16 == 0) && (!mask || (uintptr_t)mask % 16 == 0) && ((uintptr_t)cols % 16 == 0);
    if (variant == 3 && vec) {
        // ---- variant 3: coefficients once per workgroup, channel quads streamed through one LDS buffer ----------
        // rows per tile: <= 3 items per thread (items = th*Wo/4 * K <= 768); channels per workgroup: as many as keep
        // >= ~3 workgroups per CU in flight (the quad loop amortises the set-up, the grid must still fill 256 CUs)
        int step3 = 1;
        while ((step3 * g->Wo) % 4 != 0) ++step3;
        const int max_items = auto_items ? auto_items : 768;    // 1..3 items per thread
        int th3 = (max_items * 4) / (g->Wo * K);
        th3 = max(step3, (th3 / step3) * step3);
        th3 = min(th3, g->Ho);
        if (th3 < g->Ho) th3 = max(step3, (th3 / step3) * step3);
        const int items3 = (th3 * g->Wo / 4) * K;
        const int R3 = (th3 - 1) * g->sh + (g->kh - 1) * g->dh + 2 + 2 * halo;
        const size_t lds3 = (size_t)R3 * LW * sizeof(float4);
        if ((th3 * g->Wo) % 4 == 0 && items3 <= 768 && lds3 <= 160 * 1024) {
            const int tiles3 = stm_cdiv(g->Ho, th3);
            int cch3 = 4;
            for (int c = 8; c <= Cg && c <= 64; c += 4)
                if (Cg % c == 0 && (int64_t)tiles3 * (g->C / c) * g->B >= 768) cch3 = c;
            if (auto_variant && Cg % 32 == 0) cch3 = 32;      // sweep optimum for the wide layers this path is chosen for
            STM_REQUIRE(cch3 % 4 == 0 && Cg % cch3 == 0, STM_EINVAL, "stm_deform_im2col_f32: channels per block %d invalid",
                        cch3);
            a.th = th3; a.cch = cch3; a.R = R3; a.LW = LW; a.halo = halo; a.tiles_y = tiles3;
            dim3 grid3(tiles3 * (g->C / cch3) * g->B);   // 1-D: the kernel maps ids to tiles XCD-aware
            const int it = stm_cdiv(items3, 256);
            auto launch = [&](auto kern) {
                if (lds3 > 48 * 1024)
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int)lds3);
                hipLaunchKernelGGL(kern, grid3, dim3(256), lds3, stm_hs(stream), a);
            };
            {
                if (it <= 1) launch(deform_im2col_lds3<1>);
                else if (it == 2) launch(deform_im2col_lds3<2>);
                else launch(deform_im2col_lds3<3>);
                STM_CHECK_LAUNCH("deform_im2col_lds3");
                return STM_OK;
            }
        }
    }
    // Tile choice (scripts/bench_kernels.py --env-sweep on MI355X): 8 channels per workgroup and as many output rows
    // as fit the LDS budget (up to 12) minimise halo re-reads; when 8 channels do not leave room for >= 4 rows
    // (wide stride-2 layers) fall back to 4 channels.  With 16-byte stores th*Wo must be a multiple of 4.
    const size_t lds_budget = 64 * 1024;
    auto rows_of = [&](int th_) { return (th_ - 1) * g->sh + (g->kh - 1) * g->dh + 2 + 2 * halo; };
    auto fits = [&](int th_, int cch_) { return (size_t)(cch_ / 4) * rows_of(th_) * LW * sizeof(float4) <= lds_budget; };
    int step = 1;
    if (vec) while ((step * g->Wo) % 4 != 0) ++step;  // step in {1,2,4}
    int cch = (Cg % 8 == 0) ? 8 : 4;
    int th = min(12, g->Ho);
    while (th > 4 && !fits(th, cch)) --th;
    if (!fits(th, cch) && cch == 8) {
        cch = 4;
        th = min(8, g->Ho);
        while (th > 1 && !fits(th, cch)) --th;
    }
    th = min(th, g->Ho);
    if (vec) {
        th = ((th + step - 1) / step) * step;
        if (th > g->Ho) th = g->Ho;  // last tile = remaining rows; HWo % 4 == 0 keeps it aligned
        if ((th * g->Wo) % 4 != 0) vec = false;
    }
    const int R = rows_of(th);
    const size_t quad_bytes = (size_t)R * LW * sizeof(float4);
    STM_REQUIRE(cch % 4 == 0 && Cg % cch == 0, STM_EINVAL, "stm_deform_im2col_f32: channels per block %d invalid", cch);
    size_t lds = (size_t)(cch / 4) * quad_bytes;
    if (lds > 160 * 1024) {  // a single quad of this tile does not fit: use the direct kernel
        return stm_deform_im2col_f32(x, offset, off_bstride, mask, mask_bstride, mask_is_logit, cols, g, 1, stream);
    }
    a.th = th; a.cch = cch; a.R = R; a.LW = LW; a.halo = halo; a.tiles_y = stm_cdiv(g->Ho, th);
    dim3 grid(a.tiles_y * (g->C / cch), g->B);
    auto launch2 = [&](auto kern, int threads) {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(threads), lds, stm_hs(stream), a);
    };
    if (vec) launch2(deform_im2col_lds<4, 256>, 256);
    else launch2(deform_im2col_lds<1, 256>, 256);
    STM_CHECK_LAUNCH("deform_im2col_lds");
    return STM_OK;
}

extern "C" int stm_validate_deform_geom(const stm_deform_geom* g) { return validate_geom(g, "stm_deform_geom"); }


// ---- planar variant for the inference graph (stmask_amd/planar.py): the same sampling arithmetic, but NHWC fp32 in (what the
// planar 1x1 convolution in front of it writes), the raw conv_offset_mask output pixel-major [pixels][3K] (what the planar
// offset convolution writes), and the columns out in the planar activation format -- three bf16 planes, channel-slab major,
// K index = tap * C + channel -- so that the deformable convolution's GEMM is a planar 1x1 convolution over 9C channels on
// the bf16 matrix cores (csrc/conv_bf16x.hip) instead of the fp32 GEMM, and no layout change is left either side.
// C/8 lanes per output pixel, 8 channels per lane: every corner read is a run of C contiguous floats, every store 16 bytes,
// the coefficients of a (pixel, tap) are computed by one lane and broadcast inside the pixel's lane group.
// Values are bit-identical to the NCHW kernels above (same expression order), then split exactly.
namespace {

typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));

struct SampleArgs {
    const float* x;      // [B, H, W, x_ld >= C]: pixel-major, C channels used
    const float* om;     // [B*Ho*Wo, om_ld]: 2K offsets (dy, dx per tap), then K mask logits in the modulated (DCNv2) form
    uint8_t* out;        // planes [3][K*C/32][out_np][32] bf16
    int B, H, W, C, Ho, Wo, sh, sw, ph, pw, dh, dw;
    int x_ld, kw, out_pix0;      // pixel stride of x (floats); kernel width (tap k = (k / kw, k % kw)); first output pixel in the planes
    int om_ld, out_np, M, fmt;   // fmt 0: three bf16 planes, 1: two fp16 planes, 2: one fp16 plane
    int* range_flag;             // fmt 1: raised when a sampled value has no fp16 representation (may be null)
    int xcd, per_xcd, nt;        // XCD-contiguous workgroup order (workgroups per XCD); nontemporal column stores
    int prefetch;                // streaming touch of the centre pixels ahead of the gathers
    long long out_pstride;   // bytes
};

typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_planes_f16(float a, float b, unsigned& p0, unsigned& p1)
{
    const f32x2v v = {a, b};
    const f16x2v h = __builtin_convertvector(v, f16x2v);
    const f32x2v r1 = (v - __builtin_convertvector(h, f32x2v)) * 2048.0f;   // STM_F16_LOW_SCALE of conv_bf16x.hip
    const f16x2v l = __builtin_convertvector(r1, f16x2v);
    p0 = __builtin_bit_cast(unsigned, h);
    p1 = __builtin_bit_cast(unsigned, l);
}

__device__ __forceinline__ void split2_planes(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2)
{
    const f32x2v v = {a, b};
    const bf16x2v h = __builtin_convertvector(v, bf16x2v);
    const f32x2v r1 = v - __builtin_convertvector(h, f32x2v);
    const bf16x2v m = __builtin_convertvector(r1, bf16x2v);
    const f32x2v r2 = r1 - __builtin_convertvector(m, f32x2v);
    const bf16x2v l = __builtin_convertvector(r2, bf16x2v);
    p0 = __builtin_bit_cast(unsigned, h);
    p1 = __builtin_bit_cast(unsigned, m);
    p2 = __builtin_bit_cast(unsigned, l);
}

// lanes per pixel: C = 8 * LPP (16, 32 or 64 lanes -> 4, 2 or 1 pixels per wave), 8 channels per lane; K = kh * kw taps (< LPP);
// MASK: modulated (dcn_v2.DCN: sigmoid mask logits after the offsets) or plain (mmcv DeformConv2d of FeatureAlign)
template <int LPP, int K = 9, bool MASK = true>
__global__ __launch_bounds__(256) void dcn_sample_planar_kernel(const SampleArgs a)
{
    constexpr int PPW = 64 / LPP;
    static_assert(K <= LPP, "one sub-lane per tap");
    const int lane = threadIdx.x & 63, sl = lane % LPP;
    // workgroup ids are dealt round-robin to the 8 XCDs: give each XCD a contiguous run of pixels, so that an input row is
    // gathered through one L2 instead of all eight
    int bid = blockIdx.x;
    if (a.xcd) {
        bid = (blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
        if ((int)(blockIdx.x >> 3) >= a.per_xcd) return;
    }
    const int m = (bid * 4 + (threadIdx.x >> 6)) * PPW + lane / LPP;
    if (bid * 4 * PPW >= a.M) return;
    const bool live = m < a.M;
    const int mm = live ? m : a.M - 1;                       // dead pixel groups shadow the last pixel, stores masked
    const int b = mm / (a.Ho * a.Wo);
    const int rem = mm - b * (a.Ho * a.Wo);
    const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
    const float* xb = a.x + (size_t)b * a.H * a.W * a.x_ld + sl * 8;
    // Touch the undeformed centre pixel of this output pixel first: a coalesced streaming load (the lanes of a pixel read its
    // C contiguous floats, consecutive pixels consecutive lines) that brings the neighbourhood the gathers are about to hit
    // into L2 with full memory-level parallelism.  With the input cold in HBM the gathers' own first-touch misses cost as
    // much as the whole rest of the kernel (230 vs 119 us on layer2 at batch 32); the value is only kept alive, never used.
    typedef float f32x4p __attribute__((ext_vector_type(4)));
    f32x4p pf = {0.f, 0.f, 0.f, 0.f};
    if (a.prefetch) {
        const int cy = min(max(ho * a.sh - a.ph + a.dh, 0), a.H - 1), cx = min(max(wo * a.sw - a.pw + a.dw, 0), a.W - 1);
        pf = *reinterpret_cast<const f32x4p*>(xb + (size_t)(cy * a.W + cx) * a.x_ld);
    }
    // sub-lane k (< 9) of each pixel group prepares tap k: corner weights with the mask folded in and clamped corner
    // offsets; the tap loop broadcasts them inside the group, so the per-tap work is 8 vector loads, 32 FMAs, the split and
    // three 16-byte stores per lane -- not LPP copies of the coefficient arithmetic
    float cw1 = 0.f, cw2 = 0.f, cw3 = 0.f, cw4 = 0.f;
    int ca1 = 0, ca2 = 0, ca3 = 0, ca4 = 0;
    if (sl < K) {
        const int i = sl / a.kw, j = sl - a.kw * i;
        const float* omp = a.om + (size_t)mm * a.om_ld;
        const float dy = omp[2 * sl], dx = omp[2 * sl + 1];
        const float mk = MASK ? sigmoidf_dev(omp[2 * K + sl]) : 1.0f;
        const float fy = (float)(ho * a.sh - a.ph + i * a.dh) + dy;
        const float fx = (float)(wo * a.sw - a.pw + j * a.dw) + dx;
        if (fy > -1.0f && fx > -1.0f && fy < (float)a.H && fx < (float)a.W) {
            const float fl_y = floorf(fy), fl_x = floorf(fx);
            const int h_low = (int)fl_y, w_low = (int)fl_x, h_high = h_low + 1, w_high = w_low + 1;
            const float lh = fy - fl_y, lw = fx - fl_x, hh = 1.0f - lh, hw = 1.0f - lw;
            const bool t = h_low >= 0, l = w_low >= 0, bt = h_high <= a.H - 1, r = w_high <= a.W - 1;
            const int hl = max(h_low, 0), wl = max(w_low, 0), hh_i = min(h_high, a.H - 1), wh_i = min(w_high, a.W - 1);
            cw1 = (t && l) ? hh * hw * mk : 0.f;
            cw2 = (t && r) ? hh * lw * mk : 0.f;
            cw3 = (bt && l) ? lh * hw * mk : 0.f;
            cw4 = (bt && r) ? lh * lw * mk : 0.f;
            ca1 = (hl * a.W + wl) * a.x_ld;
            ca2 = (hl * a.W + wh_i) * a.x_ld;
            ca3 = (hh_i * a.W + wl) * a.x_ld;
            ca4 = (hh_i * a.W + wh_i) * a.x_ld;
        }
    }
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float w1 = __shfl(cw1, k, LPP), w2 = __shfl(cw2, k, LPP), w3 = __shfl(cw3, k, LPP), w4 = __shfl(cw4, k, LPP);
        const int a1 = __shfl(ca1, k, LPP), a2 = __shfl(ca2, k, LPP), a3 = __shfl(ca3, k, LPP), a4 = __shfl(ca4, k, LPP);
        float x1[8], x2[8], x3[8], x4[8], v[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4v q1 = *reinterpret_cast<const f32x4v*>(xb + a1 + 4 * h), q2 = *reinterpret_cast<const f32x4v*>(xb + a2 + 4 * h);
            const f32x4v q3 = *reinterpret_cast<const f32x4v*>(xb + a3 + 4 * h), q4 = *reinterpret_cast<const f32x4v*>(xb + a4 + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) { x1[4 * h + e] = q1[e]; x2[4 * h + e] = q2[e]; x3[4 * h + e] = q3[e]; x4[4 * h + e] = q4[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = bilerp(w1, w2, w3, w4, x1[e], x2[e], x3[e], x4[e]);
        unsigned q0[4], q1[4], q2[4] = {0, 0, 0, 0};
        if (a.fmt >= 1) {
            unsigned mag = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) mag = max(mag, __builtin_bit_cast(unsigned, v[e]) & 0x7fffffffu);
            if (mag > 0x477fe000u && a.range_flag) *reinterpret_cast<volatile int*>(a.range_flag) = 1;   // > 65504, inf, nan
#pragma unroll
            for (int e = 0; e < 4; ++e) split2_planes_f16(v[2 * e], v[2 * e + 1], q0[e], q1[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) split2_planes(v[2 * e], v[2 * e + 1], q0[e], q1[e], q2[e]);
        }
        const u32x4v p0 = {q0[0], q0[1], q0[2], q0[3]}, p1 = {q1[0], q1[1], q1[2], q1[3]}, p2 = {q2[0], q2[1], q2[2], q2[3]};
        const int kc = k * a.C + sl * 8;                     // K index of the lane's first channel (8 | kc: inside one slab)
        uint8_t* o = a.out + (((size_t)(kc >> 5) * a.out_np + a.out_pix0 + mm) * 32 + (kc & 31)) * 2;
        if (live && a.nt) {          // the columns are far larger than L2 and read once, later: keep them out of the gathers' way
            __builtin_nontemporal_store(p0, reinterpret_cast<u32x4v*>(o));
            if (a.fmt != 2) __builtin_nontemporal_store(p1, reinterpret_cast<u32x4v*>(o + a.out_pstride));
            if (a.fmt == 0) __builtin_nontemporal_store(p2, reinterpret_cast<u32x4v*>(o + 2 * a.out_pstride));
        } else if (live) {
            *reinterpret_cast<u32x4v*>(o) = p0;
            if (a.fmt != 2) *reinterpret_cast<u32x4v*>(o + a.out_pstride) = p1;
            if (a.fmt == 0) *reinterpret_cast<u32x4v*>(o + 2 * a.out_pstride) = p2;
        }
    }
    asm volatile("" ::"v"(pf));      // the touch above stays in the program
}


// ---- round 4: the same sampler written straight-line for the fp16 two-plane columns the inference graph uses.  The kernel above carries its
// plane format, its store policy and the liveness of a pixel group as RUN-TIME branches inside the unrolled tap loop (~10 branches and exec-mask
// rebuilds per tap) and keeps one tap of loads in flight; this one fixes them at compile time (FMT = 1, nontemporal stores, dead groups masked by
// the exec mask of the stores only), takes the range check out of the loop (one running maximum, tested once), broadcasts a tap's coefficients
// with a DPP row share where a pixel's lane group is one row of 16 lanes (C = 128: no LDS crossbar instruction), and -- PIPE = 1 -- issues tap
// k + 1's eight corner loads before tap k is blended (96-110 VGPRs: 4 waves per SIMD instead of 8, two taps of loads in flight per wave).
// Same bilerp(), same split: columns bit-identical to the kernel above.  DS_ABL (make variant ... VFLAGS=-DDS_ABL=n, RESULTS ARE WRONG):
// 1 no column stores, 2 corners not loaded (constants), 4 every corner reads the tap's first corner (L1-resident gathers).
#ifndef DS_ABL
#define DS_ABL 0
#endif
template <int LPP, int K, bool MASK, int PIPE>
__global__ __launch_bounds__(256) void dcn_sample_planar_f16x2_kernel(const SampleArgs a)
{
    constexpr int PPW = 64 / LPP;
    static_assert(K <= LPP, "one sub-lane per tap");
    const int lane = threadIdx.x & 63, sl = lane % LPP;
    int bid = blockIdx.x;
    if (a.xcd) {
        bid = (blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
        if ((int)(blockIdx.x >> 3) >= a.per_xcd) return;
    }
    const int m = (bid * 4 + (threadIdx.x >> 6)) * PPW + lane / LPP;
    if (bid * 4 * PPW >= a.M) return;
    const bool live = m < a.M;
    const int mm = live ? m : a.M - 1;
    const int b = mm / (a.Ho * a.Wo);
    const int rem = mm - b * (a.Ho * a.Wo);
    const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
    const float* xb = a.x + (size_t)b * a.H * a.W * a.x_ld + sl * 8;
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    f32x4v pf = {0.f, 0.f, 0.f, 0.f};
    if (a.prefetch) {
        const int cy = min(max(ho * a.sh - a.ph + a.dh, 0), a.H - 1), cx = min(max(wo * a.sw - a.pw + a.dw, 0), a.W - 1);
        pf = *reinterpret_cast<const f32x4v*>(xb + (size_t)(cy * a.W + cx) * a.x_ld);
    }
    float cw1 = 0.f, cw2 = 0.f, cw3 = 0.f, cw4 = 0.f;
    int ca1 = 0, ca2 = 0, ca3 = 0, ca4 = 0;
    if (sl < K) {
        const int i = sl / a.kw, j = sl - a.kw * i;
        const float* omp = a.om + (size_t)mm * a.om_ld;
        const float dy = omp[2 * sl], dx = omp[2 * sl + 1];
        const float mk = MASK ? sigmoidf_dev(omp[2 * K + sl]) : 1.0f;
        const float fy = (float)(ho * a.sh - a.ph + i * a.dh) + dy;
        const float fx = (float)(wo * a.sw - a.pw + j * a.dw) + dx;
        if (fy > -1.0f && fx > -1.0f && fy < (float)a.H && fx < (float)a.W) {
            const float fl_y = floorf(fy), fl_x = floorf(fx);
            const int h_low = (int)fl_y, w_low = (int)fl_x, h_high = h_low + 1, w_high = w_low + 1;
            const float lh = fy - fl_y, lw = fx - fl_x, hh = 1.0f - lh, hw = 1.0f - lw;
            const bool t = h_low >= 0, l = w_low >= 0, bt = h_high <= a.H - 1, r = w_high <= a.W - 1;
            const int hl = max(h_low, 0), wl = max(w_low, 0), hh_i = min(h_high, a.H - 1), wh_i = min(w_high, a.W - 1);
            cw1 = (t && l) ? hh * hw * mk : 0.f;
            cw2 = (t && r) ? hh * lw * mk : 0.f;
            cw3 = (bt && l) ? lh * hw * mk : 0.f;
            cw4 = (bt && r) ? lh * lw * mk : 0.f;
            ca1 = (hl * a.W + wl) * a.x_ld;
            ca2 = (hl * a.W + wh_i) * a.x_ld;
            ca3 = (hh_i * a.W + wl) * a.x_ld;
            ca4 = (hh_i * a.W + wh_i) * a.x_ld;
        }
    }
    // a tap's coefficient from the sub-lane that prepared it: a DPP row share where the lane group IS a row of 16 lanes, else the crossbar
    auto bcast_i = [&](int v, int k) -> int {
        if constexpr (LPP == 16) {
            switch (k) {      // (the DPP control is an immediate)
#define STM_RS(n_) case n_: return __builtin_amdgcn_update_dpp(0, v, 0x150 + n_, 0xf, 0xf, false);
                STM_RS(0) STM_RS(1) STM_RS(2) STM_RS(3) STM_RS(4) STM_RS(5) STM_RS(6) STM_RS(7) STM_RS(8) STM_RS(9) STM_RS(10) STM_RS(11) STM_RS(12)
                STM_RS(13) STM_RS(14) STM_RS(15)
#undef STM_RS
                default: return v;
            }
        } else {
            return __shfl(v, k, LPP);
        }
    };
    auto bcast_f = [&](float v, int k) { return __builtin_bit_cast(float, bcast_i(__builtin_bit_cast(int, v), k)); };
    struct Tap { f32x4v q[4][2]; };
    auto load_tap = [&](int k, Tap& t) {
        const int a1 = bcast_i(ca1, k), a2 = bcast_i(ca2, k), a3 = bcast_i(ca3, k), a4 = bcast_i(ca4, k);
        const int ad[4] = {a1, (DS_ABL & 4) ? a1 : a2, (DS_ABL & 4) ? a1 : a3, (DS_ABL & 4) ? a1 : a4};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (DS_ABL & 2) t.q[c][h] = f32x4v{1.f + c, 2.f, 3.f + h, 4.f};
                else t.q[c][h] = *reinterpret_cast<const f32x4v*>(xb + ad[c] + 4 * h);
            }
    };
    unsigned mag = 0;
    const bool nt = a.nt != 0;
    auto blend_store = [&](int k, const Tap& t) {
        const float w1 = bcast_f(cw1, k), w2 = bcast_f(cw2, k), w3 = bcast_f(cw3, k), w4 = bcast_f(cw4, k);
        float v[8];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * h + e] = bilerp(w1, w2, w3, w4, t.q[0][h][e], t.q[1][h][e], t.q[2][h][e], t.q[3][h][e]);
        unsigned q0[4], q1[4];
#pragma unroll
        for (int e = 0; e < 8; ++e) mag = max(mag, __builtin_bit_cast(unsigned, v[e]) & 0x7fffffffu);
#pragma unroll
        for (int e = 0; e < 4; ++e) split2_planes_f16(v[2 * e], v[2 * e + 1], q0[e], q1[e]);
        const u32x4v p0 = {q0[0], q0[1], q0[2], q0[3]}, p1 = {q1[0], q1[1], q1[2], q1[3]};
        const int kc = k * a.C + sl * 8;
        uint8_t* o = a.out + (((size_t)(kc >> 5) * a.out_np + a.out_pix0 + mm) * 32 + (kc & 31)) * 2;
        if (DS_ABL & 1) { asm volatile("" ::"v"(p0), "v"(p1)); return; }
        if (live) {
            if (nt) {
                __builtin_nontemporal_store(p0, reinterpret_cast<u32x4v*>(o));
                __builtin_nontemporal_store(p1, reinterpret_cast<u32x4v*>(o + a.out_pstride));
            } else {
                *reinterpret_cast<u32x4v*>(o) = p0;
                *reinterpret_cast<u32x4v*>(o + a.out_pstride) = p1;
            }
        }
    };
    if constexpr (PIPE) {
        Tap t0, t1;
        load_tap(0, t0);
#pragma unroll
        for (int k = 0; k < K; k += 2) {
            if (k + 1 < K) load_tap(k + 1, t1);
            blend_store(k, t0);
            if (k + 2 < K) load_tap(k + 2, t0);
            if (k + 1 < K) blend_store(k + 1, t1);
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            Tap t;
            load_tap(k, t);
            blend_store(k, t);
        }
    }
    if (mag > 0x477fe000u && a.range_flag) *reinterpret_cast<volatile int*>(a.range_flag) = 1;   // > 65504, inf, nan: no fp16 plane representation
    asm volatile("" ::"v"(pf));
}


// ---- LDS-staged form of the planar sampler (the form BASELINE.json's north star names: "deformable im2col with LDS-staged input
// tiles and coalesced HBM offset reads").  The register-gather kernel above fetches every bilinear corner through the vector-memory
// path: 4 corners x 9 taps x C x 4 B = 18 KB per output pixel at C = 128 for 4.6 KB of columns, at the 64 B/clk a CU's L1 moves --
// 460 cycles per pixel measured, 3.9 TB/s algorithmic.  Here a workgroup owns a tile of TH x TW output pixels and walks the channels
// in chunks of CCH:
//   A. once: the coefficients of its (pixel, tap) pairs -- corner weights with the mask folded in, the four clamped corner coordinates
//      -- computed exactly as above and kept in LDS (24 B per pair); offsets / mask logits are read once, pixel-major rows;
//   B. per chunk: the input rectangle the tile can reach with offsets up to HALO pixels is staged once, coalesced (64-byte runs per
//      pixel, consecutive pixels consecutive), and the corners are gathered from LDS (256 B/clk); the columns leave as 16-byte
//      nontemporal stores, 16 consecutive pixels of one (tap, channel slab) = 512 B or 1 KB contiguous per plane.
// A pair with a contributing corner outside the staged rectangle (an offset beyond HALO) gathers that tap from global memory
// instead: same values, any offset.  The blend is the same bilerp() on the same operands: columns bit-identical to the kernel above.
constexpr int SL_TW = 16, SL_HALO = 3, SL_CCH = 32;
#ifndef SL_ABL
#define SL_ABL 0     // diagnostic builds (make EXTRA=-DSL_ABL=n, RESULTS ARE WRONG): 1 no column stores, 2 no corner gathers, 4 no staging
#endif
struct SampleLdsArgs {
    SampleArgs s;
    int th, rh, rw, tiles_y, tiles_x, kh;
};

struct SlCoef { float w[4]; short y0, x0, y1, x1; unsigned flags; };   // flags bit 0: gather this pair from global memory

template <bool MASK, int TH>
__global__ __launch_bounds__(256) void dcn_sample_planar_lds_kernel(const SampleLdsArgs aa)
{
    const SampleArgs& a = aa.s;
    extern __shared__ __align__(16) uint8_t sl_smem[];
    const int K = aa.kh * a.kw;
    constexpr int TP = TH * SL_TW;                                   // pixels of the tile (a power of two: item indices split with shifts)
    constexpr int SL_MAXQ = 16;                                      // staged float4 per thread and chunk (rectangles of up to 512 pixels)
    SlCoef* coef = reinterpret_cast<SlCoef*>(sl_smem);               // [TP][K]
    float* reg = reinterpret_cast<float*>(sl_smem + (((size_t)TP * K * sizeof(SlCoef) + 15) & ~(size_t)15));   // [rh][rw][CCH]
    const int64_t nblk = (int64_t)a.B * aa.tiles_y * aa.tiles_x;
    const int64_t blk = a.xcd ? stm_xcd_block(nblk) : (int64_t)blockIdx.x;
    if (blk < 0 || blk >= nblk) return;
    const int tx = (int)(blk % aa.tiles_x), ty = (int)((blk / aa.tiles_x) % aa.tiles_y), b = (int)(blk / ((int64_t)aa.tiles_x * aa.tiles_y));
    const int oy0 = ty * TH, ox0 = tx * SL_TW;
    const int in_y0 = oy0 * a.sh - a.ph - SL_HALO, in_x0 = ox0 * a.sw - a.pw - SL_HALO;
    const int tid = threadIdx.x;
    const float* xb = a.x + (size_t)b * a.H * a.W * a.x_ld;

    // ---- A. coefficients
    for (int it = tid; it < TP * K; it += 256) {
        const int p = it / K, k = it - p * K;
        const int ho = oy0 + p / SL_TW, wo = ox0 + (p % SL_TW);
        SlCoef c;
        c.w[0] = c.w[1] = c.w[2] = c.w[3] = 0.f;
        c.y0 = c.x0 = c.y1 = c.x1 = 0;
        c.flags = 0;
        if (ho < a.Ho && wo < a.Wo) {
            const int i = k / a.kw, j = k - a.kw * i;
            const float* omp = a.om + ((size_t)(b * a.Ho + ho) * a.Wo + wo) * a.om_ld;
            const float dy = omp[2 * k], dx = omp[2 * k + 1];
            const float mk = MASK ? sigmoidf_dev(omp[2 * K + k]) : 1.0f;
            const float fy = (float)(ho * a.sh - a.ph + i * a.dh) + dy;
            const float fx = (float)(wo * a.sw - a.pw + j * a.dw) + dx;
            if (fy > -1.0f && fx > -1.0f && fy < (float)a.H && fx < (float)a.W) {
                const float fl_y = floorf(fy), fl_x = floorf(fx);
                const int h_low = (int)fl_y, w_low = (int)fl_x, h_high = h_low + 1, w_high = w_low + 1;
                const float lh = fy - fl_y, lw = fx - fl_x, hh = 1.0f - lh, hw = 1.0f - lw;
                const bool t = h_low >= 0, l = w_low >= 0, bt = h_high <= a.H - 1, r = w_high <= a.W - 1;
                const int hl = max(h_low, 0), wl = max(w_low, 0), hh_i = min(h_high, a.H - 1), wh_i = min(w_high, a.W - 1);
                c.w[0] = (t && l) ? hh * hw * mk : 0.f;
                c.w[1] = (t && r) ? hh * lw * mk : 0.f;
                c.w[2] = (bt && l) ? lh * hw * mk : 0.f;
                c.w[3] = (bt && r) ? lh * lw * mk : 0.f;
                c.y0 = (short)hl; c.x0 = (short)wl; c.y1 = (short)hh_i; c.x1 = (short)wh_i;
                // all four (clamped) corners inside the staged rectangle?  (clamped corners carry weight 0 but are still read)
                const bool in = hl >= in_y0 && hh_i < in_y0 + aa.rh && wl >= in_x0 && wh_i < in_x0 + aa.rw;
                c.flags = in ? 0u : 1u;
            }
        }
        coef[it] = c;
    }

    typedef float f32x4v __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    constexpr int QPP = SL_CCH / 4;                                   // float4 per staged pixel
    constexpr int GPP = SL_CCH / 8;                                   // 8-channel groups per pixel and chunk
    const int nreg = aa.rh * aa.rw;
    // staging plan, once: float4 number tid + 256 u of the rectangle -> element offset in the image (-1: outside, stays zero)
    int g_off[SL_MAXQ];
#pragma unroll
    for (int u = 0; u < SL_MAXQ; ++u) {
        const int it = tid + 256 * u;
        g_off[u] = -1;
        if (it < nreg * QPP) {
            const int rp = it / QPP, q = it - rp * QPP;
            const int ry = rp / aa.rw;
            const int y = in_y0 + ry, x = in_x0 + (rp - ry * aa.rw);
            if ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W) g_off[u] = (y * a.W + x) * a.x_ld + 4 * q;
        }
    }
    // ---- B1. the rectangle of a chunk travels global -> registers -> LDS; the loads of chunk c + 1 are in flight while chunk c is
    // blended.  Outside the image: zeros (never blended with a non-zero weight)
    f32x4v st[SL_MAXQ];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int u = 0; u < SL_MAXQ; ++u) {
            st[u] = f32x4v{0.f, 0.f, 0.f, 0.f};
            if (!(SL_ABL & 4) && g_off[u] >= 0) st[u] = *reinterpret_cast<const f32x4v*>(xb + g_off[u] + c0);
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < a.C; c0 += SL_CCH) {
        __syncthreads();                                              // previous chunk consumed (first pass: coefficients written)
#pragma unroll
        for (int u = 0; u < SL_MAXQ; ++u)
            if (tid + 256 * u < nreg * QPP) *reinterpret_cast<f32x4v*>(reg + (size_t)(tid + 256 * u) * 4) = st[u];
        __syncthreads();
        if (c0 + SL_CCH < a.C) fetch(c0 + SL_CCH);
        // ---- B2. blend: item = (tap, pixel, 8-channel group); a wave's 64 items = 32 pixels x 2 groups of one tap
        for (int it = tid; it < K * TP * GPP; it += 256) {
            const int g8 = it % GPP, p = (it / GPP) % TP, k = it / (GPP * TP);     // (all powers of two)
            const int ho = oy0 + p / SL_TW, wo = ox0 + (p % SL_TW);
            if (ho >= a.Ho || wo >= a.Wo) continue;
            const SlCoef c = coef[p * K + k];
            float x1[8], x2[8], x3[8], x4[8], v[8];
            if (c.flags & 1u) {
                const float* gx = xb + c0 + 8 * g8;
                const size_t a1 = ((size_t)c.y0 * a.W + c.x0) * a.x_ld, a2 = ((size_t)c.y0 * a.W + c.x1) * a.x_ld;
                const size_t a3 = ((size_t)c.y1 * a.W + c.x0) * a.x_ld, a4 = ((size_t)c.y1 * a.W + c.x1) * a.x_ld;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4v q1 = *reinterpret_cast<const f32x4v*>(gx + a1 + 4 * h), q2 = *reinterpret_cast<const f32x4v*>(gx + a2 + 4 * h);
                    const f32x4v q3 = *reinterpret_cast<const f32x4v*>(gx + a3 + 4 * h), q4 = *reinterpret_cast<const f32x4v*>(gx + a4 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { x1[4 * h + e] = q1[e]; x2[4 * h + e] = q2[e]; x3[4 * h + e] = q3[e]; x4[4 * h + e] = q4[e]; }
                }
            } else if (SL_ABL & 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { x1[e] = 1.f; x2[e] = 2.f; x3[e] = 3.f; x4[e] = (float)e; }
            } else {
                const float* lx = reg + 8 * g8;
                const int r0 = (c.y0 - in_y0) * aa.rw - in_x0, r1 = (c.y1 - in_y0) * aa.rw - in_x0;
                const int a1 = (r0 + c.x0) * SL_CCH, a2 = (r0 + c.x1) * SL_CCH, a3 = (r1 + c.x0) * SL_CCH, a4 = (r1 + c.x1) * SL_CCH;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4v q1 = *reinterpret_cast<const f32x4v*>(lx + a1 + 4 * h), q2 = *reinterpret_cast<const f32x4v*>(lx + a2 + 4 * h);
                    const f32x4v q3 = *reinterpret_cast<const f32x4v*>(lx + a3 + 4 * h), q4 = *reinterpret_cast<const f32x4v*>(lx + a4 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { x1[4 * h + e] = q1[e]; x2[4 * h + e] = q2[e]; x3[4 * h + e] = q3[e]; x4[4 * h + e] = q4[e]; }
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = bilerp(c.w[0], c.w[1], c.w[2], c.w[3], x1[e], x2[e], x3[e], x4[e]);
            unsigned q0[4], q1[4], q2[4] = {0, 0, 0, 0};
            if (a.fmt >= 1) {
                unsigned mag = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) mag = max(mag, __builtin_bit_cast(unsigned, v[e]) & 0x7fffffffu);
                if (mag > 0x477fe000u && a.range_flag) *reinterpret_cast<volatile int*>(a.range_flag) = 1;
#pragma unroll
                for (int e = 0; e < 4; ++e) split2_planes_f16(v[2 * e], v[2 * e + 1], q0[e], q1[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) split2_planes(v[2 * e], v[2 * e + 1], q0[e], q1[e], q2[e]);
            }
            const u32x4v p0 = {q0[0], q0[1], q0[2], q0[3]}, p1 = {q1[0], q1[1], q1[2], q1[3]}, p2 = {q2[0], q2[1], q2[2], q2[3]};
            const int kcol = k * a.C + c0 + 8 * g8;                  // K index of the item's first channel
            const size_t mm = ((size_t)b * a.Ho + ho) * a.Wo + wo;
            uint8_t* o = a.out + (((size_t)(kcol >> 5) * a.out_np + a.out_pix0 + mm) * 32 + (kcol & 31)) * 2;
            if ((SL_ABL & 1) && q0[0] != 0x12345678u) continue;
            if (a.nt) {
                __builtin_nontemporal_store(p0, reinterpret_cast<u32x4v*>(o));
                if (a.fmt != 2) __builtin_nontemporal_store(p1, reinterpret_cast<u32x4v*>(o + a.out_pstride));
                if (a.fmt == 0) __builtin_nontemporal_store(p2, reinterpret_cast<u32x4v*>(o + 2 * a.out_pstride));
            } else {
                *reinterpret_cast<u32x4v*>(o) = p0;
                if (a.fmt != 2) *reinterpret_cast<u32x4v*>(o + a.out_pstride) = p1;
                if (a.fmt == 0) *reinterpret_cast<u32x4v*>(o + 2 * a.out_pstride) = p2;
            }
        }
    }
}

}  // namespace

extern "C" int stm_dcn_sample_planar_fmt_f32(const float* x, const float* offset_mask, int om_ld, void* planes, int out_np,
                                             long long out_plane_stride, const stm_deform_geom* g, int fmt, stm_stream_t stream);
extern "C" int stm_dcn_sample_planar_f32(const float* x, const float* offset_mask, int om_ld, void* planes, int out_np,
                                         long long out_plane_stride, const stm_deform_geom* g, stm_stream_t stream)
{
    return stm_dcn_sample_planar_fmt_f32(x, offset_mask, om_ld, planes, out_np, out_plane_stride, g, 0, stream);
}

extern "C" int stm_deform_sample_planar_f32(const float* x, int x_ld, const float* offsets, int om_ld, int has_mask, void* planes, int out_np,
                                            int out_pixel_offset, long long out_plane_stride, const stm_deform_geom* g, int fmt,
                                            stm_stream_t stream);
extern "C" int stm_dcn_sample_planar_fmt_f32(const float* x, const float* offset_mask, int om_ld, void* planes, int out_np,
                                             long long out_plane_stride, const stm_deform_geom* g, int fmt, stm_stream_t stream)
{
    STM_REQUIRE(g, STM_ENULL, "stm_dcn_sample_planar_f32: NULL argument");
    return stm_deform_sample_planar_f32(x, g->C, offset_mask, om_ld, 1, planes, out_np, 0, out_plane_stride, g, fmt, stream);
}

extern "C" int stm_deform_sample_planar_f32(const float* x, int x_ld, const float* offsets, int om_ld, int has_mask, void* planes, int out_np,
                                            int out_pixel_offset, long long out_plane_stride, const stm_deform_geom* g, int fmt,
                                            stm_stream_t stream)
{
    const char* who = "stm_deform_sample_planar_f32";
    STM_REQUIRE(fmt >= 0 && fmt <= 2, STM_EINVAL, "%s: fmt must be 0, 1 or 2", who);
    STM_REQUIRE(x && offsets && planes && g, STM_ENULL, "%s: NULL argument", who);
    const int K = g->kh * g->kw;
    STM_REQUIRE(g->dg == 1 && (K == 9 || (K == 15 && !has_mask)), STM_EUNSUPPORTED,
                "%s: one deformable group; 3x3 taps, or 3x5 / 5x3 without mask (got %dx%d, dg %d)", who, g->kh, g->kw, g->dg);
    STM_REQUIRE(g->C == 128 || g->C == 256 || g->C == 512, STM_EUNSUPPORTED, "%s: C must be 128, 256 or 512 (got %d)", who, g->C);
    STM_REQUIRE(has_mask || g->C == 256, STM_EUNSUPPORTED, "%s: the mask-free form is built for C = 256 (got %d)", who, g->C);
    STM_REQUIRE(g->B > 0 && g->H > 0 && g->W > 0 && g->Ho > 0 && g->Wo > 0 && om_ld >= (has_mask ? 3 : 2) * K && x_ld >= g->C && x_ld % 4 == 0 &&
                out_pixel_offset >= 0, STM_EINVAL, "%s: bad geometry", who);
    const int64_t M = (int64_t)g->B * g->Ho * g->Wo;
    STM_REQUIRE(M < ((int64_t)1 << 30) && (int64_t)g->B * g->H * g->W * x_ld < ((int64_t)1 << 31), STM_EUNSUPPORTED,
                "%s: tensor too large for 32-bit indexing", who);
    STM_REQUIRE(out_np <= 0 || out_pixel_offset + M <= out_np, STM_EINVAL, "%s: output pixels [%d, %lld) exceed the planes (%d)", who,
                out_pixel_offset, (long long)(out_pixel_offset + M), out_np);
    SampleArgs a;
    a.x = x; a.om = offsets; a.out = static_cast<uint8_t*>(planes);
    a.B = g->B; a.H = g->H; a.W = g->W; a.C = g->C; a.Ho = g->Ho; a.Wo = g->Wo;
    a.sh = g->sh; a.sw = g->sw; a.ph = g->ph; a.pw = g->pw; a.dh = g->dh; a.dw = g->dw;
    a.x_ld = x_ld; a.kw = g->kw; a.out_pix0 = out_pixel_offset;
    a.om_ld = om_ld; a.M = (int)M; a.out_np = out_np > 0 ? out_np : (int)M; a.fmt = fmt; a.range_flag = stm_internal_range_flag();
    a.out_pstride = (out_plane_stride > 0 ? out_plane_stride : (long long)(K * g->C / 32) * a.out_np * 32) * 2;
    const int ppw = 512 / g->C;                                  // pixels per wave
    const int nblk = stm_cdiv(M, 4 * ppw);
    // XCD-contiguous block order, nontemporal column stores (-1: by size), centre-pixel touch: A/B switches until round 6, all on since round 2
    const int env_xcd = 1, env_nt = -1, env_prefetch = 1;
    a.xcd = env_xcd;
    // nontemporal column stores: 248 -> 115 us on layer2 at batch 32 together with the XCD order (5.6 TB/s algorithmic), but
    // 34 -> 45 us with 512 channels (one pixel per wave, 1-KB runs per tap) -- so up to 256 channels only
    a.nt = env_nt >= 0 ? env_nt : (g->C <= 256 ? 1 : 0);
    a.per_xcd = stm_cdiv(nblk, 8);
    a.prefetch = env_prefetch;
    // LDS-staged form (STM_DCN_LDS=1; default off): tiles of 4 x 16 (stride 1) or 2 x 16 (stride 2) output pixels, 32-channel chunks.
    // Measured at batch 32 on cold inputs (scripts/ab_dcn_lds.py, profiles/r03_dcn_sampler_ab.txt): 241 vs 253 us and 183 vs 177 us on
    // the two layer2 shapes, 1.1-2.7x SLOWER on the smaller layers (few tiles, two waves per SIMD).  Its ablations say why the staging
    // does not pay: without the column stores 107 us, without the LDS gathers the same, with neither stores, gathers nor staging 83 us --
    // the per-value arithmetic (blend, plane split, range check: ~10 VALU per value, 1152 values per pixel) and the loop around it are
    // the floor of either form, and the register-gather kernel hides its gathers behind that arithmetic with 8 waves per SIMD.
    const int env_lds = STM_ENV_INT("STM_DCN_LDS", 0);
    if (env_lds && g->dh == 1 && g->dw == 1 && g->sh == g->sw && (g->sh == 1 || g->sh == 2) && g->C % SL_CCH == 0 && g->H < 32768 && g->W < 32768) {
        SampleLdsArgs aa;
        aa.s = a;
        aa.kh = g->kh;
        aa.th = g->sh == 1 ? 4 : 2;
        aa.rh = (aa.th - 1) * g->sh + (g->kh - 1) + 2 + 2 * SL_HALO;
        aa.rw = (SL_TW - 1) * g->sw + (g->kw - 1) + 2 + 2 * SL_HALO;
        aa.tiles_y = stm_cdiv(g->Ho, aa.th);
        aa.tiles_x = stm_cdiv(g->Wo, SL_TW);
        const size_t coef_b = ((size_t)aa.th * SL_TW * K * sizeof(SlCoef) + 15) & ~(size_t)15;
        const size_t lds = coef_b + (size_t)aa.rh * aa.rw * SL_CCH * sizeof(float);
        if (lds <= 80 * 1024 && aa.rh * aa.rw * (SL_CCH / 4) <= 16 * 256) {
            const int64_t nb = (int64_t)g->B * aa.tiles_y * aa.tiles_x;
            const dim3 grid_l(a.xcd ? stm_xcd_grid(nb) : (unsigned)nb);
            static std::atomic<int> reserved[4][32];
            int dev = 0;
            const bool have_dev = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 32;
            const int which = (has_mask ? 2 : 0) + (aa.th == 4 ? 1 : 0);   // (tile height 4: stride 1, 2: stride 2)
            if (!have_dev || reserved[which][dev].load(std::memory_order_relaxed) < (int)lds) {
                const void* fn = has_mask ? (aa.th == 4 ? reinterpret_cast<const void*>(dcn_sample_planar_lds_kernel<true, 4>)
                                                        : reinterpret_cast<const void*>(dcn_sample_planar_lds_kernel<true, 2>))
                                          : (aa.th == 4 ? reinterpret_cast<const void*>(dcn_sample_planar_lds_kernel<false, 4>)
                                                        : reinterpret_cast<const void*>(dcn_sample_planar_lds_kernel<false, 2>));
                STM_REQUIRE(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess, STM_ELAUNCH,
                            "%s: cannot reserve %zu bytes of LDS", who, lds);
                if (have_dev) reserved[which][dev].store((int)lds, std::memory_order_relaxed);
            }
            if (has_mask && aa.th == 4) hipLaunchKernelGGL((dcn_sample_planar_lds_kernel<true, 4>), grid_l, dim3(256), lds, stm_hs(stream), aa);
            else if (has_mask) hipLaunchKernelGGL((dcn_sample_planar_lds_kernel<true, 2>), grid_l, dim3(256), lds, stm_hs(stream), aa);
            else if (aa.th == 4) hipLaunchKernelGGL((dcn_sample_planar_lds_kernel<false, 4>), grid_l, dim3(256), lds, stm_hs(stream), aa);
            else hipLaunchKernelGGL((dcn_sample_planar_lds_kernel<false, 2>), grid_l, dim3(256), lds, stm_hs(stream), aa);
            STM_CHECK_LAUNCH("dcn_sample_planar_lds_kernel");
            return STM_OK;
        }
    }
    const dim3 grid(a.xcd ? 8 * a.per_xcd : nblk);
    // Sampler form: 0 = the run-time-format kernel (rounds 1-3: 64 registers, 8 waves per SIMD); the straight-line fp16x2 kernel: 1 = registers as
    // the compiler likes (116: 4 waves per SIMD, it hoists the next taps' loads by itself), 2 = explicit one-tap look-ahead (143: 3 waves).  Default
    // (-1): form 2 on the stride-2 layers of 256 / 512 channels, form 0 elsewhere -- profiles/r04_dcn_sampler_forms.txt: 123 vs 127 us and 65 vs 77 us
    // there, 20-50 % slower on the stride-1 layers; forms capped to 6 / 8 waves per SIMD spilled (148 / 220 B per lane) and ran at half the rate.
    int variant = -1;        // (-1: the rule below; 0 / 1 / 2 forced one form everywhere -- an environment switch until round 6)
    if (variant < 0) variant = (g->sh == 2 && g->C >= 256) ? 2 : 0;
    if (variant && fmt == 1 && has_mask && K == 9 && (g->C == 128 || g->C == 256 || g->C == 512)) {
#define STM_DSV(LPP_) \
        if (variant == 2) hipLaunchKernelGGL((dcn_sample_planar_f16x2_kernel<LPP_, 9, true, 1>), grid, dim3(256), 0, stm_hs(stream), a); \
        else hipLaunchKernelGGL((dcn_sample_planar_f16x2_kernel<LPP_, 9, true, 0>), grid, dim3(256), 0, stm_hs(stream), a);
        if (g->C == 128) { STM_DSV(16) } else if (g->C == 256) { STM_DSV(32) } else { STM_DSV(64) }
#undef STM_DSV
        STM_CHECK_LAUNCH("dcn_sample_planar_f16x2_kernel");
        return STM_OK;
    }
    if (!has_mask && K == 15) hipLaunchKernelGGL((dcn_sample_planar_kernel<32, 15, false>), grid, dim3(256), 0, stm_hs(stream), a);
    else if (!has_mask) hipLaunchKernelGGL((dcn_sample_planar_kernel<32, 9, false>), grid, dim3(256), 0, stm_hs(stream), a);
    else if (g->C == 128) hipLaunchKernelGGL((dcn_sample_planar_kernel<16>), grid, dim3(256), 0, stm_hs(stream), a);
    else if (g->C == 256) hipLaunchKernelGGL((dcn_sample_planar_kernel<32>), grid, dim3(256), 0, stm_hs(stream), a);
    else hipLaunchKernelGGL((dcn_sample_planar_kernel<64>), grid, dim3(256), 0, stm_hs(stream), a);
    STM_CHECK_LAUNCH("dcn_sample_planar_kernel");
    return STM_OK;
}

// `_f16` form of the planar sampler (BASELINE config 5): columns as ONE fp16 plane
extern "C" int stm_dcn_sample_planar_f16(const float* x, const float* offset_mask, int om_ld, void* planes, int out_np, long long out_plane_stride,
                                         const stm_deform_geom* g, stm_stream_t stream)
{
    return stm_dcn_sample_planar_fmt_f32(x, offset_mask, om_ld, planes, out_np, out_plane_stride, g, 2, stream);
}
