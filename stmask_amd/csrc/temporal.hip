// temporal.hip -- temporal-fusion kernels for gfx950: spatial correlation (patch sampler) and RoIAlign.
//
// Correlation replaces spatial_correlation_sampler.spatial_correlation_sample as called by
// layers/modules/track_to_segment_head.py:53-59 (kernel_size 1, patch 11, stride 1, padding 0) plus the
// reference's follow-up `/ C` and leaky_relu_(0.1) (:60-62).  2.4 MB of traffic and 59.5 MFLOP per frame pair:
// bandwidth/latency bound, so the kernel is organised around LDS reuse, not MFMA.
//
//   One workgroup = one output row y of one image.  Work item = (4 consecutive x, one displacement row i):
//   it needs f1[c][y][x0..x0+3] and the 14 values f2[c][y+i-5][x0-5..x0+8] -- a sliding window held in
//   registers -- and produces 4 x 11 dot products (44 FMAs per channel for 5 ds_read_b128/b64).  Channels are
//   staged through LDS in chunks (f2 rows zero-padded by 5 on each side so no border tests remain).  A wave
//   carries 32 items x 2 channel halves: lanes l and l+32 work on the same item with different channels and
//   the halves are summed at the end with a wavefront shuffle (ds_swizzle / permlane), then scaled and
//   leaky-ReLU'd in registers before one coalesced store.
#include "stm_common.h"

namespace {


// IN_NHWC: f1 / f2 are channels-last [B][H][W][C] (C % 4 == 0) -- the layout the trunk's fp32 outputs have; a staging unit is then
// (f2 row, x, 4 channels) instead of (channel, f2 row, 4 x): same LDS image, same arithmetic, no NCHW copy of the feature maps.
// CCK: channels staged per chunk -- 16 for rows of up to 44 pixels, 8 for rows of up to 88 (the 46x80 P4 level of 736x1280 frames, BASELINE
// config 5, which fell to the generic kernel before: 3.5 ms per step at 4 clips).
// MAXU: float4 staging units per thread and chunk (8 covers every row length the launcher sends here, 7 covers W <= 40 at CCK = 16 -- the 24x40 level
// of 384x640 frames); WPS: waves per SIMD the register allocation must leave room for.  Round 4: the kernel sat at 188 VGPRs = 2 waves per SIMD, so
// the 768 workgroups of a 32-clip step ran as 1.5 rounds over 512 slots; with the staging plan packed into one register per unit (global offset in
// the low 20 bits, LDS word offset above), 7 units and a 168-register budget all 768 are resident at once.
template <int P, bool IN_NHWC, int CCK = 16, int MAXU = 8, int WPS = 2>
__global__ __launch_bounds__(256, WPS) void corr_patch_tiled(const float* __restrict__ f1, const float* __restrict__ f2,
                                                        float* __restrict__ out, int C, int H, int W, float scale,
                                                        float slope, int B, int out_ld)
{
    constexpr int R = P / 2;
    constexpr int WIN = 4 + P - 1;          // f2 values per item per channel (14 for P = 11)
    extern __shared__ float smem[];
    const int Wq = W / 4;                   // x quads (W % 4 == 0 on this path)
    const int LW2 = ((W + 2 * R + 3) / 4) * 4;   // padded f2 row length (multiple of 4)
    float* f1s = smem;                      // [CCK][W]
    float* f2s = smem + CCK * W;       // [CCK][P][LW2], index x + R

    // One workgroup per output row; the 11 workgroups whose windows share an f2 row must meet in ONE L2: workgroup ids are
    // dealt round-robin to the 8 XCDs, so the (image, row) list is cut into 8 contiguous runs, one per XCD (stm_xcd_block).
    // With plain (row, image) grid order every XCD's L2 fetched every row: 297 MB from HBM for 62.9 MB of inputs at batch 32
    // (rocprofv3 FETCH_SIZE, round 1).
    const int64_t blk = stm_xcd_block((int64_t)H * B);
    if (blk < 0) return;
    const int y = (int)(blk % H), b = (int)(blk / H);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int half = lane >> 5;             // channel half handled by this lane
    const int slot = wave * 32 + (lane & 31);  // 128 item slots per pass
    const int items = Wq * P;
    const int64_t HW = (int64_t)H * W;
    const float* f1b = f1 + (int64_t)b * C * HW;
    const float* f2b = f2 + (int64_t)b * C * HW;
    constexpr int CQ = CCK / 4;        // channel quads per chunk (channels-last staging)

    // ---- staging plan, computed ONCE: unit u = (channel c, f2 row rr, x quad) -> one float4 global load + 4 LDS words.
    //      Only in-image rows are ever loaded; halo columns and out-of-image rows are zeroed once and never touched again.
    const int units = CCK * P * Wq;
    // plan[u] = global offset in float4 units (low 18 bits; NHWC: (pixel * C + 4 cq) / 4, NCHW: (c * HW + pixel) / 4 -- both < 2^18 for the maps
    // the launcher admits) | LDS word offset << 18; 0xffffffff = nothing to stage
    unsigned plan[MAXU];
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
        int id = tid + u * 256;
        int g_off_u = -1, l_off_u = 0;
        if (id < units) {
            if constexpr (IN_NHWC) {
                // unit = (f2 row rr, x, channel quad): 16 bytes of 4 channels at one pixel; the quad is the fastest index, so the
                // CQ lanes of a pixel read 64 contiguous bytes
                const int cq = id % CQ, rem = id / CQ;
                const int rr = rem / W, x = rem - rr * W;
                const int yy = y + rr - R;
                if (yy >= 0 && yy < H) {
                    g_off_u = (yy * W + x) * C + cq * 4;
                    l_off_u = (cq * 4 * P + rr) * LW2 + R + x;
                }
            } else {
                int c = id / (P * Wq), rem = id - c * (P * Wq);
                int rr = rem / Wq, xq = rem - rr * Wq;
                int yy = y + rr - R;
                if (yy >= 0 && yy < H) {
                    g_off_u = c * (int)HW + yy * W + xq * 4;
                    l_off_u = (c * P + rr) * LW2 + R + xq * 4;
                }
            }
        }
        plan[u] = g_off_u < 0 ? 0xffffffffu : ((unsigned)g_off_u >> 2) | ((unsigned)l_off_u << 18);
    }
    const bool f1_unit = IN_NHWC ? tid < CQ * W : tid < CCK * Wq;
    const int f1_c = IN_NHWC ? (tid % CQ) * 4 : tid / Wq;            // first channel of the unit
    const int f1_xq = IN_NHWC ? tid / CQ : tid - f1_c * Wq;          // its x (channels-last) / x quad
    for (int idx = tid; idx < CCK * P * LW2; idx += 256) f2s[idx] = 0.0f;

    float4 pf[MAXU], pf1;
    auto prefetch = [&](int c0) {           // channels beyond C read as zero
#pragma unroll
        for (int u = 0; u < MAXU; ++u) {
            pf[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (plan[u] != 0xffffffffu) {
                const int g4 = (int)(plan[u] & 0x3ffffu) * 4;
                if constexpr (IN_NHWC) {
                    const int c = ((tid + u * 256) % CQ) * 4;     // C % 4 == 0: a quad is inside or outside as a whole
                    if (c0 + c < C) pf[u] = *reinterpret_cast<const float4*>(f2b + g4 + c0);
                } else {
                    int c = (tid + u * 256) / (P * Wq);
                    if (c0 + c < C) pf[u] = *reinterpret_cast<const float4*>(f2b + (int64_t)c0 * HW + g4);
                }
            }
        }
        pf1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f1_unit && c0 + f1_c < C) {
            if constexpr (IN_NHWC) pf1 = *reinterpret_cast<const float4*>(f1b + ((int64_t)y * W + f1_xq) * C + c0 + f1_c);
            else pf1 = *reinterpret_cast<const float4*>(f1b + (int64_t)(c0 + f1_c) * HW + (int64_t)y * W + f1_xq * 4);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < MAXU; ++u)
            if (plan[u] != 0xffffffffu) {
                float* d = f2s + (plan[u] >> 18);
                constexpr int ST = IN_NHWC ? 0 : 1;              // word stride between the unit's 4 values: channels (P * LW2) or x (1)
                const int st = ST ? 1 : P * LW2;
                d[0] = pf[u].x; d[st] = pf[u].y; d[2 * st] = pf[u].z; d[3 * st] = pf[u].w;
            }
        if (f1_unit) {
            if constexpr (IN_NHWC) {
                float* d = f1s + f1_c * W + f1_xq;
                d[0] = pf1.x; d[W] = pf1.y; d[2 * W] = pf1.z; d[3 * W] = pf1.w;
            } else {
                *reinterpret_cast<float4*>(f1s + f1_c * W + f1_xq * 4) = pf1;
            }
        }
    };

    for (int pass0 = 0; pass0 < items; pass0 += 128) {
        const int item = pass0 + slot;
        const bool active = item < items;
        const int i = active ? item / Wq : 0;        // displacement row
        const int xq = active ? item - i * Wq : 0;   // x quad
        const int x0 = xq * 4;
        float acc[4][P];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int j = 0; j < P; ++j) acc[p][j] = 0.0f;

        prefetch(0);
        for (int c0 = 0; c0 < C; c0 += CCK) {
            __syncthreads();                 // previous chunk fully consumed (and the zero fill done)
            commit();
            __syncthreads();
            if (c0 + CCK < C) prefetch(c0 + CCK);   // next chunk's loads fly behind the FMAs below
            if (active) {
#pragma unroll 2
                for (int cc = 0; cc < CCK / 2; ++cc) {
                    const int c = half * (CCK / 2) + cc;
                    const float4 a = *reinterpret_cast<const float4*>(&f1s[c * W + x0]);
                    const float* wrow = &f2s[(c * P + i) * LW2 + x0];  // window starts at x0 - R  (index x0)
                    float win[WIN];
#pragma unroll
                    for (int t = 0; t + 3 < WIN; t += 4) {
                        float4 v = *reinterpret_cast<const float4*>(wrow + t);
                        win[t] = v.x; win[t + 1] = v.y; win[t + 2] = v.z; win[t + 3] = v.w;
                    }
#pragma unroll
                    for (int t = (WIN / 4) * 4; t < WIN; ++t) win[t] = wrow[t];
                    const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                    for (int p = 0; p < 4; ++p)
#pragma unroll
                        for (int j = 0; j < P; ++j) acc[p][j] = fmaf(av[p], win[p + j], acc[p][j]);
                }
            }
        }
        // sum the two channel halves held by lanes l and l + 32 (wavefront shuffle), activate, store
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int j = 0; j < P; ++j) {
                float v = acc[p][j];
                v += __shfl_xor(v, 32, 64);
                acc[p][j] = v;
            }
        if (active && half == 0) {
#pragma unroll
            for (int j = 0; j < P; ++j) {
                float o[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float v = acc[p][j] * scale;
                    o[p] = v < 0.0f ? v * slope : v;
                }
                if (out_ld == 0) {
                    float* op = out + ((((int64_t)b * P + i) * P + j) * H + y) * W + x0;
                    *reinterpret_cast<float4*>(op) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
                    // channels-last [B][H][W][out_ld], channel i * P + j: what the RoI kernel of the temporal fusion gathers (a
                    // pixel's 121 displacements in 4 cache lines instead of 121)
                    float* op = out + (((int64_t)b * H + y) * W + x0) * out_ld + i * P + j;
#pragma unroll
                    for (int p = 0; p < 4; ++p) op[(int64_t)p * out_ld] = o[p];
                }
            }
        }
    }
}

// Generic fallback (any patch size / patch dilation): one thread per output, channel loop from global.
__global__ void corr_patch_generic(const float* __restrict__ f1, const float* __restrict__ f2, float* __restrict__ out,
                                   int B, int C, int H, int W, int P, int dil, float scale, float slope, int out_ld, int in_nhwc)
{
    int64_t total = (int64_t)B * P * P * H * W;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    int x = t % W;
    int64_t r = t / W;
    int y = r % H; r /= H;
    int j = r % P; r /= P;
    int i = r % P;
    int b = (int)(r / P);
    int y2 = y + (i - P / 2) * dil, x2 = x + (j - P / 2) * dil;
    float acc = 0.0f;
    if (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) {
        const int64_t HW = (int64_t)H * W;
        const int64_t cs = in_nhwc ? 1 : HW, ps = in_nhwc ? C : 1;     // channel / pixel strides of the inputs
        const float* p1 = f1 + (int64_t)b * C * HW + ((int64_t)y * W + x) * ps;
        const float* p2 = f2 + (int64_t)b * C * HW + ((int64_t)y2 * W + x2) * ps;
        for (int c = 0; c < C; ++c) acc = fmaf(p1[c * cs], p2[c * cs], acc);
    }
    acc *= scale;
    const float v = acc < 0.0f ? acc * slope : acc;
    if (out_ld == 0) out[t] = v;
    else out[(((int64_t)b * H + y) * W + x) * out_ld + i * P + j] = v;
}

// ---------------------------------------------------------------------------------------- RoIAlign
__device__ __forceinline__ float bilinear_roi(const float* __restrict__ im, int H, int W, float y, float x)
{
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.0f;
    if (y <= 0.0f) y = 0.0f;
    if (x <= 0.0f) x = 0.0f;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
    if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
    float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.0f - ly, hx = 1.0f - lx;
    return hy * hx * im[y_low * W + x_low] + hy * lx * im[y_low * W + x_high] + ly * hx * im[y_high * W + x_low] +
           ly * lx * im[y_high * W + x_high];
}

// one thread per output element; consecutive threads walk (px, py, c) so a wave shares one RoI and touches
// neighbouring channels of one small feature window (L1/L2 resident: 633 x 24 x 40 floats = 2.4 MB)
__global__ void roi_align_avg_kernel(const float* __restrict__ feat, const float* __restrict__ rois,
                                     float* __restrict__ out, int C, int H, int W, int n, int PH, int PW, float scale,
                                     int sampling_ratio, int aligned, int xcd)
{
    int64_t total = (int64_t)n * C * PH * PW;
    // RoIs arrive sorted by image: XCD-contiguous block order keeps an image's feature maps in one L2
    const int64_t blk = xcd ? stm_xcd_block((total + blockDim.x - 1) / blockDim.x) : (int64_t)blockIdx.x;
    if (blk < 0) return;
    int64_t t = blk * blockDim.x + threadIdx.x;
    if (t >= total) return;
    int px = t % PW;
    int64_t r = t / PW;
    int py = r % PH; r /= PH;
    int c = r % C;
    int ri = (int)(r / C);
    const float* roi = rois + 5 * ri;
    int b = (int)roi[0];
    float offset = aligned ? 0.5f : 0.0f;
    float sw_ = roi[1] * scale - offset, sh_ = roi[2] * scale - offset;
    float ew_ = roi[3] * scale - offset, eh_ = roi[4] * scale - offset;
    float rw = ew_ - sw_, rh = eh_ - sh_;
    if (!aligned) { rw = fmaxf(rw, 1.0f); rh = fmaxf(rh, 1.0f); }
    float bh = rh / (float)PH, bw = rw / (float)PW;
    int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
    int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
    float count = (float)max(gh * gw, 1);
    const float* im = feat + ((int64_t)b * C + c) * H * W;
    float acc = 0.0f;
    for (int iy = 0; iy < gh; ++iy) {
        float y = sh_ + (float)py * bh + ((float)iy + 0.5f) * bh / (float)gh;
        for (int ix = 0; ix < gw; ++ix) {
            float x = sw_ + (float)px * bw + ((float)ix + 0.5f) * bw / (float)gw;
            acc += bilinear_roi(im, H, W, y, x);
        }
    }
    out[t] = acc / count;
}

}  // namespace

namespace {
int corr_patch_launch(const float* f1, const float* f2, float* out, int B, int C, int H, int W, int P, int dil, float scale, float leaky_slope,
                      int out_ld, int in_nhwc, stm_stream_t stream)
{
    STM_REQUIRE(f1 && f2 && out, STM_ENULL, "stm_corr_patch_f32: f1/f2/out must be non-NULL");
    STM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, STM_EINVAL, "stm_corr_patch_f32: empty input");
    STM_REQUIRE(P > 0 && (P & 1) && dil > 0, STM_EINVAL, "stm_corr_patch_f32: patch_size must be odd, dilation > 0");
    const int force = STM_ENV_INT("STM_CORR_VARIANT", 0);        // 1: the generic kernel (tests); 2: round 3's 2-waves-per-SIMD form of the tiled kernel (A/B)
    // tiled kernel: P = 11, rows of whole float4 quads, staging plan of <= 8 units per thread (W <= 44)
    // tiled kernel: P = 11, rows of whole float4 quads, staging plan of <= 8 units per thread: 16-channel chunks up to W = 44, 8-channel up to 88
    const int cck = 16 * 11 * (W / 4) <= 8 * 256 ? 16 : 8;
    bool tiled = (P == 11 && dil == 1 && H <= 65535 && B <= 65535 && W % 4 == 0 && cck * 11 * (W / 4) <= 8 * 256 &&
                  ((uintptr_t)out % 16 == 0) && ((uintptr_t)f1 % 16 == 0) && ((uintptr_t)f2 % 16 == 0) &&
                  (int64_t)C * H * W < ((int64_t)1 << 20));      // (the packed staging plan: float4 offsets of one image below 2^18)
    if (force == 1 || (in_nhwc && C % 4 != 0)) tiled = false;
    if (tiled) {
        int LW2 = ((W + 10 + 3) / 4) * 4;
        size_t lds = (size_t)(cck * W + cck * 11 * LW2) * sizeof(float);
        if (lds <= 64 * 1024) {
            const dim3 grid(stm_xcd_grid((int64_t)H * B));
            if (in_nhwc && cck == 16 && force != 2 && 16 * 11 * (W / 4) <= 7 * 256)
                hipLaunchKernelGGL((corr_patch_tiled<11, true, 16, 7, 3>), grid, dim3(256), lds, stm_hs(stream), f1, f2, out, C, H, W, scale, leaky_slope, B, out_ld);
            else if (in_nhwc && cck == 16)
                hipLaunchKernelGGL((corr_patch_tiled<11, true, 16>), grid, dim3(256), lds, stm_hs(stream), f1, f2, out, C, H, W, scale, leaky_slope, B, out_ld);
            else if (in_nhwc)
                hipLaunchKernelGGL((corr_patch_tiled<11, true, 8>), grid, dim3(256), lds, stm_hs(stream), f1, f2, out, C, H, W, scale, leaky_slope, B, out_ld);
            else if (cck == 16)
                hipLaunchKernelGGL((corr_patch_tiled<11, false, 16>), grid, dim3(256), lds, stm_hs(stream), f1, f2, out, C, H, W, scale, leaky_slope, B, out_ld);
            else
                hipLaunchKernelGGL((corr_patch_tiled<11, false, 8>), grid, dim3(256), lds, stm_hs(stream), f1, f2, out, C, H, W, scale, leaky_slope, B, out_ld);
            STM_CHECK_LAUNCH("corr_patch_tiled");
            return STM_OK;
        }
    }
    int64_t total = (int64_t)B * P * P * H * W;
    hipLaunchKernelGGL(corr_patch_generic, dim3(stm_cdiv(total, 256)), dim3(256), 0, stm_hs(stream), f1, f2, out, B, C, H,
                       W, P, dil, scale, leaky_slope, out_ld, in_nhwc);
    STM_CHECK_LAUNCH("corr_patch_generic");
    return STM_OK;
}
}  // namespace

extern "C" int stm_corr_patch_f32(const float* f1, const float* f2, float* out, int B, int C, int H, int W, int P,
                                  int dil, float scale, float leaky_slope, stm_stream_t stream)
{
    return corr_patch_launch(f1, f2, out, B, C, H, W, P, dil, scale, leaky_slope, 0, 0, stream);
}

extern "C" int stm_corr_patch_nhwc_f32(const float* f1, const float* f2, float* out, int B, int C, int H, int W, int P, int dil, float scale,
                                       float leaky_slope, int out_ld, int in_nhwc, stm_stream_t stream)
{
    STM_REQUIRE(out_ld >= P * P, STM_EINVAL, "stm_corr_patch_nhwc_f32: out_ld (%d) must hold the %d displacement channels", out_ld, P * P);
    return corr_patch_launch(f1, f2, out, B, C, H, W, P, dil, scale, leaky_slope, out_ld, in_nhwc ? 1 : 0, stream);
}

extern "C" int stm_roi_align_avg_f32(const float* feat, const float* rois, float* out, int B, int C, int H, int W, int n,
                                     int PH, int PW, float spatial_scale, int sampling_ratio, int aligned,
                                     stm_stream_t stream)
{
    STM_REQUIRE(n >= 0, STM_EINVAL, "stm_roi_align_avg_f32: n=%d", n);
    if (n == 0) return STM_OK;
    STM_REQUIRE(feat && rois && out, STM_ENULL, "stm_roi_align_avg_f32: feat/rois/out must be non-NULL");
    STM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && PH > 0 && PW > 0, STM_EINVAL, "stm_roi_align_avg_f32: bad sizes");
    int64_t total = (int64_t)n * C * PH * PW;
    const int xcd = 1;        // XCD-contiguous block order (a switch until round 6)
    hipLaunchKernelGGL(roi_align_avg_kernel, dim3(xcd ? stm_xcd_grid(stm_cdiv(total, 256)) : stm_cdiv(total, 256)), dim3(256), 0, stm_hs(stream),
                       feat, rois, out, C, H, W, n, PH, PW, spatial_scale, sampling_ratio, aligned, xcd);
    STM_CHECK_LAUNCH("roi_align_avg_kernel");
    return STM_OK;
}

// ---- TemporalNet's tail: AvgPool2d((7, 7)) + fc + fc_coeff (track_to_segment_head.py:33-37) -------------------------------------------------
// Input: the pooled SUMS stm_conv2d_planar_windows_pool_f32 accumulated (pool_fix[n][C], unsigned 32.32 fixed point).  One workgroup per eight RoIs:
// mean = sum / npix (one rounding, in double), kept in LDS; output o = bias[o] + sum_k mean[k] w[o][k]: a lane adds its k = lane, lane + 64, ... in
// order, the 64 lane sums are folded row by row with DPP moves and the four row totals added -- a fixed order, the same on every run.  With `clear` the consumed sums are zeroed for the
// next step (no separate memset launch).
namespace {
constexpr int TPF_R = 8;      // RoIs per workgroup: a row of the stacked weight matrix is read once per 8 RoIs
__global__ __launch_bounds__(256) void temporal_pool_fc_kernel(unsigned long long* __restrict__ pool_fix, int n, int C, double inv, const float* __restrict__ w,
                                                              const float* __restrict__ bias, int n_out, int n_first, float* __restrict__ out,
                                                              float* __restrict__ out2, float* __restrict__ pooled_out, int clear)
{
    extern __shared__ float mean[];                       // [TPF_R][C]
    const int b0 = blockIdx.x * TPF_R, nr = min(TPF_R, n - b0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long* rows = pool_fix + (size_t)b0 * C;
    // (the RoIs' rows are contiguous: one linear sweep, eight loads in flight per thread -- with the clearing store between two loads of the plain loop
    // the compiler kept one load in flight and the sweep took most of the kernel's 59 us)
    const int total = nr * C;
    for (int i0 = tid; i0 < total; i0 += 8 * 256) {
        unsigned long long x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = i0 + u * 256 < total ? rows[i0 + u * 256] : 0ull;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = i0 + u * 256;
            if (idx < total) {
                const float m = (float)((double)x[u] * inv);
                mean[idx] = m;
                if (pooled_out) pooled_out[(size_t)b0 * C + idx] = m;
                if (clear) rows[idx] = 0ull;
            }
        }
    }
    for (int idx = total + tid; idx < TPF_R * C; idx += 256) mean[idx] = 0.0f;
    __syncthreads();
    for (int o = wave; o < n_out; o += 4) {
        const float* wr = w + (size_t)o * C;
        float s[TPF_R];
#pragma unroll
        for (int r = 0; r < TPF_R; ++r) s[r] = 0.0f;
        for (int k0 = lane; k0 < C; k0 += 64 * 8) {                   // eight weight loads in flight per lane (same order of the sums as a plain loop)
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = k0 + 64 * u < C ? wr[k0 + 64 * u] : 0.0f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + 64 * u;
                if (k < C) {
#pragma unroll
                    for (int r = 0; r < TPF_R; ++r) s[r] = __builtin_fmaf(mean[r * C + k], wv[u], s[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < TPF_R; ++r) s[r] = stm_wave_sum(s[r]);        // (DPP row sums + four row totals: no LDS round trips)
        if (lane == 0) {
            const float bo = bias ? bias[o] : 0.0f;
#pragma unroll
            for (int r = 0; r < TPF_R; ++r) {
                if (r >= nr) break;
                const float v = s[r] + bo;
                if (o < n_first) out[(size_t)(b0 + r) * n_first + o] = v;
                else out2[(size_t)(b0 + r) * (n_out - n_first) + (o - n_first)] = v;
            }
        }
    }
}
}  // namespace

extern "C" int stm_temporal_pool_fc_f32(unsigned long long* pool_fix, int n, int C, int npix, const float* weight, const float* bias, int n_out,
                                        int n_first, float* out, float* out2, float* pooled_out, int clear, stm_stream_t stream)
{
    const char* who = "stm_temporal_pool_fc_f32";
    STM_REQUIRE(n >= 0, STM_EINVAL, "%s: n=%d", who, n);
    if (n == 0) return STM_OK;
    STM_REQUIRE(pool_fix && weight && out, STM_ENULL, "%s: pool_fix / weight / out must be non-NULL", who);
    STM_REQUIRE(C > 0 && C <= 2048 && npix > 0 && n_out > 0, STM_EINVAL, "%s: C (%d, at most 2048), npix (%d), n_out (%d)", who, C, npix, n_out);
    if (!out2) n_first = n_out;
    STM_REQUIRE(n_first >= 0 && n_first <= n_out, STM_EINVAL, "%s: n_first (%d) must lie in [0, n_out]", who, n_first);
    hipLaunchKernelGGL(temporal_pool_fc_kernel, dim3(stm_cdiv(n, TPF_R)), dim3(256), (size_t)TPF_R * C * sizeof(float), stm_hs(stream), pool_fix, n, C,
                       1.0 / (4294967296.0 * (double)npix), weight, bias, n_out, n_first, out, out2, pooled_out, clear);
    STM_CHECK_LAUNCH("temporal_pool_fc_kernel");
    return STM_OK;
}
