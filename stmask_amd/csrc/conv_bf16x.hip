// conv_bf16x.hip -- dense 2-D convolution as an implicit GEMM on the 16-bit matrix cores with fp32-equivalent accuracy: every
// fp32 operand travels as PLANES whose sum is the value --
//   format 0: three bf16 planes (x = p0 + p1 + p2, each the round-to-nearest of the running residual), six plane products
//             x*w ~= x0 w0 + x0 w1 + x1 w0 + x1 w1 + x0 w2 + x2 w0 (dropped terms <= 2^-24 |x w|), any fp32 range;
//   format 1: two fp16 planes, x = h + l / 2048 (22 significand bits for 6.1e-5 <= |x| <= 65504), three products
//             h h + (h l + l h) / 2048 -- half the matrix work at the same tested error (2e-6 of sum |x w| against fp64);
//   format 2: ONE fp16 plane, one product: genuine fp16 convolution with fp32 accumulation (BASELINE config 5; 1e-3 of sum |x w|)
// -- accumulated in fp32 by v_mfma_f32_16x16x32_{bf16,f16}.  The fp32 matrix peak of the chip is 157 TFLOP/s, the 16-bit peak 2.5 P.
//
// This is the reference's nn.Conv2d on the hot path -- Bottleneck 1x1/3x3 (backbone.py:38-58), the GEMM of dcn_v2.DCN
// (backbone.py:20-26,45), FPN (FPN.py:68-108), proto-net (make_net.py:5-59), the PredictionModule_FC tower
// (prediction_head_FC.py:146-195) and TemporalNet (track_to_segment_head.py:10-37), SURVEY.md section 8 rows a1-a5 / a10 / f4 --
// with the eval-mode BatchNorm folded into the weights, and bias, residual add and ReLU fused into the epilogue.
//
// Activations stay split BETWEEN layers ([planes][C/32][pixels][32], include/stmask_hip.h): each element is split once, by its
// producer's epilogue, and staging a K-slab (one tap, 32 channels) is pure LDS-DMA.  Tiles: 128*MG pixels x 64*NJ channels, 4*MG
// waves; LDS rows of 64 B with the 16-B chunk index XORed by f((row >> 2) & 3), conflict-free for the 16x16x32 fragment reads;
// two LDS buffers, or a three-buffer ring with fragment prefetch (fp16 formats); split-K for grids that would idle the chip.
// DESIGN.md section 4 has the history of this kernel and what bounds it.
#include "planar_common.h"
#include <algorithm>
#include <type_traits>
#include <atomic>
#include <mutex>
#include <unordered_map>

namespace {

// Write 8 consecutive channels of one pixel into a planar tensor: dst = address of the 16-byte group in plane 0, plane_b = bytes
// between planes.  fmt 0: three bf16 planes; 1: two fp16 planes (h, (x - h) * 2048); 2: ONE fp16 plane (h only -- the genuine
// fp16 activation format of BASELINE config 5; plane 0 of a fmt-1 tensor is a valid fmt-2 tensor).  nt: nontemporal stores.
__device__ __forceinline__ void store_planes8(uint8_t* dst, size_t plane_b, const float (&v)[8], int fmt, int* range_flag, bool nt)
{
    unsigned q0[4], q1[4], q2[4];
    if (fmt == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) split2(f32x2{v[2 * e], v[2 * e + 1]}, q0[e], q1[e], q2[e]);
    } else {
        f16_range_check8(v, range_flag);
#pragma unroll
        for (int e = 0; e < 4; ++e) split2_f16(f32x2{v[2 * e], v[2 * e + 1]}, q0[e], q1[e]);
    }
    const int np = fmt == 0 ? 3 : (fmt == 1 ? 2 : 1);
    if (nt) {
        __builtin_nontemporal_store(u32x4{q0[0], q0[1], q0[2], q0[3]}, reinterpret_cast<u32x4*>(dst));
        if (np > 1) __builtin_nontemporal_store(u32x4{q1[0], q1[1], q1[2], q1[3]}, reinterpret_cast<u32x4*>(dst + plane_b));
        if (np > 2) __builtin_nontemporal_store(u32x4{q2[0], q2[1], q2[2], q2[3]}, reinterpret_cast<u32x4*>(dst + 2 * plane_b));
    } else {
        *reinterpret_cast<u32x4*>(dst) = u32x4{q0[0], q0[1], q0[2], q0[3]};
        if (np > 1) *reinterpret_cast<u32x4*>(dst + plane_b) = u32x4{q1[0], q1[1], q1[2], q1[3]};
        if (np > 2) *reinterpret_cast<u32x4*>(dst + 2 * plane_b) = u32x4{q2[0], q2[1], q2[2], q2[3]};
    }
}

// the sticky fp16-range flag, one per device (a process that drives several GPUs registers one flag on each)
constexpr int STM_MAX_DEVICES = 32;
static int* g_range_flags[STM_MAX_DEVICES] = {};
static int* current_range_flag()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= STM_MAX_DEVICES) return nullptr;
    return g_range_flags[dev];
}

// ---- planar variant: the activation arrives ALREADY split, as NPL bf16 planes [NPL][B*H*W][x_ld] (the format the
// epilogue below writes), so staging a K-slab is pure data movement: LDS-DMA (buffer_load ... lds, 16 B per lane,
// out-of-range offsets deliver the zero padding) for the activation rows and global_load_lds for the pre-tiled weights,
// no registers and no VALU beyond a handful of address operations.  Each element is split once, where it is produced,
// instead of kh*kw*n_tiles times in the consumers' loaders -- measured on the register-staged kernels above, the
// split's VALU stream crawls (2.8x slower) whenever the SIMD's other wave is issuing MFMAs, and that, not the matrix
// pipe, set their speed.  Two LDS buffers; slab s+1 streams in while the MFMAs run on slab s; one barrier per slab.
// Workgroup = 4*MG waves as (2*MG) x 2, tile = 128*MG pixels x 128 channels.
struct PlanarArgs {
    const uint8_t* xp;      // [NPL][C/32][x_np][32] bf16 (channel-slab major: a pixel's 32-channel slab is one 64-B line)
    const uint8_t* wp;
    const float* bias;
    const float* res_f32;   // [M][res_ld] or null
    const uint8_t* res_pl;  // [3][Cout/32][res_np][32] bf16 or null
    float* out_f32;         // [M][out_ld] or null
    uint8_t* out_pl;        // [3][Cout/32][out_np][32] bf16 or null
    int B, H, W, C, Ho, Wo, Cout;
    int kh, kw, sh, sw, ph, pw;
    int x_ld, out_ld, res_ld;                        // fp32 tensors only (pixels x ld)
    int x_np, out_np, res_np;                        // pixels per channel slab of the planar buffers
    int relu;
    int M, n_tiles, m_tiles, slabs;
    int nsub;                // 0, or the channel tiles of one pixel tile that run TOGETHER on an XCD (see the tile map)
    unsigned plane_bytes;   // bytes of one input plane that may be addressed (buffer range)
    long long x_pstride, out_pstride, res_pstride;   // bytes between planes
    int groups, ntpg, cout_g;                        // grouped conv: n-tiles per group, output channels per group
    int group_real[8];                               // output channels per group that are not zero padding (MFMA tiles past them are skipped)
    int n_levels;                                    // > 0: pixels are the concatenation of n_levels images sizes
    int lvl_start[9], lvl_h[8], lvl_w[8];
    int pointwise;            // 1x1 / stride 1 / no padding / one image size: input pixel == output pixel, no coordinate arithmetic
    float inv_hw, inv_w;      // 1 / (Ho * Wo), 1 / Wo for the pixel -> (image, row, column) split of single-size launches
    int fmt;                  // input / residual planes: 0 three bf16 planes (six products), 1 two fp16 planes (three), 2 one fp16 plane (one)
    int out_fmt;              // format of out_pl (normally fmt; a fmt-2 layer may write fmt 1 for a consumer that wants both planes)
    float out_scale;          // 1 / (power-of-two weight scale of the packed image)
    int* range_flag;          // fmt 1: set to 1 when an output has no fp16 representation (see f16_range_check8); may be null
    int nt_out;               // fmt 1: nontemporal plane stores (outputs far larger than L2)
    int splitk, kslabs, ldp;  // split-K: K-slabs per split, fp32 partial sums [splitk][M][ldp] in `partial`
    float* partial;
    int vec_epilogue;       // Cout, out_ld, res_ld multiples of 8 and 16-byte aligned pointers: vector epilogue
    // two-source 1x1 convolution (DUAL instantiations): K-slabs [0, c1_slabs) come from xp (one pixel per output pixel), the others from
    // xp2, a second planar tensor of B images of H2 x W2 read at stride s2 -- conv3 of a ResNet stage's first bottleneck and its
    // projection shortcut as ONE product over the concatenated channels, W = [W3 | Wds]
    const uint8_t* xp2;
    long long x2_pstride;
    unsigned plane2_bytes;
    int c1_slabs, x2_np, H2, W2, s2;
    // window launches (stm_conv_geom.win_w > 0): the launch computes a Ho x Wo window of every output image; output pixel (b, oy, ox) is row
    // b * win_hw + oy * win_w + ox + win_off of the output tensors
    int win_w, win_hw, win_off;
};
// several window launches in ONE grid (CLS instantiation; stm_conv2d_planar_windows_f32): class c = tiles [tile0, next tile0) with its own
// weights, sub-kernel, window and pixel count -- they replace wp / kh / kw / ph / pw / Ho / Wo / M / slabs / win_off per tile.  (A type of
// its own: half a KB more of kernel arguments on EVERY convolution launch cost the single-stream step 0.1 ms.)
struct PlanarArgsCls : PlanarArgs {
    int n_cls, cls_tiles;
    // pooled output (stm_conv2d_planar_windows_pool_f32): instead of writing its pixels the launch adds relu(conv + bias) of every output image's pixels
    // into pool[image][channel] as unsigned 32.32 fixed point -- integer sums, so the order in which the classes' workgroups arrive does not matter
    unsigned long long* pool;
    int pool_ld;
    struct Cls { const uint8_t* wp; int kh, kw, ph, pw, Ho, Wo, M, slabs, win_off, tile0; float inv_hw, inv_w; } cls[9];
};

// Epilogue shared by the planar kernels: bias + residual (+ReLU) in fp32, then fp32 NHWC and/or the three bf16 planes.
// Each wave first parks its 64 x (32*NJ) accumulator tile in LDS (free now) -- park32 / park16 know the C/D register layout
// of the MFMA shape used -- and the tail re-reads it pixel-major, 8 consecutive channels per lane, so that (Cout and the
// leading dimensions multiples of 8) every global access is a 16-byte vector: 24 stores per thread for the three planes
// instead of 192 two-byte ones.  Otherwise the same 8-channel segments are written element by element, guarded.
template <int NJ>
__device__ __forceinline__ float* park_base(uint8_t* smem, int wave) { return reinterpret_cast<float*>(smem) + wave * (64 * (32 * NJ + 4)); }

template <int NJ>   // v_mfma_f32_16x16x32: col = lane & 15, row = 4 (lane >> 4) + r
__device__ __forceinline__ void park16(f32x4 (&acc)[4][2 * NJ], f32x4 (&accl)[4][2 * NJ], uint8_t* smem, int wave, int lane, float ls = 1.0f)
{
    constexpr int EP_LD = 32 * NJ + 4;
    float* park = park_base<NJ>(smem, wave);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2 * NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                park[(i * 16 + 4 * (lane >> 4) + r) * EP_LD + j * 16 + (lane & 15)] = acc[i][j][r] + accl[i][j][r] * ls;
}

// bias + residual + ReLU + stores of one 8-channel segment (pixel m, channels co .. co+7, nvalid of them real); v = raw sums
// SCALE_BIAS false: the caller has applied out_scale and the bias already.  FMT >= 0: the plane format is known at compile time (the
// convolution kernels: it follows from their template parameters) and the branches on it fold away; -1: read it from the arguments.
template <bool SCALE_BIAS = true, int FMT = -1>
__device__ __forceinline__ void epilogue_store8(const PlanarArgs& a, int m, int co, int nvalid, float (&v)[8], bool have_pre = false,
                                                f16x8 pre0 = f16x8{}, f16x8 pre1 = f16x8{})   // pre0 / pre1: this segment's fp16 residual planes, already loaded
{
    const int fmt = FMT >= 0 ? FMT : a.fmt;
    const int ofmt = (FMT == 0 || FMT == 1) ? FMT : a.out_fmt;      // only a format-2 layer may write another format (1)
    const size_t opl = (size_t)(a.out_pstride >> 1), rpl = (size_t)(a.res_pstride >> 1);   // plane strides in elements
    // element index of (pixel m, channel co) in a slab-major planar buffer with np pixels per slab
    auto pidx = [](int m_, int co_, int np) { return ((size_t)(co_ >> 5) * np + m_) * 32 + (co_ & 31); };
    int mo = m;                                          // output row (differs from the pixel index of the launch in window launches)
    if (a.win_w) {
        const int hw = a.Ho * a.Wo;
        int b = (int)((float)m * a.inv_hw);
        int rem = m - b * hw;
        if (rem < 0) { --b; rem += hw; } else if (rem >= hw) { ++b; rem -= hw; }
        int oy = (int)((float)rem * a.inv_w);
        const int t = rem - oy * a.Wo;
        if (t < 0) --oy; else if (t >= a.Wo) ++oy;
        mo = b * a.win_hw + oy * a.win_w + (rem - oy * a.Wo) + a.win_off;
    }
    if constexpr (SCALE_BIAS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e], a.out_scale, (a.bias && e < nvalid) ? a.bias[co + e] : 0.0f);
    }
    if (a.vec_epilogue) {                              // implies nvalid == 8
        if (a.res_f32) {
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(a.res_f32 + (size_t)m * a.res_ld + co);
            const f32x4 r1 = *reinterpret_cast<const f32x4*>(a.res_f32 + (size_t)m * a.res_ld + co + 4);
            v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
        }
        if (a.res_pl) {
            const size_t ri = pidx(m, co, a.res_np) * 2;
            if (fmt == 2) {
                const f16x8 p0 = have_pre ? pre0 : *reinterpret_cast<const f16x8*>(a.res_pl + ri);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)p0[e];
            } else if (fmt == 1) {
                const f16x8 p0 = have_pre ? pre0 : *reinterpret_cast<const f16x8*>(a.res_pl + ri);
                const f16x8 p1 = have_pre ? pre1 : *reinterpret_cast<const f16x8*>(a.res_pl + ri + rpl * 2);
                // l / 2048 is exact, so fma(l, 1/2048, h) is the same single rounding as h + l * (1/2048) (v_fma_mix_f32: no converts)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += __builtin_fmaf((float)p1[e], 1.0f / STM_F16_LOW_SCALE, (float)p0[e]);
            } else {
                const bf16x8 p0 = *reinterpret_cast<const bf16x8*>(a.res_pl + ri);
                const bf16x8 p1 = *reinterpret_cast<const bf16x8*>(a.res_pl + ri + rpl * 2);
                const bf16x8 p2 = *reinterpret_cast<const bf16x8*>(a.res_pl + ri + rpl * 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += ((float)p0[e] + (float)p1[e]) + (float)p2[e];
            }
        }
        if (a.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaxf(v[e], 0.0f);   // (nan -> 0 and -0 -> +0, as `v > 0 ? v : 0` gives)
        }
        if (a.out_f32) {
            float* o = a.out_f32 + (size_t)mo * a.out_ld + co;
            *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
        if (a.out_pl) {
            uint8_t* o = a.out_pl + pidx(mo, co, a.out_np) * 2;
            store_planes8(o, opl * 2, v, ofmt, a.range_flag, a.nt_out != 0);
        }
        return;
    }
    __bf16* outp = reinterpret_cast<__bf16*>(a.out_pl);
    const __bf16* resp = reinterpret_cast<const __bf16*>(a.res_pl);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (e >= nvalid) break;
        float x = v[e];
        if (a.res_f32) x += a.res_f32[(size_t)m * a.res_ld + co + e];
        if (resp) {
            const size_t ri = pidx(m, co + e, a.res_np);
            if (fmt >= 1) {
                const _Float16* rh = reinterpret_cast<const _Float16*>(a.res_pl);
                x += (float)rh[ri];
                if (fmt == 1) x += (float)rh[ri + rpl] * (1.0f / STM_F16_LOW_SCALE);
            } else {
                x += ((float)resp[ri] + (float)resp[ri + rpl]) + (float)resp[ri + 2 * rpl];
            }
        }
        if (a.relu) x = x > 0.0f ? x : 0.0f;
        if (a.out_f32) a.out_f32[(size_t)mo * a.out_ld + co + e] = x;
        if (outp) {
            const size_t oi = pidx(mo, co + e, a.out_np);
            if (ofmt >= 1) {
                _Float16* oh = reinterpret_cast<_Float16*>(a.out_pl);
                const _Float16 h = (_Float16)x;
                oh[oi] = h;
                if (ofmt == 1) oh[oi + opl] = (_Float16)((x - (float)h) * STM_F16_LOW_SCALE);
                if (!(fabsf(x) <= 65504.0f) && a.range_flag) *reinterpret_cast<volatile int*>(a.range_flag) = 1;
            } else {
                const __bf16 h = (__bf16)x;
                const float r1 = x - (float)h;
                const __bf16 mid = (__bf16)r1;
                outp[oi] = h;
                outp[oi + opl] = mid;
                outp[oi + 2 * opl] = (__bf16)(r1 - (float)mid);
            }
        }
    }
}

template <int NJ, int FMT = -1>
__device__ __forceinline__ void planar_epilogue_tail(const PlanarArgs& a, uint8_t* smem, int wave, int lane, int m0, int n0g, int grp,
                                                     int wm, int wn)
{
    const int fmt = FMT >= 0 ? FMT : a.fmt;
    constexpr int EP_LD = 32 * NJ + 4;                 // floats per parked pixel row (+ 4 pad)
    constexpr int LPR = 4 * NJ;                        // lanes per pixel row (8 channels each)
    const float* park = park_base<NJ>(smem, wave);
    // same wave reads back what it wrote: no workgroup barrier needed, only the LDS counter
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int seg = lane % LPR, prow = lane / LPR;
    const int cog = n0g + wn * (32 * NJ) + seg * 8;    // channel within the group
    const int co = grp * a.cout_g + cog;
    const int nvalid = min(8, a.cout_g - cog);         // channels of this segment that exist (<= 0: none)
    if (nvalid <= 0) return;
    // fp16 residual planes: all passes' loads first.  Taken pass by pass they sit behind the previous pass's stores (the
    // compiler must assume the output aliases the residual), one 32-byte load per lane in flight -- the HBM-bound layers
    // (expanding 1x1 convolutions of the bottlenecks) then run at the latency of four dependent round trips.  The fragment
    // and accumulator registers are dead here, so the 8 * LPR registers cost nothing.
    // (64-channel tiles only: with LPR = 8 the 64 extra registers push the 256 x 128 kernel into scratch -- 380 -> 1065 us)
    // bias of this lane's 8 channels: loaded once, not once per pass and element
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = 0.0f;
    if (a.bias) {
        if (nvalid == 8 && a.vec_epilogue) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + co), b1 = *reinterpret_cast<const f32x4*>(a.bias + co + 4);
            bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w; bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (e < nvalid) bias8[e] = a.bias[co + e];
        }
    }
    const bool pre = NJ == 1 && a.vec_epilogue && a.res_pl != nullptr && fmt >= 1;
    f16x8 r0[LPR], r1[LPR];
    if (pre) {
#pragma unroll
        for (int pass = 0; pass < LPR; ++pass) {
            const int m = min(m0 + wm * 64 + pass * (64 / LPR) + prow, a.M - 1);
            const size_t ri = (((size_t)(co >> 5) * a.res_np + m) * 32 + (co & 31)) * 2;
            r0[pass] = *reinterpret_cast<const f16x8*>(a.res_pl + ri);
            r1[pass] = fmt == 1 ? *reinterpret_cast<const f16x8*>(a.res_pl + ri + a.res_pstride) : r0[pass];   // fmt 2: one plane
        }
    }
#pragma unroll
    for (int pass = 0; pass < LPR; ++pass) {
        const int pr = pass * (64 / LPR) + prow;
        const int m = m0 + wm * 64 + pr;
        if (m >= a.M) continue;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(park + pr * EP_LD + seg * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(park + pr * EP_LD + seg * 8 + 4);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        // out_scale is a power of two: v * out_scale is exact, so the fused form rounds exactly like multiply-then-add (one
        // instruction per element instead of two under -ffp-contract=off)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e], a.out_scale, bias8[e]);
        if (pre) epilogue_store8<false, FMT>(a, m, co, nvalid, v, true, r0[pass], r1[pass]);
        else epilogue_store8<false, FMT>(a, m, co, nvalid, v);
    }
}

// split-K: add the partial sums of one (pixel, 8-channel segment) and run the epilogue; thread per segment
__global__ __launch_bounds__(256) void planar_splitk_finish_kernel(const PlanarArgs a, int bn)
{
    const int segs_g = (a.cout_g + 7) >> 3;                            // segments per group
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)a.M * a.groups * segs_g;
    if (idx >= total) return;
    const int sg = (int)(idx % (a.groups * segs_g));
    const int m = (int)(idx / (a.groups * segs_g));
    const int grp = sg / segs_g, cog = (sg - grp * segs_g) * 8;
    const int col = grp * a.ntpg * bn + cog;                           // column in the tile-padded partial matrix
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // four parts' loads in flight at a time (one dependent round trip per part made this kernel 6 us on 15 000 threads: the
    // single-stream step has 47 of them); the parts are still added in part order
    const float* p = a.partial + (size_t)m * a.ldp + col;
    const size_t pstride = (size_t)a.M * a.ldp;
    for (int k0 = 0; k0 < a.splitk; k0 += 4) {
        f32x4 q0[4], q1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* pk = p + (size_t)min(k0 + j, a.splitk - 1) * pstride;
            q0[j] = *reinterpret_cast<const f32x4*>(pk);
            q1[j] = *reinterpret_cast<const f32x4*>(pk + 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (k0 + j < a.splitk) {
                v[0] += q0[j].x; v[1] += q0[j].y; v[2] += q0[j].z; v[3] += q0[j].w;
                v[4] += q1[j].x; v[5] += q1[j].y; v[6] += q1[j].z; v[7] += q1[j].w;
            }
        }
    }
    epilogue_store8(a, m, grp * a.cout_g + cog, min(8, a.cout_g - cog), v);
}

// NPL planes per operand; tile = 128*MG pixels x 64*NJ channels; DT = 1: fp16 planes (v_mfma_f32_16x16x32_f16), 0: bf16 planes
// (v_mfma_f32_16x16x32_bf16 -- the 16x16x32 shape spends less energy per flop than 32x32x16 and the chip is power-limited here:
// 593 vs 656 us on the 145-GF proto layer); ST = 3: three-buffer LDS ring with fragment prefetch (fp16 formats), 2: two buffers.
// ABL (builds with -DSTM_ABLATE only): timing ablations of the ring loop, RESULTS ARE WRONG -- 1 no DMA, 2 no barrier, 4 no
// fragment reads, 8 no wait for the DMAs, 16 activation DMA on every third slab only (the traffic of a kx-reuse staging), 32 no
// activation DMA, 64 no weight DMA.
// AvgPool2d over the whole output image folded into the epilogue of a window-set launch (TemporalNet: track_to_segment_head.py:30-33).  The wave owns 64
// class pixels x 64 channels, parked in LDS: lane = channel, the rows are walked once; the pixels of one image are consecutive rows of a class (hw of
// them), so the running sum is flushed whenever the image index advances, and at the end of the block (an image cut by a block or class boundary arrives
// in several pieces).  Each piece is an fp32 sum in row order, converted exactly to 32.32 fixed point and added with an integer atomic: the total does
// not depend on the order of arrival.  mw = first class pixel of the wave's rows, col = the lane's output channel.
__device__ __forceinline__ void pooled_epilogue(const PlanarArgs& a, unsigned long long* pool, int pool_ld, uint8_t* smem, int wave, int lane, int mw, int col)
{
    constexpr int EP_LD = 32 * 2 + 4;
    const float* park = park_base<2>(smem, wave);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int hw = a.Ho * a.Wo;
    const int rows = min(64, a.M - mw);
    int b = mw / hw, r = mw - b * hw;
    const float bias = a.bias ? a.bias[col] : 0.0f;
    float sum = 0.0f;
    for (int row = 0; row < rows; ++row) {
        sum += __builtin_fmaxf(__builtin_fmaf(park[row * EP_LD + lane], a.out_scale, bias), 0.0f);
        if (++r == hw || row == rows - 1) {
            if (!(sum < 4.0e9f) && a.range_flag) *reinterpret_cast<volatile int*>(a.range_flag) = 1;     // (also NaN)
            const unsigned hi = (unsigned)sum;
            const unsigned lo = (unsigned)((sum - (float)hi) * 4294967296.0f);
            __hip_atomic_fetch_add(pool + (size_t)b * pool_ld + col, ((unsigned long long)hi << 32) | lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sum = 0.0f;
            if (r == hw) { r = 0; ++b; }
        }
    }
}

// (Tried and removed, round 2: loading the residual planes of a 64-channel tile BEFORE the K loop -- no change on the HBM-bound
// expanding 1x1 convolutions, 337 vs 345 us, for 32 more live registers and one wave per SIMD less; and resident workgroups
// walking the tiles instead of one workgroup per tile -- 404 -> 451 us.  Neither the epilogue's second memory round trip nor
// workgroup launch overhead is what holds these layers at 3.3 TB/s.)
template <int NPL, int MG, int NJ, int DT = 0, int ST = 2, int ABL = 0, bool DUAL = false, bool CLS = false>
__global__ __launch_bounds__(512, 1) void conv_planar_kernel(const typename std::conditional<CLS, PlanarArgsCls, PlanarArgs>::type a_in)
{
#if defined(__HIP_DEVICE_COMPILE__)   // the LDS-DMA builtins exist only in the device pass; the host pass needs just the launch stub
    extern __shared__ __align__(16) uint8_t smem[];
    constexpr int BM = CV_BM * MG;
    constexpr int BN = 64 * NJ;                      // output channels per workgroup: waves 2 wide, 32*NJ columns each
    constexpr int WPL = BN * 64;                     // bytes of one weight plane of a slab
    constexpr int XBUF = NPL * BM * 64, WBUF = NPL * WPL, BUF = XBUF + WBUF;
    constexpr int NWAVES = 4 * MG;
    constexpr int WDMA = (BN / 16) * NPL / NWAVES;   // weight DMA instructions (1 KB each) per wave per slab
    static_assert((BN / 16) * NPL % NWAVES == 0, "weight tile must split evenly over the waves");
    static_assert(ST == 2 || (ST == 3 && DT == 1 && NPL <= 2), "the three-buffer ring is built for the fp16 formats");

    int tiles_ = a_in.m_tiles * a_in.n_tiles * a_in.splitk;
    if constexpr (CLS) tiles_ = a_in.cls_tiles;
    const int tiles = tiles_;
    const int per_xcd = (tiles + 7) >> 3;
    int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    // Which tiles share an XCD at one time decides what its 4-MB L2 must fetch: the 32 workgroups an XCD runs together are 32 / n_tiles pixel
    // tiles x all n_tiles channel tiles in the plain order, i.e. every K-slab of EVERY channel tile's weights for only 4 pixel tiles at
    // n_tiles = 8 (rocprofv3: 3.3 GB fetched per TemporalNet launch for 0.45 GB of input -- the 19-MB weight set streams from the Infinity Cache
    // once per 4 pixel tiles).  With nsub (2 or 4) the order inside a group of (32 / nsub) pixel tiles is: nsub channel tiles of each pixel
    // tile, then the next nsub, ... -- 32 / nsub pixel tiles x nsub channel tiles together.  Same tiles, same results.
    auto regroup = [&](int j, int nt_all, int& mt_o, int& nt_o) {
        const int pg = 32 / a_in.nsub, G = pg * nt_all;
        const int gq = j / G, r = j - gq * G;
        const int nh = r >> 5, rr = r & 31;
        mt_o = gq * pg + rr / a_in.nsub;
        nt_o = nh * a_in.nsub + rr % a_in.nsub;
    };
    if constexpr (CLS) {
        // the classes' tiles differ in length (4, 6 or 9 taps): a contiguous run of tile ids per XCD would give one XCD the short classes and
        // another the long one.  Pixel tiles are dealt round-robin to the XCDs instead, each with all its channel tiles (they share its rows).
        const int j = blockIdx.x >> 3;
        int jm = j / a_in.n_tiles, nt_ = j - jm * a_in.n_tiles;
        if (a_in.nsub) regroup(j, a_in.n_tiles, jm, nt_);
        logical = (jm * 8 + (int)(blockIdx.x & 7)) * a_in.n_tiles + nt_;
    } else if (a_in.nsub) {
        int mt_, nt_;
        regroup(logical, a_in.n_tiles, mt_, nt_);
        logical = mt_ * a_in.n_tiles + nt_;        // (a trailing partial group maps past `tiles` only where the plain order would too: see launch)
    }
    if (logical >= tiles) return;
    // CLS: the grid is the concatenation of several window launches (classes); this tile's class supplies the fields that differ
    PlanarArgs a_cls;
    const PlanarArgs* ap = &a_in;
    if constexpr (CLS) {
        int c = 0;
#pragma unroll
        for (int i = 1; i < 9; ++i)
            if (i < a_in.n_cls && logical >= a_in.cls[i].tile0) c = i;
        a_cls = static_cast<const PlanarArgs&>(a_in);
        a_cls.wp = a_in.cls[c].wp;
        a_cls.kh = a_in.cls[c].kh; a_cls.kw = a_in.cls[c].kw; a_cls.ph = a_in.cls[c].ph; a_cls.pw = a_in.cls[c].pw;
        a_cls.Ho = a_in.cls[c].Ho; a_cls.Wo = a_in.cls[c].Wo; a_cls.M = a_in.cls[c].M;
        a_cls.slabs = a_in.cls[c].slabs; a_cls.kslabs = a_in.cls[c].slabs; a_cls.win_off = a_in.cls[c].win_off;
        a_cls.inv_hw = a_in.cls[c].inv_hw; a_cls.inv_w = a_in.cls[c].inv_w;
        logical -= a_in.cls[c].tile0;
        ap = &a_cls;
    }
    const PlanarArgs& a = *ap;
    // (integer divisions cost ~40 instructions each on this ISA and a short-K tile is only a few hundred MFMA cycles long: the
    // common cases -- no split-K, one n-tile, one group -- take none)
    const int ksp = a.splitk == 1 ? 0 : logical % a.splitk;      // split-K part (the parts of one tile sit on one XCD)
    const int tile = a.splitk == 1 ? logical : logical / a.splitk;
    const int mt = a.n_tiles == 1 ? tile : (a.n_tiles == 2 ? tile >> 1 : (a.n_tiles == 4 ? tile >> 2 : tile / a.n_tiles));
    const int nt = tile - mt * a.n_tiles;
    const int m0 = mt * BM;
    if (m0 >= a.M) return;                           // (pixel tiles the regrouped tile map pads the grid with)
    const int s_begin = ksp * a.kslabs, s_end = min(a.slabs, s_begin + a.kslabs);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % (2 * MG), wn = wave / (2 * MG);

    // grouped convolution: n-tile -> group; the group reads its own C input channels and writes its own cout_g outputs
    const int grp = a.groups == 1 ? 0 : nt / a.ntpg;
    const int n0g = (nt - grp * a.ntpg) * BN;           // first output channel of this tile within its group
    const int creal = a.group_real[grp & 7];            // real (not zero-padding) output channels of the group
    // DMA duties of this lane: activation row groups 2*wave and 2*wave+1 (16 rows x 64 B each, all planes).  Per row, computed
    // once: the byte offset of tap (0, 0) within a channel slab (`base`, may be negative), the row pitch in bytes (`wl64`), and
    // one validity bit per tap (`vmask`: the tap's source pixel lies inside the image -- zero padding otherwise).  A K-slab's
    // address is then base + ky * wl64 + kx * 64 + slab offset and its out-of-range flag one bit extract: ~5 VALU per row and
    // slab instead of the coordinate arithmetic (this loop is instruction-bound on the 64-channel tiles and on short K).
    int base[2], wl64[2];
    unsigned vmask[2];
    int base2[2] = {0, 0};                               // DUAL: byte offset of the row's source pixel in the second tensor
    const int slot = lane & 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (2 * wave + i) * 16 + (lane >> 2);
        const int m = m0 + r;
        const bool ok = m < a.M;
        const int mm = ok ? m : 0;
        const int swz16 = (slot ^ swz(r)) << 4;
        if constexpr (DUAL) {
            // output pixel (b, oy, ox) of B x Ho x Wo reads pixel (b, oy * s2, ox * s2) of the second tensor
            const int hw = a.Ho * a.Wo;
            int b = (int)((float)mm * a.inv_hw);
            int rem = mm - b * hw;
            if (rem < 0) { --b; rem += hw; } else if (rem >= hw) { ++b; rem -= hw; }
            int oy = (int)((float)rem * a.inv_w);
            const int t = rem - oy * a.Wo;
            if (t < 0) --oy; else if (t >= a.Wo) ++oy;
            const int ox = rem - oy * a.Wo;
            base2[i] = ((b * a.H2 + oy * a.s2) * a.W2 + ox * a.s2) * 64 + swz16;
        }
        if (a.pointwise) {
            // 1x1 convolution without stride or padding: input pixel == output pixel, one tap, always inside
            base[i] = mm * 64 + swz16;
            wl64[i] = 0;
            vmask[i] = ok ? 1u : 0u;
            continue;
        }
        int H = a.H, W = a.W, Ho = a.Ho, Wo = a.Wo, first = 0, local = mm;
        if (a.n_levels > 0) {
            // the pixel axis concatenates several image sizes (the FPN levels a shared head runs over): find this
            // pixel's level; stride 1 / same padding there, so input and output pixel indices coincide
#pragma unroll
            for (int l = 0; l < 8; ++l)
                if (l < a.n_levels && mm >= a.lvl_start[l]) { first = a.lvl_start[l]; H = a.lvl_h[l]; W = a.lvl_w[l]; }
            Ho = H; Wo = W; local = mm - first;
        }
        int b, rem, oy, ox;
        if (a.n_levels > 0) {
            b = local / (Ho * Wo);
            rem = local - b * (Ho * Wo);
            oy = rem / Wo;
        } else {
            // reciprocal multiply + one correction step instead of two integer divisions (exact: pixel counts < 2^24)
            const int hw = Ho * Wo;
            b = (int)((float)local * a.inv_hw);
            rem = local - b * hw;
            if (rem < 0) { --b; rem += hw; } else if (rem >= hw) { ++b; rem -= hw; }
            oy = (int)((float)rem * a.inv_w);
            int t = rem - oy * Wo;
            if (t < 0) --oy; else if (t >= Wo) ++oy;
        }
        ox = rem - oy * Wo;
        const int iy0 = oy * a.sh - a.ph, ix0 = ox * a.sw - a.pw;
        unsigned xbits = 0, vm = 0;
        for (int kx = 0; kx < a.kw; ++kx) xbits |= ((unsigned)(ix0 + kx) < (unsigned)W ? 1u : 0u) << kx;
        for (int ky = 0; ky < a.kh; ++ky)
            if ((unsigned)(iy0 + ky) < (unsigned)H) vm |= xbits << (ky * a.kw);
        vmask[i] = ok ? vm : 0u;
        wl64[i] = W * 64;
        // byte offset of (image origin + tap (0, 0), logical chunk) within one channel slab of a plane
        base[i] = (first + b * H * W + iy0 * W + ix0) * 64 + swz16;
    }
    __amdgpu_buffer_rsrc_t xr[NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p)
        xr[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.xp) + (size_t)p * a.x_pstride, 0, (int)a.plane_bytes, 0x00020000);
    const uint8_t* wtile = a.wp + (size_t)nt * a.slabs * WBUF;
    const int taps = a.kh * a.kw, cslabs = a.C / CV_BK;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;

    // K-slab -> (channel slab, ky, kx), kept as counters: a division per slab would cost more than the slab's DMA issue
    int s_c = 0, s_ky = 0, s_kx = 0;
    if (s_begin != 0) {
        const int s_tap = s_begin % taps;
        s_c = s_begin / taps;
        s_ky = s_tap / a.kw;
        s_kx = s_tap - s_ky * a.kw;
    }
    int s_t = s_ky * a.kw + s_kx;                                      // tap index of the next slab to stage
    auto dma_x = [&](int buf, bool issue = true) {
        uint8_t* xb = smem + buf * BUF;
        if constexpr (DUAL) {
            // (uniform selects, no branch: the slab's source tensor, its buffer descriptor and the row offsets within it)
            const bool seg2 = s_c >= a.c1_slabs;
            const uint8_t* pb = seg2 ? a.xp2 : a.xp;
            const long long pstr = seg2 ? a.x2_pstride : a.x_pstride;
            const int nrec = (int)(seg2 ? a.plane2_bytes : a.plane_bytes);
            const int uni2 = seg2 ? (s_c - a.c1_slabs) * (a.x2_np * 64) : s_c * (a.x_np * 64);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned oob = (vmask[i] & 1u) ^ 1u;
                const unsigned off = (unsigned)((seg2 ? base2[i] : base[i]) + uni2) | (oob << 31);
#pragma unroll
                for (int p = 0; p < NPL; ++p) {
                    const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(pb) + (size_t)p * pstr, 0, nrec, 0x00020000);
                    if (issue) __builtin_amdgcn_raw_ptr_buffer_load_lds(xs, (lds_ptr)(xb + p * (BM * 64) + (2 * wave + i) * 1024), 16, off, 0, 0, 0);
                }
            }
            ++s_c;
            return;
        }
        const int uni = (grp * cslabs + s_c) * (a.x_np * 64) + s_kx * 64;  // uniform: this K-slab's channel slab + the tap's column
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // branch-free (a select the compiler turns into an exec-masked branch would split the MFMA block)
            const unsigned oob = ((vmask[i] >> s_t) & 1u) ^ 1u;
            const unsigned off = (unsigned)(base[i] + s_ky * wl64[i] + uni) | (oob << 31);
#pragma unroll
            for (int p = 0; p < NPL; ++p)
                if (issue) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr[p], (lds_ptr)(xb + p * (BM * 64) + (2 * wave + i) * 1024), 16, off, 0, 0, 0);
        }
        ++s_t;
        if (++s_kx == a.kw) {
            s_kx = 0;
            if (++s_ky == a.kh) { s_ky = 0; s_t = 0; ++s_c; }
        }
    };
    // weight pieces as buffer loads: the per-lane part of the address is 16 lane (one loop-invariant register), slab and piece go into the scalar offset
    // (round 4, measured on conv_planar_kx3_kernel: -0.8 % against global_load_lds with a 64-bit per-lane address; CV_WBUFFER=0 restores that form)
#ifndef CV_WBUFFER
#define CV_WBUFFER 1
#endif
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(wtile), 0, a.slabs * WBUF, 0x00020000);
    const int lane16 = lane * 16;
    auto dma_w = [&](int slab, int buf) {
        uint8_t* wb = smem + buf * BUF + XBUF;
#pragma unroll
        for (int j = 0; j < WDMA; ++j) {
            const int wi = wave + NWAVES * j;
            // (default cache policy: the nontemporal hint on this stream, which every CU re-reads from L2, measured -1.5 %)
#if CV_WBUFFER
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr)(wb + wi * 1024), 16, lane16, slab * WBUF + wi * 1024, 0, 0);
#else
            __builtin_amdgcn_global_load_lds((glb_ptr)(wtile + (size_t)slab * WBUF + wi * 1024 + lane16), (lds_ptr)(wb + wi * 1024), 16, 0, 0);
#endif
        }
    };

    f32x4 acc16[4][2 * NJ], accl16[4][2 * NJ];   // main products / plane-correction products (v_mfma_f32_16x16x32 C layout)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2 * NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc16[i][j][r] = 0.0f; accl16[i][j][r] = 0.0f; }

    if constexpr (ST == 3) {
        // ---- three-buffer ring (fp16 format) ------------------------------------------------------------------------
        // With two buffers every K-slab opens with a barrier followed by a burst of fragment reads (12 ds_read_b128 per
        // wave, 8 waves) that the matrix pipe waits for: the PMC passes show it busy 48 % of the kernel's cycles, 57 % with
        // the DMA removed, bank conflicts ~0.  Here slab s+2 is in flight while slab s is multiplied, so slab s+1 is
        // already resident during the second half of slab s and its fragments are read then, between the MFMAs that still
        // work on slab s: the activation fragments of its first half into the registers the first half of slab s just
        // freed, each weight fragment as soon as the last MFMA of slab s that uses its predecessor has issued (the MFMAs
        // run column-tile-major for that).  One barrier per slab as before, placed mid-slab.
        constexpr int NDMA = 2 * NPL + WDMA;             // DMA instructions per wave and slab
        constexpr int MPC = NPL == 2 ? 6 : 2;            // MFMAs per 16-column tile and half: 2 row tiles x (3 plane products | 1)
        constexpr int NM = 2 * NJ * MPC;                 // MFMAs per half
#define MM16(x_, y_, c_) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x_), __builtin_bit_cast(f16x8, y_), c_, 0, 0, 0)
        const int r16 = lane & 15, kc = lane >> 4;
        int xoff[4], woff[2 * NJ];                       // fragment byte offsets within a buffer (plane 0)
#pragma unroll
        for (int i = 0; i < 4; ++i) xoff[i] = lds_off(wm * 64 + i * 16 + r16, kc);
#pragma unroll
        for (int j = 0; j < 2 * NJ; ++j) woff[j] = XBUF + lds_off(wn * (32 * NJ) + j * 16 + r16, kc);
        dma_x(0);
        dma_w(s_begin, 0);
        dma_x(1);
        dma_w(min(s_begin + 1, s_end - 1), 1);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NDMA) : "memory");
        bf16x8 bf[2 * NJ][NPL], af0[2][NPL], af1[2][NPL];
#pragma unroll
        for (int j = 0; j < 2 * NJ; ++j)
#pragma unroll
            for (int p = 0; p < NPL; ++p) bf[j][p] = *reinterpret_cast<const bf16x8*>(smem + woff[j] + p * WPL);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < NPL; ++p) af0[i][p] = *reinterpret_cast<const bf16x8*>(smem + xoff[i] + p * (BM * 64));
        int cur = 0, nxt = 1, dmb = 2;                   // ring positions of slab s, slab s+1 and of the DMA target (slab s+2)
        // first half of a slab: row tiles 0, 1 (fragments already in registers); reads the second half's activation
        // fragments and starts slab S_+2 on its way in
#define RING_HALF0(S_, DMA_)                                                                                                        \
        {                                                                                                                       \
            const uint8_t* xs = smem + cur * BUF;                                                                               \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                       \
                _Pragma("unroll") for (int p = 0; p < NPL; ++p)                                                                 \
                    if (!(ABL & 4)) af1[i][p] = *reinterpret_cast<const bf16x8*>(xs + xoff[2 + i] + p * (BM * 64));            \
            if (DMA_ && !(ABL & 1)) {                                                                                           \
                dma_x(dmb, !(ABL & 32) && (!(ABL & 16) || ((S_) + 2) % 3 == 0));                                                \
                if (!(ABL & 64)) dma_w(min((S_) + 2, s_end - 1), dmb);                                                          \
            }                                                                                                                   \
            _Pragma("unroll") for (int j = 0; j < 2 * NJ; ++j) {                                                                \
                if constexpr (NPL == 2) {                                                                                       \
                    /* the two MFMAs of a correction accumulator are kept two instructions apart */                             \
                    const f32x4 c0 = MM16(af0[0][NPL - 1], bf[j][0], accl16[0][j]);                                             \
                    const f32x4 c1 = MM16(af0[1][NPL - 1], bf[j][0], accl16[1][j]);                                             \
                    acc16[0][j] = MM16(af0[0][0], bf[j][0], acc16[0][j]);                                                       \
                    accl16[0][j] = MM16(af0[0][0], bf[j][NPL - 1], c0);                                                         \
                    acc16[1][j] = MM16(af0[1][0], bf[j][0], acc16[1][j]);                                                       \
                    accl16[1][j] = MM16(af0[1][0], bf[j][NPL - 1], c1);                                                         \
                } else {                                                                                                        \
                    acc16[0][j] = MM16(af0[0][0], bf[j][0], acc16[0][j]);                                                       \
                    acc16[1][j] = MM16(af0[1][0], bf[j][0], acc16[1][j]);                                                       \
                }                                                                                                               \
            }                                                                                                                   \
            _Pragma("unroll") for (int k = 0; k < NM; ++k) {                                                                    \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    /* one MFMA (16 cycles) */                \
                if (k < 2 * NPL) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   /* second half's activation fragments */  \
                if (DMA_) {                                                                                                     \
                    __builtin_amdgcn_sched_group_barrier(0x006, 3, 0);                /* VALU / SALU of the DMA addresses */    \
                    if (k % (NM / NDMA) == NM / NDMA - 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   /* a DMA */      \
                }                                                                                                               \
            }                                                                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                                  \
        }
        // second half: row tiles 2, 3; with PRE_ the fragments of the next slab (buffer nxt) replace this slab's as they
        // retire (the MFMAs run column-tile-major for that)
#define RING_HALF1(PRE_, S_, DMA_)                                                                                                      \
        {                                                                                                                       \
            const uint8_t* xn = smem + nxt * BUF;                                                                               \
            if (DMA_ && !(ABL & 1)) {                                                                                           \
                dma_x(dmb, !(ABL & 32) && (!(ABL & 16) || ((S_) + 2) % 3 == 0));                                                \
                if (!(ABL & 64)) dma_w(min((S_) + 2, s_end - 1), dmb);                                                          \
            }                                                                                                                   \
            if (PRE_ && !(ABL & 4)) {                                                                                           \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                   \
                    _Pragma("unroll") for (int p = 0; p < NPL; ++p)                                                             \
                        af0[i][p] = *reinterpret_cast<const bf16x8*>(xn + xoff[i] + p * (BM * 64));                             \
            }                                                                                                                   \
            _Pragma("unroll") for (int j = 0; j < 2 * NJ; ++j) {                                                                \
                if constexpr (NPL == 2) {                                                                                       \
                    const f32x4 c0 = MM16(af1[0][NPL - 1], bf[j][0], accl16[2][j]);                                             \
                    const f32x4 c1 = MM16(af1[1][NPL - 1], bf[j][0], accl16[3][j]);                                             \
                    acc16[2][j] = MM16(af1[0][0], bf[j][0], acc16[2][j]);                                                       \
                    accl16[2][j] = MM16(af1[0][0], bf[j][NPL - 1], c0);                                                         \
                    acc16[3][j] = MM16(af1[1][0], bf[j][0], acc16[3][j]);                                                       \
                    accl16[3][j] = MM16(af1[1][0], bf[j][NPL - 1], c1);                                                         \
                } else {                                                                                                        \
                    acc16[2][j] = MM16(af1[0][0], bf[j][0], acc16[2][j]);                                                       \
                    acc16[3][j] = MM16(af1[1][0], bf[j][0], acc16[3][j]);                                                       \
                }                                                                                                               \
                if (PRE_ && !(ABL & 4)) {                                                                                       \
                    _Pragma("unroll") for (int p = 0; p < NPL; ++p)                                                             \
                        bf[j][p] = *reinterpret_cast<const bf16x8*>(xn + woff[j] + p * WPL);                                    \
                }                                                                                                               \
            }                                                                                                                   \
            if (PRE_) {                                                                                                         \
                _Pragma("unroll") for (int k = 0; k < NM; ++k) {                                                                \
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                          \
                    if (k < 2 * NPL) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                         \
                    else if (k >= MPC && ((k - MPC) % MPC) < NPL) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);            \
                    if (DMA_) {                                                                                                 \
                        __builtin_amdgcn_sched_group_barrier(0x006, 3, 0);                                                      \
                        if (k % (NM / NDMA) == NM / NDMA - 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                \
                    }                                                                                                           \
                }                                                                                                               \
                __builtin_amdgcn_sched_barrier(0);                                                                              \
            }                                                                                                                   \
        }
        // (Tried: the two waves that share a SIMD issuing their DMAs in different halves -- early waves in the first half with
        // the DMAs left in flight over the barrier, late waves right after it -- so that one of the pair always feeds the
        // matrix pipe while the other sits in a DMA's issue.  445 vs 429 us on the 145-GF layer, and 256 VGPRs: rejected.)
        RING_HALF0(s_begin, true);
        for (int s = s_begin; s < s_end - 1; ++s) {
            // slab s+1 (issued one slab ago) has landed for this wave, then for every wave; slab s+2's DMAs stay in flight
            if constexpr (ABL & 8) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // no wait for the DMAs to land
            else if constexpr (ABL & 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NDMA) : "memory");
            else if constexpr ((ABL & 48) != 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(WDMA) : "memory");   // fewer A DMAs in flight
            else if constexpr ((ABL & 64) != 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * NPL) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NDMA) : "memory");
            RING_HALF1(true, s, false);
            cur = nxt; nxt = dmb; dmb = dmb == 2 ? 0 : dmb + 1;
            RING_HALF0(s + 1, true);
        }
        RING_HALF1(false, 0, false);
#undef RING_HALF0
#undef RING_HALF1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's last (unused) slabs must land before the epilogue reuses the LDS
#undef MM16
    } else {
    dma_x(s_begin & 1);
    dma_w(s_begin, s_begin & 1);
    for (int s = s_begin; s < s_end; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of slab s has landed
        __syncthreads();                   // ... everyone's has, and buffer (s+1)&1 is no longer read
        const uint8_t* xs = smem + (s & 1) * BUF;
        const uint8_t* ws = xs + XBUF;
        // Source order matters: an LDS-DMA is an LDS store to the compiler, so fragment reads cannot move above it.
        // Each half (k-step) therefore reads its fragments first and is followed by its share of slab s+1's DMA, and the
        // schedule directives deal the DMA's address arithmetic and issue out between that half's MFMAs.  Issued as one
        // block ahead of the MFMAs the same ~60 instructions took 850-2800 cycles: the SIMD's other wave is issuing
        // MFMAs then and takes the issue slots.  Past the last slab the DMA re-reads it into the idle buffer (no branch).
        {
            // 16x16x32: one instruction covers the slab's 32 channels; lane (l & 15, l >> 4) holds row l & 15, chunk l >> 4
#define MFMA16(x_, y_, c_, i0_, i1_, i2_) \
    (DT == 1 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x_), __builtin_bit_cast(f16x8, y_), c_, 0, 0, 0) \
             : __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_, y_, c_, 0, 0, 0))
            const int r16 = lane & 15, kc = lane >> 4;
            // 16-column tiles of this wave that hold real output channels (wave-uniform): narrow layers and zero-padded
            // groups skip the MFMAs of the others
            const int jn = min(2 * NJ, (creal - (n0g + wn * (32 * NJ)) + 15) >> 4);
            bf16x8 bf[2 * NJ][NPL];
#pragma unroll
            for (int j = 0; j < 2 * NJ; ++j)
#pragma unroll
                for (int p = 0; p < NPL; ++p)
                    bf[j][p] = *reinterpret_cast<const bf16x8*>(ws + p * WPL + lds_off(wn * (32 * NJ) + j * 16 + r16, kc));
            if (NJ == 2 || jn == 2 * NJ) {   // every column tile is real (always, for the 128-wide tiles): one branch-free
                                             // block -- the schedule directives need one, and a second copy of the
                                             // 128-wide body spills registers
    #pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    bf16x8 af[2][NPL];
    #pragma unroll
                    for (int i = 0; i < 2; ++i)
    #pragma unroll
                        for (int p = 0; p < NPL; ++p)
                            af[i][p] = *reinterpret_cast<const bf16x8*>(xs + p * (BM * 64) + lds_off(wm * 64 + (2 * hf + i) * 16 + r16, kc));
                    if (hf == 0) dma_x((s + 1) & 1);
                    else dma_w(min(s + 1, s_end - 1), (s + 1) & 1);
    #pragma unroll
                    for (int i = 0; i < 2; ++i)
    #pragma unroll
                        for (int j = 0; j < 2 * NJ; ++j) {
                            if constexpr (NPL >= 2) {
                                f32x4 c = accl16[2 * hf + i][j];
                                if constexpr (NPL == 3) {
                                    c = MFMA16(af[i][NPL - 1], bf[j][0], c, 0, 0, 0);
                                    c = MFMA16(af[i][0], bf[j][NPL - 1], c, 0, 0, 0);
                                    c = MFMA16(af[i][1], bf[j][1], c, 0, 0, 0);
                                }
                                c = MFMA16(af[i][1], bf[j][0], c, 0, 0, 0);
                                accl16[2 * hf + i][j] = MFMA16(af[i][0], bf[j][1], c, 0, 0, 0);
                            }
                            acc16[2 * hf + i][j] = MFMA16(af[i][0], bf[j][0], acc16[2 * hf + i][j], 0, 0, 0);
                        }
    #pragma unroll
                    for (int k = 0; k < 8 * NJ * NPL; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    // one MFMA (16 cycles)
                        if (hf == 0 && (k & 1)) __builtin_amdgcn_sched_group_barrier(0x006, 3, 0);
                        else if (k % 4 == 1) __builtin_amdgcn_sched_group_barrier(0x006, 1, 0);
                        if (k % 6 == 3) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // a DMA
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if constexpr (NJ == 1) {  // narrow layer / zero-padded group (64-wide tiles): guarded column tiles
    #pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    bf16x8 af[2][NPL];
    #pragma unroll
                    for (int i = 0; i < 2; ++i)
    #pragma unroll
                        for (int p = 0; p < NPL; ++p)
                            af[i][p] = *reinterpret_cast<const bf16x8*>(xs + p * (BM * 64) + lds_off(wm * 64 + (2 * hf + i) * 16 + r16, kc));
                    if (hf == 0) dma_x((s + 1) & 1);
                    else dma_w(min(s + 1, s_end - 1), (s + 1) & 1);
    #pragma unroll
                    for (int i = 0; i < 2; ++i)
    #pragma unroll
                        for (int j = 0; j < 2 * NJ; ++j) {
                            if (j >= jn) continue;
                            if constexpr (NPL >= 2) {
                                f32x4 c = accl16[2 * hf + i][j];
                                if constexpr (NPL == 3) {
                                    c = MFMA16(af[i][NPL - 1], bf[j][0], c, 0, 0, 0);
                                    c = MFMA16(af[i][0], bf[j][NPL - 1], c, 0, 0, 0);
                                    c = MFMA16(af[i][1], bf[j][1], c, 0, 0, 0);
                                }
                                c = MFMA16(af[i][1], bf[j][0], c, 0, 0, 0);
                                accl16[2 * hf + i][j] = MFMA16(af[i][0], bf[j][1], c, 0, 0, 0);
                            }
                            acc16[2 * hf + i][j] = MFMA16(af[i][0], bf[j][0], acc16[2 * hf + i][j], 0, 0, 0);
                        }
    #pragma unroll
                    for (int k = 0; k < 8 * NJ * NPL; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    // one MFMA (16 cycles)
                        if (hf == 0 && (k & 1)) __builtin_amdgcn_sched_group_barrier(0x006, 3, 0);
                        else if (k % 4 == 1) __builtin_amdgcn_sched_group_barrier(0x006, 1, 0);
                        if (k % 6 == 3) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // a DMA
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
}
    }
    }
#undef MFMA16

    __syncthreads();                                   // all fragment reads of the last slab are done
    constexpr float LS = DT == 1 ? 1.0f / STM_F16_LOW_SCALE : 1.0f;   // fp16 planes: the corrections carry the low-plane scale
    park16<NJ>(acc16, accl16, smem, wave, lane, LS);
    if constexpr (CLS && NJ == 2) {
        if (a_in.pool) {
            pooled_epilogue(a, a_in.pool, a_in.pool_ld, smem, wave, lane, m0 + wm * 64, nt * BN + wn * 64 + lane);
            return;
        }
    }
    if (a.splitk > 1) {
        // raw fp32 partial sums of this K range; planar_splitk_finish_kernel adds the parts and runs the epilogue
        constexpr int EP_LD = 32 * NJ + 4, LPR = 4 * NJ;
        const float* park = park_base<NJ>(smem, wave);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int seg = lane % LPR, prow = lane / LPR;
        const int col = nt * BN + wn * (32 * NJ) + seg * 8;          // column in the tile-padded space
#pragma unroll
        for (int pass = 0; pass < LPR; ++pass) {
            const int pr = pass * (64 / LPR) + prow;
            const int m = m0 + wm * 64 + pr;
            if (m >= a.M) continue;
            float* o = a.partial + ((size_t)ksp * a.M + m) * a.ldp + col;
            *reinterpret_cast<f32x4*>(o) = *reinterpret_cast<const f32x4*>(park + pr * EP_LD + seg * 8);
            *reinterpret_cast<f32x4*>(o + 4) = *reinterpret_cast<const f32x4*>(park + pr * EP_LD + seg * 8 + 4);
        }
    } else {
        planar_epilogue_tail<NJ, DT == 1 ? (NPL == 2 ? 1 : 2) : 0>(a, smem, wave, lane, m0, n0g, grp, wm, wn);
    }
#endif
}

// ---- kx-reuse staging inside the ring loop (round 4, third attempt: see DESIGN section 4 item 3 for the two that lost) --------------------------
// For a stride-1 convolution with kw = 3 the 256 x 128 ring kernel above stages the activation tile once per TAP: three of the nine (or 3 kh)
// K-slabs of a channel slab fetch the same BM pixels shifted by one.  Here a STAGE is one (channel slab, ky): the BM + 2 consecutive pixels of the flat
// pixel run m0 - 1 .. m0 + BM (each at its own row oy + ky - ph) are staged ONCE, into one of two 34-KB buffers, and serve kx = 0, 1, 2: tap kx of
// tile pixel r is staged row r + kx.  Where that neighbour is in another image row (ox + kx - 1 outside [0, W)) the lane's read address points at an
// always-zero staged row instead (rows 258 .. 271 of a buffer are filled by out-of-range DMAs): no masking instruction in the loop.  Weights keep
// the three-slot ring, one slot per tap (slot = kx).  Per wave and tap: 11 / 3 LDS-DMA instructions instead of 6, 1 / 2.8 of the activation bytes.
// What is different from the two earlier attempts: the three taps of a stage are unrolled with their read offsets in three register sets (the ring
// kernel turned out to use 213 registers, not 256: the twelve offsets fit), so there is no per-tap register rotation and no run-time tap index; a
// stage's DMAs are issued in the first half of taps 0 and 1 and are older than the weight slab the barrier of tap 2 waits for, so the counted waits
// stay compile-time constants (5, 4, 2); every wave issues the same number of DMAs (the 17th row group is fetched by every wave: same bytes, same place).
// And the loop is written as `for (t < T - 1) { three taps } + last stage` -- with `for (;;) { ...; if (++t == T) break; ... }` around the same macros
// the register allocator spilled 660 bytes per lane (fragments hoisted over the exit edge); this form takes 250 registers and no scratch.
// Same products in the same order as conv_planar_kernel<2, 2, 2, 1, 3>: bit-equal outputs (tests/test_gpu_conv.py).
// Measured (profiles/r04_kx3_ab.txt, r04_kx3_ablations.txt): the 3x3 layers -6 % (1417 -> 1335 us proto-net layer, 1801 -> 1683 us tower layer at batch 32),
// the step 22.2 -> 21.8 ms.  Timing ablations of THIS kernel: no DMA at all 870 us, weight DMAs only 960 us, plus the stage's 40 pieces 1335 us -- and with
// those 40 pieces reading hot weight lines instead of activations still 1257 us: what the staging costs is the DMA INSTRUCTIONS a wave issues between
// its MFMAs (a marginal piece costs four times what the first 16 per tap and CU cost), not the bytes, their source, or HBM latency (all stages reading
// channel slab 0: same time).
constexpr int KX3_XROWS = 272, KX3_XPL = KX3_XROWS * 64, KX3_ABUF = 2 * KX3_XPL, KX3_WBUF = 2 * 128 * 64, KX3_W0 = 2 * KX3_ABUF;
constexpr int KX3_ZROW = 260;
constexpr int KX3_LDS_LOOP = KX3_W0 + 3 * KX3_WBUF;
#ifndef KX3_SCHED
#define KX3_SCHED 24     // 0: no schedule hints (experiment)
#endif
#ifndef KX3_WBUFFER
#define KX3_WBUFFER 1    // weight pieces as buffer loads (scalar offset + 16 lane: 242 registers, -0.8 %); 0 = global_load_lds with a 64-bit per-lane address
#endif
#ifndef KX3_ABL
#define KX3_ABL 0        // timing builds (RESULTS WRONG): 1 no DMA at all, 2 no activation DMA, 4 no fragment reads, 8 every stage stages channel slab 0 (L2-resident rows), 16 activation pieces read weight bytes instead (same count of DMAs, hot lines), 32 activation pieces land behind the weight ring (another LDS region), 64 the ring pre-filled once with random fp16 values
#endif
template <int ABL = 0, bool WIN = false>
__global__ __launch_bounds__(512, 1) void conv_planar_kx3_kernel(const typename std::conditional<WIN, PlanarArgsCls, PlanarArgs>::type a_in)
{
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __align__(16) uint8_t smem[];
    constexpr int NJ = 2, BM = 256, BN = 128, WPL = BN * 64;
    // staged rows per activation buffer: BM + 2 (17 row groups of 16) -- or, WIN, the kw = 3 classes of a window set (TemporalNet's border classes with
    // all three column taps: the 5-pixel-wide windows of a 7x7 map): a tile's 256 class pixels are at most 53 window rows, each staged as its Wo + 2 input
    // pixels (371 rows, 24 groups), and tap kx of pixel (window row q, ox) is staged row (Wo + 2) q + ox + kx -- every tap inside the map, no zero row
    constexpr int XROWS = WIN ? 384 : KX3_XROWS, XPL = XROWS * 64, ABUF = 2 * XPL, W0 = 2 * ABUF;
    int logical, n_tiles_ = a_in.n_tiles, cls_c = 0;
    if constexpr (WIN) {
        // tile -> class, as conv_planar_kernel<..., CLS>: pixel tiles dealt round-robin to the XCDs, each with all its channel tiles
        const int j = blockIdx.x >> 3;
        const int jm = j / n_tiles_, nt_ = j - jm * n_tiles_;
        logical = (jm * 8 + (int)(blockIdx.x & 7)) * n_tiles_ + nt_;
        if (logical >= a_in.cls_tiles) return;
#pragma unroll
        for (int i = 1; i < 9; ++i)
            if (i < a_in.n_cls && logical >= a_in.cls[i].tile0) cls_c = i;
        logical -= a_in.cls[cls_c].tile0;
    } else {
        const int tiles = a_in.m_tiles * n_tiles_;
        const int per_xcd = (tiles + 7) >> 3;
        logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
        if (logical >= tiles) return;
    }
    const PlanarArgs& a = a_in;
    // the fields a class replaces, as scalars (the class-adjusted copy of the whole argument block is only built for the epilogue, after the loop: kept
    // live across the loop it went to scratch memory)
    int KH = a.kh, PH = a.ph, PW = a.pw, HO = a.Ho, WO = a.Wo, MM = a.M, SLABS = a.slabs;
    const uint8_t* WP = a.wp;
    if constexpr (WIN) {
        KH = a_in.cls[cls_c].kh; PH = a_in.cls[cls_c].ph; PW = a_in.cls[cls_c].pw; HO = a_in.cls[cls_c].Ho; WO = a_in.cls[cls_c].Wo;
        MM = a_in.cls[cls_c].M; SLABS = a_in.cls[cls_c].slabs; WP = a_in.cls[cls_c].wp;
    }
    const int mt = n_tiles_ == 1 ? logical : (n_tiles_ == 2 ? logical >> 1 : (n_tiles_ == 4 ? logical >> 2 : logical / n_tiles_));
    const int nt = logical - mt * n_tiles_;
    const int m0 = mt * BM;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int grp = a.groups == 1 ? 0 : nt / a.ntpg;
    const int n0g = (nt - grp * a.ntpg) * BN;
    const int cslabs = a.C / CV_BK;

    // pixel m of the launch's pixel axis -> (valid, first pixel of its level, image, row, column, level size)
    auto decode = [&](int m, int& first, int& b, int& oy, int& ox, int& H, int& W) -> bool {
        const bool ok = m >= 0 && m < MM;
        const int mm = ok ? m : 0;
        H = a.H; W = a.W; first = 0;
        int local = mm;
        if (a.n_levels > 0) {
#pragma unroll
            for (int l = 0; l < 8; ++l)
                if (l < a.n_levels && mm >= a.lvl_start[l]) { first = a.lvl_start[l]; H = a.lvl_h[l]; W = a.lvl_w[l]; }
            local = mm - first;
            b = local / (H * W);
            const int rem = local - b * (H * W);
            oy = rem / W;
            ox = rem - oy * W;
        } else {
            const int hw = H * W;
            b = (int)((float)local * a.inv_hw);
            int rem = local - b * hw;
            if (rem < 0) { --b; rem += hw; } else if (rem >= hw) { ++b; rem -= hw; }
            oy = (int)((float)rem * a.inv_w);
            int t = rem - oy * W;
            if (t < 0) --oy; else if (t >= W) ++oy;
            ox = rem - oy * W;
        }
        return ok;
    };

    // DMA duties of this lane: staged rows j = 16 q + (lane >> 2) of row groups q = wave, wave + 8 and 16 (WIN: wave + 16); staged row j is launch pixel
    // m0 - 1 + j -- WIN: input pixel xi - pw of window row wr0 + j / Ws, xi = j mod Ws
    int base[3], wlv[3];          // wlv = row pitch in bytes (a multiple of 64) | one validity bit per ky in the low six bits
    const int slot = lane & 3;
    const int Ws = WO + 2, wr0 = WIN ? m0 / WO : 0;        // (WIN) staged pixels per window row; first window row of the tile
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int q = i == 0 ? wave : (i == 1 ? wave + 8 : (WIN ? wave + 16 : 16));
        const int j = q * 16 + (lane >> 2);
        if constexpr (WIN) {
            const int qr = j / Ws, xi = j - qr * Ws;
            const int wr = wr0 + qr;                             // window row (image b, window line oy) of this staged pixel
            const bool ok = wr * WO < MM;
            const int b = wr / HO, oy = wr - b * HO;
            wlv[i] = a.W * 64 + (ok ? 63 : 0);                   // every tap of a class lies inside the map
            base[i] = ((b * a.H + oy - PH) * a.W + (xi - PW)) * 64 + ((slot ^ swz(j)) << 4);
        } else {
            int first, b, oy, ox, H, W;
            const bool ok = decode(m0 - 1 + j, first, b, oy, ox, H, W) && j < BM + 2;
            unsigned vm = 0;
            for (int ky = 0; ky < KH; ++ky)
                if ((unsigned)(oy - PH + ky) < (unsigned)H) vm |= 1u << ky;
            wlv[i] = W * 64 + (int)(ok ? vm : 0u);
            base[i] = (first + b * H * W + (oy - PH) * W + ox) * 64 + ((slot ^ swz(j)) << 4);
        }
    }
    __amdgpu_buffer_rsrc_t xr[2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
        xr[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.xp) + (size_t)p * a.x_pstride, 0, (int)a.plane_bytes, 0x00020000);
    const uint8_t* wtile = WP + (size_t)nt * SLABS * KX3_WBUF;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;

    // fragment read offsets: activation row tile i at tap kx = staged rows 64 wm + 16 i + r16 + kx, or the zero row where the tap leaves the image row
    const int r16 = lane & 15, kc = lane >> 4;
    // (row tiles 16 apart share their chunk swizzle -- it depends on bits 2, 3 of the row -- so tile i / column tile j is a constant 1 KB further:
    // one register for the weight fragments and one for the unmasked middle tap instead of four each)
    int xo[3][4], woff0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wm * 64 + i * 16 + r16;
        if constexpr (WIN) {
            const int m = m0 + r;
            const int wr = m / WO, ox = m - wr * WO;
            const int sr = (wr - wr0) * Ws + ox;                 // staged row of tap 0
            xo[0][i] = lds_off(sr, kc); xo[1][i] = lds_off(sr + 1, kc); xo[2][i] = lds_off(sr + 2, kc);
        } else {
            int first, b, oy, ox, H, W;
            decode(m0 + r, first, b, oy, ox, H, W);
            const int z = lds_off(KX3_ZROW, kc);
            xo[0][i] = ox == 0 ? z : lds_off(r, kc);
            xo[1][i] = lds_off(r + 1, kc);           // (= xo[1][0] + 1024 i: only xo[1][0] stays live)
            xo[2][i] = ox == W - 1 ? z : lds_off(r + 2, kc);
        }
    }
    woff0 = W0 + lds_off(wn * 64 + r16, kc);

    // stage counters of the NEXT activation stage to issue: (channel slab, ky); stage index and its buffer
    int st_c = 0, st_ky = 0, st_buf = 0;
    auto dma_a = [&](int i, int p) {          // piece (row group of duty i, plane p) of the stage (st_c, st_ky) into buffer st_buf
        const int q = i == 0 ? wave : (i == 1 ? wave + 8 : (WIN ? wave + 16 : 16));
        const unsigned oob = (((unsigned)wlv[i] >> st_ky) & 1u) ^ 1u;
        const unsigned off = (unsigned)(base[i] + st_ky * (wlv[i] & ~63) + (grp * cslabs + ((ABL & 8) ? 0 : st_c)) * (a.x_np * 64)) | (oob << 31);
        if constexpr ((ABL & 16) != 0) {        // the same number of DMAs into the same LDS places, but reading weight-tile bytes the way dma_w does
            __builtin_amdgcn_global_load_lds((glb_ptr)(wtile + (size_t)(q & 15) * 1024 + lane * 16), (lds_ptr)(smem + st_buf * ABUF + p * XPL + q * 1024), 16, 0, 0);
        } else if constexpr ((ABL & 32) != 0) { // the real activation pieces, landing in the spare LDS behind the weight ring instead of the activation buffers
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr[p], (lds_ptr)(smem + W0 + 3 * KX3_WBUF + ((q + 17 * p) & 15) * 1024), 16, off, 0, 0, 0);
        } else
        if (!(ABL & 3)) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr[p], (lds_ptr)(smem + st_buf * ABUF + p * XPL + q * 1024), 16, off, 0, 0, 0);
    };
    auto stage_advance = [&]() {
        st_buf ^= 1;
        if (++st_ky == KH) { st_ky = 0; ++st_c; }
    };
    // weight pieces as buffer loads: per-lane part of the address = 16 lane (one loop-invariant register), the rest in the scalar offset
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(wtile), 0, SLABS * KX3_WBUF, 0x00020000);
    const int lane16 = lane * 16;
    auto dma_w = [&](int slab, int wslot) {
        uint8_t* wb = smem + W0 + wslot * KX3_WBUF;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int wi = wave + 8 * j;
#if KX3_WBUFFER
            if (!(ABL & 1)) __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr)(wb + wi * 1024), 16, lane16, slab * KX3_WBUF + wi * 1024, 0, 0);
#else
            if (!(ABL & 1)) __builtin_amdgcn_global_load_lds((glb_ptr)(wtile + (size_t)slab * KX3_WBUF + wi * 1024 + lane16), (lds_ptr)(wb + wi * 1024), 16, 0, 0);
#endif
        }
    };

    f32x4 acc16[4][4], accl16[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc16[i][j][r] = 0.0f; accl16[i][j][r] = 0.0f; }

#define KX3_XO(K_, I_) ((K_) == 1 && !WIN ? xo[1][0] + (I_) * 1024 : xo[K_][I_])
#define MM16(x_, y_, c_) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x_), __builtin_bit_cast(f16x8, y_), c_, 0, 0, 0)
    const int S = SLABS;                       // K-slabs = 3 taps x stages
    if constexpr ((ABL & 64) != 0) {
        // timing build: the whole ring pre-filled ONCE with pseudo-random fp16 values in [0.5, 2) of both signs -- with ABL 1 | 64 the MFMAs then run on
        // operands that look like data although nothing is staged (an LDS that is never written hands the matrix pipe constant fragments, and a pipe
        // fed constants draws much less power: at the board's power limit that reads as a far faster loop)
        for (int i = tid; i < (W0 + 3 * KX3_WBUF) / 4; i += 512) {
            unsigned h = (unsigned)i * 2654435761u + (unsigned)blockIdx.x * 40503u;
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            reinterpret_cast<unsigned*>(smem)[i] = (h & 0x87ff87ffu) | 0x38003800u;
        }
        __syncthreads();
    }
    // prologue: stage 0 whole, weight slabs 0 and 1
    dma_a(0, 0); dma_a(0, 1); dma_a(1, 0); dma_a(1, 1);
    if (WIN) { dma_a(2, 0); dma_a(2, 1); } else dma_a(2, wave & 1);
    stage_advance();
    dma_w(0, 0);
    dma_w(min(1, S - 1), 1);
    asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
    bf16x8 bf[4][2], af0[2][2], af1[2][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p) bf[j][p] = *reinterpret_cast<const bf16x8*>(smem + woff0 + j * 1024 + p * WPL);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) af0[i][p] = *reinterpret_cast<const bf16x8*>(smem + KX3_XO(0, i) + p * XPL);
    int s = 0;                                   // K-slab of the tap being multiplied
    // first half of tap KX_: row tiles 0, 1 (fragments in registers); reads the second half's activation fragments (same stage, same tap) and issues
    // this tap's DMAs: the next stage's pieces (taps 0 and 1 only), then weight slab s + 2 into the slot tap KX_ - 1 just left
#define KX3_HALF0(KX_)                                                                                                              \
    {                                                                                                                               \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                               \
            _Pragma("unroll") for (int p = 0; p < 2; ++p)                                                                           \
                if (!(ABL & 4)) af1[i][p] = *reinterpret_cast<const bf16x8*>(smem + KX3_XO(KX_, 2 + i) + p * XPL);                 \
        if (KX_ == 0) { dma_a(0, 0); dma_a(0, 1); dma_a(1, 0); }                                                 \
        if (KX_ == 1) { dma_a(1, 1); if (WIN) { dma_a(2, 0); dma_a(2, 1); } else dma_a(2, wave & 1); stage_advance(); }                                      \
        dma_w(min(s + 2, S - 1), (KX_ + 2) % 3);                                                               \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                             \
            const f32x4 c0 = MM16(af0[0][1], bf[j][0], accl16[0][j]);                                                               \
            const f32x4 c1 = MM16(af0[1][1], bf[j][0], accl16[1][j]);                                                               \
            acc16[0][j] = MM16(af0[0][0], bf[j][0], acc16[0][j]);                                                                   \
            accl16[0][j] = MM16(af0[0][0], bf[j][1], c0);                                                                           \
            acc16[1][j] = MM16(af0[1][0], bf[j][0], acc16[1][j]);                                                                   \
            accl16[1][j] = MM16(af0[1][0], bf[j][1], c1);                                                                           \
        }                                                                                                                           \
        constexpr int ND = KX_ == 0 ? 5 : (KX_ == 1 ? (WIN ? 5 : 4) : 2);                                                                       \
        _Pragma("unroll") for (int k = 0; k < KX3_SCHED; ++k) {                                                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                      \
            if (k < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                           \
            __builtin_amdgcn_sched_group_barrier(0x006, 3, 0);                                                                      \
            if (k % (24 / ND) == 24 / ND - 1 && k / (24 / ND) < ND) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);              \
        }                                                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                                          \
    }
    // second half: row tiles 2, 3; with PRE_ the fragments of the next tap replace this tap's as they retire: activation rows 0, 1 at the next tap's
    // shift (the offsets already point into the next stage's buffer when KX_ = 2), weight fragments from the next tap's slot
#define KX3_HALF1(KX_, PRE_)                                                                                                        \
    {                                                                                                                               \
        if (PRE_ && !(ABL & 4)) {                                                                                                   \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                           \
                _Pragma("unroll") for (int p = 0; p < 2; ++p)                                                                       \
                    af0[i][p] = *reinterpret_cast<const bf16x8*>(smem + KX3_XO((KX_ + 1) % 3, i) + p * XPL);                        \
        }                                                                                                                           \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                             \
            const f32x4 c0 = MM16(af1[0][1], bf[j][0], accl16[2][j]);                                                               \
            const f32x4 c1 = MM16(af1[1][1], bf[j][0], accl16[3][j]);                                                               \
            acc16[2][j] = MM16(af1[0][0], bf[j][0], acc16[2][j]);                                                                   \
            accl16[2][j] = MM16(af1[0][0], bf[j][1], c0);                                                                           \
            acc16[3][j] = MM16(af1[1][0], bf[j][0], acc16[3][j]);                                                                   \
            accl16[3][j] = MM16(af1[1][0], bf[j][1], c1);                                                                           \
            if (PRE_ && !(ABL & 4)) {                                                                                               \
                _Pragma("unroll") for (int p = 0; p < 2; ++p)                                                                       \
                    bf[j][p] = *reinterpret_cast<const bf16x8*>(smem + woff0 + j * 1024 + ((KX_ + 1) % 3) * KX3_WBUF + p * WPL);             \
            }                                                                                                                       \
        }                                                                                                                           \
        if (PRE_) {                                                                                                                 \
            _Pragma("unroll") for (int k = 0; k < KX3_SCHED; ++k) {                                                                        \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                  \
                if (k < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                       \
                else if (k >= 6 && ((k - 6) % 6) < 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                            \
            }                                                                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                                                      \
        }                                                                                                                           \
    }
#define KX3_BARRIER(N_) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((ABL & 2) ? 2 : (N_)) : "memory")     // (ABL 2: only the weight slab s + 2 is in flight)
    const int T = S / 3;                          // stages
    KX3_HALF0(0);
#define KX3_FLIP(T_)                                                                      \
        {                                                                                 \
            const int d = ((T_) & 1) ? -ABUF : ABUF;                              \
            _Pragma("unroll") for (int k = 0; k < 3; ++k)                                 \
                _Pragma("unroll") for (int i = 0; i < (k == 1 && !WIN ? 1 : 4); ++i) xo[k][i] += d; \
        }
    for (int t = 0; t < T - 1; ++t) {
        KX3_BARRIER(5);
        KX3_HALF1(0, true);
        ++s;
        KX3_HALF0(1);
        KX3_BARRIER(WIN ? 5 : 4);
        KX3_HALF1(1, true);
        ++s;
        KX3_HALF0(2);
        // the activation fragments of this stage are all in registers: from here on the read offsets address the other buffer
        KX3_FLIP(t);
        KX3_BARRIER(2);
        KX3_HALF1(2, true);
        ++s;
        KX3_HALF0(0);
    }
    KX3_BARRIER(5);
    KX3_HALF1(0, true);
    ++s;
    KX3_HALF0(1);
    KX3_BARRIER(WIN ? 5 : 4);
    KX3_HALF1(1, true);
    ++s;
    KX3_HALF0(2);
    KX3_BARRIER(2);
#undef KX3_FLIP
    KX3_HALF1(2, false);
#undef KX3_XO
#undef KX3_HALF0
#undef KX3_HALF1
#undef KX3_BARRIER
#undef MM16
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stage and the weight slabs issued past the end must land before the epilogue reuses the LDS
    __syncthreads();
    park16<NJ>(acc16, accl16, smem, wave, lane, 1.0f / STM_F16_LOW_SCALE);
    if constexpr (WIN) {
        PlanarArgs a_cls = static_cast<const PlanarArgs&>(a_in);
        a_cls.Ho = HO; a_cls.Wo = WO; a_cls.M = MM;
        a_cls.win_off = a_in.cls[cls_c].win_off; a_cls.inv_hw = a_in.cls[cls_c].inv_hw; a_cls.inv_w = a_in.cls[cls_c].inv_w;
        if (a_in.pool) pooled_epilogue(a_cls, a_in.pool, a_in.pool_ld, smem, wave, lane, m0 + wm * 64, nt * BN + wn * 64 + lane);
        else planar_epilogue_tail<NJ, 1>(a_cls, smem, wave, lane, m0, n0g, grp, wm, wn);
    } else {
        planar_epilogue_tail<NJ, 1>(a, smem, wave, lane, m0, n0g, grp, wm, wn);
    }
#endif
}

// Streaming kernels below: workgroup ids are dealt round-robin to the 8 XCDs; a launch of 8 * per_xcd workgroups maps id ->
// (id & 7) * per_xcd + (id >> 3), so each XCD works on a contiguous run of pixels (neighbouring rows share input lines in
// one L2 instead of eight).  Returns -1 for the padding workgroups.
__device__ __forceinline__ int64_t xcd_contiguous_block(int64_t nblocks)
{
    const int64_t per_xcd = (nblocks + 7) >> 3;
    const int64_t b = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    return b < nblocks ? b : -1;
}

// fp32 [n pixels][C] (NHWC) -> three bf16 planes [3][C/32][n][32] (entry into the planar format from a foreign producer);
// thread = 8 channels of one pixel
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, uint8_t* __restrict__ planes, int64_t n, int C,
                                                           int fmt, int* range_flag)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c8n = C >> 3;
    if (idx >= n * c8n) return;
    const int64_t pix = idx / c8n;
    const int c8 = (int)(idx - pix * c8n);
    const float* src = x + pix * C + c8 * 8;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(src), a1 = *reinterpret_cast<const f32x4*>(src + 4);
    const size_t plane_b = (size_t)n * C * 2;
    uint8_t* dst = planes + (((size_t)(c8 >> 2) * n + pix) * 32 + (c8 & 3) * 8) * 2;
    const float v8[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    store_planes8(dst, plane_b, v8, fmt, range_flag, false);
}

// Bilinear resize (F.interpolate(mode="bilinear", align_corners=False): make_net.py's InterpolateModule between the proto-net
// convolutions) of an fp32 NHWC tensor straight into planes: thread = 8 channels of one output pixel; the fp32 upsampled
// tensor (4x the input for the proto-net's x2) is never written or re-read.  Same expression order as the ATen kernel.
__global__ __launch_bounds__(256) void resize_bilinear_planes_kernel(const float* __restrict__ x, uint8_t* __restrict__ planes, int B, int H,
                                                                    int W, int C, int Ho, int Wo, float sy, float sx, int fmt,
                                                                    int* range_flag)
{
    const int c8n = C >> 3;
    const int64_t n = (int64_t)B * Ho * Wo;
    const int64_t blk = xcd_contiguous_block((n * c8n + 255) >> 8);
    if (blk < 0) return;
    const int64_t idx = blk * 256 + threadIdx.x;
    if (idx >= n * c8n) return;
    const int64_t pix = idx / c8n;
    const int c8 = (int)(idx - pix * c8n);
    const int b = (int)(pix / ((int64_t)Ho * Wo));
    const int rem = (int)(pix - (int64_t)b * Ho * Wo);
    const int oy = rem / Wo, ox = rem - oy * Wo;
    // area_pixel_compute_source_index(scale, dst, align_corners=false, cubic=false): max(0, scale * (dst + 0.5) - 0.5)
    const float fy = fmaxf(sy * ((float)oy + 0.5f) - 0.5f, 0.0f), fx = fmaxf(sx * ((float)ox + 0.5f) - 0.5f, 0.0f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly1 = fy - (float)y0, lx1 = fx - (float)x0, ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    const float* base = x + (size_t)b * H * W * C + c8 * 8;
    float v[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(base + ((size_t)y0 * W + x0) * C + 4 * h);
        const f32x4 bq = *reinterpret_cast<const f32x4*>(base + ((size_t)y0 * W + x1) * C + 4 * h);
        const f32x4 c = *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * W + x0) * C + 4 * h);
        const f32x4 d = *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * W + x1) * C + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * h + e] = ly0 * (lx0 * a[e] + lx1 * bq[e]) + ly1 * (lx0 * c[e] + lx1 * d[e]);
    }
    const size_t plane_b = (size_t)n * C * 2;
    uint8_t* dst = planes + (((size_t)(c8 >> 2) * n + pix) * 32 + (c8 & 3) * 8) * 2;
    store_planes8(dst, plane_b, v, fmt, range_flag, true);
}

// ResNet stem tail (backbone.py:73: relu(bn1(conv1)) -> MaxPool2d(3, 2, 1)) on the raw fp32 NHWC convolution output, written as
// planes for layer1: y = relu(max over the 3x3 window (stride 2, pad 1) + folded-BN bias).  The bias add and the ReLU are
// monotone and the bias is per channel, so they commute with the max exactly.  thread = 8 channels of one output pixel.
__global__ __launch_bounds__(256) void bias_relu_maxpool_planes_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                                      uint8_t* __restrict__ planes, int B, int H, int W, int C, int Ho, int Wo,
                                                                      int fmt, int* range_flag)
{
    const int c8n = C >> 3;
    const int64_t n = (int64_t)B * Ho * Wo;
    const int64_t blk = xcd_contiguous_block((n * c8n + 255) >> 8);
    if (blk < 0) return;
    const int64_t idx = blk * 256 + threadIdx.x;
    if (idx >= n * c8n) return;
    const int64_t pix = idx / c8n;
    const int c8 = (int)(idx - pix * c8n);
    const int b = (int)(pix / ((int64_t)Ho * Wo));
    const int rem = (int)(pix - (int64_t)b * Ho * Wo);
    const int oy = rem / Wo, ox = rem - oy * Wo;
    const float* base = x + (size_t)b * H * W * C + c8 * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = -__builtin_inff();
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int iy = 2 * oy - 1 + dy;
        if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int ix = 2 * ox - 1 + dx;
            if ((unsigned)ix >= (unsigned)W) continue;
            const f32x4 a = *reinterpret_cast<const f32x4*>(base + ((size_t)iy * W + ix) * C);
            const f32x4 c = *reinterpret_cast<const f32x4*>(base + ((size_t)iy * W + ix) * C + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = fmaxf(v[e], a[e]); v[4 + e] = fmaxf(v[4 + e], c[e]); }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float t = v[e] + (bias ? bias[c8 * 8 + e] : 0.0f);
        v[e] = t > 0.0f ? t : 0.0f;
    }
    const size_t plane_b = (size_t)n * C * 2;
    uint8_t* dst = planes + (((size_t)(c8 >> 2) * n + pix) * 32 + (c8 & 3) * 8) * 2;
    store_planes8(dst, plane_b, v, fmt, range_flag, true);
}

// CandidateShift's RoI features (TF_utils.py:30-39: relu(cat(corr, T2S_prev, T2S)) -> mmcv roi_align 7x7, aligned, adaptive
// sampling grid) written straight into the planes TemporalNet's first convolution reads: no concatenated feature map, no
// fp32 RoI tensor, no pad / permute / split passes.  Channel order of the planes: [T2S_prev (C1) | T2S (C1) | corr (Cc) |
// zeros] -- the two feature maps are NHWC, so a lane's 8 channels are two 16-byte loads per corner and the lanes of a
// wave read one contiguous run; the correlation volume is NCHW (strided, 19 % of the channels).  The arithmetic is
// roi_align_avg_kernel's, operation for operation (temporal.hip), with the ReLU applied to the sampled inputs.
struct RoiPlanesArgs {
    const float* t2s_prev;   // [B][H][W][C1]
    const float* t2s;        // [B][H][W][C1]
    const float* corr;       // [B][Cc][H][W], or channels-last [B][H][W][corr_ld] when corr_ld > 0
    int corr_ld;
    const float* rois;       // [n][5] = (image, x1, y1, x2, y2) in feature-map pixels
    uint8_t* planes;         // [P][Cpad/32][n*PH*PW][32]
    int n, H, W, C1, Cc, Cpad, PH, PW, fmt;
    int* range_flag;
};

__global__ __launch_bounds__(256) void roi_align_planes_kernel(const RoiPlanesArgs a)
{
    const int gpp = a.Cpad >> 3;                                   // 8-channel groups per output pixel
    const int64_t npix = (int64_t)a.n * a.PH * a.PW;
    const int64_t blk = xcd_contiguous_block((npix * gpp + 255) >> 8);
    if (blk < 0) return;
    const int64_t idx = blk * 256 + threadIdx.x;
    if (idx >= npix * gpp) return;
    const int64_t pix = idx / gpp;
    const int g = (int)(idx - pix * gpp);
    const int ri = (int)(pix / (a.PH * a.PW));
    const int pp = (int)(pix - (int64_t)ri * a.PH * a.PW);
    const int py = pp / a.PW, px = pp - py * a.PW;
    const float* roi = a.rois + 5 * ri;
    const int b = (int)roi[0];
    const float sw_ = roi[1] - 0.5f, sh_ = roi[2] - 0.5f, ew_ = roi[3] - 0.5f, eh_ = roi[4] - 0.5f;   // aligned, scale 1
    const float rw = ew_ - sw_, rh = eh_ - sh_;
    const float bh = rh / (float)a.PH, bw = rw / (float)a.PW;
    const int gh = (int)ceilf(rh / (float)a.PH), gw = (int)ceilf(rw / (float)a.PW);
    const float count = (float)max(gh * gw, 1);
    const int c0 = g * 8;
    // source of this lane's 8 channels
    const bool from_prev = c0 < a.C1, from_cur = !from_prev && c0 < 2 * a.C1;
    const float* nhwc = from_prev ? a.t2s_prev + (size_t)b * a.H * a.W * a.C1 + c0
                                  : a.t2s + (size_t)b * a.H * a.W * a.C1 + (c0 - a.C1);
    const int cc0 = c0 - 2 * a.C1;                                 // first correlation channel of the lane (NCHW source)
    const float* nchw = a.corr + ((size_t)b * a.Cc + (cc0 > 0 ? cc0 : 0)) * a.H * a.W;
    const float* cl = a.corr + (size_t)b * a.H * a.W * a.corr_ld + (cc0 > 0 ? cc0 : 0);        // channels-last source of the lane's 8 channels
    const int HW = a.H * a.W;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int iy = 0; iy < gh; ++iy) {
        const float ys = sh_ + (float)py * bh + ((float)iy + 0.5f) * bh / (float)gh;
        for (int ix = 0; ix < gw; ++ix) {
            const float xs = sw_ + (float)px * bw + ((float)ix + 0.5f) * bw / (float)gw;
            float y = ys, x = xs;
            if (y < -1.0f || y > (float)a.H || x < -1.0f || x > (float)a.W) continue;     // the sample contributes 0
            if (y <= 0.0f) y = 0.0f;
            if (x <= 0.0f) x = 0.0f;
            int y_low = (int)y, x_low = (int)x, y_high, x_high;
            if (y_low >= a.H - 1) { y_high = y_low = a.H - 1; y = (float)y_low; } else y_high = y_low + 1;
            if (x_low >= a.W - 1) { x_high = x_low = a.W - 1; x = (float)x_low; } else x_high = x_low + 1;
            const float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.0f - ly, hx = 1.0f - lx;
            const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
            const int o1 = y_low * a.W + x_low, o2 = y_low * a.W + x_high, o3 = y_high * a.W + x_low, o4 = y_high * a.W + x_high;
            float v1[8], v2[8], v3[8], v4[8];
            if (from_prev || from_cur) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 q1 = *reinterpret_cast<const f32x4*>(nhwc + (size_t)o1 * a.C1 + 4 * h);
                    const f32x4 q2 = *reinterpret_cast<const f32x4*>(nhwc + (size_t)o2 * a.C1 + 4 * h);
                    const f32x4 q3 = *reinterpret_cast<const f32x4*>(nhwc + (size_t)o3 * a.C1 + 4 * h);
                    const f32x4 q4 = *reinterpret_cast<const f32x4*>(nhwc + (size_t)o4 * a.C1 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v1[4 * h + e] = q1[e]; v2[4 * h + e] = q2[e]; v3[4 * h + e] = q3[e]; v4[4 * h + e] = q4[e]; }
                }
            } else if (a.corr_ld > 0) {
                // (channels past Cc of the padded row are whatever the buffer holds: masked, never used)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    // a 4-channel group that lies wholly in the zero padding past Cc is not loaded at all: its address may be
                    // beyond the padded row (Cpad rounds 2*C1 + Cc up to 32, corr_ld only Cc up to 8) or, on the last pixel, the buffer
                    const int hofs = cc0 + 4 * h < a.Cc ? 4 * h : -cc0;          // all-padding group: re-read channel 0 (masked below)
                    const f32x4 q1 = *reinterpret_cast<const f32x4*>(cl + (size_t)o1 * a.corr_ld + hofs);
                    const f32x4 q2 = *reinterpret_cast<const f32x4*>(cl + (size_t)o2 * a.corr_ld + hofs);
                    const f32x4 q3 = *reinterpret_cast<const f32x4*>(cl + (size_t)o3 * a.corr_ld + hofs);
                    const f32x4 q4 = *reinterpret_cast<const f32x4*>(cl + (size_t)o4 * a.corr_ld + hofs);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool real = cc0 + 4 * h + e < a.Cc;
                        v1[4 * h + e] = real ? q1[e] : 0.0f; v2[4 * h + e] = real ? q2[e] : 0.0f;
                        v3[4 * h + e] = real ? q3[e] : 0.0f; v4[4 * h + e] = real ? q4[e] : 0.0f;
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool real = cc0 + e < a.Cc;
                    const float* im = nchw + (size_t)(real ? e : 0) * HW;
                    v1[e] = real ? im[o1] : 0.0f; v2[e] = real ? im[o2] : 0.0f; v3[e] = real ? im[o3] : 0.0f; v4[e] = real ? im[o4] : 0.0f;
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e)
                acc[e] += w1 * fmaxf(v1[e], 0.0f) + w2 * fmaxf(v2[e], 0.0f) + w3 * fmaxf(v3[e], 0.0f) + w4 * fmaxf(v4[e], 0.0f);
        }
    }
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = acc[e] / count;
    const size_t plane_b = (size_t)npix * a.Cpad * 2;
    uint8_t* dst = a.planes + (((size_t)(g >> 2) * npix + pix) * 32 + (g & 3) * 8) * 2;
    store_planes8(dst, plane_b, v, a.fmt, a.range_flag, false);
}

// The same, laid out for the memory system (round 5; channels-last correlation volume only).  The kernel above gives a wave 64 channel groups of ONE
// pixel: its plane stores are sixteen 64-byte pieces in sixteen channel slabs per instruction, and nothing of what neighbouring bins share (the corners of
// adjacent bins of a RoI are the same feature pixels) is reused inside a workgroup.  Here a workgroup owns 16 consecutive output pixels (two to three rows
// of a 7 x 7 RoI grid) and every channel slab: lane = 4 pixel + chunk, wave w takes slabs w, w + 4, ... -- a plane store is 1 KB contiguous (16 pixels x
// 64 B), a corner read 16 full 128-byte lines, and the five slabs of a lane share its RoI arithmetic.  Same expressions per output value: bit-equal.
__global__ __launch_bounds__(256) void roi_align_planes_tiled_kernel(const RoiPlanesArgs a)
{
    const int64_t npix = (int64_t)a.n * a.PH * a.PW;
    const int64_t blk = xcd_contiguous_block((npix + 15) >> 4);
    if (blk < 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t pix_ = blk * 16 + (lane >> 2);
    const bool live = pix_ < npix;
    const int64_t pix = live ? pix_ : npix - 1;
    const int ck = lane & 3;
    const int ri = (int)(pix / (a.PH * a.PW));
    const int pp = (int)(pix - (int64_t)ri * a.PH * a.PW);
    const int py = pp / a.PW, px = pp - py * a.PW;
    const float* roi = a.rois + 5 * ri;
    const int b = (int)roi[0];
    const float sw_ = roi[1] - 0.5f, sh_ = roi[2] - 0.5f, ew_ = roi[3] - 0.5f, eh_ = roi[4] - 0.5f;   // aligned, scale 1
    const float rw = ew_ - sw_, rh = eh_ - sh_;
    const float bh = rh / (float)a.PH, bw = rw / (float)a.PW;
    const int gh = (int)ceilf(rh / (float)a.PH), gw = (int)ceilf(rw / (float)a.PW);
    const float count = (float)max(gh * gw, 1);
    const size_t plane_b = (size_t)npix * a.Cpad * 2;
    const int nslabs = a.Cpad >> 5;
    for (int s = wave; s < nslabs; s += 4) {
        const int g = s * 4 + ck;
        const int c0 = g * 8;
        const bool from_prev = c0 < a.C1, from_cur = !from_prev && c0 < 2 * a.C1;
        const float* nhwc = from_prev ? a.t2s_prev + (size_t)b * a.H * a.W * a.C1 + c0
                                      : a.t2s + (size_t)b * a.H * a.W * a.C1 + (c0 - a.C1);
        const int cc0 = c0 - 2 * a.C1;                             // first correlation channel of the lane
        const float* cl = a.corr + (size_t)b * a.H * a.W * a.corr_ld + (cc0 > 0 ? cc0 : 0);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int iy = 0; iy < gh; ++iy) {
            const float ys = sh_ + (float)py * bh + ((float)iy + 0.5f) * bh / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
                const float xs = sw_ + (float)px * bw + ((float)ix + 0.5f) * bw / (float)gw;
                float y = ys, x = xs;
                if (y < -1.0f || y > (float)a.H || x < -1.0f || x > (float)a.W) continue;     // the sample contributes 0
                if (y <= 0.0f) y = 0.0f;
                if (x <= 0.0f) x = 0.0f;
                int y_low = (int)y, x_low = (int)x, y_high, x_high;
                if (y_low >= a.H - 1) { y_high = y_low = a.H - 1; y = (float)y_low; } else y_high = y_low + 1;
                if (x_low >= a.W - 1) { x_high = x_low = a.W - 1; x = (float)x_low; } else x_high = x_low + 1;
                const float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.0f - ly, hx = 1.0f - lx;
                const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                const int o1 = y_low * a.W + x_low, o2 = y_low * a.W + x_high, o3 = y_high * a.W + x_low, o4 = y_high * a.W + x_high;
                float v1[8], v2[8], v3[8], v4[8];
                if (from_prev || from_cur) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x4 q1 = *reinterpret_cast<const f32x4*>(nhwc + (size_t)o1 * a.C1 + 4 * h);
                        const f32x4 q2 = *reinterpret_cast<const f32x4*>(nhwc + (size_t)o2 * a.C1 + 4 * h);
                        const f32x4 q3 = *reinterpret_cast<const f32x4*>(nhwc + (size_t)o3 * a.C1 + 4 * h);
                        const f32x4 q4 = *reinterpret_cast<const f32x4*>(nhwc + (size_t)o4 * a.C1 + 4 * h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v1[4 * h + e] = q1[e]; v2[4 * h + e] = q2[e]; v3[4 * h + e] = q3[e]; v4[4 * h + e] = q4[e]; }
                    }
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int hofs = cc0 + 4 * h < a.Cc ? 4 * h : -cc0;          // all-padding group: re-read channel 0 (masked below)
                        const f32x4 q1 = *reinterpret_cast<const f32x4*>(cl + (size_t)o1 * a.corr_ld + hofs);
                        const f32x4 q2 = *reinterpret_cast<const f32x4*>(cl + (size_t)o2 * a.corr_ld + hofs);
                        const f32x4 q3 = *reinterpret_cast<const f32x4*>(cl + (size_t)o3 * a.corr_ld + hofs);
                        const f32x4 q4 = *reinterpret_cast<const f32x4*>(cl + (size_t)o4 * a.corr_ld + hofs);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const bool real = cc0 + 4 * h + e < a.Cc;
                            v1[4 * h + e] = real ? q1[e] : 0.0f; v2[4 * h + e] = real ? q2[e] : 0.0f;
                            v3[4 * h + e] = real ? q3[e] : 0.0f; v4[4 * h + e] = real ? q4[e] : 0.0f;
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    acc[e] += w1 * fmaxf(v1[e], 0.0f) + w2 * fmaxf(v2[e], 0.0f) + w3 * fmaxf(v3[e], 0.0f) + w4 * fmaxf(v4[e], 0.0f);
            }
        }
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc[e] / count;
        if (live) {
            uint8_t* dst = a.planes + (((size_t)s * npix + pix) * 32 + ck * 8) * 2;
            store_planes8(dst, plane_b, v, a.fmt, a.range_flag, false);
        }
    }
}

// ... and with the RoI's feature patch in LDS (round 5; STM_ROI_TILED=2: built, bit-equal, NOT faster -- kept as the measured alternative).  The bins of a RoI sample the same few feature pixels over
// and over (49 bins x gh gw samples x 4 corners over a patch of a few dozen pixels): through L1 that is 7 GB of gathers per step for 82 MB of feature
// maps.  One workgroup per RoI: per 32-channel slab the patch (rows / columns any sample of the RoI can touch, ReLU applied, channels past Cc zero) is
// copied into LDS once -- 144-byte rows: consecutive pixels start 4 banks apart -- and the 49 bins x 4 chunks gather from there; the output pixels of a
// RoI are consecutive, so a slab's 49 x 64 B leave as one 3-KB run.  Same expressions in the same order per output value (relu(v) is taken when the
// patch is staged instead of at every use: the same value): bit-equal to the kernels above.  RoIs whose patch exceeds RP_MAX pixels (a tenth of the
// frame or more at 24 x 40) take the tiled kernel's direct loads inside the same launch.
constexpr int RP_MAX = 288, RP_PITCH = 144;
__global__ __launch_bounds__(256) void roi_align_planes_lds_kernel(const RoiPlanesArgs a)
{
    extern __shared__ __align__(16) uint8_t rp_smem[];
    const int64_t blk = xcd_contiguous_block(a.n);
    if (blk < 0) return;
    const int ri = (int)blk;
    const int tid = threadIdx.x;
    const int64_t npix = (int64_t)a.n * a.PH * a.PW;
    const int nb = a.PH * a.PW;                                       // bins (output pixels) of the RoI
    const float* roi = a.rois + 5 * ri;
    const int b = (int)roi[0];
    const float sw_ = roi[1] - 0.5f, sh_ = roi[2] - 0.5f, ew_ = roi[3] - 0.5f, eh_ = roi[4] - 0.5f;   // aligned, scale 1
    const float rw = ew_ - sw_, rh = eh_ - sh_;
    const float bh = rh / (float)a.PH, bw = rw / (float)a.PW;
    const int gh = (int)ceilf(rh / (float)a.PH), gw = (int)ceilf(rw / (float)a.PW);
    const float count = (float)max(gh * gw, 1);
    // feature rows / columns a sample of this RoI can read: samples lie in [sh_, eh_] x [sw_, ew_], are clamped to [0, H - 1] x [0, W - 1]
    // and read (low, low + 1)
    const int y0 = min(max((int)floorf(fminf(sh_, eh_)), 0), a.H - 1), y1 = min(max((int)floorf(fmaxf(sh_, eh_)) + 1, y0), a.H - 1);
    const int x0 = min(max((int)floorf(fminf(sw_, ew_)), 0), a.W - 1), x1 = min(max((int)floorf(fmaxf(sw_, ew_)) + 1, x0), a.W - 1);
    const int ph = y1 - y0 + 1, pw = x1 - x0 + 1;
    const bool staged = ph * pw <= RP_MAX;                            // (uniform over the workgroup)
    const int pp = tid >> 2, ck = tid & 3;                            // bin and 8-channel chunk of the compute phase
    const int py = pp / a.PW, px = pp - py * a.PW;
    const size_t plane_b = (size_t)npix * a.Cpad * 2;
    const int nslabs = a.Cpad >> 5;
    for (int s = 0; s < nslabs; ++s) {
        const int c0s = s * 32;                                       // first channel of the slab
        const bool from_prev = c0s < a.C1, from_cur = !from_prev && c0s < 2 * a.C1;
        const int cc0s = c0s - 2 * a.C1;                              // first correlation channel of the slab
        const float* src = from_prev ? a.t2s_prev + (size_t)b * a.H * a.W * a.C1 + c0s
                         : from_cur ? a.t2s + (size_t)b * a.H * a.W * a.C1 + (c0s - a.C1)
                                    : a.corr + (size_t)b * a.H * a.W * a.corr_ld + cc0s;
        const int ld = (from_prev || from_cur) ? a.C1 : a.corr_ld;
        if (staged) {
            __syncthreads();                                          // the previous slab's gathers are done
            for (int i = tid; i < ph * pw * 8; i += 256) {
                const int q = i >> 3, c4 = i & 7;                     // patch pixel, 4-channel piece
                const int qy = q / pw, qx = q - qy * pw;
                f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                const int ch = c4 * 4;
                if (from_prev || from_cur || cc0s + ch < a.Cc) {      // (a piece wholly past Cc is not loaded: it may lie beyond the padded row)
                    v = *reinterpret_cast<const f32x4*>(src + (size_t)((y0 + qy) * a.W + x0 + qx) * ld + ch);
                    if (!(from_prev || from_cur)) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (cc0s + ch + e >= a.Cc) v[e] = 0.0f;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
                *reinterpret_cast<f32x4*>(rp_smem + q * RP_PITCH + c4 * 16) = v;
            }
            __syncthreads();
        }
        if (pp < nb) {
            const int c0 = c0s + ck * 8, cc0 = c0 - 2 * a.C1;
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int iy = 0; iy < gh; ++iy) {
                const float ys = sh_ + (float)py * bh + ((float)iy + 0.5f) * bh / (float)gh;
                for (int ix = 0; ix < gw; ++ix) {
                    const float xs = sw_ + (float)px * bw + ((float)ix + 0.5f) * bw / (float)gw;
                    float y = ys, x = xs;
                    if (y < -1.0f || y > (float)a.H || x < -1.0f || x > (float)a.W) continue;     // the sample contributes 0
                    if (y <= 0.0f) y = 0.0f;
                    if (x <= 0.0f) x = 0.0f;
                    int y_low = (int)y, x_low = (int)x, y_high, x_high;
                    if (y_low >= a.H - 1) { y_high = y_low = a.H - 1; y = (float)y_low; } else y_high = y_low + 1;
                    if (x_low >= a.W - 1) { x_high = x_low = a.W - 1; x = (float)x_low; } else x_high = x_low + 1;
                    const float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.0f - ly, hx = 1.0f - lx;
                    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                    float v1[8], v2[8], v3[8], v4[8];
                    if (staged) {
                        const uint8_t* r1 = rp_smem + ((y_low - y0) * pw + x_low - x0) * RP_PITCH + ck * 32;
                        const uint8_t* r2 = rp_smem + ((y_low - y0) * pw + x_high - x0) * RP_PITCH + ck * 32;
                        const uint8_t* r3 = rp_smem + ((y_high - y0) * pw + x_low - x0) * RP_PITCH + ck * 32;
                        const uint8_t* r4 = rp_smem + ((y_high - y0) * pw + x_high - x0) * RP_PITCH + ck * 32;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const f32x4 q1 = *reinterpret_cast<const f32x4*>(r1 + 16 * h), q2 = *reinterpret_cast<const f32x4*>(r2 + 16 * h);
                            const f32x4 q3 = *reinterpret_cast<const f32x4*>(r3 + 16 * h), q4 = *reinterpret_cast<const f32x4*>(r4 + 16 * h);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v1[4 * h + e] = q1[e]; v2[4 * h + e] = q2[e]; v3[4 * h + e] = q3[e]; v4[4 * h + e] = q4[e]; }
                        }
                    } else {
                        const int o1 = y_low * a.W + x_low, o2 = y_low * a.W + x_high, o3 = y_high * a.W + x_low, o4 = y_high * a.W + x_high;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int chh = ck * 8 + 4 * h;
                            const bool skip = !(from_prev || from_cur) && cc0 + 4 * h >= a.Cc;
                            const int hofs = skip ? 0 : chh;
                            const f32x4 q1 = *reinterpret_cast<const f32x4*>(src + (size_t)o1 * ld + hofs), q2 = *reinterpret_cast<const f32x4*>(src + (size_t)o2 * ld + hofs);
                            const f32x4 q3 = *reinterpret_cast<const f32x4*>(src + (size_t)o3 * ld + hofs), q4 = *reinterpret_cast<const f32x4*>(src + (size_t)o4 * ld + hofs);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const bool real = from_prev || from_cur || cc0 + 4 * h + e < a.Cc;
                                v1[4 * h + e] = real ? fmaxf(q1[e], 0.0f) : 0.0f; v2[4 * h + e] = real ? fmaxf(q2[e], 0.0f) : 0.0f;
                                v3[4 * h + e] = real ? fmaxf(q3[e], 0.0f) : 0.0f; v4[4 * h + e] = real ? fmaxf(q4[e], 0.0f) : 0.0f;
                            }
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += w1 * v1[e] + w2 * v2[e] + w3 * v3[e] + w4 * v4[e];
                }
            }
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[e] / count;
            const int64_t pix = (int64_t)ri * nb + pp;
            uint8_t* dst = a.planes + (((size_t)s * npix + pix) * 32 + ck * 8) * 2;
            store_planes8(dst, plane_b, v, a.fmt, a.range_flag, false);
        }
    }
}

// Stem entry (backbone.py:73, the 7x7 / stride-2 convolution on the 3-channel frame): the kw * Cin = 21 values one kernel row
// reads for output column ox are contiguous in the NHWC frame, starting at column sw*ox - pw.  This kernel lays them out as
// the 32-channel slab of a planar tensor R[b][y][ox][32] (channels >= kw*Cin zero, columns outside the frame zero), which
// turns the stem into a (kh x 1) convolution with stride (sh, 1) over R on the planar kernel: K = kh * 32 = 224 for 147
// real products, no im2col buffer, no library call.  thread = 8 channels of one R pixel.
__global__ __launch_bounds__(256) void stem_rows_planes_kernel(const float* __restrict__ x, uint8_t* __restrict__ planes, int B, int H, int W,
                                                              int Cin, int kw, int sw, int pw, int Wo, int fmt, int* range_flag)
{
    const int64_t n = (int64_t)B * H * Wo;
    const int64_t blk = xcd_contiguous_block((n * 4 + 255) >> 8);
    if (blk < 0) return;
    const int64_t idx = blk * 256 + threadIdx.x;
    if (idx >= n * 4) return;
    const int64_t pix = idx >> 2;
    const int g = (int)(idx & 3);
    const int ox = (int)(pix % Wo);
    const int64_t row = pix / Wo;                                   // b * H + y
    const float* src = x + row * (int64_t)W * Cin;
    const int c0 = (sw * ox - pw) * Cin;                            // first float of the patch within the frame row
    const int lim = W * Cin, real = kw * Cin;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int j = g * 8 + e, c = c0 + j;
        v[e] = (j < real && c >= 0 && c < lim) ? src[c] : 0.0f;
    }
    const size_t plane_b = (size_t)n * 32 * 2;
    uint8_t* dst = planes + ((size_t)pix * 32 + g * 8) * 2;
    store_planes8(dst, plane_b, v, fmt, range_flag, false);
}

// Weights [Cout][Cin][kh][kw] fp32 -> packed [n_tile][slab][plane][row 0..127][swizzled 16-B chunk][8 bf16]; rows past
// Cout are zero.  One thread per (n_tile, slab, row, chunk).
__global__ __launch_bounds__(256) void conv_pack_weights_kernel(const float* __restrict__ w, uint8_t* __restrict__ wp, int Cout,
                                                                int C, int kh, int kw, int slabs, int n_tiles, int npl, int bn, int fmt,
                                                                float wscale)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int per_tile = bn * 4;                       // (row, chunk) pairs of one slab tile
    const int64_t total = (int64_t)n_tiles * slabs * per_tile;
    if (idx >= total) return;
    const int chunk = (int)(idx & 3), row = (int)((idx >> 2) % bn);
    const int slab = (int)((idx / per_tile) % slabs), nt = (int)((idx / per_tile) / slabs);
    const int taps = kh * kw;
    const int cs = slab / taps, tap = slab - cs * taps, c0 = cs * CV_BK + chunk * 8;   // K order: channel slab outer, tap inner
    const int co = nt * bn + row;
    unsigned pl[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f32x2 v = {0.0f, 0.0f};
        if (co < Cout) {
            v.x = w[((size_t)co * C + c0 + 2 * e) * (kh * kw) + tap];
            v.y = w[((size_t)co * C + c0 + 2 * e + 1) * (kh * kw) + tap];
        }
        if (fmt >= 1) { pl[2][e] = 0; split2_f16(v * wscale, pl[0][e], pl[1][e]); }
        else split2(v, pl[0][e], pl[1][e], pl[2][e]);
    }
    const int wpl = bn * 64;
    uint8_t* dst = wp + ((size_t)nt * slabs + slab) * (npl * wpl) + lds_off(row, chunk);
    for (int p = 0; p < npl; ++p) {
        u32x4 o = {pl[p][0], pl[p][1], pl[p][2], pl[p][3]};
        *reinterpret_cast<u32x4*>(dst + p * wpl) = o;
    }
}

// Launch tunables: read from the environment ONCE (first launch), never per launch.  Defaults are the measured best (DESIGN.md
// section 9); the variables exist for A/B runs.  stm_debug_reload_tunables() (capi.hip) makes the next launch re-read them.
std::atomic<long long> g_kx3_launches{0};      // stm_debug_launch_count(0)
struct ConvTunables {
    int ring = 3;          // STM_CONV_RING: 2 = two-buffer loop on the 128-wide tiles, 3 = three-buffer ring (fp16 formats)
    int ring64_small = 512; // grids up to this many workgroups take the ring on 128 x 64 tiles whatever K (round 2 sweep; a switch until round 6)
    int ring64 = 3;        // STM_CONV_RING64: 2 never / 4 always the ring on 128 x 64 tiles, 3 = by K length (rule below)
    int splitk = 0;        // STM_CONV_SPLITK: force this many K parts (0 = rule)
    int sk_rule = 0;       // 1 = round 1's split-K rule of the 64-wide tiles for every layer (kept for the record; a switch until round 6)
    int sk_target = 256;   // workgroups the split-K rule of the 64-wide tiles aims at (one per CU; a switch until round 6)
    int mg = 0;            // STM_CONV_MG: force 128 (1) or 256 (2) pixel tiles
    long long nt_mb = 0;   // nontemporal plane stores for outputs of at least this many MB (0 = off: no gain measured in rounds 2-4)
    int scalar_epilogue = 0;   // element-wise epilogue stores (a cross-check form)
    int abl = 0;           // STM_CONV_ABL (builds with -DSTM_ABLATE only)
    int nsub = 0;          // 2 / 4 = channel tiles of a pixel tile that share an XCD's L2 at one time (regrouped tile map: no effect measured in round 4)
    int kx3 = 1;           // STM_CONV_KX3: 0 = stride-1 kw = 3 layers on the 256 x 128 ring tiles stay on conv_planar_kernel (1: conv_planar_kx3_kernel, kx-reuse staging)
};
ConvTunables read_tunables()
{
    ConvTunables t;
    auto geti = [](const char* name, long long dflt) { const char* e = getenv(name); return e ? atoll(e) : dflt; };
    t.ring = (int)geti("STM_CONV_RING", t.ring);
    t.ring64 = (int)geti("STM_CONV_RING64", t.ring64);
    t.splitk = (int)geti("STM_CONV_SPLITK", 0);
    t.mg = (int)geti("STM_CONV_MG", 0);
    t.kx3 = (int)geti("STM_CONV_KX3", t.kx3);
#ifdef STM_ABLATE
    t.abl = (int)geti("STM_CONV_ABL", 0);
#endif
    return t;
}
const ConvTunables& tunables()
{
    static ConvTunables t;
    static int gen = -1;
    const int g = stm_env_generation();
    if (gen != g) { t = read_tunables(); gen = g; }
    return t;
}

template <int NPL, int MG, int NJ, int DT = 0, int ST = 2, int ABL = 0, bool DUAL = false, bool CLS = false>
int launch_planar(const typename std::conditional<CLS, PlanarArgsCls, PlanarArgs>::type& a, int tiles, stm_stream_t stream)
{
    size_t lds = (size_t)ST * (NPL * CV_BM * MG * 64 + NPL * (64 * NJ) * 64);
    const size_t park = (size_t)4 * MG * 64 * (32 * NJ + 4) * sizeof(float);   // the epilogue parks one 64 x 32NJ tile per wave
    if (lds < park) lds = park;
    // per instantiation AND per device (the attribute belongs to the device's copy of the function); it is sticky, so it is set once --
    // setting it at each launch only costs host time.  Relaxed atomics: two host threads racing here both set the same value.
    static std::atomic<bool> lds_reserved[STM_MAX_DEVICES];
    int dev = 0;
    const bool have_dev = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < STM_MAX_DEVICES;
    if (!have_dev || !lds_reserved[dev].load(std::memory_order_relaxed)) {
        STM_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_planar_kernel<NPL, MG, NJ, DT, ST, ABL, DUAL, CLS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds) == hipSuccess, STM_ELAUNCH, "stm_conv2d_planar_f32: cannot reserve %zu bytes of LDS", lds);
        if (have_dev) lds_reserved[dev].store(true, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((conv_planar_kernel<NPL, MG, NJ, DT, ST, ABL, DUAL, CLS>), dim3(8 * stm_cdiv(tiles, 8)), dim3(256 * MG), lds, stm_hs(stream), a);
    STM_CHECK_LAUNCH("conv_planar_kernel");
    return STM_OK;
}

bool geom_ok(const stm_conv_geom* g, const char* who)
{
    if (!g) { stm_set_error("%s: geometry is NULL", who); return false; }
    if (g->B <= 0 || g->H <= 0 || g->W <= 0 || g->C <= 0 || g->Cout <= 0 || g->kh <= 0 || g->kw <= 0 || g->sh <= 0 ||
        g->sw <= 0 || g->Ho <= 0 || g->Wo <= 0 || (g->win_w <= 0 && (g->ph < 0 || g->pw < 0))) {
        stm_set_error("%s: bad geometry", who);
        return false;
    }
    if (g->C % CV_BK != 0) { stm_set_error("%s: input channels (%d) must be a multiple of 32", who, g->C); return false; }
    if (g->win_w > 0) {
        // window launch: Ho x Wo outputs per image starting at (win_y0, win_x0) of the win_h x win_w output image; ph / pw may be negative
        if (g->win_h <= 0 || g->win_y0 < 0 || g->win_x0 < 0 || g->win_y0 + g->Ho > g->win_h || g->win_x0 + g->Wo > g->win_w || g->n_levels > 0) {
            stm_set_error("%s: the %dx%d window at (%d, %d) leaves the %dx%d output image", who, g->Ho, g->Wo, g->win_y0, g->win_x0, g->win_h, g->win_w);
            return false;
        }
    } else if (g->Ho != (g->H + 2 * g->ph - g->kh) / g->sh + 1 || g->Wo != (g->W + 2 * g->pw - g->kw) / g->sw + 1) {
        stm_set_error("%s: Ho/Wo do not match the convolution arithmetic", who);
        return false;
    }
    if (g->planes < 1 || g->planes > 3) { stm_set_error("%s: planes must be 1, 2 or 3", who); return false; }
    return true;
}

}  // namespace

int* stm_internal_range_flag() { return current_range_flag(); }
extern "C" int stm_planar_set_range_flag(int* device_flag)
{
    int dev = 0;
    STM_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < STM_MAX_DEVICES, STM_EINVAL,
                "stm_planar_set_range_flag: no current device (or more than %d devices)", STM_MAX_DEVICES);
    g_range_flags[dev] = device_flag;       // the flag of the CURRENT device: kernels launched on device d raise flag d
    return STM_OK;
}

extern "C" size_t stm_conv_packed_weight_bytes_tiled(int Cout, int Cin, int kh, int kw, int planes, int tile_n)
{
    if (Cout <= 0 || Cin <= 0 || Cin % CV_BK || kh <= 0 || kw <= 0 || (planes < 1 || planes > 3) || (tile_n != 64 && tile_n != 128))
        return 0;
    return (size_t)stm_cdiv(Cout, tile_n) * (kh * kw * (Cin / CV_BK)) * planes * (tile_n * 64);
}

extern "C" size_t stm_conv_packed_weight_bytes(int Cout, int Cin, int kh, int kw, int planes)
{
    return stm_conv_packed_weight_bytes_tiled(Cout, Cin, kh, kw, planes, CV_BN);
}

extern "C" int stm_conv_pack_weights_fmt_f32(const float* weight, void* packed, int Cout, int Cin, int kh, int kw, int tile_n, int fmt,
                                             float wscale, stm_stream_t stream);

extern "C" int stm_conv_pack_weights_tiled_f32(const float* weight, void* packed, int Cout, int Cin, int kh, int kw, int planes,
                                               int tile_n, stm_stream_t stream)
{
    if (planes == 3) return stm_conv_pack_weights_fmt_f32(weight, packed, Cout, Cin, kh, kw, tile_n, 0, 1.0f, stream);
    STM_REQUIRE(weight && packed, STM_ENULL, "stm_conv_pack_weights_f32: weight/packed must be non-NULL");
    STM_REQUIRE(stm_conv_packed_weight_bytes_tiled(Cout, Cin, kh, kw, planes, tile_n) > 0, STM_EINVAL,
                "stm_conv_pack_weights_f32: bad sizes Cout=%d Cin=%d (multiple of 32) k=%dx%d planes=%d tile_n=%d (64 or 128)", Cout,
                Cin, kh, kw, planes, tile_n);
    STM_REQUIRE((uintptr_t)packed % 16 == 0, STM_EINVAL, "stm_conv_pack_weights_f32: packed buffer must be 16-byte aligned");
    const int slabs = kh * kw * (Cin / CV_BK), n_tiles = stm_cdiv(Cout, tile_n);
    const int64_t total = (int64_t)n_tiles * slabs * tile_n * 4;
    hipLaunchKernelGGL(conv_pack_weights_kernel, dim3(stm_cdiv(total, 256)), dim3(256), 0, stm_hs(stream), weight,
                       static_cast<uint8_t*>(packed), Cout, Cin, kh, kw, slabs, n_tiles, planes, tile_n, 0, 1.0f);
    STM_CHECK_LAUNCH("conv_pack_weights_kernel");
    return STM_OK;
}

extern "C" int stm_conv_pack_weights_fmt_f32(const float* weight, void* packed, int Cout, int Cin, int kh, int kw, int tile_n, int fmt,
                                             float wscale, stm_stream_t stream)
{
    STM_REQUIRE(weight && packed, STM_ENULL, "stm_conv_pack_weights_fmt_f32: weight/packed must be non-NULL");
    STM_REQUIRE(fmt >= 0 && fmt <= 2, STM_EINVAL, "stm_conv_pack_weights_fmt_f32: fmt must be 0 (bf16 x 3), 1 (fp16 x 2) or 2 (fp16 x 1)");
    const int planes = fmt == 1 ? 2 : (fmt == 2 ? 1 : 3);
    STM_REQUIRE(stm_conv_packed_weight_bytes_tiled(Cout, Cin, kh, kw, planes, tile_n) > 0, STM_EINVAL,
                "stm_conv_pack_weights_fmt_f32: bad sizes Cout=%d Cin=%d (multiple of 32) k=%dx%d tile_n=%d (64 or 128)", Cout, Cin, kh, kw,
                tile_n);
    STM_REQUIRE((uintptr_t)packed % 16 == 0, STM_EINVAL, "stm_conv_pack_weights_fmt_f32: packed buffer must be 16-byte aligned");
    STM_REQUIRE(fmt == 0 || (wscale > 0.0f && wscale < 3.0e38f), STM_EINVAL, "stm_conv_pack_weights_fmt_f32: bad weight scale");
    const int slabs = kh * kw * (Cin / CV_BK), n_tiles = stm_cdiv(Cout, tile_n);
    const int64_t total = (int64_t)n_tiles * slabs * tile_n * 4;
    hipLaunchKernelGGL(conv_pack_weights_kernel, dim3(stm_cdiv(total, 256)), dim3(256), 0, stm_hs(stream), weight,
                       static_cast<uint8_t*>(packed), Cout, Cin, kh, kw, slabs, n_tiles, planes, tile_n, fmt, fmt >= 1 ? wscale : 1.0f);
    STM_CHECK_LAUNCH("conv_pack_weights_kernel");
    return STM_OK;
}

extern "C" int stm_conv_pack_weights_f32(const float* weight, void* packed, int Cout, int Cin, int kh, int kw, int planes,
                                         stm_stream_t stream)
{
    return stm_conv_pack_weights_tiled_f32(weight, packed, Cout, Cin, kh, kw, planes, CV_BN, stream);
}

extern "C" int stm_split_planes_fmt_f32(const float* x, void* planes, int64_t n_pixels, int C, int fmt, stm_stream_t stream);
extern "C" int stm_split_bf16_planes_f32(const float* x, void* planes, int64_t n_pixels, int C, stm_stream_t stream)
{
    return stm_split_planes_fmt_f32(x, planes, n_pixels, C, 0, stream);
}

extern "C" int stm_split_planes_fmt_f32(const float* x, void* planes, int64_t n_pixels, int C, int fmt, stm_stream_t stream)
{
    STM_REQUIRE(fmt >= 0 && fmt <= 2, STM_EINVAL, "stm_split_planes_fmt_f32: fmt must be 0 (bf16 x 3), 1 (fp16 x 2) or 2 (fp16 x 1)");
    STM_REQUIRE(x && planes, STM_ENULL, "stm_split_bf16_planes_f32: x/planes must be non-NULL");
    STM_REQUIRE(n_pixels > 0 && C > 0 && C % 32 == 0, STM_EINVAL, "stm_split_bf16_planes_f32: n_pixels (%lld) > 0 and C (%d) a multiple of 32",
                (long long)n_pixels, C);
    STM_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)planes % 16 == 0, STM_EINVAL, "stm_split_bf16_planes_f32: 16-byte alignment required");
    hipLaunchKernelGGL(split_planes_kernel, dim3(stm_cdiv(n_pixels * (C / 8), 256)), dim3(256), 0, stm_hs(stream), x,
                       static_cast<uint8_t*>(planes), n_pixels, C, fmt, current_range_flag());
    STM_CHECK_LAUNCH("split_planes_kernel");
    return STM_OK;
}

extern "C" int stm_resize_bilinear_planes_f32(const float* x, void* planes, int B, int H, int W, int C, int Ho, int Wo, int fmt,
                                              stm_stream_t stream)
{
    STM_REQUIRE(fmt >= 0 && fmt <= 2, STM_EINVAL, "stm_resize_bilinear_planes_f32: fmt must be 0, 1 or 2");
    STM_REQUIRE(x && planes, STM_ENULL, "stm_resize_bilinear_planes_f32: x/planes must be non-NULL");
    STM_REQUIRE(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 32 == 0, STM_EINVAL,
                "stm_resize_bilinear_planes_f32: sizes must be positive and C (%d) a multiple of 32", C);
    STM_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)planes % 16 == 0, STM_EINVAL, "stm_resize_bilinear_planes_f32: 16-byte alignment required");
    const int64_t n = (int64_t)B * Ho * Wo;
    // scale as ATen computes it for align_corners=false without an explicit scale factor: input size / output size
    const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
    hipLaunchKernelGGL(resize_bilinear_planes_kernel, dim3(8 * stm_cdiv(stm_cdiv(n * (C / 8), 256), 8)), dim3(256), 0, stm_hs(stream), x,
                       static_cast<uint8_t*>(planes), B, H, W, C, Ho, Wo, sy, sx, fmt, current_range_flag());
    STM_CHECK_LAUNCH("resize_bilinear_planes_kernel");
    return STM_OK;
}

extern "C" int stm_bias_relu_maxpool_planes_f32(const float* x, const float* bias, void* planes, int B, int H, int W, int C, int fmt,
                                                stm_stream_t stream)
{
    STM_REQUIRE(fmt >= 0 && fmt <= 2, STM_EINVAL, "stm_bias_relu_maxpool_planes_f32: fmt must be 0, 1 or 2");
    STM_REQUIRE(x && planes, STM_ENULL, "stm_bias_relu_maxpool_planes_f32: x/planes must be non-NULL");
    STM_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 32 == 0, STM_EINVAL,
                "stm_bias_relu_maxpool_planes_f32: sizes must be positive and C (%d) a multiple of 32", C);
    STM_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)planes % 16 == 0, STM_EINVAL, "stm_bias_relu_maxpool_planes_f32: 16-byte alignment required");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;     // MaxPool2d(kernel 3, stride 2, padding 1), floor mode
    const int64_t n = (int64_t)B * Ho * Wo;
    hipLaunchKernelGGL(bias_relu_maxpool_planes_kernel, dim3(8 * stm_cdiv(stm_cdiv(n * (C / 8), 256), 8)), dim3(256), 0, stm_hs(stream), x, bias,
                       static_cast<uint8_t*>(planes), B, H, W, C, Ho, Wo, fmt, current_range_flag());
    STM_CHECK_LAUNCH("bias_relu_maxpool_planes_kernel");
    return STM_OK;
}

extern "C" int stm_roi_align_planes_nhwc_f32(const float* t2s_prev, const float* t2s, const float* corr, int corr_ld, const float* rois,
                                             void* planes, int B, int H, int W, int C1, int Cc, int n, int PH, int PW, int fmt, stm_stream_t stream);
extern "C" int stm_roi_align_planes_f32(const float* t2s_prev, const float* t2s, const float* corr, const float* rois, void* planes, int B,
                                        int H, int W, int C1, int Cc, int n, int PH, int PW, int fmt, stm_stream_t stream)
{
    return stm_roi_align_planes_nhwc_f32(t2s_prev, t2s, corr, 0, rois, planes, B, H, W, C1, Cc, n, PH, PW, fmt, stream);
}

extern "C" int stm_roi_align_planes_nhwc_f32(const float* t2s_prev, const float* t2s, const float* corr, int corr_ld, const float* rois,
                                             void* planes, int B, int H, int W, int C1, int Cc, int n, int PH, int PW, int fmt, stm_stream_t stream)
{
    STM_REQUIRE(corr_ld == 0 || (corr_ld >= (Cc + 7) / 8 * 8 && corr_ld % 4 == 0 && (uintptr_t)corr % 16 == 0), STM_EINVAL,
                "stm_roi_align_planes_nhwc_f32: corr_ld must be 0 (NCHW) or a multiple of 4 >= Cc rounded up to 8, corr 16-byte aligned");
    STM_REQUIRE(fmt >= 0 && fmt <= 2, STM_EINVAL, "stm_roi_align_planes_f32: fmt must be 0, 1 or 2");
    STM_REQUIRE(t2s_prev && t2s && corr && rois && planes, STM_ENULL, "stm_roi_align_planes_f32: NULL argument");
    STM_REQUIRE(B > 0 && H > 0 && W > 0 && C1 > 0 && C1 % 8 == 0 && Cc > 0 && n > 0 && PH > 0 && PW > 0, STM_EINVAL,
                "stm_roi_align_planes_f32: bad sizes (C1 = %d must be a multiple of 8)", C1);
    STM_REQUIRE((uintptr_t)t2s_prev % 16 == 0 && (uintptr_t)t2s % 16 == 0 && (uintptr_t)planes % 16 == 0, STM_EINVAL,
                "stm_roi_align_planes_f32: 16-byte alignment required");
    RoiPlanesArgs a;
    a.t2s_prev = t2s_prev; a.t2s = t2s; a.corr = corr; a.corr_ld = corr_ld; a.rois = rois; a.planes = static_cast<uint8_t*>(planes);
    a.n = n; a.H = H; a.W = W; a.C1 = C1; a.Cc = Cc; a.Cpad = (2 * C1 + Cc + 31) / 32 * 32; a.PH = PH; a.PW = PW; a.fmt = fmt;
    a.range_flag = current_range_flag();
    const int64_t threads = (int64_t)n * PH * PW * (a.Cpad / 8);
    // STM_ROI_TILED: 0 the first kernel (one pixel's channel groups per wave), 1 (default) the tiled kernel, 2 the RoI's patch in LDS -- bit-equal,
    // and slower: 399 vs 272 us at 32 clips, 149 vs 80 at 8 (twenty slabs of stage / barrier / gather / barrier per workgroup, 196 of 256 lanes at work)
    const int roi_form = STM_ENV_INT("STM_ROI_TILED", 1);
    if (corr_ld > 0 && roi_form == 2 && PH * PW * 4 <= 256 && C1 % 32 == 0 && corr_ld % 32 == 0 && a.Cpad == 2 * C1 + corr_ld) {
        hipLaunchKernelGGL(roi_align_planes_lds_kernel, dim3(8 * stm_cdiv(n, 8)), dim3(256), (size_t)RP_MAX * RP_PITCH, stm_hs(stream), a);
        STM_CHECK_LAUNCH("roi_align_planes_lds_kernel");
        return STM_OK;
    }
    if (corr_ld > 0 && roi_form) {
        const int64_t tiles = ((int64_t)n * PH * PW + 15) >> 4;
        hipLaunchKernelGGL(roi_align_planes_tiled_kernel, dim3(8 * stm_cdiv(tiles, 8)), dim3(256), 0, stm_hs(stream), a);
        STM_CHECK_LAUNCH("roi_align_planes_tiled_kernel");
        return STM_OK;
    }
    hipLaunchKernelGGL(roi_align_planes_kernel, dim3(8 * stm_cdiv(stm_cdiv(threads, 256), 8)), dim3(256), 0, stm_hs(stream), a);
    STM_CHECK_LAUNCH("roi_align_planes_kernel");
    return STM_OK;
}

extern "C" int stm_stem_rows_planes_f32(const float* x, void* planes, int B, int H, int W, int Cin, int kw, int sw, int pw, int fmt,
                                        stm_stream_t stream)
{
    STM_REQUIRE(fmt >= 0 && fmt <= 2, STM_EINVAL, "stm_stem_rows_planes_f32: fmt must be 0, 1 or 2");
    STM_REQUIRE(x && planes, STM_ENULL, "stm_stem_rows_planes_f32: x/planes must be non-NULL");
    STM_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && kw > 0 && sw > 0 && pw >= 0 && kw * Cin <= 32, STM_EINVAL,
                "stm_stem_rows_planes_f32: sizes must be positive and kw * Cin (%d) at most 32", kw * Cin);
    STM_REQUIRE((uintptr_t)planes % 16 == 0, STM_EINVAL, "stm_stem_rows_planes_f32: 16-byte alignment required");
    const int Wo = (W + 2 * pw - kw) / sw + 1;
    STM_REQUIRE(Wo > 0 && (int64_t)H * W * Cin < ((int64_t)1 << 31), STM_EINVAL, "stm_stem_rows_planes_f32: bad geometry");
    const int64_t n = (int64_t)B * H * Wo;
    hipLaunchKernelGGL(stem_rows_planes_kernel, dim3(8 * stm_cdiv(stm_cdiv(n * 4, 256), 8)), dim3(256), 0, stm_hs(stream), x,
                       static_cast<uint8_t*>(planes), B, H, W, Cin, kw, sw, pw, Wo, fmt, current_range_flag());
    STM_CHECK_LAUNCH("stem_rows_planes_kernel");
    return STM_OK;
}

namespace {
struct WinSet { const void* const* packed; const stm_conv_window* win; int n; unsigned long long* pool; };
struct DualSrc { const void* x2; int C2, H2, W2, s2; long long x2_np, x2_plane_stride; };
int conv2d_planar_impl(const void* x_planes, const void* packed_weight, const float* bias, const float* residual_f32,
                       const void* residual_planes, float* out_f32, void* out_planes, const stm_conv_geom* g,
                       int relu, void* workspace, size_t workspace_bytes, stm_stream_t stream, const DualSrc* dual, const WinSet* wset);
}  // namespace

extern "C" int stm_conv2d_planar_ws_f32(const void* x_planes, const void* packed_weight, const float* bias, const float* residual_f32,
                                        const void* residual_planes, float* out_f32, void* out_planes, const stm_conv_geom* g,
                                        int relu, void* workspace, size_t workspace_bytes, stm_stream_t stream)
{
    return conv2d_planar_impl(x_planes, packed_weight, bias, residual_f32, residual_planes, out_f32, out_planes, g, relu, workspace,
                              workspace_bytes, stream, nullptr, nullptr);
}

// Two-source 1x1 convolution: y = W [x1 ; x2(stride s2)] + bias (+ residual) -- the last 1x1 convolution of a ResNet stage's first
// bottleneck and its projection shortcut (backbone.py:38-58: out = conv3(...); out += downsample(x); relu) as ONE product over the
// concatenated channels.  g describes the output (B, Ho, Wo = H, W of the first source; kh = kw = 1, stride 1, no padding; C = C1 + C2
// = the weight's input channels); x_planes holds the first C1 = g->C - C2 channels at one pixel per output pixel (x_np / x_plane_stride
// of g), x2_planes the other C2 channels as B images of H2 x W2 read at stride s2 (Ho = (H2 - 1) / s2 + 1).  The projection's output
// tensor -- 4 * planes channels written, then read back as conv3's residual: 1 GB per launch pair in layer1 at batch 32 -- never exists.
extern "C" int stm_conv2d_planar_dual_f32(const void* x_planes, const void* x2_planes, int C2, int H2, int W2, int s2, long long x2_np,
                                          long long x2_plane_stride, const void* packed_weight, const float* bias, const float* residual_f32,
                                          const void* residual_planes, float* out_f32, void* out_planes, const stm_conv_geom* g, int relu,
                                          void* workspace, size_t workspace_bytes, stm_stream_t stream)
{
    STM_REQUIRE(x2_planes && g, STM_ENULL, "stm_conv2d_planar_dual_f32: x2_planes / geometry must be non-NULL");
    DualSrc d{x2_planes, C2, H2, W2, s2, x2_np, x2_plane_stride};
    return conv2d_planar_impl(x_planes, packed_weight, bias, residual_f32, residual_planes, out_f32, out_planes, g, relu, workspace,
                              workspace_bytes, stream, &d, nullptr);
}

// Several window launches (stm_conv_geom.win_*) of ONE layer as one grid: window i has its own sub-kernel (kh x kw taps, packed weights
// packed[i], all under the same weight scale) and its own Ho x Wo rectangle at (y0, x0) of every win_h x win_w output image.  g gives what
// the windows share: B, H, W, C, Cout, sh = sw = 1, fmt 1, planes, tile_n 128, out_scale, win_h / win_w, the buffer geometry.  The tiles of
// all windows form one grid, so the small windows (a row or a corner of a 7x7 RoI map) do not pay a launch and a partial last round each:
// TemporalNet's 3x3 layers run their nine border classes -- the taps that can be inside the map, 361 of 441 tap-pixels -- at the
// efficiency of the single padded launch.  Same sums as that launch (a skipped tap added exact zeros): bit-equal.
extern "C" int stm_conv2d_planar_windows_f32(const void* x_planes, const void* const* packed_weights, const stm_conv_window* windows, int n_windows,
                                             const float* bias, float* out_f32, void* out_planes, const stm_conv_geom* g, int relu, stm_stream_t stream)
{
    const char* who = "stm_conv2d_planar_windows_f32";
    STM_REQUIRE(packed_weights && windows && g, STM_ENULL, "%s: NULL argument", who);
    STM_REQUIRE(n_windows >= 1 && n_windows <= 9, STM_EINVAL, "%s: 1 .. 9 windows", who);
    STM_REQUIRE(g->win_w > 0 && g->win_h > 0 && g->fmt == 1 && g->sh == 1 && g->sw == 1 && (g->groups <= 1) && g->n_levels <= 0 &&
                (g->tile_n == 0 || g->tile_n == 128), STM_EUNSUPPORTED, "%s: fp16x2 planes, stride 1, one group, 128-channel tiles, win_h / win_w set", who);
    for (int i = 0; i < n_windows; ++i) STM_REQUIRE(packed_weights[i], STM_ENULL, "%s: packed weight %d is NULL", who, i);
    stm_conv_geom g0 = *g;                       // window 0 stands in for the common checks and fields
    g0.kh = windows[0].kh; g0.kw = windows[0].kw; g0.ph = windows[0].ph; g0.pw = windows[0].pw;
    g0.Ho = windows[0].Ho; g0.Wo = windows[0].Wo; g0.win_y0 = windows[0].y0; g0.win_x0 = windows[0].x0;
    WinSet ws{packed_weights, windows, n_windows, nullptr};
    return conv2d_planar_impl(x_planes, packed_weights[0], bias, nullptr, nullptr, out_f32, out_planes, &g0, relu, nullptr, 0, stream, nullptr, &ws);
}

// The same window set with ReLU and the average pool over each output image folded into the epilogue (TemporalNet's conv3 + ReLU + AvgPool2d((7, 7)):
// track_to_segment_head.py:30-33): nothing is written per pixel; pool_fix[image][channel] (row length Cout, unsigned 64-bit, 32.32 fixed point)
// receives the SUM over the image's win_h x win_w pixels of relu(conv + bias), added to what it holds -- the caller zeroes it (stm_temporal_pool_fc_f32
// does, when it consumes it).  Integer accumulation: the result does not depend on the order in which the workgroups finish.  Sums must stay below
// 2^32 (the per-device range flag is raised otherwise, as for an fp16 overflow).
extern "C" int stm_conv2d_planar_windows_pool_f32(const void* x_planes, const void* const* packed_weights, const stm_conv_window* windows, int n_windows,
                                                  const float* bias, unsigned long long* pool_fix, const stm_conv_geom* g, stm_stream_t stream)
{
    const char* who = "stm_conv2d_planar_windows_pool_f32";
    STM_REQUIRE(packed_weights && windows && g && pool_fix, STM_ENULL, "%s: NULL argument", who);
    STM_REQUIRE(n_windows >= 1 && n_windows <= 9, STM_EINVAL, "%s: 1 .. 9 windows", who);
    STM_REQUIRE(g->win_w > 0 && g->win_h > 0 && g->fmt == 1 && g->sh == 1 && g->sw == 1 && (g->groups <= 1) && g->n_levels <= 0 &&
                (g->tile_n == 0 || g->tile_n == 128) && g->Cout % 128 == 0, STM_EUNSUPPORTED,
                "%s: fp16x2 planes, stride 1, one group, whole 128-channel tiles, win_h / win_w set", who);
    STM_REQUIRE((uintptr_t)pool_fix % 8 == 0, STM_EINVAL, "%s: pool_fix must be 8-byte aligned", who);
    for (int i = 0; i < n_windows; ++i) STM_REQUIRE(packed_weights[i], STM_ENULL, "%s: packed weight %d is NULL", who, i);
    stm_conv_geom g0 = *g;
    g0.kh = windows[0].kh; g0.kw = windows[0].kw; g0.ph = windows[0].ph; g0.pw = windows[0].pw;
    g0.Ho = windows[0].Ho; g0.Wo = windows[0].Wo; g0.win_y0 = windows[0].y0; g0.win_x0 = windows[0].x0;
    WinSet ws{packed_weights, windows, n_windows, pool_fix};
    return conv2d_planar_impl(x_planes, packed_weights[0], bias, nullptr, nullptr, nullptr, nullptr, &g0, 1, nullptr, 0, stream, nullptr, &ws);
}

namespace {
int conv2d_planar_impl(const void* x_planes, const void* packed_weight, const float* bias, const float* residual_f32,
                       const void* residual_planes, float* out_f32, void* out_planes, const stm_conv_geom* g,
                       int relu, void* workspace, size_t workspace_bytes, stm_stream_t stream, const DualSrc* dual, const WinSet* wset)
{
    const char* who = dual ? "stm_conv2d_planar_dual_f32" : "stm_conv2d_planar_f32";
    STM_REQUIRE(x_planes && packed_weight && (out_f32 || out_planes || (wset && wset->pool)), STM_ENULL,
                "%s: x_planes/packed_weight and at least one output must be non-NULL", who);
    STM_REQUIRE(g, STM_ENULL, "%s: geometry is NULL", who);
    const int groups = g->groups > 0 ? g->groups : 1;
    STM_REQUIRE(g->C > 0 && g->C % CV_BK == 0 && g->Cout > 0 && g->Cout % groups == 0 && g->kh > 0 && g->kw > 0 &&
                (g->planes >= 1 && g->planes <= 3), STM_EINVAL, "%s: bad channel / kernel / planes arguments", who);
    const int cout_g = g->Cout / groups;
    const int bn = g->tile_n ? g->tile_n : CV_BN;      // must be the tile width the weights were packed for
    STM_REQUIRE(bn == 64 || bn == 128, STM_EINVAL, "%s: tile_n must be 64 or 128", who);
    STM_REQUIRE(groups == 1 || cout_g % bn == 0, STM_EINVAL, "%s: Cout per group (%d) must be a multiple of the tile width %d", who, cout_g, bn);
    int64_t M, in_pixels;
    if (g->n_levels > 0) {
        STM_REQUIRE(g->n_levels <= 8 && g->sh == 1 && g->sw == 1 && 2 * g->ph == g->kh - 1 && 2 * g->pw == g->kw - 1, STM_EINVAL,
                    "%s: multi-level launches need stride 1 and same padding (<= 8 levels)", who);
        STM_REQUIRE(g->lvl_start[0] == 0, STM_EINVAL, "%s: lvl_start[0] must be 0", who);
        for (int l = 0; l < g->n_levels; ++l) {
            const int n = g->lvl_start[l + 1] - g->lvl_start[l];
            STM_REQUIRE(g->lvl_h[l] > 0 && g->lvl_w[l] > 0 && n > 0 && n % (g->lvl_h[l] * g->lvl_w[l]) == 0, STM_EINVAL,
                        "%s: level %d: %d pixels is not a whole number of %dx%d images", who, l, n, g->lvl_h[l], g->lvl_w[l]);
        }
        M = in_pixels = g->lvl_start[g->n_levels];
    } else {
        if (!geom_ok(g, who)) return STM_EINVAL;
        M = (int64_t)g->B * g->Ho * g->Wo;
        in_pixels = (int64_t)g->B * g->H * g->W;
    }
    // fp32 tensors: [pixels][ld]; planar buffers: [plane][channel slab][np pixels][32]
    const int out_ld = g->out_ld ? g->out_ld : g->Cout, res_ld = g->res_ld ? g->res_ld : g->Cout;
    STM_REQUIRE(out_ld >= g->Cout && res_ld >= g->Cout, STM_EINVAL, "%s: bad leading dimensions out_ld=%d res_ld=%d", who, out_ld, res_ld);
    const bool win = g->n_levels <= 0 && g->win_w > 0;
    const int64_t out_rows = win ? (int64_t)g->B * g->win_h * g->win_w : M;       // rows of the output tensors
    const int64_t x_np = g->x_np ? g->x_np : in_pixels, out_np = g->out_np ? g->out_np : out_rows, res_np = g->res_np ? g->res_np : M;
    STM_REQUIRE(!win || (!residual_f32 && !residual_planes && !dual), STM_EUNSUPPORTED, "%s: window launches take no residual / second source", who);
    STM_REQUIRE(x_np >= in_pixels && out_np >= out_rows && res_np >= M, STM_EINVAL, "%s: x_np / out_np / res_np smaller than the pixel count", who);
    STM_REQUIRE((uintptr_t)x_planes % 16 == 0 && (uintptr_t)packed_weight % 16 == 0, STM_EINVAL,
                "%s: x_planes and packed_weight must be 16-byte aligned", who);
    if (dual) {
        STM_REQUIRE(groups == 1 && g->n_levels <= 0 && g->kh == 1 && g->kw == 1 && g->sh == 1 && g->sw == 1 && g->ph == 0 && g->pw == 0 &&
                    g->H == g->Ho && g->W == g->Wo && (g->fmt == 1 || g->fmt == 2), STM_EUNSUPPORTED,
                    "%s: 1x1 / stride 1 / one group / fp16 plane formats only", who);
        STM_REQUIRE(dual->C2 > 0 && dual->C2 % CV_BK == 0 && dual->C2 < g->C && dual->s2 >= 1 && dual->H2 > 0 && dual->W2 > 0 &&
                    (dual->H2 - 1) / dual->s2 + 1 == g->Ho && (dual->W2 - 1) / dual->s2 + 1 == g->Wo, STM_EINVAL,
                    "%s: second source %d channels of %dx%d at stride %d does not match the %dx%d output", who, dual->C2, dual->H2, dual->W2, dual->s2,
                    g->Ho, g->Wo);
    }
    const int64_t x_slabs = dual ? (int64_t)(g->C - dual->C2) / CV_BK : (int64_t)groups * (g->C / CV_BK);
    const int64_t plane_bytes = x_slabs * x_np * 64;          // the channel slabs this launch may address, from x_planes
    STM_REQUIRE(plane_bytes < ((int64_t)1 << 31), STM_EUNSUPPORTED, "%s: plane larger than 2 GiB", who);
    STM_REQUIRE(M < ((int64_t)1 << 30), STM_EUNSUPPORTED, "%s: too many output pixels", who);
    const int64_t o_slabs = stm_cdiv(g->Cout, 32);
    const int64_t xps = g->x_plane_stride ? g->x_plane_stride : x_slabs * x_np * 32;
    const int64_t ops = g->out_plane_stride ? g->out_plane_stride : o_slabs * out_np * 32;
    const int64_t rps = g->res_plane_stride ? g->res_plane_stride : o_slabs * res_np * 32;
    STM_REQUIRE(xps % 8 == 0, STM_EINVAL, "%s: x_plane_stride must be a multiple of 8 elements", who);
    const int x_ld = 0;
    PlanarArgs a;
    a.xp = static_cast<const uint8_t*>(x_planes); a.wp = static_cast<const uint8_t*>(packed_weight); a.bias = bias;
    a.res_f32 = residual_f32; a.res_pl = static_cast<const uint8_t*>(residual_planes);
    a.out_f32 = out_f32; a.out_pl = static_cast<uint8_t*>(out_planes);
    a.B = g->B; a.H = g->H; a.W = g->W; a.C = g->C; a.Ho = g->Ho; a.Wo = g->Wo; a.Cout = g->Cout;
    a.kh = g->kh; a.kw = g->kw; a.sh = g->sh; a.sw = g->sw; a.ph = g->ph; a.pw = g->pw;
    a.x_ld = x_ld; a.out_ld = out_ld; a.res_ld = res_ld; a.relu = relu;
    a.x_np = (int)x_np; a.out_np = (int)out_np; a.res_np = (int)res_np;
    a.M = (int)M; a.n_tiles = stm_cdiv(g->Cout, bn); a.slabs = g->kh * g->kw * (g->C / CV_BK);
    a.nsub = 0;
    a.plane_bytes = (unsigned)plane_bytes;
    a.x_pstride = xps * 2; a.out_pstride = ops * 2; a.res_pstride = rps * 2;
    a.groups = groups; a.cout_g = cout_g; a.ntpg = stm_cdiv(cout_g, bn);
    for (int i = 0; i < 8; ++i) a.group_real[i] = (i < groups && g->group_cout[i] > 0 && g->group_cout[i] < cout_g) ? g->group_cout[i] : cout_g;
    a.n_levels = g->n_levels > 0 ? g->n_levels : 0;
    for (int l = 0; l < 8; ++l) { a.lvl_start[l] = g->lvl_start[l]; a.lvl_h[l] = g->lvl_h[l]; a.lvl_w[l] = g->lvl_w[l]; }
    a.lvl_start[8] = g->lvl_start[8];
    const ConvTunables& tn = tunables();
    a.vec_epilogue = (cout_g % 8 == 0) && (!out_f32 || out_ld % 4 == 0) && (!residual_f32 || res_ld % 4 == 0) && ((uintptr_t)out_f32 % 16 == 0) &&
                     ((uintptr_t)out_planes % 16 == 0) && ((uintptr_t)residual_f32 % 16 == 0) && ((uintptr_t)residual_planes % 16 == 0) &&
                     (ops % 8 == 0) && (rps % 8 == 0) && !tn.scalar_epilogue;
    STM_REQUIRE(g->kh * g->kw <= 32, STM_EUNSUPPORTED, "%s: more than 32 taps", who);
    a.pointwise = (g->n_levels <= 0 && !win && g->kh == 1 && g->kw == 1 && g->sh == 1 && g->sw == 1 && g->ph == 0 && g->pw == 0) ? 1 : 0;
    a.win_w = win ? g->win_w : 0; a.win_hw = win ? g->win_h * g->win_w : 0; a.win_off = win ? g->win_y0 * g->win_w + g->win_x0 : 0;
    a.inv_hw = g->n_levels > 0 ? 0.0f : 1.0f / (float)((int64_t)g->Ho * g->Wo);
    a.inv_w = g->n_levels > 0 ? 0.0f : 1.0f / (float)g->Wo;
    STM_REQUIRE(g->n_levels > 0 || M < ((int64_t)1 << 24), STM_EUNSUPPORTED, "%s: more than 2^24 output pixels in one launch", who);
    a.fmt = (g->fmt >= 0 && g->fmt <= 2) ? g->fmt : 0;
    a.out_fmt = g->out_fmt_plus1 > 0 ? g->out_fmt_plus1 - 1 : a.fmt;
    STM_REQUIRE(a.out_fmt >= 0 && a.out_fmt <= 2 && (a.out_fmt == a.fmt || (a.fmt == 2 && a.out_fmt == 1)), STM_EINVAL,
                "%s: output format %d cannot be produced by a format-%d layer", who, a.out_fmt, a.fmt);
    a.out_scale = (a.fmt >= 1 && g->out_scale > 0.0f) ? g->out_scale : 1.0f;
    a.range_flag = current_range_flag();
    a.nt_out = tn.nt_mb > 0 && M * (int64_t)g->Cout * 4 >= tn.nt_mb * 1000000;
    const int want_planes = a.fmt == 0 ? 3 : (a.fmt == 1 ? 2 : 1);
    STM_REQUIRE(g->planes == want_planes || (a.fmt == 0 && g->planes == 2), STM_EINVAL, "%s: format %d has %d planes (planes = %d)", who, a.fmt,
                want_planes, g->planes);
    a.splitk = 1; a.kslabs = a.slabs; a.partial = nullptr; a.ldp = a.n_tiles * bn;
    a.xp2 = nullptr; a.x2_pstride = 0; a.plane2_bytes = 0; a.c1_slabs = a.slabs; a.x2_np = 0; a.H2 = a.W2 = 0; a.s2 = 1;
    if (dual) {
        const int64_t in2 = (int64_t)g->B * dual->H2 * dual->W2;
        const int64_t np2 = dual->x2_np ? dual->x2_np : in2;
        const int64_t slabs2 = dual->C2 / CV_BK;
        const int64_t ps2 = dual->x2_plane_stride ? dual->x2_plane_stride : slabs2 * np2 * 32;
        STM_REQUIRE(np2 >= in2 && slabs2 * np2 * 64 < ((int64_t)1 << 31) && ps2 % 8 == 0 && (uintptr_t)dual->x2 % 16 == 0, STM_EINVAL,
                    "%s: bad second-source plane geometry", who);
        a.xp2 = static_cast<const uint8_t*>(dual->x2); a.x2_pstride = ps2 * 2; a.plane2_bytes = (unsigned)(slabs2 * np2 * 64);
        a.c1_slabs = (int)x_slabs; a.x2_np = (int)np2; a.H2 = dual->H2; a.W2 = dual->W2; a.s2 = dual->s2;
    }
    // split-K for grids that would leave most CUs idle over a long K (small feature maps: ResNet stages 3/4, P5-P7):
    // parts write fp32 partial sums into the caller's workspace, planar_splitk_finish_kernel adds them and runs the epilogue.
    // (A fused reduction -- the part drawing a tile's last ticket adds the parts -- was built and measured twice.  Round 1 with
    // device-scope fences: an L2 write-back / invalidate per workgroup, 665 vs 919 frames/s.  Round 2 without any fence -- the
    // parts' sums written and read as agent-scope relaxed atomics (sc1 accesses), store -> vmcnt(0) -> ticket ... ticket -> load
    // -- to save the 47 finishing launches of a single-stream step (12 % of its GPU time): 359 vs 461 frames/s at 1 clip, 965 vs
    // 1047 at 8, 1201 vs 1214 at 32.  One workgroup adding a tile's parts at the end of the kernel is a longer tail than the
    // finishing kernel's few thousand threads after it; removed again.)
    // (128 x 64 tiles) every group fills its 64-channel tiles: the layer can take the ring loop; the narrow ones (output layers
    // with 41 / 5 / 32 real channels of 64, offset convolutions) keep the two-buffer loop with its skip of all-padding column tiles
    bool full = cout_g % 64 == 0;
    for (int gi = 0; gi < groups && gi < 8; ++gi) full = full && a.group_real[gi] == cout_g;
    auto plan_splitk = [&](int tiles) {
        int sk = tn.splitk;
        if (sk <= 0) {
            sk = 1;
            if (bn == 64) {
                // 128 x 64 tiles: grids of at most half a workgroup per CU are split until about one workgroup per CU exists,
                // with at least 16 K-slabs per part (10 on the tiniest grids).  Re-measured in round 2 with every configuration
                // replayed from a HIP graph (scripts/sweep_small_m.py; round 1's rule -- up to 256 tiles, three workgroups per
                // CU, 10 slabs per part -- dated from before the ring loop took the small grids): 240 tiles x 32 slabs 31.9 us
                // split in 3 vs 20.2 unsplit, 240 x 72 slabs 49.5 vs 39.8, 120 tiles x 64 slabs 29.8 (6 parts) vs 24.4 (2),
                // 64 x 64 19.4 (6) vs 17.3 (4), 240 x 36 (layer2's 3x3 at 4 clips) 34.3 vs 20.0.
                // The narrow layers (two-buffer loop) keep round 1's rule: 120 tiles x 72 slabs 36.8 us in 6 parts, 55.2 in 2.
                if (tn.sk_rule == 1 || !full) {
                    if (tiles <= 256 && a.slabs >= 24) sk = (int)std::min<int64_t>(std::min<int64_t>(8, 768 / tiles), a.slabs / 10);
                } else if (tiles <= 128 && a.slabs >= 24)
                    sk = (int)std::min<int64_t>(std::min<int64_t>(8, tn.sk_target / tiles), a.slabs / (tiles <= 40 ? 10 : 16));
            } else if (tiles < 128 && a.slabs >= 24) sk = (int)std::min<int64_t>(std::min<int64_t>(8, 256 / tiles), a.slabs / 12);
        }
        if (sk < 2) return;
        const int per = stm_cdiv(a.slabs, sk);
        sk = stm_cdiv(a.slabs, per);
        if (sk < 2 || !workspace || (size_t)sk * M * a.ldp * sizeof(float) > workspace_bytes || ((uintptr_t)workspace % 16)) return;
        a.splitk = sk; a.kslabs = per; a.partial = static_cast<float*>(workspace);
    };
    auto finish_splitk = [&]() -> int {
        if (a.splitk < 2) return STM_OK;
        const int64_t total = M * groups * ((cout_g + 7) / 8);
        hipLaunchKernelGGL(planar_splitk_finish_kernel, dim3(stm_cdiv(total, 256)), dim3(256), 0, stm_hs(stream), a, bn);
        STM_CHECK_LAUNCH("planar_splitk_finish_kernel");
        return STM_OK;
    };
    if (wset) {
        STM_REQUIRE(bn == 128 && a.fmt == 1 && !dual && win, STM_EUNSUPPORTED, "%s: window sets run on the 256 x 128 ring tiles of the fp16x2 format", who);
        // Two grids: the windows whose sub-kernel has all three column taps inside the map (kw = 3: the interior-column border classes, 79 % of the
        // products of a 3x3 layer on 7x7 maps) take conv_planar_kx3_kernel<., WIN> -- one staged run per (channel slab, ky) and window row serves the
        // three taps -- the others (two column taps: the left / right columns) stay on conv_planar_kernel<..., CLS>.  Disjoint output rows.
        PlanarArgsCls ac[2];
        int t0[2] = {0, 0}, ncls[2] = {0, 0};
        for (int k = 0; k < 2; ++k) {
            static_cast<PlanarArgs&>(ac[k]) = a;
            ac[k].pool = wset->pool;
            ac[k].pool_ld = g->Cout;
        }
        for (int i = 0; i < wset->n; ++i) {
            const stm_conv_window& w = wset->win[i];
            STM_REQUIRE(w.kh > 0 && w.kw > 0 && w.kh * w.kw <= 32 && w.Ho > 0 && w.Wo > 0 && w.y0 >= 0 && w.x0 >= 0 && w.y0 + w.Ho <= g->win_h &&
                            w.x0 + w.Wo <= g->win_w, STM_EINVAL, "%s: window %d (%dx%d at %d, %d, kernel %dx%d) is not inside the %dx%d output", who, i, w.Ho,
                        w.Wo, w.y0, w.x0, w.kh, w.kw, g->win_h, g->win_w);
            STM_REQUIRE((uintptr_t)wset->packed[i] % 16 == 0, STM_EINVAL, "%s: packed weight %d is not 16-byte aligned", who, i);
            const int64_t Mc = (int64_t)g->B * w.Ho * w.Wo;
            STM_REQUIRE(Mc < ((int64_t)1 << 24), STM_EUNSUPPORTED, "%s: more than 2^24 output pixels in one window", who);
            // kx-reuse form: every tap inside the input map (no zero fill to stage), and a tile's window rows with their two halo pixels fit 384 staged rows
            const bool taps_inside = -w.ph >= 0 && -w.ph + (w.Ho - 1) + (w.kh - 1) <= g->H - 1 && -w.pw >= 0 && -w.pw + (w.Wo - 1) + (w.kw - 1) <= g->W - 1;
            const int k = (tn.kx3 && w.kw == 3 && w.kh <= 6 && taps_inside && (255 / w.Wo + 2) * (w.Wo + 2) <= 384) ? 1 : 0;
            PlanarArgsCls::Cls& c = ac[k].cls[ncls[k]++];
            c.wp = static_cast<const uint8_t*>(wset->packed[i]);
            c.kh = w.kh; c.kw = w.kw; c.ph = w.ph; c.pw = w.pw; c.Ho = w.Ho; c.Wo = w.Wo; c.M = (int)Mc;
            c.slabs = w.kh * w.kw * (g->C / CV_BK); c.win_off = w.y0 * g->win_w + w.x0; c.tile0 = t0[k];
            c.inv_hw = 1.0f / (float)(w.Ho * w.Wo); c.inv_w = 1.0f / (float)w.Wo;
            t0[k] += stm_cdiv(Mc, 2 * CV_BM) * a.n_tiles;
        }
        for (int k = 0; k < 2; ++k) {
            if (!ncls[k]) continue;
            for (int i = ncls[k]; i < 9; ++i) ac[k].cls[i] = ac[k].cls[0];
            ac[k].n_cls = ncls[k];
            ac[k].cls_tiles = t0[k];
            ac[k].m_tiles = 0;
            ac[k].nsub = 0;
            if (k == 0) {
                ac[0].nsub = ((tn.nsub == 2 || tn.nsub == 4) && a.n_tiles > tn.nsub && a.n_tiles % tn.nsub == 0 && g->groups == 1) ? tn.nsub : 0;
                // (grid: every XCD gets the same number of pixel tiles x all channel tiles -- a multiple of 32 / nsub of them under the regrouped tile
                // map, so that its groups are whole; ids past the last tile leave at once)
                const int pg = ac[0].nsub ? 32 / ac[0].nsub : 1;
#ifndef CLS_ABL
#define CLS_ABL 0     // timing build (RESULTS WRONG): 16 = the window-set kernel issues its activation DMAs on every third K-slab only
#endif
                const int rc0 = launch_planar<2, 2, 2, 1, 3, CLS_ABL, false, true>(ac[0], stm_cdiv(stm_cdiv(t0[0] / a.n_tiles, 8), pg) * pg * 8 * a.n_tiles, stream);
                if (rc0 != STM_OK) return rc0;
            } else {
                static std::atomic<bool> kx3w_reserved[STM_MAX_DEVICES];
                constexpr size_t lds = 2 * 2 * 384 * 64 + 3 * KX3_WBUF;      // two 48-KB activation buffers + the weight ring (the epilogue's park needs less)
                int dev = 0;
                const bool have_dev = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < STM_MAX_DEVICES;
                if (!have_dev || !kx3w_reserved[dev].load(std::memory_order_relaxed)) {
                    STM_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_planar_kx3_kernel<KX3_ABL, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    (int)lds) == hipSuccess, STM_ELAUNCH, "%s: cannot reserve %zu bytes of LDS", who, lds);
                    if (have_dev) kx3w_reserved[dev].store(true, std::memory_order_relaxed);
                }
                hipLaunchKernelGGL((conv_planar_kx3_kernel<KX3_ABL, true>), dim3(stm_cdiv(t0[1] / a.n_tiles, 8) * 8 * a.n_tiles), dim3(512), lds, stm_hs(stream), ac[1]);
                STM_CHECK_LAUNCH("conv_planar_kx3_kernel<WIN>");
                g_kx3_launches.fetch_add(1, std::memory_order_relaxed);
            }
        }
        return STM_OK;
    }
    int rc;
    if (bn == 64) {
        // 128 x 64 tiles, 72 KB of LDS: two independent workgroups per CU, each one's barrier / staging gaps filled by
        // the other's MFMAs
        a.m_tiles = stm_cdiv(M, CV_BM);
        plan_splitk(a.m_tiles * a.n_tiles);
        const int tiles = a.m_tiles * a.n_tiles * a.splitk;
        // the three-buffer ring (72 KB: still two workgroups per CU) has no skip of zero-padded column tiles: `full` layers only
        // measured in the graph (bench.py --layer-table): the ring wins on the short K loops (<= 36 slabs: 35 -> 30 us,
        // 46 -> 36 us, 45 -> 37 us), where its two-slab head start hides the first DMA latency, and loses on the long and
        // the split-K ones (100 -> 114 us at 72 slabs), where two resident two-buffer workgroups already cover each other;
        // under 12 slabs the layer is HBM-bound and the two-buffer loop's 48 KB (three workgroups per CU) wins: 302 vs 389 us
        // grids of at most two workgroups per CU (single-stream / small batches: every layer; layer4 at 8 clips) have no further
        // resident workgroup to cover a workgroup's staging gaps: the ring whatever K.  Swept 0 / 128 / 256 / 512 / 1024 on one
        // box (scripts/sweep_ring64_small.sh): 1 clip 399 / 421 / 436 / 456 / 452 frames/s, 2 clips 628 .. 662, 4 clips 836 ..
        // 880, 8 clips flat (1053), 32 clips 1219 .. 1226.
        const bool small_grid = tiles <= tn.ring64_small;
        const bool ring = full && (tn.ring64 == 3 ? (small_grid || (a.splitk == 1 && a.slabs >= 12 && a.slabs <= 40)) : tn.ring64 == 4);
        if (dual) {
            if (a.fmt == 2) rc = ring ? launch_planar<1, 1, 1, 1, 3, 0, true>(a, tiles, stream) : launch_planar<1, 1, 1, 1, 2, 0, true>(a, tiles, stream);
            else rc = ring ? launch_planar<2, 1, 1, 1, 3, 0, true>(a, tiles, stream) : launch_planar<2, 1, 1, 1, 2, 0, true>(a, tiles, stream);
        } else if (a.fmt == 2) rc = ring ? launch_planar<1, 1, 1, 1, 3>(a, tiles, stream) : launch_planar<1, 1, 1, 1>(a, tiles, stream);
        else if (a.fmt == 1) rc = ring ? launch_planar<2, 1, 1, 1, 3>(a, tiles, stream) : launch_planar<2, 1, 1, 1>(a, tiles, stream);
        else rc = g->planes == 3 ? launch_planar<3, 1, 1>(a, tiles, stream) : launch_planar<2, 1, 1>(a, tiles, stream);
        return rc != STM_OK ? rc : finish_splitk();
    }
    // 256-pixel tiles once there are enough of them, else 128-pixel tiles.  (A cost model of rounds x tile time x measured
    // efficiency was tried for this choice and for the 64-channel tile: 478-483 frames/s against 502-509 with these plain
    // thresholds in the same session -- rejected.  So was a kx-reuse kernel that stages each activation row once per kernel
    // row: 1.8x fewer DMA bytes, no gain -- 654 vs 670 us on the 145-GF proto layer -- removed.  Round 2 rebuilt it on the
    // three-buffer ring -- one staged copy of BM + 2 pixels per (channel slab, ky) serving kx = 0, 1, 2 at LDS row r + kx, an
    // all-zero LDS row for the padding columns, stage parts spread over the three K-slabs, bit-identical results: half the
    // activation DMA instructions, 1/2.8 of their L2 reads, and 1.5-2.5 % SLOWER on every 3x3 layer (proto 1465 vs 1440 us,
    // tower 1850 vs 1807, TemporalNet conv3 2993 vs 2953).  Its own ablations: no stage DMA at all 1406 -> 1188 us, no
    // per-tap register rotation 1379, never waiting for a stage to land 1378.  The cost of the activation staging follows
    // the number of DMA instructions issued (~30 clocks of a wave's issue each), not the bytes they fetch or their latency,
    // and the bookkeeping of the reuse eats what the fewer instructions give back -- removed again.  The three-slab body
    // unrolled (compile-time taps, no rotation) made the register allocator migrate the accumulators: 53 quads, 54 spills.)
    const int64_t t2 = (int64_t)stm_cdiv(M, 2 * CV_BM) * a.n_tiles;
    const int mg = tn.mg ? tn.mg : (t2 >= 192 ? 2 : 1);
    a.m_tiles = stm_cdiv(M, CV_BM * mg);
    plan_splitk(a.m_tiles * a.n_tiles);
    // regrouped tile map (nsub > 0): whole groups of 32 / nsub pixel tiles -- the padding tiles leave at once
    a.nsub = ((tn.nsub == 2 || tn.nsub == 4) && a.splitk == 1 && a.n_tiles > tn.nsub && a.n_tiles % tn.nsub == 0 && g->groups == 1 &&
              a.m_tiles >= 4 * (32 / tn.nsub)) ? tn.nsub : 0;
    if (a.nsub) a.m_tiles = stm_cdiv(a.m_tiles, 32 / a.nsub) * (32 / a.nsub);
    const int tiles = a.m_tiles * a.n_tiles * a.splitk;
    const bool ring = tn.ring == 3;
#ifdef STM_ABLATE
    if (a.fmt == 1 && ring && mg == 2 && tn.abl) {      // timing ablations of the ring loop: RESULTS ARE WRONG
        const int abl = tn.abl;
        rc = abl == 1 ? launch_planar<2, 2, 2, 1, 3, 1>(a, tiles, stream) : abl == 2 ? launch_planar<2, 2, 2, 1, 3, 2>(a, tiles, stream)
           : abl == 3 ? launch_planar<2, 2, 2, 1, 3, 3>(a, tiles, stream) : abl == 5 ? launch_planar<2, 2, 2, 1, 3, 5>(a, tiles, stream)
           : abl == 8 ? launch_planar<2, 2, 2, 1, 3, 8>(a, tiles, stream) : abl == 16 ? launch_planar<2, 2, 2, 1, 3, 16>(a, tiles, stream)
           : abl == 32 ? launch_planar<2, 2, 2, 1, 3, 32>(a, tiles, stream) : abl == 64 ? launch_planar<2, 2, 2, 1, 3, 64>(a, tiles, stream)
                      : launch_planar<2, 2, 2, 1, 3, 7>(a, tiles, stream);
        return rc != STM_OK ? rc : finish_splitk();
    }
#endif
    if (tn.kx3 && a.fmt == 1 && ring && mg == 2 && !dual && !win && a.splitk == 1 && a.kw == 3 && a.pw == 1 && a.sh == 1 && a.sw == 1 &&
        a.slabs % 3 == 0 && a.kh <= 6 && 2 * a.ph == a.kh - 1 && full) {
        // ("same" padding in height too: the kernel decodes a pixel index ONCE and uses it for the output and the staged input rows alike, so
        // Ho == H and Wo == W are part of its contract -- a 3x3 layer with padding (0, 1) stays on conv_planar_kernel)
        // kx-reuse staging (conv_planar_kx3_kernel): one staged run of BM + 2 pixels per (channel slab, ky) serves the three taps of a kernel row
        static std::atomic<bool> kx3_reserved[STM_MAX_DEVICES];
        constexpr size_t lds = 8 * 64 * (64 + 4) * sizeof(float) > (size_t)KX3_LDS_LOOP ? 8 * 64 * (64 + 4) * sizeof(float) : (size_t)KX3_LDS_LOOP;
        int dev = 0;
        const bool have_dev = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < STM_MAX_DEVICES;
        if (!have_dev || !kx3_reserved[dev].load(std::memory_order_relaxed)) {
            STM_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_planar_kx3_kernel<KX3_ABL, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) ==
                            hipSuccess, STM_ELAUNCH, "%s: cannot reserve %zu bytes of LDS", who, lds);
            if (have_dev) kx3_reserved[dev].store(true, std::memory_order_relaxed);
        }
        hipLaunchKernelGGL((conv_planar_kx3_kernel<KX3_ABL, false>), dim3(8 * stm_cdiv(tiles, 8)), dim3(512), lds, stm_hs(stream), a);
        STM_CHECK_LAUNCH("conv_planar_kx3_kernel");
        g_kx3_launches.fetch_add(1, std::memory_order_relaxed);
        return STM_OK;
    }
    if (dual) {      // (the 128-wide tiles of the fp16 formats always take the ring loop here)
        if (a.fmt == 2) rc = mg == 2 ? launch_planar<1, 2, 2, 1, 3, 0, true>(a, tiles, stream) : launch_planar<1, 1, 2, 1, 3, 0, true>(a, tiles, stream);
        else rc = mg == 2 ? launch_planar<2, 2, 2, 1, 3, 0, true>(a, tiles, stream) : launch_planar<2, 1, 2, 1, 3, 0, true>(a, tiles, stream);
    } else if (a.fmt == 2) {
        if (ring) rc = mg == 2 ? launch_planar<1, 2, 2, 1, 3>(a, tiles, stream) : launch_planar<1, 1, 2, 1, 3>(a, tiles, stream);
        else rc = mg == 2 ? launch_planar<1, 2, 2, 1>(a, tiles, stream) : launch_planar<1, 1, 2, 1>(a, tiles, stream);
    } else if (a.fmt == 1) {
        if (ring) rc = mg == 2 ? launch_planar<2, 2, 2, 1, 3>(a, tiles, stream) : launch_planar<2, 1, 2, 1, 3>(a, tiles, stream);
        else rc = mg == 2 ? launch_planar<2, 2, 2, 1>(a, tiles, stream) : launch_planar<2, 1, 2, 1>(a, tiles, stream);
    } else if (g->planes == 3) rc = mg == 2 ? launch_planar<3, 2, 2>(a, tiles, stream) : launch_planar<3, 1, 2>(a, tiles, stream);
    else rc = mg == 2 ? launch_planar<2, 2, 2>(a, tiles, stream) : launch_planar<2, 1, 2>(a, tiles, stream);
    return rc != STM_OK ? rc : finish_splitk();
}
}  // namespace

extern "C" long long stm_debug_launch_count(int which)
{
    return which == 0 ? g_kx3_launches.load(std::memory_order_relaxed) : (which == 1 ? stm_internal_fused_dcn_launches() : -1);
}

extern "C" int stm_conv2d_planar_f32(const void* x_planes, const void* packed_weight, const float* bias, const float* residual_f32,
                                     const void* residual_planes, float* out_f32, void* out_planes, const stm_conv_geom* g,
                                     int relu, stm_stream_t stream)
{
    return stm_conv2d_planar_ws_f32(x_planes, packed_weight, bias, residual_f32, residual_planes, out_f32, out_planes, g, relu, nullptr,
                                    0, stream);
}

// ---- `_f16` entry points ("fp16 variants _f16 for config 5") ------------------------------------------
// The genuine-fp16 convolution path of BASELINE config 5 under its own names: one fp16 plane per tensor (plane format 2), one
// v_mfma_f32_16x16x32_f16 product, fp32 accumulation / bias / residual / ReLU.  Thin forms of the format-aware entries above.
extern "C" int stm_split_planes_f16(const float* x, void* planes, int64_t n_pixels, int C, stm_stream_t stream)
{
    return stm_split_planes_fmt_f32(x, planes, n_pixels, C, 2, stream);
}

extern "C" int stm_conv_pack_weights_f16(const float* weight, void* packed, int Cout, int Cin, int kh, int kw, int tile_n, float wscale,
                                         stm_stream_t stream)
{
    return stm_conv_pack_weights_fmt_f32(weight, packed, Cout, Cin, kh, kw, tile_n, 2, wscale, stream);
}

extern "C" int stm_conv2d_planar_f16(const void* x_planes, const void* packed_weight, const float* bias, const float* residual_f32,
                                     const void* residual_planes, float* out_f32, void* out_planes, const stm_conv_geom* g, int relu,
                                     void* workspace, size_t workspace_bytes, stm_stream_t stream)
{
    STM_REQUIRE(g, STM_ENULL, "stm_conv2d_planar_f16: geometry is NULL");
    stm_conv_geom h = *g;
    h.fmt = 2;
    h.planes = 1;
    return stm_conv2d_planar_ws_f32(x_planes, packed_weight, bias, residual_f32, residual_planes, out_f32, out_planes, &h, relu, workspace,
                                    workspace_bytes, stream);
}
