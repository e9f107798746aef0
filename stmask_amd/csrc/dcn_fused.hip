// dcn_fused.hip -- deformable convolution as ONE kernel: bilinear sampler -> fp16 plane split -> MFMA product, no column buffer.
//
// Replaces, on the inference graph, the pair  stm_deform_sample_planar_f32 (columns [pixels][taps*C] as planes: 78 % of the bytes a
// DCN layer moved)  +  stm_conv2d_planar_f32 as a 1x1 convolution over taps*C channels  --  i.e. dcn_v2.DCN.forward
// (backbone.py:20-26,45 of the reference: modulated 3x3, bias, the bottleneck's ReLU in the epilogue) and mmcv.ops.DeformConv2d
// as FeatureAlign calls it (Featurealign.py:27-31,72: no mask, no bias, 3x3 / 3x5 / 5x3).  SURVEY.md section 8 rows a1 / a7; the fused
// byte formula of section 8(d): input + offsets + output + weights, no columns.
//
// Shape of the kernel (gfx950):
//   * a workgroup = 8 waves = a TH x TW patch of <= 128 output pixels of one image x 128 output channels.  Patches are 2-D so that the rows
//     the taps of neighbouring pixels gather overlap: per 32-channel slab the patch's neighbourhood is ~(TH + 4) x (TW + 4) pixels x 128 B =
//     30 KB, and the K loop runs channel-slab outer / tap inner (the order stm_conv_pack_weights_fmt_f32 packs a kh x kw weight in), so the
//     9 x 4 corner reads of a pixel mostly hit the CU's L1 instead of going to L2 36 times.
//   * roles: waves 0-3 PRODUCE (32 pixels each per K-slab: corner gathers, bilinear blend, fp16 plane split, into an operand ring in LDS; they
//     also stream the weight slabs into a weight ring by LDS-DMA), waves 4-7 CONSUME (64 channels x 64 pixels each: 48 v_mfma_f32_16x16x32_f16
//     per K-slab, D[channel][pixel] = W . X^T, fragments from both rings, read one phase ahead of their use).  One wave of each role per
//     SIMD; one barrier per K-slab; three-slot rings; the gathers stay in flight across the barriers (counted vmcnt).
//   * the gather's lane layout is lane = 4 pixel + quarter, 64 contiguous bytes per pixel and instruction: the texture path coalesces over
//     ADJACENT lanes only (scripts/l1_gather_probe.hip: 69 B/clk/CU from L1 and 61 from L2 in that layout, 18 with the matrix operand's
//     lane = pixel + 16 chunk -- the first version of this kernel gathered straight into the operand registers and took 4 400 cycles per
//     K-slab), so the sampled values cross LDS once on their way to the matrix layout.
//   * per-(pixel, tap) coefficients -- four corner weights with the mask folded in, four clamped byte offsets -- are computed once per tile
//     into LDS (same expressions as dcn_sample_planar_kernel: the sampled values are bit-identical to the unfused sampler's).
//   * epilogue: the consumers park their accumulators as fp32 [pixel][channel] in LDS, all eight waves apply out_scale, bias, ReLU and the
//     plane split and write 1-KB runs (16 pixels x 64 B) per store instruction.
// What bounds it (profiles/r05_dcn_fused_forms.txt, r06_dcn_fused_ablations.txt, r06_power_probe.txt): the consumers alone (MFMAs on real data + fragment reads + weight
// DMA) take 101 us per 256-channel layer at batch 32, the producers alone 73, both 161 -- the sides add.  Not because of the power cap (round 5's reading): the kernel holds
// 2.15-2.27 GHz at 1.36-1.39 kW where the planar kernels sit at 1.9 GHz.  They meet on the SIMD's issue port -- a producer VALU instruction costs ~18 clocks while the consumer
// wave on the same SIMD issues MFMAs back to back: removing the 52-instruction blend saves more time than removing the gathers -- and on the LDS (fragment reads 64 KB + weight
// DMA 32 KB per K-slab against 98 KB per 768 matrix clocks).  Against the pair it replaces: x1.25.
#include "planar_common.h"
#include <atomic>

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int DF_NS = 3;              // ring slots
constexpr int DF_KMAX = 15;           // taps the coefficient tables hold (3x3, 3x5, 5x3)
// Two tile shapes.  NARROW: 128 pixels x 128 channels (two 16-pixel units per producer wave, consumer waves 2 (channel halves) x 2 (pixel halves)).
// WIDE, for layers with Cout a multiple of 256: 64 pixels x 256 channels (one unit per producer wave, the four consumer waves take 64 channels each of
// all 64 pixels).  A 256-channel layer on narrow tiles samples every pixel TWICE (once per channel tile) -- gathers, blend and split are the larger
// half of the kernel's time -- while its weight slabs cross L2 -> LDS once per 128 pixels; wide tiles sample once and stream the weights once per 64
// pixels: 2.4 MB less on-chip traffic per 128 pixels of a 256-channel layer, and half the blend arithmetic.
template <int WIDE>
struct DfShape {
    static constexpr int TP = WIDE ? 64 : 128;        // pixels per tile
    static constexpr int BN = WIDE ? 256 : 128;       // output channels per tile
    static constexpr int NU = WIDE ? 1 : 2;           // 16-pixel units per producer wave
    static constexpr int NH = WIDE ? 2 : 1;           // 128-channel weight tiles (stm_conv_pack_weights_fmt_f32, tile_n 128) per K-slab
    static constexpr int PARK_LD = BN + 4;            // floats per parked pixel row
    static constexpr int COEFA = DF_KMAX * TP * 16;   // bytes of one coefficient table
    static constexpr int RING = 2 * COEFA;            // first byte behind the two tables
};

struct FusedArgs {
    const float* x;        // [B, H, W, x_ld >= C] fp32, pixel-major
    const float* om;       // [B*Ho*Wo, om_ld]: 2K offsets (dy, dx per tap), then K mask logits (MASK)
    const uint8_t* wp;     // stm_conv_pack_weights_fmt_f32 image of the [Cout][C][kh][kw] weight, tile_n 128
    const float* bias;     // [Cout] or null
    uint8_t* out;          // planes [P][Cout/32][out_np][32]
    int B, H, W, C, Ho, Wo, Cout;
    int kh, kw, sh, sw, ph, pw, dh, dw;
    int x_ld, om_ld;
    int out_np, out_pix0;
    long long out_pstride;             // bytes
    int relu;
    float out_scale;
    int out_fmt;                       // 1: two fp16 planes, 2: one
    int* range_flag;
    int TH, TW, tiles_x, tiles_y, m_tiles, n_tiles;
    int K, cslabs, slabs;              // taps, C / 32, K * cslabs
    unsigned x_bytes;                  // addressable bytes of x
};

__device__ __forceinline__ float df_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }   // (deform_im2col.hip: sigmoidf_dev)

// one quarter (two channels) of a lane's 8 sampled values of a K-slab: blend the four corners (deform_im2col.hip bilerp(): w1 v1, then three
// fmas), split into the fp16 planes h = RN16(v), l = RN16((v - h) * 2048) -- planar_common.h split2_f16, value for value:
// v * 2048 and h * 2048 are exact, so fma(h, -2048, v * 2048) is (v - h) * 2048 without a rounding of its own (v_fma_mixlo / mixhi_f16)
template <int NPL, int Q>
__device__ __forceinline__ void blend_part(const f32x4 (&X)[4][2], const f32x4 w, unsigned (&ph)[4], unsigned (&pl)[4])
{
    constexpr int h = Q >> 1, e = Q & 1;
    const f32x2 w1 = {w.x, w.x}, w2 = {w.y, w.y}, w3 = {w.z, w.z}, w4 = {w.w, w.w};
    const f32x2 x1 = {X[0][h][2 * e], X[0][h][2 * e + 1]}, x2 = {X[1][h][2 * e], X[1][h][2 * e + 1]};
    const f32x2 x3 = {X[2][h][2 * e], X[2][h][2 * e + 1]}, x4 = {X[3][h][2 * e], X[3][h][2 * e + 1]};
    f32x2 v = w1 * x1;
    v = __builtin_elementwise_fma(w2, x2, v);
    v = __builtin_elementwise_fma(w3, x3, v);
    v = __builtin_elementwise_fma(w4, x4, v);
    // (No range bookkeeping here since round 6: a sampled value beyond fp16's range becomes inf / nan in the h plane, every product it enters is inf or nan, and the
    // output pass tests the PRE-activation value -- df_store_tile.  Two vector instructions per pair less in a stream where each costs ~18 clocks beside the MFMA wave.)
    const f16x2 hh = __builtin_convertvector(v, f16x2);
    ph[Q] = __builtin_bit_cast(unsigned, hh);
    if constexpr (NPL == 2) {
        const f32x2 vs = v * STM_F16_LOW_SCALE;
        f16x2 ll;
        ll.x = (_Float16)__builtin_fmaf((float)hh.x, -STM_F16_LOW_SCALE, vs.x);
        ll.y = (_Float16)__builtin_fmaf((float)hh.y, -STM_F16_LOW_SCALE, vs.y);
        pl[Q] = __builtin_bit_cast(unsigned, ll);
    }
}

// Coefficients of a tile: (pixel, tap) -> four corner weights (mask folded in), four corner byte offsets into x.  Expressions and their order are
// dcn_sample_planar_kernel's (deform_im2col.hip): same weights, same clamped corners, bit for bit.  All 512 threads of the workgroup.
template <bool MASK, int TP>
__device__ __forceinline__ void df_coefficients(const FusedArgs& a, int b, int oy0, int ox0, uint8_t* coefW, uint8_t* coefA, int tid)
{
    const int K = a.K;
    const int tp = a.TH * a.TW;
    // the (up to four) items of a thread: their offset / mask loads all issued before the first is used -- one memory round trip per tile, not one per item
    constexpr int NIT = (DF_KMAX * TP + 511) / 512;
    float dyv[NIT], dxv[NIT], mkv[NIT];
#pragma unroll
    for (int n = 0; n < NIT; ++n) {
        const int it = tid + n * 512;
        const int t = it & (TP - 1), k = it / TP;
        const int ty = t / a.TW, tx = t - ty * a.TW;
        const int ho = oy0 + ty, wo = ox0 + tx;
        dyv[n] = dxv[n] = mkv[n] = 0.0f;
        if (k < K && t < tp && ho < a.Ho && wo < a.Wo) {
            const float* omp = a.om + (size_t)((b * a.Ho + ho) * a.Wo + wo) * a.om_ld;
            dyv[n] = omp[2 * k];
            dxv[n] = omp[2 * k + 1];
            if (MASK) mkv[n] = omp[2 * K + k];
        }
    }
#pragma unroll
    for (int n = 0; n < NIT; ++n) {
        const int it = tid + n * 512;
        const int t = it & (TP - 1), k = it / TP;
        if (k >= K) break;
        const int ty = t / a.TW, tx = t - ty * a.TW;
        const int ho = oy0 + ty, wo = ox0 + tx;
        float cw1 = 0.f, cw2 = 0.f, cw3 = 0.f, cw4 = 0.f;
        int ca1 = 0, ca2 = 0, ca3 = 0, ca4 = 0;
        if (t < tp && ho < a.Ho && wo < a.Wo) {
            const int i = k / a.kw, j = k - a.kw * i;
            const float dy = dyv[n], dx = dxv[n];
            const float mk = MASK ? df_sigmoid(mkv[n]) : 1.0f;
            const float fy = (float)(ho * a.sh - a.ph + i * a.dh) + dy;
            const float fx = (float)(wo * a.sw - a.pw + j * a.dw) + dx;
            if (fy > -1.0f && fx > -1.0f && fy < (float)a.H && fx < (float)a.W) {
                const float fl_y = floorf(fy), fl_x = floorf(fx);
                const int h_low = (int)fl_y, w_low = (int)fl_x, h_high = h_low + 1, w_high = w_low + 1;
                const float lh = fy - fl_y, lw = fx - fl_x, hh = 1.0f - lh, hw = 1.0f - lw;
                const bool tt = h_low >= 0, l = w_low >= 0, bt = h_high <= a.H - 1, r = w_high <= a.W - 1;
                const int hl = max(h_low, 0), wl = max(w_low, 0), hh_i = min(h_high, a.H - 1), wh_i = min(w_high, a.W - 1);
                cw1 = (tt && l) ? hh * hw * mk : 0.f;
                cw2 = (tt && r) ? hh * lw * mk : 0.f;
                cw3 = (bt && l) ? lh * hw * mk : 0.f;
                cw4 = (bt && r) ? lh * lw * mk : 0.f;
                const int rowb = b * a.H;
                ca1 = (((rowb + hl) * a.W + wl) * a.x_ld) * 4;
                ca2 = (((rowb + hl) * a.W + wh_i) * a.x_ld) * 4;
                ca3 = (((rowb + hh_i) * a.W + wl) * a.x_ld) * 4;
                ca4 = (((rowb + hh_i) * a.W + wh_i) * a.x_ld) * 4;
            }
        }
        *reinterpret_cast<f32x4*>(coefW + (k * TP + t) * 16) = f32x4{cw1, cw2, cw3, cw4};
        *reinterpret_cast<i32x4*>(coefA + (k * TP + t) * 16) = i32x4{ca1, ca2, ca3, ca4};
    }
}

// Output pass of a tile: the TP pixels x BN channels parked in LDS as fp32 [pixel][PARK_LD] (pixel = tile-linear index) -> out_scale, bias, ReLU,
// plane split, 16-byte stores.  Narrow: wave w writes pixels 16 w .. 16 w + 15; wide: pixels 16 (w & 3) .. + 15 of channel half w >> 2.  One 32-channel
// slab per pass: 1-KB runs where the pixels are one image row.
template <int WIDE>
__device__ __forceinline__ void df_store_tile(const FusedArgs& a, const float* park_all, int wave, int lane, int b, int oy0, int ox0, int nt)
{
    constexpr int DF_PARK_LD = DfShape<WIDE>::PARK_LD, DF_BN = DfShape<WIDE>::BN;
    const int px = lane >> 2, cs = lane & 3;
    const int t = (WIDE ? (wave & 3) : wave) * 16 + px;
    const int ch0 = WIDE ? (wave >> 2) * 128 : 0;
    const int ty = t / a.TW, tx = t - ty * a.TW;
    const int ho = oy0 + ty, wo = ox0 + tx;
    const bool live = t < a.TH * a.TW && ho < a.Ho && wo < a.Wo;
    const int mo = a.out_pix0 + (b * a.Ho + ho) * a.Wo + wo;
    const float* park = park_all + t * DF_PARK_LD;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = ch0 + 32 * i + 8 * cs;
        const int co = nt * DF_BN + ch;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(park + ch);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(park + ch + 4);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        f32x4 b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
        if (a.bias) { b0 = *reinterpret_cast<const f32x4*>(a.bias + co); b1 = *reinterpret_cast<const f32x4*>(a.bias + co + 4); }
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e], a.out_scale, bb[e]);
        // the fp16 range guard on the PRE-activation value (a ReLU would turn a nan into 0): |v| > 65504, inf or nan -- which is also where a sampled value
        // beyond fp16's range ends up (its h plane is inf: every product with it is inf or nan), so the producers carry no range bookkeeping of their own
        if (live) f16_range_check8(v, a.range_flag);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (a.relu) v[e] = __builtin_fmaxf(v[e], 0.0f);
        if (live) {
            unsigned q0[4], q1[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) split2_f16(f32x2{v[2 * e], v[2 * e + 1]}, q0[e], q1[e]);
            uint8_t* o = a.out + (((size_t)(co >> 5) * a.out_np + mo) * 32 + (co & 31)) * 2;
            *reinterpret_cast<u32x4*>(o) = u32x4{q0[0], q0[1], q0[2], q0[3]};
            if (a.out_fmt == 1) *reinterpret_cast<u32x4*>(o + a.out_pstride) = u32x4{q1[0], q1[1], q1[2], q1[3]};
        }
    }
}

#ifndef DF_CPRIO
#define DF_CPRIO 0
#endif
#ifndef DF_PPRIO
#define DF_PPRIO 3
#endif
#ifndef DF_PSCHED
#define DF_PSCHED 6      // vector instructions between two gathers of the producer stream (0: the compiler's order)
#endif
#ifndef DF_WAUX
#define DF_WAUX 0        // cache policy bits of the weight DMA (2 = nt) -- timing experiments
#endif
#ifndef DF_ABL
#define DF_ABL 0         // timing builds (RESULTS WRONG): 1 no corner gathers, 2 no blend / split / staging, 4 no MFMAs, 8 no weight DMA, 16 no fragment reads
#endif

// LDS map (bytes): coefficient tables [15 taps][128 pixels] x 16 B weights, then offsets; the weight ring; the operand ring (sampled planes of a
// K-slab: [plane][128 pixel rows][64 B], the planar kernels' chunk swizzle).  The parked output tile (67.6 KB) overlays it all after the K loop.
template <int NPL, bool MASK, int WIDE>
__global__ __launch_bounds__(512, 1) void dcn_fused_kernel(const FusedArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __align__(16) uint8_t smem[];
    typedef DfShape<WIDE> S;
    constexpr int DF_TP = S::TP, NU = S::NU, DF_COEFA = S::COEFA, DF_RING = S::RING, DF_PARK_LD = S::PARK_LD;
    constexpr int WPL = 128 * 64, WT = NPL * WPL;              // bytes of one plane of a 128-channel weight tile's K-slab / of the tile's K-slab
    constexpr int WBUF = S::NH * WT;                           // ... of one K-slab of weights in the ring: [128-channel tile][plane][128 rows][64 B]
    constexpr int BPL = DF_TP * 64, BBUF = NPL * BPL;          // ... of one sampled plane / of one K-slab of sampled values
    constexpr int BRING = DF_RING + DF_NS * WBUF;
    constexpr int NPW = WBUF / 1024 / 4;                       // weight DMA pieces (1 KB) per producer wave and slab
    const int K = a.K;
    uint8_t* const coefW = smem;
    uint8_t* const coefA = smem + DF_COEFA;

    const int tiles = a.m_tiles * a.n_tiles;
    const int per_xcd = (tiles + 7) >> 3;
    const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || logical >= tiles) return;
    const int mt = a.n_tiles == 1 ? logical : (a.n_tiles == 2 ? logical >> 1 : (a.n_tiles == 4 ? logical >> 2 : logical / a.n_tiles));
    const int nt = logical - mt * a.n_tiles;
    const int tpi = a.tiles_x * a.tiles_y;
    const int b = mt / tpi, rt = mt - b * tpi;
    const int tyi = rt / a.tiles_x, txi = rt - tyi * a.tiles_x;
    const int oy0 = tyi * a.TH, ox0 = txi * a.TW;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    typedef __attribute__((address_space(3))) void* lds_ptr;

    // Roles.  Waves 0-3 PRODUCE: each samples 32 of the tile's 128 pixels per K-slab -- gathers the four corners, blends, splits into fp16 planes --
    // writes them into the operand ring and streams the weight slabs into the weight ring.  Waves 4-7 CONSUME: each multiplies 64 channels x 64 pixels
    // (4 x 4 tiles of v_mfma_f32_16x16x32_f16, D[channel][pixel] = W . X^T), fragments from both rings.  One wave of each role per SIMD: the
    // producer's VALU / memory stream and the consumer's MFMA stream run beside each other by themselves, instead of being dealt into one stream
    // (the first form of this kernel -- every wave sampling its own 16 pixels and multiplying them by all 128 channels -- ran its pipes one after
    // the other: LDS fragment reads 36 us + weight DMA 20 + MFMA 54 + gathers 64 = the 171 us it took per layer; profiles/r05_dcn_fused_forms.txt).
    // One barrier per K-slab for all eight waves: at barrier s the operand slab s and the weight slab s are complete (producers wrote / landed them
    // one to two slabs ago) and ring slot (s + 2) % 3 is free (the consumers have drained their reads of slab s - 1).
    df_coefficients<MASK, DF_TP>(a, b, oy0, ox0, coefW, coefA, tid);
    __syncthreads();

    f32x4 acc[4][4], accl[4][4];                                  // consumer: [channel tile][pixel tile], main / correction products
    if (wave < 4) {
        // ================================================ producer =======================================================================
        // lane = 4 gp + gq: pixel gp of a 16-pixel unit, quarter gq; a corner's 128-byte channel slab arrives as two instructions of 64 contiguous
        // bytes per pixel (four adjacent lanes share a line: scripts/l1_gather_probe.hip -- 69 B/clk/CU from L1, 61 from L2; the matrix layout's
        // lane = pixel + 16 chunk gets 18).  The lane then holds channels 4 gq .. + 3 and 16 + 4 gq .. + 3 of its pixel; two units per wave.
        const int pw = wave;
        if (DF_PPRIO) __builtin_amdgcn_s_setprio(DF_PPRIO);         // the producer's vector / memory instructions win the SIMD's issue port over the consumer's MFMAs
        const int gp = lane >> 2, gq = lane & 3;
        const int gq16 = gq * 16;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
        const uint8_t* wtile = a.wp + (size_t)(nt * S::NH) * a.slabs * WT;          // this tile's NH consecutive 128-channel weight tiles
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(wtile), 0, S::NH * a.slabs * WT, 0x00020000);
        const int lane16 = lane * 16;
        auto dma_w = [&](int slab, int slot) {
            if constexpr ((DF_ABL & 8) != 0) return;
            uint8_t* wb = smem + DF_RING + slot * WBUF;
#pragma unroll
            for (int j = 0; j < NPW; ++j) {
                const int wi = pw + 4 * j;                                        // 1-KB piece of the ring slot
                const int wh_ = wi / (WT / 1024), wp_ = wi - wh_ * (WT / 1024);   // weight tile, piece inside its K-slab
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr)(wb + wi * 1024), 16, lane16, (wh_ * a.slabs + slab) * WT + wp_ * 1024, 0, DF_WAUX);
            }
        };
        const int cpix0 = (pw * 16 * NU + gp) * 16, cpix1 = cpix0 + 16 * 16;     // the lane's pixel of unit 0 / 1 inside a tap's coefficient block
        // operand-ring row of the lane's pixel (unit 0; unit 1 = + 16 rows), channels 4 gq (+ 16 h): chunk 2 h + (gq >> 1), second half for odd gq
        const int row0 = pw * 16 * NU + gp;
        const int st_w0 = BRING + lds_off(row0, gq >> 1) + 8 * (gq & 1), st_w1 = BRING + lds_off(row0, 2 + (gq >> 1)) + 8 * (gq & 1);
        f32x4 XA[NU][4][2], XB[NU][4][2];                           // corner values [unit][corner][half], two sets
        f32x4 WA[NU], WB[NU];                                       // corner weights [unit] of the sets
#pragma unroll
        for (int u = 0; u < NU; ++u) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { XA[u][c][0] = XA[u][c][1] = XB[u][c][0] = XB[u][c][1] = f32x4{0, 0, 0, 0}; }
            WA[u] = WB[u] = f32x4{0, 0, 0, 0};
        }
        int g_tap = 0, g_soff = 0;                                  // tap and channel-slab byte offset of the next slab to gather
#define DF_GATHER(X_, W_)                                                                                                          \
        {                                                                                                                          \
            _Pragma("unroll") for (int u = 0; u < NU; ++u) {                                                                       \
                W_[u] = *reinterpret_cast<const f32x4*>(coefW + g_tap * (DF_TP * 16) + (u ? cpix1 : cpix0));                       \
                const i32x4 ca_ = *reinterpret_cast<const i32x4*>(coefA + g_tap * (DF_TP * 16) + (u ? cpix1 : cpix0));             \
                if (!(DF_ABL & 1)) {                                                                                               \
                    _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                                \
                        const int vo = ca_[c] + gq16;                                                                              \
                        X_[u][c][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, vo, g_soff, 0));         \
                        X_[u][c][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, vo + 64, g_soff, 0));    \
                    }                                                                                                              \
                }                                                                                                                  \
            }                                                                                                                      \
            if (++g_tap == K) { g_tap = 0; g_soff += 128; }                                                                        \
        }
        // blend + split of a gathered slab, both units, into operand-ring slot SLOT_
#define DF_BLEND(X_, W_, SLOT_)                                                                                                    \
        if (!(DF_ABL & 2)) {                                                                                                       \
            typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));                                                           \
            uint8_t* const bs_ = smem + (SLOT_) * BBUF;                                                                            \
            _Pragma("unroll") for (int u = 0; u < NU; ++u) {                                                                       \
                unsigned ph[4], pl[4];                                                                                             \
                blend_part<NPL, 0>(X_[u], W_[u], ph, pl);                                                                    \
                blend_part<NPL, 1>(X_[u], W_[u], ph, pl);                                                                    \
                blend_part<NPL, 2>(X_[u], W_[u], ph, pl);                                                                    \
                blend_part<NPL, 3>(X_[u], W_[u], ph, pl);                                                                    \
                *reinterpret_cast<u32x2_*>(bs_ + st_w0 + u * 1024) = u32x2_{ph[0], ph[1]};                                         \
                *reinterpret_cast<u32x2_*>(bs_ + st_w1 + u * 1024) = u32x2_{ph[2], ph[3]};                                         \
                if constexpr (NPL == 2) {                                                                                          \
                    *reinterpret_cast<u32x2_*>(bs_ + BPL + st_w0 + u * 1024) = u32x2_{pl[0], pl[1]};                               \
                    *reinterpret_cast<u32x2_*>(bs_ + BPL + st_w1 + u * 1024) = u32x2_{pl[2], pl[3]};                               \
                }                                                                                                                  \
            }                                                                                                                      \
        }
        // Iteration i: weight DMA of slab i + 2, gathers of slab i + 3, blend of slab i + 2 (gathered an iteration ago) into operand slot
        // (i + 2) % 3; the barrier that ENDS it certifies slab i + 2 (the consumers prefetch the fragments of slab i + 2 during step i + 1).  Past
        // the last slab the same operations run on clamped / out-of-range addresses into free slots: the counted wait relies on a constant number
        // of vector-memory operations per iteration -- younger than this iteration's weight DMA are only its 16 gathers.
#define DF_PROD(I_, XG_, WG_, XV_, WV_)                                                                                             \
        {                                                                                                                          \
            dma_w(min((I_) + 2, a.slabs - 1), slot2);                                                                              \
            __builtin_amdgcn_sched_barrier(0);               /* the counted wait assumes this issue order */                       \
            DF_GATHER(XG_, WG_);                                                                                                   \
            DF_BLEND(XV_, WV_, slot2);                                                                                             \
            /* the gathers dealt out between the blend's arithmetic: issued as a block they fill the texture unit's queue, the wave stalls at   \
               issue behind them, and its ~110 vector instructions only start when the queue has drained */                        \
            if (DF_PSCHED) {                                                                                                       \
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * NU, 0);                                                            \
                _Pragma("unroll") for (int k = 0; k < 8 * NU; ++k) {                                                                   \
                    __builtin_amdgcn_sched_group_barrier(0x002, DF_PSCHED, 0);                                                     \
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                             \
                }                                                                                                                  \
            }                                                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                                     \
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(8 * NU) : "memory");                                  \
            slot2 = slot2 == 2 ? 0 : slot2 + 1;                                                                                    \
        }
        // prologue = iterations -3, -2, -1: G(0) | W(0), G(1), blend 0 | W(1), G(2), blend 1; barrier 0 certifies slabs 0 and 1
        DF_GATHER(XA, WA);
        __builtin_amdgcn_sched_barrier(0);
        dma_w(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        DF_GATHER(XB, WB);
        DF_BLEND(XA, WA, 0);
        __builtin_amdgcn_sched_barrier(0);
        dma_w(min(1, a.slabs - 1), 1);
        __builtin_amdgcn_sched_barrier(0);
        DF_GATHER(XA, WA);
        DF_BLEND(XB, WB, 1);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(8 * NU) : "memory");
        int slot2 = 2;
        for (int i = 0; i < a.slabs; i += 2) {
            DF_PROD(i, XB, WB, XA, WA);
            DF_PROD(i + 1, XA, WA, XB, WB);
        }
#undef DF_PROD
#undef DF_BLEND
#undef DF_GATHER
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the rings' last (unused) slabs have landed before the LDS is reused
    } else {
        // ================================================ consumer =======================================================================
        if (DF_CPRIO) __builtin_amdgcn_s_setprio(DF_CPRIO);
        const int cw = wave - 4;
        const int chh = cw & 1, pxh = WIDE ? 0 : cw >> 1, wh = WIDE ? cw >> 1 : 0;   // 64-channel half of weight tile wh, 64-pixel half of the tile
        const int p = lane & 15, q = lane >> 4;                      // fragment row (channel of a weight tile / pixel of an operand tile), K chunk
        const int aoff = DF_RING + wh * WT + lds_off(p, q) + chh * (64 * 64);  // + i * 1024: channel tile i of this wave
        const int boff = BRING + lds_off(p, q) + pxh * (64 * 64);    // + j * 1024: pixel tile j of this wave
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc[i][j][r] = 0.0f; accl[i][j][r] = 0.0f; }
#define DF_MM(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a_), __builtin_bit_cast(f16x8, b_), c_, 0, 0, 0)
#define DF_MM32(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_), __builtin_bit_cast(f16x8, b_), c_, 0, 0, 0)
        // (DF_ABL & 64, timing only: the same flops as 6 v_mfma_f32_32x32x16_f16 per phase instead of 12 v_mfma_f32_16x16x32_f16, on whatever the fragment
        // registers hold -- does the matrix instruction's shape change what the producer waves on the same SIMDs get of the issue port?)
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        f32x16 acc32[2][2], accl32[2][2];
        if (DF_ABL & 64) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { acc32[i][j][r] = 0.0f; accl32[i][j][r] = 0.0f; }
        }
        // Fragments are read one phase ahead of the MFMAs that use them, across the barriers: a step = four phases (channel tiles), each 12 MFMAs;
        // phase i reads the weight fragments of channel tile i + 1 (of the NEXT slab in the last phase) into the other of two fragment slots, and
        // two of the next slab's eight operand fragments into the other operand set.  (Read at the top of the step -- barrier, 16 reads, wait,
        // 48 MFMAs -- the consumer sat through ~450 cycles of LDS time per slab with an idle matrix pipe: 116 us per layer alone instead of ~80.)
        u32x4 ah[2], al[2];                                          // weight fragments, two slots (channel tile i in slot i & 1)
        u32x4 bhA[4], blA[4], bhB[4], blB[4];                        // operand fragments of a slab, two sets
#pragma unroll
        for (int t = 0; t < 4; ++t) bhA[t] = blA[t] = bhB[t] = blB[t] = u32x4{0, 0, 0, 0};
        ah[0] = ah[1] = al[0] = al[1] = u32x4{0, 0, 0, 0};
#define DF_RD_A(SLOT_, WS_, I_)                                                                                                    \
        if (!(DF_ABL & 16)) {                                                                                                      \
            ah[SLOT_] = *reinterpret_cast<const u32x4*>((WS_) + (I_) * 1024);                                                      \
            if constexpr (NPL == 2) al[SLOT_] = *reinterpret_cast<const u32x4*>((WS_) + WPL + (I_) * 1024);                        \
        }
#define DF_RD_B(BH_, BL_, BS_, J_)                                                                                                 \
        if (!(DF_ABL & 16)) {                                                                                                      \
            BH_[J_] = *reinterpret_cast<const u32x4*>((BS_) + (J_) * 1024);                                                        \
            if constexpr (NPL == 2) BL_[J_] = *reinterpret_cast<const u32x4*>((BS_) + BPL + (J_) * 1024);                          \
        }
        // channel tile I_ (fragments in slot I_ & 1) against the four pixel tiles: the four x_l w_h products, the four x_h w_h, then the four
        // x_h w_l, each 8 MFMAs behind the product whose accumulator it continues -- conv_planar_kernel's sums in its order per accumulator
#define DF_TILE(I_, BH_, BL_)                                                                                                      \
        if (DF_ABL & 64) {                                                                                                         \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) accl32[(I_) >> 1][j] = DF_MM32(ah[(I_) & 1], BL_[j], accl32[(I_) >> 1][j]); \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) acc32[(I_) >> 1][j] = DF_MM32(ah[(I_) & 1], BH_[j], acc32[(I_) >> 1][j]);   \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) accl32[(I_) >> 1][j] = DF_MM32(al[(I_) & 1], BH_[j], accl32[(I_) >> 1][j]); \
        } else if (!(DF_ABL & 4)) {                                                                                                \
            if constexpr (NPL == 2) {                                                                                              \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) accl[I_][j] = DF_MM(ah[(I_) & 1], BL_[j], accl[I_][j]);              \
            }                                                                                                                      \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[I_][j] = DF_MM(ah[(I_) & 1], BH_[j], acc[I_][j]);                    \
            if constexpr (NPL == 2) {                                                                                              \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) accl[I_][j] = DF_MM(al[(I_) & 1], BH_[j], accl[I_][j]);              \
            }                                                                                                                      \
        }
        constexpr int RPP = NPL == 2 ? 6 : 3;                        // LDS reads per phase: one weight tile, two operand tiles (x planes)
#define DF_PHASE_SCHED()                                                                                                           \
        __builtin_amdgcn_sched_group_barrier(0x100, RPP, 0);                                                                       \
        __builtin_amdgcn_sched_group_barrier(0x008, (DF_ABL & 64) ? 6 : NPL == 2 ? 12 : 4, 0);                                     \
        __builtin_amdgcn_sched_barrier(0);
        // step: slab s in operand set (BHC_, BLC_) and weight slot 0 (tile 0); slab s + 1 (certified by the barrier that opened the step) is
        // prefetched into (BHN_, BLN_)
#define DF_CSTEP(BHC_, BLC_, BHN_, BLN_)                                                                                            \
        {                                                                                                                          \
            const uint8_t* ws = smem + aoff + cur * WBUF;                                                                          \
            const uint8_t* wn = smem + aoff + nxt * WBUF;                                                                          \
            const uint8_t* bn = smem + boff + nxt * BBUF;                                                                          \
            DF_RD_A(1, ws, 1); DF_RD_B(BHN_, BLN_, bn, 0); DF_RD_B(BHN_, BLN_, bn, 1);                                             \
            DF_TILE(0, BHC_, BLC_);                                                                                                \
            DF_PHASE_SCHED();                                                                                                      \
            DF_RD_A(0, ws, 2); DF_RD_B(BHN_, BLN_, bn, 2); DF_RD_B(BHN_, BLN_, bn, 3);                                             \
            DF_TILE(1, BHC_, BLC_);                                                                                                \
            DF_PHASE_SCHED();                                                                                                      \
            DF_RD_A(1, ws, 3);                                                                                                     \
            DF_TILE(2, BHC_, BLC_);                                                                                                \
            __builtin_amdgcn_sched_group_barrier(0x100, NPL, 0);                                                                   \
            __builtin_amdgcn_sched_group_barrier(0x008, (DF_ABL & 64) ? 6 : NPL == 2 ? 12 : 4, 0);                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                                     \
            DF_RD_A(0, wn, 0);                                                                                                     \
            DF_TILE(3, BHC_, BLC_);                                                                                                \
            __builtin_amdgcn_sched_group_barrier(0x100, NPL, 0);                                                                   \
            __builtin_amdgcn_sched_group_barrier(0x008, (DF_ABL & 64) ? 6 : NPL == 2 ? 12 : 4, 0);                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                                     \
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                                        \
            cur = nxt; nxt = nxt == 2 ? 0 : nxt + 1;                                                                               \
        }
        asm volatile("s_barrier" ::: "memory");                      // barrier 0: slabs 0 and 1 are complete
        {
            const uint8_t* ws = smem + aoff, *bs = smem + boff;
            DF_RD_A(0, ws, 0);
            DF_RD_B(bhA, blA, bs, 0); DF_RD_B(bhA, blA, bs, 1); DF_RD_B(bhA, blA, bs, 2); DF_RD_B(bhA, blA, bs, 3);
        }
        int cur = 0, nxt = 1;
        for (int s = 0; s < a.slabs; s += 2) {
            DF_CSTEP(bhA, blA, bhB, blB);
            DF_CSTEP(bhB, blB, bhA, blA);
        }
        if (DF_ABL & 64) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { acc[i][j][r & 3] += acc32[i][j][r]; accl[i][j][r & 3] += accl32[i][j][r]; }
        }
#undef DF_MM32
#undef DF_CSTEP
#undef DF_PHASE_SCHED
#undef DF_TILE
#undef DF_RD_B
#undef DF_RD_A
#undef DF_MM
    }

    // ---- epilogue: the consumers park their 64 x 64 tiles as fp32 [pixel][channel], then all eight waves write the tile out ----------------------
    __syncthreads();                                                // nobody reads or fills the rings or the coefficient tables any more
    float* const park_all = reinterpret_cast<float*>(smem);
    if (wave >= 4) {
        const int cw = wave - 4, chh = (WIDE ? cw : cw & 1), pxh = WIDE ? 0 : cw >> 1, p = lane & 15, g = lane >> 4;
        constexpr float LS = 1.0f / STM_F16_LOW_SCALE;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = acc[i][j];
                if constexpr (NPL == 2) v = acc[i][j] + accl[i][j] * LS;
                *reinterpret_cast<f32x4*>(park_all + (pxh * 64 + j * 16 + p) * DF_PARK_LD + chh * 64 + i * 16 + 4 * g) = v;
            }
    }
    __syncthreads();
    df_store_tile<WIDE>(a, park_all, wave, lane, b, oy0, ox0, nt);
#endif
}

// tile patch TH x TW (<= 128 pixels) of a Ho x Wo output image: the one that wastes the fewest tile pixels, then the squarest
void pick_patch(int Ho, int Wo, int& TH, int& TW, int DF_TP)
{
    double best = -1.0;
    int bth = 8, btw = 16, bper = 1 << 30;
    for (int tw = 4; tw <= DF_TP; ++tw) {
        const int th = DF_TP / tw;
        if (th < 1) break;
        const int thc = th > Ho ? Ho : th, twc = tw > Wo ? Wo : tw;
        const long long tiles = (long long)stm_cdiv(Ho, thc) * stm_cdiv(Wo, twc);
        const double eff = (double)Ho * Wo / ((double)tiles * DF_TP);
        const int per = thc + twc;                      // half perimeter: the neighbourhood the taps gather grows with it
        if (eff > best + 1e-9 || (eff > best - 1e-9 && per < bper)) { best = eff; bth = thc; btw = twc; bper = per; }
    }
    TH = bth; TW = btw;
}

template <int NPL, bool MASK, int WIDE>
int df_launch(const FusedArgs& a, size_t lds, hipStream_t stream, const char* who)
{
    static std::atomic<int> reserved[32];
    int dev = 0;
    const bool have_dev = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 32;
    if (!have_dev || reserved[dev].load(std::memory_order_relaxed) < (int)lds) {
        STM_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(dcn_fused_kernel<NPL, MASK, WIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) ==
                        hipSuccess, STM_ELAUNCH, "%s: cannot reserve %zu bytes of LDS", who, lds);
        if (have_dev) reserved[dev].store((int)lds, std::memory_order_relaxed);
    }
    const dim3 grid(8 * stm_cdiv((int64_t)a.m_tiles * a.n_tiles, 8));
    hipLaunchKernelGGL((dcn_fused_kernel<NPL, MASK, WIDE>), grid, dim3(512), lds, stream, a);
    STM_CHECK_LAUNCH("dcn_fused_kernel");
    return STM_OK;
}

std::atomic<long long> g_fused_launches{0};

}  // namespace

long long stm_internal_fused_dcn_launches() { return g_fused_launches.load(std::memory_order_relaxed); }

extern "C" int stm_deform_conv_fused_planar_supported(const stm_deform_geom* g, int Cout, int has_mask, int fmt)
{
    if (!g) return 0;
    const int K = g->kh * g->kw;
    if (g->dg != 1 || K < 1 || K > 15 || g->C % 64 != 0 || Cout % 128 != 0 || (fmt != 1 && fmt != 2)) return 0;
    if (has_mask && K != 9) return 0;
    return 1;
}

extern "C" int stm_deform_conv_fused_planar_f32(const float* x, int x_ld, const float* offsets, int om_ld, int has_mask, const void* packed_weight,
                                                const float* bias, void* out_planes, int out_np, int out_pixel_offset, long long out_plane_stride,
                                                int Cout, int relu, float out_scale, const stm_deform_geom* g, int fmt, int out_fmt, stm_stream_t stream)
{
    const char* who = "stm_deform_conv_fused_planar_f32";
    STM_REQUIRE(x && offsets && packed_weight && out_planes && g, STM_ENULL, "%s: NULL argument", who);
    STM_REQUIRE(fmt == 1 || fmt == 2, STM_EUNSUPPORTED, "%s: fmt must be 1 (fp16 x 2) or 2 (fp16 x 1); the bf16 x 3 format keeps the sampler + product pair", who);
    STM_REQUIRE(out_fmt == fmt || (fmt == 2 && out_fmt == 1), STM_EUNSUPPORTED, "%s: out_fmt must be fmt (or 1 under fmt 2)", who);
    const int K = g->kh * g->kw;
    STM_REQUIRE(stm_deform_conv_fused_planar_supported(g, Cout, has_mask, fmt), STM_EUNSUPPORTED,
                "%s: one deformable group, <= 15 taps (9 with mask), C a multiple of 64, Cout a multiple of 128 (got %dx%d, dg %d, C %d, Cout %d)", who,
                g->kh, g->kw, g->dg, g->C, Cout);
    STM_REQUIRE(g->B > 0 && g->H > 0 && g->W > 0 && g->Ho > 0 && g->Wo > 0 && om_ld >= (has_mask ? 3 : 2) * K && x_ld >= g->C && x_ld % 4 == 0 &&
                out_pixel_offset >= 0 && g->sh > 0 && g->sw > 0, STM_EINVAL, "%s: bad geometry", who);
    STM_REQUIRE(((uintptr_t)x | (uintptr_t)packed_weight | (uintptr_t)out_planes | (uintptr_t)bias) % 16 == 0, STM_EINVAL, "%s: pointers must be 16-byte aligned", who);
    const int64_t M = (int64_t)g->B * g->Ho * g->Wo;
    const int64_t xb = (int64_t)g->B * g->H * g->W * x_ld * 4;
    STM_REQUIRE(M < ((int64_t)1 << 30) && xb < ((int64_t)1 << 31), STM_EUNSUPPORTED, "%s: tensor too large for 32-bit offsets", who);
    STM_REQUIRE(out_np <= 0 || out_pixel_offset + M <= out_np, STM_EINVAL, "%s: output pixels [%d, %lld) exceed the planes (%d)", who, out_pixel_offset,
                (long long)(out_pixel_offset + M), out_np);
    FusedArgs a;
    a.x = x; a.om = offsets; a.wp = static_cast<const uint8_t*>(packed_weight); a.bias = bias; a.out = static_cast<uint8_t*>(out_planes);
    a.B = g->B; a.H = g->H; a.W = g->W; a.C = g->C; a.Ho = g->Ho; a.Wo = g->Wo; a.Cout = Cout;
    a.kh = g->kh; a.kw = g->kw; a.sh = g->sh; a.sw = g->sw; a.ph = g->ph; a.pw = g->pw; a.dh = g->dh; a.dw = g->dw;
    a.x_ld = x_ld; a.om_ld = om_ld;
    a.out_np = out_np > 0 ? out_np : (int)M; a.out_pix0 = out_pixel_offset;
    a.out_pstride = (out_plane_stride > 0 ? out_plane_stride : (long long)(Cout / 32) * a.out_np * 32) * 2;
    a.relu = relu; a.out_scale = out_scale > 0.0f ? out_scale : 1.0f; a.out_fmt = out_fmt;
    a.range_flag = stm_internal_range_flag();
    // tile shape: 64 pixels x 256 channels where the layer has them (every pixel is then sampled once, not once per 128 channels); STM_DCN_FUSED_WIDE=0: always 128 x 128
    const bool wide = Cout % 256 == 0 && STM_ENV_INT("STM_DCN_FUSED_WIDE", 1) != 0;
    const int TP = wide ? DfShape<1>::TP : DfShape<0>::TP, BN = wide ? DfShape<1>::BN : DfShape<0>::BN;
    int th = 0, tw = 0;
    if (th <= 0 || tw <= 0 || th * tw > TP) pick_patch(g->Ho, g->Wo, th, tw, TP);
    a.TH = th; a.TW = tw;
    a.tiles_y = stm_cdiv(g->Ho, th); a.tiles_x = stm_cdiv(g->Wo, tw);
    a.m_tiles = g->B * a.tiles_y * a.tiles_x; a.n_tiles = Cout / BN;
    a.K = K; a.cslabs = g->C / 32; a.slabs = K * a.cslabs;
    a.x_bytes = (unsigned)xb;
    const int npl = fmt == 1 ? 2 : 1;
    size_t lds = (size_t)DF_KMAX * TP * 32 + (size_t)DF_NS * npl * BN * 64 + (size_t)DF_NS * npl * TP * 64;   // coefficient tables, weight ring, operand ring
    const size_t park = (size_t)TP * (BN + 4) * sizeof(float);
    if (lds < park) lds = park;
    int rc;
    const hipStream_t hs = stm_hs(stream);
    if (wide) {
        if (npl == 2) rc = has_mask ? df_launch<2, true, 1>(a, lds, hs, who) : df_launch<2, false, 1>(a, lds, hs, who);
        else rc = has_mask ? df_launch<1, true, 1>(a, lds, hs, who) : df_launch<1, false, 1>(a, lds, hs, who);
    } else {
        if (npl == 2) rc = has_mask ? df_launch<2, true, 0>(a, lds, hs, who) : df_launch<2, false, 0>(a, lds, hs, who);
        else rc = has_mask ? df_launch<1, true, 0>(a, lds, hs, who) : df_launch<1, false, 0>(a, lds, hs, who);
    }
    if (rc != STM_OK) return rc;
    g_fused_launches.fetch_add(1, std::memory_order_relaxed);
    return STM_OK;
}
