// stem_fused.hip -- the ResNet stem in ONE kernel: conv1 (7x7, stride 2, padding 3, 3 -> 64 channels, eval BatchNorm folded into
// weight and bias) -> ReLU -> MaxPool2d(3, stride 2, padding 1), reference backbone.py:61-75 (ResNetBackbone.__init__ / forward:
// self.conv1, self.bn1, self.relu, self.maxpool) -- SURVEY.md section 8 row a2.  fp32 NHWC frame in, planar activations out (the
// format layer1's convolutions read: include/stmask_hip.h).
//
// Round 2 ran the stem as three launches: a row-patch tensor R (the 21 values a kernel row reads per output column, padded to a
// 32-channel slab: 503 MB written and read at batch 32), the (7 x 1) planar convolution over R (its fp32 output another 503 MB written
// and read), and the bias + ReLU + max-pool kernel: 2.2 GB of traffic, 0.78 ms per step, for a layer whose input is 94 MB and whose
// output is 126 MB.  Here a workgroup owns a tile of PH x PW pooled pixels:
//   1. the input patch it needs ((4 PH + 7) x (4 PW + 9) pixels) is read once, split into fp16 planes (x = h + l / 2048, as every
//      planar tensor) and laid out in LDS as [row][pixel][4 channels] -- RGB + a zero, 8 bytes per pixel and plane -- so that the 7
//      pixels a kernel row reads for a conv column are 28 consecutive fp16: ONE 16-byte-aligned K-slab of a 16x16x32 MFMA
//      (kx = 7 and channel 3 carry zero weights).  No im2col / row-patch tensor exists anywhere;
//   2. the conv outputs of the (2 PH + 1) x (2 PW + 1) positions the pool windows cover are computed transposed, D[channel][position]
//      = W x X^T: wave w keeps the weight fragments of channels 16 w .. 16 w + 15 for all 7 kernel rows in registers (56 VGPRs, loaded
//      once) and reads activation fragments from LDS;
//   3. the 3x3 / stride-2 max runs on the accumulators: across positions with wavefront shuffles inside the 16-lane groups,
//      across conv rows as running maxima in registers (positions outside the conv output are -inf: MaxPool2d's padding); then
//      + bias, ReLU -- monotone and per channel, so they commute with the max exactly -- split, plane stores.  (A first version parked
//      the conv tile in LDS for the pool: 155 KB of LDS, one workgroup per CU, every phase's latency exposed: 400 us at batch 32.)
// Arithmetic: fp32-equivalent like the planar convolutions (three MFMA products per reference product: w_h x_h + (w_h x_l + w_l x_h) /
// 2048, fp32 accumulation, power-of-two weight scale removed after the sum); format 2 keeps one plane / one product.
#include "planar_common.h"

#include <atomic>

namespace {

constexpr int ST_PH = 4, ST_PW = 23;                 // pooled pixels per workgroup: 2 PW + 1 = 47 conv columns = 3 MFMA tiles of 16
constexpr int ST_CR = 2 * ST_PH + 1;                 // conv rows per workgroup (9)
constexpr int ST_CT = (2 * ST_PW + 1 + 15) / 16;     // conv column tiles per conv row (3)
constexpr int ST_CC = 16 * ST_CT;                    // conv columns computed per row (48; the last one is never read)
constexpr int ST_IR = 2 * ST_CR + 5;                 // input rows of the patch (23)
constexpr int ST_IS = 2 * ST_CC + 6;                 // input pixel slots per patch row (102): slot s = input column ix0 + s
constexpr int ST_ROWB = ST_IS * 8;                   // bytes per patch row and plane
constexpr int ST_XPL = ST_IR * ST_ROWB;              // bytes per plane of the patch
constexpr int ST_MAX_DEVICES = 32;

struct StemArgs {
    const float* x;          // [B][H][W][3]
    const uint8_t* wp;       // [Cout/16][7][plane][64 lanes][8 fp16]: every lane's A fragment, contiguous
    const float* bias;       // [Cout] or null
    uint8_t* out;            // [planes][Cout/32][B*Hp*Wp][32]
    int B, H, W, Hc, Wc, Hp, Wp;
    int tiles_y, tiles_x;
    long long out_pstride;   // bytes between output planes
    float out_scale;
    int out_fmt;
    int* range_flag;
};

template <int NPL>
__global__ __launch_bounds__(256) void stem_fused_kernel(const StemArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __align__(16) uint8_t smem[];
    uint8_t* xin = smem;                                         // [NPL][IR][IS][4] fp16
    const int64_t nblk = (int64_t)a.B * a.tiles_y * a.tiles_x;
    const int64_t blk = stm_xcd_block(nblk);
    if (blk < 0) return;
    const int tx = (int)(blk % a.tiles_x), ty = (int)((blk / a.tiles_x) % a.tiles_y), b = (int)(blk / ((int64_t)a.tiles_x * a.tiles_y));
    const int py0 = ty * ST_PH, px0 = tx * ST_PW;
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;              // first conv row / column of the tile
    const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;              // first input row / column of the patch
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- weight fragments of this wave's 16 channels, all 7 kernel rows: registers for the whole kernel
    f16x8 wh[7], wl[7];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
        const uint8_t* src = a.wp + (((size_t)wave * 7 + ky) * NPL) * 1024 + lane * 16;
        wh[ky] = *reinterpret_cast<const f16x8*>(src);
        if constexpr (NPL == 2) wl[ky] = *reinterpret_cast<const f16x8*>(src + 1024);
    }

    // ---- 1. input patch -> fp16 planes in LDS, [row][slot][4]; outside the frame: zeros (the convolution's padding)
    const float* xb = a.x + (size_t)b * a.H * a.W * 3;
    for (int idx = tid; idx < ST_IR * ST_IS; idx += 256) {
        const int r = idx / ST_IS, s = idx - r * ST_IS;
        const int iy = iy0 + r, ix = ix0 + s;
        float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f;
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
            const float* p = xb + ((size_t)iy * a.W + ix) * 3;
            v0 = p[0]; v1 = p[1]; v2 = p[2];
        }
        unsigned h0, h1, l0, l1;
        split2_f16(f32x2{v0, v1}, h0, l0);
        split2_f16(f32x2{v2, 0.0f}, h1, l1);
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2*>(xin + r * ST_ROWB + s * 8) = u32x2{h0, h1};
        if constexpr (NPL == 2) *reinterpret_cast<u32x2*>(xin + ST_XPL + r * ST_ROWB + s * 8) = u32x2{l0, l1};
        if (a.range_flag) {
            const unsigned m = max(max(__builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1)) & 0x7fffffffu,
                                   __builtin_bit_cast(unsigned, v2) & 0x7fffffffu);
            if (m > 0x477fe000u) *reinterpret_cast<volatile int*>(a.range_flag) = 1;
        }
    }
    __syncthreads();

    // ---- 2. conv positions, one conv row (3 column tiles) at a time.  B fragment of lane (position r16 of tile ct, chunk kc), kernel
    // row ky: 8 fp16 = pixels 2 kc, 2 kc + 1 of the 7-pixel window of conv column 16 ct + r16, i.e. patch slots 2 (16 ct + r16) + 2 kc ..:
    // byte offset 16 (16 ct + r16 + kc) -- always 16-byte aligned because the patch starts at an odd input column.
    // ---- 3. the pool runs on the accumulators, no parked tile: a lane holds 4 channels (16 wave + 4 kc ..) of conv position (row cr,
    // column 16 ct + r16).  Horizontally, pooled column pc = 8 ct + r16 / 2 (even r16) covers r16, r16 + 1, r16 + 2 of the same register
    // -- lanes r16 + 1, r16 + 2 of the 16-lane group, or lanes 0 / 1 of the NEXT tile's register for r16 = 14 -- fetched with wavefront
    // shuffles; vertically, pooled row pr = max of conv rows 2 pr, 2 pr + 1, 2 pr + 2, kept as running maxima in registers.  Positions
    // outside the conv output are -inf (MaxPool2d's padding).
    const int r16 = lane & 15, kc = lane >> 4;
    const int boff = 16 * (r16 + kc);
    const float ls = NPL == 2 ? 1.0f / STM_F16_LOW_SCALE : 0.0f;
    const float NEG = -__builtin_inff();
    bool colok[ST_CT];
#pragma unroll
    for (int ct = 0; ct < ST_CT; ++ct) colok[ct] = (unsigned)(cx0 + 16 * ct + r16) < (unsigned)a.Wc;
    float b4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) b4[r] = a.bias ? a.bias[16 * wave + 4 * kc + r] : 0.0f;
    const size_t n_out = (size_t)a.B * a.Hp * a.Wp;
    const int ch0 = 16 * wave + 4 * kc;
    f32x4 top[ST_CT];                     // horizontally pooled values of conv row 2 pr (carried from the previous pooled row's last row)
    f32x4 run[ST_CT];                     // running maximum of the current pooled row
#pragma unroll
    for (int ct = 0; ct < ST_CT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) { top[ct][r] = NEG; run[ct][r] = NEG; }
    for (int cr = 0; cr < ST_CR; ++cr) {
        f32x4 acc[ST_CT], accl[ST_CT];
#pragma unroll
        for (int ct = 0; ct < ST_CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[ct][r] = 0.0f; accl[ct][r] = 0.0f; }
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            const uint8_t* row = xin + (2 * cr + ky) * ST_ROWB + boff;
#pragma unroll
            for (int ct = 0; ct < ST_CT; ++ct) {
                const f16x8 xh = *reinterpret_cast<const f16x8*>(row + 256 * ct);
                if constexpr (NPL == 2) {
                    const f16x8 xl = *reinterpret_cast<const f16x8*>(row + ST_XPL + 256 * ct);
                    accl[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ky], xl, accl[ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ky], xh, acc[ct], 0, 0, 0);
                    accl[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ky], xh, accl[ct], 0, 0, 0);
                } else {
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ky], xh, acc[ct], 0, 0, 0);
                }
            }
        }
        const bool rowok = (unsigned)(cy0 + cr) < (unsigned)a.Hc;       // wave-uniform
        f32x4 v[ST_CT], hp[ST_CT];
#pragma unroll
        for (int ct = 0; ct < ST_CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float c = (NPL == 2 ? acc[ct][r] + accl[ct][r] * ls : acc[ct][r]) * a.out_scale;
                v[ct][r] = (rowok && colok[ct]) ? c : NEG;
            }
        // horizontal 3-max at even positions: neighbours r16 + 1, r16 + 2 (the next tile's positions 0, 1 beyond position 15)
#pragma unroll
        for (int ct = 0; ct < ST_CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float nx = ct + 1 < ST_CT ? v[ct + 1 < ST_CT ? ct + 1 : ct][r] : NEG;
                float n1 = __shfl(v[ct][r], (r16 + 1) & 15, 16), n2 = __shfl(v[ct][r], (r16 + 2) & 15, 16);
                const float w1 = __shfl(nx, (r16 + 1) & 15, 16), w2 = __shfl(nx, (r16 + 2) & 15, 16);
                if (r16 + 1 > 15) n1 = w1;
                if (r16 + 2 > 15) n2 = w2;
                hp[ct][r] = fmaxf(fmaxf(v[ct][r], n1), n2);
            }
        // vertical: conv row cr is row (cr - 2 pr) of pooled row pr = cr / 2 (rows 0, 1) and row 2 of pooled row pr - 1
        const bool even = (cr & 1) == 0;
#pragma unroll
        for (int ct = 0; ct < ST_CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) run[ct][r] = fmaxf(run[ct][r], hp[ct][r]);
        if (even && cr > 0) {
            // pooled row pr = cr / 2 - 1 is complete: bias + ReLU + split + store (8 bytes per lane and plane: 4 channels of one pixel)
            const int pr = cr / 2 - 1, py = py0 + pr;
#pragma unroll
            for (int ct = 0; ct < ST_CT; ++ct) {
                const int pc = 8 * ct + (r16 >> 1), px = px0 + pc;
                if ((r16 & 1) == 0 && pc < ST_PW && py < a.Hp && px < a.Wp) {
                    float o[4];
                    unsigned m4 = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float t = run[ct][r] + b4[r];
                        o[r] = t > 0.0f ? t : 0.0f;
                        m4 = max(m4, __builtin_bit_cast(unsigned, o[r]) & 0x7fffffffu);
                    }
                    if (m4 > 0x477fe000u && a.range_flag) *reinterpret_cast<volatile int*>(a.range_flag) = 1;
                    unsigned h0, h1, l0, l1;
                    split2_f16(f32x2{o[0], o[1]}, h0, l0);
                    split2_f16(f32x2{o[2], o[3]}, h1, l1);
                    const size_t pix = ((size_t)b * a.Hp + py) * a.Wp + px;
                    uint8_t* dst = a.out + (((size_t)(ch0 >> 5) * n_out + pix) * 32 + (ch0 & 31)) * 2;
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
                    if (a.out_fmt == 1) *reinterpret_cast<u32x2*>(dst + a.out_pstride) = u32x2{l0, l1};
                }
            }
            // the row just finished is also the first row of the next pooled row
#pragma unroll
            for (int ct = 0; ct < ST_CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) run[ct][r] = hp[ct][r];
        }
    }
    (void)top;
#endif
}

// weight [Cout][3][7][7] fp32 -> [Cout/16][ky][plane][lane][8 fp16]: lane (r16, kc) holds k = 8 kc + e <-> pixel kx = 2 kc + e / 4,
// channel e % 4 of kernel row ky for output channel 16 n + r16; kx = 7 and channel 3 are zero
__global__ __launch_bounds__(256) void stem_pack_kernel(const float* __restrict__ w, uint8_t* __restrict__ wp, int Cout, int npl, float wscale)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int total = (Cout / 16) * 7 * 64;
    if (idx >= total) return;
    const int lane = idx & 63, ky = (idx >> 6) % 7, n = idx / (64 * 7);
    const int r16 = lane & 15, kc = lane >> 4;
    const int co = 16 * n + r16;
    unsigned pl[2][4];
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
        float v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e = 2 * e2 + h, kx = 2 * kc + e / 4, c = e % 4;
            v[h] = (kx < 7 && c < 3) ? w[(((size_t)co * 3 + c) * 7 + ky) * 7 + kx] : 0.0f;
        }
        split2_f16(f32x2{v[0], v[1]} * wscale, pl[0][e2], pl[1][e2]);
    }
    uint8_t* dst = wp + (((size_t)n * 7 + ky) * npl) * 1024 + lane * 16;
    for (int p = 0; p < npl; ++p) *reinterpret_cast<u32x4*>(dst + p * 1024) = u32x4{pl[p][0], pl[p][1], pl[p][2], pl[p][3]};
}

}  // namespace

extern "C" size_t stm_stem_packed_weight_bytes(int Cout, int fmt)
{
    if (Cout != 64 || (fmt != 1 && fmt != 2)) return 0;
    return (size_t)(Cout / 16) * 7 * (fmt == 1 ? 2 : 1) * 1024;
}

extern "C" int stm_stem_pack_weights_f32(const float* weight, void* packed, int Cout, int fmt, float wscale, stm_stream_t stream)
{
    STM_REQUIRE(weight && packed, STM_ENULL, "stm_stem_pack_weights_f32: weight/packed must be non-NULL");
    STM_REQUIRE(stm_stem_packed_weight_bytes(Cout, fmt) > 0, STM_EUNSUPPORTED, "stm_stem_pack_weights_f32: Cout must be 64 and fmt 1 or 2 (got %d, %d)", Cout, fmt);
    STM_REQUIRE((uintptr_t)packed % 16 == 0 && wscale > 0.0f && wscale < 3.0e38f, STM_EINVAL, "stm_stem_pack_weights_f32: alignment / weight scale");
    const int total = (Cout / 16) * 7 * 64;
    hipLaunchKernelGGL(stem_pack_kernel, dim3(stm_cdiv(total, 256)), dim3(256), 0, stm_hs(stream), weight, static_cast<uint8_t*>(packed), Cout,
                       fmt == 1 ? 2 : 1, wscale);
    STM_CHECK_LAUNCH("stem_pack_kernel");
    return STM_OK;
}

extern "C" int stm_stem_fused_f32(const float* x, const void* packed_weight, const float* bias, void* out_planes, int B, int H, int W, int Cout,
                                  int fmt, int out_fmt, float out_scale, stm_stream_t stream)
{
    const char* who = "stm_stem_fused_f32";
    STM_REQUIRE(x && packed_weight && out_planes, STM_ENULL, "%s: x/packed_weight/out_planes must be non-NULL", who);
    STM_REQUIRE(Cout == 64 && (fmt == 1 || fmt == 2), STM_EUNSUPPORTED, "%s: 64 output channels and plane format 1 or 2 (got %d, %d)", who, Cout, fmt);
    STM_REQUIRE(out_fmt == fmt || (fmt == 2 && out_fmt == 1), STM_EINVAL, "%s: output format %d cannot be produced by a format-%d layer", who, out_fmt, fmt);
    STM_REQUIRE(B > 0 && H >= 7 && W >= 7 && (int64_t)B * H * W * 3 < ((int64_t)1 << 40), STM_EINVAL, "%s: bad frame batch", who);
    STM_REQUIRE((uintptr_t)packed_weight % 16 == 0 && (uintptr_t)out_planes % 16 == 0, STM_EINVAL, "%s: 16-byte alignment required", who);
    StemArgs a;
    a.x = x; a.wp = static_cast<const uint8_t*>(packed_weight); a.bias = bias; a.out = static_cast<uint8_t*>(out_planes);
    a.B = B; a.H = H; a.W = W;
    a.Hc = (H + 6 - 7) / 2 + 1; a.Wc = (W + 6 - 7) / 2 + 1;          // conv1: 7x7, stride 2, padding 3
    a.Hp = (a.Hc - 1) / 2 + 1; a.Wp = (a.Wc - 1) / 2 + 1;            // MaxPool2d(3, 2, 1), floor mode
    a.tiles_y = stm_cdiv(a.Hp, ST_PH); a.tiles_x = stm_cdiv(a.Wp, ST_PW);
    a.out_pstride = (long long)(Cout / 32) * B * a.Hp * a.Wp * 64;
    a.out_scale = out_scale > 0.0f ? out_scale : 1.0f;
    a.out_fmt = out_fmt;
    a.range_flag = stm_internal_range_flag();
    const int npl = fmt == 1 ? 2 : 1;
    const size_t lds = (size_t)npl * ST_XPL;
    static std::atomic<bool> reserved[2][ST_MAX_DEVICES];
    int dev = 0;
    const bool have_dev = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < ST_MAX_DEVICES;
    if (!have_dev || !reserved[npl - 1][dev].load(std::memory_order_relaxed)) {
        const void* fn = npl == 2 ? reinterpret_cast<const void*>(stem_fused_kernel<2>) : reinterpret_cast<const void*>(stem_fused_kernel<1>);
        STM_REQUIRE(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess, STM_ELAUNCH,
                    "%s: cannot reserve %zu bytes of LDS", who, lds);
        if (have_dev) reserved[npl - 1][dev].store(true, std::memory_order_relaxed);
    }
    const int64_t nblk = (int64_t)B * a.tiles_y * a.tiles_x;
    if (npl == 2) hipLaunchKernelGGL(stem_fused_kernel<2>, dim3(stm_xcd_grid(nblk)), dim3(256), lds, stm_hs(stream), a);
    else hipLaunchKernelGGL(stem_fused_kernel<1>, dim3(stm_xcd_grid(nblk)), dim3(256), lds, stm_hs(stream), a);
    STM_CHECK_LAUNCH("stem_fused_kernel");
    return STM_OK;
}
