// conv_kxr.hip -- stride-1 kh x kw convolution (kw >= 3) with FEW output channels per group (<= 64) on the 16-bit matrix cores, in
// the planar operand formats of conv_bf16x.hip (format 1: two fp16 planes / three products, fp32-equivalent; format 2: one fp16
// plane), for the layers whose time is set by the L2 -> LDS staging rate rather than by the matrix pipe:
//   * the shared head's output layers (prediction_head_FC.py:146-195: conf 41 / centerness + bbox 5 / mask 32 channels, three
//     kernel shapes 3x3, 3x5, 5x3, each group reading its own 256-channel tower) -- SURVEY.md section 8 rows a5 / f4;
//   * the DCN layers' 27-channel offset / mask convolutions at stride 1 (backbone.py:20-26), the 64 -> 64 3x3 convolutions of layer1
//     (backbone.py:38-58) -- rows a1 / a2.
// On conv_planar_kernel's 128 x 64 tiles these layers stage 24 KB per K-slab and workgroup for at most 48 matrix instructions per
// wave and run at the rate the CUs can pull rows from L2 into LDS (17 TB/s chip-wide: 8.8 GB per launch of the 3x3 / 3x5 output
// layers at batch 32, 510 us).  This kernel moves 3-4x fewer bytes:
//   * kx-reuse: per (channel slab, ky) the BM + kw - 1 consecutive input pixels of a FLAT run of BM = 256 output pixels are
//     staged ONCE and serve all kw taps -- tap kx of output pixel m is staged row (m - m0) + kx.  Where that neighbour lies in
//     another image row (x + kx - pw outside [0, W)) the lane reads an always-zero LDS row instead; rows of the ky-shifted run that
//     fall outside the image are zero-filled by the DMA's range check.  No masking instruction in the loop;
//   * weight tiles as wide as the group's real channels (16 * NC, NC = 1..4) instead of 64, and 256 pixels per tile;
//   * every wave multiplies: the four waves split the PIXELS (64 each) and take all NC channel tiles, computed transposed --
//     D[channel][pixel] = W x X^T, weights as the A operand -- so a lane ends with 4 consecutive channels of one pixel and the
//     epilogue stores 16-byte fp32 vectors (or 8-byte plane pieces) straight from registers, no LDS round trip.
// Groups with different NC share one launch (job table; the heavy groups' tiles are dealt first).
// Same arithmetic as conv_planar_kernel: per K-slab acc += w_h x_h, accl += w_h x_l then w_l x_h, K order (channel slab, ky, kx),
// fp32 accumulation, acc + accl / 2048, power-of-two weight scale taken out in the epilogue.  tests/test_gpu_conv.py holds both
// kernels to the same bound against the fp64 oracle.
#include "planar_common.h"

#include <algorithm>
#include <atomic>

namespace {

constexpr int KX_CONSUMERS = 4, KX_PRODUCERS = 2, KX_THREADS = 64 * (KX_CONSUMERS + KX_PRODUCERS);
constexpr int KX_MAX_JOBS = 4;
constexpr int KX_MAX_DEVICES = 32;
constexpr int KX_LDS_MAX = 160 * 1024;

// Tile shape of a (kw, planes, channel tiles) combination: each of the 4 consumer waves takes PT pixel tiles of 16 (workgroup tile
// BM = 64 PT flat pixels), the LDS holds a ring of D stages.  A stage (one (channel slab, ky): BM + 16 staged pixel rows per plane and
// the kw weight tiles) must be fetched two stages ahead of its use -- the first touch of the activations comes from HBM, and a stage
// is what a CU has to keep in flight to cover that latency at the L2 -> LDS rate -- so D >= 3 decides PT: the widest tile whose ring fits.
constexpr int kx_stage_bytes(int kw, int npl, int nc, int pt) { return npl * (64 * pt + 16) * 64 + kw * npl * nc * 1024; }
constexpr int kx_pt(int kw, int npl, int nc)
{
    if (nc >= 4 && npl == 2) return 2;     // four channel tiles: 64 accumulator registers per wave leave room for the second fragment set
#ifndef KX_PT_MAX
#define KX_PT_MAX 4      // (experiments: narrower tiles leave room for a deeper ring)
#endif
    for (int pt = KX_PT_MAX; pt > 2; --pt)
        if (3 * kx_stage_bytes(kw, npl, nc, pt) <= KX_LDS_MAX) return pt;
    return 2;
}
constexpr int kx_depth(int kw, int npl, int nc)
{
    const int d = KX_LDS_MAX / kx_stage_bytes(kw, npl, nc, kx_pt(kw, npl, nc));
    return d > 4 ? 4 : d;
}

struct KxrJob {
    const uint8_t* wp;     // packed weights of this group: [stage = (channel slab, ky)][kx][plane][16 * nc rows][64 B, chunk-swizzled]
    const float* bias;     // bias of the group's channel 0 (bias_n entries) or null
    int bias_n;
    int x_slab0;           // first input channel slab of the group
    int out_ch0;           // first output channel of the group in the output tensors
    int nc;                // 16-channel tiles
    int tile0;             // first tile id of this job (ids are padded to a multiple of 8 per job)
    int per_xcd;           // tile ids of this job per XCD: id -> tile (id & 7) * per_xcd + (id >> 3)
    int lvl_tile0[9];      // first tile of each level, in this job's tile size
};

struct KxrArgs {
    const uint8_t* xp;     // [NPL][slabs][x_np][32] fp16 planes
    float* out_f32;        // [M][out_ld] or null
    uint8_t* out_pl;       // [planes][Cout/32][out_np][32] or null
    KxrJob job[KX_MAX_JOBS];
    int n_jobs, total;     // workgroups of all jobs
    int cslabs, kh, ph, pw;
    int x_np, out_ld, out_np;
    long long x_pstride, out_pstride;     // bytes between planes
    unsigned plane_bytes;
    int relu, out_fmt;
    float out_scale;
    int* range_flag;
    int n_lvl;
    int lvl_start[9], lvl_h[8], lvl_w[8];
};

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

#define KX_MM(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x32_f16(a_, b_, c_, 0, 0, 0)
#ifndef KX_ABL
#define KX_ABL 0      // diagnostic builds (make EXTRA=-DKX_ABL=n, RESULTS ARE WRONG): 1 no DMA, 2 no fragment reads / MFMAs, 4 no activation DMA, 8 no weight DMA
#endif

// Workgroup = 4 consumer waves (16 PT pixels each, all NC channel tiles: LDS fragment reads and MFMAs, nothing else in their loop) + 2
// producer waves that issue every LDS-DMA of the workgroup.  With one consumer wave per SIMD an LDS-DMA issued from the MFMA stream
// stalls that SIMD's matrix pipe for the ~60 cycles the instruction takes to issue (15 of them per stage and wave: a third of the
// stage); a wave that does nothing else issues one every ~25 cycles.  Ring of D stage buffers, one barrier per stage:
//   producers:  DMA(stage s) -> wait until stage s - (D - 2) has landed (the later ones stay in flight) -> barrier s - (D - 2)
//   consumers:  barrier k -> fragments + MFMAs of stage k
// DMA(s) goes into the buffer of stage s - D, which every consumer left before it reached barrier s - D + 1 -- the last barrier the
// producers passed before issuing it.
template <int KW, int NPL, int NC>
__device__ __forceinline__ void kxr_body(const KxrArgs& a, const KxrJob& jb, int tile_first, int n_seq, int tile_step, uint8_t* smem)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int PT = kx_pt(KW, NPL, NC), D = kx_depth(KW, NPL, NC);
    constexpr int BM = 64 * PT, XROWS = BM + 16, NRG = XROWS / 16;
    constexpr int HALO = KW - 1;
    constexpr int XPL = XROWS * 64;             // bytes of one plane of the staged rows
    constexpr int XBUF = NPL * XPL;
    constexpr int WT = NC * 16 * 64;            // bytes of one (kx, plane) weight tile
    constexpr int WBUF = KW * NPL * WT;
    constexpr int BUF = XBUF + WBUF;
    constexpr int NWD = KW * NPL * NC;          // weight DMA instructions per stage (1 KB each)
    constexpr int NXD = NRG * NPL;              // activation DMA instructions per stage
    constexpr int XDW = (NXD + KX_PRODUCERS - 1) / KX_PRODUCERS, WDW = (NWD + KX_PRODUCERS - 1) / KX_PRODUCERS;
    constexpr int NPD = XDW + WDW;              // DMA instructions per producer wave and stage
    static_assert(BM + HALO <= XROWS - 1, "the always-zero row must exist");
    static_assert(D >= 2 && D * BUF <= KX_LDS_MAX && (D - 2) * NPD <= 63, "ring does not fit");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = a.cslabs * a.kh;
    // The workgroup is persistent over a sequence of this job's tiles (tile_first, tile_first + tile_step, ...: n_seq of them): the ring
    // of stages runs straight through the tile boundaries, so the producers are already fetching the next tile while the consumers
    // store the previous one -- per tile only the setup of the new pixel run remains of the workgroup's start-up cost.
    struct TileGeo { int H, W, lstart, lend, m0; };
    // Tile ids are dealt round-robin to the 8 XCDs (id & 7 = blockIdx.x & 7: the grid and every job's first id are multiples of 8); the
    // map gives each XCD -- each L2 -- a contiguous run of the job's tiles, so the rows a tile's ky stages share with its neighbours are
    // filled into one L2 once.  Ids of the padding map to tiles behind the last level: every access of such a tile is out of range.
    auto tile_geo = [&](int id) {
        const int tile = (id & 7) * jb.per_xcd + (id >> 3);
        int lvl = 0;
#pragma unroll
        for (int l = 1; l < 8; ++l)
            if (l < a.n_lvl && tile >= jb.lvl_tile0[l]) lvl = l;
        TileGeo g;
        g.H = a.lvl_h[lvl]; g.W = a.lvl_w[lvl]; g.lstart = a.lvl_start[lvl]; g.lend = a.lvl_start[lvl + 1];
        g.m0 = g.lstart + (tile - jb.lvl_tile0[lvl]) * BM;
        return g;
    };

    if (wave >= KX_CONSUMERS) {
        // ---------------------------------------------------------------------------------------------------------- producer
        // activation pieces p * NRG + rg (plane p, row group rg) and weight pieces are dealt round-robin to the producers.  Staged row
        // j = input pixel m0 - pw + j of the ky-shifted run; per row: byte offset of ky = 0 within a channel slab, one validity bit per
        // ky (the shifted pixel lies in the same image; everything else, and rows >= BM + kw - 1, is zero-filled by the range check)
        const int pw_ = wave - KX_CONSUMERS;
        __amdgpu_buffer_rsrc_t xr[NPL];
#pragma unroll
        for (int p = 0; p < NPL; ++p)
            xr[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.xp) + (size_t)p * a.x_pstride, 0, (int)a.plane_bytes, 0x00020000);
        int dbase[XDW];
        unsigned dmask[XDW];
        int W = 1;
        auto setup = [&](int tile) {
            const TileGeo tg = tile_geo(tile);
            W = tg.W;
            const int HW = tg.H * tg.W;
#pragma unroll
            for (int i = 0; i < XDW; ++i) {
                const int piece = min(pw_ + KX_PRODUCERS * i, NXD - 1);
                const int rg = piece % NRG;
                const int j = rg * 16 + (lane >> 2);
                const int q = tg.m0 - a.pw + j;
                const bool okq = j < BM + HALO && q >= tg.lstart && q < tg.lend;
                const int local = okq ? q - tg.lstart : 0;
                const int y = (local % HW) / tg.W;
                unsigned vm = 0;
                for (int ky = 0; ky < a.kh; ++ky)
                    if ((unsigned)(y + ky - a.ph) < (unsigned)tg.H) vm |= 1u << ky;
                dmask[i] = okq ? vm : 0u;
                dbase[i] = (q - a.ph * tg.W) * 64 + (((lane & 3) ^ swz(j)) << 4);
            }
        };
        int cs = 0, ky = 0, buf = 0, s_in = 0, t_in = 0;       // (channel slab, ky), ring slot, stage and tile of the next stage to issue
        auto issue = [&]() {
            if (s_in == 0) setup(tile_first + t_in * tile_step);
            uint8_t* xb = smem + buf * BUF;
            const int uni = (jb.x_slab0 + cs) * (a.x_np * 64) + ky * (W * 64);
#pragma unroll
            for (int i = 0; i < XDW; ++i) {
                const int piece = min(pw_ + KX_PRODUCERS * i, NXD - 1);       // (an uneven split fetches its last piece twice)
                const int p = piece / NRG, rg = piece - p * NRG;
                const unsigned oob = ((dmask[i] >> ky) & 1u) ^ 1u;
                const unsigned off = (unsigned)(dbase[i] + uni) | (oob << 31);
                if (!(KX_ABL & 5)) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr[NPL == 2 ? (p & 1) : 0], (lds_ptr)(xb + p * XPL + rg * 1024), 16, off, 0, 0, 0);
            }
            uint8_t* wb = xb + XBUF;
            const uint8_t* wsrc = jb.wp + (size_t)s_in * WBUF;
#pragma unroll
            for (int k = 0; k < WDW; ++k) {
                const int idx = min(pw_ + KX_PRODUCERS * k, NWD - 1);
                if (!(KX_ABL & 9)) __builtin_amdgcn_global_load_lds((glb_ptr)(wsrc + idx * 1024 + lane * 16), (lds_ptr)(wb + idx * 1024), 16, 0, 0);
            }
            if (++ky == a.kh) { ky = 0; ++cs; }
            if (++buf == D) buf = 0;
            if (++s_in == S) { s_in = 0; cs = 0; ky = 0; ++t_in; }
        };
        const int G = S * n_seq;                             // stages of the whole sequence
#if defined(KX_SAFE_WAIT)
        // (make EXTRA=-DKX_SAFE_WAIT) cross-check build: vmcnt(0) for stage g - 1 BEFORE stage g goes out.  Round 3 believed the counted wait
        // below could hand over a stage that had not landed; scripts/ldsdma_order_probe.hip shows it cannot (LDS-DMA completes in issue order,
        // mixed buffer_load ... lds / global_load_lds included: 0 stale words in 3.1e9 per variant, profiles/r04_ldsdma_order_probe.txt).  What
        // that form really did was delay the restaging of a slot the consumers were still reading -- see the consumers' barrier.
        if (G > 0) issue();
        for (int g = 1; g < G; ++g) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            issue();
            asm volatile("s_barrier" ::: "memory");
        }
        if (G > 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        return;
#else
        // head: D - 2 stages go out before the first wait
        for (int g = 0; g < D - 2 && g < G; ++g) issue();
        for (int g = D - 2; g < G; ++g) {
            issue();
            // stage g - (D - 2) has landed; the D - 2 stages behind it stay in flight across the barrier
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((D - 2) * NPD) : "memory");
        }
        // tail: the last D - 2 stages (issued above, or all of them when G < D - 1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int k = max(G - (D - 2), 0); k < G; ++k) asm volatile("s_barrier" ::: "memory");
        return;
#endif
    }

    // -------------------------------------------------------------------------------------------------------------- consumer
    // B operand (activations): lane (pixel r16 of pixel tile t, chunk kc), tap kx -> staged row + kx, or the always-zero row where the
    // tap leaves the image row.  A operand (weights): lane (channel r16 of tile c, chunk kc).
    const int r16 = lane & 15, kc = lane >> 4;
    const int aoff = XBUF + lds_off(r16, kc);
    int boff[PT][KW];
    f32x4 acc[NC][PT], accl[NC][PT];

    constexpr int NRD = (PT + NC) * NPL;        // fragment reads per tap
    constexpr int NM = PT * NC * (NPL == 2 ? 3 : 1);   // MFMAs per tap
    constexpr bool PF = NC * PT <= 12 && !(NC == 4 && PT == 3);     // the second fragment set must fit beside the accumulators
    constexpr int NS = PF ? 2 : 1;
    f16x8 bh[NS][PT], bl[NS][PT], ah[NS][NC], al[NS][NC];
    auto read_frags = [&](const uint8_t* xs, int kx, int set) {
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            bh[set][t] = *reinterpret_cast<const f16x8*>(xs + boff[t][kx]);
            if constexpr (NPL == 2) bl[set][t] = *reinterpret_cast<const f16x8*>(xs + XPL + boff[t][kx]);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            ah[set][c] = *reinterpret_cast<const f16x8*>(xs + aoff + ((kx * NPL) * NC + c) * 1024);
            if constexpr (NPL == 2) al[set][c] = *reinterpret_cast<const f16x8*>(xs + aoff + ((kx * NPL + 1) * NC + c) * 1024);
        }
    };
    auto mfmas = [&](int set) {
        if constexpr (NPL == 2) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int t = 0; t < PT; ++t) accl[c][t] = KX_MM(ah[set][c], bl[set][t], accl[c][t]);
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int t = 0; t < PT; ++t) acc[c][t] = KX_MM(ah[set][c], bh[set][t], acc[c][t]);
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int t = 0; t < PT; ++t) accl[c][t] = KX_MM(al[set][c], bh[set][t], accl[c][t]);
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int t = 0; t < PT; ++t) acc[c][t] = KX_MM(ah[set][c], bh[set][t], acc[c][t]);
        }
    };
    // bias of this lane's channels, once per job: loaded in the epilogue they cost four dependent global round trips per tile (~5 us on
    // six-stage tiles: more than the tile's MFMAs)
    float b4[NC][4];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = 16 * c + 4 * kc + r;
            b4[c][r] = (jb.bias && ch < jb.bias_n) ? jb.bias[ch] : 0.0f;
        }
    int buf = 0;
    for (int t_seq = 0; t_seq < n_seq; ++t_seq) {
    const TileGeo tg = tile_geo(tile_first + t_seq * tile_step);
    const int lstart = tg.lstart, lend = tg.lend, m0 = tg.m0, W = tg.W;
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        const int row0 = 16 * PT * wave + 16 * t + r16;
        const int m = m0 + row0;
        const bool okm = m < lend;
        const int x = okm ? (m - lstart) % W : 0;
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            const bool v = okm && (unsigned)(x + kx - a.pw) < (unsigned)W;
            boff[t][kx] = lds_off(v ? row0 + kx : BM + HALO, kc);
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int t = 0; t < PT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[c][t][r] = 0.0f; accl[c][t][r] = 0.0f; }
    for (int s = 0; s < S; ++s) {
        // barrier s: stage s is in LDS, and every consumer has LEFT stage s - 1 -- which takes the lgkmcnt(0) in front of the barrier: the
        // producers restage that slot as soon as the barrier opens, and a fragment read that is still queued in the LDS pipe then returns
        // the NEW stage's bytes (measured on conv_chain.hip's copy of this ring: 34 wrong outputs in 19 200 launches beside a second
        // process without the drain, 0 with it; profiles/r04_ring_stress_chain_variants.txt).  The compiler sinks the stage's last MFMAs
        // below an undrained barrier together with the lgkmcnt wait of their fragments.
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const uint8_t* xs = smem + buf * BUF;
        if (++buf == D) buf = 0;
        if constexpr ((KX_ABL & 2) != 0) continue;
        if constexpr (PF) {
            read_frags(xs, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
                const int cur = kx & 1;
                if (kx + 1 < KW) read_frags(xs, kx + 1, cur ^ 1);      // the next tap's fragments arrive under this tap's MFMAs
                mfmas(cur);
                // schedule: one fragment read of the next tap behind each of the first MFMAs
#pragma unroll
                for (int k = 0; k < NM; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (kx + 1 < KW && k < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
                read_frags(xs, kx, 0);
                mfmas(0);
            }
        }
    }

    // ---- epilogue, straight from the accumulators: lane holds channels 16 c + 4 kc .. + 3 of pixel m0 + 16 PT wave + 16 t + r16
    const float ls = NPL == 2 ? 1.0f / STM_F16_LOW_SCALE : 0.0f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = 16 * c + 4 * kc;                    // channel within the group
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            const int m = m0 + 16 * PT * wave + 16 * t + r16;
            if (m >= lend) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sum = NPL == 2 ? acc[c][t][r] + accl[c][t][r] * ls : acc[c][t][r];
                v[r] = __builtin_fmaf(sum, a.out_scale, b4[c][r]);
                if (a.relu) v[r] = __builtin_fmaxf(v[r], 0.0f);
            }
            const int oc = jb.out_ch0 + ch;
            if ((KX_ABL & 16) && v[0] != 12345.678f) continue;      // ablation: no output stores
            if (a.out_f32) *reinterpret_cast<f32x4*>(a.out_f32 + (size_t)m * a.out_ld + oc) = f32x4{v[0], v[1], v[2], v[3]};
            if (a.out_pl) {
                unsigned m4 = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) m4 = max(m4, __builtin_bit_cast(unsigned, v[r]) & 0x7fffffffu);
                if (m4 > 0x477fe000u && a.range_flag) *reinterpret_cast<volatile int*>(a.range_flag) = 1;
                unsigned h0, h1, l0, l1;
                split2_f16(f32x2{v[0], v[1]}, h0, l0);
                split2_f16(f32x2{v[2], v[3]}, h1, l1);
                uint8_t* o = a.out_pl + (((size_t)(oc >> 5) * a.out_np + m) * 32 + (oc & 31)) * 2;
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<u32x2*>(o) = u32x2{h0, h1};
                if (a.out_fmt == 1) *reinterpret_cast<u32x2*>(o + a.out_pstride) = u32x2{l0, l1};
            }
        }
    }
    }   // tiles of the sequence
#endif
}

// One launch covers every group (job) of a layer, whatever its channel-tile count: the jobs are numbered widest first (their tiles take
// longest), the body is selected per run of tiles.  (Register budget 256: six waves per workgroup put two on two of the SIMDs.  With 512
// the compiler splits the file into VGPRs and AGPRs and moves accumulators between them at every stage.)
template <int KW, int NPL>
__global__ __launch_bounds__(KX_THREADS) void conv_kxr_kernel(const KxrArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __align__(16) uint8_t smem[];
    // persistent workgroups: workgroup w takes the global tiles w, w + grid, w + 2 grid, ...; the tiles are numbered job by job (widest
    // job first), so its list is a few runs of tiles of one job each -- one call of that job's body per run
    const int grid = gridDim.x;
    int g = blockIdx.x;
    while (g < a.total) {
        int j = 0;
#pragma unroll
        for (int i = 1; i < KX_MAX_JOBS; ++i)
            if (i < a.n_jobs && g >= a.job[i].tile0) j = i;
        const KxrJob& jb = a.job[j];
        const int jend = j + 1 < a.n_jobs ? a.job[j + 1].tile0 : a.total;     // first global tile behind this job
        const int n_seq = (jend - g + grid - 1) / grid;
        const int tile = g - jb.tile0;
        switch (jb.nc) {
            case 1: kxr_body<KW, NPL, 1>(a, jb, tile, n_seq, grid, smem); break;
            case 2: kxr_body<KW, NPL, 2>(a, jb, tile, n_seq, grid, smem); break;
            case 3: kxr_body<KW, NPL, 3>(a, jb, tile, n_seq, grid, smem); break;
            default: kxr_body<KW, NPL, 4>(a, jb, tile, n_seq, grid, smem); break;
        }
        g += n_seq * grid;
        __syncthreads();        // the next job's ring starts on LDS this one has left
    }
#endif
}

// weights [Cout][C][kh][kw] fp32 (grouped: group g = rows [g * cout_g, ...)) -> the blob of one job:
// [cs][ky][kx][plane][row 0 .. 16 nc)[chunk ^ swz(row)][8 fp16]; rows past the group's real channels are zero.
__global__ __launch_bounds__(256) void kxr_pack_kernel(const float* __restrict__ w, uint8_t* __restrict__ wp, int row0, int creal, int nc, int C,
                                                       int kh, int kw, int npl, float wscale)
{
    const int rows = nc * 16;
    const int64_t total = (int64_t)(C / 32) * kh * kw * rows * 4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int chunk = (int)(idx & 3), row = (int)((idx >> 2) % rows);
    const int tap = (int)((idx / (rows * 4)) % (kh * kw)), cs = (int)(idx / ((int64_t)rows * 4 * kh * kw));
    const int ky = tap / kw, kx = tap - ky * kw;
    unsigned pl[2][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f32x2 v = {0.0f, 0.0f};
        if (row < creal) {
            const size_t b = ((size_t)(row0 + row) * C + cs * 32 + chunk * 8 + 2 * e) * (kh * kw) + tap;
            v.x = w[b];
            v.y = w[b + (size_t)kh * kw];
        }
        split2_f16(v * wscale, pl[0][e], pl[1][e]);
    }
    const size_t stage = (size_t)cs * kh + ky;
    const size_t wt = (size_t)rows * 64;
    uint8_t* dst = wp + ((stage * kw + kx) * npl) * wt + lds_off(row, chunk);
    for (int p = 0; p < npl; ++p) *reinterpret_cast<u32x4*>(dst + p * wt) = u32x4{pl[p][0], pl[p][1], pl[p][2], pl[p][3]};
}

struct KxrPlan {
    int n_jobs, nc[KX_MAX_JOBS], creal[KX_MAX_JOBS], cout_g, npl, slabs_kh;   // slabs_kh = (C / 32) * kh stages
    size_t woff[KX_MAX_JOBS + 1];
};

// can the kernel take kw / format / nc at all (a three-stage ring must fit)
bool kxr_ring_fits(int kw, int npl, int nc) { return kx_depth(kw, npl, nc) >= 3; }

// Which layers the kernel takes, and how its weights are laid out.  Returns false with the error string set otherwise.
bool kxr_plan(const stm_conv_geom* g, KxrPlan* pl, const char* who)
{
    if (!g) { stm_set_error("%s: geometry is NULL", who); return false; }
    const int groups = g->groups > 0 ? g->groups : 1;
    if (g->C <= 0 || g->C % 32 || g->Cout <= 0 || g->Cout % groups || groups > KX_MAX_JOBS) {
        stm_set_error("%s: C (%d) must be a multiple of 32, Cout (%d) a multiple of groups (%d <= %d)", who, g->C, g->Cout, groups, KX_MAX_JOBS);
        return false;
    }
    if (!(g->kw == 3 || g->kw == 5) || g->kh < 1 || g->kh > 8 || g->sh != 1 || g->sw != 1 || g->ph < 0 || g->pw < 0 || g->pw >= g->kw || g->ph >= g->kh) {
        stm_set_error("%s: stride 1, kw = 3 or 5, kh <= 8, padding smaller than the kernel", who);
        return false;
    }
    if (g->fmt != 1 && g->fmt != 2) { stm_set_error("%s: fp16 plane formats (1, 2) only", who); return false; }
    pl->n_jobs = groups;
    pl->cout_g = g->Cout / groups;
    pl->npl = g->fmt == 1 ? 2 : 1;
    pl->slabs_kh = (g->C / 32) * g->kh;
    pl->woff[0] = 0;
    for (int i = 0; i < groups; ++i) {
        const int real = (g->group_cout[i] > 0 && g->group_cout[i] < pl->cout_g) ? g->group_cout[i] : pl->cout_g;
        if (real > 64) { stm_set_error("%s: at most 64 output channels per group (group %d has %d)", who, i, real); return false; }
        pl->creal[i] = real;
        pl->nc[i] = (real + 15) / 16;
        if (!kxr_ring_fits(g->kw, pl->npl, pl->nc[i])) {
            stm_set_error("%s: %d output channels with kw = %d: no three-stage ring fits the LDS", who, real, g->kw);
            return false;
        }
        pl->woff[i + 1] = pl->woff[i] + (size_t)pl->slabs_kh * g->kw * pl->npl * pl->nc[i] * 1024;
    }
    return true;
}

template <int KW, int NPL>
int kxr_launch(KxrArgs a, stm_stream_t stream)
{
    std::stable_sort(a.job, a.job + a.n_jobs, [](const KxrJob& x, const KxrJob& y) { return x.nc > y.nc; });    // widest first
    size_t lds = 0;
    a.total = 0;
    for (int i = 0; i < a.n_jobs; ++i) {
        KxrJob& j = a.job[i];
        const int nc = j.nc;
        const int pt = kx_pt(KW, NPL, nc), d = kx_depth(KW, NPL, nc);
        STM_REQUIRE(d >= 3, STM_EUNSUPPORTED, "stm_conv2d_planar_kxr_f32: %d channel tiles x kw = %d: no three-stage ring fits the LDS", nc, KW);
        lds = std::max(lds, (size_t)d * kx_stage_bytes(KW, NPL, nc, pt));
        j.tile0 = a.total;
        int t0 = 0;
        for (int l = 0; l < a.n_lvl; ++l) { j.lvl_tile0[l] = t0; t0 += stm_cdiv(a.lvl_start[l + 1] - a.lvl_start[l], 64 * pt); }
        j.lvl_tile0[a.n_lvl] = t0;
        j.per_xcd = (t0 + 7) / 8;
        a.total += 8 * j.per_xcd;
    }
    static std::atomic<int> reserved[KX_MAX_DEVICES];      // bytes reserved so far, per instantiation and device
    int dev = 0;
    const bool have_dev = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < KX_MAX_DEVICES;
    if (!have_dev || reserved[dev].load(std::memory_order_relaxed) < (int)lds) {
        STM_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kxr_kernel<KW, NPL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds) == hipSuccess, STM_ELAUNCH, "stm_conv2d_planar_kxr_f32: cannot reserve %zu bytes of LDS", lds);
        if (have_dev) reserved[dev].store((int)lds, std::memory_order_relaxed);
    }
    // one persistent workgroup per CU (the ring takes most of a CU's LDS)
    static std::atomic<int> n_cus[KX_MAX_DEVICES];
    int cus = have_dev ? n_cus[dev].load(std::memory_order_relaxed) : 0;
    if (cus <= 0) {
        hipDeviceProp_t prop;
        cus = (have_dev && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        if (have_dev) n_cus[dev].store(cus, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((conv_kxr_kernel<KW, NPL>), dim3(std::min(a.total, std::max(cus / 8 * 8, 8))), dim3(KX_THREADS), lds, stm_hs(stream), a);
    STM_CHECK_LAUNCH("conv_kxr_kernel");
    return STM_OK;
}

}  // namespace

extern "C" size_t stm_conv_kxr_packed_bytes(const stm_conv_geom* g)
{
    KxrPlan pl;
    return kxr_plan(g, &pl, "stm_conv_kxr_packed_bytes") ? pl.woff[pl.n_jobs] : 0;
}

extern "C" int stm_conv_pack_weights_kxr_f32(const float* weight, void* packed, const stm_conv_geom* g, float wscale, stm_stream_t stream)
{
    const char* who = "stm_conv_pack_weights_kxr_f32";
    STM_REQUIRE(weight && packed, STM_ENULL, "%s: weight/packed must be non-NULL", who);
    KxrPlan pl;
    if (!kxr_plan(g, &pl, who)) return STM_EINVAL;
    STM_REQUIRE((uintptr_t)packed % 16 == 0, STM_EINVAL, "%s: packed buffer must be 16-byte aligned", who);
    STM_REQUIRE(wscale > 0.0f && wscale < 3.0e38f, STM_EINVAL, "%s: bad weight scale", who);
    for (int i = 0; i < pl.n_jobs; ++i) {
        const int64_t total = (int64_t)(g->C / 32) * g->kh * g->kw * pl.nc[i] * 16 * 4;
        hipLaunchKernelGGL(kxr_pack_kernel, dim3(stm_cdiv(total, 256)), dim3(256), 0, stm_hs(stream), weight, static_cast<uint8_t*>(packed) + pl.woff[i],
                           i * pl.cout_g, pl.creal[i], pl.nc[i], g->C, g->kh, g->kw, pl.npl, wscale);
        STM_CHECK_LAUNCH("kxr_pack_kernel");
    }
    return STM_OK;
}

extern "C" int stm_conv2d_planar_kxr_f32(const void* x_planes, const void* packed_weight, const float* bias, float* out_f32, void* out_planes,
                                         const stm_conv_geom* g, int relu, stm_stream_t stream)
{
    const char* who = "stm_conv2d_planar_kxr_f32";
    STM_REQUIRE(x_planes && packed_weight && (out_f32 || out_planes), STM_ENULL, "%s: x_planes/packed_weight and at least one output must be non-NULL", who);
    KxrPlan pl;
    if (!kxr_plan(g, &pl, who)) return STM_EINVAL;
    KxrArgs a;
    memset(&a, 0, sizeof(a));
    int64_t M;
    if (g->n_levels > 0) {
        STM_REQUIRE(g->n_levels <= 8 && g->lvl_start[0] == 0, STM_EINVAL, "%s: at most 8 levels, lvl_start[0] = 0", who);
        a.n_lvl = g->n_levels;
        for (int l = 0; l < g->n_levels; ++l) {
            const int n = g->lvl_start[l + 1] - g->lvl_start[l];
            STM_REQUIRE(g->lvl_h[l] > 0 && g->lvl_w[l] > 0 && n > 0 && n % (g->lvl_h[l] * g->lvl_w[l]) == 0, STM_EINVAL,
                        "%s: level %d: %d pixels is not a whole number of %dx%d images", who, l, n, g->lvl_h[l], g->lvl_w[l]);
            a.lvl_start[l] = g->lvl_start[l]; a.lvl_h[l] = g->lvl_h[l]; a.lvl_w[l] = g->lvl_w[l];
        }
        a.lvl_start[g->n_levels] = g->lvl_start[g->n_levels];
        M = g->lvl_start[g->n_levels];
    } else {
        STM_REQUIRE(g->B > 0 && g->H > 0 && g->W > 0, STM_EINVAL, "%s: bad image batch", who);
        STM_REQUIRE(g->Ho == g->H + 2 * g->ph - g->kh + 1 && g->Wo == g->W + 2 * g->pw - g->kw + 1 && g->Ho == g->H && g->Wo == g->W, STM_EUNSUPPORTED,
                    "%s: the kernel keeps the image size (padding (k - 1) / 2)", who);
        a.n_lvl = 1;
        M = (int64_t)g->B * g->H * g->W;
        a.lvl_start[0] = 0; a.lvl_start[1] = (int)M; a.lvl_h[0] = g->H; a.lvl_w[0] = g->W;
    }
    STM_REQUIRE(2 * g->ph == g->kh - 1 && 2 * g->pw == g->kw - 1, STM_EUNSUPPORTED, "%s: same padding only", who);
    STM_REQUIRE(M < ((int64_t)1 << 30), STM_EUNSUPPORTED, "%s: too many pixels", who);
    const int groups = pl.n_jobs;
    const int64_t x_np = g->x_np ? g->x_np : M, out_np = g->out_np ? g->out_np : M;
    STM_REQUIRE(x_np >= M && out_np >= M, STM_EINVAL, "%s: x_np / out_np smaller than the pixel count", who);
    const int64_t x_slabs = (int64_t)groups * (g->C / 32);
    const int64_t plane_bytes = x_slabs * x_np * 64;
    STM_REQUIRE(plane_bytes < ((int64_t)1 << 31), STM_EUNSUPPORTED, "%s: plane larger than 2 GiB", who);
    const int64_t xps = g->x_plane_stride ? g->x_plane_stride : x_slabs * x_np * 32;
    const int64_t ops = g->out_plane_stride ? g->out_plane_stride : (int64_t)stm_cdiv(g->Cout, 32) * out_np * 32;
    const int out_ld = g->out_ld ? g->out_ld : g->Cout;
    STM_REQUIRE((uintptr_t)x_planes % 16 == 0 && (uintptr_t)packed_weight % 16 == 0 && xps % 8 == 0, STM_EINVAL, "%s: 16-byte alignment required", who);
    STM_REQUIRE(!out_f32 || ((uintptr_t)out_f32 % 16 == 0 && out_ld % 4 == 0), STM_EINVAL, "%s: out_f32 must be 16-byte aligned with out_ld a multiple of 4", who);
    STM_REQUIRE(!out_planes || ((uintptr_t)out_planes % 16 == 0 && ops % 8 == 0), STM_EINVAL, "%s: out_planes must be 16-byte aligned", who);
    a.xp = static_cast<const uint8_t*>(x_planes); a.out_f32 = out_f32; a.out_pl = static_cast<uint8_t*>(out_planes);
    a.n_jobs = groups;
    int nc_max = 1;
    for (int i = 0; i < groups; ++i) {
        KxrJob& j = a.job[i];
        j.wp = static_cast<const uint8_t*>(packed_weight) + pl.woff[i];
        j.bias = bias ? bias + (size_t)i * pl.cout_g : nullptr;
        j.bias_n = pl.cout_g;
        j.x_slab0 = i * (g->C / 32);
        j.out_ch0 = i * pl.cout_g;
        j.nc = pl.nc[i];
        nc_max = std::max(nc_max, j.nc);
        // every 16-channel tile is written whole: the output row must hold it
        STM_REQUIRE(j.out_ch0 + 16 * j.nc <= (out_f32 ? out_ld : g->Cout) || !out_f32, STM_EINVAL,
                    "%s: group %d writes channels [%d, %d) but an output row has %d", who, i, j.out_ch0, j.out_ch0 + 16 * j.nc, out_ld);
        STM_REQUIRE(!out_planes || j.out_ch0 + 16 * j.nc <= stm_cdiv(g->Cout, 32) * 32, STM_EINVAL, "%s: group %d leaves the output planes", who, i);
    }
    a.cslabs = g->C / 32; a.kh = g->kh; a.ph = g->ph; a.pw = g->pw;
    a.x_np = (int)x_np; a.out_ld = out_ld; a.out_np = (int)out_np;
    a.x_pstride = xps * 2; a.out_pstride = ops * 2;
    a.plane_bytes = (unsigned)plane_bytes;
    a.relu = relu;
    a.out_fmt = g->out_fmt_plus1 > 0 ? g->out_fmt_plus1 - 1 : g->fmt;
    STM_REQUIRE(a.out_fmt == g->fmt || (g->fmt == 2 && a.out_fmt == 1), STM_EINVAL, "%s: output format %d cannot be produced by a format-%d layer", who,
                a.out_fmt, g->fmt);
    a.out_scale = g->out_scale > 0.0f ? g->out_scale : 1.0f;
    a.range_flag = stm_internal_range_flag();
    (void)nc_max;
    if (g->fmt == 1) return g->kw == 3 ? kxr_launch<3, 2>(a, stream) : kxr_launch<5, 2>(a, stream);
    return g->kw == 3 ? kxr_launch<3, 1>(a, stream) : kxr_launch<5, 1>(a, stream);
}
