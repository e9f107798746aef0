// conv_chain.hip -- the tail of one ResNet bottleneck and the head of the next as ONE kernel (reference backbone.py:38-58, Bottleneck.forward):
//
//     mid2 = relu(conv2(mid1) + b2)          3x3, P -> P, stride 1        (BatchNorm folded)
//     y    = relu(conv3(mid2) + b3 + x)      1x1, P -> 4P, + identity shortcut
//     z    = relu(conv1'(y) + b1')           1x1, 4P -> P, the NEXT block's first convolution (optional)
//
// for the 64-channel bottlenecks of layer1 (P = 64).  As three launches these layers move 1.9 GB per block at batch 32 (mid2 written and
// read, y written, read by conv1' and read again as the next shortcut) and run at the HBM rate: 519 us.  Chained, the 3x3's input (126 MB),
// the shortcut (503 MB) and the two outputs (503 + 126 MB) cross HBM once.  No halo is recomputed: the only tensor that crosses a kernel
// boundary between two 3x3 convolutions is the 64-channel z.
//
// Built on conv_kxr.hip's machinery (flat 128-pixel tiles, kx-reuse staging of the 3x3's input rows, two producer waves issuing every
// LDS-DMA, four consumer waves of 32 pixels, ring of three stage buffers, persistent workgroups) plus what the TRANSPOSED product buys:
// a consumer wave owns its 32 pixels for all channels, and the accumulator layout of D[channel][pixel] = W X^T -- a lane holds 4
// consecutive channels of one pixel -- is, after bias / ReLU / plane split, exactly a B-operand fragment of the next product if that
// product's K order is permuted accordingly (slab k' = 8 kc + j  <->  channel 4 kc + j of the slab's first 16-channel tile for j < 4, of
// its second tile for j >= 4).  The permutation is baked into the packed weights of conv3 and conv1', so mid2 and y feed the next
// product straight from registers: no LDS round trip, no barrier.  conv3 runs in eight half-groups of 32 output channels -- one K-slab
// of conv1' each -- whose weights (conv3's two tiles, conv1's slab) arrive through the same ring as eight more stages of 16 KB.
//
// Arithmetic = the three planar convolutions: per K-slab acc += w_h x_h, accl += w_h x_l then w_l x_h, fp32 accumulation,
// (acc + accl / 2048) * out_scale + bias, residual added as h + l / 2048, ReLU, results split into fp16 planes exactly as the separate
// kernels' epilogues do.  Only the order of the 32 products INSIDE an MFMA differs (the K permutation): fp32-rounding-level differences.
#include "planar_common.h"

#include <algorithm>
#include <atomic>

namespace {

constexpr int CH_P = 64;                  // bottleneck width
constexpr int CH_PT = 2;                  // pixel tiles of 16 per consumer wave
constexpr int CH_BM = 64 * CH_PT;         // 128 flat pixels per workgroup tile
constexpr int CH_XROWS = CH_BM + 16, CH_NRG = CH_XROWS / 16;
constexpr int CH_XPL = CH_XROWS * 64, CH_XBUF = 2 * CH_XPL;                 // staged rows: two planes
constexpr int CH_WT = 4 * 16 * 64, CH_WBUF = 3 * 2 * CH_WT;                 // conv2 weights of one (channel slab, ky): 3 taps x 2 planes x 64 rows
constexpr int CH_BUF = CH_XBUF + CH_WBUF;                                   // 43 008 B per ring slot
constexpr int CH_D = 3;                                                     // ring slots (the producers' wait is written for three)
// tail stage j = 0 .. 8: [conv3 weights of half-group j: 8 KB | conv1' weights of K-slab j - 1: 8 KB]
constexpr int CH_TAIL_STAGES = 9, CH_TW3 = 0, CH_TW1 = 8192, CH_TW_BYTES = 16384;
// projection-shortcut form (a stage's first block: y = relu(conv3(mid2) + proj(x0) + b)): conv3 runs over four K-slabs -- mid2's two and the two of
// the 64-channel block input x0 -- so a tail stage carries 16 KB of conv3 tiles [q 2][s 4][plane 2] and conv1's 8 KB behind them
constexpr int CH_TW1_P = 16384, CH_TWP_BYTES = 24576;
constexpr int CH_RSLOTS = 2;              // half-groups of the shortcut tensor in flight per consumer wave (registers)
// behind the ring: the consumers' mid2 fragments (8 KB per wave: 32 registers the tail cannot spare) and the three bias vectors
constexpr int CH_PARK_OFF = CH_D * CH_BUF, CH_BIAS_OFF = CH_PARK_OFF + 4 * 8192, CH_LDS = CH_BIAS_OFF + (64 + 256 + 64) * 4;
static_assert(CH_LDS <= 160 * 1024, "LDS");
constexpr int CH_CONSUMERS = 4, CH_PRODUCERS = 2, CH_THREADS = 64 * (CH_CONSUMERS + CH_PRODUCERS);
constexpr int CH_MAX_DEVICES = 32;

struct ChainArgs {
    const uint8_t* xin;      // mid1 planes [2][2][np_in][32]
    const uint8_t* res;      // shortcut planes [2][8][np_res][32]
    uint8_t* y;              // [2][8][np_y][32]
    uint8_t* z;              // [2][2][np_z][32] or null (no chained conv1')
    const uint8_t* w2;       // conv2, conv_kxr layout: [stage (slab, ky)][kx][plane][64 rows][64 B]
    const uint8_t* wt;       // tail: [stage 9][conv3 of half-group j: tile 2, slab 2, plane 2 x 1 KB | conv1' of K-slab j - 1: tile 4, plane 2 x 1 KB]
    const float *b2, *b3, *b1;
    float scale2, scale3, scale1;
    int B, H, W, M;
    int np_in, np_res, np_y, np_z;
    long long ps_in, ps_res, ps_y, ps_z;      // bytes between planes
    unsigned plane_bytes_in;
    int tiles;
    int* range_flag;
    unsigned long long* dbg;
};

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#ifndef CH_NT
#define CH_NT 0           // cache policy of the shortcut loads and the y / z stores: 2 = streaming (nt), 0 = default
#endif
#ifndef CH_TIMING
#define CH_TIMING 0       // 1: consumer wave 0 of workgroup 0 records s_memtime around every barrier into the buffer given to stm_debug_chain_timing
#endif
#ifndef CH_PROBE
#define CH_PROBE 0        // 1: every wave records {HW_ID | XCC_ID at start and end, largest s_memrealtime gap between two of its barriers, the stage of that gap}
                          //    into the buffer given to stm_debug_chain_timing: [block][wave][4] u64 -- shows a wave that was context-switched (CWSR) mid-kernel
#endif
#ifndef CH_STORE_NOPS
#define CH_STORE_NOPS 4   // wait states pinned behind every 16-byte y / z store (0: none = round 3's code; see store16 below)
#endif
#ifndef CH_STORE_VOFF
#define CH_STORE_VOFF 0   // 1: the stores' slab offset is added to the VGPR offset and soffset is 0 -- the form llvm's hazard recogniser pads by itself
#endif
#ifndef CH_COUNTED
#define CH_COUNTED 1      // the producers' landing wait as a counted vmcnt with the next stage in flight; 0 = vmcnt(0) before the next stage goes out
#endif
#ifndef CH_LGKM
#define CH_LGKM 1         // consumers drain their LDS reads (s_waitcnt lgkmcnt(0)) before every barrier; 0 = round 3's barrier (diagnosis: a ring slot is
                          // then restaged under ds_reads that are still queued)
#endif
#ifndef CH_ZX
#define CH_ZX 0        // diagnosis of the z-producing instantiation: 1 no z stores, 2 no conv1' MFMAs, 4 no schedule hints in the tail
#endif
#ifndef CH_ABL
#define CH_ABL 0      // diagnostic builds (make EXTRA=-DCH_ABL=n, RESULTS ARE WRONG): 1 no stores, 2 no shortcut loads, 4 no tail MFMAs, 8 no tail epilogue, 16 no 3x3 MFMAs, 32 no DMA
#endif
#define CH_MM(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x32_f16(a_, b_, c_, 0, 0, 0)

__device__ __forceinline__ f16x8 pack_frag(u32x2 lo, u32x2 hi)
{
    const u32x4 q = {lo.x, lo.y, hi.x, hi.y};
    return __builtin_bit_cast(f16x8, q);
}

// The accumulator layout gives lane (pixel r16, row kc) channels 4 kc .. 4 kc + 3 of each 16-channel tile: 8 bytes, and a store
// instruction writes 32-byte pieces of sixteen 64-byte slab rows.  Two gfx950 row swaps turn the lane's values of the two tiles of a
// 32-channel slab (x0: tile 0, x1: tile 1) into channels 8 kc .. 8 kc + 7 of the slab: 16 bytes per lane, one KB contiguous per
// instruction -- a quarter of the L2 write requests.  v_permlane32_swap exchanges rows 2, 3 of its first operand with rows 0, 1 of
// the second, v_permlane16_swap rows 1, 3 of the first with rows 0, 2 of the second; both are their own inverse.
__device__ __forceinline__ u32x4 slab_gather(u32x2 x0, u32x2 x1)
{
    const u32x2 a = __builtin_amdgcn_permlane32_swap(x0.x, x1.x, false, false), b = __builtin_amdgcn_permlane32_swap(x0.y, x1.y, false, false);
    const u32x2 a2 = __builtin_amdgcn_permlane16_swap(a.x, a.y, false, false), b2 = __builtin_amdgcn_permlane16_swap(b.x, b.y, false, false);
    return u32x4{a2.x, b2.x, a2.y, b2.y};
}
__device__ __forceinline__ void slab_scatter(u32x4 v, u32x2& x0, u32x2& x1)
{
    const u32x2 a = __builtin_amdgcn_permlane16_swap(v.x, v.z, false, false), b = __builtin_amdgcn_permlane16_swap(v.y, v.w, false, false);
    const u32x2 a2 = __builtin_amdgcn_permlane32_swap(a.x, a.y, false, false), b2 = __builtin_amdgcn_permlane32_swap(b.x, b.y, false, false);
    x0 = u32x2{a2.x, b2.x};
    x1 = u32x2{a2.y, b2.y};
}

// four fp32 -> the two fp16 planes of four consecutive channels (8 bytes per plane)
__device__ __forceinline__ void split4_f16(const float (&v)[4], u32x2& h, u32x2& l)
{
    unsigned h0, h1, l0, l1;
    split2_f16(f32x2{v[0], v[1]}, h0, l0);
    split2_f16(f32x2{v[2], v[3]}, h1, l1);
    h = u32x2{h0, h1};
    l = u32x2{l0, l1};
}

// A 16-byte buffer store reads its data registers over several cycles AFTER it has issued.  The ISA's wait-state table asks for wait states before
// a VALU instruction overwrites them and exempts MUBUF stores with an SGPR soffset; llvm's hazard recogniser follows it (GCNHazardRecognizer::
// createsVALUHazard), so `buffer_store_dwordx4 v[24:27], v160, s[48:51], s14 offen` / `v_mov_b32 v24, v16` came out back to back in the
// z-producing instantiations, whose register pressure makes consecutive stores share their data registers.  On MI355X the exemption does not
// hold when the texture path is back-pressured: lanes 12-15 / 28-31 / 44-47 / 60-63 of such a store -- the last 256 B of the wave's 1 KB --
// pick up the NEW contents (scripts/vmem_store_war_probe.hip, profiles/r04_store_war_probe.txt; in the kernel: wrong y / z in exactly those
// pixels, only beside a second process, only in the instantiations with the sequence: scripts/ring_stress.py, scripts/lint_store_war.py).
// Every store therefore goes through store16: the wait states are pinned behind it (nothing crosses a sched_barrier).
__device__ __forceinline__ void store16(u32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
{
#if CH_STORE_VOFF
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff + (unsigned)soff, 0, CH_NT);
#else
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, CH_NT);
#endif
#if CH_STORE_NOPS > 0
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop %0" ::"n"(CH_STORE_NOPS - 1));
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// Workgroup = 4 consumer waves (32 pixels each, every channel) + 2 producer waves issuing all LDS-DMA: activations and weights of the
// 3x3's six (channel slab, ky) stages, then per tile nine tail stages with conv3's and conv1's weights.  Ring of three slots, one
// barrier per stage (conv_kxr.hip).  The SHORTCUT tensor -- 80 % of the bytes read -- does not go through the ring: two stages of
// look-ahead are ~1.5 us, less than a loaded HBM round trip, and a ring stalled on it ran the kernel at 2.5 TB/s.  Each consumer lane
// loads the shortcut values of its own accumulator elements (8 bytes per plane, tile and pixel) straight into registers, four
// half-groups (64 registers) ahead of their use: the slot a stage's epilogue frees is refilled at once with the half-group four stages
// on, across tile boundaries -- the next tile's first four half-groups arrive under this tile's last stages and the next 3x3.
// The consumers' tail is software-pipelined by one stage: stage j issues conv3 of half-group j, then conv1' of K-slab j - 1 (whose B
// fragments the previous stage's epilogue left in registers), then the epilogue of half-group j -- on the VALU while conv1's MFMAs drain.
template <bool HAS_Z, bool PROJ>
__global__ __launch_bounds__(CH_THREADS) void conv_chain_kernel(const ChainArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __align__(16) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grid = gridDim.x;
    // Persistent workgroups over the tile ids blockIdx.x, + grid, ...; ids are dealt round-robin to the 8 XCDs, so id -> tile
    // (id & 7) * per_xcd + (id >> 3) gives each XCD (= each L2) a contiguous run of pixels: the rows a tile's three ky stages read are the
    // rows of its neighbours, one L2 fill instead of three (measured: 943 MB fetched for 629 MB of input before).  Ids past the padded
    // count map to tiles behind M: every access of such a tile is out of range.
    const int per_xcd = (a.tiles + 7) >> 3;
    const int n_seq = (8 * per_xcd - (int)blockIdx.x + grid - 1) / grid;
    if (n_seq <= 0) return;
    auto tile_of = [&](int id) { return (id & 7) * per_xcd + (id >> 3); };
    constexpr int S = (CH_P / 32) * 3;                                        // main stages per tile: (channel slab, ky)
    constexpr int SG = S + CH_TAIL_STAGES;                                    // stages per tile
    const int HW = a.H * a.W;
#if CH_PROBE
    const unsigned pr_id0 = __builtin_amdgcn_s_getreg((31 << 11) | 4) ^ (__builtin_amdgcn_s_getreg((31 << 11) | 20) << 28);     // HW_ID, XCC_ID
    unsigned long long pr_prev = __builtin_amdgcn_s_memrealtime(), pr_max = 0;
    int pr_stage = 0, pr_at = -1;
    auto probe = [&]() {
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        if (t - pr_prev > pr_max) { pr_max = t - pr_prev; pr_at = pr_stage; }
        pr_prev = t;
        ++pr_stage;
    };
    auto probe_end = [&]() {
        const unsigned id1 = __builtin_amdgcn_s_getreg((31 << 11) | 4) ^ (__builtin_amdgcn_s_getreg((31 << 11) | 20) << 28);
        if (a.dbg && lane == 0) {
            unsigned long long* d = a.dbg + ((size_t)blockIdx.x * (CH_CONSUMERS + CH_PRODUCERS) + wave) * 4;
            d[0] = (unsigned long long)pr_id0 | ((unsigned long long)id1 << 32);
            d[1] = pr_max; d[2] = (unsigned long long)(unsigned)pr_at; d[3] = (unsigned long long)pr_stage;
        }
    };
#else
    auto probe = [&]() {};
    auto probe_end = [&]() {};
#endif
    float* bias_lds = reinterpret_cast<float*>(smem + CH_BIAS_OFF);           // b2 [64] | b3 [256] | b1' [64]
    for (int i = tid; i < 384; i += CH_THREADS)
        bias_lds[i] = i < 64 ? (a.b2 ? a.b2[i] : 0.0f) : i < 320 ? (a.b3 ? a.b3[i - 64] : 0.0f) : ((HAS_Z && a.b1) ? a.b1[i - 320] : 0.0f);
    __syncthreads();

    if (wave >= CH_CONSUMERS) {
        // ------------------------------------------------------------------------------------------------------------ producer
        constexpr int NXD = CH_NRG * 2, XDW = (NXD + CH_PRODUCERS - 1) / CH_PRODUCERS;       // 18 activation pieces: 9 per producer
        constexpr int NWD = 3 * 2 * 4, WDW = NWD / CH_PRODUCERS;                              // 24 weight pieces: 12 per producer
        constexpr int TWB = PROJ ? CH_TWP_BYTES : CH_TW_BYTES, TW1 = PROJ ? CH_TW1_P : CH_TW1;
        constexpr int TWD = TWB / 1024 / CH_PRODUCERS;                                        // 16 (24) tail weight pieces: 8 (12) per producer
        const int pw_ = wave - CH_CONSUMERS;
        // activations: producer p stages plane p (nine row groups of 16 staged rows); the weight pieces are dealt round-robin
        static_assert(CH_PRODUCERS == 2 && XDW == CH_NRG, "one activation plane per producer");
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.xin) + (size_t)pw_ * a.ps_in, 0, (int)a.plane_bytes_in, 0x00020000);
        // (the weights go through buffer descriptors too: one kind of instruction for every LDS-DMA of a wave)
        const __amdgpu_buffer_rsrc_t w2r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.w2), 0, S * CH_WBUF, 0x00020000);
        const __amdgpu_buffer_rsrc_t wtr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.wt), 0, CH_TAIL_STAGES * TWB, 0x00020000);
        int dbase[XDW];
        unsigned dmask[XDW];
        auto setup = [&](int tile) {
            const int m0 = tile * CH_BM;
#pragma unroll
            for (int i = 0; i < XDW; ++i) {
                const int rg = i;
                const int j = rg * 16 + (lane >> 2);
                const int q = m0 - 1 + j;
                const bool okq = j < CH_BM + 2 && q >= 0 && q < a.M;
                const int local = okq ? q : 0;
                const int y = (local % HW) / a.W;
                unsigned vm = 0;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
                    if ((unsigned)(y + ky - 1) < (unsigned)a.H) vm |= 1u << ky;
                dmask[i] = okq ? vm : 0u;
                dbase[i] = (q - a.W) * 64 + (((lane & 3) ^ swz(j)) << 4);
            }
        };
        int buf = 0, s_in = 0, t_in = 0;
        // issues the next stage; returns its kind: 0 main (21 DMAs per producer), 1 tail (8), 2 last tail (4)
        auto issue = [&]() -> int {
            uint8_t* sb = smem + buf * CH_BUF;
            if (++buf == CH_D) buf = 0;
            int kind = 0;
            if (s_in < S) {
                if (s_in == 0) setup(tile_of((int)blockIdx.x + t_in * grid));
                const int cs = s_in / 3, ky = s_in - 3 * cs;
                const int uni = cs * (a.np_in * 64) + ky * (a.W * 64);
#pragma unroll
                for (int i = 0; i < XDW; ++i) {
                    const unsigned oob = ((dmask[i] >> ky) & 1u) ^ 1u;
                    const unsigned off = (unsigned)(dbase[i] + uni) | (oob << 31);
                    if (!(CH_ABL & 32)) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr)(sb + pw_ * CH_XPL + i * 1024), 16, off, 0, 0, 0);
                }
                const int wsrc = s_in * CH_WBUF;        // byte offset in the packed conv2 weights
#pragma unroll
                for (int k = 0; k < WDW; ++k) {
                    const int idx = pw_ + CH_PRODUCERS * k;
                    if (!(CH_ABL & 32)) __builtin_amdgcn_raw_ptr_buffer_load_lds(w2r, (lds_ptr)(sb + CH_XBUF + idx * 1024), 16, lane * 16, wsrc + idx * 1024, 0, 0);
                }
            } else {
                const int j = s_in - S;
                const int wsrc = j * TWB;               // byte offset in the packed tail weights
                if (j < CH_TAIL_STAGES - 1) {
                    kind = 1;
#pragma unroll
                    for (int k = 0; k < TWD; ++k) {
                        const int idx = pw_ + CH_PRODUCERS * k;
                        if (!(CH_ABL & 32)) __builtin_amdgcn_raw_ptr_buffer_load_lds(wtr, (lds_ptr)(sb + idx * 1024), 16, lane * 16, wsrc + idx * 1024, 0, 0);
                    }
                } else {
                    kind = 2;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int idx = TW1 / 1024 + pw_ + CH_PRODUCERS * k;       // conv1' part only
                        if (!(CH_ABL & 32)) __builtin_amdgcn_raw_ptr_buffer_load_lds(wtr, (lds_ptr)(sb + idx * 1024), 16, lane * 16, wsrc + idx * 1024, 0, 0);
                    }
                }
            }
            if (++s_in == SG) { s_in = 0; ++t_in; }
            return kind;
        };
        const int G = SG * n_seq;
        // Stage g goes out after barrier g - 2 (its slot held stage g - 3, which every consumer has left: they drain lgkmcnt in front of their
        // barriers) and must have landed at barrier g.  The landing wait is a COUNTED vmcnt that leaves only the stage just issued outstanding:
        // a wave's loads, stores and LDS-DMAs complete in issue order (MI355X_MICROARCH.md; asked of the hardware by scripts/ldsdma_order_probe.hip:
        // 0 stale words in 3.1e9 per variant, stages of HBM rows followed by L2-resident pieces, out-of-range pieces and mixed instruction kinds
        // included -- profiles/r04_ldsdma_order_probe.txt).
        // History: round 3 ran this loop with UNDRAINED consumer barriers, saw one pipeline pass in ten differ, blamed the counted wait and
        // went to vmcnt(0)-before-issue (CH_COUNTED=0).  The wait was innocent: with the counted form the producers restage slot (g + 1) % 3
        // the moment barrier g - 1 opens, while the consumers' last fragment reads of stage g - 2 -- issued before that barrier, their MFMAs
        // sunk below it by the compiler -- can still be queued in the LDS pipe, and a weight piece from L2 lands within a few hundred cycles.
        // vmcnt(0)-before-issue only delays the restaging by the landing time of the stage in flight.  Measured beside a second process
        // (scripts/ring_stress.py, 1 200 launches per case): counted + undrained 34 wrong outputs in 19 200 launches (whole 16-pixel fragments,
        // every instantiation), counted + drained 0, vmcnt(0)-before-issue + drained 0 (profiles/r04_ring_stress_chain_variants.txt).
#if CH_COUNTED
        issue();
        for (int g = 1; g < G; ++g) {
            const int kind = issue();
            if (kind == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XDW + WDW) : "memory");
            else if (kind == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TWD) : "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            probe();
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        probe_end();
        return;
#endif
        // (CH_COUNTED=0, the cross-check form)
        issue();                                     // G >= 15
        for (int g = 1; g < G; ++g) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // stage g - 1 has landed
            issue();                                              // stage g flies across the barrier
            asm volatile("s_barrier" ::: "memory");               // barrier g - 1
            probe();
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        probe_end();
        return;
    }

    // ---------------------------------------------------------------------------------------------------------------- consumer
    const int r16 = lane & 15, kc = lane >> 4;
    const int aoff0 = lds_off(r16, kc);                    // a weight fragment's offset inside its 1-KB tile
    const float ls = 1.0f / STM_F16_LOW_SCALE;
    int boff[CH_PT][3];
    f32x4 acc[4][CH_PT], accl[4][CH_PT];
    // shortcut values of this lane's accumulator elements: slot = half-group mod 4, [channel tile q][pixel tile t], high / low plane
    u32x4 rsh[CH_RSLOTS][CH_PT], rsl[CH_RSLOTS][CH_PT];     // as loaded: channels 8 kc .. 8 kc + 7 of the slab (slab_scatter at use)
    // shortcut / y / z through buffer descriptors covering both planes: a pixel behind M gets bit 31 in its offset -- loads return 0, stores
    // are dropped, and the epilogues stay free of branches (one basic block per stage: the scheduler can put the VALU work under the MFMAs)
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.res), 0, (int)(a.ps_res + (long long)a.np_res * (PROJ ? 128 : 512)), 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)(a.ps_y + (long long)a.np_y * 512), 0x00020000);
    const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(HAS_Z ? a.z : a.y, 0, HAS_Z ? (int)(a.ps_z + (long long)a.np_z * 128) : 0, 0x00020000);
    auto px_off = [&](int m) { return (unsigned)(m * 64 + 16 * kc) | (m >= a.M ? 0x80000000u : 0u); };     // byte offset of (pixel m, channel 8 kc) in a slab
    auto load_res = [&](int slot, int hg, const unsigned (&po)[CH_PT]) {
#pragma unroll
        for (int t = 0; t < CH_PT; ++t) {
            if (CH_ABL & 2) { rsh[slot][t] = u32x4{0u, 0u, 0u, 0u}; rsl[slot][t] = u32x4{0u, 0u, 0u, 0u}; continue; }
            rsh[slot][t] = __builtin_amdgcn_raw_buffer_load_b128(rr, po[t], hg * a.np_res * 64, CH_NT);
            rsl[slot][t] = __builtin_amdgcn_raw_buffer_load_b128(rr, po[t], hg * a.np_res * 64 + (int)a.ps_res, CH_NT);
        }
    };
    if constexpr (!PROJ) {
        unsigned po[CH_PT];
#pragma unroll
        for (int t = 0; t < CH_PT; ++t) po[t] = px_off(tile_of((int)blockIdx.x) * CH_BM + 16 * CH_PT * wave + 16 * t + r16);
#pragma unroll
        for (int k = 0; k < CH_RSLOTS; ++k) load_res(k, k, po);
    }
    // projection form: the block input x0 of this lane's pixels, B fragments of conv3's K-slabs 2 and 3 in their natural order (chunk kc
    // = channels 8 kc ..: exactly the 16 bytes a lane loads); they take the registers of the shortcut slots, fetched at the tile's start
    u32x4 (&xfh)[CH_RSLOTS][CH_PT] = rsh, (&xfl)[CH_RSLOTS][CH_PT] = rsl;
    unsigned rng = 0;          // largest magnitude bits this lane produced (all values are post-ReLU: non-negative)
    int buf = 0;
    int dbg_i = 0;
    auto tick = [&]() {
        if (CH_TIMING && a.dbg && blockIdx.x == 0 && wave == 0) { const unsigned long long t = __builtin_readcyclecounter(); if (lane == 0) a.dbg[dbg_i] = t; ++dbg_i; }
    };
    for (int t_seq = 0; t_seq < n_seq; ++t_seq) {
        const int m0 = tile_of((int)blockIdx.x + t_seq * grid) * CH_BM;
        const int m0_next = tile_of((int)blockIdx.x + (t_seq + 1) * grid) * CH_BM;
        tick();
        int mpx[CH_PT];
        unsigned po[CH_PT], po_next[CH_PT];
#pragma unroll
        for (int t = 0; t < CH_PT; ++t) {
            const int row0 = 16 * CH_PT * wave + 16 * t + r16;
            const int m = m0 + row0;
            mpx[t] = m;
            po[t] = px_off(m);
            po_next[t] = px_off(m0_next + row0);            // (behind the last tile: out of range, nothing is fetched)
            if constexpr (PROJ) {
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    xfh[sl][t] = __builtin_amdgcn_raw_buffer_load_b128(rr, po[t], sl * a.np_res * 64, 0);
                    xfl[sl][t] = __builtin_amdgcn_raw_buffer_load_b128(rr, po[t], sl * a.np_res * 64 + (int)a.ps_res, 0);
                }
            }
            const bool okm = m < a.M;
            const int x = okm ? m % a.W : 0;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const bool v = okm && (unsigned)(x + kx - 1) < (unsigned)a.W;
                boff[t][kx] = lds_off(v ? row0 + kx : CH_BM + 2, kc);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < CH_PT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc[c][t][r] = 0.0f; accl[c][t][r] = 0.0f; }

        // ---- conv2: six stages of three taps
        for (int s = 0; s < S; ++s) {
            tick();
            if (CH_LGKM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            probe();
            tick();
            const uint8_t* xs = smem + buf * CH_BUF;
            if (++buf == CH_D) buf = 0;
            if constexpr ((CH_ABL & 16) != 0) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                f16x8 bh[CH_PT], bl[CH_PT];
#pragma unroll
                for (int t = 0; t < CH_PT; ++t) {
                    bh[t] = *reinterpret_cast<const f16x8*>(xs + boff[t][kx]);
                    bl[t] = *reinterpret_cast<const f16x8*>(xs + CH_XPL + boff[t][kx]);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f16x8 ah = *reinterpret_cast<const f16x8*>(xs + CH_XBUF + ((kx * 2) * 4 + c) * 1024 + aoff0);
                    const f16x8 al = *reinterpret_cast<const f16x8*>(xs + CH_XBUF + ((kx * 2 + 1) * 4 + c) * 1024 + aoff0);
#pragma unroll
                    for (int t = 0; t < CH_PT; ++t) {
                        accl[c][t] = CH_MM(ah, bl[t], accl[c][t]);
                        acc[c][t] = CH_MM(ah, bh[t], acc[c][t]);
                        accl[c][t] = CH_MM(al, bh[t], accl[c][t]);
                    }
                }
            }
        }
        // ---- mid2 = relu(conv2 + b2), split: B fragments of conv3's two K-slabs (slab s = channel tiles 2s, 2s + 1), parked in this
        // wave's own 8 KB of LDS (fragment (s, t, plane) at 1 KB each, lane-contiguous): every tail stage reads them back
        uint8_t* park = smem + CH_PARK_OFF + wave * 8192 + lane * 16;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < CH_PT; ++t) {
                u32x2 h[2], l[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int c = 2 * s + q;
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_lds + 16 * c + 4 * kc);
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = __builtin_fmaxf(__builtin_fmaf(acc[c][t][r] + accl[c][t][r] * ls, a.scale2, bv[r]), 0.0f);
                        rng = max(rng, __builtin_bit_cast(unsigned, v[r]));          // (mid2 never leaves the kernel, but an overflowed plane would zero y silently)
                    }
                    split4_f16(v, h[q], l[q]);
                }
                *reinterpret_cast<f16x8*>(park + ((s * 2 + t) * 2) * 1024) = pack_frag(h[0], h[1]);
                *reinterpret_cast<f16x8*>(park + ((s * 2 + t) * 2 + 1) * 1024) = pack_frag(l[0], l[1]);
            }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < CH_PT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc[c][t][r] = 0.0f; accl[c][t][r] = 0.0f; }       // now conv1's accumulators
        // ---- tail: conv3 in eight half-groups of 32 output channels (= one K-slab of conv1'), pipelined by one stage
        f16x8 yh[CH_PT], yl[CH_PT];
#pragma unroll
        for (int t = 0; t < CH_PT; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) { yh[t][e] = (_Float16)0.0f; yl[t][e] = (_Float16)0.0f; }
#pragma unroll
        for (int j = 0; j < CH_TAIL_STAGES; ++j) {       // (unrolled: the shortcut slots are registers)
            tick();
            if (CH_LGKM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            probe();
            tick();
            const uint8_t* wb = smem + buf * CH_BUF;
            if (++buf == CH_D) buf = 0;
            // Order of a stage: conv3 tile q = 0 | conv3 tile q = 1 with the epilogue of q = 0 | conv1' of K-slab j - 1 with the epilogue of
            // q = 1.  Each epilogue is ~90 VALU instructions behind 12 / 24 independent MFMAs.
            f32x4 a3[2][CH_PT], a3l[2][CH_PT];
            u32x2 h[2][CH_PT], l[2][CH_PT];
            auto conv3_tile = [&](int q) {
#pragma unroll
                for (int t = 0; t < CH_PT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { a3[q][t][r] = 0.0f; a3l[q][t][r] = 0.0f; }
                if (CH_ABL & 4) return;
                constexpr int NS = PROJ ? 4 : 2;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const f16x8 ah = *reinterpret_cast<const f16x8*>(wb + CH_TW3 + ((q * NS + s) * 2) * 1024 + aoff0);
                    const f16x8 al = *reinterpret_cast<const f16x8*>(wb + CH_TW3 + ((q * NS + s) * 2 + 1) * 1024 + aoff0);
#pragma unroll
                    for (int t = 0; t < CH_PT; ++t) {
                        const f16x8 mh = s < 2 ? *reinterpret_cast<const f16x8*>(park + (((s & 1) * 2 + t) * 2) * 1024) : __builtin_bit_cast(f16x8, xfh[s & 1][t]);
                        const f16x8 ml = s < 2 ? *reinterpret_cast<const f16x8*>(park + (((s & 1) * 2 + t) * 2 + 1) * 1024) : __builtin_bit_cast(f16x8, xfl[s & 1][t]);
                        a3l[q][t] = CH_MM(ah, ml, a3l[q][t]);
                        a3[q][t] = CH_MM(ah, mh, a3[q][t]);
                        a3l[q][t] = CH_MM(al, mh, a3l[q][t]);
                    }
                }
            };
            // y = relu(conv3 + b3 + shortcut) of channel tile q: its planes for the store and for the B fragment of conv1's K-slab j
            u32x2 rq_h[2][CH_PT], rq_l[2][CH_PT];
            auto epilogue = [&](int q) {
                if (CH_ABL & 8) {
#pragma unroll
                    for (int t = 0; t < CH_PT; ++t) { h[q][t] = u32x2{0u, 0u}; l[q][t] = u32x2{0u, 0u}; }
                    return;
                }
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_lds + 64 + 32 * j + 16 * q + 4 * kc);
#pragma unroll
                for (int t = 0; t < CH_PT; ++t) {
                    const f16x4 rh = __builtin_bit_cast(f16x4, rq_h[q][t]);
                    const f16x4 rl = __builtin_bit_cast(f16x4, rq_l[q][t]);
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = __builtin_fmaf(a3[q][t][r] + a3l[q][t][r] * ls, a.scale3, bv[r]);
                        if constexpr (!PROJ) v[r] += __builtin_fmaf((float)rl[r], ls, (float)rh[r]);
                        v[r] = __builtin_fmaxf(v[r], 0.0f);
                        rng = max(rng, __builtin_bit_cast(unsigned, v[r]));
                    }
                    split4_f16(v, h[q][t], l[q][t]);
                }
            };
            if (j < CH_TAIL_STAGES - 1) {
#pragma unroll
                for (int t = 0; t < CH_PT; ++t) {
                    if constexpr (PROJ) { rq_h[0][t] = rq_h[1][t] = rq_l[0][t] = rq_l[1][t] = u32x2{0u, 0u}; continue; }
                    slab_scatter(rsh[j % CH_RSLOTS][t], rq_h[0][t], rq_h[1][t]);
                    slab_scatter(rsl[j % CH_RSLOTS][t], rq_l[0][t], rq_l[1][t]);
                }
                conv3_tile(0);
                __builtin_amdgcn_sched_barrier(0);
                conv3_tile(1);
                epilogue(0);
#pragma unroll
                for (int k = 0; k < ((CH_ZX & 4) ? 0 : (PROJ ? 24 : 12)); ++k) {               // one MFMA of tile 1, then a share of tile 0's epilogue
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, PROJ ? 4 : 8, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (j > 0 && HAS_Z && !(CH_ABL & 4) && !(CH_ZX & 2)) {
                // conv1', K-slab j - 1: the y fragments of the previous stage
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f16x8 ah = *reinterpret_cast<const f16x8*>(wb + (PROJ ? CH_TW1_P : CH_TW1) + (c * 2) * 1024 + aoff0);
                    const f16x8 al = *reinterpret_cast<const f16x8*>(wb + (PROJ ? CH_TW1_P : CH_TW1) + (c * 2 + 1) * 1024 + aoff0);
#pragma unroll
                    for (int t = 0; t < CH_PT; ++t) {
                        accl[c][t] = CH_MM(ah, yl[t], accl[c][t]);
                        acc[c][t] = CH_MM(ah, yh[t], acc[c][t]);
                        accl[c][t] = CH_MM(al, yh[t], accl[c][t]);
                    }
                }
            }
            if (j < CH_TAIL_STAGES - 1) {
                epilogue(1);
#pragma unroll
                for (int k = 0; k < ((CH_ZX & 4) ? 0 : 24); ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < CH_PT; ++t) {
                    yh[t] = pack_frag(h[0][t], h[1][t]);
                    yl[t] = pack_frag(l[0][t], l[1][t]);
                    if (!(CH_ABL & 1)) {
                        store16(slab_gather(h[0][t], h[1][t]), yr, po[t], j * a.np_y * 64);
                        store16(slab_gather(l[0][t], l[1][t]), yr, po[t], j * a.np_y * 64 + (int)a.ps_y);
                    }
                }
                // the slot is free: refill it with the half-group four stages on (this tile's, or the next tile's first four)
                if constexpr (!PROJ) {
                    if (j < 8 - CH_RSLOTS) load_res(j % CH_RSLOTS, j + CH_RSLOTS, po);
                    else load_res(j % CH_RSLOTS, j + CH_RSLOTS - 8, po_next);
                }
            }
        }
        // ---- z = relu(conv1' + b1')
        if constexpr (HAS_Z) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                for (int t = 0; t < CH_PT; ++t) {
                    u32x2 zh[2], zl[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int c = 2 * sl + q;
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_lds + 320 + 16 * c + 4 * kc);
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[r] = __builtin_fmaxf(__builtin_fmaf(acc[c][t][r] + accl[c][t][r] * ls, a.scale1, bv[r]), 0.0f);
                            rng = max(rng, __builtin_bit_cast(unsigned, v[r]));
                        }
                        split4_f16(v, zh[q], zl[q]);
                    }
                    if (!(CH_ABL & 1) && !(CH_ZX & 1)) {
                        store16(slab_gather(zh[0], zh[1]), zr, po[t], sl * a.np_z * 64);
                        store16(slab_gather(zl[0], zl[1]), zr, po[t], sl * a.np_z * 64 + (int)a.ps_z);
                    }
                }
        }
    }
    tick();
    probe_end();
    if (rng > 0x477fe000u && a.range_flag) *reinterpret_cast<volatile int*>(a.range_flag) = 1;
#endif
}

// Tail weights, stage j = 0 .. 8 of 16 KB (24 KB in the projection form):
//   conv3 tiles [q 2][s NS][plane 2] x 1 KB of half-group j (rows = output channels 32 j + 16 q + r16; zeros for j = 8): K-slabs s = 0, 1 are
//   the 64 channels of mid2 in the CHAINED order -- value e of chunk kc = channel 32 s + (e < 4 ? 4 kc + e : 16 + 4 kc + e - 4) -- and, in the
//   projection form (NS = 4), s = 2, 3 the 64 channels of the block input in their natural order (8 kc + e) from the projection weight wds,
//   conv1' tiles [c 4][plane 2] x 1 KB of K-slab j - 1 (rows = output channels 16 c + r16, inputs 32 (j - 1) .. in the chained order; zeros for j = 0),
// each 1-KB tile in the fragment layout lds_off(row, chunk kc).  One thread per (stage, tile, row, chunk).
__global__ __launch_bounds__(256) void chain_pack_tail_kernel(const float* __restrict__ w3, const float* __restrict__ wds, const float* __restrict__ w1,
                                                              uint8_t* __restrict__ wt, float ws3, float ws1)
{
    const int ns = wds ? 4 : 2, n3 = 2 * ns, ntl = n3 + 4;
    const int idx = blockIdx.x * 256 + threadIdx.x;                // 9 stages x (8 | 12) tiles x 16 rows x 4 chunks
    if (idx >= CH_TAIL_STAGES * ntl * 64) return;
    const int kcq = idx & 3, row = (idx >> 2) & 15, tl = (idx >> 6) % ntl, j = (idx >> 6) / ntl;
    unsigned pl[2][4];
    const bool is3 = tl < n3;                                      // conv3 tiles (q, s) first, then conv1' tiles c
    const int q = tl / ns, s = tl - q * ns, c = tl - n3;
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
        float v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e = 2 * e2 + h;
            const int perm = e < 4 ? 4 * kcq + e : 16 + 4 * kcq + (e - 4);
            const int orow = 32 * j + 16 * q + row;
            if (is3 && s < 2) v[h] = j < 8 ? w3[(size_t)orow * CH_P + 32 * s + perm] * ws3 : 0.0f;
            else if (is3) v[h] = j < 8 ? wds[(size_t)orow * CH_P + 32 * (s - 2) + 8 * kcq + e] * ws3 : 0.0f;
            else v[h] = (w1 && j > 0) ? w1[(size_t)(16 * c + row) * (4 * CH_P) + 32 * (j - 1) + perm] * ws1 : 0.0f;
        }
        split2_f16(f32x2{v[0], v[1]}, pl[0][e2], pl[1][e2]);
    }
    uint8_t* base = wt + (size_t)j * (wds ? CH_TWP_BYTES : CH_TW_BYTES);
    uint8_t* tile = is3 ? base + CH_TW3 + ((q * ns + s) * 2) * 1024 : base + (wds ? CH_TW1_P : CH_TW1) + (c * 2) * 1024;
    *reinterpret_cast<u32x4*>(tile + lds_off(row, kcq)) = u32x4{pl[0][0], pl[0][1], pl[0][2], pl[0][3]};
    *reinterpret_cast<u32x4*>(tile + 1024 + lds_off(row, kcq)) = u32x4{pl[1][0], pl[1][1], pl[1][2], pl[1][3]};
}

}  // namespace

#if CH_TIMING || CH_PROBE
static unsigned long long* g_chain_dbg = nullptr;
extern "C" void stm_debug_chain_timing(void* p) { g_chain_dbg = static_cast<unsigned long long*>(p); }
#endif

extern "C" size_t stm_chain_tail_weight_bytes(void) { return (size_t)CH_TAIL_STAGES * CH_TW_BYTES; }
extern "C" size_t stm_chain_tail_weight_bytes_proj(void) { return (size_t)CH_TAIL_STAGES * CH_TWP_BYTES; }

static int chain_pack_tail(const char* who, const float* w3, const float* wds, const float* w1_next, void* packed, float wscale3, float wscale1,
                           stm_stream_t stream)
{
    STM_REQUIRE(w3 && packed, STM_ENULL, "%s: w3 / packed must be non-NULL", who);
    STM_REQUIRE((uintptr_t)packed % 16 == 0 && wscale3 > 0.0f && (!w1_next || wscale1 > 0.0f), STM_EINVAL, "%s: alignment / scales", who);
    hipLaunchKernelGGL(chain_pack_tail_kernel, dim3(stm_cdiv(CH_TAIL_STAGES * (wds ? 12 : 8) * 64, 256)), dim3(256), 0, stm_hs(stream), w3, wds, w1_next,
                       static_cast<uint8_t*>(packed), wscale3, wscale1);
    STM_CHECK_LAUNCH("chain_pack_tail_kernel");
    return STM_OK;
}

// conv3 weight [256][64] and (optional) the next conv1 weight [64][256], both 1x1 OIHW fp32, times their power-of-two scales
extern "C" int stm_chain_pack_tail_f32(const float* w3, const float* w1_next, void* packed, float wscale3, float wscale1, stm_stream_t stream)
{
    return chain_pack_tail("stm_chain_pack_tail_f32", w3, nullptr, w1_next, packed, wscale3, wscale1, stream);
}

// projection form: conv3 [256][64] and the shortcut's 1x1 projection [256][64] share wscale3 (they are one product over [mid2 ; x0])
extern "C" int stm_chain_pack_tail_proj_f32(const float* w3, const float* wds, const float* w1_next, void* packed, float wscale3, float wscale1,
                                            stm_stream_t stream)
{
    STM_REQUIRE(wds, STM_ENULL, "stm_chain_pack_tail_proj_f32: wds must be non-NULL");
    return chain_pack_tail("stm_chain_pack_tail_proj_f32", w3, wds, w1_next, packed, wscale3, wscale1, stream);
}

static int chain_launch(const char* who, bool proj, const void* mid1_planes, const void* x_planes, void* y_planes, void* z_planes, const void* w2_packed,
                        const void* tail_packed, const float* b2, const float* b3, const float* b1_next, float out_scale2, float out_scale3,
                        float out_scale1, int B, int H, int W, stm_stream_t stream)
{
    STM_REQUIRE(mid1_planes && x_planes && y_planes && w2_packed && tail_packed, STM_ENULL, "%s: NULL argument", who);
    STM_REQUIRE(B > 0 && H > 0 && W > 0 && (int64_t)B * H * W < ((int64_t)1 << 24), STM_EINVAL, "%s: bad image batch", who);
    const int64_t M = (int64_t)B * H * W;
    STM_REQUIRE(2 * 8 * M * 64 < ((int64_t)1 << 31), STM_EUNSUPPORTED, "%s: more than 2 GiB per tensor", who);
    for (const void* p : {mid1_planes, x_planes, (const void*)y_planes, (const void*)z_planes, w2_packed, tail_packed})
        STM_REQUIRE((uintptr_t)p % 16 == 0, STM_EINVAL, "%s: 16-byte alignment required", who);
    ChainArgs a;
    a.xin = static_cast<const uint8_t*>(mid1_planes); a.res = static_cast<const uint8_t*>(x_planes);
    a.y = static_cast<uint8_t*>(y_planes); a.z = static_cast<uint8_t*>(z_planes);
    a.w2 = static_cast<const uint8_t*>(w2_packed); a.wt = static_cast<const uint8_t*>(tail_packed);
    a.b2 = b2; a.b3 = b3; a.b1 = b1_next;
    a.scale2 = out_scale2 > 0.0f ? out_scale2 : 1.0f; a.scale3 = out_scale3 > 0.0f ? out_scale3 : 1.0f; a.scale1 = out_scale1 > 0.0f ? out_scale1 : 1.0f;
    a.B = B; a.H = H; a.W = W; a.M = (int)M;
    a.np_in = a.np_res = a.np_y = a.np_z = (int)M;
    a.ps_in = 2 * M * 64; a.ps_res = (proj ? 2 : 8) * M * 64; a.ps_y = 8 * M * 64; a.ps_z = 2 * M * 64;
    a.plane_bytes_in = (unsigned)(2 * M * 64);
    a.tiles = stm_cdiv(M, CH_BM);
    a.range_flag = stm_internal_range_flag();
#if CH_TIMING || CH_PROBE
    a.dbg = g_chain_dbg;
#else
    a.dbg = nullptr;
#endif
    const size_t lds = CH_LDS;
    static std::atomic<bool> reserved[CH_MAX_DEVICES];
    static std::atomic<int> n_cus[CH_MAX_DEVICES];
    int dev = 0;
    const bool have_dev = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < CH_MAX_DEVICES;
    if (!have_dev || !reserved[dev].load(std::memory_order_relaxed)) {
        const void* fns[4] = {reinterpret_cast<const void*>(conv_chain_kernel<true, false>), reinterpret_cast<const void*>(conv_chain_kernel<false, false>),
                              reinterpret_cast<const void*>(conv_chain_kernel<true, true>), reinterpret_cast<const void*>(conv_chain_kernel<false, true>)};
        for (const void* fn : fns)
            STM_REQUIRE(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess, STM_ELAUNCH,
                        "%s: cannot reserve %zu bytes of LDS", who, lds);
        if (have_dev) reserved[dev].store(true, std::memory_order_relaxed);
    }
    int cus = have_dev ? n_cus[dev].load(std::memory_order_relaxed) : 0;
    if (cus <= 0) {
        hipDeviceProp_t prop;
        cus = (have_dev && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        if (have_dev) n_cus[dev].store(cus, std::memory_order_relaxed);
    }
    const int grid = std::min((a.tiles + 7) / 8 * 8, cus / 8 * 8 > 0 ? cus / 8 * 8 : 8);       // a multiple of 8: the id -> tile map needs id & 7 = XCD
    const dim3 g(grid), b(CH_THREADS);
    if (proj) {
        if (a.z) hipLaunchKernelGGL((conv_chain_kernel<true, true>), g, b, lds, stm_hs(stream), a);
        else hipLaunchKernelGGL((conv_chain_kernel<false, true>), g, b, lds, stm_hs(stream), a);
    } else {
        if (a.z) hipLaunchKernelGGL((conv_chain_kernel<true, false>), g, b, lds, stm_hs(stream), a);
        else hipLaunchKernelGGL((conv_chain_kernel<false, false>), g, b, lds, stm_hs(stream), a);
    }
    STM_CHECK_LAUNCH("conv_chain_kernel");
    return STM_OK;
}

// mid1 [B, H, W, 64] planes, shortcut x [B, H, W, 256] planes -> y [B, H, W, 256] planes (and z [B, H, W, 64] planes when z_planes and the
// packed tail holds the next block's conv1).  w2_packed: stm_conv_pack_weights_kxr_f32 of the 3x3 weight [64, 64, 3, 3] (one group, 64
// real channels, fmt 1); tail_packed: stm_chain_pack_tail_f32.  All tensors dense.
extern "C" int stm_bottleneck_chain_f32(const void* mid1_planes, const void* x_planes, void* y_planes, void* z_planes, const void* w2_packed,
                                        const void* tail_packed, const float* b2, const float* b3, const float* b1_next, float out_scale2,
                                        float out_scale3, float out_scale1, int B, int H, int W, stm_stream_t stream)
{
    return chain_launch("stm_bottleneck_chain_f32", false, mid1_planes, x_planes, y_planes, z_planes, w2_packed, tail_packed, b2, b3, b1_next, out_scale2,
                        out_scale3, out_scale1, B, H, W, stream);
}

// the same for a stage's FIRST block at stride 1: x0 [B, H, W, 64] planes is the block's input, the shortcut its 1x1 projection, folded into
// conv3's product (tail_packed: stm_chain_pack_tail_proj_f32; b3 = conv3's bias + the projection's)
extern "C" int stm_bottleneck_chain_proj_f32(const void* mid1_planes, const void* x0_planes, void* y_planes, void* z_planes, const void* w2_packed,
                                             const void* tail_packed, const float* b2, const float* b3, const float* b1_next, float out_scale2,
                                             float out_scale3, float out_scale1, int B, int H, int W, stm_stream_t stream)
{
    return chain_launch("stm_bottleneck_chain_proj_f32", true, mid1_planes, x0_planes, y_planes, z_planes, w2_packed, tail_packed, b2, b3, b1_next,
                        out_scale2, out_scale3, out_scale1, B, H, W, stream);
}
