// vmem_store_war_probe.hip -- does a 16-byte MUBUF store read its data registers AFTER the next VALU instruction has overwritten them?
//
// The ISA's manual wait-state table (and llvm's GCNHazardRecognizer::createsVALUHazard, which hipcc relies on) says: a VMEM store of more than
// 64 bits followed by a VALU write of its data VGPRs needs wait states in between -- EXCEPT a MUBUF store whose soffset operand is an SGPR, for
// which the compiler inserts nothing.  conv_chain_kernel<true, .> (csrc/conv_chain.hip) is full of exactly that sequence,
//
//     buffer_store_dwordx4 v[24:27], v160, s[48:51], s14 offen
//     v_mov_b32_e32 v24, v16                      ; the next store's data, same registers, no wait state
//
// and its y / z planes came out wrong beside a second process -- always in lanes 12-15, 28-31, 44-47, 60-63 of a store, i.e. the last 256 bytes
// of each 1-KB wave store (scripts/ring_stress.py, profiles/r04_ring_stress_*.txt).  This program asks the hardware directly:
//
//   every wave writes pattern A to v[200:203], stores them (1 KB per wave and iteration, each to its own address), overwrites v[200:203] with
//   pattern B after W wait states (W = 0, 1, 2, 3, 5, 9), and stores B elsewhere.  A checker then counts the words of the FIRST store's region that
//   hold B: each one is a store that read its data after the overwrite.  Forms: buffer stores with the soffset in an SGPR (no compiler protection) or 0 (the
//   form the compiler pads), global stores with a 64-bit VGPR address or an SGPR base, 16 / 12 / 8 bytes per lane; 8 waves per CU streaming stores
//   (texture-path back-pressure) or one.
//
// build: hipcc -O2 --offload-arch=gfx950 scripts/vmem_store_war_probe.hip -o scripts/bin/vmem_store_war_probe      run: scripts/bin/vmem_store_war_probe [iterations]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } \
    } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline unsigned pat_a(unsigned slot, unsigned lane, unsigned d) { return (((slot * 64u + lane) * 4u + d) * 2654435761u) & 0x7fffffffu; }

struct Result {
    unsigned long long b_words, other_words, checked;
    unsigned long long by_lane[64];
};

#define NOPS_0 ""
#define NOPS_1 "s_nop 0\n\t"
#define NOPS_2 "s_nop 1\n\t"
#define NOPS_3 "s_nop 2\n\t"
#define NOPS_5 "s_nop 4\n\t"
#define NOPS_9 "s_nop 8\n\t"

// store forms: data always v[200:203] (the x2 / x3 forms store its first 2 / 3 registers, 16 bytes apart per lane all the same)
enum Form { BUF4_SGPR = 0, BUF4_IMM = 1, GLB4_VADDR = 2, GLB4_SADDR = 3, BUF2_SGPR = 4, GLB2_SADDR = 5, BUF3_SGPR = 6, N_FORMS = 7 };
__host__ __device__ constexpr int form_dwords(int f) { return (f == BUF2_SGPR || f == GLB2_SADDR) ? 2 : (f == BUF3_SGPR ? 3 : 4); }
static const char* form_name(int f)
{
    switch (f) {
        case BUF4_SGPR: return "buffer_store_dwordx4, soffset in an SGPR   ";
        case BUF4_IMM: return "buffer_store_dwordx4, soffset 0            ";
        case GLB4_VADDR: return "global_store_dwordx4 v[addr64], off        ";
        case GLB4_SADDR: return "global_store_dwordx4 v_off, s[base]        ";
        case BUF2_SGPR: return "buffer_store_dwordx2, soffset in an SGPR   ";
        case GLB2_SADDR: return "global_store_dwordx2 v_off, s[base]        ";
        default: return "buffer_store_dwordx3, soffset in an SGPR   ";
    }
}

#define SEQ(ST1, NOPS, ST2)                                                                                                                          \
    asm volatile("v_mov_b32 v200, %[a0]\n\tv_mov_b32 v201, %[a1]\n\tv_mov_b32 v202, %[a2]\n\tv_mov_b32 v203, %[a3]\n\t"                            \
                 "s_nop 7\n\t" ST1 "\n\t" NOPS                                                                                                      \
                 "v_mov_b32 v200, %[b0]\n\tv_mov_b32 v201, %[b1]\n\tv_mov_b32 v202, %[b2]\n\tv_mov_b32 v203, %[b3]\n\t"                            \
                 "s_nop 7\n\t" ST2 "\n\t"                                                                                                           \
                 "s_nop 7\n\t"                                                                                                                      \
                 :                                                                                                                                  \
                 : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [b0] "v"(a0 | 0x80000000u), [b1] "v"(a1 | 0x80000000u),                  \
                   [b2] "v"(a2 | 0x80000000u), [b3] "v"(a3 | 0x80000000u), [vo] "v"(vo), [vo2] "v"(vo2), [rs] "s"(rs), [so] "s"(so),                \
                   [ga] "v"(ga), [ga2] "v"(ga2), [sb] "s"(ba)                                                                                       \
                 : "v200", "v201", "v202", "v203", "memory")

#define SEQ_FORM(NOPS)                                                                                                                               \
    if constexpr (FORM == BUF4_SGPR) SEQ("buffer_store_dwordx4 v[200:203], %[vo], %[rs], %[so] offen", NOPS, "buffer_store_dwordx4 v[200:203], %[vo2], %[rs], %[so] offen"); \
    else if constexpr (FORM == BUF4_IMM) SEQ("buffer_store_dwordx4 v[200:203], %[vo], %[rs], 0 offen", NOPS, "buffer_store_dwordx4 v[200:203], %[vo2], %[rs], 0 offen"); \
    else if constexpr (FORM == GLB4_VADDR) SEQ("global_store_dwordx4 %[ga], v[200:203], off", NOPS, "global_store_dwordx4 %[ga2], v[200:203], off"); \
    else if constexpr (FORM == GLB4_SADDR) SEQ("global_store_dwordx4 %[vo], v[200:203], %[sb]", NOPS, "global_store_dwordx4 %[vo2], v[200:203], %[sb]"); \
    else if constexpr (FORM == BUF2_SGPR) SEQ("buffer_store_dwordx2 v[200:201], %[vo], %[rs], %[so] offen", NOPS, "buffer_store_dwordx2 v[200:201], %[vo2], %[rs], %[so] offen"); \
    else if constexpr (FORM == GLB2_SADDR) SEQ("global_store_dwordx2 %[vo], v[200:201], %[sb]", NOPS, "global_store_dwordx2 %[vo2], v[200:201], %[sb]"); \
    else SEQ("buffer_store_dwordx3 v[200:202], %[vo], %[rs], %[so] offen", NOPS, "buffer_store_dwordx3 v[200:202], %[vo2], %[rs], %[so] offen")

template <int W, int FORM>
__global__ __launch_bounds__(256) void store_kernel(unsigned* buf, unsigned half_bytes, int iters, unsigned slots_per_iter)
{
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    // the buffer descriptor by hand (base, stride 0, 2^31 - 1 records, raw dword format): inline asm wants it as four SGPRs
    const unsigned long long ba = reinterpret_cast<unsigned long long>(buf);
    const u32x4 rs = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ba), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((ba >> 32) & 0xffffu)), 0x7fffffffu,
                      0x00020000u};
    constexpr bool SGOFF = FORM == BUF4_SGPR || FORM == BUF2_SGPR || FORM == BUF3_SGPR;     // the slot's byte offset rides in the SGPR soffset
    for (int it = 0; it < iters; ++it) {
        const unsigned slot = (unsigned)it * slots_per_iter + wave;
        const unsigned a0 = pat_a(slot, lane, 0), a1 = pat_a(slot, lane, 1), a2 = pat_a(slot, lane, 2), a3 = pat_a(slot, lane, 3);
        const unsigned so = SGOFF ? slot * 1024u : 0u;
        const unsigned vo = lane * 16u + (SGOFF ? 0u : slot * 1024u), vo2 = vo + half_bytes;
        const unsigned long long ga = ba + (unsigned long long)slot * 1024u + lane * 16u, ga2 = ga + half_bytes;
        if constexpr (W == 0) { SEQ_FORM(NOPS_0); }
        if constexpr (W == 1) { SEQ_FORM(NOPS_1); }
        if constexpr (W == 2) { SEQ_FORM(NOPS_2); }
        if constexpr (W == 3) { SEQ_FORM(NOPS_3); }
        if constexpr (W == 5) { SEQ_FORM(NOPS_5); }
        if constexpr (W == 9) { SEQ_FORM(NOPS_9); }
    }
}

__global__ void check_kernel(const unsigned* buf, size_t words, Result* res, int nd)
{
    unsigned long long nb = 0, no = 0, nc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned slot = (unsigned)(i >> 8), lane = (unsigned)(i >> 2) & 63u, d = (unsigned)i & 3u;
        if ((int)d >= nd) continue;                 // (the x2 / x3 forms leave the rest of a lane's 16 bytes untouched)
        const unsigned a = pat_a(slot, lane, d), w = buf[i];
        ++nc;
        if (w == a) continue;
        if (w == (a | 0x80000000u)) { ++nb; atomicAdd(&res->by_lane[lane], 1ull); }
        else ++no;
    }
    atomicAdd(&res->b_words, nb);
    atomicAdd(&res->other_words, no);
    atomicAdd(&res->checked, nc);
}

template <int W, int FORM>
static void run(unsigned* buf, size_t half_bytes, Result* dres, int iters, int grid, int threads)
{
    const unsigned slots_per_iter = (unsigned)grid * threads / 64;
    const size_t used = (size_t)iters * slots_per_iter * 1024;
    CK(hipMemset(buf, 0xff, used));
    CK(hipMemset(dres, 0, sizeof(Result)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((store_kernel<W, FORM>), dim3(grid), dim3(threads), 0, 0, buf, (unsigned)half_bytes, iters, slots_per_iter);
    CK(hipGetLastError());
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    hipLaunchKernelGGL(check_kernel, dim3(2048), dim3(256), 0, 0, buf, used / 4, dres, form_dwords(FORM));
    CK(hipDeviceSynchronize());
    Result r;
    CK(hipMemcpy(&r, dres, sizeof(r), hipMemcpyDeviceToHost));
    printf("%s %d wait state(s) before the overwrite  %4d x %3d threads: %12llu words stored, %10llu hold the NEW register contents, %llu other; %.2f TB/s",
           form_name(FORM), W, grid, threads, r.checked, r.b_words, r.other_words, 2.0 * used * form_dwords(FORM) / 4 / (ms * 1e-3) / 1e12);
    if (r.b_words) {
        printf("; lanes:");
        for (int l = 0; l < 64; ++l)
            if (r.by_lane[l]) printf(" %d", l);
    }
    printf("\n");
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t half_bytes = (size_t)1 << 30;           // A stores in [0, 1 GB), B stores in [1 GB, 2 GB): 32-bit buffer offsets
    unsigned* buf;
    Result* dres;
    CK(hipMalloc(&buf, 2 * half_bytes));
    CK(hipMalloc(&dres, sizeof(Result)));
    printf("%s, %d CUs; every wave: store v[200:203] (16 B per lane, 1 KB), W wait states, overwrite v[200:203], store again; %d iterations per wave\n", prop.gcnArchName, cus, iters);
    for (int rep = 0; rep < 2; ++rep) {
        const int it8 = iters, it1 = iters * 4;
        // 8 waves per CU (two 256-thread blocks): the store path saturated
#define ROW(F) run<0, F>(buf, half_bytes, dres, it8, cus * 2, 256); run<1, F>(buf, half_bytes, dres, it8, cus * 2, 256); \
               run<2, F>(buf, half_bytes, dres, it8, cus * 2, 256); run<3, F>(buf, half_bytes, dres, it8, cus * 2, 256);
        ROW(BUF4_SGPR) run<5, BUF4_SGPR>(buf, half_bytes, dres, it8, cus * 2, 256); run<9, BUF4_SGPR>(buf, half_bytes, dres, it8, cus * 2, 256);
        ROW(BUF4_IMM) ROW(GLB4_VADDR) ROW(GLB4_SADDR) ROW(BUF3_SGPR) ROW(BUF2_SGPR) ROW(GLB2_SADDR)
        // one wave per CU: an idle store path
        run<0, BUF4_SGPR>(buf, half_bytes, dres, it1, cus, 64);
        run<0, BUF4_IMM>(buf, half_bytes, dres, it1, cus, 64);
        run<0, GLB4_SADDR>(buf, half_bytes, dres, it1, cus, 64);
    }
    return 0;
}
