// ldsdma_order_probe.hip -- is a counted `s_waitcnt vmcnt(N)` a landing guarantee for LDS-DMA on gfx950?
//
// /opt/skills/guides/MI355X_MICROARCH.md:512 says loads, stores, atomics and LDS-DMA count together in issue order.  Round 3 suspected the
// opposite after a ring kernel misbehaved ("Bottleneck chain", finding 1).  This program asks the hardware directly:
//
//   wave 0 of a workgroup poisons an LDS region, then issues
//     stage A: NA LDS-DMA pieces (1 KB each) from a COLD buffer (far larger than the Infinity Cache, every piece read once: HBM misses),
//     stage B: NB LDS-DMA pieces from a HOT buffer (a few KB, the same for every workgroup: L2 hits),
//   waits `s_waitcnt vmcnt(NB)` -- "all but my NB youngest operations are done" -- and checks stage A's bytes in LDS
//     * by itself, right after the wait (same-wave visibility), and
//     * by wave 1 behind an s_barrier that wave 0 joins after its wait (the ring kernels' producer -> consumer hand-over).
//   Any poison or foreign word found there contradicts in-order completion (or the visibility of a counted-off DMA).
//
// Variants (all in one run): stage B as buffer_load ... lds or as global_load_lds (mixed instruction kinds in one wave, which conv_kxr.hip
// uses), stage A with out-of-range pieces interleaved (offset bit 31: the zero-fill path of the convolution's padding), plain register loads
// and stores in between (the consumers' shortcut loads / y stores).  A control with vmcnt(NA + NB) -- no wait at all -- must FIND poison,
// otherwise the probe cannot see what it is looking for.
//
// build: hipcc -O2 --offload-arch=gfx950 scripts/ldsdma_order_probe.hip -o scripts/bin/ldsdma_order_probe     run: scripts/bin/ldsdma_order_probe [iterations]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } \
    } while (0)

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

constexpr int NA = 16, NB = 8;                   // pieces per stage (1 KB per piece: 64 lanes x 16 B)
constexpr unsigned POISON = 0xDEADBEEFu;
constexpr int LDS_BYTES = (NA + NB) * 1024;

__device__ __forceinline__ unsigned word_of(size_t dword_index, unsigned seed) { return (unsigned)(dword_index * 2654435761u) ^ seed; }

__global__ void fill_kernel(unsigned* p, size_t n, unsigned seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = word_of(i, seed);
}

struct Result {
    unsigned long long same_wave_bad, cross_wave_bad, after_full_wait_bad, checks;
    unsigned sample[8];
};

// MODE bit 0: stage B by global_load_lds (else buffer_load lds); bit 1: every fourth stage-A piece is out of range (must read as zeros);
// bit 2: register loads + stores between the stages; bit 3: CONTROL -- the wait leaves stage A in flight too
template <int MODE>
__global__ __launch_bounds__(128) void probe_kernel(const unsigned* cold, size_t cold_dwords, const unsigned* hot, unsigned* scratch, Result* res,
                                                    int iters, unsigned seed_cold, unsigned seed_hot)
{
    extern __shared__ __align__(16) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(cold), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(hot), 0, NB * 1024, 0x00020000);
    unsigned long long bad_same = 0, bad_cross = 0, bad_full = 0, checks = 0;
    unsigned smp = 0, reg_v = 0, reg_sum = 0;
    const size_t pieces_total = cold_dwords / 256;      // 1 KB pieces in the cold buffer (kept below 2 GB: 32-bit buffer offsets)
    for (int it = 0; it < iters; ++it) {
        // poison (both waves, their halves), then everybody sees it
        for (int i = threadIdx.x; i < LDS_BYTES / 4; i += 128) reinterpret_cast<volatile unsigned*>(smem)[i] = POISON;
        __syncthreads();
        // this iteration's cold pieces: a pseudo-random walk, unique per (block, iteration, piece)
        size_t piece0 = ((size_t)blockIdx.x * 7919u + (size_t)it * 104729u) * NA % (pieces_total - NA);
        if (wave == 0) {
#pragma unroll
            for (int k = 0; k < NA; ++k) {
                const bool oob = (MODE & 2) && (k & 3) == 3;
                const unsigned off = (unsigned)((piece0 + k) * 1024 + lane * 16) | (oob ? 0x80000000u : 0u);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(cr, (lds_ptr)(smem + k * 1024), 16, off, 0, 0, 0);
            }
            if (MODE & 4) {
                // register traffic of the kind the ring kernels' consumers have in flight (counted in the same vmcnt)
                reg_v = hot[(lane + it) & 63];                                    // (used only after the checks: no wait is forced here)
                scratch[(size_t)blockIdx.x * 64 + lane] = (unsigned)it;
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                if (MODE & 1) __builtin_amdgcn_global_load_lds((glb_ptr)(reinterpret_cast<const uint8_t*>(hot) + k * 1024 + lane * 16), (lds_ptr)(smem + (NA + k) * 1024), 16, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(hr, (lds_ptr)(smem + (NA + k) * 1024), 16, lane * 16, k * 1024, 0, 0);
            }
            if (MODE & 8) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB) : "memory");
            // same-wave check of stage A, inline asm so that the compiler adds no wait of its own in front of the reads
#pragma unroll
            for (int k = 0; k < NA; ++k) {
                unsigned w0;
                const unsigned addr = (unsigned)(k * 1024 + lane * 16);
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w0) : "v"(addr) : "memory");
                const bool oob = (MODE & 2) && (k & 3) == 3;
                const unsigned exp = oob ? 0u : word_of(((piece0 + k) * 1024 + lane * 16) / 4, seed_cold);
                if (w0 != exp) { ++bad_same; smp = w0; }
                ++checks;
            }
        }
        asm volatile("s_barrier" ::: "memory");
        if (wave == 1) {
#pragma unroll
            for (int k = 0; k < NA; ++k) {
                unsigned w0;
                const unsigned addr = (unsigned)(k * 1024 + lane * 16 + 4);
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w0) : "v"(addr) : "memory");
                const bool oob = (MODE & 2) && (k & 3) == 3;
                const unsigned exp = oob ? 0u : word_of(((piece0 + k) * 1024 + lane * 16 + 4) / 4, seed_cold);
                if (w0 != exp) { ++bad_cross; smp = w0; }
                ++checks;
            }
        }
        if (wave == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        reg_sum += reg_v;
        asm volatile("s_barrier" ::: "memory");
        // sanity: with everything landed both stages must be exact
        for (int i = threadIdx.x; i < (NA + NB) * 256; i += 128) {
            const int k = i / 256, d = i % 256;
            unsigned w0;
            const unsigned addr = (unsigned)(i * 4);
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w0) : "v"(addr) : "memory");
            const bool oob = (MODE & 2) && k < NA && (k & 3) == 3;
            const unsigned exp = k < NA ? (oob ? 0u : word_of((piece0 + k) * 256 + d, seed_cold)) : word_of((size_t)(k - NA) * 256 + d, seed_hot);
            if (w0 != exp) ++bad_full;
        }
        __syncthreads();
    }
    // totals
    for (int o = 32; o > 0; o >>= 1) {
        bad_same += __shfl_down(bad_same, o);
        bad_cross += __shfl_down(bad_cross, o);
        bad_full += __shfl_down(bad_full, o);
        checks += __shfl_down(checks, o);
    }
    if (lane == 0) {
        atomicAdd(&res->same_wave_bad, bad_same);
        atomicAdd(&res->cross_wave_bad, bad_cross);
        atomicAdd(&res->after_full_wait_bad, bad_full);
        atomicAdd(&res->checks, checks);
    }
    if (smp) res->sample[(blockIdx.x + wave) & 7] = smp;
    if (reg_sum == 0x7fffffffu) res->sample[7] = reg_sum;       // keeps the register load alive
}

template <int MODE>
static void run(const char* what, const unsigned* cold, size_t cold_dwords, const unsigned* hot, unsigned* scratch, Result* dres, int iters, int grid,
                unsigned seed_cold, unsigned seed_hot)
{
    CK(hipMemset(dres, 0, sizeof(Result)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe_kernel<MODE>, dim3(grid), dim3(128), LDS_BYTES, 0, cold, cold_dwords, hot, scratch, dres, iters, seed_cold, seed_hot);
    CK(hipGetLastError());
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    Result r;
    CK(hipMemcpy(&r, dres, sizeof(r), hipMemcpyDeviceToHost));
    printf("%-74s words checked %12llu | stale after vmcnt(NB): same wave %8llu, other wave behind s_barrier %8llu | after vmcnt(0) %llu | %.1f ms, %.2f TB/s of DMA\n",
           what, r.checks, r.same_wave_bad, r.cross_wave_bad, r.after_full_wait_bad, ms, (double)grid * iters * (NA + NB) * 1024 / (ms * 1e-3) / 1e12);
    if (r.same_wave_bad || r.cross_wave_bad) printf("    sample stale words: %08x %08x %08x %08x\n", r.sample[0], r.sample[1], r.sample[2], r.sample[3]);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount * 4;
    const size_t cold_bytes = (size_t)1800 << 20;        // 1.8 GB: no piece is an Infinity-Cache hit, offsets fit 31 bits
    const size_t cold_dwords = cold_bytes / 4;
    unsigned *cold, *hot, *scratch;
    Result* dres;
    CK(hipMalloc(&cold, cold_bytes));
    CK(hipMalloc(&hot, NB * 1024));
    CK(hipMalloc(&scratch, (size_t)grid * 64 * 4));
    CK(hipMalloc(&dres, sizeof(Result)));
    const unsigned seed_cold = 0x1234567u, seed_hot = 0x89abcdeu;
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, cold, cold_dwords, seed_cold);
    hipLaunchKernelGGL(fill_kernel, dim3(8), dim3(256), 0, 0, hot, (size_t)NB * 256, seed_hot);
    CK(hipDeviceSynchronize());
    printf("%s, %d CUs, grid %d x 128 threads, %d iterations, stage A = %d cold 1-KB pieces, stage B = %d hot pieces, wait = s_waitcnt vmcnt(%d)\n", prop.gcnArchName,
           prop.multiProcessorCount, grid, iters, NA, NB, NB);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("A buffer_load lds (HBM) | B buffer_load lds (L2)", cold, cold_dwords, hot, scratch, dres, iters, grid, seed_cold, seed_hot);
        run<1>("A buffer_load lds (HBM) | B global_load_lds (L2): mixed instruction kinds", cold, cold_dwords, hot, scratch, dres, iters, grid, seed_cold, seed_hot);
        run<2>("A with every 4th piece out of range (zero fill) | B buffer_load lds", cold, cold_dwords, hot, scratch, dres, iters, grid, seed_cold, seed_hot);
        run<3>("A with out-of-range pieces | B global_load_lds", cold, cold_dwords, hot, scratch, dres, iters, grid, seed_cold, seed_hot);
        run<5>("A | register load + store | B global_load_lds", cold, cold_dwords, hot, scratch, dres, iters, grid, seed_cold, seed_hot);
        run<7>("A with out-of-range pieces | register load + store | B global_load_lds", cold, cold_dwords, hot, scratch, dres, iters, grid, seed_cold, seed_hot);
        run<8>("CONTROL: no wait for stage A at all (vmcnt(NA + NB)) -- must find stale words", cold, cold_dwords, hot, scratch, dres, iters, grid, seed_cold, seed_hot);
    }
    return 0;
}
