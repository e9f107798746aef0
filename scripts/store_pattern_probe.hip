// store_pattern_probe.hip -- at what rate can a kernel write the deformable sampler's columns?  (round 4: the sampler's store stream alone -- corner loads
// compiled out -- runs at 4.35 TB/s on the C = 128 layers while torch.fill_ writes the same bytes at 6.8.)
// Column tensor of layer2.2 at batch 32: planes [2][36 slabs][M = 122 880 px][32 ch] fp16 = 566 MB.  Patterns:
//   A  the sampler's: a wave = 4 pixels x 128 channels of one tap: one store instruction = four 256-B segments in four slab planes
//   B  a wave = 16 pixels x 32 channels (one slab) of one tap: one store instruction = 1 KB contiguous; a workgroup's four waves take the tap's four slabs
//   C  as B, but a workgroup owns 64 consecutive pixels: each wave writes 4 KB contiguous per (tap, slab) before it moves on
//   D  linear fill of the same bytes (1 KB per wave instruction, consecutive waves consecutive KB)
// each with plain and nontemporal stores.  build: hipcc -O2 --offload-arch=gfx950 scripts/store_pattern_probe.hip -o scripts/bin/store_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int M = 122880, SLABS = 36, TAPS = 9;
constexpr size_t PLANE = (size_t)SLABS * M * 64;      // bytes of one plane

template <bool NT> __device__ __forceinline__ void st(uint8_t* p, u32x4 v)
{
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
    else *reinterpret_cast<u32x4*>(p) = v;
}

template <int PAT, bool NT>
__global__ __launch_bounds__(256) void k(uint8_t* out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4 v = {(unsigned)threadIdx.x, blockIdx.x, 3u, 4u};
    if (PAT == 0) {            // A: workgroup = 16 pixels; lane = (pixel p of 4, sub-lane sl of 16)
        const int m = (blockIdx.x * 4 + wave) * 4 + lane / 16, sl = lane % 16;
        for (int t = 0; t < TAPS; ++t) {
            uint8_t* o = out + ((size_t)(t * 4 + sl / 4) * M + m) * 64 + (sl % 4) * 16;
            st<NT>(o, v); st<NT>(o + PLANE, v);
        }
    } else if (PAT == 1) {     // B: workgroup = 16 pixels; wave = slab of the tap; lane = (pixel of 16, chunk of 4)
        const int m = blockIdx.x * 16 + lane / 4;
        for (int t = 0; t < TAPS; ++t) {
            uint8_t* o = out + ((size_t)(t * 4 + wave) * M + m) * 64 + (lane % 4) * 16;
            st<NT>(o, v); st<NT>(o + PLANE, v);
        }
    } else if (PAT == 2) {     // C: workgroup = 64 pixels; wave = slab; four runs of 16 pixels per (tap, slab)
        for (int t = 0; t < TAPS; ++t)
            for (int r = 0; r < 4; ++r) {
                const int m = blockIdx.x * 64 + r * 16 + lane / 4;
                uint8_t* o = out + ((size_t)(t * 4 + wave) * M + m) * 64 + (lane % 4) * 16;
                st<NT>(o, v); st<NT>(o + PLANE, v);
            }
    } else {                   // D: linear
        const size_t total = 2 * PLANE / 16;
        for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) st<NT>(out + i * 16, v);
    }
}

template <int PAT, bool NT> static void run(const char* name, uint8_t* out, int grid)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<PAT, NT>), dim3(grid), dim3(256), 0, 0, out);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<PAT, NT>), dim3(grid), dim3(256), 0, 0, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-88s %s: %7.1f us  %.2f TB/s\n", name, NT ? "nontemporal" : "plain      ", ms * 100, 2.0 * PLANE / (ms * 1e-4) / 1e12);
}

int main()
{
    uint8_t* out; CK(hipMalloc(&out, 2 * PLANE));
    for (int rep = 0; rep < 2; ++rep) {
        run<0, true>("A  4 px x 128 ch per wave: four 256-B segments per store instruction (the sampler's)", out, M / 16);
        run<0, false>("A", out, M / 16);
        run<1, true>("B  16 px x 32 ch per wave: 1 KB contiguous per store instruction", out, M / 16);
        run<1, false>("B", out, M / 16);
        run<2, true>("C  as B, workgroup = 64 px: 4 KB contiguous per wave and (tap, slab)", out, M / 64);
        run<2, false>("C", out, M / 64);
        run<3, true>("D  linear fill", out, 256 * 16);
        run<3, false>("D", out, 256 * 16);
    }
    return 0;
}
