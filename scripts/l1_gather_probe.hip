// l1_gather_probe.hip -- what a CU's vector-memory path delivers for the corner gathers of the deformable sampler (csrc/dcn_fused.hip,
// csrc/deform_im2col.hip): 16-byte loads per lane from 128-byte pixel slabs, the pixel chosen per lane group, as a function of HOW the lanes of
// one instruction are laid over the slabs:
//   pattern 0: lane = pixel + 16 * piece   (the MFMA operand layout: every lane of a quad in a different line)
//   pattern 1: lane = 4 * pixel + quarter, 64 contiguous bytes per pixel and instruction (two instructions cover a slab)          [dcn_fused v1]
//   pattern 2: lane = 4 * pixel + quarter, pieces at stride 32 B (the old sampler: 8 contiguous channels per lane, two loads)
//   pattern 3: lane = 8 * pixel + piece, the whole 128-byte slab of 8 pixels per instruction
//   pattern 4: lane = 16 * pixel + piece, 256 contiguous bytes (two adjacent slabs) of 4 pixels per instruction
// for footprints that fit L1 (16 KB per workgroup), L2 (1 MB per workgroup) or neither.  One 512-thread workgroup per CU, 8 loads in flight per wave.
// build: hipcc -O3 --offload-arch=gfx950 -o scripts/bin/l1_gather_probe scripts/l1_gather_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(512, 1) void probe(const float* __restrict__ x, float* __restrict__ out, int npix_mask, int iters, unsigned bytes)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)bytes, 0x00020000);
    // per-lane pixel group and byte offset inside the 128-byte slab for the two loads of a "corner"
    int grp, o0, o1;
    if (PAT == 0) { grp = lane & 15; o0 = (lane >> 4) * 32; o1 = o0 + 16; }
    else if (PAT == 1) { grp = lane >> 2; o0 = (lane & 3) * 16; o1 = o0 + 64; }
    else if (PAT == 2) { grp = lane >> 2; o0 = (lane & 3) * 32; o1 = o0 + 16; }
    else if (PAT == 3) { grp = lane >> 3; o0 = (lane & 7) * 16; o1 = o0; }          // second load: another pixel (grp + 8), same piece
    else { grp = lane >> 4; o0 = (lane & 15) * 16; o1 = o0; }                       // second load: another pixel pair
    const int wg_base = (blockIdx.x * 977) & npix_mask;
    unsigned h = (blockIdx.x * 8 + wave) * 2654435761u + grp * 40503u;
    f32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        f32x4 v[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            h = h * 1664525u + 1013904223u;
            const int p0 = (wg_base + (int)((h >> 8) & (unsigned)npix_mask)) & npix_mask;
            int p1 = p0;
            if (PAT == 3 || PAT == 4) { const unsigned h2 = h * 22695477u + 1u; p1 = (wg_base + (int)((h2 >> 8) & (unsigned)npix_mask)) & npix_mask; }
            const int step = PAT == 4 ? 256 : 128;
            v[2 * c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, p0 * step + o0, 0, 0));
            v[2 * c + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, p1 * step + o1, 0, 0));
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) acc += v[c];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[threadIdx.x] = acc.x;
}

template <int PAT>
static void run(const float* x, float* out, int npix, int iters, unsigned bytes, const char* what)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int cus = 256;
    probe<PAT><<<cus, 512>>>(x, out, npix - 1, 64, bytes);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<PAT><<<cus, 512>>>(x, out, npix - 1, iters, bytes);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double total = (double)cus * 8 * iters * 8 * 1024.0;      // bytes delivered to registers
    printf("pattern %d %-34s footprint %8d B/WG-window: %7.2f TB/s = %6.1f GB/s per CU = %5.1f B/clk/CU at 2.1 GHz (%.3f ms)\n", PAT, what, npix * (PAT == 4 ? 256 : 128),
           total / ms * 1e-9, total / ms * 1e-6 / cus, total / ms * 1e-6 / cus / 2.1, ms);
}

int main()
{
    const size_t bytes = 512u << 20;
    float* x; float* out;
    hipMalloc(&x, bytes); hipMalloc(&out, 4096);
    hipMemset(x, 0, bytes);
    for (int npix : {128, 8192, 1 << 21}) {          // 16 KB (L1), 1 MB (L2), 256 MB (beyond L2: Infinity Cache / HBM)
        const int iters = npix > 100000 ? 400 : 2000;
        run<0>(x, out, npix, iters, (unsigned)bytes, "lane = pixel + 16 piece");
        run<1>(x, out, npix, iters, (unsigned)bytes, "quad = 64 contiguous B");
        run<2>(x, out, npix, iters, (unsigned)bytes, "quad = 4 x 16 B at stride 32");
        run<3>(x, out, npix, iters, (unsigned)bytes, "8 lanes = one 128-B slab");
        run<4>(x, out, npix / 2, iters, (unsigned)bytes, "16 lanes = 256 contiguous B");
    }
    return 0;
}
