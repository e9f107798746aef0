// gemm_f32.hip -- fp32 GEMM on the gfx950 matrix pipe (v_mfma_f32_32x32x2_f32) with fused bias / ReLU.
//
// The GEMM half of the deformable convolution: y[b] = W[O, C*K] x cols[b][C*K, Ho*Wo] + bias
// (dcn_v2 / mmcv call cuBLAS sgemm here; backbone.py:45, Featurealign.py:72).  fp32-in / fp32-accumulate
// MFMA is an exact k-ordered fmaf chain (MI355X guide §3), so results are deterministic and independent of
// the tiling.  MFMA-bound: 157 TFLOP/s dense fp32 peak.
//
// Tiling: workgroup 256 threads = 4 waves (2x2); block tile BM x BN, wave tile (BM/2) x (BN/2) made of 32x32
// MFMA tiles; BK = 16.  A (weights, [M][K] row-major) is transposed on the way into LDS (As[BK][BM+2]: the +2
// pad makes the transposing ds_write_b32 conflict-free), B (columns, [K][N] row-major) is copied as is
// (Bs[BK][BN]).  Operand fetch is one conflict-free ds_read_b32 per 32x32x2 MFMA operand.  The next K-slab is
// prefetched global->registers while the current one is multiplied.
#include "stm_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BK = 16;

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_bias_f32_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                            const float* __restrict__ bias, float* __restrict__ Cm,
                                                            int M, int N, int Kd, int64_t b_bs, int64_t c_bs,
                                                            int relu, int n_tiles)
{
    constexpr int TM = BM / 64;       // 32x32 tiles per wave along M
    constexpr int TN = BN / 64;       // along N
    constexpr int LDA = BM + 2;       // padded leading dimension of As
    constexpr int A_F4 = BM * BK / 4 / 256;  // float4 loads per thread for the A slab
    constexpr int B_F4 = BN * BK / 4 / 256;  // for the B slab
    static_assert(A_F4 >= 1 && B_F4 >= 1, "tile too small for 256 threads");

    __shared__ float As[BK * LDA];
    __shared__ float Bs[BK * BN];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_n = blockIdx.x % n_tiles, tile_m = blockIdx.x / n_tiles;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int batch = blockIdx.y;
    const float* Bp = Bm + (int64_t)batch * b_bs;
    float* Cp = Cm + (int64_t)batch * c_bs;

    const bool n_vec = (N % 4 == 0);  // rows of B are 16-byte aligned
    const bool k_vec = (Kd % 4 == 0);

    float4 ra[A_F4], rb[B_F4];

    auto load_slab = [&](int k0) {
#pragma unroll
        for (int t = 0; t < A_F4; ++t) {
            int f = tid + t * 256;          // float4 index in the [BM][BK/4] slab
            int row = f / (BK / 4), kq = f % (BK / 4);
            int gm = m0 + row, gk = k0 + kq * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gm < M) {
                const float* p = A + (int64_t)gm * Kd + gk;
                if (k_vec && gk + 3 < Kd) {
                    v = *reinterpret_cast<const float4*>(p);
                } else {
                    if (gk < Kd) v.x = p[0];
                    if (gk + 1 < Kd) v.y = p[1];
                    if (gk + 2 < Kd) v.z = p[2];
                    if (gk + 3 < Kd) v.w = p[3];
                }
            }
            ra[t] = v;
        }
#pragma unroll
        for (int t = 0; t < B_F4; ++t) {
            int f = tid + t * 256;          // float4 index in the [BK][BN/4] slab
            int kr = f / (BN / 4), nq = f % (BN / 4);
            int gk = k0 + kr, gn = n0 + nq * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gk < Kd) {
                const float* p = Bp + (int64_t)gk * N + gn;
                if (n_vec && gn + 3 < N) {
                    v = *reinterpret_cast<const float4*>(p);
                } else {
                    if (gn < N) v.x = p[0];
                    if (gn + 1 < N) v.y = p[1];
                    if (gn + 2 < N) v.z = p[2];
                    if (gn + 3 < N) v.w = p[3];
                }
            }
            rb[t] = v;
        }
    };

    auto store_slab = [&]() {
#pragma unroll
        for (int t = 0; t < A_F4; ++t) {
            int f = tid + t * 256;
            int row = f / (BK / 4), kq = f % (BK / 4);
            As[(kq * 4 + 0) * LDA + row] = ra[t].x;
            As[(kq * 4 + 1) * LDA + row] = ra[t].y;
            As[(kq * 4 + 2) * LDA + row] = ra[t].z;
            As[(kq * 4 + 3) * LDA + row] = ra[t].w;
        }
#pragma unroll
        for (int t = 0; t < B_F4; ++t) {
            int f = tid + t * 256;
            int kr = f / (BN / 4), nq = f % (BN / 4);
            *reinterpret_cast<float4*>(&Bs[kr * BN + nq * 4]) = rb[t];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int lrow = lane & 31, lk = lane >> 5;
    const int a_base = wm * (BM / 2) + lrow;
    const int b_base = wn * (BN / 2) + lrow;

    load_slab(0);
    for (int k0 = 0; k0 < Kd; k0 += BK) {
        __syncthreads();  // previous slab fully consumed
        store_slab();
        __syncthreads();
        if (k0 + BK < Kd) load_slab(k0 + BK);  // prefetch the next slab behind the MFMAs
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = As[(kk + lk) * LDA + a_base + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = Bs[(kk + lk) * BN + b_base + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            int gn = n0 + wn * (BN / 2) + j * 32 + lrow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int gm = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (gm < M && gn < N) {
                    float v = acc[i][j][r];
                    if (bias) v = v + bias[gm];
                    if (relu) v = v > 0.0f ? v : 0.0f;
                    Cp[(int64_t)gm * N + gn] = v;
                }
            }
        }
}

}  // namespace

extern "C" int stm_gemm_bias_f32(const float* A, const float* Bmat, const float* bias, float* Cmat, int M, int N, int K,
                                 int batch, int64_t b_bstride, int64_t c_bstride, int relu, stm_stream_t stream)
{
    STM_REQUIRE(A && Bmat && Cmat, STM_ENULL, "stm_gemm_bias_f32: A/B/C must be non-NULL");
    STM_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0, STM_EINVAL, "stm_gemm_bias_f32: bad sizes M=%d N=%d K=%d batch=%d",
                M, N, K, batch);
    STM_REQUIRE(batch <= 65535, STM_EINVAL, "stm_gemm_bias_f32: batch %d > 65535", batch);
    STM_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)Bmat % 16 == 0) && (b_bstride % 4 == 0 || N % 4 != 0),
                STM_EINVAL, "stm_gemm_bias_f32: A and B must be 16-byte aligned");
    // tile choice: big tiles when the grid still fills 256 CUs, otherwise 64x64 for parallelism
    int64_t big = (int64_t)stm_cdiv(M, 128) * stm_cdiv(N, 128) * batch;
    const char* force = getenv("STM_GEMM_TILE");
    int tile = force ? atoi(force) : (big >= 512 && M >= 128 ? 128 : 64);
    if (tile == 128) {
        int nt = stm_cdiv(N, 128);
        dim3 grid(nt * stm_cdiv(M, 128), batch);
        hipLaunchKernelGGL((gemm_bias_f32_kernel<128, 128>), grid, dim3(256), 0, stm_hs(stream), A, Bmat, bias, Cmat, M,
                           N, K, b_bstride, c_bstride, relu, nt);
    } else {
        int nt = stm_cdiv(N, 64);
        dim3 grid(nt * stm_cdiv(M, 64), batch);
        hipLaunchKernelGGL((gemm_bias_f32_kernel<64, 64>), grid, dim3(256), 0, stm_hs(stream), A, Bmat, bias, Cmat, M, N,
                           K, b_bstride, c_bstride, relu, nt);
    }
    STM_CHECK_LAUNCH("gemm_bias_f32_kernel");
    return STM_OK;
}

extern "C" size_t stm_deform_conv_workspace_bytes(const stm_deform_geom* g)
{
    if (!g) return 0;
    return (size_t)g->B * g->C * g->kh * g->kw * g->Ho * g->Wo * sizeof(float);
}

extern "C" int stm_deform_conv_fwd_f32(const float* x, const float* offset, int64_t off_bstride, const float* mask,
                                       int64_t mask_bstride, int mask_is_logit, const float* weight, const float* bias,
                                       float* y, int O, int relu, const stm_deform_geom* g, void* workspace,
                                       size_t workspace_bytes, stm_stream_t stream)
{
    STM_REQUIRE(g, STM_ENULL, "stm_deform_conv_fwd_f32: geometry is NULL");
    STM_REQUIRE(weight && y, STM_ENULL, "stm_deform_conv_fwd_f32: weight/y must be non-NULL");
    STM_REQUIRE(O > 0, STM_EINVAL, "stm_deform_conv_fwd_f32: O=%d", O);
    size_t need = stm_deform_conv_workspace_bytes(g);
    STM_REQUIRE(workspace && workspace_bytes >= need, STM_EWORKSPACE,
                "stm_deform_conv_fwd_f32: workspace %zu bytes < required %zu", workspace_bytes, need);
    float* cols = static_cast<float*>(workspace);
    int rc = stm_deform_im2col_f32(x, offset, off_bstride, mask, mask_bstride, mask_is_logit, cols, g, 0, stream);
    if (rc) return rc;
    const int CK = g->C * g->kh * g->kw, HWo = g->Ho * g->Wo;
    return stm_gemm_bias_f32(weight, cols, bias, y, O, HWo, CK, g->B, (int64_t)CK * HWo, (int64_t)O * HWo, relu, stream);
}
