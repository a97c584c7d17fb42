// Ping-pong MFMA weight-gradient GEMM for gfx950 (bf16 operands, fp32 accumulate and output):
//
//   dW[N, K] (fp32 slab per reduction split) = dY[M, N]^T X[M, K]      (+ fused db = sum_m dY)
//
// for the large Linear layers of the EgoVLPv2 hot path (M = B*S = 25 096 tokens is the REDUCTION; autograd of
// video_transformer.py:53,56,120,152 and of the patch-embedding conv :82).  Both operands are reduction-major in memory (a row of
// dY / X is one token), so a K-tile of 64 tokens is DMA-staged AS STORED (buffer_load ... lds; tokens past M read as zero through
// the buffer descriptor) and the MFMA fragments -- 8 consecutive tokens of one column -- are gathered with the gfx950
// transposing LDS read ds_read_b64_tr_b16, as in gemm_wgrad_ring_kernel (egv_gemm2.hip: same 16-byte-chunk swizzle
// c ^ f(token), same fragment addressing).  What changes is the schedule, which is the one of the persistent forward kernel
// (egv_gemm3.hip): 256 x 256 output tile per workgroup, 8 waves = 2 x 4 with 128 x 64 wave tiles, 4 phases per K-tile of
// {tr-read one sub-tile ; stage one 16 KB unit ; counted vmcnt ; barrier ; 16 MFMAs ; barrier}, the two wave rows one barrier
// apart, units staged 5 phases ahead of their read.  One workgroup = one (tile, reduction split) item; the launcher splits the
// reduction so that the items fill the CUs once.  The 256 x 256 tile halves the operand traffic per flop of the 256 x 128 ring
// kernel (the in-step PMC pass shows that kernel at 1.9x its algorithmic bytes and 20 % MFMA-busy).
//
// Units of a K-tile (64 tokens x 128 columns = 16 KB each): U0 = dY columns read in phase 0 (sub-tile 0 of both wave rows),
// U1 = X columns of phase 0 (first 32 of every wave column), U2 = X columns of phase 1, U3 = dY columns of phase 2.
// Requirements (launcher): N % 256 == 0, K % 256 == 0, split length % 64 == 0, 16-byte aligned rows, byte offsets < 2^31.
#include "egv_wgrad_core.h"

namespace egv {

__global__ __launch_bounds__(512) void gemm_wgrad_pp_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = wave_id();
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;

    // item = (tile, split): consecutive workgroup ids of one XCD (id % 8) share the output tile's operand columns
    const int ntile = g.tiles_m * g.tiles_n;
    const int nz = (g.K + g.k_per_split - 1) / g.k_per_split;
    const int nitem = ntile * nz;
    const int item = xcd_remap(blockIdx.x, nitem);
    const int tile = item % ntile, split = item / ntile;
    const int m0 = (tile / g.tiles_n) * 256, n0 = (tile % g.tiles_n) * 256;      // output rows (dY columns) / columns (X columns)
    const int kbeg = split * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int KT = (kend - kbeg + 63) >> 6;

    f32x4_t acc[8][4];
    f32x4_t bacc[2];                                               // column sums of dY (bias gradient): rows w4_row(wr, wc, s, 0) + fr of the tile
    const bool want_colsum = (g.colsum != nullptr) && (n0 == 0);
    w4_mainloop(g.A, g.B, g.lda, g.ldb, g.K, m0, n0, kbeg, KT, want_colsum, smem, acc, bacc);

    // ---- bias gradient partials: [split][N] (rows of dW); every register and token group of a lane holds the same sum
    if (want_colsum && fg == 0) {
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            const int row = m0 + w4_row(wr, wc, sb, 0) + fr;
            if (row < g.M) g.colsum[(size_t)split * g.M + row] = bacc[sb][0];
        }
    }
    // ---- epilogue: fp32 slab [N, K] of this split; lane (fr, fg) of fragment (s, i, t, j'): row .. + fr, 4 consecutive columns
    float* C = reinterpret_cast<float*>(g.C) + (size_t)split * g.slab_stride;
    const float sc = g.e.scale * (g.e.gate ? *g.e.gate : 1.0f);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m0 + w4_row(wr, wc, s, i) + fr;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    const int col = n0 + w4_col(wc, t, jp) + fg * 4;
                    f32x4_t v = acc[s * 4 + i][t * 2 + jp];
                    v[0] *= sc; v[1] *= sc; v[2] *= sc; v[3] *= sc;
                    if (row < g.M && col < g.N) *reinterpret_cast<f32x4_t*>(C + (size_t)row * g.ldc + col) = v;
                }
        }
}

}  // namespace egv
using namespace egv;

// returns 1 if the ping-pong weight-gradient kernel took the call (g as prepared by egv_gemm_wgrad: M/N = output rows / columns,
// K = reduction, k_per_split set, C = slab base or dW)
int egv_gemm4_launch(const GemmArgs& g, int nz, hipStream_t st) {
    if ((g.M % 256) || (g.N % 256) || (g.k_per_split % 64) || !g.a_vec_ok || !g.b_vec_ok || (g.ldc % 4)) return 0;
    if ((long long)g.K * g.lda * 2 >= (1LL << 31) || (long long)g.K * g.ldb * 2 >= (1LL << 31)) return 0;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wgrad_pp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS);
        attr = true;
    }
    GemmArgs a = g;
    a.tiles_m = g.M / 256;
    a.tiles_n = g.N / 256;
    if ((g.K + g.k_per_split - 1) / g.k_per_split != nz) return 0;
    hipLaunchKernelGGL(gemm_wgrad_pp_kernel, dim3(a.tiles_m * a.tiles_n * nz), dim3(512), W4_LDS, st, a);
    return 1;
}
