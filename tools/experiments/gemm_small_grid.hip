// EXPERIMENT, NOT IN THE PRODUCT BUILD (profiles/round5_experiments.md section 14): built, tested against fp64 (8 shapes x epilogue kinds, row
// independence, run-to-run bits), measured, removed.  Isolated 7.58 ms per step for the 314 text-side launches against 8.15 ms on the
// 128 x 128 ring kernel; inside the two-stream step 19.5 against 16.9 ms and the step +2.2 ms (72.9 against 70.7 ms, alternating): the
// text stream's kernels wait for CUs the persistent grids of the other streams hold, and 192-576 short workgroups take longer to find
// them than 12-48.  To build it again: add this file to csrc/build.sh and call egv_gemm6_launch from egv_gemm before egv_gemm2_launch
// (commit 5a325e7 has the wiring, the switches EGV_GEMM_SMALL_M / EGV_GEMM_SMALL_SPLITK and the tests).
//
// Small-grid bf16 NT GEMM for gfx950: C[M,N] = epilogue(A[M,K] B[N,K]^T) for the Linears over the TEXT rows of a step (RobertaLayer's
// query / key / value, attention outputs, intermediate / output dense and their data gradients: roberta.py:226-236, 338-349, 380-420;
// M = B * 32 = 256 ... 768 rows).  As 128 x 128 tiles these are 12-48 workgroups that each walk K alone through 24-96 barrier-separated
// steps (19-32 us per launch on an idle chip, 29-75 us inside the step: profiles/round5_experiments.md section 13) -- latency, not work.
// Here a launch is (M / 64) x (N / 64) [x 4 K-slices when K >= 2048] workgroups of four independent waves:
//   * a wave owns 32 x 32 outputs and loads BOTH operands straight into MFMA fragment layout (16 bytes per lane, rows of 128 bytes
//     per 64-wide K chunk; no LDS, no barrier in the K loop), three chunks in flight; the loads are agent-scope (sc1): operands
//     every CU reads out of freshly recycled memory were seen to return a stale line through the plain path (experiment log 11);
//   * the weight fragment is the A operand of the MFMA, so a lane ends up with four CONSECUTIVE output columns of one row and the
//     epilogue is gemm_epilogue4 of the generic kernel (bias, saved pre-activation, activation, gate, two residuals, activation
//     derivative -- every form egv_gemm accepts);
//   * K >= 2048 (the 3072-wide MLP products): four K-slices per tile, summed inside the launch in slice order by the tile's last
//     arriver (cdna_hip_programming.md, hand-off in its counter form: sc1 slabs, every wave drains, one relaxed agent-scope ticket;
//     the reducer acquires once and reads with sc1 loads).  The slice count depends on K alone: a row's bits do not depend on M.
// No workgroup needs more than 1 KiB of LDS or 128 registers: the grid fits beside the persistent grids of the other streams.
#include "egv_gemm.h"
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

namespace egv {
namespace {
constexpr int S6_AUX = 16;                      // sc1 (agent scope) on the slab traffic
#ifndef S6_LOAD_AUX
#define S6_LOAD_AUX 16                          // ... and on the operand loads (tools/variant_build.sh: 0 = plain loads, experiment)
#endif
constexpr int S6_SLAB = 64 * 64;                // floats per (tile, slice) partial

__device__ __forceinline__ f32x4_t s6_mfma(u32x4_t a, u32x4_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

template <int NSPLIT>
__global__ __launch_bounds__(256) void gemm_small_kernel(const GemmArgs g, float* __restrict__ slabs, int* __restrict__ cnt, int kper,
                                                         unsigned int a_bytes, unsigned int b_bytes) {
    __shared__ int sh_last;
    const int tid = threadIdx.x, lane = tid & 63, w = wave_id();
    const int fr = lane & 15, fg = lane >> 4;
    const int tile = blockIdx.x;
    const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
    const int m0 = tm * 64 + (w >> 1) * 32, n0 = tn * 64 + (w & 1) * 32;
    const int split = NSPLIT > 1 ? (int)blockIdx.y : 0;
    const int KT = kper >> 6;                                      // 64-wide chunks of this slice
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), 0, (int)b_bytes, 0x00020000);
    // lane (fr, fg): row fr of a 16-row fragment, 16-byte pieces fg (first MFMA of a chunk) and fg + 4 (second) of the row's 128 bytes
    unsigned int xo[2], wo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        xo[i] = (unsigned int)(min(m0 + i * 16 + fr, g.M - 1) * g.lda + fg * 8) * 2u;      // rows past M: the last row again (never stored)
        wo[i] = (unsigned int)((n0 + i * 16 + fr) * g.ldb + fg * 8) * 2u;
    }
    const unsigned int kb0 = (unsigned int)(split * kper) * 2u;

    u32x4_t xa[3][2][2], wb[3][2][2];                              // [ring slot][fragment][K half]
    auto load = [&](auto slot, int kc) {
        constexpr int S = decltype(slot)::value;
        const unsigned int kb = kb0 + (unsigned int)kc * 128u;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                xa[S][i][h] = __builtin_amdgcn_raw_buffer_load_b128(rA, xo[i] + h * 64u, kb, S6_LOAD_AUX);
                wb[S][i][h] = __builtin_amdgcn_raw_buffer_load_b128(rB, wo[i] + h * 64u, kb, S6_LOAD_AUX);
            }
    };
    f32x4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](auto slot) {
        constexpr int S = decltype(slot)::value;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = s6_mfma(wb[S][j][h], xa[S][i][h], acc[i][j]);   // D[n = fg*4 + r][m = fr]
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    // chunk c lives in ring slot c % 3; past the last chunk the last one is fetched again (no branch around a load)
    load(I0{}, 0);
    load(I1{}, min(1, KT - 1));
    for (int kc = 0;;) {
        load(I2{}, min(kc + 2, KT - 1));
        compute(I0{});
        if (++kc >= KT) break;
        load(I0{}, min(kc + 2, KT - 1));
        compute(I1{});
        if (++kc >= KT) break;
        load(I1{}, min(kc + 2, KT - 1));
        compute(I2{});
        if (++kc >= KT) break;
    }

    if constexpr (NSPLIT > 1) {
        // ---- publish this slice's partial (fragment q of thread tid at float4 index q * 256 + tid), draw a ticket
        float* tile_slabs = slabs + (size_t)tile * NSPLIT * S6_SLAB;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tile_slabs, 0, NSPLIT * S6_SLAB * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[i][j]), rs,
                                                       (unsigned int)(((i * 2 + j) * 256 + tid) * 16), (unsigned int)(split * S6_SLAB * 4), S6_AUX);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every storing wave drains its write-through stores
        __syncthreads();
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == NSPLIT - 1;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next launch on this stream
            }
            sh_last = last;
        }
        __syncthreads();
        if (!sh_last) return;
        // ---- last arriver: every slice's partial (its own included) in slice order
        u32x4_t p[4][NSPLIT];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int z = 0; z < NSPLIT; ++z)
                p[q][z] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned int)((q * 256 + tid) * 16), (unsigned int)(z * S6_SLAB * 4), S6_AUX);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4_t s = __builtin_bit_cast(f32x4_t, p[q][0]);
#pragma unroll
            for (int z = 1; z < NSPLIT; ++z) s += __builtin_bit_cast(f32x4_t, p[q][z]);
            acc[q >> 1][q & 1] = s;
        }
    }

    const float gate = g.e.gate ? *g.e.gate : 1.0f;
    bf16_t* C = reinterpret_cast<bf16_t*>(g.C);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + i * 16 + fr;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            gemm_epilogue4<bf16_t, bf16_t>(g, C, m, n0 + j * 16 + fg * 4, v, gate);
        }
    }
}

// slabs and tickets of the K-sliced launches: one pool per (device, stream), grown on demand (launches of one stream are ordered, so a
// pool is never shared by two launches in flight); tickets are zeroed when the pool is made and put back by each tile's reducer
struct S6Pool { float* slabs = nullptr; size_t slab_floats = 0; int* cnt = nullptr; size_t ncnt = 0; };
std::mutex g_s6_mu;
std::map<std::pair<int, hipStream_t>, S6Pool> g_s6_pools;

bool s6_pool(hipStream_t st, size_t slab_floats, size_t ntile, S6Pool& out) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_s6_mu);
    S6Pool& p = g_s6_pools[{dev, st}];
    if (p.slab_floats < slab_floats) {
        // the old block may still be read by a launch in flight on this stream: it is left allocated (pools grow a few times at most)
        const size_t want = slab_floats < ((size_t)4 << 20) ? ((size_t)4 << 20) : slab_floats;
        float* q = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&q), want * sizeof(float)) != hipSuccess) return false;
        p.slabs = q; p.slab_floats = want;
    }
    if (p.ncnt < ntile) {
        const size_t want = ntile < 4096 ? 4096 : ntile;
        int* q = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&q), want * sizeof(int)) != hipSuccess) return false;
        if (hipMemset(q, 0, want * sizeof(int)) != hipSuccess) return false;
        p.cnt = q; p.ncnt = want;
    }
    out = p;
    return true;
}
}  // namespace
}  // namespace egv
using namespace egv;

// non-zero if the call was enqueued here.  egv_gemm (egv_gemm.hip) asks before the DMA-staged kernels
int egv_gemm6_launch(const GemmArgs& g0, int a_trans, int b_trans, int out_f32, hipStream_t st) {
    static const int max_m = egv_cfg_int("EGV_GEMM_SMALL_M", 768);
    if (max_m <= 0 || a_trans || b_trans || out_f32) return 0;
    if (g0.M > max_m || g0.M < 17 || (g0.N % 64) || (g0.K % 64) || g0.N > 8192 || !g0.a_vec_ok || !g0.b_vec_ok) return 0;
    if ((long long)g0.M * g0.lda * 2 >= (1LL << 31) || (long long)g0.N * g0.ldb * 2 >= (1LL << 31)) return 0;   // 32-bit buffer range
    GemmArgs g = g0;
    g.tiles_m = (g.M + 63) / 64;
    g.tiles_n = g.N / 64;
    const int ntile = g.tiles_m * g.tiles_n;
    const unsigned int a_bytes = (unsigned int)((long long)g.M * g.lda * 2), b_bytes = (unsigned int)((long long)g.N * g.ldb * 2);
    static const int split_k = egv_cfg_int("EGV_GEMM_SMALL_SPLITK", 2048);
    const int nsplit = (g.K >= split_k && (g.K % 256) == 0) ? 4 : 1;     // a function of K alone: a row's bits do not depend on M
    if (nsplit == 1) {
        hipLaunchKernelGGL((gemm_small_kernel<1>), dim3(ntile), dim3(256), 0, st, g, (float*)nullptr, (int*)nullptr, g.K, a_bytes, b_bytes);
        return 1;
    }
    S6Pool p;
    if (!s6_pool(st, (size_t)ntile * nsplit * S6_SLAB, (size_t)ntile, p)) return 0;
    hipLaunchKernelGGL((gemm_small_kernel<4>), dim3(ntile, nsplit), dim3(256), 0, st, g, p.slabs, p.cnt, g.K / nsplit, a_bytes, b_bytes);
    return 1;
}
