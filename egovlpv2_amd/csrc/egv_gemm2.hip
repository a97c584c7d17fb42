// 256-row MFMA ring GEMMs for gfx950 (bf16 storage, fp32 accumulate): the Linear layers of the EgoVLPv2 hot path whose shape or
// epilogue the persistent ping-pong kernel (egv_gemm3.hip) does not take -- residual / GELU' / gated epilogues, the text-side
// grids -- and every weight gradient (M = B*S = 25 096 tokens).
//
//   * gemm_ring_kernel (NT, forward + dgrad): 512-thread workgroups, 256x128 (or 128x128) tiles, K step 32, NS-stage LDS ring filled
//     by global_load_lds_dwordx4 with NS-1 K-tiles in flight, counted s_waitcnt vmcnt + one raw s_barrier per K step, XOR swizzle
//     applied on the DMA's per-lane source address and on the fragment read, epilogue through a per-wave LDS transpose;
//   * gemm_wgrad_ring_kernel (TN, dW = dY^T X): both operands are reduction-major in memory; K-tiles are DMA-staged as stored
//     (buffer_load ... lds, reduction tail zero through the descriptor) and the fragments gathered with ds_read_b64_tr_b16;
//     split over the reduction into fp32 slabs, bias gradient from the fragments in registers.
// Ragged edges: rows beyond M/N are clamped on load (their products are never stored); NT operands require K % 64 == 0 (true
// for every Linear of the model), otherwise the caller falls back to the generic 128x128 kernel (egv_gemm.hip).
#include "egv_gemm.h"
#include <cstdlib>

namespace egv {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int WGM_, int WGN_, int MI_, int NI_>
struct Cfg {
    static constexpr int WGM = WGM_, WGN = WGN_, MI = MI_, NI = NI_;
    static constexpr int BM = WGM * MI * 16, BN = WGN * NI * 16;
    static constexpr int STAGE = (BM + BN) * 128;      // bytes per LDS stage (rows of 64 bf16)
};
using CfgB = Cfg<4, 2, 4, 4>;   // 256 x 128
using CfgC = Cfg<4, 2, 2, 4>;   // 128 x 128 (3 workgroups per CU with a 3-stage ring: fills the wave-quantisation tail of N=768 GEMMs)

// ------------------------------------------------------------------------------------------------
// NT ring kernel: K step 32 (64-byte LDS rows), NS-stage LDS ring filled by global_load_lds with NS-1 K-tiles in flight,
// counted s_waitcnt vmcnt (never 0 in steady state) + one raw s_barrier per K step (cdna_hip_programming.md T3+T4).
// LDS image per operand tile: rows of 64 B = 4 chunks of 16 B; chunk c of row r is stored at chunk c ^ (((r >> 2) & 1) * 3),
// which makes the 16-lane ds_read_b128 service groups of gfx950 conflict free for a 64-byte pitch.
// ------------------------------------------------------------------------------------------------
template <int ROWS>
__device__ __forceinline__ void ring_dma(const bf16_t* base, int rows_total, int ld, int row0, int k0, unsigned char* s, int wave,
                                         int lane) {
    const int rl = lane >> 2;                                  // row inside the 16-row piece
    const int c = (lane & 3) ^ (((rl >> 2) & 1) * 3);
#pragma unroll
    for (int p = wave; p < ROWS / 16; p += 8) {
        const int gr = min(row0 + p * 16 + rl, rows_total - 1);
        const bf16_t* src = base + (size_t)gr * ld + k0 + c * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(s + p * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ unsigned int pack2f(float a, float b) { return (unsigned int)f2bf(a) | ((unsigned int)f2bf(b) << 16); }

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename CFG, int NS>
__global__ __launch_bounds__(512) void gemm_ring_kernel(const GemmArgs g) {
    constexpr int BM = CFG::BM, BN = CFG::BN, MI = CFG::MI, NI = CFG::NI;
    constexpr int STG = (BM + BN) * 64;                        // bytes per stage
    constexpr int P = (BM + BN) / 128;                         // DMA instructions per wave per K-tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id();
    const int wm = wave / CFG::WGN, wn = wave % CFG::WGN;
    const int fr = lane & 15, fg = lane >> 4;

    const int ntile = g.tiles_m * g.tiles_n;
    const int t = xcd_remap(blockIdx.x, ntile);
    const int tm = t / g.tiles_n, tn = t % g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A);
    const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B);

    f32x4_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nt = g.K / 32;
    auto issue = [&](int kt, int slot) {
        unsigned char* sA = smem + slot * STG;
        ring_dma<BM>(A, g.M, g.lda, m0, kt * 32, sA, wave, lane);
        ring_dma<BN>(B, g.N, g.ldb, n0, kt * 32, sA + BM * 64, wave, lane);
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nt) issue(s, s);

    const int frag_off = fr * 64 + ((fg ^ (((fr >> 2) & 1) * 3)) * 16);
    int slot = 0, islot = NS - 1;
    for (int kt = 0; kt < nt; ++kt) {
        const int rem = nt - 1 - kt;
        if (rem >= NS - 2) wait_vmcnt<P*(NS - 2)>();
        else if (NS > 3 && rem == 1) wait_vmcnt<P>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + NS - 1 < nt) issue(kt + NS - 1, islot);
        const unsigned char* sA = smem + slot * STG + frag_off;
        const unsigned char* sB = sA + BM * 64;
        bf16x8_t b[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(sB + (wn * NI * 16 + j * 16) * 64);
        bf16x8_t a[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(sA + (wm * MI * 16 + i * 16) * 64);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        slot = (slot + 1 == NS) ? 0 : slot + 1;
        islot = (islot + 1 == NS) ? 0 : islot + 1;
    }

    // ---------------- epilogue through LDS ----------------
    // Each wave transposes its accumulators 16 rows at a time through a private fp32 LDS slab so that every lane owns 8
    // CONSECUTIVE output columns of one row: bias is two float4 loads per lane for the whole tile, and residual / aux / pre /
    // C accesses are 16-byte vectors forming 128-byte row segments (8 lanes per row) instead of 8-byte scattered ones.
    static_assert(NI == 4, "wave tile must be 64 columns wide");
    constexpr int EP = 68;                                     // floats per LDS slab row (64 + 4 pad)
    __syncthreads();                                           // every wave is done with the operand stages
    float* slab = reinterpret_cast<float*>(smem) + wave * 16 * EP;
    bf16_t* C = reinterpret_cast<bf16_t*>(g.C);
    const GemmEpi& e = g.e;
    const float gate = e.gate ? *e.gate : 1.0f;
    const bf16_t* R1 = reinterpret_cast<const bf16_t*>(e.res1);
    const bf16_t* R2 = reinterpret_cast<const bf16_t*>(e.res2);
    const bf16_t* AUX = reinterpret_cast<const bf16_t*>(e.aux);
    bf16_t* PRE = reinterpret_cast<bf16_t*>(e.pre);
    const int er = lane >> 3, ec = (lane & 7) * 8;             // read-back: row er (+8), columns ec .. ec+7 of the wave tile
    const int ncol = n0 + wn * 64 + ec;
    const bool col_ok = ncol < g.N;                            // N % 8 == 0 is a launch precondition
    float bias8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bias8[k] = 0.f;
    if (e.bias && col_ok) {
        const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(e.bias + ncol);
        const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(e.bias + ncol + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { bias8[k] = b0[k]; bias8[4 + k] = b1[k]; }
    }
#pragma clang loop unroll(full)
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) *reinterpret_cast<f32x4_t*>(slab + fr * EP + ni * 16 + fg * 4) = acc[mi][ni];
        // same-wave LDS write -> read: program order + lgkmcnt is enough, no barrier needed
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int row = half * 8 + er;
            const int m = m0 + wm * MI * 16 + mi * 16 + row;
            const f32x4_t x0 = *reinterpret_cast<const f32x4_t*>(slab + row * EP + ec);
            const f32x4_t x1 = *reinterpret_cast<const f32x4_t*>(slab + row * EP + ec + 4);
            if (m < g.M && col_ok) {
                float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                const size_t ro = (size_t)m * e.ldr + ncol;
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = v[k] * e.scale + bias8[k];
                if (PRE && e.act == 4) {                           // EGV_ACT_GELU_D: save gelu'(x), return gelu(x)
                    float dv[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) gelu_pair_fast_f(v[k], v[k], dv[k]);
                    u32x4_t o = {pack2f(dv[0], dv[1]), pack2f(dv[2], dv[3]), pack2f(dv[4], dv[5]), pack2f(dv[6], dv[7])};
                    *reinterpret_cast<u32x4_t*>(PRE + ro) = o;
                } else if (PRE) {
                    u32x4_t o = {pack2f(v[0], v[1]), pack2f(v[2], v[3]), pack2f(v[4], v[5]), pack2f(v[6], v[7])};
                    *reinterpret_cast<u32x4_t*>(PRE + ro) = o;
                }
                if (PRE && e.act == 4) {
                } else if (e.act == 1 || e.act == 4) {             // GELU: bf16-mode fast form (egv_common.h)
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = gelu_fast_f(v[k]);
                } else if (e.act) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = apply_act(v[k], e.act);
                }
                if (e.gate) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) { v[k] *= gate; asm volatile("" : "+v"(v[k])); }   // not contracted with the residual add (bit-equal across the GEMM kernels)
                }
                if (R1) {
                    const u32x4_t r = *reinterpret_cast<const u32x4_t*>(R1 + ro);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v[2 * k] += __uint_as_float(r[k] << 16); v[2 * k + 1] += __uint_as_float(r[k] & 0xffff0000u); }
                }
                if (R2) {
                    const u32x4_t r = *reinterpret_cast<const u32x4_t*>(R2 + ro);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v[2 * k] += __uint_as_float(r[k] << 16); v[2 * k + 1] += __uint_as_float(r[k] & 0xffff0000u); }
                }
                if (e.dact == 1) {
                    const u32x4_t r = *reinterpret_cast<const u32x4_t*>(AUX + ro);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[2 * k] *= dgelu_fast_f(__uint_as_float(r[k] << 16));
                        v[2 * k + 1] *= dgelu_fast_f(__uint_as_float(r[k] & 0xffff0000u));
                    }
                } else if (e.dact) {
                    const u32x4_t r = *reinterpret_cast<const u32x4_t*>(AUX + ro);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[2 * k] *= apply_dact(__uint_as_float(r[k] << 16), e.dact);
                        v[2 * k + 1] *= apply_dact(__uint_as_float(r[k] & 0xffff0000u), e.dact);
                    }
                }
                u32x4_t o = {pack2f(v[0], v[1]), pack2f(v[2], v[3]), pack2f(v[4], v[5]), pack2f(v[6], v[7])};
                *reinterpret_cast<u32x4_t*>(C + (size_t)m * g.ldc + ncol) = o;
            }
        }
    }
}

#ifdef EGV_INSTRUMENT
// instrumentation build only (build.sh EGV_INSTRUMENT=1; tools/gemm_pp_stamps.py): buffer for the per-K-tile timestamps of the
// STAMPS variant of gemm_pp_kernel.  Neither the symbol nor the variant exists in the release library.
float* g_timing_buf = nullptr;
extern "C" int egv_debug_timing(void* buf) { g_timing_buf = (float*)buf; return 0; }
#endif

template <typename CFG, int NS>
static void launch_ring(GemmArgs g, hipStream_t st) {
    g.tiles_m = (g.M + CFG::BM - 1) / CFG::BM;
    g.tiles_n = (g.N + CFG::BN - 1) / CFG::BN;
    const size_t lds = (size_t)NS * (CFG::BM + CFG::BN) * 64;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring_kernel<CFG, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_ring_kernel<CFG, NS>), dim3(g.tiles_m * g.tiles_n), dim3(512), lds, st, g);
}

// ------------------------------------------------------------------------------------------------
// wgrad ring kernel: both operands are stored [reduction, rows] (dY [M,N], X [M,K]).  The K-tile (32 reduction rows x
// BM / BN contiguous columns) is DMA-staged AS STORED (buffer_load ... lds, out-of-range reduction rows read as zero
// through the buffer descriptor) and the MFMA fragments (8 consecutive reduction elements of one column) are gathered
// with the gfx950 transposing LDS read ds_read_b64_tr_b16 (2 per fragment).  16-byte chunk c of reduction row k is
// stored at chunk c ^ f(k), f(k) = 2*((k & 3) + 4*((k >> 3) & 1)): the 32 lanes of a half-wave then cover all 64 banks
// exactly once per transposing read.  Same NS-stage ring / counted-vmcnt schedule as gemm_ring_kernel; the column sums
// of the A operand (bias gradient) are accumulated from the fragments that are already in registers.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__device__ __forceinline__ int swz_k(int k) { return 2 * ((k & 3) + 4 * ((k >> 3) & 1)); }

// COLS contiguous columns per reduction row; one 1-KiB piece = 1024 / (COLS*2) reduction rows
template <int COLS>
__device__ __forceinline__ void wring_dma(__amdgpu_buffer_rsrc_t rsrc, int ld, int col0, int k0, unsigned char* s, int wave, int lane) {
    constexpr int CH = COLS / 8;                 // 16-byte chunks per reduction row
    constexpr int RPP = 64 / CH;                 // reduction rows per piece
    constexpr int NPIECE = 32 / RPP;             // pieces per K-tile
    const int kr = lane / CH, cp = lane % CH;
#pragma unroll
    for (int p = wave; p < NPIECE; p += 8) {
        const int k = p * RPP + kr;
        const int c = cp ^ swz_k(k);
        const int voff = ((k0 + k) * ld + col0 + c * 8) * 2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(s + p * 1024), 16, voff, 0, 0, 0);
    }
}

// fragment: 8 consecutive reduction rows (g*8 .. g*8+7) of column idx0 + fr, from a [32][COLS] tile
template <int COLS>
__device__ __forceinline__ bf16x8_t wfrag(const unsigned char* s, int idx0, int fr, int fg) {
    const int c16 = (idx0 >> 3) + ((fr >> 1) & 1);
    const int f = 2 * ((fr >> 2) + 4 * (fg & 1));
    const int krow = fg * 8 + (fr >> 2);
    const unsigned char* p = s + krow * (COLS * 2) + ((c16 ^ f) * 16) + (fr & 1) * 8;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + 4 * COLS * 2));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
}

template <typename CFG, int NS>
__global__ __launch_bounds__(512) void gemm_wgrad_ring_kernel(const GemmArgs g) {
    constexpr int BM = CFG::BM, BN = CFG::BN, MI = CFG::MI, NI = CFG::NI;
    constexpr int STG = (BM + BN) * 64;                        // bytes per stage (32 reduction rows)
    constexpr int P = ((BM + BN) * 64) / (8 * 1024);           // DMA instructions per wave per K-tile
    static_assert((BM * 64) % (8 * 1024) == 0 && (BN * 64) % (8 * 1024) == 0, "pieces must divide evenly over 8 waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id();
    const int wm = wave / CFG::WGN, wn = wave % CFG::WGN;
    const int fr = lane & 15, fg = lane >> 4;

    const int ntile = g.tiles_m * g.tiles_n;
    const int t = xcd_remap(blockIdx.x, ntile);
    const int tm = t / g.tiles_n, tn = t % g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);

    // buffer descriptors over the whole operands: reduction rows >= K read as zero
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), 0, g.K * g.lda * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), 0, g.K * g.ldb * 2, 0x00020000);

    f32x4_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float bsum[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) bsum[i] = 0.f;
    const bool want_colsum = (g.colsum != nullptr) && (tn == 0) && (wn == 0);

    const int nt = (kend - kbeg + 31) / 32;
    auto issue = [&](int kt, int slot) {
        unsigned char* sA = smem + slot * STG;
        wring_dma<BM>(ra, g.lda, m0, kbeg + kt * 32, sA, wave, lane);
        wring_dma<BN>(rb, g.ldb, n0, kbeg + kt * 32, sA + BM * 64, wave, lane);
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nt) issue(s, s);

    int slot = 0, islot = NS - 1;
    for (int kt = 0; kt < nt; ++kt) {
        const int rem = nt - 1 - kt;
        if (rem >= NS - 2) wait_vmcnt<P*(NS - 2)>();
        else if (NS > 3 && rem == 1) wait_vmcnt<P>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + NS - 1 < nt) issue(kt + NS - 1, islot);
        const unsigned char* sA = smem + slot * STG;
        const unsigned char* sB = sA + BM * 64;
        bf16x8_t b[NI], a[MI];
#pragma unroll
        for (int j = 0; j < NI; ++j) b[j] = wfrag<BN>(sB, wn * NI * 16 + j * 16, fr, fg);
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = wfrag<BM>(sA, wm * MI * 16 + i * 16, fr, fg);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (want_colsum) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const u32x4_t u = __builtin_bit_cast(u32x4_t, a[i]);
#pragma unroll
                for (int d = 0; d < 4; ++d) bsum[i] += __uint_as_float(u[d] << 16) + __uint_as_float(u[d] & 0xffff0000u);
            }
        }
        slot = (slot + 1 == NS) ? 0 : slot + 1;
        islot = (islot + 1 == NS) ? 0 : islot + 1;
    }

    if (want_colsum) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            float v = bsum[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int row = m0 + wm * MI * 16 + i * 16 + fr;
            if (fg == 0 && row < g.M) g.colsum[(size_t)blockIdx.z * g.M + row] = v;
        }
    }

    float* C = reinterpret_cast<float*>(g.C) + (size_t)blockIdx.z * g.slab_stride;
    const float gate = g.e.gate ? *g.e.gate : 1.0f;
#pragma clang loop unroll(full)
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + wm * MI * 16 + mi * 16 + fr;
        if (m >= g.M) continue;
#pragma clang loop unroll(full)
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + wn * NI * 16 + ni * 16 + fg * 4;
            if (n >= g.N) continue;
            float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
            gemm_epilogue4<bf16_t, float>(g, C, m, n, v, gate);
        }
    }
}

template <typename CFG, int NS>
static void launch_wgrad_ring(GemmArgs g, int nz, hipStream_t st) {
    g.tiles_m = (g.M + CFG::BM - 1) / CFG::BM;
    g.tiles_n = (g.N + CFG::BN - 1) / CFG::BN;
    const size_t lds = (size_t)NS * (CFG::BM + CFG::BN) * 64;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wgrad_ring_kernel<CFG, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_wgrad_ring_kernel<CFG, NS>), dim3(g.tiles_m * g.tiles_n, 1, nz), dim3(512), lds, st, g);
}

}  // namespace egv
using namespace egv;

int egv_gemm3_launch(const egv::GemmArgs& g, hipStream_t st);

int egv_gemm2_launch(const GemmArgs& g, int a_trans, int b_trans, int out_f32, int nz, hipStream_t st) {
    if (!((a_trans == 0 && b_trans == 0) || (a_trans == 1 && b_trans == 1))) return 0;
    if (g.M < 128 || g.N < 64) return 0;
    if (!a_trans) {
        if ((g.K % 64) || !g.a_vec_ok || !g.b_vec_ok) return 0;       // DMA staging needs aligned, whole K steps
        if (out_f32) return 0;
        auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        if ((g.N % 8) || (g.ldc % 8) || (g.e.ldr % 8) || !al16(g.C) || !al16(g.e.res1) || !al16(g.e.res2) || !al16(g.e.pre) ||
            !al16(g.e.aux) || (g.e.bias && !al16(g.e.bias)))
            return 0;
    } else {
        if (!out_f32) return 0;                                        // wgrad writes fp32 (slabs or dW)
        if (!g.a_vec_ok || !g.b_vec_ok || (g.M % 8) || (g.N % 8)) return 0;
        if ((long long)g.K * g.lda * 2 >= (1LL << 31) || (long long)g.K * g.ldb * 2 >= (1LL << 31)) return 0;   // 32-bit buffer range
        if (g.k_per_split % 32) return 0;
        launch_wgrad_ring<CfgB, 3>(g, nz, st);
        return 1;
    }
    {
        static const int pp = egv_cfg_int("EGV_GEMM_PP", 1);     // persistent ping-pong kernel (egv_gemm3.hip) for large grids; 0 = ring kernels only
        if (pp && (long long)((g.M + 255) / 256) * ((g.N + 255) / 256) >= 64 && egv_gemm3_launch(g, st)) return 2;   // 2: persistent kernel
        const long long tb = (long long)((g.M + 255) / 256) * ((g.N + 127) / 128);
        if (tb <= 128) { launch_ring<CfgC, 6>(g, st); return 3; }   // 3: small-grid ring     // latency-bound small grids (text tokens): 128x128 tiles, 5 K-tiles in flight
        else launch_ring<CfgB, 3>(g, st);      // 256x128 tile x 3 stages = 72 KB: 2 workgroups per CU (epilogue of one overlaps the K loop of the other)
    }
    return 1;
}
