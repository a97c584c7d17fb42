// Main loop of the ping-pong MFMA weight-gradient kernels for gfx950 (bf16 operands, fp32 accumulate):
//
//   acc[256 x 256] += dY[kbeg .. kbeg + 64*KT, m0 .. m0+255]^T  X[kbeg .. , n0 .. n0+255]
//
// shared by gemm_wgrad_pp_kernel (egv_gemm4.hip: one (tile, split) item per workgroup, fp32 slabs summed by a second launch) and
// gemm_wgrad_group_kernel (egv_gemm5.hip: several weight gradients in one launch, splits summed by the last arriver).
// Both operands are reduction-major in memory (a row of dY / X is one token), so a K-tile of 64 tokens is DMA-staged AS STORED
// (buffer_load ... lds; tokens past the end read as zero through the buffer descriptor) and the MFMA fragments -- 8 consecutive
// tokens of one column -- are gathered with the gfx950 transposing LDS read ds_read_b64_tr_b16.  Schedule: 8 waves = 2 x 4 with
// 128 x 64 wave tiles, 4 phases per K-tile of {tr-read one sub-tile ; stage one 16 KB unit ; counted vmcnt ; barrier ; 16 MFMAs ;
// barrier}, the two wave rows one barrier apart, units staged 5 phases ahead of their read.
//
// Units of a K-tile (64 tokens x 128 columns = 16 KB each): U0 = dY columns read in phase 0 (sub-tile 0 of both wave rows: tile rows
// 0..127), U1 = X columns of phase 0 (sub-tile 0 of every wave column: tile columns 0..127), U2 = X columns of phase 1 (128..255),
// U3 = dY columns of phase 2 (128..255).
//
// Round 6 (W4_TWO_PHASE, default): TWO phases per K-tile instead of four.  The measured loop time fits  2 x sum_p max(load segment p,
// MFMA segment) + ~42 cycles per barrier pair  (complete 1.42 us, no fragment reads 1.26, no DMAs 1.18, no MFMAs 1.04 per unit of
// 256 x 256 x 64): with four phases the load segments carry 24 / 8 / 16 / 0 transposing reads + 2 DMA pieces each, so the first and the
// third outlast their 16-MFMA segments and every K-tile pays eight barrier pairs.  Two phases -- {A sub 0, B sub 0, B sub 1: 32 reads, 4
// pieces | 34 MFMAs} and {A sub 1: 16 reads, 4 pieces | 34 MFMAs} -- keep the same 64 fragment registers, halve the barrier pairs and
// put each load segment beside an MFMA segment of its own length.  Staging order is unchanged (U2 U3 of kt+1, then U0 U1 of kt+2), two
// units per phase; a unit's slot is restaged ONE phase after its last read, which is legal because every phase retires its LDS reads
// (s_waitcnt lgkmcnt(0), free: they were issued ahead of four DMA pieces) before its first barrier (cdna_hip_programming.md, WAR rule);
// the waits are vmcnt(8) in the first phase (A sub 1 of this K-tile, staged two phases ago) and vmcnt(6) in the second (U0 U1 of the next
// K-tile, staged two phases ago, and its U2 -- the B operand, i.e. a bf16 activation panel every tile of a row re-reads -- staged in the
// previous phase; its U3 may still be in flight).
#pragma once
#include "egv_gemm.h"
#ifndef W4_TWO_PHASE
#define W4_TWO_PHASE 1
#endif

namespace egv {

typedef __attribute__((address_space(3))) void* lptr4_t;
typedef __attribute__((ext_vector_type(4))) short w4_s16x4_t;
typedef __attribute__((ext_vector_type(8))) short w4_s16x8_t;

template <int N> __device__ __forceinline__ void w4_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int W4_UNIT = 16384;
constexpr int W4_BUF = 4 * W4_UNIT;
constexpr int W4_LDS = 2 * W4_BUF + 8192;      // + 1 KiB per wave: target of the DMAs issued past the last K-tile

__device__ __forceinline__ int w4_swz(int k) { return 2 * ((k & 3) + 4 * ((k >> 3) & 1)); }

// Which 128 columns of the 256-column tile side make up a staged unit.  W4_CONTIG (default): sub-tile s of wave row wr is tile rows
// s*128 + wr*64 .. +63 and sub-tile t of wave column wc is tile columns t*128 + wc*32 .. +31, so a unit is 128 CONSECUTIVE columns
// of the operand: every DMA lane group of 16 fetches one contiguous 256-byte token row (two whole cache lines).  W4_CONTIG=0 is
// the round-3 map (a wave's 128 x 64 tile contiguous: units gather 128-byte pieces of dY and 64-byte pieces of X, so every cache
// line of X is requested by two DMA instructions of different phases).
#ifndef W4_CONTIG
#define W4_CONTIG 1
#endif
// cache policy of the operand DMAs (build-time experiment hook: " nt", " sc1" ...; both measured slower than the default in the step)
#ifndef W4_LD_AUX
#define W4_LD_AUX ""
#endif
// (wave column wc keeps row block i ^ wc in its fragment slot i, so that slot 0 is the block whose bias gradient it owns)
__device__ __forceinline__ int w4_row(int wr, int wc, int s, int i) { return (W4_CONTIG ? s * 128 + wr * 64 : wr * 128 + s * 64) + (i ^ wc) * 16; }
__device__ __forceinline__ int w4_col(int wc, int t, int jp) { return W4_CONTIG ? t * 128 + wc * 32 + jp * 16 : wc * 64 + t * 32 + jp * 16; }

// fragment: tokens fg*8 .. fg*8+7 (of a 32-token half) of unit column idx0 + fr; unit image = [token][128 columns], 256-byte rows
__device__ __forceinline__ bf16x8_t w4_frag(const unsigned char* s, int idx0, int fr, int fg) {
    const int c16 = (idx0 >> 3) + ((fr >> 1) & 1);
    const int f = 2 * ((fr >> 2) + 4 * (fg & 1));
    const int krow = fg * 8 + (fr >> 2);
    const unsigned char* p = s + krow * 256 + ((c16 ^ f) * 16) + (fr & 1) * 8;
    const w4_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w4_s16x4_t*)(p));
    const w4_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w4_s16x4_t*)(p + 4 * 256));
    const w4_s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
}

// A = dY [Ktot tokens, lda], B = X [Ktot tokens, ldb]; the workgroup's tile is dY columns m0.., X columns n0..; tokens
// kbeg .. kbeg + 64*KT (those >= Ktot read as zero).  On return acc holds the tile (fragment (s*4+i, t*2+jp) of lane (fr, fg):
// row m0 + w4_row(wr, wc, s, i) + fr, columns n0 + w4_col(wc, t, jp) + fg*4 .. +3) and, with want_colsum (workgroup-uniform), every
// register of bacc[s] the column sum of dY (bias gradient) of row m0 + w4_row(wr, wc, s, 0) + fr: wave column wc owns the row
// block of its fragment slot 0.  The sums are formed on the matrix pipe -- one MFMA per K-half with an all-ones first operand, 4 of them per
// K-tile and wave on top of the 64 -- because the vector-ALU form of round 3 (192 shift / mask / packed-add instructions per K-tile
// in the waves of column 0 only) made those waves, and through the barriers every tile that has a bias gradient, late.
// Every DMA has landed and all waves are past the last barrier of the loop (LDS may be reused after one more barrier).
__device__ __forceinline__ void w4_mainloop(const void* A, const void* B, int lda, int ldb, int Ktot, int m0, int n0, int kbeg, int KT,
                                            bool want_colsum, unsigned char* smem, f32x4_t (&acc)[8][4], f32x4_t (&bacc)[2]) {
    const int lane = threadIdx.x & 63;
    const int wave = wave_id();
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;

    // ---- staging: unit piece P = p*8 + wave (p = 0, 1) holds tokens P*4 .. P*4+3; lane: token P*4 + (lane>>4), chunk position lane&15
    // source chunk c = position ^ f(token): unit column c*8 -> tile column.  Byte offset = token*ld*2 + column*2 (range-checked:
    // tokens >= Ktot lie past the descriptor's end and read as zero).
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(A), 0, Ktot * lda * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(B), 0, Ktot * ldb * 2, 0x00020000);
    unsigned int soff[4][2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int tok = (p * 8 + wave) * 4 + (lane >> 4);
        const int c = (lane & 15) ^ w4_swz(tok & 31);
        const int u = c * 8;                                       // unit column of this lane's chunk
        const int colA0 = W4_CONTIG ? u : (u >> 6) * 128 + (u & 63), colA1 = colA0 + (W4_CONTIG ? 128 : 64);   // A units: both wave rows' sub-tile
        const int colB0 = W4_CONTIG ? u : (u >> 5) * 64 + (u & 31), colB1 = colB0 + (W4_CONTIG ? 128 : 32);    // B units: every wave column's half
        soff[0][p] = (unsigned int)((kbeg + tok) * lda + m0 + colA0) * 2u;
        soff[3][p] = (unsigned int)((kbeg + tok) * lda + m0 + colA1) * 2u;
        soff[1][p] = (unsigned int)((kbeg + tok) * ldb + n0 + colB0) * 2u;
        soff[2][p] = (unsigned int)((kbeg + tok) * ldb + n0 + colB1) * 2u;
    }
    const unsigned int lds0 = (unsigned int)(unsigned long long)(lptr4_t)smem;
    const unsigned int dummy_lds = lds0 + 2 * W4_BUF + wave * 1024;
    int s_kt = 0;                                                  // K-tile of the unit to be issued next
    auto dma = [&](__amdgpu_buffer_rsrc_t rs, unsigned int voff, unsigned int lds_dst) {
        unsigned int keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen" W4_LD_AUX " lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(rs), "s"(lds_dst)
                     : "memory");
    };
    auto stage_unit = [&](int u) {                                 // u compile-time
        const bool live = s_kt < KT;
        const unsigned int dst = lds0 + (s_kt & 1) * W4_BUF + u * W4_UNIT + wave * 1024;
        const unsigned int koff = (unsigned int)s_kt * 64u * (unsigned int)((u == 0 || u == 3) ? lda : ldb) * 2u;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const unsigned int voff = live ? soff[u][p] + koff : 0x80000000u;      // past the end: reads nothing, lands in the dummy slab
            const unsigned int d = live ? dst + p * 8192 : dummy_lds;
            dma((u == 0 || u == 3) ? ra : rb, voff, __builtin_amdgcn_readfirstlane(d));
        }
    };

    bf16x8_t af[4][2], bf0[2][2], bf1[2][2];
    bacc[0] = bacc[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, u32x4_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});

    // prologue: U0..U3 of K-tile 0, U0 U1 of K-tile 1
    stage_unit(0); stage_unit(1); stage_unit(2); stage_unit(3);
    ++s_kt;
    stage_unit(0); stage_unit(1);
    w4_wait_vmcnt<W4_TWO_PHASE ? 6 : 8>();                        // U0 U1 (two phases: and U2) of K-tile 0 have landed
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();                    // the second wave row runs one barrier behind the first

#define W4_BS(S, KH)                                                                                                       \
    do {                                                                                                                   \
        if (want_colsum) bacc[S] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[0][KH], bacc[S], 0, 0, 0);             \
    } while (0)
// CS: this phase's A fragments are new (phases 0 and 2): their column sums go first (K-half 0) and last (K-half 1) in the segment
#define W4_MFMA(S, BF, T, ZERO, CS)                                                                                        \
    do {                                                                                                                   \
        PP_SETPRIO(1);                                                                                                     \
        if (CS) W4_BS(S, 0);                                                                                               \
        if (W4_EXP != 3)                                                                                                   \
        _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                      \
        _Pragma("unroll") for (int jp = 0; jp < 2; ++jp)                                                                   \
            acc[(S) * 4 + i][(T) * 2 + jp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                      \
                BF[jp][kh], af[i][kh], (ZERO) ? f32x4_t{0.f, 0.f, 0.f, 0.f} : acc[(S) * 4 + i][(T) * 2 + jp], 0, 0, 0);    \
        if (CS) W4_BS(S, 1);                                                                                               \
        PP_SETPRIO(0);                                                                                                     \
    } while (0)

    // W4_EXP (timing ablations, results wrong): 1 = fragments read in the first K-tile only, 2 = no DMA inside the loop, 3 = no MFMAs
#ifndef W4_EXP
#define W4_EXP 0
#endif
#define W4_RD(first) (W4_EXP != 1 || (first))
#define W4_STAGE(u) do { if (W4_EXP != 2) stage_unit(u); } while (0)
#if W4_TWO_PHASE
    for (int kt = 0; kt < KT; ++kt) {
        const unsigned char* buf = smem + (kt & 1) * W4_BUF;
        const bool first = kt == 0;
        // ---- phase A: A sub 0 (U0), B sub 0 (U1), B sub 1 (U2); stage U2 U3 of kt+1; quadrants (0,0) (0,1)
        {
            const unsigned char* pa = buf + 0 * W4_UNIT;
            const unsigned char* pb0 = buf + 1 * W4_UNIT;
            const unsigned char* pb1 = buf + 2 * W4_UNIT;
            if (W4_RD(first)) {
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    bf0[jp][0] = w4_frag(pb0, wc * 32 + jp * 16, fr, fg);
                    bf0[jp][1] = w4_frag(pb0 + 8192, wc * 32 + jp * 16, fr, fg);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    af[i][0] = w4_frag(pa, wr * 64 + (i ^ wc) * 16, fr, fg);
                    af[i][1] = w4_frag(pa + 8192, wr * 64 + (i ^ wc) * 16, fr, fg);
                }
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    bf1[jp][0] = w4_frag(pb1, wc * 32 + jp * 16, fr, fg);
                    bf1[jp][1] = w4_frag(pb1 + 8192, wc * 32 + jp * 16, fr, fg);
                }
            }
            W4_STAGE(2);
            W4_STAGE(3);
            ++s_kt;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this phase's reads are retired before its first barrier: their slots are restaged next phase
            w4_wait_vmcnt<8>();
            __builtin_amdgcn_s_barrier();
            if (first) { W4_MFMA(0, bf0, 0, kh == 0, true); W4_MFMA(0, bf1, 1, kh == 0, false); }
            else { W4_MFMA(0, bf0, 0, false, true); W4_MFMA(0, bf1, 1, false, false); }
            __builtin_amdgcn_s_barrier();
        }
        // ---- phase B: A sub 1 (U3); stage U0 U1 of kt+2; quadrants (1,1) (1,0)
        {
            const unsigned char* pa = buf + 3 * W4_UNIT;
            if (W4_RD(first))
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i][0] = w4_frag(pa, wr * 64 + (i ^ wc) * 16, fr, fg);
                af[i][1] = w4_frag(pa + 8192, wr * 64 + (i ^ wc) * 16, fr, fg);
            }
            W4_STAGE(0);
            W4_STAGE(1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            w4_wait_vmcnt<6>();
            __builtin_amdgcn_s_barrier();
            if (first) { W4_MFMA(1, bf1, 1, kh == 0, true); W4_MFMA(1, bf0, 0, kh == 0, false); }
            else { W4_MFMA(1, bf1, 1, false, true); W4_MFMA(1, bf0, 0, false, false); }
            __builtin_amdgcn_s_barrier();
        }
    }
#else
    for (int kt = 0; kt < KT; ++kt) {
        const unsigned char* buf = smem + (kt & 1) * W4_BUF;
        const bool first = kt == 0;
        // ---- phase 0: A sub 0 (U0) + B sub 0 (U1); stage U2 of kt+1; quadrant (0,0)
        {
            const unsigned char* pa = buf + 0 * W4_UNIT;
            const unsigned char* pb = buf + 1 * W4_UNIT;
            if (W4_RD(first))
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                bf0[jp][0] = w4_frag(pb, wc * 32 + jp * 16, fr, fg);
                bf0[jp][1] = w4_frag(pb + 8192, wc * 32 + jp * 16, fr, fg);
            }
            if (W4_RD(first))
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i][0] = w4_frag(pa, wr * 64 + (i ^ wc) * 16, fr, fg);
                af[i][1] = w4_frag(pa + 8192, wr * 64 + (i ^ wc) * 16, fr, fg);
            }
            W4_STAGE(2);
            w4_wait_vmcnt<8>();
            __builtin_amdgcn_s_barrier();
            if (first) W4_MFMA(0, bf0, 0, kh == 0, true); else W4_MFMA(0, bf0, 0, false, true);
            __builtin_amdgcn_s_barrier();
        }
        // ---- phase 1: B sub 1 (U2); stage U3 of kt+1; quadrant (0,1)
        {
            const unsigned char* pb = buf + 2 * W4_UNIT;
            if (W4_RD(first))
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                bf1[jp][0] = w4_frag(pb, wc * 32 + jp * 16, fr, fg);
                bf1[jp][1] = w4_frag(pb + 8192, wc * 32 + jp * 16, fr, fg);
            }
            W4_STAGE(3);
            ++s_kt;
            w4_wait_vmcnt<8>();
            __builtin_amdgcn_s_barrier();
            if (first) W4_MFMA(0, bf1, 1, kh == 0, false); else W4_MFMA(0, bf1, 1, false, false);
            __builtin_amdgcn_s_barrier();
        }
        // ---- phase 2: A sub 1 (U3); stage U0 of kt+2; quadrant (1,1)
        {
            const unsigned char* pa = buf + 3 * W4_UNIT;
            if (W4_RD(first))
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i][0] = w4_frag(pa, wr * 64 + (i ^ wc) * 16, fr, fg);
                af[i][1] = w4_frag(pa + 8192, wr * 64 + (i ^ wc) * 16, fr, fg);
            }
            W4_STAGE(0);
            w4_wait_vmcnt<8>();
            __builtin_amdgcn_s_barrier();
            if (first) W4_MFMA(1, bf1, 1, kh == 0, true); else W4_MFMA(1, bf1, 1, false, true);
            __builtin_amdgcn_s_barrier();
        }
        // ---- phase 3: no reads; stage U1 of kt+2; quadrant (1,0)
        {
            W4_STAGE(1);
            w4_wait_vmcnt<8>();
            __builtin_amdgcn_s_barrier();
            if (first) W4_MFMA(1, bf0, 0, kh == 0, false); else W4_MFMA(1, bf0, 0, false, false);
            __builtin_amdgcn_s_barrier();
        }
    }
#endif
#undef W4_MFMA
#undef W4_RD
#undef W4_STAGE
#undef W4_BS
    if (wr == 0) __builtin_amdgcn_s_barrier();
    w4_wait_vmcnt<0>();                                            // the dummy DMAs
}

}  // namespace egv
