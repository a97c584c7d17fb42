// Persistent ping-pong MFMA GEMM for gfx950 (bf16 storage, fp32 accumulate): forward and dgrad of the large Linear layers
// of the EgoVLPv2 hot path (M = B*S = 25 096 video tokens; SURVEY.md K2/K5/K7/K9: qkv, proj, fc1, fc2 and their dgrads).
//
//   C[M,N] = epilogue( A[M,K] * B[N,K]^T ),  both operands K-contiguous (NT form).
//
// Structure (cdna_hip_programming.md "256^2 8-phase template", rebuilt for short-K / output-heavy shapes):
//   * one workgroup per CU (grid = #CUs), 8 waves = 2 (M) x 4 (N), 256 x 256 output tile, wave tile 128 x 64 =
//     8 x 4 v_mfma_f32_16x16x32_bf16 accumulators (128 VGPRs), K step 64;
//   * PERSISTENT: a workgroup walks tiles t = first, first + gridDim, ...; the staging cursor runs 5 phases ahead of the
//     compute cursor and simply continues into the next tile (no prologue bubble per tile);
//   * a K-tile (256 x 64 of A, 256 x 64 of B = 64 KB, two buffers = 128 KB LDS) is staged as four 16 KB UNITS chosen by
//     WHEN they are read, not by where they sit in the tile:  U0 = A rows read in phase 0 (sub-tile 0 of both wave rows),
//     U1 = B rows read in phase 0, U2 = B rows read in phase 1, U3 = A rows read in phase 2.  Every unit is staged by all
//     8 waves with 2 global_load_lds_dwordx4 each (1 KiB per wave instruction), exactly one unit per phase, 5 phases
//     (6 for U0) before its read.  The wait at the end of every phase's load segment is a COUNTED s_waitcnt vmcnt(N) that
//     leaves the four youngest units (and whatever epilogue traffic is younger) in flight: nothing is ever drained;
//   * 4 phases per K-tile, each = {ds_read sub-tile fragments ; stage one unit ; vmcnt(N) ; barrier ; 16 MFMAs (one
//     64 x 32 quadrant of the wave tile over K = 64) ; barrier}.  The two wave rows run staggered by one barrier
//     (ping-pong): while waves 0-3 are in their MFMA segment, waves 4-7 (same SIMDs) are in their LDS/DMA segment;
//   * LDS image of a unit: 128 rows x 128 B, 16-byte chunk c of row r stored at chunk c ^ (r & 7) (XOR applied on the
//     per-lane SOURCE address of the DMA and on the fragment read): conflict-free ds_read_b128;
//   * the B rows of a wave are permuted when staged (operand row q of fragment (t, j') holds column t*32 + (q>>2)*8 + j'*4 +
//     (q&3) of the wave's 64) so that a lane's accumulators of fragments (t,0), (t,1) are 8 CONSECUTIVE output columns and the
//     four lane groups of a row cover 64 contiguous bytes: the epilogue is register-only with 16-byte loads / stores;
//   * DEFERRED, QUADRANT-WISE EPILOGUE: the finished tile's quadrant q is converted and stored in the load segment of phase q
//     of the NEXT tile's first K-tile, right before that quadrant's accumulators are re-initialised -- the epilogue's VALU
//     work and stores run beside the other wave row's MFMAs instead of stalling the matrix pipe.  Bias (and the residual)
//     are not added in the epilogue at all: they are the C operand of a tile's first MFMAs, loaded one tile (two phases)
//     ahead, so their latency is never exposed.
// Requirements (checked by the launcher, otherwise the 256x128 ring kernel of egv_gemm2.hip takes the call):
//   K % 64 == 0, K >= 192, N % 64 == 0, 16-byte aligned pointers / leading dims, byte offsets < 2^31, scale == 1.
#include "egv_gemm.h"
#include <cstdlib>

// cache policy of the epilogue's stores / one-time operand loads (aux bits of the buffer instructions: 2 = nt)
#ifndef PP_ST_AUX
#define PP_ST_AUX 0
#endif
#ifndef PP_LD_AUX
#define PP_LD_AUX 0
#endif
// spread epilogue of the plain kinds (round 6): 2 = beside the wave's own MFMAs, 1 = in the load segments; -DPP_EPI2=0 restores round 5's
// deferred quadrant-pair epilogue for A/B builds
#ifndef PP_EPI2
#define PP_EPI2 1
#endif
#ifndef PP_E2X
#define PP_E2X 0
#endif
// GELU / GELU' of the epilogues two values at a time (packed fp32 operations; -DPP_GELU_PK=0: the scalar forms, for A/B builds)
#ifdef EGV_GELU_OLD
#undef PP_GELU_PK
#define PP_GELU_PK 0
#endif
#ifndef PP_GELU_PK
#define PP_GELU_PK 1
#endif

namespace egv {

typedef __attribute__((address_space(3))) void* lptr3_t;

template <int N> __device__ __forceinline__ void pp_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int PP_UNIT = 16384;            // bytes per staged unit
constexpr int PP_BUF = 4 * PP_UNIT;       // bytes per K-tile buffer
constexpr int PP_LDS = 2 * PP_BUF + 16384; // + 1 KiB per wave of dummy DMA target (DMAs issued past the last tile) + 1 KiB per wave: the tile's 256 bias values
constexpr int PP_LDS_MX = PP_LDS + 8192;   // MX-fp8 form: + a 4-deep ring of 2 KiB block-scale slabs (8 pieces of 256 B: A blocks (wr, sub), B blocks wc)
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;

struct PPTile {
    int m0, n0;
};
__device__ __forceinline__ PPTile pp_tile(int t, int tiles_n, int bm) {
    PPTile r;
    r.m0 = (t / tiles_n) * bm;
    r.n0 = (t % tiles_n) * 256;
    return r;
}

// K-tile kinds: what else (besides the 2 DMA instructions of a phase) a wave puts on the vector-memory queue in each phase
// decides the count of the phase's s_waitcnt.
enum { PP_PLAIN = 0, PP_LAST = 1, PP_FIRST_CHAIN = 2, PP_FIRST_COLD = 3, PP_SECOND_CHAIN = 4, PP_SECOND_COLD = 5 };

// SPREAD epilogue of the plain kinds (round 6, `EPI2`): the finished tile's 16-byte output vectors -- NV = 2 (IM + IM1) per lane, in quadrant
// order -- are converted and stored a few at a time over consecutive phases around the tile boundary; segment sg carries vectors
// [sg NV / NSEG, (sg + 1) NV / NSEG).  Quadrant q (0..3 in the order the phases compute them; IM vectors each for 0, 1 and IM1 for 2, 3) is
// final after the MFMAs of phase q of the last K-tile and re-initialised by the MFMAs of phase q of the next tile's first K-tile.
//   mode 1 (7 segments): in the LOAD segments of the last K-tile's phases 1..3 and the first K-tile's phases 0..3 (before the phase's unit)
//   mode 2 (6 segments): in the MFMA segments of the last K-tile's phases 1..3 and the first K-tile's phases 0..2 -- beside the wave's own
//          MFMAs, which leave the vector ALU and the store path idle (a load segment that carries epilogue work lengthens the phase for
//          both wave rows: measured +600 cycles per phase in mode 1)
__host__ __device__ constexpr int pp_e2_nseg(int mode) { return mode == 2 ? 6 : 7; }
__host__ __device__ constexpr int pp_e2_beg(int sg, int nv, int mode) { return sg * nv / pp_e2_nseg(mode); }
__host__ __device__ constexpr int pp_e2_cnt(int sg, int nv, int mode) { return (sg + 1) * nv / pp_e2_nseg(mode) - sg * nv / pp_e2_nseg(mode); }
// segment sg may touch quadrant q only inside [q, q + 3] (mode 1) / [q, q + 2] (mode 2: segment sg >= 3 runs BESIDE the MFMAs that
// re-initialise quadrant sg - 3)
__host__ __device__ constexpr bool pp_e2_ok(int im, int im1, int mode) {
    const int nv = 2 * (im + im1), ns = pp_e2_nseg(mode);
    for (int sg = 0; sg < ns; ++sg)
        for (int v = pp_e2_beg(sg, nv, mode); v < pp_e2_beg(sg, nv, mode) + pp_e2_cnt(sg, nv, mode); ++v) {
            const int qd = v < im ? 0 : v < 2 * im ? 1 : v < 2 * im + im1 ? 2 : 3;
            if (sg < qd || sg > qd + (mode == 2 ? 2 : 3)) return false;
        }
    return pp_e2_beg(ns, nv, mode) == nv;
}
// extra vector-memory operations issued in the LOAD segment of phase p of a K-tile of kind `kind`, before the phase's unit (NSP0 / NSP1
// stores for the quadrant pair of A sub-tile 0 / 1 of a deferred epilogue, 1 bias DMA per tile); with a BULK epilogue (residual / GELU'
// operand kinds) the tile's NST stores are issued between the last K-tile and the next tile's first one
// (MX-fp8 form: + the K-tile's scale DMA, issued in phase 3 right BEFORE that phase's unit)
// e2nv > 0: spread epilogue with e2nv vectors per tile (the bias DMA then rides in phase 0 of a tile's SECOND K-tile)
__host__ __device__ constexpr int pp_extra(int kind, int p, int NSP0, int NSP1, bool bulk, bool mx = false, int e2nv = 0, int e2m = 0) {
    const int s = (mx && p == 3) ? 1 : 0;
    if (e2nv > 0) {
        if (kind == PP_SECOND_CHAIN || kind == PP_SECOND_COLD) return s + (p == 0 ? 1 : 0);
        if (e2m == 2) return s;
        if (kind == PP_LAST) return s + (p >= 1 ? pp_e2_cnt(p - 1, e2nv, e2m) : 0);
        if (kind == PP_FIRST_CHAIN) return s + pp_e2_cnt(3 + p, e2nv, e2m);
        return s;
    }
    if (kind == PP_LAST) return s + (p == 1 ? 1 : 0);
    if (kind == PP_FIRST_CHAIN && !bulk) return s + (p == 0 ? NSP0 : p == 2 ? NSP1 : 0);
    return s;
}
// ... and in the MFMA segment of phase p (after the phase's unit and its wait): the stores of the spread epilogue in mode 2
__host__ __device__ constexpr int pp_extra_m(int kind, int p, int e2nv, int e2m) {
    if (e2nv <= 0 || e2m != 2) return 0;
    if (kind == PP_LAST) return p >= 1 ? pp_e2_cnt(p - 1, e2nv, e2m) : 0;
    if (kind == PP_FIRST_CHAIN) return p <= 2 ? pp_e2_cnt(3 + p, e2nv, e2m) : 0;
    return 0;
}
__host__ __device__ constexpr int pp_prev_kind(int kind) {
    return kind == PP_FIRST_CHAIN ? PP_LAST : kind == PP_SECOND_CHAIN ? PP_FIRST_CHAIN : kind == PP_SECOND_COLD ? PP_FIRST_COLD : PP_PLAIN;
}
// operations younger than the unit staged 4 phases ago, at the wait of phase p: the DMAs of the last 4 phases + the load-segment extras of
// those phases (issued BEFORE their unit) + the MFMA-segment extras of phases p - 4 .. p - 1 (issued after their unit).  A count that is too
// small only waits longer; one that is too large is a race: PLAIN stands for "the K-tile before this one" of PLAIN and LAST, which is
// SECOND_* when K is 192 or 256 -- whose only extra (spread epilogue: the bias DMA of phase 0) is older than every window that reaches back
// into it.
__host__ __device__ constexpr int pp_nwait(int kind, int p, int NSP0, int NSP1, bool bulk, bool mx = false, int e2nv = 0, int e2m = 0) {
    int n = 8;
    for (int q = 0; q <= p; ++q) n += pp_extra(kind, q, NSP0, NSP1, bulk, mx, e2nv, e2m);
    for (int q = 0; q < p; ++q) n += pp_extra_m(kind, q, e2nv, e2m);
    if (kind != PP_FIRST_COLD) {                    // before a cold first K-tile there is only the prologue (nothing younger)
        for (int q = p + 1; q < 4; ++q) n += pp_extra(pp_prev_kind(kind), q, NSP0, NSP1, bulk, mx, e2nv, e2m);
        for (int q = p; q < 4; ++q) n += pp_extra_m(pp_prev_kind(kind), q, e2nv, e2m);
    } else if (mx && p < 3) n += 1;                 // ... except the prologue's scale DMA of K-tile 1, issued where a phase 3 would have
    if (kind == PP_FIRST_CHAIN && bulk) n += NSP0 + NSP1;  // the previous tile's bulk epilogue (all of its stores) sits between the tiles
    return n;
}

// Epilogue kinds are compile-time (register budget):
//   X1K : 0 none, 1 residual res1 added (forward of proj / fc2), 2 activation-derivative operand aux multiplied (dgrad)
//   PREK: also store the pre-activation (fc1) -- doubles the stores of a tile
//   ACTK: e.act may be non-zero
//   IM, IM1: 16-row fragments per wave row of A sub-tile 0 (read in phase 0, quadrants (0,0) (0,1)) and of A sub-tile 1 (read in phase
//         2, quadrants (1,1) (1,0)); each 4, 3 or 2: tile height (IM + IM1) x 32 rows = 256, 224, 192, 160 or 128.  4+4 = 256-row tiles;
//         3+3 = 192-row tiles (wave tile 96 x 64, 12 MFMAs per phase, 12 KB A units: round 2) for the N = 768 GEMMs, whose 98 x 3 =
//         294 tiles of 256 rows leave the second round of a 224-workgroup grid one third full; round 5: the launcher picks the height
//         whose tile count fills the last round of the walk best (M = 25 096 x N = 768 on 256 CUs: 471 tiles of 160 rows = two rounds
//         of 0.63 instead of 393 tiles of 192 rows = two rounds of 0.75; N = 2304: 1017 tiles of 224 rows = four rounds of 0.875
//         instead of 882 tiles of 256 rows = four rounds of 1).  A row's result does not depend on the height (same K order).
//   MX  : MX-fp8 operands (egv_mx.hip): a K-tile is 128 e4m3 elements -- the SAME 128-byte row pieces, staging, swizzle and
//         fragment reads (v_mfma_scale_f32_16x16x128_f8f6f4 takes k = 16 fg .. +15 in registers 0-3 and 64 + 16 fg .. +15 in
//         registers 4-7 of lane group fg: exactly the two 16-byte chunks a lane reads for the two bf16 K-halves), half as many
//         MFMAs of twice the length, twice the flops per staged byte; block scales arrive by one 256-byte LDS-DMA per wave and
//         K-tile (4-deep ring) and are read as one dword per lane and sub-tile
//   QOUT: (MX only) the output is ALSO written in MX-fp8 form -- codes e.oq[M][N] and role-0 scale bytes e.os, quantised from the
//         bf16-rounded values exactly as egv_quant_mx would quantise C: it is the A operand of the next Linear (fc1 -> fc2 forward,
//         fc2 -> fc1 data gradient), whose quantiser launch disappears.  A 32-column block of a row is the 8 columns of the four
//         lanes fr, fr + 16, fr + 32, fr + 48
template <int X1K, bool PREK, bool ACTK, bool STAMPS = false, int IM = 4, bool MX = false, bool QOUT = false, int IM1 = IM>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const GemmArgs g, int ntiles) {
    static_assert(!QOUT || MX, "quantised output only in the MX-fp8 form");
    static_assert(IM >= 2 && IM <= 4 && IM1 >= 2 && IM1 <= 4, "A sub-tiles of 4, 3 or 2 fragments");
    static_assert(!MX || (IM == 3 && IM1 == 3), "MX-fp8 form: 192-row tiles only (48-row scale blocks; the 256-row form does not fit 256 registers)");
    constexpr unsigned int ES = MX ? 1u : 2u;                      // bytes per operand element
    constexpr int IMX = IM > IM1 ? IM : IM1;
    constexpr int BM = (IM + IM1) * 32, WM = (IM + IM1) * 16;      // tile rows, rows per wave row
    constexpr int SM0 = IM * 16, SM1 = IM1 * 16;                   // rows per wave row of A sub-tile 0 / 1
#define PP_IMS(S) ((S) == 0 ? IM : IM1)                            /* fragments of sub-tile S */
#define PP_AOFF(S) ((S) == 0 ? 0 : IM)                             /* first accumulator row block of sub-tile S */
#define PP_ROFF(S) ((S) == 0 ? 0 : SM0)                            /* first row (inside a wave row) of sub-tile S */
    // stores per pair of quadrants (one pair_epilogue call): 2 (4 with the saved pre-activation) per 16-row fragment, + 2 code stores
    // and 1 scale store when the output is also emitted in MX-fp8 form
#ifndef EGV_PP_EXP
#define EGV_PP_EXP 0
#endif
    // EGV_PP_EXP (tools/pp_exp.sh, never in the product build): 1 = the epilogue stores are dropped; 3 = the residual / GELU' operand loads are dropped
    // (2, round 4: stores trickled through the next tile's K-tiles -- slower, removed)
    constexpr int NSP0 = EGV_PP_EXP == 1 ? 0 : (PREK ? 4 : 2) * IM + (QOUT ? 3 * IM : 0);
    constexpr int NSP1 = EGV_PP_EXP == 1 ? 0 : (PREK ? 4 : 2) * IM1 + (QOUT ? 3 * IM1 : 0);
#ifndef PP_PLAIN_BULK
#define PP_PLAIN_BULK 0
#endif
    // residual / GELU' operand kinds: epilogue in one piece at the tile's end.  (Round 5 built the GELU' operand kind with the deferred
    // quadrant-wise epilogue of the plain kinds, the saved pre-activation of a quadrant pair loaded at the head of the load segment that
    // converts it: SLOWER, fc2 data gradient 154-157 -> 167-171 us -- the wait for those loads sits inside a barrier-locked phase, so the
    // other wave row idles for a memory latency four times per tile; profiles/round5_experiments.md.)
    // -DPP_PLAIN_BULK=1 (experiment): the plain kinds with the one-piece epilogue too (within 1 % either way).
    constexpr bool BULK = X1K != 0 || PP_PLAIN_BULK;
    // plain kinds (bias or nothing; every tile height): the spread epilogue above
    constexpr bool EPI2 = PP_EPI2 && !BULK && !PREK && !ACTK && !MX && !QOUT && EGV_PP_EXP == 0;
    constexpr int E2M = EPI2 ? PP_EPI2 : 0;                        // 1: load segments, 2: MFMA segments
    constexpr int NV = 2 * (IM + IM1);                             // 16-byte output vectors per lane and tile
    constexpr int E2NV = EPI2 ? NV : 0;
    static_assert(!EPI2 || pp_e2_ok(IM, IM1, E2M), "spread epilogue: a vector outside its quadrant's window");
    constexpr unsigned int OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id();
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;
#ifdef PP_SETPRIO_STATIC                                            // experiment: the later-dispatched half of the workgroup at priority 1 throughout, no per-segment flips
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    const int KT = MX ? g.K >> 7 : g.K >> 6;
    const GemmEpi& e = g.e;

    // ---- tile walk: workgroup w takes tiles first, first + G, ... of an order in which the 32 workgroups of one XCD (w % 8)
    // hold 32 CONSECUTIVE tiles of every round (m-major): they share A panels through that XCD's L2.
    const int G = gridDim.x;
    const int wg = blockIdx.x;
    // (any G: XCD x = wg % 8 holds G / 8 workgroups, one more for x < G % 8 -- a grid beside a resident weight-gradient launch
    // takes exactly the CUs that launch leaves)
    const int first = (wg & 7) * (G >> 3) + min(wg & 7, G & 7) + (wg >> 3);
    const int my_tiles = first < ntiles ? (ntiles - 1 - first) / G + 1 : 0;
    if (my_tiles == 0) return;

    // ---- staging geometry of this lane: unit row rho = (p*8 + wave)*8 + (lane>>3), source chunk (lane&7) ^ (lane>>3)
    const int srow = lane >> 3;
    const unsigned int schunk = ((lane & 7) ^ srow) * 16;         // byte offset inside the 128-byte row piece of a K-tile
    unsigned int soff[4][2];                                      // byte offsets (from A / B) of this lane's source rows: [unit][piece]
    unsigned int sc_tile = 0;                                     // MX: this wave's scale piece of the staging tile (byte offset in K-tile 0)
    const unsigned int sc_kstride = MX ? (unsigned int)(wave < 4 ? ((g.M + 191) / 192) * 4 : (g.N + 63) >> 6) * 256u : 0u;   // per K-tile
    const unsigned char* sc_base = wave < 4 ? g.sa : g.sb;        // waves 0-3 fetch the A blocks (wr', sub') = (wave >> 1, wave & 1), waves 4-7 the B blocks wc' = wave - 4
    auto set_stage_tile = [&](int t) {
        const PPTile tl = pp_tile(t, g.tiles_n, BM);
        if constexpr (MX) sc_tile = (unsigned int)(wave < 4 ? tl.m0 / 48 + wave : min((tl.n0 >> 6) + wave - 4, ((g.N + 63) >> 6) - 1)) * 256u;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int rho = (p * 8 + wave) * 8 + srow;            // 0..127
            // A unit of sub-tile S (2 x IMS x 16 rows: the sub-tile of both wave rows): unit row ra -> tile row (ra / SMS) * WM + ROFF(S) +
            // ra % SMS; source chunk (lane & 7) ^ (ra & 7) (the reads' swizzle).  IMS = 4: 128 rows, two pieces of 8 rows per wave;
            // 3: 96 rows, this wave stages rows wave*12 .. +11 -- piece 0 = 8 rows, piece 1 = 4 rows (lanes 0..31 only); 2: 64 rows,
            // one piece of 8 rows per wave (piece 1 is a dummy DMA that keeps the count of vector-memory operations per unit)
            auto a_off = [&](int ims, int roff) -> unsigned int {
                const int sms = ims * 16;
                const int ra = ims == 4 ? rho : ims == 3 ? min(wave * 12 + p * 8 + srow, 95) : wave * 8 + srow;
                const int row = min(tl.m0 + (ra / sms) * WM + roff + (ra % sms), g.M - 1);
                const unsigned int sch = ((lane & 7) ^ (ra & 7)) * 16;
                return (ims == 2 && p == 1) ? 0u : __umul24((unsigned int)row, (unsigned int)g.lda * ES) + sch;   // (24-bit factors by the launcher's checks: one v_mad_u32_u24, no 64-bit register pair)
            };
            soff[0][p] = a_off(IM, 0);
            soff[3][p] = a_off(IM1, SM0);
            // B units: rho = wc'*32 + j'*16 + q -> column wc'*64 + sub*32 + (q>>2)*8 + j'*4 + (q&3)
            const int wcp = rho >> 5, jp = (rho >> 4) & 1, q = rho & 15;
            const int cb0 = min(tl.n0 + wcp * 64 + (q >> 2) * 8 + jp * 4 + (q & 3), g.N - 1);
            const int cb1 = min(tl.n0 + wcp * 64 + 32 + (q >> 2) * 8 + jp * 4 + (q & 3), g.N - 1);
            soff[1][p] = __umul24((unsigned int)cb0, (unsigned int)g.ldb * ES) + schunk;
            soff[2][p] = __umul24((unsigned int)cb1, (unsigned int)g.ldb * ES) + schunk;
        }
    };
    // staging cursor: units are issued in the fixed order U0(kt) U1(kt) U2(kt) U3(kt) U0(kt+1) ... across tiles
    int s_tile_seq = 0;                                           // which of my tiles the cursor is in
    int s_kt = 0;                                                 // K-tile inside that tile
    int s_gkt = 0;                                                // global K-tile count (buffer parity)
    set_stage_tile(first);

    // LDS-DMA issued from inline asm: the compiler then keeps no LDS-DMA bookkeeping of its own (with the builtin it drains
    // vmcnt(0) ahead of the first ds_read after every epilogue); every wait for staged data is one of the counted
    // pp_wait_vmcnt below.  M0 (the DMA's LDS base) is saved / restored inside the statement (cdna_hip_programming.md 5.7).
    const unsigned int lds0 = (unsigned int)(unsigned long long)(lptr3_t)smem;
    const unsigned int dummy_lds = lds0 + 2 * PP_BUF + wave * 1024;
    auto glds = [&](const void* base, unsigned int voff, unsigned int lds_dst) {
        unsigned int keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(base), "s"(lds_dst)
                     : "memory");
    };
    auto glds_half = [&](const void* base, unsigned int voff, unsigned int lds_dst) {      // lanes 0..31 only (512 bytes)
        unsigned int keep;
        unsigned long long ex;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_mov_b64 %1, exec\n\ts_mov_b32 exec_hi, 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                     "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(ex)
                     : "v"(voff), "s"(base), "s"(lds_dst)
                     : "memory");
    };
    auto stage_scales = [&]() {                                   // MX: the cursor's K-tile, 256 B per wave into ring slot s_gkt & 3
        if constexpr (MX) {
            const unsigned int live = s_tile_seq < my_tiles ? ~0u : 0u;
            const unsigned int dst = lds0 + 2 * PP_BUF + 16384 + (s_gkt & 3) * 2048 + wave * 256;
            const unsigned int d = dummy_lds + ((dst - dummy_lds) & live);
            unsigned int l4 = lane;
            asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(l4));   // recomputed per use: not a register held across the K loop
            const unsigned int voff = (l4 + sc_tile + (unsigned int)s_kt * sc_kstride) & live;
            unsigned int keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(voff), "s"(sc_base), "s"(__builtin_amdgcn_readfirstlane(d))
                         : "memory");
        }
    };
    auto stage_unit = [&](int u) {                                // u is a compile-time constant at every call site
        const unsigned int live = s_tile_seq < my_tiles ? ~0u : 0u;   // past my last tile: harmless DMAs into the dummy slab keep
                                                                       // the vmcnt arithmetic uniform
        const unsigned int dst = lds0 + (s_gkt & 1) * PP_BUF + u * PP_UNIT + wave * 1024;
        const void* base = (u == 0 || u == 3) ? g.A : g.B;
        const unsigned int koff = (unsigned int)s_kt * 128u;
        const int ims = u == 0 ? IM : u == 3 ? IM1 : 4;           // (B units: 128 rows)
        if (ims == 3) {                                           // 96-row A unit: 1.5 KiB per wave
            const unsigned int dsta = lds0 + (s_gkt & 1) * PP_BUF + u * PP_UNIT + wave * 1536;
            glds(base, (soff[u][0] + koff) & live, __builtin_amdgcn_readfirstlane(dummy_lds + ((dsta - dummy_lds) & live)));
            glds_half(base, (soff[u][1] + koff) & live, __builtin_amdgcn_readfirstlane(dummy_lds + ((dsta + 1024 - dummy_lds) & live)));
            return;
        }
        if (ims == 2) {                                           // 64-row A unit: 1 KiB per wave + a dummy piece (the waits count two operations per unit)
            glds(base, (soff[u][0] + koff) & live, __builtin_amdgcn_readfirstlane(dummy_lds + ((dst - dummy_lds) & live)));
            glds(base, 0u, __builtin_amdgcn_readfirstlane(dummy_lds));
            return;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const unsigned int voff = (soff[u][p] + koff) & live;
            const unsigned int d = dummy_lds + ((dst + p * 8192 - dummy_lds) & live);
            glds(base, voff, __builtin_amdgcn_readfirstlane(d));
        }
    };
    auto advance_cursor = [&]() {                                 // after U3 of a K-tile was issued
        ++s_gkt;
        if (++s_kt == KT) {
            s_kt = 0;
            ++s_tile_seq;
            if (s_tile_seq < my_tiles) set_stage_tile(first + s_tile_seq * G);
        }
    };

    // ---- fragment read offsets (bytes inside a K-tile buffer)
    const int swz0 = ((0 * 4 + fg) ^ (fr & 7)) * 16, swz1 = ((1 * 4 + fg) ^ (fr & 7)) * 16;
    const int a_base0 = (wr * SM0 + fr) * 128;                    // + i*2048 ; unit U0 (sub 0)
    const int a_base1 = (wr * SM1 + fr) * 128;                    // + i*2048 ; unit U3 (sub 1)
    const int b_base = (wc * 32 + fr) * 128;                      // + j'*2048 ; unit U1 (sub 0) / U2 (sub 1)

    // ---- epilogue operands: buffer descriptors (range = whole matrix; offset 2^31 is out of range by construction: such
    // loads return 0 and such stores are dropped, so rows / columns outside the matrix need no branches)
    auto mk = [&](const void* p, long long bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? (int)bytes : 0, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rs_c = mk(g.C, (long long)g.M * g.ldc * 2);
    const __amdgpu_buffer_rsrc_t rs_pre = mk(e.pre, (long long)g.M * e.ldr * 2);
    const __amdgpu_buffer_rsrc_t rs_x = mk((X1K == 1 || X1K == 3) ? e.res1 : e.aux, (long long)g.M * e.ldr * 2);
    const __amdgpu_buffer_rsrc_t rs_x2 = mk(X1K == 3 ? e.res2 : nullptr, (long long)g.M * e.ldr * 2);
    const __amdgpu_buffer_rsrc_t rs_bias = mk(e.bias, (long long)g.N * 4);
    const __amdgpu_buffer_rsrc_t rs_q = mk(QOUT ? e.oq : nullptr, (long long)g.M * g.N);
    const __amdgpu_buffer_rsrc_t rs_s = mk(QOUT ? e.os : nullptr, (long long)(g.N >> 7) * (((g.M + 191) / 192) * 4) * 256);
    const float gate = e.gate ? *e.gate : 1.0f;
    const bool has_gate = e.gate != nullptr;

    // quadrant q (= the phase that computes it): (s, t) = (0,0) (0,1) (1,1) (1,0)
    // element (s, i, t) of a lane: row m0 + wr*128 + s*64 + i*16 + fr, columns n0 + wc*64 + t*32 + fg*8 .. +7.
    // Byte offset = per-lane constant + a wave-uniform term (one v_add per access, nothing kept per tile).  Rows >= M fall
    // outside the descriptor's range by themselves; columns >= N (N % 64 == 0: both halves of a lane agree) are sent there.
    const int lane_col = wc * 64 + fg * 8;
    const unsigned int lane_c = (unsigned int)((wr * WM + fr) * g.ldc + lane_col) * 2u;
    const unsigned int lane_r = (unsigned int)((wr * WM + fr) * e.ldr + lane_col) * 2u;
    // stores: a lane pair (rows fr, fr^8) trades halves so that ONE store instruction writes 8 CONSECUTIVE rows x 128 contiguous bytes (the
    // wave's whole 64-column slab of a row) instead of 16 rows x 64: the CU's store path is issue-bound on lines per instruction
    // (measured: 12.4k -> 8.7k cycles for the 128 KB of a tile).  Lane (fr, fg): row (fr & 7) [+8 for the second store of
    // a pair], columns (fr >> 3)*32 + fg*8.
    // (the per-lane parts are recomputed from the lane id at every use -- `opaque_lane` hides it from hoisting -- because
    // hipcc otherwise keeps them in registers across the whole K loop and spills; a spill reload is a vector-memory operation
    // that would break the counted waits)
    auto opaque_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
    auto p_off = [&](const PPTile& tl, int s, int i, int second, bool out) -> unsigned int {
        const int l = opaque_lane();
        const int fr_ = l & 15, fg_ = l >> 4;
        const int ld = out ? g.ldc : e.ldr;
        const int pair_col = wc * 64 + (fr_ >> 3) * 32 + fg_ * 8;
        const unsigned int lane_part = (__umul24((unsigned int)(wr * WM + (fr_ & 7)), (unsigned int)ld) + (unsigned int)pair_col) * 2u;
        const unsigned int sterm = (unsigned int)((tl.m0 + PP_ROFF(s) + i * 16 + second * 8) * ld + tl.n0) * 2u;
        return (tl.n0 + pair_col < g.N) ? lane_part + sterm : OOB;
    };
    // (a, b) of this lane = (t = 0, t = 1) vectors of its row  ->  (first, second) store vectors
    auto pair_swap = [&](u32x4_t t0, u32x4_t t1, u32x4_t& first, u32x4_t& second) {
        const bool odd = fr & 8;                                  // upper half of the 16-lane row
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned int send = odd ? t0[k] : t1[k];
            const unsigned int got = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)send, 0x128, 0xF, 0xF, false);   // row_ror:8 = lane ^ 8 inside each 16-lane row
            first[k] = odd ? got : t0[k];
            second[k] = odd ? t1[k] : got;
        }
    };
    auto q_off = [&](const PPTile& tl, bool valid, int s, int i, int t, bool out) -> unsigned int {
        const int ld = out ? g.ldc : e.ldr;
        const unsigned int sterm = (unsigned int)((tl.m0 + PP_ROFF(s) + i * 16) * ld + tl.n0 + t * 32) * 2u;
        return (valid && tl.n0 + lane_col < g.N) ? (out ? lane_c : lane_r) + sterm : OOB;
    };

    // bias of the tile whose accumulators are initialised next: one LDS-DMA per wave and tile (256 fp32 values = 1 KiB) into the
    // wave's own slab, issued in the last K-tile of the previous tile; no register is held and no compiler-visible load exists
    // in the main loop (hipcc would cover a pending one with vmcnt(0) at the loop headers)
    const unsigned int bias_lds = lds0 + 2 * PP_BUF + 8192 + wave * 1024;
    const float* bias_slab = reinterpret_cast<const float*>(smem + 2 * PP_BUF + 8192 + wave * 1024);
    const bool has_bias = e.bias != nullptr;
    auto load_bias = [&](const PPTile& tb) {
        const unsigned int voff = (unsigned int)min(tb.n0 + lane * 4, g.N - 4) * 4u;
        if constexpr (EPI2)    // without a bias the slab holds zeros (written once, below) and the DMA -- kept for the counted waits -- lands in the dummy slab
            glds(has_bias ? (const void*)e.bias : g.B, has_bias ? voff : 0u, __builtin_amdgcn_readfirstlane(has_bias ? bias_lds : dummy_lds));
        else
            glds(has_bias ? (const void*)e.bias : g.B, has_bias ? voff : 0u, __builtin_amdgcn_readfirstlane(bias_lds));
    };
    if constexpr (EPI2) {
        if (!has_bias) *reinterpret_cast<f32x4_t*>(smem + 2 * PP_BUF + 8192 + wave * 1024 + lane * 16) = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    f32x4_t acc[IM + IM1][4];                                     // (AOFF(s)+i, t*2+j'); written by the first MFMAs of every tile
    bf16x8_t af[IMX][2], bf0[2][2], bf1[2][2];
    int sc_a0 = 0, sc_a1 = 0, sc_b = 0;                           // MX: block scales of the K-tile (A sub-tiles 0 / 1: byte i; B: byte t*2+j')
    auto sc_read = [&](int slot, int piece) {                     // this lane's dword of a 256-byte scale piece
        unsigned int l4 = lane;
        asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(l4));
        return *reinterpret_cast<const int*>(smem + 2 * PP_BUF + 16384 + slot * 2048 + piece * 256 + l4);
    };

    // convert + store the two quadrants (s, 0), (s, 1) of the finished tile `tl`: 8 full-line stores (16 with the pre-activation)
    u32x4_t xop2[X1K == 3 ? 2 : 1][X1K == 3 ? IMX : 1][2];     // second residual (gated i2t projection: x + gate * y + skip)
    u32x4_t xop[2][IMX][2];                                        // bulk epilogue: residual / GELU' operand vectors (s, i, t) of the tile
    auto pair_epilogue = [&](int s, const PPTile& tl) {           // s compile-time
        // the finished tile's bias from this wave's LDS slab, read where it is added (8 registers live instead of 16 across the pair: the
        // GELU + saved pre-activation kind spilled 34 registers with the hoisted form)
        auto bias_t = [&](int t, float (&b8)[8]) {
            if constexpr (X1K == 2) {                             // a data gradient has no bias (checked by the launcher): x + 0.f == x for every x the sum can be
#pragma unroll
                for (int k = 0; k < 8; ++k) b8[k] = 0.f;
            } else {
                const float* bp = bias_slab + wc * 64 + (opaque_lane() >> 4) * 8 + t * 32;
                f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(bp), b1 = *reinterpret_cast<const f32x4_t*>(bp + 4);
                if (!has_bias) b0 = b1 = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k) { b8[k] = b0[k]; b8[4 + k] = b1[k]; }
            }
        };
#pragma unroll
        for (int i = 0; i < PP_IMS(s); ++i) {
            u32x4_t f, sec;
            if (PREK) {                                           // pre-activation first, in its own pass (register budget)
                u32x4_t pr[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f32x4_t a0 = acc[PP_AOFF(s) + i][t * 2 + 0], a1 = acc[PP_AOFF(s) + i][t * 2 + 1];
                    float b8[8];
                    bias_t(t, b8);
                    float pv[8] = {a0[0] + b8[0], a0[1] + b8[1], a0[2] + b8[2], a0[3] + b8[3],
                                   a1[0] + b8[4], a1[1] + b8[5], a1[2] + b8[6], a1[3] + b8[7]};
                    if (e.act == 4) {                             // EGV_ACT_GELU_D: the saved tensor is gelu'(x) (one more FMA beside the GELU below)
#pragma unroll
                        for (int k = 0; k < 8; ++k) { float gg; gelu_pair_fast_f(pv[k], gg, pv[k]); }
                    }
                    pr[t] = u32x4_t{pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3]), pack_bf16x2(pv[4], pv[5]), pack_bf16x2(pv[6], pv[7])};
                }
                pair_swap(pr[0], pr[1], f, sec);
#if EGV_PP_EXP == 1
                asm volatile("" :: "v"(f), "v"(sec));
#else
                __builtin_amdgcn_raw_buffer_store_b128(f, rs_pre, p_off(tl, s, i, 0, false), 0, PP_ST_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(sec, rs_pre, p_off(tl, s, i, 1, false), 0, PP_ST_AUX);
#endif
            }
            u32x4_t o[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float v[8], b8[8];
                bias_t(t, b8);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[k] = acc[PP_AOFF(s) + i][t * 2 + 0][k] + b8[k];              // (sum over K) + bias: the ring kernels' order, bit for bit
                    v[4 + k] = acc[PP_AOFF(s) + i][t * 2 + 1][k] + b8[4 + k];
                }
                if (ACTK) {                                       // GELU only (the launcher sends other activations to the ring kernels)
#if PP_GELU_PK
#pragma unroll
                    for (int k = 0; k < 4; ++k) {                 // two values per packed-fp32 operation: the bits of gelu_fast_f
                        const gelu_f32x2_t r2 = gelu_fast_f2(gelu_f32x2_t{v[2 * k], v[2 * k + 1]});
                        v[2 * k] = r2[0]; v[2 * k + 1] = r2[1];
                    }
#else
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = gelu_fast_f(v[k]);
#endif
                }
                if (X1K == 3 || has_gate) {                       // (x * 1.0f is x: skipping the multiply changes no bit)
#pragma unroll
                    for (int k = 0; k < 8; ++k) { v[k] *= gate; asm volatile("" : "+v"(v[k])); }   // not contracted with the residual add (bit-equal across the GEMM kernels)
                }
                if (X1K == 1 || X1K == 3) {
                    const u32x4_t r = xop[s][i][t];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v[2 * k] += __uint_as_float(r[k] << 16); v[2 * k + 1] += __uint_as_float(r[k] & 0xffff0000u); }
                }
                if constexpr (X1K == 3) {
                    const u32x4_t r = xop2[s][i][t];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v[2 * k] += __uint_as_float(r[k] << 16); v[2 * k + 1] += __uint_as_float(r[k] & 0xffff0000u); }
                }
                if (X1K == 2) {
                    const u32x4_t r = xop[s][i][t];                 // GELU' only (other derivatives: ring kernels)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
#if PP_GELU_PK
                        const gelu_f32x2_t d2 = dgelu_fast_f2(gelu_f32x2_t{__uint_as_float(r[k] << 16), __uint_as_float(r[k] & 0xffff0000u)});
                        v[2 * k] *= d2[0];
                        v[2 * k + 1] *= d2[1];
#else
                        v[2 * k] *= dgelu_fast_f(__uint_as_float(r[k] << 16));
                        v[2 * k + 1] *= dgelu_fast_f(__uint_as_float(r[k] & 0xffff0000u));
#endif
                    }
                }
                o[t] = u32x4_t{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
            }
            if constexpr (QOUT) {
                const int l = opaque_lane();
                const int fr_ = l & 15, fg_ = l >> 4;
                const int row = tl.m0 + wr * WM + PP_ROFF(s) + i * 16 + fr_;
                const int col = tl.n0 + wc * 64 + fg_ * 8;                     // + t * 32
                int e8[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float r[8], amax = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        r[2 * k] = __uint_as_float(o[t][k] << 16);
                        r[2 * k + 1] = __uint_as_float(o[t][k] & 0xffff0000u);
                        amax = fmaxf(amax, fmaxf(fabsf(r[2 * k]), fabsf(r[2 * k + 1])));
                    }
                    amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
                    amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
                    const unsigned int bits = __float_as_uint(amax);
                    int e = (int)(bits >> 23) - 8 + ((bits & 0x7fffffu) > 0x600000u ? 1 : 0);
                    e = e < 0 ? 0 : (e > 254 ? 254 : e);
                    e8[t] = e;
                    u32x2_t c2;
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        float a4[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) a4[k] = fminf(fmaxf(__builtin_amdgcn_ldexpf(r[h2 * 4 + k], 127 - e), -448.f), 448.f);
                        int pk = __builtin_amdgcn_cvt_pk_fp8_f32(a4[0], a4[1], 0, false);
                        pk = __builtin_amdgcn_cvt_pk_fp8_f32(a4[2], a4[3], pk, true);
                        c2[h2] = (unsigned int)pk;
                    }
                    const unsigned int qoff = (row < g.M && col < g.N) ? (unsigned int)(row * g.N + col + t * 32) : OOB;
                    __builtin_amdgcn_raw_buffer_store_b64(c2, rs_q, qoff, 0, 0);
                }
                {   // scale bytes: lanes fg = 0 / 1 write the block of t = 0 / 1 (one store instruction)
                    const int kb = ((tl.n0 + wc * 64) >> 5) + fg_;
                    const int blk = row / 48, rb = row - blk * 48, nblk = ((g.M + 191) / 192) * 4;
                    const bool on = fg_ < 2 && row < g.M && tl.n0 + wc * 64 < g.N;
                    const unsigned int so = (unsigned int)((((kb >> 2) * nblk + blk) * 4 + (kb & 3)) * 64 + (rb & 15) * 4 + (rb >> 4));
                    // (a buffer store with an out-of-range offset for the lanes that have nothing to write: the number of vector-memory
                    // instructions per epilogue must not depend on the data -- the counted waits rely on it)
                    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(fg_ == 0 ? e8[0] : e8[1]), rs_s, on ? so : OOB, 0, 0);
                }
            }
            pair_swap(o[0], o[1], f, sec);
#if EGV_PP_EXP == 1
            asm volatile("" :: "v"(f), "v"(sec));
#else
            __builtin_amdgcn_raw_buffer_store_b128(f, rs_c, p_off(tl, s, i, 0, true), 0, PP_ST_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(sec, rs_c, p_off(tl, s, i, 1, true), 0, PP_ST_AUX);
#endif
        }
    };

    // ---- spread epilogue (EPI2): segment sg of tile `tl`.  A vector = fragment i of quadrant (s, t): row m0 + wr*WM + ROFF(s) + i*16 + fr, columns
    // n0 + wc*64 + t*32 + fg*8 .. +7 -- the MFMA C layout as it is (16 rows x 64 contiguous bytes per store instruction: the store path takes
    // the same 32 cycles per instruction as for the 8-rows-x-128-bytes form of the pair-swapped epilogue, tools/probe_store.hip), so a vector
    // costs 8 adds (bias; + 0.f without one: bit-identical to the other GEMM kernels), 4 packed conversions, one address add and the store --
    // about 14 instructions against ~50 of the pair-swapped form with its per-fragment bias reads from LDS.
    const unsigned int lane_c2 = (unsigned int)((wr * WM + fr) * g.ldc + lane_col) * 2u;
    auto epi2_seg = [&](int sg, const PPTile& tl) {               // sg compile-time at every call site
        const int vb = pp_e2_beg(sg, NV, E2M), ve = vb + pp_e2_cnt(sg, NV, E2M);
        const unsigned int col_oob = tl.n0 + wc * 64 < g.N ? 0u : OOB;   // wave-uniform (N % 64 == 0); or-ed in: no branch
        f32x4_t bA = {0.f, 0.f, 0.f, 0.f}, bB = {0.f, 0.f, 0.f, 0.f};
        int bt = -1;
#pragma unroll
        for (int v = vb; v < ve; ++v) {
            const int qd = v < IM ? 0 : v < 2 * IM ? 1 : v < 2 * IM + IM1 ? 2 : 3;
            const int i = v - (qd == 0 ? 0 : qd == 1 ? IM : qd == 2 ? 2 * IM : 2 * IM + IM1);
            const int s_ = qd >> 1, t_ = (qd == 1 || qd == 2) ? 1 : 0;
            if (t_ != bt) {                                       // (compile-time after unrolling) the 8 bias values of this lane's columns
                const float* bp = bias_slab + wc * 64 + fg * 8 + t_ * 32;
                bA = *reinterpret_cast<const f32x4_t*>(bp);
                bB = *reinterpret_cast<const f32x4_t*>(bp + 4);
                bt = t_;
            }
            f32x4_t a0 = acc[PP_AOFF(s_) + i][t_ * 2 + 0] + bA, a1 = acc[PP_AOFF(s_) + i][t_ * 2 + 1] + bB;
            if (has_gate) {                                       // (wave-uniform; the gated data gradient of proj_i2t: the product is rounded on its own, as in every GEMM kernel)
#pragma unroll
                for (int k = 0; k < 4; ++k) { a0[k] *= gate; a1[k] *= gate; }
            }
            const u32x4_t o = {pack_bf16x2(a0[0], a0[1]), pack_bf16x2(a0[2], a0[3]), pack_bf16x2(a1[0], a1[1]), pack_bf16x2(a1[2], a1[3])};
            const unsigned int sterm = ((unsigned int)((tl.m0 + PP_ROFF(s_) + i * 16) * g.ldc + tl.n0 + t_ * 32) * 2u) | col_oob;
#if PP_E2X == 1                                                       /* experiment: conversion without the store */
            asm volatile("" :: "v"(o), "v"(lane_c2 + sterm));
#elif PP_E2X == 2                                                     /* experiment: the store without the conversion (raw accumulator bits) */
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, a0), rs_c, lane_c2 + sterm, 0, PP_ST_AUX);
#else
            __builtin_amdgcn_raw_buffer_store_b128(o, rs_c, lane_c2 + sterm, 0, PP_ST_AUX);
#endif
        }
    };

    // ---- prologue: the first tile's bias (and residual quadrants 0, 1) first, then U0..U3 of K-tile 0 and U0 U1 of K-tile 1
    // (the units the steady-state schedule would have issued before phase 0)
    PPTile cur = pp_tile(first, g.tiles_n, BM);
    stage_scales();
    stage_unit(0); stage_unit(1); stage_unit(2); stage_unit(3);
    advance_cursor();
    stage_unit(0);
    stage_scales();
    stage_unit(1);
    pp_wait_vmcnt<8 + (MX ? 1 : 0)>();                            // U0(0), U1(0) (and the scales of K-tile 0) landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();

    // one K-tile = 4 phases.  KIND selects the vmcnt counts and the extra work of the phases:
    //   FIRST_CHAIN: quadrant epilogues of the previous tile `prev` + accumulator init (bias / residual) + operand loads of
    //                quadrants 2, 3;  FIRST_COLD: the same without a previous tile;  LAST: next tile's bias and the operand
    //                loads of quadrants 0, 1 (of `nxt` for a residual, of `cur` for a GELU' operand).
// MX form: the scaled MFMAs of a cold tile start from a zero C and depend on nothing but their fragments -- without a scheduling
// fence at the phase boundaries hipcc moves them across the barriers (and spills accumulators to make room)
#define PP_PIN() do { if constexpr (MX) __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_CAT8(X) __builtin_shufflevector(__builtin_bit_cast(i32x4_t, (X)[0]), __builtin_bit_cast(i32x4_t, (X)[1]), 0, 1, 2, 3, 4, 5, 6, 7)
#define PP_MX1(S, BF, T, ZERO, I, JP)                                                                                      \
    acc[PP_AOFF(S) + (I)][(T) * 2 + (JP)] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                              \
        PP_CAT8(BF[JP]), PP_CAT8(af[I]), (ZERO) ? f32x4_t{0.f, 0.f, 0.f, 0.f} : acc[PP_AOFF(S) + (I)][(T) * 2 + (JP)], 0, 0, \
        (T) * 2 + (JP), sc_b, (I), (S) == 0 ? sc_a0 : sc_a1)
#define PP_MFMA(S, BF, T, ZERO, PH)                                                                                         \
    do {                                                                                                                   \
        PP_SETPRIO(1);                                                                                                     \
        if constexpr (MX) {                                                                                                \
            /* pinned inside the phase: every MFMA reads sc_b (defined here, after the barrier) and its result is consumed */ \
            /* by an empty volatile statement before the closing barrier -- hipcc otherwise sinks all 32 to the loop's end */ \
            asm volatile("" : "+v"(sc_b));                                                                                 \
            PP_MX1(S, BF, T, ZERO, 0, 0); PP_MX1(S, BF, T, ZERO, 0, 1); PP_MX1(S, BF, T, ZERO, 1, 0); PP_MX1(S, BF, T, ZERO, 1, 1); \
            PP_MX1(S, BF, T, ZERO, 2, 0); PP_MX1(S, BF, T, ZERO, 2, 1);                                                    \
            if constexpr (IM == 4) { PP_MX1(S, BF, T, ZERO, IM - 1, 0); PP_MX1(S, BF, T, ZERO, IM - 1, 1); }               \
            _Pragma("unroll") for (int i = 0; i < IM; ++i)                                                                 \
            _Pragma("unroll") for (int jp = 0; jp < 2; ++jp) asm volatile("" : "+v"(acc[PP_AOFF(S) + i][(T) * 2 + jp]));   \
        } else {                                                                                                           \
        _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                                   \
        _Pragma("unroll") for (int i = 0; i < PP_IMS(S); ++i)                                                              \
        _Pragma("unroll") for (int jp = 0; jp < 2; ++jp)                                                                   \
            acc[PP_AOFF(S) + i][(T) * 2 + jp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                   \
                BF[jp][kh], af[i][kh], ((ZERO) && kh == 0) ? f32x4_t{0.f, 0.f, 0.f, 0.f} : acc[PP_AOFF(S) + i][(T) * 2 + jp], 0, 0, 0); \
        }                                                                                                                  \
        /* spread epilogue, mode 2: segment PH - 1 of the tile this K-tile finishes / segment 3 + PH of the previous tile, beside the MFMAs above */ \
        if (LASTK && E2M == 2 && (PH) >= 1) epi2_seg((PH) - 1, cur);                                                       \
        if (CHAIN && E2M == 2 && (PH) <= 2) epi2_seg(3 + (PH), prev);                                                      \
        PP_SETPRIO(0);                                                                                                     \
    } while (0)
#define PP_KTILE(KIND, BUFIDX)                                                                                             \
    do {                                                                                                                   \
        constexpr bool FIRSTK = (KIND) == PP_FIRST_CHAIN || (KIND) == PP_FIRST_COLD;                                       \
        constexpr bool CHAIN = (KIND) == PP_FIRST_CHAIN;                                                                   \
        constexpr bool LASTK = (KIND) == PP_LAST;                                                                          \
        constexpr bool SECONDK = (KIND) == PP_SECOND_CHAIN || (KIND) == PP_SECOND_COLD;                                    \
        const unsigned char* buf = smem + ((BUFIDX) & 1) * PP_BUF;                                                         \
        /* ---------------- phase 0: read A sub 0 (U0) + B sub 0 (U1); stage U2 of kt+1; quadrant 0 = (0,0) */             \
        {                                                                                                                  \
            if (CHAIN && !BULK && !EPI2) pp_wait_vmcnt<6 + (MX ? 1 : 0)>();   /* the bias DMA of LAST phase 1 (this wave's own slab) landed */ \
            if (CHAIN && !BULK && !EPI2) pair_epilogue(0, prev);                                                           \
            if (CHAIN && !BULK && !EPI2) PP_PIN();                                                                         \
            if (CHAIN && E2M == 1) epi2_seg(3, prev);                                                                          \
            if (SECONDK && EPI2) load_bias(cur);        /* this tile's bias: read from LAST phase 1 on (>= 5 phases later) */ \
            const unsigned char* pa = buf + 0 * PP_UNIT + a_base0;                                                         \
            const unsigned char* pb = buf + 1 * PP_UNIT + b_base;                                                          \
            _Pragma("unroll") for (int jp = 0; jp < 2; ++jp) {                                                             \
                bf0[jp][0] = *reinterpret_cast<const bf16x8_t*>(pb + jp * 2048 + swz0);                                    \
                bf0[jp][1] = *reinterpret_cast<const bf16x8_t*>(pb + jp * 2048 + swz1);                                    \
            }                                                                                                              \
            _Pragma("unroll") for (int i = 0; i < IM; ++i) {                                                               \
                af[i][0] = *reinterpret_cast<const bf16x8_t*>(pa + i * 2048 + swz0);                                       \
                af[i][1] = *reinterpret_cast<const bf16x8_t*>(pa + i * 2048 + swz1);                                       \
            }                                                                                                              \
            if constexpr (MX) {                                                                                            \
                sc_a0 = sc_read((BUFIDX) & 3, wr * 2 + 0);                                                                 \
                sc_b = sc_read((BUFIDX) & 3, 4 + wc);                                                                      \
            }                                                                                                              \
            stage_unit(2);                                                                                                 \
            pp_wait_vmcnt<pp_nwait(KIND, 0, NSP0, NSP1, BULK, MX, E2NV, E2M)>();                                                           \
            __builtin_amdgcn_s_barrier();                                                                                  \
            PP_MFMA(0, bf0, 0, FIRSTK, 0);                                                                                \
            __builtin_amdgcn_s_barrier();                                                                                  \
        }                                                                                                                  \
        /* ---------------- phase 1: read B sub 1 (U2); stage U3 of kt+1; quadrant 1 = (0,1) */                            \
        {                                                                                                                  \
            if (LASTK && !EPI2) load_bias(cur);                                                                           \
            if (LASTK && E2M == 1) epi2_seg(0, cur);                                                                          \
            if (CHAIN && E2M == 1) epi2_seg(4, prev);                                                                          \
            const unsigned char* pb = buf + 2 * PP_UNIT + b_base;                                                          \
            _Pragma("unroll") for (int jp = 0; jp < 2; ++jp) {                                                             \
                bf1[jp][0] = *reinterpret_cast<const bf16x8_t*>(pb + jp * 2048 + swz0);                                    \
                bf1[jp][1] = *reinterpret_cast<const bf16x8_t*>(pb + jp * 2048 + swz1);                                    \
            }                                                                                                              \
            stage_unit(3);                                                                                                 \
            advance_cursor();                                                                                              \
            pp_wait_vmcnt<pp_nwait(KIND, 1, NSP0, NSP1, BULK, MX, E2NV, E2M)>();                                                           \
            __builtin_amdgcn_s_barrier();                                                                                  \
            PP_MFMA(0, bf1, 1, FIRSTK, 1);                                                                                \
            __builtin_amdgcn_s_barrier();                                                                                  \
        }                                                                                                                  \
        /* ---------------- phase 2: read A sub 1 (U3); stage U0 of kt+2; quadrant 2 = (1,1) */                            \
        {                                                                                                                  \
            if (CHAIN && !BULK && !EPI2) pair_epilogue(1, prev);                                                           \
            if (CHAIN && !BULK && !EPI2) PP_PIN();                                                                         \
            if (LASTK && E2M == 1) epi2_seg(1, cur);                                                                          \
            if (CHAIN && E2M == 1) epi2_seg(5, prev);                                                                          \
            const unsigned char* pa = buf + 3 * PP_UNIT + a_base1;                                                         \
            _Pragma("unroll") for (int i = 0; i < IM1; ++i) {                                                              \
                af[i][0] = *reinterpret_cast<const bf16x8_t*>(pa + i * 2048 + swz0);                                       \
                af[i][1] = *reinterpret_cast<const bf16x8_t*>(pa + i * 2048 + swz1);                                       \
            }                                                                                                              \
            if constexpr (MX) sc_a1 = sc_read((BUFIDX) & 3, wr * 2 + 1);                                                   \
            stage_unit(0);                                                                                                 \
            pp_wait_vmcnt<pp_nwait(KIND, 2, NSP0, NSP1, BULK, MX, E2NV, E2M)>();                                                           \
            __builtin_amdgcn_s_barrier();                                                                                  \
            PP_MFMA(1, bf1, 1, FIRSTK, 2);                                                                                \
            __builtin_amdgcn_s_barrier();                                                                                  \
        }                                                                                                                  \
        /* ---------------- phase 3: no reads; stage U1 of kt+2; quadrant 3 = (1,0) */                                     \
        {                                                                                                                  \
            if (LASTK && E2M == 1) epi2_seg(2, cur);                                                                          \
            if (CHAIN && E2M == 1) epi2_seg(6, prev);                                                                          \
            stage_scales();                                                                                                \
            stage_unit(1);                                                                                                 \
            pp_wait_vmcnt<pp_nwait(KIND, 3, NSP0, NSP1, BULK, MX, E2NV, E2M)>();                                                           \
            __builtin_amdgcn_s_barrier();                                                                                  \
            PP_MFMA(1, bf0, 0, FIRSTK, 3);                                                                                \
            __builtin_amdgcn_s_barrier();                                                                                  \
        }                                                                                                                  \
    } while (0)

    PPTile prev = cur, nxt = cur;
    bool have_next = false;
    for (int ts = 0; ts < my_tiles; ++ts) {
        have_next = ts + 1 < my_tiles;
        nxt = pp_tile(first + (have_next ? ts + 1 : ts) * G, g.tiles_n, BM);
        // the second wave row runs one barrier behind the first inside a tile (ping-pong); the skew is applied per tile (and
        // undone by the first row at the tile's end) so that both rows cross the tile boundary together
        if (wr == 1) __builtin_amdgcn_s_barrier();
#ifdef EGV_INSTRUMENT
#define PP_STAMP(KT_) do { if (STAMPS && g.colsum && lane == 0 && (wave & 3) == 0 && ts < 4 && (KT_) < 16)                      \
            reinterpret_cast<long long*>(g.colsum)[((blockIdx.x * 2 + wr) * 4 + ts) * 16 + (KT_)] = __builtin_readcyclecounter(); } while (0)
#else
#define PP_STAMP(KT_) do { } while (0)
#endif
        PP_STAMP(0);
        const int kb = ts * KT;
        if (ts == 0) {
            PP_KTILE(PP_FIRST_COLD, kb);
            PP_STAMP(1);
            PP_KTILE(PP_SECOND_COLD, kb + 1);
        } else {
            PP_KTILE(PP_FIRST_CHAIN, kb);
            PP_STAMP(1);
            PP_KTILE(PP_SECOND_CHAIN, kb + 1);
        }
        for (int kt = 2; kt < KT - 1; ++kt) { PP_STAMP(kt); PP_KTILE(PP_PLAIN, kb + kt); }
        PP_STAMP(KT - 1);
        PP_KTILE(PP_LAST, kb + KT - 1);
        PP_STAMP(KT);
        if (wr == 0) __builtin_amdgcn_s_barrier();
        if (BULK) {
            // all 16 operand vectors first (a load issued after a store would wait for it), then convert + store; the stores
            // drain under the next tile's first K-tile (counted in its waits)
            if constexpr (X1K != 0)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < PP_IMS(s); ++i)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#if EGV_PP_EXP == 3                                                   // experiment: no operand loads (the VALU work and the stores stay)
                    { xop[s][i][t] = u32x4_t{0x3f803f80u, 0x3f003f00u, 0xbf80bf80u, 0x40004000u}; asm volatile("" : "+v"(xop[s][i][t])); }
#else
                        xop[s][i][t] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, q_off(cur, true, s, i, t, false), 0, PP_LD_AUX);
#endif
            if constexpr (X1K == 3) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int i = 0; i < PP_IMS(s); ++i)
#pragma unroll
                        for (int t = 0; t < 2; ++t)
                            xop2[s][i][t] = __builtin_amdgcn_raw_buffer_load_b128(rs_x2, q_off(cur, true, s, i, t, false), 0, 0);
            }
            pp_wait_vmcnt<6 + (MX ? 1 : 0) + ((EGV_PP_EXP == 3 || X1K == 0) ? 0 : (X1K == 3 ? 4 : 2) * (IM + IM1))>();                // the bias DMA of LAST phase 1 (16 operand loads are younger)
            pair_epilogue(0, cur);
            pair_epilogue(1, cur);
        }
        prev = cur;
        cur = nxt;
    }
#undef PP_KTILE
#undef PP_IMS
#undef PP_AOFF
#undef PP_ROFF
#undef PP_MFMA
#undef PP_MX1
#undef PP_PIN
#undef PP_CAT8
#undef PP_STAMP
    // ---- the last tile's epilogue
    pp_wait_vmcnt<0>();                                           // its bias slab (and the dummy DMAs)
    if constexpr (EPI2) {                                         // the quadrants the last K-tile's segments did not reach
        epi2_seg(3, prev); epi2_seg(4, prev); epi2_seg(5, prev);
        if constexpr (E2M == 1) epi2_seg(6, prev);
    } else if (!BULK) {
        pair_epilogue(0, prev);
        pair_epilogue(1, prev);
    }
    pp_wait_vmcnt<0>();
}

}  // namespace egv
using namespace egv;

// CUs the persistent grids of THIS host thread may plan for (0: all).  The block executor lowers it for the duration of a
// backward call whose weight gradients run as a persistent launch on a granted share of the chip (egv_gemm5.hip): a grid
// planned for CUs it cannot get would run its surplus workgroups as a second, nearly empty round.
static thread_local int g_cu_limit = 0;
static thread_local int g_cu_slack = -1;            // -1: EGV_PP_LIMIT_SLACK
void egv_gemm_set_cu_slack(int n) { g_cu_slack = n; }
extern thread_local int egv_prof_cus_hint;     // egv_api.cpp: the grid of a persistent launch, for the per-launch profile records
void egv_gemm_set_cu_limit(int n) { g_cu_limit = n; }

// returns 1 if the persistent ping-pong kernel took the call
#ifdef EGV_INSTRUMENT
namespace egv { extern float* g_timing_buf; }   // tools/gemm_pp_stamps.py (egv_debug_timing)
#endif
int egv_gemm3_launch(const GemmArgs& gin, hipStream_t st) {
    GemmArgs g = gin;
#ifdef EGV_INSTRUMENT
    g.colsum = egv::g_timing_buf;
    static const bool stamps = egv_cfg_on("EGV_PP_STAMPS", false);     // the STAMPS variant of the plain kind
#else
    constexpr bool stamps = false;
#endif
    const GemmEpi& e = g.e;
    if ((g.K % 64) || g.K < 192 || (g.N % 64) || !g.a_vec_ok || !g.b_vec_ok) return 0;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if ((g.ldc % 8) || (e.ldr % 8) || !al16(g.C) || !al16(e.res1) || !al16(e.pre) || !al16(e.aux) || (e.bias && !al16(e.bias))) return 0;
    if ((long long)g.M * g.lda >= (1LL << 30) || (long long)g.N * g.ldb >= (1LL << 30)) return 0;   // 32-bit byte offsets
    if ((long long)g.M * g.ldc >= (1LL << 30) || (long long)g.M * e.ldr >= (1LL << 30)) return 0;
    if (g.M >= (1 << 24) || g.N >= (1 << 24) || g.lda >= (1 << 23) || g.ldb >= (1 << 23)) return 0;   // 24-bit factors of the staging offsets
    if (e.scale != 1.0f) return 0;                                // the bias rides in as the accumulators' initial value
    if ((e.act && e.act != 1 && e.act != 4) || (e.dact && (e.dact != 1 || e.bias))) return 0;   // epilogues are built for GELU / GELU' (a data gradient: no bias) only
    if (e.res2 && (!e.res1 || e.dact || e.act)) return 0;             // gated two-residual form; e.pre = the saved pre-gate value
    if (e.res1 && !e.res2 && (e.gate || e.dact || e.act || e.pre)) return 0;
    if (e.dact && (e.act || e.pre)) return 0;
    if (e.pre && !e.act && !e.res2) return 0;
    g.tiles_n = (g.N + 255) / 256;
    static int ncu_dev = 0;
    int ncu = ncu_dev;
    if (!ncu) {
        hipDeviceProp_t prop;
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipGetDeviceProperties(&prop, dev);
        ncu = prop.multiProcessorCount > 0 ? (prop.multiProcessorCount / 8) * 8 : 256;
        // The persistent workgroups own their CU (144 KB LDS, 512 threads x 252 VGPRs): kernels of the text / weight-gradient
        // streams can only run beside them on CUs the grid leaves free.  The grid is trimmed per call (below) to the smallest
        // size that keeps the number of rounds; EGV_PP_CUS caps it (with untrimmed grids 7/8 of the CUs measured best).
        if (const int cap = egv_cfg_int("EGV_PP_CUS", 0)) ncu = (cap / 8) * 8;
        if (ncu < 8) ncu = 8;
        ncu_dev = ncu;
    }
    const int ncu_all = ncu;
    // A CU limit (the backward's grids beside a grouped weight-gradient launch) is a plan, not a fence: up to `slack` CUs beyond it are
    // taken when that removes a whole round of the walk (1176 tiles of the fc1 / fc2 class: 8 rounds on 160 CUs, 7 on 168) -- a few
    // workgroups then start late, behind the weight-gradient workgroups that hold their CUs, instead of every workgroup walking one
    // tile more (measured on configs[2]: limit 160 -> 168, 78.1 -> 76.7 ms per step)
    // -- which only pays when the companion launch ends early in the walk: a late workgroup walks ALL of its tiles after the others.
    // Where the companion is known to stay (fused blocks: its launch starts with the call and outlasts the first GEMMs) the
    // caller sets the slack to 0 (egv_gemm_set_cu_slack) and the limit is exact: a grid of 147 or 148 workgroups beside a
    // 108-workgroup launch instead of 152 of which four start after everything else has finished (fc2 data gradient 480 -> 240 us).
    static const int slack_env = egv_cfg_int("EGV_PP_LIMIT_SLACK", 16);
    const int slack = g_cu_slack >= 0 ? g_cu_slack : slack_env;
    int ncu_soft = 0;
    if (g_cu_limit > 0 && g_cu_limit < ncu_all) {
        ncu = slack > 0 ? (g_cu_limit >= 8 ? (g_cu_limit / 8) * 8 : 8) : (g_cu_limit >= 8 ? g_cu_limit : 8);
        ncu_soft = ncu + slack < ncu_all ? ncu + slack : ncu_all;
    }
    // tile height: the candidate whose walk is shortest -- rounds x (rows + c0), c0 = what a tile costs whatever its height (B loads,
    // fixed epilogue work: 56 rows' worth, from the measured 6 % penalty of 192- against 256-row tiles).  Built: plain kinds 256 / 224 /
    // 192 / 160 / 128 rows; residual kinds 256 / 192; gated two-residual 192; the GELU kinds 256 only.
    static const bool allow192 = egv_cfg_on("EGV_PP_BM192", true);
    static const int mixed_mode = egv_cfg_int("EGV_PP_MIXED", 0);   // 224- / 160- / 128-row tiles (round 5): 1 = grids that plan for the whole chip, 2 = also under a CU limit
    const bool allow_mixed = mixed_mode >= 2 || (mixed_mode == 1 && !(g_cu_limit > 0 && g_cu_limit < ncu_all));
    static const double c0 = egv_cfg_f64("EGV_PP_TILE_C0", 56.0);
    const bool kind192 = !e.dact && (!e.pre || e.res2) && !e.act && !stamps;
    const bool kind_plain = kind192 && !e.res1 && !e.res2 && !e.pre;
    auto rounds_of = [&](int t) {
        const int r = (t + ncu - 1) / ncu;
        return ncu_soft > ncu && (t + ncu_soft - 1) / ncu_soft < r ? (t + ncu_soft - 1) / ncu_soft : r;
    };
    const int t256 = ((g.M + 255) / 256) * g.tiles_n;
    int bm = 256;
    if (t256 >= ncu) {                                             // (small problems: one partial round of the tallest tile)
        double best = (double)rounds_of(t256) * (256.0 + c0);
        const int cand[4] = {224, 192, 160, 128};
        for (int c = 0; c < 4; ++c) {
            const int h = cand[c];
            const bool ok = h == 192 ? (allow192 && kind192) : (allow_mixed && kind_plain);
            if (!ok) continue;
            const double cost = (double)rounds_of(((g.M + h - 1) / h) * g.tiles_n) * (h + c0);
            if (cost < best * 0.999) { best = cost; bm = h; }
        }
    }
    if (const int f = egv_cfg_int("EGV_PP_FORCE_BM", 0)) {         // tests: a given height wherever the kind is built for it (read per call)
        if ((f == 256) || (f == 192 && kind192) || ((f == 224 || f == 160 || f == 128) && kind_plain)) bm = f;
    }
    if (e.res2) bm = 192;                                         // gated two-residual form (i2t projection): built for the 192-row tiles only
    g.tiles_m = (g.M + bm - 1) / bm;
    const int ntiles = g.tiles_m * g.tiles_n;
    // the smallest grid (multiple of 8: the XCD-aware walk) that keeps the number of rounds: the walk takes as long, and the CUs it
    // does not take serve the companion streams
    int rounds = (ntiles + ncu - 1) / ncu;
    if (ncu_soft > ncu && (ntiles + ncu_soft - 1) / ncu_soft < rounds) {
        rounds = (ntiles + ncu_soft - 1) / ncu_soft;
        ncu = ncu_soft;
    }
    static const bool trim = egv_cfg_on("EGV_PP_TRIM", true);
    int grid = trim ? (((ntiles + rounds - 1) / rounds + 7) / 8) * 8 : ncu;
    if (grid > ncu) grid = ncu;
    if (ntiles < grid) grid = ((ntiles + 7) / 8) * 8;
#define PP_GO(X, P, AC, ST, I0, I1)                                                                                      \
    do {                                                                                                                 \
        static bool attr = false;                                                                                        \
        if (!attr) {                                                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pp_kernel<X, P, AC, ST, I0, false, false, I1>), \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS);                               \
            attr = true;                                                                                                 \
        }                                                                                                                \
        egv_prof_cus_hint = grid;                                                                                        \
        hipLaunchKernelGGL((gemm_pp_kernel<X, P, AC, ST, I0, false, false, I1>), dim3(grid), dim3(512), PP_LDS, st, g, ntiles); \
        return 1;                                                                                                        \
    } while (0)
    static const int res_min_k = egv_cfg_int("EGV_PP_RES_MINK", 1536);
    if (e.res1 && g.K < res_min_k && !(bm == 192 && g.K >= res_min_k / 2)) return 0;   // short-K residual GEMMs on 256-row tiles: the 2-workgroup ring kernel hides their epilogue better
    if (e.res2 && !(allow192 && kind192)) return 0;
    if (e.res2 && e.pre) PP_GO(3, true, false, false, 3, 3);
    if (e.res2) PP_GO(3, false, false, false, 3, 3);
    if (e.res1 && bm == 192) PP_GO(1, false, false, false, 3, 3);
    if (e.res1) PP_GO(1, false, false, false, 4, 4);
    if (e.dact) PP_GO(2, false, false, false, 4, 4);
    if (e.pre) PP_GO(0, true, true, false, 4, 4);
    if (e.act) PP_GO(0, false, true, false, 4, 4);
#ifdef EGV_INSTRUMENT
    if (stamps) PP_GO(0, false, false, true, 4, 4);
#endif
    if (bm == 224) PP_GO(0, false, false, false, 4, 3);
    if (bm == 192) PP_GO(0, false, false, false, 3, 3);
    if (bm == 160) PP_GO(0, false, false, false, 3, 2);
    if (bm == 128) PP_GO(0, false, false, false, 2, 2);
    PP_GO(0, false, false, false, 4, 4);
#undef PP_GO
}

// ---- MX-fp8 form (BASELINE.json configs[4]: "fp8 MFMA weight path") ---------------------------------------------------------
// C[M,N] (bf16) = epi( sum_k sa[m,k/32] A[m,k] * sb[n,k/32] B[n,k] ): A, B e4m3 codes (row pitch = K bytes), sa / sb the E8M0 block
// scales in the lane order written by egv_quant_mx (role 0 / role 1).  Same epilogues as egv_gemm (bias, activation, saved
// pre-activation, gate-free residual, activation-derivative operand).  There is no other kernel behind it: shapes the persistent
// kernel does not take are an error.
void* egv_prof_begin(void* stream);
void egv_prof_end(void* handle, void* stream, double flops, int kind, double bytes);
extern "C" int egv_gemm_mx(int M, int N, int K, const void* Aq, const void* Ascales, const void* Bq, const void* Bscales, void* C, int ldc,
                           const float* bias, int act, const void* res1, void* pre, const void* aux, int dact, int ldr, void* out_q,
                           void* out_scales, void* stream) {
    EGV_CHECK(M > 0 && N > 0 && K > 0 && Aq && Ascales && Bq && Bscales && C, "egv_gemm_mx: null / empty operand");
    EGV_CHECK((K % 128) == 0 && K >= 384 && (N % 64) == 0, "egv_gemm_mx: K %% 128 == 0, K >= 384, N %% 64 == 0 required (N=%d K=%d)", N, K);
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!ldr) ldr = ldc;
    EGV_CHECK(al16(Aq) && al16(Bq) && al16(C) && al16(res1) && al16(pre) && al16(aux) && al16(bias) && (reinterpret_cast<uintptr_t>(Ascales) & 3) == 0 &&
              (reinterpret_cast<uintptr_t>(Bscales) & 3) == 0 && (ldc % 8) == 0 && (ldr % 8) == 0, "egv_gemm_mx: 16-byte aligned operands and ldc, ldr %% 8 == 0 required");
    EGV_CHECK((long long)M * K < (1LL << 31) && (long long)N * K < (1LL << 31) && (long long)M * ldc < (1LL << 30) && (long long)M * ldr < (1LL << 30),
              "egv_gemm_mx: operand beyond 32-bit byte offsets");
    EGV_CHECK(M < (1 << 24) && N < (1 << 24) && K < (1 << 23), "egv_gemm_mx: M, N < 2^24 and K < 2^23 required");
    EGV_CHECK(!(res1 && (dact || act || pre)) && !(dact && (act || pre)) && !(pre && !act), "egv_gemm_mx: epilogue combination not built");
    EGV_CHECK(!dact || aux, "egv_gemm_mx: dact without aux");
    EGV_CHECK((act == 0 || act == 1 || act == 4) && (dact == 0 || (dact == 1 && !bias)), "egv_gemm_mx: epilogues are built for GELU / GELU' (without bias) only");
    EGV_CHECK((out_q == nullptr) == (out_scales == nullptr), "egv_gemm_mx: out_q and out_scales come together");
    EGV_CHECK(!out_q || (((pre && act) || dact) && (N % 128) == 0 && ldc == N && al16(out_q)),
              "egv_gemm_mx: the quantised output is built for the GELU (saved pre-activation) and the GELU' epilogues, N %% 128 == 0, ldc == N");
    GemmArgs g{};
    g.A = Aq; g.B = Bq; g.C = C;
    g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldb = K; g.ldc = ldc;
    g.a_vec_ok = g.b_vec_ok = g.c_vec_ok = 1;
    g.k_per_split = K;
    g.e.bias = bias; g.e.res1 = res1; g.e.pre = pre; g.e.aux = aux; g.e.act = act; g.e.dact = dact; g.e.ldr = ldr; g.e.scale = 1.0f;
    g.sa = reinterpret_cast<const unsigned char*>(Ascales);
    g.sb = reinterpret_cast<const unsigned char*>(Bscales);
    g.mx = 1;
    g.e.oq = reinterpret_cast<unsigned char*>(out_q);
    g.e.os = reinterpret_cast<unsigned char*>(out_scales);
    g.tiles_n = (N + 255) / 256;
    g.tiles_m = (M + 191) / 192;
    const int ntiles = g.tiles_m * g.tiles_n;
    static int ncu_dev = 0;
    if (!ncu_dev) {
        hipDeviceProp_t prop;
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipGetDeviceProperties(&prop, dev);
        ncu_dev = prop.multiProcessorCount > 0 ? (prop.multiProcessorCount / 8) * 8 : 256;
        if (ncu_dev < 8) ncu_dev = 8;
    }
    int ncu = ncu_dev;
    if (g_cu_limit > 0 && g_cu_limit < ncu) ncu = g_cu_limit >= 8 ? (g_cu_limit / 8) * 8 : 8;
    const int rounds = (ntiles + ncu - 1) / ncu;
    int grid = (((ntiles + rounds - 1) / rounds + 7) / 8) * 8;
    if (grid > ncu) grid = ncu;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    void* ph = egv_prof_begin(stream);
#define MX_LAUNCH_Q(X, P, AC, QO)                                                                                          \
    do {                                                                                                                   \
        static bool attr = false;                                                                                          \
        if (!attr) {                                                                                                       \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pp_kernel<X, P, AC, false, 3, true, QO>),         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_MX);                              \
            attr = true;                                                                                                   \
        }                                                                                                                  \
        egv_prof_cus_hint = grid;                                                                                          \
        hipLaunchKernelGGL((gemm_pp_kernel<X, P, AC, false, 3, true, QO>), dim3(grid), dim3(512), PP_LDS_MX, st, g, ntiles); \
    } while (0)
#define MX_LAUNCH(X, P, AC) MX_LAUNCH_Q(X, P, AC, false)
    if (out_q && dact) MX_LAUNCH_Q(2, false, false, true);
    else if (out_q) MX_LAUNCH_Q(0, true, true, true);
    else if (res1) MX_LAUNCH(1, false, false);
    else if (dact) MX_LAUNCH(2, false, false);
    else if (pre) MX_LAUNCH(0, true, true);
    else if (act) MX_LAUNCH(0, false, true);
    else MX_LAUNCH(0, false, false);
#undef MX_LAUNCH
#undef MX_LAUNCH_Q
    // algorithmic bytes: fp8 operands + scales once, the bf16 output and every bf16 epilogue operand once
    const double abytes = (double)M * K * (1.0 + 1.0 / 32) + (double)N * K * (1.0 + 1.0 / 32) +
                          2.0 * M * N * (1 + (res1 != nullptr) + (pre != nullptr) + (aux != nullptr)) + (out_q ? (double)M * N * (1.0 + 1.0 / 32) : 0.0);
    egv_prof_end(ph, stream, 2.0 * M * N * K, 16, abytes);
    EGV_LAUNCH_CHECK();
    return 0;
}
