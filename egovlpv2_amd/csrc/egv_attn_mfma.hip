// MFMA (bf16) grouped attention for gfx950: divided space / time attention, RoBERTa self attention and the
// small-key cross attention of the EgoVLPv2 hot path, head_dim 64.
//
// One workgroup per (problem, head[, own-tile chunk]).  The OTHER side of the kernel (keys for fwd/dQ, queries for
// dK/dV; <= 224 rows incl. the optional extra CLS row) is staged once in LDS -- row-major (144-byte pitch, conflict-free
// ds_read_b128 fragments) and/or transposed ([64 d][rows], built with an in-register 8x8 bf16 block transpose, read as
// two ds_read_b64 per fragment) -- and every wave walks 16-row tiles of the OWN side with v_mfma_f32_16x16x32_bf16:
//
//   fwd : S^T = K Q^T (A = K rows from LDS, B = Q rows from global)  -> whole score row in registers (no online softmax),
//         softmax over the lane-local 4*NT values + 2 xor-shuffles, P stays in registers as the B operand of
//         O^T = V^T P^T (A = V^T from LDS).  The reduction index of the second MFMA is permuted (k = [tile 2kk rows g*4..,
//         tile 2kk+1 rows g*4..]) so that the C layout of the first MFMA is already the B layout of the second.
//   dQ  : S^T, dP^T = V dO^T, dS^T = P^T o (dP^T - delta), dQ^T = K^T dS^T.
//   dKV : S = Q K^T, dP = dO V^T (A = Q / dO rows from LDS, B = own K / V rows from global), dV^T = dO^T P, dK^T = Q^T dS.
//
// Every output row is written exactly once as 8-byte packed bf16 (4 consecutive head columns per lane).
#include "egv_attn.h"
#include <cstdlib>

#ifndef EGV_KK_BARRIER
#define EGV_KK_BARRIER
#endif
namespace egv {

constexpr int RP = 144;                       // row pitch (bytes) of row-major tiles: 64 bf16 + 16 B pad

__host__ __device__ constexpr int vt_pitch(int NT) { return NT * 32 + 16; }     // bytes: NT*16 rows of bf16 + pad

__device__ __forceinline__ long long other_row(const AttnArgs& a, const RowSet& rs, int b, int g, int j) {
    if (a.extra) return (j == 0) ? ((long long)b * a.extra_bs + a.extra_row) : rs_row(rs, b, g, j - 1);
    return rs_row(rs, b, g, j);
}

// stage other-side rows [0, ntot) of one head into a row-major LDS tile (rows >= ntot are zero)
template <int NT, int NTHR>
__device__ __forceinline__ void stage_rows(unsigned char* s, const bf16_t* base, int ld, int off, const AttnArgs& a,
                                           const RowSet& rs, int b, int g, int ntot, int tid, int r0 = 0) {
    for (int c = tid; c < NT * 16 * 8; c += NTHR) {
        const int r = c >> 3, v = c & 7;
        u32x4_t x = {0u, 0u, 0u, 0u};
        if (r < ntot) x = *reinterpret_cast<const u32x4_t*>(base + other_row(a, rs, b, g, r0 + r) * ld + off + v * 8);
        *reinterpret_cast<u32x4_t*>(s + r * RP + v * 16) = x;
    }
}

// stage the same rows transposed: sT[d][row] (bf16, pitch vt_pitch(NT)); 8x8 block transpose in registers
template <int NT, int NTHR>
__device__ __forceinline__ void stage_rows_t(unsigned char* s, const bf16_t* base, int ld, int off, const AttnArgs& a,
                                             const RowSet& rs, int b, int g, int ntot, int tid, int r0 = 0) {
    constexpr int VP = vt_pitch(NT);
    for (int u = tid; u < NT * 2 * 8; u += NTHR) {      // NT*16/8 row blocks x 8 d blocks
        const int kb = u >> 3, db = u & 7;
        u32x4_t r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = kb * 8 + j;
            r[j] = u32x4_t{0u, 0u, 0u, 0u};
            if (row < ntot) r[j] = *reinterpret_cast<const u32x4_t*>(base + other_row(a, rs, b, g, r0 + row) * ld + off + db * 8);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            u32x4_t o;
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) {
                const unsigned int x = r[2 * d4][e >> 1], y = r[2 * d4 + 1][e >> 1];
                o[d4] = (e & 1) ? ((x >> 16) | (y & 0xffff0000u)) : ((x & 0xffffu) | (y << 16));
            }
            *reinterpret_cast<u32x4_t*>(s + (db * 8 + e) * VP + kb * 16) = o;
        }
    }
}

__device__ __forceinline__ bf16x8_t ld_frag_row(const unsigned char* s, int row, int ks, int fg) {
    return *reinterpret_cast<const bf16x8_t*>(s + row * RP + ks * 64 + fg * 16);
}

// fragment of a transposed tile for k-step kk: elements [row-tile 2kk rows g*4..+3 | row-tile 2kk+1 rows g*4..+3] at column d
template <int NT>
__device__ __forceinline__ bf16x8_t ld_frag_t(const unsigned char* s, int d, int kk, int fg) {
    constexpr int VP = vt_pitch(NT);
    const u32x2_t lo = *reinterpret_cast<const u32x2_t*>(s + d * VP + ((2 * kk) * 16 + fg * 4) * 2);
    const u32x2_t hi = *reinterpret_cast<const u32x2_t*>(s + d * VP + ((2 * kk + 1) * 16 + fg * 4) * 2);
    u32x4_t v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8_t, v);
}

// A-operand fragment (row = tile row fr, 8 consecutive head dims of k-step ks) gathered from a TRANSPOSED tile sT[d][row]
// with the gfx950 transposing LDS read: lane fr supplies the address of (d0 + fr/4, row0 + (fr%4)*4 ..+3) and receives
// column row0 + fr of the 4 x 16 block (layout verified on hardware by tools/probe_tr.hip).
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
template <int NT>
__device__ __forceinline__ bf16x8_t ld_frag_tr(const unsigned char* sT, int row0, int ks, int fr, int fg) {
    constexpr int VP = vt_pitch(NT);
    const unsigned char* p = sT + (ks * 32 + fg * 8 + (fr >> 2)) * VP + (row0 + (fr & 3) * 4) * 2;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + 4 * VP));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
}

__device__ __forceinline__ bf16x8_t ld_frag_global(const bf16_t* p, bool valid) {
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (valid) v = *reinterpret_cast<const u32x4_t*>(p);
    return __builtin_bit_cast(bf16x8_t, v);
}

__device__ __forceinline__ unsigned int pack2(float a, float b) { return pack_bf16x2(a, b); }

// NL > 0: the number of live 16-row tiles of the other side is known at compile time (NL - 1 full tiles + one possibly partial
// tile); the unrolled per-tile code then carries no liveness branches at all.  NL == 0: decided at run time from ntot.
template <int NL> __device__ __forceinline__ bool tile_live(int t, int ntot) { if constexpr (NL > 0) return t < NL; else return t * 16 < ntot; }
template <int NL> __device__ __forceinline__ bool tile_partial(int t, int ntot) { if constexpr (NL > 0) return t == NL - 1; else return t * 16 + 16 > ntot; }
template <int NL> __device__ __forceinline__ bool pair_live(int kk, int ntot) { if constexpr (NL > 0) return 2 * kk < NL; else return kk * 32 < ntot; }

constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ bf16x8_t pack8(const f32x4_t& a, const f32x4_t& b) {
    u32x4_t v = {pack2(a[0], a[1]), pack2(a[2], a[3]), pack2(b[0], b[1]), pack2(b[2], b[3])};
    return __builtin_bit_cast(bf16x8_t, v);
}

__device__ __forceinline__ float grp_max(float v) {      // across the 4 lane groups (lanes l, l^16, l^32, l^48)
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float grp_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

__device__ __forceinline__ void st_bf16x4(bf16_t* p, float a, float b, float c, float d) {
    u32x2_t v = {pack2(a, b), pack2(c, d)};
    *reinterpret_cast<u32x2_t*>(p) = v;
}

// ------------------------------------------------------------------------------------------------
// forward.  NT = 16-row tiles of the key side (even).  grid (own chunks, problems, heads), NW waves,
// TPW own tiles per wave per workgroup.
// ------------------------------------------------------------------------------------------------
// FL: bit 0 = additive key mask present, bit 1 = probability dropout on; the video-side launches (neither) get a loop
// without the per-element conditionals.  Scores are kept in the log2 domain (scale * log2(e) folded into one multiply).
template <int NT, int NW, int FL, int NL = 0>
__global__ __launch_bounds__(64 * NW) void attn_fwd_mfma_kernel(const AttnArgs a, int tiles_per_wg) {
    constexpr int VP = vt_pitch(NT);
    constexpr bool MASK = (FL & 1) != 0, DROP = (FL & 2) != 0;
    const float sc2 = a.scale * LOG2E;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sK = smem;                       // [NT*16][RP]
    unsigned char* sVt = smem + NT * 16 * RP;       // [64][VP]

    const int tid = threadIdx.x, lane = tid & 63, w = wave_id();
    const int fr = lane & 15, fg = lane >> 4;
    const int p = blockIdx.y, b = p / a.G, g = p % a.G, h = blockIdx.z;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.Q);
    const bf16_t* K = reinterpret_cast<const bf16_t*>(a.K);
    const bf16_t* V = reinterpret_cast<const bf16_t*>(a.V);
    bf16_t* O = reinterpret_cast<bf16_t*>(a.O);
    const int hq = a.qoff + h * HD, hk = a.koff + h * HD, hv = a.voff + h * HD, ho = a.ooff + h * HD;
    // keys incl. the extra CLS key; with nsplit > 1 this workgroup covers the chunk [r0, r0 + ntot) of them and writes
    // partial softmax states (combined by attn_fwd_combine_kernel)
    const int nall = a.k.n + a.extra;
    const int split = blockIdx.x % a.nsplit;
    const int per = a.nsplit > 1 ? ((((nall + a.nsplit - 1) / a.nsplit) + 15) & ~15) : nall;
    const int r0 = split * per;
    const int ntot = max(0, min(per, nall - r0));

    stage_rows<NT, 64 * NW>(sK, K, a.ldk, hk, a, a.k, b, g, ntot, tid, r0);
    stage_rows_t<NT, 64 * NW>(sVt, V, a.ldv, hv, a, a.k, b, g, ntot, tid, r0);
    __syncthreads();

    const int nqt = (a.q.n + 15) >> 4;
    // cls_out (unsplit launch with an extra row and a workspace): the extra row is also a QUERY of this group -- the CLS query of
    // the divided attention sees all S keys = the union of the groups' keys -- handled as one more own tile whose one live row
    // leaves a partial softmax state (max, sum, 64 outputs) per group for attn_fwd_combine_kernel; the CLS KEY counts in group 0
    const bool cls_out = !MASK && a.nsplit == 1 && a.ws != nullptr && a.extra;
    const int t0 = (blockIdx.x / a.nsplit) * tiles_per_wg;
    const int t1 = min(nqt + (cls_out ? 1 : 0), t0 + tiles_per_wg);
    for (int qt = t0 + w; qt < t1; qt += NW) {
        const bool is_cls = qt == nqt;                                          // uniform
        const int q = qt * 16 + fr;
        const bool qv = is_cls ? fr == 0 : q < a.q.n;
        const long long qrow = is_cls ? ((long long)b * a.extra_bs + a.extra_row) : (qv ? rs_row(a.q, b, g, q) : 0);
        const bf16x8_t q0 = ld_frag_global(Q + qrow * a.ldq + hq + fg * 8, qv);
        const bf16x8_t q1 = ld_frag_global(Q + qrow * a.ldq + hq + 32 + fg * 8, qv);
        f32x4_t s[NT];
        float m = -INFINITY, l = 0.f;
        if constexpr (!MASK) {
            // pass A: all QK^T MFMAs back to back (no VALU reads an accumulator before the whole batch has issued)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
                if (tile_live<NL>(t, ntot)) {                                  // uniform: tiles past the last key are never touched
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag_row(sK, t * 16 + fr, 0, fg), q0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag_row(sK, t * 16 + fr, 1, fg), q1, acc, 0, 0, 0);
                }
                s[t] = acc;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (is_cls && g != 0 && fg == 0) s[0][0] = -INFINITY;              // the CLS key (key 0) is counted in group 0
            // pass B: row maximum of the raw scores (the scale is positive, so it commutes with max)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (tile_live<NL>(t, ntot)) {
                    if (tile_partial<NL>(t, ntot)) {                         // uniform: only the last, partial tile pays for the select
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int r = 0; r < 4; ++r) s[t][r] = (t * 16 + fg * 4 + r < ntot) ? s[t][r] : -INFINITY;
                    }
                    m = fmaxf(m, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
                }
            }
            m = grp_max(m) * sc2;
            // pass C: p = exp2(score * scale * log2e - m): one FMA and one v_exp_f32 per element
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (tile_live<NL>(t, ntot)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = exp2_fast(fmaf(s[t][r], sc2, -m));
                        s[t][r] = e;
                        l += e;
                    }
                } else {
                    s[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                }
            }
        } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            if (tile_live<NL>(t, ntot)) {                                      // uniform: tiles past the last key are never touched
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag_row(sK, t * 16 + fr, 0, fg), q0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag_row(sK, t * 16 + fr, 1, fg), q1, acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = t * 16 + fg * 4 + r;              // index inside this chunk; r0 + key is the global key index
                    float v = acc[r] * sc2;
                    if (a.mask && r0 + key >= a.extra && key < ntot) v += a.mask[(long long)b * a.mask_ld + r0 + key - a.extra] * LOG2E;
                    acc[r] = v;
                }
                if (tile_partial<NL>(t, ntot)) {                             // uniform: only the last, partial tile pays for the select
                    asm volatile("" ::: "memory");                    // keep this a scalar branch (no if-conversion into 56 selects)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = (t * 16 + fg * 4 + r < ntot) ? acc[r] : -INFINITY;
                }
                m = fmaxf(m, fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])));
            } else {
                acc = f32x4_t{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            }
            s[t] = acc;
            if (t & 1) __builtin_amdgcn_sched_barrier(0);
        }
        m = grp_max(m);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (tile_live<NL>(t, ntot)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = exp2_fast(s[t][r] - m);
                    s[t][r] = e;
                    l += e;
                }
            } else {
                s[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
        }
        }
        l = grp_sum(l);
        if (DROP) {                                   // dropout on the normalised probabilities: l keeps the full sum
            if (a.drop_p > 0.f) {
                const long long qid = (long long)p * a.q.n + q;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[t][r] *= drop_mult(a, qid, r0 + t * 16 + fg * 4 + r, h);
            }
        }
        f32x4_t o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < NT / 2; ++kk) {
            if (pair_live<NL>(kk, ntot)) {
                const bf16x8_t pf = pack8(s[2 * kk], s[2 * kk + 1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag_t<NT>(sVt, dt * 16 + fr, kk, fg), pf, o[dt], 0, 0, 0);
            }
            EGV_KK_BARRIER
        }
        if (is_cls) {
            if (qv) {                                                          // ws[group][b][head][66]: the layout of a G-way split, one query per sample
                float* dst = a.ws + (((long long)g * (gridDim.y / a.G) + b) * a.H + h) * 66;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4_t*>(dst + 2 + dt * 16 + fg * 4) = o[dt];
                if (fg == 0) { dst[0] = m * LN2; dst[1] = l; }
            }
        } else if (qv && a.nsplit > 1) {
            const long long nrows = (long long)gridDim.y * a.q.n;
            const long long orow = (long long)p * a.q.n + q;
            float* dst = a.ws + (((long long)split * nrows + orow) * a.H + h) * 66;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4_t*>(dst + 2 + dt * 16 + fg * 4) = o[dt];
            if (fg == 0) { dst[0] = m * LN2; dst[1] = l; }
        } else if (qv) {
            const float inv = 1.0f / l;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                st_bf16x4(O + qrow * a.ldo + ho + dt * 16 + fg * 4, o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
            if (a.O32) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    *reinterpret_cast<f32x4_t*>(a.O32 + qrow * a.ldo + ho + dt * 16 + fg * 4) = f32x4_t{o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv};
            }
            if (a.lse && fg == 0) a.lse[qrow * a.H + h] = m * LN2 + __logf(l);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, query-owned: dQ and delta = rowsum(dO * O)
// ------------------------------------------------------------------------------------------------
template <int NT, int NW, int FL, int NL = 0>
__global__ __launch_bounds__(64 * NW) void attn_dq_mfma_kernel(const AttnArgs a, int tiles_per_wg) {
    constexpr int VP = vt_pitch(NT);
    constexpr bool MASK = (FL & 1) != 0, DROP = (FL & 2) != 0;
    const float sc2 = a.scale * LOG2E;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sKt = smem;                      // [64][VP]  K transposed (A operand of S^T via transposing reads, and of dQ^T)
    unsigned char* sVt = sKt + 64 * VP;             // [64][VP]  V transposed (A operand of dP^T via transposing reads)

    const int tid = threadIdx.x, lane = tid & 63, w = wave_id();
    const int fr = lane & 15, fg = lane >> 4;
    const int p = blockIdx.y, b = p / a.G, g = p % a.G, h = blockIdx.z;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.Q);
    const bf16_t* K = reinterpret_cast<const bf16_t*>(a.K);
    const bf16_t* V = reinterpret_cast<const bf16_t*>(a.V);
    const bf16_t* O = reinterpret_cast<const bf16_t*>(a.O);
    const bf16_t* dO = reinterpret_cast<const bf16_t*>(a.dO);
    bf16_t* dQ = reinterpret_cast<bf16_t*>(a.dQ);
    const int hq = a.qoff + h * HD, hk = a.koff + h * HD, hv = a.voff + h * HD, ho = a.ooff + h * HD, hdq = a.dqoff + h * HD;
    const int nall = a.k.n + a.extra;
    const int split = blockIdx.x % a.nsplit;
    const int per = a.nsplit > 1 ? ((((nall + a.nsplit - 1) / a.nsplit) + 15) & ~15) : nall;
    const int r0 = split * per;
    const int ntot = max(0, min(per, nall - r0));

    stage_rows_t<NT, 64 * NW>(sKt, K, a.ldk, hk, a, a.k, b, g, ntot, tid, r0);
    stage_rows_t<NT, 64 * NW>(sVt, V, a.ldv, hv, a, a.k, b, g, ntot, tid, r0);
    __syncthreads();

    // cls_out (one-wave, one-tile groups: the 17-row time attention; unsplit launch with an extra key and a workspace): the extra
    // KEY's gradient under this group's queries -- dK = sum_q dS[q, cls] Q[q], dV = sum_q P[q, cls] dO[q] -- is left as an fp32
    // partial per group (slots 1, 2 of ws[problem][head][3][64]) for attn_cls_reduce_kernel; the column is computed here anyway
    const bool cls_out = NT == 2 && NW == 1 && !MASK && !DROP && a.nsplit == 1 && a.ws != nullptr && a.extra;
    float p_cls = 0.f, ds_cls = 0.f;

    const int nqt = (a.q.n + 15) >> 4;
    const int t0 = (blockIdx.x / a.nsplit) * tiles_per_wg;
    const int t1 = min(nqt, t0 + tiles_per_wg);
    for (int qt = t0 + w; qt < t1; qt += NW) {
        const int q = qt * 16 + fr;
        const bool qv = q < a.q.n;
        const long long qrow = qv ? rs_row(a.q, b, g, q) : 0;
        const bf16x8_t q0 = ld_frag_global(Q + qrow * a.ldq + hq + fg * 8, qv);
        const bf16x8_t q1 = ld_frag_global(Q + qrow * a.ldq + hq + 32 + fg * 8, qv);
        const bf16x8_t g0 = ld_frag_global(dO + qrow * a.ldo + ho + fg * 8, qv);
        const bf16x8_t g1 = ld_frag_global(dO + qrow * a.ldo + ho + 32 + fg * 8, qv);
        // (O is only read for delta: not at all where delta comes from the probability rows themselves, below)
        const bool need_o = !(NT == 2 && a.nsplit == 1) && !a.O32;
        const bf16x8_t o0 = ld_frag_global(O + qrow * a.ldo + ho + fg * 8, qv && need_o);
        const bf16x8_t o1 = ld_frag_global(O + qrow * a.ldo + ho + 32 + fg * 8, qv && need_o);
        float dl = 0.f;
        if (a.O32) {                                                          // delta from the fp32 values of O (egv_attn_desc::O32)
            const float* o32 = a.O32 + qrow * a.ldo + ho + fg * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += qv ? (float)g0[e] * o32[e] + (float)g1[e] * o32[32 + e] : 0.f;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += (float)g0[e] * (float)o0[e] + (float)g1[e] * (float)o1[e];
        }
        dl = grp_sum(dl);
        const float lse2 = qv ? a.lse[qrow * a.H + h] * LOG2E : 0.f;
        if (qv && fg == 0 && split == 0 && !(NT == 2 && a.nsplit == 1)) a.delta[qrow * a.H + h] = dl;

        f32x4_t o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < NT / 2; ++kk) {
            f32x4_t pr[2];
            [[maybe_unused]] f32x4_t pjv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = 2 * kk + u;
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                if (tile_live<NL>(t, ntot)) {                                  // uniform: dead key tiles contribute dS = 0
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag_tr<NT>(sKt, t * 16, 0, fr, fg), q0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag_tr<NT>(sKt, t * 16, 1, fr, fg), q1, acc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag_tr<NT>(sVt, t * 16, 0, fr, fg), g0, dp, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag_tr<NT>(sVt, t * 16, 1, fr, fg), g1, dp, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = t * 16 + fg * 4 + r;
                        float v = acc[r] * sc2 - lse2;
                        if (MASK) {
                            if (a.mask && r0 + key >= a.extra && key < ntot) v += a.mask[(long long)b * a.mask_ld + r0 + key - a.extra] * LOG2E;
                        }
                        const float pj = exp2_fast(v);
                        float dpv = dp[r];
                        if (DROP) {
                            if (a.drop_p > 0.f) dpv *= drop_mult(a, (long long)p * a.q.n + q, r0 + key, h);
                        }
                        if constexpr (NT == 2) {                              // both key tiles first: delta may come from them (below)
                            pjv[u][r] = key < ntot ? pj : 0.f;
                            acc[r] = dpv;
                        } else {
                            acc[r] = pj * (dpv - dl);
                            if (t == 0 && r == 0) { p_cls = pj; ds_cls = acc[0]; }     // key 0 = the extra key (lanes fg == 0)
                        }
                    }
                    if (NT != 2 && tile_partial<NL>(t, ntot)) {              // uniform: the partial tile zeroes its padding keys
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[r] = (t * 16 + fg * 4 + r < ntot) ? acc[r] : 0.f;
                    }
                } else if constexpr (NT == 2) {
                    pjv[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                }
                pr[u] = acc;
            }
            if constexpr (NT == 2) {
                // <= 32 keys: a query's whole probability row and dP row are in the registers of its four lanes, so
                //   delta = sum_j P_j dP_j
                // is formed from the very values dS is formed from (unsplit launches) instead of rowsum(dO o O) with the bf16-rounded O:
                // that rounding is a COMMON offset under all keys of a row and ends up, times the mean key, in dQ -- the same effect as in
                // the text -> image attention (egv_attn_desc::O32), measured there as 3-5 x the reference-under-autocast's q / k gradient
                // error.  The stored delta (read by egv_attn_bwd_dkv) is this one.
                float dd = dl;
                if (a.nsplit == 1) {
                    float sdl = 0.f;
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sdl += pjv[u][r] * pr[u][r];
                    dd = grp_sum(sdl);
                    if (qv && fg == 0) a.delta[qrow * a.H + h] = dd;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pr[u][r] = pjv[u][r] * (pr[u][r] - dd);
                p_cls = pjv[0][0];
                ds_cls = pr[0][0];
            }
            if (pair_live<NL>(kk, ntot)) {                                     // dQ^T += K^T dS^T for this pair of key tiles
                const bf16x8_t dsf = pack8(pr[0], pr[1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag_t<NT>(sKt, dt * 16 + fr, kk, fg), dsf, o[dt], 0, 0, 0);
            }
            EGV_KK_BARRIER
        }
        if (cls_out) {
            // lane (fr, fg) holds Q / dO [query fr][fg*8 .. +7 | 32 + fg*8 ..]: weight by the query's dS / P of the extra key
            // (lane fr of the first row) and sum over the 16 queries of the row with DPP exchanges
            float dc = __shfl(ds_cls, fr, 64), pc = __shfl(p_cls, fr, 64);
            if (!qv) dc = pc = 0.f;
            float* pw = a.ws + ((long long)p * a.H + h) * 3 * HD;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v4[4] = {dc * (float)q0[e], dc * (float)q1[e], pc * (float)g0[e], pc * (float)g1[e]};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v4[k] += dpp_mov_f<0xB1>(v4[k]);
                    v4[k] += dpp_mov_f<0x4E>(v4[k]);
                    v4[k] += dpp_mov_f<0x141>(v4[k]);
                    v4[k] += dpp_mov_f<0x140>(v4[k]);
                }
                if (fr == 0) {
                    pw[HD + fg * 8 + e] = v4[0] * a.scale;
                    pw[HD + 32 + fg * 8 + e] = v4[1] * a.scale;
                    pw[2 * HD + fg * 8 + e] = v4[2];
                    pw[2 * HD + 32 + fg * 8 + e] = v4[3];
                }
            }
        }
        if (qv && a.nsplit > 1) {
            const long long nrows = (long long)gridDim.y * a.q.n;
            const long long orow = (long long)p * a.q.n + q;
            float* dst = a.ws + (((long long)split * nrows + orow) * a.H + h) * HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4_t*>(dst + dt * 16 + fg * 4) = o[dt] * a.scale;
        } else if (qv) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                st_bf16x4(dQ + qrow * a.lddq + hdq + dt * 16 + fg * 4, o[dt][0] * a.scale, o[dt][1] * a.scale, o[dt][2] * a.scale,
                          o[dt][3] * a.scale);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, key-owned: dK, dV.  own = keys (no extra), other = queries [extra CLS query ; row set], NT query tiles.
// ------------------------------------------------------------------------------------------------
template <int NT, int NW, int FL, int NL = 0>
__global__ __launch_bounds__(64 * NW) void attn_dkv_mfma_kernel(const AttnArgs a, int tiles_per_wg) {
    constexpr int VP = vt_pitch(NT);
    constexpr bool MASK = (FL & 1) != 0, DROP = (FL & 2) != 0;
    const float sc2 = a.scale * LOG2E;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sQt = smem;                       // [64][VP]  Q transposed
    unsigned char* sGt = sQt + 64 * VP;              // [64][VP]  dO transposed
    float* sL = reinterpret_cast<float*>(sGt + 64 * VP);   // [NT*16] lse
    float* sD = sL + NT * 16;                        // [NT*16] delta

    const int tid = threadIdx.x, lane = tid & 63, w = wave_id();
    const int fr = lane & 15, fg = lane >> 4;
    const int p = blockIdx.y, b = p / a.G, g = p % a.G, h = blockIdx.z;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.Q);
    const bf16_t* K = reinterpret_cast<const bf16_t*>(a.K);
    const bf16_t* V = reinterpret_cast<const bf16_t*>(a.V);
    const bf16_t* dO = reinterpret_cast<const bf16_t*>(a.dO);
    bf16_t* dK = reinterpret_cast<bf16_t*>(a.dK);
    bf16_t* dV = reinterpret_cast<bf16_t*>(a.dV);
    const int hq = a.qoff + h * HD, hk = a.koff + h * HD, hv = a.voff + h * HD, ho = a.ooff + h * HD;
    const int hdk = a.dkoff + h * HD, hdv = a.dvoff + h * HD;
    // queries incl. the extra CLS query; with nsplit > 1 this workgroup covers the chunk [r0, r0 + ntot) of them
    const int nall = a.q.n + a.extra;
    const int split = blockIdx.x % a.nsplit;
    const int per = a.nsplit > 1 ? ((((nall + a.nsplit - 1) / a.nsplit) + 15) & ~15) : nall;
    const int r0 = split * per;
    const int ntot = max(0, min(per, nall - r0));

    stage_rows_t<NT, 64 * NW>(sQt, Q, a.ldq, hq, a, a.q, b, g, ntot, tid, r0);
    stage_rows_t<NT, 64 * NW>(sGt, dO, a.ldo, ho, a, a.q, b, g, ntot, tid, r0);
    // cls_out (as in attn_dq_mfma_kernel): the extra QUERY's gradient over this group's keys, dQ = sum_k dS[cls, k] K[k], is left
    // as an fp32 partial (slot 0 of ws[problem][head][3][64]); its delta is computed here (nobody else has to provide it)
    const bool cls_out = NT == 2 && NW == 1 && !MASK && !DROP && a.nsplit == 1 && a.ws != nullptr && a.extra;
    for (int i = tid; i < NT * 16; i += 64 * NW) {
        float l = INFINITY, d = 0.f;                 // padded query rows: exp(s - inf) = 0
        if (i < ntot) {
            const long long row = other_row(a, a.q, b, g, r0 + i);
            l = a.lse[row * a.H + h] * LOG2E;          // log2 domain, like the scores
            if (cls_out && i == 0) {
                const bf16_t* Oc = reinterpret_cast<const bf16_t*>(a.O);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const bf16x8_t x = *reinterpret_cast<const bf16x8_t*>(dO + row * a.ldo + ho + c * 8);
                    const bf16x8_t y = *reinterpret_cast<const bf16x8_t*>(Oc + row * a.ldo + ho + c * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) d += (float)x[e] * (float)y[e];
                }
            } else {
                d = a.delta[row * a.H + h];
            }
        }
        sL[i] = l;
        sD[i] = d;
    }
    __syncthreads();
    float ds_cls = 0.f;

    const int nkt = (a.k.n + 15) >> 4;
    const int t0 = (blockIdx.x / a.nsplit) * tiles_per_wg;
    const int t1 = min(nkt, t0 + tiles_per_wg);
    for (int kt = t0 + w; kt < t1; kt += NW) {
        const int key = kt * 16 + fr;
        const bool kv = key < a.k.n;
        const long long krow = kv ? rs_row(a.k, b, g, key) : 0;
        const bf16x8_t k0 = ld_frag_global(K + krow * a.ldk + hk + fg * 8, kv);
        const bf16x8_t k1 = ld_frag_global(K + krow * a.ldk + hk + 32 + fg * 8, kv);
        const bf16x8_t v0 = ld_frag_global(V + krow * a.ldv + hv + fg * 8, kv);
        const bf16x8_t v1 = ld_frag_global(V + krow * a.ldv + hv + 32 + fg * 8, kv);
        float mk = 0.f;
        if (MASK) {
            if (a.mask && kv) mk = a.mask[(long long)b * a.mask_ld + key] * LOG2E;
        }
        f32x4_t ov[4], ok[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) ov[dt] = ok[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < NT / 2; ++kk) {
            f32x4_t pr[2], dr[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = 2 * kk + u;
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                if (tile_live<NL>(t, ntot)) {                                  // uniform: dead query tiles contribute nothing
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag_tr<NT>(sQt, t * 16, 0, fr, fg), k0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag_tr<NT>(sQt, t * 16, 1, fr, fg), k1, acc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag_tr<NT>(sGt, t * 16, 0, fr, fg), v0, dp, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag_tr<NT>(sGt, t * 16, 1, fr, fg), v1, dp, 0, 0, 0);
                    const f32x4_t lse = *reinterpret_cast<const f32x4_t*>(sL + t * 16 + fg * 4);
                    const f32x4_t dl = *reinterpret_cast<const f32x4_t*>(sD + t * 16 + fg * 4);
                    // columns of padding keys (kv false) hold finite garbage; they are never stored.  Padding query rows of
                    // the partial tile have lse = +inf in sL, i.e. p = 0.
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pj = exp2_fast(acc[r] * sc2 + mk - lse[r]);
                        float mu = 1.0f;
                        if (DROP) {
                            if (a.drop_p > 0.f) mu = drop_mult(a, (long long)p * a.q.n + r0 + t * 16 + fg * 4 + r, key, h);
                        }
                        acc[r] = DROP ? pj * mu : pj;
                        dp[r] = pj * ((DROP ? dp[r] * mu : dp[r]) - dl[r]);
                    }
                    if (t == 0) ds_cls = dp[0];                                // query 0 = the extra query (lanes fg == 0)
                }
                pr[u] = acc;
                dr[u] = dp;
            }
            if (pair_live<NL>(kk, ntot)) {                                     // dV^T += dO^T P, dK^T += Q^T dS for this pair of query tiles
                const bf16x8_t pf = pack8(pr[0], pr[1]);
                const bf16x8_t df = pack8(dr[0], dr[1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    ov[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag_t<NT>(sGt, dt * 16 + fr, kk, fg), pf, ov[dt], 0, 0, 0);
                    ok[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag_t<NT>(sQt, dt * 16 + fr, kk, fg), df, ok[dt], 0, 0, 0);
                }
            }
            EGV_KK_BARRIER
        }
        if (cls_out) {
            float dc = __shfl(ds_cls, fr, 64);
            if (!kv) dc = 0.f;
            float* pw = a.ws + ((long long)p * a.H + h) * 3 * HD;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v2[2] = {dc * (float)k0[e], dc * (float)k1[e]};
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    v2[k] += dpp_mov_f<0xB1>(v2[k]);
                    v2[k] += dpp_mov_f<0x4E>(v2[k]);
                    v2[k] += dpp_mov_f<0x141>(v2[k]);
                    v2[k] += dpp_mov_f<0x140>(v2[k]);
                }
                if (fr == 0) {
                    pw[fg * 8 + e] = v2[0] * a.scale;
                    pw[32 + fg * 8 + e] = v2[1] * a.scale;
                }
            }
        }
        if (kv && a.nsplit > 1) {
            // fp32 partials: ws[split][P * k.n own rows][H][2][64]  (summed by attn_dkv_reduce_kernel)
            const long long nrows = (long long)gridDim.y * a.k.n;
            const long long orow = (long long)p * a.k.n + key;
            float* dst = a.ws + (((long long)split * nrows + orow) * a.H + h) * 2 * HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                *reinterpret_cast<f32x4_t*>(dst + dt * 16 + fg * 4) = ok[dt] * a.scale;
                *reinterpret_cast<f32x4_t*>(dst + HD + dt * 16 + fg * 4) = ov[dt];
            }
        } else if (kv) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                st_bf16x4(dV + krow * a.lddv + hdv + dt * 16 + fg * 4, ov[dt][0], ov[dt][1], ov[dt][2], ov[dt][3]);
                st_bf16x4(dK + krow * a.lddk + hdk + dt * 16 + fg * 4, ok[dt][0] * a.scale, ok[dt][1] * a.scale, ok[dt][2] * a.scale,
                          ok[dt][3] * a.scale);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, fused: dQ, dK, dV (and delta) of the divided space attention in ONE pass over the score tiles.
//
// Key-owned like attn_dkv_mfma_kernel: the queries [extra CLS query ; row set] sit transposed in LDS (Q^T, dO^T), every one of
// the 8 waves owns up to two 16-key tiles (K / V fragments from global, kept in registers) -- the row set's key tiles
// round-robin over the waves, plus one more tile that holds only the extra CLS KEY (its dK / dV belong to the one-key launch;
// its column is needed for dQ).  Per pair of query tiles a wave computes S = Q K^T and dP = dO V^T once, P and dS once, and uses
// them three times:
//   dV^T += dO^T P, dK^T += Q^T dS      (reduction over the queries = the C rows: the C layout IS the B operand, as above)
//   dQ^T += K^T dS^T                     (reduction over the wave's 32 keys = the C columns: the dS tile goes through a
//                                         per-wave LDS scratch [key][query] and comes back with the transposing read)
// dQ is summed over the waves in an fp32 LDS accumulator [query pair][32][64] WITHOUT atomics: in step s wave w works on query
// pair (s + w) mod 8, so no two waves touch the same pair in a step, a barrier separates the steps, and the order in which the
// waves add into a pair is fixed -- the result does not depend on scheduling.  One bf16 store pass at the end.
// delta = rowsum(dO o O) of every query is computed while staging; the row-set queries' values are stored for the CLS-key launch
// that follows (the CLS query's own delta is stored by the one-query launch, which may run beside this kernel).
// Against the dQ + dK/dV kernel pair: one staging of Q / dO instead of two stagings of two operands each, 10 instead of 14
// MFMAs and one exp instead of two per 16 x 16 score tile.
// ------------------------------------------------------------------------------------------------
constexpr int DSP = 48;                          // pitch (bytes) of a dS^T scratch row: 16 queries (bf16) + 16 B pad
constexpr int XP = 68;                           // float pitch of a dQ accumulator row (64 head dims + pad)
constexpr int XPAIR = 32 * XP * 4;               // accumulator bytes of one query pair
constexpr int FNW = 8;                           // waves of the fused kernel

// transposing fragment read from a tile T[a][b] (b contiguous, `pitch` bytes per a): lane (fr, fg) receives
// T[a0 + fg*8 .. +7][b0 + fr]
__device__ __forceinline__ bf16x8_t ld_tr_ab(const unsigned char* s, int pitch, int a0, int b0, int fr, int fg) {
    const unsigned char* p = s + (a0 + fg * 8 + (fr >> 2)) * pitch + (b0 + (fr & 3) * 4) * 2;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + 4 * pitch));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
}

// workgroup barrier that orders LDS traffic only (global stores stay in flight, unlike __syncthreads)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NT>
__global__ __launch_bounds__(64 * FNW) void attn_bwd_fused_kernel(const AttnArgs a) {
    constexpr int VP = vt_pitch(NT);
    constexpr int NW = FNW, NTHR = 64 * NW, NP = NT / 2;
    static_assert(NP <= NW, "one query pair per wave and step");
    const float sc2 = a.scale * LOG2E;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sQt = smem;                       // [64][VP]  Q transposed
    unsigned char* sGt = sQt + 64 * VP;              // [64][VP]  dO transposed
    float* sL = reinterpret_cast<float*>(sGt + 64 * VP);   // [NT*16] lse (log2 domain)
    float* sD = sL + NT * 16;                        // [NT*16] delta
    unsigned char* sDS = reinterpret_cast<unsigned char*>(sD + NT * 16);      // [NW][2 query tiles][32 keys][DSP]
    unsigned char* sAcc = sDS + NW * 2 * 32 * DSP;   // [NP][32][XP] fp32 dQ accumulators; first the K rows of the own tiles ([NW][32][RP])
    static_assert(NW * 32 * RP <= NP * XPAIR, "K scratch fits the accumulator");

    const int tid = threadIdx.x, lane = tid & 63, w = wave_id();
    const int fr = lane & 15, fg = lane >> 4;
    const int p = blockIdx.y, b = p / a.G, g = p % a.G, h = blockIdx.z;
    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.Q);
    const bf16_t* K = reinterpret_cast<const bf16_t*>(a.K);
    const bf16_t* V = reinterpret_cast<const bf16_t*>(a.V);
    const bf16_t* O = reinterpret_cast<const bf16_t*>(a.O);
    const bf16_t* dO = reinterpret_cast<const bf16_t*>(a.dO);
    bf16_t* dQ = reinterpret_cast<bf16_t*>(a.dQ);
    bf16_t* dK = reinterpret_cast<bf16_t*>(a.dK);
    bf16_t* dV = reinterpret_cast<bf16_t*>(a.dV);
    const int hq = a.qoff + h * HD, hk = a.koff + h * HD, hv = a.voff + h * HD, ho = a.ooff + h * HD;
    const int hdq = a.dqoff + h * HD, hdk = a.dkoff + h * HD, hdv = a.dvoff + h * HD;
    const int ntot = a.q.n + a.extra;               // queries incl. the extra CLS query (row 0)
    const bool cls_out = a.ws != nullptr && a.extra;     // also produce the extra row's gradients (as per-group partials)

    stage_rows_t<NT, NTHR>(sQt, Q, a.ldq, hq, a, a.q, b, g, ntot, tid);
    stage_rows_t<NT, NTHR>(sGt, dO, a.ldo, ho, a, a.q, b, g, ntot, tid);
    for (int i = tid; i < NT * 16; i += NTHR) {
        float l = INFINITY, d = 0.f;                 // padded query rows: exp(s - inf) = 0
        if (i < ntot) {
            const long long row = other_row(a, a.q, b, g, i);
            l = a.lse[row * a.H + h] * LOG2E;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const bf16x8_t x = *reinterpret_cast<const bf16x8_t*>(dO + row * a.ldo + ho + c * 8);
                const bf16x8_t y = *reinterpret_cast<const bf16x8_t*>(O + row * a.ldo + ho + c * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) d += (float)x[e] * (float)y[e];
            }
            if (!(a.extra && i == 0)) a.delta[row * a.H + h] = d;     // the CLS query's delta is stored by its own (one-query) launch
        }
        sL[i] = l;
        sD[i] = d;
    }

    // own key tiles: j = w, w + NW; j < nkt: row-set keys, j == nkt (with extra): the CLS key alone
    const int nkt = (a.k.n + 15) >> 4;
    bf16x8_t kf[2][2], vf[2][2];
    bool live[2], kval[2], part[2];
    long long krow[2];
    unsigned char* ksc = sAcc + w * 32 * RP;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int j = w + o * NW;
        const bool patch = j < nkt;
        live[o] = j < nkt + a.extra;
        const int key = j * 16 + fr;
        kval[o] = patch ? key < a.k.n : (live[o] && fr == 0);
        part[o] = !patch || (j * 16 + 16 > a.k.n);
        krow[o] = patch ? rs_row(a.k, b, g, kval[o] ? key : 0) : ((long long)b * a.extra_bs + a.extra_row);
        kf[o][0] = ld_frag_global(K + krow[o] * a.ldk + hk + fg * 8, kval[o]);
        kf[o][1] = ld_frag_global(K + krow[o] * a.ldk + hk + 32 + fg * 8, kval[o]);
        vf[o][0] = ld_frag_global(V + krow[o] * a.ldv + hv + fg * 8, kval[o]);
        vf[o][1] = ld_frag_global(V + krow[o] * a.ldv + hv + 32 + fg * 8, kval[o]);
        *reinterpret_cast<bf16x8_t*>(ksc + (o * 16 + fr) * RP + fg * 16) = kf[o][0];
        *reinterpret_cast<bf16x8_t*>(ksc + (o * 16 + fr) * RP + 64 + fg * 16) = kf[o][1];
    }
    __syncthreads();
    bf16x8_t kT[4];                                  // K^T fragments of the wave's 32 keys: A operand of dQ^T
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) kT[dt] = ld_tr_ab(ksc, RP, 0, dt * 16, fr, fg);
    lds_barrier();                                   // the scratch becomes the dQ accumulator

    f32x4_t ov[2][4], ok[2][4];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) ov[o][dt] = ok[o][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    unsigned char* dsc = sDS + w * 2 * 32 * DSP;
    const int npair = (ntot + 31) >> 5;              // live query pairs

#pragma unroll 1
    for (int step = 0; step < NW; ++step) {
        const int kk = (step + w) & (NW - 1);
        if (kk < npair) {                            // uniform per wave
            f32x4_t pr[2][2], dr[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = 2 * kk + u;
                const bool tl = t * 16 < ntot;
                bf16x8_t qa0, qa1, ga0, ga1;
                f32x4_t lse, dl;
                if (tl) {
                    qa0 = ld_frag_tr<NT>(sQt, t * 16, 0, fr, fg);
                    qa1 = ld_frag_tr<NT>(sQt, t * 16, 1, fr, fg);
                    ga0 = ld_frag_tr<NT>(sGt, t * 16, 0, fr, fg);
                    ga1 = ld_frag_tr<NT>(sGt, t * 16, 1, fr, fg);
                    lse = *reinterpret_cast<const f32x4_t*>(sL + t * 16 + fg * 4);
                    dl = *reinterpret_cast<const f32x4_t*>(sD + t * 16 + fg * 4);
                }
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    f32x4_t acc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    if (tl && live[o]) {                                               // uniform per wave
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa0, kf[o][0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa1, kf[o][1], acc, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga0, vf[o][0], dp, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga1, vf[o][1], dp, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pj = exp2_fast(acc[r] * sc2 - lse[r]);       // padding query rows: lse = +inf -> 0
                            acc[r] = pj;
                            dp[r] = pj * (dp[r] - dl[r]);
                        }
                        if (part[o]) {                                                 // padding key columns must not reach dQ
                            asm volatile("" ::: "memory");
#pragma unroll
                            for (int r = 0; r < 4; ++r) { acc[r] = kval[o] ? acc[r] : 0.f; dp[r] = kval[o] ? dp[r] : 0.f; }
                        }
                    }
                    if (cls_out && g != 0 && kk == 0 && u == 0 && (w + o * NW == nkt) && lane == 0) {
                        acc[0] = 0.f;                      // (CLS query, CLS key) belongs to every group: counted in group 0 only
                        dp[0] = 0.f;
                    }
                    pr[u][o] = acc;
                    dr[u][o] = dp;
                    // dS^T scratch: [key o*16 + fr][queries fg*4 .. +3]
                    u32x2_t pk = {pack2(dp[0], dp[1]), pack2(dp[2], dp[3])};
                    *reinterpret_cast<u32x2_t*>(dsc + u * 32 * DSP + (o * 16 + fr) * DSP + fg * 8) = pk;
                }
            }
            // dV^T += dO^T P, dK^T += Q^T dS over this pair of query tiles
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8_t gt = ld_frag_t<NT>(sGt, dt * 16 + fr, kk, fg);
                const bf16x8_t qt = ld_frag_t<NT>(sQt, dt * 16 + fr, kk, fg);
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    if (live[o]) {
                        ov[o][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt, pack8(pr[0][o], pr[1][o]), ov[o][dt], 0, 0, 0);
                        ok[o][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, pack8(dr[0][o], dr[1][o]), ok[o][dt], 0, 0, 0);
                    }
                }
            }
            // dQ^T of this query pair += K^T dS^T over this wave's 32 keys (step 0: the wave is the pair's first writer)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float* accp = reinterpret_cast<float*>(sAcc + kk * XPAIR);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bf16x8_t dst = ld_tr_ab(dsc + u * 32 * DSP, DSP, 0, 0, fr, fg);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    float* ap = accp + (u * 16 + fr) * XP + dt * 16 + fg * 4;
                    f32x4_t c = {0.f, 0.f, 0.f, 0.f};
                    if (step != 0) c = *reinterpret_cast<const f32x4_t*>(ap);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kT[dt], dst, c, 0, 0, 0);
                    *reinterpret_cast<f32x4_t*>(ap) = c;
                }
            }
        }
        lds_barrier();
    }

    // the extra (CLS) row's share of this group: dQ of the CLS query over this group's keys, dK / dV of the CLS key under this
    // group's queries -> fp32 partials [problem][head][3][64], summed over the groups by attn_cls_reduce_kernel
    if (cls_out) {
        float* pw = a.ws + ((long long)p * a.H + h) * 3 * HD;
        if (tid < 16) {
            const f32x4_t v = *reinterpret_cast<const f32x4_t*>(sAcc + (tid * 4) * 4);
            *reinterpret_cast<f32x4_t*>(pw + tid * 4) = v * a.scale;
        }
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            if (w + o * NW == nkt && fr == 0) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    *reinterpret_cast<f32x4_t*>(pw + HD + dt * 16 + fg * 4) = ok[o][dt] * a.scale;
                    *reinterpret_cast<f32x4_t*>(pw + 2 * HD + dt * 16 + fg * 4) = ov[o][dt];
                }
            }
        }
    }
    // dQ: accumulators -> bf16 rows (16 threads per query, 4 head dims each)
    for (int e = tid; e < npair * 32 * 16; e += NTHR) {
        const int i = e >> 4, d4 = (e & 15) * 4;
        if (i < ntot && !(a.extra && i == 0)) {
            const f32x4_t v = *reinterpret_cast<const f32x4_t*>(sAcc + (i >> 5) * XPAIR + ((i & 31) * XP + d4) * 4);
            const long long row = rs_row(a.q, b, g, i - a.extra);
            st_bf16x4(dQ + row * a.lddq + hdq + d4, v[0] * a.scale, v[1] * a.scale, v[2] * a.scale, v[3] * a.scale);
        }
    }
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int j = w + o * NW;
        if (j < nkt && kval[o]) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                st_bf16x4(dV + krow[o] * a.lddv + hdv + dt * 16 + fg * 4, ov[o][dt][0], ov[o][dt][1], ov[o][dt][2], ov[o][dt][3]);
                st_bf16x4(dK + krow[o] * a.lddk + hdk + dt * 16 + fg * 4, ok[o][dt][0] * a.scale, ok[o][dt][1] * a.scale,
                          ok[o][dt][2] * a.scale, ok[o][dt][3] * a.scale);
            }
        }
    }
}

// dQ / dK / dV of the extra (CLS) row = sum over the G groups of attn_bwd_fused_kernel's partials, in group order
// self_term: the partials hold neither side of the (extra query, extra key) pair (dQ + dK/dV kernel pair: the extra row is
// never an OWN row there) -- its three contributions are added here from the row itself
__global__ __launch_bounds__(256) void attn_cls_reduce_kernel(const AttnArgs a, int self_term) {
    __shared__ float red[4][3][HD];
    const int b = blockIdx.x, h = blockIdx.y, d = threadIdx.x & 63, w = threadIdx.x >> 6;
    // four waves take the groups round-robin (196 groups for the time attention: a single wave's loop was 100 us of load
    // latency), eight loads in flight each; combined in wave order -- the sum does not depend on scheduling
    float s[3] = {0.f, 0.f, 0.f};
    const float* base = a.ws + ((long long)b * a.G * a.H + h) * 3 * HD + d;
    const long long gstride = (long long)a.H * 3 * HD;
    int g = w;
    for (; g + 28 < a.G; g += 32) {
        float v[8][3];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < 3; ++k) v[u][k] = base[(g + 4 * u) * gstride + k * HD];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < 3; ++k) s[k] += v[u][k];
    }
    for (; g < a.G; g += 4)
#pragma unroll
        for (int k = 0; k < 3; ++k) s[k] += base[g * gstride + k * HD];
#pragma unroll
    for (int k = 0; k < 3; ++k) red[w][k][d] = s[k];
    __syncthreads();
    if (w != 0) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) s[k] = (red[0][k][d] + red[1][k][d]) + (red[2][k][d] + red[3][k][d]);
    const long long row = (long long)b * a.extra_bs + a.extra_row;
    if (self_term) {
        const float qc = bf2f(reinterpret_cast<const bf16_t*>(a.Q)[row * a.ldq + a.qoff + h * HD + d].v);
        const float kc = bf2f(reinterpret_cast<const bf16_t*>(a.K)[row * a.ldk + a.koff + h * HD + d].v);
        const float vc = bf2f(reinterpret_cast<const bf16_t*>(a.V)[row * a.ldv + a.voff + h * HD + d].v);
        const float gc = bf2f(reinterpret_cast<const bf16_t*>(a.dO)[row * a.ldo + a.ooff + h * HD + d].v);
        const float oc = bf2f(reinterpret_cast<const bf16_t*>(a.O)[row * a.ldo + a.ooff + h * HD + d].v);
        const float sc = wave_sum_dpp(qc * kc) * a.scale, dp = wave_sum_dpp(gc * vc), dl = wave_sum_dpp(gc * oc);
        const float pj = __expf(sc - a.lse[row * a.H + h]);
        const float ds = pj * (dp - dl);
        s[0] += ds * kc * a.scale;
        s[1] += ds * qc * a.scale;
        s[2] += pj * gc;
    }
    reinterpret_cast<bf16_t*>(a.dQ)[row * a.lddq + a.dqoff + h * HD + d].v = f2bf(s[0]);
    reinterpret_cast<bf16_t*>(a.dK)[row * a.lddk + a.dkoff + h * HD + d].v = f2bf(s[1]);
    reinterpret_cast<bf16_t*>(a.dV)[row * a.lddv + a.dvoff + h * HD + d].v = f2bf(s[2]);
}

template <int NT> constexpr size_t fused_lds() {
    return (size_t)2 * 64 * vt_pitch(NT) + 2 * NT * 16 * 4 + (size_t)FNW * 2 * 32 * DSP + (size_t)(NT / 2) * XPAIR;
}

template <int NT> constexpr size_t fwd_lds() { return (size_t)NT * 16 * RP + 64 * vt_pitch(NT); }
template <int NT> constexpr size_t dq_lds() { return (size_t)2 * 64 * vt_pitch(NT); }
template <int NT> constexpr size_t dkv_lds() { return (size_t)2 * 64 * vt_pitch(NT) + 2 * NT * 16 * 4; }

template <typename KFn>
static void set_lds(KFn k, size_t bytes) {
    if (bytes > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace egv
using namespace egv;

static bool aligned_ok(const AttnArgs& a) {
    auto ok8 = [](int x) { return (x % 8) == 0; };
    return ok8(a.ldq) && ok8(a.ldk) && ok8(a.ldv) && ok8(a.ldo) && ok8(a.qoff) && ok8(a.koff) && ok8(a.voff) && ok8(a.ooff);
}

// own tiles per workgroup: all of them for small problems, chunks of 16 for long own sides (cross attention)
static inline void own_split(int n_own, int& nw, int& tpw, int& chunks) {
    const int tiles = (n_own + 15) / 16;
    nw = tiles >= 4 ? 4 : (tiles >= 2 ? 2 : 1);
    tpw = tiles <= 16 ? tiles : 16;
    chunks = (tiles + tpw - 1) / tpw;
}

#define EGV_MFMA_LAUNCH(KERNEL, LDSFN, NTV, NLV)                                                                 \
    do {                                                                                                     \
        const size_t lds = LDSFN<NTV>();                                                                     \
        dim3 grid(chunks, B * a.G, a.H);                                                                     \
        if (a.mask || a.drop_p > 0.f) {                                                                      \
            if (nw == 4) { set_lds(KERNEL<NTV, 4, 3, NLV>, lds); hipLaunchKernelGGL((KERNEL<NTV, 4, 3, NLV>), grid, dim3(256), lds, st, a, tpw); } \
            else if (nw == 2) { set_lds(KERNEL<NTV, 2, 3, NLV>, lds); hipLaunchKernelGGL((KERNEL<NTV, 2, 3, NLV>), grid, dim3(128), lds, st, a, tpw); } \
            else { set_lds(KERNEL<NTV, 1, 3, NLV>, lds); hipLaunchKernelGGL((KERNEL<NTV, 1, 3, NLV>), grid, dim3(64), lds, st, a, tpw); } \
        } else {                                                                                             \
            if (nw == 4) { set_lds(KERNEL<NTV, 4, 0, NLV>, lds); hipLaunchKernelGGL((KERNEL<NTV, 4, 0, NLV>), grid, dim3(256), lds, st, a, tpw); } \
            else if (nw == 2) { set_lds(KERNEL<NTV, 2, 0, NLV>, lds); hipLaunchKernelGGL((KERNEL<NTV, 2, 0, NLV>), grid, dim3(128), lds, st, a, tpw); } \
            else { set_lds(KERNEL<NTV, 1, 0, NLV>, lds); hipLaunchKernelGGL((KERNEL<NTV, 1, 0, NLV>), grid, dim3(64), lds, st, a, tpw); } \
        }                                                                                                    \
    } while (0)

// the unsplit launch also treats the extra row as a QUERY of every group (partial softmax states in a.ws, one per group): the
// groups must partition the keys the extra query sees (divided attention: query row set == key row set), long key side only
bool egv_attn_fwd_cls_ok(const AttnArgs& a) {
    const int nall = a.k.n + a.extra;
    const bool same = a.q.bs == a.k.bs && a.q.base == a.k.base && a.q.gs == a.k.gs && a.q.is == a.k.is && a.q.n == a.k.n;
    return aligned_ok(a) && a.nsplit == 1 && a.ws && a.extra && a.extra_row == 0 && !a.mask && a.drop_p <= 0.f && same && nall > 64 &&
           nall <= 224 && (a.q.n + 15) / 16 <= 15;
}

int egv_attn_fwd_mfma(const AttnArgs& ain, int B, hipStream_t st) {
    AttnArgs a = ain;
    const bool cls = egv_attn_fwd_cls_ok(a);
    if (!cls && a.nsplit == 1) a.ws = nullptr;                     // the kernel keys the extra-query path on ws
    const int nall = a.k.n + a.extra;
    const int ntot = a.nsplit > 1 ? ((((nall + a.nsplit - 1) / a.nsplit) + 15) & ~15) : nall;
    if (!aligned_ok(a) || ntot > 288) return 0;
    if (a.nsplit > 1 && (a.q.n > 64 || !a.ws)) return 0;          // split form: short query side only (text -> image)
    int nw, tpw, chunks;
    own_split(a.q.n, nw, tpw, chunks);
    chunks *= a.nsplit;
    if (ntot > 224) EGV_MFMA_LAUNCH(attn_fwd_mfma_kernel, fwd_lds, 18, 0);      // 256 patches + CLS (14 x 14 patches of a 224^2 frame)
    else if (ntot <= 32) EGV_MFMA_LAUNCH(attn_fwd_mfma_kernel, fwd_lds, 2, 0);
    else if (ntot <= 64) EGV_MFMA_LAUNCH(attn_fwd_mfma_kernel, fwd_lds, 4, 0);
    else if (ntot > 192 && ntot <= 208 && a.nsplit == 1) { if (cls) ++tpw; EGV_MFMA_LAUNCH(attn_fwd_mfma_kernel, fwd_lds, 14, 13); }     // 196 patches + CLS: 13 live tiles known at compile time (-20 % on the forward; measured slower on the two backward kernels)
    else { if (cls) ++tpw; EGV_MFMA_LAUNCH(attn_fwd_mfma_kernel, fwd_lds, 14, 0); }
    return cls ? 2 : 1;
}

// the dQ + dK/dV kernel pair leaves the extra row's gradients as per-group partials in a.ws (one-wave, one-tile groups: the
// 17-row time attention) when the launch is unsplit, bf16, without mask / dropout, and the extra row is row 0 of the sample
bool egv_attn_bwd_pair_cls_ok(const AttnArgs& a) {
    return aligned_ok(a) && a.nsplit == 1 && a.ws && a.extra && a.extra_row == 0 && !a.mask && a.drop_p <= 0.f && a.k.n + a.extra <= 32 &&
           a.q.n <= 16 && a.k.n <= 16 && a.O && a.dO;
}
void egv_attn_bwd_cls_reduce_launch(const AttnArgs& a, int B, int self_term, hipStream_t st) {
    hipLaunchKernelGGL(attn_cls_reduce_kernel, dim3(B, a.H), dim3(256), 0, st, a, self_term);
}

int egv_attn_dq_mfma(const AttnArgs& ain, int B, hipStream_t st) {
    AttnArgs a = ain;
    if (a.nsplit == 1 && !egv_attn_bwd_pair_cls_ok(a)) a.ws = nullptr;
    const int nall = a.k.n + a.extra;
    const int ntot = a.nsplit > 1 ? ((((nall + a.nsplit - 1) / a.nsplit) + 15) & ~15) : nall;
    if (!aligned_ok(a) || (a.lddq % 4) || (a.dqoff % 4) || ntot > 288) return 0;
    if (a.nsplit > 1 && (a.q.n > 64 || !a.ws)) return 0;
    int nw, tpw, chunks;
    own_split(a.q.n, nw, tpw, chunks);
    chunks *= a.nsplit;
    if (ntot > 224) EGV_MFMA_LAUNCH(attn_dq_mfma_kernel, dq_lds, 18, 0);
    else if (ntot <= 32) EGV_MFMA_LAUNCH(attn_dq_mfma_kernel, dq_lds, 2, 0);
    else if (ntot <= 64) EGV_MFMA_LAUNCH(attn_dq_mfma_kernel, dq_lds, 4, 0);
    else EGV_MFMA_LAUNCH(attn_dq_mfma_kernel, dq_lds, 14, 0);
    return 1;
}

int egv_attn_dkv_mfma(const AttnArgs& ain, int B, hipStream_t st) {
    AttnArgs a = ain;
    if (a.nsplit == 1 && !egv_attn_bwd_pair_cls_ok(a)) a.ws = nullptr;
    const int nall = a.q.n + a.extra;
    const int ntot = a.nsplit > 1 ? ((((nall + a.nsplit - 1) / a.nsplit) + 15) & ~15) : nall;
    if (!aligned_ok(a) || (a.lddk % 4) || (a.lddv % 4) || (a.dkoff % 4) || (a.dvoff % 4) || ntot > 288) return 0;
    if (a.nsplit > 1 && (a.k.n > 64 || !a.ws)) return 0;          // split form: short key side only
    int nw, tpw, chunks;
    own_split(a.k.n, nw, tpw, chunks);
    chunks *= a.nsplit;
    if (ntot > 224) EGV_MFMA_LAUNCH(attn_dkv_mfma_kernel, dkv_lds, 18, 0);
    else if (ntot <= 32) EGV_MFMA_LAUNCH(attn_dkv_mfma_kernel, dkv_lds, 2, 0);
    else if (ntot <= 64) EGV_MFMA_LAUNCH(attn_dkv_mfma_kernel, dkv_lds, 4, 0);
    else EGV_MFMA_LAUNCH(attn_dkv_mfma_kernel, dkv_lds, 14, 0);
    return 1;
}

// dQ + dK/dV of a divided-attention launch in one kernel (no mask, no dropout, no split); 1 if enqueued.  Only the long
// other side (space attention: 196 patches + CLS) -- for the 17-row time attention a one-wave fused kernel measured no faster
// than the pair (both are bound by the global round trips of one wave, and the pair runs at twice the occupancy).
int egv_attn_bwd_fused_mfma(const AttnArgs& a, int B, hipStream_t st) {
    // groups of at most one 16-row tile (the 17-row time attention): the one-wave-per-group kernel of egv_attn_time.hip, which
    // leaves the CLS row's gradients as per-group partials; the (CLS, CLS) term is added by the reduction
    static const bool time_fused = egv_cfg_on("EGV_ATTN_TIME_FUSED", true);
    if (time_fused && a.q.n <= 16 && a.ws && a.extra && egv_attn_time_bwd(a, B, st)) {
        hipLaunchKernelGGL(attn_cls_reduce_kernel, dim3(B, a.H), dim3(256), 0, st, a, 1);
        return 1;
    }
    // long groups (space attention): two-phase kernel on row-major LDS images (egv_attn_space.hip)
    if (a.delta && a.lse && egv_attn_space_bwd(a, B, st)) {
        if (a.ws && a.extra) hipLaunchKernelGGL(attn_cls_reduce_kernel, dim3(B, a.H), dim3(256), 0, st, a, 0);
        return 1;
    }
    const int ntot = a.q.n + a.extra;
    const int own = (a.k.n + 15) / 16 + a.extra;
    if (!aligned_ok(a) || (a.lddq % 4) || (a.dqoff % 4) || (a.lddk % 4) || (a.lddv % 4) || (a.dkoff % 4) || (a.dvoff % 4)) return 0;
    if (a.mask || a.drop_p > 0.f || a.nsplit > 1 || ntot > 224 || ntot <= 64 || own > 2 * FNW || !a.delta || !a.lse) return 0;
    if (a.ws && a.extra && a.extra_row != 0) return 0;            // partials of the extra row: written for the CLS-first layout only
    constexpr size_t lds = fused_lds<14>();
    set_lds(attn_bwd_fused_kernel<14>, lds);
    hipLaunchKernelGGL((attn_bwd_fused_kernel<14>), dim3(1, B * a.G, a.H), dim3(64 * FNW), lds, st, a);
    if (a.ws && a.extra) hipLaunchKernelGGL(attn_cls_reduce_kernel, dim3(B, a.H), dim3(256), 0, st, a, 0);
    return 1;
}
