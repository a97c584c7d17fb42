// Divided SPACE attention for gfx950 (bf16, head_dim 64), forward and one-launch backward: the patch queries of one frame over
// [the frame's patches ; the CLS key] and the CLS query's share of that frame (VarAttention core, video_transformer.py:121-150,
// "(b h f) n d").  One workgroup of four waves per (sample, frame, head); built like the time-attention kernels of
// egv_attn_time.hip, for a key side of up to NT 16-row tiles:
//   * the other side of a phase sits in LDS as ROW-MAJOR images (128-byte rows, 16-byte chunk c of row r stored at chunk c ^ (r & 7): conflict-free 16-byte row
//     reads AND conflict-free transposing reads; a 16-byte pad per row was 45 % bank-conflict cycles on the transposing reads) copied as stored,
//     16 bytes per lane -- no register transposes; rows >= n + 1 are zeros; the CLS row is row 0 of every image, so it needs no
//     tile, launch or code path of its own: as a key it is one more column, as a query one more row of the first query tile (whose
//     result leaves as a per-group partial instead of an output row; the (CLS, CLS) pair is counted in group 0 only);
//   * A operands of the first products are 16-byte row reads, the transposed A operands of the second products (V^T, K^T, dO^T,
//     Q^T) come from the same images with ds_read_b64_tr_b16; the C layout of a first product is the B operand of the second
//     (reduction index permuted: [tile 2kk rows fg*4..+3 | tile 2kk+1 rows fg*4..+3]);
//   * own-side fragments come straight from global memory in fragment layout through buffer descriptors (32-bit offsets, out-of-range
//     = zero: no predication), outputs leave as 16-byte row pieces through a per-wave LDS scratch;
//   * backward = two phases over the same LDS: keys staged -> every wave's query tiles (dQ), then queries staged -> every wave's key
//     tiles (dK, dV); no cross-wave reduction, no accumulator in LDS, two workgroups per CU.
#include "egv_attn.h"
#include <cstdlib>

namespace egv {

namespace {
typedef __attribute__((ext_vector_type(4))) short s_s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s_s16x8_t;
constexpr int RB = 128;                         // row bytes of the LDS images (64 bf16, no padding: XOR-swizzled chunks)
constexpr float S_LOG2E = 1.4426950408889634f, S_LN2 = 0.6931471805599453f;
constexpr unsigned int S_OOB = 0x80000000u;
constexpr int SNW = 8;                          // waves per workgroup

// what the kernels need of an AttnArgs, precomputed on the host (a compact argument block: the kernels are short of scalar registers)
struct SpaceArgs {
    const void* qkv; const void* o; const void* dO; void* dqkv;
    float* lse; float* delta; float* ws;
    int ld, ldo, ldd;                           // leading dimensions (elements) of qkv, O / dO, dqkv
    int qcol, kcol, vcol, ocol, dqcol, dkcol, dvcol;   // first column (elements, from the tensor's base) of head 0
    int H, G, n, nb;
    int bs, base, gs, is, cls_bs;               // row(b, g, i) = b*bs + base + g*gs + i*is; CLS row = b*cls_bs
    float scale;
    unsigned int qkv_bytes, o_bytes, dqkv_bytes, lse_bytes;
};

__device__ __forceinline__ float s_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ bf16x8_t s_bf(u32x4_t v) { return __builtin_bit_cast(bf16x8_t, v); }
__device__ __forceinline__ bf16x8_t s_pack8(const f32x4_t& a, const f32x4_t& b) {
    u32x4_t v = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
    return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ float s_grp_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }
__device__ __forceinline__ float s_grp_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ f32x4_t s_mfma(bf16x8_t a, bf16x8_t b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float s_dot16(u32x4_t a0, u32x4_t a1, u32x4_t b0, u32x4_t b1) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        s = fmaf(__uint_as_float(a0[k] << 16), __uint_as_float(b0[k] << 16), s);
        s = fmaf(__uint_as_float(a0[k] & 0xffff0000u), __uint_as_float(b0[k] & 0xffff0000u), s);
        s = fmaf(__uint_as_float(a1[k] << 16), __uint_as_float(b1[k] << 16), s);
        s = fmaf(__uint_as_float(a1[k] & 0xffff0000u), __uint_as_float(b1[k] & 0xffff0000u), s);
    }
    return s;
}
// per-lane byte offsets inside an image (computed once per kernel)
struct SLane {
    int rf;                                     // row fragment: row fr, chunk fg (chunk fg + 4 sits at rf ^ 64); + t * 2048 per tile
    int tr[4];                                  // transposing read for head dims dt*16 ..: row fg*4 + (fr >> 2), 8 bytes; + t * 2048 per tile
};
__device__ __forceinline__ SLane s_lane(int fr, int fg) {
    SLane L;
    L.rf = fr * RB + ((fg ^ (fr & 7)) << 4);
    const int r = fg * 4 + (fr >> 2), x = r & 7, b = (fr >> 1) & 1;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) L.tr[dt] = r * RB + (((dt * 2 + b) ^ x) << 4) + (fr & 1) * 8;
    return L;
}
// row fragment (A operand of a first product): row t*16 + fr, 16-byte chunks fg and fg + 4
__device__ __forceinline__ void s_rowfrag(const unsigned char* img, int t, const SLane& L, bf16x8_t& lo, bf16x8_t& hi) {
    const unsigned char* p = img + t * 16 * RB;
    lo = *reinterpret_cast<const bf16x8_t*>(p + L.rf);
    hi = *reinterpret_cast<const bf16x8_t*>(p + (L.rf ^ 64));
}
// transposed fragment (A operand of a second product) for head dims dt*16 .. +15 over the row pair (t0, t0 + 1):
// [X^T[d][rows t0*16 + fg*4 .. +3] | X^T[d][rows (t0+1)*16 + fg*4 .. +3]] (second tile past the image: zeros)
__device__ __forceinline__ bf16x8_t s_trfrag(const unsigned char* img, int t0, bool has1, int dt, const SLane& L) {
    const unsigned char* p = img + t0 * 16 * RB + L.tr[dt];
    const s_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s_s16x4_t*)(p));
    s_s16x4_t hi = {0, 0, 0, 0};
    if (has1) hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s_s16x4_t*)(p + 16 * RB));
    const s_s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ void s_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// o[dt] (C layout: row fr, head dims dt*16 + fg*4 .. +3) -> 16-byte row pieces through the wave's scratch image (one 16-row tile)
__device__ __forceinline__ void s_store_rows(unsigned char* scr, __amdgpu_buffer_rsrc_t r, unsigned int off, const f32x4_t (&o)[4], float s,
                                             const SLane& L, int fr, int fg) {
    s_wave_sync();                                                 // the previous use of the scratch is over in every lane
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const u32x2_t pk = {pack_bf16x2(o[dt][0] * s, o[dt][1] * s), pack_bf16x2(o[dt][2] * s, o[dt][3] * s)};
        *reinterpret_cast<u32x2_t*>(scr + fr * RB + (((dt * 2 + (fg >> 1)) ^ (fr & 7)) << 4) + (fg & 1) * 8) = pk;
    }
    s_wave_sync();
    const u32x4_t lo = *reinterpret_cast<const u32x4_t*>(scr + L.rf), hi = *reinterpret_cast<const u32x4_t*>(scr + (L.rf ^ 64));
    __builtin_amdgcn_raw_buffer_store_b128(lo, r, off, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(hi, r, off == S_OOB ? S_OOB : off + 64u, 0, 0);
}

struct SGeom {
    int h, b, g, n, row0, is, cls;
};
__device__ __forceinline__ SGeom s_geom(const SpaceArgs& a) {
    SGeom q;
    const int prob = blockIdx.x;
    q.h = prob % a.H;
    const int pg = prob / a.H;
    q.b = pg / a.G; q.g = pg % a.G;
    q.n = a.n;
    q.row0 = q.b * a.bs + a.base + q.g * a.gs;
    q.is = a.is;
    q.cls = q.b * a.cls_bs;
    return q;
}
// token-matrix row of index i of the group's row list [CLS row ; patch rows 0 .. n-1], -1 past it (the CLS row first: the order in
// which attn_fwd_mfma_kernel sums the keys -- outputs stay bit-identical to that kernel's)
__device__ __forceinline__ int s_row_of(const SGeom& q, int i) { return i == 0 ? q.cls : (i <= q.n ? q.row0 + (i - 1) * q.is : -1); }

// copy rows [0, NT*16) of the group's row list (64 columns from column `col` of a tensor) into an image
template <int NT>
__device__ __forceinline__ void s_stage(unsigned char* img, __amdgpu_buffer_rsrc_t r, int ld, int col, const SGeom& q, int tid) {
    constexpr int CH = NT * 16 * 8;
    constexpr int IT = (CH + 64 * SNW - 1) / (64 * SNW);
    u32x4_t v[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = tid + it * 64 * SNW;
        const int row = s_row_of(q, c >> 3);
        v[it] = __builtin_amdgcn_raw_buffer_load_b128(r, (row >= 0 && c < CH) ? (unsigned int)(row * ld + col + (c & 7) * 8) * 2u : S_OOB, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = tid + it * 64 * SNW;
        const int row = c >> 3;
        if (c < CH) *reinterpret_cast<u32x4_t*>(img + row * RB + (((c & 7) ^ (row & 7)) << 4)) = v[it];
    }
}
// tiles whose scores need a validity select: with EXACT (every one of the NT tiles live, only the last one partial) tile 0 (the CLS
// row) and the last tile; otherwise every tile (run-time row count)
template <int NT, bool EXACT> __device__ __forceinline__ bool s_mask_tile(int t) { return EXACT ? (t == 0 || t == NT - 1) : true; }
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// forward
template <int NT, bool EXACT>
__global__ __launch_bounds__(64 * SNW) void attn_space_fwd_kernel(const SpaceArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int IMG = NT * 16 * RB;
    constexpr int NP = (NT + 1) / 2;
    unsigned char* sK = smem;
    unsigned char* sV = smem + IMG;
    const int tid = threadIdx.x, lane = tid & 63, w = wave_id();
    const int fr = lane & 15, fg = lane >> 4;
    unsigned char* scr = smem + 2 * IMG + w * 16 * RB;
    const SLane L = s_lane(fr, fg);
    const SGeom q = s_geom(a);
    const int nk = q.n + 1;                                        // keys incl. the CLS key (index 0)
    const float sc2 = a.scale * S_LOG2E;
    const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.qkv), 0, (int)a.qkv_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.o), 0, (int)a.o_bytes, 0x00020000);

    // own query tiles of this wave: w, w + SNW, ... (the CLS query is index 0 of the row list); fragments of the first one now
    const int nqt = (nk + 15) >> 4;
    auto qoff = [&](int qt) {
        const int row = s_row_of(q, qt * 16 + fr);
        return (row >= 0 && qt < nqt) ? (unsigned int)(row * a.ld + a.qcol + q.h * HD + fg * 8) * 2u : S_OOB;
    };
    const unsigned int of0 = qoff(w);
    u32x4_t q0 = __builtin_amdgcn_raw_buffer_load_b128(rQ, of0, 0, 0), q1 = __builtin_amdgcn_raw_buffer_load_b128(rQ, of0 == S_OOB ? S_OOB : of0 + 64u, 0, 0);
    s_stage<NT>(sK, rQ, a.ld, a.kcol + q.h * HD, q, tid);
    s_stage<NT>(sV, rQ, a.ld, a.vcol + q.h * HD, q, tid);
    __syncthreads();

    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    for (int qt = w; qt < nqt; qt += SNW) {
        // next tile's fragments first (their latency hides behind this tile)
        const unsigned int on = qoff(qt + SNW);
        const u32x4_t nq0 = __builtin_amdgcn_raw_buffer_load_b128(rQ, on, 0, 0), nq1 = __builtin_amdgcn_raw_buffer_load_b128(rQ, on == S_OOB ? S_OOB : on + 64u, 0, 0);
        const int qi = qt * 16 + fr;
        const bool is_cls = qi == 0;                                // this lane's query is the CLS query
        f32x4_t s[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            bf16x8_t k0, k1;
            s_rowfrag(sK, t, L, k0, k1);
            s[t] = s_mfma(k1, s_bf(q1), s_mfma(k0, s_bf(q0), zero));
        }
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (s_mask_tile<NT, EXACT>(t)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = t * 16 + fg * 4 + r;
                    const bool dead = key >= nk || (key == 0 && is_cls && q.g != 0);   // the (CLS, CLS) pair belongs to group 0
                    s[t][r] = dead ? -INFINITY : s[t][r];
                }
            }
            m = fmaxf(m, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
        }
        m = s_grp_max(m) * sc2;
        float l = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = s_exp2(fmaf(s[t][r], sc2, -m));
                s[t][r] = e;
                l += e;
            }
        l = s_grp_sum(l);
        f32x4_t o[4] = {zero, zero, zero, zero};
#pragma unroll
        for (int kk = 0; kk < NP; ++kk) {
            const bf16x8_t pf = s_pack8(s[2 * kk], 2 * kk + 1 < NT ? s[2 * kk + 1 < NT ? 2 * kk + 1 : 0] : zero);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = s_mfma(s_trfrag(sV, 2 * kk, 2 * kk + 1 < NT, dt, L), pf, o[dt]);
        }
        const int row = s_row_of(q, qi);
        const bool patch = qi >= 1 && qi <= q.n;
        if (is_cls && a.ws) {                                       // the CLS query's partial state over this group's keys
            float* dst = a.ws + (((long long)q.g * a.nb + q.b) * a.H + q.h) * 66;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4_t*>(dst + 2 + dt * 16 + fg * 4) = o[dt];
            if (fg == 0) { dst[0] = m * S_LN2; dst[1] = l; }
        }
        const float inv = 1.0f / l;
        s_store_rows(scr, rO, patch ? (unsigned int)(row * a.ldo + a.ocol + q.h * HD + fg * 8) * 2u : S_OOB, o, inv, L, fr, fg);
        if (a.lse && patch && fg == 0) a.lse[(long long)row * a.H + q.h] = m * S_LN2 + __logf(l);
        q0 = nq0; q1 = nq1;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward: dQ, dK, dV of the group's rows, delta of its queries, and (a.ws set) the group's share of the CLS row's three gradients
// as fp32 partials ws[sample * G + group][head][3][64] (dQ, dK, dV: attn_cls_reduce_kernel(self_term = 0) sums them)
// (-DSPACE_BWD_WPE=n: experiment hook -- waves per SIMD the backward kernel is compiled for; 4 = 128 registers = two workgroups per CU)
#ifndef SPACE_BWD_WPE
#define SPACE_BWD_WPE 4
#endif
template <int NT, bool EXACT>
__global__ __launch_bounds__(64 * SNW, (NT <= 14 ? SPACE_BWD_WPE : 2)) void attn_space_bwd_kernel(const SpaceArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int IMG = NT * 16 * RB;
    constexpr int NP = (NT + 1) / 2;
    unsigned char* im0 = smem;                                     // phase A: K   phase B: Q
    unsigned char* im1 = smem + IMG;                               // phase A: V   phase B: dO
    float* sL = reinterpret_cast<float*>(smem + 2 * IMG);          // lse (log2 domain) of every query of the row list (+inf past it)
    float* sD = sL + NT * 16;                                      // delta
    const int tid = threadIdx.x, lane = tid & 63, w = wave_id();
    const int fr = lane & 15, fg = lane >> 4;
    unsigned char* scr = smem + 2 * IMG + 2 * NT * 16 * 4 + w * 16 * RB;
    const SLane L = s_lane(fr, fg);
    const SGeom q = s_geom(a);
    const int nk = q.n + 1;                                        // rows of the list: the CLS row (index 0) + patches
    const int nlt = (nk + 15) >> 4;                                // live tiles
    const float sc2 = a.scale * S_LOG2E;
    const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.qkv), 0, (int)a.qkv_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.o), 0, (int)a.o_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dO), 0, (int)a.o_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(a.dqkv, 0, (int)a.dqkv_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rL = __builtin_amdgcn_make_buffer_rsrc(a.lse, 0, (int)a.lse_bytes, 0x00020000);
    auto ld2 = [&](__amdgpu_buffer_rsrc_t r, unsigned int off, u32x4_t& x0, u32x4_t& x1) {
        x0 = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
        x1 = __builtin_amdgcn_raw_buffer_load_b128(r, off == S_OOB ? S_OOB : off + 64u, 0, 0);
    };
    auto foff = [&](int t, int ld, int col) {                      // fragment offset of row t*16 + fr of the row list
        const int row = s_row_of(q, t * 16 + fr);
        return (row >= 0 && t < nlt) ? (unsigned int)(row * ld + col + q.h * HD + fg * 8) * 2u : S_OOB;
    };
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    float* pw = a.ws ? a.ws + ((long long)(q.b * a.G + q.g) * a.H + q.h) * 3 * HD : nullptr;

    // ================= phase A: keys in LDS, every wave's query tiles -> dQ (lane = query) =================
    u32x4_t q0, q1, g0, g1, o0, o1;
    ld2(rQ, foff(w, a.ld, a.qcol), q0, q1);
    ld2(rG, foff(w, a.ldo, a.ocol), g0, g1);
    ld2(rO, foff(w, a.ldo, a.ocol), o0, o1);
    auto lse_of = [&](int t) {                                     // lse (log2 domain) of row t*16 + fr; past the list: +inf -> exp2(s - inf) = 0
        const int row = s_row_of(q, t * 16 + fr);
        // (a descriptor load: `cond ? a.lse[i] : inf` compiles to a branch around a global load followed by vmcnt(0), which also waits for
        // the fragment prefetches issued just before it)
        const bool live = row >= 0 && t < nlt;
        const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rL, live ? (unsigned int)(row * a.H + q.h) * 4u : S_OOB, 0, 0));
        return live ? v * S_LOG2E : INFINITY;
    };
    float lse2 = lse_of(w);                                        // (loaded a tile ahead like the fragments: a load waited for inside the
                                                                   // loop would also wait for the prefetches issued before it)
    s_stage<NT>(im0, rQ, a.ld, a.kcol + q.h * HD, q, tid);
    s_stage<NT>(im1, rQ, a.ld, a.vcol + q.h * HD, q, tid);
    __syncthreads();
    for (int qt = w; qt < NT; qt += SNW) {
        u32x4_t nq0, nq1, ng0, ng1, no0, no1;                      // next tile's fragments: their latency hides behind this tile
        ld2(rQ, foff(qt + SNW, a.ld, a.qcol), nq0, nq1);
        ld2(rG, foff(qt + SNW, a.ldo, a.ocol), ng0, ng1);
        ld2(rO, foff(qt + SNW, a.ldo, a.ocol), no0, no1);
        const float nlse2 = lse_of(qt + SNW);
        const int qi = qt * 16 + fr;
        const int row = s_row_of(q, qi);
        const bool patch = qi >= 1 && qi <= q.n, is_cls = qi == 0;
        const float dl = s_grp_sum(s_dot16(g0, g1, o0, o1));
        if (fg == 0) { sL[qi] = lse2; sD[qi] = dl; }
        if (patch && fg == 0) a.delta[(long long)row * a.H + q.h] = dl;
        if (qt < nlt) {                                             // uniform
            f32x4_t dq[4] = {zero, zero, zero, zero};
#pragma unroll
            for (int kk = 0; kk < NP; ++kk) {
                f32x4_t ds[2] = {zero, zero};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = 2 * kk + u;
                    if (t < NT) {
                        bf16x8_t k0, k1, v0, v1;
                        s_rowfrag(im0, t, L, k0, k1);
                        s_rowfrag(im1, t, L, v0, v1);
                        const f32x4_t sc = s_mfma(k1, s_bf(q1), s_mfma(k0, s_bf(q0), zero));
                        const f32x4_t dp = s_mfma(v1, s_bf(g1), s_mfma(v0, s_bf(g0), zero));
#pragma unroll
                        for (int r = 0; r < 4; ++r) ds[u][r] = s_exp2(fmaf(sc[r], sc2, -lse2)) * (dp[r] - dl);
                        if (s_mask_tile<NT, EXACT>(t)) {            // the tile holds the CLS key or padding
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int key = t * 16 + fg * 4 + r;
                                const bool dead = key >= nk || (key == 0 && is_cls && q.g != 0);   // (CLS, CLS): group 0 only
                                ds[u][r] = dead ? 0.f : ds[u][r];
                            }
                        }
                    }
                }
                const bf16x8_t bd = s_pack8(ds[0], ds[1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dq[dt] = s_mfma(s_trfrag(im0, 2 * kk, 2 * kk + 1 < NT, dt, L), bd, dq[dt]);
            }
            if (is_cls && pw) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4_t*>(pw + dt * 16 + fg * 4) = dq[dt] * a.scale;
            }
            s_store_rows(scr, rD, patch ? (unsigned int)(row * a.ldd + a.dqcol + q.h * HD + fg * 8) * 2u : S_OOB, dq, a.scale, L, fr, fg);
        }
        q0 = nq0; q1 = nq1; g0 = ng0; g1 = ng1; o0 = no0; o1 = no1;
        lse2 = nlse2;
    }
    // ================= phase B: queries in LDS, every wave's key tiles -> dK, dV (lane = key) =================
    u32x4_t k0, k1, v0, v1;
    ld2(rQ, foff(w, a.ld, a.kcol), k0, k1);
    ld2(rQ, foff(w, a.ld, a.vcol), v0, v1);
    __syncthreads();                                               // every wave is done with the key images; sL / sD are complete
    s_stage<NT>(im0, rQ, a.ld, a.qcol + q.h * HD, q, tid);
    s_stage<NT>(im1, rG, a.ldo, a.ocol + q.h * HD, q, tid);
    __syncthreads();
    for (int kt = w; kt < nlt; kt += SNW) {
        u32x4_t nk0, nk1, nv0, nv1;
        ld2(rQ, foff(kt + SNW, a.ld, a.kcol), nk0, nk1);
        ld2(rQ, foff(kt + SNW, a.ld, a.vcol), nv0, nv1);
        const int ki = kt * 16 + fr;
        const int row = s_row_of(q, ki);
        const bool patch = ki >= 1 && ki <= q.n, is_cls = ki == 0;
        f32x4_t dv[4] = {zero, zero, zero, zero}, dk[4] = {zero, zero, zero, zero};
#pragma unroll
        for (int kk = 0; kk < NP; ++kk) {
            f32x4_t pp[2] = {zero, zero}, ds[2] = {zero, zero};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = 2 * kk + u;
                if (t < NT) {
                    bf16x8_t a0, a1, b0, b1;
                    s_rowfrag(im0, t, L, a0, a1);
                    s_rowfrag(im1, t, L, b0, b1);
                    const f32x4_t sc = s_mfma(a1, s_bf(k1), s_mfma(a0, s_bf(k0), zero));
                    const f32x4_t dp = s_mfma(b1, s_bf(v1), s_mfma(b0, s_bf(v0), zero));
                    const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(sL + t * 16 + fg * 4), d4 = *reinterpret_cast<const f32x4_t*>(sD + t * 16 + fg * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {                  // queries past the list: lse = +inf -> p = 0
                        const float p = s_exp2(fmaf(sc[r], sc2, -l4[r]));
                        pp[u][r] = p;
                        ds[u][r] = p * (dp[r] - d4[r]);
                    }
                    if (t == 0 && q.g != 0 && is_cls && fg == 0) { pp[u][0] = 0.f; ds[u][0] = 0.f; }   // (CLS query, CLS key) is group 0's
                }
            }
            const bf16x8_t bp = s_pack8(pp[0], pp[1]), bd = s_pack8(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dv[dt] = s_mfma(s_trfrag(im1, 2 * kk, 2 * kk + 1 < NT, dt, L), bp, dv[dt]);
                dk[dt] = s_mfma(s_trfrag(im0, 2 * kk, 2 * kk + 1 < NT, dt, L), bd, dk[dt]);
            }
        }
        if (is_cls && pw) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                *reinterpret_cast<f32x4_t*>(pw + HD + dt * 16 + fg * 4) = dk[dt] * a.scale;
                *reinterpret_cast<f32x4_t*>(pw + 2 * HD + dt * 16 + fg * 4) = dv[dt];
            }
        }
        s_store_rows(scr, rD, patch ? (unsigned int)(row * a.ldd + a.dvcol + q.h * HD + fg * 8) * 2u : S_OOB, dv, 1.0f, L, fr, fg);
        s_store_rows(scr, rD, patch ? (unsigned int)(row * a.ldd + a.dkcol + q.h * HD + fg * 8) * 2u : S_OOB, dk, a.scale, L, fr, fg);
        k0 = nk0; k1 = nk1; v0 = nv0; v1 = nv1;
    }
}

template <int NT> constexpr size_t s_fwd_lds() { return (size_t)2 * NT * 16 * RB + SNW * 16 * RB; }
template <int NT> constexpr size_t s_bwd_lds() { return (size_t)2 * NT * 16 * RB + 2 * NT * 16 * 4 + SNW * 16 * RB; }

}  // namespace egv
using namespace egv;

void egv_attn_cls_combine_launch(const AttnArgs& a, int B, hipStream_t st);       // egv_attn_time.hip

static bool space_on() {
    static const bool on = egv_cfg_on("EGV_ATTN_SPACE_NEW", true);
    return on;
}
// the kernels' argument block from an AttnArgs; false if the launch is not one they cover
static bool space_args(const AttnArgs& a, int B, bool backward, SpaceArgs& s) {
    const bool same = a.q.bs == a.k.bs && a.q.base == a.k.base && a.q.gs == a.k.gs && a.q.is == a.k.is && a.q.n == a.k.n;
    auto ok8 = [](long long x) { return (x % 8) == 0; };
    if (!(same && a.q.n >= 17 && a.q.n + 1 <= 17 * 16 && a.extra && a.extra_row == 0 && !a.mask && a.drop_p <= 0.f && a.nsplit == 1 && a.O)) return false;
    if (!(ok8(a.ldq) && ok8(a.ldo) && ok8(a.qoff) && ok8(a.koff) && ok8(a.voff) && ok8(a.ooff) && a.ldq == a.ldk && a.ldq == a.ldv)) return false;
    const long long rows = (long long)B * a.extra_bs;
    if (rows * a.ldq * 2 >= (1LL << 31) || rows * a.ldo * 2 >= (1LL << 31)) return false;
    // Q, K, V are column slices of one token matrix (the fused qkv buffer): one buffer descriptor serves the three
    const long long kd = (reinterpret_cast<const char*>(a.K) - reinterpret_cast<const char*>(a.Q)) / 2, vd = (reinterpret_cast<const char*>(a.V) - reinterpret_cast<const char*>(a.Q)) / 2;
    if (kd < 0 || vd < 0 || kd + a.koff + a.H * HD > a.ldq || vd + a.voff + a.H * HD > a.ldq || !ok8(kd) || !ok8(vd)) return false;
    s = SpaceArgs{};
    s.qkv = a.Q; s.o = a.O; s.dO = a.dO; s.dqkv = a.dQ; s.lse = a.lse; s.delta = a.delta; s.ws = a.ws;
    s.ld = a.ldq; s.ldo = a.ldo;
    s.qcol = a.qoff; s.kcol = (int)kd + a.koff; s.vcol = (int)vd + a.voff; s.ocol = a.ooff;
    s.H = a.H; s.G = a.G; s.n = a.q.n; s.nb = B;
    s.bs = (int)a.q.bs; s.base = (int)a.q.base; s.gs = (int)a.q.gs; s.is = (int)a.q.is; s.cls_bs = (int)a.extra_bs;
    s.scale = a.scale;
    s.qkv_bytes = (unsigned int)(rows * a.ldq * 2); s.o_bytes = (unsigned int)(rows * a.ldo * 2);
    if (backward) {
        if (!a.dO || !a.lse || !a.delta || !a.dQ || !a.dK || !a.dV) return false;
        if (!(ok8(a.lddq) && ok8(a.dqoff) && ok8(a.dkoff) && ok8(a.dvoff) && a.lddq == a.lddk && a.lddq == a.lddv)) return false;
        if (rows * a.lddq * 2 >= (1LL << 31)) return false;
        const long long dkd = (reinterpret_cast<const char*>(a.dK) - reinterpret_cast<const char*>(a.dQ)) / 2, dvd = (reinterpret_cast<const char*>(a.dV) - reinterpret_cast<const char*>(a.dQ)) / 2;
        if (dkd < 0 || dvd < 0 || dkd + a.dkoff + a.H * HD > a.lddq || dvd + a.dvoff + a.H * HD > a.lddq || !ok8(dkd) || !ok8(dvd)) return false;
        s.ldd = a.lddq; s.dqcol = a.dqoff; s.dkcol = (int)dkd + a.dkoff; s.dvcol = (int)dvd + a.dvoff;
        s.dqkv_bytes = (unsigned int)(rows * a.lddq * 2);
        s.lse_bytes = (unsigned int)(rows * a.H * 4);
    }
    return true;
}
// the CLS query is served too (partials + combination) when a.ws is set
bool egv_attn_space_fwd_ok(const AttnArgs& a, int B) {
    SpaceArgs s;
    return space_on() && space_args(a, B, false, s);
}

template <int NT, bool EXACT>
static void launch_space_fwd(const SpaceArgs& s, int nwg, hipStream_t st) {
    constexpr size_t lds = s_fwd_lds<NT>();
    static bool attr = false;
    if (!attr && lds > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_space_fwd_kernel<NT, EXACT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL((attn_space_fwd_kernel<NT, EXACT>), dim3(nwg), dim3(64 * SNW), lds, st, s);
}
template <int NT, bool EXACT>
static void launch_space_bwd(const SpaceArgs& s, int nwg, hipStream_t st) {
    constexpr size_t lds = s_bwd_lds<NT>();
    static bool attr = false;
    if (!attr && lds > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_space_bwd_kernel<NT, EXACT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL((attn_space_bwd_kernel<NT, EXACT>), dim3(nwg), dim3(64 * SNW), lds, st, s);
}
// nt == 3: the 33-row groups of the 32-frame time attention (BASELINE.json configs[3]: CLS + 32 frames of one patch position); without
// its own instance they ran on the five-tile form with run-time row counts (two dead key tiles per query tile).  -DSPACE_NT3=0: that form again
#ifndef SPACE_NT3
#define SPACE_NT3 1
#endif
#define SPACE_DISPATCH(FN)                                                   \
    do {                                                                     \
        const int nt = (s.n + 1 + 15) / 16;                                  \
        if (nt == 13) FN<13, true>(s, nwg, st);                              \
        else if (nt == 3 && SPACE_NT3) FN<3, true>(s, nwg, st);              \
        else if (nt == 14) FN<14, true>(s, nwg, st);                         \
        else if (nt == 17) FN<17, true>(s, nwg, st);                         \
        else if (nt == 5) FN<5, true>(s, nwg, st);                           \
        else if (nt < 5) FN<5, false>(s, nwg, st);                           \
        else if (nt < 13) FN<13, false>(s, nwg, st);                         \
        else FN<17, false>(s, nwg, st);                                      \
    } while (0)

// 1: group rows written; 2: also the CLS row (partials in a.ws combined); 0: shape not covered
int egv_attn_space_fwd(const AttnArgs& a, int B, hipStream_t st) {
    SpaceArgs s;
    if (!space_on() || !space_args(a, B, false, s)) return 0;
    const int nwg = B * a.G * a.H;
    SPACE_DISPATCH(launch_space_fwd);
    if (!a.ws) return 1;
    egv_attn_cls_combine_launch(a, B, st);
    return 2;
}

// 1 if enqueued (group rows, their delta and -- with a.ws -- the CLS row's per-group partials; the caller sums those)
int egv_attn_space_bwd(const AttnArgs& a, int B, hipStream_t st) {
    SpaceArgs s;
    if (!space_on() || !space_args(a, B, true, s)) return 0;
    const int nwg = B * a.G * a.H;
    SPACE_DISPATCH(launch_space_bwd);
    return 1;
}
