// Divided SPACE attention for gfx950 (bf16, head_dim 64), forward and one-launch backward: the patch queries of one frame over
// [the frame's patches ; the CLS key] and the CLS query's share of that frame (VarAttention core, video_transformer.py:121-150,
// "(b h f) n d").  One workgroup of four waves per (sample, frame, head); built like the time-attention kernels of
// egv_attn_time.hip, for a key side of up to NT 16-row tiles:
//   * the other side of a phase sits in LDS as ROW-MAJOR images (144-byte pitch: conflict-free 16-byte row reads) copied as stored,
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
constexpr int SP = 144;                         // row pitch (bytes) of the LDS images
constexpr float S_LOG2E = 1.4426950408889634f, S_LN2 = 0.6931471805599453f;
constexpr unsigned int S_OOB = 0x80000000u;
constexpr int SNW = 8;                          // waves per workgroup

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
// row fragment (A operand of a first product): row t*16 + fr of an image, 16-byte chunks fg and fg + 4
__device__ __forceinline__ void s_rowfrag(const unsigned char* img, int t, int fr, int fg, bf16x8_t& lo, bf16x8_t& hi) {
    const unsigned char* p = img + (t * 16 + fr) * SP + fg * 16;
    lo = *reinterpret_cast<const bf16x8_t*>(p);
    hi = *reinterpret_cast<const bf16x8_t*>(p + 64);
}
// transposed fragment (A operand of a second product) for head dims dt*16 .. +15 over the row pair (t0, t1):
// [X^T[d][rows t0*16 + fg*4 .. +3] | X^T[d][rows t1*16 + fg*4 .. +3]] (t1 past the image: zeros)
__device__ __forceinline__ bf16x8_t s_trfrag(const unsigned char* img, int t0, bool has1, int dt, int fr, int fg) {
    const unsigned char* p = img + (t0 * 16 + fg * 4 + (fr >> 2)) * SP + (dt * 16 + (fr & 3) * 4) * 2;
    const s_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s_s16x4_t*)(p));
    s_s16x4_t hi = {0, 0, 0, 0};
    if (has1) hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s_s16x4_t*)(p + 16 * SP));
    const s_s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ void s_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// o[dt] (C layout: row fr, head dims dt*16 + fg*4 .. +3) -> 16-byte row pieces through the wave's scratch image
__device__ __forceinline__ void s_store_rows(unsigned char* scr, __amdgpu_buffer_rsrc_t r, unsigned int off, const f32x4_t (&o)[4], float s,
                                             int fr, int fg) {
    s_wave_sync();                                                 // the previous use of the scratch is over in every lane
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const u32x2_t pk = {pack_bf16x2(o[dt][0] * s, o[dt][1] * s), pack_bf16x2(o[dt][2] * s, o[dt][3] * s)};
        *reinterpret_cast<u32x2_t*>(scr + fr * SP + dt * 32 + fg * 8) = pk;
    }
    s_wave_sync();
    const u32x4_t lo = *reinterpret_cast<const u32x4_t*>(scr + fr * SP + fg * 16), hi = *reinterpret_cast<const u32x4_t*>(scr + fr * SP + 64 + fg * 16);
    __builtin_amdgcn_raw_buffer_store_b128(lo, r, off, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(hi, r, off == S_OOB ? S_OOB : off + 64u, 0, 0);
}

struct SGeom {
    int h, b, g, n, row0, is, cls;
};
__device__ __forceinline__ SGeom s_geom(const AttnArgs& a) {
    SGeom q;
    const int prob = blockIdx.x;
    q.h = prob % a.H;
    const int pg = prob / a.H;
    q.b = pg / a.G; q.g = pg % a.G;
    q.n = a.q.n;
    q.row0 = (int)(q.b * a.q.bs + a.q.base + q.g * a.q.gs);
    q.is = (int)a.q.is;
    q.cls = (int)(q.b * a.extra_bs + a.extra_row);
    return q;
}
// token-matrix row of index i of the group's row list [CLS row ; patch rows 0 .. n-1], -1 past it (the CLS row first: the order in
// which attn_fwd_mfma_kernel sums the keys -- outputs stay bit-identical to that kernel's)
__device__ __forceinline__ int s_row_of(const SGeom& q, int i) { return i == 0 ? q.cls : (i <= q.n ? q.row0 + (i - 1) * q.is : -1); }

// copy rows [0, NT*16) of the group's row list (one head's 64 columns of `base`) into an image
template <int NT>
__device__ __forceinline__ void s_stage(unsigned char* img, __amdgpu_buffer_rsrc_t r, int ld, int coff, const SGeom& q, int tid) {
    constexpr int CH = NT * 16 * 8;
    constexpr int IT = (CH + 64 * SNW - 1) / (64 * SNW);
    u32x4_t v[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = tid + it * 64 * SNW;
        const int row = s_row_of(q, c >> 3);
        v[it] = __builtin_amdgcn_raw_buffer_load_b128(r, (row >= 0 && c < CH) ? (unsigned int)(row * ld + coff + (c & 7) * 8) * 2u : S_OOB, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = tid + it * 64 * SNW;
        if (c < CH) *reinterpret_cast<u32x4_t*>(img + (c >> 3) * SP + (c & 7) * 16) = v[it];
    }
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// forward
template <int NT>
__global__ __launch_bounds__(64 * SNW) void attn_space_fwd_kernel(const AttnArgs a, int nb, unsigned int qkv_bytes, unsigned int o_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int IMG = NT * 16 * SP;
    constexpr int NP = (NT + 1) / 2;
    unsigned char* sK = smem;
    unsigned char* sV = smem + IMG;
    const int tid = threadIdx.x, lane = tid & 63, w = wave_id();
    const int fr = lane & 15, fg = lane >> 4;
    unsigned char* scr = smem + 2 * IMG + w * 16 * SP;
    const SGeom q = s_geom(a);
    const int nk = q.n + 1;                                        // keys incl. the CLS key (index 0)
    const float sc2 = a.scale * S_LOG2E;
    auto mk = [&](const void* p, unsigned int bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000); };
    const __amdgpu_buffer_rsrc_t rQ = mk(a.Q, qkv_bytes), rK = mk(a.K, qkv_bytes), rV = mk(a.V, qkv_bytes), rO = mk(a.O, o_bytes);

    // own query tiles of this wave: w, w + 4, ... (the CLS query is index n of the row list); fragments of the first one now
    const int nqt = (nk + 15) >> 4;
    auto qoff = [&](int qt) {
        const int row = s_row_of(q, qt * 16 + fr);
        return (row >= 0 && qt < nqt) ? (unsigned int)(row * a.ldq + a.qoff + q.h * HD + fg * 8) * 2u : S_OOB;
    };
    u32x4_t q0 = __builtin_amdgcn_raw_buffer_load_b128(rQ, qoff(w), 0, 0), q1 = __builtin_amdgcn_raw_buffer_load_b128(rQ, qoff(w) == S_OOB ? S_OOB : qoff(w) + 64u, 0, 0);
    s_stage<NT>(sK, rK, a.ldk, a.koff + q.h * HD, q, tid);
    s_stage<NT>(sV, rV, a.ldv, a.voff + q.h * HD, q, tid);
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
            s_rowfrag(sK, t, fr, fg, k0, k1);
            s[t] = s_mfma(k1, s_bf(q1), s_mfma(k0, s_bf(q0), zero));
        }
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t == 0 || t * 16 + 16 > nk) {                      // uniform: tiles that hold the CLS key or padding
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
            for (int dt = 0; dt < 4; ++dt) o[dt] = s_mfma(s_trfrag(sV, 2 * kk, 2 * kk + 1 < NT, dt, fr, fg), pf, o[dt]);
        }
        const int row = s_row_of(q, qi);
        const bool patch = qi >= 1 && qi <= q.n;
        if (is_cls && a.ws) {                                       // the CLS query's partial state over this group's keys
            float* dst = a.ws + (((long long)q.g * nb + q.b) * a.H + q.h) * 66;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4_t*>(dst + 2 + dt * 16 + fg * 4) = o[dt];
            if (fg == 0) { dst[0] = m * S_LN2; dst[1] = l; }
        }
        const float inv = 1.0f / l;
        s_store_rows(scr, rO, patch ? (unsigned int)(row * a.ldo + a.ooff + q.h * HD + fg * 8) * 2u : S_OOB, o, inv, fr, fg);
        if (a.lse && patch && fg == 0) a.lse[(long long)row * a.H + q.h] = m * S_LN2 + __logf(l);
        q0 = nq0; q1 = nq1;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward: dQ, dK, dV of the group's rows, delta of its queries, and (a.ws set) the group's share of the CLS row's three gradients
// as fp32 partials ws[sample * G + group][head][3][64] (dQ, dK, dV: attn_cls_reduce_kernel(self_term = 0) sums them)
template <int NT>
__global__ __launch_bounds__(64 * SNW, (NT <= 14 ? 4 : 2)) void attn_space_bwd_kernel(const AttnArgs a, unsigned int qkv_bytes, unsigned int o_bytes,
                                                                     unsigned int dqkv_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int IMG = NT * 16 * SP;
    constexpr int NP = (NT + 1) / 2;
    unsigned char* im0 = smem;                                     // phase A: K   phase B: Q
    unsigned char* im1 = smem + IMG;                               // phase A: V   phase B: dO
    float* sL = reinterpret_cast<float*>(smem + 2 * IMG);          // lse (log2 domain) of every query of the row list (+inf past it)
    float* sD = sL + NT * 16;                                      // delta
    const int tid = threadIdx.x, lane = tid & 63, w = wave_id();
    const int fr = lane & 15, fg = lane >> 4;
    unsigned char* scr = smem + 2 * IMG + 2 * NT * 16 * 4 + w * 16 * SP;
    const SGeom q = s_geom(a);
    const int nk = q.n + 1;                                        // rows of the list: the CLS row (index 0) + patches
    const int nlt = (nk + 15) >> 4;                                // live tiles
    const float sc2 = a.scale * S_LOG2E;
    auto mk = [&](const void* p, unsigned int bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000); };
    const __amdgpu_buffer_rsrc_t rQ = mk(a.Q, qkv_bytes), rK = mk(a.K, qkv_bytes), rV = mk(a.V, qkv_bytes);
    const __amdgpu_buffer_rsrc_t rO = mk(a.O, o_bytes), rG = mk(a.dO, o_bytes);
    const __amdgpu_buffer_rsrc_t rDQ = mk(a.dQ, dqkv_bytes), rDK = mk(a.dK, dqkv_bytes), rDV = mk(a.dV, dqkv_bytes);
    auto ld2 = [&](__amdgpu_buffer_rsrc_t r, unsigned int off, u32x4_t& x0, u32x4_t& x1) {
        x0 = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
        x1 = __builtin_amdgcn_raw_buffer_load_b128(r, off == S_OOB ? S_OOB : off + 64u, 0, 0);
    };
    auto foff = [&](int t, int ld, int coff) {                     // fragment offset of row t*16 + fr of the row list
        const int row = s_row_of(q, t * 16 + fr);
        return (row >= 0 && t < nlt) ? (unsigned int)(row * ld + coff + q.h * HD + fg * 8) * 2u : S_OOB;
    };
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    float* pw = a.ws ? a.ws + ((long long)(q.b * a.G + q.g) * a.H + q.h) * 3 * HD : nullptr;

    // ================= phase A: keys in LDS, every wave's query tiles -> dQ (lane = query) =================
    u32x4_t q0, q1, g0, g1, o0, o1;
    ld2(rQ, foff(w, a.ldq, a.qoff), q0, q1);
    ld2(rG, foff(w, a.ldo, a.ooff), g0, g1);
    ld2(rO, foff(w, a.ldo, a.ooff), o0, o1);
    s_stage<NT>(im0, rK, a.ldk, a.koff + q.h * HD, q, tid);
    s_stage<NT>(im1, rV, a.ldv, a.voff + q.h * HD, q, tid);
    __syncthreads();
    for (int qt = w; qt < NT; qt += SNW) {
        u32x4_t nq0, nq1, ng0, ng1, no0, no1;                      // next tile's fragments: their latency hides behind this tile
        ld2(rQ, foff(qt + SNW, a.ldq, a.qoff), nq0, nq1);
        ld2(rG, foff(qt + SNW, a.ldo, a.ooff), ng0, ng1);
        ld2(rO, foff(qt + SNW, a.ldo, a.ooff), no0, no1);
        const int qi = qt * 16 + fr;
        const int row = s_row_of(q, qi);
        const bool patch = qi >= 1 && qi <= q.n, is_cls = qi == 0;
        const float lse2 = row >= 0 ? a.lse[(long long)row * a.H + q.h] * S_LOG2E : INFINITY;    // past the list: exp2(s - inf) = 0
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
                        s_rowfrag(im0, t, fr, fg, k0, k1);
                        s_rowfrag(im1, t, fr, fg, v0, v1);
                        const f32x4_t sc = s_mfma(k1, s_bf(q1), s_mfma(k0, s_bf(q0), zero));
                        const f32x4_t dp = s_mfma(v1, s_bf(g1), s_mfma(v0, s_bf(g0), zero));
#pragma unroll
                        for (int r = 0; r < 4; ++r) ds[u][r] = s_exp2(fmaf(sc[r], sc2, -lse2)) * (dp[r] - dl);
                        if (t == 0 || t * 16 + 16 > nk) {           // uniform: the tile holds the CLS key or padding
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
                for (int dt = 0; dt < 4; ++dt) dq[dt] = s_mfma(s_trfrag(im0, 2 * kk, 2 * kk + 1 < NT, dt, fr, fg), bd, dq[dt]);
            }
            if (is_cls && pw) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4_t*>(pw + dt * 16 + fg * 4) = dq[dt] * a.scale;
            }
            s_store_rows(scr, rDQ, patch ? (unsigned int)(row * a.lddq + a.dqoff + q.h * HD + fg * 8) * 2u : S_OOB, dq, a.scale, fr, fg);
        }
        q0 = nq0; q1 = nq1; g0 = ng0; g1 = ng1; o0 = no0; o1 = no1;
    }
    // ================= phase B: queries in LDS, every wave's key tiles -> dK, dV (lane = key) =================
    u32x4_t k0, k1, v0, v1;
    ld2(rK, foff(w, a.ldk, a.koff), k0, k1);
    ld2(rV, foff(w, a.ldv, a.voff), v0, v1);
    __syncthreads();                                               // every wave is done with the key images; sL / sD are complete
    s_stage<NT>(im0, rQ, a.ldq, a.qoff + q.h * HD, q, tid);
    s_stage<NT>(im1, rG, a.ldo, a.ooff + q.h * HD, q, tid);
    __syncthreads();
    for (int kt = w; kt < nlt; kt += SNW) {
        u32x4_t nk0, nk1, nv0, nv1;
        ld2(rK, foff(kt + SNW, a.ldk, a.koff), nk0, nk1);
        ld2(rV, foff(kt + SNW, a.ldv, a.voff), nv0, nv1);
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
                    s_rowfrag(im0, t, fr, fg, a0, a1);
                    s_rowfrag(im1, t, fr, fg, b0, b1);
                    const f32x4_t sc = s_mfma(a1, s_bf(k1), s_mfma(a0, s_bf(k0), zero));
                    const f32x4_t dp = s_mfma(b1, s_bf(v1), s_mfma(b0, s_bf(v0), zero));
                    const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(sL + t * 16 + fg * 4), d4 = *reinterpret_cast<const f32x4_t*>(sD + t * 16 + fg * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
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
                dv[dt] = s_mfma(s_trfrag(im1, 2 * kk, 2 * kk + 1 < NT, dt, fr, fg), bp, dv[dt]);
                dk[dt] = s_mfma(s_trfrag(im0, 2 * kk, 2 * kk + 1 < NT, dt, fr, fg), bd, dk[dt]);
            }
        }
        if (is_cls && pw) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                *reinterpret_cast<f32x4_t*>(pw + HD + dt * 16 + fg * 4) = dk[dt] * a.scale;
                *reinterpret_cast<f32x4_t*>(pw + 2 * HD + dt * 16 + fg * 4) = dv[dt];
            }
        }
        s_store_rows(scr, rDV, patch ? (unsigned int)(row * a.lddv + a.dvoff + q.h * HD + fg * 8) * 2u : S_OOB, dv, 1.0f, fr, fg);
        s_store_rows(scr, rDK, patch ? (unsigned int)(row * a.lddk + a.dkoff + q.h * HD + fg * 8) * 2u : S_OOB, dk, a.scale, fr, fg);
        k0 = nk0; k1 = nk1; v0 = nv0; v1 = nv1;
    }
}

template <int NT> constexpr size_t s_bwd_lds() { return (size_t)2 * NT * 16 * SP + 2 * NT * 16 * 4 + SNW * 16 * SP; }
template <int NT> constexpr size_t s_fwd_lds() { return (size_t)2 * NT * 16 * SP + SNW * 16 * SP; }

}  // namespace egv
using namespace egv;

void egv_attn_cls_combine_launch(const AttnArgs& a, int B, hipStream_t st);       // egv_attn_time.hip

static bool space_shape_ok(const AttnArgs& a, int B) {
    const bool same = a.q.bs == a.k.bs && a.q.base == a.k.base && a.q.gs == a.k.gs && a.q.is == a.k.is && a.q.n == a.k.n;
    auto ok8 = [](int x) { return (x % 8) == 0; };
    if (!(same && a.q.n >= 17 && a.q.n + 1 <= 17 * 16 && a.extra && a.extra_row == 0 && !a.mask && a.drop_p <= 0.f && a.nsplit == 1 && a.O)) return false;
    if (!(ok8(a.ldq) && ok8(a.ldk) && ok8(a.ldv) && ok8(a.ldo) && ok8(a.qoff) && ok8(a.koff) && ok8(a.voff) && ok8(a.ooff) && a.ldq == a.ldk &&
          a.ldq == a.ldv))
        return false;
    const long long rows = (long long)B * a.extra_bs;
    return rows * a.ldq * 2 < (1LL << 31) && rows * a.ldo * 2 < (1LL << 31);
}
static bool space_on() {
    static const bool on = !getenv("EGV_ATTN_SPACE_NEW") || atoi(getenv("EGV_ATTN_SPACE_NEW")) != 0;
    return on;
}
// the CLS query is served too (partials + combination) when a.ws is set
bool egv_attn_space_fwd_ok(const AttnArgs& a, int B) { return space_on() && space_shape_ok(a, B); }

template <int NT>
static void launch_space_fwd(const AttnArgs& a, int B, unsigned int qb, unsigned int ob, hipStream_t st) {
    constexpr size_t lds = s_fwd_lds<NT>();
    static bool attr = false;
    if (!attr && lds > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_space_fwd_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL((attn_space_fwd_kernel<NT>), dim3(B * a.G * a.H), dim3(64 * SNW), lds, st, a, B, qb, ob);
}

// 1: group rows written; 2: also the CLS row (partials in a.ws combined); 0: shape not covered
int egv_attn_space_fwd(const AttnArgs& a, int B, hipStream_t st) {
    if (!egv_attn_space_fwd_ok(a, B)) return 0;
    const long long rows = (long long)B * a.extra_bs;
    const unsigned int qb = (unsigned int)(rows * a.ldq * 2), ob = (unsigned int)(rows * a.ldo * 2);
    const int nt = (a.q.n + 1 + 15) / 16;
    if (nt <= 5) launch_space_fwd<5>(a, B, qb, ob, st);
    else if (nt <= 13) launch_space_fwd<13>(a, B, qb, ob, st);
    else if (nt <= 14) launch_space_fwd<14>(a, B, qb, ob, st);
    else launch_space_fwd<17>(a, B, qb, ob, st);
    if (!a.ws) return 1;
    egv_attn_cls_combine_launch(a, B, st);
    return 2;
}

template <int NT>
static void launch_space_bwd(const AttnArgs& a, int B, unsigned int qb, unsigned int ob, unsigned int db, hipStream_t st) {
    constexpr size_t lds = s_bwd_lds<NT>();
    static bool attr = false;
    if (!attr && lds > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_space_bwd_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL((attn_space_bwd_kernel<NT>), dim3(B * a.G * a.H), dim3(64 * SNW), lds, st, a, qb, ob, db);
}

// 1 if enqueued (group rows, their delta and -- with a.ws -- the CLS row's per-group partials; the caller sums those)
int egv_attn_space_bwd(const AttnArgs& a, int B, hipStream_t st) {
    if (!space_on() || !space_shape_ok(a, B)) return 0;
    if (!a.dO || !a.lse || !a.delta || !a.dQ || !a.dK || !a.dV) return 0;
    if ((a.lddq % 8) || (a.lddk % 8) || (a.lddv % 8) || (a.dqoff % 8) || (a.dkoff % 8) || (a.dvoff % 8) || a.lddq != a.lddk || a.lddq != a.lddv) return 0;
    const long long rows = (long long)B * a.extra_bs;
    if (rows * a.lddq * 2 >= (1LL << 31)) return 0;
    const unsigned int qb = (unsigned int)(rows * a.ldq * 2), ob = (unsigned int)(rows * a.ldo * 2), db = (unsigned int)(rows * a.lddq * 2);
    const int nt = (a.q.n + 1 + 15) / 16;
    if (nt <= 5) launch_space_bwd<5>(a, B, qb, ob, db, st);
    else if (nt <= 13) launch_space_bwd<13>(a, B, qb, ob, db, st);
    else if (nt <= 14) launch_space_bwd<14>(a, B, qb, ob, db, st);
    else launch_space_bwd<17>(a, B, qb, ob, db, st);
    return 1;
}
