// One-launch backward of the divided TIME attention for gfx950 (bf16, head_dim 64): dQ, dK, dV of the <= 16 frames of one patch
// position AND the group's share of the CLS row's three gradients (VarAttention core, video_transformer.py:121-150, "(b h n) f d").
//
// A group is tiny -- 16 queries x (16 keys + the CLS key), plus the CLS query over the group's 16 keys -- and there are B*N*H of them
// (18 816 at B = 8): the generic kernel pair of egv_attn_mfma.hip (query-owned dQ + key-owned dK/dV, each staging the other side
// through a register transpose) spent 60 % of the chip's VALU issue slots on address arithmetic, predication and staging
// (rocprofv3: 1 100 VALU instructions per wave and launch) for 2 % MFMA use.  Here ONE wave owns one (sample, patch, head):
//   * every operand row is fetched once, straight into MFMA fragment layout (lane (fr, fg): row fr, 16-byte chunks fg and fg + 4),
//     through buffer descriptors with 32-bit offsets -- rows >= n and the lanes that hold no CLS data read zeros by construction
//     (out-of-range offsets), there is no predication and no 64-bit address arithmetic;
//   * scores are formed twice on the (idle) matrix pipe, once per C layout: S^T = K Q^T (lane = query: feeds dQ^T = K^T dS^T) and
//     S = Q K^T (lane = key: feeds dV^T = dO^T P and dK^T = Q^T dS); the C layout of the first product IS the B operand of the second
//     (reduction index permuted: [patch rows fg*4..+3 | CLS tile rows fg*4..+3]), the CLS row rides along as a second 16-row tile
//     whose only live row is row 0;
//   * the transposed A operands (K^T, dO^T, Q^T of the 16 patch rows) come from a 2 KB row-major LDS image per operand with the
//     gfx950 transposing read ds_read_b64_tr_b16; the CLS column of those operands is one bf16 per lane;
//   * no workgroup barrier, no atomics: a workgroup is four independent waves.
// The CLS row's gradients leave as fp32 partials ws[group][head][3][64] (dQ, dK, dV; the layout of attn_bwd_fused_kernel) and are
// summed over the groups, with the (CLS, CLS) term added, by attn_cls_reduce_kernel(self_term = 1).
#include "egv_attn.h"
#include <cstdlib>

namespace egv {

namespace {
typedef __attribute__((ext_vector_type(4))) short t_s16x4_t;
typedef __attribute__((ext_vector_type(8))) short t_s16x8_t;
constexpr int TP = 144;                         // row pitch (bytes) of the LDS images: 64 bf16 + 16 B pad
constexpr int T_IMG = 16 * TP;                  // one operand image (16 rows)
constexpr int T_WLDS = 3 * T_IMG + 3 * 128 + 128;   // K, dO, Q images + the three CLS rows (bf16 x 64) + lse / delta of the 16 queries
constexpr float T_LOG2E = 1.4426950408889634f;
constexpr unsigned int T_OOB = 0x80000000u;

__device__ __forceinline__ float exp2_fast_t(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ bf16x8_t t_bf(u32x4_t v) { return __builtin_bit_cast(bf16x8_t, v); }
__device__ __forceinline__ bf16x8_t t_pack8(const f32x4_t& a, const f32x4_t& b) {
    u32x4_t v = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
    return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ float t_grp_sum(float v) {        // over the four lane groups of a column (lanes l, l^16, l^32, l^48)
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ float t_dot16(u32x4_t a0, u32x4_t a1, u32x4_t b0, u32x4_t b1) {
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
// A operand of a second-stage product for head dims dt*16 .. +15: [X^T[d][patch rows fg*4 .. +3] | x_cls[d], 0, 0, 0 (fg == 0)]
__device__ __forceinline__ bf16x8_t t_afrag(const unsigned char* img, const unsigned char* cls, int dt, int fr, int fg) {
    const unsigned char* p = img + (fg * 4 + (fr >> 2)) * TP + (dt * 16 + (fr & 3) * 4) * 2;
    const t_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) t_s16x4_t*)(p));
    const short c = fg == 0 ? *reinterpret_cast<const short*>(cls + (dt * 16 + fr) * 2) : (short)0;
    const t_s16x8_t v = {lo[0], lo[1], lo[2], lo[3], c, 0, 0, 0};
    return __builtin_bit_cast(bf16x8_t, v);
}
// o[dt] (C layout: row fr, head dims dt*16 + fg*4 .. +3) -> bf16 row pieces of 16 bytes through a (free) LDS image, so that a store
// instruction writes 16 rows x 64 contiguous bytes (as the loads read them) instead of 16 rows x 32
__device__ __forceinline__ void t_store_rows(unsigned char* img, __amdgpu_buffer_rsrc_t r, unsigned int off, const f32x4_t (&o)[4], float s,
                                             int fr, int fg) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const u32x2_t pk = {pack_bf16x2(o[dt][0] * s, o[dt][1] * s), pack_bf16x2(o[dt][2] * s, o[dt][3] * s)};
        *reinterpret_cast<u32x2_t*>(img + fr * TP + dt * 32 + fg * 8) = pk;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const u32x4_t lo = *reinterpret_cast<const u32x4_t*>(img + fr * TP + fg * 16), hi = *reinterpret_cast<const u32x4_t*>(img + fr * TP + 64 + fg * 16);
    __builtin_amdgcn_raw_buffer_store_b128(lo, r, off, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(hi, r, off + (off == T_OOB ? 0u : 64u), 0, 0);
}
__device__ __forceinline__ f32x4_t t_mfma(bf16x8_t a, bf16x8_t b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
}  // namespace

// grid: ceil(nprob / 4) workgroups of 4 waves; problem = (b * G + g) * H + h (heads fastest: the 12 waves that read the 12 head
// slices of the same 16 rows run side by side)
__global__ __launch_bounds__(256) void attn_time_bwd_kernel(const AttnArgs a, int nprob, unsigned int qkv_bytes, unsigned int o_bytes,
                                                            unsigned int dqkv_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, w = wave_id();
    const int fr = lane & 15, fg = lane >> 4;
    const int prob = blockIdx.x * 4 + w;
    if (prob >= nprob) return;                                     // (waves are independent: no workgroup barrier below)
    unsigned char* sK = smem + w * T_WLDS;
    unsigned char* sG = sK + T_IMG;
    unsigned char* sQ = sG + T_IMG;
    unsigned char* cK = sQ + T_IMG;                                // CLS rows: K, dO, Q (64 bf16 each)
    unsigned char* cG = cK + 128;
    unsigned char* cQ = cG + 128;
    float* sL = reinterpret_cast<float*>(cQ + 128);                // lse (log2 domain) of the 16 queries, then their delta
    float* sD = sL + 16;

    const int h = prob % a.H, pg = prob / a.H, b = pg / a.G, g = pg % a.G;
    const int n = a.q.n;                                           // live rows (frames) of the group, <= 16
    const bool valid = fr < n;
    const int row = (int)(b * a.q.bs + a.q.base + g * a.q.gs) + fr * (int)a.q.is;
    const int cls = (int)(b * a.extra_bs + a.extra_row);
    const float sc2 = a.scale * T_LOG2E;

    auto mk = [&](const void* p, unsigned int bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000); };
    const __amdgpu_buffer_rsrc_t rQ = mk(a.Q, qkv_bytes), rK = mk(a.K, qkv_bytes), rV = mk(a.V, qkv_bytes);
    const __amdgpu_buffer_rsrc_t rO = mk(a.O, o_bytes), rG = mk(a.dO, o_bytes);
    auto ld = [&](__amdgpu_buffer_rsrc_t r, unsigned int off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); };

    // ---- every operand once, fragment layout: patch rows (lane row fr) and the CLS row (lanes fr == 0)
    const unsigned int oq = valid ? (unsigned int)(row * a.ldq + a.qoff + h * HD + fg * 8) * 2u : T_OOB;
    const unsigned int ok_ = valid ? (unsigned int)(row * a.ldk + a.koff + h * HD + fg * 8) * 2u : T_OOB;
    const unsigned int ov_ = valid ? (unsigned int)(row * a.ldv + a.voff + h * HD + fg * 8) * 2u : T_OOB;
    const unsigned int oo = valid ? (unsigned int)(row * a.ldo + a.ooff + h * HD + fg * 8) * 2u : T_OOB;
    const bool c0 = fr == 0;
    const unsigned int cq = c0 ? (unsigned int)(cls * a.ldq + a.qoff + h * HD + fg * 8) * 2u : T_OOB;
    const unsigned int ck = c0 ? (unsigned int)(cls * a.ldk + a.koff + h * HD + fg * 8) * 2u : T_OOB;
    const unsigned int cv = c0 ? (unsigned int)(cls * a.ldv + a.voff + h * HD + fg * 8) * 2u : T_OOB;
    const unsigned int co = c0 ? (unsigned int)(cls * a.ldo + a.ooff + h * HD + fg * 8) * 2u : T_OOB;
    const u32x4_t q0 = ld(rQ, oq), q1 = ld(rQ, oq + 64), k0 = ld(rK, ok_), k1 = ld(rK, ok_ + 64), v0 = ld(rV, ov_), v1 = ld(rV, ov_ + 64);
    const u32x4_t g0 = ld(rG, oo), g1 = ld(rG, oo + 64), o0 = ld(rO, oo), o1 = ld(rO, oo + 64);
    const u32x4_t qc0 = ld(rQ, cq), qc1 = ld(rQ, cq + 64), kc0 = ld(rK, ck), kc1 = ld(rK, ck + 64), vc0 = ld(rV, cv), vc1 = ld(rV, cv + 64);
    const u32x4_t gc0 = ld(rG, co), gc1 = ld(rG, co + 64), oc0 = ld(rO, co), oc1 = ld(rO, co + 64);
    const float lse2 = valid ? a.lse[(long long)row * a.H + h] * T_LOG2E : INFINITY;    // padding queries: exp2(s - inf) = 0
    const float lsec = a.lse[(long long)cls * a.H + h] * T_LOG2E;

    // delta = rowsum(dO o O): of the lane's query (all four lane groups of a column end up with it) and of the CLS query (uniform)
    const float dl = t_grp_sum(t_dot16(g0, g1, o0, o1));
    const float dlc = readlane_f(t_grp_sum(t_dot16(gc0, gc1, oc0, oc1)), 0);
    if (valid && fg == 0) a.delta[(long long)row * a.H + h] = dl;

    // ---- LDS images (row-major) for the transposing reads; CLS rows; lse / delta for the key-lane layout
    *reinterpret_cast<u32x4_t*>(sK + fr * TP + fg * 16) = k0;
    *reinterpret_cast<u32x4_t*>(sK + fr * TP + 64 + fg * 16) = k1;
    *reinterpret_cast<u32x4_t*>(sG + fr * TP + fg * 16) = g0;
    *reinterpret_cast<u32x4_t*>(sG + fr * TP + 64 + fg * 16) = g1;
    *reinterpret_cast<u32x4_t*>(sQ + fr * TP + fg * 16) = q0;
    *reinterpret_cast<u32x4_t*>(sQ + fr * TP + 64 + fg * 16) = q1;
    if (c0) {
        *reinterpret_cast<u32x4_t*>(cK + fg * 16) = kc0;
        *reinterpret_cast<u32x4_t*>(cK + 64 + fg * 16) = kc1;
        *reinterpret_cast<u32x4_t*>(cG + fg * 16) = gc0;
        *reinterpret_cast<u32x4_t*>(cG + 64 + fg * 16) = gc1;
        *reinterpret_cast<u32x4_t*>(cQ + fg * 16) = qc0;
        *reinterpret_cast<u32x4_t*>(cQ + 64 + fg * 16) = qc1;
    }
    if (fg == 0) { sL[fr] = lse2; sD[fr] = dl; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    const long long pw_off = ((long long)pg * a.H + h) * 3 * HD;
    const __amdgpu_buffer_rsrc_t rDQ = mk(a.dQ, dqkv_bytes), rDK = mk(a.dK, dqkv_bytes), rDV = mk(a.dV, dqkv_bytes);

    // ================= lane = query: S^T[key][query] -> dQ^T = K^T dS^T =================
    {
        // patch keys x patch queries | CLS key x patch queries | patch keys x CLS query
        f32x4_t s_pp = t_mfma(t_bf(k1), t_bf(q1), t_mfma(t_bf(k0), t_bf(q0), zero));
        f32x4_t d_pp = t_mfma(t_bf(v1), t_bf(g1), t_mfma(t_bf(v0), t_bf(g0), zero));
        f32x4_t s_cp = t_mfma(t_bf(kc1), t_bf(q1), t_mfma(t_bf(kc0), t_bf(q0), zero));
        f32x4_t d_cp = t_mfma(t_bf(vc1), t_bf(g1), t_mfma(t_bf(vc0), t_bf(g0), zero));
        f32x4_t s_pc = t_mfma(t_bf(k1), t_bf(qc1), t_mfma(t_bf(k0), t_bf(qc0), zero));
        f32x4_t d_pc = t_mfma(t_bf(v1), t_bf(gc1), t_mfma(t_bf(v0), t_bf(gc0), zero));
        f32x4_t ds_pp, ds_cp = zero, ds_pc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool kv = fg * 4 + r < n;                          // padding keys contribute nothing
            const float p = exp2_fast_t(fmaf(s_pp[r], sc2, -lse2));
            ds_pp[r] = kv ? p * (d_pp[r] - dl) : 0.f;
            const float pc = exp2_fast_t(fmaf(s_pc[r], sc2, -lsec));
            ds_pc[r] = kv ? pc * (d_pc[r] - dlc) : 0.f;
        }
        {   // the CLS key is row 0 of its tile: element r = 0 of lane group 0
            const float p = exp2_fast_t(fmaf(s_cp[0], sc2, -lse2));
            ds_cp[0] = fg == 0 ? p * (d_cp[0] - dl) : 0.f;
        }
        const bf16x8_t bP = t_pack8(ds_pp, ds_cp), bC = t_pack8(ds_pc, zero);
        f32x4_t dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x8_t ak = t_afrag(sK, cK, dt, fr, fg);
            dq[dt] = t_mfma(ak, bP, zero);
            const f32x4_t dqc = t_mfma(ak, bC, zero);
            if (c0) *reinterpret_cast<f32x4_t*>(a.ws + pw_off + dt * 16 + fg * 4) = dqc * a.scale;
        }
        t_store_rows(sK, rDQ, valid ? (unsigned int)(row * a.lddq + a.dqoff + h * HD + fg * 8) * 2u : T_OOB, dq, a.scale, fr, fg);   // (the K image is free now)
    }
    // ================= lane = key: S[query][key] -> dV^T = dO^T P, dK^T = Q^T dS =================
    {
        f32x4_t s_pp = t_mfma(t_bf(q1), t_bf(k1), t_mfma(t_bf(q0), t_bf(k0), zero));       // patch queries x patch keys
        f32x4_t d_pp = t_mfma(t_bf(g1), t_bf(v1), t_mfma(t_bf(g0), t_bf(v0), zero));
        f32x4_t s_cp = t_mfma(t_bf(qc1), t_bf(k1), t_mfma(t_bf(qc0), t_bf(k0), zero));     // CLS query x patch keys (row 0 of the tile)
        f32x4_t d_cp = t_mfma(t_bf(gc1), t_bf(v1), t_mfma(t_bf(gc0), t_bf(v0), zero));
        f32x4_t s_pc = t_mfma(t_bf(q1), t_bf(kc1), t_mfma(t_bf(q0), t_bf(kc0), zero));     // patch queries x CLS key (column 0: lanes fr == 0)
        f32x4_t d_pc = t_mfma(t_bf(g1), t_bf(vc1), t_mfma(t_bf(g0), t_bf(vc0), zero));
        const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(sL + fg * 4), dl4 = *reinterpret_cast<const f32x4_t*>(sD + fg * 4);
        f32x4_t p_pp, ds_pp, p_pc, ds_pc, p_cp = zero, ds_cp = zero;
#pragma unroll
        for (int r = 0; r < 4; ++r) {                                // query fg*4 + r (padding queries: lse = +inf -> p = 0)
            const float p = exp2_fast_t(fmaf(s_pp[r], sc2, -l4[r]));
            p_pp[r] = p;
            ds_pp[r] = p * (d_pp[r] - dl4[r]);
            const float pc = exp2_fast_t(fmaf(s_pc[r], sc2, -l4[r]));
            p_pc[r] = pc;
            ds_pc[r] = pc * (d_pc[r] - dl4[r]);
        }
        {
            const float p = exp2_fast_t(fmaf(s_cp[0], sc2, -lsec));
            p_cp[0] = fg == 0 ? p : 0.f;
            ds_cp[0] = fg == 0 ? p * (d_cp[0] - dlc) : 0.f;
        }
        const bf16x8_t bvP = t_pack8(p_pp, p_cp), bvC = t_pack8(p_pc, zero), bkP = t_pack8(ds_pp, ds_cp), bkC = t_pack8(ds_pc, zero);
        f32x4_t dv[4], dk[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x8_t ag = t_afrag(sG, cG, dt, fr, fg), aq = t_afrag(sQ, cQ, dt, fr, fg);
            dv[dt] = t_mfma(ag, bvP, zero);
            dk[dt] = t_mfma(aq, bkP, zero);
            const f32x4_t dvc = t_mfma(ag, bvC, zero), dkc = t_mfma(aq, bkC, zero);
            if (c0) {
                *reinterpret_cast<f32x4_t*>(a.ws + pw_off + HD + dt * 16 + fg * 4) = dkc * a.scale;
                *reinterpret_cast<f32x4_t*>(a.ws + pw_off + 2 * HD + dt * 16 + fg * 4) = dvc;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // every lane's transposing reads of the two images are done
        __builtin_amdgcn_wave_barrier();
        t_store_rows(sG, rDV, valid ? (unsigned int)(row * a.lddv + a.dvoff + h * HD + fg * 8) * 2u : T_OOB, dv, 1.0f, fr, fg);
        t_store_rows(sQ, rDK, valid ? (unsigned int)(row * a.lddk + a.dkoff + h * HD + fg * 8) * 2u : T_OOB, dk, a.scale, fr, fg);
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Forward of the same groups: O and lse of the <= 16 patch queries over [CLS key ; the group's keys] AND the CLS query's partial
// softmax state over the group's keys (m, l, o[64] per (group, sample, head): the layout attn_fwd_mfma_kernel leaves for the space
// attention; the CLS key itself is counted in group 0) -- the one-query launch over all S keys (attn1_fwd_kernel) is not needed.
// Scores in the lane = query layout (S^T = K Q^T); the probabilities are the B operand of O^T = V^T P^T as they stand.
__global__ __launch_bounds__(256) void attn_time_fwd_kernel(const AttnArgs a, int nprob, int nb, unsigned int qkv_bytes, unsigned int o_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, w = wave_id();
    const int fr = lane & 15, fg = lane >> 4;
    const int prob = blockIdx.x * 4 + w;
    if (prob >= nprob) return;
    unsigned char* sV = smem + w * (T_IMG + 256);
    unsigned char* cV = sV + T_IMG;
    float* sP = reinterpret_cast<float*>(cV + 128);                // the CLS query's 17 probabilities (fp32)

    const int h = prob % a.H, pg = prob / a.H, b = pg / a.G, g = pg % a.G;
    const int n = a.q.n;
    const bool valid = fr < n;
    const int row = (int)(b * a.q.bs + a.q.base + g * a.q.gs) + fr * (int)a.q.is;
    const int cls = (int)(b * a.extra_bs + a.extra_row);
    const float sc2 = a.scale * T_LOG2E;
    constexpr float T_LN2 = 0.6931471805599453f;

    auto mk = [&](const void* p, unsigned int bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000); };
    const __amdgpu_buffer_rsrc_t rQ = mk(a.Q, qkv_bytes), rK = mk(a.K, qkv_bytes), rV = mk(a.V, qkv_bytes), rO = mk(a.O, o_bytes);
    auto ld = [&](__amdgpu_buffer_rsrc_t r, unsigned int off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); };
    const unsigned int oq = valid ? (unsigned int)(row * a.ldq + a.qoff + h * HD + fg * 8) * 2u : T_OOB;
    const unsigned int ok_ = valid ? (unsigned int)(row * a.ldk + a.koff + h * HD + fg * 8) * 2u : T_OOB;
    const unsigned int ov_ = valid ? (unsigned int)(row * a.ldv + a.voff + h * HD + fg * 8) * 2u : T_OOB;
    const bool c0 = fr == 0;
    const unsigned int cq = c0 ? (unsigned int)(cls * a.ldq + a.qoff + h * HD + fg * 8) * 2u : T_OOB;
    const unsigned int ck = c0 ? (unsigned int)(cls * a.ldk + a.koff + h * HD + fg * 8) * 2u : T_OOB;
    const unsigned int cv = c0 ? (unsigned int)(cls * a.ldv + a.voff + h * HD + fg * 8) * 2u : T_OOB;
    const u32x4_t q0 = ld(rQ, oq), q1 = ld(rQ, oq + 64), k0 = ld(rK, ok_), k1 = ld(rK, ok_ + 64), v0 = ld(rV, ov_), v1 = ld(rV, ov_ + 64);
    const u32x4_t qc0 = ld(rQ, cq), qc1 = ld(rQ, cq + 64), kc0 = ld(rK, ck), kc1 = ld(rK, ck + 64), vc0 = ld(rV, cv), vc1 = ld(rV, cv + 64);

    *reinterpret_cast<u32x4_t*>(sV + fr * TP + fg * 16) = v0;
    *reinterpret_cast<u32x4_t*>(sV + fr * TP + 64 + fg * 16) = v1;
    if (c0) {
        *reinterpret_cast<u32x4_t*>(cV + fg * 16) = vc0;
        *reinterpret_cast<u32x4_t*>(cV + 64 + fg * 16) = vc1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    f32x4_t s_pp = t_mfma(t_bf(k1), t_bf(q1), t_mfma(t_bf(k0), t_bf(q0), zero));       // patch keys x patch queries
    f32x4_t s_cp = t_mfma(t_bf(kc1), t_bf(q1), t_mfma(t_bf(kc0), t_bf(q0), zero));     // CLS key (row 0) x patch queries
    f32x4_t s_pc = t_mfma(t_bf(k1), t_bf(qc1), t_mfma(t_bf(k0), t_bf(qc0), zero));     // patch keys x CLS query (column 0: lanes fr == 0)
    const float s_cc = readlane_f(t_grp_sum(t_dot16(qc0, qc1, kc0, kc1)), 0);          // CLS query . CLS key
    // patch queries: softmax over [CLS key ; live patch keys] (raw scores; the positive scale commutes with max)
    float m = fg == 0 ? s_cp[0] : -INFINITY, mc = (g == 0) ? s_cc : -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool kv = fg * 4 + r < n;
        s_pp[r] = kv ? s_pp[r] : -INFINITY;
        s_pc[r] = kv ? s_pc[r] : -INFINITY;
        m = fmaxf(m, s_pp[r]);
        mc = fmaxf(mc, s_pc[r]);
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64)) * sc2;
    mc = fmaxf(mc, __shfl_xor(mc, 16, 64));
    mc = fmaxf(mc, __shfl_xor(mc, 32, 64)) * sc2;                 // (meaningful in lanes fr == 0)
    f32x4_t p_pp, p_pc, p_cp = zero, p_cc = zero;
    float l = 0.f, lc = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        p_pp[r] = exp2_fast_t(fmaf(s_pp[r], sc2, -m));
        l += p_pp[r];
        p_pc[r] = exp2_fast_t(fmaf(s_pc[r], sc2, -mc));
        lc += p_pc[r];
    }
    if (fg == 0) {
        p_cp[0] = exp2_fast_t(fmaf(s_cp[0], sc2, -m));
        l += p_cp[0];
        if (g == 0) { p_cc[0] = exp2_fast_t(fmaf(s_cc, sc2, -mc)); lc += p_cc[0]; }
    }
    l = t_grp_sum(l);
    lc = t_grp_sum(lc);
    const bf16x8_t bP = t_pack8(p_pp, p_cp), bC = t_pack8(p_pc, p_cc);
    const float inv = 1.0f / l;
    float* dst = a.ws + (((long long)g * nb + b) * a.H + h) * 66;
    // The CLS query's partial output in fp32 on the vector ALUs (17 x 64 products per group): its probabilities are NOT rounded
    // to bf16 -- the CLS row of the last block is the pooled video embedding, and the one-query launch this kernel replaces
    // (attn1_fwd_kernel) kept them in fp32 as well; bf16 probabilities here moved the bf16-mode embedding error from 1.07 x to
    // 1.14 x the reference-under-autocast's (tests/test_model_parity.py)
    if (c0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sP[fg * 4 + r] = p_pc[r];
        if (fg == 0) sP[16] = p_cc[0];
    }
    (void)bC;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        float oc = sP[16] * bf2f(*reinterpret_cast<const unsigned short*>(cV + lane * 2));
#pragma unroll
        for (int k = 0; k < 16; ++k) oc = fmaf(sP[k], bf2f(*reinterpret_cast<const unsigned short*>(sV + k * TP + lane * 2)), oc);
        dst[2 + lane] = oc;
    }
    f32x4_t o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = t_mfma(t_afrag(sV, cV, dt, fr, fg), bP, zero);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // every lane's reads of the V image are done
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] *= inv;                      // (per-lane factor: the row's 1 / l)
    t_store_rows(sV, rO, valid ? (unsigned int)(row * a.ldo + a.ooff + h * HD + fg * 8) * 2u : T_OOB, o, 1.0f, fr, fg);
    if (a.lse && valid && fg == 0) a.lse[(long long)row * a.H + h] = m * T_LN2 + __logf(l);
    if (lane == 0) { dst[0] = mc * T_LN2; dst[1] = lc; }
}

// the CLS query's G partial states (m, l, o[64]) -> its output row and lse.  One workgroup per (sample, head); the four waves take
// the groups round-robin with eight loads in flight (G = 196 for the time attention: a single wave's serial loop is 2 x 196 dependent
// round trips), combined in wave order: the result does not depend on scheduling.
constexpr int CW = 16;                          // waves of the combine workgroup
__global__ __launch_bounds__(64 * CW) void attn_cls_combine_kernel(const AttnArgs a, int nb) {
    __shared__ float red[CW][66];
    const int b = blockIdx.x, h = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long gstride = (long long)nb * a.H * 66;
    const float* base = a.ws + ((long long)b * a.H + h) * 66;
    float M = -INFINITY;
    for (int g = threadIdx.x; g < a.G; g += 64 * CW) M = fmaxf(M, base[g * gstride]);
    M = wave_max(M);
    red[w][0] = M;
    __syncthreads();
    M = red[0][0];
#pragma unroll
    for (int i = 1; i < CW; ++i) M = fmaxf(M, red[i][0]);
    __syncthreads();
    float L = 0.f, o = 0.f;
    for (int g0 = w; g0 < a.G; g0 += CW * 8) {
        float mm[8], ll[8], oo[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int g = g0 + CW * u;
            const float* src = base + (g < a.G ? g : 0) * gstride;
            mm[u] = g < a.G ? src[0] : -INFINITY; ll[u] = src[1]; oo[u] = src[2 + lane];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float c = __expf(mm[u] - M);                       // (groups past the end: exp(-inf) = 0)
            L += ll[u] * c;
            o += oo[u] * c;
        }
    }
    red[w][1] = L;
    red[w][2 + lane] = o;
    __syncthreads();
    if (w != 0) return;
    L = red[0][1];
    o = red[0][2 + lane];
#pragma unroll
    for (int i = 1; i < CW; ++i) { L += red[i][1]; o += red[i][2 + lane]; }
    const long long row = (long long)b * a.extra_bs + a.extra_row;
    reinterpret_cast<bf16_t*>(a.O)[row * a.ldo + a.ooff + h * HD + lane].v = f2bf(o / L);
    if (a.lse && lane == 0) a.lse[row * a.H + h] = M + __logf(L);
}

}  // namespace egv
using namespace egv;

// 1 if the launch was enqueued (the caller still runs attn_cls_reduce_kernel(self_term = 1) over the partials)
int egv_attn_time_bwd(const AttnArgs& a, int B, hipStream_t st) {
    const bool same = a.q.bs == a.k.bs && a.q.base == a.k.base && a.q.gs == a.k.gs && a.q.is == a.k.is && a.q.n == a.k.n;
    auto ok8 = [](int x) { return (x % 8) == 0; };
    if (!same || a.q.n > 16 || a.q.n < 1 || !a.extra || a.extra_row != 0 || !a.ws || a.mask || a.drop_p > 0.f || a.nsplit > 1) return 0;
    if (!a.O || !a.dO || !a.lse || !a.delta || !a.dQ || !a.dK || !a.dV) return 0;
    if (!(ok8(a.ldq) && ok8(a.ldk) && ok8(a.ldv) && ok8(a.ldo) && ok8(a.qoff) && ok8(a.koff) && ok8(a.voff) && ok8(a.ooff))) return 0;
    if ((a.lddq % 4) || (a.lddk % 4) || (a.lddv % 4) || (a.dqoff % 4) || (a.dkoff % 4) || (a.dvoff % 4)) return 0;
    if (a.ldq != a.ldk || a.ldq != a.ldv || a.lddq != a.lddk || a.lddq != a.lddv) return 0;
    const long long rows = (long long)B * a.extra_bs;               // rows of the token matrices (extra_bs = rows per sample)
    const long long qkv_b = rows * a.ldq * 2, o_b = rows * a.ldo * 2, dq_b = rows * a.lddq * 2;
    if (qkv_b >= (1LL << 31) || o_b >= (1LL << 31) || dq_b >= (1LL << 31)) return 0;      // 32-bit byte offsets
    const int nprob = B * a.G * a.H;
    const size_t lds = 4 * (size_t)T_WLDS;
    hipLaunchKernelGGL(attn_time_bwd_kernel, dim3((nprob + 3) / 4), dim3(256), lds, st, a, nprob, (unsigned int)qkv_b, (unsigned int)o_b,
                       (unsigned int)dq_b);
    return 1;
}

static bool time_fwd_shape_ok(const AttnArgs& a) {
    const bool same = a.q.bs == a.k.bs && a.q.base == a.k.base && a.q.gs == a.k.gs && a.q.is == a.k.is && a.q.n == a.k.n;
    auto ok8 = [](int x) { return (x % 8) == 0; };
    return same && a.q.n <= 16 && a.q.n >= 1 && a.extra && a.extra_row == 0 && a.ws && !a.mask && a.drop_p <= 0.f && a.nsplit == 1 && a.O &&
           ok8(a.ldq) && ok8(a.ldk) && ok8(a.ldv) && ok8(a.ldo) && ok8(a.qoff) && ok8(a.koff) && ok8(a.voff) && ok8(a.ooff) && a.ldq == a.ldk &&
           a.ldq == a.ldv;
}
bool egv_attn_time_fwd_ok(const AttnArgs& a, int B) {
    static const bool on = egv_cfg_on("EGV_ATTN_TIME_FUSED", true);
    if (!on || !time_fwd_shape_ok(a)) return false;
    const long long rows = (long long)B * a.extra_bs;
    return rows * a.ldq * 2 < (1LL << 31) && rows * a.ldo * 2 < (1LL << 31);
}
// 1 if enqueued: O / lse of the group rows, the CLS query's partial states in a.ws AND their combination (O / lse of the CLS row)
int egv_attn_time_fwd(const AttnArgs& a, int B, hipStream_t st) {
    if (!egv_attn_time_fwd_ok(a, B)) return 0;
    const long long rows = (long long)B * a.extra_bs;
    const int nprob = B * a.G * a.H;
    const size_t lds = 4 * (size_t)(T_IMG + 256);
    hipLaunchKernelGGL(attn_time_fwd_kernel, dim3((nprob + 3) / 4), dim3(256), lds, st, a, nprob, B, (unsigned int)(rows * a.ldq * 2),
                       (unsigned int)(rows * a.ldo * 2));
    hipLaunchKernelGGL(attn_cls_combine_kernel, dim3(B, a.H), dim3(64 * CW), 0, st, a, B);
    return 1;
}

void egv_attn_cls_combine_launch(const AttnArgs& a, int B, hipStream_t st) {
    hipLaunchKernelGGL(attn_cls_combine_kernel, dim3(B, a.H), dim3(64 * CW), 0, st, a, B);
}
