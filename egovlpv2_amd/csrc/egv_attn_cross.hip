// Attention of MANY queries over FEW keys for gfx950 (bf16, head_dim 64, <= 32 keys): the image-to-text cross attention of a fused
// SpaceTimeBlock (video_transformer.py:155-185 -- 25 096 video tokens attend to 32 text tokens under the additive text mask), forward
// in one launch and backward (dQ, dK, dV) in one launch plus a partial sum.
//
// The generic kernels (egv_attn_mfma.hip) run this shape as a query-owned dQ launch, a key-owned dK/dV launch split over 15 query
// chunks and its reduction -- 50 + 44 + 9 us for a problem whose operands are 77 MB of reads and 39 MB of writes.  Here, as in
// egv_attn_time.hip:
//   * K and V of one (sample, head) -- 32 rows x 128 bytes each -- are fetched once per wave, straight into MFMA fragment layout
//     (lane (fr, fg): row fr, 16-byte chunks fg and fg + 4), and stay in registers while the wave walks 32 queries at a time;
//   * scores are formed on the matrix pipe in the lane = query layout, S^T = K Q^T: the softmax statistics, delta and
//     dQ^T = K^T dS^T (the C layout of the first product IS the B operand of the second with the reduction index permuted:
//     [tile 0 rows fg*4..+3 | tile 1 rows fg*4..+3]; the transposed A operands come from row-major LDS images through
//     ds_read_b64_tr_b16 in the same permutation).  P and dS then go once, as bf16, through two small LDS images [query][key]
//     whose transposing reads are the B operands of dV^T = dO^T P and dK^T = Q^T dS -- the scores are not formed a second time
//     in the lane = key layout (as egv_attn_time.hip does for its 16 x 17 groups): with 32 keys per query the second softmax
//     pass was 40 % of the kernel's vector-ALU work, and the kernel is bound by exactly that;
//   * all keys of a row sit in the two tiles, so delta = sum_j P_j dP_j is formed from the products themselves (no O read, no
//     common rounding offset: DESIGN.md section 4);
//   * dK^T / dV^T accumulate in registers over the wave's queries, are summed over the workgroup's four waves through LDS and leave
//     as ONE fp32 partial per workgroup; a second small launch sums the partials in a fixed order (deterministic, no atomics).
// Rows past the end of a sample and keys past k_n are read as zeros through the buffer descriptors (out-of-range offsets); padding
// queries carry lse = +inf, padding keys a mask of -inf.
#include "egv_attn.h"
#include <cstdlib>

namespace egv {

namespace {
typedef __attribute__((ext_vector_type(4))) short x_s16x4_t;
typedef __attribute__((ext_vector_type(8))) short x_s16x8_t;
constexpr int XP = 144;                          // row pitch (bytes) of the LDS images: 64 bf16 + 16 B pad
constexpr int X_IMG = 32 * XP;                   // one 32-row operand image
constexpr int X_STAGE = 16 * XP;                 // output staging image of a wave (16 rows)
constexpr int X_NW = 4;                          // waves per workgroup
constexpr int XPP = 96;                          // row pitch of the P / dS images (32 bf16 + 32 B: conflict-free transposing reads)
constexpr int X_PIMG = 32 * XPP;
constexpr int X_WLDS = 2 * X_IMG + X_STAGE + 2 * X_PIMG;   // backward, per wave: dO and Q images, staging, P and dS of the 32 queries
constexpr int X_RED_PITCH = 68;                  // float pitch of the cross-wave sum (conflict-free f32x4 rows)
constexpr int X_BWD_LDS = X_IMG + X_NW * X_WLDS; // + the K image shared by the workgroup
constexpr int X_FWD_LDS = X_IMG + X_NW * X_STAGE;
static_assert(32 * X_RED_PITCH * 4 <= X_WLDS, "a wave's area holds one 32 x 64 fp32 matrix for the cross-wave sum");
constexpr float X_LOG2E = 1.4426950408889634f;
constexpr float X_LN2 = 0.6931471805599453f;
constexpr unsigned int X_OOB = 0x80000000u;

__device__ __forceinline__ float x_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ bf16x8_t x_bf(u32x4_t v) { return __builtin_bit_cast(bf16x8_t, v); }
__device__ __forceinline__ bf16x8_t x_pack8(const f32x4_t& a, const f32x4_t& b) {
    u32x4_t v = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
    return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ float x_grp_sum(float v) {        // over the four lane groups of a column (lanes l, l^16, l^32, l^48)
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ float x_grp_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ f32x4_t x_mfma(bf16x8_t a, bf16x8_t b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
// A operand of a second-stage product for head dims dt*16 .. +15 over 32 rows of a row-major image:
// [X^T[d][rows fg*4 .. +3] | X^T[d][rows 16 + fg*4 .. +3]]
__device__ __forceinline__ bf16x8_t x_afrag(const unsigned char* img, int dt, int fr, int fg) {
    const unsigned char* p = img + (fg * 4 + (fr >> 2)) * XP + (dt * 16 + (fr & 3) * 4) * 2;
    const x_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) x_s16x4_t*)(p));
    const x_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) x_s16x4_t*)(p + 16 * XP));
    const x_s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}
// B operand of dV^T / dK^T for keys kt*16 .. +15 from a [query][key] image: [X[q rows fg*4 .. +3][key] | X[q rows 16 + fg*4 .. +3][key]]
__device__ __forceinline__ bf16x8_t x_bfrag(const unsigned char* img, int kt, int fr, int fg) {
    const unsigned char* p = img + (fg * 4 + (fr >> 2)) * XPP + (kt * 16 + (fr & 3) * 4) * 2;
    const x_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) x_s16x4_t*)(p));
    const x_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) x_s16x4_t*)(p + 16 * XPP));
    const x_s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ void x_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// o[dt] (C layout: column = row fr of the output, head dims dt*16 + fg*4 .. +3) -> 16 rows x 128 contiguous bytes through the wave's
// staging image: a store instruction writes 16 rows x 64 contiguous bytes
__device__ __forceinline__ void x_store_rows(unsigned char* img, __amdgpu_buffer_rsrc_t r, unsigned int off, const f32x4_t (&o)[4], float s,
                                             int fr, int fg) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const u32x2_t pk = {pack_bf16x2(o[dt][0] * s, o[dt][1] * s), pack_bf16x2(o[dt][2] * s, o[dt][3] * s)};
        *reinterpret_cast<u32x2_t*>(img + fr * XP + dt * 32 + fg * 8) = pk;
    }
    x_wave_sync();
    const u32x4_t lo = *reinterpret_cast<const u32x4_t*>(img + fr * XP + fg * 16), hi = *reinterpret_cast<const u32x4_t*>(img + fr * XP + 64 + fg * 16);
    __builtin_amdgcn_raw_buffer_store_b128(lo, r, off, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(hi, r, off + (off == X_OOB ? 0u : 64u), 0, 0);
}
}  // namespace

// grid: (workgroups per (sample, group, head), B*G*H); a workgroup's wave w takes the 32-query tiles (it * 4 + w), it < iters
__global__ __launch_bounds__(256) void attn_fewkeys_fwd_kernel(const AttnArgs a, int iters, unsigned int q_bytes, unsigned int kv_bytes,
                                                               unsigned int o_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, w = wave_id();
    const int fr = lane & 15, fg = lane >> 4;
    const int h = blockIdx.y % a.H, pg = blockIdx.y / a.H, b = pg / a.G, g = pg % a.G;
    unsigned char* sV = smem;
    unsigned char* sS = smem + X_IMG + w * X_STAGE;
    const int nq = a.q.n, nk = a.k.n;
    const float sc2 = a.scale * X_LOG2E;
    const int krow0 = (int)(b * a.k.bs + a.k.base + g * a.k.gs), qrow0 = (int)(b * a.q.bs + a.q.base + g * a.q.gs);

    auto mk = [&](const void* p, unsigned int bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000); };
    const __amdgpu_buffer_rsrc_t rQ = mk(a.Q, q_bytes), rK = mk(a.K, kv_bytes), rV = mk(a.V, kv_bytes), rO = mk(a.O, o_bytes);
    auto ld = [&](__amdgpu_buffer_rsrc_t r, unsigned int off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); };
    // K / V: 8 KB that every workgroup of the (sample, head) reads -- at agent scope (sc1), see wgrad_small_m_kernel in egv_gemm.hip
    auto ld_kv = [&](__amdgpu_buffer_rsrc_t r, unsigned int off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16); };

    u32x4_t k[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int key = kt * 16 + fr;
        const unsigned int off = key < nk ? (unsigned int)((krow0 + key) * a.ldk + a.koff + h * HD + fg * 8) * 2u : X_OOB;
        k[kt][0] = ld_kv(rK, off);
        k[kt][1] = ld_kv(rK, off + 64);
    }
    if (w < 2) {                                                    // the V image: waves 0 and 1 bring one 16-row tile each
        const int key = w * 16 + fr;
        const unsigned int off = key < nk ? (unsigned int)((krow0 + key) * a.ldv + a.voff + h * HD + fg * 8) * 2u : X_OOB;
        const u32x4_t v0 = ld_kv(rV, off), v1 = ld_kv(rV, off + 64);
        *reinterpret_cast<u32x4_t*>(sV + key * XP + fg * 16) = v0;
        *reinterpret_cast<u32x4_t*>(sV + key * XP + 64 + fg * 16) = v1;
    }
    // additive mask (log2 domain) of the lane's keys kt*16 + fg*4 + r: descriptor loads, all in flight together (a conditional
    // global load per key compiles to a branch and a full wait each: ten serial round trips before the first query)
    const __amdgpu_buffer_rsrc_t rM = mk(a.mask ? (const void*)(a.mask + (long long)b * a.mask_ld) : (const void*)a.K, a.mask ? (unsigned int)nk * 4u : 0u);
    float mq[2][4];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mq[kt][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rM, (unsigned int)(kt * 16 + fg * 4 + r) * 4u, 0, 0));
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mq[kt][r] = kt * 16 + fg * 4 + r < nk ? mq[kt][r] * X_LOG2E : -INFINITY;
    __syncthreads();
    bf16x8_t av[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) av[dt] = x_afrag(sV, dt, fr, fg);

    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    const int q_begin = blockIdx.x * (X_NW * iters * 32);
    for (int it = 0; it < iters; ++it) {
        const int q0 = q_begin + (it * X_NW + w) * 32;
        if (q0 >= nq) break;
        u32x4_t q[2][2];
        unsigned int oo[2];
        int rows[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int qi = q0 + t * 16 + fr;
            const bool valid = qi < nq;
            rows[t] = valid ? qrow0 + qi : -1;
            const unsigned int oq = valid ? (unsigned int)((qrow0 + qi) * a.ldq + a.qoff + h * HD + fg * 8) * 2u : X_OOB;
            oo[t] = valid ? (unsigned int)((qrow0 + qi) * a.ldo + a.ooff + h * HD + fg * 8) * 2u : X_OOB;
            q[t][0] = ld(rQ, oq);
            q[t][1] = ld(rQ, oq + 64);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4_t s0 = x_mfma(x_bf(k[0][1]), x_bf(q[t][1]), x_mfma(x_bf(k[0][0]), x_bf(q[t][0]), zero));     // keys 0..15 x the tile's queries
            f32x4_t s1 = x_mfma(x_bf(k[1][1]), x_bf(q[t][1]), x_mfma(x_bf(k[1][0]), x_bf(q[t][0]), zero));     // keys 16..31
            float m = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s0[r] = fmaf(s0[r], sc2, mq[0][r]);
                s1[r] = fmaf(s1[r], sc2, mq[1][r]);
                m = fmaxf(m, fmaxf(s0[r], s1[r]));
            }
            m = x_grp_max(m);
            if (m == -INFINITY) m = 0.f;                              // (a row whose keys are all masked out by -inf: zeros, not NaN)
            f32x4_t p0, p1;
            float l = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p0[r] = x_exp2(s0[r] - m);
                p1[r] = x_exp2(s1[r] - m);
                l += p0[r] + p1[r];
            }
            l = x_grp_sum(l);
            const bf16x8_t bP = x_pack8(p0, p1);
            f32x4_t o[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = x_mfma(av[dt], bP, zero);
            x_store_rows(sS, rO, oo[t], o, l > 0.f ? 1.0f / l : 0.f, fr, fg);
            if (a.lse && rows[t] >= 0 && fg == 0) a.lse[(long long)rows[t] * a.H + h] = m * X_LN2 + __logf(l);
        }
    }
}

// grid as above; part: fp32 partial dK^T-then-dV^T sums, [problem][workgroup][dK | dV][32 keys][64]
__global__ __launch_bounds__(256) void attn_fewkeys_bwd_kernel(const AttnArgs a, int iters, unsigned int q_bytes, unsigned int kv_bytes,
                                                               unsigned int o_bytes, unsigned int dq_bytes, unsigned int lse_bytes, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, w = wave_id();
    const int fr = lane & 15, fg = lane >> 4;
    const int h = blockIdx.y % a.H, pg = blockIdx.y / a.H, b = pg / a.G, g = pg % a.G;
    unsigned char* sK = smem;
    unsigned char* sG = smem + X_IMG + w * X_WLDS;                  // dO rows of the wave's 32 queries
    unsigned char* sQ = sG + X_IMG;
    unsigned char* sS = sQ + X_IMG;
    unsigned char* sP = sS + X_STAGE;                               // P and dS of the wave's 32 queries, [query][key] bf16
    unsigned char* sDS = sP + X_PIMG;
    const int nq = a.q.n, nk = a.k.n;
    const float sc2 = a.scale * X_LOG2E;
    const int krow0 = (int)(b * a.k.bs + a.k.base + g * a.k.gs), qrow0 = (int)(b * a.q.bs + a.q.base + g * a.q.gs);

    auto mk = [&](const void* p, unsigned int bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000); };
    const __amdgpu_buffer_rsrc_t rQ = mk(a.Q, q_bytes), rK = mk(a.K, kv_bytes), rV = mk(a.V, kv_bytes), rG = mk(a.dO, o_bytes);
    const __amdgpu_buffer_rsrc_t rDQ = mk(a.dQ, dq_bytes);
    auto ld = [&](__amdgpu_buffer_rsrc_t r, unsigned int off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); };
    // K / V: 8 KB that every workgroup of the (sample, head) reads -- at agent scope (sc1), see wgrad_small_m_kernel in egv_gemm.hip
    auto ld_kv = [&](__amdgpu_buffer_rsrc_t r, unsigned int off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16); };

    u32x4_t k[2][2], v[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int key = kt * 16 + fr;
        const unsigned int ok_ = key < nk ? (unsigned int)((krow0 + key) * a.ldk + a.koff + h * HD + fg * 8) * 2u : X_OOB;
        const unsigned int ov_ = key < nk ? (unsigned int)((krow0 + key) * a.ldv + a.voff + h * HD + fg * 8) * 2u : X_OOB;
        k[kt][0] = ld_kv(rK, ok_);
        k[kt][1] = ld_kv(rK, ok_ + 64);
        v[kt][0] = ld_kv(rV, ov_);
        v[kt][1] = ld_kv(rV, ov_ + 64);
    }
    if (w < 2) {                                                    // the K image (for K^T): waves 0 and 1 write one tile each
        *reinterpret_cast<u32x4_t*>(sK + (w * 16 + fr) * XP + fg * 16) = w == 0 ? k[0][0] : k[1][0];
        *reinterpret_cast<u32x4_t*>(sK + (w * 16 + fr) * XP + 64 + fg * 16) = w == 0 ? k[0][1] : k[1][1];
    }
    // additive mask (log2 domain) of the lane's keys kt*16 + fg*4 + r.  Descriptor loads, all in flight together (no mask: a
    // zero-length descriptor reads zeros)
    const __amdgpu_buffer_rsrc_t rM = mk(a.mask ? (const void*)(a.mask + (long long)b * a.mask_ld) : (const void*)a.K, a.mask ? (unsigned int)nk * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rL = mk(a.lse, lse_bytes);
    float mq[2][4];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mq[kt][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rM, (unsigned int)(kt * 16 + fg * 4 + r) * 4u, 0, 0));
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mq[kt][r] = kt * 16 + fg * 4 + r < nk ? mq[kt][r] * X_LOG2E : -INFINITY;
    __syncthreads();

    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    f32x4_t dk[2][4], dv[2][4];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dk[kt][dt] = zero; dv[kt][dt] = zero; }

    const int q_begin = blockIdx.x * (X_NW * iters * 32);
    // the wave's operands of one trip: Q and dO rows of 32 queries in fragment layout, their lse, the dQ store offsets.  The NEXT trip's
    // loads are issued before this trip's arithmetic (the kernel is bound by the latency of a trip, two waves per SIMD)
    struct Trip { u32x4_t q[2][2], g[2][2]; float lse2[2]; unsigned int odq[2]; };
    auto fetch = [&](int q0, Trip& tr) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int qi = q0 + t * 16 + fr;
            const bool valid = qi < nq;
            const int row = qrow0 + qi;
            const unsigned int oq = valid ? (unsigned int)(row * a.ldq + a.qoff + h * HD + fg * 8) * 2u : X_OOB;
            const unsigned int og = valid ? (unsigned int)(row * a.ldo + a.ooff + h * HD + fg * 8) * 2u : X_OOB;
            tr.odq[t] = valid ? (unsigned int)(row * a.lddq + a.dqoff + h * HD + fg * 8) * 2u : X_OOB;
            tr.q[t][0] = ld(rQ, oq);
            tr.q[t][1] = ld(rQ, oq + 64);
            tr.g[t][0] = ld(rG, og);
            tr.g[t][1] = ld(rG, og + 64);
            tr.lse2[t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rL, valid ? (unsigned int)(row * a.H + h) * 4u : X_OOB, 0, 0));   // (no branch, no wait)
        }
    };
    Trip nx;
    if (q_begin + w * 32 < nq) fetch(q_begin + w * 32, nx);
    for (int it = 0; it < iters; ++it) {
        const int q0 = q_begin + (it * X_NW + w) * 32;
        if (q0 >= nq) break;                                        // (wave-uniform; the wave still takes part in the sum below)
        u32x4_t q[2][2], gq[2][2];
        float lse2[2];
        unsigned int odq[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            q[t][0] = nx.q[t][0]; q[t][1] = nx.q[t][1]; gq[t][0] = nx.g[t][0]; gq[t][1] = nx.g[t][1];
            lse2[t] = nx.odq[t] != X_OOB ? nx.lse2[t] * X_LOG2E : INFINITY;      // padding queries: exp2(s - inf) = 0
            odq[t] = nx.odq[t];
        }
        if (it + 1 < iters && q0 + X_NW * 32 < nq) fetch(q0 + X_NW * 32, nx);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            *reinterpret_cast<u32x4_t*>(sQ + (t * 16 + fr) * XP + fg * 16) = q[t][0];
            *reinterpret_cast<u32x4_t*>(sQ + (t * 16 + fr) * XP + 64 + fg * 16) = q[t][1];
            *reinterpret_cast<u32x4_t*>(sG + (t * 16 + fr) * XP + fg * 16) = gq[t][0];
            *reinterpret_cast<u32x4_t*>(sG + (t * 16 + fr) * XP + 64 + fg * 16) = gq[t][1];
        }
        // ================= lane = query: S^T[key][query] -> delta, dQ^T = K^T dS^T =================
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4_t s[2], d[2], ds[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                s[kt] = x_mfma(x_bf(k[kt][1]), x_bf(q[t][1]), x_mfma(x_bf(k[kt][0]), x_bf(q[t][0]), zero));
                d[kt] = x_mfma(x_bf(v[kt][1]), x_bf(gq[t][1]), x_mfma(x_bf(v[kt][0]), x_bf(gq[t][0]), zero));
            }
            float dl = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = x_exp2(fmaf(s[kt][r], sc2, mq[kt][r] - lse2[t]));
                    s[kt][r] = p;
                    dl = fmaf(p, d[kt][r], dl);
                }
            dl = x_grp_sum(dl);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) ds[kt][r] = s[kt][r] * (d[kt][r] - dl);
            const bf16x8_t bP = x_pack8(ds[0], ds[1]);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {                         // row t*16 + fr, keys kt*16 + fg*4 .. +3
                const u32x4_t dsv = __builtin_bit_cast(u32x4_t, bP);
                *reinterpret_cast<u32x2_t*>(sP + (t * 16 + fr) * XPP + (kt * 16 + fg * 4) * 2) = u32x2_t{pack_bf16x2(s[kt][0], s[kt][1]), pack_bf16x2(s[kt][2], s[kt][3])};
                *reinterpret_cast<u32x2_t*>(sDS + (t * 16 + fr) * XPP + (kt * 16 + fg * 4) * 2) = u32x2_t{dsv[kt * 2], dsv[kt * 2 + 1]};
            }
            f32x4_t dq[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dq[dt] = x_mfma(x_afrag(sK, dt, fr, fg), bP, zero);
            x_store_rows(sS, rDQ, odq[t], dq, a.scale, fr, fg);
        }
        x_wave_sync();
        // ================= dV^T = dO^T P, dK^T = Q^T dS: P and dS of the 32 queries from their images, lane = key =================
        bf16x8_t bV[2], bK[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            bV[kt] = x_bfrag(sP, kt, fr, fg);
            bK[kt] = x_bfrag(sDS, kt, fr, fg);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x8_t ag = x_afrag(sG, dt, fr, fg), aq = x_afrag(sQ, dt, fr, fg);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                dv[kt][dt] = x_mfma(ag, bV[kt], dv[kt][dt]);
                dk[kt][dt] = x_mfma(aq, bK[kt], dk[kt][dt]);
            }
        }
        x_wave_sync();                                               // the images are rewritten by the next trip
    }

    // ---- sum of the four waves' dK^T, then dV^T (lane (fr, fg): key kt*16 + fr, head dims dt*16 + fg*4 .. +3), one partial per workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem + X_IMG + w * X_WLDS);
    float* dst = part + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 2 * 32 * HD;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<f32x4_t*>(red + (kt * 16 + fr) * X_RED_PITCH + dt * 16 + fg * 4) = pass == 0 ? dk[kt][dt] : dv[kt][dt];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = threadIdx.x + j * 256, key = e >> 4, c4 = e & 15;
            f32x4_t s = zero;
#pragma unroll
            for (int ww = 0; ww < X_NW; ++ww)
                s += *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(smem + X_IMG + ww * X_WLDS) + key * X_RED_PITCH + c4 * 4);
            *reinterpret_cast<f32x4_t*>(dst + pass * 32 * HD + key * HD + c4 * 4) = s;
        }
        __syncthreads();
    }
}

// four workgroups per (sample, group, head), one float4 of dK or dV per thread: dK = scale * sum of the partials, dV = their sum, in
// workgroup order (eight loads in flight)
__global__ __launch_bounds__(256) void attn_fewkeys_reduce_kernel(const AttnArgs a, int nwg, const float* __restrict__ part) {
    const int prob = blockIdx.x >> 2;
    const int h = prob % a.H, pg = prob / a.H, b = pg / a.G, g = pg % a.G;
    const long long krow0 = b * a.k.bs + a.k.base + g * a.k.gs;
    const int e = (blockIdx.x & 3) * 256 + threadIdx.x;             // float4 index over [dK | dV][32 keys][16]
    const int pass = e >> 9, key = (e & 511) >> 4, c4 = e & 15;
    if (key >= a.k.n) return;
    const float* src = part + (long long)prob * nwg * 2 * 32 * HD + pass * 32 * HD + key * HD + c4 * 4;
    f32x4_t s = {0.f, 0.f, 0.f, 0.f};
    for (int w0 = 0; w0 < nwg; w0 += 8) {
        f32x4_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4_t*>(src + (long long)(w0 + u < nwg ? w0 + u : w0) * 2 * 32 * HD);
#pragma unroll
        for (int u = 0; u < 8; ++u) if (w0 + u < nwg) s += v[u];
    }
    const float sc = pass == 0 ? a.scale : 1.0f;
    const u32x2_t pk = {pack_bf16x2(s[0] * sc, s[1] * sc), pack_bf16x2(s[2] * sc, s[3] * sc)};
    bf16_t* out = reinterpret_cast<bf16_t*>(pass == 0 ? a.dK : a.dV);
    const long long off = (krow0 + key) * (pass == 0 ? a.lddk : a.lddv) + (pass == 0 ? a.dkoff : a.dvoff) + h * HD + c4 * 4;
    *reinterpret_cast<u32x2_t*>(out + off) = pk;
}


// =====================================================================================================================
// The mirror image: FEW queries (<= 32) over MANY keys -- the text-to-image cross attention of a fused RobertaLayer (roberta.py:241-327:
// 32 text tokens attend to the 25 096 video tokens of their sample, attention-probability dropout :313, no mask).  Queries (and, in
// the backward, dO, lse and delta) sit in registers as the B operands of S^T = K Q^T / dP^T = V dO^T; a wave streams 32 keys per trip
// with K and V rows fetched straight into fragment layout.  Forward: online softmax per query (the running maximum is shared by the
// four lane groups of a query, so one rescale factor serves its whole output row), O^T += V^T (P o D)^T with V^T from an LDS image;
// wave states are combined per workgroup and leave as (m, l, o[64]) partials, a second launch combines them and writes O (bf16, and
// fp32 for the backward's delta), lse.  Backward: dS^T in the lane = query layout gives dQ^T += K^T dS^T (accumulated over the wave's keys,
// summed like dK / dV above); P o D and dS go through the [query][key] LDS images and come back as the B operands of dV^T = dO^T (P o D),
// dK^T = Q^T dS, which are stored per key row.  delta = rowsum(dO o O) from the fp32 O.
namespace {
constexpr int Y_WLDS_F = X_IMG + 32 * X_RED_PITCH * 4;                      // forward, per wave: V image + (m, l, -, -, o[64]) of 32 queries
constexpr int Y_FWD_LDS = X_NW * Y_WLDS_F;
constexpr int Y_WLDS_B = X_IMG + 2 * X_PIMG + X_STAGE;                      // backward, per wave: K image, P o D and dS images, staging
constexpr int Y_BWD_LDS = 2 * X_IMG + X_NW * Y_WLDS_B;                      // + Q and dO images shared by the workgroup
static_assert(32 * X_RED_PITCH * 4 <= Y_WLDS_B, "a wave's area holds the 32 x 64 fp32 dQ for the cross-wave sum");
__device__ __forceinline__ unsigned int y_fmix(unsigned int h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
// drop_mult(a, qrow, krow, h) of egv_attn.h with the query part of the hash (hq) hoisted
__device__ __forceinline__ float y_drop(unsigned int hq, int krow, float p, float keep) {
    const unsigned int x = y_fmix(hq ^ ((unsigned int)krow * 0x85EBCA77u + 0x165667B1u));
    return (float)(x >> 8) * (1.0f / 16777216.0f) >= p ? keep : 0.0f;
}
}  // namespace

// grid: (workgroups per problem, B*G*H); wave w of a workgroup takes the 32-key tiles (it * 4 + w), it < iters.
// part: [problem][workgroup][32 queries][68]: m (log2 domain), l, -, -, o[64] (unnormalised)
__global__ __launch_bounds__(256) void attn_fewq_fwd_kernel(const AttnArgs a, int iters, unsigned int q_bytes, unsigned int kv_bytes,
                                                            float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, w = wave_id();
    const int fr = lane & 15, fg = lane >> 4;
    const int h = blockIdx.y % a.H, pg = blockIdx.y / a.H, b = pg / a.G, g = pg % a.G;
    unsigned char* sV = smem + w * Y_WLDS_F;
    float* sT = reinterpret_cast<float*>(sV + X_IMG);
    const int nq = a.q.n, nk = a.k.n;
    const float sc2 = a.scale * X_LOG2E;
    const int krow0 = (int)(b * a.k.bs + a.k.base + g * a.k.gs), qrow0 = (int)(b * a.q.bs + a.q.base + g * a.q.gs);
    auto mk = [&](const void* p, unsigned int bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000); };
    const __amdgpu_buffer_rsrc_t rQ = mk(a.Q, q_bytes), rK = mk(a.K, kv_bytes), rV = mk(a.V, kv_bytes);
    auto ld = [&](__amdgpu_buffer_rsrc_t r, unsigned int off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); };
    auto ld_q = [&](__amdgpu_buffer_rsrc_t r, unsigned int off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16); };   // agent scope: every workgroup of the problem reads them

    u32x4_t q[2][2];
    unsigned int hq[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int qi = t * 16 + fr;
        const unsigned int off = qi < nq ? (unsigned int)((qrow0 + qi) * a.ldq + a.qoff + h * HD + fg * 8) * 2u : X_OOB;
        q[t][0] = ld_q(rQ, off);
        q[t][1] = ld_q(rQ, off + 64);
        hq[t] = a.drop_seed ^ y_fmix((unsigned int)(pg * nq + qi) * 0x9E3779B1u + (unsigned int)h);
    }
    const bool drop = a.drop_p > 0.f;
    const float keep = drop ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
    f32x4_t o[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[t][dt] = zero;

    struct Trip { u32x4_t k[2][2], v[2][2]; };
    auto fetch = [&](int key0, Trip& tr) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int key = key0 + kt * 16 + fr;
            const unsigned int ok_ = key < nk ? (unsigned int)((krow0 + key) * a.ldk + a.koff + h * HD + fg * 8) * 2u : X_OOB;
            const unsigned int ov_ = key < nk ? (unsigned int)((krow0 + key) * a.ldv + a.voff + h * HD + fg * 8) * 2u : X_OOB;
            tr.k[kt][0] = ld(rK, ok_);
            tr.k[kt][1] = ld(rK, ok_ + 64);
            tr.v[kt][0] = ld(rV, ov_);
            tr.v[kt][1] = ld(rV, ov_ + 64);
        }
    };
    const int k_begin = blockIdx.x * (X_NW * iters * 32);
    Trip nx;
    if (k_begin + w * 32 < nk) fetch(k_begin + w * 32, nx);
    for (int it = 0; it < iters; ++it) {
        const int key0 = k_begin + (it * X_NW + w) * 32;
        if (key0 >= nk) break;
        u32x4_t k[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            k[kt][0] = nx.k[kt][0]; k[kt][1] = nx.k[kt][1];
            *reinterpret_cast<u32x4_t*>(sV + (kt * 16 + fr) * XP + fg * 16) = nx.v[kt][0];
            *reinterpret_cast<u32x4_t*>(sV + (kt * 16 + fr) * XP + 64 + fg * 16) = nx.v[kt][1];
        }
        if (it + 1 < iters && key0 + X_NW * 32 < nk) fetch(key0 + X_NW * 32, nx);
        x_wave_sync();
        bf16x8_t av[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) av[dt] = x_afrag(sV, dt, fr, fg);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4_t s[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) s[kt] = x_mfma(x_bf(k[kt][1]), x_bf(q[t][1]), x_mfma(x_bf(k[kt][0]), x_bf(q[t][0]), zero));
            float mt = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[kt][r] = key0 + kt * 16 + fg * 4 + r < nk ? s[kt][r] * sc2 : -INFINITY;
                    mt = fmaxf(mt, s[kt][r]);
                }
            mt = x_grp_max(mt);                                      // (key0 < nk: at least one live key, the maximum is finite)
            const float mn = fmaxf(m[t], mt);
            const float alpha = x_exp2(m[t] - mn);
            m[t] = mn;
            float ls = 0.f;
            f32x4_t pd[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = x_exp2(s[kt][r] - mn);
                    ls += p;
                    pd[kt][r] = drop ? p * y_drop(hq[t], key0 + kt * 16 + fg * 4 + r, a.drop_p, keep) : p;
                }
            l[t] = fmaf(l[t], alpha, ls);                            // (this lane group's keys; summed over the groups at the end)
            const bf16x8_t bP = x_pack8(pd[0], pd[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[t][dt] = x_mfma(av[dt], bP, o[t][dt] * alpha);
        }
        x_wave_sync();                                               // the V image is rewritten by the next trip
    }
    // ---- wave states -> workgroup state -> partial
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float lt = x_grp_sum(l[t]);
        float* row = sT + (t * 16 + fr) * X_RED_PITCH;
        if (fg == 0) { row[0] = m[t]; row[1] = lt; }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4_t*>(row + 4 + dt * 16 + fg * 4) = o[t][dt];
    }
    __syncthreads();
    {
        const int qi = threadIdx.x >> 3, c8 = (threadIdx.x & 7) * 8;
        float M = -INFINITY;
#pragma unroll
        for (int ww = 0; ww < X_NW; ++ww) M = fmaxf(M, reinterpret_cast<const float*>(smem + ww * Y_WLDS_F + X_IMG)[qi * X_RED_PITCH]);
        float L = 0.f;
        f32x4_t o0 = zero, o1 = zero;
#pragma unroll
        for (int ww = 0; ww < X_NW; ++ww) {
            const float* row = reinterpret_cast<const float*>(smem + ww * Y_WLDS_F + X_IMG) + qi * X_RED_PITCH;
            const float f = row[0] == -INFINITY ? 0.f : x_exp2(row[0] - M);
            L = fmaf(row[1], f, L);
            o0 += *reinterpret_cast<const f32x4_t*>(row + 4 + c8) * f;
            o1 += *reinterpret_cast<const f32x4_t*>(row + 4 + c8 + 4) * f;
        }
        float* dst = part + (((long long)blockIdx.y * gridDim.x + blockIdx.x) * 32 + qi) * X_RED_PITCH;
        if (c8 == 0) { dst[0] = M; dst[1] = L; }
        *reinterpret_cast<f32x4_t*>(dst + 4 + c8) = o0;
        *reinterpret_cast<f32x4_t*>(dst + 4 + c8 + 4) = o1;
    }
}

// one workgroup per problem: O = sum_w o_w 2^(m_w - M) / sum_w l_w 2^(m_w - M), lse; thread = (query, 8 head dims)
__global__ __launch_bounds__(256) void attn_fewq_combine_kernel(const AttnArgs a, int nwg, const float* __restrict__ part) {
    const int h = blockIdx.x % a.H, pg = blockIdx.x / a.H, b = pg / a.G, g = pg % a.G;
    const int qi = threadIdx.x >> 3, c8 = (threadIdx.x & 7) * 8;
    if (qi >= a.q.n) return;
    const float* src = part + ((long long)blockIdx.x * nwg * 32 + qi) * X_RED_PITCH;
    const long long wstride = 32LL * X_RED_PITCH;
    float M = -INFINITY;
    for (int wg = 0; wg < nwg; ++wg) M = fmaxf(M, src[wg * wstride]);
    float L = 0.f;
    f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
    for (int wg = 0; wg < nwg; ++wg) {
        const float* row = src + wg * wstride;
        const float f = row[0] == -INFINITY ? 0.f : x_exp2(row[0] - M);
        L = fmaf(row[1], f, L);
        o0 += *reinterpret_cast<const f32x4_t*>(row + 4 + c8) * f;
        o1 += *reinterpret_cast<const f32x4_t*>(row + 4 + c8 + 4) * f;
    }
    const float inv = 1.0f / L;
    o0 *= inv; o1 *= inv;
    const long long row = b * a.q.bs + a.q.base + g * a.q.gs + qi;
    const long long off = row * a.ldo + a.ooff + h * HD + c8;
    const u32x4_t pk = {pack_bf16x2(o0[0], o0[1]), pack_bf16x2(o0[2], o0[3]), pack_bf16x2(o1[0], o1[1]), pack_bf16x2(o1[2], o1[3])};
    *reinterpret_cast<u32x4_t*>(reinterpret_cast<bf16_t*>(a.O) + off) = pk;
    if (a.O32) {
        *reinterpret_cast<f32x4_t*>(a.O32 + off) = o0;
        *reinterpret_cast<f32x4_t*>(a.O32 + off + 4) = o1;
    }
    if (a.lse && c8 == 0) a.lse[row * a.H + h] = (M + __log2f(L)) * X_LN2;
}

// backward: dK, dV per streamed key row, dQ as one fp32 partial per workgroup ([problem][workgroup][32 queries][64])
__global__ __launch_bounds__(256) void attn_fewq_bwd_kernel(const AttnArgs a, int iters, unsigned int q_bytes, unsigned int kv_bytes,
                                                            unsigned int o_bytes, unsigned int dkv_bytes, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, w = wave_id();
    const int fr = lane & 15, fg = lane >> 4;
    const int h = blockIdx.y % a.H, pg = blockIdx.y / a.H, b = pg / a.G, g = pg % a.G;
    unsigned char* sQ = smem;                                       // shared: Q and dO rows of the 32 queries
    unsigned char* sG = smem + X_IMG;
    unsigned char* sK = smem + 2 * X_IMG + w * Y_WLDS_B;            // per wave: K rows of the trip, P o D, dS, staging
    unsigned char* sP = sK + X_IMG;
    unsigned char* sDS = sP + X_PIMG;
    unsigned char* sS = sDS + X_PIMG;
    const int nq = a.q.n, nk = a.k.n;
    const float sc2 = a.scale * X_LOG2E;
    const int krow0 = (int)(b * a.k.bs + a.k.base + g * a.k.gs), qrow0 = (int)(b * a.q.bs + a.q.base + g * a.q.gs);
    auto mk = [&](const void* p, unsigned int bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000); };
    const __amdgpu_buffer_rsrc_t rQ = mk(a.Q, q_bytes), rK = mk(a.K, kv_bytes), rV = mk(a.V, kv_bytes), rG = mk(a.dO, o_bytes), rO = mk(a.O, o_bytes);
    const __amdgpu_buffer_rsrc_t rDK = mk(a.dK, dkv_bytes), rDV = mk(a.dV, dkv_bytes);
    auto ld = [&](__amdgpu_buffer_rsrc_t r, unsigned int off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); };
    auto ld_q = [&](__amdgpu_buffer_rsrc_t r, unsigned int off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16); };

    u32x4_t q[2][2], gq[2][2];
    float lse2[2], dl[2];
    unsigned int hq[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int qi = t * 16 + fr;
        const bool valid = qi < nq;
        const int row = qrow0 + qi;
        const unsigned int oq = valid ? (unsigned int)(row * a.ldq + a.qoff + h * HD + fg * 8) * 2u : X_OOB;
        const unsigned int og = valid ? (unsigned int)(row * a.ldo + a.ooff + h * HD + fg * 8) * 2u : X_OOB;
        q[t][0] = ld_q(rQ, oq);
        q[t][1] = ld_q(rQ, oq + 64);
        gq[t][0] = ld_q(rG, og);
        gq[t][1] = ld_q(rG, og + 64);
        hq[t] = a.drop_seed ^ y_fmix((unsigned int)(pg * nq + qi) * 0x9E3779B1u + (unsigned int)h);
        // delta = rowsum(dO o O): this lane holds head dims fg*8 .. +7 and 32 + fg*8 .. +7 of query qi; O in fp32 when the forward kept it
        float d = 0.f;
        float ov[16];
        if (a.O32) {
            const float* op = a.O32 + ((long long)(valid ? row : qrow0) * a.ldo + a.ooff + h * HD + fg * 8);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const f32x4_t v4 = *reinterpret_cast<const f32x4_t*>(op + c * 32 + e * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) ov[c * 8 + e * 4 + j] = v4[j];
                }
        } else {
            const u32x4_t o0 = ld_q(rO, og), o1 = ld_q(rO, og + 64);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ov[2 * j] = __uint_as_float(o0[j] << 16); ov[2 * j + 1] = __uint_as_float(o0[j] & 0xffff0000u);
                ov[8 + 2 * j] = __uint_as_float(o1[j] << 16); ov[8 + 2 * j + 1] = __uint_as_float(o1[j] & 0xffff0000u);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            d = fmaf(__uint_as_float(gq[t][0][j] << 16), ov[2 * j], d);
            d = fmaf(__uint_as_float(gq[t][0][j] & 0xffff0000u), ov[2 * j + 1], d);
            d = fmaf(__uint_as_float(gq[t][1][j] << 16), ov[8 + 2 * j], d);
            d = fmaf(__uint_as_float(gq[t][1][j] & 0xffff0000u), ov[8 + 2 * j + 1], d);
        }
        dl[t] = valid ? x_grp_sum(d) : 0.f;
        lse2[t] = valid ? a.lse[(long long)row * a.H + h] * X_LOG2E : INFINITY;      // (prologue: one round trip per wave)
    }
    if (w < 2) {
        *reinterpret_cast<u32x4_t*>(sQ + (w * 16 + fr) * XP + fg * 16) = w == 0 ? q[0][0] : q[1][0];
        *reinterpret_cast<u32x4_t*>(sQ + (w * 16 + fr) * XP + 64 + fg * 16) = w == 0 ? q[0][1] : q[1][1];
    } else {
        *reinterpret_cast<u32x4_t*>(sG + ((w - 2) * 16 + fr) * XP + fg * 16) = w == 2 ? gq[0][0] : gq[1][0];
        *reinterpret_cast<u32x4_t*>(sG + ((w - 2) * 16 + fr) * XP + 64 + fg * 16) = w == 2 ? gq[0][1] : gq[1][1];
    }
    __syncthreads();
    const bool drop = a.drop_p > 0.f;
    const float keep = drop ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    f32x4_t dq[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[t][dt] = zero;

    struct Trip { u32x4_t k[2][2], v[2][2]; unsigned int odk[2], odv[2]; };
    auto fetch = [&](int key0, Trip& tr) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int key = key0 + kt * 16 + fr;
            const bool valid = key < nk;
            const unsigned int ok_ = valid ? (unsigned int)((krow0 + key) * a.ldk + a.koff + h * HD + fg * 8) * 2u : X_OOB;
            const unsigned int ov_ = valid ? (unsigned int)((krow0 + key) * a.ldv + a.voff + h * HD + fg * 8) * 2u : X_OOB;
            tr.odk[kt] = valid ? (unsigned int)((krow0 + key) * a.lddk + a.dkoff + h * HD + fg * 8) * 2u : X_OOB;
            tr.odv[kt] = valid ? (unsigned int)((krow0 + key) * a.lddv + a.dvoff + h * HD + fg * 8) * 2u : X_OOB;
            tr.k[kt][0] = ld(rK, ok_);
            tr.k[kt][1] = ld(rK, ok_ + 64);
            tr.v[kt][0] = ld(rV, ov_);
            tr.v[kt][1] = ld(rV, ov_ + 64);
        }
    };
    const int k_begin = blockIdx.x * (X_NW * iters * 32);
    Trip nx;
    if (k_begin + w * 32 < nk) fetch(k_begin + w * 32, nx);
    for (int it = 0; it < iters; ++it) {
        const int key0 = k_begin + (it * X_NW + w) * 32;
        if (key0 >= nk) break;
        u32x4_t k[2][2], v[2][2];
        unsigned int odk[2], odv[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            k[kt][0] = nx.k[kt][0]; k[kt][1] = nx.k[kt][1]; v[kt][0] = nx.v[kt][0]; v[kt][1] = nx.v[kt][1];
            odk[kt] = nx.odk[kt]; odv[kt] = nx.odv[kt];
            *reinterpret_cast<u32x4_t*>(sK + (kt * 16 + fr) * XP + fg * 16) = k[kt][0];
            *reinterpret_cast<u32x4_t*>(sK + (kt * 16 + fr) * XP + 64 + fg * 16) = k[kt][1];
        }
        if (it + 1 < iters && key0 + X_NW * 32 < nk) fetch(key0 + X_NW * 32, nx);
        x_wave_sync();
        // ---- lane = query: S^T[key][query] -> P o D, dS; dQ^T += K^T dS^T
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4_t s[2], dp[2], ds[2], pd[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                s[kt] = x_mfma(x_bf(k[kt][1]), x_bf(q[t][1]), x_mfma(x_bf(k[kt][0]), x_bf(q[t][0]), zero));
                dp[kt] = x_mfma(x_bf(v[kt][1]), x_bf(gq[t][1]), x_mfma(x_bf(v[kt][0]), x_bf(gq[t][0]), zero));
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = key0 + kt * 16 + fg * 4 + r;
                    const float p = key < nk ? x_exp2(fmaf(s[kt][r], sc2, -lse2[t])) : 0.f;     // padding queries: lse = +inf -> 0
                    const float mu = drop ? y_drop(hq[t], key, a.drop_p, keep) : 1.0f;
                    pd[kt][r] = p * mu;
                    ds[kt][r] = p * (mu * dp[kt][r] - dl[t]);
                }
            const bf16x8_t bD = x_pack8(ds[0], ds[1]), bP = x_pack8(pd[0], pd[1]);
            const u32x4_t dsv = __builtin_bit_cast(u32x4_t, bD), pdv = __builtin_bit_cast(u32x4_t, bP);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {                         // row t*16 + fr (query), keys kt*16 + fg*4 .. +3
                *reinterpret_cast<u32x2_t*>(sP + (t * 16 + fr) * XPP + (kt * 16 + fg * 4) * 2) = u32x2_t{pdv[kt * 2], pdv[kt * 2 + 1]};
                *reinterpret_cast<u32x2_t*>(sDS + (t * 16 + fr) * XPP + (kt * 16 + fg * 4) * 2) = u32x2_t{dsv[kt * 2], dsv[kt * 2 + 1]};
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dq[t][dt] = x_mfma(x_afrag(sK, dt, fr, fg), bD, dq[t][dt]);
        }
        x_wave_sync();
        // ---- lane = key: dV^T = dO^T (P o D), dK^T = Q^T dS over the 32 queries; one 16-key tile at a time
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const bf16x8_t bV = x_bfrag(sP, kt, fr, fg), bK = x_bfrag(sDS, kt, fr, fg);
            f32x4_t dvt[4], dkt[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dvt[dt] = x_mfma(x_afrag(sG, dt, fr, fg), bV, zero);
                dkt[dt] = x_mfma(x_afrag(sQ, dt, fr, fg), bK, zero);
            }
            x_store_rows(sS, rDV, odv[kt], dvt, 1.0f, fr, fg);
            x_store_rows(sS, rDK, odk[kt], dkt, a.scale, fr, fg);
        }
        x_wave_sync();                                               // the images are rewritten by the next trip
    }
    // ---- dQ^T of the four waves (lane (fr, fg): query t*16 + fr, head dims dt*16 + fg*4 .. +3) -> one partial per workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem + 2 * X_IMG + w * Y_WLDS_B);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4_t*>(red + (t * 16 + fr) * X_RED_PITCH + dt * 16 + fg * 4) = dq[t][dt];
    __syncthreads();
    float* dst = part + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 32 * HD;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int e = threadIdx.x + j * 256, qi = e >> 4, c4 = e & 15;
        f32x4_t sum = zero;
#pragma unroll
        for (int ww = 0; ww < X_NW; ++ww)
            sum += *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(smem + 2 * X_IMG + ww * Y_WLDS_B) + qi * X_RED_PITCH + c4 * 4);
        *reinterpret_cast<f32x4_t*>(dst + qi * HD + c4 * 4) = sum;
    }
}

// dQ = scale * sum of the workgroup partials in workgroup order; two workgroups per problem, one float4 per thread
__global__ __launch_bounds__(256) void attn_fewq_reduce_kernel(const AttnArgs a, int nwg, const float* __restrict__ part) {
    const int prob = blockIdx.x >> 1;
    const int h = prob % a.H, pg = prob / a.H, b = pg / a.G, g = pg % a.G;
    const int e = (blockIdx.x & 1) * 256 + threadIdx.x, qi = e >> 4, c4 = e & 15;
    if (qi >= a.q.n) return;
    const float* src = part + (long long)prob * nwg * 32 * HD + qi * HD + c4 * 4;
    f32x4_t s = {0.f, 0.f, 0.f, 0.f};
    for (int w0 = 0; w0 < nwg; w0 += 8) {
        f32x4_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4_t*>(src + (long long)(w0 + u < nwg ? w0 + u : w0) * 32 * HD);
#pragma unroll
        for (int u = 0; u < 8; ++u) if (w0 + u < nwg) s += v[u];
    }
    const u32x2_t pk = {pack_bf16x2(s[0] * a.scale, s[1] * a.scale), pack_bf16x2(s[2] * a.scale, s[3] * a.scale)};
    const long long row = b * a.q.bs + a.q.base + g * a.q.gs + qi;
    *reinterpret_cast<u32x2_t*>(reinterpret_cast<bf16_t*>(a.dQ) + row * a.lddq + a.dqoff + h * HD + c4 * 4) = pk;
}

}  // namespace egv
using namespace egv;

namespace {
int fewkeys_iters() {
    static const int it = egv_cfg_int("EGV_ATTN_FEWKEYS_ITERS", 4);
    return it < 1 ? 1 : (it > 64 ? 64 : it);
}
bool fewkeys_shape_ok(const AttnArgs& a, int B, bool bwd) {
    static const bool on = egv_cfg_on("EGV_ATTN_FEWKEYS", true);
    auto ok8 = [](int x) { return (x % 8) == 0; };
    if (!on || a.extra || a.drop_p > 0.f || a.nsplit > 1 || a.O32) return false;
    if (a.k.n < 1 || a.k.n > 32 || a.q.n < 128 || a.q.is != 1 || a.k.is != 1) return false;
    if (!(ok8(a.ldq) && ok8(a.ldk) && ok8(a.ldv) && ok8(a.ldo) && ok8(a.qoff) && ok8(a.koff) && ok8(a.voff) && ok8(a.ooff))) return false;
    if (a.mask && a.mask_ld < a.k.n) return false;
    if (bwd) {
        if (!a.dO || !a.lse || !a.dQ || !a.dK || !a.dV || !ok8(a.lddq) || !ok8(a.dqoff)) return false;
        if ((a.lddk % 4) || (a.lddv % 4) || (a.dkoff % 4) || (a.dvoff % 4)) return false;
    } else if (!a.O) return false;
    // 32-bit byte offsets: the row sets must describe matrices of < 2 GB (rows of the last sample / group included)
    const long long qrows = (long long)(B - 1) * a.q.bs + a.q.base + (long long)(a.G - 1) * a.q.gs + a.q.n;
    const long long krows = (long long)(B - 1) * a.k.bs + a.k.base + (long long)(a.G - 1) * a.k.gs + a.k.n;
    const long long ldq_max = a.ldq > a.ldo ? a.ldq : a.ldo, ldk_max = a.ldk > a.ldv ? a.ldk : a.ldv;
    if (qrows * ldq_max * 2 >= (1LL << 31) || krows * ldk_max * 2 >= (1LL << 31)) return false;
    if (bwd && qrows * a.lddq * 2 >= (1LL << 31)) return false;
    return true;
}
inline long long q_rows(const AttnArgs& a, int B) { return (long long)(B - 1) * a.q.bs + a.q.base + (long long)(a.G - 1) * a.q.gs + a.q.n; }
inline long long k_rows(const AttnArgs& a, int B) { return (long long)(B - 1) * a.k.bs + a.k.base + (long long)(a.G - 1) * a.k.gs + a.k.n; }
}  // namespace

// workgroups per (sample, group, head) and the bytes of the backward's partial sums
int egv_attn_fewkeys_nwg(int q_n) {
    const int per = X_NW * fewkeys_iters() * 32;
    return (q_n + per - 1) / per;
}
extern "C" long long egv_attn_fewkeys_workspace_bytes(int B, int G, int H, int q_n) {
    return (long long)B * G * H * egv_attn_fewkeys_nwg(q_n) * 2 * 32 * HD * 4;
}

// 1 if enqueued
int egv_attn_fewkeys_fwd(const AttnArgs& a, int B, hipStream_t st) {
    if (!fewkeys_shape_ok(a, B, false)) return 0;
    const long long qr = q_rows(a, B), kr = k_rows(a, B);
    const int nwg = egv_attn_fewkeys_nwg(a.q.n);
    const long long kvb = kr * (a.ldk > a.ldv ? a.ldk : a.ldv) * 2;
    hipLaunchKernelGGL(attn_fewkeys_fwd_kernel, dim3(nwg, B * a.G * a.H), dim3(256), X_FWD_LDS, st, a, fewkeys_iters(), (unsigned int)(qr * a.ldq * 2),
                       (unsigned int)kvb, (unsigned int)(qr * a.ldo * 2));
    return 1;
}

// 1 if enqueued (dQ, dK, dV; a.ws: >= egv_attn_fewkeys_workspace_bytes, checked by the caller)
int egv_attn_fewkeys_bwd(const AttnArgs& a, int B, hipStream_t st) {
    if (!a.ws || !fewkeys_shape_ok(a, B, true)) return 0;
    const long long qr = q_rows(a, B), kr = k_rows(a, B);
    const int nwg = egv_attn_fewkeys_nwg(a.q.n);
    const long long kvb = kr * (a.ldk > a.ldv ? a.ldk : a.ldv) * 2;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fewkeys_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, X_BWD_LDS);
        attr = true;
    }
    hipLaunchKernelGGL(attn_fewkeys_bwd_kernel, dim3(nwg, B * a.G * a.H), dim3(256), X_BWD_LDS, st, a, fewkeys_iters(), (unsigned int)(qr * a.ldq * 2),
                       (unsigned int)kvb, (unsigned int)(qr * a.ldo * 2), (unsigned int)(qr * a.lddq * 2), (unsigned int)(qr * a.H * 4), a.ws);
    hipLaunchKernelGGL(attn_fewkeys_reduce_kernel, dim3(B * a.G * a.H * 4), dim3(256), 0, st, a, nwg, a.ws);
    return 1;
}

// ---- few queries over many keys (text -> image)
namespace {
int fewq_iters() {
    static const int it = egv_cfg_int("EGV_ATTN_FEWQ_ITERS", 6);
    return it < 1 ? 1 : (it > 64 ? 64 : it);
}
bool fewq_shape_ok(const AttnArgs& a, int B, bool bwd) {
    static const bool on = egv_cfg_on("EGV_ATTN_FEWQ", true);
    auto ok8 = [](int x) { return (x % 8) == 0; };
    if (!on || a.extra || a.mask || a.q.n < 1 || a.q.n > 32 || a.k.n < 512 || a.q.is != 1 || a.k.is != 1) return false;
    if (!(ok8(a.ldq) && ok8(a.ldk) && ok8(a.ldv) && ok8(a.ldo) && ok8(a.qoff) && ok8(a.koff) && ok8(a.voff) && ok8(a.ooff))) return false;
    if (!a.O || !a.ws) return false;
    if (bwd) {
        if (!a.dO || !a.lse || !a.dQ || !a.dK || !a.dV || !ok8(a.lddk) || !ok8(a.lddv) || !ok8(a.dkoff) || !ok8(a.dvoff)) return false;
        if ((a.lddq % 4) || (a.dqoff % 4) || a.lddk != a.lddv) return false;
    }
    const long long qr = q_rows(a, B), kr = k_rows(a, B);
    const long long ldq_max = a.ldq > a.ldo ? a.ldq : a.ldo, ldk_max = a.ldk > a.ldv ? a.ldk : a.ldv;
    if (qr * ldq_max * 2 >= (1LL << 31) || kr * ldk_max * 2 >= (1LL << 31)) return false;
    if (bwd && kr * a.lddk * 2 >= (1LL << 31)) return false;
    return true;
}
}  // namespace
int egv_attn_fewq_nwg(int k_n) {
    const int per = X_NW * fewq_iters() * 32;
    return (k_n + per - 1) / per;
}
// bytes of the partials: forward (m, l, o) states, backward dQ sums (the larger of the two: one buffer serves both)
extern "C" long long egv_attn_fewq_workspace_bytes(int B, int G, int H, int k_n) {
    return (long long)B * G * H * egv_attn_fewq_nwg(k_n) * 32 * X_RED_PITCH * 4;
}
// 1 if enqueued (O, optional O32, lse); a.ws >= egv_attn_fewq_workspace_bytes (checked by the caller)
int egv_attn_fewq_fwd(const AttnArgs& a, int B, hipStream_t st) {
    if (!fewq_shape_ok(a, B, false)) return 0;
    const long long qr = q_rows(a, B), kr = k_rows(a, B);
    const int nwg = egv_attn_fewq_nwg(a.k.n);
    hipLaunchKernelGGL(attn_fewq_fwd_kernel, dim3(nwg, B * a.G * a.H), dim3(256), Y_FWD_LDS, st, a, fewq_iters(), (unsigned int)(qr * a.ldq * 2),
                       (unsigned int)(kr * (a.ldk > a.ldv ? a.ldk : a.ldv) * 2), a.ws);
    hipLaunchKernelGGL(attn_fewq_combine_kernel, dim3(B * a.G * a.H), dim3(256), 0, st, a, nwg, a.ws);
    return 1;
}
// 1 if enqueued (dQ, dK, dV)
int egv_attn_fewq_bwd(const AttnArgs& a, int B, hipStream_t st) {
    if (!fewq_shape_ok(a, B, true)) return 0;
    const long long qr = q_rows(a, B), kr = k_rows(a, B);
    const int nwg = egv_attn_fewq_nwg(a.k.n);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fewq_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, Y_BWD_LDS);
        attr = true;
    }
    hipLaunchKernelGGL(attn_fewq_bwd_kernel, dim3(nwg, B * a.G * a.H), dim3(256), Y_BWD_LDS, st, a, fewq_iters(), (unsigned int)(qr * a.ldq * 2),
                       (unsigned int)(kr * (a.ldk > a.ldv ? a.ldk : a.ldv) * 2), (unsigned int)(qr * a.ldo * 2), (unsigned int)(kr * a.lddk * 2), a.ws);
    hipLaunchKernelGGL(attn_fewq_reduce_kernel, dim3(B * a.G * a.H * 2), dim3(256), 0, st, a, nwg, a.ws);
    return 1;
}
