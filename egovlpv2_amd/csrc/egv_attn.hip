// Grouped softmax attention for the EgoVLPv2 hot path, head_dim = 64.
//
// All attention variants of the path are "a set of query rows attends to a set of key rows" where both
// sets are affine row ranges of the (B*S, d) token matrices, optionally with ONE extra row prepended on
// the other side (the CLS token):
//   divided space attention : queries = patches of frame f,   keys = [CLS ; patches of frame f]
//   divided time attention  : queries = patch n of all frames, keys = [CLS ; patch n of all frames]  (row stride N)
//   CLS query               : query = CLS,                    keys = all S rows
//   image->text / text->image cross attention, RoBERTa self attention: plain row ranges (+ additive key mask)
// so the (b h)(f n) d / (b h n) f d permutes, the r-fold repeat of the CLS key/value and the concatenations of
// video_transformer.py:121-150 are never materialised: kernels gather 128-byte head rows straight from the
// fused qkv buffer.
//
// Backward is split the FlashAttention-1 way into a query-owned kernel (dQ) and a key-owned kernel (dK, dV);
// with the "extra row" symmetric on both sides every output row is written exactly once (no atomics):
//   dQ : groups with extra CLS key, CLS query over all keys
//   dKV: groups with extra CLS query, CLS key over all queries (other side split across workgroups + fp32 reduce)
//
// v1 compute mapping (exact fp32 VALU, shared by the f32 and bf16 storage types): lane = key for q.k^T
// (key row in 64 VGPRs, q broadcast with v_readlane), lane = d for p.V (value column in 64 VGPRs).
#include "egv_attn.h"

namespace egv {

// stage `rows` head rows (64 elements each) of a [.., ld] matrix into an fp32 LDS tile; rows >= nvalid are zero
template <typename T, int NT>
__device__ __forceinline__ void stage_tile(float* s, const T* base, int ld, int off, const AttnArgs& a, const RowSet& rs,
                                           int b, int g, int j0, int ntotal, int tid) {
    for (int c = tid; c < TK * 16; c += NT) {
        const int r = c >> 4, v = c & 15;
        const int j = j0 + r;
        float x[4] = {0.f, 0.f, 0.f, 0.f};
        if (j < ntotal) {
            long long row;
            if (a.extra) row = (j == 0) ? ((long long)b * a.extra_bs + a.extra_row) : rs_row(rs, b, g, j - 1);
            else row = rs_row(rs, b, g, j);
            ld4(base + row * ld + off + v * 4, x);
        }
        *reinterpret_cast<f32x4_t*>(s + r * LDT + v * 4) = f32x4_t{x[0], x[1], x[2], x[3]};
    }
}

__device__ __forceinline__ void load_row64(const float* s, int r, float (&o)[HD]) {
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        f32x4_t t = *reinterpret_cast<const f32x4_t*>(s + r * LDT + v * 4);
        o[v * 4] = t[0]; o[v * 4 + 1] = t[1]; o[v * 4 + 2] = t[2]; o[v * 4 + 3] = t[3];
    }
}

__device__ __forceinline__ float dot_bcast(float v, const float (&row)[HD]) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s = fmaf(readlane_f(v, d), row[d], s);
    return s;
}

// ------------------------------------------------------------------------------------------------
// forward: own = queries
// ------------------------------------------------------------------------------------------------
template <typename T, int NW>
__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(const AttnArgs a) {
    constexpr int NT = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sK = smem;
    float* sV = sK + TK * LDT;
    float* sQ = sV + TK * LDT;                 // [NW][QPW][64]
    float* sO = sQ + NW * QPW * HD;            // [NW][QPW][64]
    float* sM = sO + NW * QPW * HD;            // [NW][QPW][2]

    const int tid = threadIdx.x, lane = tid & 63, w = wave_id();
    const int p = blockIdx.y, b = p / a.G, g = p % a.G, h = blockIdx.z;
    const int split = blockIdx.x % a.nsplit;
    const int q0 = (blockIdx.x / a.nsplit) * (NW * QPW) + w * QPW;
    const int nq = min(QPW, a.q.n - q0);       // may be <= 0 for trailing waves
    const T* Q = reinterpret_cast<const T*>(a.Q);
    const T* K = reinterpret_cast<const T*>(a.K);
    const T* V = reinterpret_cast<const T*>(a.V);
    const int hq = a.qoff + h * HD, hk = a.koff + h * HD, hv = a.voff + h * HD, ho = a.ooff + h * HD;

    for (int i = 0; i < nq; ++i) {
        const long long row = rs_row(a.q, b, g, q0 + i);
        sQ[(w * QPW + i) * HD + lane] = Elem<T>::ld(Q + row * a.ldq + hq + lane) * a.scale;
        sO[(w * QPW + i) * HD + lane] = 0.f;
        if (lane == 0) {
            sM[(w * QPW + i) * 2] = -INFINITY;
            sM[(w * QPW + i) * 2 + 1] = 0.f;
        }
    }
    const int ntot = a.k.n + a.extra;
    const int per = (((ntot + TK - 1) / TK + a.nsplit - 1) / a.nsplit) * TK;      // key range of this split (tile aligned)
    const int jbeg = split * per, jend = min(ntot, jbeg + per);
    for (int j0 = jbeg; j0 < jend; j0 += TK) {
        __syncthreads();
        stage_tile<T, NT>(sK, K, a.ldk, hk, a, a.k, b, g, j0, jend, tid);
        stage_tile<T, NT>(sV, V, a.ldv, hv, a, a.k, b, g, j0, jend, tid);
        __syncthreads();
        if (nq > 0) {
            float krow[HD], vcol[HD];
            load_row64(sK, lane, krow);
#pragma unroll
            for (int kk = 0; kk < TK; ++kk) vcol[kk] = sV[kk * LDT + lane];
            const int j = j0 + lane;
            const bool valid = j < jend;
            float mk = 0.f;
            if (a.mask && valid && !(a.extra && j == 0)) mk = a.mask[(long long)b * a.mask_ld + (j - a.extra)];
            for (int i = 0; i < nq; ++i) {
                const float qv = sQ[(w * QPW + i) * HD + lane];
                float s = dot_bcast(qv, krow) + mk;
                s = valid ? s : -INFINITY;
                const float m_old = sM[(w * QPW + i) * 2], l_old = sM[(w * QPW + i) * 2 + 1];
                const float m_new = fmaxf(m_old, wave_max(s));
                const float pj = valid ? __expf(s - m_new) : 0.f;
                const float corr = __expf(m_old - m_new);
                const float l_new = l_old * corr + wave_sum(pj);
                const float pd = a.drop_p > 0.f ? pj * drop_mult(a, (long long)p * a.q.n + q0 + i, j - a.extra, h) : pj;
                float o = sO[(w * QPW + i) * HD + lane] * corr;
#pragma unroll
                for (int kk = 0; kk < TK; ++kk) o = fmaf(readlane_f(pd, kk), vcol[kk], o);
                sO[(w * QPW + i) * HD + lane] = o;
                if (lane == 0) {
                    sM[(w * QPW + i) * 2] = m_new;
                    sM[(w * QPW + i) * 2 + 1] = l_new;
                }
            }
        }
    }
    T* O = reinterpret_cast<T*>(a.O);
    for (int i = 0; i < nq; ++i) {
        const long long row = rs_row(a.q, b, g, q0 + i);
        const float m = sM[(w * QPW + i) * 2], l = sM[(w * QPW + i) * 2 + 1];
        if (a.nsplit == 1) {
            Elem<T>::st(O + row * a.ldo + ho + lane, sO[(w * QPW + i) * HD + lane] / l);
            if (a.lse && lane == 0) a.lse[row * a.H + h] = m + __logf(l);
        } else {
            // partial (m, l, o[64]) per split: ws[split][P * q.n own rows][H][66]
            const long long nrows = (long long)gridDim.y * a.q.n;
            const long long orow = (long long)p * a.q.n + q0 + i;
            float* dst = a.ws + (((long long)split * nrows + orow) * a.H + h) * 66;
            dst[2 + lane] = sO[(w * QPW + i) * HD + lane];
            if (lane == 0) { dst[0] = m; dst[1] = l; }
        }
    }
}

// combine the per-split partial softmax states of attn_fwd_kernel (one wave per (own row, head))
template <typename T>
__global__ void attn_fwd_combine_kernel(const AttnArgs a, int P) {
    const int lane = threadIdx.x & 63;
    const long long nrows = (long long)P * a.q.n;
    const long long idx = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (idx >= nrows * a.H) return;
    const long long orow = idx / a.H;
    const int h = (int)(idx % a.H);
    const int p = (int)(orow / a.q.n), i = (int)(orow % a.q.n);
    const int b = p / a.G, g = p % a.G;
    float M = -INFINITY;
    for (int s = 0; s < a.nsplit; ++s) M = fmaxf(M, a.ws[(((long long)s * nrows + orow) * a.H + h) * 66]);
    float Lsum = 0.f, o = 0.f;
    for (int s = 0; s < a.nsplit; ++s) {
        const float* src = a.ws + (((long long)s * nrows + orow) * a.H + h) * 66;
        const float c = __expf(src[0] - M);
        Lsum += src[1] * c;
        o += src[2 + lane] * c;
    }
    const long long row = rs_row(a.q, b, g, i);
    Elem<T>::st(reinterpret_cast<T*>(a.O) + row * a.ldo + a.ooff + h * HD + lane, o / Lsum);
    if (a.O32) a.O32[row * a.ldo + a.ooff + h * HD + lane] = o / Lsum;
    if (a.lse && lane == 0) a.lse[row * a.H + h] = M + __logf(Lsum);
}

// ------------------------------------------------------------------------------------------------
// backward, query-owned: dQ (+ delta = rowsum(dO * O))
// ------------------------------------------------------------------------------------------------
template <typename T, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_dq_kernel(const AttnArgs a) {
    constexpr int NT = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sK = smem;
    float* sV = sK + TK * LDT;
    float* sQ = sV + TK * LDT;                 // [NW][QPW][64] scaled q
    float* sDO = sQ + NW * QPW * HD;           // [NW][QPW][64]
    float* sDQ = sDO + NW * QPW * HD;          // [NW][QPW][64]
    float* sS = sDQ + NW * QPW * HD;           // [NW][QPW][2]: lse, delta

    const int tid = threadIdx.x, lane = tid & 63, w = wave_id();
    const int p = blockIdx.y, b = p / a.G, g = p % a.G, h = blockIdx.z;
    const int split = blockIdx.x % a.nsplit;
    const int q0 = (blockIdx.x / a.nsplit) * (NW * QPW) + w * QPW;
    const int nq = min(QPW, a.q.n - q0);
    const T* Q = reinterpret_cast<const T*>(a.Q);
    const T* K = reinterpret_cast<const T*>(a.K);
    const T* V = reinterpret_cast<const T*>(a.V);
    const T* O = reinterpret_cast<const T*>(a.O);
    const T* dO = reinterpret_cast<const T*>(a.dO);
    const int hq = a.qoff + h * HD, hk = a.koff + h * HD, hv = a.voff + h * HD, ho = a.ooff + h * HD;

    for (int i = 0; i < nq; ++i) {
        const long long row = rs_row(a.q, b, g, q0 + i);
        const float dov = Elem<T>::ld(dO + row * a.ldo + ho + lane);
        const float ov = Elem<T>::ld(O + row * a.ldo + ho + lane);
        sQ[(w * QPW + i) * HD + lane] = Elem<T>::ld(Q + row * a.ldq + hq + lane) * a.scale;
        sDO[(w * QPW + i) * HD + lane] = dov;
        sDQ[(w * QPW + i) * HD + lane] = 0.f;
        const float dl = wave_sum(dov * ov);
        if (lane == 0) {
            sS[(w * QPW + i) * 2] = a.lse[row * a.H + h];
            sS[(w * QPW + i) * 2 + 1] = dl;
            if (split == 0) a.delta[row * a.H + h] = dl;
        }
    }
    const int ntot = a.k.n + a.extra;
    const int per = (((ntot + TK - 1) / TK + a.nsplit - 1) / a.nsplit) * TK;
    const int jbeg = split * per, jend = min(ntot, jbeg + per);
    for (int j0 = jbeg; j0 < jend; j0 += TK) {
        __syncthreads();
        stage_tile<T, NT>(sK, K, a.ldk, hk, a, a.k, b, g, j0, jend, tid);
        stage_tile<T, NT>(sV, V, a.ldv, hv, a, a.k, b, g, j0, jend, tid);
        __syncthreads();
        if (nq > 0) {
            float krow[HD], vrow[HD];
            load_row64(sK, lane, krow);
            load_row64(sV, lane, vrow);
            const int j = j0 + lane;
            const bool valid = j < jend;
            float mk = 0.f;
            if (a.mask && valid && !(a.extra && j == 0)) mk = a.mask[(long long)b * a.mask_ld + (j - a.extra)];
            for (int i = 0; i < nq; ++i) {
                const float qv = sQ[(w * QPW + i) * HD + lane];
                const float dov = sDO[(w * QPW + i) * HD + lane];
                const float lse = sS[(w * QPW + i) * 2], dl = sS[(w * QPW + i) * 2 + 1];
                const float s = dot_bcast(qv, krow) + mk;
                const float pj = valid ? __expf(s - lse) : 0.f;
                float dp = dot_bcast(dov, vrow);
                if (a.drop_p > 0.f) dp *= drop_mult(a, (long long)p * a.q.n + q0 + i, j - a.extra, h);
                const float ds = pj * (dp - dl);
                float acc = sDQ[(w * QPW + i) * HD + lane];
#pragma unroll
                for (int kk = 0; kk < TK; ++kk) acc = fmaf(readlane_f(ds, kk), sK[kk * LDT + lane], acc);
                sDQ[(w * QPW + i) * HD + lane] = acc;
            }
        }
    }
    T* dQ = reinterpret_cast<T*>(a.dQ);
    const int hdq = a.dqoff + h * HD;
    for (int i = 0; i < nq; ++i) {
        const long long row = rs_row(a.q, b, g, q0 + i);
        if (a.nsplit == 1) {
            Elem<T>::st(dQ + row * a.lddq + hdq + lane, sDQ[(w * QPW + i) * HD + lane] * a.scale);
        } else {
            const long long nrows = (long long)gridDim.y * a.q.n;
            const long long orow = (long long)p * a.q.n + q0 + i;
            a.ws[(((long long)split * nrows + orow) * a.H + h) * HD + lane] = sDQ[(w * QPW + i) * HD + lane] * a.scale;
        }
    }
}

template <typename T>
__global__ void attn_dq_combine_kernel(const AttnArgs a, int P) {
    const int lane = threadIdx.x & 63;
    const long long nrows = (long long)P * a.q.n;
    const long long idx = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (idx >= nrows * a.H) return;
    const long long orow = idx / a.H;
    const int h = (int)(idx % a.H);
    const int p = (int)(orow / a.q.n), i = (int)(orow % a.q.n);
    const int b = p / a.G, g = p % a.G;
    float acc = 0.f;
    for (int s = 0; s < a.nsplit; ++s) acc += a.ws[(((long long)s * nrows + orow) * a.H + h) * HD + lane];
    const long long row = rs_row(a.q, b, g, i);
    Elem<T>::st(reinterpret_cast<T*>(a.dQ) + row * a.lddq + a.dqoff + h * HD + lane, acc);
}

// ------------------------------------------------------------------------------------------------
// backward, key-owned: dK, dV.  own = keys (one tile of <= 64 keys per workgroup, lane = key),
// other = queries (optional extra CLS query first), distributed over the waves (and over blockIdx.x % nsplit).
// ------------------------------------------------------------------------------------------------
template <typename T, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_dkv_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sK = smem;                          // own tile [64][LDT]
    float* sV = sK + TK * LDT;
    float* sR = smem;                          // reduction buffer [64][129], aliases sK/sV after the main loop
    constexpr int LDR = 129;
    static_assert(2 * TK * LDT >= TK * 129, "reduction buffer must fit in the K/V tiles");

    const int tid = threadIdx.x, lane = tid & 63, w = wave_id();
    const int p = blockIdx.y, b = p / a.G, g = p % a.G, h = blockIdx.z;
    const int tile = blockIdx.x / a.nsplit, split = blockIdx.x % a.nsplit;
    const int k0 = tile * TK;
    const int nk = min(TK, a.k.n - k0);
    const T* Q = reinterpret_cast<const T*>(a.Q);
    const T* K = reinterpret_cast<const T*>(a.K);
    const T* V = reinterpret_cast<const T*>(a.V);
    const T* dO = reinterpret_cast<const T*>(a.dO);
    const int hq = a.qoff + h * HD, hk = a.koff + h * HD, hv = a.voff + h * HD, ho = a.ooff + h * HD;

    // stage own K/V tile (no extra on the own side)
    for (int c = tid; c < TK * 16; c += 64 * NW) {
        const int r = c >> 4, v = c & 15;
        float x[4] = {0.f, 0.f, 0.f, 0.f}, y[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < nk) {
            const long long row = rs_row(a.k, b, g, k0 + r);
            ld4(K + row * a.ldk + hk + v * 4, x);
            ld4(V + row * a.ldv + hv + v * 4, y);
        }
        *reinterpret_cast<f32x4_t*>(sK + r * LDT + v * 4) = f32x4_t{x[0], x[1], x[2], x[3]};
        *reinterpret_cast<f32x4_t*>(sV + r * LDT + v * 4) = f32x4_t{y[0], y[1], y[2], y[3]};
    }
    __syncthreads();

    float dk[HD], dv[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) dk[d] = dv[d] = 0.f;
    const bool valid = lane < nk;
    float mk = 0.f;
    if (a.mask && valid) mk = a.mask[(long long)b * a.mask_ld + k0 + lane];

    const int ntot = a.q.n + a.extra;
    const int per = (ntot + a.nsplit - 1) / a.nsplit;
    const int i0 = split * per, i1 = min(ntot, i0 + per);
    for (int i = i0 + w; i < i1; i += NW) {
        long long row;
        if (a.extra) row = (i == 0) ? ((long long)b * a.extra_bs + a.extra_row) : rs_row(a.q, b, g, i - 1);
        else row = rs_row(a.q, b, g, i);
        const float qv = Elem<T>::ld(Q + row * a.ldq + hq + lane) * a.scale;
        const float dov = Elem<T>::ld(dO + row * a.ldo + ho + lane);
        const float lse = a.lse[row * a.H + h], dl = a.delta[row * a.H + h];
        float s = mk, dp = 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const f32x4_t kx = *reinterpret_cast<const f32x4_t*>(sK + lane * LDT + v * 4);
            const f32x4_t vx = *reinterpret_cast<const f32x4_t*>(sV + lane * LDT + v * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s = fmaf(readlane_f(qv, v * 4 + e), kx[e], s);
                dp = fmaf(readlane_f(dov, v * 4 + e), vx[e], dp);
            }
        }
        const float pj0 = valid ? __expf(s - lse) : 0.f;
        const float mu = a.drop_p > 0.f ? drop_mult(a, (long long)p * a.q.n + i - a.extra, k0 + lane, h) : 1.0f;
        const float pj = pj0 * mu;
        const float ds = pj0 * (dp * mu - dl);
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            dv[d] = fmaf(pj, readlane_f(dov, d), dv[d]);
            dk[d] = fmaf(ds, readlane_f(qv, d), dk[d]);     // qv carries the softmax scale
        }
    }
    // cross-wave reduction through LDS (lane = key writes column d: pitch 129 -> conflict free)
    __syncthreads();                           // every wave is done reading sK/sV
    for (int ww = 0; ww < NW; ++ww) {
        if (w == ww) {
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                if (ww == 0) {
                    sR[lane * LDR + d] = dk[d];
                    sR[lane * LDR + HD + d] = dv[d];
                } else {
                    sR[lane * LDR + d] += dk[d];
                    sR[lane * LDR + HD + d] += dv[d];
                }
            }
        }
        __syncthreads();
    }
    if (a.nsplit == 1) {
        T* dK = reinterpret_cast<T*>(a.dK);
        T* dV = reinterpret_cast<T*>(a.dV);
        const int hdk = a.dkoff + h * HD, hdv = a.dvoff + h * HD;
        for (int r = w; r < nk; r += NW) {
            const long long row = rs_row(a.k, b, g, k0 + r);
            Elem<T>::st(dK + row * a.lddk + hdk + lane, sR[r * LDR + lane]);
            Elem<T>::st(dV + row * a.lddv + hdv + lane, sR[r * LDR + HD + lane]);
        }
    } else {
        // ws layout: [split][P * k.n own rows][H][2][64] fp32
        const long long nrows = (long long)gridDim.y * a.k.n;
        for (int r = w; r < nk; r += NW) {
            const long long orow = (long long)p * a.k.n + k0 + r;
            float* dst = a.ws + (((long long)split * nrows + orow) * a.H + h) * 2 * HD;
            dst[lane] = sR[r * LDR + lane];
            dst[HD + lane] = sR[r * LDR + HD + lane];
        }
    }
}

template <typename T>
__global__ void attn_dkv_reduce_kernel(const AttnArgs a, int P) {
    // one wave per (own row, head)
    const int lane = threadIdx.x & 63;
    const long long nrows = (long long)P * a.k.n;
    const long long idx = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (idx >= nrows * a.H) return;
    const long long orow = idx / a.H;
    const int h = (int)(idx % a.H);
    const int p = (int)(orow / a.k.n), i = (int)(orow % a.k.n);
    const int b = p / a.G, g = p % a.G;
    float sk = 0.f, sv = 0.f;
    for (int s = 0; s < a.nsplit; ++s) {
        const float* src = a.ws + (((long long)s * nrows + orow) * a.H + h) * 2 * HD;
        sk += src[lane];
        sv += src[HD + lane];
    }
    const long long row = rs_row(a.k, b, g, i);
    T* dK = reinterpret_cast<T*>(a.dK);
    T* dV = reinterpret_cast<T*>(a.dV);
    Elem<T>::st(dK + row * a.lddk + a.dkoff + h * HD + lane, sk);
    Elem<T>::st(dV + row * a.lddv + a.dvoff + h * HD + lane, sv);
}

// ------------------------------------------------------------------------------------------------
// "one row against many" kernels (CLS query over all S keys; CLS key under all S queries).  lane = other-side row:
// every lane keeps a private online-softmax / gradient accumulator over its rows and the 64 lanes are merged once per
// wave; partial states go to the same fp32 workspaces as the split launches above (combined by the same kernels).
// grid (nsplit, problems, heads), one wave per workgroup; each split covers a contiguous range of other rows.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void load_row_f(const T* p, float (&o)[HD]) {
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        float x[4];
        ld4(p + v * 4, x);
        o[v * 4] = x[0]; o[v * 4 + 1] = x[1]; o[v * 4 + 2] = x[2]; o[v * 4 + 3] = x[3];
    }
}

__device__ __forceinline__ long long other_row_v(const AttnArgs& a, const RowSet& rs, int b, int g, int j) {
    if (a.extra) return (j == 0) ? ((long long)b * a.extra_bs + a.extra_row) : rs_row(rs, b, g, j - 1);
    return rs_row(rs, b, g, j);
}

// 8 lanes share one other-side row (16 B = 8 head dims per lane -> 128-byte coalesced row reads); a wave walks 8 rows at a
// time, a workgroup (4 waves) 32 rows.  Dot products are reduced over the 8-lane group with 3 xor-shuffles; every lane
// accumulates its own 8 output dims; the 8 groups of a wave are merged with 3 more shuffle steps at the end.
template <typename T>
__device__ __forceinline__ void ld8(const T* p, float (&o)[8]) {
    float a[4], b[4];
    ld4(p, a);
    ld4(p + 4, b);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}
__device__ __forceinline__ float grp8_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    return v + __shfl_xor(v, 4, 64);
}
__device__ __forceinline__ float xgrp_sum(float v) {           // across the 8 row groups of a wave (same sub-lane)
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ float xgrp_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 8, 64));
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}

template <typename T>
__global__ __launch_bounds__(256) void attn1_fwd_kernel(const AttnArgs a) {
    __shared__ float red[4][66];
    const int lane = threadIdx.x & 63, w = wave_id(), sub = lane & 7, grp = lane >> 3;
    const int split = blockIdx.x, p = blockIdx.y, b = p / a.G, g = p % a.G, h = blockIdx.z;
    const T* Q = reinterpret_cast<const T*>(a.Q);
    const T* K = reinterpret_cast<const T*>(a.K);
    const T* V = reinterpret_cast<const T*>(a.V);
    const long long qrow = rs_row(a.q, b, g, 0);
    float q[8];
    ld8(Q + qrow * a.ldq + a.qoff + h * HD + sub * 8, q);
#pragma unroll
    for (int d = 0; d < 8; ++d) q[d] *= a.scale;
    const int ntot = a.k.n + a.extra;
    const int per = (ntot + a.nsplit - 1) / a.nsplit;
    const int j0 = split * per, j1 = min(ntot, j0 + per);
    float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) o[d] = 0.f;
    for (int j = j0 + w * 8 + grp; j < j1; j += 32) {
        const long long row = other_row_v(a, a.k, b, g, j);
        float kr[8], vr[8];
        ld8(K + row * a.ldk + a.koff + h * HD + sub * 8, kr);
        ld8(V + row * a.ldv + a.voff + h * HD + sub * 8, vr);
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) s = fmaf(q[d], kr[d], s);
        s = grp8_sum(s);
        if (a.mask && !(a.extra && j == 0)) s += a.mask[(long long)b * a.mask_ld + (j - a.extra)];
        const float mn = fmaxf(m, s);
        const float c = __expf(m - mn), pj = __expf(s - mn);
        l = l * c + pj;
#pragma unroll
        for (int d = 0; d < 8; ++d) o[d] = fmaf(o[d], c, pj * vr[d]);
        m = mn;
    }
    // merge the 8 groups of the wave, then the 4 waves through LDS
    const float M = xgrp_max(m);
    const float c = (m == -INFINITY) ? 0.f : __expf(m - M);
    const float Lw = xgrp_sum(l * c);
#pragma unroll
    for (int d = 0; d < 8; ++d) o[d] = xgrp_sum(o[d] * c);
    if (grp == 0) {
#pragma unroll
        for (int d = 0; d < 8; ++d) red[w][2 + sub * 8 + d] = o[d];
        if (sub == 0) { red[w][0] = M; red[w][1] = Lw; }
    }
    __syncthreads();
    if (w == 0) {
        const float MM = fmaxf(fmaxf(red[0][0], red[1][0]), fmaxf(red[2][0], red[3][0]));
        float L = 0.f, od = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            const float cw = (red[ww][0] == -INFINITY) ? 0.f : __expf(red[ww][0] - MM);
            L += red[ww][1] * cw;
            od += red[ww][2 + lane] * cw;
        }
        const long long nrows = (long long)gridDim.y * a.q.n;
        const long long orow = (long long)p * a.q.n;
        float* dst = a.ws + (((long long)split * nrows + orow) * a.H + h) * 66;
        dst[2 + lane] = od;
        if (lane == 0) { dst[0] = MM; dst[1] = L; }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn1_dq_kernel(const AttnArgs a) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = wave_id(), sub = lane & 7, grp = lane >> 3;
    const int split = blockIdx.x, p = blockIdx.y, b = p / a.G, g = p % a.G, h = blockIdx.z;
    const T* Q = reinterpret_cast<const T*>(a.Q);
    const T* K = reinterpret_cast<const T*>(a.K);
    const T* V = reinterpret_cast<const T*>(a.V);
    const T* O = reinterpret_cast<const T*>(a.O);
    const T* dO = reinterpret_cast<const T*>(a.dO);
    const long long qrow = rs_row(a.q, b, g, 0);
    float q[8], go[8], ov[8];
    ld8(Q + qrow * a.ldq + a.qoff + h * HD + sub * 8, q);
    ld8(dO + qrow * a.ldo + a.ooff + h * HD + sub * 8, go);
    if (a.O32) {                                                   // delta from the fp32 values of O (egv_attn_desc::O32)
        const float* o32 = a.O32 + qrow * a.ldo + a.ooff + h * HD + sub * 8;
#pragma unroll
        for (int d = 0; d < 8; ++d) ov[d] = o32[d];
    } else {
        ld8(O + qrow * a.ldo + a.ooff + h * HD + sub * 8, ov);
    }
    float dl = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) { dl = fmaf(go[d], ov[d], dl); q[d] *= a.scale; }
    dl = grp8_sum(dl);
    const float lse = a.lse[qrow * a.H + h];
    if (split == 0 && threadIdx.x == 0) a.delta[qrow * a.H + h] = dl;
    const int ntot = a.k.n + a.extra;
    const int per = (ntot + a.nsplit - 1) / a.nsplit;
    const int j0 = split * per, j1 = min(ntot, j0 + per);
    float acc[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[d] = 0.f;
    for (int j = j0 + w * 8 + grp; j < j1; j += 32) {
        const long long row = other_row_v(a, a.k, b, g, j);
        float kr[8], vr[8];
        ld8(K + row * a.ldk + a.koff + h * HD + sub * 8, kr);
        ld8(V + row * a.ldv + a.voff + h * HD + sub * 8, vr);
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) { s = fmaf(q[d], kr[d], s); dp = fmaf(go[d], vr[d], dp); }
        s = grp8_sum(s);
        dp = grp8_sum(dp);
        if (a.mask && !(a.extra && j == 0)) s += a.mask[(long long)b * a.mask_ld + (j - a.extra)];
        const float ds = __expf(s - lse) * (dp - dl);
#pragma unroll
        for (int d = 0; d < 8; ++d) acc[d] = fmaf(ds, kr[d], acc[d]);
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[d] = xgrp_sum(acc[d]);
    if (grp == 0) {
#pragma unroll
        for (int d = 0; d < 8; ++d) red[w][sub * 8 + d] = acc[d];
    }
    __syncthreads();
    if (w == 0) {
        const long long nrows = (long long)gridDim.y * a.q.n;
        const long long orow = (long long)p * a.q.n;
        a.ws[(((long long)split * nrows + orow) * a.H + h) * HD + lane] = (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) * a.scale;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn1_dkv_kernel(const AttnArgs a) {
    __shared__ float red[4][128];
    const int lane = threadIdx.x & 63, w = wave_id(), sub = lane & 7, grp = lane >> 3;
    const int split = blockIdx.x, p = blockIdx.y, b = p / a.G, g = p % a.G, h = blockIdx.z;
    const T* Q = reinterpret_cast<const T*>(a.Q);
    const T* K = reinterpret_cast<const T*>(a.K);
    const T* V = reinterpret_cast<const T*>(a.V);
    const T* dO = reinterpret_cast<const T*>(a.dO);
    const long long krow = rs_row(a.k, b, g, 0);
    float kc[8], vc[8];
    ld8(K + krow * a.ldk + a.koff + h * HD + sub * 8, kc);
    ld8(V + krow * a.ldv + a.voff + h * HD + sub * 8, vc);
    const float mk = a.mask ? a.mask[(long long)b * a.mask_ld] : 0.f;
    const int ntot = a.q.n + a.extra;
    const int per = (ntot + a.nsplit - 1) / a.nsplit;
    const int i0 = split * per, i1 = min(ntot, i0 + per);
    float dk[8], dv[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) dk[d] = dv[d] = 0.f;
    for (int i = i0 + w * 8 + grp; i < i1; i += 32) {
        const long long row = other_row_v(a, a.q, b, g, i);
        float qr[8], gr[8];
        ld8(Q + row * a.ldq + a.qoff + h * HD + sub * 8, qr);
        ld8(dO + row * a.ldo + a.ooff + h * HD + sub * 8, gr);
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) { s = fmaf(qr[d], kc[d], s); dp = fmaf(gr[d], vc[d], dp); }
        s = grp8_sum(s);
        dp = grp8_sum(dp);
        const float pj = __expf(s * a.scale + mk - a.lse[row * a.H + h]);
        const float ds = pj * (dp - a.delta[row * a.H + h]) * a.scale;
#pragma unroll
        for (int d = 0; d < 8; ++d) { dv[d] = fmaf(pj, gr[d], dv[d]); dk[d] = fmaf(ds, qr[d], dk[d]); }
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) { dk[d] = xgrp_sum(dk[d]); dv[d] = xgrp_sum(dv[d]); }
    if (grp == 0) {
#pragma unroll
        for (int d = 0; d < 8; ++d) { red[w][sub * 8 + d] = dk[d]; red[w][64 + sub * 8 + d] = dv[d]; }
    }
    __syncthreads();
    if (w < 2) {
        const long long nrows = (long long)gridDim.y * a.k.n;
        const long long orow = (long long)p * a.k.n;
        float* dst = a.ws + (((long long)split * nrows + orow) * a.H + h) * 2 * HD;
        const int c = w * 64 + lane;
        dst[c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    }
}

static inline size_t fwd_smem(int nw) { return (size_t)(2 * TK * LDT + 2 * nw * QPW * HD + nw * QPW * 2) * 4; }
static inline size_t dq_smem(int nw) { return (size_t)(2 * TK * LDT + 3 * nw * QPW * HD + nw * QPW * 2) * 4; }
static inline size_t dkv_smem() { return (size_t)(2 * TK * LDT) * 4; }

}  // namespace egv
using namespace egv;

static int check_desc(const egv_attn_desc* d, const char* who) {
    EGV_CHECK(d->B > 0 && d->G > 0 && d->H > 0 && d->q_n > 0 && d->k_n > 0, "%s: bad problem shape", who);
    EGV_CHECK(!(d->drop_p > 0.f) || (d->extra == 0 && d->drop_p < 1.f && d->q_n > 1 && d->k_n > 1),
              "%s: attention dropout is implemented for plain row sets (no extra CLS row, no single-row problems)", who);
    EGV_CHECK((d->ldq % 4 == 0) && (d->ldk % 4 == 0) && (d->ldv % 4 == 0) && (d->qoff % 4 == 0) && (d->koff % 4 == 0) &&
                  (d->voff % 4 == 0), "%s: leading dims / head offsets must be multiples of 4 elements", who);
    return 0;
}

#define EGV_ATTN_LAUNCH(KERNEL, NWV, SMEM, GRID)                                                                 \
    do {                                                                                                         \
        if (dtype == EGV_BF16) {                                                                                 \
            if (NWV == 1) hipLaunchKernelGGL((KERNEL<bf16_t, 1>), GRID, dim3(64), SMEM, st, a);                    \
            else if (NWV == 2) hipLaunchKernelGGL((KERNEL<bf16_t, 2>), GRID, dim3(128), SMEM, st, a);              \
            else hipLaunchKernelGGL((KERNEL<bf16_t, 4>), GRID, dim3(256), SMEM, st, a);                            \
        } else {                                                                                                 \
            if (NWV == 1) hipLaunchKernelGGL((KERNEL<float, 1>), GRID, dim3(64), SMEM, st, a);                     \
            else if (NWV == 2) hipLaunchKernelGGL((KERNEL<float, 2>), GRID, dim3(128), SMEM, st, a);               \
            else hipLaunchKernelGGL((KERNEL<float, 4>), GRID, dim3(256), SMEM, st, a);                             \
        }                                                                                                        \
    } while (0)

static inline int pick_nw(int n_own) {
    const int t = (n_own + QPW - 1) / QPW;
    return t >= 4 ? 4 : (t >= 2 ? 2 : 1);
}

// fp32 workspace for a split launch: which = 0 fwd (own = queries), 1 dq (own = queries), 2 dkv (own = keys)
extern "C" long long egv_attn_split_workspace_bytes(int which, int B, int G, int H, int n_own, int nsplit) {
    if (nsplit <= 1) return 0;
    const long long per = which == 0 ? 66 : (which == 1 ? HD : 2 * HD);
    return (long long)nsplit * B * G * n_own * H * per * 4;
}

// egv_attn_fwd with nsplit <= 1, an extra row and a workspace of this size also computes the extra row AS A QUERY over the
// union of the groups' keys (the CLS query of the divided attention) when egv_attn_fwd_covers_extra says so
extern "C" long long egv_attn_fwd_extra_workspace_bytes(int B, int G, int H) { return (long long)B * G * H * 66 * 4; }
extern "C" int egv_attn_fwd_covers_extra(int dtype, const egv_attn_desc* d) {
    if (dtype != EGV_BF16 || !d || !d->ws) return 0;
    // a workspace too small for the partial states covers nothing: egv_attn_fwd then computes the group rows only and the caller
    // must issue the one-query launch itself (the extra row of O and lse would otherwise stay unwritten)
    if (d->ws_bytes < egv_attn_fwd_extra_workspace_bytes(d->B, d->G, d->H)) return 0;
    AttnArgs a = to_args(d);
    return (egv_attn_fwd_cls_ok(a) || egv_attn_time_fwd_ok(a, d->B) || egv_attn_space_fwd_ok(a, d->B)) ? 1 : 0;
}

static int egv_attn_fwd_impl(int dtype, const egv_attn_desc* d, void* stream) {
    if (check_desc(d, "egv_attn_fwd")) return -1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    AttnArgs a = to_args(d);
    if (dtype == EGV_BF16 && a.nsplit == 1) {
        if (egv_attn_fewkeys_fwd(a, d->B, st)) {    // many queries over <= 32 keys (image -> text cross attention): egv_attn_cross.hip
            EGV_LAUNCH_CHECK();
            return 0;
        }
        if (d->ws && d->ws_bytes >= egv_attn_fwd_extra_workspace_bytes(d->B, d->G, d->H) && egv_attn_time_fwd(a, d->B, st)) {
            EGV_LAUNCH_CHECK();                  // <= 16-row groups (time attention): group rows and the CLS query in one launch + its combination
            return 0;
        }
        {   // long groups (space attention): row-major LDS images, the CLS row as row n of them (egv_attn_space.hip)
            AttnArgs a2 = a;
            if (!(d->ws && d->ws_bytes >= egv_attn_fwd_extra_workspace_bytes(d->B, d->G, d->H))) a2.ws = nullptr;
            if (egv_attn_space_fwd(a2, d->B, st)) {
                EGV_LAUNCH_CHECK();
                return 0;
            }
        }
        if (d->ws && egv_attn_fwd_cls_ok(a))
            EGV_CHECK(d->ws_bytes >= egv_attn_fwd_extra_workspace_bytes(d->B, d->G, d->H), "egv_attn_fwd: workspace too small for the extra row's partial states");
        const int r = egv_attn_fwd_mfma(a, d->B, st);
        if (r) {
            EGV_LAUNCH_CHECK();
            if (r == 2) {                        // the extra row as a query: combine its G partial states (one per group)
                AttnArgs c = a;
                c.q = RowSet{a.extra_bs, a.extra_row, 0, 1, 1};
                c.nsplit = a.G;
                c.G = 1;
                const long long n = (long long)d->B * d->H;
                hipLaunchKernelGGL(attn_fwd_combine_kernel<bf16_t>, dim3((int)((n + 3) / 4)), dim3(256), 0, st, c, d->B);
                EGV_LAUNCH_CHECK();
            }
            return 0;
        }
    }
    if (dtype == EGV_BF16 && d->ws && d->q_n <= 32 && d->ws_bytes >= egv_attn_fewq_workspace_bytes(d->B, d->G, d->H, d->k_n) &&
        egv_attn_fewq_fwd(a, d->B, st)) {              // <= 32 queries over many keys (text -> image cross attention): egv_attn_cross.hip
        EGV_LAUNCH_CHECK();
        return 0;
    }
    if (a.nsplit > 1)
        EGV_CHECK(d->ws && d->ws_bytes >= egv_attn_split_workspace_bytes(0, d->B, d->G, d->H, d->q_n, a.nsplit),
                  "egv_attn_fwd: workspace too small");
    if (d->q_n == 1 && a.nsplit > 1) {
        dim3 grid1(a.nsplit, d->B * d->G, d->H);
        if (dtype == EGV_BF16) hipLaunchKernelGGL(attn1_fwd_kernel<bf16_t>, grid1, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(attn1_fwd_kernel<float>, grid1, dim3(256), 0, st, a);
    } else if (dtype == EGV_BF16 && a.nsplit > 1 && egv_attn_fwd_mfma(a, d->B, st)) {
        // short query side against a long key side (text -> image): MFMA kernel per key chunk, partial softmax states
    } else {
        const int nw = pick_nw(d->q_n);
        dim3 grid(((d->q_n + nw * QPW - 1) / (nw * QPW)) * a.nsplit, d->B * d->G, d->H);
        const size_t sm = fwd_smem(nw);
        EGV_ATTN_LAUNCH(attn_fwd_kernel, nw, sm, grid);
    }
    EGV_LAUNCH_CHECK();
    if (a.nsplit > 1) {
        const long long n = (long long)d->B * d->G * d->q_n * d->H;
        const int blocks = (int)((n + 3) / 4);
        if (dtype == EGV_BF16) hipLaunchKernelGGL(attn_fwd_combine_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, a, d->B * d->G);
        else hipLaunchKernelGGL(attn_fwd_combine_kernel<float>, dim3(blocks), dim3(256), 0, st, a, d->B * d->G);
        EGV_LAUNCH_CHECK();
    }
    return 0;
}

static int egv_attn_bwd_dq_impl(int dtype, const egv_attn_desc* d, void* stream) {
    if (check_desc(d, "egv_attn_bwd_dq")) return -1;
    EGV_CHECK(d->lse && d->delta && d->dO && d->dQ, "egv_attn_bwd_dq: missing lse/delta/dO/dQ");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    AttnArgs a = to_args(d);
    if (dtype == EGV_BF16 && a.nsplit == 1 && egv_attn_dq_mfma(a, d->B, st)) {
        EGV_LAUNCH_CHECK();
        return 0;
    }
    if (a.nsplit > 1)
        EGV_CHECK(d->ws && d->ws_bytes >= egv_attn_split_workspace_bytes(1, d->B, d->G, d->H, d->q_n, a.nsplit),
                  "egv_attn_bwd_dq: workspace too small");
    if (d->q_n == 1 && a.nsplit > 1) {
        dim3 grid1(a.nsplit, d->B * d->G, d->H);
        if (dtype == EGV_BF16) hipLaunchKernelGGL(attn1_dq_kernel<bf16_t>, grid1, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(attn1_dq_kernel<float>, grid1, dim3(256), 0, st, a);
    } else if (dtype == EGV_BF16 && a.nsplit > 1 && egv_attn_dq_mfma(a, d->B, st)) {
    } else {
        const int nw = pick_nw(d->q_n);
        dim3 grid(((d->q_n + nw * QPW - 1) / (nw * QPW)) * a.nsplit, d->B * d->G, d->H);
        const size_t sm = dq_smem(nw);
        EGV_ATTN_LAUNCH(attn_bwd_dq_kernel, nw, sm, grid);
    }
    EGV_LAUNCH_CHECK();
    if (a.nsplit > 1) {
        const long long n = (long long)d->B * d->G * d->q_n * d->H;
        const int blocks = (int)((n + 3) / 4);
        if (dtype == EGV_BF16) hipLaunchKernelGGL(attn_dq_combine_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, a, d->B * d->G);
        else hipLaunchKernelGGL(attn_dq_combine_kernel<float>, dim3(blocks), dim3(256), 0, st, a, d->B * d->G);
        EGV_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" long long egv_attn_bwd_dkv_workspace_bytes(int B, int G, int H, int k_n, int nsplit) {
    if (nsplit <= 1) return 0;
    return (long long)nsplit * B * G * k_n * H * 2 * HD * 4;
}

static int egv_attn_bwd_dkv_impl(int dtype, const egv_attn_desc* d, void* stream) {
    if (check_desc(d, "egv_attn_bwd_dkv")) return -1;
    EGV_CHECK(d->lse && d->delta && d->dO && d->dK && d->dV, "egv_attn_bwd_dkv: missing lse/delta/dO/dK/dV");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    AttnArgs a = to_args(d);
    if (dtype == EGV_BF16 && a.nsplit == 1 && egv_attn_dkv_mfma(a, d->B, st)) {
        EGV_LAUNCH_CHECK();
        return 0;
    }
    if (a.nsplit > 1)
        EGV_CHECK(d->ws && d->ws_bytes >= egv_attn_bwd_dkv_workspace_bytes(d->B, d->G, d->H, d->k_n, a.nsplit),
                  "egv_attn_bwd_dkv: workspace too small");
    if (d->k_n == 1 && a.nsplit > 1) {
        dim3 grid1(a.nsplit, d->B * d->G, d->H);
        if (dtype == EGV_BF16) hipLaunchKernelGGL(attn1_dkv_kernel<bf16_t>, grid1, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(attn1_dkv_kernel<float>, grid1, dim3(256), 0, st, a);
    } else if (dtype == EGV_BF16 && a.nsplit > 1 && egv_attn_dkv_mfma(a, d->B, st)) {
        // long query side, short key side (image->text cross attention): MFMA kernel per query chunk, fp32 partials
    } else {
        const int ntot = d->q_n + d->extra;
        const int per = (ntot + a.nsplit - 1) / a.nsplit;
        const int nw = per >= 4 ? 4 : (per >= 2 ? 2 : 1);
        const int tiles = (d->k_n + TK - 1) / TK;
        dim3 grid(tiles * a.nsplit, d->B * d->G, d->H);
        const size_t sm = dkv_smem();
        EGV_ATTN_LAUNCH(attn_bwd_dkv_kernel, nw, sm, grid);
    }
    EGV_LAUNCH_CHECK();
    if (a.nsplit > 1) {
        const long long n = (long long)d->B * d->G * d->k_n * d->H;
        const int blocks = (int)((n + 3) / 4);
        if (dtype == EGV_BF16) hipLaunchKernelGGL(attn_dkv_reduce_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, a, d->B * d->G);
        else hipLaunchKernelGGL(attn_dkv_reduce_kernel<float>, dim3(blocks), dim3(256), 0, st, a, d->B * d->G);
        EGV_LAUNCH_CHECK();
    }
    return 0;
}

// dQ, dK, dV and delta of one grouped launch in a single kernel (bf16, other side <= 224 rows, no mask / dropout / split).
// Returns 0 if enqueued, 1 if the shape is not covered (nothing enqueued: use egv_attn_bwd_dq + egv_attn_bwd_dkv), -1 on error.
// Independent of the one-query (CLS) launches: it computes every delta it needs and stores those of the row-set queries.
// With a workspace (d->ws, egv_attn_bwd_fused_workspace_bytes) and an extra row it ALSO produces that row's dQ, dK and dV (per-
// group partials summed by a second small launch), i.e. it then replaces the two one-row launches as well.
extern "C" long long egv_attn_bwd_fused_workspace_bytes(int B, int G, int H) { return (long long)B * G * H * 3 * HD * 4; }
static int egv_attn_bwd_fused_impl(int dtype, const egv_attn_desc* d, void* stream) {
    if (check_desc(d, "egv_attn_bwd_fused")) return -1;
    EGV_CHECK(d->lse && d->delta && d->dO && d->dQ && d->dK && d->dV, "egv_attn_bwd_fused: missing lse/delta/dO/dQ/dK/dV");
    if (dtype != EGV_BF16) return 1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    AttnArgs a = to_args(d);
    a.nsplit = 1;
    if (d->ws && d->extra)
        EGV_CHECK(d->ws_bytes >= egv_attn_bwd_fused_workspace_bytes(d->B, d->G, d->H), "egv_attn_bwd_fused: workspace too small");
    if (!d->extra && d->ws && d->ws_bytes >= egv_attn_fewkeys_workspace_bytes(d->B, d->G, d->H, d->q_n) && egv_attn_fewkeys_bwd(a, d->B, st)) {
        EGV_LAUNCH_CHECK();                      // many queries over <= 32 keys, mask allowed (egv_attn_cross.hip)
        return 0;
    }
    if (!d->extra && d->ws && d->q_n <= 32 && d->ws_bytes >= egv_attn_fewq_workspace_bytes(d->B, d->G, d->H, d->k_n) && egv_attn_fewq_bwd(a, d->B, st)) {
        EGV_LAUNCH_CHECK();                      // <= 32 queries over many keys, dropout allowed (egv_attn_cross.hip)
        return 0;
    }
    if (!egv_attn_bwd_fused_mfma(a, d->B, st)) return 1;
    EGV_LAUNCH_CHECK();
    return 0;
}

// The dQ + dK/dV kernel pair of a grouped launch whose groups are a single 16-row tile (the 17-row time attention) can leave the
// extra row's three gradients as per-group partials in d->ws (egv_attn_bwd_fused_workspace_bytes) when this returns 1: call
// egv_attn_bwd_dq and egv_attn_bwd_dkv with the workspace set, then egv_attn_bwd_extra_reduce(..., self_term = 1) -- the
// one-query and one-key launches of the extra row are then not needed (and egv_attn_bwd_dkv does not read the extra row's delta).
extern "C" int egv_attn_bwd_pair_covers_extra(int dtype, const egv_attn_desc* d) {
    if (dtype != EGV_BF16 || !d || !d->ws || d->ws_bytes < egv_attn_bwd_fused_workspace_bytes(d->B, d->G, d->H)) return 0;
    AttnArgs a = to_args(d);
    return egv_attn_bwd_pair_cls_ok(a) ? 1 : 0;
}
extern "C" int egv_attn_bwd_extra_reduce(int dtype, const egv_attn_desc* d, int self_term, void* stream) {
    if (check_desc(d, "egv_attn_bwd_extra_reduce")) return -1;
    EGV_CHECK(dtype == EGV_BF16 && d->ws && d->extra && d->dQ && d->dK && d->dV && d->lse && d->dO && d->O,
              "egv_attn_bwd_extra_reduce: bf16 launch with workspace, extra row, O, dO, lse and the three gradient tensors");
    EGV_CHECK(d->ws_bytes >= egv_attn_bwd_fused_workspace_bytes(d->B, d->G, d->H), "egv_attn_bwd_extra_reduce: workspace too small");
    AttnArgs a = to_args(d);
    egv_attn_bwd_cls_reduce_launch(a, d->B, self_term, reinterpret_cast<hipStream_t>(stream));
    EGV_LAUNCH_CHECK();
    return 0;
}

// ---- C entry points: the launch logic above, bracketed by the HIP-event instrumentation of bench.py's roofline leg (a no-op unless
// egv_prof_enable(1)): kind 20 forward, 21 dQ, 22 dK/dV, 23 one-pass backward; bytes = algorithmic traffic (each operand once)
void* egv_prof_begin(void* stream);
void egv_prof_end(void* handle, void* stream, double flops, int kind, double bytes);
static void attn_work(int dtype, const egv_attn_desc* d, int kind, double& flops, double& bytes) {
    const double rq = (double)d->B * d->G * d->q_n + (double)d->B * (d->extra ? 1 : 0), rk = (double)d->B * d->G * d->k_n + (double)d->B * (d->extra ? 1 : 0);
    const double row = (double)d->H * 64 * (dtype == EGV_BF16 ? 2 : 4);
    const double pairs = (double)d->B * d->G * d->H * ((double)d->q_n + (d->extra ? 1 : 0)) * ((double)d->k_n + (d->extra ? 1 : 0));
    const double mm = 2.0 * 64 * pairs;                           // one q k^T-sized product
    if (kind == 20) { flops = 2 * mm; bytes = row * (2 * rq + 2 * rk); }
    else if (kind == 21) { flops = 3 * mm; bytes = row * (3 * rq + 2 * rk); }
    else if (kind == 22) { flops = 4 * mm; bytes = row * (2 * rq + 4 * rk); }
    else { flops = 5 * mm; bytes = row * (4 * rq + 4 * rk); }
}
#define EGV_ATTN_ENTRY(NAME, KIND)                                                              \
    extern "C" int NAME(int dtype, const egv_attn_desc* d, void* stream) {                      \
        void* ph = egv_prof_begin(stream);                                                      \
        const int rc = NAME##_impl(dtype, d, stream);                                           \
        if (ph) {                                                                               \
            double fl = 0, by = 0;                                                              \
            if (d && rc == 0) attn_work(dtype, d, KIND, fl, by);                                \
            egv_prof_end(ph, stream, fl, rc == 0 ? KIND : -1, by);                              \
        }                                                                                       \
        return rc;                                                                              \
    }
EGV_ATTN_ENTRY(egv_attn_fwd, 20)
EGV_ATTN_ENTRY(egv_attn_bwd_dq, 21)
EGV_ATTN_ENTRY(egv_attn_bwd_dkv, 22)
EGV_ATTN_ENTRY(egv_attn_bwd_fused, 23)
#undef EGV_ATTN_ENTRY
