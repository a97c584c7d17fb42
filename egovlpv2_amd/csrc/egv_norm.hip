// HBM-bound row kernels of the EgoVLPv2 hot path: LayerNorm fwd/bwd (SURVEY.md K2/K7/K8 prologues),
// column sums (bias gradients), dot reductions (gate gradients), dtype casts.
// One 64-lane wavefront per token row, 8-16 B vector accesses, fp32 statistics.
#include "egv_common.h"
#include <cstdlib>

namespace egv {

constexpr int LN_MAXV = 4;   // up to 4 vectors of 4 elements per lane -> D <= 1024

// y2 (optional): a bf16 copy of the output -- the fp32 residual stream of the text tower keeps y in fp32 and feeds the next Linear
// (a bf16 MFMA GEMM) from the copy
// MXQ (bf16 only, D % 128 == 0): also emit the MX-fp8 form of the output (egv_mx.hip: e4m3 codes q[M][D] and the E8M0 scale bytes in
// the role-0 lane order) -- the A operand of the Linear that follows, quantised from the bf16-rounded values exactly as egv_quant_mx
// would quantise y: a 32-element block is the 4 x 8 elements of 8 neighbouring lanes
template <typename T, bool MXQ = false>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ stats, int M, int D, float eps, bf16_t* __restrict__ y2 = nullptr,
                                                            unsigned char* __restrict__ mxq = nullptr, unsigned char* __restrict__ mxs = nullptr) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave_id();
    if (row >= M) return;
    const T* xr = x + (size_t)row * D;
    float v[LN_MAXV][4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int c = (j * 64 + lane) * 4;
        if (c < D) {
            ld4(xr + c, v[j]);
            s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
        } else {
            v[j][0] = v[j][1] = v[j][2] = v[j][3] = 0.f;
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int c = (j * 64 + lane) * 4;
        if (c < D) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[j][e] - mean;
                q += d * d;
            }
        }
    }
    const float var = wave_sum(q) / (float)D;
    const float rstd = rsqrtf(var + eps);
    if (stats && lane == 0) {
        stats[2 * (size_t)row] = mean;
        stats[2 * (size_t)row + 1] = rstd;
    }
    T* yr = y + (size_t)row * D;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int c = (j * 64 + lane) * 4;
        if (c < D) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mean) * rstd * gamma[c + e] + beta[c + e];
            st4(yr + c, o);
            if (y2) st4(y2 + (size_t)row * D + c, o);
            if constexpr (MXQ) {
                float r[4], amax = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { r[e] = bf2f(f2bf(o[e])); amax = fmaxf(amax, fabsf(r[e])); }
                amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
                amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
                amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
                const unsigned int bits = __float_as_uint(amax);
                int e8 = (int)(bits >> 23) - 8 + ((bits & 0x7fffffu) > 0x600000u ? 1 : 0);
                e8 = e8 < 0 ? 0 : (e8 > 254 ? 254 : e8);
                const int sh = 127 - e8;
                float a4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) a4[e] = fminf(fmaxf(__builtin_amdgcn_ldexpf(r[e], sh), -448.f), 448.f);
                int pk = __builtin_amdgcn_cvt_pk_fp8_f32(a4[0], a4[1], 0, false);
                pk = __builtin_amdgcn_cvt_pk_fp8_f32(a4[2], a4[3], pk, true);
                *reinterpret_cast<int*>(mxq + (size_t)row * D + c) = pk;
                if ((lane & 7) == 0) {
                    const int kb = c >> 5, blk = row / 48, rb = row - blk * 48, nblk = ((M + 191) / 192) * 4;
                    mxs[(((size_t)(kb >> 2) * nblk + blk) * 4 + (kb & 3)) * 64 + (rb & 15) * 4 + (rb >> 4)] = (unsigned char)e8;
                }
            }
        }
    }
}

// fp32 residual stream of the video tower in the bf16 mode (EGV_BLOCK_RES_F32 of egv_vblock_fwd): the residual sums of a
// SpaceTimeBlock (video_transformer.py:218,222,226) and the LayerNorm that reads them are formed in fp32, as torch.autocast keeps
// them (trainer_egoclip.py:143) -- s = base + d1 + d2 + gate * dg with base in fp32 (or, at the head of the stream, bf16) and the
// Linear outputs d1 / d2 / dg in bf16; s leaves as fp32 (sum32) and / or bf16 (sum16: what the backward pass and the GEMMs read),
// y = LayerNorm(s) in bf16 is the next Linear's operand.  One wave per row, every operand read once.
#ifndef EGV_SUMLN_ROWS
#define EGV_SUMLN_ROWS 1
#endif
template <int R>                                                   // rows per wave, all of their loads in flight together
__global__ __launch_bounds__(256) void sum_ln_kernel(const float* __restrict__ base32, const bf16_t* __restrict__ base16,
                                                     const bf16_t* d1, const bf16_t* d2, const bf16_t* dg,      // (sum16 may alias one of them)
                                                     const float* __restrict__ gate,
                                                     float* __restrict__ sum32, bf16_t* sum16, bf16_t* __restrict__ y,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ stats, int M, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + wave_id()) * R;
    if (row0 >= M) return;
    const float gt = (dg && gate) ? *gate : 1.0f;
    float v[R][LN_MAXV][4];
    float t1[R][LN_MAXV][4], t2[R][LN_MAXV][4], t3[R][LN_MAXV][4];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const size_t ro = (size_t)min(row0 + r, M - 1) * D;         // (a row past the end re-reads the last one and stores nothing)
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int c = (j * 64 + lane) * 4;
            if (c < D) {
                if (base32) ld4(base32 + ro + c, v[r][j]);
                else ld4(base16 + ro + c, v[r][j]);
                if (d1) ld4(d1 + ro + c, t1[r][j]);
                if (d2) ld4(d2 + ro + c, t2[r][j]);
                if (dg) ld4(dg + ro + c, t3[r][j]);
            }
        }
    }
    float s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool live = row0 + r < M;
        const size_t ro = (size_t)min(row0 + r, M - 1) * D;
        s[r] = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int c = (j * 64 + lane) * 4;
            if (c < D) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = v[r][j][e];
                    if (d1) a += t1[r][j][e];
                    if (d2) a += t2[r][j][e];
                    if (dg) a += gt * t3[r][j][e];
                    v[r][j][e] = a;
                }
                if (live) {
                    if (sum32) st4(sum32 + ro + c, v[r][j]);
                    if (sum16) st4(sum16 + ro + c, v[r][j]);
                }
                s[r] += v[r][j][0] + v[r][j][1] + v[r][j][2] + v[r][j][3];
            } else {
                v[r][j][0] = v[r][j][1] = v[r][j][2] = v[r][j][3] = 0.f;
            }
        }
    }
    if (!y) return;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r;
        const float mean = wave_sum(s[r]) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int c = (j * 64 + lane) * 4;
            if (c < D) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dlt = v[r][j][e] - mean;
                    q += dlt * dlt;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
        if (row >= M) continue;
        if (stats && lane == 0) {
            stats[2 * (size_t)row] = mean;
            stats[2 * (size_t)row + 1] = rstd;
        }
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int c = (j * 64 + lane) * 4;
            if (c < D) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[r][j][e] - mean) * rstd * gamma[c + e] + beta[c + e];
                st4(y + (size_t)row * D + c, o);
            }
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  partial dgamma/dbeta per workgroup.
// TD / TX / TA / TO: types of dy, x, the skip gradients and dx (all T in the uniform modes; the fp32 residual stream of the text
// tower has a bf16 dy -- a data gradient leaving a GEMM -- beside fp32 x / skips / dx).  dy_b (optional, fp32): a second part of
// the incoming gradient, dy = dy + dy_b.
template <typename TD, typename TX = TD, typename TA = TD, typename TO = TD>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const TD* __restrict__ dy, const TX* __restrict__ x,
                                                            const float* __restrict__ stats, const float* __restrict__ gamma,
                                                            const TA* __restrict__ add, const TA* __restrict__ add2, TO* __restrict__ dx,
                                                            float* __restrict__ partial, int M, int D, int rows_per_block,
                                                            const float* __restrict__ dy_b = nullptr) {
    __shared__ float red[4][2][LN_MAXV * 256];
    const int lane = threadIdx.x & 63;
    const int w = wave_id();
    float dg[LN_MAXV][4], db[LN_MAXV][4], gm[LN_MAXV][4];
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int c = (j * 64 + lane) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dg[j][e] = 0.f;
            db[j][e] = 0.f;
            gm[j][e] = (c < D) ? gamma[c + e] : 0.f;
        }
    }
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    // two rows per wave and iteration (independent register sets): twice the loads in flight -- the single-row loop ran at
    // 2.1 TB/s on the 116 MB of a video-token call, latency-bound on its 6 eight-byte loads per wave
    for (int row0 = r0 + w; row0 < r1; row0 += 8) {
        float xh[2][LN_MAXV][4], g[2][LN_MAXV][4], av[2][LN_MAXV][4];
        float mean[2], rstd[2], s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
        bool live[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = row0 + u * 4;
            live[u] = row < r1;
            const int rr = live[u] ? row : r0;
            mean[u] = stats[2 * (size_t)rr];
            rstd[u] = stats[2 * (size_t)rr + 1];
#pragma unroll
            for (int j = 0; j < LN_MAXV; ++j) {
                const int c = (j * 64 + lane) * 4;
                if (c < D) {
                    float xv[4], dv[4] = {0.f, 0.f, 0.f, 0.f};
                    ld4(x + (size_t)rr * D + c, xv);
                    if (dy) ld4(dy + (size_t)rr * D + c, dv);
                    if (dy_b) {
                        float d2[4];
                        ld4(dy_b + (size_t)rr * D + c, d2);
#pragma unroll
                        for (int e = 0; e < 4; ++e) dv[e] += d2[e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[u][j][e] = 0.f;
                    if (add) {
                        float a[4];
                        ld4(add + (size_t)rr * D + c, a);
#pragma unroll
                        for (int e = 0; e < 4; ++e) av[u][j][e] = a[e];
                    }
                    if (add2) {
                        float a[4];
                        ld4(add2 + (size_t)rr * D + c, a);
#pragma unroll
                        for (int e = 0; e < 4; ++e) av[u][j][e] += a[e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        xh[u][j][e] = (xv[e] - mean[u]) * rstd[u];
                        g[u][j][e] = dv[e] * gm[j][e];
                        s1[u] += g[u][j][e];
                        s2[u] += g[u][j][e] * xh[u][j][e];
                        if (live[u]) {
                            dg[j][e] += dv[e] * xh[u][j][e];
                            db[j][e] += dv[e];
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) xh[u][j][e] = g[u][j][e] = av[u][j][e] = 0.f;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            s1[u] = wave_sum(s1[u]) / (float)D;
            s2[u] = wave_sum(s2[u]) / (float)D;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!live[u]) continue;
            const int row = row0 + u * 4;
#pragma unroll
            for (int j = 0; j < LN_MAXV; ++j) {
                const int c = (j * 64 + lane) * 4;
                if (c < D) {
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = rstd[u] * (g[u][j][e] - s1[u] - xh[u][j][e] * s2[u]) + av[u][j][e];
                    st4(dx + (size_t)row * D + c, o);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[w][0][(j * 64 + lane) * 4 + e] = dg[j][e];
            red[w][1][(j * 64 + lane) * 4 + e] = db[j][e];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            a += red[ww][0][c];
            b += red[ww][1][c];
        }
        partial[(size_t)blockIdx.x * 2 * D + c] = a;
        partial[(size_t)blockIdx.x * 2 * D + D + c] = b;
    }
}

// bf16 specialisation for D = NJ * 256 (every lane owns NJ full 4-column vectors: no column predicates), R rows per wave and
// iteration, branch-free.  Measured in the training step (where the rows come from HBM; a micro-benchmark replays them from the
// last-level cache and shows the opposite): 92.8 -> 90.4 ms per step against layernorm_bwd_kernel<bf16>.  The rows stay PACKED (bf16 pairs) in registers between the statistics pass and the output pass -- 6
// instead of 12 registers per row and operand at D = 768 -- so more rows (x, dy and the addends) are in flight per wave, and the
// row sums use DPP exchanges instead of the LDS butterfly.  The order of the per-lane partial sums, of the rows inside a wave
// (w, w + 4, w + 8, ...) and of the cross-wave combination is the one of layernorm_bwd_kernel.
__device__ __forceinline__ void unpack4(const u32x2_t v, float (&o)[4]) {
    o[0] = __uint_as_float(v[0] << 16); o[1] = __uint_as_float(v[0] & 0xffff0000u);
    o[2] = __uint_as_float(v[1] << 16); o[3] = __uint_as_float(v[1] & 0xffff0000u);
}

template <int NJ, int RR, int NADD>
__device__ __forceinline__ void ln_bwd_rows(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const float* __restrict__ stats,
                                            const bf16_t* __restrict__ add, const bf16_t* __restrict__ add2, bf16_t* __restrict__ dx,
                                            int row0, unsigned int lane4, const float (&gm)[NJ][4], float (&dg)[NJ][4],
                                            float (&db)[NJ][4]) {
    constexpr int D = NJ * 256;
    u32x2_t px[RR][NJ], pd[RR][NJ], pa[RR][NJ], pb[RR][NJ];
    float mean[RR], rstd[RR], s1[RR], s2[RR];
#pragma unroll
    for (int u = 0; u < RR; ++u) {
        const size_t rr = row0 + u * 4;
        mean[u] = stats[2 * rr];
        rstd[u] = stats[2 * rr + 1];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {                             // uniform row base (scalar) + one lane offset for all loads
            px[u][j] = *reinterpret_cast<const u32x2_t*>(x + rr * D + j * 256 + lane4);
            pd[u][j] = *reinterpret_cast<const u32x2_t*>(dy + rr * D + j * 256 + lane4);
            if constexpr (NADD >= 1) pa[u][j] = *reinterpret_cast<const u32x2_t*>(add + rr * D + j * 256 + lane4);
            if constexpr (NADD >= 2) pb[u][j] = *reinterpret_cast<const u32x2_t*>(add2 + rr * D + j * 256 + lane4);
        }
    }
#pragma unroll
    for (int u = 0; u < RR; ++u) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float xv[4], dv[4];
            unpack4(px[u][j], xv);
            unpack4(pd[u][j], dv);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (xv[e] - mean[u]) * rstd[u];
                const float g = dv[e] * gm[j][e];
                a1 += g;
                a2 += g * xh;
                dg[j][e] += dv[e] * xh;
                db[j][e] += dv[e];
            }
        }
        s1[u] = a1;
        s2[u] = a2;
    }
#pragma unroll
    for (int u = 0; u < RR; ++u) {
        s1[u] = wave_sum_dpp(s1[u]) / (float)D;
        s2[u] = wave_sum_dpp(s2[u]) / (float)D;
    }
    // the output pass unpacks the rows again: without this the compiler keeps the floats of the statistics pass alive
#pragma unroll
    for (int u = 0; u < RR; ++u)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            asm volatile("" : "+v"(px[u][j][0]), "+v"(px[u][j][1]), "+v"(pd[u][j][0]), "+v"(pd[u][j][1]));
#pragma unroll
    for (int u = 0; u < RR; ++u) {
        const size_t row = row0 + u * 4;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float xv[4], dv[4], o[4];
            unpack4(px[u][j], xv);
            unpack4(pd[u][j], dv);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (xv[e] - mean[u]) * rstd[u];
                o[e] = rstd[u] * (dv[e] * gm[j][e] - s1[u] - xh * s2[u]);
            }
            if constexpr (NADD >= 1) {
                float av[4];
                unpack4(pa[u][j], av);
                if constexpr (NADD >= 2) {
                    float bv[4];
                    unpack4(pb[u][j], bv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[e] += bv[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += av[e];
            }
            st4(dx + row * D + j * 256 + lane4, o);
        }
    }
}

template <int NJ, int R, int NADD>
__global__ __launch_bounds__(256, 2) void layernorm_bwd_bf16_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                    const float* __restrict__ stats, const float* __restrict__ gamma,
                                                                    const bf16_t* __restrict__ add, const bf16_t* __restrict__ add2,
                                                                    bf16_t* __restrict__ dx, float* __restrict__ partial, int M,
                                                                    int rows_per_block) {
    constexpr int D = NJ * 256;
    __shared__ float red[4][2][D];
    const int lane = threadIdx.x & 63;
    const int w = wave_id();
    const unsigned int lane4 = lane * 4;
    float dg[NJ][4], db[NJ][4], gm[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dg[j][e] = 0.f;
            db[j][e] = 0.f;
            gm[j][e] = gamma[j * 256 + lane4 + e];
        }
    }
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    int row0 = r0 + w;
    for (; row0 + 4 * (R - 1) < r1; row0 += 4 * R) ln_bwd_rows<NJ, R, NADD>(dy, x, stats, add, add2, dx, row0, lane4, gm, dg, db);
    for (; row0 < r1; row0 += 4) ln_bwd_rows<NJ, 1, NADD>(dy, x, stats, add, add2, dx, row0, lane4, gm, dg, db);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[w][0][j * 256 + lane4 + e] = dg[j][e];
            red[w][1][j * 256 + lane4 + e] = db[j][e];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            a += red[ww][0][c];
            b += red[ww][1][c];
        }
        partial[(size_t)blockIdx.x * 2 * D + c] = a;
        partial[(size_t)blockIdx.x * 2 * D + D + c] = b;
    }
}

// out[c] = scale * gate * sum_p partial[p][c]   (deterministic order).  64 columns per workgroup, the P partial rows are
// split over 16 waves and combined through LDS in a fixed order.
__global__ __launch_bounds__(1024) void colsum_partials_kernel(const float* __restrict__ partial, float* __restrict__ out, int P,
                                                               int ncol, int stride, float scale, const float* gate) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, w = wave_id();
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < ncol) {
        // eight loads in flight per thread (a dependent load-add chain over P / 16 rows was 12 us for 3 MB); summed in row order
        int p = w;
        for (; p + 7 * 16 < P; p += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(p + u * 16) * stride + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; p < P; p += 16) s += partial[(size_t)p * stride + c];
    }
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && c < ncol) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][lane];
        out[c] = t * scale * (gate ? *gate : 1.0f);
    }
}

// The same sum for up to eight partial buffers in ONE launch (blockIdx.y = buffer): the LayerNorm parameter gradients of a block's backward
// call, whose three or four reductions are otherwise dependent 6-us launches on the calling stream although nothing in the call reads
// their results.  Per buffer the code, the row split and the order of the sums are those of colsum_partials_kernel: the same bits.
struct ColsumBatch {
    const float* partial[8];
    float* out[8];            // [dgamma ; dbeta] contiguous (out2 == nullptr) or dgamma
    float* out2[8];           // dbeta when it is not out + D
    int P[8];
    int D, n;
};
__global__ __launch_bounds__(1024) void colsum_partials_batch_kernel(const ColsumBatch b) {
    __shared__ float red[16][64];
    const int e = blockIdx.y;
    const float* __restrict__ partial = b.partial[e];
    const int P = b.P[e], ncol = 2 * b.D, stride = 2 * b.D;
    const int lane = threadIdx.x & 63, w = wave_id();
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < ncol) {
        int p = w;
        for (; p + 7 * 16 < P; p += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(p + u * 16) * stride + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; p < P; p += 16) s += partial[(size_t)p * stride + c];
    }
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && c < ncol) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][lane];
        t = t * 1.0f;
        if (b.out2[e] && c >= b.D) b.out2[e][c - b.D] = t;
        else b.out[e][c] = t;
    }
}

// partial column sums of X[M,N]: grid (ceil(N/256), chunks); each lane owns 4 columns, waves split rows.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, float* __restrict__ partial, int M, int N, int ld,
                                                     int rows_per_block) {
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, w = wave_id();
    const int c = blockIdx.x * 256 + lane * 4;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const bool vec = ((ld & 3) == 0) && (c + 3 < N);
    for (int r = r0 + w; r < r1; r += 4) {
        if (vec) {
            float v[4];
            ld4(X + (size_t)r * ld + c, v);
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < N) s[e] += Elem<T>::ld(X + (size_t)r * ld + c + e);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[w][lane * 4 + e] = s[e];
    __syncthreads();
    const int cc = blockIdx.x * 256 + threadIdx.x;
    if (cc < N) partial[(size_t)blockIdx.y * N + cc] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

template <typename T>
__global__ __launch_bounds__(256) void dot_partial_kernel(const T* __restrict__ a, const T* __restrict__ b, float* __restrict__ partial,
                                                          long long n) {
    __shared__ float red[4];
    float s = 0.f;
    const long long nv = n / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
        float x[4], y[4];
        ld4(a + i * 4, x);
        ld4(b + i * 4, y);
        s += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long long i = nv * 4 + threadIdx.x;
        s += Elem<T>::ld(a + i) * Elem<T>::ld(b + i);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[wave_id()] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void sum_partials_kernel(const float* __restrict__ partial, float* __restrict__ out, int P, float scale) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < P; i += 256) s += partial[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[wave_id()] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) * scale;
}

template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, long long n) {
    const long long nv = n / 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        float v[4];
        ld4(src + i * 4, v);
        st4(dst + i * 4, v);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long long i = nv * 4 + threadIdx.x;
        Elem<D>::st(dst + i, Elem<S>::ld(src + i));
    }
}

// dst[c][r] (bf16) = src[r][c] (fp32): transposed compute copy of a weight matrix (dgrad runs in the NT form on it)
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int R, int Cc) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < Cc) ? src[(size_t)r * Cc + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < Cc && r < R) dst[(size_t)c * R + r].v = f2bf(tile[tx][i]);
    }
}

// dst[C,R] (T2) = transpose(src[R,C] (T1, row pitch ld)) through a padded 64x64 LDS tile
template <typename T1, typename T2>
__global__ __launch_bounds__(256) void transpose_kernel(const T1* __restrict__ src, T2* __restrict__ dst, int R, int Cc, int ld) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < Cc) ? Elem<T1>::ld(src + (size_t)r * ld + c) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < Cc && r < R) Elem<T2>::st(dst + (size_t)c * R + r, tile[tx][i]);
    }
}

}  // namespace egv
using namespace egv;

extern "C" int egv_transpose(int dtype_src, int dtype_dst, const void* src, void* dst, int R, int Cc, int ld, void* stream) {
    EGV_CHECK(R > 0 && Cc > 0 && ld >= Cc, "egv_transpose: bad shape R=%d C=%d ld=%d", R, Cc, ld);
    const dim3 grid((Cc + 63) / 64, (R + 63) / 64);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype_src == EGV_BF16 && dtype_dst == EGV_BF16)
        hipLaunchKernelGGL((transpose_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, R, Cc, ld);
    else if (dtype_src == EGV_F32 && dtype_dst == EGV_BF16)
        hipLaunchKernelGGL((transpose_kernel<float, bf16_t>), grid, dim3(256), 0, st, (const float*)src, (bf16_t*)dst, R, Cc, ld);
    else if (dtype_src == EGV_F32 && dtype_dst == EGV_F32)
        hipLaunchKernelGGL((transpose_kernel<float, float>), grid, dim3(256), 0, st, (const float*)src, (float*)dst, R, Cc, ld);
    else
        EGV_CHECK(false, "egv_transpose: unsupported dtype pair %d -> %d", dtype_src, dtype_dst);
    EGV_LAUNCH_CHECK();
    return 0;
}

extern "C" int egv_cast_transpose(const float* src, void* dst, int R, int Cc, void* stream) {
    hipLaunchKernelGGL(cast_transpose_kernel, dim3((Cc + 63) / 64, (R + 63) / 64), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src,
                       (bf16_t*)dst, R, Cc);
    EGV_LAUNCH_CHECK();
    return 0;
}

void* egv_prof_begin(void* stream);
void egv_prof_end(void* handle, void* stream, double flops, int kind, double bytes);
struct LnProf {                 // bench.py roofline leg: kind 30 LayerNorm forward, 31 backward; bytes = every operand once
    void* h; void* st; int kind; double bytes;
    LnProf(void* stream, int k, double b) : h(egv_prof_begin(stream)), st(stream), kind(k), bytes(b) {}
    ~LnProf() { if (h) egv_prof_end(h, st, 0.0, kind, bytes); }
};

extern "C" int egv_layernorm_fwd(int dtype, const void* x, void* y, const float* gamma, const float* beta, float* stats,
                                 int M, int D, float eps, void* stream) {
    EGV_CHECK(D % 4 == 0 && D <= LN_MAXV * 256, "egv_layernorm_fwd: D=%d unsupported", D);
    EGV_CHECK(M > 0, "egv_layernorm_fwd: M=%d", M);
    LnProf prof(stream, 30, 2.0 * M * D * (dtype == EGV_BF16 ? 2 : 4));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((M + 3) / 4);
    if (dtype == EGV_BF16)
        hipLaunchKernelGGL(layernorm_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, gamma, beta, stats, M, D, eps);
    else
        hipLaunchKernelGGL(layernorm_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (float*)y, gamma, beta, stats, M, D, eps);
    EGV_LAUNCH_CHECK();
    return 0;
}

// bf16 LayerNorm that also writes the MX-fp8 form of its output (role 0: the A operand of the next Linear; BASELINE.json configs[4])
extern "C" int egv_layernorm_fwd_mx(const void* x, void* y, const float* gamma, const float* beta, float* stats, void* q, void* scales,
                                    int M, int D, float eps, void* stream) {
    EGV_CHECK(D % 128 == 0 && D <= LN_MAXV * 256 && M > 0 && q && scales, "egv_layernorm_fwd_mx: M=%d D=%d unsupported", M, D);
    LnProf prof(stream, 30, 5.0 * M * D);
    hipLaunchKernelGGL((layernorm_fwd_kernel<bf16_t, true>), dim3((M + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       (const bf16_t*)x, (bf16_t*)y, gamma, beta, stats, M, D, eps, (bf16_t*)nullptr, (unsigned char*)q, (unsigned char*)scales);
    EGV_LAUNCH_CHECK();
    return 0;
}

// fp32 residual stream of the text tower in the bf16 mode (bf16 GEMM operands, LayerNorm input / output and residual sums in fp32:
// what torch.autocast does with nn.LayerNorm, trainer/trainer_egoclip.py:143; roberta.py:336-345, :417-426): y fp32 and, when
// y16 != NULL, a bf16 copy for the next Linear
extern "C" int egv_layernorm_fwd_res32(const float* x, float* y, void* y16, const float* gamma, const float* beta, float* stats, int M, int D,
                                       float eps, void* stream) {
    EGV_CHECK(D % 4 == 0 && D <= LN_MAXV * 256 && M > 0, "egv_layernorm_fwd_res32: M=%d D=%d unsupported", M, D);
    hipLaunchKernelGGL(layernorm_fwd_kernel<float>, dim3((M + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y, gamma, beta, stats,
                       M, D, eps, (bf16_t*)y16);
    EGV_LAUNCH_CHECK();
    return 0;
}

extern "C" int egv_sum_layernorm(const float* base32, const void* base16, const void* d1, const void* d2, const void* dg, const float* gate,
                                 float* sum32, void* sum16, void* y, const float* gamma, const float* beta, float* stats, int M, int D,
                                 float eps, void* stream) {
    EGV_CHECK(D % 4 == 0 && D <= LN_MAXV * 256 && M > 0, "egv_sum_layernorm: M=%d D=%d unsupported", M, D);
    EGV_CHECK((base32 != nullptr) != (base16 != nullptr), "egv_sum_layernorm: exactly one of base32 / base16");
    EGV_CHECK(!y || (gamma && beta), "egv_sum_layernorm: LayerNorm output without its affine terms");
    EGV_CHECK(y || sum32 || sum16, "egv_sum_layernorm: no output");
    double bytes = (double)M * D * ((base32 ? 4 : 2) + (d1 ? 2 : 0) + (d2 ? 2 : 0) + (dg ? 2 : 0) + (sum32 ? 4 : 0) + (sum16 ? 2 : 0) + (y ? 2 : 0));
    LnProf prof(stream, 30, bytes);
    constexpr int R = EGV_SUMLN_ROWS;
    hipLaunchKernelGGL(sum_ln_kernel<R>, dim3((M + 4 * R - 1) / (4 * R)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), base32, (const bf16_t*)base16,
                       (const bf16_t*)d1, (const bf16_t*)d2, (const bf16_t*)dg, gate, sum32, (bf16_t*)sum16, (bf16_t*)y, gamma, beta, stats, M, D, eps);
    EGV_LAUNCH_CHECK();
    return 0;
}

static inline int ln_bwd_blocks(int M) {
    int nb = (M + 3) / 4;
    static const int cap = egv_cfg_int("EGV_LN_BLOCKS", 512);
    return nb > cap ? cap : nb;          // 2 workgroups per CU (two rows in flight per wave); the partial-sum reduction reads nb rows
}

extern "C" long long egv_layernorm_bwd_workspace_bytes(int M, int D) { return (long long)ln_bwd_blocks(M) * 2 * D * 4; }

// ---- deferred reduction of the LayerNorm parameter-gradient partials (block executor, egv_block.cpp) -------------------------------
// Between egv_ln_bwd_defer_begin(ws, bytes) and egv_ln_bwd_defer_flush(stream) every egv_layernorm_bwd2 call of this host thread (bf16 / fp32
// forms of this file with the [nb][2][D] partial layout) writes its partials into its own slice of `ws` and does NOT launch its
// reduction; the flush sums all of them in one launch.  A call that does not fit (more than eight, slice too small) reduces at once as before.
namespace {
struct LnDefer {
    bool on = false;
    char* base = nullptr;
    size_t cap = 0, off = 0;
    egv::ColsumBatch b{};
};
thread_local LnDefer g_ln_defer;
}  // namespace
void egv_ln_bwd_defer_begin(void* ws, long long bytes) {
    static const bool enabled = egv_cfg_on("EGV_LN_DEFER", true);
    LnDefer& d = g_ln_defer;
    d.on = enabled && ws && bytes > 0;
    d.base = (char*)ws; d.cap = (size_t)(bytes > 0 ? bytes : 0); d.off = 0;
    d.b = egv::ColsumBatch{};
}
int egv_ln_bwd_defer_flush(void* stream) {
    LnDefer& d = g_ln_defer;
    d.on = false;
    if (d.b.n == 0) return 0;
    hipLaunchKernelGGL(colsum_partials_batch_kernel, dim3((2 * d.b.D + 63) / 64, d.b.n), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), d.b);
    d.b.n = 0;
    EGV_LAUNCH_CHECK();
    return 0;
}
// a slice for nb2 partial rows of 2 D floats, or nullptr (-> immediate reduction)
static float* ln_defer_slot(int nb2, int D, float* dgamma, float* dbeta) {
    LnDefer& d = g_ln_defer;
    if (!d.on || d.b.n >= 8 || (d.b.n > 0 && d.b.D != D)) return nullptr;
    const size_t need = (size_t)nb2 * 2 * D * 4;
    const size_t off = (d.off + 255) & ~(size_t)255;
    if (off + need > d.cap) return nullptr;
    float* p = (float*)(d.base + off);
    d.off = off + need;
    const int e = d.b.n++;
    d.b.partial[e] = p; d.b.out[e] = dgamma; d.b.out2[e] = (dbeta == dgamma + D) ? nullptr : dbeta; d.b.P[e] = nb2; d.b.D = D;
    return p;
}

extern "C" int egv_layernorm_bwd2(int dtype, const void* dy, const void* x, const float* stats, const float* gamma,
                                  const void* add, const void* add2, void* dx, float* dgamma, float* dbeta, int M, int D,
                                  void* workspace, void* stream);
extern "C" int egv_layernorm_bwd(int dtype, const void* dy, const void* x, const float* stats, const float* gamma,
                                 const void* add, void* dx, float* dgamma, float* dbeta, int M, int D, void* workspace,
                                 void* stream) {
    return egv_layernorm_bwd2(dtype, dy, x, stats, gamma, add, nullptr, dx, dgamma, dbeta, M, D, workspace, stream);
}
// dx = LN'(dy) + add + add2: the second addend is the other skip path of a divided space-time block (x feeds the time
// residual AND the space residual, video_transformer.py:218,222)
extern "C" int egv_layernorm_bwd2(int dtype, const void* dy, const void* x, const float* stats, const float* gamma,
                                  const void* add, const void* add2, void* dx, float* dgamma, float* dbeta, int M, int D,
                                  void* workspace, void* stream) {
    EGV_CHECK(D % 4 == 0 && D <= LN_MAXV * 256, "egv_layernorm_bwd: D=%d unsupported", D);
    LnProf prof(stream, 31, (3.0 + (add != nullptr) + (add2 != nullptr)) * M * D * (dtype == EGV_BF16 ? 2 : 4));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nb = ln_bwd_blocks(M);
    const int rpb = (M + nb - 1) / nb;
    const int nb2 = (M + rpb - 1) / rpb;
    float* partial = (float*)workspace;
    float* const slot = ln_defer_slot(nb2, D, dgamma, dbeta);        // block executor: the reduction waits for the call's flush
    if (slot) partial = slot;
    static const int packed = egv_cfg_int("EGV_LN_PACKED", 1);
    if (dtype == EGV_BF16 && packed && D == 1024) {                  // ViT-L / RoBERTa-large width: two packed rows per wave
        const bf16_t *pdy = (const bf16_t*)dy, *pxx = (const bf16_t*)x, *pa = (const bf16_t*)add, *pb = (const bf16_t*)add2;
        if (pb && !pa) { pa = pb; pb = nullptr; }
        if (pb) hipLaunchKernelGGL((layernorm_bwd_bf16_kernel<4, 2, 2>), dim3(nb2), dim3(256), 0, st, pdy, pxx, stats, gamma, pa, pb, (bf16_t*)dx, partial, M, rpb);
        else if (pa) hipLaunchKernelGGL((layernorm_bwd_bf16_kernel<4, 2, 1>), dim3(nb2), dim3(256), 0, st, pdy, pxx, stats, gamma, pa, pb, (bf16_t*)dx, partial, M, rpb);
        else hipLaunchKernelGGL((layernorm_bwd_bf16_kernel<4, 2, 0>), dim3(nb2), dim3(256), 0, st, pdy, pxx, stats, gamma, pa, pb, (bf16_t*)dx, partial, M, rpb);
    }
    else if (dtype == EGV_BF16 && packed && D == 768) {
        const bf16_t *pdy = (const bf16_t*)dy, *pxx = (const bf16_t*)x, *pa = (const bf16_t*)add, *pb = (const bf16_t*)add2;
        if (pb && !pa) { pa = pb; pb = nullptr; }
        if (pb) hipLaunchKernelGGL((layernorm_bwd_bf16_kernel<3, 4, 2>), dim3(nb2), dim3(256), 0, st, pdy, pxx, stats, gamma, pa, pb, (bf16_t*)dx, partial, M, rpb);
        else if (pa) hipLaunchKernelGGL((layernorm_bwd_bf16_kernel<3, 4, 1>), dim3(nb2), dim3(256), 0, st, pdy, pxx, stats, gamma, pa, pb, (bf16_t*)dx, partial, M, rpb);
        else hipLaunchKernelGGL((layernorm_bwd_bf16_kernel<3, 4, 0>), dim3(nb2), dim3(256), 0, st, pdy, pxx, stats, gamma, pa, pb, (bf16_t*)dx, partial, M, rpb);
    }
    else if (dtype == EGV_BF16)
        hipLaunchKernelGGL(layernorm_bwd_kernel<bf16_t>, dim3(nb2), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, stats, gamma,
                           (const bf16_t*)add, (const bf16_t*)add2, (bf16_t*)dx, partial, M, D, rpb);
    else
        hipLaunchKernelGGL(layernorm_bwd_kernel<float>, dim3(nb2), dim3(256), 0, st, (const float*)dy, (const float*)x, stats, gamma,
                           (const float*)add, (const float*)add2, (float*)dx, partial, M, D, rpb);
    EGV_LAUNCH_CHECK();
    if (slot) return 0;
    // partial layout [nb2][2][D]: columns 0..D-1 = dgamma, D..2D-1 = dbeta
    if (dbeta == dgamma + D) {                  // caller keeps [dgamma ; dbeta] in one buffer: one reduction over 2D columns
        hipLaunchKernelGGL(colsum_partials_kernel, dim3((2 * D + 63) / 64), dim3(1024), 0, st, (const float*)partial, dgamma, nb2,
                           2 * D, 2 * D, 1.0f, (const float*)nullptr);
    } else {
        hipLaunchKernelGGL(colsum_partials_kernel, dim3((D + 63) / 64), dim3(1024), 0, st, (const float*)partial, dgamma, nb2, D,
                           2 * D, 1.0f, (const float*)nullptr);
        hipLaunchKernelGGL(colsum_partials_kernel, dim3((D + 63) / 64), dim3(1024), 0, st, (const float*)partial + D, dbeta, nb2, D,
                           2 * D, 1.0f, (const float*)nullptr);
    }
    EGV_LAUNCH_CHECK();
    return 0;
}

// dx (fp32) = LN'(dy16 + dy32) + add32 over fp32 x: dy16 (bf16, a data gradient leaving a GEMM) and dy32 (fp32, the residual path)
// may each be NULL, not both
extern "C" int egv_layernorm_bwd_res32(const void* dy16, const float* dy32, const float* x, const float* stats, const float* gamma,
                                       const float* add32, float* dx, float* dgamma, float* dbeta, int M, int D, void* workspace, void* stream) {
    EGV_CHECK(D % 4 == 0 && D <= LN_MAXV * 256 && (dy16 || dy32), "egv_layernorm_bwd_res32: D=%d unsupported or no gradient", D);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nb = ln_bwd_blocks(M);
    const int rpb = (M + nb - 1) / nb;
    const int nb2 = (M + rpb - 1) / rpb;
    float* partial = (float*)workspace;
    if (dy16)
        hipLaunchKernelGGL((layernorm_bwd_kernel<bf16_t, float, float, float>), dim3(nb2), dim3(256), 0, st, (const bf16_t*)dy16, x, stats, gamma, add32,
                           (const float*)nullptr, dx, partial, M, D, rpb, dy32);
    else
        hipLaunchKernelGGL((layernorm_bwd_kernel<float, float, float, float>), dim3(nb2), dim3(256), 0, st, dy32, x, stats, gamma, add32,
                           (const float*)nullptr, dx, partial, M, D, rpb, (const float*)nullptr);
    EGV_LAUNCH_CHECK();
    if (dbeta == dgamma + D) {
        hipLaunchKernelGGL(colsum_partials_kernel, dim3((2 * D + 63) / 64), dim3(1024), 0, st, (const float*)partial, dgamma, nb2, 2 * D, 2 * D, 1.0f,
                           (const float*)nullptr);
    } else {
        hipLaunchKernelGGL(colsum_partials_kernel, dim3((D + 63) / 64), dim3(1024), 0, st, (const float*)partial, dgamma, nb2, D, 2 * D, 1.0f, (const float*)nullptr);
        hipLaunchKernelGGL(colsum_partials_kernel, dim3((D + 63) / 64), dim3(1024), 0, st, (const float*)partial + D, dbeta, nb2, D, 2 * D, 1.0f, (const float*)nullptr);
    }
    EGV_LAUNCH_CHECK();
    return 0;
}

static inline int colsum_chunks(int M) {
    int c = (M + 255) / 256;
    return c > 32 ? 32 : (c < 1 ? 1 : c);
}
extern "C" long long egv_colsum_workspace_bytes(int M, int N) { return (long long)colsum_chunks(M) * N * 4; }

// out[n] (fp32) = scale * gate * sum_m X[m,n]
extern "C" int egv_colsum(int dtype, const void* X, int M, int N, int ld, float* out, float scale, const float* gate,
                          void* workspace, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int ch = colsum_chunks(M);
    const int rpb = (M + ch - 1) / ch;
    const int ch2 = (M + rpb - 1) / rpb;
    dim3 grid((N + 255) / 256, ch2);
    if (dtype == EGV_BF16)
        hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)X, (float*)workspace, M, N, ld, rpb);
    else
        hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, st, (const float*)X, (float*)workspace, M, N, ld, rpb);
    EGV_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_partials_kernel, dim3((N + 63) / 64), dim3(1024), 0, st, (const float*)workspace, out, ch2, N, N, scale, gate);
    EGV_LAUNCH_CHECK();
    return 0;
}

// out[0] (fp32) = scale * sum_i a[i] * b[i];  workspace >= 1024 floats
extern "C" int egv_dot(int dtype, const void* a, const void* b, long long n, float* out, float scale, void* workspace,
                       void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    long long nb = (n / 4 + 255) / 256;
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    if (dtype == EGV_BF16)
        hipLaunchKernelGGL(dot_partial_kernel<bf16_t>, dim3((int)nb), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (float*)workspace, n);
    else
        hipLaunchKernelGGL(dot_partial_kernel<float>, dim3((int)nb), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)workspace, n);
    EGV_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, out, (int)nb, scale);
    EGV_LAUNCH_CHECK();
    return 0;
}

// dst (dtype_dst) = cast(src (dtype_src)); pointers 16-byte aligned
extern "C" int egv_cast(int dtype_src, int dtype_dst, const void* src, void* dst, long long n, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    long long nb = (n / 4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    if (dtype_src == EGV_F32 && dtype_dst == EGV_BF16)
        hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3((int)nb), dim3(256), 0, st, (const float*)src, (bf16_t*)dst, n);
    else if (dtype_src == EGV_BF16 && dtype_dst == EGV_F32)
        hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3((int)nb), dim3(256), 0, st, (const bf16_t*)src, (float*)dst, n);
    else if (dtype_src == EGV_F32 && dtype_dst == EGV_F32)
        hipLaunchKernelGGL((cast_kernel<float, float>), dim3((int)nb), dim3(256), 0, st, (const float*)src, (float*)dst, n);
    else
        hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3((int)nb), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, n);
    EGV_LAUNCH_CHECK();
    return 0;
}
