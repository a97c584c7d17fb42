// MFMA GEMM for gfx950 with fused epilogues -- the dominant kernel of the EgoVLPv2 hot path
// (~97 % of the FLOPs).
//
//   C[M,N] = epilogue( sum_k Aop[m,k] * Bop[n,k] )
//
// One template covers the three GEMM forms of a Linear layer y = x W^T + b (W stored [N_out, K_in]):
//   forward   y  = x  W^T : A = x  [M,K]  (AT=0),  B = W  [N,K]       (BT=0)
//   dgrad     dx = dy W   : A = dy [M,N'] (AT=0),  B = W  [N',K'] read as [red, out] (BT=1)
//   wgrad     dW = dy^T x : A = dy [M',N] read as [red, out] (AT=1), B = x [M',K] (BT=1)
// AT/BT = 1 means the operand is stored [reduction, rows] (rows contiguous) and is transposed on the
// way into LDS by an in-register VEC x VEC block transpose, so the MFMA main loop is identical for
// all three forms and no transposed copies of weights or activations ever touch HBM.
//
// Tiling (v1): 128x128 output tile per 256-thread workgroup (4 waves, 2x2, 64x64 per wave = 4x4
// MFMA 16x16 tiles), K step = 128 bytes per row (64 bf16 / 32 f32), register-prefetched global
// loads -> padded LDS (144 B pitch: conflict-free ds_read_b128 fragments) -> MFMA
// (v_mfma_f32_16x16x32_bf16 / exact-fp32 v_mfma_f32_16x16x4_f32), fp32 accumulation.
// Operands are swapped in the MFMA (D = W_frag * X_frag^T) so that every lane owns 4 consecutive
// output columns of one row: bias / residual / aux loads and the store are 8-16 B vectors.
#include "egv_gemm.h"
#include <cstdlib>

namespace egv {

constexpr int BM = 128, BN = 128;
constexpr int PITCH = 144;          // bytes per LDS row: 128 B of K + 16 B pad
constexpr int ROWB = 128;           // payload bytes per LDS row

template <typename T> struct Tile;
template <> struct Tile<bf16_t> { static constexpr int BK = 64; };
template <> struct Tile<float> { static constexpr int BK = 32; };

// ---- 16-byte chunk load with zero fill (row-contiguous run of VEC elements starting at p) ----
template <typename T>
__device__ __forceinline__ u32x4_t load_chunk(const T* p, int n_valid, bool vec_ok) {
    constexpr int VEC = Elem<T>::VEC;
    u32x4_t r = {0u, 0u, 0u, 0u};
    if (n_valid >= VEC && vec_ok) {
        r = *reinterpret_cast<const u32x4_t*>(p);
    } else if (n_valid > 0) {
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < n_valid) r[e] = reinterpret_cast<const unsigned int*>(p)[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (e < n_valid) r[e >> 1] |= ((unsigned int)reinterpret_cast<const unsigned short*>(p)[e]) << ((e & 1) * 16);
        }
    }
    return r;
}

// Per-thread staging registers for one operand tile (128 rows x 128 B).
// Direct: 1024 chunks / 256 threads = 4 chunks.  Transposed: units of VEC chunks (VEC x VEC blocks).
template <typename T, int TR>
struct Stager {
    static constexpr int VEC = Elem<T>::VEC;
    static constexpr int BK = Tile<T>::BK;
    static constexpr int NCH = TR ? VEC : 4;
    u32x4_t r[NCH];

    // op: base pointer; rows: extent of the (non-reduction) row dimension; ld: leading dim;
    // row0: first tile row; k0: first reduction index of this K step; kend: end of the reduction range.
    __device__ __forceinline__ void load(const T* op, int rows, int ld, int row0, int k0, int kend, bool vec_ok, int tid) {
        if constexpr (!TR) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = tid + i * 256;
                const int row = c >> 3, cc = c & 7;
                const int grow = row0 + row, gk = k0 + cc * VEC;
                int nv = kend - gk;
                if (grow >= rows) nv = 0;
                r[i] = load_chunk<T>(op + (size_t)grow * ld + gk, nv, vec_ok);
            }
        } else {
            constexpr int RB = 128 / VEC;          // row blocks per tile
            constexpr int UNITS = 8 * RB;          // 8 k-blocks of VEC
            if (tid < UNITS) {
                const int kb = tid / RB, rb = tid % RB;
                const int grow0 = row0 + rb * VEC;
                const int nv_rows = rows - grow0;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const int gk = k0 + kb * VEC + j;
                    const int nv = (gk < kend) ? nv_rows : 0;
                    r[j] = load_chunk<T>(op + (size_t)gk * ld + grow0, nv, vec_ok);
                }
            }
        }
    }

    __device__ __forceinline__ void store(unsigned char* s, int tid) const {
        if constexpr (!TR) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = tid + i * 256;
                const int row = c >> 3, cc = c & 7;
                *reinterpret_cast<u32x4_t*>(s + row * PITCH + cc * 16) = r[i];
            }
        } else {
            constexpr int RB = 128 / VEC;
            constexpr int UNITS = 8 * RB;
            if (tid < UNITS) {
                const int kb = tid / RB, rb = tid % RB;
                if constexpr (sizeof(T) == 4) {
                    // r[j][e] = X[k = kb*4+j][row = rb*4+e]  ->  out[e][j]
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        u32x4_t o = {r[0][e], r[1][e], r[2][e], r[3][e]};
                        *reinterpret_cast<u32x4_t*>(s + (rb * 4 + e) * PITCH + kb * 16) = o;
                    }
                } else {
                    // r[j] holds rows rb*8 .. +7 (two per dword) at k = kb*8 + j.
                    // out[row e].dword[d] = (lo: r[2d] elem e, hi: r[2d+1] elem e)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        u32x4_t o;
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const unsigned int a = r[2 * d][e >> 1], b = r[2 * d + 1][e >> 1];
                            o[d] = (e & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
                        }
                        *reinterpret_cast<u32x4_t*>(s + (rb * 8 + e) * PITCH + kb * 16) = o;
                    }
                }
            }
        }
    }
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    // one 128-byte LDS row segment = 64 bf16 = 2 MFMA k-steps of 32
    static __device__ __forceinline__ void run(const unsigned char* sA, const unsigned char* sB, int wm, int wn, int lane,
                                               f32x4_t (&acc)[4][4]) {
        const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = *reinterpret_cast<const bf16x8_t*>(sA + (wm * 64 + i * 16 + fr) * PITCH + ks * 64 + fg * 16);
                b[i] = *reinterpret_cast<const bf16x8_t*>(sB + (wn * 64 + i * 16 + fr) * PITCH + ks * 64 + fg * 16);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);
        }
    }
};
template <> struct Mma<float> {
    // 128-byte row = 32 f32 = 2 groups of 16 k; lane group g reads k = ks*16 + g*4 .. +3 (one b128) and
    // feeds element e to MFMA step e (k order inside the reduction is free as long as A and B agree).
    static __device__ __forceinline__ void run(const unsigned char* sA, const unsigned char* sB, int wm, int wn, int lane,
                                               f32x4_t (&acc)[4][4]) {
        const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f32x4_t a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = *reinterpret_cast<const f32x4_t*>(sA + (wm * 64 + i * 16 + fr) * PITCH + ks * 64 + fg * 16);
                b[i] = *reinterpret_cast<const f32x4_t*>(sB + (wn * 64 + i * 16 + fr) * PITCH + ks * 64 + fg * 16);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[ni][e], a[mi][e], acc[mi][ni], 0, 0, 0);
        }
    }
};

template <typename T, int AT, int BT, typename OutT>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs g) {
    constexpr int BK = Tile<T>::BK;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 128 * PITCH];
    unsigned char* sA = smem;
    unsigned char* sB = smem + 128 * PITCH;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = wave_id();
    const int wm = wid >> 1, wn = wid & 1;

    const int ntile = g.tiles_m * g.tiles_n;
    const int t = xcd_remap(blockIdx.x, ntile);
    const int tm = t / g.tiles_n, tn = t % g.tiles_n;     // N tiles fastest: neighbours share the A panel in L2
    const int m0 = tm * BM, n0 = tn * BN;

    const int kbeg = blockIdx.z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);

    const T* A = reinterpret_cast<const T*>(g.A);
    const T* B = reinterpret_cast<const T*>(g.B);

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    Stager<T, AT> stA;
    Stager<T, BT> stB;
    // when a transposed bf16 operand only has 128 units, give B's units to threads 128..255
    const int tidB = (BT && AT) ? ((tid + 128) & 255) : tid;

    stA.load(A, g.M, g.lda, m0, kbeg, kend, g.a_vec_ok != 0, tid);
    stB.load(B, g.N, g.ldb, n0, kbeg, kend, g.b_vec_ok != 0, tidB);

    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        stA.store(sA, tid);
        stB.store(sB, tidB);
        __syncthreads();
        if (k0 + BK < kend) {
            stA.load(A, g.M, g.lda, m0, k0 + BK, kend, g.a_vec_ok != 0, tid);
            stB.load(B, g.N, g.ldb, n0, k0 + BK, kend, g.b_vec_ok != 0, tidB);
        }
        Mma<T>::run(sA, sB, wm, wn, lane, acc);
        __syncthreads();
    }

    // ---------------- epilogue ----------------
    OutT* C = reinterpret_cast<OutT*>(g.C) + (size_t)blockIdx.z * g.slab_stride;
    const float gate = g.e.gate ? *g.e.gate : 1.0f;
    const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 64 + mi * 16 + fr;
        if (m >= g.M) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + ni * 16 + fg * 4;
            if (n >= g.N) continue;
            float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
            gemm_epilogue4<T, OutT>(g, C, m, n, v, gate);
        }
    }
}

// sum split-K slabs: out[i] = scale * gate * sum_z slab[z][i]   (fp32, deterministic order)
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ out, long long n, int nz,
                                    long long stride, float scale, const float* gate, const float* __restrict__ slabs2,
                                    float* __restrict__ out2, long long n2) {
    const float gsc = scale * (gate ? *gate : 1.0f);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < nz; ++z) s += slabs[(size_t)z * stride + i];
        out[i] = s * gsc;
    }
    // optional second, small reduction (bias-gradient partials [nz][n2]) in the same launch
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < nz; ++z) s += slabs2[(size_t)z * n2 + i];
        out2[i] = s * gsc;
    }
}

template <typename T, int AT, int BT, typename OutT>
static int launch_gemm(const GemmArgs& g, int nz, hipStream_t st) {
    dim3 grid(g.tiles_m * g.tiles_n, 1, nz);
    hipLaunchKernelGGL((gemm_kernel<T, AT, BT, OutT>), grid, dim3(256), 0, st, g);
    return 0;
}

}  // namespace egv

using namespace egv;

void* egv_prof_begin(void* stream);
void egv_prof_end(void* handle, void* stream, double flops, int kind, double bytes);

static inline int vec_ok(const void* p, int ld, int dtype) {
    const int vec = dtype == EGV_BF16 ? 8 : 4;
    return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && (ld % vec == 0);
}

// See include/egovlp_hip.h for the contract.
// ------------------------------------------------------------------------------------------------
// Skinny bf16 Linear, M <= 16 rows (the projection heads over B pooled rows: [8, 4096] x [4096, 4096]^T, model.py:105-115): as a tiled GEMM
// this is N / 128 workgroups that each walk K alone (86 us for a 33.5 MB weight); here one wave owns FOUR output columns, streams their
// weight rows once (16 bytes per lane and row, 512 k per step) against the <= 16 activation rows (cache-resident) and reduces across
// the lanes at the end: N / 16 workgroups, bound by one pass over the weight.  A row's sum does not depend on the other rows
// (batch-independent bit for bit, like the tiled kernels).  y = act(x W^T + b), act in {none, relu, tanh}.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dot8_bf16(const u32x4_t& a, const u32x4_t& b, float acc) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        acc = fmaf(__uint_as_float(a[k] << 16), __uint_as_float(b[k] << 16), acc);
        acc = fmaf(__uint_as_float(a[k] & 0xffff0000u), __uint_as_float(b[k] & 0xffff0000u), acc);
    }
    return acc;
}
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb,
                                                          bf16_t* __restrict__ C, int ldc, const float* __restrict__ bias, int act,
                                                          int M, int N, int K, bf16_t* __restrict__ pre) {
    const int lane = threadIdx.x & 63;
    const int n0 = (blockIdx.x * 4 + wave_id()) * 4;
    if (n0 >= N) return;
    float acc[16][4];
#pragma unroll
    for (int m = 0; m < 16; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[m][c] = 0.f;
    for (int k0 = lane * 8; k0 < K; k0 += 512) {
        u32x4_t w[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            w[c] = n0 + c < N ? *reinterpret_cast<const u32x4_t*>(B + (long long)(n0 + c) * ldb + k0) : u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (m < M) {
                // agent-scope loads of the <= 16 activation rows every wave of the launch reads (see wgrad_small_m_kernel: plain loads of a
                // small, freshly recycled buffer from every CU were seen to return a stale line)
                const unsigned long long* ap = reinterpret_cast<const unsigned long long*>(A + (long long)m * lda + k0);
                const unsigned long long x01 = __hip_atomic_load(ap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long x23 = __hip_atomic_load(ap + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const u32x4_t x = {(unsigned int)x01, (unsigned int)(x01 >> 32), (unsigned int)x23, (unsigned int)(x23 >> 32)};
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[m][c] = dot8_bf16(x, w[c], acc[m][c]);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        if (m < M) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float v = wave_sum(acc[m][c]);
                if (lane == 0 && n0 + c < N) {
                    const float z = v + (bias ? bias[n0 + c] : 0.f);
                    if (pre) pre[(long long)m * ldc + n0 + c].v = f2bf(z);       // saved pre-activation (GELU: the MLP's fc1 on the CLS rows)
                    C[(long long)m * ldc + n0 + c].v = f2bf(act == 1 ? gelu_fast_f(z) : apply_act(z, act));   // GELU: the form of every bf16 kernel
                }
            }
        }
    }
}

extern "C" int egv_gemm(int dtype, int a_trans, int b_trans, int M, int N, int K,
                        const void* A, int lda, const void* B, int ldb, void* C, int ldc, int out_f32,
                        const float* bias, int act, const float* gate, const void* res1, const void* res2,
                        void* pre, const void* aux, int dact, int ldr, float scale, void* stream) {
    EGV_CHECK(dtype == EGV_F32 || dtype == EGV_BF16, "egv_gemm: bad dtype %d", dtype);
    EGV_CHECK(M > 0 && N > 0 && K > 0, "egv_gemm: bad shape %d %d %d", M, N, K);
    EGV_CHECK(!(a_trans && !b_trans), "egv_gemm: (a_trans=1,b_trans=0) is not a form this path uses");
    GemmArgs g;
    g.A = A; g.B = B; g.C = C;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.a_vec_ok = vec_ok(A, lda, dtype);
    g.b_vec_ok = vec_ok(B, ldb, dtype);
    g.c_vec_ok = out_f32 ? vec_ok(C, ldc, EGV_F32) : (((reinterpret_cast<uintptr_t>(C) & 15) == 0) && (ldc % 4 == 0));
    g.k_per_split = K;
    g.slab_stride = 0;
    g.colsum = nullptr;
    g.tiles_m = (M + BM - 1) / BM;
    g.tiles_n = (N + BN - 1) / BN;
    g.e.bias = bias; g.e.gate = gate; g.e.res1 = res1; g.e.res2 = res2; g.e.pre = pre; g.e.aux = aux;
    g.e.act = act; g.e.dact = dact; g.e.ldr = ldr ? ldr : ldc; g.e.scale = scale;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const double es_ = dtype == EGV_BF16 ? 2.0 : 4.0;
    // algorithmic bytes: operands + output once, plus every epilogue operand that is read or written
    const double abytes = es_ * ((double)M * K + (double)N * K + (double)M * N * (1 + (res1 != nullptr) + (res2 != nullptr) + (pre != nullptr) + (aux != nullptr)));
    void* ph = egv_prof_begin(stream);
    if (dtype == EGV_BF16 && !a_trans && !b_trans && !out_f32 && M <= 16 && N >= 512 && K >= 512 && (K % 8) == 0 && g.a_vec_ok && g.b_vec_ok &&
        !gate && !res1 && !res2 && !aux && !dact && scale == 1.0f && (act == 0 || act == 1 || act == 2 || act == 3) && (!pre || (act == 1 && g.e.ldr == ldc))) {
        hipLaunchKernelGGL(gemm_skinny_kernel, dim3((N + 15) / 16), dim3(256), 0, st, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, (bf16_t*)C, ldc,
                           bias, act, M, N, K, (bf16_t*)pre);
        egv_prof_end(ph, stream, 2.0 * M * N * K, 7, abytes);        // 7 = skinny (<= 16 rows)
        EGV_LAUNCH_CHECK();
        return 0;
    }
    const int took = dtype == EGV_BF16 ? egv_gemm2_launch(g, a_trans, b_trans, out_f32, 1, st) : 0;
    if (took) {
        egv_prof_end(ph, stream, 2.0 * M * N * K, took == 2 ? 12 : (took == 3 ? 13 : 8), abytes);   // 8 ring 256x128, 12 persistent ping-pong, 13 ring 128x128
        EGV_LAUNCH_CHECK();
        return 0;
    }
#define EGV_DISPATCH(TT)                                                                   \
    if (!a_trans && !b_trans) {                                                            \
        if (out_f32) launch_gemm<TT, 0, 0, float>(g, 1, st); else launch_gemm<TT, 0, 0, TT>(g, 1, st); \
    } else if (!a_trans && b_trans) {                                                      \
        if (out_f32) launch_gemm<TT, 0, 1, float>(g, 1, st); else launch_gemm<TT, 0, 1, TT>(g, 1, st); \
    } else {                                                                               \
        if (out_f32) launch_gemm<TT, 1, 1, float>(g, 1, st); else launch_gemm<TT, 1, 1, TT>(g, 1, st); \
    }
    if (dtype == EGV_BF16) { EGV_DISPATCH(bf16_t) } else { EGV_DISPATCH(float) }
#undef EGV_DISPATCH
    egv_prof_end(ph, stream, 2.0 * M * N * K, (dtype == EGV_BF16 ? 0 : 4) + (b_trans ? 1 : 0), abytes);
    EGV_LAUNCH_CHECK();
    return 0;
}

// split count of the wgrad reduction: minimise (MFMA time / wave-quantisation efficiency) + slab write/read time
static inline int wgrad_splits(int tiles, int M, long long out_elems, double flops, int slots) {
    int best = 1;
    double best_cost = 1e30;
    for (int nz = 1; nz <= 64; ++nz) {
        if (nz > 1 && M / nz < 512) break;
        const long long wgs = (long long)tiles * nz;
        const double eff = (double)wgs / (double)(((wgs + slots - 1) / slots) * slots);
        const double t_mma = flops / 450e12 / eff;
        const double t_slab = nz > 1 ? (2.0 * nz * out_elems * 4.0) / 4.0e12 : 0.0;
        const double cost = t_mma + t_slab;
        if (cost < best_cost * 0.97) { best_cost = cost; best = nz; }
    }
    return best;
}

int egv_gemm4_launch(const egv::GemmArgs& g, int nz, hipStream_t st);
// ping-pong weight-gradient kernel (egv_gemm4.hip): 256x256 tiles, one (tile, split) item per CU
static inline bool wgrad_use_pp(int dtype, int M, int N, int K) {
    static const int on = egv_cfg_int("EGV_WGRAD_PP", 1);
    return on && dtype == EGV_BF16 && (N % 256) == 0 && (K % 256) == 0 && M >= 4096 && (N / 256) * (K / 256) <= 128;
}
static inline int wgrad_pp_splits(int M, int N, int K) {
    const int tiles = (N / 256) * (K / 256);
    static const int items = egv_cfg_int("EGV_WGRAD_ITEMS", 224);   // 7/8 of the CUs: the rest serve the other streams (measured 92.6 -> 91.8 ms per step)
    int nz = items / tiles;
    while (nz > 1 && M / nz < 512) --nz;
    return nz < 1 ? 1 : nz;
}
static inline bool wgrad_use_gemm2(int dtype, int M, int N, int K) { return dtype == EGV_BF16 && N >= 128 && K >= 64 && M >= 256; }

static inline int wgrad_plan(int dtype, int M, int N, int K, bool& v2) {
    v2 = wgrad_use_gemm2(dtype, M, N, K);
    if (v2) {
        const int tb = ((N + 255) / 256) * ((K + 127) / 128);
        return wgrad_splits(tb, M, (long long)N * K, 2.0 * M * N * K, 512);
    }
    return wgrad_splits(((N + BM - 1) / BM) * ((K + BN - 1) / BN), M, (long long)N * K, 2.0 * M * N * K, 512);
}

extern "C" long long egv_gemm_wgrad_workspace_bytes(int N, int K, int M) {
    // fp32 slabs [nz][N,K] + [nz][N] column-sum partials + the colsum fallback workspace
    bool v2;
    const int nz = wgrad_plan(EGV_BF16, M, N, K, v2);
    bool v2f;
    const int nzf = wgrad_plan(EGV_F32, M, N, K, v2f);
    long long z = nz > nzf ? nz : nzf;
    if (wgrad_use_pp(EGV_BF16, M, N, K) && wgrad_pp_splits(M, N, K) > z) z = wgrad_pp_splits(M, N, K);
    return z * N * K * 4 + z * N * 4 + (long long)32 * N * 4 + 4096;
}

extern "C" int egv_colsum(int dtype, const void* X, int M, int N, int ld, float* out, float scale, const float* gate,
                          void* workspace, void* stream);

// Weight (and bias) gradient of a Linear over a handful of rows (M <= 16: the B CLS rows of the last block of a video pass, the pooled
// rows of the heads): dW[n, k] = sum_m dY[m, n] X[m, k] is M multiply-adds per output element and N*K*4 bytes of stores -- an outer
// product, bound by writing dW once.  One thread per (n, four consecutive k): the M x 4 values of X and the M values of dY come from the
// caches (both operands are a few KB), the sum runs over m in order (deterministic), dbias[n] = sum_m dY[m, n] rides with the k = 0 thread.
// As a 128 x 128-tile MFMA GEMM the same gradient was 144 tiles whose K loop has one step: 20 launches of 39 us (up to 250 us in the step) per
// training step, every one of them on the latency-bound chain between two block calls.
#ifndef SMALLM_DIAG
#define SMALLM_DIAG 0
#endif
namespace egv {
__global__ __launch_bounds__(256) void wgrad_small_m_kernel(const bf16_t* __restrict__ dY, int ldy, const bf16_t* __restrict__ X, int ldx,
                                                            float* __restrict__ dW, float* __restrict__ dbias, int M, int N, int K, float scale,
                                                            const float* __restrict__ gate) {
    const int k4n = K >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)N * k4n) return;
    const int n = (int)(idx / k4n), k = (int)(idx % k4n) * 4;
    const float sc = scale * (gate ? *gate : 1.0f);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, sb = 0.f;
    // diagnostic builds (tools/variant_build.sh): which cache holds the stale line?  3 = plain loads behind an invalidate of the CU's vector L1,
    // 4 = behind an agent-scope invalidate (vector L1 + the non-coherent lines of this XCD's L2), 5 = plain loads after a delay
#if SMALLM_DIAG == 3
    asm volatile("buffer_inv sc0\n\ts_waitcnt vmcnt(0)" ::: "memory");
#elif SMALLM_DIAG == 4
    asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
#elif SMALLM_DIAG == 5   // plain loads ~25 us after the wave's start: does a producer that is still finishing explain the stale line?
    for (int i = 0; i < 6; ++i) __builtin_amdgcn_s_sleep(127);
    asm volatile("" ::: "memory");
#endif
    for (int m = 0; m < M; ++m) {
#if SMALLM_DIAG >= 2    // plain cached loads (the form that read stale lines)
        const float d = bf2f(dY[(size_t)m * ldy + n].v);
        const u32x2_t xv = *reinterpret_cast<const u32x2_t*>(X + (size_t)m * ldx + k);
#else
        // Agent-scope loads (they do not hit in the CU's vector L1).  With plain loads one wave in ~10^5 of this launch -- 16 384 short
        // workgroups that all read the same 128 KB, spread over every CU while persistent kernels of the other streams hold the CUs --
        // got ONE stale 128-byte line of x (the address's previous content within the step): 10-30 wrong elements in one row of a
        // 16.7 M-element gradient in one run out of three, inputs identical in memory before and after (tools/repro_check.py;
        // profiles/round5_experiments.md section 11).  The operands are 128 KB: the L1 is worth nothing here.
        const float d = bf2f(__hip_atomic_load(reinterpret_cast<const unsigned short*>(dY) + (size_t)m * ldy + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const unsigned long long xq = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(X + (size_t)m * ldx + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u32x2_t xv = {(unsigned int)xq, (unsigned int)(xq >> 32)};
#endif
        a0 = fmaf(d, __uint_as_float(xv[0] << 16), a0);
        a1 = fmaf(d, __uint_as_float(xv[0] & 0xffff0000u), a1);
        a2 = fmaf(d, __uint_as_float(xv[1] << 16), a2);
        a3 = fmaf(d, __uint_as_float(xv[1] & 0xffff0000u), a3);
        sb += d;
    }
    *reinterpret_cast<f32x4_t*>(dW + (size_t)n * K + k) = f32x4_t{a0 * sc, a1 * sc, a2 * sc, a3 * sc};
    if (dbias && k == 0) dbias[n] = sb * sc;
}
}  // namespace egv

// dW[N,K] (fp32) = scale * gate * dY[M,N]^T X[M,K], reduction over M split across blockIdx.z;
// dbias[N] (fp32, optional) = scale * gate * sum_m dY[m, :]  (fused: the wgrad kernel already streams dY).
extern "C" int egv_gemm_wgrad(int dtype, int M, int N, int K, const void* dY, int ldy, const void* X, int ldx,
                              float* dW, float* dbias, float scale, const float* gate, void* workspace, long long workspace_bytes,
                              void* stream) {
    EGV_CHECK(dtype == EGV_F32 || dtype == EGV_BF16, "egv_gemm_wgrad: bad dtype %d", dtype);
    EGV_CHECK(M > 0 && N > 0 && K > 0, "egv_gemm_wgrad: bad shape");
    EGV_CHECK(workspace && workspace_bytes >= egv_gemm_wgrad_workspace_bytes(N, K, M), "egv_gemm_wgrad: workspace too small");
    static const bool small_m = egv_cfg_on("EGV_WGRAD_SMALL_M", true);
    if (small_m && dtype == EGV_BF16 && M <= 16 && (K % 4) == 0 && (ldx % 4) == 0 && (reinterpret_cast<uintptr_t>(X) & 7) == 0 && (reinterpret_cast<uintptr_t>(dW) & 15) == 0) {
        void* ph0 = egv_prof_begin(stream);
        const long long nthr = (long long)N * (K >> 2);
        hipLaunchKernelGGL(wgrad_small_m_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           (const bf16_t*)dY, ldy, (const bf16_t*)X, ldx, dW, dbias, M, N, K, scale, gate);
        egv_prof_end(ph0, stream, 2.0 * M * N * K, 9, 2.0 * ((double)M * N + (double)M * K) + 4.0 * N * K);   // 9 = small-M outer product
        EGV_LAUNCH_CHECK();
        return 0;
    }
    bool v2;
    int nz = wgrad_plan(dtype, M, N, K, v2);
    const bool pp = wgrad_use_pp(dtype, M, N, K);
    if (pp) nz = wgrad_pp_splits(M, N, K);
    const int bk = dtype == EGV_BF16 ? 64 : 32;
    int kper = (M + nz - 1) / nz;
    kper = ((kper + bk - 1) / bk) * bk;
    nz = (M + kper - 1) / kper;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* slabs = (float*)workspace;
    float* bias_part = slabs + (size_t)nz * N * K;          // [nz][N]
    void* cs_ws = bias_part + (size_t)nz * N;
    GemmArgs g;
    g.A = dY; g.B = X;
    g.M = N; g.N = K; g.K = M;           // output rows = N (of dY), cols = K (of X), reduction = M
    g.lda = ldy; g.ldb = ldx; g.ldc = K;
    g.a_vec_ok = vec_ok(dY, ldy, dtype);
    g.b_vec_ok = vec_ok(X, ldx, dtype);
    g.k_per_split = kper;
    g.tiles_m = (N + BM - 1) / BM;
    g.tiles_n = (K + BN - 1) / BN;
    g.colsum = nullptr;
    g.e = GemmEpi{};
    g.e.ldr = K;
    if (nz == 1) {
        g.C = dW; g.slab_stride = 0;
        g.c_vec_ok = vec_ok(dW, K, EGV_F32);
        g.e.scale = scale; g.e.gate = gate;
    } else {
        g.C = slabs; g.slab_stride = (long long)N * K;
        g.c_vec_ok = vec_ok(slabs, K, EGV_F32) && (((long long)N * K) % 4 == 0);
        g.e.scale = 1.0f;
    }
    void* ph = egv_prof_begin(stream);
    bool bias_fused = false;
    if (v2) {
        // one split, no scaling: the kernel's column sums ARE the bias gradient (no reduction / copy launch)
        const bool bias_direct = dbias && nz == 1 && scale == 1.0f && gate == nullptr;
        if (dbias) g.colsum = bias_direct ? dbias : bias_part;
        const bool took_pp = pp && egv_gemm4_launch(g, nz, st);
        if (took_pp || egv_gemm2_launch(g, 1, 1, 1, nz, st)) {
            egv_prof_end(ph, stream, 2.0 * M * N * K, took_pp ? 14 : 10,       // 14 ping-pong 256x256, 10 ring 256x128
                         2.0 * ((double)M * N + (double)M * K) + 4.0 * N * K);
            bias_fused = dbias != nullptr;
            if (bias_direct) dbias = nullptr;          // done
        } else {
            v2 = false;
            g.colsum = nullptr;
        }
    }
    if (!v2) {
        if (dtype == EGV_BF16) launch_gemm<bf16_t, 1, 1, float>(g, nz, st);
        else launch_gemm<float, 1, 1, float>(g, nz, st);
        egv_prof_end(ph, stream, 2.0 * M * N * K, (dtype == EGV_BF16 ? 0 : 4) + 2, (dtype == EGV_BF16 ? 2.0 : 4.0) * ((double)M * N + (double)M * K) + 4.0 * N * K);
    }
    EGV_LAUNCH_CHECK();
    const bool fuse_bias_reduce = dbias && bias_fused && nz > 1;
    if (nz > 1) {
        const long long n = (long long)N * K;
        int blocks = (int)((n + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, st, (const float*)slabs, dW, n, nz,
                           (long long)N * K, scale, gate, fuse_bias_reduce ? (const float*)bias_part : (const float*)nullptr,
                           fuse_bias_reduce ? dbias : (float*)nullptr, fuse_bias_reduce ? (long long)N : 0LL);
        EGV_LAUNCH_CHECK();
    }
    if (dbias && !fuse_bias_reduce) {
        if (bias_fused) {
            const int blocks = (N + 255) / 256;
            hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, st, (const float*)bias_part, dbias, (long long)N, nz,
                               (long long)N, scale, gate, (const float*)nullptr, (float*)nullptr, 0LL);
            EGV_LAUNCH_CHECK();
        } else {
            return egv_colsum(dtype, dY, M, N, ldy, dbias, scale, gate, cs_ws, stream);
        }
    }
    return 0;
}
