// Shared declarations of the GEMM kernels (generic 128x128 path: egv_gemm.hip; 256-row glds path: egv_gemm2.hip).
#pragma once
#include "egv_common.h"

namespace egv {

struct GemmEpi {
    const float* bias;   // [N] fp32 or null
    const float* gate;   // device scalar (alpha gate) or null
    const void* res1;    // [M,N] T or null
    const void* res2;    // [M,N] T or null
    void* pre;           // [M,N] T: save (acc + bias) before activation, or null
    const void* aux;     // [M,N] T: operand of the activation derivative (backward), or null
    int act;             // 0 none, 1 gelu(erf), 2 relu, 3 tanh
    int dact;            // 0 none, 1 gelu'(aux = pre-activation), 2 relu'(aux = output), 3 tanh'(aux = output)
    int ldr;             // leading dim of res1/res2/pre/aux
    float scale;         // host scalar applied to the accumulator first
    unsigned char* oq = nullptr;   // MX-fp8 form only: the output also as e4m3 codes [M][N] ...
    unsigned char* os = nullptr;   // ... and role-0 scale bytes (egv_mx.hip)
};

struct GemmArgs {
    const void* A;
    const void* B;
    void* C;
    int M, N, K;
    int lda, ldb, ldc;
    int a_vec_ok, b_vec_ok, c_vec_ok;   // 16-byte vector path allowed (pointer + leading dim aligned)
    int k_per_split;                    // K range per blockIdx.z (== K when not split)
    long long slab_stride;              // elements between split slabs of C
    int tiles_m, tiles_n;
    float* colsum;                      // wgrad only: fp32 [gridDim.z][M] partial column sums of the A operand (dY), or null
    GemmEpi e;
    // MX-fp8 form of the persistent kernel (egv_gemm3.hip, egv_mx.hip): A / B hold e4m3 codes (lda / ldb in BYTES = elements), the
    // E8M0 block scales come in the lane order of egv_mx.hip (role 0 for A, role 1 for B)
    const unsigned char* sa = nullptr;
    const unsigned char* sb = nullptr;
    int mx = 0;
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 1 || act == 4) return gelu_f(v);
    if (act == 2) return fmaxf(v, 0.0f);
    if (act == 3) return tanhf(v);
    return v;
}
__device__ __forceinline__ float apply_dact(float aux, int dact) {
    if (dact == 1) return dgelu_f(aux);
    if (dact == 2) return aux > 0.0f ? 1.0f : 0.0f;
    if (dact == 3) return 1.0f - aux * aux;
    if (dact == 4) return aux;                   // EGV_ACT_GELU_D: aux already holds gelu'(x)
    return 1.0f;
}


// Epilogue for one lane-owned group of 4 consecutive output columns (m, n .. n+3); v = raw accumulators.
template <typename T, typename OutT>
__device__ __forceinline__ void gemm_epilogue4(const GemmArgs& g, OutT* C, int m, int n, float (&v)[4], float gate) {
    const GemmEpi& e = g.e;
    const T* R1 = reinterpret_cast<const T*>(e.res1);
    const T* R2 = reinterpret_cast<const T*>(e.res2);
    const T* AUX = reinterpret_cast<const T*>(e.aux);
    T* PRE = reinterpret_cast<T*>(e.pre);
    const bool rvec = (e.ldr & 3) == 0;
    const bool full = (n + 3 < g.N);
    const size_t ro = (size_t)m * e.ldr + n;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= e.scale;
    if (e.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (full || n + r < g.N) v[r] += e.bias[n + r];
    }
    if (PRE) {
        float pv[4] = {v[0], v[1], v[2], v[3]};
        if (e.act == 4) {                        // EGV_ACT_GELU_D: the saved tensor is gelu'(x)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (sizeof(T) == 2) { float gg; gelu_pair_fast_f(v[r], gg, pv[r]); }
                else pv[r] = dgelu_f(v[r]);
            }
        }
        if (full && rvec) st4(PRE + ro, pv);
        else
            for (int r = 0; r < 4; ++r)
                if (n + r < g.N) Elem<T>::st(PRE + ro + r, pv[r]);
    }
    if (e.act) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (sizeof(T) == 2 && (e.act == 1 || e.act == 4)) ? gelu_fast_f(v[r]) : apply_act(v[r], e.act);   // bf16 mode: the same GELU form in every kernel
    }
    if (e.gate) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] *= gate; asm volatile("" : "+v"(v[r])); }   // product rounded on its own in every GEMM kernel: never contracted with the residual add
    }
    if (R1) {
        float x[4] = {0.f, 0.f, 0.f, 0.f};
        if (full && rvec) ld4(R1 + ro, x);
        else
            for (int r = 0; r < 4; ++r)
                if (n + r < g.N) x[r] = Elem<T>::ld(R1 + ro + r);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += x[r];
    }
    if (R2) {
        float x[4] = {0.f, 0.f, 0.f, 0.f};
        if (full && rvec) ld4(R2 + ro, x);
        else
            for (int r = 0; r < 4; ++r)
                if (n + r < g.N) x[r] = Elem<T>::ld(R2 + ro + r);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += x[r];
    }
    if (e.dact) {
        float x[4] = {0.f, 0.f, 0.f, 0.f};
        if (full && rvec) ld4(AUX + ro, x);
        else
            for (int r = 0; r < 4; ++r)
                if (n + r < g.N) x[r] = Elem<T>::ld(AUX + ro + r);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= (sizeof(T) == 2 && e.dact == 1) ? dgelu_fast_f(x[r]) : apply_dact(x[r], e.dact);
    }
    const size_t co = (size_t)m * g.ldc + n;
    if (full && g.c_vec_ok) st4(C + co, v);
    else
        for (int r = 0; r < 4; ++r)
            if (n + r < g.N) Elem<OutT>::st(C + co + r, v[r]);
}

}  // namespace egv

// egv_gemm2.hip: non-zero if a DMA-staged kernel covered the call (and enqueued it): 1 ring 256x128 / wgrad, 2 persistent ping-pong, 3 ring 128x128
int egv_gemm2_launch(const egv::GemmArgs& g, int a_trans, int b_trans, int out_f32, int nz, hipStream_t st);
