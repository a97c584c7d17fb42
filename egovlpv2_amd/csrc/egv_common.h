// Common device helpers for the gfx950 (MI355X / CDNA4) kernels of the EgoVLPv2 hot path.
// Wavefront = 64 lanes everywhere; no other architecture is supported (no dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define EGV_F32 0
#define EGV_BF16 1

namespace egv {

struct bf16_t { unsigned short v; };

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }
// round-to-nearest-even, NaN preserved (same rounding torch uses for float -> bfloat16)
// (gfx950 has the conversion in hardware: v_cvt_pk_bf16_f32, one instruction per pair)
typedef __attribute__((ext_vector_type(2))) float egv_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 egv_bf16x2_t;
__device__ __forceinline__ unsigned short f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(egv_f32x2_t{lo, hi}, egv_bf16x2_t));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int VEC = 4;   // elements per 16-byte chunk
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int VEC = 8;
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(p->v); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { p->v = f2bf(v); }
};

// 4 consecutive elements <-> 4 floats (8 B for bf16, 16 B for f32); p must be aligned to the vector.
__device__ __forceinline__ void ld4(const float* p, float (&o)[4]) {
    f32x4_t v = *reinterpret_cast<const f32x4_t*>(p);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
}
__device__ __forceinline__ void ld4(const bf16_t* p, float (&o)[4]) {
    u32x2_t v = *reinterpret_cast<const u32x2_t*>(p);
    o[0] = __uint_as_float(v[0] << 16); o[1] = __uint_as_float(v[0] & 0xffff0000u);
    o[2] = __uint_as_float(v[1] << 16); o[3] = __uint_as_float(v[1] & 0xffff0000u);
}
__device__ __forceinline__ void st4(float* p, const float (&o)[4]) {
    f32x4_t v = {o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4_t*>(p) = v;
}
__device__ __forceinline__ void st4(bf16_t* p, const float (&o)[4]) {
    u32x2_t v;
    v[0] = pack_bf16x2(o[0], o[1]);
    v[1] = pack_bf16x2(o[2], o[3]);
    *reinterpret_cast<u32x2_t*>(p) = v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// wave-wide sum with DPP lane exchanges inside the 16-lane rows (no LDS traffic, unlike the bpermute butterfly of wave_sum) and
// four v_readlane for the rows; every lane returns the same value.  All 64 lanes must be active.
template <int CTRL> __device__ __forceinline__ float dpp_mov_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_mov_f<0xB1>(v);       // quad_perm [1,0,3,2]
    v += dpp_mov_f<0x4E>(v);       // quad_perm [2,3,0,1]
    v += dpp_mov_f<0x141>(v);      // row_half_mirror
    v += dpp_mov_f<0x140>(v);      // row_mirror
    return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float dgelu_f(float x) {
    // d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

#ifdef EGV_GELU_OLD
// GELU / GELU' for the bf16 storage mode: Phi(x) from the Abramowitz-Stegun 7.1.26 rational form of erf (absolute error
// < 1.5e-7, far below the 2^-9 rounding of the bf16 value that is stored).  The negative tail is formed without
// cancellation (Phi(x<0) = q, Phi(x>=0) = 1 - q) and the same exp(-x^2/2) serves the density in the derivative; about half
// the VALU work of erff + expf, which matters in the fc1 / fc2-dgrad epilogues (64 values per thread per tile).
__device__ __forceinline__ float phi_tail_q(float x, float& e) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);      // exp(-x^2 / 2)
    return 0.5f * p * t * e;                                            // = 0.5 * erfc(|x| / sqrt 2)
}
__device__ __forceinline__ float gelu_fast_f(float x) {
    float e;
    const float q = phi_tail_q(x, e);
    return x * (x < 0.f ? q : 1.0f - q);
}
__device__ __forceinline__ float dgelu_fast_f(float x) {
    float e;
    const float q = phi_tail_q(x, e);
    return (x < 0.f ? q : 1.0f - q) + x * 0.39894228040143267794f * e;
}

// GELU and its derivative from ONE evaluation of the tail / density (the pair the EGV_ACT_GELU_D epilogues store)
__device__ __forceinline__ void gelu_pair_fast_f(float x, float& g, float& d) {
    float e;
    const float q = phi_tail_q(x, e);
    const float cdf = x < 0.f ? q : 1.0f - q;
    g = x * cdf;
    d = cdf + x * 0.39894228040143267794f * e;
}

#else
// GELU / GELU' for the bf16 storage mode.  Phi(x) = 1 / (1 + 2^(x P(|x|))): the logit of the normal distribution function is x times a
// smooth even function, P is its degree-5 fit in |x| (minimax on the RELATIVE error of x Phi(x) over |x| <= 6 with a floor of 2e-3 on the
// magnitude; -log2(e) folded into the coefficients; tools/fit_gelu.py).  x Phi(x) within 4.0e-5 relative (6.6e-6 absolute), the
// derivative within 1.8e-5 absolute: below 1/25 of the 2^-9 rounding of the bf16 value that is stored.  10 VALU instructions per
// value (5 FMA, 2 transcendentals, no select) against 19 for a rational erfc form: the fc1 / fc2-dgrad epilogues (128 values per
// thread and tile) are bound by VALU issue, not by the MFMA pipe (profiles/round4_experiments.md section 6).  x P(|x|) is monotone,
// +-huge inputs give x or -0, no NaN for finite x.
__device__ __forceinline__ float phi_fast_f(float x) {
    const float a = fabsf(x);
    float p = fmaf(-0.0004328800132498145f, a, 0.005315648391842842f);
    p = fmaf(p, a, -0.014343290589749813f);
    p = fmaf(p, a, -0.08734285086393356f);
    p = fmaf(p, a, -0.00952006783336401f);
    p = fmaf(p, a, -2.300459146499634f);
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * p));
}
__device__ __forceinline__ float gelu_fast_f(float x) { return x * phi_fast_f(x); }
__device__ __forceinline__ float dgelu_fast_f(float x) {
    const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);      // exp(-x^2 / 2)
    return fmaf(x * 0.39894228040143267794f, e, phi_fast_f(x));
}

// GELU and its derivative from ONE evaluation of Phi (the pair the EGV_ACT_GELU_D epilogues store)
__device__ __forceinline__ void gelu_pair_fast_f(float x, float& g, float& d) {
    const float cdf = phi_fast_f(x);
    const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);
    g = x * cdf;
    d = fmaf(x * 0.39894228040143267794f, e, cdf);
}

// The same three functions on TWO values at a time: the polynomial, the products and the sums as packed fp32 operations (v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32: two IEEE operations per lane and issue slot -- the bits of the scalar forms above), the transcendentals
// per value.  5 + 2 issue slots per value instead of 8 + 2: the GELU epilogues of the persistent GEMM are bound by VALU issue.
typedef float gelu_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gelu_f32x2_t phi_fast_f2(gelu_f32x2_t x) {
    const gelu_f32x2_t a = {fabsf(x[0]), fabsf(x[1])};
    auto c = [](float v) { return gelu_f32x2_t{v, v}; };
    gelu_f32x2_t p = __builtin_elementwise_fma(c(-0.0004328800132498145f), a, c(0.005315648391842842f));
    p = __builtin_elementwise_fma(p, a, c(-0.014343290589749813f));
    p = __builtin_elementwise_fma(p, a, c(-0.08734285086393356f));
    p = __builtin_elementwise_fma(p, a, c(-0.00952006783336401f));
    p = __builtin_elementwise_fma(p, a, c(-2.300459146499634f));
    const gelu_f32x2_t z = x * p;
    const gelu_f32x2_t d = c(1.0f) + gelu_f32x2_t{__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])};
    return gelu_f32x2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
__device__ __forceinline__ gelu_f32x2_t gelu_fast_f2(gelu_f32x2_t x) { return x * phi_fast_f2(x); }
__device__ __forceinline__ gelu_f32x2_t dgelu_fast_f2(gelu_f32x2_t x) {
    const gelu_f32x2_t z = x * x * gelu_f32x2_t{-0.72134752044448170368f, -0.72134752044448170368f};
    const gelu_f32x2_t e = {__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])};
    return __builtin_elementwise_fma(x * gelu_f32x2_t{0.39894228040143267794f, 0.39894228040143267794f}, e, phi_fast_f2(x));
}

#endif

// priority of a wave inside its MFMA segment (ping-pong GEMM kernels).  -DPP_SETPRIO_OFF: experiment hook
#ifdef PP_SETPRIO_OFF
#define PP_SETPRIO(X) do { } while (0)
#elif defined(PP_SETPRIO_HI)
#define PP_SETPRIO(X) __builtin_amdgcn_s_setprio((X) ? 3 : 0)
#else
#define PP_SETPRIO(X) __builtin_amdgcn_s_setprio(X)
#endif

// bijective XCD-aware remap of a linear workgroup id (cdna_hip_programming.md §5 template):
// consecutive logical tiles land on the same XCD (= same L2) instead of round-robin over the 8 XCDs.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    const int q = nwg / nx, r = nwg % nx;
    const int xcd = bid % nx, idx = bid / nx;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

}  // namespace egv

// ---- host side error plumbing shared by the C ABI translation units ----
extern "C" const char* egv_last_error(void);
// run-time switches (egv_api.cpp: the one table of names and defaults; the environment overrides one switch at a time)
int egv_cfg_int(const char* name, int def);
double egv_cfg_f64(const char* name, double def);
bool egv_cfg_on(const char* name, bool def);
void egv_set_error(const char* fmt, ...);
#define EGV_CHECK(cond, ...)                      \
    do {                                          \
        if (!(cond)) {                            \
            egv_set_error(__VA_ARGS__);           \
            return -1;                            \
        }                                         \
    } while (0)
#define EGV_LAUNCH_CHECK()                                                   \
    do {                                                                     \
        hipError_t e__ = hipGetLastError();                                  \
        if (e__ != hipSuccess) {                                             \
            egv_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
            return -2;                                                       \
        }                                                                    \
    } while (0)
