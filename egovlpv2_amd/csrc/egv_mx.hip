// MX-fp8 operand preparation for the block-scaled weight GEMMs of BASELINE.json configs[4] (ViT-L/14, "fp8 MFMA weight path").
//
// Format (OCP Microscaling v1.0, MXFP8 E4M3): 32 consecutive elements along the contraction dimension share one E8M0 scale
// 2^(s - 127); elements are OCP e4m3fn (gfx950's native fp8).  The scale of a block is the smallest power of two that brings the
// block's largest magnitude to <= 448 (the e4m3 maximum): with amax = m * 2^E, 1 <= m < 2, that is 2^(E - 8) for m <= 1.75 and
// 2^(E - 7) above -- nothing saturates, and at least the top 3 mantissa bits of the largest element survive.  Elements are scaled
// exactly (v_ldexp_f32) and converted with v_cvt_pk_fp8_f32 (round to nearest even; measured on gfx950: no saturation, values
// above 464 become NaN -- the scale rule keeps every value <= 448, a clamp guards inf inputs).
//
// Outputs: q[R][K] fp8 codes, row-major (the GEMM stages them exactly like its bf16 operands: 128-byte row pieces by LDS-DMA),
// and the scale bytes in the order the v_mfma_scale_f32_16x16x128_f8f6f4 lanes want them (measured lane mapping: lane (fr, fg) of
// a 16-row fragment supplies the scale of row fr, 32-element block fg, in the byte its op_sel names):
//     s[ktile][row block][fg = k-block in the 128-wide K-tile][fr][4 bytes]
// so that the 256 bytes of (K-tile, row block) are one coalesced 4-byte-per-lane load for a wave, each lane receiving the
// bytes of ITS fragments.  Which row of the block sits in (fr, byte) depends on the operand's role in the GEMM tile:
//   role 0 (A: activations / output gradients; blocks of 48 rows = one sub-tile of a wave row of the 192-row tile):
//           row = byte * 16 + fr, byte = fragment i = 0..2 (byte 3 unused)
//   role 1 (B: weights; blocks of 64 rows): row = (byte >> 1) * 32 + (fr >> 2) * 8 + (byte & 1) * 4 + (fr & 3)   (the B-row
//           permutation of egv_gemm3.hip that makes a lane's accumulators 8 consecutive output columns)
// Bytes of rows past R are never written: allocate the array once, filled with 0x7f (scale 1).
#include "egv_common.h"

namespace egv {

struct MxRec {             // 40 bytes, one per tensor of a batched launch
    const bf16_t* src;     // [R, K] bf16, row pitch ld elements
    unsigned char* q;      // [R, K] fp8 e4m3 codes, row pitch K
    unsigned char* s;      // scale bytes (layout above)
    int R, K, ld, role;
};

__host__ __device__ inline int mx_nblk(int R, int role) { return role == 0 ? ((R + 191) / 192) * 4 : (R + 63) / 64; }

__device__ __forceinline__ void mx_quant_block(const MxRec& rec, int idx) {
    const int KB = rec.K >> 5;
    const int row = idx / KB, kb = idx - row * KB;
    const u32x4_t* src = reinterpret_cast<const u32x4_t*>(rec.src + (size_t)row * rec.ld + kb * 32);
    u32x4_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = src[i];
    float v[32];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[i * 8 + 2 * k] = __uint_as_float(w[i][k] << 16);
            v[i * 8 + 2 * k + 1] = __uint_as_float(w[i][k] & 0xffff0000u);
            amax = fmaxf(amax, fmaxf(fabsf(v[i * 8 + 2 * k]), fabsf(v[i * 8 + 2 * k + 1])));
        }
    const unsigned int bits = __float_as_uint(amax);
    int e8 = (int)(bits >> 23) - 8 + ((bits & 0x7fffffu) > 0x600000u ? 1 : 0);       // biased exponent of the block scale
    e8 = e8 < 0 ? 0 : (e8 > 254 ? 254 : e8);
    const int sh = 127 - e8;
    u32x4_t o[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = fminf(fmaxf(__builtin_amdgcn_ldexpf(v[i * 4 + k], sh), -448.f), 448.f);
        int p = __builtin_amdgcn_cvt_pk_fp8_f32(a[0], a[1], 0, false);
        p = __builtin_amdgcn_cvt_pk_fp8_f32(a[2], a[3], p, true);
        o[i >> 2][i & 3] = (unsigned int)p;
    }
    u32x4_t* dst = reinterpret_cast<u32x4_t*>(rec.q + (size_t)row * rec.K + kb * 32);
    dst[0] = o[0];
    dst[1] = o[1];
    const int ktile = kb >> 2, fg = kb & 3;
    int blk, fr, byte;
    if (rec.role == 0) { blk = row / 48; const int rb = row - blk * 48; fr = rb & 15; byte = rb >> 4; }
    else { blk = row >> 6; const int rb = row & 63, x = rb & 31; fr = ((x >> 3) << 2) | (x & 3); byte = (rb >> 5) * 2 + ((x >> 2) & 1); }
    rec.s[(((size_t)ktile * mx_nblk(rec.R, rec.role) + blk) * 4 + fg) * 64 + fr * 4 + byte] = (unsigned char)e8;
}

__global__ __launch_bounds__(256) void quant_mx_kernel(const MxRec rec) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx < (long long)rec.R * (rec.K >> 5)) mx_quant_block(rec, (int)idx);
}

__global__ __launch_bounds__(256) void quant_mx_batch_kernel(const MxRec* __restrict__ table, const int* __restrict__ prefix, int ntensors) {
    const int c = blockIdx.x;
    int lo = 0, hi = ntensors;                       // largest t with prefix[t] <= c
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid] <= c) lo = mid; else hi = mid;
    }
    const MxRec rec = table[lo];
    const long long idx = (long long)(c - prefix[lo]) * 256 + threadIdx.x;
    if (idx < (long long)rec.R * (rec.K >> 5)) mx_quant_block(rec, (int)idx);
}

}  // namespace egv
using namespace egv;

void* egv_prof_begin(void* stream);
void egv_prof_end(void* handle, void* stream, double flops, int kind, double bytes);

extern "C" long long egv_mx_scale_bytes(int R, int K, int role) {
    if (R <= 0 || K <= 0 || (K % 128) || (role != 0 && role != 1)) return -1;
    return (long long)(K / 128) * mx_nblk(R, role) * 256;
}

// x: bf16 [R, K] (row pitch ld elements, 16-byte aligned rows); q: [R, K] bytes; scales: egv_mx_scale_bytes(R, K, role) bytes
extern "C" int egv_quant_mx(const void* x, int R, int K, int ld, void* q, void* scales, int role, void* stream) {
    EGV_CHECK(x && q && scales && R > 0 && K > 0, "egv_quant_mx: null / empty operand");
    EGV_CHECK((K % 128) == 0 && (ld % 8) == 0 && (role == 0 || role == 1), "egv_quant_mx: K %% 128, ld %% 8, role in {0,1} required (K=%d ld=%d role=%d)", K, ld, role);
    EGV_CHECK(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(q)) & 15) == 0, "egv_quant_mx: 16-byte aligned pointers required");
    EGV_CHECK((long long)R * (K / 32) < (1LL << 31), "egv_quant_mx: too many blocks");
    MxRec rec{reinterpret_cast<const bf16_t*>(x), reinterpret_cast<unsigned char*>(q), reinterpret_cast<unsigned char*>(scales), R, K, ld, role};
    const long long nb = ((long long)R * (K / 32) + 255) / 256;
    void* ph = egv_prof_begin(stream);
    hipLaunchKernelGGL(quant_mx_kernel, dim3((unsigned)nb), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), rec);
    egv_prof_end(ph, stream, 0.0, 32, 3.0 * R * K + (double)R * K / 32);
    EGV_LAUNCH_CHECK();
    return 0;
}

// table: device array of MxRec {const bf16* src; u8* q; u8* s; int R, K, ld, role} (40 bytes per tensor); prefix: device
// int32[ntensors + 1], prefix[t] = number of 256-block workgroups before tensor t (a tensor takes ceil(R * K / 32 / 256)).
extern "C" int egv_quant_mx_batch(const void* table, const int* prefix, int ntensors, int nblocks, void* stream) {
    EGV_CHECK(table && prefix && ntensors > 0 && nblocks > 0, "egv_quant_mx_batch: empty table");
    hipLaunchKernelGGL(quant_mx_batch_kernel, dim3(nblocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), (const MxRec*)table, prefix,
                       ntensors);
    EGV_LAUNCH_CHECK();
    return 0;
}
