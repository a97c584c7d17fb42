// Fused multi-tensor AdamW for the EgoVLPv2 training step: the update of
// transformers==4.30.0 `AdamW` as used by set_optim_schedule.py:108 (eps=1e-8, betas=(0.9,0.98), correct_bias=True,
// weight decay applied AFTER the Adam update with the plain lr).  HBM-bound: 16 B read + 12 B written per parameter.
// One launch per parameter group; a device table with one record per tensor plus a chunk prefix array maps workgroups to
// 16K-element slices (binary search over <= a few hundred tensors).
#include "egv_common.h"

namespace egv {

struct OptChunk {
    float* p;
    const float* g;
    float* m;
    float* v;
    int n;
    int pad;
};

constexpr int OPT_CHUNK = 16384;

__global__ __launch_bounds__(256) void adamw_kernel(const OptChunk* __restrict__ table, const int* __restrict__ prefix, int ntensors,
                                                    float lr, float step_size, float beta1, float beta2, float eps,
                                                    float weight_decay, float grad_scale) {
    const int c = blockIdx.x;
    int lo = 0, hi = ntensors;                       // largest t with prefix[t] <= c
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid] <= c) lo = mid; else hi = mid;
    }
    OptChunk ch = table[lo];
    const int off = (c - prefix[lo]) * OPT_CHUNK;
    ch.p += off; ch.g += off; ch.m += off; ch.v += off;
    ch.n = min(OPT_CHUNK, ch.n - off);
    // 16-byte vectors only when all four pointers of this tensor are 16-byte aligned: gradients that are views into a DDP bucket
    // (gradient_as_bucket_view) start at arbitrary element offsets
    const bool al = ((reinterpret_cast<uintptr_t>(ch.p) | reinterpret_cast<uintptr_t>(ch.g) | reinterpret_cast<uintptr_t>(ch.m) |
                      reinterpret_cast<uintptr_t>(ch.v)) & 15) == 0;
    const int nv = al ? (ch.n >> 2) : 0;
    for (int i = threadIdx.x; i < nv; i += 256) {
        f32x4_t p = reinterpret_cast<f32x4_t*>(ch.p)[i];
        const f32x4_t g = reinterpret_cast<const f32x4_t*>(ch.g)[i];
        f32x4_t m = reinterpret_cast<f32x4_t*>(ch.m)[i];
        f32x4_t v = reinterpret_cast<f32x4_t*>(ch.v)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gg = g[e] * grad_scale;
            m[e] = m[e] * beta1 + gg * (1.0f - beta1);
            v[e] = v[e] * beta2 + gg * gg * (1.0f - beta2);
            float x = p[e] - step_size * (m[e] / (sqrtf(v[e]) + eps));
            x -= lr * weight_decay * x;
            p[e] = x;
        }
        reinterpret_cast<f32x4_t*>(ch.p)[i] = p;
        reinterpret_cast<f32x4_t*>(ch.m)[i] = m;
        reinterpret_cast<f32x4_t*>(ch.v)[i] = v;
    }
    for (int i = (nv << 2) + threadIdx.x; i < ch.n; i += 256) {
        const float gg = ch.g[i] * grad_scale;
        const float m = ch.m[i] * beta1 + gg * (1.0f - beta1);
        const float v = ch.v[i] * beta2 + gg * gg * (1.0f - beta2);
        float x = ch.p[i] - step_size * (m / (sqrtf(v) + eps));
        x -= lr * weight_decay * x;
        ch.p[i] = x; ch.m[i] = m; ch.v[i] = v;
    }
}

// ---- multi-tensor weight preparation: bf16 W and W^T copies of many fp32 [R, C] weights in one launch -----------------
struct CastRec {          // 32 bytes, one per tensor; R and C are multiples of 64
    const float* src;
    bf16_t* dst;          // [R, C] bf16
    bf16_t* dst_t;        // [C, R] bf16
    int R, C;
};

struct CastRecLd {        // 40 bytes: the same with a row pitch for the transposed copy (merged same-input projections)
    const float* src;
    bf16_t* dst;
    bf16_t* dst_t;        // element (c, r) at dst_t[c * ldt + r]
    int R, C, ldt, pad;
};
struct SegRec { const float* src; float* dst; long long n; };

__global__ __launch_bounds__(256) void copy_segments_kernel(const SegRec* __restrict__ table) {
    const SegRec s = table[blockIdx.x];
    for (long long i = threadIdx.x; i < s.n; i += 256) s.dst[i] = s.src[i];
}

template <typename REC>
__global__ __launch_bounds__(256) void cast_weights_kernel(const REC* __restrict__ table, const int* __restrict__ prefix, int ntensors) {
    __shared__ float tile[64][65];
    const int c = blockIdx.x;
    int lo = 0, hi = ntensors;                       // largest t with prefix[t] <= c
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid] <= c) lo = mid; else hi = mid;
    }
    const REC rec = table[lo];
    int ldt = rec.R;
    if constexpr (sizeof(REC) == sizeof(CastRecLd)) ldt = reinterpret_cast<const CastRecLd&>(rec).ldt;
    const int t = c - prefix[lo], tiles_c = rec.C >> 6;
    const int r0 = (t / tiles_c) << 6, c0 = (t % tiles_c) << 6;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;          // 16 lanes x 4 floats per 64-wide row, 16 rows per pass
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ty + i * 16;
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(rec.src + (size_t)(r0 + r) * rec.C + c0 + tx * 4);
        u32x2_t o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<u32x2_t*>(rec.dst + (size_t)(r0 + r) * rec.C + c0 + tx * 4) = o;
        tile[r][tx * 4 + 0] = v[0]; tile[r][tx * 4 + 1] = v[1]; tile[r][tx * 4 + 2] = v[2]; tile[r][tx * 4 + 3] = v[3];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cc = ty + i * 16;                                  // column of the source tile = row of the transposed copy
        u32x2_t o = {pack_bf16x2(tile[tx * 4 + 0][cc], tile[tx * 4 + 1][cc]), pack_bf16x2(tile[tx * 4 + 2][cc], tile[tx * 4 + 3][cc])};
        *reinterpret_cast<u32x2_t*>(rec.dst_t + (size_t)(c0 + cc) * ldt + r0 + tx * 4) = o;
    }
}

}  // namespace egv
using namespace egv;

// table: device array of {const float* src; bf16* dst; bf16* dst_t; int R; int C} (32 bytes per tensor, R % 64 == C % 64 == 0,
// pointers 16-byte aligned); prefix: device int32[ntensors + 1], prefix[t] = number of 64x64 tiles before tensor t.
extern "C" int egv_cast_weights(const void* table, const int* prefix, int ntensors, int ntiles, void* stream) {
    EGV_CHECK(table && prefix && ntensors > 0 && ntiles > 0, "egv_cast_weights: empty table");
    hipLaunchKernelGGL(cast_weights_kernel<CastRec>, dim3(ntiles), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), (const CastRec*)table,
                       prefix, ntensors);
    EGV_LAUNCH_CHECK();
    return 0;
}

extern "C" int egv_cast_weights_ld(const void* table, const int* prefix, int ntensors, int ntiles, void* stream) {
    EGV_CHECK(table && prefix && ntensors > 0 && ntiles > 0, "egv_cast_weights_ld: empty table");
    hipLaunchKernelGGL(cast_weights_kernel<CastRecLd>, dim3(ntiles), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), (const CastRecLd*)table,
                       prefix, ntensors);
    EGV_LAUNCH_CHECK();
    return 0;
}

extern "C" int egv_copy_segments(const void* table, int nseg, void* stream) {
    EGV_CHECK(table && nseg > 0, "egv_copy_segments: empty table");
    hipLaunchKernelGGL(copy_segments_kernel, dim3(nseg), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), (const SegRec*)table);
    EGV_LAUNCH_CHECK();
    return 0;
}

// table: device array of {float* p; const float* g; float* m; float* v; int n; int pad} (32 bytes, one per tensor, pointers
// 16-byte aligned); prefix: device int32[ntensors + 1], prefix[t] = number of 16384-element chunks before tensor t.
// step_size = lr * sqrt(1 - beta2^t) / (1 - beta1^t) (computed by the host per step).
extern "C" int egv_adamw_step(const void* table, const int* prefix, int ntensors, int nchunks, float lr, float step_size, float beta1,
                              float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
    EGV_CHECK(table && prefix && ntensors > 0 && nchunks > 0, "egv_adamw_step: empty table");
    hipLaunchKernelGGL(adamw_kernel, dim3(nchunks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), (const OptChunk*)table, prefix,
                       ntensors, lr, step_size, beta1, beta2, eps, weight_decay, grad_scale);
    EGV_LAUNCH_CHECK();
    return 0;
}
