// Shared declarations of the attention kernels (VALU reference-precision path: egv_attn.hip; MFMA bf16 path:
// egv_attn_mfma.hip).
#pragma once
#include "egv_common.h"

namespace egv {

constexpr int HD = 64;          // head dim
constexpr int TK = 64;          // other-side rows per LDS tile
constexpr int LDT = 68;         // float pitch of LDS tiles (conflict-free b128 row reads and b32 column reads)
constexpr int QPW = 8;          // own rows per wave (keeps every kernel under 64 KB of LDS)

struct RowSet {
    long long bs, base, gs, is;   // row(b,g,i) = b*bs + base + g*gs + i*is
    int n;
};

struct AttnArgs {
    const void* Q; const void* K; const void* V; void* O; const void* dO;
    void* dQ; void* dK; void* dV;
    int ldq, ldk, ldv, ldo, lddq, lddk, lddv;
    int qoff, koff, voff, ooff, dqoff, dkoff, dvoff;
    float* lse; float* delta; int H;
    RowSet q, k;
    int extra; long long extra_bs, extra_row;   // one extra row prepended on the OTHER side of the launched kernel
    float scale;
    const float* mask; int mask_ld;             // additive mask over key index: mask[b*mask_ld + i]
    int G;
    int nsplit; float* ws;                      // split of the other-side loop + fp32 partial slabs
    float drop_p; unsigned int drop_seed;       // attention-probability dropout (roberta.py:313); 0 = off
    float* O32;                                 // optional fp32 copy of O (same ld / offsets): written by the forward, read for delta
};

// counter-based dropout mask: a pure function of (seed, global query row, global key row, head), so forward, dQ and dK/dV
// kernels (MFMA or VALU, any tiling) regenerate the same mask.  Returns the multiplier 0 or 1/(1-p).
__device__ __forceinline__ unsigned int fmix32(unsigned int h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ float drop_mult(const AttnArgs& a, long long qrow, long long krow, int h) {
    if (a.drop_p <= 0.f) return 1.0f;
    unsigned int x = a.drop_seed ^ fmix32((unsigned int)qrow * 0x9E3779B1u + (unsigned int)h);
    x = fmix32(x ^ ((unsigned int)krow * 0x85EBCA77u + 0x165667B1u));
    const float u = (float)(x >> 8) * (1.0f / 16777216.0f);
    return u >= a.drop_p ? 1.0f / (1.0f - a.drop_p) : 0.0f;
}

__device__ __forceinline__ long long rs_row(const RowSet& r, int b, int g, int i) {
    return (long long)b * r.bs + r.base + (long long)g * r.gs + (long long)i * r.is;
}

}  // namespace egv

// Flat C description of one attention launch (see include/egovlp_hip.h).
struct egv_attn_desc {
    const void* Q; const void* K; const void* V; void* O; const void* dO; void* dQ; void* dK; void* dV;
    int ldq, ldk, ldv, ldo, lddq, lddk, lddv;
    int qoff, koff, voff, ooff, dqoff, dkoff, dvoff;
    float* lse; float* delta;
    int B, G, H;
    long long q_bs, q_base, q_gs, q_is; int q_n;
    long long k_bs, k_base, k_gs, k_is; int k_n;
    int extra; long long extra_bs, extra_row;
    float scale;
    const float* mask; int mask_ld;
    int nsplit; float* ws; long long ws_bytes;
    float drop_p; unsigned int drop_seed;
    float* O32;
};

static inline egv::AttnArgs to_args(const egv_attn_desc* d) {
    egv::AttnArgs a;
    a.Q = d->Q; a.K = d->K; a.V = d->V; a.O = d->O; a.dO = d->dO; a.dQ = d->dQ; a.dK = d->dK; a.dV = d->dV;
    a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv; a.ldo = d->ldo; a.lddq = d->lddq; a.lddk = d->lddk; a.lddv = d->lddv;
    a.qoff = d->qoff; a.koff = d->koff; a.voff = d->voff; a.ooff = d->ooff; a.dqoff = d->dqoff; a.dkoff = d->dkoff; a.dvoff = d->dvoff;
    a.lse = d->lse; a.delta = d->delta; a.H = d->H;
    a.q = egv::RowSet{d->q_bs, d->q_base, d->q_gs, d->q_is, d->q_n};
    a.k = egv::RowSet{d->k_bs, d->k_base, d->k_gs, d->k_is, d->k_n};
    a.extra = d->extra; a.extra_bs = d->extra_bs; a.extra_row = d->extra_row;
    a.scale = d->scale; a.mask = d->mask; a.mask_ld = d->mask_ld; a.G = d->G;
    a.nsplit = d->nsplit > 0 ? d->nsplit : 1; a.ws = d->ws;
    a.drop_p = d->drop_p; a.drop_seed = d->drop_seed;
    a.O32 = d->O32;
    return a;
}


// MFMA launchers (egv_attn_mfma.hip): return 1 if the problem shape is covered (and the kernel was enqueued), else 0.
int egv_attn_fwd_mfma(const egv::AttnArgs& a, int B, hipStream_t st);     // 2: the extra row's partial states were written too
bool egv_attn_fwd_cls_ok(const egv::AttnArgs& a);
int egv_attn_dq_mfma(const egv::AttnArgs& a, int B, hipStream_t st);
int egv_attn_dkv_mfma(const egv::AttnArgs& a, int B, hipStream_t st);
int egv_attn_bwd_fused_mfma(const egv::AttnArgs& a, int B, hipStream_t st);
bool egv_attn_bwd_pair_cls_ok(const egv::AttnArgs& a);
bool egv_attn_space_fwd_ok(const egv::AttnArgs& a, int B);
int egv_attn_space_fwd(const egv::AttnArgs& a, int B, hipStream_t st);  // egv_attn_space.hip: 1 group rows, 2 also the CLS row (a.ws set), 0 not covered
int egv_attn_space_bwd(const egv::AttnArgs& a, int B, hipStream_t st);  // one-launch backward of those groups (+ the CLS row's partials with a.ws)
bool egv_attn_time_fwd_ok(const egv::AttnArgs& a, int B);
int egv_attn_time_fwd(const egv::AttnArgs& a, int B, hipStream_t st);   // egv_attn_time.hip: forward of those groups incl. the CLS query (partials + combination)
int egv_attn_time_bwd(const egv::AttnArgs& a, int B, hipStream_t st);   // egv_attn_time.hip: one-launch backward of the <= 16-row groups (time attention)
int egv_attn_fewkeys_fwd(const egv::AttnArgs& a, int B, hipStream_t st);   // egv_attn_cross.hip: many queries over <= 32 keys (image -> text), 1 if enqueued
int egv_attn_fewkeys_bwd(const egv::AttnArgs& a, int B, hipStream_t st);   // ... dQ, dK, dV in one launch + the partial sum (a.ws: egv_attn_fewkeys_workspace_bytes)
extern "C" long long egv_attn_fewkeys_workspace_bytes(int B, int G, int H, int q_n);
int egv_attn_fewq_fwd(const egv::AttnArgs& a, int B, hipStream_t st);      // egv_attn_cross.hip: <= 32 queries over many keys (text -> image), 1 if enqueued (a.ws: egv_attn_fewq_workspace_bytes)
int egv_attn_fewq_bwd(const egv::AttnArgs& a, int B, hipStream_t st);      // ... dQ, dK, dV in one launch + the partial sum of dQ
extern "C" long long egv_attn_fewq_workspace_bytes(int B, int G, int H, int k_n);
void egv_attn_bwd_cls_reduce_launch(const egv::AttnArgs& a, int B, int self_term, hipStream_t st);
