// Block-level entry points: one C-ABI call runs a whole SpaceTimeBlock (video_transformer.py:214-228, VarAttention :117-187)
// or a whole RobertaLayer (roberta.py:444-505) forward or backward -- the sequence of kernel launches the Python layer used to
// issue one ctypes call / one autograd node at a time.  Nothing is computed here: every step is one of the kernels behind
// include/egovlp_hip.h; this file owns the ORDER, the saved-activation layout, the scratch plan and the two-stream
// choreography of the backward pass (weight gradients run beside the data-gradient chain).
//
// Memory: `save` (forward writes, backward reads) and `ws` (scratch) are caller-allocated; sizes from the *_bytes queries.
// Streams: all work is enqueued on `stream`; with `stream2` != NULL the weight-gradient GEMMs of a backward call are enqueued
// on stream2 between an event recorded on `stream` (their operands are ready) and a final join (stream waits for stream2), so
// when the call returns every output is ordered on `stream` like single-stream work.  The events come from a small
// library-owned pool (created once, never destroyed); no other state is kept.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/egovlp_hip.h"

void egv_set_error(const char* fmt, ...);
int egv_cfg_int(const char* name, int def);          // run-time switches: the one table in egv_api.cpp
bool egv_cfg_on(const char* name, bool def);
void egv_gemm_set_cu_limit(int n);                  // egv_gemm3.hip: CUs the persistent forward / dgrad grids of this thread plan for
void egv_gemm_set_cu_slack(int n);                  // ... and how many CUs beyond that plan a grid may take to save a round (-1: EGV_PP_LIMIT_SLACK)
void egv_ln_bwd_defer_begin(void* ws, long long bytes);     // egv_norm.hip: the LayerNorm parameter-gradient partials of a call summed by ONE launch
int egv_ln_bwd_defer_flush(void* stream);
extern "C" int egv_layernorm_bwd2(int dtype, const void* dy, const void* x, const float* stats, const float* gamma,
                                  const void* add, const void* add2, void* dx, float* dgamma, float* dbeta, int M, int D,
                                  void* workspace, void* stream);

namespace {

#define BCHK(call)                    \
    do {                              \
        int rc__ = (call);            \
        if (rc__ != 0) return rc__;   \
    } while (0)

inline size_t al(size_t x) { return (x + 255) & ~size_t(255); }

struct Bump {
    char* base;
    size_t off = 0, cap;
    Bump(void* p, long long c) : base((char*)p), cap((size_t)c) {}
    void* take(size_t n) { return base + take_off(n); }
    size_t take_off(size_t n) {               // offset form (layout computation: base == nullptr)
        off = al(off);
        const size_t r = off;
        off += n;
        return r;
    }
    bool ok() const { return off <= cap; }
};

// four partial buffers of a LayerNorm backward (egv_layernorm_bwd_workspace_bytes each, 256-byte aligned slices)
size_t ln_defer_bytes(int M, int D) { return 4 * (al((size_t)egv_layernorm_bwd_workspace_bytes(M, D)) + 256); }
struct LnDeferScope {                               // egv_ln_bwd_defer_begin ... flush; an early return of the call ends the deferral
    bool open;
    LnDeferScope(void* ws, long long bytes) : open(true) { egv_ln_bwd_defer_begin(ws, bytes); }
    int flush(void* stream) { open = false; return egv_ln_bwd_defer_flush(stream); }
    ~LnDeferScope() { if (open) egv_ln_bwd_defer_begin(nullptr, 0); }
};

// ---- event ring (stream fork / join inside one call) ----
hipEvent_t next_event() {
    static thread_local std::vector<hipEvent_t> ring;
    static thread_local size_t pos = 0;
    if (ring.empty()) {
        ring.resize(64);
        for (auto& e : ring) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    }
    hipEvent_t e = ring[pos];
    pos = (pos + 1) % ring.size();
    return e;
}

struct Fork {                 // weight-gradient stream of one backward call
    hipStream_t main, side;
    bool used = false;
    Fork(void* m, void* s, int M) : main((hipStream_t)m), side((s && M >= 4096) ? (hipStream_t)s : (hipStream_t)m) {}
    bool forked() const { return side != main; }
    // order the side stream after everything enqueued on main so far; returns the stream to launch the weight gradient on
    void* begin() {
        if (forked()) {
            hipEvent_t e = next_event();
            (void)hipEventRecord(e, main);
            (void)hipStreamWaitEvent(side, e, 0);
            used = true;
        }
        return (void*)side;
    }
    void join() {
        if (forked() && used) {
            hipEvent_t e = next_event();
            (void)hipEventRecord(e, side);
            (void)hipStreamWaitEvent(main, e, 0);
        }
    }
};

inline int esz(int dtype) { return dtype == EGV_BF16 ? 2 : 4; }
inline char* at(const void* p, size_t bytes) { return (char*)p + bytes; }

// y = act(x W^T + b) (+ gate, residuals, saved pre-activation): forward of one Linear
int lin_fwd(int dt, int M, int N, int K, const void* x, const void* w, const float* b, void* y, int act, const float* gate,
            const void* r1, const void* r2, void* pre, void* st) {
    return egv_gemm(dt, 0, 0, M, N, K, x, K, w, K, y, N, 0, b, act, gate, r1, r2, pre, nullptr, 0, N, 1.0f, st);
}
// dx[M,K] = gate * (dz[M,N] W[N,K]) * act'(aux): NT form on the transposed copy when there is one
int lin_dgrad(int dt, int M, int N, int K, const void* dz, const void* w, const void* wt, void* dx, const float* gate, const void* aux,
              int dact, void* st) {
    if (wt) return egv_gemm(dt, 0, 0, M, K, N, dz, N, wt, N, dx, K, 0, nullptr, 0, gate, nullptr, nullptr, nullptr, aux, dact, K, 1.0f, st);
    return egv_gemm(dt, 0, 1, M, K, N, dz, N, w, K, dx, K, 0, nullptr, 0, gate, nullptr, nullptr, nullptr, aux, dact, K, 1.0f, st);
}
int lin_wgrad(int dt, int M, int N, int K, const void* dz, int ldz, const void* x, float* dw, float* db, const float* gate, void* ws,
              long long wsb, void* st) {
    return egv_gemm_wgrad(dt, M, N, K, dz, ldz, x, K, dw, db, 1.0f, gate, ws, wsb, st);
}

void rowset(long long& bs, long long& base, long long& gs, long long& is, int& n, long long a, long long b, long long c, long long d, int e) {
    bs = a; base = b; gs = c; is = d; n = e;
}

// activation code of the video MLP (fc1 forward epilogue / fc2 data-gradient epilogue).  EGV_GELU_DERIV=1 (bf16 mode): the tensor
// saved for the backward is gelu'(x) instead of x (EGV_ACT_GELU_D: the backward epilogue multiplies instead of evaluating erf and
// exp for 77 M elements per block).  Measured on configs[2] (alternating A/B): 74.2 -> 74.9 ms per step -- the fc2 data-gradient
// epilogue is not bound by its VALU work, and the forward epilogue pays for the second value -- so it stays OFF.
int mlp_act(int dt) {
    static const bool on = egv_cfg_on("EGV_GELU_DERIV", false);
    return (on && dt == EGV_BF16) ? EGV_ACT_GELU_D : EGV_ACT_GELU;
}

int nsplit_for(int n_other) {
    if (n_other <= 224) return 1;
    int s = n_other / 384;
    if (s > 32) s = 32;
    if (s < 2) s = 2;
    return s;
}

// ---- divided space / time attention on the fused qkv buffer [M, 3D] (VarAttention core, video_transformer.py:121-150) ----
struct Divided {
    int dt, B, Fr, N, H, D, S, M;
    bool space;
    void fill(egv_attn_desc& d, const void* qkv, void* O, float* lse) const {
        std::memset(&d, 0, sizeof(d));
        const int es = esz(dt);
        d.Q = qkv; d.K = at(qkv, (size_t)D * es); d.V = at(qkv, (size_t)2 * D * es); d.O = O;
        d.ldq = d.ldk = d.ldv = 3 * D; d.ldo = D;
        d.lse = lse;
        d.B = B; d.H = H;
        d.scale = 0.125f;                       // 64^-0.5 (head_dim 64)
        d.nsplit = 1;
    }
    void groups(egv_attn_desc& d) const {
        d.G = space ? Fr : N;
        if (space) { rowset(d.q_bs, d.q_base, d.q_gs, d.q_is, d.q_n, S, 1, N, 1, N); rowset(d.k_bs, d.k_base, d.k_gs, d.k_is, d.k_n, S, 1, N, 1, N); }
        else { rowset(d.q_bs, d.q_base, d.q_gs, d.q_is, d.q_n, S, 1, 1, N, Fr); rowset(d.k_bs, d.k_base, d.k_gs, d.k_is, d.k_n, S, 1, 1, N, Fr); }
        d.extra = 1; d.extra_bs = S; d.extra_row = 0;
    }
    long long ws_bytes() const {
        const int ns = nsplit_for(S);
        long long a = egv_attn_split_workspace_bytes(0, B, 1, H, 1, ns), b = egv_attn_split_workspace_bytes(1, B, 1, H, 1, ns),
                  c = egv_attn_bwd_dkv_workspace_bytes(B, 1, H, 1, ns);
        long long m = a > b ? a : b;
        const long long f = egv_attn_bwd_fused_workspace_bytes(B, space ? Fr : N, H), f2 = egv_attn_fwd_extra_workspace_bytes(B, space ? Fr : N, H);
        if (f > m) m = f;
        if (f2 > m) m = f2;
        return m > c ? m : c;
    }
    int fwd(const void* qkv, void* O, float* lse, void* ws, long long wsb, void* st) const {
        egv_attn_desc d;
        fill(d, qkv, O, lse);
        groups(d);
        static const bool fused_cls = egv_cfg_on("EGV_ATTN_FUSED_CLS", true);
        if (fused_cls && wsb >= egv_attn_fwd_extra_workspace_bytes(B, d.G, H)) { d.ws = (float*)ws; d.ws_bytes = wsb; }
        const bool covers = d.ws && egv_attn_fwd_covers_extra(dt, &d);
        if (!covers) { d.ws = nullptr; d.ws_bytes = 0; }
        BCHK(egv_attn_fwd(dt, &d, st));
        if (covers) return 0;                    // the group launch handled the CLS query too
        fill(d, qkv, O, lse);                    // CLS query over all S keys
        d.G = 1;
        rowset(d.q_bs, d.q_base, d.q_gs, d.q_is, d.q_n, S, 0, 0, 1, 1);
        rowset(d.k_bs, d.k_base, d.k_gs, d.k_is, d.k_n, S, 0, 0, 1, S);
        d.nsplit = nsplit_for(S);
        d.ws = (float*)ws; d.ws_bytes = wsb;
        return egv_attn_fwd(dt, &d, st);
    }
    int bwd(const void* qkv, const void* O, float* lse, const void* dO, void* dqkv, float* delta, void* ws, long long wsb, void* st) const {
        const int es = esz(dt);
        auto grads = [&](egv_attn_desc& d) {
            d.dO = dO; d.dQ = dqkv; d.dK = at(dqkv, (size_t)D * es); d.dV = at(dqkv, (size_t)2 * D * es);
            d.lddq = d.lddk = d.lddv = 3 * D;
            d.delta = delta;
        };
        egv_attn_desc d;
        const int ns = nsplit_for(S);
        // groups: dQ, dK, dV in one pass where the shape allows it (bf16 space attention); with the workspace that kernel also
        // produces the CLS row's gradients (per-group partials + one small sum) and nothing else is launched
        static const bool fused = egv_cfg_on("EGV_ATTN_FUSED_BWD", true);
        static const bool fused_cls = egv_cfg_on("EGV_ATTN_FUSED_CLS", true);
        fill(d, qkv, const_cast<void*>(O), lse); grads(d); groups(d);
        if (fused_cls && wsb >= egv_attn_bwd_fused_workspace_bytes(B, d.G, H)) { d.ws = (float*)ws; d.ws_bytes = wsb; }
        int fr = fused ? egv_attn_bwd_fused(dt, &d, st) : 1;
        if (fr < 0) return fr;
        if (fr == 0 && d.ws) return 0;
        if (fr == 1 && d.ws && egv_attn_bwd_pair_covers_extra(dt, &d)) {   // time attention: the kernel pair + the CLS partial sum
            BCHK(egv_attn_bwd_dq(dt, &d, st));
            BCHK(egv_attn_bwd_dkv(dt, &d, st));
            return egv_attn_bwd_extra_reduce(dt, &d, 1, st);
        }
        const bool groups_done = fr == 0;
        fill(d, qkv, const_cast<void*>(O), lse); grads(d);  // CLS query over all S keys (also its delta, which the key-owned group launch reads)
        d.G = 1;
        rowset(d.q_bs, d.q_base, d.q_gs, d.q_is, d.q_n, S, 0, 0, 1, 1);
        rowset(d.k_bs, d.k_base, d.k_gs, d.k_is, d.k_n, S, 0, 0, 1, S);
        d.nsplit = ns; d.ws = (float*)ws; d.ws_bytes = wsb;
        BCHK(egv_attn_bwd_dq(dt, &d, st));
        if (!groups_done) {
            fill(d, qkv, const_cast<void*>(O), lse); grads(d); groups(d);
            BCHK(egv_attn_bwd_dq(dt, &d, st));
            BCHK(egv_attn_bwd_dkv(dt, &d, st));
        }
        fill(d, qkv, const_cast<void*>(O), lse); grads(d);  // CLS key <- all S queries
        d.G = 1;
        rowset(d.q_bs, d.q_base, d.q_gs, d.q_is, d.q_n, S, 0, 0, 1, S);
        rowset(d.k_bs, d.k_base, d.k_gs, d.k_is, d.k_n, S, 0, 0, 1, 1);
        d.nsplit = ns; d.ws = (float*)ws; d.ws_bytes = wsb;
        return egv_attn_bwd_dkv(dt, &d, st);
    }
};

// ---- plain softmax(scale q k^T + mask) v per (batch, head) with optional probability dropout ----
struct Plain {
    int dt, B, H, D, nq, nk;
    float scale, drop_p;
    unsigned int drop_seed;
    const float* mask;                              // [B, nk] additive fp32 or NULL
    int fwd_ns() const { return nk <= 224 ? 1 : (nk + 223) / 224; }
    bool mask_ok_fused() const { return drop_p <= 0.f && nq >= 128; }
    int dkv_ns() const { return nq <= 224 ? 1 : (nq + 223) / 224; }
    long long ws_bytes() const {
        long long a = egv_attn_split_workspace_bytes(0, B, 1, H, nq, fwd_ns()), b = egv_attn_split_workspace_bytes(1, B, 1, H, nq, fwd_ns()),
                  c = egv_attn_bwd_dkv_workspace_bytes(B, 1, H, nk, dkv_ns());
        long long m = a > b ? a : b;
        if (nk <= 32 && egv_attn_fewkeys_workspace_bytes(B, 1, H, nq) > m) m = egv_attn_fewkeys_workspace_bytes(B, 1, H, nq);
        if (nq <= 32 && nk >= 512 && egv_attn_fewq_workspace_bytes(B, 1, H, nk) > m) m = egv_attn_fewq_workspace_bytes(B, 1, H, nk);
        return m > c ? m : c;
    }
    void fill(egv_attn_desc& d, const void* q, int ldq, const void* k, const void* v, int ldkv, void* O, float* lse) const {
        std::memset(&d, 0, sizeof(d));
        d.Q = q; d.K = k; d.V = v; d.O = O;
        d.ldq = ldq; d.ldk = d.ldv = ldkv; d.ldo = D;
        d.lse = lse;
        d.B = B; d.G = 1; d.H = H;
        rowset(d.q_bs, d.q_base, d.q_gs, d.q_is, d.q_n, nq, 0, 0, 1, nq);
        rowset(d.k_bs, d.k_base, d.k_gs, d.k_is, d.k_n, nk, 0, 0, 1, nk);
        d.scale = scale;
        d.mask = mask; d.mask_ld = mask ? nk : 0;
        d.nsplit = 1;
        d.drop_p = drop_p; d.drop_seed = drop_seed;
    }
    // o32 (bf16 storage, optional): fp32 values of O -- written by the forward, the source of delta in the backward (egv_attn_desc::O32)
    int fwd(const void* q, int ldq, const void* k, const void* v, int ldkv, void* O, float* lse, void* ws, long long wsb, void* st, float* o32 = nullptr) const {
        egv_attn_desc d;
        fill(d, q, ldq, k, v, ldkv, O, lse);
        d.nsplit = fwd_ns(); d.ws = (float*)ws; d.ws_bytes = wsb;
        d.O32 = dt == EGV_BF16 ? o32 : nullptr;
        return egv_attn_fwd(dt, &d, st);
    }
    int bwd(const void* q, int ldq, const void* k, const void* v, int ldkv, const void* O, float* lse, const void* dO, void* dq, int lddq,
            void* dk, void* dv, int lddkv, float* delta, void* ws, long long wsb, void* st, const float* o32 = nullptr) const {
        egv_attn_desc d;
        fill(d, q, ldq, k, v, ldkv, const_cast<void*>(O), lse);
        d.O32 = dt == EGV_BF16 ? const_cast<float*>(o32) : nullptr;
        d.dO = dO; d.dQ = dq; d.dK = dk; d.dV = dv; d.lddq = lddq; d.lddk = d.lddv = lddkv; d.delta = delta;
        d.nsplit = fwd_ns(); d.ws = (float*)ws; d.ws_bytes = wsb;
        if ((nk <= 32 && mask_ok_fused()) || (nq <= 32 && nk >= 512 && !mask)) {   // many queries over <= 32 keys, or <= 32 queries over many keys: one launch (egv_attn_cross.hip)
            const int r = egv_attn_bwd_fused(dt, &d, st);
            if (r <= 0) return r;
        }
        BCHK(egv_attn_bwd_dq(dt, &d, st));
        d.nsplit = dkv_ns();
        return egv_attn_bwd_dkv(dt, &d, st);
    }
};

// =====================================================================================================================
// SpaceTimeBlock
// =====================================================================================================================
enum { VW_TQKV = 0, VW_TPROJ, VW_SQKV, VW_SPROJ, VW_FC1, VW_FC2, VW_KV_I2T, VW_Q_I2T, VW_PROJ_I2T };
enum { VL_NORM3 = 0, VL_NORM1, VL_NORM2, VL_NORM_I2T };

struct VLayout {           // saved activations of one block (byte offsets into `save`)
    size_t stats3, h3, qkv_t, tctx, lse_t, tr, stats1, h1, qkv_s, sctx, lse_s, sr, stats2, h2, pre, act;
    size_t s, kv, stats_i, hs, q, o, lse_x, pg;     // fused only
    size_t total;
};

VLayout vlayout(const egv_vblock_desc* d) {
    VLayout L{};
    Bump b(nullptr, 0);
    const size_t es = esz(d->dtype);
    const size_t S = 1 + (size_t)d->F * d->N, M = (size_t)d->B * S, D = d->D, Hd = d->Hd, H = d->H;
    auto T = [&](size_t n) { return b.take_off(n); };
    L.stats3 = T(M * 8); L.h3 = T(M * D * es); L.qkv_t = T(M * 3 * D * es); L.tctx = T(M * D * es); L.lse_t = T(M * H * 4);
    L.tr = T(M * D * es); L.stats1 = T(M * 8); L.h1 = T(M * D * es); L.qkv_s = T(M * 3 * D * es);
    if (d->flags & EGV_BLOCK_HEAD) {                     // the head form stops at the space attention's qkv: the MLP / context slots (0.3 GB at configs[2]) are never written or read
        L.total = al(b.off);
        return L;
    }
    L.sctx = T(M * D * es);
    L.lse_s = T(M * H * 4); L.sr = T(M * D * es); L.stats2 = T(M * 8); L.h2 = T(M * D * es); L.pre = T(M * Hd * es); L.act = T(M * Hd * es);
    if (d->L > 0) {
        const size_t BL = (size_t)d->B * d->L;
        L.s = T(M * D * es); L.kv = T(BL * 2 * D * es); L.stats_i = T(M * 8); L.hs = T(M * D * es); L.q = T(M * D * es);
        L.o = T(M * D * es); L.lse_x = T(M * H * 4); L.pg = T(M * D * es);
    }
    L.total = al(b.off);
    return L;
}

// the weight gradients of a block go out as ONE grouped launch (egv_gemm5.hip) at the end of the backward call when the shapes
// allow it: bf16, D and Hd multiples of 256, enough tokens
// MX-fp8 forward / dgrad GEMMs (BASELINE.json configs[4]): bf16 block, flag set, shapes the MX kernel takes
bool vfp8_on(const egv_vblock_desc* d) {
    return (d->flags & EGV_BLOCK_FP8) && d->dtype == EGV_BF16 && (d->D % 128) == 0 && (d->Hd % 128) == 0 && d->D >= 384;
}
// scratch of one quantised A operand: codes [M, Kmax] + scale bytes
size_t vfp8_ws_bytes(const egv_vblock_desc* d) {
    if (!vfp8_on(d)) return 0;
    const long long M = (long long)d->B * (1 + (long long)d->F * d->N);
    const int Kmax = d->Hd > 3 * d->D ? d->Hd : 3 * d->D;
    // + a second operand: the MX-fp8 form of the MLP's hidden activation / its gradient, written by the producing GEMM's epilogue
    return al((size_t)M * Kmax) + al((size_t)egv_mx_scale_bytes((int)M, Kmax, 0)) + al((size_t)M * d->Hd) +
           al((size_t)egv_mx_scale_bytes((int)M, d->Hd, 0)) + 8192;
}
struct Fp8 {                                     // quantise-then-GEMM for one block call
    bool on;
    int M;
    void* q;
    void* s;
    void* st;
    void* q2 = nullptr;                          // second operand buffer (codes / scales of an [M, Hd] tensor written by a GEMM epilogue)
    void* s2 = nullptr;
    // y = epi(x W^T): x bf16 [M, K] -> MX codes, then the block-scaled GEMM on (wq, wq_s) [N, K]
    int lin(int N, int K, const void* x, const void* wq, const void* wq_s, const float* b, void* y, int act, const void* r1, void* pre,
            const void* aux, int dact) const {
        if (egv_quant_mx(x, M, K, K, q, s, 0, st)) return -1;
        return egv_gemm_mx(M, N, K, q, s, wq, wq_s, y, N, b, act, r1, pre, aux, dact, N, nullptr, nullptr, st);
    }
    // the same, and the output also lands in (q2, s2) in MX-fp8 form (GELU + saved pre-activation, or GELU' epilogue)
    int lin_qout(int N, int K, const void* x, bool x_is_quantised, const void* wq, const void* wq_s, const float* b, void* y, int act, void* pre,
                 const void* aux, int dact) const {
        if (!x_is_quantised && egv_quant_mx(x, M, K, K, q, s, 0, st)) return -1;
        return egv_gemm_mx(M, N, K, q, s, wq, wq_s, y, N, b, act, nullptr, pre, aux, dact, N, q2, s2, st);
    }
    // a Linear whose A operand already sits in (q2, s2)
    int lin_from_q2(int N, int K, const void* wq, const void* wq_s, const float* b, void* y, const void* r1) const {
        return egv_gemm_mx(M, N, K, q2, s2, wq, wq_s, y, N, b, 0, r1, nullptr, nullptr, 0, N, nullptr, nullptr, st);
    }
};

bool vgroup_ok(const egv_vblock_desc* d) {
    static const bool on = egv_cfg_on("EGV_WGRAD_GROUP", true);
    const long long M = (long long)d->B * (1 + (long long)d->F * d->N);
    return on && d->dtype == EGV_BF16 && (d->D % 256) == 0 && (d->Hd % 256) == 0 && M >= 4096;
}
int device_cus() {
    static int n = 0;
    if (!n) {
        hipDeviceProp_t prop;
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipGetDeviceProperties(&prop, dev);
        n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return n;
}
// CUs granted to the persistent grouped weight-gradient launch of a block when it runs beside the data-gradient chain of the
// next block: two thirds of a CU per output tile -- the launch is then two phases, one whole tile per workgroup and one half tile
// per workgroup (one reduction split for a third of the tiles), and takes about as long as that chain on the remaining CUs
// (measured on configs[2]: 96 of 256 CUs for the 144 tiles of an unfused block; 88 / 104 are 2 / 5 ms per step worse).
// EGV_WGRAD_CUS overrides.
int vgroup_cus(const egv_vblock_desc* d) {
    static const int forced = egv_cfg_int("EGV_WGRAD_CUS", 0);
    static const int forced_fused = egv_cfg_int("EGV_WGRAD_CUS_FUSED", 0);
    if (d->L > 0 && forced_fused > 0) return forced_fused;
    if (forced > 0) return forced;
    const int tD = d->D / 256, tH = d->Hd / 256;
    // A fused block gets the grant of the unfused form (96 CUs at ViT-B for 162 instead of 144 tiles): its data-gradient chain is a
    // third longer, so the launch has the time, and the 8 CUs this leaves beside the chain's 152-workgroup grids are what the text
    // layer's backward kernels (a dozen workgroups each, on the text stream) run on -- with 104 or 108 every CU is owned by a
    // persistent workgroup and the text layer, whose gradient the NEXT video block call waits for, only gets CUs between launches
    // (alternating A/B: 73.3 -> 72.2 ms per step).
    // (EGV_BLOCK_HEAD: only the two qkv gradients and the time projection's)
    const int ntile = (d->flags & EGV_BLOCK_HEAD) ? tD * tD + 2 * 3 * tD * tD : 2 * tD * tH + 2 * tD * tD + 2 * 3 * tD * tD;
    int g = ((2 * ntile + 2) / 3 / 8) * 8;                         // a multiple of 8: so is what it leaves the XCD-aware grids of the calling stream
    const int cap = device_cus() / 2;
    return g > cap ? cap : g;
}
// Deferred grouped weight gradients pay when the group fits beside the data-gradient chain: measured on the ViT-L/14 geometry
// (256 output tiles per block, bf16) every CU split loses to one launch per gradient on the companion stream (190-256 vs 183 ms per
// step), while with MX-fp8 data-gradient GEMMs -- a chain a third shorter -- the split wins (167 vs 213 ms).  EGV_WGRAD_DEFER_MAXTILES
// overrides the bf16 limit.
bool vdefer_ok(const egv_vblock_desc* d) {
    static const int max_tiles = egv_cfg_int("EGV_WGRAD_DEFER_MAXTILES", 192);
    const int tD = d->D / 256, tH = d->Hd / 256;
    const int ntile = 2 * tD * tH + 2 * tD * tD + 2 * 3 * tD * tD + (d->L > 0 ? 2 * tD * tD : 0);
    return vfp8_on(d) || ntile <= max_tiles;
}
struct CuLimit {                                    // scoped egv_gemm_set_cu_limit / egv_gemm_set_cu_slack
    bool on;
    explicit CuLimit(int n, int slack = -1) : on(n > 0) { if (on) { egv_gemm_set_cu_limit(n); egv_gemm_set_cu_slack(slack); } }
    ~CuLimit() { if (on) { egv_gemm_set_cu_limit(0); egv_gemm_set_cu_slack(-1); } }
};
long long vgroup_ws_bytes(const egv_vblock_desc* d) {
    if (!vgroup_ok(d)) return 0;
    const int M = d->B * (1 + d->F * d->N), D = d->D, Hd = d->Hd;
    egv_wgrad_problem pr[8];
    int n = 0;
    auto add = [&](int N, int K) { pr[n] = egv_wgrad_problem{}; pr[n].N = N; pr[n].K = K; ++n; };
    if (d->flags & EGV_BLOCK_HEAD) { add(3 * D, D); add(D, D); add(3 * D, D); }
    else {
        add(D, Hd); add(Hd, D); add(D, D); add(3 * D, D); add(D, D); add(3 * D, D);
        if (d->L > 0) { add(D, D); add(D, D); }
    }
    const long long b = egv_gemm_wgrad_grouped_workspace_bytes(M, n, pr, vgroup_cus(d));
    long long b0 = egv_gemm_wgrad_grouped_workspace_bytes(M, n, pr, (device_cus() * 7) / 8);     // single-stream mode
    if (const int tc = egv_cfg_int("EGV_WGRAD_TAIL_CUS", 0)) {                                    // the tail launch's own grant
        const long long b1 = egv_gemm_wgrad_grouped_workspace_bytes(M, n, pr, tc);
        if (b1 > b0) b0 = b1;
    }
    return b > b0 ? b : (b0 > 0 ? b0 : 0);
}

}  // namespace

extern "C" long long egv_vblock_save_bytes(const egv_vblock_desc* d) { return (long long)vlayout(d).total; }
extern "C" long long egv_vblock_qkv_s_offset(const egv_vblock_desc* d) { return d ? (long long)vlayout(d).qkv_s : -1; }
extern "C" int egv_vblock_next_slots(const egv_vblock_desc* d, long long* stats3_off, long long* h3_off) {
    if (!d || !stats3_off || !h3_off) { egv_set_error("egv_vblock_next_slots: null argument"); return -1; }
    const VLayout L = vlayout(d);
    *stats3_off = (long long)L.stats3;
    *h3_off = (long long)L.h3;
    return 0;
}

extern "C" long long egv_vblock_ws_bytes(const egv_vblock_desc* d, int backward) {
    const size_t es = esz(d->dtype);
    const size_t S = 1 + (size_t)d->F * d->N, M = (size_t)d->B * S, D = d->D, Hd = d->Hd, H = d->H;
    Divided dv{d->dtype, d->B, d->F, d->N, d->H, d->D, (int)S, (int)M, true};
    long long attn = dv.ws_bytes();
    {   // the time attention's groups (one per patch position) need the larger CLS-partial workspace
        Divided dvt{d->dtype, d->B, d->F, d->N, d->H, d->D, (int)S, (int)M, false};
        if (dvt.ws_bytes() > attn) attn = dvt.ws_bytes();
    }
    if (d->L > 0) {
        Plain p{d->dtype, d->B, d->H, d->D, (int)S, d->L, 0.125f, 0.f, 0u, nullptr};
        if (p.ws_bytes() > attn) attn = p.ws_bytes();
    }
    size_t tot = al((size_t)attn) + 4096 + vfp8_ws_bytes(d);
    if (!backward && (d->flags & EGV_BLOCK_RES_F32)) tot += al(M * D * es);      // the space projection's output beside the fp32 stream
    if (backward) {
        long long wg = 0;
        auto mx = [&](long long v) { if (v > wg) wg = v; };
        mx(egv_gemm_wgrad_workspace_bytes(3 * (int)D, (int)D, (int)M)); mx(egv_gemm_wgrad_workspace_bytes((int)D, (int)D, (int)M));
        mx(egv_gemm_wgrad_workspace_bytes((int)Hd, (int)D, (int)M)); mx(egv_gemm_wgrad_workspace_bytes((int)D, (int)Hd, (int)M));
        if (d->L > 0) mx(egv_gemm_wgrad_workspace_bytes(2 * (int)D, (int)D, d->B * d->L));
        mx(vgroup_ws_bytes(d));
        tot += al((size_t)wg) + al((size_t)egv_layernorm_bwd_workspace_bytes((int)M, (int)D)) + al(ln_defer_bytes((int)M, (int)D));
        tot += al(M * Hd * es) + 8 * al(M * D * es) + 2 * al(M * 3 * D * es) + 3 * al(M * H * 4) + 4096;
        if (d->L > 0) tot += 4 * al(M * D * es) + al((size_t)d->B * d->L * 2 * D * es) + 4096;
    }
    return (long long)tot + 65536;
}

extern "C" int egv_vblock_fwd(const egv_vblock_desc* d) {
    const int dt = d->dtype;
    const int S = 1 + d->F * d->N, M = d->B * S, D = d->D, Hd = d->Hd;
    const bool fused = d->L > 0;
    const bool head = (d->flags & EGV_BLOCK_HEAD) != 0;             // stop after the space attention's qkv projection (its slot of `save` is the result)
    if (!(d->x && (d->out || head) && d->save && d->ws)) { egv_set_error("egv_vblock_fwd: null buffer"); return -1; }
    if (head && !(d->flags & EGV_BLOCK_RES_F32)) { egv_set_error("egv_vblock_fwd: EGV_BLOCK_HEAD is built for the fp32 residual stream form"); return -1; }
    const VLayout L = vlayout(d);
    if ((long long)L.total > d->save_bytes) { egv_set_error("egv_vblock_fwd: save buffer too small"); return -1; }
    char* sv = (char*)d->save;
    void* st = d->stream;
    Bump ws(d->ws, d->ws_bytes);
    Divided dvt{dt, d->B, d->F, d->N, d->H, D, S, M, false}, dvs{dt, d->B, d->F, d->N, d->H, D, S, M, true};
    long long awb = dvs.ws_bytes() > dvt.ws_bytes() ? dvs.ws_bytes() : dvt.ws_bytes();
    Plain px{dt, d->B, d->H, D, S, d->L, 0.125f, 0.f, 0u, d->y_mask};
    if (fused && px.ws_bytes() > awb) awb = px.ws_bytes();
    void* aws = ws.take((size_t)awb);
    Fp8 f8{vfp8_on(d), M, nullptr, nullptr, st};
    if (f8.on) {
        const int Kmax = Hd > 3 * D ? Hd : 3 * D;
        f8.q = ws.take((size_t)M * Kmax);
        f8.s = ws.take((size_t)egv_mx_scale_bytes(M, Kmax, 0));
        f8.q2 = ws.take((size_t)M * Hd);
        f8.s2 = ws.take((size_t)egv_mx_scale_bytes(M, Hd, 0));
    }
    if (!ws.ok()) { egv_set_error("egv_vblock_fwd: workspace too small"); return -1; }
    CuLimit fwd_limit(d->fwd_cus > 0 && d->fwd_cus < device_cus() ? d->fwd_cus : 0, 0);   // (exact: the rest of the chip belongs to the other chain)
    static const bool epi_q = egv_cfg_on("EGV_MX_EPI_QUANT", true);
    const bool mlp_chain = epi_q && f8.on && d->wq[VW_FC1] && d->wq_s[VW_FC1] && d->wq[VW_FC2] && d->wq_s[VW_FC2];
    // EGV_BLOCK_INFER: no backward call follows -- what only the backward pass reads is not written.  That is the MLP's pre-activation
    // (M x Hd: as large as the activation itself; fc1 then runs the GELU epilogue without the second store).  Not with MX-fp8
    // operands (their fc1 epilogue is built with the saved pre-activation only).
    void* const fc1_pre = ((d->flags & EGV_BLOCK_INFER) && !f8.on) ? nullptr : (void*)(sv + L.pre);
    // one Linear over the M video tokens: MX-fp8 when the desc carries the quantised weight, bf16 otherwise
    auto lin = [&](int w, int N, int K, const void* x, void* y, int act, const void* r1, void* pre) -> int {
        if (f8.on && d->wq[w] && d->wq_s[w]) return f8.lin(N, K, x, d->wq[w], d->wq_s[w], d->b[w], y, act, r1, pre, nullptr, 0);
        return lin_fwd(dt, M, N, K, x, d->w[w], d->b[w], y, act, nullptr, r1, nullptr, pre, st);
    };
    // LayerNorm followed by a Linear on its output: in the MX-fp8 mode the LayerNorm kernel writes the quantised operand itself
    // (the same codes and scales egv_quant_mx would produce from h), so the Linear needs no quantiser launch
    auto ln_lin = [&](int ln, const void* xin, void* h, float* stats, int w, int N, int K, void* y, int act, void* pre) -> int {
        static const bool ln_mx = egv_cfg_on("EGV_LN_MX", true);
        if (ln_mx && f8.on && d->wq[w] && d->wq_s[w] && K == D) {
            if (egv_layernorm_fwd_mx(xin, h, d->ln_g[ln], d->ln_b[ln], stats, f8.q, f8.s, M, D, d->eps, st)) return -1;
            if (w == VW_FC1 && mlp_chain)       // fc1's GELU epilogue also writes the MX-fp8 form of the activation: fc2's operand
                return egv_gemm_mx(M, N, K, f8.q, f8.s, d->wq[w], d->wq_s[w], y, N, d->b[w], act, nullptr, pre, nullptr, 0, N, f8.q2, f8.s2, st);
            return egv_gemm_mx(M, N, K, f8.q, f8.s, d->wq[w], d->wq_s[w], y, N, d->b[w], act, nullptr, pre, nullptr, 0, N, nullptr, nullptr, st);
        }
        if (egv_layernorm_fwd(dt, xin, h, d->ln_g[ln], d->ln_b[ln], stats, M, D, d->eps, st)) return -1;
        return lin(w, N, K, h, y, act, nullptr, pre);
    };

    if (d->flags & EGV_BLOCK_RES_F32) {
        // fp32 residual stream (bf16 GEMM operands and outputs, the three residual sums and the LayerNorm inputs in fp32: what
        // torch.autocast keeps in fp32, trainer_egoclip.py:143).  d->x32 = the stream's fp32 value at the block input (NULL: d->x is
        // exact), d->out32 receives the fp32 output; d->x / d->out / the saved tr, sr are their bf16 roundings -- the backward pass
        // reads those and is unchanged.  The Linears run WITHOUT their residual epilogues; every sum is formed by the kernel that
        // normalises it (egv_sum_layernorm) from the fp32 base and the bf16 Linear outputs.
        if (dt != EGV_BF16 || f8.on || (!d->out32 && !head)) {
            egv_set_error("egv_vblock_fwd: EGV_BLOCK_RES_F32 needs a bf16 block without MX-fp8 operands and an out32 buffer");
            return -1;
        }
        void* ys = ws.take((size_t)M * D * esz(dt));
        if (!ws.ok()) { egv_set_error("egv_vblock_fwd: workspace too small"); return -1; }
        const float* x32 = d->x32;
        auto sumln = [&](const void* d1, const void* d2, const void* dg, float* s32, void* s16, void* y, int ln, float* stats) -> int {
            return egv_sum_layernorm(x32, x32 ? nullptr : d->x, d1, d2, dg, dg ? d->alpha : nullptr, s32, s16, y, y ? d->ln_g[ln] : nullptr,
                                     y ? d->ln_b[ln] : nullptr, stats, M, D, d->eps, st);
        };
        if (!(d->flags & EGV_BLOCK_H3_READY))                     // (else the previous block's output pass left norm3(x) and its statistics in place)
            BCHK(sumln(nullptr, nullptr, nullptr, nullptr, nullptr, sv + L.h3, VL_NORM3, (float*)(sv + L.stats3)));
        BCHK(lin(VW_TQKV, 3 * D, D, sv + L.h3, sv + L.qkv_t, 0, nullptr, nullptr));
        BCHK(dvt.fwd(sv + L.qkv_t, sv + L.tctx, (float*)(sv + L.lse_t), aws, awb, st));
        BCHK(lin(VW_TPROJ, D, D, sv + L.tctx, sv + L.tr, 0, nullptr, nullptr));                    // the projection, then tr = x + it in place
        // (EGV_BLOCK_INFER: the bf16 roundings of the two inner sums, tr and sr, are read by the backward pass only -- not written)
        const bool lean = (d->flags & EGV_BLOCK_INFER) != 0;
        BCHK(sumln(sv + L.tr, nullptr, nullptr, nullptr, lean ? nullptr : sv + L.tr, sv + L.h1, VL_NORM1, (float*)(sv + L.stats1)));
        BCHK(lin(VW_SQKV, 3 * D, D, sv + L.h1, sv + L.qkv_s, 0, nullptr, nullptr));
        if (head) return 0;                                        // the caller goes on with the CLS query alone (model.py: _video_block_tail)
        BCHK(dvs.fwd(sv + L.qkv_s, sv + L.sctx, (float*)(sv + L.lse_s), aws, awb, st));
        const void *a1, *ag = nullptr;                                                             // sr = x + a1 (+ alpha * ag)
        if (!fused) {
            BCHK(lin(VW_SPROJ, D, D, sv + L.sctx, ys, 0, nullptr, nullptr));
            a1 = ys;
        } else {
            const int BL = d->B * d->L;
            BCHK(lin(VW_SPROJ, D, D, sv + L.sctx, sv + L.s, 0, nullptr, nullptr));
            BCHK(lin_fwd(dt, BL, 2 * D, D, d->y, d->w[VW_KV_I2T], d->b[VW_KV_I2T], sv + L.kv, 0, nullptr, nullptr, nullptr, nullptr, st));
            BCHK(ln_lin(VL_NORM_I2T, sv + L.s, sv + L.hs, (float*)(sv + L.stats_i), VW_Q_I2T, D, D, sv + L.q, 0, nullptr));
            BCHK(px.fwd(sv + L.q, D, sv + L.kv, at(sv + L.kv, (size_t)D * esz(dt)), 2 * D, sv + L.o, (float*)(sv + L.lse_x), aws, awb, st));
            BCHK(lin_fwd(dt, M, D, D, sv + L.o, d->w[VW_PROJ_I2T], d->b[VW_PROJ_I2T], sv + L.pg, 0, nullptr, nullptr, nullptr, nullptr, st));
            a1 = sv + L.s; ag = sv + L.pg;
        }
        BCHK(sumln(a1, nullptr, ag, nullptr, lean ? nullptr : sv + L.sr, sv + L.h2, VL_NORM2, (float*)(sv + L.stats2)));
        BCHK(lin(VW_FC1, Hd, D, sv + L.h2, sv + L.act, mlp_act(dt), nullptr, fc1_pre));
        BCHK(lin(VW_FC2, D, Hd, sv + L.act, d->out, 0, nullptr, nullptr));                         // the MLP's output, then out = sr + it in place
        // the block's output sum, in fp32 and bf16 -- and, when the caller names the next block's norm3 and save slots, that LayerNorm too
        const bool fold = d->next_h && d->next_g && d->next_b && d->next_stats;
        return egv_sum_layernorm(x32, x32 ? nullptr : d->x, a1, d->out, ag, ag ? d->alpha : nullptr, d->out32, d->out, fold ? d->next_h : nullptr,
                                 fold ? d->next_g : nullptr, fold ? d->next_b : nullptr, fold ? d->next_stats : nullptr, M, D, d->eps, st);
    }
    // temporal attention (video_transformer.py:217-218): x + proj(attn(qkv(norm3 x)))
    BCHK(ln_lin(VL_NORM3, d->x, sv + L.h3, (float*)(sv + L.stats3), VW_TQKV, 3 * D, D, sv + L.qkv_t, 0, nullptr));
    BCHK(dvt.fwd(sv + L.qkv_t, sv + L.tctx, (float*)(sv + L.lse_t), aws, awb, st));
    BCHK(lin(VW_TPROJ, D, D, sv + L.tctx, sv + L.tr, 0, d->x, nullptr));
    // spatial attention (:219-222): residual from x, not from the time residual
    BCHK(ln_lin(VL_NORM1, sv + L.tr, sv + L.h1, (float*)(sv + L.stats1), VW_SQKV, 3 * D, D, sv + L.qkv_s, 0, nullptr));
    BCHK(dvs.fwd(sv + L.qkv_s, sv + L.sctx, (float*)(sv + L.lse_s), aws, awb, st));
    if (!fused) {
        BCHK(lin(VW_SPROJ, D, D, sv + L.sctx, sv + L.sr, 0, d->x, nullptr));
    } else {
        // image-to-text cross attention (:155-185): s = proj(ctx); q from norm_i2t_i(s), k|v from the text states; x + s + alpha*proj_i2t(o)
        const int BL = d->B * d->L;
        BCHK(lin(VW_SPROJ, D, D, sv + L.sctx, sv + L.s, 0, nullptr, nullptr));
        BCHK(lin_fwd(dt, BL, 2 * D, D, d->y, d->w[VW_KV_I2T], d->b[VW_KV_I2T], sv + L.kv, 0, nullptr, nullptr, nullptr, nullptr, st));
        BCHK(ln_lin(VL_NORM_I2T, sv + L.s, sv + L.hs, (float*)(sv + L.stats_i), VW_Q_I2T, D, D, sv + L.q, 0, nullptr));
        BCHK(px.fwd(sv + L.q, D, sv + L.kv, at(sv + L.kv, (size_t)D * esz(dt)), 2 * D, sv + L.o, (float*)(sv + L.lse_x), aws, awb, st));
        BCHK(lin_fwd(dt, M, D, D, sv + L.o, d->w[VW_PROJ_I2T], d->b[VW_PROJ_I2T], sv + L.sr, 0, d->alpha, sv + L.s, d->x, sv + L.pg, st));
    }
    // MLP (:226): sr + fc2(gelu(fc1(norm2 sr)))
    BCHK(ln_lin(VL_NORM2, sv + L.sr, sv + L.h2, (float*)(sv + L.stats2), VW_FC1, Hd, D, sv + L.act, mlp_act(dt), fc1_pre));
    if (mlp_chain) BCHK(f8.lin_from_q2(D, Hd, d->wq[VW_FC2], d->wq_s[VW_FC2], d->b[VW_FC2], d->out, sv + L.sr));
    else BCHK(lin(VW_FC2, D, Hd, sv + L.act, d->out, 0, sv + L.sr, nullptr));
    return 0;
}

// 1 if egv_vblock_bwd(d) with EGV_BLOCK_NO_JOIN set would return with weight-gradient work still running on d->stream2 (the caller
// then owes the join and must keep ws / save / dout alive until stream2 has drained), 0 if the call joins by itself anyway
extern "C" int egv_vblock_bwd_defers(const egv_vblock_desc* d) {
    const long long M = (long long)d->B * (1 + (long long)d->F * d->N);
    return (vgroup_ok(d) && vdefer_ok(d) && d->stream2 && d->stream2 != d->stream && M >= 4096) ? 1 : 0;
}

// Linear slots whose weight gradients egv_vblock_bwd(d) forms in its grouped launch (those over the M video tokens) -- the slots a caller
// may ask to ACCUMULATE (acc_mask); 0 when the call would take the one-launch-per-gradient form
extern "C" unsigned int egv_vblock_bwd_groups(const egv_vblock_desc* d) {
    if (!vgroup_ok(d)) return 0u;
    if (d->flags & EGV_BLOCK_HEAD) return (1u << VW_TQKV) | (1u << VW_TPROJ) | (1u << VW_SQKV);
    unsigned int m = (1u << VW_TQKV) | (1u << VW_TPROJ) | (1u << VW_SQKV) | (1u << VW_SPROJ) | (1u << VW_FC1) | (1u << VW_FC2);
    if (d->L > 0) m |= (1u << VW_Q_I2T) | (1u << VW_PROJ_I2T);
    return m;
}

extern "C" int egv_vblock_bwd(const egv_vblock_desc* d) {
    const int dt = d->dtype;
    const size_t es = esz(dt);
    const int S = 1 + d->F * d->N, M = d->B * S, D = d->D, Hd = d->Hd, H = d->H;
    const bool fused = d->L > 0;
    if (!(d->x && d->dout && d->dx && d->save && d->ws)) { egv_set_error("egv_vblock_bwd: null buffer"); return -1; }
    const bool head = (d->flags & EGV_BLOCK_HEAD) != 0;             // dout = the gradient of the space attention's qkv [M, 3D]; the call starts there
    const VLayout L = vlayout(d);
    const char* sv = (const char*)d->save;
    void* st = d->stream;
    Fork fk(d->stream, d->stream2, M);
    Bump ws(d->ws, d->ws_bytes);
    Divided dvt{dt, d->B, d->F, d->N, d->H, D, S, M, false}, dvs{dt, d->B, d->F, d->N, d->H, D, S, M, true};
    long long awb = dvs.ws_bytes() > dvt.ws_bytes() ? dvs.ws_bytes() : dvt.ws_bytes();
    Plain px{dt, d->B, d->H, D, S, d->L, 0.125f, 0.f, 0u, d->y_mask};
    if (fused && px.ws_bytes() > awb) awb = px.ws_bytes();
    void* aws = ws.take((size_t)awb);
    long long wgb = 0;
    {
        auto mx = [&](long long v) { if (v > wgb) wgb = v; };
        mx(egv_gemm_wgrad_workspace_bytes(3 * D, D, M)); mx(egv_gemm_wgrad_workspace_bytes(D, D, M));
        mx(egv_gemm_wgrad_workspace_bytes(Hd, D, M)); mx(egv_gemm_wgrad_workspace_bytes(D, Hd, M));
        if (fused) mx(egv_gemm_wgrad_workspace_bytes(2 * D, D, d->B * d->L));
        mx(vgroup_ws_bytes(d));
    }
    void* wgw = ws.take((size_t)wgb);                       // weight-gradient slabs: the side stream runs them one after another
    void* lnw = ws.take((size_t)egv_layernorm_bwd_workspace_bytes(M, D));
    // the call's three (fused: four) LayerNorm backward passes keep their parameter-gradient partials until ONE reduction launch at the end
    // of the call (nothing in the call reads dgamma / dbeta): two or three dependent 6-us launches less on the calling stream per call
    LnDeferScope ln_defer(ws.take(ln_defer_bytes(M, D)), (long long)ln_defer_bytes(M, D));
    void* dpre = ws.take((size_t)M * Hd * es);
    void* dh2 = ws.take((size_t)M * D * es);
    void* d_sr = ws.take((size_t)M * D * es);
    void* d_sctx = ws.take((size_t)M * D * es);
    void* dqkv_s = ws.take((size_t)M * 3 * D * es);
    if (head) dqkv_s = const_cast<void*>(d->dout);
    void* dh1 = ws.take((size_t)M * D * es);
    void* d_tr = ws.take((size_t)M * D * es);
    void* d_tctx = ws.take((size_t)M * D * es);
    void* dqkv_t = ws.take((size_t)M * 3 * D * es);
    void* dh3 = ws.take((size_t)M * D * es);
    float* delta_s = (float*)ws.take((size_t)M * H * 4);
    float* delta_t = (float*)ws.take((size_t)M * H * 4);
    void *d_s = nullptr, *dhs = nullptr, *dq = nullptr, *d_o = nullptr, *dkv = nullptr;
    float* delta_x = nullptr;
    if (fused) {
        d_s = ws.take((size_t)M * D * es); dhs = ws.take((size_t)M * D * es); dq = ws.take((size_t)M * D * es); d_o = ws.take((size_t)M * D * es);
        dkv = ws.take((size_t)d->B * d->L * 2 * D * es);
        delta_x = (float*)ws.take((size_t)M * H * 4);
    }
    void* dotw = ws.take(4096);
    Fp8 f8{vfp8_on(d), M, nullptr, nullptr, st};
    if (f8.on) {
        const int Kmax = Hd > 3 * D ? Hd : 3 * D;
        f8.q = ws.take((size_t)M * Kmax);
        f8.s = ws.take((size_t)egv_mx_scale_bytes(M, Kmax, 0));
        f8.q2 = ws.take((size_t)M * Hd);
        f8.s2 = ws.take((size_t)egv_mx_scale_bytes(M, Hd, 0));
    }
    if (!ws.ok()) { egv_set_error("egv_vblock_bwd: workspace too small (%zu > %zu)", ws.off, ws.cap); return -1; }
    static const bool epi_q = egv_cfg_on("EGV_MX_EPI_QUANT", true);
    const bool mlp_chain = epi_q && f8.on && d->wtq[VW_FC1] && d->wtq_s[VW_FC1] && d->wtq[VW_FC2] && d->wtq_s[VW_FC2];
    // dx[M,K] = (dz[M,N] W[N,K]) * act'(aux) over the M video tokens: MX-fp8 on the quantised transposed weight when the desc
    // carries it (the output gradient is quantised along N, the contraction), bf16 otherwise
    auto dgrad = [&](int w, int N, int K, const void* dz, void* dx, const void* aux, int dact) -> int {
        if (f8.on && d->wtq[w] && d->wtq_s[w]) return f8.lin(K, N, dz, d->wtq[w], d->wtq_s[w], nullptr, dx, 0, nullptr, nullptr, aux, dact);
        return lin_dgrad(dt, M, N, K, dz, d->w[w], d->wt[w], dx, nullptr, aux, dact, st);
    };
    // weight gradients over the M video tokens: collected and launched together at the end of the call (every operand -- saved
    // activations, scratch, dout -- stays untouched until then), or one launch each on the side stream as soon as ready
    // Where the weight gradients run:
    //  * companion stream + EGV_BLOCK_NO_JOIN (the caller joins once per backward pass): ONE persistent grouped launch on a granted
    //    share of the CUs, started at the end of this call, running beside the NEXT call's data-gradient chain -- whose persistent
    //    GEMM grids plan for the remaining CUs (a grid planned for CUs it cannot get runs its surplus as a second, nearly empty round);
    //  * no companion stream: one grouped launch on the calling stream (7/8 of the CUs: the chip-wide rate of this kernel peaks there);
    //  * companion stream, joined inside the call (DistributedDataParallel, gradient accumulation, hooks): one launch per gradient
    //    as soon as its operands exist, fp32 slabs + reduction launch (egv_gemm4.hip).
    const bool side_group = vgroup_ok(d) && vdefer_ok(d) && fk.forked() && (d->flags & EGV_BLOCK_NO_JOIN);
    const bool group = vgroup_ok(d) && (side_group || !fk.forked());
    static const int main_limit = egv_cfg_int("EGV_WGRAD_MAIN_LIMIT", 0);
    // fused blocks: the group's launch starts with this call (the calling stream first waits for the text layer's gradient) and
    // is resident through the MLP's data gradients -- their grids take exactly the CUs it leaves (EGV_PP_LIMIT_SLACK_FUSED)
    static const int slack_fused = egv_cfg_int("EGV_PP_LIMIT_SLACK_FUSED", 0);
    CuLimit cu_limit(side_group ? (main_limit > 0 ? main_limit : device_cus() - vgroup_cus(d)) : 0, fused ? slack_fused : -1);
    egv_wgrad_problem grp[8];
    int ngrp = 0;
    auto wgrad = [&](int N, int K, const void* dz, const void* x, int w, const float* gate, int rows) -> int {
        if (group && rows == M) {
            egv_wgrad_problem& q = grp[ngrp++];
            q.dy = dz; q.ldy = N; q.x = x; q.ldx = K; q.dw = d->dw[w]; q.db = d->db[w]; q.gate = gate; q.N = N; q.K = K;
            q.accumulate = (d->acc_mask >> w) & 1;
            return 0;
        }
        if ((d->acc_mask >> w) & 1) { egv_set_error("egv_vblock_bwd: acc_mask bit %d set for a weight gradient outside the grouped launch", w); return -1; }
        return lin_wgrad(dt, rows, N, K, dz, N, x, d->dw[w], d->db[w], gate, wgw, wgb, fk.begin());
    };

    if (!head) {
    // ---- MLP: out = sr + fc2(gelu(pre)), pre = fc1(h2)
    BCHK(wgrad(D, Hd, d->dout, sv + L.act, VW_FC2, nullptr, M));
    if (mlp_chain) {
        // fc2's data gradient writes dpre = (dout W2) * gelu'(pre) in bf16 (the weight gradient of fc1 reads it) AND in MX-fp8 form:
        // the operand of fc1's data gradient, without a quantiser launch
        BCHK(f8.lin_qout(Hd, D, d->dout, false, d->wtq[VW_FC2], d->wtq_s[VW_FC2], nullptr, dpre, 0, nullptr, sv + L.pre, mlp_act(dt)));
        BCHK(wgrad(Hd, D, dpre, sv + L.h2, VW_FC1, nullptr, M));
        BCHK(f8.lin_from_q2(D, Hd, d->wtq[VW_FC1], d->wtq_s[VW_FC1], nullptr, dh2, nullptr));
    } else {
        BCHK(dgrad(VW_FC2, D, Hd, d->dout, dpre, sv + L.pre, mlp_act(dt)));
        BCHK(wgrad(Hd, D, dpre, sv + L.h2, VW_FC1, nullptr, M));
        BCHK(dgrad(VW_FC1, Hd, D, dpre, dh2, nullptr, 0));
    }
    // d_sr = LN2'(dh2) + dout (skip path of the MLP residual)
    BCHK(egv_layernorm_bwd2(dt, dh2, sv + L.sr, (const float*)(sv + L.stats2), d->ln_g[VL_NORM2], d->dout, nullptr, d_sr, d->dln_g[VL_NORM2],
                            d->dln_b[VL_NORM2], M, D, lnw, st));
    const void* d_sproj_out = d_sr;                          // gradient at the output of attn.proj
    if (fused) {
        // sr = alpha * P + s + x, P = proj_i2t(o) (saved as pg)
        const int BL = d->B * d->L;
        BCHK(egv_dot(dt, d_sr, sv + L.pg, (long long)M * D, d->dalpha, 1.0f, dotw, st));
        BCHK(wgrad(D, D, d_sr, sv + L.o, VW_PROJ_I2T, d->alpha, M));
        BCHK(lin_dgrad(dt, M, D, D, d_sr, d->w[VW_PROJ_I2T], d->wt[VW_PROJ_I2T], d_o, d->alpha, nullptr, 0, st));
        BCHK(px.bwd(sv + L.q, D, sv + L.kv, at(sv + L.kv, (size_t)D * es), 2 * D, sv + L.o, (float*)const_cast<char*>(sv + L.lse_x), d_o, dq, D, dkv,
                    at(dkv, (size_t)D * es), 2 * D, delta_x, aws, awb, st));
        BCHK(wgrad(D, D, dq, sv + L.hs, VW_Q_I2T, nullptr, M));
        BCHK(dgrad(VW_Q_I2T, D, D, dq, dhs, nullptr, 0));
        BCHK(egv_layernorm_bwd2(dt, dhs, sv + L.s, (const float*)(sv + L.stats_i), d->ln_g[VL_NORM_I2T], d_sr, nullptr, d_s, d->dln_g[VL_NORM_I2T],
                                d->dln_b[VL_NORM_I2T], M, D, lnw, st));
        BCHK(lin_wgrad(dt, BL, 2 * D, D, dkv, 2 * D, d->y, d->dw[VW_KV_I2T], d->db[VW_KV_I2T], nullptr, wgw, wgb, fk.begin()));
        if (d->dy) BCHK(lin_dgrad(dt, BL, 2 * D, D, dkv, d->w[VW_KV_I2T], d->wt[VW_KV_I2T], d->dy, nullptr, nullptr, 0, st));
        d_sproj_out = d_s;
    }
    // fused blocks: by now (MLP and image-to-text part done) the grouped launch of the previous call has finished or is in its last
    // phase: the remaining GEMMs of the chain plan for the whole chip (EGV_FUSED_LIMIT_LIFT; alternating A/B 72.2 -> 71.9 ms per step)
    static const bool lift = egv_cfg_on("EGV_FUSED_LIMIT_LIFT", true);
    if (fused && lift && cu_limit.on) egv_gemm_set_cu_limit(0);
    // ---- spatial attention
    BCHK(wgrad(D, D, d_sproj_out, sv + L.sctx, VW_SPROJ, nullptr, M));
    BCHK(dgrad(VW_SPROJ, D, D, d_sproj_out, d_sctx, nullptr, 0));
    BCHK(dvs.bwd(sv + L.qkv_s, sv + L.sctx, (float*)const_cast<char*>(sv + L.lse_s), d_sctx, dqkv_s, delta_s, aws, awb, st));
    }   // !head
    BCHK(wgrad(3 * D, D, dqkv_s, sv + L.h1, VW_SQKV, nullptr, M));
    BCHK(dgrad(VW_SQKV, 3 * D, D, dqkv_s, dh1, nullptr, 0));
    BCHK(egv_layernorm_bwd2(dt, dh1, sv + L.tr, (const float*)(sv + L.stats1), d->ln_g[VL_NORM1], nullptr, nullptr, d_tr, d->dln_g[VL_NORM1],
                            d->dln_b[VL_NORM1], M, D, lnw, st));
    // ---- temporal attention
    BCHK(wgrad(D, D, d_tr, sv + L.tctx, VW_TPROJ, nullptr, M));
    BCHK(dgrad(VW_TPROJ, D, D, d_tr, d_tctx, nullptr, 0));
    BCHK(dvt.bwd(sv + L.qkv_t, sv + L.tctx, (float*)const_cast<char*>(sv + L.lse_t), d_tctx, dqkv_t, delta_t, aws, awb, st));
    BCHK(wgrad(3 * D, D, dqkv_t, sv + L.h3, VW_TQKV, nullptr, M));
    BCHK(dgrad(VW_TQKV, 3 * D, D, dqkv_t, dh3, nullptr, 0));
    // dx = LN3'(dh3) + d_sr + d_tr: x feeds norm3, the time residual and the space residual
    // (EGV_BLOCK_HEAD: the space residual's gradient reaches x through the caller's own graph, not through this call)
    BCHK(egv_layernorm_bwd2(dt, dh3, d->x, (const float*)(sv + L.stats3), d->ln_g[VL_NORM3], head ? nullptr : d_sr, d_tr, d->dx, d->dln_g[VL_NORM3],
                            d->dln_b[VL_NORM3], M, D, lnw, st));
    BCHK(ln_defer.flush(st));
    const bool tail = side_group && (d->flags & EGV_BLOCK_TAIL);      // no data-gradient chain follows: the launch may have the chip
    static const int tail_cus = egv_cfg_int("EGV_WGRAD_TAIL_CUS", 0);  // grant of the tail launch (0: 7/8 of the CUs)
    const int free_cus = (tail && tail_cus > 0) ? tail_cus : (device_cus() * 7) / 8;
    if (ngrp) BCHK(egv_gemm_wgrad_grouped(dt, M, ngrp, grp, (side_group && !tail) ? vgroup_cus(d) : free_cus, wgw, wgb, fk.begin()));
    if (!side_group) fk.join();
    return 0;
}

// =====================================================================================================================
// RobertaLayer
// =====================================================================================================================
namespace {
enum { TW_Q = 0, TW_K, TW_V, TW_AO, TW_FC1, TW_FC2, TW_CQ, TW_CK, TW_CV, TW_CO };
enum { TL_ATT = 0, TL_OUT };

struct TLayout {
    size_t q, k, v, ctx, lse, a0, a0d, cq, ck, cv, cctx, cctx32, lse_c, pg, a_pre, stats0, a, pre, act, f0, f_pre, stats1, hid16, a16;
    size_t total;
};

// EGV_BLOCK_RES_F32 (bf16 mode only): the residual stream -- layer input / output, the two pre-LayerNorm sums and the output of the
// attention LayerNorm -- is fp32, GEMM operands and outputs stay bf16 (what torch.autocast does, trainer/trainer_egoclip.py:143)
// merged same-input projections (bf16 mode, pointers supplied by the caller)
inline bool tmq(const egv_tlayer_desc* d) { return d->dtype == EGV_BF16 && d->w_qkv != nullptr; }
inline bool tmc(const egv_tlayer_desc* d) { return d->dtype == EGV_BF16 && d->S > 0 && d->w_ckv != nullptr; }
// the weight gradients over the B*L text rows as ONE grouped launch (egv_gemm5.hip) at the end of the backward call
inline bool tgroup(const egv_tlayer_desc* d) {
    static const bool on = egv_cfg_on("EGV_TEXT_WGRAD_GROUP", true);
    return on && d->dtype == EGV_BF16 && (d->D % 256) == 0 && (d->Hd % 256) == 0 && d->B * d->L >= 64;
}
inline bool tres32(const egv_tlayer_desc* d) { return (d->flags & EGV_BLOCK_RES_F32) && d->dtype == EGV_BF16; }

TLayout tlayout(const egv_tlayer_desc* d) {
    TLayout L{};
    Bump b(nullptr, 0);
    const size_t es = esz(d->dtype);
    const size_t rs = tres32(d) ? 4 : es;                       // element size of the residual-stream tensors
    const size_t BL = (size_t)d->B * d->L, BS = (size_t)d->B * d->S, D = d->D, Hd = d->Hd, H = d->H;
    const bool fused = d->S > 0, drop = d->drop_p > 0.f;
    auto T = [&](size_t n) { return b.take_off(n); };
    L.q = T(BL * 3 * D * es); L.k = L.q + BL * D * es; L.v = L.q + 2 * BL * D * es;     // one [BL, 3D] matrix when the projections are merged
    L.ctx = T(BL * D * es); L.lse = T(BL * H * 4);
    if (fused || drop || tres32(d)) L.a0 = T(BL * D * es);
    if (fused && drop) L.a0d = T(BL * D * es);
    if (fused) {
        L.cq = T(BL * D * es); L.ck = T(BS * 2 * D * es); L.cv = L.ck + BS * D * es; L.cctx = T(BL * D * es); L.cctx32 = T(BL * D * 4); L.lse_c = T(BL * H * 4);
        L.pg = T(BL * D * es);
    }
    L.a_pre = T(BL * D * rs); L.stats0 = T(BL * 8); L.a = T(BL * D * rs); L.pre = T(BL * Hd * es); L.act = T(BL * Hd * es);
    if (drop || tres32(d)) L.f0 = T(BL * D * es);
    L.f_pre = T(BL * D * rs); L.stats1 = T(BL * 8);
    if (tres32(d)) { L.hid16 = T(BL * D * es); L.a16 = T(BL * D * es); }
    L.total = al(b.off);
    return L;
}

struct TPlan {               // attention problems of one layer
    Plain self, cross;
    long long awb;
};
TPlan tplan(const egv_tlayer_desc* d) {
    TPlan p{};
    const bool drop = d->drop_p > 0.f;
    p.self = Plain{d->dtype, d->B, d->H, d->D, d->L, d->L, 0.125f, drop ? d->drop_p : 0.f, d->seeds[0], d->mask};
    p.cross = Plain{d->dtype, d->B, d->H, d->D, d->L, d->S, 0.125f, drop ? d->drop_p : 0.f, d->seeds[2], nullptr};
    p.awb = p.self.ws_bytes();
    if (d->S > 0 && p.cross.ws_bytes() > p.awb) p.awb = p.cross.ws_bytes();
    return p;
}
}  // namespace

namespace {
long long tgroup_ws_bytes(const egv_tlayer_desc* d) {
    if (!tgroup(d)) return 0;
    const int D = d->D, Hd = d->Hd;
    egv_wgrad_problem pr[8];
    int n = 0;
    auto add = [&](int N, int K) { pr[n] = egv_wgrad_problem{}; pr[n].N = N; pr[n].K = K; ++n; };
    add(D, Hd); add(Hd, D); add(D, D);
    if (tmq(d)) add(3 * D, D); else { add(D, D); add(D, D); add(D, D); }
    if (d->S > 0) { add(D, D); add(D, D); }
    const long long b = egv_gemm_wgrad_grouped_workspace_bytes(d->B * d->L, n, pr, 0);
    return b > 0 ? b : 0;
}
}  // namespace

extern "C" long long egv_tlayer_save_bytes(const egv_tlayer_desc* d) { return (long long)tlayout(d).total; }

extern "C" long long egv_tlayer_ws_bytes(const egv_tlayer_desc* d, int backward) {
    const size_t es = esz(d->dtype);
    const size_t BL = (size_t)d->B * d->L, BS = (size_t)d->B * d->S, D = d->D, Hd = d->Hd, H = d->H;
    size_t tot = al((size_t)tplan(d).awb) + al(BL * D * es) + 4096;
    if (backward) {
        long long wg = 0;
        auto mx = [&](long long v) { if (v > wg) wg = v; };
        mx(egv_gemm_wgrad_workspace_bytes((int)D, (int)D, (int)BL)); mx(egv_gemm_wgrad_workspace_bytes((int)Hd, (int)D, (int)BL));
        mx(egv_gemm_wgrad_workspace_bytes((int)D, (int)Hd, (int)BL));
        if (d->S > 0) mx(egv_gemm_wgrad_workspace_bytes(2 * (int)D, (int)D, (int)BS));
        tot += al((size_t)tgroup_ws_bytes(d)) + 256;
        tot += al((size_t)wg) + al((size_t)egv_layernorm_bwd_workspace_bytes((int)BL, (int)D));
        tot += 17 * al(BL * D * es) + 3 * al(BL * D * 4) + al(BL * Hd * es) + 2 * al(BL * H * 4) + 4096;
        if (d->S > 0) tot += 3 * al(BS * D * es) + 4096;
    }
    return (long long)tot + 65536;
}

extern "C" int egv_tlayer_fwd(const egv_tlayer_desc* d) {
    const int dt = d->dtype;
    const int BL = d->B * d->L, BS = d->B * d->S, D = d->D, Hd = d->Hd;
    const bool fused = d->S > 0, drop = d->drop_p > 0.f;
    const float p = d->drop_p;
    if (!(d->hid && d->out && d->save && d->ws)) { egv_set_error("egv_tlayer_fwd: null buffer"); return -1; }
    const TLayout L = tlayout(d);
    if ((long long)L.total > d->save_bytes) { egv_set_error("egv_tlayer_fwd: save buffer too small"); return -1; }
    char* sv = (char*)d->save;
    void* st = d->stream;
    const TPlan tp = tplan(d);
    Bump ws(d->ws, d->ws_bytes);
    void* aws = ws.take((size_t)tp.awb);
    if (!ws.ok()) { egv_set_error("egv_tlayer_fwd: workspace too small"); return -1; }
    const long long n = (long long)BL * D;

    const bool r32 = tres32(d);
    // fp32 residual stream: the Linears read a bf16 copy of the (fp32) layer input; every residual sum is formed in fp32 by the
    // mixed-type dropout/add kernel (p = 0: a plain add) instead of a GEMM epilogue
    const void* hid_op = d->hid;
    if (r32) {
        BCHK(egv_cast(EGV_F32, EGV_BF16, d->hid, sv + L.hid16, n, st));
        hid_op = sv + L.hid16;
    }
    const bool mq = tmq(d), mc = tmc(d);
    const size_t es_ = esz(dt);
    char* qp = sv + L.q;
    char* kp = mq ? qp + (size_t)D * es_ : sv + L.k;
    char* vp = mq ? qp + (size_t)2 * D * es_ : sv + L.v;
    const int ldqkv = mq ? 3 * D : D;
    if (mq) {
        BCHK(lin_fwd(dt, BL, 3 * D, D, hid_op, d->w_qkv, d->b_qkv, qp, 0, nullptr, nullptr, nullptr, nullptr, st));
    } else {
        BCHK(lin_fwd(dt, BL, D, D, hid_op, d->w[TW_Q], d->b[TW_Q], qp, 0, nullptr, nullptr, nullptr, nullptr, st));
        BCHK(lin_fwd(dt, BL, D, D, hid_op, d->w[TW_K], d->b[TW_K], kp, 0, nullptr, nullptr, nullptr, nullptr, st));
        BCHK(lin_fwd(dt, BL, D, D, hid_op, d->w[TW_V], d->b[TW_V], vp, 0, nullptr, nullptr, nullptr, nullptr, st));
    }
    BCHK(tp.self.fwd(qp, ldqkv, kp, vp, ldqkv, sv + L.ctx, (float*)(sv + L.lse), aws, tp.awb, st));
    if (!fused) {
        if (r32) {
            BCHK(lin_fwd(dt, BL, D, D, sv + L.ctx, d->w[TW_AO], d->b[TW_AO], sv + L.a0, 0, nullptr, nullptr, nullptr, nullptr, st));
            BCHK(egv_dropout_add_mixed(EGV_BF16, sv + L.a0, nullptr, (const float*)d->hid, EGV_F32, sv + L.a_pre, n, drop ? p : 0.f, d->seeds[1], st));
        } else if (!drop) {
            BCHK(lin_fwd(dt, BL, D, D, sv + L.ctx, d->w[TW_AO], d->b[TW_AO], sv + L.a_pre, 0, nullptr, d->hid, nullptr, nullptr, st));   // dense(ctx) + hidden
        } else {
            BCHK(lin_fwd(dt, BL, D, D, sv + L.ctx, d->w[TW_AO], d->b[TW_AO], sv + L.a0, 0, nullptr, nullptr, nullptr, nullptr, st));
            BCHK(egv_dropout_add(dt, sv + L.a0, d->hid, nullptr, sv + L.a_pre, n, p, d->seeds[1], st));
        }
    } else {
        BCHK(lin_fwd(dt, BL, D, D, sv + L.ctx, d->w[TW_AO], d->b[TW_AO], sv + L.a0, 0, nullptr, nullptr, nullptr, nullptr, st));
        const char* a0x = sv + L.a0;
        if (drop) {
            BCHK(egv_dropout_add(dt, sv + L.a0, nullptr, nullptr, sv + L.a0d, n, p, d->seeds[1], st));
            a0x = sv + L.a0d;
        }
        BCHK(lin_fwd(dt, BL, D, D, a0x, d->w[TW_CQ], d->b[TW_CQ], sv + L.cq, 0, nullptr, nullptr, nullptr, nullptr, st));
        char* ckp = sv + L.ck;
        char* cvp = mc ? ckp + (size_t)D * es_ : sv + L.cv;
        if (mc) {
            BCHK(lin_fwd(dt, BS, 2 * D, D, d->enc, d->w_ckv, d->b_ckv, ckp, 0, nullptr, nullptr, nullptr, nullptr, st));
        } else {
            BCHK(lin_fwd(dt, BS, D, D, d->enc, d->w[TW_CK], d->b[TW_CK], ckp, 0, nullptr, nullptr, nullptr, nullptr, st));
            BCHK(lin_fwd(dt, BS, D, D, d->enc, d->w[TW_CV], d->b[TW_CV], cvp, 0, nullptr, nullptr, nullptr, nullptr, st));
        }
        BCHK(tp.cross.fwd(sv + L.cq, D, ckp, cvp, mc ? 2 * D : D, sv + L.cctx, (float*)(sv + L.lse_c), aws, tp.awb, st, (float*)(sv + L.cctx32)));
        if (!drop && !r32) {
            // alpha_t2i * dense(cctx) + a0 + hidden (roberta.py:486-488)
            BCHK(lin_fwd(dt, BL, D, D, sv + L.cctx, d->w[TW_CO], d->b[TW_CO], sv + L.a_pre, 0, d->alpha, a0x, d->hid, sv + L.pg, st));
        } else {
            // the gate commutes with the keep mask: y = alpha * dense(cctx) in the GEMM epilogue (pre-gate value saved), then dropout + adds
            void* y = ws.take((size_t)n * esz(dt));
            if (!ws.ok()) { egv_set_error("egv_tlayer_fwd: workspace too small"); return -1; }
            BCHK(lin_fwd(dt, BL, D, D, sv + L.cctx, d->w[TW_CO], d->b[TW_CO], y, 0, d->alpha, nullptr, nullptr, sv + L.pg, st));
            if (r32) BCHK(egv_dropout_add_mixed(EGV_BF16, y, a0x, (const float*)d->hid, EGV_F32, sv + L.a_pre, n, drop ? p : 0.f, d->seeds[3], st));
            else BCHK(egv_dropout_add(dt, y, a0x, d->hid, sv + L.a_pre, n, p, d->seeds[3], st));
        }
    }
    const void* a_op = sv + L.a;
    if (r32) {
        BCHK(egv_layernorm_fwd_res32((const float*)(sv + L.a_pre), (float*)(sv + L.a), sv + L.a16, d->ln_g[TL_ATT], d->ln_b[TL_ATT], (float*)(sv + L.stats0), BL, D, d->eps, st));
        a_op = sv + L.a16;
    } else {
        BCHK(egv_layernorm_fwd(dt, sv + L.a_pre, sv + L.a, d->ln_g[TL_ATT], d->ln_b[TL_ATT], (float*)(sv + L.stats0), BL, D, d->eps, st));
    }
    BCHK(lin_fwd(dt, BL, Hd, D, a_op, d->w[TW_FC1], d->b[TW_FC1], sv + L.act, EGV_ACT_GELU, nullptr, nullptr, nullptr, sv + L.pre, st));
    if (r32) {
        BCHK(lin_fwd(dt, BL, D, Hd, sv + L.act, d->w[TW_FC2], d->b[TW_FC2], sv + L.f0, 0, nullptr, nullptr, nullptr, nullptr, st));
        BCHK(egv_dropout_add_mixed(EGV_BF16, sv + L.f0, nullptr, (const float*)(sv + L.a), EGV_F32, sv + L.f_pre, n, drop ? p : 0.f, d->seeds[4], st));
        return egv_layernorm_fwd_res32((const float*)(sv + L.f_pre), (float*)d->out, nullptr, d->ln_g[TL_OUT], d->ln_b[TL_OUT], (float*)(sv + L.stats1), BL, D, d->eps, st);
    }
    if (!drop) {
        BCHK(lin_fwd(dt, BL, D, Hd, sv + L.act, d->w[TW_FC2], d->b[TW_FC2], sv + L.f_pre, 0, nullptr, sv + L.a, nullptr, nullptr, st));
    } else {
        BCHK(lin_fwd(dt, BL, D, Hd, sv + L.act, d->w[TW_FC2], d->b[TW_FC2], sv + L.f0, 0, nullptr, nullptr, nullptr, nullptr, st));
        BCHK(egv_dropout_add(dt, sv + L.f0, sv + L.a, nullptr, sv + L.f_pre, n, p, d->seeds[4], st));
    }
    BCHK(egv_layernorm_fwd(dt, sv + L.f_pre, d->out, d->ln_g[TL_OUT], d->ln_b[TL_OUT], (float*)(sv + L.stats1), BL, D, d->eps, st));
    return 0;
}

// as egv_vblock_bwd_groups: the gradients over the B*L text rows (a merged q | k | v gradient is slot TW_Q's problem: bits 1, 2 follow bit 0)
extern "C" unsigned int egv_tlayer_bwd_groups(const egv_tlayer_desc* d) {
    if (!tgroup(d)) return 0u;
    unsigned int m = (1u << TW_Q) | (1u << TW_K) | (1u << TW_V) | (1u << TW_AO) | (1u << TW_FC1) | (1u << TW_FC2);
    if (d->S > 0) m |= (1u << TW_CQ) | (1u << TW_CO);
    return m;
}

extern "C" int egv_tlayer_bwd(const egv_tlayer_desc* d) {
    const int dt = d->dtype;
    const size_t es = esz(dt);
    const int BL = d->B * d->L, BS = d->B * d->S, D = d->D, Hd = d->Hd, H = d->H;
    const bool fused = d->S > 0, drop = d->drop_p > 0.f;
    const float p = d->drop_p;
    if (!(d->hid && d->dout && d->dhid && d->save && d->ws)) { egv_set_error("egv_tlayer_bwd: null buffer"); return -1; }
    const TLayout L = tlayout(d);
    const char* sv = (const char*)d->save;
    void* st = d->stream;
    Fork fk(d->stream, d->stream2, fused ? BS : BL);
    const TPlan tp = tplan(d);
    Bump ws(d->ws, d->ws_bytes);
    void* aws = ws.take((size_t)tp.awb);
    long long wgb = 0;
    {
        auto mx = [&](long long v) { if (v > wgb) wgb = v; };
        mx(egv_gemm_wgrad_workspace_bytes(D, D, BL)); mx(egv_gemm_wgrad_workspace_bytes(Hd, D, BL)); mx(egv_gemm_wgrad_workspace_bytes(D, Hd, BL));
        if (fused) mx(egv_gemm_wgrad_workspace_bytes(2 * D, D, BS));
    }
    void* wgw = ws.take((size_t)wgb);
    void* lnw = ws.take((size_t)egv_layernorm_bwd_workspace_bytes(BL, D));
    const size_t nb = (size_t)BL * D * es;
    void* df_pre = ws.take(nb); void* df0 = ws.take(nb); void* dpre = ws.take((size_t)BL * Hd * es); void* da = ws.take(nb);
    void* da_pre = ws.take(nb); void* da0 = ws.take(nb); void* dctx = ws.take(nb);
    const bool mq = tmq(d), mc = tmc(d);
    char* dqkv = (char*)ws.take(3 * nb);                          // [BL, 3D] when the projections are merged, else three [BL, D] blocks
    void *dq = dqkv, *dk = mq ? dqkv + (size_t)D * es : dqkv + nb, *dv = mq ? dqkv + (size_t)2 * D * es : dqkv + 2 * nb;
    const int lddqkv = mq ? 3 * D : D;
    void* t1 = ws.take(nb); void* t2 = ws.take(nb);
    float* delta = (float*)ws.take((size_t)BL * H * 4);
    void *dyg = nullptr, *dcctx = nullptr, *dcq = nullptr, *dck = nullptr, *dcv = nullptr, *da0d = nullptr, *te = nullptr;
    float* delta_c = nullptr;
    if (fused) {
        dyg = ws.take(nb); dcctx = ws.take(nb); dcq = ws.take(nb); da0d = ws.take(nb);
        char* dckv = (char*)ws.take((size_t)2 * BS * D * es);
        dck = dckv; dcv = mc ? dckv + (size_t)D * es : dckv + (size_t)BS * D * es;
        te = ws.take((size_t)BS * D * es);
        delta_c = (float*)ws.take((size_t)BL * H * 4);
    }
    void* dotw = ws.take(4096);
    if (!ws.ok()) { egv_set_error("egv_tlayer_bwd: workspace too small (%zu > %zu)", ws.off, ws.cap); return -1; }
    const long long n = (long long)BL * D;
    // dx = dz W + res (dgrad whose output also receives a skip gradient)
    auto dgrad_res = [&](int rows, int N, int K, const void* dz, int w, void* dx, const void* res, const float* gate) -> int {
        if (d->wt[w]) return egv_gemm(dt, 0, 0, rows, K, N, dz, N, d->wt[w], N, dx, K, 0, nullptr, 0, gate, res, nullptr, nullptr, nullptr, 0, K, 1.0f, st);
        return egv_gemm(dt, 0, 1, rows, K, N, dz, N, d->w[w], K, dx, K, 0, nullptr, 0, gate, res, nullptr, nullptr, nullptr, 0, K, 1.0f, st);
    };

    // weight gradients over the B*L text rows: collected and launched together at the end of the call (ONE persistent grouped
    // launch on this stream instead of 4-6 launches of 20 us that use a dozen CUs each); those over the B*S video rows of a fused
    // layer keep their own launches on the companion stream
    const bool grp_on = tgroup(d);
    egv_wgrad_problem grp[8];
    int ngrp = 0;
    const long long gwb = grp_on ? tgroup_ws_bytes(d) : 0;
    void* gws = grp_on ? ws.take((size_t)gwb) : nullptr;
    if (!ws.ok()) { egv_set_error("egv_tlayer_bwd: workspace too small (%zu > %zu)", ws.off, ws.cap); return -1; }
    auto wgrad_ld = [&](int rows, int N, int K, const void* dz, int ldz, const void* x, int w, const float* gate) -> int {
        if (grp_on && rows == BL) {
            egv_wgrad_problem& q = grp[ngrp++];
            q.dy = dz; q.ldy = ldz; q.x = x; q.ldx = K; q.dw = d->dw[w]; q.db = d->db[w]; q.gate = gate; q.N = N; q.K = K;
            q.accumulate = (d->acc_mask >> w) & 1;
            return 0;
        }
        if ((d->acc_mask >> w) & 1) { egv_set_error("egv_tlayer_bwd: acc_mask bit %d set for a weight gradient outside the grouped launch", w); return -1; }
        return lin_wgrad(dt, rows, N, K, dz, ldz, x, d->dw[w], d->db[w], gate, wgw, wgb, fk.begin());
    };
    auto wgrad = [&](int rows, int N, int K, const void* dz, const void* x, int w, const float* gate) -> int {
        return wgrad_ld(rows, N, K, dz, N, x, w, gate);
    };
    const int ldckv = mc ? 2 * D : D;
    const char* qp = sv + L.q;
    const char* kp = mq ? qp + (size_t)D * es : sv + L.k;
    const char* vp = mq ? qp + (size_t)2 * D * es : sv + L.v;
    const int ldqkv = mq ? 3 * D : D;
    const char* ckp = fused ? sv + L.ck : nullptr;
    const char* cvp = fused ? (mc ? ckp + (size_t)D * es : sv + L.cv) : nullptr;
    // gradients of the q / k / v projections: weight gradients (one [3D, D] problem when merged) and the data gradient
    // out = dq Wq + dk Wk + dv Wv (+ res), ONE GEMM with K = 3D when merged
    auto qkv_grads = [&](const void* hid_op, void* out, const void* res) -> int {
        if (mq) {
            BCHK(wgrad_ld(BL, 3 * D, D, dqkv, 3 * D, hid_op, TW_Q, nullptr));
            if (d->wt_qkv) return egv_gemm(dt, 0, 0, BL, D, 3 * D, dqkv, 3 * D, d->wt_qkv, 3 * D, out, D, 0, nullptr, 0, nullptr, res, nullptr, nullptr, nullptr, 0, D, 1.0f, st);
            return egv_gemm(dt, 0, 1, BL, D, 3 * D, dqkv, 3 * D, d->w_qkv, D, out, D, 0, nullptr, 0, nullptr, res, nullptr, nullptr, nullptr, 0, D, 1.0f, st);
        }
        BCHK(wgrad(BL, D, D, dq, hid_op, TW_Q, nullptr));
        BCHK(wgrad(BL, D, D, dk, hid_op, TW_K, nullptr));
        BCHK(wgrad(BL, D, D, dv, hid_op, TW_V, nullptr));
        auto one = [&](const void* dz, int w, void* dx, const void* r) -> int {
            if (d->wt[w]) return egv_gemm(dt, 0, 0, BL, D, D, dz, D, d->wt[w], D, dx, D, 0, nullptr, 0, nullptr, r, nullptr, nullptr, nullptr, 0, D, 1.0f, st);
            return egv_gemm(dt, 0, 1, BL, D, D, dz, D, d->w[w], D, dx, D, 0, nullptr, 0, nullptr, r, nullptr, nullptr, nullptr, 0, D, 1.0f, st);
        };
        BCHK(one(dq, TW_Q, t1, res));
        BCHK(one(dk, TW_K, t2, t1));
        return one(dv, TW_V, out, t2);
    };
    // gradients of the text-to-image key / value projections over the video tokens
    auto ckv_grads = [&]() -> int {
        if (mc) {
            BCHK(lin_wgrad(dt, BS, 2 * D, D, dck, 2 * D, d->enc, d->dw[TW_CK], d->db[TW_CK], nullptr, wgw, wgb, fk.begin()));
            if (!d->denc) return 0;
            if (d->wt_ckv) return egv_gemm(dt, 0, 0, BS, D, 2 * D, dck, 2 * D, d->wt_ckv, 2 * D, d->denc, D, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, D, 1.0f, st);
            return egv_gemm(dt, 0, 1, BS, D, 2 * D, dck, 2 * D, d->w_ckv, D, d->denc, D, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, D, 1.0f, st);
        }
        BCHK(lin_wgrad(dt, BS, D, D, dck, D, d->enc, d->dw[TW_CK], d->db[TW_CK], nullptr, wgw, wgb, fk.begin()));
        BCHK(lin_wgrad(dt, BS, D, D, dcv, D, d->enc, d->dw[TW_CV], d->db[TW_CV], nullptr, wgw, wgb, fk.begin()));
        if (d->denc) {
            BCHK(lin_dgrad(dt, BS, D, D, dck, d->w[TW_CK], d->wt[TW_CK], te, nullptr, nullptr, 0, st));
            BCHK(dgrad_res(BS, D, D, dcv, TW_CV, d->denc, te, nullptr));
        }
        return 0;
    };
    auto finish = [&]() -> int {
        if (ngrp) BCHK(egv_gemm_wgrad_grouped(dt, BL, ngrp, grp, 0, gws, gwb, st));
        fk.join();
        return 0;
    };
    const bool r32 = tres32(d);
    if (r32) {
        // ---- fp32 residual stream: the gradients of the fp32 tensors (layer output, the two pre-LayerNorm sums, the attention
        // LayerNorm output, the layer input) are fp32; every GEMM operand is bf16.  The residual-path gradient joins a GEMM's
        // bf16 data gradient inside the LayerNorm backward (dy16 + dy32) or in a mixed-type add, never in a GEMM epilogue.
        const size_t nb4 = (size_t)BL * D * 4;
        float* df_pre32 = (float*)ws.take(nb4);
        float* da_pre32 = (float*)ws.take(nb4);
        float* s32 = (float*)ws.take(nb4);
        if (!ws.ok()) { egv_set_error("egv_tlayer_bwd: workspace too small (%zu > %zu)", ws.off, ws.cap); return -1; }
        const float pd = drop ? p : 0.f;
        // out = LN(f_pre): dout is fp32
        BCHK(egv_layernorm_bwd_res32(nullptr, (const float*)d->dout, (const float*)(sv + L.f_pre), (const float*)(sv + L.stats1), d->ln_g[TL_OUT], nullptr,
                                     df_pre32, d->dln_g[TL_OUT], d->dln_b[TL_OUT], BL, D, lnw, st));
        // f_pre = dropout(fc2(act)) + a
        BCHK(egv_dropout_add_mixed(EGV_F32, df_pre32, nullptr, nullptr, EGV_BF16, df0, n, pd, d->seeds[4], st));
        BCHK(wgrad(BL, D, Hd, df0, sv + L.act, TW_FC2, nullptr));
        BCHK(lin_dgrad(dt, BL, D, Hd, df0, d->w[TW_FC2], d->wt[TW_FC2], dpre, nullptr, sv + L.pre, EGV_ACT_GELU, st));
        BCHK(wgrad(BL, Hd, D, dpre, sv + L.a16, TW_FC1, nullptr));
        BCHK(lin_dgrad(dt, BL, Hd, D, dpre, d->w[TW_FC1], d->wt[TW_FC1], da, nullptr, nullptr, 0, st));
        // a = LN(a_pre); its gradient = fc1's data gradient (bf16) + the residual path df_pre (fp32)
        BCHK(egv_layernorm_bwd_res32(da, df_pre32, (const float*)(sv + L.a_pre), (const float*)(sv + L.stats0), d->ln_g[TL_ATT], nullptr, da_pre32,
                                     d->dln_g[TL_ATT], d->dln_b[TL_ATT], BL, D, lnw, st));
        const void* d_ao = da0;
        if (!fused) {
            BCHK(egv_dropout_add_mixed(EGV_F32, da_pre32, nullptr, nullptr, EGV_BF16, da0, n, pd, d->seeds[1], st));
        } else {
            BCHK(egv_dropout_add_mixed(EGV_F32, da_pre32, nullptr, nullptr, EGV_BF16, dyg, n, pd, d->seeds[3], st));
            const char* a0x = drop ? sv + L.a0d : sv + L.a0;
            BCHK(egv_dot(dt, dyg, sv + L.pg, n, d->dalpha, 1.0f, dotw, st));
            BCHK(wgrad(BL, D, D, dyg, sv + L.cctx, TW_CO, d->alpha));
            BCHK(lin_dgrad(dt, BL, D, D, dyg, d->w[TW_CO], d->wt[TW_CO], dcctx, d->alpha, nullptr, 0, st));
            BCHK(tp.cross.bwd(sv + L.cq, D, ckp, cvp, ldckv, sv + L.cctx, (float*)const_cast<char*>(sv + L.lse_c), dcctx, dcq, D, dck, dcv, ldckv, delta_c,
                              aws, tp.awb, st, (const float*)(sv + L.cctx32)));
            BCHK(wgrad(BL, D, D, dcq, a0x, TW_CQ, nullptr));
            BCHK(lin_dgrad(dt, BL, D, D, dcq, d->w[TW_CQ], d->wt[TW_CQ], da0d, nullptr, nullptr, 0, st));
            BCHK(ckv_grads());
            // a0 (after its dropout) feeds the cross-attention query AND the residual: (bf16 data gradient + fp32 da_pre), then the
            // dropout of attention.output
            BCHK(egv_dropout_add_mixed(EGV_BF16, da0d, nullptr, da_pre32, EGV_F32, s32, n, 0.f, 0u, st));
            BCHK(egv_dropout_add_mixed(EGV_F32, s32, nullptr, nullptr, EGV_BF16, da0, n, pd, d->seeds[1], st));
        }
        BCHK(wgrad(BL, D, D, d_ao, sv + L.ctx, TW_AO, nullptr));
        BCHK(lin_dgrad(dt, BL, D, D, d_ao, d->w[TW_AO], d->wt[TW_AO], dctx, nullptr, nullptr, 0, st));
        BCHK(tp.self.bwd(qp, ldqkv, kp, vp, ldqkv, sv + L.ctx, (float*)const_cast<char*>(sv + L.lse), dctx, dq, lddqkv, dk, dv, lddqkv, delta, aws, tp.awb, st));
        BCHK(qkv_grads(sv + L.hid16, dctx, nullptr));         // (dctx is free again: the bf16 sum of the three data gradients)
        // dhid (fp32) = data gradients of q / k / v (bf16 sum) + the residual path
        BCHK(egv_dropout_add_mixed(EGV_BF16, dctx, nullptr, da_pre32, EGV_F32, d->dhid, n, 0.f, 0u, st));
        return finish();
    }
    // out = LN(f_pre)
    BCHK(egv_layernorm_bwd2(dt, d->dout, sv + L.f_pre, (const float*)(sv + L.stats1), d->ln_g[TL_OUT], nullptr, nullptr, df_pre, d->dln_g[TL_OUT],
                            d->dln_b[TL_OUT], BL, D, lnw, st));
    // f_pre = [dropout](fc2(act)) + a
    const void* dfc2 = df_pre;
    if (drop) {
        BCHK(egv_dropout_add(dt, df_pre, nullptr, nullptr, df0, n, p, d->seeds[4], st));
        dfc2 = df0;
    }
    BCHK(wgrad(BL, D, Hd, dfc2, sv + L.act, TW_FC2, nullptr));
    BCHK(lin_dgrad(dt, BL, D, Hd, dfc2, d->w[TW_FC2], d->wt[TW_FC2], dpre, nullptr, sv + L.pre, EGV_ACT_GELU, st));
    BCHK(wgrad(BL, Hd, D, dpre, sv + L.a, TW_FC1, nullptr));
    BCHK(dgrad_res(BL, Hd, D, dpre, TW_FC1, da, df_pre, nullptr));                         // + skip gradient of the residual a
    BCHK(egv_layernorm_bwd2(dt, da, sv + L.a_pre, (const float*)(sv + L.stats0), d->ln_g[TL_ATT], nullptr, nullptr, da_pre, d->dln_g[TL_ATT],
                            d->dln_b[TL_ATT], BL, D, lnw, st));
    // a_pre = [dropout](...) + hid  (+ a0 in the fused form): da_pre flows to hid unchanged
    const void* d_ao = da_pre;                                                             // gradient at the output of attention.output.dense
    if (!fused) {
        if (drop) {
            BCHK(egv_dropout_add(dt, da_pre, nullptr, nullptr, da0, n, p, d->seeds[1], st));
            d_ao = da0;
        }
    } else {
        const void* dy_ = da_pre;
        if (drop) {
            BCHK(egv_dropout_add(dt, da_pre, nullptr, nullptr, dyg, n, p, d->seeds[3], st));
            dy_ = dyg;
        }
        const char* a0x = drop ? sv + L.a0d : sv + L.a0;
        BCHK(egv_dot(dt, dy_, sv + L.pg, n, d->dalpha, 1.0f, dotw, st));
        BCHK(wgrad(BL, D, D, dy_, sv + L.cctx, TW_CO, d->alpha));
        BCHK(lin_dgrad(dt, BL, D, D, dy_, d->w[TW_CO], d->wt[TW_CO], dcctx, d->alpha, nullptr, 0, st));
        BCHK(tp.cross.bwd(sv + L.cq, D, ckp, cvp, ldckv, sv + L.cctx, (float*)const_cast<char*>(sv + L.lse_c), dcctx, dcq, D, dck, dcv, ldckv, delta_c,
                          aws, tp.awb, st, (const float*)(sv + L.cctx32)));
        BCHK(wgrad(BL, D, D, dcq, a0x, TW_CQ, nullptr));
        BCHK(dgrad_res(BL, D, D, dcq, TW_CQ, da0d, da_pre, nullptr));                       // a0 also feeds the residual
        BCHK(ckv_grads());
        d_ao = da0d;
        if (drop) {
            BCHK(egv_dropout_add(dt, da0d, nullptr, nullptr, da0, n, p, d->seeds[1], st));
            d_ao = da0;
        }
    }
    BCHK(wgrad(BL, D, D, d_ao, sv + L.ctx, TW_AO, nullptr));
    BCHK(lin_dgrad(dt, BL, D, D, d_ao, d->w[TW_AO], d->wt[TW_AO], dctx, nullptr, nullptr, 0, st));
    BCHK(tp.self.bwd(qp, ldqkv, kp, vp, ldqkv, sv + L.ctx, (float*)const_cast<char*>(sv + L.lse), dctx, dq, lddqkv, dk, dv, lddqkv, delta, aws, tp.awb, st));
    BCHK(qkv_grads(d->hid, d->dhid, da_pre));
    return finish();
}
