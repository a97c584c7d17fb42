/* libegovlp_hip.so -- C ABI of the MI355X (gfx950) kernels behind the EgoVLPv2 pre-training hot path.
 *
 * The reference (facebookresearch/EgoVLPv2) has no FFI: its hot path is a composition of stock ATen ops
 * inside Python nn.Modules.  Each entry point below replaces one such
 * composition; the reference lines it replaces are cited per function (paths relative to
 * EgoVLPv2/ in the reference tree).  The Python host code in egovlpv2_amd/ binds these with ctypes
 * (egovlpv2_amd/_lib.py) and mirrors the reference module API (FrozenInTime.forward()/infer(),
 * sim_matrix, EgoNCE, AllGather_multi) on top of them.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless stated otherwise; the caller owns all memory, outputs and
 *    workspaces are caller-allocated (sizes from the *_workspace_bytes queries), nothing is allocated,
 *    freed or synchronised inside the library;
 *  - `stream` is a hipStream_t (as void*); all work is enqueued on it, stream-ordered;
 *  - `dtype` selects the STORAGE type of activations: EGV_F32 (exact fp32 MFMA path, used for the 1e-3
 *    parity gate) or EGV_BF16 (bf16 MFMA path, the throughput configuration); arithmetic accumulates in
 *    fp32 in both; parameters, biases, gates, LayerNorm affine terms, statistics and parameter gradients
 *    are always fp32;
 *  - matrices are row-major with an explicit leading dimension in ELEMENTS; activation pointers must be
 *    16-byte aligned and leading dims multiples of 4 elements for the vector paths (scalar fallbacks
 *    exist in the GEMM for ragged shapes);
 *  - return value: 0 on success, negative on error, message via egv_last_error(); errors are argument
 *    errors detected on the host or launch failures -- there is no CPU fallback of any kind.
 */
#ifndef EGOVLP_HIP_H
#define EGOVLP_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define EGV_F32 0
#define EGV_BF16 1

#define EGV_ACT_NONE 0
#define EGV_ACT_GELU 1 /* exact erf GELU: nn.GELU video_transformer.py:43, ACT2FN["gelu"] roberta.py:402 */
#define EGV_ACT_RELU 2 /* txt_proj / vid_proj, model.py:105-115 */
#define EGV_ACT_TANH 3 /* heads.Pooler, heads.py:15-25 */
#define EGV_ACT_GELU_D 4 /* GELU whose saved tensor is the DERIVATIVE gelu'(x) instead of x: as act (with pre) the forward GEMM stores
                           gelu'(acc + bias) in pre; as dact the data-gradient GEMM multiplies by aux as it stands.  Same bytes, but
                           the backward epilogue (video_transformer.py:43 under autograd) has no erf / exp left to compute */

int egv_abi_version(void);
const char* egv_last_error(void);
/* run-time switches of the library: one line "NAME\tdefault\tcurrent\tmeaning" per switch (egv_api.cpp holds the one table; an
 * environment variable of the same name overrides a switch -- an A/B aid, the defaults are the tested configuration) */
const char* egv_config_dump(void);

/* A non-blocking HIP stream for companion work (weight gradients, the text tower) at HIP priority `priority`:
 * -1 high, 0 normal, 1 low (hipDeviceGetStreamPriorityRange on gfx950).  Low-priority companions only take the CUs the
 * calling stream's kernels leave free.  The handle is a hipStream_t; it lives until process exit. */
int egv_stream_create(int priority, void** stream);

/* ---- GEMM with fused epilogue: every nn.Linear / Conv2d(k=s=16) on the path ------------------------
 * C[M,N] = epi( sum_k Aop[m,k] * Bop[n,k] ),  Aop = A[M,K] (a_trans=0) or A[K,M] (a_trans=1), same for B.
 *   forward  x W^T + b : a_trans=0, b_trans=0        (video_transformer.py:53,56,120,152,160,166,183;
 *                                                       roberta.py:257-270,341,404,423; model.py:105-115,279-290;
 *                                                       heads.py:23,34,48-49; patch embed :82 after egv_im2col)
 *   dgrad    dy W      : a_trans=0, b_trans=1        (autograd of the same lines)
 * epi: v = scale*acc + bias[n]; pre[m,n] = v (optional save); v = act(v); v *= *gate (optional device
 * scalar: alpha_i2t / alpha_t2i, video_transformer.py:185, roberta.py:486); v += res1[m,n] + res2[m,n]
 * (residual adds video_transformer.py:218,222,226; roberta.py:424,488); v *= act'(aux[m,n]) (backward,
 * dact = EGV_ACT_*: aux is the saved pre-activation for GELU, the forward output for RELU/TANH).
 * C is dtype unless out_f32.  res1/res2/pre/aux share leading dim ldr (0 -> ldc). */
int egv_gemm(int dtype, int a_trans, int b_trans, int M, int N, int K,
             const void* A, int lda, const void* B, int ldb, void* C, int ldc, int out_f32,
             const float* bias, int act, const float* gate, const void* res1, const void* res2,
             void* pre, const void* aux, int dact, int ldr, float scale, void* stream);

/* wgrad: dW[N,K] (fp32) = scale * (*gate) * dY[M,N]^T X[M,K]; reduction over the M tokens is split across
 * workgroups into fp32 slabs in `workspace` and summed in a fixed order (deterministic).  If dbias != NULL it also
 * returns dbias[N] (fp32) = scale * (*gate) * sum_m dY[m,:] from the same pass over dY (bias gradient of the Linear). */
long long egv_gemm_wgrad_workspace_bytes(int N, int K, int M);
int egv_gemm_wgrad(int dtype, int M, int N, int K, const void* dY, int ldy, const void* X, int ldx,
                   float* dW, float* dbias, float scale, const float* gate, void* workspace, long long workspace_bytes,
                   void* stream);

/* Grouped weight gradients: the dW / db of SEVERAL Linear layers over the same M tokens in ONE launch (bf16 operands; N and K
 * multiples of 256; autograd of video_transformer.py:53,56,120,152,166,183 for one SpaceTimeBlock).  With all tiles of a block in
 * one launch two to four reduction splits fill the chip; the splits of a tile are summed INSIDE the launch, in split order
 * (deterministic), by the tile's last arriver -- no slab round trip, no reduction launch.  dw[N,K] (row pitch K) and db[N] (may
 * be NULL) are fp32 and scaled by *gate when gate != NULL.  cus: the launch is PERSISTENT with at most `cus` workgroups (one per
 * CU; <= 0: all CUs): a caller that runs it beside other work grants it a share of the chip instead of letting long-running
 * workgroups take every CU that falls free; with at least `cus` tiles in the group there are no reduction splits at all. */
typedef struct egv_wgrad_problem {
    const void* dy; int ldy;            /* [M, N], row pitch ldy elements */
    const void* x; int ldx;             /* [M, K] */
    float* dw; float* db;
    const float* gate;
    int N, K;
    int accumulate;                     /* 1: dw += ..., db += ... (beta = 1: a block used several times per step adds a later use's gradient
                                           into the first use's buffer inside the launch -- existing + (sum of the splits, gated): the bits of
                                           a separate fp32 add of two buffers); 0: overwrite */
} egv_wgrad_problem;
long long egv_gemm_wgrad_grouped_workspace_bytes(int M, int nprob, const egv_wgrad_problem* problems, int cus);   /* -1: group not supported */
int egv_gemm_wgrad_grouped(int dtype, int M, int nprob, const egv_wgrad_problem* problems, int cus, void* workspace,
                           long long workspace_bytes, void* stream);
/* The split-reduction counters of the grouped launch live in a pool per (device, stream) that is zeroed once and put back by the
 * launches themselves.  A launch whose wait for a publisher times out (never observed; a device fault) leaves its counters alone,
 * writes NaN and marks the pool: every later launch on that pool writes NaN too (loud, not silently wrong) until this call zeroes
 * the calling device's pools, stream-ordered on each pool's own stream. */
int egv_gemm_wgrad_group_reset(void);

/* ---- LayerNorm (video_transformer.py:196,207,210,304,115; roberta.py:160,336,417; model.py:155;
 * BertPredictionHeadTransform.LayerNorm heads.py:41).  stats = [M][2] fp32 {mean, rstd} (may be NULL in
 * inference).  bwd: dx = LN'(dy) (+ add if not NULL); dgamma/dbeta fp32 [D]. */
int egv_layernorm_fwd(int dtype, const void* x, void* y, const float* gamma, const float* beta, float* stats,
                      int M, int D, float eps, void* stream);
long long egv_layernorm_bwd_workspace_bytes(int M, int D);
int egv_layernorm_bwd(int dtype, const void* dy, const void* x, const float* stats, const float* gamma,
                      const void* add, void* dx, float* dgamma, float* dbeta, int M, int D, void* workspace, void* stream);
/* dx = LN'(dy) + add + add2 (either may be NULL): the input of a divided space-time block feeds norm3 AND both residual sums
 * (video_transformer.py:218,222), so its gradient collects two skip paths while the LayerNorm backward writes dx */
int egv_layernorm_bwd2(int dtype, const void* dy, const void* x, const float* stats, const float* gamma,
                       const void* add, const void* add2, void* dx, float* dgamma, float* dbeta, int M, int D, void* workspace,
                       void* stream);

/* out[n] (fp32) = scale * (*gate) * sum_m X[m,n]  -- bias gradients */
long long egv_colsum_workspace_bytes(int M, int N);
int egv_colsum(int dtype, const void* X, int M, int N, int ld, float* out, float scale, const float* gate,
               void* workspace, void* stream);
/* out[0] (fp32) = scale * sum_i a[i]*b[i]  -- gradients of the scalar gates; workspace >= 4 KiB */
int egv_dot(int dtype, const void* a, const void* b, long long n, float* out, float scale, void* workspace, void* stream);
/* out = dy * act'(aux)  (kind = EGV_ACT_*; aux = forward output for RELU/TANH, pre-activation for GELU) */
int egv_act_bwd(int dtype, const void* dy, const void* aux, void* out, long long n, int kind, void* stream);
/* hidden-state dropout fused with the residual adds it feeds (roberta.py:203,342,422): y = keep(i)/(1-p) * x + r1 + r2;
 * the backward of x is the same call on dy with r1 = r2 = NULL.  mask = counter-based function of (seed, element index). */
int egv_dropout_add(int dtype, const void* x, const void* r1, const void* r2, void* y, long long n, float p, unsigned int seed, void* stream);
/* ---- fp32 residual stream of the text tower inside the bf16 mode: bf16 GEMM operands and outputs, LayerNorm input / output and the
 * residual sums in fp32 -- what torch.autocast does (trainer/trainer_egoclip.py:143) with roberta.py:336-345, :417-426.
 * dropout_add_mixed: y (ytype) = keep(i)/(1-p) * x (xtype) + r1 (bf16, may be NULL) + r2 (fp32, may be NULL), same mask function as
 * egv_dropout_add.  layernorm_fwd_res32: y fp32 and (y16 != NULL) its bf16 copy, the next Linear's operand.  layernorm_bwd_res32:
 * dx (fp32) = LN'(dy16 + dy32) + add32; dy16 bf16 / dy32 fp32 may each be NULL (not both), add32 may be NULL. */
int egv_dropout_add_mixed(int xtype, const void* x, const void* r1_bf16, const float* r2_f32, int ytype, void* y, long long n, float p,
                          unsigned int seed, void* stream);
int egv_layernorm_fwd_res32(const float* x, float* y, void* y16, const float* gamma, const float* beta, float* stats, int M, int D,
                            float eps, void* stream);
int egv_layernorm_bwd_res32(const void* dy16, const float* dy32, const float* x, const float* stats, const float* gamma,
                            const float* add32, float* dx, float* dgamma, float* dbeta, int M, int D, void* workspace, void* stream);
/* fp32 residual stream of the video tower inside the bf16 mode (EGV_BLOCK_RES_F32 of egv_vblock_fwd): a residual sum of
 * SpaceTimeBlock.forward (video_transformer.py:218,222,226) formed in fp32 by the kernel that normalises it --
 * s = base + d1 + d2 + (*gate) * dg, base = base32 (fp32) or base16 (bf16; exactly one of the two), d1 / d2 / dg = bf16 Linear outputs
 * (each may be NULL; gate NULL = 1); outputs, each optional: sum32 = s, sum16 = bf16(s), y = bf16(LayerNorm(s)) with stats[M][2] =
 * (mean, rstd).  sum16 may alias d1 / d2 / dg (element-wise in place). */
int egv_sum_layernorm(const float* base32, const void* base16, const void* d1, const void* d2, const void* dg, const float* gate,
                      float* sum32, void* sum16, void* y, const float* gamma, const float* beta, float* stats, int M, int D, float eps,
                      void* stream);
int egv_cast(int dtype_src, int dtype_dst, const void* src, void* dst, long long n, void* stream);
/* dst[C][R] (bf16) = src[R][C] (fp32): transposed bf16 compute copy of a weight, so that dgrad runs in the NT form */
int egv_cast_transpose(const float* src, void* dst, int R, int C, void* stream);
/* dst[C][R] (dtype_dst) = src[R][C] (dtype_src, row pitch ld elements); pairs bf16->bf16, f32->bf16, f32->f32.  Used by the
   MLM decoder's input gradient (heads.py:44-50 backward): dlogits^T and dx^T feed / leave the split-K weight-gradient kernel. */
int egv_transpose(int dtype_src, int dtype_dst, const void* src, void* dst, int R, int C, int ld, void* stream);

/* ---- grouped attention, head_dim 64 (video_transformer.py:35-39,117-150,155-182; roberta.py:257-327) ----
 * Query rows and key rows are affine row sets of token matrices:
 *     row(b, g, i) = b*bs + base + g*gs + i*is,   b < B, g < G, i < n
 * and `extra` prepends one row (b*extra_bs + extra_row: the CLS token) on the OTHER side of the launched
 * kernel: the key side for egv_attn_fwd / egv_attn_bwd_dq, the query side for egv_attn_bwd_dkv.
 * Q/K/V/O (and gradients) are [rows, ld] matrices; head h lives in columns off + 64*h .. +63.
 * lse/delta: fp32 [query rows of the whole tensor][H].  mask: additive fp32 over the key index
 * (mask[b*mask_ld + i]), applies to non-extra keys.  scale multiplies q (video_transformer.py:123,171-172;
 * roberta.py:303).  egv_attn_bwd_dq must run before egv_attn_bwd_dkv (it produces delta). */
typedef struct egv_attn_desc {
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
    int nsplit; float* ws; long long ws_bytes;   /* split of the other-side loop, fp32 partial slabs */
    float drop_p; unsigned int drop_seed;        /* attention-probability dropout (roberta.py:313): P~ = P * keep/(1-p); the mask is a
                                                    counter-based function of (seed, query row, key row, head); 0 = off */
    float* O32;                                  /* optional (bf16 storage): fp32 values of O, [rows, ldo] with the offsets of O.  egv_attn_fwd
                                                    writes it (split form and the unsplit MFMA kernel), egv_attn_bwd_dq forms
                                                    delta = rowsum(dO o O) from it.  A query over thousands of keys (text -> image) has nearly
                                                    uniform probabilities: dS = P o (dP - delta) is a small difference and the bf16 rounding of
                                                    O puts a COMMON offset into delta -- 3-5 x the q / k gradient error of the reference
                                                    under autocast, measured (tools/bf16_grad_error.py); NULL = O itself is used */
} egv_attn_desc;
/* nsplit > 1 splits the OTHER side of a launch across workgroups (fp32 partials in ws, combined in a fixed order):
 * needed when one own row meets thousands of other rows (CLS query/key over all S tokens, text<->video cross attention).
 * which = 0 fwd, 1 dq, 2 dkv; n_own = q_n (fwd, dq) or k_n (dkv).  With nsplit == 1 and dtype == EGV_BF16, problems whose
 * other side has <= 288 rows (256 patches of a 14 x 14-patch frame + CLS) run on the MFMA kernels (csrc/egv_attn_mfma.hip). */
long long egv_attn_split_workspace_bytes(int which, int B, int G, int H, int n_own, int nsplit);
int egv_attn_fwd(int dtype, const egv_attn_desc* d, void* stream);
/* An unsplit egv_attn_fwd launch with an extra row, query row set == key row set and a workspace of
 * egv_attn_fwd_extra_workspace_bytes also computes the extra row AS A QUERY over the union of the groups' keys -- the CLS query
 * of the divided attention (video_transformer.py:129) -- from per-group partial softmax states, when egv_attn_fwd_covers_extra
 * returns 1 for the descriptor (bf16, 65..224 keys per group, no mask / dropout); the one-query launch is then not needed. */
long long egv_attn_fwd_extra_workspace_bytes(int B, int G, int H);
int egv_attn_fwd_covers_extra(int dtype, const egv_attn_desc* d);
int egv_attn_bwd_dq(int dtype, const egv_attn_desc* d, void* stream);
long long egv_attn_bwd_dkv_workspace_bytes(int B, int G, int H, int k_n, int nsplit);
int egv_attn_bwd_dkv(int dtype, const egv_attn_desc* d, void* stream);
/* dQ + dK/dV + delta of one grouped launch in a single kernel (bf16 divided video attention: no mask, dropout or split).
 * 0 = enqueued, 1 = shape not covered (nothing enqueued; call the two functions above), -1 = error.  It stores delta of the
 * row-set queries only and does not read d->delta.  With d->ws (>= egv_attn_bwd_fused_workspace_bytes) and an extra row at
 * extra_row == 0 it also writes that row's dQ / dK / dV (sum of per-group partials in group order: deterministic), replacing
 * the one-query egv_attn_bwd_dq and the one-key egv_attn_bwd_dkv launches of the CLS row. */
long long egv_attn_bwd_fused_workspace_bytes(int B, int G, int H);
int egv_attn_bwd_fused(int dtype, const egv_attn_desc* d, void* stream);
/* Many queries over <= 32 keys (the image-to-text cross attention, video_transformer.py:155-185: q_n >= 128, unit row strides, additive
 * key mask allowed, no extra row / dropout): egv_attn_fwd takes a one-launch kernel for it, and egv_attn_bwd_fused one launch for dQ, dK, dV
 * plus a fixed-order sum of per-workgroup partials when d->ws holds >= egv_attn_fewkeys_workspace_bytes (d->delta is not read). */
long long egv_attn_fewkeys_workspace_bytes(int B, int G, int H, int q_n);
/* The mirror image, <= 32 queries over >= 512 keys (text-to-image cross attention, roberta.py:241-327: no mask, dropout allowed): with
 * d->ws >= egv_attn_fewq_workspace_bytes egv_attn_fwd runs one streaming launch + a combination (O, optional O32, lse), and
 * egv_attn_bwd_fused one launch for dK, dV (stored per key) and dQ (fixed-order sum of per-workgroup partials); delta comes from O32
 * when it is set, else from O. */
long long egv_attn_fewq_workspace_bytes(int B, int G, int H, int k_n);
/* The same for launches whose groups are one 16-row tile (the 17-row time attention), on the dQ + dK/dV kernel pair: when
 * egv_attn_bwd_pair_covers_extra returns 1 for a descriptor with d->ws set, egv_attn_bwd_dq and egv_attn_bwd_dkv leave the extra
 * row's gradients as per-group partials and egv_attn_bwd_extra_reduce(self_term = 1) sums them (adding the extra-query x
 * extra-key term, which no group owns) -- the one-query / one-key launches of the CLS row are not needed. */
int egv_attn_bwd_pair_covers_extra(int dtype, const egv_attn_desc* d);
int egv_attn_bwd_extra_reduce(int dtype, const egv_attn_desc* d, int self_term, void* stream);

/* ---- patch embedding pre/post (video_transformer.py:78-83,356-371; model.py:212-231,296-317) ---- */
int egv_im2col(int dtype, const float* video, void* out, int BF, int C, int H, int W, int P, void* stream);
/* The same patch matrix from uint8 clips [BF][3][H][W]: (x/255 - mean[c]) / std[c] while patchifying = ToTensor + Normalize of
   the reference's input transform (data_loader/transforms.py:17-19) on the device.  mean3 / std3: HOST float[3]. */
int egv_im2col_u8(int dtype, const unsigned char* video, void* out, int BF, int C, int H, int W, int P, const float* mean3,
                  const float* std3, void* stream);
int egv_assemble_tokens(int dtype, const void* patch, const float* cls, const float* pos, const float* temporal,
                        void* out, int B, int F, int N, int D, void* stream);
long long egv_assemble_tokens_bwd_workspace_bytes(int F, int N, int D);
int egv_assemble_tokens_bwd(int dtype, const void* dX, void* dpatch, float* dcls, float* dpos, float* dtemporal,
                            int B, int F, int N, int D, void* workspace, void* stream);

/* ---- RoBERTa embeddings (roberta.py:174-204,881-892): out = word[id] + type[0] + position[posid] ---- */
int egv_text_embed_fwd(int dtype, const long long* ids, const float* word, const float* pos, const float* type,
                       void* out, int B, int L, int D, int pad_id, void* stream);
int egv_text_embed_bwd(int dtype, const long long* ids, const void* de, float* dword, float* dpos, int B, int L, int D,
                       int pad_id, void* stream);

/* ---- cross entropy (model.py:414-418,478): lse/row_loss fp32 [R]; bwd writes coef*(softmax-onehot),
 * zero for ignored rows and for padded columns [V, Vpad) ---- */
int egv_ce_fwd(int dtype, const void* logits, const long long* labels, float* lse, float* row_loss, int R, int V, int ld,
               long long ignore_index, void* stream);
int egv_ce_bwd(int dtype, const void* logits, const long long* labels, const float* lse, const float* coef, void* dlogits,
               int R, int V, int Vpad, int ld, long long ignore_index, void* stream);

/* ---- sim_matrix + EgoNCE (model.py:576-584; loss.py:40-61), fp32 ---- */
int egv_l2norm_fwd(const float* x, float* y, float* nrm, int n, int d, float eps, void* stream);
int egv_l2norm_bwd(const float* dy, const float* y, const float* nrm, float* dx, int n, int d, float eps, void* stream);
/* the matrix product of sim_matrix (model.py:582-583) for the small matrices of the EgoNCE branch (n x m entries, one wave each):
   sim[n,m] = a[n,d] b[m,d]^T;  backward out[rows,d] = g other[cols,d] with g = ds[rows,cols] (trans = 0) or ds[cols,rows]^T (trans = 1) */
int egv_sim_small_fwd(const float* a, const float* b, float* sim, int n, int m, int d, void* stream);
int egv_sim_small_bwd(const float* ds, const float* other, float* out, int rows, int cols, int d, int trans, void* stream);
int egv_egonce_fwd(const float* x, const float* sim_v, const float* sim_n, int n, float temperature, int noun, int verb,
                   float* stats /* [2n][4] */, float* loss, unsigned char* mask_bool /* [n][n] or NULL */, void* stream);
int egv_egonce_bwd(const float* x, const float* sim_v, const float* sim_n, const float* stats, const float* gout, float* dx,
                   int n, float temperature, int noun, int verb, void* stream);

/* bf16 W [R,C] and W^T [C,R] compute copies of many fp32 master weights in ONE launch (the per-step weight preparation of
   the bf16 mode; replaces ~400 per-tensor cast launches).  table: device array of 32-byte records
   {const float* src; void* dst; void* dst_t; int R; int C} with R % 64 == C % 64 == 0; prefix: device int32[ntensors + 1],
   prefix[t] = number of 64x64 tiles before tensor t; ntiles = prefix[ntensors]. */
int egv_cast_weights(const void* table, const int* prefix, int ntensors, int ntiles, void* stream);
/* The same with a row pitch for the transposed copy: 40-byte records {const float* src; void* dst; void* dst_t; int R; int C; int ldt;
   int pad} -- dst_t[c * ldt + r]; several weights that share their input (query / key / value) then land side by side in ONE
   [C, sum R] transposed matrix (and, with consecutive dst blocks, in ONE [sum R, C] matrix). */
int egv_cast_weights_ld(const void* table, const int* prefix, int ntensors, int ntiles, void* stream);
/* fp32 segment copies in one launch (the concatenated biases of merged projections): table = device array of 24-byte records
   {const float* src; float* dst; long long n} */
int egv_copy_segments(const void* table, int nseg, void* stream);
/* ---- MX-fp8 weight GEMMs (BASELINE.json configs[4]: ViT-L/14 + RoBERTa-large, "fp8 MFMA weight path"; the reference has no
 * fp8 code: what it replaces are the bf16 forward / dgrad GEMMs of video_transformer.py:53,56,120,152,166,183 under a block-scaled
 * fp8 format).  Format: OCP MXFP8 E4M3 -- e4m3fn codes, one E8M0 scale 2^(s-127) per 32 consecutive elements of the contraction
 * dimension, s = the smallest power of two that brings the block's largest magnitude to <= 448 (nothing saturates).
 * egv_quant_mx: x bf16 [R,K] (row pitch ld) -> q [R,K] codes (row pitch K) + the scale bytes in the lane order of the
 * v_mfma_scale_f32_16x16x128_f8f6f4 consumers (csrc/egv_mx.hip): role 0 = the GEMM's A operand (activations, output gradients),
 * role 1 = its B operand (weights).  Scale rows past R are not written: fill the array with 0x7f once.
 * egv_quant_mx_batch: many tensors in one launch -- 40-byte records {const void* src; void* q; void* scales; int R, K, ld, role},
 * prefix[t] = workgroups before tensor t (a tensor takes ceil(R * K / 32 / 256)).
 * egv_gemm_mx: C[M,N] (bf16) = epi( A B^T ) with both operands quantised along K; epilogue as egv_gemm (bias, act, saved
 * pre-activation `pre`, residual res1, activation-derivative operand aux/dact).  K % 128 == 0, K >= 384, N % 64 == 0. */
/* bf16 LayerNorm (as egv_layernorm_fwd) that also writes the MX-fp8 form of its output y (role 0): bit-identical to
   egv_quant_mx(y) in a separate pass; D % 128 == 0 */
int egv_layernorm_fwd_mx(const void* x, void* y, const float* gamma, const float* beta, float* stats, void* q, void* scales,
                         int M, int D, float eps, void* stream);
long long egv_mx_scale_bytes(int R, int K, int role);
int egv_quant_mx(const void* x, int R, int K, int ld, void* q, void* scales, int role, void* stream);
int egv_quant_mx_batch(const void* table, const int* prefix, int ntensors, int nblocks, void* stream);
int egv_gemm_mx(int M, int N, int K, const void* Aq, const void* Ascales, const void* Bq, const void* Bscales, void* C, int ldc,
                const float* bias, int act, const void* res1, void* pre, const void* aux, int dact, int ldr,
                void* out_q, void* out_scales,   /* optional (both or neither; GELU+pre and GELU' epilogues, N % 128 == 0): C also in MX-fp8
                                                    form, role 0 -- the A operand of the next Linear, bit-identical to egv_quant_mx(C) */
                void* stream);

/* ---- fused multi-tensor AdamW (set_optim_schedule.py:108 -> transformers 4.30 AdamW: eps on sqrt(v) without bias
 * correction of the denominator, step_size = lr*sqrt(1-b2^t)/(1-b1^t), weight decay p -= lr*wd*p AFTER the update).
 * table: device array of 32-byte records {float* p; const float* g; float* m; float* v; int n; int pad}, one per tensor;
 * prefix: device int32[ntensors+1] = running count of 16384-element chunks (one workgroup per chunk). ---- */
int egv_adamw_step(const void* table, const int* prefix, int ntensors, int nchunks, float lr, float step_size, float beta1,
                   float beta2, float eps, float weight_decay, float grad_scale, void* stream);


/* ---- block-level entry points (csrc/egv_block.cpp): ONE call = one SpaceTimeBlock / one RobertaLayer, forward or backward.
 * They issue the same kernels as the entry points above, in the reference's order; what they add is the saved-activation
 * layout, the scratch plan and the stream choreography, so that the host issues ~130 calls per training step instead of
 * ~4 600.  save/ws: caller-allocated, sizes from the *_bytes queries (ws: forward and backward sizes differ).  All work is
 * enqueued on `stream`; with stream2 != NULL the weight-gradient GEMMs of a backward call run on stream2, forked from and
 * joined back into `stream` inside the call (events from a library-owned pool), so every output is ordered on `stream`
 * when the call returns.  flags & EGV_BLOCK_NO_JOIN (backward, stream2 != NULL): the call returns WITHOUT the join -- dw / db
 * (not the LayerNorm / gate gradients, not dx / dy) are then ordered on stream2 only, and ws / save / dout must stay untouched
 * until stream2 has drained; the caller joins once per backward pass (hipops.py does).  Weights: w[i] = compute-dtype copy W[N,K]; wt[i] = its transpose W^T[K,N] (bf16 mode, may be
 * NULL: dgrad then reads W as a [reduction, out] operand); biases, LayerNorm affine terms, gates and ALL gradients fp32.
 * dln_b[i] must be dln_g[i] + D (one [2][D] buffer per LayerNorm).
 *
 * SpaceTimeBlock.forward (video_transformer.py:214-228) with VarAttention (:117-187) and Mlp (:42-58) on the flat token
 * matrix x[B*S, D], S = 1 + F*N.  L > 0 selects the fused form: y[B*L, D] = text states, y_mask[B, L] additive fp32 key mask,
 * weights 6..8 = qkv_text_i2t (2D x D), qkv_i2t, proj_i2t, LayerNorm 3 = norm_i2t_i, alpha = alpha_i2t (:155-185).
 * Weight order: timeattn.qkv, timeattn.proj, attn.qkv, attn.proj, mlp.fc1, mlp.fc2; LayerNorm order: norm3, norm1, norm2. */
#define EGV_BLOCK_NO_JOIN 1
#define EGV_BLOCK_RES_F32 2
      /* egv_tlayer_*: hid / out / dout / dhid are fp32 (the text tower's fp32 residual stream), dtype = EGV_BF16;
         egv_vblock_fwd: the fp32 stream rides beside the bf16 tensors (x32 / out32 below) */
#define EGV_BLOCK_FP8 4       /* egv_vblock_*: MX-fp8 forward / dgrad GEMMs where the desc carries quantised weights */
#define EGV_BLOCK_TAIL 8      /* egv_vblock_bwd with EGV_BLOCK_NO_JOIN: nothing but the join follows this call on the calling stream (the last block of a
                                 backward pass): its grouped weight-gradient launch gets 7/8 of the CUs instead of its share */
typedef struct egv_vblock_desc {
    int dtype, B, F, N, H, D, Hd, L;
    float eps;
    const void* x; void* out;                       /* [B*S, D] */
    const void* y; const float* y_mask;             /* fused only */
    void* save; long long save_bytes;
    void* ws; long long ws_bytes;
    const void* w[9]; const void* wt[9]; const float* b[9];
    const float* ln_g[4]; const float* ln_b[4];
    const float* alpha;
    /* backward */
    const void* dout; void* dx; void* dy;           /* dy [B*L, D]: gradient of the text states (fused), may be NULL */
    float* dw[9]; float* db[9]; float* dln_g[4]; float* dln_b[4]; float* dalpha;
    void* stream; void* stream2;
    int flags;                                      /* EGV_BLOCK_* */
    /* EGV_BLOCK_FP8 (bf16 blocks): MX-fp8 copies of the Linear weights (egv_quant_mx role 1) -- wq[i] / wq_s[i] = codes / scales of
     * w[i] [N,K] (forward), wtq[i] / wtq_s[i] of wt[i] [K,N] (data gradient).  A Linear with both operands present runs its forward /
     * dgrad over the M video tokens as egv_gemm_mx on activations quantised on the fly (role 0); NULL entries, the weight gradients,
     * the gated i2t projection and the B*L-row text projections stay bf16. */
    const void* wq[9]; const void* wq_s[9]; const void* wtq[9]; const void* wtq_s[9];
    /* EGV_BLOCK_RES_F32 (egv_vblock_fwd, bf16 blocks without EGV_BLOCK_FP8): the residual stream in fp32, as torch.autocast keeps it
     * (trainer_egoclip.py:143; the sums at video_transformer.py:218,222,226 and the LayerNorms that read them).  x32 [B*S, D] = the
     * stream's fp32 value at the block input (NULL: x is exact), out32 receives the fp32 output; x / out stay the bf16 roundings
     * (GEMM operands, and all egv_vblock_bwd reads).  The backward pass ignores both fields. */
    const float* x32; float* out32;
    /* EGV_BLOCK_RES_F32, optional: the NEXT block's first LayerNorm (its norm3, video_transformer.py:217) folded into this call's
     * output pass -- the kernel that forms out = sr + fc2(..) in fp32 has the whole row in registers and also writes
     * LayerNorm(out; next_g, next_b) (bf16) to next_h and (mean, rstd) to next_stats: the h3 / stats3 slots of the SAVE BUFFER OF THE
     * NEXT CALL (egv_vblock_next_slots), which is then made with EGV_BLOCK_H3_READY and skips its own LayerNorm pass over the fp32
     * stream (one 115 MB pass per block at configs[2]).  Values are those of the separate pass, bit for bit (same kernel, same row). */
    const float* next_g; const float* next_b; void* next_h; float* next_stats;
    /* egv_vblock_fwd: CUs the persistent GEMM grids of this call plan for (0: all) -- a caller that runs two independent block
     * chains on two streams lets each chain's GEMMs take a share of the chip, so that the other chain's HBM-bound kernels
     * (LayerNorm, attention) find free CUs beside them instead of queueing behind a grid that owns every CU. */
    int fwd_cus;
    /* egv_vblock_bwd: bit w set = the weight / bias gradient of Linear slot w is ADDED to dw[w] / db[w] (egv_wgrad_problem::accumulate) -- honoured by
     * the grouped weight-gradient launch only (egv_vblock_bwd_groups(d) == 1); a caller must not set bits otherwise. */
    unsigned int acc_mask;
} egv_vblock_desc;
#define EGV_BLOCK_HEAD 32     /* egv_vblock_fwd / _bwd (EGV_BLOCK_RES_F32 form): only the part of the block every output row depends on -- norm3, the
                                 time attention with its projection and residual, norm1 and the space attention's qkv projection
                                 (video_transformer.py:217-219, :120) -- for a block whose output is read at the CLS rows only (the last block
                                 of a tower: video_transformer.py:392-394, model.py:275): forward stops there, the result is the qkv_s slot of
                                 `save` (egv_vblock_qkv_s_offset; out / out32 unused); backward takes dout = the gradient of that [M, 3D] matrix
                                 and returns dx without the space residual's share.  The caller runs the CLS query, the output projection and
                                 the MLP on B rows (model.py: _video_block_tail) instead of on all M: 66 % of the block's matrix work is dead. */
#define EGV_BLOCK_INFER 64    /* egv_vblock_fwd: no egv_vblock_bwd call will be made on this call's `save` (inference, validation: torch.no_grad()) -- what only
                                 the backward pass reads is not written: the MLP's pre-activation [M, Hd] (fc1 runs its GELU epilogue without the second
                                 store; not with EGV_BLOCK_FP8) and, with EGV_BLOCK_RES_F32, the bf16 roundings of the two inner residual sums.
                                 Outputs are bitwise those of a call without the flag. */
#define EGV_BLOCK_H3_READY 16 /* egv_vblock_fwd with EGV_BLOCK_RES_F32: the h3 / stats3 slots of `save` were filled by the previous call (next_h / next_stats) */
long long egv_vblock_save_bytes(const egv_vblock_desc* d);
/* byte offsets of the stats3 [M][2] fp32 and h3 [M, D] slots inside a save buffer of this geometry */
int egv_vblock_next_slots(const egv_vblock_desc* d, long long* stats3_off, long long* h3_off);
long long egv_vblock_qkv_s_offset(const egv_vblock_desc* d);   /* byte offset of the space attention's qkv [M, 3D] slot inside `save` */
long long egv_vblock_ws_bytes(const egv_vblock_desc* d, int backward);
int egv_vblock_fwd(const egv_vblock_desc* d);
int egv_vblock_bwd(const egv_vblock_desc* d);
/* 1 if egv_vblock_bwd(d) with EGV_BLOCK_NO_JOIN would return with weight-gradient work still running on d->stream2 */
int egv_vblock_bwd_defers(const egv_vblock_desc* d);
/* bit mask of the Linear slots whose weight gradients egv_vblock_bwd(d) would form in ONE grouped launch (and may therefore accumulate: acc_mask); 0: none */
unsigned int egv_vblock_bwd_groups(const egv_vblock_desc* d);

/* RobertaLayer.forward (roberta.py:444-505) on hid[B*L, D]: self attention (:257-327, separate q/k/v Linears, additive key
 * mask, probability dropout), RobertaSelfOutput (:335-345), optional text-to-image cross attention over enc[B*S, D] (video
 * tokens, no mask, :486-488: alpha_t2i * dense(cctx) + a0 + hidden, no LayerNorm inside), intermediate + output (:397-426).
 * drop_p > 0 (train mode): hidden dropout before every residual add and attention-probability dropout; seeds[0..5] = one
 * 32-bit seed per site: 0 self-attention probabilities, 1 attention.output, 2 crossattention probabilities,
 * 3 crossattention.output, 4 output (5 unused).  Weight order: query, key, value, attention.output.dense, intermediate.dense,
 * output.dense, crossattention_t2i.self.{query,key,value}, crossattention_t2i.output.dense; LayerNorm order:
 * attention.output.LayerNorm, output.LayerNorm. */
typedef struct egv_tlayer_desc {
    int dtype, B, L, H, D, Hd, S;                   /* S > 0: fused layer, enc has B*S rows */
    float eps, drop_p;
    unsigned int seeds[6];
    const void* hid; void* out;                     /* [B*L, D] */
    const float* mask;                              /* [B, L] additive fp32 */
    const void* enc;
    void* save; long long save_bytes;
    void* ws; long long ws_bytes;
    const void* w[10]; const void* wt[10]; const float* b[10];
    const float* ln_g[2]; const float* ln_b[2];
    const float* alpha;
    /* backward */
    const void* dout; void* dhid; void* denc;       /* denc [B*S, D]: gradient of the video tokens (fused), may be NULL */
    float* dw[10]; float* db[10]; float* dln_g[2]; float* dln_b[2]; float* dalpha;
    void* stream; void* stream2;
    int flags;                                      /* EGV_BLOCK_* */
    /* Same-input projections as ONE GEMM (bf16 mode; each may be NULL -> three / two separate GEMMs): w_qkv = [query ; key ; value]
     * rows [3D, D], wt_qkv its transpose [D, 3D], b_qkv [3D] (roberta.py:257-270 reads hidden_states three times); w_ckv = text-to-image
     * [key ; value] [2D, D] over the video tokens (roberta.py:241-242,274-277), wt_ckv [D, 2D], b_ckv [2D].  With w_qkv set, dw[1], dw[2]
     * must follow dw[0] contiguously ([3D, D] fp32) and db[1], db[2] follow db[0]; with w_ckv set, dw[8] follows dw[7] and db[8] db[7]. */
    const void* w_qkv; const void* wt_qkv; const float* b_qkv;
    const void* w_ckv; const void* wt_ckv; const float* b_ckv;
    unsigned int acc_mask;                          /* egv_tlayer_bwd: as egv_vblock_desc::acc_mask (slots 0..9; a merged q | k | v gradient follows bit 0, k | v of t2i bit 7) */
} egv_tlayer_desc;
long long egv_tlayer_save_bytes(const egv_tlayer_desc* d);
long long egv_tlayer_ws_bytes(const egv_tlayer_desc* d, int backward);
int egv_tlayer_fwd(const egv_tlayer_desc* d);
int egv_tlayer_bwd(const egv_tlayer_desc* d);
unsigned int egv_tlayer_bwd_groups(const egv_tlayer_desc* d);   /* as egv_vblock_bwd_groups */

/* ---- instrumentation: HIP-event timing of the GEMM launches on their own stream (bench.py roofline) ---- */
int egv_prof_enable(int on);
int egv_prof_reset(void);
/* synchronises the recorded events; returns the number of (flops, ms) records copied to the HOST arrays */
int egv_prof_collect(double* flops, float* ms, int* kind, int max_records);
/* the same plus the algorithmic bytes of each launch (operands, output and epilogue operands once) */
int egv_prof_collect2(double* flops, double* bytes, float* ms, int* kind, int max_records);
/* ... and the number of workgroups (= CUs) each persistent launch was planned for (0 for non-persistent kernels) */
int egv_prof_collect3(double* flops, double* bytes, float* ms, int* kind, int* cus, int max_records);

#ifdef __cplusplus
}
#endif
#endif
