// Layout / embedding / loss kernels of the EgoVLPv2 hot path (SURVEY.md K1 pre/post, K8 embeddings,
// K11 EgoNCE, K12 cross-entropy).  All are HBM- or latency-bound; fp32 arithmetic throughout.
#include "egv_common.h"

namespace egv {

// ------------------------------------------------------------------------------------------------
// K1 (pre): patchify.  video f32 [BF, C, H, W] -> patches T [BF * gh * gw, C * P * P] with the (c, ph, pw)
// flattening of the Conv2d weight (video_transformer.py:76,82), so patch embedding is one MFMA GEMM.
// One thread moves 4 consecutive pixels (16-byte read).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void im2col_kernel(const float* __restrict__ video, T* __restrict__ out, int BF, int C, int H, int W, int P) {
    const int gw = W / P, gh = H / P;
    const int rowlen = C * P * P;
    const long long total = (long long)BF * C * H * (W / 4);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x4 = (int)(i % (W / 4));
        long long r = i / (W / 4);
        const int y = (int)(r % H); r /= H;
        const int c = (int)(r % C);
        const int bf = (int)(r / C);
        const int x = x4 * 4;
        float v[4];
        ld4(video + (((long long)bf * C + c) * H + y) * W + x, v);
        const int py = y / P, ph = y % P;
        if ((P & 3) == 0) {
            const int px = x / P, pw = x % P;
            const long long orow = ((long long)bf * gh + py) * gw + px;
            st4(out + orow * rowlen + (c * P + ph) * P + pw, v);
        } else {                                               // patch width not a multiple of 4 (14 x 14 patches): the four pixels
#pragma unroll                                                 // may belong to two patches and the row is not 16-byte aligned
            for (int e = 0; e < 4; ++e) {
                const int px = (x + e) / P, pw = (x + e) % P;
                const long long orow = ((long long)bf * gh + py) * gw + px;
                Elem<T>::st(out + orow * rowlen + (c * P + ph) * P + pw, v[e]);
            }
        }
    }
}

// The same patch rows from uint8 clips: (x / 255 - mean[c]) / std[c] applied while patchifying, i.e. the ToTensor +
// Normalize of the reference's input transform (data_loader/transforms.py:17-19) moved onto the device -- the host then
// ships 1 byte per pixel instead of 4.  One thread moves 4 consecutive pixels (4-byte read).
template <typename T>
__global__ void im2col_u8_kernel(const unsigned char* __restrict__ video, T* __restrict__ out, int BF, int C, int H, int W, int P,
                                 float m0, float m1, float m2, float s0, float s1, float s2) {
    const int gw = W / P, gh = H / P;
    const int rowlen = C * P * P;
    const long long total = (long long)BF * C * H * (W / 4);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x4 = (int)(i % (W / 4));
        long long r = i / (W / 4);
        const int y = (int)(r % H); r /= H;
        const int c = (int)(r % C);
        const int bf = (int)(r / C);
        const int x = x4 * 4;
        const unsigned int px4 = *reinterpret_cast<const unsigned int*>(video + (((long long)bf * C + c) * H + y) * W + x);
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), inv = 1.0f / (c == 0 ? s0 : (c == 1 ? s1 : s2));
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ((float)((px4 >> (8 * e)) & 0xffu) * (1.0f / 255.0f) - mean) * inv;
        const int py = y / P, ph = y % P;
        if ((P & 3) == 0) {
            const int px = x / P, pw = x % P;
            const long long orow = ((long long)bf * gh + py) * gw + px;
            st4(out + orow * rowlen + (c * P + ph) * P + pw, v);
        } else {                                               // patch width not a multiple of 4 (14 x 14 patches): the four pixels
#pragma unroll                                                 // may belong to two patches and the row is not 16-byte aligned
            for (int e = 0; e < 4; ++e) {
                const int px = (x + e) / P, pw = (x + e) % P;
                const long long orow = ((long long)bf * gh + py) * gw + px;
                Elem<T>::st(out + orow * rowlen + (c * P + ph) * P + pw, v[e]);
            }
        }
    }
}

// K1 (post): tokens[b, 0] = cls + pos[0];  tokens[b, 1 + f*N + n] = patch[(b*F + f)*N + n] + pos[1 + n] + temporal[f]
// (video_transformer.py:360-371 / model.py:217-231).  One wave per output row.
template <typename T>
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const T* __restrict__ patch, const float* __restrict__ cls,
                                                              const float* __restrict__ pos, const float* __restrict__ temporal,
                                                              T* __restrict__ out, int B, int F, int N, int D) {
    const int lane = threadIdx.x & 63;
    const int S = 1 + F * N;
    const long long row = (long long)blockIdx.x * 4 + wave_id();
    if (row >= (long long)B * S) return;
    const int b = (int)(row / S), s = (int)(row % S);
    for (int c = lane * 4; c < D; c += 256) {
        float o[4];
        if (s == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = cls[c + e] + pos[c + e];
        } else {
            const int f = (s - 1) / N, n = (s - 1) % N;
            float x[4];
            ld4(patch + ((long long)(b * F + f) * N + n) * D + c, x);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = x[e] + pos[(long long)(1 + n) * D + c + e] + temporal[(long long)f * D + c + e];
        }
        st4(out + row * D + c, o);
    }
}

// backward of assemble: dpatch rows (copy), E[f*N+n] = sum_b dX[b, 1+f*N+n]  (fp32), dcls = sum_b dX[b,0]
template <typename T>
__global__ __launch_bounds__(256) void assemble_bwd_kernel(const T* __restrict__ dX, T* __restrict__ dpatch, float* __restrict__ E,
                                                           float* __restrict__ dcls, int B, int F, int N, int D) {
    const int lane = threadIdx.x & 63;
    const int S = 1 + F * N;
    const int s = blockIdx.x * 4 + wave_id();          // token index within a sample
    if (s >= S) return;
    for (int c = lane * 4; c < D; c += 256) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < B; ++b) {
            float x[4];
            ld4(dX + ((long long)b * S + s) * D + c, x);
            if (s > 0) st4(dpatch + ((long long)b * F * N + (s - 1)) * D + c, x);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += x[e];
        }
        if (s == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) dcls[c + e] = acc[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) E[(long long)(s - 1) * D + c + e] = acc[e];
        }
    }
}

// dpos[0] = dcls; dpos[1+n] = sum_f E[f*N+n];  dtemporal[f] = sum_n E[f*N+n]
__global__ void posemb_grad_kernel(const float* __restrict__ E, const float* __restrict__ dcls, float* __restrict__ dpos,
                                   float* __restrict__ dtemporal, int F, int N, int D) {
    const int r = blockIdx.x;                           // 0..N: pos rows, N+1..N+F: temporal rows
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        if (r == 0) {
            dpos[c] = dcls[c];
        } else if (r <= N) {
            float s = 0.f;
            for (int f = 0; f < F; ++f) s += E[((long long)f * N + (r - 1)) * D + c];
            dpos[(long long)r * D + c] = s;
        } else {
            // (196 rows per sum: four partial sums and sixteen loads in flight -- one dependent chain of 196 loads took 160 us)
            const int f = r - N - 1;
            const float* e = E + (long long)f * N * D + c;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int n = 0;
#pragma unroll 4
            for (; n + 3 < N; n += 4) {
                s0 += e[(long long)n * D];
                s1 += e[(long long)(n + 1) * D];
                s2 += e[(long long)(n + 2) * D];
                s3 += e[(long long)(n + 3) * D];
            }
            for (; n < N; ++n) s0 += e[(long long)n * D];
            dtemporal[(long long)f * D + c] = (s0 + s1) + (s2 + s3);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K8 embeddings: e[b,l] = word[id] + type[0] + position[posid],  posid = cumsum(id != pad)*(id != pad) + pad
// (roberta.py:174-204, :881-892).  One wave per token.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int position_id(const long long* ids_row, int l, int pad) {
    if (ids_row[l] == pad) return pad;
    int c = 0;
    for (int t = 0; t <= l; ++t) c += (ids_row[t] != pad) ? 1 : 0;
    return c + pad;
}

template <typename T>
__global__ __launch_bounds__(256) void text_embed_kernel(const long long* __restrict__ ids, const float* __restrict__ word,
                                                         const float* __restrict__ pos, const float* __restrict__ type,
                                                         T* __restrict__ out, int BL, int L, int D, int pad) {
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * 4 + wave_id();
    if (tok >= BL) return;
    const int b = tok / L, l = tok % L;
    const long long id = ids[tok];
    const int pid = position_id(ids + (long long)b * L, l, pad);
    for (int c = lane * 4; c < D; c += 256) {
        float w[4], p[4], t[4], o[4];
        ld4(word + id * D + c, w);
        ld4(pos + (long long)pid * D + c, p);
        ld4(type + c, t);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = w[e] + t[e] + p[e];
        st4(out + (long long)tok * D + c, o);
    }
}

// Gradient of the embedding tables (rows of the padding index get no gradient, as with nn.Embedding(padding_idx=1) in
// roberta.py:154,168-171).  No atomics: one wave per token t; the wave whose token is the FIRST occurrence of its word id (or of
// its position id) owns that table row -- it adds the gradient rows of every token with the same id in token order and writes
// the sum once, so the result does not depend on scheduling.  dword / dpos must be zero-filled by the caller (untouched rows).
// B*L is a few hundred tokens: every wave scans the id list (L2-resident) itself; position ids are computed once per
// workgroup into LDS.
template <typename T>
__global__ __launch_bounds__(256) void text_embed_bwd_kernel(const long long* __restrict__ ids, const T* __restrict__ de,
                                                             float* __restrict__ dword, float* __restrict__ dpos, int BL, int L,
                                                             int D, int pad, int accumulate) {
    extern __shared__ int pids[];                                  // [BL] position id of every token
    for (int t = threadIdx.x; t < BL; t += blockDim.x) pids[t] = position_id(ids + (long long)(t / L) * L, t % L, pad);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * 4 + wave_id();
    if (tok >= BL) return;
    const long long id = ids[tok];
    const int pid = pids[tok];
    // table 0: word embeddings keyed by id; table 1: position embeddings keyed by pid
#pragma unroll
    for (int table = 0; table < 2; ++table) {
        const long long key = table == 0 ? id : (long long)pid;
        if (key == pad) continue;                                  // wave-uniform
        bool first = true;
        for (int t0 = 0; t0 < tok && first; t0 += 64) {
            const int t = t0 + lane;
            const bool same = t < tok && (table == 0 ? ids[t] : (long long)pids[t]) == key;
            if (__ballot(same)) first = false;
        }
        if (!first) continue;
        float* out = (table == 0 ? dword : dpos) + key * D;
        for (int cb = 0; cb < D; cb += 256) {                      // wave-uniform loops: every lane takes part in the ballots
            const int c = cb + lane * 4;
            const bool live = c < D;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int t0 = tok; t0 < BL; t0 += 64) {
                const int t = t0 + lane;
                unsigned long long m = __ballot(t < BL && (table == 0 ? ids[t] : (long long)pids[t]) == key);
                while (m) {                                        // matching tokens in increasing order
                    const int j = __builtin_ctzll(m);
                    m &= m - 1;
                    if (live) {
                        float g[4];
                        ld4(de + (long long)(t0 + j) * D + c, g);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[e] += g[e];
                    }
                }
            }
            if (live) {
                if (accumulate) {                                  // a later chunk of a long token list (chunks run one after the other)
                    float prev[4];
                    ld4(out + c, prev);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += prev[e];
                }
                st4(out + c, acc);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K12 cross-entropy over the (padded) vocabulary.  One workgroup per row.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* __restrict__ logits, const long long* __restrict__ labels,
                                                     float* __restrict__ lse_out, float* __restrict__ row_loss, int V, int ld,
                                                     long long ignore_index) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    const T* x = logits + (long long)r * ld;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < V; c += 256) m = fmaxf(m, Elem<T>::ld(x + c));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[wave_id()] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = threadIdx.x; c < V; c += 256) s += __expf(Elem<T>::ld(x + c) - m);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[wave_id()] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float lse = m + __logf(red[0] + red[1] + red[2] + red[3]);
        lse_out[r] = lse;
        const long long lab = labels[r];
        // labels outside [0, V) other than ignore_index would read out of bounds: treated as ignored (torch raises; the host
        // wrapper checks the dtype, the values are the data collator's)
        row_loss[r] = (lab == ignore_index || lab < 0 || lab >= V) ? 0.f : (lse - Elem<T>::ld(x + lab));
    }
}

// dlogits[r, v] = coef * (softmax - onehot) for valid rows; 0 for ignored rows and for padded columns v in [V, Vpad)
template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* __restrict__ logits, const long long* __restrict__ labels,
                                                     const float* __restrict__ lse, const float* __restrict__ coef,
                                                     T* __restrict__ dlogits, int V, int Vpad, int ld, long long ignore_index) {
    const int r = blockIdx.x;
    const long long lab = labels[r];
    const float cf = (lab == ignore_index || lab < 0 || lab >= V) ? 0.f : coef[0];
    const float l = lse[r];
    const T* x = logits + (long long)r * ld;
    T* d = dlogits + (long long)r * ld;
    for (int c = threadIdx.x; c < Vpad; c += 256) {
        float g = 0.f;
        if (c < V && cf != 0.f) g = cf * (__expf(Elem<T>::ld(x + c) - l) - (c == lab ? 1.0f : 0.0f));
        Elem<T>::st(d + c, g);
    }
}

// ------------------------------------------------------------------------------------------------
// K11 sim_matrix + EgoNCE (model.py:576-584, loss.py:40-61); fp32.
// ------------------------------------------------------------------------------------------------
// y = x / max(||x||, eps); inv[r] = 1 / max(||x||, eps); nrm[r] = ||x||
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ nrm,
                                                         int n, int d, float eps) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    float s = 0.f;
    for (int c = threadIdx.x; c < d; c += 256) {
        const float v = x[(long long)r * d + c];
        s += v * v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[wave_id()] = s;
    __syncthreads();
    const float nr = sqrtf(red[0] + red[1] + red[2] + red[3]);
    const float inv = 1.0f / fmaxf(nr, eps);
    for (int c = threadIdx.x; c < d; c += 256) y[(long long)r * d + c] = x[(long long)r * d + c] * inv;
    if (threadIdx.x == 0) nrm[r] = nr;
}

// sim[i, j] = <a[i, :], b[j, :]> for the small similarity matrices of the EgoNCE branch (n x m <= a few thousand entries over d =
// 118 ... 4096 features): ONE WAVE PER ENTRY.  As a GEMM this is a single 128 x 128 output tile whose K loop one workgroup walks alone
// -- 530 us for the 8 x 8 x 4096 text-video matrix, three such products per EgoNCE tail -- here every entry is a strided dot product
// with a wave reduction (fixed order: deterministic).
__global__ __launch_bounds__(256) void sim_small_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ sim,
                                                            int n, int m, int d) {
    const int e = blockIdx.x * 4 + wave_id();
    if (e >= n * m) return;
    const int i = e / m, j = e % m, lane = threadIdx.x & 63;
    const float* pa = a + (long long)i * d;
    const float* pb = b + (long long)j * d;
    float s0 = 0.f, s1 = 0.f;
    int c = lane;
    for (; c + 64 < d; c += 128) { s0 += pa[c] * pb[c]; s1 += pa[c + 64] * pb[c + 64]; }
    if (c < d) s0 += pa[c] * pb[c];
    const float s = wave_sum(s0 + s1);
    if (lane == 0) sim[e] = s;
}
// out[i, c] = sum_j g(i, j) * other[j, c],  g(i, j) = ds[i * ld + j] (trans = 0: gradient of the row operand) or ds[j * ld + i]
// (trans = 1: gradient of the column operand); rows values of i, cols values of j (<= a few dozen)
__global__ __launch_bounds__(256) void sim_small_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ other, float* __restrict__ out,
                                                            int rows, int cols, int d, int ld, int trans) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)rows * d) return;
    const int i = (int)(t / d), c = (int)(t % d);
    float s = 0.f;
    for (int j = 0; j < cols; ++j) s += (trans ? ds[(long long)j * ld + i] : ds[(long long)i * ld + j]) * other[(long long)j * d + c];
    out[t] = s;
}

// dx = (dy - y * <y, dy>) / ||x||   if ||x|| > eps,  else dy / eps
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ nrm, float* __restrict__ dx, int n, int d, float eps) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    float s = 0.f;
    for (int c = threadIdx.x; c < d; c += 256) s += dy[(long long)r * d + c] * y[(long long)r * d + c];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[wave_id()] = s;
    __syncthreads();
    const float dot = red[0] + red[1] + red[2] + red[3];
    const float nr = nrm[r];
    for (int c = threadIdx.x; c < d; c += 256) {
        const long long i = (long long)r * d + c;
        dx[i] = (nr > eps) ? (dy[i] - y[i] * dot) / nr : dy[i] / eps;
    }
}

// stats: block r < n -> row r of x ; block n + c -> column c of x.
// Writes st[4*k + {0,1,2}] = {max, den, num} where num = sum exp((x - max)/T) * mask, den = sum exp((x - max)/T).
// mask for row i:    m[i][j] = (sim_v[i][j] * sim_n[i][j] + (i==j)) > 0   (combined per the noun/verb flags)
// mask for column j: uses m[j][i]  (loss.py:58 multiplies softmax(x^T) by the UNtransposed mask)
__device__ __forceinline__ bool egonce_mask(const float* sv, const float* sn, int i, int j, int n, int noun, int verb) {
    float m = (i == j) ? 1.0f : 0.0f;
    if (noun && verb) m += sv[(long long)i * n + j] * sn[(long long)i * n + j];
    else if (noun) m += sn[(long long)i * n + j];
    else if (verb) m += sv[(long long)i * n + j];
    return m > 0.0f;
}

__global__ __launch_bounds__(256) void egonce_stats_kernel(const float* __restrict__ x, const float* __restrict__ sv,
                                                           const float* __restrict__ sn, float* __restrict__ st,
                                                           unsigned char* __restrict__ mask_out, int n, float invT, int noun, int verb) {
    __shared__ float red[4], red2[4];
    const int k = blockIdx.x;
    const bool is_row = k < n;
    const int a = is_row ? k : k - n;
    float m = -INFINITY;
    for (int t = threadIdx.x; t < n; t += 256) {
        const float v = is_row ? x[(long long)a * n + t] : x[(long long)t * n + a];
        m = fmaxf(m, v);
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[wave_id()] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float den = 0.f, num = 0.f;
    for (int t = threadIdx.x; t < n; t += 256) {
        const float v = is_row ? x[(long long)a * n + t] : x[(long long)t * n + a];
        const float e = __expf((v - m) * invT);
        const bool mk = egonce_mask(sv, sn, a, t, n, noun, verb);       // m[a][t] in both cases
        den += e;
        num += mk ? e : 0.f;
        if (is_row && mask_out) mask_out[(long long)a * n + t] = mk ? 1 : 0;
    }
    den = wave_sum(den);
    num = wave_sum(num);
    if ((threadIdx.x & 63) == 0) { red[wave_id()] = den; red2[wave_id()] = num; }
    __syncthreads();
    if (threadIdx.x == 0) {
        st[4 * k] = m;
        st[4 * k + 1] = red[0] + red[1] + red[2] + red[3];
        st[4 * k + 2] = red2[0] + red2[1] + red2[2] + red2[3];
    }
}

// loss = -(1/n) sum_i log(num_i/den_i) - (1/n) sum_j log(numc_j/denc_j)
__global__ __launch_bounds__(256) void egonce_loss_kernel(const float* __restrict__ st, float* __restrict__ loss, int n) {
    __shared__ float red[4];
    float s = 0.f;
    for (int k = threadIdx.x; k < 2 * n; k += 256) s += __logf(st[4 * k + 2] / st[4 * k + 1]);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[wave_id()] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = -(red[0] + red[1] + red[2] + red[3]) / (float)n;
}

// dx[i][j] = gout * (-(1/n) invT) * [ e_r (m[i][j]/num_i - 1/den_i) + e_c (m[j][i]/numc_j - 1/denc_j) ]
__global__ void egonce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ sv, const float* __restrict__ sn,
                                  const float* __restrict__ st, const float* __restrict__ gout, float* __restrict__ dx, int n,
                                  float invT, int noun, int verb) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * n) return;
    const int i = (int)(idx / n), j = (int)(idx % n);
    const float v = x[idx];
    const float er = __expf((v - st[4 * i]) * invT);
    const float ec = __expf((v - st[4 * (n + j)]) * invT);
    const float mr = egonce_mask(sv, sn, i, j, n, noun, verb) ? 1.0f : 0.0f;
    const float mc = egonce_mask(sv, sn, j, i, n, noun, verb) ? 1.0f : 0.0f;
    const float tr = er * (mr / st[4 * i + 2] - 1.0f / st[4 * i + 1]);
    const float tc = ec * (mc / st[4 * (n + j) + 2] - 1.0f / st[4 * (n + j) + 1]);
    dx[idx] = gout[0] * (-invT / (float)n) * (tr + tc);
}

// out = dy * act'(aux): relu' / tanh' use the forward OUTPUT as aux, gelu' the saved pre-activation
template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ aux, T* __restrict__ out, long long n, int kind) {
    const long long nv = n / 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        float g[4], a[4], o[4];
        ld4(dy + i * 4, g);
        ld4(aux + i * 4, a);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float d = 1.0f;
            if (kind == 1) d = sizeof(T) == 2 ? dgelu_fast_f(a[e]) : dgelu_f(a[e]);
            else if (kind == 2) d = a[e] > 0.f ? 1.0f : 0.0f;
            else if (kind == 3) d = 1.0f - a[e] * a[e];
            o[e] = g[e] * d;
        }
        st4(out + i * 4, o);
    }
}

__device__ __forceinline__ unsigned int fmix32_e(unsigned int h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
// y[i] = keep(i)/(1-p) * x[i] + r1[i] + r2[i]   (hidden-state dropout + the residual adds that follow it)
template <typename T>
__global__ void dropout_add_kernel(const T* __restrict__ x, const T* __restrict__ r1, const T* __restrict__ r2, T* __restrict__ y,
                                   long long n, float p, unsigned int seed) {
    const float inv = 1.0f / (1.0f - p);
    const long long nv = n / 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        float v[4], a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
        ld4(x + i * 4, v);
        if (r1) ld4(r1 + i * 4, a);
        if (r2) ld4(r2 + i * 4, b);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned int h = fmix32_e(seed ^ fmix32_e((unsigned int)(i * 4 + e) * 0x9E3779B1u + 0x7F4A7C15u));
            const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
            o[e] = (u >= p ? v[e] * inv : 0.f) + a[e] + b[e];
        }
        st4(y + i * 4, o);
    }
}

// mixed-type form for the fp32 residual stream of the text tower: y (TY) = keep(i)/(1-p) * x (TX) + r1 (bf16) + r2 (fp32); the keep
// decision is the one of dropout_add_kernel (same function of seed and element index), so forward (bf16 dense output -> fp32 sum)
// and backward (fp32 gradient -> bf16 operand of the dense layer's gradients) agree
template <typename TX, typename TY>
__global__ void dropout_add_mixed_kernel(const TX* __restrict__ x, const bf16_t* __restrict__ r1, const float* __restrict__ r2, TY* __restrict__ y,
                                         long long n, float p, unsigned int seed) {
    const float inv = 1.0f / (1.0f - p);
    const long long nv = n / 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        float v[4], a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
        ld4(x + i * 4, v);
        if (r1) ld4(r1 + i * 4, a);
        if (r2) ld4(r2 + i * 4, b);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float kept = v[e];
            if (p > 0.f) {
                const unsigned int h = fmix32_e(seed ^ fmix32_e((unsigned int)(i * 4 + e) * 0x9E3779B1u + 0x7F4A7C15u));
                const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
                kept = u >= p ? v[e] * inv : 0.f;
            }
            o[e] = kept + a[e] + b[e];
        }
        st4(y + i * 4, o);
    }
}

}  // namespace egv
using namespace egv;

#define EGV_ST reinterpret_cast<hipStream_t>(stream)

extern "C" int egv_im2col(int dtype, const float* video, void* out, int BF, int C, int H, int W, int P, void* stream) {
    EGV_CHECK(W % 4 == 0 && H % P == 0 && W % P == 0, "egv_im2col: unsupported geometry H=%d W=%d P=%d", H, W, P);
    const long long total = (long long)BF * C * H * (W / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (dtype == EGV_BF16) hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(blocks), dim3(256), 0, EGV_ST, video, (bf16_t*)out, BF, C, H, W, P);
    else hipLaunchKernelGGL(im2col_kernel<float>, dim3(blocks), dim3(256), 0, EGV_ST, video, (float*)out, BF, C, H, W, P);
    EGV_LAUNCH_CHECK();
    return 0;
}

extern "C" int egv_im2col_u8(int dtype, const unsigned char* video, void* out, int BF, int C, int H, int W, int P, const float* mean3,
                             const float* std3, void* stream) {
    EGV_CHECK(C == 3 && W % 4 == 0 && H % P == 0 && W % P == 0, "egv_im2col_u8: unsupported geometry C=%d H=%d W=%d P=%d", C, H, W, P);
    EGV_CHECK(mean3 && std3 && std3[0] > 0.f && std3[1] > 0.f && std3[2] > 0.f, "egv_im2col_u8: mean/std (host float[3]) required");
    const long long total = (long long)BF * C * H * (W / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (dtype == EGV_BF16)
        hipLaunchKernelGGL(im2col_u8_kernel<bf16_t>, dim3(blocks), dim3(256), 0, EGV_ST, video, (bf16_t*)out, BF, C, H, W, P, mean3[0], mean3[1],
                           mean3[2], std3[0], std3[1], std3[2]);
    else
        hipLaunchKernelGGL(im2col_u8_kernel<float>, dim3(blocks), dim3(256), 0, EGV_ST, video, (float*)out, BF, C, H, W, P, mean3[0], mean3[1],
                           mean3[2], std3[0], std3[1], std3[2]);
    EGV_LAUNCH_CHECK();
    return 0;
}

extern "C" int egv_assemble_tokens(int dtype, const void* patch, const float* cls, const float* pos, const float* temporal,
                                   void* out, int B, int F, int N, int D, void* stream) {
    EGV_CHECK(D % 4 == 0, "egv_assemble_tokens: D %% 4");
    const long long rows = (long long)B * (1 + F * N);
    dim3 grid((unsigned)((rows + 3) / 4));
    if (dtype == EGV_BF16)
        hipLaunchKernelGGL(assemble_tokens_kernel<bf16_t>, grid, dim3(256), 0, EGV_ST, (const bf16_t*)patch, cls, pos, temporal, (bf16_t*)out, B, F, N, D);
    else
        hipLaunchKernelGGL(assemble_tokens_kernel<float>, grid, dim3(256), 0, EGV_ST, (const float*)patch, cls, pos, temporal, (float*)out, B, F, N, D);
    EGV_LAUNCH_CHECK();
    return 0;
}

extern "C" long long egv_assemble_tokens_bwd_workspace_bytes(int F, int N, int D) { return ((long long)F * N + 1) * D * 4; }

// dpatch T [B*F*N, D]; dcls_model fp32 [D] (gradient of the cls parameter), dpos fp32 [1+N, D], dtemporal fp32 [F, D]
extern "C" int egv_assemble_tokens_bwd(int dtype, const void* dX, void* dpatch, float* dcls, float* dpos, float* dtemporal,
                                       int B, int F, int N, int D, void* workspace, void* stream) {
    EGV_CHECK(D % 4 == 0, "egv_assemble_tokens_bwd: D %% 4");
    const int S = 1 + F * N;
    float* E = (float*)workspace;
    dim3 grid((S + 3) / 4);
    if (dtype == EGV_BF16)
        hipLaunchKernelGGL(assemble_bwd_kernel<bf16_t>, grid, dim3(256), 0, EGV_ST, (const bf16_t*)dX, (bf16_t*)dpatch, E, dcls, B, F, N, D);
    else
        hipLaunchKernelGGL(assemble_bwd_kernel<float>, grid, dim3(256), 0, EGV_ST, (const float*)dX, (float*)dpatch, E, dcls, B, F, N, D);
    EGV_LAUNCH_CHECK();
    hipLaunchKernelGGL(posemb_grad_kernel, dim3(1 + N + F), dim3(256), 0, EGV_ST, (const float*)E, (const float*)dcls, dpos, dtemporal, F, N, D);
    EGV_LAUNCH_CHECK();
    return 0;
}

extern "C" int egv_text_embed_fwd(int dtype, const long long* ids, const float* word, const float* pos, const float* type,
                                  void* out, int B, int L, int D, int pad_id, void* stream) {
    EGV_CHECK(D % 4 == 0, "egv_text_embed_fwd: D %% 4");
    dim3 grid((B * L + 3) / 4);
    if (dtype == EGV_BF16)
        hipLaunchKernelGGL(text_embed_kernel<bf16_t>, grid, dim3(256), 0, EGV_ST, ids, word, pos, type, (bf16_t*)out, B * L, L, D, pad_id);
    else
        hipLaunchKernelGGL(text_embed_kernel<float>, grid, dim3(256), 0, EGV_ST, ids, word, pos, type, (float*)out, B * L, L, D, pad_id);
    EGV_LAUNCH_CHECK();
    return 0;
}

extern "C" int egv_text_embed_bwd(int dtype, const long long* ids, const void* de, float* dword, float* dpos, int B, int L, int D,
                                  int pad_id, void* stream) {
    EGV_CHECK((D % 4) == 0 && L > 0 && L <= 16384, "egv_text_embed_bwd: D must be a multiple of 4 and L <= 16384 tokens per sequence");
    // the kernel keeps the position ids of its token list in LDS (64 KB = 16384 tokens): longer lists go in chunks of whole
    // sequences, one launch after the other on the stream, every chunk after the first adding to what the earlier ones wrote --
    // still no atomics, still a fixed summation order (chunk order, then token order)
    const int seqs = 16384 / L;
    const size_t esz = dtype == EGV_BF16 ? 2 : 4;
    for (int b0 = 0; b0 < B; b0 += seqs) {
        const int nb = B - b0 < seqs ? B - b0 : seqs;
        const int bl = nb * L;
        const long long* idc = ids + (long long)b0 * L;
        const char* dec = reinterpret_cast<const char*>(de) + (size_t)b0 * L * D * esz;
        dim3 grid((bl + 3) / 4);
        const size_t lds = (size_t)bl * sizeof(int);
        if (dtype == EGV_BF16)
            hipLaunchKernelGGL(text_embed_bwd_kernel<bf16_t>, grid, dim3(256), lds, EGV_ST, idc, (const bf16_t*)dec, dword, dpos, bl, L, D, pad_id, b0 > 0 ? 1 : 0);
        else
            hipLaunchKernelGGL(text_embed_bwd_kernel<float>, grid, dim3(256), lds, EGV_ST, idc, (const float*)dec, dword, dpos, bl, L, D, pad_id, b0 > 0 ? 1 : 0);
    }
    EGV_LAUNCH_CHECK();
    return 0;
}

extern "C" int egv_ce_fwd(int dtype, const void* logits, const long long* labels, float* lse, float* row_loss, int R, int V, int ld,
                          long long ignore_index, void* stream) {
    if (dtype == EGV_BF16)
        hipLaunchKernelGGL(ce_fwd_kernel<bf16_t>, dim3(R), dim3(256), 0, EGV_ST, (const bf16_t*)logits, labels, lse, row_loss, V, ld, ignore_index);
    else
        hipLaunchKernelGGL(ce_fwd_kernel<float>, dim3(R), dim3(256), 0, EGV_ST, (const float*)logits, labels, lse, row_loss, V, ld, ignore_index);
    EGV_LAUNCH_CHECK();
    return 0;
}

extern "C" int egv_ce_bwd(int dtype, const void* logits, const long long* labels, const float* lse, const float* coef, void* dlogits,
                          int R, int V, int Vpad, int ld, long long ignore_index, void* stream) {
    if (dtype == EGV_BF16)
        hipLaunchKernelGGL(ce_bwd_kernel<bf16_t>, dim3(R), dim3(256), 0, EGV_ST, (const bf16_t*)logits, labels, lse, coef, (bf16_t*)dlogits, V, Vpad, ld, ignore_index);
    else
        hipLaunchKernelGGL(ce_bwd_kernel<float>, dim3(R), dim3(256), 0, EGV_ST, (const float*)logits, labels, lse, coef, (float*)dlogits, V, Vpad, ld, ignore_index);
    EGV_LAUNCH_CHECK();
    return 0;
}

extern "C" int egv_l2norm_fwd(const float* x, float* y, float* nrm, int n, int d, float eps, void* stream) {
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(n), dim3(256), 0, EGV_ST, x, y, nrm, n, d, eps);
    EGV_LAUNCH_CHECK();
    return 0;
}

extern "C" int egv_l2norm_bwd(const float* dy, const float* y, const float* nrm, float* dx, int n, int d, float eps, void* stream) {
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(n), dim3(256), 0, EGV_ST, dy, y, nrm, dx, n, d, eps);
    EGV_LAUNCH_CHECK();
    return 0;
}

extern "C" int egv_sim_small_fwd(const float* a, const float* b, float* sim, int n, int m, int d, void* stream) {
    EGV_CHECK(n > 0 && m > 0 && d > 0 && (long long)n * m <= (1 << 20), "egv_sim_small_fwd: 1 <= n * m <= 2^20 entries");
    hipLaunchKernelGGL(sim_small_fwd_kernel, dim3((n * m + 3) / 4), dim3(256), 0, EGV_ST, a, b, sim, n, m, d);
    EGV_LAUNCH_CHECK();
    return 0;
}

extern "C" int egv_sim_small_bwd(const float* ds, const float* other, float* out, int rows, int cols, int d, int trans, void* stream) {
    EGV_CHECK(rows > 0 && cols > 0 && d > 0, "egv_sim_small_bwd: bad shape");
    const long long tot = (long long)rows * d;
    hipLaunchKernelGGL(sim_small_bwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, EGV_ST, ds, other, out, rows, cols, d,
                       trans ? rows : cols, trans);
    EGV_LAUNCH_CHECK();
    return 0;
}

// stats: fp32 [2n][4] workspace kept for backward
extern "C" int egv_egonce_fwd(const float* x, const float* sim_v, const float* sim_n, int n, float temperature, int noun, int verb,
                              float* stats, float* loss, unsigned char* mask_bool, void* stream) {
    EGV_CHECK(n > 0 && temperature > 0.f, "egv_egonce_fwd: bad args");
    hipLaunchKernelGGL(egonce_stats_kernel, dim3(2 * n), dim3(256), 0, EGV_ST, x, sim_v, sim_n, stats, mask_bool, n, 1.0f / temperature, noun, verb);
    hipLaunchKernelGGL(egonce_loss_kernel, dim3(1), dim3(256), 0, EGV_ST, (const float*)stats, loss, n);
    EGV_LAUNCH_CHECK();
    return 0;
}

extern "C" int egv_egonce_bwd(const float* x, const float* sim_v, const float* sim_n, const float* stats, const float* gout, float* dx,
                              int n, float temperature, int noun, int verb, void* stream) {
    const long long tot = (long long)n * n;
    hipLaunchKernelGGL(egonce_bwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, EGV_ST, x, sim_v, sim_n, stats, gout, dx, n,
                       1.0f / temperature, noun, verb);
    EGV_LAUNCH_CHECK();
    return 0;
}

extern "C" int egv_act_bwd(int dtype, const void* dy, const void* aux, void* out, long long n, int kind, void* stream) {
    EGV_CHECK(n % 4 == 0, "egv_act_bwd: n %% 4");
    long long nb = (n / 4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    if (dtype == EGV_BF16)
        hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, dim3((int)nb), dim3(256), 0, EGV_ST, (const bf16_t*)dy, (const bf16_t*)aux, (bf16_t*)out, n, kind);
    else
        hipLaunchKernelGGL(act_bwd_kernel<float>, dim3((int)nb), dim3(256), 0, EGV_ST, (const float*)dy, (const float*)aux, (float*)out, n, kind);
    EGV_LAUNCH_CHECK();
    return 0;
}

extern "C" int egv_dropout_add_mixed(int xtype, const void* x, const void* r1_bf16, const float* r2_f32, int ytype, void* y, long long n, float p,
                                     unsigned int seed, void* stream) {
    EGV_CHECK(n % 4 == 0 && p >= 0.f && p < 1.f, "egv_dropout_add_mixed: n %% 4 == 0 and 0 <= p < 1 required");
    long long nb = (n / 4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    const bf16_t* r1 = (const bf16_t*)r1_bf16;
    if (xtype == EGV_BF16 && ytype == EGV_F32)
        hipLaunchKernelGGL((dropout_add_mixed_kernel<bf16_t, float>), dim3((int)nb), dim3(256), 0, EGV_ST, (const bf16_t*)x, r1, r2_f32, (float*)y, n, p, seed);
    else if (xtype == EGV_F32 && ytype == EGV_BF16)
        hipLaunchKernelGGL((dropout_add_mixed_kernel<float, bf16_t>), dim3((int)nb), dim3(256), 0, EGV_ST, (const float*)x, r1, r2_f32, (bf16_t*)y, n, p, seed);
    else if (xtype == EGV_F32 && ytype == EGV_F32)
        hipLaunchKernelGGL((dropout_add_mixed_kernel<float, float>), dim3((int)nb), dim3(256), 0, EGV_ST, (const float*)x, r1, r2_f32, (float*)y, n, p, seed);
    else
        hipLaunchKernelGGL((dropout_add_mixed_kernel<bf16_t, bf16_t>), dim3((int)nb), dim3(256), 0, EGV_ST, (const bf16_t*)x, r1, r2_f32, (bf16_t*)y, n, p, seed);
    EGV_LAUNCH_CHECK();
    return 0;
}

extern "C" int egv_dropout_add(int dtype, const void* x, const void* r1, const void* r2, void* y, long long n, float p, unsigned int seed,
                               void* stream) {
    EGV_CHECK(n % 4 == 0 && p >= 0.f && p < 1.f, "egv_dropout_add: n %% 4 == 0 and 0 <= p < 1 required");
    long long nb = (n / 4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    if (dtype == EGV_BF16)
        hipLaunchKernelGGL(dropout_add_kernel<bf16_t>, dim3((int)nb), dim3(256), 0, EGV_ST, (const bf16_t*)x, (const bf16_t*)r1, (const bf16_t*)r2, (bf16_t*)y, n, p, seed);
    else
        hipLaunchKernelGGL(dropout_add_kernel<float>, dim3((int)nb), dim3(256), 0, EGV_ST, (const float*)x, (const float*)r1, (const float*)r2, (float*)y, n, p, seed);
    EGV_LAUNCH_CHECK();
    return 0;
}
