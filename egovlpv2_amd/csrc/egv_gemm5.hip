// Grouped weight-gradient launch for gfx950: the weight (and bias) gradients of SEVERAL Linear layers that share the token axis
//
//   dW_p[N_p, K_p] (fp32) = gate_p * dY_p[M, N_p]^T X_p[M, K_p],   db_p[N_p] = gate_p * sum_m dY_p[m, :]        p = 0 .. nprob-1
//
// in ONE persistent kernel launch that is balanced over the CUs it is given, without the fp32 slab round trip of the
// one-GEMM-per-launch form (egv_gemm4.hip + reduce_slabs_kernel: a 768 x 768 gradient alone has 9 output tiles, so filling the chip
// meant 24-28 reduction splits, 57 MB of slabs written and re-read per launch and 14-K-tile main loops; autograd of
// video_transformer.py:53,56,120,152,166,183).  A SpaceTimeBlock has six (fused: eight) such gradients = 144 (162) tiles of
// 256 x 256 and KT = 393 K-tiles of 64 tokens each.  The host plans PHASES: in a phase, G workgroups (one per CU) each run one
// (tile, reduction split) item; a phase with S splits covers floor(G / S) tiles and takes KT / S K-tiles of time.  144 tiles on
// 256 CUs: 128 tiles x 2 splits, then 16 tiles x 16 splits = 196.5 + 24.6 = 221.1 K-tile times -- the ideal 144 * 393 / 256 --
// with 196-K-tile main loops (egv_wgrad_core.h) for 8/9 of the work.
//
// The splits of a tile are summed inside the launch: a workgroup that is not the tile's last arriver publishes its partial tile
// (fp32, lane-linear 1 KiB-per-wave-instruction image, write-through `sc1` stores) and bumps the tile's `done` counter; the last
// arriver keeps its partial in registers, waits until the others have published (they ARRIVED before it, so they are running:
// no workgroup ever waits for one that may not be resident -- two such launches sharing a GPU cannot deadlock), acquires, adds the
// partials IN SPLIT ORDER (its own at its split's position: the sum does not depend on who arrives last) and writes dW / db.
// Hand-off per cdna_hip_programming.md Guideline 16 (R1: sc1 payload, every storing wave drains, one lane bumps an agent-scope
// counter; consumer: one relaxed poll loop, ONE agent acquire, plain loads); the counters are zeroed by a memset node ahead of
// every launch (or, pooled per (device, stream), put back by the last arriver of a tile).  Nothing depends on dispatch order or XCD placement.
#include "egv_wgrad_core.h"
#include <mutex>
#include "../../include/egovlp_hip.h"
#include <cstdlib>

namespace egv {

constexpr int WG_MAXP = 12;
constexpr int WG_SLAB_FLOATS = 256 * 256 + 256;           // partial tile + partial column sums of its 256 rows

struct WgProb {
    const void* A;          // dY [M, lda] bf16
    const void* B;          // X  [M, ldb] bf16
    float* C;               // dW [rows, ldc] fp32
    float* db;              // [rows] fp32 or null
    const float* gate;      // device scalar or null
    int lda, ldb, ldc;
    int tiles_m, tiles_n;   // row (N_p / 256) and column (K_p / 256) tiles
    int tile0;              // first global tile index of this problem
    int accumulate;         // 1: C += ..., db += ... (a later use of a block adds into the first use's buffer)
};
constexpr int WG_MAXPH = 8;
struct WgPhase {
    int tile0, ntile;       // tiles tile0 .. tile0 + ntile - 1 of the group
    int nsplit, kper;       // reduction splits of every tile of this phase; tokens per split (multiple of 64)
    long long slab0;        // first slab (in slabs of WG_SLAB_FLOATS floats) of this phase: [tile - tile0][split]
};
struct WgGroup {
    WgProb p[WG_MAXP];
    WgPhase ph[WG_MAXPH];
    int nprob, nphase, ntile, M;
    float* slabs;
    int* cnt;               // [2][ntile]: arrive, done
    int* err;               // sticky error word of the counter pool (null: counters zeroed per launch)
};

__global__ __launch_bounds__(512) void gemm_wgrad_group_kernel(const WgGroup g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wave_id();
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;

    // PERSISTENT: one workgroup per granted CU (a workgroup owns its CU: 136 KB of LDS, 512 threads x 226 VGPRs); in every phase
    // workgroup w runs item w = (split, tile) of the phase.  The workgroups of one XCD (blockIdx % 8: their L2) hold CONSECUTIVE
    // tiles of one split -- tiles of one row of one gradient read the same dY columns of the same tokens at the same time.
    const int G = gridDim.x;
    const int w = xcd_remap(blockIdx.x, G);
    for (int phase = 0; phase < g.nphase; ++phase) {
    const WgPhase& PH = g.ph[phase];
    if (w >= PH.ntile * PH.nsplit) continue;
    __syncthreads();                                               // LDS and the hand-off words of the previous item are free
    const int nsplit = PH.nsplit;
    const int tile = PH.tile0 + w % PH.ntile, split = w / PH.ntile;
    int pi = 0;
#pragma unroll
    for (int q = 1; q < WG_MAXP; ++q)
        if (q < g.nprob && g.p[q].tile0 <= tile) pi = q;
    const WgProb& P = g.p[pi];
    const int local = tile - P.tile0;
    // wide gradients (more column than row tiles: fc2) are walked column-major, so that consecutive tiles still form squares
    const int m0 = (P.tiles_m < P.tiles_n ? local % P.tiles_m : local / P.tiles_n) * 256;
    const int n0 = (P.tiles_m < P.tiles_n ? local / P.tiles_m : local % P.tiles_n) * 256;
    const int kbeg = split * PH.kper;
    const int kend = min(g.M, kbeg + PH.kper);
    const int KT = (kend - kbeg + 63) >> 6;

    f32x4_t acc[8][4];
    f32x4_t bacc[2];                                               // bias gradient: column sums of dY, rows w4_row(wr, wc, s, 0) + fr of the tile
    const bool has_db = (P.db != nullptr) && (n0 == 0);
    w4_mainloop(P.A, P.B, P.lda, P.ldb, g.M, m0, n0, kbeg, KT, has_db, smem, acc, bacc);

    // ---- who sums this tile?
    int my_rank = 0;
    if (nsplit > 1) {
        __syncthreads();                                           // every wave is out of the main loop: LDS is free
        int* sh = reinterpret_cast<int*>(smem);
        if (tid == 0) sh[0] = __hip_atomic_fetch_add(g.cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        my_rank = sh[0];
    }
    float* tile_slabs = g.slabs + ((size_t)PH.slab0 + (size_t)(tile - PH.tile0) * nsplit) * WG_SLAB_FLOATS;
    if (my_rank < nsplit - 1) {
        // ---- publisher: partial tile -> slab [tile][split], fragment q of thread tid at float4 index q*512 + tid
        float* slab = tile_slabs + (size_t)split * WG_SLAB_FLOATS;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, WG_SLAB_FLOATS * 4, 0x00020000);
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[a][b]), rs, (unsigned int)(((a * 4 + b) * 512 + tid) * 16), 0, 16);   // aux 16 = sc1
        if (has_db && fg == 0) {
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(bacc[sb][0]), rs,
                                                      (unsigned int)((256 * 256 + w4_row(wr, wc, sb, 0) + fr) * 4), 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every storing wave drains its write-through stores
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(g.cnt + g.ntile + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        continue;
    }
    // ---- last arriver: wait for the published partials (their owners arrived earlier, i.e. are running), acquire, sum in split order
    bool ok = true;
    if (nsplit > 1) {
        if (tid == 0) {
            unsigned int spins = 0;
            while (__hip_atomic_load(g.cnt + g.ntile + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nsplit - 1) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1u << 26)) { ok = false; break; }   // never seen; a lost publisher must not hang the device
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (ok) {
                // every publisher of this tile has made both of its increments: the counters go back to zero for the next launch on
                // this stream (they live in a per-(device, stream) pool that is zeroed once, not in front of every launch)
                __hip_atomic_store(g.cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(g.cnt + g.ntile + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (g.err) {
                // a publisher never showed up: its increments may still come, so the counters are NOT put back (a later launch could
                // otherwise read a complete count and sum unpublished slabs in silence) -- the pool is marked broken instead: every
                // launch that uses it from now on poisons what it writes, until the host re-creates the pool (egv_gemm_wgrad_group_reset)
                __hip_atomic_store(g.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (g.err && __hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ok = false;
            reinterpret_cast<int*>(smem)[1] = ok ? 1 : 0;
        }
        __syncthreads();
        ok = reinterpret_cast<int*>(smem)[1] != 0;
    }
    const float sc = P.gate ? *P.gate : 1.0f;
    const float poison = ok ? 0.f : __builtin_nanf("");
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m0 + w4_row(wr, wc, s, i) + fr;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    const int a = s * 4 + i, b = t * 2 + jp;
                    f32x4_t v = acc[a][b];
                    if (nsplit > 1) {
                        const size_t off = (size_t)((a * 4 + b) * 512 + tid) * 4;
                        f32x4_t sum = split == 0 ? v : *reinterpret_cast<const f32x4_t*>(tile_slabs + off);
                        for (int z = 1; z < nsplit; ++z) {
                            const f32x4_t pz = z == split ? v : *reinterpret_cast<const f32x4_t*>(tile_slabs + (size_t)z * WG_SLAB_FLOATS + off);
                            sum += pz;
                        }
                        v = sum;
                    }
                    v[0] = v[0] * sc + poison; v[1] = v[1] * sc + poison; v[2] = v[2] * sc + poison; v[3] = v[3] * sc + poison;
                    const int col = n0 + w4_col(wc, t, jp) + fg * 4;
                    f32x4_t* dst = reinterpret_cast<f32x4_t*>(P.C + (size_t)row * P.ldc + col);
                    if (P.accumulate) v = *dst + v;                 // (written by an earlier launch on this stream; this tile's only writer now)
                    *dst = v;
                }
        }
    if (has_db && fg == 0) {
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            const int rl = w4_row(wr, wc, sb, 0) + fr;
            float v = bacc[sb][0];
            if (nsplit > 1) {
                float sum = split == 0 ? v : tile_slabs[256 * 256 + rl];
                for (int z = 1; z < nsplit; ++z) sum += z == split ? v : tile_slabs[(size_t)z * WG_SLAB_FLOATS + 256 * 256 + rl];
                v = sum;
            }
            v = v * sc + poison;
            P.db[m0 + rl] = P.accumulate ? P.db[m0 + rl] + v : v;
        }
    }
    }   // phases
}

}  // namespace egv
extern thread_local int egv_prof_cus_hint;     // egv_api.cpp
using namespace egv;

void* egv_prof_begin(void* stream);
void egv_prof_end(void* handle, void* stream, double flops, int kind, double bytes);

static int group_cus(int cus) {
    static int ncu = 0;
    if (!ncu) {
        hipDeviceProp_t prop;
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipGetDeviceProperties(&prop, dev);
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return (cus <= 0 || cus > ncu) ? ncu : cus;
}

// Phases for R remaining tiles on G workgroups, KT K-tiles per tile: either all of them now with floor(G / R) splits, or
// floor(G / s) of them with s = ceil(G / R) splits and the rest later -- whichever takes fewer K-tile times in total.
static constexpr int WG_MAXS = 16;
static double plan_phases(int R, int G, int KT, int depth, int* tiles, int* splits, int& n) {
    if (R <= 0) { n = 0; return 0.0; }
    if (R >= G) {                                                  // a full round of whole tiles
        int t2[WG_MAXPH], s2[WG_MAXPH], n2 = 0;
        const double rest = (depth + 1 < WG_MAXPH) ? plan_phases(R - G, G, KT, depth + 1, t2, s2, n2) : 1e30;
        tiles[0] = G; splits[0] = 1;
        for (int i = 0; i < n2; ++i) { tiles[1 + i] = t2[i]; splits[1 + i] = s2[i]; }
        n = 1 + n2;
        return KT + rest;
    }
    int sf = G / R;
    if (sf > WG_MAXS) sf = WG_MAXS;
    while (sf > 1 && KT / sf < 8) --sf;
    const double all_now = (double)((KT + sf - 1) / sf);
    int sc = (G + R - 1) / R;
    double split_more = 1e30;
    int t2[WG_MAXPH], s2[WG_MAXPH], n2 = 0;
    if (sc > sf && sc <= WG_MAXS && KT / sc >= 8 && depth + 2 < WG_MAXPH && G / sc >= 1 && G / sc < R)
        split_more = (double)((KT + sc - 1) / sc) + plan_phases(R - G / sc, G, KT, depth + 1, t2, s2, n2);
    if (split_more < all_now) {
        tiles[0] = G / sc; splits[0] = sc;
        for (int i = 0; i < n2; ++i) { tiles[1 + i] = t2[i]; splits[1 + i] = s2[i]; }
        n = 1 + n2;
        return split_more;
    }
    tiles[0] = R; splits[0] = sf;
    n = 1;
    return all_now;
}

struct GroupPlan {
    int ntile, nphase, G;
    int tiles[WG_MAXPH], splits[WG_MAXPH], kper[WG_MAXPH];
    long long nslab;
};
static int group_plan(int M, int nprob, const egv_wgrad_problem* pr, int cus, GroupPlan& gp) {
    if (nprob < 1 || nprob > WG_MAXP || M < 64) return 0;
    gp.ntile = 0;
    for (int i = 0; i < nprob; ++i) {
        if ((pr[i].N % 256) || (pr[i].K % 256) || pr[i].N <= 0 || pr[i].K <= 0) return 0;
        gp.ntile += (pr[i].N / 256) * (pr[i].K / 256);
    }
    gp.G = cus;
    const int KT = (M + 63) / 64;
    plan_phases(gp.ntile, cus, KT, 0, gp.tiles, gp.splits, gp.nphase);
    {   // the recursion stops at WG_MAXPH phases: a plan that does not cover every tile (few CUs, many tiles) is no plan --
        // the caller then takes the one-gradient-per-launch form instead of leaving dW tiles unwritten
        int covered = 0;
        for (int i = 0; i < gp.nphase; ++i) covered += gp.tiles[i];
        if (covered != gp.ntile || gp.nphase < 1 || gp.nphase > WG_MAXPH) return 0;
    }
    gp.nslab = 0;
    for (int i = 0; i < gp.nphase; ++i) {
        int ns = gp.splits[i];
        int kper = (((M + ns - 1) / ns + 63) / 64) * 64;
        ns = (M + kper - 1) / kper;                                // (a split may come out empty after rounding to K-tiles)
        gp.splits[i] = ns; gp.kper[i] = kper;
        if (ns > 1) gp.nslab += (long long)gp.tiles[i] * ns;
    }
    return 1;
}

// arrive / done counters of the reduction splits: a per-stream pool, zeroed when it is created; the last arriver of a tile puts its two
// counters back to zero, so a launch finds them clean without a memset in front of it (two launches on one stream are ordered,
// launches on different streams have different pools)
// The pool's last word is its sticky error flag (set by a launch whose spin timed out: the counters are then left as they are and every
// later launch on the pool poisons its output); egv_gemm_wgrad_group_reset zeroes the pools of the calling device.
namespace {
struct WgPool { int dev; hipStream_t st; int* p; size_t n; };
WgPool g_pools[32];
int g_npool = 0;
std::mutex g_pool_mu;
}
static int* group_counters(hipStream_t st, size_t n_ints, int** err) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (int i = 0; i < g_npool; ++i)
        if (g_pools[i].dev == dev && g_pools[i].st == st && g_pools[i].n >= n_ints + 1) { *err = g_pools[i].p + g_pools[i].n - 1; return g_pools[i].p; }
    if (g_npool == 32) return nullptr;
    const size_t n = n_ints + 1 < 4096 ? 4096 : n_ints + 1;
    int* p = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(int)) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, n * sizeof(int)) != hipSuccess) { (void)hipFree(p); return nullptr; }
    g_pools[g_npool++] = WgPool{dev, st, p, n};
    *err = p + n - 1;
    return p;
}

// Zero the split-reduction counter pools (and their sticky error words) of the calling device, stream-ordered on each pool's stream.
// For a host that has seen NaN weight gradients after a device fault and wants to go on without restarting the process.
extern "C" int egv_gemm_wgrad_group_reset(void) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (int i = 0; i < g_npool; ++i)
        if (g_pools[i].dev == dev) EGV_CHECK(hipMemsetAsync(g_pools[i].p, 0, g_pools[i].n * sizeof(int), g_pools[i].st) == hipSuccess, "egv_gemm_wgrad_group_reset: memset failed");
    return 0;
}

extern "C" long long egv_gemm_wgrad_grouped_workspace_bytes(int M, int nprob, const egv_wgrad_problem* problems, int cus) {
    GroupPlan gp;
    if (!group_plan(M, nprob, problems, group_cus(cus), gp)) return -1;
    return gp.nslab * WG_SLAB_FLOATS * 4 + (long long)2 * gp.ntile * 4 + 256;
}

extern "C" int egv_gemm_wgrad_grouped(int dtype, int M, int nprob, const egv_wgrad_problem* pr, int cus, void* workspace,
                                      long long workspace_bytes, void* stream) {
    EGV_CHECK(dtype == EGV_BF16, "egv_gemm_wgrad_grouped: bf16 operands only");
    GroupPlan gp;
    cus = group_cus(cus);
    EGV_CHECK(pr && group_plan(M, nprob, pr, cus, gp), "egv_gemm_wgrad_grouped: unsupported group (1..%d problems, N and K multiples of 256, M >= 64)", WG_MAXP);
    EGV_CHECK(workspace && workspace_bytes >= egv_gemm_wgrad_grouped_workspace_bytes(M, nprob, pr, cus), "egv_gemm_wgrad_grouped: workspace too small");
    WgGroup g{};
    const int ntile = gp.ntile;
    g.nprob = nprob; g.ntile = ntile; g.M = M; g.nphase = gp.nphase;
    {
        int t0 = 0;
        long long s0 = 0;
        for (int i = 0; i < gp.nphase; ++i) {
            g.ph[i].tile0 = t0; g.ph[i].ntile = gp.tiles[i]; g.ph[i].nsplit = gp.splits[i]; g.ph[i].kper = gp.kper[i]; g.ph[i].slab0 = s0;
            t0 += gp.tiles[i];
            if (gp.splits[i] > 1) s0 += (long long)gp.tiles[i] * gp.splits[i];
        }
    }
    int t0 = 0;
    double flops = 0, bytes = 0;
    for (int i = 0; i < nprob; ++i) {
        const egv_wgrad_problem& q = pr[i];
        auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        EGV_CHECK(q.dy && q.x && q.dw && al16(q.dy) && al16(q.x) && al16(q.dw) && (q.ldy % 8) == 0 && (q.ldx % 8) == 0 && q.ldy >= q.N && q.ldx >= q.K,
                  "egv_gemm_wgrad_grouped: problem %d: operands must be 16-byte aligned with leading dimensions multiples of 8", i);
        EGV_CHECK((long long)M * q.ldy * 2 < (1LL << 31) && (long long)M * q.ldx * 2 < (1LL << 31), "egv_gemm_wgrad_grouped: problem %d: operand too large", i);
        g.p[i].A = q.dy; g.p[i].B = q.x; g.p[i].C = q.dw; g.p[i].db = q.db; g.p[i].gate = q.gate;
        g.p[i].lda = q.ldy; g.p[i].ldb = q.ldx; g.p[i].ldc = q.K;
        g.p[i].tiles_m = q.N / 256;
        g.p[i].tiles_n = q.K / 256;
        g.p[i].tile0 = t0;
        g.p[i].accumulate = q.accumulate != 0;
        t0 += (q.N / 256) * (q.K / 256);
        flops += 2.0 * M * q.N * q.K;
        bytes += 2.0 * ((double)M * q.N + (double)M * q.K) + 4.0 * q.N * q.K * (q.accumulate ? 2 : 1);
    }
    g.slabs = (float*)workspace;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    g.err = nullptr;
    g.cnt = gp.nslab > 0 ? group_counters(st, (size_t)2 * ntile, &g.err) : nullptr;
    const bool pooled = g.cnt != nullptr;
    if (!pooled) g.cnt = (int*)((char*)workspace + (size_t)gp.nslab * WG_SLAB_FLOATS * 4);   // (no pool: counters in the workspace, zeroed per launch)
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wgrad_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS);
        attr = true;
    }
    void* ph = egv_prof_begin(stream);
    if (gp.nslab > 0 && !pooled) (void)hipMemsetAsync(g.cnt, 0, (size_t)2 * ntile * 4, st);
    int nwg = 0;
    for (int i = 0; i < gp.nphase; ++i) nwg = gp.tiles[i] * gp.splits[i] > nwg ? gp.tiles[i] * gp.splits[i] : nwg;
    egv_prof_cus_hint = nwg;
    hipLaunchKernelGGL(gemm_wgrad_group_kernel, dim3(nwg), dim3(512), W4_LDS, st, g);
    egv_prof_end(ph, stream, flops, 15, bytes);                    // 15 = grouped ping-pong weight gradient
    EGV_LAUNCH_CHECK();
    return 0;
}
