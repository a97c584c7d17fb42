// Error plumbing, ABI version and HIP-event instrumentation for libegovlp_hip.so
// (C ABI declared in include/egovlp_hip.h).
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

extern "C" const char* egv_last_error(void) { return g_err; }

void egv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int egv_abi_version(void) { return 6; }

// ---- switches --------------------------------------------------------------------------------------------------------------
// Every run-time switch of the library in ONE table (name, default, what it does).  The defaults are the configuration that is
// benchmarked and tested; an environment variable of the same name overrides one switch (A/B aid).  egv_cfg_* is the only place
// that reads the environment: a name that is not in the table, or a call site that states another default than the table, is a
// programming error and aborts.  egv_config_dump() lists name, default and current value (tests/test_abi_and_host.py pins the
// defaults; INTEGRATION.md explains the switches).
#include <cstdlib>
#include <cstring>
#include <string>
namespace {
struct Switch { const char* name; double def; const char* doc; };
const Switch g_switches[] = {
    {"EGV_ATTN_FEWKEYS", 1, "one-launch forward / backward of many queries over <= 32 keys (image-to-text cross attention), egv_attn_cross.hip"},
    {"EGV_ATTN_FEWKEYS_ITERS", 4, "... 32-query tiles per wave (a workgroup covers 128 x this many queries and leaves one partial dK / dV)"},
    {"EGV_ATTN_FEWQ", 1, "one-launch forward / backward of <= 32 queries over many keys (text-to-image cross attention), egv_attn_cross.hip"},
    {"EGV_ATTN_FEWQ_ITERS", 6, "... 32-key tiles per wave (a workgroup covers 128 x this many keys and leaves one partial state / dQ)"},
    {"EGV_ATTN_TIME_FUSED", 1, "one-launch forward / backward of the <= 16-row attention groups (time attention), egv_attn_time.hip"},
    {"EGV_ATTN_SPACE_NEW", 1, "space attention on row-major LDS images, egv_attn_space.hip (0: the kernels of egv_attn_mfma.hip)"},
    {"EGV_ATTN_FUSED_CLS", 1, "the group launches also serve the CLS row (per-group partials + one small sum)"},
    {"EGV_ATTN_FUSED_BWD", 1, "one-launch attention backward where a kernel covers the shape (0: dQ + dK/dV kernel pair)"},
    {"EGV_WGRAD_SMALL_M", 1, "weight gradients over <= 16 rows as an outer product (wgrad_small_m_kernel) instead of a one-K-step MFMA GEMM"},
    {"EGV_WGRAD_PP", 1, "ping-pong 256x256 weight-gradient kernel (0: 256x128 ring kernel)"},
    {"EGV_WGRAD_ITEMS", 224, "(tile, split) items of a one-gradient-per-launch weight gradient: 7/8 of the CUs"},
    {"EGV_GEMM_PP", 1, "persistent ping-pong GEMM for large grids (0: DMA-ring kernels only)"},
    {"EGV_PP_STAMPS", 0, "instrumentation build only: per-K-tile cycle stamps of the persistent GEMM"},
    {"EGV_PP_CUS", 0, "cap of the persistent GEMM's grid (0: all CUs)"},
    {"EGV_PP_LIMIT_SLACK", 16, "CUs a persistent grid may take beyond its CU limit when that removes a round of its tile walk"},
    {"EGV_PP_LIMIT_SLACK_FUSED", 0, "the same inside a fused video block's backward call (its weight-gradient launch stays resident: exact limit)"},
    {"EGV_PP_BM192", 1, "192-row tiles where they shorten the walk"},
    {"EGV_PP_MIXED", 0, "224- / 160- / 128-row tiles (A sub-tiles of different heights) for the plain kinds where they shorten the walk: 1 = grids planned for the whole chip, 2 = also under a CU limit, 0 = off (isolated launches gain 2-7 %, the step nothing: profiles/round5_experiments.md section 4)"},
    {"EGV_PP_FORCE_BM", 0, "tests: tile height of the persistent GEMM wherever the epilogue kind is built for it (0: chosen per call)"},
    {"EGV_PP_TILE_C0", 56, "height-independent cost of a tile of the persistent GEMM, in rows (tile cost = rows + this)"},
    {"EGV_PP_TRIM", 1, "grid trimmed to the smallest size that keeps the round count"},
    {"EGV_PP_RES_MINK", 1536, "shortest K for which a residual-epilogue GEMM takes the persistent kernel on 256-row tiles"},
    {"EGV_LN_BLOCKS", 512, "workgroup cap of the LayerNorm backward"},
    {"EGV_LN_DEFER", 1, "video block backward: the LayerNorm parameter-gradient partials of the call's passes are summed by one launch at its end"},
    {"EGV_LN_PACKED", 1, "packed four-rows-per-wave LayerNorm backward (bf16, D = 768 / 1024)"},
    {"EGV_GELU_DERIV", 0, "video MLP saves gelu'(x) instead of x in the bf16 mode (EGV_ACT_GELU_D; measured slower, off)"},
    {"EGV_WGRAD_GROUP", 1, "all weight gradients of a video block call as one persistent grouped launch"},
    {"EGV_WGRAD_CUS", 0, "CU grant of the grouped weight-gradient launch (0: 2/3 CU per output tile)"},
    {"EGV_WGRAD_TAIL_CUS", 0, "CU grant of the LAST grouped weight-gradient launch of a backward pass (EGV_BLOCK_TAIL; 0: 7/8 of the CUs)"},
    {"EGV_WGRAD_CUS_FUSED", 0, "CU grant of a fused block's grouped launch (0: as EGV_WGRAD_CUS / 2/3 CU per output tile)"},
    {"EGV_FUSED_LIMIT_LIFT", 1, "fused block backward: the GEMMs after the image-to-text part plan for the whole chip"},
    {"EGV_WGRAD_DEFER_MAXTILES", 192, "largest group whose launch is left running beside the next block call"},
    {"EGV_WGRAD_MAIN_LIMIT", 0, "CU limit of the calling stream's grids beside a grouped launch (0: the CUs the grant leaves)"},
    {"EGV_TEXT_WGRAD_GROUP", 1, "weight gradients over the text rows of a RoBERTa layer as one grouped launch"},
    {"EGV_MX_EPI_QUANT", 1, "MX-fp8 path: GELU / GELU' epilogues also emit the quantised form of their output"},
    {"EGV_LN_MX", 1, "MX-fp8 path: LayerNorm forward also writes the quantised form of its output"},
};
const Switch* find_switch(const char* name) {
    for (const Switch& s : g_switches)
        if (!std::strcmp(s.name, name)) return &s;
    return nullptr;
}
std::string g_dump;
}  // namespace

double egv_cfg_f64(const char* name, double def) {
    const Switch* s = find_switch(name);
    if (!s || s->def != def) {
        std::fprintf(stderr, "egovlp_hip: switch %s (default %g) is not in the switch table of egv_api.cpp with that default\n", name, def);
        std::abort();
    }
    const char* v = std::getenv(name);
    return (v && *v) ? std::atof(v) : def;
}
int egv_cfg_int(const char* name, int def) { return (int)egv_cfg_f64(name, (double)def); }
bool egv_cfg_on(const char* name, bool def) { return egv_cfg_f64(name, def ? 1.0 : 0.0) != 0.0; }

// "NAME default current doc\n" per switch (tab separated)
extern "C" const char* egv_config_dump(void) {
    g_dump.clear();
    char line[512];
    for (const Switch& s : g_switches) {
        const char* v = std::getenv(s.name);
        std::snprintf(line, sizeof(line), "%s\t%g\t%g\t%s\n", s.name, s.def, (v && *v) ? std::atof(v) : s.def, s.doc);
        g_dump += line;
    }
    return g_dump.c_str();
}

extern "C" int egv_stream_create(int priority, void** stream) {
    if (!stream) { egv_set_error("egv_stream_create: null output pointer"); return -1; }
    int least = 0, greatest = 0;
    hipDeviceGetStreamPriorityRange(&least, &greatest);          // numerically: greatest <= least
    if (priority > least) priority = least;
    if (priority < greatest) priority = greatest;
    hipStream_t s = nullptr;
    const hipError_t e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority);
    if (e != hipSuccess) { egv_set_error("egv_stream_create: %s", hipGetErrorString(e)); return -1; }
    *stream = reinterpret_cast<void*>(s);
    return 0;
}

// ---- per-launch HIP-event timing of the GEMM kernels, on the stream they are launched on ----
namespace {
struct Rec { hipEvent_t a, b; double flops, bytes; int kind, cus; };
std::mutex g_mu;
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e; hipEventCreate(&e); return e;
}
}  // namespace

bool egv_prof_on() { return g_on; }
// CUs a persistent launch was planned for (its grid): set by the launcher right before egv_prof_end picks it up
thread_local int egv_prof_cus_hint = 0;

void* egv_prof_begin(void* stream) {
    if (!g_on) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    Rec r; r.a = get_event(); r.b = get_event(); r.flops = 0; r.bytes = 0; r.kind = 0; r.cus = 0;
    hipEventRecord(r.a, reinterpret_cast<hipStream_t>(stream));
    g_recs.push_back(r);
    return reinterpret_cast<void*>(g_recs.size());      // 1-based handle
}

void egv_prof_end(void* handle, void* stream, double flops, int kind, double bytes) {
    const int cus = egv_prof_cus_hint;
    egv_prof_cus_hint = 0;
    if (!handle) return;
    std::lock_guard<std::mutex> lk(g_mu);
    Rec& r = g_recs[reinterpret_cast<size_t>(handle) - 1];
    r.flops = flops; r.kind = kind; r.bytes = bytes; r.cus = cus;
    hipEventRecord(r.b, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int egv_prof_enable(int on) { g_on = on != 0; return 0; }

extern "C" int egv_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
    g_recs.clear();
    return 0;
}

extern "C" int egv_prof_collect3(double* flops, double* bytes, float* ms, int* kind, int* cus, int max_records);
extern "C" int egv_prof_collect2(double* flops, double* bytes, float* ms, int* kind, int max_records) {
    return egv_prof_collect3(flops, bytes, ms, kind, nullptr, max_records);
}
extern "C" int egv_prof_collect(double* flops, float* ms, int* kind, int max_records) {
    return egv_prof_collect2(flops, nullptr, ms, kind, max_records);
}
extern "C" int egv_prof_collect3(double* flops, double* bytes, float* ms, int* kind, int* cus, int max_records) {
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    for (auto& r : g_recs) {
        if (n >= max_records) break;
        hipEventSynchronize(r.b);
        float t = 0.f;
        hipEventElapsedTime(&t, r.a, r.b);
        flops[n] = r.flops; ms[n] = t; kind[n] = r.kind;
        if (bytes) bytes[n] = r.bytes;
        if (cus) cus[n] = r.cus;
        ++n;
    }
    return n;
}
