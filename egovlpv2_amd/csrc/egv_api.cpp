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

extern "C" int egv_abi_version(void) { return 4; }

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
