// Probes for the "partition the chip by whole XCDs" question (round 5):
//   (1) how do the bits of a hipExtStreamCreateWithCUMask mask map to (XCC, SE, CU)?  The masks used here leave NO XCC empty
//       under either hypothesis (bit i -> XCC i % 8, or 32 consecutive bits per XCC): a queue whose XCC has no CU may never drain.
//   (2) side-by-side factor of pure matrix-pipe work (no memory traffic at all): an MFMA loop on 96 CUs alone, on 256 CUs, and
//       on 96 + 160 CUs from two streams at once -- with the effective shader clock of every run (s_memtime ticks per 100 MHz
//       s_memrealtime tick).  If 96 CUs slow down beside 160 busy ones here, the coupling is power / clock, not L2.
//   (3) the same pair with the two launches on CU-masked streams (3 + 5 XCDs; workgroups that land on a foreign XCC exit at once).
// hipcc --offload-arch=gfx950 -O3 tools/probe_cumask.hip -o /tmp/probe_cumask && /tmp/probe_cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <map>
#include <set>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ unsigned int xcc_id() {
    unsigned int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}
__device__ __forceinline__ unsigned int hw_id() {
    unsigned int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    return v;
}

// one record per workgroup: xcc | hw_id
__global__ void where_kernel(unsigned int* out, int spin) {
    extern __shared__ unsigned char lds[];
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = xcc_id();
        out[blockIdx.x * 2 + 1] = hw_id();
    }
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { }
    if (lds[threadIdx.x] == 123 && spin < 0) out[0] = 1;
}

// MFMA loop: 8 waves, 32 accumulators of v_mfma_f32_16x16x32_bf16 per wave, operands = random bf16 bit patterns from a hash
// rec[wg] = {xcc, t_wall_start, t_wall_end, shader ticks}; xcc_mask: workgroups on an XCC outside the mask exit at once
__global__ __launch_bounds__(512) void mfma_kernel(long long* rec, int iters, unsigned int xcc_mask, float* sink) {
    extern __shared__ unsigned char lds[];
    const unsigned int xcc = xcc_id();
    if (!((xcc_mask >> xcc) & 1u)) {
        if (threadIdx.x == 0) { rec[blockIdx.x * 4] = -1 - (long long)xcc; rec[blockIdx.x * 4 + 1] = rec[blockIdx.x * 4 + 2] = rec[blockIdx.x * 4 + 3] = 0; }
        return;
    }
    unsigned int h = (threadIdx.x + 1) * 2654435761u ^ (blockIdx.x * 40503u);
    unsigned int w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h = h * 1664525u + 1013904223u;
        // bf16 pairs with exponents around 1.0 and random sign / mantissa: full-range toggling like real activations
        w[i] = (h & 0x807f807fu) | 0x3f003f00u | ((h >> 3) & 0x00800080u);
    }
    bf16x8_t a, b;
    memcpy(&a, &w[0], 16);
    memcpy(&b, &w[4], 16);
    f32x4_t acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const long long w0 = wall_clock64();
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    const long long c1 = clock64();
    const long long w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) sink[0] = s + lds[0];
    if (threadIdx.x == 0) {
        rec[blockIdx.x * 4] = xcc;
        rec[blockIdx.x * 4 + 1] = w0;
        rec[blockIdx.x * 4 + 2] = w1;
        rec[blockIdx.x * 4 + 3] = c1 - c0;
    }
}

// streaming reader: every workgroup walks its slice of a large buffer with 16-byte loads (HBM traffic, no matrix work)
__global__ __launch_bounds__(512) void stream_kernel(const uint4* buf, size_t n16, int passes, long long* rec, float* sink) {
    extern __shared__ unsigned char lds[];
    const long long w0 = wall_clock64();
    unsigned int acc = 0;
    for (int p = 0; p < passes; ++p)
        for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 512) {
            const uint4 v = buf[i];
            acc ^= v.x + v.y + v.z + v.w;
        }
    const long long w1 = wall_clock64();
    if (acc == 0x12345678u) sink[1] = (float)acc + lds[0];
    if (threadIdx.x == 0) { rec[blockIdx.x * 4] = xcc_id(); rec[blockIdx.x * 4 + 1] = w0; rec[blockIdx.x * 4 + 2] = w1; rec[blockIdx.x * 4 + 3] = 0; }
}

static void report_where(const char* tag, hipStream_t st, unsigned int* dout) {
    const int N = 4096;
    CK(hipMemsetAsync(dout, 0xff, N * 8, st));
    hipLaunchKernelGGL(where_kernel, dim3(N), dim3(64), 60000, st, dout, 2000);     // 60 KB of LDS: two per CU, 20 us each
    CK(hipStreamSynchronize(st));
    std::vector<unsigned int> h(N * 2);
    CK(hipMemcpy(h.data(), dout, N * 8, hipMemcpyDeviceToHost));
    std::map<int, std::set<unsigned int>> cus;
    std::map<int, int> wgs;
    int agree = 0;
    for (int i = 0; i < N; ++i) {
        const int x = h[2 * i] & 15;
        const unsigned int hw = h[2 * i + 1];
        cus[x].insert(hw & 0xff00u);               // cu_id [11:8], sh_id [12], se_id [15:13]
        wgs[x]++;
        if (x == i % 8) ++agree;
    }
    printf("%s: CUs used per XCC:", tag);
    for (int x = 0; x < 8; ++x) printf(" %d", (int)cus[x].size());
    printf("   workgroups per XCC:");
    for (int x = 0; x < 8; ++x) printf(" %d", wgs[x]);
    printf("   (block b on XCC b %% 8: %d of %d)\n", agree, N);
}

struct Run { double us, ghz; int n; };
static Run summarize(const std::vector<long long>& r, int nwg) {
    long long lo = 1LL << 62, hi = 0;
    double ticks = 0, wall = 0;
    int n = 0;
    for (int i = 0; i < nwg; ++i) {
        if (r[i * 4] < 0) continue;
        lo = std::min(lo, r[i * 4 + 1]);
        hi = std::max(hi, r[i * 4 + 2]);
        ticks += (double)r[i * 4 + 3];
        wall += (double)(r[i * 4 + 2] - r[i * 4 + 1]);
        ++n;
    }
    Run x;
    x.n = n;
    x.us = n ? (hi - lo) / 100.0 : 0;                 // 100 MHz
    x.ghz = n ? ticks / wall * 0.1 : 0;               // shader ticks per 10 ns
    return x;
}

int main(int argc, char** argv) {
    int dev = 0;
    CK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    unsigned int* dout;
    CK(hipMalloc(&dout, 4096 * 8));
    hipStream_t plain;
    CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
    report_where("no mask", plain, dout);

    // mask A: bits i with i % 8 in {0,1,2}, plus bits 3..7 (no XCC empty under either hypothesis)
    // mask B: the complement of {i % 8 in {0,1,2}}, plus bits 0..2
    unsigned int mA[8], mB[8];
    memset(mA, 0, sizeof(mA));
    memset(mB, 0, sizeof(mB));
    for (int i = 0; i < 256; ++i) {
        const bool lowx = (i % 8) < 3;
        if (lowx || i < 8) mA[i / 32] |= 1u << (i % 32);
        if (!lowx || i < 8) mB[i / 32] |= 1u << (i % 32);
    }
    hipStream_t sA, sB;
    CK(hipExtStreamCreateWithCUMask(&sA, 8, mA));
    CK(hipExtStreamCreateWithCUMask(&sB, 8, mB));
    report_where("mask A (i%8<3 + bits 0..7)", sA, dout);
    report_where("mask B (i%8>=3 + bits 0..7)", sB, dout);
    // mask C: the first 96 bits + one bit in every later group of 32 (contiguous hypothesis: XCCs 0-2 full, one CU elsewhere)
    unsigned int mC[8];
    memset(mC, 0, sizeof(mC));
    for (int i = 0; i < 256; ++i)
        if (i < 96 || (i % 32) == 0) mC[i / 32] |= 1u << (i % 32);
    hipStream_t sC;
    CK(hipExtStreamCreateWithCUMask(&sC, 8, mC));
    report_where("mask C (bits 0..95 + every 32nd)", sC, dout);

    // ---- MFMA side-by-side
    long long *recA, *recB;
    float* sink;
    CK(hipMalloc(&recA, 256 * 32));
    CK(hipMalloc(&recB, 256 * 32));
    CK(hipMalloc(&sink, 64));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const int iters = argc > 1 ? atoi(argv[1]) : 6000;       // 6000 x 32 MFMAs x 2 waves per SIMD x 16 cycles ~ 2.9 ms at 2.1 GHz
    auto run = [&](const char* tag, int gA, hipStream_t stA, unsigned int maskA, int gB, hipStream_t stB, unsigned int maskB) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            if (gA) hipLaunchKernelGGL(mfma_kernel, dim3(gA), dim3(512), 100 * 1024, stA, recA, iters, maskA, sink);
            if (gB) hipLaunchKernelGGL(mfma_kernel, dim3(gB), dim3(512), 100 * 1024, stB, recB, iters, maskB, sink);
            CK(hipDeviceSynchronize());
            std::vector<long long> ra(256 * 4), rb(256 * 4);
            CK(hipMemcpy(ra.data(), recA, 256 * 32, hipMemcpyDeviceToHost));
            CK(hipMemcpy(rb.data(), recB, 256 * 32, hipMemcpyDeviceToHost));
            Run a = summarize(ra, gA), b = gB ? summarize(rb, gB) : Run{0, 0, 0};
            printf("%-44s rep %d: A %3d wgs %8.1f us %.3f GHz", tag, rep, a.n, a.us, a.ghz);
            if (gB) printf("   B %3d wgs %8.1f us %.3f GHz", b.n, b.us, b.ghz);
            printf("\n");
        }
    };
    run("96 alone", 96, s1, 0xff, 0, s2, 0);
    run("160 alone", 160, s1, 0xff, 0, s2, 0);
    run("256 alone", 256, s1, 0xff, 0, s2, 0);
    run("96 + 160 two streams", 96, s1, 0xff, 160, s2, 0xff);
    // masked streams, whole-XCD partition: 256-workgroup grids, foreign-XCC workgroups exit at once
    run("XCC 0-2 alone (masked stream A)", 256, sA, 0x07, 0, sB, 0);
    run("XCC 3-7 alone (masked stream B)", 256, sB, 0xf8, 0, sA, 0);
    run("XCC 0-2 | XCC 3-7 (masked streams)", 256, sA, 0x07, 256, sB, 0xf8);
    // disjoint whole-XCD partition with one reserved CU per XCC for the other side's foreign workgroups (bit i -> XCC i % 8, CU i / 8)
    unsigned int mA2[8], mB2[8];
    memset(mA2, 0, sizeof(mA2));
    memset(mB2, 0, sizeof(mB2));
    for (int i = 0; i < 256; ++i) {
        const bool lowx = (i % 8) < 3, first = (i / 8) == 0;
        if ((lowx && !first) || (!lowx && first)) mA2[i / 32] |= 1u << (i % 32);
        else mB2[i / 32] |= 1u << (i % 32);
    }
    hipStream_t sA2, sB2;
    CK(hipExtStreamCreateWithCUMask(&sA2, 8, mA2));
    CK(hipExtStreamCreateWithCUMask(&sB2, 8, mB2));
    report_where("mask A2 (XCC 0-2 w/o CU 0, CU 0 of XCC 3-7)", sA2, dout);
    report_where("mask B2 (complement)", sB2, dout);
    run("A2 alone", 256, sA2, 0x07, 0, sB2, 0);
    run("B2 alone", 256, sB2, 0xf8, 0, sA2, 0);
    run("A2 | B2 (disjoint masked streams)", 256, sA2, 0x07, 256, sB2, 0xf8);
    {   // MFMA on 96 CUs beside an HBM streaming reader on 160 (no matrix work there): memory-side / power coupling without MFMA power
        uint4* big;
        const size_t bytes = (size_t)2 << 30;
        CK(hipMalloc(&big, bytes));
        CK(hipMemset(big, 1, bytes));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(mfma_kernel, dim3(96), dim3(512), 100 * 1024, s1, recA, iters, 0xffu, sink);
            hipLaunchKernelGGL(stream_kernel, dim3(160), dim3(512), 100 * 1024, s2, big, bytes / 16, 6, recB, sink);
            CK(hipDeviceSynchronize());
            std::vector<long long> ra(256 * 4), rb(256 * 4);
            CK(hipMemcpy(ra.data(), recA, 256 * 32, hipMemcpyDeviceToHost));
            CK(hipMemcpy(rb.data(), recB, 256 * 32, hipMemcpyDeviceToHost));
            Run a = summarize(ra, 96), b = summarize(rb, 160);
            printf("96 MFMA + 160 streaming readers            rep %d: A %3d wgs %8.1f us %.3f GHz   B %3d wgs %8.1f us = %.2f TB/s\n", rep, a.n, a.us, a.ghz, b.n,
                   b.us, 6.0 * bytes / b.us * 1e-6);
        }
    }
    printf("done\n");
    return 0;
}
