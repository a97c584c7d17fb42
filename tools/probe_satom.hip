// probe: do scalar memory atomics (s_atomic_add, returning) work on gfx950?  hipcc --offload-arch=gfx950 -O3 tools/probe_satom.hip -o /tmp/probe_satom && /tmp/probe_satom
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(unsigned int* ctr, unsigned int* out) {
    unsigned int v = 1;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ctr) : "memory");
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int main() {
    const int N = 4096;
    unsigned int *ctr, *out;
    hipMalloc(&ctr, 64); hipMalloc(&out, N * 4);
    hipMemset(ctr, 0, 64);
    hipLaunchKernelGGL(k, dim3(N), dim3(256), 0, 0, ctr, out);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<unsigned int> h(N); unsigned int c;
    hipMemcpy(h.data(), out, N * 4, hipMemcpyDeviceToHost); hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    bool ok = c == N * 4u;        // 4 waves per workgroup each add 1
    printf("counter %u (expect %u)  min %u max %u  %s\n", c, N * 4u, h.front(), h.back(), ok ? "OK" : "MISMATCH");
    return 0;
}
