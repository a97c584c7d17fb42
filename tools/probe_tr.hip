// probe: semantics of ds_read_b64_tr_b16 on gfx950 (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(const int* addr, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
    int h_addr[64]; unsigned short h_out[256];
    // hypothesis: out[i][j] = X[4j + i/4][i%4] per 16-lane group, X[t] = 4 elements at lane t's address
    for (int l = 0; l < 64; ++l) { int g = l >> 4, t = l & 15; h_addr[l] = g * 1000 + (t / 4) * 100 + (t % 4) * 4; }  // rows 100 apart
    int* d_addr; unsigned short* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        int g = l >> 4, i = l & 15;
        for (int j = 0; j < 4; ++j) {
            int t = 4 * j + i / 4, e = i % 4;
            int exp = (g * 1000 + (t / 4) * 100 + (t % 4) * 4) + e;     // X[t][e]
            // equivalently element (row j, col i) of the 4x16 block
            if (h_out[l * 4 + j] != exp) ++bad;
        }
    }
    printf("hypothesis mismatches: %d\n", bad);
    for (int l = 0; l < 20; ++l) printf("lane %d: %d %d %d %d\n", l, h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3]);
    return 0;
}
