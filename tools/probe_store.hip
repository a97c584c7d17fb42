// Probe: what does a wave's buffer_store_dwordx4 cost as a function of the lane -> address map?  (round 5: the persistent GEMM's
// epilogue stores its 128 KB tile in ~8.7k cycles = 68 cycles per 1 KiB store instruction, four times a load's 16.)
//   map 0: lane-linear            lane l -> base + l*16                         (1 KiB contiguous)
//   map 1: the GEMM epilogue's    row l&7, 16-byte chunk (l>>3)                 (8 rows x 128 B, a line's 8 chunks 8 lanes apart)
//   map 2: line-major             row l>>3, chunk l&7                           (8 rows x 128 B, a line = 8 ADJACENT lanes)
//   map 3: C-layout half lines    row l&15, chunk l>>4                          (16 rows x 64 B)
//   map 4: line pairs             row l>>4 (4 rows), chunk l&15                 (4 rows x 256 B)
// Rows are `pitch` bytes apart (1536 = a [M,768] bf16 matrix).  Every workgroup (8 waves) writes its own region; `per` stores per wave
// back to back, repeated; cycles per store instruction per CU = elapsed shader ticks * waves / stores.
// hipcc --offload-arch=gfx950 -O3 tools/probe_store.hip -o tools/probe/probe_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

template <int MAP>
__global__ __launch_bounds__(512) void store_kernel(unsigned char* out, int pitch, int reps, int per, long long* ticks, int aux_nt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the workgroup's region: 8 waves x per stores x (rows of the map) rows
    int row, chunk, rows_per;
    if (MAP == 0) { row = 0; chunk = lane; rows_per = 1; }
    else if (MAP == 1) { row = lane & 7; chunk = lane >> 3; rows_per = 8; }
    else if (MAP == 2) { row = lane >> 3; chunk = lane & 7; rows_per = 8; }
    else if (MAP == 3) { row = lane & 15; chunk = lane >> 4; rows_per = 16; }
    else { row = lane >> 4; chunk = lane & 15; rows_per = 4; }
    const size_t region = (size_t)pitch * rows_per * per * 8;             // bytes a workgroup covers per rep (rows x pitch)
    unsigned char* base = out + (size_t)blockIdx.x * region + (size_t)wave * per * rows_per * pitch;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    u32x4_t v = {(unsigned)lane, (unsigned)wave, 3u, 4u};
    __syncthreads();
    const long long c0 = clock64();
    for (int r = 0; r < reps; ++r) {
        for (int i = 0; i < per; ++i) {
            const unsigned int off = (unsigned int)((i * rows_per + row) * pitch + chunk * 16 + (MAP == 0 ? 0 : (r & 3) * 128));
            if (aux_nt) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 2);
            else __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
        }
        v[2] += 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long c1 = clock64();
    if (threadIdx.x == 0) ticks[blockIdx.x] = c1 - c0;
}

int main() {
    CK(hipSetDevice(0));
    const int pitch = 1536, per = 16, reps = 64;
    unsigned char* out;
    const size_t bytes = (size_t)256 * pitch * 16 * per * 8 + (1 << 20);
    CK(hipMalloc(&out, bytes));
    long long* ticks;
    CK(hipMalloc(&ticks, 256 * 8));
    auto run = [&](int map, int grid, int nt) {
        for (int rep = 0; rep < 2; ++rep) {
            if (map == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(grid), dim3(512), 0, 0, out, pitch, reps, per, ticks, nt);
            if (map == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(grid), dim3(512), 0, 0, out, pitch, reps, per, ticks, nt);
            if (map == 2) hipLaunchKernelGGL(store_kernel<2>, dim3(grid), dim3(512), 0, 0, out, pitch, reps, per, ticks, nt);
            if (map == 3) hipLaunchKernelGGL(store_kernel<3>, dim3(grid), dim3(512), 0, 0, out, pitch, reps, per, ticks, nt);
            if (map == 4) hipLaunchKernelGGL(store_kernel<4>, dim3(grid), dim3(512), 0, 0, out, pitch, reps, per, ticks, nt);
            CK(hipDeviceSynchronize());
        }
        std::vector<long long> h(grid);
        CK(hipMemcpy(h.data(), ticks, grid * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double med = (double)h[grid / 2];
        const double n = (double)reps * per * 8;          // store instructions per CU
        printf("map %d grid %3d %s: %7.1f ticks per store instruction (1 KiB) per CU = %5.1f B/tick/CU   (median workgroup %.0f ticks)\n", map, grid,
               nt ? "nt" : "  ", med / n, 1024.0 * n / med, med);
    };
    for (int grid : {1, 32, 256})
        for (int map = 0; map < 5; ++map) run(map, grid, 0);
    for (int map = 0; map < 5; ++map) run(map, 256, 1);
    printf("done\n");
    return 0;
}
