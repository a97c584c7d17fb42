// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x fp8 e4m3): operand element order, scale lane mapping, opsel bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void probe(const unsigned char* A, const unsigned char* B, const unsigned int* SA, const unsigned int* SB, float* C, int sel) {
    const int lane = threadIdx.x, fr = lane & 15, fg = lane >> 4;
    i32x8 a, b;
    for (int w = 0; w < 8; ++w) {
        a[w] = *reinterpret_cast<const int*>(A + fr * 128 + (w >> 2) * 64 + fg * 16 + (w & 3) * 4);
        b[w] = *reinterpret_cast<const int*>(B + fr * 128 + (w >> 2) * 64 + fg * 16 + (w & 3) * 4);
    }
    const int sa = SA[lane], sb = SB[lane];
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    if (sel == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    if (sel == 1) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 1, sa, 2, sb);
    if (sel == 2) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 3, sa, 1, sb);
    for (int r = 0; r < 4; ++r) C[lane * 4 + r] = c[r];
}

static float e4m3(unsigned char v) {
    int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7);
    return s ? -x : x;
}
int main() {
    std::vector<unsigned char> A(16 * 128), B(16 * 128);
    std::vector<unsigned int> SA(64), SB(64);
    srand(1);
    for (auto& v : A) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v ^= 1; }
    for (auto& v : B) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v ^= 1; }
    for (auto& v : SA) { v = 0; for (int k = 0; k < 4; ++k) v |= (unsigned)(120 + rand() % 14) << (8 * k); }
    for (auto& v : SB) { v = 0; for (int k = 0; k < 4; ++k) v |= (unsigned)(120 + rand() % 14) << (8 * k); }
    unsigned char *dA, *dB; unsigned int *dSA, *dSB; float* dC;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dSA, 256); hipMalloc(&dSB, 256); hipMalloc(&dC, 1024);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipMemcpy(dSA, SA.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dSB, SB.data(), 256, hipMemcpyHostToDevice);
    const int selA[3] = {0, 1, 3}, selB[3] = {0, 2, 1};
    for (int sel = 0; sel < 3; ++sel) {
        probe<<<1, 64>>>(dA, dB, dSA, dSB, dC, sel);
        std::vector<float> C(256);
        hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
        // operand row i = lane&15, lane group g = lane>>4: registers 0-3 hold k = 16 g .. +15, registers 4-7 hold k = 64 + 16 g .. +15; scale of (row i, block g) = byte opsel of lane i + 16 g;
        // D[i][j] (i: A row, j: B row) in lane (j + 16*(i/4)), reg i%4
        double maxerr = 0, maxref = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double ref = 0;
                for (int g = 0; g < 4; ++g) {
                    double s = 0;
                    for (int k = 0; k < 32; ++k) s += (double)e4m3(A[i * 128 + g * 32 + k]) * e4m3(B[j * 128 + g * 32 + k]);
                    const int ea = (SA[i + 16 * g] >> (8 * selA[sel])) & 0xff, eb = (SB[j + 16 * g] >> (8 * selB[sel])) & 0xff;
                    ref += s * ldexp(1.0, ea - 127) * ldexp(1.0, eb - 127);
                }
                const double got = C[(j + 16 * (i / 4)) * 4 + (i % 4)];
                maxerr = fmax(maxerr, fabs(got - ref)); maxref = fmax(maxref, fabs(ref));
            }
        printf("sel %d: max |err| %.3e of max |ref| %.3e\n", sel, maxerr, maxref);
    }
    return 0;
}
