// Exploratory probe: which operand bytes does the scale byte of lane L apply to?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void probe(const unsigned char* A, const unsigned char* B, const unsigned int* SA, const unsigned int* SB, float* C) {
    const int lane = threadIdx.x;
    i32x8 a, b;
    for (int w = 0; w < 8; ++w) {   // lane-linear: lane's 32 bytes
        a[w] = *reinterpret_cast<const int*>(A + lane * 32 + w * 4);
        b[w] = *reinterpret_cast<const int*>(B + lane * 32 + w * 4);
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, (int)SA[lane], 0, (int)SB[lane]);
    for (int r = 0; r < 4; ++r) C[lane * 4 + r] = c[r];
}
unsigned char *dA, *dB; unsigned int *dSA, *dSB; float* dC;
std::vector<float> run(const std::vector<unsigned char>& A, const std::vector<unsigned char>& B, const std::vector<unsigned int>& SA, const std::vector<unsigned int>& SB) {
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(dSA, SA.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dSB, SB.data(), 256, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dSA, dSB, dC);
    std::vector<float> C(256); hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
    return C;
}
int main() {
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dSA, 256); hipMalloc(&dSB, 256); hipMalloc(&dC, 1024);
    std::vector<unsigned char> A(2048, 0x38), B(2048, 0x38);
    std::vector<unsigned int> S1(64, 0x7f7f7f7fu);
    auto base = run(A, B, S1, S1);
    printf("all ones: D[0]=%g D[255]=%g\n", base[0], base[255]);
    for (int which = 0; which < 2; ++which)
    for (int L : {0, 5, 16, 37, 63}) {
        auto SA = S1, SB = S1;
        (which ? SB : SA)[L] = 0x7f7f7f80u;
        auto c = run(A, B, SA, SB);
        printf("%s scale lane %d byte0 x2: ", which ? "B" : "A", L);
        int cnt = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (c[l * 4 + r] != base[l * 4 + r]) { if (cnt < 4) printf("[lane %d reg %d: +%g] ", l, r, c[l * 4 + r] - base[l * 4 + r]); ++cnt; }
        printf(" (%d outputs changed)\n", cnt);
        // which operand bytes are in the scaled block: zero 16-byte halves of each lane
        for (int l2 = 0; l2 < 64; ++l2) for (int h = 0; h < 2; ++h) {
            auto X = which ? B : A;
            for (int k = 0; k < 16; ++k) X[l2 * 32 + h * 16 + k] = 0;
            auto c0 = which ? run(A, X, S1, S1) : run(X, B, S1, S1);
            auto c1 = which ? run(A, X, SA, SB) : run(X, B, SA, SB);
            double d = 0, d0 = 0;
            for (int q = 0; q < 256; ++q) { d += c1[q] - c0[q]; d0 += c[q] - base[q]; }
            if (d != d0) printf("    bytes of lane %d half %d are in the block (diff sum %g -> %g)\n", l2, h, d0, d);
        }
    }
    // byte 1..3 via opsel is checked by the main probe once the mapping is known
    return 0;
}
