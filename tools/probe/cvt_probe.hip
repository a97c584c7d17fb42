// Probe of v_cvt_pk_fp8_f32 on gfx950: OCP e4m3fn, rounding, saturation, subnormals.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstdint>
#include <cstring>
__global__ void cvt(const float* x, unsigned char* y, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { int r = __builtin_amdgcn_cvt_pk_fp8_f32(x[i], 0.f, 0, false); y[i] = r & 0xff; }
}
static float e4m3(unsigned char v) {
    int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    if (e == 15 && m == 7) return NAN;
    float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7);
    return s ? -x : x;
}
// RNE to e4m3fn with saturation to 448
static unsigned char ref(float f) {
    unsigned char s = std::signbit(f) ? 0x80 : 0; float a = fabsf(f);
    if (std::isnan(f)) return s | 0x7f;
    if (a >= 448.f) { return s | 0x7e; }
    // enumerate
    int best = 0; float bd = 1e30f;
    for (int c = 0; c < 0x7f; ++c) { float d = fabsf(e4m3(c) - a); if (d < bd || (d == bd && !(c & 1))) { bd = d; best = c; } }
    return s | best;
}
int main() {
    std::vector<float> x;
    for (int e = -14; e <= 10; ++e) for (int m = 0; m < 64; ++m) { float v = ldexpf(1.f + m / 64.f, e); x.push_back(v); x.push_back(-v); }
    x.push_back(0.f); x.push_back(448.f); x.push_back(449.f); x.push_back(464.f); x.push_back(465.f); x.push_back(480.f); x.push_back(1e6f); x.push_back(INFINITY);
    int n = x.size(); float* dx; unsigned char* dy; hipMalloc(&dx, n * 4); hipMalloc(&dy, n);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    cvt<<<(n + 255) / 256, 256>>>(dx, dy, n);
    std::vector<unsigned char> y(n); hipMemcpy(y.data(), dy, n, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) { unsigned char r = ref(x[i]); if (r != y[i]) { if (bad < 20) printf("x=%g hw=0x%02x (%g) ref=0x%02x (%g)\n", x[i], y[i], e4m3(y[i]), r, e4m3(r)); ++bad; } }
    printf("cvt: %d of %d differ from RNE-saturating e4m3fn\n", bad, n);
    return 0;
}
