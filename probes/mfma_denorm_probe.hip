// Does v_mfma_f32_16x16x32_f16 keep float16 SUBNORMAL inputs?  B = raw byte values in the mantissa field
// (n * 2^-24), A = small integers; compare the result with the exact sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float *out) {
    const int lane = threadIdx.x;
    u16x8 braw, araw;
    h16x8 a;
    for (int i = 0; i < 8; ++i) {
        braw[i] = (unsigned short)((lane * 7 + i * 13) & 0xff);      // subnormal f16: n * 2^-24
        a[i] = (_Float16)(float)(1 + ((lane + i) & 3));
    }
    h16x8 b = __builtin_bit_cast(h16x8, braw);
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[r];
}
int main() {
    float *d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // reference: D[i][j] = sum_k A[i][k] B[k][j]; A lane (i = lane&15, kg = lane>>4) holds k = 8 kg .. 8 kg + 7;
    // B lane (j = lane&15, kg) likewise; D lane (j = lane&15, rows 4 (lane>>4) + r)
    double maxerr = 0, maxref = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
            const int j = lane & 15, i = 4 * (lane >> 4) + r;
            double ref = 0;
            for (int kg = 0; kg < 4; ++kg)
                for (int e = 0; e < 8; ++e) {
                    const int la = i + 16 * kg, lb = j + 16 * kg;
                    const double av = 1 + ((la + e) & 3);
                    const double bv = ((lb * 7 + e * 13) & 0xff) * std::ldexp(1.0, -24);
                    ref += av * bv;
                }
            maxerr = std::fmax(maxerr, std::fabs(h[lane * 4 + r] - ref));
            maxref = std::fmax(maxref, ref);
        }
    printf("max |mfma - exact| = %.3e, max exact = %.3e (%s)\n", maxerr, maxref,
           maxerr == 0 ? "subnormal float16 inputs are kept, products exact" : "NOT exact: flushed or layout differs");
    printf("sample: got %.6e\n", h[5]);
    return 0;
}
