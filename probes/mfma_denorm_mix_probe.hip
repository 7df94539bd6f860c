// One v_mfma_f32_16x16x32_f16 with the operand mix of the sparse kernel: K slots [lo, lo, hi, hi | ...] of float16
// subnormals (bytes * 2^-24) against [w, w, 256 w, 256 w | ...], w = 78.125, lo = 100, hi = 1 .. 15, into C = 0 and
// C = 3: is the sum exact?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float *out, int hi, float c0, int normal_b) {
    const int lane = threadIdx.x;
    u16x8 braw = {100, 100, (unsigned short)hi, (unsigned short)hi, 100, 100, (unsigned short)hi, (unsigned short)hi};
    const _Float16 w = (_Float16)78.125f, w256 = (_Float16)20000.0f;
    h16x8 a = {w, w, w256, w256, w, w, w256, w256}, b;
    if (normal_b) for (int i = 0; i < 8; ++i) b[i] = (_Float16)(float)braw[i];
    else b = __builtin_bit_cast(h16x8, braw);
    f32x4 acc = {c0, c0, c0, c0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    out[lane] = acc[0];
}
int main() {
    float *d; hipMalloc(&d, 64 * 4);
    for (int nb = 0; nb < 2; ++nb)
        for (float c0 : {0.0f, 3.0f})
            for (int hi : {1, 7, 15, 200}) {
                hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, hi, nb ? c0 * 16777216.0f : c0, nb);
                float h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
                const double sc = nb ? 1.0 : std::ldexp(1.0, -24);
                const double exact = (nb ? c0 * 16777216.0 : c0) + 4 * (2 * 78.125 * 100 + 2 * 20000.0 * hi) * sc;
                printf("B %s C=%g hi=%3d: got %.9e exact %.9e rel diff %+.2e\n", nb ? "normal   " : "subnormal", c0, hi,
                       h[0], exact, (h[0] - exact) / exact);
            }
    return 0;
}
