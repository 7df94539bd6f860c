// v_mfma_f32_16x16x32_f16 with a float16-subnormal B (b * 2^-24) against SMALL A values 2^-k: down to which
// magnitude is the product kept?  One nonzero product per output.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float *out, int kexp, int normal_b) {
    const int lane = threadIdx.x;
    u16x8 braw = {0, 0, 0, 0, 0, 0, 0, 0};
    h16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane < 16) {
        a[0] = (_Float16)ldexpf(1.0f, -kexp);          // 2^-kexp (subnormal float16 for kexp > 14)
        if (normal_b) b[0] = (_Float16)200.0f; else braw[0] = 200;
    }
    if (!normal_b) b = __builtin_bit_cast(h16x8, braw);
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    out[lane] = acc[0];
}
int main() {
    float *d; hipMalloc(&d, 64 * 4);
    for (int nb = 0; nb < 2; ++nb)
        for (int kexp : {0, 5, 10, 14, 15, 18, 20, 22, 24}) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, kexp, nb);
            float h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            const double exact = std::ldexp(200.0, -kexp) * (nb ? 1.0 : std::ldexp(1.0, -24));
            printf("B %s, A = 2^-%-2d: got %.6e exact %.6e %s\n", nb ? "normal   " : "subnormal", kexp, h[0], exact,
                   h[0] == (float)exact ? "ok" : "DIFFERENT");
        }
    return 0;
}
