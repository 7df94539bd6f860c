// Accumulation error of v_mfma_f32_16x16x32_f16 with the B operand (bytes 0..255) given (a) as float16
// subnormals b * 2^-24 and (b) as normal float16 numbers b, over chains of N MFMAs into one accumulator,
// against the exact sum (float64 on the host).  A = weights in (0, 1] scaled by 64 (as the sparse kernel does).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
__device__ __host__ inline unsigned rnd(unsigned x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }
template <bool SUB>
__global__ void k(float *out, const _Float16 *wts, int n) {
    const int lane = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    unsigned st = 12345u + lane * 977u;
    for (int it = 0; it < n; ++it) {
        h16x8 a, b;
        u16x8 braw;
        for (int i = 0; i < 8; ++i) {
            st = rnd(st);
            const bool hi = (i & 2) != 0;                                 // slots [lo lo hi hi | lo lo hi hi]
            const unsigned byte = hi ? (st & 0xfu) : (st & 0xffu);       // 12-bit pixels: hi < 16
            a[i] = hi ? wts[(it * 64 + lane) * 8 + i - 2] * (_Float16)256.0f : wts[(it * 64 + lane) * 8 + i];
            if (SUB) braw[i] = (unsigned short)byte; else b[i] = (_Float16)(float)byte;
        }
        if (SUB) b = __builtin_bit_cast(h16x8, braw);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = SUB ? acc[r] * 16777216.0f : acc[r];
}
int main() {
    const int n = 54;                      // 54 MFMAs x 32 = 1728 products per output (a C4 ring: ~430 x 4)
    std::vector<_Float16> w((size_t)n * 64 * 8);
    unsigned s = 777;
    for (auto &x : w) { s = rnd(s); x = (_Float16)(64.0f * ((s & 0xffff) + 1) / 65536.0f); }
    _Float16 *dw; float *d;
    hipMalloc(&dw, w.size() * 2); hipMalloc(&d, 256 * 4);
    hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice);
    for (int sub = 0; sub < 2; ++sub) {
        if (sub) hipLaunchKernelGGL(k<true>, dim3(1), dim3(64), 0, 0, d, dw, n);
        else hipLaunchKernelGGL(k<false>, dim3(1), dim3(64), 0, 0, d, dw, n);
        float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        double maxrel = 0, mean = 0;
        for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 4; ++r) {
                const int j = lane & 15, i = 4 * (lane >> 4) + r;
                double ref = 0;
                for (int kg = 0; kg < 4; ++kg) {
                    const int la = i + 16 * kg, lb = j + 16 * kg;
                    unsigned st = 12345u + lb * 977u;
                    for (int it = 0; it < n; ++it)
                        for (int e = 0; e < 8; ++e) {
                            st = rnd(st);
                            const bool hi = (e & 2) != 0;
                            const double av = hi ? 256.0 * (double)w[((size_t)it * 64 + la) * 8 + e - 2]
                                                 : (double)w[((size_t)it * 64 + la) * 8 + e];
                            ref += av * (double)(hi ? (st & 0xfu) : (st & 0xffu));
                        }
                }
                const double rel = (h[lane * 4 + r] - ref) / ref;
                maxrel = std::fmax(maxrel, std::fabs(rel));
                mean += rel / 256;
            }
        printf("%s B: max rel err %.3e, mean rel err %+.3e\n", sub ? "subnormal" : "normal   ", maxrel, mean);
    }
    return 0;
}
