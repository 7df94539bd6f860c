// Issue rate of the float16 matrix instructions on gfx950: v_mfma_f32_16x16x16_f16 (K = 16) against
// v_mfma_f32_16x16x32_f16 (K = 32) and, for scale, v_mfma_f32_16x16x4_f32.  One wave per SIMD, 8 independent
// accumulators, cycles per instruction from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)

template <int KIND>
__global__ void __launch_bounds__(256) k(float *out, unsigned long long *cyc, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    const float x = (float)threadIdx.x;
    h16x4 a4 = {(_Float16)x, (_Float16)1.f, (_Float16)2.f, (_Float16)3.f}, b4 = a4;
    h16x8 a8 = {(_Float16)x, 1, 2, 3, 4, 5, 6, 7}, b8 = a8;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x + 1.f, acc[j], 0, 0, 0);
            if (KIND == 1) acc[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[j], 0, 0, 0);
            if (KIND == 2) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[j], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    float *out; unsigned long long *cyc, h;
    CHECK(hipMalloc(&out, 1024 * 256 * 4)); CHECK(hipMalloc(&cyc, 8));
    const int iters = 20000;
    const char *names[3] = {"v_mfma_f32_16x16x4_f32", "v_mfma_f32_16x16x16_f16", "v_mfma_f32_16x16x32_f16"};
    for (int kind = 0; kind < 3; ++kind)
        for (int blocks : {1, 256}) {
            hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            for (int r = 0; r < 2; ++r) {
                CHECK(hipEventRecord(a));
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
                if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
                if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
                CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            }
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            printf("%-26s blocks %3d: %.2f counter ticks per MFMA (one wave per SIMD), %.3f ms, %.2f ns per MFMA per SIMD\n",
                   names[kind], blocks, (double)h / (iters * 8.0), ms, ms * 1e6 / (iters * 8.0));
        }
    return 0;
}
