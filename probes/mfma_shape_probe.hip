// v_mfma_f32_16x16x4_f32 against v_mfma_f32_32x32x2_f32 ("the 32x32x2 kernel body for C5"): issue rate, and -- sampled
// by the caller from hwmon while each phase runs for ~1.5 s -- board power and clock.  Both shapes do 32 frames x 32 columns x 32
// pixels per step: 32 instructions of 32 pipe cycles (2 tiles x 2 groups x 8) or 16 of 64, from the SAME number of fragment registers
// (A: 16 floats per lane, B: 16) -- per flop the two shapes read the same LDS bytes once a 16x16 kernel re-uses its fragments for two
// frame tiles and two column groups, as k_dense_lds / k_dense_fold do.  LDS = 1: the fragments are re-read from LDS every step
// (8 ds_read_b128 per 1024 pipe cycles, the kernels' rate).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)

template <int SHAPE, int LDS>
__global__ void __launch_bounds__(256) k(float *out, int iters) {
    __shared__ f32x4 frag[256 * 8];
    for (int i = 0; i < 8; ++i) frag[threadIdx.x * 8 + i] = f32x4{1.f + threadIdx.x, 2.f, 3.f, 4.f + i};
    __syncthreads();
    f32x4 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = frag[threadIdx.x * 8 + i]; b[i] = frag[threadIdx.x * 8 + 4 + i]; }
    f32x4 acc4[4] = {};
    f32x16 acc16 = {};
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = frag[((threadIdx.x + it) & 255) * 8 + i];
                b[i] = frag[((threadIdx.x + it) & 255) * 8 + 4 + i];
            }
        }
        if (SHAPE == 16) {
            // tiles (a[0..1] / a[2..3]) x groups (b[0..1] / b[2..3]) x 8 pixels of the lane
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                    for (int g = 0; g < 2; ++g)
                        acc4[tl * 2 + g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tl * 2 + (j >> 2)][j & 3], b[g * 2 + (j >> 2)][j & 3],
                                                                               acc4[tl * 2 + g], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                acc16 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j >> 2][j & 3], b[j >> 2][j & 3], acc16, 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc4[i][0] + acc4[i][3];
    for (int i = 0; i < 16; ++i) s += acc16[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static double now() { return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count(); }

template <int SHAPE, int LDS> int run(const char *name, float *out) {
    const int iters = 20000, blocks = 256;                    // one wave per SIMD, like the kernels
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<SHAPE, LDS>), dim3(blocks), dim3(256), 0, 0, out, iters);
    CHECK(hipDeviceSynchronize());
    const double t0 = now();
    float ms_sum = 0; int n = 0;
    while (now() - t0 < 1.5) {
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 4; ++r) hipLaunchKernelGGL((k<SHAPE, LDS>), dim3(blocks), dim3(256), 0, 0, out, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms_sum += ms; n += 4;
    }
    const double t1 = now();
    const double flop = (double)blocks * 4 * iters * 2.0 * 32 * 32 * 32 * n;
    printf("PHASE %.3f %.3f %-44s %.1f TFLOP/s (%.1f %% of 157.3)\n", t0, t1, name, flop / (ms_sum * 1e-3) / 1e12,
           flop / (ms_sum * 1e-3) / 1e12 / 157.3 * 100);
    fflush(stdout);
    return 0;
}

int main() {
    float *out; CHECK(hipMalloc(&out, 256 * 256 * 4));
    run<16, 0>("16x16x4, operands in registers", out);
    run<32, 0>("32x32x2, operands in registers", out);
    run<16, 1>("16x16x4, fragments re-read from LDS", out);
    run<32, 1>("32x32x2, fragments re-read from LDS", out);
    run<16, 0>("16x16x4, operands in registers (again)", out);
    return 0;
}
