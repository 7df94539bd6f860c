// Peak issue rate of v_mfma_f32_16x16x4_f32 on this box: pure MFMA loop vs MFMA + cvt operand prep.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)

template <int NACC, bool CVT>
__global__ void __launch_bounds__(256) k(float *out, const unsigned *in, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    unsigned raw = in[threadIdx.x];
    float a = (float)threadIdx.x, b = 1.0f + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            float av = a;
            if (CVT) { av = (float)((raw >> (j & 15)) & 0xffffu); }
            acc[j % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc[j % NACC], 0, 0, 0);
        }
        raw += it;
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool CVT> int run(const char *name, int blocks, float *out, unsigned *in) {
    const int iters = 4000;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    std::vector<float> ts;
    for (int r = 0; r < 6; ++r) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((k<NACC, CVT>), dim3(blocks), dim3(256), 0, 0, out, in, iters);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (r) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double mfma = (double)blocks * 4 * iters * 32;            // wave-level MFMA instructions
    const double tf = mfma * 2048 / (ts[2] * 1e-3) / 1e12;
    // cycles per MFMA per SIMD at 2.4 GHz nominal: waves per SIMD = blocks*4/1024
    printf("%-40s blocks=%d  %.3f ms  %.1f TFLOP/s  (%.1f%% of 157.3)\n", name, blocks, ts[2], tf, tf / 157.3 * 100);
    return 0;
}

int main() {
    float *out; unsigned *in;
    CHECK(hipMalloc(&out, 4096 * 256 * 4)); CHECK(hipMalloc(&in, 1024)); CHECK(hipMemset(in, 1, 1024));
    run<4, false>("pure mfma, 4 acc, 1 wave/SIMD", 256, out, in);
    run<4, false>("pure mfma, 4 acc, 2 waves/SIMD", 512, out, in);
    run<2, false>("pure mfma, 2 acc, 2 waves/SIMD", 512, out, in);
    run<2, true>("mfma + cvt, 2 acc, 2 waves/SIMD", 512, out, in);
    run<4, true>("mfma + cvt, 4 acc, 2 waves/SIMD", 512, out, in);
    run<1, false>("pure mfma, 1 acc, 2 waves/SIMD", 512, out, in);
    return 0;
}
