// LDS atomic-add throughput probe: how many ds_add_f32 (no return) per cycle does a CU sustain,
// conflict-free (lane-contiguous) -- input for the "accumulators in LDS" sparse design.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(512) k_probe(float *out, int iters, int rows) {
    extern __shared__ float acc[];                       // rows x 64 floats
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < rows * 64; i += blockDim.x) acc[i] = 0.f;
    __syncthreads();
    float v = 1.0f + lane;
    int r = wave * 7;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            r = (r + 13) & (rows - 1);                       // a different accumulator row each time
            float *p = acc + r * 64 + lane;
            if (MODE == 0) {
                __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (MODE == 1) {
                *p += v;                                  // plain read-modify-write
            } else {
                v += *p;                                  // read only
            }
        }
    }
    __syncthreads();
    if (MODE == 2) acc[threadIdx.x] = v;
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[threadIdx.x];
}

int main() {
    const int rows = 512, iters = 4096, blocks = 256 * 1, threads = 512;
    float *out;
    hipMalloc(&out, blocks * threads * sizeof(float));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const char *names[3] = {"ds_add_f32 (atomic, no return)", "read-modify-write", "read only"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(blocks), dim3(threads), rows * 256, 0, out, iters, rows);
            if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(blocks), dim3(threads), rows * 256, 0, out, iters, rows);
            if (mode == 2) hipLaunchKernelGGL(k_probe<2>, dim3(blocks), dim3(threads), rows * 256, 0, out, iters, rows);
            hipEventRecord(b);
            hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        const double ops = (double)blocks * (threads / 64) * iters * 8;       // wave-level ops
        printf("%-34s %.3f ms  %.2f G wave-ops/s  = %.1f cycles per wave-op per CU at 2.4 GHz (1 block/CU)\n",
               names[mode], ms, ops / ms / 1e6, ms * 1e-3 * 2.4e9 / (ops / blocks));
    }
    return 0;
}
