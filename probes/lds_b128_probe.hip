// ds_read_b128 by 64 lanes (n = lane & 15, kg = lane >> 4) for the mask-fragment address patterns of k_dense_lds:
// which swizzles are served without bank conflicts?  (round 4: SQ_LDS_BANK_CONFLICT scales with the number of
// column groups of the 128-pixel slot layouts, 4 extra cycles per fragment read; the 256-pixel layout has none)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int P>
__global__ void __launch_bounds__(256) k(float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < 32768 / 4; i += 256) ((float *)lds)[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, n = lane & 15, kg = lane >> 4;
    f32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            int addr;
            if (P == 0) addr = n * 1024 + (((kg * 16 + x) ^ n) << 4);                    // 256-pixel slot (NG = 1)
            else if (P == 1) addr = n * 512 + (((kg * 8 + x) ^ n) << 4);                 // 128-pixel slot, as shipped
            else if (P == 2) addr = n * 512 + ((kg * 8 + (x ^ (n & 7))) << 4);           // XOR inside the kg segment
            else if (P == 3) addr = n * 528 + ((kg * 8 + x) << 4);                       // padded rows, no XOR
            else if (P == 4) addr = n * 512 + ((kg * 8 + (x ^ (n & 7) ^ (n >> 3))) << 4);
            else addr = n * 512 + ((((kg ^ (n >> 3)) * 8) + (x ^ (n & 7))) << 4);        // P == 5
            acc += *(const f32x4 *)(lds + (addr & 32767));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float *out;
    CHECK(hipMalloc(&out, 256 * 256 * 4));
    const int iters = 20000;
    const char *names[6] = {"256-px slot, unit ^ n", "128-px slot, unit ^ n (shipped)", "128-px, x ^ (n & 7)",
                            "128-px, rows padded by 16 B", "128-px, x ^ (n&7) ^ (n>>3)", "128-px, (kg ^ (n>>3)), x ^ (n&7)"};
    for (int p = 0; p < 6; ++p) {
        hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        float ms = 0;
        for (int r = 0; r < 2; ++r) {
            CHECK(hipEventRecord(a));
            if (p == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 32768, 0, out, iters);
            if (p == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 32768, 0, out, iters);
            if (p == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 32768, 0, out, iters);
            if (p == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 32768, 0, out, iters);
            if (p == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 32768, 0, out, iters);
            if (p == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(256), 32768, 0, out, iters);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            CHECK(hipEventElapsedTime(&ms, a, b));
        }
        printf("%-36s %.3f ms  %.2f ns per ds_read_b128 per CU (4 waves)\n", names[p], ms, ms * 1e6 / (iters * 8.0 * 4.0));
    }
    return 0;
}
