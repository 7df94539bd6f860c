// HBM read-streaming probe: what does the access pattern of k_dense_mfma cost by itself?
//   v0: every wave instruction reads 1 KiB contiguous (lane i: 16 B at i*16)       plain loads
//   v1: same, non-temporal
//   v2: fragment pattern of the MFMA kernel: 16 rows (frames, 128 KiB apart) x 64 B per
//       instruction, 8 consecutive instructions walk 512 B of each row               plain loads
//   v3: same, non-temporal
// Every variant reads the whole 8 GiB buffer once, 8 x 16-B loads in flight per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4 *p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

// linear: block b handles a contiguous slab; each wave walks 8 KiB per step
template <bool NT>
__global__ void __launch_bounds__(256) k_linear(const u32x4 *buf, size_t n_vec, unsigned *out) {
    const size_t per_block = n_vec / gridDim.x;
    const u32x4 *base = buf + (size_t)blockIdx.x * per_block;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned acc = 0;
    // wave w walks its quarter of the slab in steps of 8 x 64 vec
    const size_t per_wave = per_block / 4;
    const u32x4 *wb = base + (size_t)wave * per_wave;
    for (size_t i = 0; i + 512 <= per_wave; i += 512) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ld<NT>(wb + i + u * 64 + lane);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    }
    if (acc == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// fragment: wave owns 16 rows of row_vec vectors; lane (m = lane&15, kg = lane>>4)
template <bool NT>
__global__ void __launch_bounds__(256) k_fragment(const u32x4 *buf, size_t row_vec, int rows_per_wave_block,
                                                   unsigned *out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = lane & 15, kg = lane >> 4;
    const size_t row0 = ((size_t)blockIdx.x * 4 + wave) * 32;       // MT=2: 32 rows per wave
    unsigned acc = 0;
    const u32x4 *r0 = buf + (row0 + m) * row_vec + kg;
    const u32x4 *r1 = buf + (row0 + 16 + m) * row_vec + kg;
    for (size_t c = 0; c < row_vec; c += 32) {          // 32 vec = 512 B = 256 px of a row
        u32x4 v[16];
#pragma unroll
        for (int b = 0; b < 8; ++b) { v[2 * b] = ld<NT>(r0 + c + b * 4); v[2 * b + 1] = ld<NT>(r1 + c + b * 4); }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    }
    if (acc == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const size_t rows = 65536, row_bytes = 131072;
    const size_t bytes = rows * row_bytes;
    u32x4 *buf; unsigned *out;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&out, 1 << 22));
    CHECK(hipMemset(buf, 1, bytes));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const size_t n_vec = bytes / 16, row_vec = row_bytes / 16;
    auto run = [&](const char *name, auto launch) {
        std::vector<float> ts;
        for (int i = 0; i < 12; ++i) {
            CHECK(hipEventRecord(a)); launch(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (i >= 2) ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        printf("%-44s median %.3f ms  %.0f GB/s\n", name, ts[ts.size() / 2], bytes / ts[ts.size() / 2] / 1e6);
    };
    for (int blocks : {512, 1024, 2048, 4096}) {
        char nm[96];
        snprintf(nm, 96, "v0 linear plain      grid=%d", blocks);
        run(nm, [&] { hipLaunchKernelGGL(k_linear<false>, dim3(blocks), dim3(256), 0, 0, buf, n_vec, out); });
        snprintf(nm, 96, "v1 linear nt         grid=%d", blocks);
        run(nm, [&] { hipLaunchKernelGGL(k_linear<true>, dim3(blocks), dim3(256), 0, 0, buf, n_vec, out); });
    }
    run("v2 fragment 16x64B plain  grid=512", [&] { hipLaunchKernelGGL(k_fragment<false>, dim3(512), dim3(256), 0, 0, buf, row_vec, 32, out); });
    run("v3 fragment 16x64B nt     grid=512", [&] { hipLaunchKernelGGL(k_fragment<true>, dim3(512), dim3(256), 0, 0, buf, row_vec, 32, out); });
    return 0;
}
