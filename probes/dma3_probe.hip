// Where does global_load_lds_dwordx3 put the 12 bytes of lane l?  (gfx950)
//   hipcc --offload-arch=gfx950 probes/dma3_probe.hip -o /tmp/dma3_probe && /tmp/dma3_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void *lp;
typedef const __attribute__((address_space(1))) void *gp;
__global__ void k(const unsigned *src, unsigned *dst) {
    extern __shared__ unsigned char lds[];
    unsigned *l = (unsigned *)lds;
    for (int i = threadIdx.x; i < 512; i += 64) l[i] = 0xdeadbeefu;
    __syncthreads();
    __builtin_amdgcn_global_load_lds((gp)(src + threadIdx.x * 3), (lp)(lds), 12, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = threadIdx.x; i < 512; i += 64) dst[i] = l[i];
}
int main() {
    unsigned h[192], *d, *o, r[512];
    for (int i = 0; i < 192; ++i) h[i] = ((i / 3) << 8) | (i % 3);       // lane << 8 | word
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, o);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    for (int i = 0; i < 200; ++i) printf("%s%04x", i % 16 ? " " : "\n", r[i] == 0xdeadbeefu ? 0xffff : r[i]);
    printf("\n");
    return 0;
}
