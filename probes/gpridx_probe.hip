// Probes for a scatter-FMA sparse kernel on gfx950 (round 4):
//  A  does VGPR index mode (s_set_gpr_idx_on, gfx9) apply to the destination / src2 of the packed
//     v_pk_fma_f32 (VOP3P)?  even and odd index values
//  B  issue cost: {index change + 4 packed FMAs + conversion + address add} per "bundle", 16 waves per CU
//  C  LDS-DMA (global_load_lds_dwordx4) to an LDS address that is only 4- / 8-byte aligned
//  D  ds_read_u16 by 64 lanes at a row pitch of 1024 / 1028 / 1032 / 1040 bytes (bank conflicts)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)

#define ACCS "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79"

__global__ void __launch_bounds__(64) k_a(float *out, int idx) {
    asm volatile(
        "v_mov_b32 v64, 0\n\tv_mov_b32 v65, 0\n\tv_mov_b32 v66, 0\n\tv_mov_b32 v67, 0\n\t"
        "v_mov_b32 v68, 0\n\tv_mov_b32 v69, 0\n\tv_mov_b32 v70, 0\n\tv_mov_b32 v71, 0\n\t"
        "v_mov_b32 v72, 0\n\tv_mov_b32 v73, 0\n\tv_mov_b32 v74, 0\n\tv_mov_b32 v75, 0\n\t"
        "v_mov_b32 v76, 0\n\tv_mov_b32 v77, 0\n\tv_mov_b32 v78, 0\n\tv_mov_b32 v79, 0\n\t"
        "v_cvt_f32_u32 v40, %0\n\tv_add_f32 v40, 1.0, v40\n\tv_mov_b32 v41, 0x42c80000\n\t"      // x = lane + 1; v41 = 100 (must NOT be used)
        "s_mov_b32 s40, 0x40400000\n\ts_mov_b32 s41, 0x40800000\n\t"   // 3, 4
        "s_mov_b32 s42, 0x40a00000\n\ts_mov_b32 s43, 0x40c00000\n\t"   // 5, 6
        "s_set_gpr_idx_on %1, 0xc\n\t"
        "v_pk_fma_f32 v[64:65], v[40:41], s[40:41], v[64:65] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 v[66:67], v[40:41], s[42:43], v[66:67] op_sel_hi:[0,1,1]\n\t"
        "s_set_gpr_idx_off\n\t"
        :: "v"(threadIdx.x), "s"(idx) : ACCS, "v40", "v41", "s40", "s41", "s42", "s43");
    float r[16];
    asm volatile("v_mov_b32 %0, v64\n\tv_mov_b32 %1, v65\n\tv_mov_b32 %2, v66\n\tv_mov_b32 %3, v67\n\t"
                 "v_mov_b32 %4, v68\n\tv_mov_b32 %5, v69\n\tv_mov_b32 %6, v70\n\tv_mov_b32 %7, v71"
                 : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]) :: ACCS);
    asm volatile("v_mov_b32 %0, v72\n\tv_mov_b32 %1, v73\n\tv_mov_b32 %2, v74\n\tv_mov_b32 %3, v75\n\t"
                 "v_mov_b32 %4, v76\n\tv_mov_b32 %5, v77\n\tv_mov_b32 %6, v78\n\tv_mov_b32 %7, v79"
                 : "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]), "=v"(r[12]), "=v"(r[13]), "=v"(r[14]), "=v"(r[15]) :: ACCS);
    for (int i = 0; i < 16; ++i) out[threadIdx.x * 16 + i] = r[i];
}

// B: per bundle {s_bfe, s_set_gpr_idx_on, 4 x v_pk_fma_f32, s_set_gpr_idx_off, v_cvt, v_add}; KIND 0 = with the
// index instructions, 1 = the same VALU work with static registers (no index mode), 2 = non-packed v_fmac x 8
template <int KIND>
__global__ void __launch_bounds__(1024) k_b(float *out, unsigned long long *cyc, int iters, unsigned hdr) {
    asm volatile(
        "v_mov_b32 v64, 0\n\tv_mov_b32 v65, 0\n\tv_mov_b32 v66, 0\n\tv_mov_b32 v67, 0\n\t"
        "v_mov_b32 v68, 0\n\tv_mov_b32 v69, 0\n\tv_mov_b32 v70, 0\n\tv_mov_b32 v71, 0\n\t"
        "v_mov_b32 v72, 0\n\tv_mov_b32 v73, 0\n\tv_mov_b32 v74, 0\n\tv_mov_b32 v75, 0\n\t"
        "v_mov_b32 v76, 0\n\tv_mov_b32 v77, 0\n\tv_mov_b32 v78, 0\n\tv_mov_b32 v79, 0\n\t"
        "v_mov_b32 v40, %0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, %0\n\t"
        "s_mov_b32 s40, 0x3f800000\n\ts_mov_b32 s41, 0x3f000000\n\ts_mov_b32 s42, 0x3e800000\n\ts_mov_b32 s43, 0x3e000000\n\t"
        "s_mov_b32 s44, 0x3f800000\n\ts_mov_b32 s45, 0x3f000000\n\ts_mov_b32 s46, 0x3e800000\n\ts_mov_b32 s47, 0x3e000000\n\t"
        :: "v"(threadIdx.x) : ACCS, "v40", "v41", "v42", "v43", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48");
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (KIND == 0)
                asm volatile(
                    "s_bfe_u32 s48, %0, 0x30010\n\t"          // 3 bits at 16: index 0 .. 7 (here: what the caller passes)
                    "s_lshl_b32 s48, s48, 1\n\t"
                    "v_cvt_f32_u32 v40, v42\n\t"
                    "v_add_u32 v43, s48, v42\n\t"
                    "s_set_gpr_idx_on s48, 0xc\n\t"
                    "v_pk_fma_f32 v[64:65], v[40:41], s[40:41], v[64:65] op_sel_hi:[0,1,1]\n\t"
                    "v_pk_fma_f32 v[66:67], v[40:41], s[42:43], v[66:67] op_sel_hi:[0,1,1]\n\t"
                    "v_pk_fma_f32 v[68:69], v[40:41], s[44:45], v[68:69] op_sel_hi:[0,1,1]\n\t"
                    "v_pk_fma_f32 v[70:71], v[40:41], s[46:47], v[70:71] op_sel_hi:[0,1,1]\n\t"
                    "s_set_gpr_idx_off\n\t"
                    :: "s"(hdr + u * 0x10000u) : ACCS, "v40", "v41", "v43", "s48", "scc");
            else if (KIND == 1)
                asm volatile(
                    "s_bfe_u32 s48, %0, 0x30010\n\t"
                    "s_lshl_b32 s48, s48, 1\n\t"
                    "v_cvt_f32_u32 v40, v42\n\t"
                    "v_add_u32 v43, s48, v42\n\t"
                    "v_pk_fma_f32 v[64:65], v[40:41], s[40:41], v[64:65] op_sel_hi:[0,1,1]\n\t"
                    "v_pk_fma_f32 v[66:67], v[40:41], s[42:43], v[66:67] op_sel_hi:[0,1,1]\n\t"
                    "v_pk_fma_f32 v[68:69], v[40:41], s[44:45], v[68:69] op_sel_hi:[0,1,1]\n\t"
                    "v_pk_fma_f32 v[70:71], v[40:41], s[46:47], v[70:71] op_sel_hi:[0,1,1]\n\t"
                    :: "s"(hdr + u * 0x10000u) : ACCS, "v40", "v41", "v43", "s48", "scc");
            else
                asm volatile(
                    "s_bfe_u32 s48, %0, 0x30010\n\t"
                    "s_lshl_b32 s48, s48, 1\n\t"
                    "v_cvt_f32_u32 v40, v42\n\t"
                    "v_add_u32 v43, s48, v42\n\t"
                    "s_set_gpr_idx_on s48, 0x8\n\t"
                    "v_fmac_f32 v64, s40, v40\n\tv_fmac_f32 v65, s41, v40\n\tv_fmac_f32 v66, s42, v40\n\tv_fmac_f32 v67, s43, v40\n\t"
                    "v_fmac_f32 v68, s44, v40\n\tv_fmac_f32 v69, s45, v40\n\tv_fmac_f32 v70, s46, v40\n\tv_fmac_f32 v71, s47, v40\n\t"
                    "s_set_gpr_idx_off\n\t"
                    :: "s"(hdr + u * 0x10000u) : ACCS, "v40", "v41", "v43", "s48", "scc");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s;
    asm volatile("v_add_f32 %0, v64, v66\n\tv_add_f32 %0, %0, v68\n\tv_add_f32 %0, %0, v70\n\tv_add_f32 %0, %0, v72\n\tv_add_f32 %0, %0, v65" : "=v"(s) :: ACCS);
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// C: LDS-DMA to a destination with a given byte misalignment
__global__ void __launch_bounds__(64) k_c(const unsigned *src, unsigned *out, int mis) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < 2048 / 4; i += 64) ((unsigned *)lds)[i] = 0xdeadbeefu;
    __syncthreads();
    typedef const __attribute__((address_space(1))) void *gptr;
    typedef __attribute__((address_space(3))) void *lptr;
    __builtin_amdgcn_global_load_lds((gptr)(src + threadIdx.x * 4), (lptr)(lds + mis), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048 / 4; i += 64) out[i] = ((unsigned *)lds)[i];
}

// D: 64 lanes read one u16 each, lane i at i * pitch + p * 2
__global__ void __launch_bounds__(1024) k_d(unsigned *out, unsigned long long *cyc, int pitch, int iters, int wide) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < 70000 / 4; i += 1024) ((unsigned *)lds)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = ((it * 8 + u) * 37) & 255;
            if (wide) acc += *(const unsigned *)(lds + lane * pitch + p * 4);
            else acc += *(const unsigned short *)(lds + lane * pitch + p * 2);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const char *which = argc > 1 ? argv[1] : "ABCD";
    auto want = [&](char c) { for (const char *p = which; *p; ++p) if (*p == c) return true; return false; };
    float *out; unsigned long long *cyc, h;
    printf("start %s\n", which);
    CHECK(hipMalloc(&out, 1024 * 1024 * 4)); CHECK(hipMalloc(&cyc, 8));
    // A
    if (want('A'))
    for (int idx : {0, 2, 4, 1, 3, 10}) {
        CHECK(hipMemset(out, 0, 64 * 16 * 4));
        hipLaunchKernelGGL(k_a, dim3(1), dim3(64), 0, 0, out, idx);
        CHECK(hipDeviceSynchronize());
        float r[64 * 16];
        CHECK(hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost));
        printf("A idx %2d  lane 1 (x = 2):", idx);
        for (int i = 0; i < 16; ++i) printf(" %g", r[16 + i]);
        bool ok = true;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 16; ++i) {
                const float x = l + 1.f;
                float want = 0.f;
                if (i == idx) want = 3 * x; else if (i == idx + 1) want = 4 * x;
                else if (i == idx + 2) want = 5 * x; else if (i == idx + 3) want = 6 * x;
                if (r[l * 16 + i] != want) ok = false;
            }
        printf("   -> %s\n", ok ? "indexed dst/src2 as expected" : "NOT as expected");
    }
    // B
    const int iters = 20000;
    const char *names[3] = {"index mode + 4 pk_fma", "static 4 pk_fma", "index mode + 8 v_fmac"};
    if (want('B'))
    for (int kind = 0; kind < 3; ++kind)
        for (int blocks : {1, 256}) {
            hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            for (int r = 0; r < 2; ++r) {
                CHECK(hipEventRecord(a));
                if (kind == 0) hipLaunchKernelGGL(k_b<0>, dim3(blocks), dim3(1024), 0, 0, out, cyc, iters, 0x10000u);
                if (kind == 1) hipLaunchKernelGGL(k_b<1>, dim3(blocks), dim3(1024), 0, 0, out, cyc, iters, 0x10000u);
                if (kind == 2) hipLaunchKernelGGL(k_b<2>, dim3(blocks), dim3(1024), 0, 0, out, cyc, iters, 0x10000u);
                CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            }
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            // 16 waves per CU = 4 per SIMD; a bundle = 6 VALU (kind 2: 10)
            printf("B %-24s blocks %3d: %.3f ms, %.2f ns per bundle per SIMD (4 waves), %.1f ticks per bundle and wave\n",
                   names[kind], blocks, ms, ms * 1e6 / (iters * 4.0 * 4.0), (double)h / (iters * 4.0));
        }
    // C
    if (want('C')) {
        unsigned *src, *o;
        CHECK(hipMalloc(&src, 4096)); CHECK(hipMalloc(&o, 4096));
        std::vector<unsigned> hs(1024);
        for (int i = 0; i < 1024; ++i) hs[i] = 0x1000 + i;
        CHECK(hipMemcpy(src, hs.data(), 4096, hipMemcpyHostToDevice));
        for (int mis : {0, 8, 4, 12}) {
            hipLaunchKernelGGL(k_c, dim3(1), dim3(64), 4096, 0, src, o, mis);
            CHECK(hipDeviceSynchronize());
            std::vector<unsigned> ho(512);
            CHECK(hipMemcpy(ho.data(), o, 2048, hipMemcpyDeviceToHost));
            int good = 0, bad = 0;
            for (int i = 0; i < 256; ++i) { if (ho[mis / 4 + i] == 0x1000u + i) ++good; else ++bad; }
            printf("C LDS-DMA dwordx4 to LDS offset %2d: %d of 256 dwords in place (%s)\n", mis, good, bad ? "BROKEN" : "ok");
        }
    }
    // D
    if (want('D'))
    for (int wide : {0, 1})
        for (int pitch : {1024, 1028, 1032, 1040, 1056}) {
            unsigned *o = (unsigned *)out;
            CHECK(hipFuncSetAttribute((const void *)k_d, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
            hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            float ms = 0;
            for (int r = 0; r < 2; ++r) {
                CHECK(hipEventRecord(a));
                hipLaunchKernelGGL(k_d, dim3(256), dim3(1024), 72 * 1024, 0, o, cyc, pitch, 20000, wide);
                CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
                CHECK(hipEventElapsedTime(&ms, a, b));
            }
            CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            printf("D ds_read_%s pitch %4d: %.3f ms, %.2f ns per read instruction per CU (16 waves)\n", wide ? "b32" : "u16", pitch,
                   ms, ms * 1e6 / (20000 * 8.0 * 16.0));
        }
    return 0;
}
