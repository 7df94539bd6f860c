// ltmi_cryst.hip -- CrystallinityUDF.process_frame for 256 x 256 frames in ONE kernel (SURVEY.md
// section 8, row f3; reference udf/crystallinity.py:73-79):
//
//     intensity[f] = sum( abs(rfft2(frame * real_mask)) * half_fourier_mask )
//
// The hipFFT route (ltmi_fft.hip) writes every frame as float32, transforms it into a 129 x 256
// half spectrum in HBM and reads that back: 128 KiB of pixels become 0.9 MiB of traffic and three
// launches.  Here a workgroup keeps ONE frame in the LDS from the pixels to the single float:
//
//   rows     a wave takes two rows a, b as the complex sequence a + i b through a 256-point Stockham
//            radix-4 transform (4 passes, lane t = butterfly t, data through 2 KiB of LDS per wave),
//            separates the two spectra (Z[k], conj Z[256 - k]: one ds_bpermute pair) and stores the
//            columns kx < K it will need -- K = the ring's outer radius + 1 -- into G[kx][y];
//   columns  a wave transforms column kx of G in place with the same passes and sums
//            |F[ky][kx]| * mask[ky][kx] into a register; the workgroup writes one float.
//
// LDS: K columns of (256 + 2) float2 + 8 x 2 KiB of row scratch = 147 KiB for K = 65 (rad_out 64);
// rings with K > CF_KMAX columns, other frame shapes, float64 pixels and fused corrections stay on
// the hipFFT route.  HBM traffic: the pixels once (+ the two masks from the L2).
//
// Bank conflicts: a pass reads the units t + 64 r (contiguous lanes: none) and writes 4 t + r,
// 16 (t >> 2) + (t & 3) + 4 r, 64 (t >> 4) + (t & 15) + 16 r: the first two put a 16-lane store
// group on 4 of its 16 bank pairs.  Unit u is therefore kept at u ^ (5 * ((u >> 4) & 3)): bits
// 5:4 XORed into bits 1:0 and 3:2 make every store group of all three passes a permutation of the
// 16 bank pairs and leave the reads contiguous within 16 lanes (model: scripts/cryst_fft_model.py).
#include "ltmi_common.h"
#include <algorithm>

namespace ltmi {

constexpr int CF_N = 256;                       // frame edge
constexpr int CF_WAVES = 8;
constexpr int CF_COL = CF_N + 2;                // float2 units per column of G: 2064 B, 16-B aligned, a
                                                // b128 store of 8 neighbouring columns hits 32 banks
constexpr int CF_SCR = CF_N;                    // float2 units of row scratch per wave
constexpr int CF_KMAX = (160 * 1024 - 256 - CF_WAVES * CF_SCR * 8) / (CF_COL * 8);

struct CfLane {                                 // per-lane constants of the four passes
    float2 tw[3][3];                            // passes p = 4, 16, 64: exp(-2 pi i (t & (p-1)) r / (4 p)), r = 1..3
    int rd;                                     // phys(t): reads of the passes 2..4 (+ 64 r)
    int wr[3][4];                               // stores of the passes 1..3
};

__device__ __forceinline__ int cf_phys(int u) { return u ^ (5 * ((u >> 4) & 3)); }

__device__ __forceinline__ float2 cf_mul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

__device__ __forceinline__ void cf_bfly(float2 (&u)[4]) {
    const float2 t0 = make_float2(u[0].x + u[2].x, u[0].y + u[2].y);
    const float2 t1 = make_float2(u[0].x - u[2].x, u[0].y - u[2].y);
    const float2 t2 = make_float2(u[1].x + u[3].x, u[1].y + u[3].y);
    const float2 t3 = make_float2(u[1].y - u[3].y, u[3].x - u[1].x);       // -i (u1 - u3)
    u[0] = make_float2(t0.x + t2.x, t0.y + t2.y);
    u[1] = make_float2(t1.x + t3.x, t1.y + t3.y);
    u[2] = make_float2(t0.x - t2.x, t0.y - t2.y);
    u[3] = make_float2(t1.x - t3.x, t1.y - t3.y);
}

// the wave's own stores become visible to its own loads in program order (DS operations of a wave
// are served in order): only the compiler has to be kept from moving them across each other
__device__ __forceinline__ void cf_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// 256-point forward transform of buf (float2 units; the input at rd0 + 64 r for lane t); the result
// stays in registers: u[r] = Z[t + 64 r].  buf is destroyed.
__device__ __forceinline__ void cf_fft256(float2 *buf, int rd0, const CfLane &c, float2 (&u)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) u[r] = buf[rd0 + 64 * r];
    cf_bfly(u);
    cf_wave_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) buf[c.wr[0][r]] = u[r];
#pragma unroll
    for (int pi = 0; pi < 3; ++pi) {
        cf_wave_sync();
#pragma unroll
        for (int r = 0; r < 4; ++r) u[r] = buf[c.rd + 64 * r];
        u[1] = cf_mul(u[1], c.tw[pi][0]);
        u[2] = cf_mul(u[2], c.tw[pi][1]);
        u[3] = cf_mul(u[3], c.tw[pi][2]);
        cf_bfly(u);
        if (pi < 2) {
            cf_wave_sync();
#pragma unroll
            for (int r = 0; r < 4; ++r) buf[c.wr[pi + 1][r]] = u[r];
        }
    }
    cf_wave_sync();
}

template <typename T, bool MASK>
__global__ void __launch_bounds__(CF_WAVES * 64)
k_cryst_fused(const T *__restrict__ tile, int64_t ld, int64_t n_frames,
              const float *__restrict__ real_mask, const float *__restrict__ mask_t, int K,
              float *__restrict__ out, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cf_smem[];
    __shared__ float part[CF_WAVES];
    const int t = threadIdx.x & 63, w = threadIdx.x >> 6;
    float2 *G = (float2 *)cf_smem;
    float2 *scr = G + K * CF_COL + w * CF_SCR;

    CfLane c;
#pragma unroll
    for (int pi = 0; pi < 3; ++pi) {
        const int p = 4 << (2 * pi);
        const int k = t & (p - 1);
#pragma unroll
        for (int r = 1; r < 4; ++r) {
            double s, co;
            sincospi(-2.0 * (double)(k * r) / (double)(4 * p), &s, &co);
            c.tw[pi][r - 1] = make_float2((float)co, (float)s);
        }
    }
    c.rd = cf_phys(t);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        c.wr[0][r] = cf_phys(4 * t + r);
        c.wr[1][r] = cf_phys(16 * (t >> 2) + (t & 3) + 4 * r);
        c.wr[2][r] = cf_phys(64 * (t >> 4) + (t & 15) + 16 * r);
    }
    // the staged row pair: element idx at idx ^ (2 * ((idx >> 4) & 1)) (16-byte halves of a lane's 32 bytes
    // swapped in every other group of 4 lanes: the b128 stores of 8 lanes then cover all 32 banks)
    const int half = (t >> 2) & 1;
    const int st0 = 4 * t + 2 * half, st1 = 4 * t + 2 - 2 * half;
    const int rd_staged = t ^ (2 * ((t >> 4) & 1));
    const int back = ((64 - t) & 63) * 4;           // ds_bpermute address: lane (-t) mod 64

    typedef T __attribute__((ext_vector_type(4))) vec_t;
    for (int64_t f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const T *src = tile + f * ld;
        // ---- rows: pairs (2 y', 2 y' + 1), y' = w + 8 i
#pragma unroll 2
        for (int i = 0; i < CF_N / 2 / CF_WAVES; ++i) {
            const int yp = w + CF_WAVES * i;
            const vec_t ra = __builtin_nontemporal_load((const vec_t *)(src + (2 * yp) * CF_N + 4 * t));
            const vec_t rb = __builtin_nontemporal_load((const vec_t *)(src + (2 * yp + 1) * CF_N + 4 * t));
            float za[4], zb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                za[j] = (float)ra[j];
                zb[j] = (float)rb[j];
            }
            if (MASK) {
                const float4 ma = *(const float4 *)(real_mask + (2 * yp) * CF_N + 4 * t);
                const float4 mb = *(const float4 *)(real_mask + (2 * yp + 1) * CF_N + 4 * t);
                za[0] *= ma.x; za[1] *= ma.y; za[2] *= ma.z; za[3] *= ma.w;
                zb[0] *= mb.x; zb[1] *= mb.y; zb[2] *= mb.z; zb[3] *= mb.w;
            }
            *(float4 *)(scr + st0) = make_float4(za[0], zb[0], za[1], zb[1]);
            *(float4 *)(scr + st1) = make_float4(za[2], zb[2], za[3], zb[3]);
            cf_wave_sync();
            float2 u[4];
            cf_fft256(scr, rd_staged, c, u);
            // two real rows out of one complex transform: with Z[k] = (a, b), Z[256 - k] = (c, d)
            //   2 A[k] = (a + c, b - d)        2 B[k] = (b + d, c - a)       (the 1/2 is applied at the end)
            {
                const float2 give = t == 0 ? u[0] : u[3];
                const float cr = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(back, __builtin_bit_cast(int, give.x)));
                const float ci = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(back, __builtin_bit_cast(int, give.y)));
                if (t < K)
                    *(float4 *)(G + t * CF_COL + 2 * yp) =
                        make_float4(u[0].x + cr, u[0].y - ci, u[0].y + ci, cr - u[0].x);
            }
            if (K > 64) {
                const float2 give = t == 0 ? u[3] : u[2];
                const float cr = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(back, __builtin_bit_cast(int, give.x)));
                const float ci = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(back, __builtin_bit_cast(int, give.y)));
                if (t + 64 < K)
                    *(float4 *)(G + (t + 64) * CF_COL + 2 * yp) =
                        make_float4(u[1].x + cr, u[1].y - ci, u[1].y + ci, cr - u[1].x);
            }
        }
        __syncthreads();
        // ---- columns kx = w + 8 i: transform in place, |F| * mask summed per lane
        float acc = 0.f;
        for (int kx = w; kx < K; kx += CF_WAVES) {
            float m[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) m[r] = mask_t[kx * CF_N + t + 64 * r];
            float2 u[4];
            cf_fft256(G + kx * CF_COL, t, c, u);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (m[r] != 0.f) acc += sqrtf(u[r].x * u[r].x + u[r].y * u[r].y) * m[r];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (t == 0) part[w] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < CF_WAVES; ++i) v += part[i];
            v *= 0.5f;
            out[f] = accumulate ? out[f] + v : v;
        }
    }
}

// mask_t[kx][ky] = half_mask[ky][kx] for the K columns of the ring (lanes = ky in the column stage)
__global__ void __launch_bounds__(256)
k_cryst_mask_t(const float *__restrict__ half_mask, int wc, int K, float *__restrict__ mask_t) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= K * CF_N) return;
    const int kx = i / CF_N, ky = i - kx * CF_N;
    mask_t[i] = half_mask[(int64_t)ky * wc + kx];
}

int cryst_fused_max_cols() { return CF_KMAX; }

template <typename T>
static int launch_fused(const void *tile, int64_t ld, int64_t n_frames, const float *real_mask,
                        const float *mask_t, int K, float *out, int accumulate, int n_cu,
                        hipStream_t stream) {
    auto kern = real_mask ? k_cryst_fused<T, true> : k_cryst_fused<T, false>;
    const int lds = K * CF_COL * 8 + CF_WAVES * CF_SCR * 8;
    int device = 0;
    LTMI_HIP(hipGetDevice(&device));
    static bool attr_set[16][2] = {{false}};          // per device (and per pixel type: one copy per T)
    if (!attr_set[device & 15][real_mask ? 1 : 0]) {
        LTMI_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024 - 256));
        attr_set[device & 15][real_mask ? 1 : 0] = true;
    }
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(n_frames, n_cu));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(CF_WAVES * 64), (size_t)lds, stream, (const T *)tile, ld,
                       n_frames, real_mask, mask_t, K, out, accumulate);
    LTMI_HIP(hipGetLastError());
    return LTMI_OK;
}

// -> LTMI_OK with *handled = true when the fused kernel ran
int cryst_fused(const void *tile, int tile_dtype, int64_t n_frames, int64_t ld, int sig_h, int sig_w,
                const float *real_mask, const float *half_mask, int n_cols, float *mask_t, float *out,
                int accumulate, int n_cu, hipStream_t stream, bool *handled) {
    *handled = false;
    if (sig_h != CF_N || sig_w != CF_N || n_cols < 1 || n_cols > CF_KMAX || !mask_t) return LTMI_OK;
    const size_t esz = (size_t)dtype_size(tile_dtype);
    if (esz > 4 || tile_dtype == LTMI_F64) return LTMI_OK;
    if ((uintptr_t)tile % (4 * esz) != 0 || ld % 4 != 0) return LTMI_OK;
    if (real_mask && (uintptr_t)real_mask % 16 != 0) return LTMI_OK;
    hipLaunchKernelGGL(k_cryst_mask_t, dim3((unsigned)((n_cols * CF_N + 255) / 256)), dim3(256), 0, stream,
                       half_mask, sig_w / 2 + 1, n_cols, mask_t);
    int rc = LTMI_E_DTYPE;
    switch (tile_dtype) {
        case LTMI_BOOL:
        case LTMI_U8: rc = launch_fused<uint8_t>(tile, ld, n_frames, real_mask, mask_t, n_cols, out, accumulate, n_cu, stream); break;
        case LTMI_I8: rc = launch_fused<int8_t>(tile, ld, n_frames, real_mask, mask_t, n_cols, out, accumulate, n_cu, stream); break;
        case LTMI_U16: rc = launch_fused<uint16_t>(tile, ld, n_frames, real_mask, mask_t, n_cols, out, accumulate, n_cu, stream); break;
        case LTMI_I16: rc = launch_fused<int16_t>(tile, ld, n_frames, real_mask, mask_t, n_cols, out, accumulate, n_cu, stream); break;
        case LTMI_U32: rc = launch_fused<uint32_t>(tile, ld, n_frames, real_mask, mask_t, n_cols, out, accumulate, n_cu, stream); break;
        case LTMI_I32: rc = launch_fused<int32_t>(tile, ld, n_frames, real_mask, mask_t, n_cols, out, accumulate, n_cu, stream); break;
        case LTMI_F32: rc = launch_fused<float>(tile, ld, n_frames, real_mask, mask_t, n_cols, out, accumulate, n_cu, stream); break;
        default: return LTMI_OK;
    }
    if (rc == LTMI_OK) *handled = true;
    return rc;
}

}  // namespace ltmi
