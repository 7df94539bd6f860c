// Dense mask application for gfx950 (MI355X):  out[f, k] (+)= sum_p tile[f, p] * masks[k, p]
//
// Replaces ApplyMasksEngine.process_flat for dense stacks (src/libertem/udf/masks.py:59-77:
// torch.mm / `flat_tile @ masks`) fused with the tile's dtype conversion
// (io/dataset/memory.py:102-105) and the `+=` into the result buffer (udf/masks.py:389-392).
//
// Two device paths:
//   * k_dense_mfma    : f32 result (also complex64 = 2 real columns per mask) for
//                       u8/i8/u16/i16/f32 tiles.  v_mfma_f32_16x16x4_f32 (exact f32 FMA chain),
//                       frames on the M axis straight from HBM into VGPRs (non-temporal 16-B
//                       loads, converted in registers), mask columns staged through LDS from a
//                       pre-swizzled device image, pixels on the K axis.
//   * k_dense_generic : every other dtype combination the reference supports (float64, complex,
//                       wrap-around integers); plain VALU, correctness first.
#include "ltmi_common.h"
#include <vector>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <typeinfo>
#include <cmath>
#include <type_traits>
#include <unordered_map>

namespace ltmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

constexpr int KC = 256;                    // pixels per staged mask chunk
constexpr int GROUP = 16;                  // mask columns per MFMA group (N of 16x16x4)
constexpr int CHUNK_FLOATS = GROUP * KC;   // 4096 floats = 16 KiB per (group, chunk)

// Position of mask value (column n of a group, pixel q of a chunk) inside the 16-KiB image
// block.  Lane (n = lane&15, kg = lane>>4) of a wave needs, for pixel block `blk` (32 px) and
// half h, the four pixels q = blk*32 + kg*8 + h*4 + {0..3} of its column n as ONE ds_read_b128.
// The 16-B unit index is XOR-ed with n so that the 16 lanes of every ds_read_b128 service group
// (which always hold 16 different n) hit 16 different 16-B slots of the 256-B LDS row:
// conflict-free without padding, and the image can be copied into LDS linearly.
__host__ __device__ static inline int img_index(int n, int q) {
    const int blk = q >> 5, kg = (q >> 3) & 3, j = q & 7;
    const int unit = (kg * 16 + blk * 2 + (j >> 2)) ^ n;
    return n * KC + unit * 4 + (j & 3);
}

// raw stack (n_masks, n_px) of f32 [cpm = 1] or interleaved complex64 [cpm = 2]  ->  image
__global__ void k_build_image(const float *__restrict__ src, float *__restrict__ img,
                              int64_t n_masks, int cpm, int64_t n_px, int n_chunks) {
    const int64_t total = n_masks * cpm * n_px;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        // i enumerates the source linearly: ((k * n_px + p) * cpm + part)
        const int part = (int)(i % cpm);
        const int64_t kp = i / cpm;
        const int64_t k = kp / n_px, p = kp % n_px;
        const int col = (int)(k * cpm + part);
        const int g = col / GROUP, n = col % GROUP;
        const int c = (int)(p / KC), q = (int)(p % KC);
        img[((size_t)g * n_chunks + c) * CHUNK_FLOATS + img_index(n, q)] = src[i];
    }
}

// the float16 images of the same geometries (k_dense_lds X16): where the float32 image has pixels
// q .. q+3 of column n in the 16-byte unit (U + 0) ^ n and q+4 .. q+7 in (U + 1) ^ n (U = kg * (kb / 16)
// + blk * 2, q = blk*32 + kg*8), the float16 image has w1 of all 8 pixels in the first unit and w2 in the
// second, w * scale[column] = w1 + w2 (float16: round to nearest, then the residual).  Index in HALVES
// relative to the start of the 16-column group block.
__host__ __device__ static inline int h16_index(int n, int q, int kb, int plane) {
    const int blk = q >> 5, kg = (q >> 3) & 3, j = q & 7;
    const int unit = (kg * (kb / 16) + blk * 2 + plane) ^ (n & (kb / 16 - 1));      // (see img2_index)
    return (n * kb + unit * 4) * 2 + j;
}

// do two float16 pieces carry the (column-scaled) weight to 2^-19 relative?  (host and device: same IEEE
// conversions, same answer -- the host lists the weights that fail, the image builder leaves them out)
__host__ __device__ static inline bool h16_pieces_exact_enough(float ws) {
    const _Float16 w1 = (_Float16)ws;
    const float r = ws - (float)w1;
    const _Float16 w2 = (_Float16)r;
    return fabsf(r - (float)w2) <= ldexpf(fabsf(ws), -19);
}

// layout: 1 = standard image (one group tile, chunks of KC), 2 = slot-major image 2 (ng groups per slot
// of kb = 128 pixels), 3 = image 3 without VALU columns (ng groups per slot, slot_floats floats per slot)
__global__ void k_build_image_h16(const float *__restrict__ src, _Float16 *__restrict__ img,
                                  int64_t n_masks, int cpm, int64_t n_px, int n_slots, int ng,
                                  int layout, int slot_floats, const float *__restrict__ scale,
                                  const float *__restrict__ amax) {
    const int64_t total = n_masks * cpm * n_px;
    const int kb = layout == 1 ? KC : 128;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int part = (int)(i % cpm);
        const int64_t kp = i / cpm;
        const int64_t k = kp / n_px, p = kp % n_px;
        const int col = (int)(k * cpm + part);
        const int g = col / GROUP, n = col % GROUP;
        const int slot = (int)(p / kb), q = (int)(p % kb);
        size_t base;                                       // floats
        if (layout == 1) base = ((size_t)g * n_slots + slot) * CHUNK_FLOATS;
        else if (layout == 2) base = ((((size_t)(g / ng) * n_slots + slot) * ng + g % ng) * GROUP) * kb;
        else base = (size_t)slot * slot_floats + (size_t)g * GROUP * kb;
        // (power-of-two scale: exact; the few small weights that two float16 pieces do not carry to 2^-19
        // relative are left out here and added in float32 by the epilogue of k_dense_lds -- same rule as the host's)
        float ws = src[i] * scale[col];
        if (ws != 0.f && fabsf(src[i]) < ldexpf(amax[col], -20) && !h16_pieces_exact_enough(ws)) ws = 0.f;
        const _Float16 w1 = (_Float16)ws;
        const _Float16 w2 = (_Float16)(ws - (float)w1);
        img[base * 2 + h16_index(n, q, kb, 0)] = w1;
        img[base * 2 + h16_index(n, q, kb, 1)] = w2;
    }
}

constexpr int DENSE_TAIL_MAX = 64;     // weights a stack's float16 images may leave to the float32 epilogue of k_dense_lds

// ---- per-input-dtype loading / conversion of 8 consecutive pixels ---------------------------
template <typename T> struct InTraits;

template <> struct InTraits<uint16_t> {
    typedef u32x4 raw_t;
    static __device__ __forceinline__ raw_t load(const uint16_t *p) {
        return __builtin_nontemporal_load((const u32x4 *)p);
    }
    static __device__ __forceinline__ void cvt(const raw_t &r, float (&f)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = (float)(r[i] & 0xffffu);
            f[2 * i + 1] = (float)(r[i] >> 16);
        }
    }
};
template <> struct InTraits<int16_t> {
    typedef u32x4 raw_t;
    static __device__ __forceinline__ raw_t load(const int16_t *p) {
        return __builtin_nontemporal_load((const u32x4 *)p);
    }
    static __device__ __forceinline__ void cvt(const raw_t &r, float (&f)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = (float)((int)(r[i] << 16) >> 16);
            f[2 * i + 1] = (float)((int)r[i] >> 16);
        }
    }
};
template <> struct InTraits<uint8_t> {
    typedef u32x2 raw_t;
    static __device__ __forceinline__ raw_t load(const uint8_t *p) {
        return __builtin_nontemporal_load((const u32x2 *)p);
    }
    static __device__ __forceinline__ void cvt(const raw_t &r, float (&f)[8]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int b = 0; b < 4; ++b) f[4 * i + b] = (float)((r[i] >> (8 * b)) & 0xffu);
        }
    }
};
template <> struct InTraits<int8_t> {
    typedef u32x2 raw_t;
    static __device__ __forceinline__ raw_t load(const int8_t *p) {
        return __builtin_nontemporal_load((const u32x2 *)p);
    }
    static __device__ __forceinline__ void cvt(const raw_t &r, float (&f)[8]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int b = 0; b < 4; ++b)
                f[4 * i + b] = (float)((int)(r[i] << (24 - 8 * b)) >> 24);
        }
    }
};
template <> struct InTraits<float> {
    struct raw_t { f32x4 a, b; };
    static __device__ __forceinline__ raw_t load(const float *p) {
        raw_t r;
        r.a = __builtin_nontemporal_load((const f32x4 *)p);
        r.b = __builtin_nontemporal_load((const f32x4 *)p + 1);
        return r;
    }
    static __device__ __forceinline__ void cvt(const raw_t &r, float (&f)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[i] = r.a[i]; f[4 + i] = r.b[i]; }
    }
};

// ---- the MFMA kernel ---------------------------------------------------------------------------
// grid = (frame tiles, ksplit, column-group tiles); block = WAVES*64.
// Each wave owns MT*16 consecutive frames (M), all waves of a block share the staged mask chunk.
template <typename T, int MT, int NG, int WAVES, bool ALIGNED>
__global__ void __launch_bounds__(WAVES * 64)
k_dense_mfma(const T *__restrict__ tile, int64_t ld, int64_t n_frames, int64_t n_px,
             const float *__restrict__ img, int n_chunks, float *__restrict__ out, int64_t ld_out,
             int n_cols, int accumulate, float *__restrict__ partials, int ksplit) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = WAVES * 64;
    constexpr int STAGE = NG * CHUNK_FLOATS;   // floats per LDS stage
    constexpr int BUNITS = STAGE / 4 / NT;     // 16-B units each thread copies per stage
    static_assert(STAGE % (4 * NT) == 0, "stage must divide evenly");
    using TR = InTraits<T>;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int m = lane & 15, kg = lane >> 4;
    const int g0 = blockIdx.z * NG;
    const int ks = blockIdx.y;

    const int n_full = ALIGNED ? (int)(n_px / KC) : 0;   // chunks readable with vector loads
    const int per = (n_chunks + ksplit - 1) / ksplit;
    const int c_begin = ks * per;
    const int c_end = min(n_chunks, c_begin + per);
    const int cf_end = min(c_end, n_full);

    const int64_t f_wave = (int64_t)blockIdx.x * (WAVES * MT * 16) + wave * (MT * 16);
    const T *rowp[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int64_t f = f_wave + mt * 16 + m;
        if (f > n_frames - 1) f = n_frames - 1;   // clamp: loads stay valid, result discarded
        rowp[mt] = tile + f * ld + kg * 8;
    }

    // acc: the running tiles of at most 4 chunks (1024 pixels); acc2: the long sums (two accumulation
    // levels keep the float32 chains short on large frames, see k_dense_lds)
    f32x4 acc[MT][NG], acc2[MT][NG];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[mt][g] = acc2[mt][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto flush_acc = [&]() {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                acc2[mt][g] += acc[mt][g];
                acc[mt][g] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    };

    // this thread's share of the linear image -> LDS copy
    const u32x4 *img_units = (const u32x4 *)img;
    auto img_unit_index = [&](int i, int c) -> int64_t {
        const int u = i * NT + tid;                  // unit within the stage
        const int g = u / (CHUNK_FLOATS / 4);
        const int w = u % (CHUNK_FLOATS / 4);
        return ((int64_t)(g0 + g) * n_chunks + c) * (CHUNK_FLOATS / 4) + w;
    };

    // LDS read offsets (floats) of this lane's column for (blk, h): base + ((blk*2+h)^m)*4
    const int lds_lane_base = m * KC + kg * 64;

    if (c_begin < cf_end) {
        typename TR::raw_t raw[MT][8];
        u32x4 breg[BUNITS];
        // prologue: mask chunk c_begin -> stage 0, frame data of chunk c_begin -> registers
#pragma unroll
        for (int i = 0; i < BUNITS; ++i) breg[i] = img_units[img_unit_index(i, c_begin)];
#pragma unroll
        for (int blk = 0; blk < 8; ++blk)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                raw[mt][blk] = TR::load(rowp[mt] + (int64_t)c_begin * KC + blk * 32);
#pragma unroll
        for (int i = 0; i < BUNITS; ++i) ((u32x4 *)lds)[i * NT + tid] = breg[i];
        __syncthreads();

        for (int c = c_begin; c < cf_end; ++c) {
            const int cn = min(c + 1, cf_end - 1);      // branch-free prefetch target
            const int s = (c - c_begin) & 1;
            if (((c - c_begin) & 3) == 3) flush_acc();
#pragma unroll
            for (int i = 0; i < BUNITS; ++i) breg[i] = img_units[img_unit_index(i, cn)];
            const float *ldsb = lds + s * STAGE + lds_lane_base;
#pragma unroll
            for (int blk = 0; blk < 8; ++blk) {
                float a[MT][8];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) TR::cvt(raw[mt][blk], a[mt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    raw[mt][blk] = TR::load(rowp[mt] + (int64_t)cn * KC + blk * 32);
                f32x4 b[NG][2];
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        b[g][h] = *(const f32x4 *)(ldsb + g * CHUNK_FLOATS +
                                                   (((blk * 2 + h) ^ m) << 2));
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int g = 0; g < NG; ++g)
                            acc[mt][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                a[mt][j], b[g][j >> 2][j & 3], acc[mt][g], 0, 0, 0);
            }
            u32x4 *ldsn = (u32x4 *)(lds + (s ^ 1) * STAGE);
#pragma unroll
            for (int i = 0; i < BUNITS; ++i) ldsn[i * NT + tid] = breg[i];
            __syncthreads();
        }
    }

    // chunks that need guarded element loads: the ragged last chunk (n_px % 256 != 0), or every
    // chunk when the rows are not 16-byte aligned
    for (int c = max(c_begin, n_full); c < c_end; ++c) {
        __syncthreads();
        if (((c - c_begin) & 3) == 3) flush_acc();
#pragma unroll
        for (int i = 0; i < BUNITS; ++i)
            ((u32x4 *)lds)[i * NT + tid] = img_units[img_unit_index(i, c)];
        __syncthreads();
        const float *ldsb = lds + lds_lane_base;
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) {
            const int64_t p0 = (int64_t)c * KC + blk * 32;   // + kg*8 is folded into rowp
            float a[MT][8];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    a[mt][j] = (p0 + kg * 8 + j < n_px) ? (float)rowp[mt][p0 + j] : 0.f;
            f32x4 b[NG][2];
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    b[g][h] = *(const f32x4 *)(ldsb + g * CHUNK_FLOATS +
                                               (((blk * 2 + h) ^ m) << 2));
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        acc[mt][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                            a[mt][j], b[g][j >> 2][j & 3], acc[mt][g], 0, 0, 0);
        }
    }

    // epilogue.  C/D layout of 16x16x4: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t f = f_wave + mt * 16 + kg * 4 + r;
                const int col = (g0 + g) * GROUP + m;
                if (f < n_frames && col < n_cols) {
                    const float v = acc[mt][g][r] + acc2[mt][g][r];
                    if (ksplit == 1) {
                        float *p = out + f * ld_out + col;
                        *p = accumulate ? (*p + v) : v;
                    } else {
                        partials[((int64_t)ks * n_frames + f) * n_cols + col] = v;
                    }
                }
            }
}

// ---- frames through LDS with full-line LDS-DMA (k_dense_lds) ---------------------------------------
// probes/stream_probe.hip: the fragment-shaped loads of k_dense_mfma (16 frame rows x 64 B per wave
// instruction) top out at ~5.8 TB/s with no compute; 1-KiB-contiguous non-temporal reads reach
// 7.0 TB/s.  Here every wave instruction is a global_load_lds_dwordx4 that moves 4 frame rows x 256 B
// (whole 128-B lines) straight into LDS; MFMA A fragments are then ds_read from there.
//
//   * a workgroup is 128 frames: 4 waves (one per SIMD) of 32 frames = two 16-frame MFMA tiles each
//     (TILES = 2; the first version had 8 waves of 16 frames: bit-identical results, but every mask
//     fragment read from LDS fed only one tile -- twice the mask-fragment LDS traffic per MFMA, and
//     these kernels sit at the board's power cap: profiles/r02_tiles.txt, 5-11 % slower on C2);
//     per wave a ring of sub-chunk slots (32 rows x 256 B = 8 KiB each).
//   * the 16 pieces (16 B) of a row are stored at piece ^ (row & 15): the 16 lanes of every
//     ds_read_b128 service group (16 different rows) hit 16 different slots.  LDS-DMA writes
//     lane-linear, so the permutation is applied to the per-lane SOURCE address.
//   * mask slots also arrive by LDS-DMA (linear copies of the pre-swizzled image), 2 slots, shared
//     by the waves of the workgroup, one s_barrier per slot.
//   * counted waits, never vmcnt(0) in the loop (DMA completes in order per wave).
//   * rows need not be 16-B aligned: the DMA reads from any element-aligned address (ltmi_common.h
//     vector_loads_ok), the piece permutation is relative to the row start.
//   * two accumulation levels (see acc2 in the kernel): the float32 chains stay <= 1024 pixels long.
constexpr int V2_ROWS = 16;                         // frames per MFMA tile
constexpr int V2_SUB_BYTES = 256;                   // bytes of a row per sub-chunk

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;

// The kernel: wave-private frame ring filled by global_load_lds, counted vmcnt, one barrier per mask
// slot, fragments double-buffered in registers, conversions batched ahead of the MFMAs; generic over
//   * T: a sub-chunk is always 256 B of a row = 256 / sizeof(T) pixels (u8 256, u16 128, f32 64);
//   * NG column groups per wave (C5: 25 complex masks = 50 real columns = 4 groups): the frame
//     fragment of a block is converted once and used for NG x 8 MFMAs.
// The mask image for NG > 1 ("image 2") is slot-major: slot k = pixels [k*KB, (k+1)*KB), KB = 128,
// holds its NG groups back to back (NG x 8 KiB), so one slot is a linear DMA copy:
//   float offset = (((gt * n_slots + k) * NG + g) * 16 + n) * KB + ((kg*(KB/16) + blk*2 + h) ^ n)*4 + (q&3)
// with blk = q >> 5, kg = (q >> 3) & 3, h = (q >> 2) & 1 inside the slot; NG = 1 uses the standard
// image (KB = 256, which this formula reproduces).
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// NE > 0 ("extras"): NG groups go through the matrix cores and NE further columns (the remainder
// of a stack with 16 NG + NE columns, e.g. 25 complex masks = 48 + 2) are accumulated on the VALU
// from the already converted frame fragments -- instead of a whole extra MFMA group of padding.
// NG = 0: stacks of at most 4 columns (CoM, single-mask analyses) use the VALU columns ONLY.
// Their slot: NG x 8 KiB of groups, then NE / 2 column pairs of 1 KiB, (pixel, column of the pair)
// floats -- one v_pk_fma_f32 per pixel and pair --, rounded up to 4 KiB.
// bytes of a mask slot of 128 pixels with ng MFMA groups + ne VALU columns (kernel and image builder)
__host__ __device__ constexpr int ext_slot_bytes(int ng, int ne) {
    return (ng * GROUP * 128 * 4 + ne * 128 * 4 + 4095) / 4096 * 4096;
}

// TILES: 16-frame tiles per wave.  With 2, a workgroup is 4 waves of 32 frames (same 128 frames, same
// LDS): one mask fragment read from LDS feeds the MFMAs of two frame tiles, i.e. half the mask-fragment
// LDS traffic per MFMA -- these kernels run into the board's power cap, so energy per frame is time.
template <int NG, int NE = 0, int TILES = 1> struct LdsCfg {
    static constexpr int KB = (NG == 1 && NE == 0) ? KC : 128;  // pixels per mask slot
    // bytes per mask slot; with extras: the NG groups + NE / 2 column pairs, rounded up to 4 KiB (each
    // of the 4 waves copies an equal share of a slot in whole 1-KiB DMA instructions): 12 / 20 / 28 KiB
    static_assert(NE == 0 || TILES == 2, "the VALU-column image is laid out for 4-wave workgroups");
    static constexpr int BSLOT = NE > 0 ? ext_slot_bytes(NG, NE) : NG * GROUP * KB * 4;
    static constexpr int EXTRA_OFF = NG * GROUP * KB;           // float offset of the extras in a slot
    static_assert(NE == 0 || (NG * GROUP * KB * 4 + NE * KB * 4 <= BSLOT), "extras must fit the slot");
    static constexpr int RING = BSLOT > 16384 ? 3 : 4;          // frame ring depth (sub-chunks)
    static constexpr int WAVES = 8 / TILES;
    static constexpr int ROWS = V2_ROWS * TILES;                // frames per wave
    static constexpr int ASLOT = ROWS * V2_SUB_BYTES;           // bytes per wave and ring slot
    static constexpr int WG_ROWS = WAVES * ROWS;                // 128 frames per workgroup
    static constexpr int LDS_BYTES = RING * WAVES * ASLOT + 2 * BSLOT;         // 160 KiB
};

// (the XOR must stay inside a lane group's kb / 16 units: with 128-pixel slots -- 8 units per group -- the
// full 4-bit n flipped the group, lanes (n, kg) and (n ^ 8, kg ^ 1) asked for the same unit of two rows 4 KiB
// apart: 4 conflict cycles per fragment read, SQ_LDS_BANK_CONFLICT 2^23 per column group and C5-sized launch
// in rounds 2 and 3; profiles/r04_lds_conflicts.txt, probes/lds_b128_probe.hip)
__host__ __device__ static inline int img2_index(int n, int q, int kb) {
    const int blk = q >> 5, kg = (q >> 3) & 3, j = q & 7;
    const int unit = (kg * (kb / 16) + blk * 2 + (j >> 2)) ^ (n & (kb / 16 - 1));
    return n * kb + unit * 4 + (j & 3);
}

// raw stack -> image 2 (see above); ng = groups per tile, n_gt = group tiles
__global__ void k_build_image2(const float *__restrict__ src, float *__restrict__ img,
                               int64_t n_masks, int cpm, int64_t n_px, int n_slots, int ng, int kb) {
    const int64_t total = n_masks * cpm * n_px;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int part = (int)(i % cpm);
        const int64_t kp = i / cpm;
        const int64_t k = kp / n_px, p = kp % n_px;
        const int col = (int)(k * cpm + part);
        const int g = col / GROUP, n = col % GROUP;
        const int gt = g / ng, gl = g % ng;
        const int slot = (int)(p / kb), q = (int)(p % kb);
        img[((((size_t)gt * n_slots + slot) * ng + gl) * GROUP) * kb + img2_index(n, q, kb)] = src[i];
    }
}

// IND (indirect rows, used for shifted masks): workgroup b processes the 128 frames
// rows[128 b .. 128 b + 127] (-1 = padding) against its own mask image wg_img[b].
// standard image (img_index layout, one group tile) of the stack shifted by (dy, dx):
// mask'[k](y, x) = mask[k](y - dy, x - dx) inside the frame, 0 outside (udf/masks.py:85-124)
__global__ void k_build_image_shifted(const float *__restrict__ src, float *__restrict__ img,
                                      int64_t n_masks, int cpm, int sig_h, int sig_w, int dy, int dx,
                                      int n_chunks) {
    const int64_t n_px = (int64_t)sig_h * sig_w;
    const int64_t total = n_masks * cpm * n_px;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int part = (int)(i % cpm);
        const int64_t kp = i / cpm;
        const int64_t k = kp / n_px, p = kp % n_px;            // destination pixel
        const int y = (int)(p / sig_w), x = (int)(p % sig_w);
        const int ys = y - dy, xs = x - dx;
        float v = 0.f;
        if (ys >= 0 && ys < sig_h && xs >= 0 && xs < sig_w)
            v = src[((k * n_px) + (int64_t)ys * sig_w + xs) * cpm + part];
        const int col = (int)(k * cpm + part);
        const int g = col / GROUP, n = col % GROUP;
        const int c = (int)(p / KC), q = (int)(p % KC);
        img[((size_t)g * n_chunks + c) * CHUNK_FLOATS + img_index(n, q)] = v;
    }
}

// ... and its float16 image for k_dense_lds X16 (w * scale = w1 + w2, scale = 1 / inv_scale: powers of two)
__global__ void k_build_image_shifted_h16(const float *__restrict__ src, _Float16 *__restrict__ img,
                                          int64_t n_masks, int cpm, int sig_h, int sig_w, int dy, int dx,
                                          int n_chunks, const float *__restrict__ inv_scale) {
    const int64_t n_px = (int64_t)sig_h * sig_w;
    const int64_t total = n_masks * cpm * n_px;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int part = (int)(i % cpm);
        const int64_t kp = i / cpm;
        const int64_t k = kp / n_px, p = kp % n_px;            // destination pixel
        const int y = (int)(p / sig_w), x = (int)(p % sig_w);
        const int ys = y - dy, xs = x - dx;
        float v = 0.f;
        if (ys >= 0 && ys < sig_h && xs >= 0 && xs < sig_w)
            v = src[((k * n_px) + (int64_t)ys * sig_w + xs) * cpm + part];
        const int col = (int)(k * cpm + part);
        const int g = col / GROUP, n = col % GROUP;
        const int c = (int)(p / KC), q = (int)(p % KC);
        const float ws = v * (1.0f / inv_scale[col]);
        const _Float16 w1 = (_Float16)ws;
        const _Float16 w2 = (_Float16)(ws - (float)w1);
        _Float16 *block = img + (((size_t)g * n_chunks + c) * CHUNK_FLOATS) * 2;
        block[h16_index(n, q, KC, 0)] = w1;
        block[h16_index(n, q, KC, 1)] = w2;
    }
}

// raw stack -> image 3 (NG groups + extras, 32-KiB slots of 128 pixels): groups as in image 2, then
// the columns >= 16 NG in pairs, [pair][pixel][2] floats
__global__ void k_build_image3(const float *__restrict__ src, float *__restrict__ img,
                               int64_t n_masks, int cpm, int64_t n_px, int n_slots, int ng,
                               int slot_floats) {
    constexpr int kb = 128;
    const int64_t total = n_masks * cpm * n_px;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int part = (int)(i % cpm);
        const int64_t kp = i / cpm;
        const int64_t k = kp / n_px, p = kp % n_px;
        const int col = (int)(k * cpm + part);
        const int slot = (int)(p / kb), q = (int)(p % kb);
        float *base = img + (size_t)slot * slot_floats;
        if (col < ng * GROUP) {
            const int g = col / GROUP, n = col % GROUP;
            base[(size_t)g * GROUP * kb + img2_index(n, q, kb)] = src[i];
        } else {
            const int e = col - ng * GROUP;             // column pairs: (pair, pixel, column of the pair)
            base[(size_t)ng * GROUP * kb + (e / 2) * (2 * kb) + q * 2 + (e & 1)] = src[i];
        }
    }
}

// X16 (unsigned 1- and 2-byte pixels, no VALU columns): the products are formed EXACTLY from float16
// operands on v_mfma_f32_16x16x32_f16 instead of converting the pixels to float32 -- a pixel is
// lo + 256 hi (two bytes, each a float16 number: 0x6400 | b is 1024 + b, one v_perm_b32 and one
// v_pk_add_f16 per pixel pair), a weight times its column's power-of-two scale (max|w| S in [2^14, 2^15))
// is w1 + w2 (two float16, 22 bits; the image holds w1 of the lane's 8 pixels in the 16-byte unit h = 0 and w2 in h = 1: same
// bytes, same addresses as the float32 image).  lo w1 + lo w2 go to one accumulator, hi w1 + hi w2 to a
// second one that counts 256-fold: 4 x 16 matrix-pipe cycles per 16 frames x 16 columns x 32 pixels where
// float32 takes 8 x 32 -- these kernels run at the board's power cap with the f32 pipe 65 % busy (C2),
// so the pipe's energy is time.  Same scheme as k_bell_flat (ltmi_bell.hip); `inv_scale`: 1 / scale
// per column, applied to the sums.
template <typename T, int NG, int ABL = 0, int IND = 0, int NE = 0, int TILES = 1, bool X16 = false>
__global__ void __launch_bounds__(512 / TILES)                  // LdsCfg::WAVES * 64
k_dense_lds(const T *__restrict__ tile, int64_t ld, int64_t n_frames, int64_t n_px,
            const float *__restrict__ img, int n_slots, float *__restrict__ out, int64_t ld_out,
            int n_cols, int accumulate, float *__restrict__ partials, int ksplit_arg,
            const int32_t *__restrict__ rows = nullptr,
            const float *const *__restrict__ wg_img = nullptr, int *__restrict__ kcount = nullptr,
            const float *__restrict__ inv_scale = nullptr, const float *__restrict__ tail_val = nullptr,
            const int32_t *__restrict__ tail_col = nullptr, const int32_t *__restrict__ tail_px = nullptr,
            int n_tail = 0) {
    static_assert(!X16 || (NE == 0 && NG >= 1 && ABL == 0 && sizeof(T) <= 2),
                  "X16: 1- / 2-byte integer pixels on the matrix cores only");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    using TR = InTraits<T>;
    using CFG = LdsCfg<NG, NE, TILES>;
    constexpr int WAVES = CFG::WAVES, RING = CFG::RING, KB = CFG::KB, BSLOT = CFG::BSLOT;
    constexpr int ROWS = CFG::ROWS, ASLOT = CFG::ASLOT;
    // bytes of a frame row per sub-chunk: 256, or 128 for 1-byte pixels against 128-pixel mask slots
    // (a sub-chunk must not straddle mask slots); rows sit SUBB bytes apart in a ring slot
    constexpr int SUBB = (sizeof(T) == 1 && KB < V2_SUB_BYTES) ? 128 : V2_SUB_BYTES;
    constexpr int PPR = SUBB / 16;                      // 16-byte pieces per row
    constexpr int RPI = 64 / PPR;                       // rows one DMA instruction (64 x 16 B) covers
    constexpr int TILE_BYTES = V2_ROWS * SUBB;          // one 16-frame tile of a ring slot
    constexpr int ND = (V2_ROWS / RPI) * TILES;         // DMA instructions per sub-chunk
    constexpr int SPX = SUBB / (int)sizeof(T);          // pixels per sub-chunk
    static_assert(SPX <= KB && KB % SPX == 0, "a sub-chunk must not straddle mask slots");
    constexpr int PER = KB / SPX;                       // sub-chunks per mask slot
    constexpr int BLKS = SPX / 32;                      // MFMA pixel blocks per sub-chunk (2/4/8)
    constexpr int NT = WAVES * 64;
    constexpr int A_BYTES = RING * WAVES * ASLOT;
    constexpr int BPW = BSLOT / WAVES;                  // mask-slot bytes each wave copies
    constexpr int NBI = BPW / 1024;                     // ... in this many DMA instructions
    constexpr int A_N = ND * (RING - 2);                // DMA instructions of the later sub-chunks
    static_assert(A_N + 2 * NBI < 64, "vmcnt is a 6-bit counter");
    // accumulators per group and frame tile (X16: [0] the low bytes' products, [1] the high bytes')
    constexpr int NACC = (NG == 1 || X16) ? 2 : 1;
    constexpr int SLOT_FLOATS = BSLOT / 4;
    constexpr bool CVT = !std::is_same<T, float>::value;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int m = lane & 15, kg = lane >> 4;
    const int ks = blockIdx.y;
    const int gt = blockIdx.z;

    // ksplit_arg < 0: the parts of a pixel-split launch take the mask slots in turn (part ks: slots ks, ks + ksplit,
    // ...) instead of a contiguous range each, so that at any moment the workgroups of a frame block read NEIGHBOURING
    // 512-byte pieces of their frames, not pieces 4 - 16 KiB apart (profiles/r05_small_tiles.txt)
    // (-ksplit_arg = ksplit | log2(run) << 16: runs of 2^.. consecutive slots in turn)
    const bool kstr = ksplit_arg < 0 && IND == 0 && ABL == 0;
    const int ksplit = ksplit_arg < 0 ? ((-ksplit_arg) & 0xffff) : ksplit_arg;
    const int run_sh = kstr ? ((-ksplit_arg) >> 16) : 0, run = 1 << run_sh;
    const int n_full = (int)(n_px / KB);                 // slots readable by DMA
    const int per = (n_slots + ksplit - 1) / ksplit;
    const int k_begin = ks * per;
    const int k_end = min(n_slots, k_begin + per);
    const int kf_end = min(k_end, n_full);
    const int n_runs = (n_full + run - 1) >> run_sh;     // runs of the whole stack, the last one may be short
    const int my_runs = n_runs > ks ? (n_runs - ks + ksplit - 1) / ksplit : 0;
    const int nk = kstr ? max(0, (my_runs << run_sh) - ((my_runs > 0 && (n_runs - 1) % ksplit == ks)
                                                         ? (n_runs << run_sh) - n_full : 0))
                        : max(0, kf_end - k_begin);
    const bool has_ragged = kstr ? (n_slots > n_full && ks == (n_full >> run_sh) % ksplit) : (k_end > n_full);
    // g-th slot of this part
    auto slot_of = [&](int g) { return kstr ? (((ks + (g >> run_sh) * ksplit) << run_sh) + (g & (run - 1))) : k_begin + g; };
    const float *img_t = IND == 1 ? wg_img[blockIdx.x] : img + (size_t)gt * n_slots * SLOT_FLOATS;

    const int64_t f_wave = (int64_t)blockIdx.x * (WAVES * ROWS) + wave * ROWS;
    // IND == 1 (shifted masks): row r of this wave is frame rows[...] of the tile, read AND written
    // there (-1 = nothing).  IND == 2 (a region of interest without a gathered copy): result row i is
    // the product of frame rows[i] of the tile, i < n_frames.
    auto frame_of = [&](int r) -> int64_t {                 // result row
        if (IND == 1) return rows[f_wave + r];
        const int64_t f = f_wave + r;
        return f < n_frames ? f : -1;
    };
    auto src_frame_of = [&](int r) -> int64_t {             // frame to read (clamped: loads stay valid)
        int64_t f = frame_of(r);
        if (IND == 2) return f < 0 ? (int64_t)rows[0] : (int64_t)rows[f];
        if (f < 0) f = IND ? 0 : n_frames - 1;
        return f;
    };
    unsigned char *a_base = lds_raw + wave * ASLOT;      // + slot * (WAVES * ASLOT)
    unsigned char *b_base = lds_raw + A_BYTES;           // + bslot * BSLOT

    constexpr int NGA = NG > 0 ? NG : 1;                // (array dimensions; NG = 0: VALU columns only)
    f32x4 acc[TILES][NGA][NACC];
    // second accumulation level: every FLUSH_PX pixels the running tiles are added to acc2 and start
    // again from zero.  One float32 chain over a whole 512 x 512 (1024 x 1024) frame drifts by 1.5e-5
    // (6e-5) of the sum on all-positive data -- round-off random walk over 32 768 (131 072) MFMA steps;
    // with chains of 128 steps + at most a few thousand second-level additions it stays below 2e-6.
    constexpr int FLUSH_PX = 1024;
    f32x4 acc2[TILES][NGA];
#pragma unroll
    for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
        for (int g = 0; g < NGA; ++g) acc2[tl][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int x = 0; x < NACC; ++x) acc[tl][g][x] = f32x4{0.f, 0.f, 0.f, 0.f};
    static_assert(NE % 2 == 0, "VALU columns come in pairs");
    // VALU column pairs: partial over this lane's pixels -- two levels: the running sum of ONE sub-chunk
    // (16 / 32 / 64 products per lane) is added to the long sum at the end of the sub-chunk, so the
    // float32 chains stay short (a single chain over 65 536 products of a 512 x 512 frame drifts by
    // ~2e-5 relative)
    f32x2 acc_e[TILES][NE > 0 ? NE / 2 : 1], acc_e2[TILES][NE > 0 ? NE / 2 : 1];
#pragma unroll
    for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
        for (int c = 0; c < (NE > 0 ? NE / 2 : 1); ++c)
            acc_e[tl][c] = acc_e2[tl][c] = f32x2{0.f, 0.f};

    // lane-constant parts of the fragment addresses
    const int a_lane = m * SUBB;                         // bytes inside a frame tile of a ring slot
    const int b_lane = m * KB;                           // floats inside a group of a mask slot
    auto b_unit = [&](int blk_in_slot, int h) {          // swizzled 16-B unit of (blk, h) for this lane
        return ((kg * (KB / 16) + blk_in_slot * 2 + h) ^ (m & (KB / 16 - 1))) << 2;
    };
    // X16: the lane's 8 raw pixels -> float16 operands of their low and high bytes
    // (signed pixels: the top byte is a signed number -- its sign bit is flipped before the perm and
    // 128 more are taken off: 1024 + (b ^ 0x80) - 1152 = b as int8)
    auto bytes_f16 = [&](const auto &r, h16x8 &lo, h16x8 &hi) {
        constexpr bool SIGNED = std::is_signed<T>::value;
        const h16x2 kbias = {(_Float16)1024.f, (_Float16)1024.f};
        const h16x2 kbias_top = SIGNED ? h16x2{(_Float16)1152.f, (_Float16)1152.f} : kbias;
        h16x2 l[4], h[4];
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned w = SIGNED ? (r[i] ^ 0x80008000u) : r[i];
                l[i] = __builtin_bit_cast(h16x2, __builtin_amdgcn_perm(0x64646464u, w, 0x04020400u)) - kbias;
                h[i] = __builtin_bit_cast(h16x2, __builtin_amdgcn_perm(0x64646464u, w, 0x04030401u)) - kbias_top;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned w = SIGNED ? (r[i] ^ 0x80808080u) : r[i];
                l[2 * i] = __builtin_bit_cast(h16x2, __builtin_amdgcn_perm(0x64646464u, w, 0x04010400u)) - kbias_top;
                l[2 * i + 1] = __builtin_bit_cast(h16x2, __builtin_amdgcn_perm(0x64646464u, w, 0x04030402u)) - kbias_top;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) h[i] = h16x2{(_Float16)0.f, (_Float16)0.f};
        }
        lo = h16x8{l[0][0], l[0][1], l[1][0], l[1][1], l[2][0], l[2][1], l[3][0], l[3][1]};
        hi = h16x8{h[0][0], h[0][1], h[1][0], h[1][1], h[2][0], h[2][1], h[3][0], h[3][1]};
    };
    // (tile tl, group g) += the block's products; bw1 / bw2: the two 16-byte units of the mask fragment
    auto mfma_x16 = [&](int tl, int g, const h16x8 &lo, const h16x8 &hi, const f32x4 &bw1, const f32x4 &bw2) {
        const h16x8 w1 = __builtin_bit_cast(h16x8, bw1), w2 = __builtin_bit_cast(h16x8, bw2);
        acc[tl][g][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lo, w2, acc[tl][g][0], 0, 0, 0);
        acc[tl][g][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lo, w1, acc[tl][g][0], 0, 0, 0);
        if constexpr (sizeof(T) == 2) {
            acc[tl][g][NACC - 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, w2, acc[tl][g][NACC - 1], 0, 0, 0);
            acc[tl][g][NACC - 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, w1, acc[tl][g][NACC - 1], 0, 0, 0);
        }
    };

    // the main loop, compiled twice where a launch may be pixel-split in turn (KSTR): the contiguous order keeps its
    // addresses as `first sub-chunk + s` -- the slot arithmetic of the in-turn order costs the full C2 launch 5 %
    // (13 % with the float32 instruction) when it sits in the copy-issue path of every launch
    auto main_loop = [&](auto IN_TURN) {
        constexpr bool KSTR = decltype(IN_TURN)::value;
        const unsigned char *src[ND];
#pragma unroll
        for (int t = 0; t < ND; ++t) {
            const int r = RPI * t + lane / PPR;
            const int64_t f = src_frame_of(r);          // (clamped; such results are discarded)
            const int piece = (lane & (PPR - 1)) ^ (r & (PPR - 1));
            src[t] = (const unsigned char *)(tile + f * ld) + piece * 16;
        }
        const unsigned char *bsrc = (const unsigned char *)img_t + wave * BPW + lane * 16;
        // sub-chunks of this part: KSTR: counted from 0 in the order they are taken; else the frame rows' own numbers
        const int S0 = KSTR ? 0 : k_begin * PER, S1 = KSTR ? nk * PER : kf_end * PER;

        auto issue_a1 = [&](int s, int slot, int t) {
            if (ABL >= 2) return;
            const int sl = min(s, S1 - 1);
            const int sc = KSTR ? slot_of(sl / PER) * PER + sl % PER : sl;         // sub-chunk of the frame rows
            unsigned char *dst = a_base + slot * (WAVES * ASLOT);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[t] + (int64_t)sc * SUBB),
                                             (lds_ptr_t)(dst + t * 1024), 16, 0, 2 /*nt*/);
        };
        auto issue_b = [&](int gidx) {                  // the part's gidx-th mask slot -> LDS slot gidx & 1
            if (ABL >= 2) return;
            const int kk = KSTR ? slot_of(min(gidx, nk - 1)) : min(k_begin + gidx, kf_end - 1);
            unsigned char *dst = b_base + (gidx & 1) * BSLOT + wave * BPW;
            const unsigned char *sp = bsrc + (int64_t)kk * BSLOT;
#pragma unroll
            for (int u = 0; u < NBI; ++u)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(sp + u * 1024),
                                                 (lds_ptr_t)(dst + u * 1024), 16, 0, 0);
        };

#pragma unroll
        for (int t = 0; t < ND; ++t) issue_a1(S0, 0, t);
        issue_b(0);
#pragma unroll
        for (int d = 1; d < RING - 1; ++d)
#pragma unroll
            for (int t = 0; t < ND; ++t) issue_a1(S0 + d, d, t);

        // One sub-chunk.  ph = i % UNROLL as a compile-time constant in the unrolled main loop
        // (ring slot, mask slot and block offsets become immediates) or -1 in the generic tail.
        auto iteration = [&](int s, auto ph) {
            constexpr int PH = decltype(ph)::value;
            const int i = s - S0;
            const int ip = PH >= 0 ? PH % PER : i % PER;           // position inside the mask slot
            // DMA issue order per wave: ... A(s) | [B at the start of every PER-th iteration]
            // A(s+1) ... A(s+RING-2).  ip == 0: the slot issued PER iterations ago and A(s) must
            // have landed -> at most min(PER, RING-2) later sub-chunks may stay in flight.
            // ip > 0: A(s) must have landed; later sub-chunks and every mask slot issued after
            // A(s) (iterations i-ip-k*PER >= i-RING+2) may stay in flight.
            if (ip == 0) {
                constexpr int N0 = ND * (PER < RING - 2 ? PER : RING - 2);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N0) : "memory");
                if (ABL < 3) __builtin_amdgcn_s_barrier();
                issue_b(i / PER + 1);
            } else {
                int nb = 0;
#pragma unroll
                for (int k = 0; k < RING; ++k) nb += (ip + k * PER <= RING - 2) ? 1 : 0;
                if (nb == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_N) : "memory");
                else if (nb == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_N + NBI) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_N + 2 * NBI) : "memory");
            }
            const int slot = PH >= 0 ? PH % RING : i % RING;
            const int nslot = PH >= 0 ? (PH + RING - 1) % RING : (i + RING - 1) % RING;
            const int bsl = PH >= 0 ? (PH / PER) & 1 : (i / PER) & 1;
            const int blk0 = ip * BLKS;                             // block offset inside the mask slot
            const unsigned char *aslot = a_base + slot * (WAVES * ASLOT) + a_lane;
            const float *bslot = (const float *)(b_base + bsl * BSLOT) + b_lane;
            auto rd_a = [&](int tl, int blk) {
                const unsigned char *at = aslot + tl * TILE_BYTES;
                const int u = blk * 4 + kg;             // 8-pixel unit of this lane inside the sub-chunk
                if constexpr (sizeof(T) == 2) {
                    return *(const typename TR::raw_t *)(at + ((u ^ m) << 4));
                } else if constexpr (sizeof(T) == 4) {
                    typename TR::raw_t r;
                    r.a = *(const f32x4 *)(at + (((2 * u) ^ m) << 4));
                    r.b = *(const f32x4 *)(at + (((2 * u + 1) ^ m) << 4));
                    return r;
                } else {
                    return *(const typename TR::raw_t *)(at + (((u >> 1) ^ (m & (PPR - 1))) << 4) + (u & 1) * 8);
                }
            };
            auto rd_b = [&](int blk, int g, int h) {
                return *(const f32x4 *)(bslot + g * (GROUP * KB) + b_unit(blk0 + blk, h));
            };
            // extras: column pair c2, pixels 2h, 2h+1 of this lane's 8 pixels of the block, as
            // (col 2 c2, col 2 c2 + 1) per pixel (same address for the 16 lanes of a kg group: LDS
            // broadcast)
            auto rd_e = [&](int blk, int c2, int h) {
                return *(const f32x4 *)(bslot - b_lane + CFG::EXTRA_OFF + c2 * (2 * KB) +
                                        ((blk0 + blk) * 32 + kg * 8) * 2 + h * 4);
            };
            typename TR::raw_t raw_c[TILES];
#pragma unroll
            for (int tl = 0; tl < TILES; ++tl) raw_c[tl] = rd_a(tl, 0);
            f32x4 b_c[NGA][2];
#pragma unroll
            for (int g = 0; g < NG; ++g) { b_c[g][0] = rd_b(0, g, 0); b_c[g][1] = rd_b(0, g, 1); }
            f32x4 e_c[NE > 0 ? NE / 2 : 1][4];
#pragma unroll
            for (int c = 0; c < NE / 2; ++c)
#pragma unroll
                for (int h = 0; h < 4; ++h) e_c[c][h] = rd_e(0, c, h);
#pragma unroll
            for (int blk = 0; blk < BLKS; ++blk) {
                typename TR::raw_t raw_n[TILES];
#pragma unroll
                for (int tl = 0; tl < TILES; ++tl) raw_n[tl] = raw_c[tl];
                f32x4 b_n[NGA][2];
#pragma unroll
                for (int g = 0; g < NG; ++g) { b_n[g][0] = b_c[g][0]; b_n[g][1] = b_c[g][1]; }
                f32x4 e_n[NE > 0 ? NE / 2 : 1][4];
#pragma unroll
                for (int c = 0; c < NE / 2; ++c)
#pragma unroll
                    for (int h = 0; h < 4; ++h) e_n[c][h] = e_c[c][h];
                if (blk + 1 < BLKS) {
#pragma unroll
                    for (int tl = 0; tl < TILES; ++tl) raw_n[tl] = rd_a(tl, blk + 1);
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        b_n[g][0] = rd_b(blk + 1, g, 0);
                        b_n[g][1] = rd_b(blk + 1, g, 1);
                    }
#pragma unroll
                    for (int c = 0; c < NE / 2; ++c)
#pragma unroll
                        for (int h = 0; h < 4; ++h) e_n[c][h] = rd_e(blk + 1, c, h);
                }
                // the DMA instructions of sub-chunk s+RING-1 are spread over the BLKS blocks
#pragma unroll
                for (int t = 0; t < ND; ++t)
                    if ((t * BLKS) / ND == blk) issue_a1(s + RING - 1, nslot, t);
                __builtin_amdgcn_sched_barrier(0);      // keep the prefetch reads above the MFMAs
                float a[TILES][8];
                if constexpr (X16) {
                    h16x8 lo[TILES], hi[TILES];
#pragma unroll
                    for (int tl = 0; tl < TILES; ++tl) bytes_f16(raw_c[tl], lo[tl], hi[tl]);
#pragma unroll
                    for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
                        for (int g = 0; g < NG; ++g) mfma_x16(tl, g, lo[tl], hi[tl], b_c[g][0], b_c[g][1]);
                } else {
#pragma unroll
                for (int tl = 0; tl < TILES; ++tl) TR::cvt(raw_c[tl], a[tl]);
                if (ABL == 1) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
                            for (int g = 0; g < NG; ++g)
                                acc[tl][g][j & (NACC - 1)][j & 3] += a[tl][j] + b_c[g][j >> 2][j & 3];
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
                            for (int g = 0; g < NG; ++g)
                                acc[tl][g][j & (NACC - 1)] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                    a[tl][j], b_c[g][j >> 2][j & 3], acc[tl][g][j & (NACC - 1)], 0, 0, 0);
                    if constexpr (NG > 0) {
                        if (CVT) __builtin_amdgcn_sched_group_barrier(0x002, 8 * TILES, 0);  // conversions
                        __builtin_amdgcn_sched_group_barrier(0x008, 8 * NG * TILES, 0);      // then the MFMAs
                    }
                }
                }
                // VALU columns, two at a time (v_pk_fma_f32: the pixel value is broadcast, the two
                // columns' mask values sit next to each other in the slot)
#pragma unroll
                for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
                    for (int c2 = 0; c2 < NE / 2; ++c2)
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const f32x4 ev = e_c[c2][j >> 1];
                            const f32x2 e2 = (j & 1) ? f32x2{ev[2], ev[3]} : f32x2{ev[0], ev[1]};
                            acc_e[tl][c2] = __builtin_elementwise_fma(f32x2{a[tl][j], a[tl][j]}, e2,
                                                                      acc_e[tl][c2]);
                        }
#pragma unroll
                for (int tl = 0; tl < TILES; ++tl) raw_c[tl] = raw_n[tl];
#pragma unroll
                for (int g = 0; g < NG; ++g) { b_c[g][0] = b_n[g][0]; b_c[g][1] = b_n[g][1]; }
#pragma unroll
                for (int c = 0; c < NE / 2; ++c)
#pragma unroll
                    for (int h = 0; h < 4; ++h) e_c[c][h] = e_n[c][h];
            }
#pragma unroll
            for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
                for (int c = 0; c < NE / 2; ++c) {
                    acc_e2[tl][c] += acc_e[tl][c];
                    acc_e[tl][c] = f32x2{0.f, 0.f};
                }
        };

        // unroll period: ring slot (i % RING) and mask-slot parity ((i / PER) & 1) both static
        constexpr int U2 = 2 * PER;
        constexpr int UNROLL = (RING % U2 == 0) ? RING : ((U2 % RING == 0) ? U2 : RING * U2 /
                               ((RING % 2 == 0 && U2 % 2 == 0) ? 2 : 1));
        static_assert(UNROLL % RING == 0 && UNROLL % U2 == 0, "unroll period");
        int s = S0;
        if constexpr (UNROLL <= 12) {
            // (the second-level flush sits between the unrolled blocks, not inside them)
            constexpr int FLUSH_BLOCKS = FLUSH_PX / (UNROLL * SPX) > 1 ? FLUSH_PX / (UNROLL * SPX) : 1;
            int blocks_done = 0;
            for (; s + UNROLL <= S1; s += UNROLL) {
                if (NG > 0 && blocks_done == FLUSH_BLOCKS) {
                    blocks_done = 0;
#pragma unroll
                    for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
                        for (int g = 0; g < NG; ++g)
#pragma unroll
                            for (int x = 0; x < NACC; ++x) {
                                // (X16: [1] holds the high bytes' products)
                                if (X16 && x == 1) acc2[tl][g] += acc[tl][g][x] * 256.f;
                                else acc2[tl][g] += acc[tl][g][x];
                                acc[tl][g][x] = f32x4{0.f, 0.f, 0.f, 0.f};
                            }
                }
                ++blocks_done;
                static_for<0, UNROLL>([&](auto I) { iteration(s + decltype(I)::value, I); });
            }
        }
        for (; s < S1; ++s) iteration(s, std::integral_constant<int, -1>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // drain the clamped prefetches
    };
    if (nk > 0) {
        if constexpr (IND == 0 && ABL == 0) {
            if (kstr) main_loop(std::true_type{});
            else main_loop(std::false_type{});
        } else {
            main_loop(std::false_type{});
        }
    }

    // ragged last slot (n_px % KB != 0): guarded element loads, mask slot staged by plain copies
    if (has_ragged) {
        const int k = n_full;
        __syncthreads();
        float *bl = (float *)b_base;
        const u32x4 *img_units = (const u32x4 *)(img_t + (size_t)k * SLOT_FLOATS);
#pragma unroll
        for (int i = 0; i < SLOT_FLOATS / 4 / NT; ++i)
            ((u32x4 *)bl)[i * NT + tid] = img_units[i * NT + tid];
        __syncthreads();
        const float *ldsb = bl + b_lane;
#pragma unroll
        for (int tl = 0; tl < TILES; ++tl) {
            const int64_t f = src_frame_of(tl * 16 + m);
            const T *rowp = tile + f * ld + kg * 8;
#pragma unroll
            for (int blk = 0; blk < KB / 32; ++blk) {
                const int64_t p0 = (int64_t)k * KB + blk * 32;
                if constexpr (X16) {
                    typename TR::raw_t raw;
#pragma unroll
                    for (int i = 0; i < (int)(8 * sizeof(T) / 4); ++i) {
                        unsigned w = 0;
#pragma unroll
                        for (int e = 0; e < (int)(4 / sizeof(T)); ++e) {
                            const int j = i * (int)(4 / sizeof(T)) + e;
                            const unsigned v = (p0 + kg * 8 + j < n_px)
                                                   ? ((unsigned)rowp[p0 + j] & (sizeof(T) == 2 ? 0xffffu : 0xffu))
                                                   : 0u;
                            w |= v << (8 * (int)sizeof(T) * e);
                        }
                        raw[i] = w;
                    }
                    h16x8 lo, hi;
                    bytes_f16(raw, lo, hi);
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        mfma_x16(tl, g, lo, hi, *(const f32x4 *)(ldsb + g * (GROUP * KB) + b_unit(blk, 0)),
                                 *(const f32x4 *)(ldsb + g * (GROUP * KB) + b_unit(blk, 1)));
                } else {
                float a[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    a[j] = (p0 + kg * 8 + j < n_px) ? (float)rowp[p0 + j] : 0.f;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    f32x4 b[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        b[h] = *(const f32x4 *)(ldsb + g * (GROUP * KB) + b_unit(blk, h));
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        acc[tl][g][j & (NACC - 1)] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                            a[j], b[j >> 2][j & 3], acc[tl][g][j & (NACC - 1)], 0, 0, 0);
                }
#pragma unroll
                for (int c = 0; c < NE; ++c)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        acc_e[tl][c / 2][c & 1] += a[j] * bl[CFG::EXTRA_OFF + (c / 2) * (2 * KB) +
                                                            (blk * 32 + kg * 8 + j) * 2 + (c & 1)];
                }
            }
        }
    }

    // The few weights the float16 image leaves out (at most DENSE_TAIL_MAX per stack; k_build_image_h16): one
    // float32 product per stored entry, like the reference's matrix product (udf/masks.py:59-77), added to
    // the column's sum.  A lane holds column m of every group and 4 * TILES frames: the pixels of an entry
    // that falls on one of its columns are fetched for all of its frames at once (independent loads).
    float tail_add[TILES][X16 ? NG : 1][4];
    if constexpr (X16) {
#pragma unroll
        for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) tail_add[tl][g][r] = 0.f;
        if (n_tail > 0 && ks == 0) {
            for (int e = 0; e < n_tail; ++e) {
                const int c = tail_col[e] - gt * NG * GROUP;
                if (c < 0 || c >= NG * GROUP || (c & (GROUP - 1)) != m) continue;
                const float w = tail_val[e];
                const int64_t px = tail_px[e];
                float x[TILES][4];
#pragma unroll
                for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        x[tl][r] = (float)tile[src_frame_of(tl * 16 + kg * 4 + r) * ld + px];
#pragma unroll
                for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
                    for (int g = 0; g < NG; ++g)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (g == c / GROUP) tail_add[tl][g][r] += w * x[tl][r];
            }
        }
    }
#pragma unroll
    for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t f = frame_of(tl * 16 + kg * 4 + r);
                const int col = (gt * NG + g) * GROUP + m;
                if (f >= 0 && col < n_cols) {
                    float v = acc[tl][g][0][r];
                    if (X16) v += 256.f * acc[tl][g][NACC - 1][r];
                    else if (NACC == 2) v += acc[tl][g][NACC - 1][r];
                    v += acc2[tl][g][r];
                    if (X16) v *= inv_scale[col];             // undo the column's power-of-two scale
                    if constexpr (X16) v += tail_add[tl][g][r];
                    if (ksplit == 1) {
                        float *p = out + f * ld_out + col;
                        *p = accumulate ? (*p + v) : v;
                    } else {
                        partials[((int64_t)ks * n_frames + f) * n_cols + col] = v;
                    }
                }
            }
    if constexpr (NE > 0) {
        // lane (m, kg) holds frame m's partial over its own pixels: sum the 4 kg lanes
#pragma unroll
        for (int tl = 0; tl < TILES; ++tl) {
            const int64_t f = frame_of(tl * 16 + m);
#pragma unroll
            for (int c = 0; c < NE; ++c) {
                float v = acc_e2[tl][c / 2][c & 1] + acc_e[tl][c / 2][c & 1];   // (+ the ragged tail)
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                const int col = NG * GROUP + c;
                if (kg == 0 && f >= 0 && col < n_cols) {
                    if (ksplit == 1) {
                        float *p = out + f * ld_out + col;
                        *p = accumulate ? (*p + v) : v;
                    } else {
                        partials[((int64_t)ks * n_frames + f) * n_cols + col] = v;
                    }
                }
            }
        }
    }
    // ---- pixel axis split over `ksplit` workgroups, optional (kcount != nullptr): the LAST of them to
    // finish a block of frames adds the partial sums up -- in the fixed order k = 0 .. ksplit - 1,
    // whoever is last, so the result does not depend on the arrival order -- instead of a second
    // launch.  Measured slower than the second launch (see partial_counters) and off by default.
    if (IND != 1 && kcount != nullptr && ksplit > 1) {
        volatile int *last_flag = (volatile int *)lds_raw;  // (the LDS is full; its buffers are dead now)
        __threadfence();                                    // this workgroup's partials are visible
        __syncthreads();
        int *cnt = kcount + (int64_t)blockIdx.z * gridDim.x + blockIdx.x;
        if (tid == 0) *last_flag = atomicAdd(cnt, 1) == ksplit - 1;
        __syncthreads();
        if (*last_flag) {
            __threadfence();
            const int64_t fb0 = (int64_t)blockIdx.x * (WAVES * ROWS);
            const int nf = (int)min<int64_t>(WAVES * ROWS, n_frames - fb0);
            const int c0 = gt * NG * GROUP;
            const int nc = min(n_cols - c0, NG * GROUP + NE);
            // (device-scope loads: the other workgroups' partials come from the L2, never from a line
            // this CU's L1 kept from an earlier launch; 8 independent loads in flight per thread)
            const int64_t kstride = n_frames * n_cols;
            // 8 outputs x 8 splits = 64 loads in flight per thread and batch: the tail is a handful of
            // L2 round trips, not one per (output, 8 splits) -- it is what every workgroup of the frame
            // block waits for.  Each output still adds its partials in the order k = 0 .. ksplit - 1.
            constexpr int OB = 8, KBATCH = 8;
            const int n_out = nf * nc;
            for (int o0 = tid; o0 < n_out; o0 += NT * OB) {
                float acc_s[OB];
                const float *src[OB];
                float *dstp[OB];
#pragma unroll
                for (int j = 0; j < OB; ++j) {
                    const int idx = min(o0 + j * NT, n_out - 1);            // (clamped: discarded below)
                    const int64_t f = fb0 + idx / nc;
                    const int col = c0 + idx % nc;
                    dstp[j] = out + f * ld_out + col;
                    src[j] = partials + f * n_cols + col;
                    acc_s[j] = accumulate ? *dstp[j] : 0.f;
                }
                for (int k0 = 0; k0 < ksplit; k0 += KBATCH) {
                    float v[OB][KBATCH];
#pragma unroll
                    for (int j = 0; j < OB; ++j)
#pragma unroll
                        for (int u = 0; u < KBATCH; ++u)
                            v[j][u] = __hip_atomic_load(src[j] + (int64_t)min(k0 + u, ksplit - 1) * kstride,
                                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int j = 0; j < OB; ++j)
#pragma unroll
                        for (int u = 0; u < KBATCH; ++u)
                            if (k0 + u < ksplit) acc_s[j] += v[j][u];
                }
#pragma unroll
                for (int j = 0; j < OB; ++j)
                    if (o0 + j * NT < n_out) *dstp[j] = acc_s[j];
            }
            if (tid == 0) *cnt = 0;                         // ready for the next launch
        }
    }
}

__global__ void k_reduce_partials(const float *__restrict__ partials, int ksplit,
                                  int64_t n_frames, int n_cols, float *__restrict__ out,
                                  int64_t ld_out, int accumulate) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_frames * n_cols) return;
    const int64_t f = idx / n_cols;
    const int col = (int)(idx % n_cols);
    float *p = out + f * ld_out + col;
    float s = accumulate ? *p : 0.f;
    // eight independent loads in flight, added in the order k = 0 .. ksplit - 1 (a loop of dependent
    // load + add pairs made this kernel 10 us of a 45-us launch of 1 024 frames: 32 L2 latencies in a row)
    const int64_t stride = n_frames * n_cols;
    for (int k0 = 0; k0 < ksplit; k0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partials[(int64_t)min(k0 + u, ksplit - 1) * stride + idx];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + u < ksplit) s += v[u];
    }
    *p = s;
}

// ---- generic path ----------------------------------------------------------------------------
// Acc types: float, double, cfloat, cdouble, uint64_t (wrap-around integer arithmetic)
template <typename A> struct AccOps;
template <> struct AccOps<float> {
    static __device__ __forceinline__ float zero() { return 0.f; }
    static __device__ __forceinline__ void fma(float &acc, float x, float m) { acc += x * m; }
    static __device__ __forceinline__ float add(float a, float b) { return a + b; }
};
template <> struct AccOps<double> {
    static __device__ __forceinline__ double zero() { return 0.; }
    static __device__ __forceinline__ void fma(double &acc, double x, double m) { acc += x * m; }
    static __device__ __forceinline__ double add(double a, double b) { return a + b; }
};
template <> struct AccOps<uint64_t> {
    static __device__ __forceinline__ uint64_t zero() { return 0; }
    static __device__ __forceinline__ void fma(uint64_t &acc, uint64_t x, uint64_t m) {
        acc += x * m;
    }
    static __device__ __forceinline__ uint64_t add(uint64_t a, uint64_t b) { return a + b; }
};
template <> struct AccOps<cfloat> {
    static __device__ __forceinline__ cfloat zero() { return cfloat{0.f, 0.f}; }
    static __device__ __forceinline__ void fma(cfloat &acc, cfloat x, cfloat m) {
        acc.re += x.re * m.re - x.im * m.im;
        acc.im += x.re * m.im + x.im * m.re;
    }
    static __device__ __forceinline__ cfloat add(cfloat a, cfloat b) {
        return cfloat{a.re + b.re, a.im + b.im};
    }
};
template <> struct AccOps<cdouble> {
    static __device__ __forceinline__ cdouble zero() { return cdouble{0., 0.}; }
    static __device__ __forceinline__ void fma(cdouble &acc, cdouble x, cdouble m) {
        acc.re += x.re * m.re - x.im * m.im;
        acc.im += x.re * m.im + x.im * m.re;
    }
    static __device__ __forceinline__ cdouble add(cdouble a, cdouble b) {
        return cdouble{a.re + b.re, a.im + b.im};
    }
};

template <typename A, typename T> struct Conv {
    static __device__ __forceinline__ A from(T v) { return (A)v; }
};
template <typename T> struct Conv<uint64_t, T> {   // integer: sign-extend like numpy astype
    static __device__ __forceinline__ uint64_t from(T v) { return (uint64_t)(int64_t)v; }
};
template <> struct Conv<uint64_t, uint64_t> {
    static __device__ __forceinline__ uint64_t from(uint64_t v) { return v; }
};
template <typename T> struct Conv<cfloat, T> {
    static __device__ __forceinline__ cfloat from(T v) { return cfloat{(float)v, 0.f}; }
};
template <typename T> struct Conv<cdouble, T> {
    static __device__ __forceinline__ cdouble from(T v) { return cdouble{(double)v, 0.}; }
};
template <> struct Conv<cfloat, cfloat> {
    static __device__ __forceinline__ cfloat from(cfloat v) { return v; }
};
template <> struct Conv<cdouble, cfloat> {
    static __device__ __forceinline__ cdouble from(cfloat v) { return cdouble{v.re, v.im}; }
};
template <> struct Conv<cdouble, cdouble> {
    static __device__ __forceinline__ cdouble from(cdouble v) { return v; }
};
template <> struct Conv<cfloat, cdouble> {
    static __device__ __forceinline__ cfloat from(cdouble v) {
        return cfloat{(float)v.re, (float)v.im};
    }
};

template <typename S, typename A> struct Store {
    static __device__ __forceinline__ void put(S *p, A v, bool accumulate) {
        *p = accumulate ? (S)(*p + (S)v) : (S)v;
    }
};
template <> struct Store<cfloat, cfloat> {
    static __device__ __forceinline__ void put(cfloat *p, cfloat v, bool accumulate) {
        if (accumulate) { v.re += p->re; v.im += p->im; }
        *p = v;
    }
};
template <> struct Store<cdouble, cdouble> {
    static __device__ __forceinline__ void put(cdouble *p, cdouble v, bool accumulate) {
        if (accumulate) { v.re += p->re; v.im += p->im; }
        *p = v;
    }
};

constexpr int GEN_MASKS = 4;   // masks per block of the generic kernel

// block (256 threads) per (frame, group of 4 masks)
template <typename TIn, typename A, typename S>
__global__ void __launch_bounds__(256)
k_dense_generic(const TIn *__restrict__ tile, int64_t ld, int64_t n_frames, int64_t n_px,
                const A *__restrict__ masks, int n_masks, S *__restrict__ out, int64_t ld_out,
                int accumulate) {
    __shared__ A red[GEN_MASKS][256];
    const int64_t f = blockIdx.x;
    const int k0 = blockIdx.y * GEN_MASKS;
    const int nk = min(GEN_MASKS, n_masks - k0);
    A acc[GEN_MASKS];
#pragma unroll
    for (int k = 0; k < GEN_MASKS; ++k) acc[k] = AccOps<A>::zero();
    const TIn *row = tile + f * ld;
    for (int64_t p = threadIdx.x; p < n_px; p += 256) {
        const A x = Conv<A, TIn>::from(row[p]);
#pragma unroll
        for (int k = 0; k < GEN_MASKS; ++k)
            if (k < nk) AccOps<A>::fma(acc[k], x, masks[(int64_t)(k0 + k) * n_px + p]);
    }
#pragma unroll
    for (int k = 0; k < GEN_MASKS; ++k) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
#pragma unroll
            for (int k = 0; k < GEN_MASKS; ++k)
                red[k][threadIdx.x] = AccOps<A>::add(red[k][threadIdx.x], red[k][threadIdx.x + s]);
        }
        __syncthreads();
    }
    if ((int)threadIdx.x < nk)
        Store<S, A>::put(out + f * ld_out + k0 + threadIdx.x, red[threadIdx.x][0], accumulate != 0);
}

// shifted masks: block (256 threads) per (frame, group of 4 masks); threads walk the overlap region
template <typename TIn, typename A, typename S>
__global__ void __launch_bounds__(256)
k_dense_shifted(const TIn *__restrict__ tile, int64_t ld, int sig_h, int sig_w,
                const int32_t *__restrict__ shifts, const A *__restrict__ masks, int n_masks,
                S *__restrict__ out, int64_t ld_out, int accumulate) {
    __shared__ A red[GEN_MASKS][256];
    const int64_t f = blockIdx.x;
    const int k0 = blockIdx.y * GEN_MASKS;
    const int nk = min(GEN_MASKS, n_masks - k0);
    const int dy = shifts[2 * f], dx = shifts[2 * f + 1];
    // frame rows [y0, y1) x cols [x0, x1) overlap the mask shifted by (dy, dx)
    const int y0 = max(0, dy), y1 = min(sig_h, sig_h + dy);
    const int x0 = max(0, dx), x1 = min(sig_w, sig_w + dx);
    const int ow = max(0, x1 - x0), oh = max(0, y1 - y0);
    const int64_t n_px = (int64_t)sig_h * sig_w;
    A acc[GEN_MASKS];
#pragma unroll
    for (int k = 0; k < GEN_MASKS; ++k) acc[k] = AccOps<A>::zero();
    const TIn *row = tile + f * ld;
    for (int64_t i = threadIdx.x; i < (int64_t)ow * oh; i += 256) {
        const int y = y0 + (int)(i / ow), x = x0 + (int)(i % ow);
        const A v = Conv<A, TIn>::from(row[(int64_t)y * sig_w + x]);
        const int64_t mp = (int64_t)(y - dy) * sig_w + (x - dx);
#pragma unroll
        for (int k = 0; k < GEN_MASKS; ++k)
            if (k < nk) AccOps<A>::fma(acc[k], v, masks[(int64_t)(k0 + k) * n_px + mp]);
    }
#pragma unroll
    for (int k = 0; k < GEN_MASKS; ++k) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
#pragma unroll
            for (int k = 0; k < GEN_MASKS; ++k)
                red[k][threadIdx.x] = AccOps<A>::add(red[k][threadIdx.x], red[k][threadIdx.x + s]);
        }
        __syncthreads();
    }
    if ((int)threadIdx.x < nk)
        Store<S, A>::put(out + f * ld_out + k0 + threadIdx.x, red[threadIdx.x][0], accumulate != 0);
}

}  // namespace ltmi

// =================================================================================================
// host side
// =================================================================================================
using namespace ltmi;

namespace ltmi {
void guard_note_unchecked(ltmi_masks *m);   // ltmi_guard.hip
int csr_destroy(ltmi_masks *m);   // ltmi_sparse.hip
int csr_set_sig_shape(ltmi_masks *m, int sig_h, int sig_w);
bool csr_has_band(const ltmi_masks *m);
bool csr_rows_ok(const ltmi_masks *m, const void *tile, int tile_dtype, int64_t ld_tile);
int csr_apply(ltmi_masks *m, const void *tile, int tile_dtype, int64_t n_frames, int64_t ld_tile,
              void *out, int64_t ld_out, int accumulate, hipStream_t stream);
}

static bool mfma_tile_dtype(int dt) {
    return dt == LTMI_U8 || dt == LTMI_I8 || dt == LTMI_U16 || dt == LTMI_I16 ||
           dt == LTMI_F32 || dt == LTMI_BOOL;
}

extern "C" int ltmi_masks_create_dense(int device, const void *masks_host, int result_dtype,
                                       int64_t n_masks, int64_t n_px, ltmi_masks **out) {
    if (!masks_host || !out || n_masks <= 0 || n_px <= 0)
        LTMI_FAIL(LTMI_E_INVALID, "ltmi_masks_create_dense: bad arguments (n_masks=%lld n_px=%lld)",
                  (long long)n_masks, (long long)n_px);
    if (dtype_size(result_dtype) == 0)
        LTMI_FAIL(LTMI_E_DTYPE, "ltmi_masks_create_dense: unknown dtype %d", result_dtype);
    LTMI_HIP(hipSetDevice(device));
    ltmi_masks *m = new (std::nothrow) ltmi_masks();
    if (!m) LTMI_FAIL(LTMI_E_NOMEM, "out of host memory");
    m->device = device;
    m->result_dtype = result_dtype;
    m->n_masks = n_masks;
    m->n_px = n_px;

    m->kind = (result_dtype == LTMI_F32 || result_dtype == LTMI_C64) ? 0 : 1;
    if (m->kind == 0) {
        // ---- MFMA image ----
        const int cpm = (result_dtype == LTMI_C64) ? 2 : 1;      // real columns per mask
        m->n_cols = (int)(n_masks * cpm);
        int groups = (m->n_cols + GROUP - 1) / GROUP;
        m->ng = groups == 1 ? 1 : (groups == 2 ? 2 : 4);
        m->n_groups = (groups + m->ng - 1) / m->ng * m->ng;
        m->n_chunks = (int)((n_px + KC - 1) / KC);
    }
    // the generic image is always kept too: it serves tile dtypes the MFMA path does not take
    {
        // accumulate type: f32 f64 c64 c128, integers -> 64-bit
        const bool is_int = result_dtype <= LTMI_I64;
        const size_t n = (size_t)n_masks * n_px;
        const size_t acc_size = is_int ? 8 : (size_t)dtype_size(result_dtype);
        std::vector<unsigned char> buf;
        const void *upload = masks_host;
        if (is_int && dtype_size(result_dtype) != 8) {
            try { buf.resize(n * 8); } catch (...) {
                ltmi_masks_destroy(m);
                LTMI_FAIL(LTMI_E_NOMEM, "out of host memory");
            }
            int64_t *d = (int64_t *)buf.data();
            for (size_t i = 0; i < n; ++i) {
                switch (result_dtype) {
                    case LTMI_BOOL: case LTMI_U8: d[i] = ((const uint8_t *)masks_host)[i]; break;
                    case LTMI_I8: d[i] = ((const int8_t *)masks_host)[i]; break;
                    case LTMI_U16: d[i] = ((const uint16_t *)masks_host)[i]; break;
                    case LTMI_I16: d[i] = ((const int16_t *)masks_host)[i]; break;
                    case LTMI_U32: d[i] = ((const uint32_t *)masks_host)[i]; break;
                    case LTMI_I32: d[i] = ((const int32_t *)masks_host)[i]; break;
                }
            }
            upload = buf.data();
        }
        if (is_int) {
            // largest |mask value|: decides whether sums stay exactly representable in f64
            uint64_t absmax = 0;
            const int64_t *d = (const int64_t *)upload;
            const bool is_u64 = result_dtype == LTMI_U64;
            for (size_t i = 0; i < n; ++i) {
                const uint64_t a = is_u64 ? (uint64_t)d[i]
                                          : (uint64_t)(d[i] < 0 ? -(d[i] + 1) + 1ull : d[i]);
                absmax = std::max(absmax, a);
            }
            int bits = 0;
            while (bits < 64 && (absmax >> bits) != 0) ++bits;
            m->mask_bits = bits;
        }
        hipError_t e = hipMalloc(&m->gmasks, n * acc_size);
        if (e == hipSuccess) e = hipMemcpy(m->gmasks, upload, n * acc_size, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            ltmi_masks_destroy(m);
            LTMI_FAIL((int)e, "uploading the mask stack failed: %s", hipGetErrorString(e));
        }
    }
    if (result_dtype == LTMI_F64 || result_dtype == LTMI_C128 ||
        (result_dtype >= LTMI_U8 && result_dtype <= LTMI_I64 && m->mask_bits <= 20)) {
        const int rc64 = ltmi::dense64_create(m);
        if (rc64 != LTMI_OK) {
            ltmi_masks_destroy(m);
            return rc64;
        }
    }
    if (m->kind == 0) {
        // MFMA image: swizzled on the device from the raw stack uploaded above (f32, or
        // interleaved (re, im) pairs) -- no host-side transform, no second upload
        const int cpm = (result_dtype == LTMI_C64) ? 2 : 1;
        const size_t n_float = (size_t)m->n_groups * m->n_chunks * CHUNK_FLOATS;
        hipError_t e = hipMalloc((void **)&m->img, n_float * sizeof(float));
        if (e == hipSuccess) e = hipMemset(m->img, 0, n_float * sizeof(float));
        if (e == hipSuccess) {
            const int64_t total = n_masks * cpm * n_px;
            const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 65535 * 16);
            hipLaunchKernelGGL(ltmi::k_build_image, dim3(blocks), dim3(256), 0, 0,
                               (const float *)m->gmasks, m->img, n_masks, cpm, n_px, m->n_chunks);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipDeviceSynchronize();
        }
        if (e == hipSuccess && m->ng > 1) {
            // slot-major image for the LDS-DMA kernel with several column groups (k_dense_lds)
            constexpr int kb = 128;
            m->n_slots2 = (int)((n_px + kb - 1) / kb);
            const size_t n2 = (size_t)m->n_groups * m->n_slots2 * GROUP * kb;
            e = hipMalloc((void **)&m->img2, n2 * sizeof(float));
            if (e == hipSuccess) e = hipMemset(m->img2, 0, n2 * sizeof(float));
            if (e == hipSuccess) {
                const int64_t total = n_masks * cpm * n_px;
                const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 65535 * 16);
                hipLaunchKernelGGL(ltmi::k_build_image2, dim3(blocks), dim3(256), 0, 0,
                                   (const float *)m->gmasks, m->img2, n_masks, cpm, n_px,
                                   m->n_slots2, m->ng, kb);
                e = hipGetLastError();
                if (e == hipSuccess) e = hipDeviceSynchronize();
            }
        }
        // Column counts the 1 / 2 / 4-group tiles serve badly get their own slot-major image
        // ("image 3"): ng3 MFMA groups + ne3 columns on the VALU instead of a group that is mostly
        // padding (k_dense_lds<T, NG, .., NE>; measured in scripts/bench_extras.py):
        //   17..18 -> 1 + 2    33..34 -> 2 + 2    35..36 -> 2 + 4    37..48 -> 3 (no padded 4th group)
        //   49..50 -> 3 + 2    51..52 -> 3 + 4    (19..20: the padded 2-group kernel is faster)
        //   1..2 -> 0 + 2      3..4 -> 0 + 4      (no matrix cores at all)
        int ng3 = 0, ne3 = 0;
        {
            const int nc = m->n_cols;
            if (nc <= 4) { ng3 = 0; ne3 = nc <= 2 ? 2 : 4; }        // VALU only
            else if (nc >= 17 && nc <= 18) { ng3 = 1; ne3 = 2; }
            else if (nc >= 33 && nc <= 36) { ng3 = 2; ne3 = nc <= 34 ? 2 : 4; }
            else if (nc >= 37 && nc <= 48) { ng3 = 3; ne3 = 0; }
            else if (nc >= 49 && nc <= 52) { ng3 = 3; ne3 = nc <= 50 ? 2 : 4; }
        }
        if (e == hipSuccess && (ng3 > 0 || ne3 > 0)) {
            constexpr int kb = 128;
            m->ng3 = ng3;
            m->n_slots3 = (int)((n_px + kb - 1) / kb);
            m->ne3 = ne3;
            const int slot_floats = (m->ne3 > 0 ? ltmi::ext_slot_bytes(ng3, m->ne3)
                                                : ng3 * GROUP * kb * 4) / 4;
            const size_t n3 = (size_t)m->n_slots3 * slot_floats;
            e = hipMalloc((void **)&m->img3, n3 * sizeof(float));
            if (e == hipSuccess) e = hipMemset(m->img3, 0, n3 * sizeof(float));
            if (e == hipSuccess) {
                const int64_t total = n_masks * cpm * n_px;
                const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 65535 * 16);
                hipLaunchKernelGGL(ltmi::k_build_image3, dim3(blocks), dim3(256), 0, 0,
                                   (const float *)m->gmasks, m->img3, n_masks, cpm, n_px,
                                   m->n_slots3, ng3, slot_floats);
                e = hipGetLastError();
                if (e == hipSuccess) e = hipDeviceSynchronize();
            }
        }
        // float16 images for 1- / 2-byte integer pixels (k_dense_lds X16) of every matrix-core layout
        // this stack has without VALU columns (one group: also for <= 4 columns): finite weights only.  LTMI_DENSE_F16=0: never.
        const bool want_h1 = m->ng == 1 && m->n_groups == 1;
        const bool want_h2 = m->ng > 1 && m->img2 != nullptr;
        const bool want_h3 = m->img3 != nullptr && m->ng3 == 3 && m->ne3 == 0;
        if (e == hipSuccess && (want_h1 || want_h2 || want_h3)) {
            const char *off = getenv("LTMI_DENSE_F16");
            const float *hm = (const float *)masks_host;      // (n_masks, n_px, cpm) float32 on the host
            std::vector<float> amax((size_t)m->n_cols, 0.f);
            bool finite = !(off && atoi(off) == 0);
            for (int64_t k = 0; k < n_masks && finite; ++k)
                for (int64_t p = 0; p < n_px && finite; ++p)
                    for (int c2 = 0; c2 < cpm; ++c2) {
                        const float v = hm[(k * n_px + p) * cpm + c2];
                        if (!std::isfinite(v)) { finite = false; break; }
                        float &mx = amax[(size_t)(k * cpm + c2)];
                        mx = std::max(mx, std::fabs(v));
                    }
            // (columns whose largest weight lies outside 2^-100 .. 2^100 cannot be brought into float16 by
            // a float32 power of two with room to spare: such stacks keep the float32 instruction)
            for (int k = 0; k < m->n_cols && finite; ++k)
                if (amax[(size_t)k] > 0.f) {
                    int ex;
                    (void)std::frexp(amax[(size_t)k], &ex);
                    if (ex < -100 || ex > 100) finite = false;
                }
            // Two float16 pieces of the scaled weight carry 22 bits down to 2^-19 of the column's largest
            // weight and an absolute 2^-39 max|w| below that (the float16 subnormal grid).  A weight whose two
            // pieces miss it by more than 2^-19 relative (only possible below 2^-20 of the maximum; weights
            // with few significant bits -- 0/1 masks, k 2^-24 random numbers -- are exact at any size) is
            // left out of the float16 images and added in float32 by the epilogue of the matrix kernel.  A stack with more
            // than DENSE_TAIL_MAX of them (smooth masks with long tails: Gaussians) keeps the float32
            // instruction altogether.
            std::vector<float> scale((size_t)m->n_cols, 1.f), inv((size_t)m->n_cols, 1.f);
            std::vector<int32_t> tail_px, tail_col;
            std::vector<float> tail_val;
            if (finite) {
                for (int k = 0; k < m->n_cols; ++k)
                    if (amax[(size_t)k] > 0.f) {
                        int ex;
                        (void)std::frexp(amax[(size_t)k], &ex);           // amax = f 2^ex, f in [0.5, 1)
                        // amax 2^sh in [16384, 32768): the largest scale whose w1 stays a finite float16.
                        // (k_bell_flat keeps 256 w S finite and scales to [64, 128); here the high bytes
                        // have their own accumulator.)
                        const int sh = std::max(-120, std::min(120, 15 - ex));
                        scale[(size_t)k] = std::ldexp(1.0f, sh);
                        inv[(size_t)k] = std::ldexp(1.0f, -sh);
                    }
                for (int64_t k = 0; k < n_masks && finite; ++k)
                    for (int64_t p = 0; p < n_px && finite; ++p)
                        for (int c2 = 0; c2 < cpm; ++c2) {
                            const float v = hm[(k * n_px + p) * cpm + c2];
                            const int col = (int)(k * cpm + c2);
                            if (v == 0.f || std::fabs(v) >= std::ldexp(amax[(size_t)col], -20)) continue;
                            if (ltmi::h16_pieces_exact_enough(v * scale[(size_t)col])) continue;
                            if ((int)tail_val.size() == ltmi::DENSE_TAIL_MAX) { finite = false; break; }
                            tail_px.push_back((int32_t)p);
                            tail_col.push_back(col);
                            tail_val.push_back(v);
                        }
            }
            if (finite) {
                float *scale_dev = nullptr, *amax_dev = nullptr;
                e = hipMalloc((void **)&m->inv_scale, inv.size() * sizeof(float));
                if (e == hipSuccess) e = hipMalloc((void **)&scale_dev, scale.size() * sizeof(float));
                if (e == hipSuccess) e = hipMalloc((void **)&amax_dev, amax.size() * sizeof(float));
                if (e == hipSuccess) e = hipMemcpy(amax_dev, amax.data(), amax.size() * sizeof(float), hipMemcpyHostToDevice);
                if (e == hipSuccess && !tail_val.empty()) {
                    const size_t nt = tail_val.size();
                    m->tail_n = (int)nt;
                    e = hipMalloc((void **)&m->tail_px, nt * sizeof(int32_t));
                    if (e == hipSuccess) e = hipMalloc((void **)&m->tail_col, nt * sizeof(int32_t));
                    if (e == hipSuccess) e = hipMalloc((void **)&m->tail_val, nt * sizeof(float));
                    if (e == hipSuccess) e = hipMemcpy(m->tail_px, tail_px.data(), nt * sizeof(int32_t), hipMemcpyHostToDevice);
                    if (e == hipSuccess) e = hipMemcpy(m->tail_col, tail_col.data(), nt * sizeof(int32_t), hipMemcpyHostToDevice);
                    if (e == hipSuccess) e = hipMemcpy(m->tail_val, tail_val.data(), nt * sizeof(float), hipMemcpyHostToDevice);
                }
                if (e == hipSuccess) e = hipMemcpy(m->inv_scale, inv.data(), inv.size() * sizeof(float), hipMemcpyHostToDevice);
                if (e == hipSuccess) e = hipMemcpy(scale_dev, scale.data(), scale.size() * sizeof(float), hipMemcpyHostToDevice);
                const int64_t total = n_masks * cpm * n_px;
                const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 65535 * 16);
                auto build = [&](float **dst, size_t n_floats, int n_slots, int ng, int layout,
                                 int slot_floats) {
                    if (e != hipSuccess) return;
                    e = hipMalloc((void **)dst, n_floats * sizeof(float));
                    if (e == hipSuccess) e = hipMemset(*dst, 0, n_floats * sizeof(float));
                    if (e != hipSuccess) return;
                    hipLaunchKernelGGL(ltmi::k_build_image_h16, dim3(blocks), dim3(256), 0, 0,
                                       (const float *)m->gmasks, (_Float16 *)*dst, n_masks, cpm, n_px,
                                       n_slots, ng, layout, slot_floats, (const float *)scale_dev,
                                       (const float *)amax_dev);
                    e = hipGetLastError();
                };
                if (want_h1) build(&m->img_h, n_float, m->n_chunks, 1, 1, 0);
                if (want_h2)
                    build(&m->img2_h, (size_t)m->n_groups * m->n_slots2 * GROUP * 128, m->n_slots2, m->ng,
                          2, 0);
                if (want_h3)
                    build(&m->img3_h, (size_t)m->n_slots3 * (3 * GROUP * 128), m->n_slots3, 3, 3,
                          3 * GROUP * 128);
                if (e == hipSuccess) e = hipDeviceSynchronize();
                if (scale_dev) (void)hipFree(scale_dev);
                if (amax_dev) (void)hipFree(amax_dev);
            }
        }
        if (e != hipSuccess) {
            ltmi_masks_destroy(m);
            LTMI_FAIL((int)e, "building the mask image failed: %s", hipGetErrorString(e));
        }
    }
    if (m->kind == 0 && m->n_cols > 4 * GROUP) {
        // wide stacks: 4-group tiles over ALL columns would pad the last tile to 64 columns; blocks
        // of 64 columns + a last block with the tile width that fits it compute less (70 columns:
        // 64 + 16 instead of 128)
        const int cpm = (result_dtype == LTMI_C64) ? 2 : 1;
        const int64_t per = 4 * GROUP / cpm;
        const size_t row_bytes = (size_t)n_px * dtype_size(result_dtype);
        for (int64_t k0 = 0; k0 < n_masks; k0 += per) {
            ltmi_masks *child = nullptr;
            const int rc = ltmi_masks_create_dense(
                device, (const unsigned char *)masks_host + (size_t)k0 * row_bytes, result_dtype,
                std::min<int64_t>(per, n_masks - k0), n_px, &child);
            if (rc != LTMI_OK) {
                ltmi_masks_destroy(m);
                return rc;
            }
            m->blocks.push_back(child);
            m->block_first.push_back(k0);
        }
    }
    *out = m;
    return LTMI_OK;
}

static void shift_cache_destroy(ltmi_masks *m);

extern "C" int ltmi_masks_destroy(ltmi_masks *m) {
    if (!m) return LTMI_OK;
    for (ltmi_masks *b : m->blocks) (void)ltmi_masks_destroy(b);
    m->blocks.clear();
    (void)hipSetDevice(m->device);
    fold_destroy(m);
    if (m->img) (void)hipFree(m->img);
    if (m->img2) (void)hipFree(m->img2);
    if (m->img3) (void)hipFree(m->img3);
    ltmi::split_destroy(m->split);
    m->split = nullptr;
    if (m->img_h) (void)hipFree(m->img_h);
    if (m->img2_h) (void)hipFree(m->img2_h);
    if (m->img3_h) (void)hipFree(m->img3_h);
    if (m->inv_scale) (void)hipFree(m->inv_scale);
    if (m->tail_px) (void)hipFree(m->tail_px);
    if (m->tail_col) (void)hipFree(m->tail_col);
    if (m->tail_val) (void)hipFree(m->tail_val);
    ltmi::dense64_destroy(m);
    shift_cache_destroy(m);
    if (m->partials) (void)hipFree(m->partials);
    if (m->gmasks) (void)hipFree(m->gmasks);
    if (m->csr) (void)ltmi::csr_destroy(m);
    ltmi::guard_destroy(m);
    delete m;
    return LTMI_OK;
}

extern "C" int ltmi_masks_set_sig_shape(ltmi_masks *m, int sig_h, int sig_w) {
    if (!m) LTMI_FAIL(LTMI_E_INVALID, "ltmi_masks_set_sig_shape: null handle");
    if (sig_h <= 0 || sig_w <= 0 || (int64_t)sig_h * sig_w != m->n_px)
        LTMI_FAIL(LTMI_E_SHAPE, "ltmi_masks_set_sig_shape: %d x %d is not the handle's %lld pixels", sig_h, sig_w,
                  (long long)m->n_px);
    LTMI_HIP(hipSetDevice(m->device));
    for (ltmi_masks *b : m->blocks) {
        const int rc = ltmi_masks_set_sig_shape(b, sig_h, sig_w);
        if (rc != LTMI_OK) return rc;
    }
    if (m->kind == 2) return ltmi::csr_set_sig_shape(m, sig_h, sig_w);
    if (m->kind != 0) return LTMI_OK;
    return fold_create(m, sig_h, sig_w);
}

extern "C" int ltmi_masks_kind(const ltmi_masks *m, int *kind) {
    if (!m || !kind) LTMI_FAIL(LTMI_E_INVALID, "ltmi_masks_kind: null argument");
    *kind = (m->kind == 2 && ltmi::csr_has_band(m)) ? 3 : m->kind;
    return LTMI_OK;
}

extern "C" int ltmi_masks_set_tuning(ltmi_masks *m, int mt, int waves, int ksplit) {
    if (!m) LTMI_FAIL(LTMI_E_INVALID, "ltmi_masks_set_tuning: null handle");
    if (mt == 0 && ((waves >= 30 && waves <= 38) || (waves >= 40 && waves <= 42))) {
        // k_dense_lds: 30 = as dispatched, 31 / 32 = timing-only ablations (no DMA / no MFMA),
        // 34 / 35 = one / two frame tiles per wave; 36 = k_dense_split (float32 frames, ltmi_split.hip);
        // 37 = the float32 matrix instruction also where the exact float16 products (X16) apply;
        // 38 = the unfolded kernels for a stack that has a row-mirror image (ltmi_fold.hip);
        // sparse stacks: 40 = as dispatched, 41 = SELL kernel even if a blocked / scatter image exists,
        // 42 = not the scatter kernel (blocked image or SELL)
        m->tune_mt = 0;
        m->tune_waves = 0;
        m->tune_ksplit = ksplit;
        m->tune_ksplit_ring = waves;
        return LTMI_OK;
    }
    if (!(mt == 0 || mt == 1 || mt == 2) || !(waves == 0 || waves == 4 || waves == 8) || ksplit < 0)
        LTMI_FAIL(LTMI_E_INVALID, "ltmi_masks_set_tuning: unsupported (mt=%d waves=%d ksplit=%d)",
                  mt, waves, ksplit);
    m->tune_mt = mt;
    m->tune_waves = waves;
    m->tune_ksplit = ksplit;
    m->tune_ksplit_ring = 0;
    return LTMI_OK;
}

extern "C" const char *ltmi_masks_last_kernel(const ltmi_masks *m) {
    return m ? m->last_kernel : "";
}

// ---- MFMA launch ---------------------------------------------------------------------------------
// the workspace of a handle: KCOUNT_BYTES of arrival counters (zero between launches: the last workgroup
// of a frame block resets its counter) followed by the partial sums
constexpr size_t KCOUNT_BYTES = 64 * 1024;
static int ensure_partials(ltmi_masks *m, size_t need, hipStream_t stream) {
    need += KCOUNT_BYTES;
    if (need > m->partials_bytes) {
        if (m->partials) {
            LTMI_HIP(hipStreamSynchronize(stream));
            LTMI_HIP(hipFree(m->partials));
            m->partials = nullptr;
            m->partials_bytes = 0;
        }
        LTMI_HIP(hipMalloc((void **)&m->partials, need));
        LTMI_HIP(hipMemsetAsync(m->partials, 0, KCOUNT_BYTES, stream));
        m->partials_bytes = need;
    }
    return LTMI_OK;
}
static inline float *partial_sums(const ltmi_masks *m) {
    return m->partials ? (float *)((char *)m->partials + KCOUNT_BYTES) : nullptr;
}
static inline int *partial_counters(const ltmi_masks *m, int64_t n_blocks) {
    // Off unless LTMI_KSPLIT_FUSED is set: measured on MI355X (profiles/r03_small_tiles.txt) the in-kernel
    // reduction by the last-arriving workgroup loses to the second launch at every size -- final build,
    // tail widened to 64 loads in flight per thread: 1 024 / 4 096 / 8 192 frames 61 / 135 / 193 us against
    // 45 / 114 / 167 us.  The ~20 us it costs whatever the split are the device-scope fences: partials
    // written under one XCD's L2 must be made visible to a workgroup on another XCD (L2 write-back +
    // invalidate inside the kernel), which the kernel boundary of the second launch does once for all.
    static const bool off = getenv("LTMI_KSPLIT_FUSED") == nullptr;
    return (m->partials && !off && n_blocks * (int64_t)sizeof(int) <= (int64_t)KCOUNT_BYTES)
               ? (int *)m->partials : nullptr;
}

template <typename T, int MT, int NG, int WAVES, bool ALIGNED>
static int launch_mfma_variant(ltmi_masks *m, const T *tile, int64_t n_frames, int64_t ld,
                               float *out, int64_t ld_out, int accumulate, int ksplit,
                               hipStream_t stream) {
    auto kern = k_dense_mfma<T, MT, NG, WAVES, ALIGNED>;
    const size_t lds_bytes = (size_t)2 * NG * CHUNK_FLOATS * sizeof(float);
    static bool attr_set[16] = {false};   // per device
    if (lds_bytes > 64 * 1024 && !attr_set[m->device & 15]) {
        LTMI_HIP(hipFuncSetAttribute((const void *)kern,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_set[m->device & 15] = true;
    }
    dim3 grid((unsigned)((n_frames + WAVES * MT * 16 - 1) / (WAVES * MT * 16)), (unsigned)ksplit,
              (unsigned)(m->n_groups / NG));
    hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds_bytes, stream, tile, ld, n_frames, m->n_px,
                       (const float *)m->img, m->n_chunks, out, ld_out, m->n_cols, accumulate,
                       partial_sums(m), ksplit);
    LTMI_HIP(hipGetLastError());
    snprintf(m->last_kernel, sizeof(m->last_kernel),
             "k_dense_mfma<%s,MT=%d,NG=%d,WAVES=%d,%s> grid=(%u,%u,%u)", typeid(T).name(), MT, NG,
             WAVES, ALIGNED ? "aligned" : "unaligned", grid.x, grid.y, grid.z);
    return LTMI_OK;
}

// frame tiles per wave (LdsCfg): 2 unless a bench run forces one of them
// (ltmi_masks_set_tuning waves code 34 = one tile / 8 waves, 35 = two tiles / 4 waves)
// kernel instantiation point.  -DLTMI_DENSE_EXP (experiment builds, scripts/dense_variant.sh): only the
// C5 kernels (float frames, 3 groups + 0 / 2 VALU columns) are compiled -- a minute instead of five
template <typename T, int NG, int ABL, int IND, int NE, int TILES, bool X16 = false>
static auto lds_kernel() -> void (*)(const T *, int64_t, int64_t, int64_t, const float *, int, float *,
                                     int64_t, int, int, float *, int, const int32_t *,
                                     const float *const *, int *, const float *, const float *,
                                     const int32_t *, const int32_t *, int) {
    // the timing-only ablations (tuning codes 31 / 32) and the one-tile-per-wave shape (34) exist for the
    // C2 kernel only -- uint16 pixels, one column group --: they are bench comparisons
    // (scripts/clock_probe.py, profiles/r02_tiles.txt), and every variant is minutes of compile time
    if constexpr ((ABL != 0 || TILES == 1) && !(std::is_same<T, uint16_t>::value && NG == 1 && NE == 0))
        return nullptr;
    else
#ifdef LTMI_DENSE_EXP
    // (C5: float frames, 3 groups; C2 and its 1-byte sibling: one group, with and without X16)
    if constexpr (!(ABL == 0 && IND == 0 && TILES == 2 &&
                    ((std::is_same<T, float>::value && NG == 3) ||
                     ((std::is_same<T, uint16_t>::value || std::is_same<T, uint8_t>::value) && NG == 1 &&
                      NE == 0))))
        return nullptr;
    else
#endif
        return k_dense_lds<T, NG, ABL, IND, NE, TILES, X16>;
}

// the float32 matrix instruction also where the exact float16 products (X16) apply: tuning code 37 on the
// handle, or LTMI_DENSE_F32_INSTR=1 in the environment (read per launch: bench.py times both arithmetics
// through Context.run_udf with the same cached handle)
static inline bool f32_instruction_only(const ltmi_masks *m) {
    if (m->tune_ksplit_ring == 37) return true;
    const char *e = getenv("LTMI_DENSE_F32_INSTR");
    return e && e[0] == '1';
}

// the parts of a pixel-split launch: mask slots in turn (negative: k_dense_lds) or a contiguous range each
// Measured (C2 stack, us per launch incl. the reduction, contiguous -> runs of 4 slots in turn): 1 024 frames (32 parts)
// 39.9 -> 35.8, 2 048 (16) 62.7 -> 55.0, 4 096 (8) 114.7 -> 94.0, but 8 192 (4 parts) 167.5 -> 177.0; 512 x 512 frames
// in 61 parts 46.6 -> 53.8 (256 frames), in 32 parts 112.9 -> 105.5; 1024 x 1024 in 64 parts 162.9 -> 150.4; float32
// frames and 48 columns: no difference (scripts/r5_run18.sh, r5_run19.sh).  In turn for 8, 16, 32, 64 parts.
// LTMI_KSPLIT_STRIDED = 0: never, v > 0: runs of 2^(v-1) slots whatever the number of parts.
static inline int ksplit_order(int ksplit) {
    static const int forced = getenv("LTMI_KSPLIT_STRIDED") ? atoi(getenv("LTMI_KSPLIT_STRIDED")) : -1;
    if (ksplit <= 1 || forced == 0) return ksplit;
    if (forced > 0) return -(ksplit | ((forced - 1) << 16));
    return (ksplit >= 8 && (ksplit & (ksplit - 1)) == 0) ? -(ksplit | (2 << 16)) : ksplit;
}

static inline int lds_tiles(const ltmi_masks *m) {
    return m->tune_ksplit_ring == 34 ? 1 : 2;
}

// generalised LDS-DMA kernel (k_dense_lds): any pixel width, 1 / 2 / 4 column groups per wave
template <typename T, int NG, int TILES>
static int launch_lds_ng_t(ltmi_masks *m, const T *tile, int64_t n_frames, int64_t ld, float *out,
                           int64_t ld_out, int accumulate, hipStream_t stream) {
    using CFG = LdsCfg<NG, 0, TILES>;
    const int abl = m->tune_ksplit_ring == 31 ? 2 : (m->tune_ksplit_ring == 32 ? 1 : 0);
    void (*kern)(const T *, int64_t, int64_t, int64_t, const float *, int, float *, int64_t, int,
                 int, float *, int, const int32_t *, const float *const *, int *, const float *,
                 const float *, const int32_t *, const int32_t *, int) =
        abl == 2 ? lds_kernel<T, NG, 2, 0, 0, TILES>()
                 : (abl == 1 ? lds_kernel<T, NG, 1, 0, 0, TILES>()
                             : lds_kernel<T, NG, 0, 0, 0, TILES>());
    const int32_t *rows = m->roi_rows;                  // ltmi_apply_masks_rows: frames through a row list
    if (rows) kern = lds_kernel<T, NG, 0, 2, 0, TILES>();
    // 1- / 2-byte integer pixels: exact float16 products (X16; tuning code 37 keeps the float32
    // instruction: tests, benches)
    bool x16 = false;
    if constexpr (sizeof(T) <= 2 && std::is_integral<T>::value) {
        x16 = (NG == 1 ? m->img_h : m->img2_h) != nullptr && abl == 0 && !f32_instruction_only(m);
        if (x16)
            kern = rows ? lds_kernel<T, NG, 0, 2, 0, TILES, true>()
                        : lds_kernel<T, NG, 0, 0, 0, TILES, true>();
        m->x16_used = x16;
    }
    if (!kern)
        LTMI_FAIL(LTMI_E_DTYPE, "k_dense_lds<%s, NG=%d>: tuning code %d (ablations / one tile per wave) is built for "
                  "uint16 tiles and one column group only", typeid(T).name(), NG, m->tune_ksplit_ring);
    static bool attr_set[16][8] = {{false}};
    const int variant = (rows ? 3 : abl) + (x16 ? 4 : 0);
    if (!attr_set[m->device & 15][variant]) {
        LTMI_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     CFG::LDS_BYTES));
        attr_set[m->device & 15][variant] = true;
    }
    const float *img = x16 ? (NG == 1 ? m->img_h : m->img2_h) : (NG == 1 ? m->img : m->img2);
    const int n_slots = NG == 1 ? m->n_chunks : m->n_slots2;
    const int64_t gx = (n_frames + CFG::WG_ROWS - 1) / CFG::WG_ROWS;
    const int64_t gz = m->n_groups / NG;
    int ksplit = m->tune_ksplit;
    if (ksplit <= 0) ksplit = choose_ksplit(gx * gz, n_slots);
    ksplit = std::max(1, std::min(ksplit, n_slots));
    {
        const int per = (n_slots + ksplit - 1) / ksplit;
        ksplit = (n_slots + per - 1) / per;
    }
    if (ksplit > 1) {
        int rc = ensure_partials(m, (size_t)ksplit * n_frames * m->n_cols * sizeof(float), stream);
        if (rc != LTMI_OK) return rc;
    }
    dim3 grid((unsigned)gx, (unsigned)ksplit, (unsigned)gz);
    int *kcount = ksplit > 1 ? partial_counters(m, gx * gz) : nullptr;
    hipLaunchKernelGGL(kern, grid, dim3(CFG::WAVES * 64), CFG::LDS_BYTES, stream, tile, ld, n_frames,
                       m->n_px, img, n_slots, out, ld_out, m->n_cols, accumulate, partial_sums(m),
                       ksplit_order(ksplit), rows, (const float *const *)nullptr, kcount,
                       x16 ? (const float *)m->inv_scale : (const float *)nullptr,
                       (const float *)m->tail_val, (const int32_t *)m->tail_col, (const int32_t *)m->tail_px,
                       x16 ? m->tail_n : 0);
    LTMI_HIP(hipGetLastError());
    snprintf(m->last_kernel, sizeof(m->last_kernel),
             "k_dense_lds<%s,NG=%d,ring=%d,tiles=%d%s%s> grid=(%u,%u,%u)", typeid(T).name(), NG,
             CFG::RING, TILES, x16 ? ",f16" : "",
             rows ? ",rows" : (abl ? (abl == 2 ? ",noDMA" : ",noMFMA") : ""), grid.x, grid.y, grid.z);
    if (ksplit > 1 && !kcount) {
        const int64_t n = n_frames * m->n_cols;
        hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           stream, (const float *)partial_sums(m), ksplit, n_frames, m->n_cols, out,
                           ld_out, accumulate);
        LTMI_HIP(hipGetLastError());
    }
    return LTMI_OK;
}

template <typename T, int NG>
static int launch_lds_ng(ltmi_masks *m, const T *tile, int64_t n_frames, int64_t ld, float *out,
                         int64_t ld_out, int accumulate, hipStream_t stream) {
    if constexpr (std::is_same<T, uint16_t>::value && NG == 1) {
        if (lds_tiles(m) != 2 && !m->roi_rows)
            return launch_lds_ng_t<T, NG, 1>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
    }
    return launch_lds_ng_t<T, NG, 2>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
}

// NG MFMA groups + NE VALU columns (stacks of 16 NG + 1..4 columns)
template <typename T, int NG, int NE, int TILES>
static int launch_lds_extras_t(ltmi_masks *m, const T *tile, int64_t n_frames, int64_t ld, float *out,
                               int64_t ld_out, int accumulate, hipStream_t stream) {
    using CFG = LdsCfg<NG, NE, TILES>;
    auto kern = lds_kernel<T, NG, 0, 0, NE, TILES>();
    const int32_t *rows = m->roi_rows;
    if (rows) kern = lds_kernel<T, NG, 0, 2, NE, TILES>();
    bool x16 = false;
    if constexpr (NE == 0 && NG >= 1 && sizeof(T) <= 2 && std::is_integral<T>::value) {
        // (exactly 3 groups: the float16 image of image 3, see launch_lds_ng_t)
        x16 = m->img3_h != nullptr && !f32_instruction_only(m);
        if (x16)
            kern = rows ? lds_kernel<T, NG, 0, 2, NE, TILES, true>()
                        : lds_kernel<T, NG, 0, 0, NE, TILES, true>();
        m->x16_used = x16;
    }
    if (!kern)
        LTMI_FAIL(LTMI_E_DTYPE, "k_dense_lds<%s, NG=%d + %d VALU columns>: variant not built (tuning code %d)",
                  typeid(T).name(), NG, NE, m->tune_ksplit_ring);
    static bool attr_set[16][4] = {{false}};
    const int variant = (rows ? 1 : 0) + (x16 ? 2 : 0);
    if (!attr_set[m->device & 15][variant]) {
        LTMI_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     CFG::LDS_BYTES));
        attr_set[m->device & 15][variant] = true;
    }
    const int n_slots = m->n_slots3;
    const int64_t gx = (n_frames + CFG::WG_ROWS - 1) / CFG::WG_ROWS;
    int ksplit = m->tune_ksplit;
    if (ksplit <= 0) ksplit = choose_ksplit(gx, n_slots);
    ksplit = std::max(1, std::min(ksplit, n_slots));
    {
        const int per = (n_slots + ksplit - 1) / ksplit;
        ksplit = (n_slots + per - 1) / per;
    }
    if (ksplit > 1) {
        int rc = ensure_partials(m, (size_t)ksplit * n_frames * m->n_cols * sizeof(float), stream);
        if (rc != LTMI_OK) return rc;
    }
    dim3 grid((unsigned)gx, (unsigned)ksplit, 1);
    int *kcount = ksplit > 1 ? partial_counters(m, gx) : nullptr;
    hipLaunchKernelGGL(kern, grid, dim3(CFG::WAVES * 64), CFG::LDS_BYTES, stream, tile, ld, n_frames,
                       m->n_px, x16 ? (const float *)m->img3_h : (const float *)m->img3, n_slots, out,
                       ld_out, m->n_cols, accumulate, partial_sums(m), ksplit_order(ksplit), rows,
                       (const float *const *)nullptr, kcount,
                       x16 ? (const float *)m->inv_scale : (const float *)nullptr,
                       (const float *)m->tail_val, (const int32_t *)m->tail_col, (const int32_t *)m->tail_px,
                       x16 ? m->tail_n : 0);
    LTMI_HIP(hipGetLastError());
    if (NE > 0)
        snprintf(m->last_kernel, sizeof(m->last_kernel),
                 "k_dense_lds<%s,NG=%d+%d VALU columns,ring=%d,tiles=%d%s> grid=(%u,%u,1)",
                 typeid(T).name(), NG, NE, CFG::RING, TILES, rows ? ",rows" : "", grid.x, grid.y);
    else
        snprintf(m->last_kernel, sizeof(m->last_kernel),
                 "k_dense_lds<%s,NG=%d,ring=%d,tiles=%d%s%s> grid=(%u,%u,1)", typeid(T).name(), NG,
                 CFG::RING, TILES, x16 ? ",f16" : "", rows ? ",rows" : "", grid.x, grid.y);
    if (ksplit > 1 && !kcount) {
        const int64_t n = n_frames * m->n_cols;
        hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           stream, (const float *)partial_sums(m), ksplit, n_frames, m->n_cols, out,
                           ld_out, accumulate);
        LTMI_HIP(hipGetLastError());
    }
    return LTMI_OK;
}

template <typename T, int NG, int NE>
static int launch_lds_extras(ltmi_masks *m, const T *tile, int64_t n_frames, int64_t ld, float *out,
                             int64_t ld_out, int accumulate, hipStream_t stream) {
    return launch_lds_extras_t<T, NG, NE, 2>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
}

// (for the kernels of other translation units that split the pixel axis: ltmi_fold.hip)
int ltmi::dense_ensure_partials(ltmi_masks *m, size_t need, hipStream_t stream) { return ensure_partials(m, need, stream); }
float *ltmi::dense_partial_sums(const ltmi_masks *m) { return partial_sums(m); }
int ltmi::dense_reduce_partials(ltmi_masks *m, int ksplit, int64_t n_frames, float *out, int64_t ld_out, int accumulate,
                                hipStream_t stream, int n_cols) {
    if (n_cols <= 0) n_cols = m->n_cols;
    const int64_t n = n_frames * n_cols;
    hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       (const float *)partial_sums(m), ksplit, n_frames, n_cols, out, ld_out, accumulate);
    LTMI_HIP(hipGetLastError());
    return LTMI_OK;
}

template <typename T>
static bool lds_kernel_applies(const ltmi_masks *m) {
    return m->n_px >= (m->ng == 1 ? KC : 128);
}

template <typename T>
static int launch_lds(ltmi_masks *m, const T *tile, int64_t n_frames, int64_t ld, float *out,
                      int64_t ld_out, int accumulate, hipStream_t stream) {
    // float32 frames of a stack whose columns are even / odd under a mirror of the detector rows: half the pixels
    // on the matrix cores (ltmi_fold.hip; the handle knows the frame shape through ltmi_masks_set_sig_shape)
    if constexpr (std::is_same<T, float>::value) {
        if (fold_takes(m, tile, ld))
            return launch_fold(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
    }
    // ... and 2-byte integer frames of such a stack when it keeps the float32 matrix instruction (no float16
    // images: more small weights than their float32 tail takes -- radial-Fourier stacks) or is told to
    if constexpr (sizeof(T) <= 2 && std::is_integral<T>::value) {
        const bool x16 = (m->img_h || m->img2_h || m->img3_h) && !f32_instruction_only(m);
        if (!x16 && m->fold && fold_takes16(m, tile, ld, (int)sizeof(T)))
            return launch_fold16(m, tile, (int)sizeof(T), std::is_signed<T>::value, n_frames, ld, out, ld_out,
                                 accumulate, stream);
    }
    if (m->ng == 1) {
        // at most 4 columns (CoM: 3, single-mask analyses: 1 or 2): all of them on the VALU -- a
        // 16-column MFMA tile would be >= 75 % padding that still costs matrix-pipe power
        // (1- / 2-byte integer pixels: the padded group with exact float16 products instead -- 1-byte
        // pixels are VALU bound on the column kernel (uint8, 3 masks: 0.54 -> 0.89 of HBM), 2-byte ones
        // gain a per cent; tuning 37 keeps the VALU columns)
        bool x16_group = false;
        if constexpr (sizeof(T) <= 2 && std::is_integral<T>::value)
            x16_group = m->img_h != nullptr && !f32_instruction_only(m);
        if (m->img3 && m->ng3 == 0 && m->tune_ksplit_ring != 33 && !x16_group) {
            if (m->ne3 == 2)
                return launch_lds_extras<T, 0, 2>(m, tile, n_frames, ld, out, ld_out, accumulate,
                                                  stream);
            return launch_lds_extras<T, 0, 4>(m, tile, n_frames, ld, out, ld_out, accumulate,
                                              stream);
        }
        return launch_lds_ng<T, 1>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
    }
    // 1- / 2-byte integer pixels: the VALU columns need float32 pixels, and with the exact float16
    // products (X16) a padded group costs less than they do -- the padded-group kernel instead
    bool x16_padded = false;
    if constexpr (sizeof(T) <= 2 && std::is_integral<T>::value)
        x16_padded = m->ne3 > 0 && m->img2_h != nullptr && !f32_instruction_only(m);
    if (m->img3 && m->ng3 > 0 && m->tune_ksplit_ring != 33 && !x16_padded) {   // 33: force the padded-group kernel (bench)
#define LTMI_EXTRAS(NG_, NE_)                                                                     \
    if (m->ng3 == NG_ && m->ne3 == NE_)                                                           \
        return launch_lds_extras<T, NG_, NE_>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
        LTMI_EXTRAS(1, 2)
        LTMI_EXTRAS(2, 2)
        LTMI_EXTRAS(2, 4)
        LTMI_EXTRAS(3, 0)
        LTMI_EXTRAS(3, 2)
        LTMI_EXTRAS(3, 4)
#undef LTMI_EXTRAS
    }
    if (m->ng == 2)
        return launch_lds_ng<T, 2>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
    return launch_lds_ng<T, 4>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
}

// ---- shifted masks through the MFMA kernel -----------------------------------------------------------
// Frames are grouped by their (dy, dx); every group gets the image of the stack shifted by that
// amount (built on the device from the raw stack, cached in the handle) and whole workgroups of
// 128 frames (row lists padded with -1).  ONE launch of k_dense_lds<.., IND> for the tile.
struct ShiftCache {
    std::unordered_map<uint64_t, float *> images;       // key (dy, dx) -> device image
    bool x16 = false;                                   // the images hold float16 pieces (k_dense_lds X16)
    size_t image_bytes = 0;
    int sig_h = 0, sig_w = 0;
    int32_t *rows_dev = nullptr;
    const float **wg_img_dev = nullptr;
    size_t rows_cap = 0, wg_cap = 0;
    // host staging of the row lists / image pointers: a small ring, so that a list is not rewritten
    // while an asynchronous copy of an earlier call may still be reading it
    static constexpr int STAGES = 4;
    std::vector<int32_t> rows_stage[STAGES];
    std::vector<const float *> wg_stage[STAGES];
    int stage = 0;
    // float64 / complex128 / exact-integer results (shifted64): images in the f64 layout, scratch for
    // the frames and results of one shift group, frame numbers of the groups
    std::unordered_map<uint64_t, double *> images64;
    size_t image64_bytes = 0;
    int sig_h64 = 0, sig_w64 = 0;
    void *gather = nullptr, *res = nullptr;
    size_t gather_bytes = 0, res_bytes = 0;
    int64_t *idx_dev = nullptr;
    size_t idx_cap = 0;
    std::vector<int64_t> idx_stage[STAGES];
};
constexpr size_t SHIFT_CACHE_BYTES = (size_t)4 << 30;   // at most 4 GiB of shifted images per handle

static void shift_cache_destroy(ltmi_masks *m) {
    ShiftCache *c = (ShiftCache *)m->shift_cache;
    if (!c) return;
    for (auto &kv : c->images) (void)hipFree(kv.second);
    for (auto &kv : c->images64) (void)hipFree(kv.second);
    if (c->gather) (void)hipFree(c->gather);
    if (c->res) (void)hipFree(c->res);
    if (c->idx_dev) (void)hipFree(c->idx_dev);
    if (c->rows_dev) (void)hipFree(c->rows_dev);
    if (c->wg_img_dev) (void)hipFree((void *)c->wg_img_dev);
    delete c;
    m->shift_cache = nullptr;
}

template <typename T>
static int launch_lds_shifted(ltmi_masks *m, const T *tile, int64_t n_frames, int64_t ld, int sig_h,
                              int sig_w, const int32_t *shifts_host, float *out, int64_t ld_out,
                              int accumulate, hipStream_t stream, bool *handled) {
    *handled = false;
    using CFG = LdsCfg<1, 0, 2>;
    ShiftCache *c = (ShiftCache *)m->shift_cache;
    if (!c) {
        c = new (std::nothrow) ShiftCache();
        if (!c) LTMI_FAIL(LTMI_E_NOMEM, "out of host memory");
        m->shift_cache = c;
    }
    const size_t img_bytes = (size_t)m->n_groups * m->n_chunks * CHUNK_FLOATS * sizeof(float);
    // 1- / 2-byte integer pixels against a stack that has column scales: exact float16 products
    bool x16 = false;
    if constexpr (sizeof(T) <= 2 && std::is_integral<T>::value)
        x16 = m->inv_scale != nullptr && m->tail_n == 0 && !f32_instruction_only(m);   // (tail pixels would have to shift along)
    if (c->sig_h != sig_h || c->sig_w != sig_w || c->image_bytes != img_bytes || c->x16 != x16) {
        c->x16 = x16;
        for (auto &kv : c->images) (void)hipFree(kv.second);
        c->images.clear();
        c->sig_h = sig_h;
        c->sig_w = sig_w;
        c->image_bytes = img_bytes;
    }
    // group the frames by shift (order of first appearance)
    std::unordered_map<uint64_t, int> group_of;
    std::vector<uint64_t> keys;
    std::vector<std::vector<int32_t>> members;
    for (int64_t f = 0; f < n_frames; ++f) {
        const uint64_t key = ((uint64_t)(uint32_t)shifts_host[2 * f] << 32) |
                             (uint32_t)shifts_host[2 * f + 1];
        auto it = group_of.find(key);
        int g;
        if (it == group_of.end()) {
            g = (int)keys.size();
            group_of.emplace(key, g);
            keys.push_back(key);
            members.emplace_back();
        } else {
            g = it->second;
        }
        members[g].push_back((int32_t)f);
    }
    // images: build the missing ones; give up (generic kernel) if the budget does not hold them
    size_t missing = 0;
    for (uint64_t key : keys) missing += c->images.count(key) ? 0 : 1;
    if ((c->images.size() + missing) * img_bytes > SHIFT_CACHE_BYTES) {
        if (keys.size() * img_bytes > SHIFT_CACHE_BYTES) return LTMI_OK;      // not handled
        for (auto &kv : c->images) (void)hipFree(kv.second);                  // start over
        c->images.clear();
    }
    const int cpm = (m->result_dtype == LTMI_C64) ? 2 : 1;
    for (uint64_t key : keys) {
        if (c->images.count(key)) continue;
        float *img = nullptr;
        LTMI_HIP(hipMalloc((void **)&img, img_bytes));
        c->images.emplace(key, img);
        LTMI_HIP(hipMemsetAsync(img, 0, img_bytes, stream));
        const int dy = (int)(int32_t)(key >> 32), dx = (int)(int32_t)(key & 0xffffffffu);
        const int64_t total = m->n_masks * cpm * m->n_px;
        const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 65535 * 16);
        if (x16)
            hipLaunchKernelGGL(k_build_image_shifted_h16, dim3(blocks), dim3(256), 0, stream,
                               (const float *)m->gmasks, (_Float16 *)img, m->n_masks, cpm, sig_h, sig_w,
                               dy, dx, m->n_chunks, (const float *)m->inv_scale);
        else
            hipLaunchKernelGGL(k_build_image_shifted, dim3(blocks), dim3(256), 0, stream,
                               (const float *)m->gmasks, img, m->n_masks, cpm, sig_h, sig_w, dy, dx,
                               m->n_chunks);
        LTMI_HIP(hipGetLastError());
    }
    // row lists, padded per group to whole workgroups
    constexpr int WG_ROWS = CFG::WG_ROWS;
    c->stage = (c->stage + 1) % ShiftCache::STAGES;
    std::vector<int32_t> &rows_host = c->rows_stage[c->stage];
    std::vector<const float *> &wg_host = c->wg_stage[c->stage];
    rows_host.clear();
    wg_host.clear();
    for (size_t g = 0; g < keys.size(); ++g) {
        const float *img = c->images[keys[g]];
        const size_t n = members[g].size();
        const size_t n_wg = (n + WG_ROWS - 1) / WG_ROWS;
        rows_host.insert(rows_host.end(), members[g].begin(), members[g].end());
        rows_host.resize(rows_host.size() + (n_wg * WG_ROWS - n), -1);
        for (size_t b = 0; b < n_wg; ++b) wg_host.push_back(img);
    }
    const size_t n_wg = wg_host.size();
    // Stacks with more than 16 real columns: one launch per 16-column group, every launch with the
    // image pointers of ITS group (the shifted image is group-major like the standard one) and its
    // own 16 result columns.  The frames are read once per group -- still one matrix-core launch per
    // group and tile instead of one VALU block per frame.
    const int n_col_groups = (m->n_cols + GROUP - 1) / GROUP;
    const size_t group_floats = (size_t)m->n_chunks * CHUNK_FLOATS;
    wg_host.resize(n_wg * n_col_groups);
    for (int gi = 1; gi < n_col_groups; ++gi)
        for (size_t b = 0; b < n_wg; ++b) wg_host[gi * n_wg + b] = wg_host[b] + gi * group_floats;
    if (c->rows_cap < rows_host.size()) {
        if (c->rows_dev) LTMI_HIP(hipFree(c->rows_dev));
        c->rows_dev = nullptr;
        c->rows_cap = rows_host.size() * 2;
        LTMI_HIP(hipMalloc((void **)&c->rows_dev, c->rows_cap * sizeof(int32_t)));
    }
    if (c->wg_cap < wg_host.size()) {
        if (c->wg_img_dev) LTMI_HIP(hipFree((void *)c->wg_img_dev));
        c->wg_img_dev = nullptr;
        c->wg_cap = wg_host.size() * 2;
        LTMI_HIP(hipMalloc((void **)&c->wg_img_dev, c->wg_cap * sizeof(float *)));
    }
    LTMI_HIP(hipMemcpyAsync(c->rows_dev, rows_host.data(), rows_host.size() * sizeof(int32_t),
                            hipMemcpyHostToDevice, stream));
    LTMI_HIP(hipMemcpyAsync((void *)c->wg_img_dev, wg_host.data(), wg_host.size() * sizeof(float *),
                            hipMemcpyHostToDevice, stream));
    auto kern = lds_kernel<T, 1, 0, 1, 0, 2>();
    if constexpr (sizeof(T) <= 2 && std::is_integral<T>::value) {
        if (x16) kern = lds_kernel<T, 1, 0, 1, 0, 2, true>();
    }
    if (!kern) LTMI_FAIL(LTMI_E_DTYPE, "k_dense_lds<%s, shifted>: variant not built", typeid(T).name());
    static bool attr_set[16][2] = {{false}};
    if (!attr_set[m->device & 15][x16 ? 1 : 0]) {
        LTMI_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     CFG::LDS_BYTES));
        attr_set[m->device & 15][x16 ? 1 : 0] = true;
    }
    for (int gi = 0; gi < n_col_groups; ++gi)
        hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(CFG::WAVES * 64), CFG::LDS_BYTES, stream,
                           tile, ld, n_frames, m->n_px, (const float *)nullptr, m->n_chunks,
                           out + gi * GROUP, ld_out, std::min(GROUP, m->n_cols - gi * GROUP),
                           accumulate, (float *)nullptr, 1, (const int32_t *)c->rows_dev,
                           (const float *const *)c->wg_img_dev + (size_t)gi * n_wg, (int *)nullptr,
                           x16 ? (const float *)m->inv_scale + gi * GROUP : (const float *)nullptr,
                           (const float *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr, 0);
    LTMI_HIP(hipGetLastError());
    snprintf(m->last_kernel, sizeof(m->last_kernel),
             "k_dense_lds<%s,NG=1,shifted%s> grid=(%zu,1,1) x %d column group(s), shift groups=%zu",
             typeid(T).name(), x16 ? ",f16" : "", n_wg, n_col_groups, keys.size());
    *handled = true;
    return LTMI_OK;
}

template <typename T>
static int launch_mfma(ltmi_masks *m, const T *tile, int64_t n_frames, int64_t ld, float *out,
                       int64_t ld_out, int accumulate, hipStream_t stream) {
    const bool aligned = (((uintptr_t)tile) % 16 == 0) && ((ld * (int64_t)sizeof(T)) % 16 == 0);
    // The LDS-DMA kernel does not need 16-B aligned rows: gfx950 serves global_load_lds_dwordx4 from
    // any element-aligned address (detectors with odd row lengths -- 515 x 515 uint16 -- run at the
    // speed of 516 x 516: profiles/r02_unaligned.txt; the direct-load fallback with guarded element
    // loads was 5.5x slower).  LTMI_ALIGNED_DMA_ONLY=1 restores the old dispatch.
    if constexpr (std::is_same<T, float>::value) {
        // opt-in (LTMI_SPLIT=1 or tuning code 36): float32 frames against two or more column groups
        // as bf16 pieces on the bf16 matrix cores (ltmi_split.hip; measured slower than k_dense_lds
        // on C5, profiles/r03_split.txt, hence not the default).  The image is made on first use.
        if (ltmi::split_selected(m->tune_ksplit_ring == 36) && m->blocks.empty() && !m->roi_rows &&
            m->result_dtype != LTMI_F64 && ltmi::split_wanted(m->n_cols, m->n_px) &&
            vector_loads_ok(tile, ld, sizeof(T)) && m->tune_mt == 0 && m->tune_waves == 0 &&
            (m->tune_ksplit_ring == 0 || m->tune_ksplit_ring == 36)) {
            if (!m->split) {
                const int cpm = (m->result_dtype == LTMI_C64) ? 2 : 1;
                const int rcs = ltmi::split_create(m->device, (const float *)m->gmasks, m->n_masks, cpm,
                                                   m->n_px, m->n_cols, &m->split);
                if (rcs != LTMI_OK) return rcs;
            }
            return ltmi::split_apply(m, m->split, tile, n_frames, ld, out, ld_out, accumulate, stream);
        }
    }
    if (vector_loads_ok(tile, ld, sizeof(T)) && m->tune_mt == 0 && m->tune_waves == 0 &&
        lds_kernel_applies<T>(m)) {
        m->x16_used = false;
        const int rc_lds = launch_lds<T>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
        if (rc_lds == LTMI_OK && m->x16_used && m->tail_n > 0) {
            const size_t l = strlen(m->last_kernel);
            snprintf(m->last_kernel + l, sizeof(m->last_kernel) - l, " +tail(%d)", m->tail_n);
        }
        return rc_lds;
    }
    int waves = m->tune_waves ? m->tune_waves : 4;
    int mt = m->tune_mt ? m->tune_mt : (n_frames >= 256 * waves * 32 ? 2 : 1);
    if (m->ng == 4) { mt = 1; }   // keep the accumulator/LDS budget in check
    const int64_t gx = (n_frames + waves * mt * 16 - 1) / (waves * mt * 16);
    const int64_t gz = m->n_groups / m->ng;
    int ksplit = m->tune_ksplit;
    if (ksplit <= 0) {
        // split K only when the frame/column tiling alone leaves CUs idle (< 2 workgroups per CU):
        // measured on C2, 512 workgroups without a split beat 1024 with one (profiles/r01_*)
        ksplit = 1;
        const int64_t wgs = gx * gz;
        if (wgs < 512) {
            ksplit = (int)std::min<int64_t>((1024 + wgs - 1) / wgs, std::max(1, m->n_chunks / 8));
        }
    }
    ksplit = std::max(1, std::min(ksplit, m->n_chunks));
    // every split must own at least one chunk
    {
        const int per = (m->n_chunks + ksplit - 1) / ksplit;
        ksplit = (m->n_chunks + per - 1) / per;
    }
    if (ksplit > 1) {
        int rc0 = ensure_partials(m, (size_t)ksplit * n_frames * m->n_cols * sizeof(float), stream);
        if (rc0 != LTMI_OK) return rc0;
    }
    int rc;
#define LTMI_VARIANT(MT_, NG_, W_)                                                                \
    (aligned ? launch_mfma_variant<T, MT_, NG_, W_, true>(m, tile, n_frames, ld, out, ld_out,      \
                                                          accumulate, ksplit, stream)              \
             : launch_mfma_variant<T, MT_, NG_, W_, false>(m, tile, n_frames, ld, out, ld_out,     \
                                                           accumulate, ksplit, stream))
    // (8-wave workgroups are a tuning choice only -- ltmi_masks_set_tuning(.., waves = 8, ..) -- and
    // compiled for uint16 pixels, the type the comparisons were made on; default: 4 waves)
    constexpr bool W8 = std::is_same<T, uint16_t>::value;
    if (waves == 8 && !W8)
        LTMI_FAIL(LTMI_E_INVALID, "k_dense_mfma with 8 waves is built for uint16 tiles only");
    if (m->ng == 1) {
        if (waves == 4) rc = (mt == 1) ? LTMI_VARIANT(1, 1, 4) : LTMI_VARIANT(2, 1, 4);
        else if constexpr (W8) rc = (mt == 1) ? LTMI_VARIANT(1, 1, 8) : LTMI_VARIANT(2, 1, 8);
        else rc = LTMI_E_INVALID;
    } else if (m->ng == 2) {
        if (waves == 4) rc = (mt == 1) ? LTMI_VARIANT(1, 2, 4) : LTMI_VARIANT(2, 2, 4);
        else if constexpr (W8) rc = (mt == 1) ? LTMI_VARIANT(1, 2, 8) : LTMI_VARIANT(2, 2, 8);
        else rc = LTMI_E_INVALID;
    } else {
        if (waves == 4) rc = LTMI_VARIANT(1, 4, 4);
        else if constexpr (W8) rc = LTMI_VARIANT(1, 4, 8);
        else rc = LTMI_E_INVALID;
    }
#undef LTMI_VARIANT
    if (rc != LTMI_OK) return rc;
    if (ksplit > 1) {
        const int64_t n = n_frames * m->n_cols;
        hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           stream, (const float *)partial_sums(m), ksplit, n_frames, m->n_cols, out,
                           ld_out, accumulate);
        LTMI_HIP(hipGetLastError());
    }
    return LTMI_OK;
}

// ---- generic launch ------------------------------------------------------------------------------
// set by ltmi_apply_masks_shifted around the generic dispatch (per thread)
struct ShiftCtx { const int32_t *shifts = nullptr; int sig_h = 0, sig_w = 0; };
static thread_local ShiftCtx g_shift;

template <typename TIn, typename A, typename S>
static int launch_generic(ltmi_masks *m, const void *tile, int64_t n_frames, int64_t ld, void *out,
                          int64_t ld_out, int accumulate, hipStream_t stream) {
    dim3 grid((unsigned)n_frames, (unsigned)((m->n_masks + GEN_MASKS - 1) / GEN_MASKS));
    if (g_shift.shifts) {
        hipLaunchKernelGGL((k_dense_shifted<TIn, A, S>), grid, dim3(256), 0, stream,
                           (const TIn *)tile, ld, g_shift.sig_h, g_shift.sig_w, g_shift.shifts,
                           (const A *)m->gmasks, (int)m->n_masks, (S *)out, ld_out, accumulate);
        LTMI_HIP(hipGetLastError());
        snprintf(m->last_kernel, sizeof(m->last_kernel), "k_dense_shifted<%s,%s> grid=(%u,%u)",
                 typeid(TIn).name(), typeid(A).name(), grid.x, grid.y);
        return LTMI_OK;
    }
    hipLaunchKernelGGL((k_dense_generic<TIn, A, S>), grid, dim3(256), 0, stream, (const TIn *)tile,
                       ld, n_frames, m->n_px, (const A *)m->gmasks, (int)m->n_masks, (S *)out,
                       ld_out, accumulate);
    LTMI_HIP(hipGetLastError());
    snprintf(m->last_kernel, sizeof(m->last_kernel), "k_dense_generic<%s,%s> grid=(%u,%u)",
             typeid(TIn).name(), typeid(A).name(), grid.x, grid.y);
    return LTMI_OK;
}

template <typename A, typename S>
static int dispatch_generic_real_in(ltmi_masks *m, const void *tile, int tile_dtype,
                                    int64_t n_frames, int64_t ld, void *out, int64_t ld_out,
                                    int accumulate, hipStream_t stream) {
    switch (tile_dtype) {
        case LTMI_BOOL:
        case LTMI_U8: return launch_generic<uint8_t, A, S>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
        case LTMI_I8: return launch_generic<int8_t, A, S>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
        case LTMI_U16: return launch_generic<uint16_t, A, S>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
        case LTMI_I16: return launch_generic<int16_t, A, S>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
        case LTMI_U32: return launch_generic<uint32_t, A, S>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
        case LTMI_I32: return launch_generic<int32_t, A, S>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
        case LTMI_U64: return launch_generic<uint64_t, A, S>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
        case LTMI_I64: return launch_generic<int64_t, A, S>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
        case LTMI_F32: return launch_generic<float, A, S>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
        case LTMI_F64: return launch_generic<double, A, S>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
    }
    LTMI_FAIL(LTMI_E_DTYPE, "tile dtype %s cannot be combined with result dtype %s",
              dtype_name(tile_dtype), dtype_name(m->result_dtype));
}

template <typename A, typename S>
static int dispatch_generic_int(ltmi_masks *m, const void *tile, int tile_dtype, int64_t n_frames,
                                int64_t ld, void *out, int64_t ld_out, int accumulate,
                                hipStream_t stream) {
    if (tile_dtype > LTMI_I64)
        LTMI_FAIL(LTMI_E_DTYPE, "float/complex tile (%s) with integer result dtype %s",
                  dtype_name(tile_dtype), dtype_name(m->result_dtype));
    return dispatch_generic_real_in<A, S>(m, tile, tile_dtype, n_frames, ld, out, ld_out,
                                          accumulate, stream);
}

static int apply_generic(ltmi_masks *m, const void *tile, int tile_dtype, int64_t n_frames,
                         int64_t ld, void *out, int64_t ld_out, int accumulate, hipStream_t stream) {
    switch (m->result_dtype) {
        case LTMI_F32: return dispatch_generic_real_in<float, float>(m, tile, tile_dtype, n_frames, ld, out, ld_out, accumulate, stream);
        case LTMI_F64: return dispatch_generic_real_in<double, double>(m, tile, tile_dtype, n_frames, ld, out, ld_out, accumulate, stream);
        case LTMI_C64:
            if (tile_dtype == LTMI_C64) return launch_generic<cfloat, cfloat, cfloat>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
            if (tile_dtype == LTMI_C128) LTMI_FAIL(LTMI_E_DTYPE, "complex128 tile with complex64 result");
            return dispatch_generic_real_in<cfloat, cfloat>(m, tile, tile_dtype, n_frames, ld, out, ld_out, accumulate, stream);
        case LTMI_C128:
            if (tile_dtype == LTMI_C64) return launch_generic<cfloat, cdouble, cdouble>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
            if (tile_dtype == LTMI_C128) return launch_generic<cdouble, cdouble, cdouble>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
            return dispatch_generic_real_in<cdouble, cdouble>(m, tile, tile_dtype, n_frames, ld, out, ld_out, accumulate, stream);
        case LTMI_BOOL: case LTMI_U8: case LTMI_I8:
            return dispatch_generic_int<uint64_t, uint8_t>(m, tile, tile_dtype, n_frames, ld, out, ld_out, accumulate, stream);
        case LTMI_U16: case LTMI_I16:
            return dispatch_generic_int<uint64_t, uint16_t>(m, tile, tile_dtype, n_frames, ld, out, ld_out, accumulate, stream);
        case LTMI_U32: case LTMI_I32:
            return dispatch_generic_int<uint64_t, uint32_t>(m, tile, tile_dtype, n_frames, ld, out, ld_out, accumulate, stream);
        case LTMI_U64: case LTMI_I64:
            return dispatch_generic_int<uint64_t, uint64_t>(m, tile, tile_dtype, n_frames, ld, out, ld_out, accumulate, stream);
    }
    LTMI_FAIL(LTMI_E_DTYPE, "unsupported result dtype %d", m->result_dtype);
}

extern "C" int ltmi_apply_masks(ltmi_masks *m, const void *tile, int tile_dtype, int64_t n_frames,
                                int64_t ld_tile, void *out, int64_t ld_out, int accumulate,
                                void *stream_) {
    if (!m) LTMI_FAIL(LTMI_E_INVALID, "ltmi_apply_masks: null handle");
    if (n_frames < 0 || ld_tile < m->n_px || ld_out < m->n_masks)
        LTMI_FAIL(LTMI_E_SHAPE, "ltmi_apply_masks: n_frames=%lld ld_tile=%lld (n_px=%lld) "
                  "ld_out=%lld (n_masks=%lld)", (long long)n_frames, (long long)ld_tile,
                  (long long)m->n_px, (long long)ld_out, (long long)m->n_masks);
    if (dtype_size(tile_dtype) == 0)
        LTMI_FAIL(LTMI_E_DTYPE, "ltmi_apply_masks: unknown tile dtype %d", tile_dtype);
    if (n_frames == 0) return LTMI_OK;
    if (!tile || !out) LTMI_FAIL(LTMI_E_INVALID, "ltmi_apply_masks: null tile/out pointer");
    // (a runtime call that failed OUTSIDE the library -- a refused host registration, say -- leaves its code as
    // the thread's sticky last error; the launch checks below must not report it for these kernels)
    (void)hipGetLastError();
    hipStream_t stream = (hipStream_t)stream_;
    LTMI_HIP(hipSetDevice(m->device));
    // float frames against a stack whose fast kernels also multiply zeros the stack does not hold (or, for a dense
    // stack held as column blocks, skip zeros it does hold): frames with non-finite results are computed again with
    // the reference's arithmetic (ltmi_guard.hip)
    if (ltmi::guard_wanted(m, tile_dtype))
        return ltmi::guard_apply(m, tile, tile_dtype, n_frames, ld_tile, out, ld_out, accumulate, stream);
    if (m->guard) ltmi::guard_note_unchecked(m);
    return ltmi::apply_masks_unguarded(m, tile, tile_dtype, n_frames, ld_tile, out, ld_out, accumulate, stream);
}

int ltmi::apply_masks_unguarded(ltmi_masks *m, const void *tile, int tile_dtype, int64_t n_frames, int64_t ld_tile,
                                void *out, int64_t ld_out, int accumulate, hipStream_t stream) {
    void *stream_ = (void *)stream;
    m->last_exact = false;
    if (m->kind == 2)
        return ltmi::csr_apply(m, tile, tile_dtype, n_frames, ld_tile, out, ld_out, accumulate,
                               stream);
    if (m->kind == 0 && mfma_tile_dtype(tile_dtype) && !m->blocks.empty() &&
        m->tune_mt == 0 && m->tune_waves == 0 && m->tune_ksplit_ring != 33) {
        const size_t elem = (size_t)dtype_size(m->result_dtype);
        for (size_t b = 0; b < m->blocks.size(); ++b) {
            ltmi_masks *c = m->blocks[b];
            c->tune_ksplit = m->tune_ksplit;
            c->tune_ksplit_ring = m->tune_ksplit_ring;
            c->roi_rows = m->roi_rows;                      // (ltmi_apply_masks_rows: every block reads them)
            const int rc = ltmi_apply_masks(c, tile, tile_dtype, n_frames, ld_tile,
                                            (unsigned char *)out + (size_t)m->block_first[b] * elem,
                                            ld_out, accumulate, stream_);
            c->roi_rows = nullptr;
            if (rc != LTMI_OK) return rc;
        }
        snprintf(m->last_kernel, sizeof(m->last_kernel), "%zu column blocks, last: %.90s",
                 m->blocks.size(), m->blocks.back()->last_kernel);
        return LTMI_OK;
    }
    if (m->kind == 0 && mfma_tile_dtype(tile_dtype)) {
        float *o = (float *)out;
        const int64_t ldo = ld_out * (m->result_dtype == LTMI_C64 ? 2 : 1);
        switch (tile_dtype) {
            case LTMI_BOOL:
            case LTMI_U8: return launch_mfma<uint8_t>(m, (const uint8_t *)tile, n_frames, ld_tile, o, ldo, accumulate, stream);
            case LTMI_I8: return launch_mfma<int8_t>(m, (const int8_t *)tile, n_frames, ld_tile, o, ldo, accumulate, stream);
            case LTMI_U16: return launch_mfma<uint16_t>(m, (const uint16_t *)tile, n_frames, ld_tile, o, ldo, accumulate, stream);
            case LTMI_I16: return launch_mfma<int16_t>(m, (const int16_t *)tile, n_frames, ld_tile, o, ldo, accumulate, stream);
            case LTMI_F32: return launch_mfma<float>(m, (const float *)tile, n_frames, ld_tile, o, ldo, accumulate, stream);
        }
    }
    if (m->result_dtype == LTMI_F64 || m->result_dtype == LTMI_C128 ||
        (m->result_dtype >= LTMI_U8 && m->result_dtype <= LTMI_I64)) {
        bool handled = false;
        const int rc = ltmi::dense64_apply(m, tile, tile_dtype, n_frames, ld_tile, out, ld_out,
                                           accumulate, stream, &handled);
        if (rc != LTMI_OK || handled) return rc;
    }
    if (m->roi_rows) LTMI_FAIL(LTMI_E_INVALID, "ltmi_apply_masks: the generic kernel does not take a row list");
    return apply_generic(m, tile, tile_dtype, n_frames, ld_tile, out, ld_out, accumulate, stream);
}

// Frames of a tile through a row list (a region of interest without a gathered copy):
//   out[i, k] (+)= sum_p tile[rows[i], p] * masks[k, p],   i < n_rows
// *handled = 0 (and nothing done) when the handle / tile has no row-list kernel: dense stacks with
// float32 / complex64 results of at most 64 real columns on the LDS-DMA kernels only.
extern "C" int ltmi_apply_masks_rows(ltmi_masks *m, const void *tile, int tile_dtype,
                                     const int32_t *rows, int64_t n_rows, int64_t ld_tile, void *out,
                                     int64_t ld_out, int accumulate, void *stream_, int *handled) {
    if (!m || !handled) LTMI_FAIL(LTMI_E_INVALID, "ltmi_apply_masks_rows: null handle / handled");
    *handled = 0;
    if (n_rows < 0 || ld_tile < m->n_px || ld_out < m->n_masks)
        LTMI_FAIL(LTMI_E_SHAPE, "ltmi_apply_masks_rows: n_rows=%lld ld_tile=%lld (n_px=%lld) "
                  "ld_out=%lld (n_masks=%lld)", (long long)n_rows, (long long)ld_tile,
                  (long long)m->n_px, (long long)ld_out, (long long)m->n_masks);
    if (dtype_size(tile_dtype) == 0)
        LTMI_FAIL(LTMI_E_DTYPE, "ltmi_apply_masks_rows: unknown tile dtype %d", tile_dtype);
    if (n_rows == 0) { *handled = 1; return LTMI_OK; }
    if (!tile || !out || !rows) LTMI_FAIL(LTMI_E_INVALID, "ltmi_apply_masks_rows: null pointer");
    if (m->kind == 2) {
        // sparse stacks: the blocked image's kernel reads frames through the row list as well
        if (n_rows >= (1ll << 31) || !ltmi::csr_rows_ok(m, tile, tile_dtype, ld_tile)) return LTMI_OK;
        m->roi_rows = rows;
        const int rc = ltmi_apply_masks(m, tile, tile_dtype, n_rows, ld_tile, out, ld_out, accumulate,
                                        stream_);
        m->roi_rows = nullptr;
        *handled = 1;
        return rc;
    }
    if (m->kind != 2 && m->img64 && m->blocks.empty() && n_rows < (1ll << 31) &&
        ltmi::dense64_rows_ok(m, tile, tile_dtype, ld_tile)) {
        // float64 / complex128 results: the f64 LDS-DMA kernel reads frames through the row list
        m->roi_rows = rows;
        const int rc = ltmi_apply_masks(m, tile, tile_dtype, n_rows, ld_tile, out, ld_out, accumulate,
                                        stream_);
        m->roi_rows = nullptr;
        *handled = 1;
        return rc;
    }
    if (m->kind != 0 || !mfma_tile_dtype(tile_dtype) || m->tune_mt != 0 ||
        m->tune_waves != 0 || m->tune_ksplit_ring == 33 || n_rows >= (1ll << 31) ||
        !vector_loads_ok(tile, ld_tile, (size_t)dtype_size(tile_dtype)))
        return LTMI_OK;
    auto lds_ok = [&](const ltmi_masks *h) {
        switch (tile_dtype) {
            case LTMI_BOOL: case LTMI_U8: case LTMI_I8: return lds_kernel_applies<uint8_t>(h);
            case LTMI_U16: case LTMI_I16: return lds_kernel_applies<uint16_t>(h);
            case LTMI_F32: return lds_kernel_applies<float>(h);
        }
        return false;
    };
    // (stacks of more than 64 real columns: every column block must take the row list)
    bool lds = m->blocks.empty() ? lds_ok(m) : true;
    for (const ltmi_masks *b : m->blocks) lds = lds && lds_ok(b);
    if (!lds) return LTMI_OK;
    m->roi_rows = rows;
    const int rc = ltmi_apply_masks(m, tile, tile_dtype, n_rows, ld_tile, out, ld_out, accumulate,
                                    stream_);
    m->roi_rows = nullptr;
    *handled = 1;
    return rc;
}

// ---- shifted masks, float64 / complex128 / exact-integer results ---------------------------------------
// The f64 matrix-core path (ltmi_dense64.hip) works off m->img64.  Frames are grouped by their shift;
// every distinct shift gets the image of the shifted stack (built on the device, cached) and the group
// runs through dense64_apply with that image in place of the unshifted one: a whole tile directly when
// it has ONE shift (a constant descan correction), otherwise group by group on gathered frames, the
// result rows scattered back.  Up to SHIFT64_SMALL_GROUPS distinct shifts whatever the tile; more -- a descan
// correction over +- 8 pixels in both directions has 289 -- while a group averages SHIFT64_MIN_AVG frames or more
// (three launches per group: below that the per-frame kernel's single launch wins) and the shifted images fit
// SHIFT_CACHE_BYTES, up to SHIFT64_MAX_GROUPS.  Anything else: not handled (per-frame kernel).
constexpr size_t SHIFT64_SMALL_GROUPS = 256;
constexpr size_t SHIFT64_MAX_GROUPS = 4096;
constexpr int64_t SHIFT64_MIN_AVG = 8;

template <typename R>
__global__ void k_scatter_rows(const R *__restrict__ src, int64_t n_rows, int n_cols,
                               const int64_t *__restrict__ rows, R *__restrict__ out, int64_t ld_out,
                               int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * n_cols) return;
    const int64_t r = i / n_cols;
    const int col = (int)(i % n_cols);
    R *p = out + rows[r] * ld_out + col;
    *p = accumulate ? (R)(*p + src[i]) : src[i];
}

static int ensure_bytes(void **buf, size_t *have, size_t need, hipStream_t stream) {
    if (*have >= need) return LTMI_OK;
    if (*buf) {
        LTMI_HIP(hipStreamSynchronize(stream));
        LTMI_HIP(hipFree(*buf));
        *buf = nullptr;
        *have = 0;
    }
    LTMI_HIP(hipMalloc(buf, need));
    *have = need;
    return LTMI_OK;
}

static int shifted64(ltmi_masks *m, const void *tile, int tile_dtype, int64_t n_frames, int64_t ld_tile,
                     int sig_h, int sig_w, const int32_t *shifts_host, void *out, int64_t ld_out,
                     int accumulate, hipStream_t stream, bool *handled) {
    *handled = false;
    ShiftCache *c = (ShiftCache *)m->shift_cache;
    if (!c) {
        c = new (std::nothrow) ShiftCache();
        if (!c) LTMI_FAIL(LTMI_E_NOMEM, "out of host memory");
        m->shift_cache = c;
    }
    // group the frames by shift (order of first appearance)
    std::unordered_map<uint64_t, int> group_of;
    std::vector<uint64_t> keys;
    std::vector<std::vector<int64_t>> members;
    for (int64_t f = 0; f < n_frames; ++f) {
        const uint64_t key = ((uint64_t)(uint32_t)shifts_host[2 * f] << 32) |
                             (uint32_t)shifts_host[2 * f + 1];
        auto it = group_of.find(key);
        int g;
        if (it == group_of.end()) {
            if (keys.size() == SHIFT64_MAX_GROUPS ||
                (keys.size() >= SHIFT64_SMALL_GROUPS && (int64_t)(keys.size() + 1) * SHIFT64_MIN_AVG > n_frames))
                return LTMI_OK;                                          // not handled
            g = (int)keys.size();
            group_of.emplace(key, g);
            keys.push_back(key);
            members.emplace_back();
        } else {
            g = it->second;
        }
        members[g].push_back(f);
    }
    const size_t img_bytes = ltmi::dense64_image_bytes(m);
    if (c->sig_h64 != sig_h || c->sig_w64 != sig_w || c->image64_bytes != img_bytes) {
        LTMI_HIP(hipStreamSynchronize(stream));
        for (auto &kv : c->images64) (void)hipFree(kv.second);
        c->images64.clear();
        c->sig_h64 = sig_h;
        c->sig_w64 = sig_w;
        c->image64_bytes = img_bytes;
    }
    size_t missing = 0;
    for (uint64_t key : keys) missing += c->images64.count(key) ? 0 : 1;
    if ((c->images64.size() + missing) * img_bytes > SHIFT_CACHE_BYTES) {
        if (keys.size() * img_bytes > SHIFT_CACHE_BYTES) return LTMI_OK;         // not handled
        LTMI_HIP(hipStreamSynchronize(stream));
        for (auto &kv : c->images64) (void)hipFree(kv.second);
        c->images64.clear();
    }
    for (uint64_t key : keys) {
        if (c->images64.count(key)) continue;
        double *img = nullptr;
        LTMI_HIP(hipMalloc((void **)&img, img_bytes));
        c->images64.emplace(key, img);
        LTMI_HIP(hipMemsetAsync(img, 0, img_bytes, stream));                      // the padding
        const int rc = ltmi::dense64_build_shifted(m, sig_h, sig_w, (int)(int32_t)(key >> 32),
                                                   (int)(int32_t)(key & 0xffffffffu), img, stream);
        if (rc != LTMI_OK) return rc;
    }
    double *const img_plain = m->img64;
    auto run = [&](double *img, const void *frames, int64_t n, int64_t ld, void *dst, int64_t ld_dst,
                   int acc, bool *ok) {
        m->img64 = img;
        const int rc = ltmi::dense64_apply(m, frames, tile_dtype, n, ld, dst, ld_dst, acc, stream, ok);
        m->img64 = img_plain;
        return rc;
    };
    if (keys.size() == 1) {                      // one shift for the whole tile: no gather, no scatter
        bool ok = false;
        const int rc = run(c->images64[keys[0]], tile, n_frames, ld_tile, out, ld_out, accumulate, &ok);
        if (rc != LTMI_OK) return rc;
        if (ok) {
            const size_t len = strlen(m->last_kernel);
            snprintf(m->last_kernel + len, sizeof(m->last_kernel) - len, " shifted, 1 group");
        }
        *handled = ok;
        return LTMI_OK;
    }
    // several shifts: frame numbers of all groups to the device, then group by group
    const size_t esz = (size_t)dtype_size(tile_dtype), rsz = (size_t)dtype_size(m->result_dtype);
    size_t largest = 0;
    c->stage = (c->stage + 1) % ShiftCache::STAGES;
    std::vector<int64_t> &idx_host = c->idx_stage[c->stage];
    idx_host.clear();
    for (auto &mem : members) {
        largest = std::max(largest, mem.size());
        idx_host.insert(idx_host.end(), mem.begin(), mem.end());
    }
    int rc = ensure_bytes(&c->gather, &c->gather_bytes, largest * (size_t)m->n_px * esz, stream);
    if (rc != LTMI_OK) return rc;
    rc = ensure_bytes(&c->res, &c->res_bytes, largest * (size_t)m->n_masks * rsz, stream);
    if (rc != LTMI_OK) return rc;
    if (c->idx_cap < idx_host.size()) {
        if (c->idx_dev) {
            LTMI_HIP(hipStreamSynchronize(stream));
            LTMI_HIP(hipFree(c->idx_dev));
        }
        c->idx_dev = nullptr;
        c->idx_cap = idx_host.size() * 2;
        LTMI_HIP(hipMalloc((void **)&c->idx_dev, c->idx_cap * sizeof(int64_t)));
    }
    LTMI_HIP(hipMemcpyAsync(c->idx_dev, idx_host.data(), idx_host.size() * sizeof(int64_t),
                            hipMemcpyHostToDevice, stream));
    size_t first = 0;
    for (size_t g = 0; g < keys.size(); ++g) {
        const int64_t n = (int64_t)members[g].size();
        const int64_t *idx = c->idx_dev + first;
        first += (size_t)n;
        rc = ltmi_gather_rows(m->device, tile, ld_tile * (int64_t)esz, idx, n, m->n_px * (int64_t)esz,
                              c->gather, stream);
        if (rc != LTMI_OK) return rc;
        bool ok = false;
        rc = run(c->images64[keys[g]], c->gather, n, m->n_px, c->res, m->n_masks, 0, &ok);
        if (rc != LTMI_OK) return rc;
        if (!ok) {
            if (g == 0) return LTMI_OK;          // (nothing written yet: the per-frame kernel takes over)
            LTMI_FAIL(LTMI_E_DTYPE, "ltmi_apply_masks_shifted_host: the float64 path stopped midway");
        }
        // result rows back to their frames (complex128 rows as 2 doubles per mask)
        const int n_cols = (int)(m->n_masks * (m->result_dtype == LTMI_C128 ? 2 : 1));
        const int64_t ldo = ld_out * (m->result_dtype == LTMI_C128 ? 2 : 1);
        const dim3 grid((unsigned)((n * n_cols + 255) / 256));
        switch (m->result_dtype == LTMI_C128 ? 8 : (int)rsz) {
            case 1: hipLaunchKernelGGL(k_scatter_rows<uint8_t>, grid, dim3(256), 0, stream, (const uint8_t *)c->res, n, n_cols, idx, (uint8_t *)out, ldo, accumulate); break;
            case 2: hipLaunchKernelGGL(k_scatter_rows<uint16_t>, grid, dim3(256), 0, stream, (const uint16_t *)c->res, n, n_cols, idx, (uint16_t *)out, ldo, accumulate); break;
            case 4: hipLaunchKernelGGL(k_scatter_rows<uint32_t>, grid, dim3(256), 0, stream, (const uint32_t *)c->res, n, n_cols, idx, (uint32_t *)out, ldo, accumulate); break;
            default:
                if (m->result_dtype == LTMI_F64 || m->result_dtype == LTMI_C128)
                    hipLaunchKernelGGL(k_scatter_rows<double>, grid, dim3(256), 0, stream, (const double *)c->res, n, n_cols, idx, (double *)out, ldo, accumulate);
                else
                    hipLaunchKernelGGL(k_scatter_rows<uint64_t>, grid, dim3(256), 0, stream, (const uint64_t *)c->res, n, n_cols, idx, (uint64_t *)out, ldo, accumulate);
                break;
        }
        LTMI_HIP(hipGetLastError());
    }
    {
        const size_t len = strlen(m->last_kernel);
        snprintf(m->last_kernel + len, sizeof(m->last_kernel) - len, " shifted, %zu groups", keys.size());
    }
    *handled = true;
    return LTMI_OK;
}

extern "C" int ltmi_apply_masks_shifted(ltmi_masks *m, const void *tile, int tile_dtype,
                                        int64_t n_frames, int64_t ld_tile, int sig_h, int sig_w,
                                        const int32_t *shifts, void *out, int64_t ld_out,
                                        int accumulate, void *stream_) {
    if (!m) LTMI_FAIL(LTMI_E_INVALID, "ltmi_apply_masks_shifted: null handle");
    if (m->kind == 2 || !m->gmasks)
        LTMI_FAIL(LTMI_E_INVALID, "ltmi_apply_masks_shifted: needs a dense mask handle");
    if (sig_h <= 0 || sig_w <= 0 || (int64_t)sig_h * sig_w != m->n_px)
        LTMI_FAIL(LTMI_E_SHAPE, "ltmi_apply_masks_shifted: sig shape (%d, %d) does not match "
                  "n_px=%lld", sig_h, sig_w, (long long)m->n_px);
    if (n_frames < 0 || ld_tile < m->n_px || ld_out < m->n_masks)
        LTMI_FAIL(LTMI_E_SHAPE, "ltmi_apply_masks_shifted: bad leading dimensions");
    if (dtype_size(tile_dtype) == 0)
        LTMI_FAIL(LTMI_E_DTYPE, "ltmi_apply_masks_shifted: unknown tile dtype %d", tile_dtype);
    if (n_frames == 0) return LTMI_OK;
    if (!tile || !out || !shifts)
        LTMI_FAIL(LTMI_E_INVALID, "ltmi_apply_masks_shifted: null pointer");
    LTMI_HIP(hipSetDevice(m->device));
    g_shift.shifts = shifts;
    g_shift.sig_h = sig_h;
    g_shift.sig_w = sig_w;
    const int rc = apply_generic(m, tile, tile_dtype, n_frames, ld_tile, out, ld_out, accumulate,
                                 (hipStream_t)stream_);
    g_shift.shifts = nullptr;
    return rc;
}

extern "C" int ltmi_apply_masks_shifted_host(ltmi_masks *m, const void *tile, int tile_dtype,
                                             int64_t n_frames, int64_t ld_tile, int sig_h,
                                             int sig_w, const int32_t *shifts_host, void *out,
                                             int64_t ld_out, int accumulate, void *stream_) {
    if (!m) LTMI_FAIL(LTMI_E_INVALID, "ltmi_apply_masks_shifted_host: null handle");
    if (m->kind == 2 || !m->gmasks)
        LTMI_FAIL(LTMI_E_INVALID, "ltmi_apply_masks_shifted_host: needs a dense mask handle");
    if (sig_h <= 0 || sig_w <= 0 || (int64_t)sig_h * sig_w != m->n_px)
        LTMI_FAIL(LTMI_E_SHAPE, "ltmi_apply_masks_shifted_host: sig shape (%d, %d) does not match "
                  "n_px=%lld", sig_h, sig_w, (long long)m->n_px);
    if (n_frames < 0 || ld_tile < m->n_px || ld_out < m->n_masks)
        LTMI_FAIL(LTMI_E_SHAPE, "ltmi_apply_masks_shifted_host: bad leading dimensions");
    if (dtype_size(tile_dtype) == 0)
        LTMI_FAIL(LTMI_E_DTYPE, "ltmi_apply_masks_shifted_host: unknown tile dtype %d", tile_dtype);
    if (n_frames == 0) return LTMI_OK;
    if (!tile || !out || !shifts_host)
        LTMI_FAIL(LTMI_E_INVALID, "ltmi_apply_masks_shifted_host: null pointer");
    LTMI_HIP(hipSetDevice(m->device));
    hipStream_t stream = (hipStream_t)stream_;
    const size_t esz = (size_t)dtype_size(tile_dtype);
    const bool aligned = vector_loads_ok(tile, ld_tile, esz);
    if (m->kind == 0 && aligned && m->n_px >= KC && n_frames < (1ll << 31)) {
        bool handled = false;
        int rc = LTMI_OK;
        const int64_t ldo = (m->result_dtype == LTMI_C64) ? 2 * ld_out : ld_out;
        switch (tile_dtype) {
            case LTMI_U8: rc = launch_lds_shifted<uint8_t>(m, (const uint8_t *)tile, n_frames, ld_tile, sig_h, sig_w, shifts_host, (float *)out, ldo, accumulate, stream, &handled); break;
            case LTMI_I8: rc = launch_lds_shifted<int8_t>(m, (const int8_t *)tile, n_frames, ld_tile, sig_h, sig_w, shifts_host, (float *)out, ldo, accumulate, stream, &handled); break;
            case LTMI_U16: rc = launch_lds_shifted<uint16_t>(m, (const uint16_t *)tile, n_frames, ld_tile, sig_h, sig_w, shifts_host, (float *)out, ldo, accumulate, stream, &handled); break;
            case LTMI_I16: rc = launch_lds_shifted<int16_t>(m, (const int16_t *)tile, n_frames, ld_tile, sig_h, sig_w, shifts_host, (float *)out, ldo, accumulate, stream, &handled); break;
            case LTMI_F32: rc = launch_lds_shifted<float>(m, (const float *)tile, n_frames, ld_tile, sig_h, sig_w, shifts_host, (float *)out, ldo, accumulate, stream, &handled); break;
            default: break;
        }
        if (rc != LTMI_OK) return rc;
        if (handled) return LTMI_OK;
    }
    if (m->img64 && tile_dtype != LTMI_C64 && tile_dtype != LTMI_C128 && n_frames < (1ll << 31)) {
        bool handled = false;
        const int rc = shifted64(m, tile, tile_dtype, n_frames, ld_tile, sig_h, sig_w, shifts_host, out,
                                 ld_out, accumulate, stream, &handled);
        if (rc != LTMI_OK) return rc;
        if (handled) return LTMI_OK;
    }
    // everything else: per-frame kernel with the shifts uploaded to the handle's scratch
    ShiftCache *c = (ShiftCache *)m->shift_cache;
    if (!c) {
        c = new (std::nothrow) ShiftCache();
        if (!c) LTMI_FAIL(LTMI_E_NOMEM, "out of host memory");
        m->shift_cache = c;
    }
    const size_t need = (size_t)n_frames * 2;
    if (c->rows_cap < need) {
        if (c->rows_dev) LTMI_HIP(hipFree(c->rows_dev));
        c->rows_dev = nullptr;
        c->rows_cap = need * 2;
        LTMI_HIP(hipMalloc((void **)&c->rows_dev, c->rows_cap * sizeof(int32_t)));
    }
    c->stage = (c->stage + 1) % ShiftCache::STAGES;
    c->rows_stage[c->stage].assign(shifts_host, shifts_host + need);
    LTMI_HIP(hipMemcpyAsync(c->rows_dev, c->rows_stage[c->stage].data(), need * sizeof(int32_t),
                            hipMemcpyHostToDevice, stream));
    return ltmi_apply_masks_shifted(m, tile, tile_dtype, n_frames, ld_tile, sig_h, sig_w, c->rows_dev,
                                    out, ld_out, accumulate, stream_);
}

