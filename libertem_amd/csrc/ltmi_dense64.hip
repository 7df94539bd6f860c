// ltmi_dense64.hip -- dense mask stacks with FLOAT64 results on the f64 matrix cores.
//
// np.result_type(input_dtype, mask_dtype) is float64 for int32 / uint32 / int64 / float64 data
// (reference udf/base.py:106-123, udf/masks.py:362) and for float64 masks; the reference then runs
// a dgemm.  Here: v_mfma_f64_16x16x4_f64 (16 frames x 16 masks x 4 pixels per instruction, exact
// f64 FMA chain).  Structure of the direct-load f32 kernel (k_dense_mfma): a wave owns 16 frames,
// lane (m, kg) loads 4 consecutive pixels of frame m straight from HBM into a rolling register ring
// (16 blocks ahead), converts them to f64 in registers; the 32-KiB mask chunk (256 px x 16 columns,
// f64, XOR-swizzled like the f32 image) is shared by the workgroup through LDS, double buffered.
#include "ltmi_common.h"
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <typeinfo>
#include <type_traits>
#include <utility>

namespace ltmi {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));

constexpr int KC64 = 256;                       // pixels per mask chunk
constexpr int CH64 = 16 * KC64;                 // doubles per (group, chunk) = 32 KiB

// double index of (column n, pixel q) inside a chunk block: a lane's 4 pixels are two 16-B units;
// XOR with n spreads the 16 lanes of a read over 16 different units (conflict-free)
__host__ __device__ static inline int img64_index(int n, int q) {
    const int blk = q >> 4, kg = (q >> 2) & 3, j = q & 3;
    const int v = (blk * 8 + kg * 2 + (j >> 1)) ^ n;
    return n * KC64 + v * 2 + (j & 1);
}

// cpm: real columns per mask (2 for complex128 stacks: interleaved (re, im) like the f32 image)
template <typename SRC>
__global__ void k_build_image64(const SRC *__restrict__ src, double *__restrict__ img,
                                int64_t n_masks, int64_t n_px, int n_chunks, int cpm) {
    const int64_t total = n_masks * n_px * cpm;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int part = (int)(i % cpm);
        const int64_t kp = i / cpm;
        const int64_t p = kp % n_px;
        const int64_t k = (kp / n_px) * cpm + part;           // real column
        const int g = (int)(k / 16), n = (int)(k % 16);
        const int c = (int)(p / KC64), q = (int)(p % KC64);
        img[((size_t)g * n_chunks + c) * CH64 + img64_index(n, q)] = (double)src[i];
    }
}

// the image of the stack shifted by (dy, dx): mask'[k](y, x) = mask[k](y - dy, x - dx) inside the frame,
// 0 outside (udf/masks.py:85-124); every element of the unpadded image is written
template <typename SRC>
__global__ void k_build_image64_shifted(const SRC *__restrict__ src, double *__restrict__ img,
                                        int64_t n_masks, int64_t n_px, int n_chunks, int cpm, int sig_h,
                                        int sig_w, int dy, int dx) {
    const int64_t total = n_masks * n_px * cpm;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int part = (int)(i % cpm);
        const int64_t kp = i / cpm;
        const int64_t p = kp % n_px, mask = kp / n_px;
        const int y = (int)(p / sig_w) - dy, x = (int)(p % sig_w) - dx;
        double v = 0.;
        if (y >= 0 && y < sig_h && x >= 0 && x < sig_w)
            v = (double)src[(mask * n_px + (int64_t)y * sig_w + x) * cpm + part];
        const int64_t k = mask * cpm + part;
        const int g = (int)(k / 16), n = (int)(k % 16);
        const int c = (int)(p / KC64), q = (int)(p % KC64);
        img[((size_t)g * n_chunks + c) * CH64 + img64_index(n, q)] = v;
    }
}

template <typename T, int WAVES, bool VEC>
__global__ void __launch_bounds__(WAVES * 64)
k_dense_mfma_f64(const T *__restrict__ tile, int64_t ld, int64_t n_frames, int64_t n_px,
                 const double *__restrict__ img, int n_chunks, double *__restrict__ out,
                 int64_t ld_out, int n_cols, int accumulate, double *__restrict__ partials,
                 int ksplit) {
    extern __shared__ __attribute__((aligned(16))) double lds64[];     // 2 stages x CH64
    constexpr int NT = WAVES * 64;
    constexpr int BUNITS = CH64 * 8 / 16 / NT;          // 16-B units per thread per chunk
    constexpr int DEPTH = sizeof(T) == 8 ? 8 : 16;      // blocks (of 16 px) in the register ring
    typedef T __attribute__((ext_vector_type(4))) raw_t;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int m = lane & 15, kg = lane >> 4;
    const int ks = blockIdx.y, g0 = blockIdx.z;

    const int n_full = VEC ? (int)(n_px / KC64) : 0;
    const int per = (n_chunks + ksplit - 1) / ksplit;
    const int c_begin = ks * per;
    const int c_end = min(n_chunks, c_begin + per);
    const int cf_end = min(c_end, n_full);

    const int64_t f_wave = (int64_t)blockIdx.x * (WAVES * 16) + wave * 16;
    int64_t f = f_wave + m;
    if (f > n_frames - 1) f = n_frames - 1;             // clamp: loads stay valid, result discarded
    const T *rowp = tile + f * ld + kg * 4;

    f64x4 acc = {0., 0., 0., 0.};
    const u32x4_ *img_units = (const u32x4_ *)img;
    auto unit = [&](int i, int c) -> int64_t {
        return ((int64_t)g0 * n_chunks + c) * (CH64 / 2) + i * NT + tid;
    };
    const int lds_lane = m * KC64;
    auto rd_b = [&](const double *stage, int blk, int h) {
        return *(const f64x2 *)(stage + lds_lane + (((blk * 8 + kg * 2 + h) ^ m) << 1));
    };

    if (c_begin < cf_end) {
        raw_t raw[DEPTH];
        u32x4_ breg[BUNITS];
        const int64_t px0 = (int64_t)c_begin * KC64;
        const int n_blocks = (cf_end - c_begin) * 16;
#pragma unroll
        for (int i = 0; i < BUNITS; ++i) breg[i] = img_units[unit(i, c_begin)];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i)
            raw[i] = __builtin_nontemporal_load((const raw_t *)(rowp + px0 + min(i, n_blocks - 1) * 16));
#pragma unroll
        for (int i = 0; i < BUNITS; ++i) ((u32x4_ *)lds64)[i * NT + tid] = breg[i];
        __syncthreads();

        for (int c = c_begin; c < cf_end; ++c) {
            const int cn = min(c + 1, cf_end - 1);
            const int s = (c - c_begin) & 1;
#pragma unroll
            for (int i = 0; i < BUNITS; ++i) breg[i] = img_units[unit(i, cn)];
            const double *stage = lds64 + s * CH64;
#pragma unroll
            for (int blk = 0; blk < 16; ++blk) {
                const raw_t r = raw[blk % DEPTH];
                const int nb = min((c - c_begin) * 16 + blk + DEPTH, n_blocks - 1);
                raw[blk % DEPTH] = __builtin_nontemporal_load((const raw_t *)(rowp + px0 + (int64_t)nb * 16));
                const f64x2 b0 = rd_b(stage, blk, 0), b1 = rd_b(stage, blk, 1);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)r[0], b0[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)r[1], b0[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)r[2], b1[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)r[3], b1[1], acc, 0, 0, 0);
            }
            u32x4_ *ldsn = (u32x4_ *)(lds64 + (s ^ 1) * CH64);
#pragma unroll
            for (int i = 0; i < BUNITS; ++i) ldsn[i * NT + tid] = breg[i];
            __syncthreads();
        }
    }

    // chunks that need guarded element loads: the ragged last chunk, or everything if unaligned
    for (int c = max(c_begin, n_full); c < c_end; ++c) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < BUNITS; ++i) ((u32x4_ *)lds64)[i * NT + tid] = img_units[unit(i, c)];
        __syncthreads();
#pragma unroll
        for (int blk = 0; blk < 16; ++blk) {
            const int64_t p0 = (int64_t)c * KC64 + blk * 16;          // + kg*4 is folded into rowp
            double a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = (p0 + kg * 4 + j < n_px) ? (double)rowp[p0 + j] : 0.;
            const f64x2 b0 = rd_b(lds64, blk, 0), b1 = rd_b(lds64, blk, 1);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b0[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b0[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b1[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b1[1], acc, 0, 0, 0);
        }
    }

    // C/D layout of 16x16x4 f64 (differs from the f32 instruction): col = lane & 15,
    // row = reg * 4 + (lane >> 4)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t fr = f_wave + r * 4 + kg;
        const int col = g0 * 16 + m;
        if (fr < n_frames && col < n_cols) {
            if (ksplit == 1) {
                double *p = out + fr * ld_out + col;
                *p = accumulate ? (*p + acc[r]) : acc[r];
            } else {
                partials[((int64_t)ks * n_frames + fr) * n_cols + col] = acc[r];
            }
        }
    }
}

// ---- frames through LDS with full-line LDS-DMA (k_dense_lds64) ---------------------------------------
// The f64 twin of k_dense_lds (ltmi_dense.hip) for every pixel size (int32 / uint32 / float32 /
// int64 / uint64 / float64 -- whose results are float64 in the reference -- and 1- / 2-byte pixels
// with float64 masks): 4 waves of
// 32 frames (two 16-frame tiles), per wave a ring of 3 sub-chunk slots (32 rows x 256 B) filled by
// global_load_lds_dwordx4 with the 16-B pieces of a row stored at piece ^ (row & 15); the 32-KiB mask
// chunks (the image of the direct-load kernel above: 256 px x 16 columns of f64) arrive by LDS-DMA
// too, double buffered, one s_barrier per chunk; counted vmcnt waits.  160 KiB of LDS.
typedef __attribute__((address_space(3))) void *lds_ptr64_t;
typedef const __attribute__((address_space(1))) void *glb_ptr64_t;

template <int I, int N, typename F> __device__ __forceinline__ void static_for64(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for64<I + 1, N>(f);
    }
}

struct Lds64Cfg {
    static constexpr int TILES = 2, WAVES = 4, RING = 3;
    static constexpr int SUB_BYTES = 256;                       // bytes of a row per sub-chunk
    static constexpr int ROWS = 16 * TILES;
    static constexpr int TSLOT = 16 * SUB_BYTES;                // one tile of a ring slot
    static constexpr int ASLOT = ROWS * SUB_BYTES;              // 8 KiB per wave and ring slot
    static constexpr int BSLOT = CH64 * 8;                      // 32 KiB per mask chunk
    static constexpr int WG_ROWS = WAVES * ROWS;                // 128 frames per workgroup
    static constexpr int LDS_BYTES = RING * WAVES * ASLOT + 2 * BSLOT;     // 160 KiB
};

template <typename T>
__global__ void __launch_bounds__(256)
k_dense_lds64(const T *__restrict__ tile, int64_t ld, int64_t n_frames, int64_t n_px,
              const double *__restrict__ img, int n_chunks, double *__restrict__ out,
              int64_t ld_out, int n_cols, int accumulate, double *__restrict__ partials,
              int ksplit,
              const int32_t *__restrict__ rows = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw64[];
    using CFG = Lds64Cfg;
    static_assert(sizeof(T) == 1 || sizeof(T) == 2 || sizeof(T) == 4 || sizeof(T) == 8, "pixel size");
    constexpr int TILES = CFG::TILES, WAVES = CFG::WAVES, RING = CFG::RING;
    constexpr int ASLOT = CFG::ASLOT, BSLOT = CFG::BSLOT, SUB = CFG::SUB_BYTES;
    constexpr int ND = 4 * TILES;                       // DMA instructions per sub-chunk
    constexpr int SPX = SUB / (int)sizeof(T);           // pixels per sub-chunk (64 / 32)
    constexpr int PER = KC64 / SPX;                     // sub-chunks per mask chunk (4 / 8)
    constexpr int BLKS = SPX / 16;                      // 16-pixel MFMA blocks per sub-chunk (4 / 2)
    constexpr int NT = WAVES * 64;
    constexpr int A_BYTES = RING * WAVES * ASLOT;
    constexpr int BPW = BSLOT / WAVES;
    constexpr int NBI = BPW / 1024;
    constexpr int A_N = ND * (RING - 2);
    static_assert(A_N + 2 * NBI < 64, "vmcnt is a 6-bit counter");
    typedef T __attribute__((ext_vector_type(sizeof(T) >= 4 ? 16 / sizeof(T) : 4))) unit_t;
    // (4- / 8-byte pixels: one 16-B piece; 1- / 2-byte pixels: this lane's 4 pixels)

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int m = lane & 15, kg = lane >> 4;
    const int ks = blockIdx.y, g0 = blockIdx.z;

    const int n_full = (int)(n_px / KC64);
    const int per = (n_chunks + ksplit - 1) / ksplit;
    const int c_begin = ks * per;
    const int c_end = min(n_chunks, c_begin + per);
    const int cf_end = min(c_end, n_full);
    const double *img_t = img + (size_t)g0 * n_chunks * CH64;

    const int64_t f_wave = (int64_t)blockIdx.x * CFG::WG_ROWS + wave * CFG::ROWS;
    unsigned char *a_base = lds_raw64 + wave * ASLOT;
    unsigned char *b_base = lds_raw64 + A_BYTES;

    f64x4 acc[TILES][2];
#pragma unroll
    for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
        for (int x = 0; x < 2; ++x) acc[tl][x] = f64x4{0., 0., 0., 0.};

    const int a_lane = m * SUB;
    const int b_lane = m * KC64;                         // doubles inside a mask chunk
    auto b_unit = [&](int blk_in_chunk, int h) {         // double offset of the swizzled 16-B unit
        return ((blk_in_chunk * 8 + kg * 2 + h) ^ m) << 1;
    };

    if (c_begin < cf_end) {
        const unsigned char *src[ND];
#pragma unroll
        for (int t = 0; t < ND; ++t) {
            const int r = 4 * t + (lane >> 4);
            int64_t f = f_wave + r;
            if (f > n_frames - 1) f = n_frames - 1;      // clamp: loads stay valid, result discarded
            if (rows) f = rows[f];                       // a region of interest: result row i = frame rows[i]
            const int piece = (lane & 15) ^ (r & 15);
            src[t] = (const unsigned char *)(tile + f * ld) + piece * 16;
        }
        const unsigned char *bsrc = (const unsigned char *)img_t + wave * BPW + lane * 16;
        const int S0 = c_begin * PER, S1 = cf_end * PER;

        auto issue_a1 = [&](int s, int slot, int t) {
            const int sc = min(s, S1 - 1);
            unsigned char *dst = a_base + slot * (WAVES * ASLOT);
            __builtin_amdgcn_global_load_lds((glb_ptr64_t)(src[t] + (int64_t)sc * SUB),
                                             (lds_ptr64_t)(dst + t * 1024), 16, 0, 2 /*nt*/);
        };
        auto issue_b = [&](int gidx) {
            const int cc = min(c_begin + gidx, cf_end - 1);
            unsigned char *dst = b_base + (gidx & 1) * BSLOT + wave * BPW;
            const unsigned char *sp = bsrc + (int64_t)cc * BSLOT;
#pragma unroll
            for (int u = 0; u < NBI; ++u)
                __builtin_amdgcn_global_load_lds((glb_ptr64_t)(sp + u * 1024),
                                                 (lds_ptr64_t)(dst + u * 1024), 16, 0, 0);
        };

#pragma unroll
        for (int t = 0; t < ND; ++t) issue_a1(S0, 0, t);
        issue_b(0);
#pragma unroll
        for (int d = 1; d < RING - 1; ++d)
#pragma unroll
            for (int t = 0; t < ND; ++t) issue_a1(S0 + d, d, t);

        // one sub-chunk; the waits follow the DMA issue order exactly as in k_dense_lds
        auto iteration = [&](int s, auto ph) {
            constexpr int PH = decltype(ph)::value;
            const int i = s - S0;
            const int ip = PH >= 0 ? PH % PER : i % PER;
            if (ip == 0) {
                constexpr int N0 = ND * (PER < RING - 2 ? PER : RING - 2);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N0) : "memory");
                __builtin_amdgcn_s_barrier();
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
            const int blk0 = ip * BLKS;                 // 16-px block offset inside the mask chunk
            const unsigned char *aslot = a_base + slot * (WAVES * ASLOT) + a_lane;
            const double *bslot = (const double *)(b_base + bsl * BSLOT) + b_lane;
            // this lane's 4 pixels (kg*4 .. kg*4+3) of block blk, tile tl, as doubles
            auto rd_a = [&](int tl, int blk, double (&a)[4]) {
                const unsigned char *at = aslot + tl * CFG::TSLOT;
                const int u = blk * 4 + kg;
                if constexpr (sizeof(T) == 4) {
                    const unit_t r = *(const unit_t *)(at + ((u ^ m) << 4));
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] = (double)r[j];
                } else if constexpr (sizeof(T) == 2) {
                    // 8 bytes of piece u / 2 (two lanes of a kg pair share a piece)
                    const unit_t r = *(const unit_t *)(at + (((u >> 1) ^ m) << 4) + (u & 1) * 8);
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] = (double)r[j];
                } else if constexpr (sizeof(T) == 1) {
                    const unit_t r = *(const unit_t *)(at + (((u >> 2) ^ m) << 4) + (u & 3) * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] = (double)r[j];
                } else {
                    const unit_t r0 = *(const unit_t *)(at + (((2 * u) ^ m) << 4));
                    const unit_t r1 = *(const unit_t *)(at + (((2 * u + 1) ^ m) << 4));
                    a[0] = (double)r0[0]; a[1] = (double)r0[1];
                    a[2] = (double)r1[0]; a[3] = (double)r1[1];
                }
            };
            auto rd_b = [&](int blk, int h) {
                return *(const f64x2 *)(bslot + b_unit(blk0 + blk, h));
            };
            double a_c[TILES][4];
#pragma unroll
            for (int tl = 0; tl < TILES; ++tl) rd_a(tl, 0, a_c[tl]);
            f64x2 b_c[2] = {rd_b(0, 0), rd_b(0, 1)};
#pragma unroll
            for (int blk = 0; blk < BLKS; ++blk) {
                double a_n[TILES][4];
                f64x2 b_n[2] = {b_c[0], b_c[1]};
#pragma unroll
                for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
                    for (int j = 0; j < 4; ++j) a_n[tl][j] = a_c[tl][j];
                if (blk + 1 < BLKS) {
#pragma unroll
                    for (int tl = 0; tl < TILES; ++tl) rd_a(tl, blk + 1, a_n[tl]);
                    b_n[0] = rd_b(blk + 1, 0);
                    b_n[1] = rd_b(blk + 1, 1);
                }
#pragma unroll
                for (int t = 0; t < ND; ++t)
                    if ((t * BLKS) / ND == blk) issue_a1(s + RING - 1, nslot, t);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int tl = 0; tl < TILES; ++tl)
                        acc[tl][j & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                            a_c[tl][j], b_c[j >> 1][j & 1], acc[tl][j & 1], 0, 0, 0);
#pragma unroll
                for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
                    for (int j = 0; j < 4; ++j) a_c[tl][j] = a_n[tl][j];
                b_c[0] = b_n[0];
                b_c[1] = b_n[1];
            }
        };

        constexpr int U2 = 2 * PER;
        constexpr int UNROLL = (U2 % RING == 0) ? U2 : RING * U2;
        int s = S0;
        if constexpr (UNROLL <= 48) {
            for (; s + UNROLL <= S1; s += UNROLL) {
                static_for64<0, UNROLL>([&](auto I) { iteration(s + decltype(I)::value, I); });
            }
        }
        for (; s < S1; ++s) iteration(s, std::integral_constant<int, -1>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // ragged last chunk (n_px % 256 != 0): guarded element loads, mask chunk staged by plain copies
    if (c_end > n_full) {
        const int c = n_full;
        __syncthreads();
        double *bl = (double *)b_base;
        const u32x4_ *img_units = (const u32x4_ *)(img_t + (size_t)c * CH64);
#pragma unroll
        for (int i = 0; i < BSLOT / 16 / NT; ++i)
            ((u32x4_ *)bl)[i * NT + tid] = img_units[i * NT + tid];
        __syncthreads();
        const double *ldsb = bl + b_lane;
#pragma unroll
        for (int tl = 0; tl < TILES; ++tl) {
            int64_t f = f_wave + tl * 16 + m;
            if (f > n_frames - 1) f = n_frames - 1;
            if (rows) f = rows[f];
            const T *rowp = tile + f * ld + kg * 4;
#pragma unroll
            for (int blk = 0; blk < KC64 / 16; ++blk) {
                const int64_t p0 = (int64_t)c * KC64 + blk * 16;
                double a[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = (p0 + kg * 4 + j < n_px) ? (double)rowp[p0 + j] : 0.;
                const f64x2 b0 = *(const f64x2 *)(ldsb + b_unit(blk, 0));
                const f64x2 b1 = *(const f64x2 *)(ldsb + b_unit(blk, 1));
                acc[tl][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b0[0], acc[tl][0], 0, 0, 0);
                acc[tl][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b0[1], acc[tl][1], 0, 0, 0);
                acc[tl][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b1[0], acc[tl][0], 0, 0, 0);
                acc[tl][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b1[1], acc[tl][1], 0, 0, 0);
            }
        }
    }

    // C/D layout of 16x16x4 f64: col = lane & 15, row = reg * 4 + (lane >> 4)
#pragma unroll
    for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t fr = f_wave + tl * 16 + r * 4 + kg;
            const int col = g0 * 16 + m;
            if (fr < n_frames && col < n_cols) {
                const double v = acc[tl][0][r] + acc[tl][1][r];
                if (ksplit == 1) {
                    double *p = out + fr * ld_out + col;
                    *p = accumulate ? (*p + v) : v;
                } else {
                    partials[((int64_t)ks * n_frames + fr) * n_cols + col] = v;
                }
            }
        }
}

__global__ void k_reduce_partials64(const double *__restrict__ partials, int ksplit,
                                    int64_t n_frames, int n_cols, double *__restrict__ out,
                                    int64_t ld_out, int accumulate) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_frames * n_cols) return;
    const int64_t f = idx / n_cols;
    const int col = (int)(idx % n_cols);
    double *p = out + f * ld_out + col;
    double s = accumulate ? *p : 0.;
    const int64_t stride = n_frames * n_cols;            // (independent loads, added in the order of k: see k_reduce_partials)
    for (int k0 = 0; k0 < ksplit; k0 += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partials[(int64_t)min(k0 + u, ksplit - 1) * stride + idx];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + u < ksplit) s += v[u];
    }
    *p = s;
}

// ---- host side ---------------------------------------------------------------------------------------
// exact integer sum (held in a double, |v| < 2^53) -> wrap-around integer of the result width,
// the arithmetic NumPy's integer matmul performs (two's complement truncation)
template <typename S>
__global__ void k_f64_to_int(const double *__restrict__ src, int64_t n_frames, int n_masks,
                             S *__restrict__ out, int64_t ld_out, int accumulate) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_frames * n_masks) return;
    const int64_t f = idx / n_masks;
    const int k = (int)(idx % n_masks);
    const S v = (S)(uint64_t)(int64_t)src[idx];
    S *p = out + f * ld_out + k;
    *p = accumulate ? (S)(*p + v) : v;
}

int dense64_create(ltmi_masks *m) {
    // m->gmasks holds the (n_masks, n_px) stack on the device: float64, or int64 for integer
    // result dtypes (then the image is only built if the masks are small, see dense64_apply)
    m->cpm64 = (m->result_dtype == LTMI_C128) ? 2 : 1;
    m->n_groups64 = (int)((m->n_masks * m->cpm64 + 15) / 16);
    m->n_chunks64 = (int)((m->n_px + KC64 - 1) / KC64);
    const size_t n = (size_t)m->n_groups64 * m->n_chunks64 * CH64;
    LTMI_HIP(hipMalloc((void **)&m->img64, n * sizeof(double)));
    LTMI_HIP(hipMemset(m->img64, 0, n * sizeof(double)));
    const int64_t total = m->n_masks * m->n_px * m->cpm64;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 65535 * 16);
    if (m->result_dtype == LTMI_F64 || m->result_dtype == LTMI_C128)
        hipLaunchKernelGGL(k_build_image64<double>, dim3(blocks), dim3(256), 0, 0,
                           (const double *)m->gmasks, m->img64, m->n_masks, m->n_px, m->n_chunks64,
                           m->cpm64);
    else
        hipLaunchKernelGGL(k_build_image64<int64_t>, dim3(blocks), dim3(256), 0, 0,
                           (const int64_t *)m->gmasks, m->img64, m->n_masks, m->n_px,
                           m->n_chunks64, 1);
    LTMI_HIP(hipGetLastError());
    LTMI_HIP(hipDeviceSynchronize());
    return LTMI_OK;
}

size_t dense64_image_bytes(const ltmi_masks *m) {
    return (size_t)m->n_groups64 * m->n_chunks64 * CH64 * sizeof(double);
}

// `img`: dense64_image_bytes(m) of device memory whose padding (columns past the stack, pixels past the
// frame) is zero; asynchronous on `stream`
int dense64_build_shifted(ltmi_masks *m, int sig_h, int sig_w, int dy, int dx, double *img,
                          hipStream_t stream) {
    const int64_t total = m->n_masks * m->n_px * m->cpm64;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 65535 * 16);
    if (m->result_dtype == LTMI_F64 || m->result_dtype == LTMI_C128)
        hipLaunchKernelGGL(k_build_image64_shifted<double>, dim3(blocks), dim3(256), 0, stream,
                           (const double *)m->gmasks, img, m->n_masks, m->n_px, m->n_chunks64, m->cpm64,
                           sig_h, sig_w, dy, dx);
    else
        hipLaunchKernelGGL(k_build_image64_shifted<int64_t>, dim3(blocks), dim3(256), 0, stream,
                           (const int64_t *)m->gmasks, img, m->n_masks, m->n_px, m->n_chunks64, 1, sig_h,
                           sig_w, dy, dx);
    LTMI_HIP(hipGetLastError());
    return LTMI_OK;
}

void dense64_destroy(ltmi_masks *m) {
    if (m->img64) (void)hipFree(m->img64);
    if (m->ws64) (void)hipFree(m->ws64);
    if (m->res64) (void)hipFree(m->res64);
    m->img64 = nullptr;
    m->ws64 = nullptr;
    m->res64 = nullptr;
}

static inline int64_t n_cols64(const ltmi_masks *m) { return m->n_masks * m->cpm64; }

static int ensure_ws64(ltmi_masks *m, size_t need, hipStream_t stream) {
    if (m->ws64_bytes < need) {
        if (m->ws64) {
            LTMI_HIP(hipStreamSynchronize(stream));
            LTMI_HIP(hipFree(m->ws64));
            m->ws64 = nullptr;
            m->ws64_bytes = 0;
        }
        LTMI_HIP(hipMalloc(&m->ws64, need));
        m->ws64_bytes = need;
    }
    return LTMI_OK;
}

// at least one full mask chunk: the LDS-DMA kernel
template <typename T>
static int launch64_lds(ltmi_masks *m, const T *tile, int64_t n_frames, int64_t ld, double *out,
                        int64_t ld_out, int accumulate, hipStream_t stream) {
    using CFG = Lds64Cfg;
    auto kern = k_dense_lds64<T>;
    static bool attr_set[16] = {false};
    if (!attr_set[m->device & 15]) {
        LTMI_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     CFG::LDS_BYTES));
        attr_set[m->device & 15] = true;
    }
    const int64_t gx = (n_frames + CFG::WG_ROWS - 1) / CFG::WG_ROWS;
    const int64_t gz = m->n_groups64;
    int ksplit = m->tune_ksplit;
    if (ksplit <= 0) ksplit = choose_ksplit(gx * gz, m->n_chunks64);
    ksplit = std::max(1, std::min(ksplit, m->n_chunks64));
    {
        const int per = (m->n_chunks64 + ksplit - 1) / ksplit;
        ksplit = (m->n_chunks64 + per - 1) / per;
    }
    if (ksplit > 1) {
        int rc = ensure_ws64(m, (size_t)ksplit * n_frames * n_cols64(m) * sizeof(double), stream);
        if (rc != LTMI_OK) return rc;
    }
    dim3 grid((unsigned)gx, (unsigned)ksplit, (unsigned)gz);
    hipLaunchKernelGGL(kern, grid, dim3(CFG::WAVES * 64), CFG::LDS_BYTES, stream, tile, ld, n_frames,
                       m->n_px, (const double *)m->img64, m->n_chunks64, out, ld_out,
                       (int)n_cols64(m), accumulate, (double *)m->ws64, ksplit, m->roi_rows);
    LTMI_HIP(hipGetLastError());
    snprintf(m->last_kernel, sizeof(m->last_kernel), "k_dense_lds64<%s%s> grid=(%u,%u,%u)",
             typeid(T).name(), m->roi_rows ? ",rows" : "", grid.x, grid.y, grid.z);
    if (ksplit > 1) {
        const int64_t n = n_frames * n_cols64(m);
        hipLaunchKernelGGL(k_reduce_partials64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           stream, (const double *)m->ws64, ksplit, n_frames, (int)n_cols64(m), out,
                           ld_out, accumulate);
        LTMI_HIP(hipGetLastError());
    }
    return LTMI_OK;
}

// integer masks x integer frames: every possible partial sum fits 2^52, the f64 FMA chain is exact
static bool int_product_is_exact(const ltmi_masks *m, int tile_dtype) {
    int data_bits;
    switch (tile_dtype) {
        case LTMI_BOOL: data_bits = 1; break;
        case LTMI_U8: case LTMI_I8: data_bits = 8; break;
        case LTMI_U16: case LTMI_I16: data_bits = 16; break;
        case LTMI_U32: case LTMI_I32: data_bits = 32; break;
        default: return false;                            // 64-bit or non-integer tiles
    }
    int k_bits = 0;
    while (((int64_t)1 << k_bits) < m->n_px) ++k_bits;
    return data_bits + m->mask_bits + k_bits <= 52;
}

// can this handle read the frames of a region of interest through a row list (ltmi_apply_masks_rows)?
// float64 / complex128 / exact-integer results through the LDS-DMA kernel (mirrors the dispatch of
// dense64_apply and launch64)
bool dense64_rows_ok(const ltmi_masks *m, const void *tile, int tile_dtype, int64_t ld) {
    if (!m->img64) return false;
    const bool int_result = m->result_dtype >= LTMI_U8 && m->result_dtype <= LTMI_I64;
    if (int_result ? !int_product_is_exact(m, tile_dtype)
                   : (m->result_dtype != LTMI_F64 && m->result_dtype != LTMI_C128)) return false;
    if (tile_dtype == LTMI_C64 || tile_dtype == LTMI_C128 || dtype_size(tile_dtype) == 0) return false;
    return m->tune_mt != 1 && m->n_px >= KC64 &&
           vector_loads_ok(tile, ld, (size_t)dtype_size(tile_dtype));
}

template <typename T>
static int launch64(ltmi_masks *m, const T *tile, int64_t n_frames, int64_t ld, double *out,
                    int64_t ld_out, int accumulate, hipStream_t stream) {
    {
        // tune_mt == 1 (ltmi_masks_set_tuning): force the direct-load kernel (bench comparison)
        // (rows need not be 16-B aligned: LDS-DMA reads from any element-aligned address)
        if (m->tune_mt != 1 && m->n_px >= KC64 && vector_loads_ok(tile, ld, sizeof(T)))
            return launch64_lds<T>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
    }
    if (m->roi_rows) LTMI_FAIL(LTMI_E_INVALID, "k_dense_mfma_f64 does not take a row list");
    constexpr int WAVES = 4;
    const bool vec = (((uintptr_t)tile) % (4 * sizeof(T)) == 0) && (ld % 4 == 0);
    const int64_t gx = (n_frames + WAVES * 16 - 1) / (WAVES * 16);
    const int64_t gz = m->n_groups64;
    int ksplit = m->tune_ksplit;
    if (ksplit <= 0) {
        ksplit = 1;
        if (gx * gz < 512)
            ksplit = (int)std::min<int64_t>((1024 + gx * gz - 1) / (gx * gz),
                                            std::max(1, m->n_chunks64 / 8));
    }
    ksplit = std::max(1, std::min(ksplit, m->n_chunks64));
    {
        const int per = (m->n_chunks64 + ksplit - 1) / ksplit;
        ksplit = (m->n_chunks64 + per - 1) / per;
    }
    if (ksplit > 1) {
        const size_t need = (size_t)ksplit * n_frames * n_cols64(m) * sizeof(double);
        if (m->ws64_bytes < need) {
            if (m->ws64) {
                LTMI_HIP(hipStreamSynchronize(stream));
                LTMI_HIP(hipFree(m->ws64));
                m->ws64 = nullptr;
                m->ws64_bytes = 0;
            }
            LTMI_HIP(hipMalloc(&m->ws64, need));
            m->ws64_bytes = need;
        }
    }
    const size_t lds = 2 * (size_t)CH64 * sizeof(double);          // 64 KiB
    auto kern = vec ? k_dense_mfma_f64<T, WAVES, true> : k_dense_mfma_f64<T, WAVES, false>;
    LTMI_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
    dim3 grid((unsigned)gx, (unsigned)ksplit, (unsigned)gz);
    hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, stream, tile, ld, n_frames, m->n_px,
                       (const double *)m->img64, m->n_chunks64, out, ld_out, (int)n_cols64(m),
                       accumulate, (double *)m->ws64, ksplit);
    LTMI_HIP(hipGetLastError());
    snprintf(m->last_kernel, sizeof(m->last_kernel), "k_dense_mfma_f64<%s,%s> grid=(%u,%u,%u)",
             typeid(T).name(), vec ? "vec" : "guarded", grid.x, grid.y, grid.z);
    if (ksplit > 1) {
        const int64_t n = n_frames * n_cols64(m);
        hipLaunchKernelGGL(k_reduce_partials64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           stream, (const double *)m->ws64, ksplit, n_frames, (int)n_cols64(m), out,
                           ld_out, accumulate);
        LTMI_HIP(hipGetLastError());
    }
    return LTMI_OK;
}

// -> LTMI_OK and *handled = true if the tile went through the f64 matrix kernel
int dense64_apply(ltmi_masks *m, const void *tile, int tile_dtype, int64_t n_frames, int64_t ld,
                  void *out, int64_t ld_out, int accumulate, hipStream_t stream, bool *handled) {
    *handled = false;
    if (!m->img64) return LTMI_OK;
    const bool int_result = m->result_dtype >= LTMI_U8 && m->result_dtype <= LTMI_I64;
    if (!int_result && m->result_dtype != LTMI_F64 && m->result_dtype != LTMI_C128) return LTMI_OK;
    // complex128 masks on REAL frames: 2 real f64 columns per mask, the result row is the interleaved
    // complex128 row (complex frames stay with the generic kernel: `default` below)
    double *o = (double *)out;
    int64_t ld_o = ld_out * m->cpm64;
    if (int_result) {
        // Integer masks x integer frames (preferred_dtype / mask_dtype integer: NumPy integer matmul,
        // wrap-around).  If every possible partial sum fits 2^52 the f64 FMA chain is EXACT, so the
        // product runs on the f64 matrix cores into a scratch buffer and is then truncated to the
        // result width -- bit-identical to integer arithmetic.  Otherwise: the integer VALU kernel.
        if (!int_product_is_exact(m, tile_dtype)) return LTMI_OK;
        const size_t need = (size_t)n_frames * m->n_masks * sizeof(double);
        if (m->res64_bytes < need) {
            if (m->res64) {
                LTMI_HIP(hipStreamSynchronize(stream));
                LTMI_HIP(hipFree(m->res64));
                m->res64 = nullptr;
                m->res64_bytes = 0;
            }
            LTMI_HIP(hipMalloc(&m->res64, need));
            m->res64_bytes = need;
        }
        o = (double *)m->res64;
        ld_o = m->n_masks;
    }
    const int acc64 = int_result ? 0 : accumulate;
    int rc;
    switch (tile_dtype) {
        case LTMI_BOOL:
        case LTMI_U8: rc = launch64<uint8_t>(m, (const uint8_t *)tile, n_frames, ld, o, ld_o, acc64, stream); break;
        case LTMI_I8: rc = launch64<int8_t>(m, (const int8_t *)tile, n_frames, ld, o, ld_o, acc64, stream); break;
        case LTMI_U16: rc = launch64<uint16_t>(m, (const uint16_t *)tile, n_frames, ld, o, ld_o, acc64, stream); break;
        case LTMI_I16: rc = launch64<int16_t>(m, (const int16_t *)tile, n_frames, ld, o, ld_o, acc64, stream); break;
        case LTMI_U32: rc = launch64<uint32_t>(m, (const uint32_t *)tile, n_frames, ld, o, ld_o, acc64, stream); break;
        case LTMI_I32: rc = launch64<int32_t>(m, (const int32_t *)tile, n_frames, ld, o, ld_o, acc64, stream); break;
        case LTMI_U64: rc = launch64<uint64_t>(m, (const uint64_t *)tile, n_frames, ld, o, ld_o, acc64, stream); break;
        case LTMI_I64: rc = launch64<int64_t>(m, (const int64_t *)tile, n_frames, ld, o, ld_o, acc64, stream); break;
        case LTMI_F32: rc = launch64<float>(m, (const float *)tile, n_frames, ld, o, ld_o, acc64, stream); break;
        case LTMI_F64: rc = launch64<double>(m, (const double *)tile, n_frames, ld, o, ld_o, acc64, stream); break;
        default: return LTMI_OK;           // complex tiles: generic kernel
    }
    if (rc != LTMI_OK) return rc;
    if (int_result) {
        const int64_t n = n_frames * m->n_masks;
        const dim3 grid((unsigned)((n + 255) / 256));
        const double *src = (const double *)m->res64;
        switch (dtype_size(m->result_dtype)) {
            case 1: hipLaunchKernelGGL(k_f64_to_int<uint8_t>, grid, dim3(256), 0, stream, src, n_frames, (int)m->n_masks, (uint8_t *)out, ld_out, accumulate); break;
            case 2: hipLaunchKernelGGL(k_f64_to_int<uint16_t>, grid, dim3(256), 0, stream, src, n_frames, (int)m->n_masks, (uint16_t *)out, ld_out, accumulate); break;
            case 4: hipLaunchKernelGGL(k_f64_to_int<uint32_t>, grid, dim3(256), 0, stream, src, n_frames, (int)m->n_masks, (uint32_t *)out, ld_out, accumulate); break;
            default: hipLaunchKernelGGL(k_f64_to_int<uint64_t>, grid, dim3(256), 0, stream, src, n_frames, (int)m->n_masks, (uint64_t *)out, ld_out, accumulate); break;
        }
        LTMI_HIP(hipGetLastError());
        const size_t len = strlen(m->last_kernel);
        snprintf(m->last_kernel + len, sizeof(m->last_kernel) - len, " exact-int");
    }
    *handled = true;
    return LTMI_OK;
}

}  // namespace ltmi
