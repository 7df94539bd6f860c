// Row-mirror fold of dense stacks: k_dense_fold and the folded image of a handle (ltmi_masks::fold).
//
// Virtual-detector stacks are often symmetric under the reflection of the detector rows about the centre the
// masks were built around: radial-Fourier masks ring(r) * exp(i o phi) (src/libertem/analysis/radialfourier.py:
// 106-146; phi = arctan2(dy, dx), utils/__init__.py:41-44) have real parts that are EVEN and imaginary parts that
// are ODD under dy -> -dy, bit for bit (arctan2, cos and sin of the math library are exactly odd / even); rings,
// disks and the centre-of-mass ramps (udf/com.py:47-97) likewise.  For a column c that is even (s = +1) or odd
// (s = -1) under y -> y' = c2 - y:
//
//     sum_p x[p] w_c[p]  =  sum over row pairs (y < y') and x of  (x[y, x] + s x[y', x]) * w_c[y, x]   (+ unpaired rows)
//
// with the ORIGINAL float32 weights of the rows y -- nothing is averaged, so a frame with a single non-zero pixel
// gives exactly pixel * weight like the unfolded product.  The matrix cores then see half of the pixels: the even
// columns against e = a + c, the odd ones against o = a - c (two v_pk_add_f32 per pixel pair of a lane).  C5
// (25 complex masks on 1024 x 1024 float32 frames): 2 + 2 groups of 16 columns over 513 x 1024 folded pixels
// instead of 3 groups + 2 VALU columns over 1024 x 1024 -- the matrix work drops from 24 (+ VALU) to 16
// v_mfma_f32_16x16x4_f32 per 32 pixels and frame tile and the kernel moves from the matrix pipe to the HBM.
// (The reflection of the columns, dx -> -dx, is NOT exact for these masks -- arctan2(dy, -dx) = pi - arctan2(dy, dx)
// rounds -- so only the row mirror is used; a stack in which one column is neither even nor odd is not folded.)
//
// Kernel: the frame layout, piece swizzle and fragment reads of k_dense_lds<float> (4 waves x 32 frames, LDS-DMA of
// 4 rows x 256 B per instruction); a STAGE is 64 folded pixels = 256 B of detector row y ("A" part) and of row y'
// ("C" part; unpaired rows: a zero page) of the wave's frames, and it is copied and multiplied in two HALF-stages
// (one 16-frame tile each): the LDS holds four half-stages of 32 KiB -- one being multiplied, three on their way
// (96 KiB per CU in flight; whole stages would leave room for one in flight only: 7.4 ms per 8192 frames of C5,
// one 64-KiB stage per 3.6 us; HBM -> registers -> LDS with two stages in flight in the register file: 8.9 ms,
// the stores and their waits stand in the matrix instructions' way) -- and the mask slot of a stage (NGE + NGO
// groups x 16 columns x 64 pixels, <= 16 KiB) double buffered behind them; one s_barrier per stage, counted waits.
#include "ltmi_common.h"
#include <vector>
#include <cstdlib>
#include <algorithm>
#include <exception>
#include <climits>
#include <cstdint>
#include <type_traits>
#include <typeinfo>

namespace ltmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;
constexpr int GROUP = 16;                  // mask columns per MFMA group (N of 16x16x4)


constexpr int FD_KB = 64;                            // folded pixels per stage / mask slot
constexpr int FD_WAVES = 4, FD_TILES = 2;
constexpr int FD_ROWS = 16 * FD_TILES;               // frames per wave
constexpr int FD_TPART = 16 * 256;                   // 4 KiB: rows y (or y') of one 16-frame tile of a wave, one stage
constexpr int FD_HALF = 2 * FD_WAVES * FD_TPART;     // 32 KiB: a half-stage = one tile of every wave, parts A and C
constexpr int FD_RING = 4;                           // half-stages in LDS: one being multiplied, three on their way
constexpr int FD_WG_ROWS = FD_WAVES * FD_ROWS;       // 128 frames per workgroup
constexpr bool FOLD_TURN_DEFAULT = false;           // stages of a pixel-split launch in turn (LTMI_FOLD_TURN=1 / 0)
constexpr bool FOLD8_DEFAULT = false;                // k_dense_fold8 (two waves per SIMD) where it applies: see launch_fold_t
__host__ __device__ constexpr int fold_slot_bytes(int ng) { return ng * GROUP * FD_KB * 4; }
__host__ __device__ constexpr int fold_lds_bytes(int ng) { return FD_RING * FD_HALF + 2 * fold_slot_bytes(ng); }

// Position of (column n of a group, pixel q of the 64-pixel slot) inside the group's 4 KiB: rows of 256 B (one
// per column) hold 16 units of 16 B; lane (n, kg) reads unit (kg, c = 2 blk + h).  All rows start on the same
// bank, so the 16 lanes of a ds_read_b128 service group -- {n 0-3, 12-15 of kg} + {n 4-11 of kg ^ 1} -- must land
// on 16 different units: the high bits are kg ^ g(n >> 2) with g = (0, 3, 2, 1), the low bits c ^ (n & 3).
__host__ __device__ static inline int fold_index(int n, int q) {
    const int blk = q >> 5, kg = (q >> 3) & 3, j = q & 7;
    const int hi = kg ^ ((4 - (n >> 2)) & 3);
    const int lo = (blk * 2 + (j >> 2)) ^ (n & 3);
    return n * FD_KB + (hi * 4 + lo) * 4 + (j & 3);
}

// raw stack -> fold image: virtual column v of the folded stack is real column colmap[v] (-1: padding), folded
// pixel pf = fy * sig_w + x is pixel rows[fy].x * sig_w + x of the stack
__global__ void k_build_fold_image(const float *__restrict__ src, float *__restrict__ img, int cpm, int64_t n_px,
                                   int sig_w, int n_fold_rows, const int2 *__restrict__ rows,
                                   const int *__restrict__ colmap, int ng) {
    const int64_t total = (int64_t)ng * GROUP * n_fold_rows * sig_w;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pf = i % ((int64_t)n_fold_rows * sig_w);
        const int v = (int)(i / ((int64_t)n_fold_rows * sig_w));
        const int col = colmap[v];
        const int fy = (int)(pf / sig_w), x = (int)(pf % sig_w);
        float w = 0.f;
        if (col >= 0) {
            const int64_t k = col / cpm, part = col % cpm;
            w = src[(k * n_px + (int64_t)rows[fy].x * sig_w + x) * cpm + part];
        }
        const int64_t slot = pf / FD_KB;
        const int q = (int)(pf % FD_KB);
        img[(slot * ng + v / GROUP) * (GROUP * FD_KB) + fold_index(v % GROUP, q)] = w;
    }
}

// even[col] / odd[col] (preset to 1) are cleared when the column is not even / odd under y -> c2 - y
__global__ void k_fold_classify(const float *__restrict__ src, int cpm, int64_t n_masks, int sig_h, int sig_w,
                                int c2, int *__restrict__ even, int *__restrict__ odd) {
    const int64_t n_px = (int64_t)sig_h * sig_w;
    const int64_t total = n_masks * cpm * n_px;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int part = (int)(i % cpm);
        const int64_t kp = i / cpm;
        const int64_t k = kp / n_px, p = kp % n_px;
        const int y = (int)(p / sig_w), x = (int)(p % sig_w);
        const int y2 = c2 - y;
        if (y2 <= y || y2 >= sig_h) continue;                         // each pair once; unpaired rows: free
        const float a = src[i], b = src[((k * n_px) + (int64_t)y2 * sig_w + x) * cpm + part];
        const int col = (int)(k * cpm + part);
        if (!(a == b)) even[col] = 0;                                  // (NaN: neither)
        if (!(a == -b)) odd[col] = 0;
    }
}

// LIST (banded stacks, see band_build below): the stack is a set of column BLOCKS with a pixel support each (the
// bins of a radial-Fourier stack with several bins); workgroup (x, y) multiplies the frames of x with block y only,
// over the stages that touch the block's support: stage s is (row y, row y' or -1, 64-pixel slot of the rows, -)
// = stage_list[s], s in [blk_off[y], blk_off[y + 1]), its mask slot is slot s of the image, the block's columns are
// colmap[y * NG * 16 ...].  No pixel split (every block writes its own columns).
template <int NGE, int NGO, int ABL = 0, bool LIST = false>
__global__ void __launch_bounds__(FD_WAVES * 64)
k_dense_fold(const float *__restrict__ tile, int64_t ld, int64_t n_frames, int spr /* stages per row */,
             const int2 *__restrict__ fold_rows, const float *__restrict__ img, int n_stages,
             float *__restrict__ out, int64_t ld_out, int n_cols, const int *__restrict__ colmap,
             int accumulate, float *__restrict__ partials, int ksplit,
             const unsigned char *__restrict__ zeros, const int32_t *__restrict__ rows,
             const int4 *__restrict__ stage_list, const int *__restrict__ blk_off) {
    constexpr int NG = NGE + NGO;
    constexpr int BSLOT = fold_slot_bytes(NG);
    constexpr int NBI = BSLOT / FD_WAVES / 1024;          // mask-slot DMA instructions per wave and stage
    constexpr int NDH = 16 / 4;                           // DMA instructions per part of a half-stage (4 rows x 256 B each)
    constexpr int NF = 2 * NDH;                           // ... per half-stage (A and C)
    static_assert(BSLOT % (FD_WAVES * 1024) == 0, "whole DMA instructions per wave");
    static_assert(2 * NF + NBI < 64, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int m = lane & 15, kg = lane >> 4;
    // LIST: blockIdx.y = block * ksplit + part of the block's stage list
    // ksplit < 0 (not LIST): the parts of a pixel-split launch take the stages IN TURN (part ks: stages ks, ks + ksplit,
    // ...) instead of a contiguous range each -- the workgroups of a frame group that run at the same time then read
    // neighbouring 256-byte pieces of the same detector rows.  Stage of loop index v: p0 + v * pstride.
    const bool turn = !LIST && ksplit < 0;
    if (ksplit < 0) ksplit = -ksplit;
    const int by = LIST ? blockIdx.y / ksplit : 0;
    const int ks = LIST ? blockIdx.y - by * ksplit : blockIdx.y;
    const int b0 = LIST ? blk_off[by] : 0, b1 = LIST ? blk_off[by + 1] : n_stages;
    const int per = (b1 - b0 + ksplit - 1) / ksplit;
    const int p0 = turn ? ks : 0, pstride = turn ? ksplit : 1;
    const int s_begin = turn ? 0 : b0 + ks * per;
    const int s_end = turn ? (n_stages - ks + ksplit - 1) / ksplit : min(b1, s_begin + per);
    if (LIST) colmap += by * (NG * GROUP);

    const int64_t f_wave = (int64_t)blockIdx.x * FD_WG_ROWS + wave * FD_ROWS;
    auto frame_of = [&](int r) -> int64_t {                 // result row (-1: none)
        const int64_t f = f_wave + r;
        return f < n_frames ? f : -1;
    };
    auto src_frame_of = [&](int r) -> int64_t {             // frame to read (clamped: loads stay valid)
        int64_t f = frame_of(r);
        if (rows) return f < 0 ? (int64_t)rows[0] : (int64_t)rows[f];
        return f < 0 ? n_frames - 1 : f;
    };

    // ring slot q: the wave's 16 rows of part A (4 KiB), then of part C
    unsigned char *a_base = lds_raw + wave * (2 * FD_TPART);                // + q * FD_HALF
    unsigned char *b_base = lds_raw + FD_RING * FD_HALF;                    // + slot * BSLOT

    f32x4 acc[FD_TILES][NG], acc2[FD_TILES][NG];
#pragma unroll
    for (int tl = 0; tl < FD_TILES; ++tl)
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[tl][g] = acc2[tl][g] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (s_begin < s_end) {
        const unsigned char *src[FD_TILES][NDH];
#pragma unroll
        for (int tl = 0; tl < FD_TILES; ++tl)
#pragma unroll
            for (int t = 0; t < NDH; ++t) {
                const int r = 4 * t + lane / 16;                            // row inside the tile
                const int piece = (lane & 15) ^ r;
                src[tl][t] = (const unsigned char *)(tile + src_frame_of(tl * 16 + r) * ld) + piece * 16;
            }
        const unsigned char *zsrc = zeros + lane * 16;
        const unsigned char *bsrc = (const unsigned char *)img + wave * (BSLOT / FD_WAVES) + lane * 16;

        // half-stages are issued in order: (stage, tile 0), (stage, tile 1), (stage + 1, tile 0) ...; past the end
        // the last stage again (clamped prefetch: every step issues the same number of copies, the waits count them)
        int iss = s_begin, iss_fy = LIST ? 0 : (p0 + s_begin * pstride) / spr, iss_xs = LIST ? 0 : (p0 + s_begin * pstride) % spr;
        int4 iss_st = LIST ? stage_list[s_begin] : int4{0, 0, 0, 0};         // (fetched one stage ahead of its use)
        auto issue_half = [&](auto TL, auto Q) {
            constexpr int tl = decltype(TL)::value, q = decltype(Q)::value;
            if (ABL >= 2) return;
            const int2 rr = LIST ? int2{iss_st.x, iss_st.y} : fold_rows[iss_fy];
            // (LIST: a window starts at pixel iss_st.z of its row, any multiple of 16)
            const int64_t x_px = LIST ? iss_st.z : iss_xs * FD_KB;
            const int64_t off_a = ((int64_t)rr.x * spr * FD_KB + x_px) * 4;
            const int64_t off_c = ((int64_t)rr.y * spr * FD_KB + x_px) * 4;
            const bool pair = rr.y >= 0;
            unsigned char *da = a_base + q * FD_HALF, *dc = da + FD_TPART;
#pragma unroll
            for (int t = 0; t < NDH; ++t)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[tl][t] + off_a), (lds_ptr_t)(da + t * 1024), 16, 0,
                                                 2 /*nt*/);
#pragma unroll
            for (int t = 0; t < NDH; ++t) {
                const unsigned char *p = pair ? src[tl][t] + off_c : zsrc;
                __builtin_amdgcn_global_load_lds((glb_ptr_t)p, (lds_ptr_t)(dc + t * 1024), 16, 0, 2 /*nt*/);
            }
            if (tl == FD_TILES - 1 && iss + 1 < s_end) {
                ++iss;
                if (LIST) iss_st = stage_list[iss];
                else {
                    iss_xs += pstride;
                    while (iss_xs >= spr) { iss_xs -= spr; ++iss_fy; }
                }
            }
        };
        auto issue_b = [&](int s, int bslot) {              // mask slot of stage s (clamped)
            if (ABL >= 2) return;
            unsigned char *db = b_base + bslot * BSLOT + wave * (BSLOT / FD_WAVES);
            const unsigned char *sp = bsrc + (int64_t)(p0 + min(s, s_end - 1) * pstride) * BSLOT;
#pragma unroll
            for (int u = 0; u < NBI; ++u)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(sp + u * 1024), (lds_ptr_t)(db + u * 1024), 16, 0, 0);
        };

        // lane-constant parts of the fragment addresses
        const int a_lane = m * 256;
        const int b_lane = m * FD_KB;                                       // floats
        const int b_hi = (kg ^ ((4 - (m >> 2)) & 3)) * 4;
        auto b_unit = [&](int blk, int h) { return (b_hi + ((blk * 2 + h) ^ (m & 3))) << 2; };

        // Fragments of one 32-pixel block of a half-stage: the lane's 8 pixels of parts A and C, its 8 weights of
        // every group.
        struct FragAC { f32x4 a[2], c[2]; };
        struct FragB { f32x4 b[NG][2]; };
        auto load_ac = [&](FragAC &fr, auto Q, int blk) {
            constexpr int q = decltype(Q)::value;
            const unsigned char *as = a_base + q * FD_HALF + a_lane;
            const unsigned char *cs = as + FD_TPART;
            const int u = blk * 4 + kg;                      // 8-pixel unit of this lane inside the part
            fr.a[0] = *(const f32x4 *)(as + (((2 * u) ^ m) << 4));
            fr.a[1] = *(const f32x4 *)(as + (((2 * u + 1) ^ m) << 4));
            fr.c[0] = *(const f32x4 *)(cs + (((2 * u) ^ m) << 4));
            fr.c[1] = *(const f32x4 *)(cs + (((2 * u + 1) ^ m) << 4));
        };
        auto load_b = [&](FragB &fr, auto BS, int blk) {
            constexpr int bslot = decltype(BS)::value;
            const float *bs = (const float *)(b_base + bslot * BSLOT) + b_lane;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                fr.b[g][0] = *(const f32x4 *)(bs + g * (GROUP * FD_KB) + b_unit(blk, 0));
                fr.b[g][1] = *(const f32x4 *)(bs + g * (GROUP * FD_KB) + b_unit(blk, 1));
            }
        };
        auto mfma_block = [&](const FragAC &fr, const FragB &fb, auto TL) {
            constexpr int tl = decltype(TL)::value;
            float e[8], o[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 ev = fr.a[h] + fr.c[h];
                const f32x4 ov = fr.a[h] - fr.c[h];
#pragma unroll
                for (int i = 0; i < 4; ++i) { e[h * 4 + i] = ev[i]; o[h * 4 + i] = ov[i]; }
            }
            if (ABL == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        acc[tl][g][j & 3] += (g < NGE ? e[j] : o[j]) + fb.b[g][j >> 2][j & 3];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        acc[tl][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                            g < NGE ? e[j] : o[j], fb.b[g][j >> 2][j & 3], acc[tl][g], 0, 0, 0);
                // the folds as one group ahead of the block's matrix instructions (6.41 - 6.60 -> 6.26 - 6.48 ms on C5)
                __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8 * NG, 0);
            }
        };

        // The loop is software pipelined over half-stages h = 2 (s - s_begin) + tl (ring slot h % 4, mask slot
        // s & 1): the fragments of block 0 of h + 1 are fetched BEFORE the matrix instructions of block 1 of h
        // are issued, so the pipe does not wait for a half-stage's waits, barrier, copy issue and first LDS
        // round trip (the unpipelined loop: 6.5 ms per 8192 frames of C5, 4.95 ms without the frame copies).
        //   step h:   read block 1 of h  |  MFMA block 0 of h  |  M(h)  |  read block 0 of h + 1  |  MFMA block 1 of h
        //   M(h):     block 1 of h is in registers -> slot h % 4 is free: issue F(h + 4);
        //             wait for F(h + 1) [tl = 1: and for B(s + 1); then barrier; issue B(s + 2)]
        // Copies are issued in the order  M(h), tl = 0: F(h + 4)   M(h), tl = 1: F(h + 4), B(s + 2), so the copies
        // younger than the ones M(h) waits for are
        //   tl = 0 (F(h + 1)):          B, F(h + 2), F(h + 3), B, F(h + 4)   -> vmcnt(3 NF + 2 NBI)
        //   tl = 1 (F(h + 1), B(s + 1)): F(h + 3), F(h + 4)                   -> vmcnt(2 NF)
        static_assert(3 * NF + 2 * NBI < 64, "vmcnt is a 6-bit counter");
        int since_flush = 0;
        // fragments of block 0 / block 1 of the current half-stage; the weights of a stage's two blocks are read
        // ONCE for both of its tiles (block 1 in the tile-0 step, block 0 ahead of it): a third fewer LDS reads
        // per matrix instruction -- the kernel runs at the board's power cap (1380 W), energy is time
        FragAC f0, f1;
        FragB w0, w1;
        auto step = [&](auto TL, auto Q, auto BS, int s) {
            constexpr int tl = decltype(TL)::value, q = decltype(Q)::value, bslot = decltype(BS)::value;
            load_ac(f1, Q, 1);
            if (tl == 0) load_b(w1, BS, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_block(f0, w0, TL);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // block 1 is in registers
            issue_half(TL, Q);                                                  // F(h + 4) -> slot h % 4
            if (tl == 0) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NF + 2 * NBI) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                load_ac(f0, std::integral_constant<int, (q + 1) & 3>{}, 0);
            } else {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NF) : "memory");
                if (ABL < 3) __builtin_amdgcn_s_barrier();   // everybody's quarter of B(s + 1) is there, B(s) is free
                issue_b(s + 2, bslot);
                __builtin_amdgcn_sched_barrier(0);
                load_ac(f0, std::integral_constant<int, (q + 1) & 3>{}, 0);
                load_b(w0, std::integral_constant<int, bslot ^ 1>{}, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_block(f1, w1, TL);
            if (tl == 1 && ++since_flush == 8) {             // second accumulation level every 512 folded pixels
                since_flush = 0;
#pragma unroll
                for (int t2 = 0; t2 < FD_TILES; ++t2)
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        acc2[t2][g] += acc[t2][g];
                        acc[t2][g] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;
        // prologue: F(0), F(1), B(s_begin), F(2), F(3), B(s_begin + 1); block 0 of half-stage 0
        issue_half(I0{}, I0{});
        issue_half(I1{}, I1{});
        issue_b(s_begin, 0);
        issue_half(I0{}, I2{});
        issue_half(I1{}, I3{});
        issue_b(s_begin + 1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NF + NBI) : "memory");
        if (ABL < 3) __builtin_amdgcn_s_barrier();
        load_ac(f0, I0{}, 0);
        load_b(w0, I0{}, 0);
        int s = s_begin;
        for (; s + 2 <= s_end; s += 2) {
            step(I0{}, I0{}, I0{}, s);
            step(I1{}, I1{}, I0{}, s);
            step(I0{}, I2{}, I1{}, s + 1);
            step(I1{}, I3{}, I1{}, s + 1);
        }
        if (s < s_end) {
            step(I0{}, I0{}, I0{}, s);
            step(I1{}, I1{}, I0{}, s);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // drain the clamped prefetches
    }

#pragma unroll
    for (int tl = 0; tl < FD_TILES; ++tl)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int col = colmap[g * GROUP + m];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t f = frame_of(tl * 16 + kg * 4 + r);
                if (f >= 0 && col >= 0) {
                    const float v = acc[tl][g][r] + acc2[tl][g][r];
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


// ---- the same product with TWO waves per SIMD (round 6) -----------------------------------------------------------------
// k_dense_fold keeps one wave per SIMD (4 waves x 2 frame tiles) and hides every wait by software pipelining inside the
// wave; the matrix pipe still idles a quarter of the time without any frame copies (4.7 ms per 8192 frames of C5 where
// the instructions alone take 3.5).  Here the two tiles of a wave become two waves on the same SIMD (8 waves x 1 tile,
// the same 128 frames, ring, mask slots and LDS layout): while one waits for its fragments, its copies or the stage
// barrier, the other issues.  Plain loop per wave and stage: fragments of both 32-pixel blocks, matrix instructions,
// the copy of its half-stage two stages ahead into the ring slot just read, wait for the next half-stage and its
// share of the next mask slot, barrier, its share of the mask slot after that.  Pays with a second read of a stage's
// weights (each tile's wave reads them itself).  Stacks of 2 or 4 column groups (mask slot shares of whole KiB).
template <int NGE, int NGO>
__global__ void __launch_bounds__(8 * 64)
k_dense_fold8(const float *__restrict__ tile, int64_t ld, int64_t n_frames, int spr /* stages per row */,
              const int2 *__restrict__ fold_rows, const float *__restrict__ img, int n_stages,
              float *__restrict__ out, int64_t ld_out, int n_cols, const int *__restrict__ colmap,
              int accumulate, float *__restrict__ partials, int ksplit,
              const unsigned char *__restrict__ zeros, const int32_t *__restrict__ rows) {
    constexpr int NG = NGE + NGO;
    constexpr int BSLOT = fold_slot_bytes(NG);
    constexpr int W8 = 8;
    constexpr int NBI = BSLOT / W8 / 1024;                // mask-slot DMA instructions per wave and stage
    constexpr int NDH = 16 / 4;                           // DMA instructions per part of a half-stage
    constexpr int NF = 2 * NDH;
    static_assert(BSLOT % (W8 * 1024) == 0, "whole DMA instructions per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int vw = wave >> 1, tl = wave & 1;              // the rows of "virtual wave" vw, frame tile tl
    const int lane = tid & 63;
    const int m = lane & 15, kg = lane >> 4;
    const int ks = blockIdx.y;
    const int per = (n_stages + ksplit - 1) / ksplit;
    const int s_begin = ks * per;
    const int s_end = min(n_stages, s_begin + per);

    const int64_t f_wave = (int64_t)blockIdx.x * FD_WG_ROWS + vw * FD_ROWS + tl * 16;
    auto frame_of = [&](int r) -> int64_t {
        const int64_t f = f_wave + r;
        return f < n_frames ? f : -1;
    };
    auto src_frame_of = [&](int r) -> int64_t {
        int64_t f = frame_of(r);
        if (rows) return f < 0 ? (int64_t)rows[0] : (int64_t)rows[f];
        return f < 0 ? n_frames - 1 : f;
    };
    // ring slot q = 2 * (stage parity) + tl: the virtual wave's 16 rows of part A (4 KiB), then of part C
    unsigned char *a_base = lds_raw + vw * (2 * FD_TPART);
    unsigned char *b_base = lds_raw + FD_RING * FD_HALF;

    f32x4 acc[NG], acc2[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = acc2[g] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (s_begin < s_end) {
        const unsigned char *src[NDH];
#pragma unroll
        for (int t = 0; t < NDH; ++t) {
            const int r = 4 * t + lane / 16;
            const int piece = (lane & 15) ^ r;
            src[t] = (const unsigned char *)(tile + src_frame_of(r) * ld) + piece * 16;
        }
        const unsigned char *zsrc = zeros + lane * 16;
        const unsigned char *bsrc = (const unsigned char *)img + wave * (BSLOT / W8) + lane * 16;
        int iss = s_begin, iss_fy = s_begin / spr, iss_xs = s_begin % spr;
        auto issue_f = [&](auto Q) {                       // this wave's half-stage of stage `iss` (clamped) -> ring slot q
            constexpr int q = decltype(Q)::value;
            const int2 rr = fold_rows[iss_fy];
            const int64_t x_px = (int64_t)iss_xs * FD_KB;
            const int64_t off_a = ((int64_t)rr.x * spr * FD_KB + x_px) * 4;
            const int64_t off_c = ((int64_t)rr.y * spr * FD_KB + x_px) * 4;
            const bool pair = rr.y >= 0;
            unsigned char *da = a_base + q * FD_HALF, *dc = da + FD_TPART;
#pragma unroll
            for (int t = 0; t < NDH; ++t)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[t] + off_a), (lds_ptr_t)(da + t * 1024), 16, 0, 2 /*nt*/);
#pragma unroll
            for (int t = 0; t < NDH; ++t) {
                const unsigned char *p = pair ? src[t] + off_c : zsrc;
                __builtin_amdgcn_global_load_lds((glb_ptr_t)p, (lds_ptr_t)(dc + t * 1024), 16, 0, 2 /*nt*/);
            }
            if (iss + 1 < s_end) {
                ++iss;
                if (++iss_xs == spr) { iss_xs = 0; ++iss_fy; }
            }
        };
        auto issue_b = [&](int s, int bslot) {
            unsigned char *db = b_base + bslot * BSLOT + wave * (BSLOT / W8);
            const unsigned char *sp = bsrc + (int64_t)min(s, s_end - 1) * BSLOT;
#pragma unroll
            for (int u = 0; u < NBI; ++u)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(sp + u * 1024), (lds_ptr_t)(db + u * 1024), 16, 0, 0);
        };
        const int a_lane = m * 256;
        const int b_lane = m * FD_KB;
        const int b_hi = (kg ^ ((4 - (m >> 2)) & 3)) * 4;
        auto b_unit = [&](int blk, int h) { return (b_hi + ((blk * 2 + h) ^ (m & 3))) << 2; };
        struct FragAC { f32x4 a[2], c[2]; };
        struct FragB { f32x4 b[NG][2]; };
        auto load_ac = [&](FragAC &fr, auto Q, int blk) {
            constexpr int q = decltype(Q)::value;
            const unsigned char *as = a_base + q * FD_HALF + a_lane;
            const unsigned char *cs = as + FD_TPART;
            const int u = blk * 4 + kg;
            fr.a[0] = *(const f32x4 *)(as + (((2 * u) ^ m) << 4));
            fr.a[1] = *(const f32x4 *)(as + (((2 * u + 1) ^ m) << 4));
            fr.c[0] = *(const f32x4 *)(cs + (((2 * u) ^ m) << 4));
            fr.c[1] = *(const f32x4 *)(cs + (((2 * u + 1) ^ m) << 4));
        };
        auto load_b = [&](FragB &fr, auto BS, int blk) {
            constexpr int bslot = decltype(BS)::value;
            const float *bs = (const float *)(b_base + bslot * BSLOT) + b_lane;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                fr.b[g][0] = *(const f32x4 *)(bs + g * (GROUP * FD_KB) + b_unit(blk, 0));
                fr.b[g][1] = *(const f32x4 *)(bs + g * (GROUP * FD_KB) + b_unit(blk, 1));
            }
        };
        auto mfma_block = [&](const FragAC &fr, const FragB &fb) {
            float e[8], o[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 ev = fr.a[h] + fr.c[h];
                const f32x4 ov = fr.a[h] - fr.c[h];
#pragma unroll
                for (int i = 0; i < 4; ++i) { e[h * 4 + i] = ev[i]; o[h * 4 + i] = ov[i]; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(g < NGE ? e[j] : o[j], fb.b[g][j >> 2][j & 3], acc[g],
                                                                  0, 0, 0);
        };
        int since_flush = 0;
        auto step = [&](auto P, int s) {
            constexpr int par = decltype(P)::value;
            using Q = std::integral_constant<int, 2 * par>;               // + tl at run time: see below
            FragAC f0, f1;
            FragB w0, w1;
            // (tl is a wave-uniform run-time value: both instantiations, one taken)
            if (tl == 0) {
                load_ac(f0, std::integral_constant<int, 2 * par>{}, 0);
                load_ac(f1, std::integral_constant<int, 2 * par>{}, 1);
            } else {
                load_ac(f0, std::integral_constant<int, 2 * par + 1>{}, 0);
                load_ac(f1, std::integral_constant<int, 2 * par + 1>{}, 1);
            }
            load_b(w0, P, 0);
            load_b(w1, P, 1);
            mfma_block(f0, w0);
            mfma_block(f1, w1);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the half-stage and the slot are in registers
            if (tl == 0) issue_f(std::integral_constant<int, 2 * par>{});  // stage s + 2 -> the ring slot just read
            else issue_f(std::integral_constant<int, 2 * par + 1>{});
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NF) : "memory");      // F(s + 1), this wave's share of B(s + 1)
            __builtin_amdgcn_s_barrier();                                  // everybody's share is there, slot `par` is free
            issue_b(s + 2, par);
            __builtin_amdgcn_sched_barrier(0);
            (void)sizeof(Q);
            if (++since_flush == 8) {                                      // second accumulation level every 512 folded pixels
                since_flush = 0;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    acc2[g] += acc[g];
                    acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        // prologue: F(s_begin), B(s_begin), F(s_begin + 1), B(s_begin + 1)
        if (tl == 0) issue_f(std::integral_constant<int, 0>{}); else issue_f(std::integral_constant<int, 1>{});
        issue_b(s_begin, 0);
        if (tl == 0) issue_f(std::integral_constant<int, 2>{}); else issue_f(std::integral_constant<int, 3>{});
        issue_b(s_begin + 1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NF + NBI) : "memory");
        __builtin_amdgcn_s_barrier();
        int s = s_begin;
        for (; s + 2 <= s_end; s += 2) {
            step(I0{}, s);
            step(I1{}, s + 1);
        }
        if (s < s_end) step(I0{}, s);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // drain the clamped prefetches
    }

#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int col = colmap[g * GROUP + m];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t f = frame_of(kg * 4 + r);
            if (f >= 0 && col >= 0) {
                const float v = acc[g][r] + acc2[g][r];
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


// ---- 2-byte integer pixels (uint16 / int16 detectors) -------------------------------------------------------------------
// Radial-Fourier stacks keep the float32 matrix instruction for integer pixels too (their zero crossings hold more
// weights than the float16 pieces' float32 tail takes), so uint16 frames x 50 columns ran at 0.28 - 0.31 of HBM, bound
// by the same 24 + VALU matrix instructions per 32 pixels as C5.  The fold applies unchanged -- the pixels are
// converted to float32 first, a + c and a - c of integers below 2^17 are exact -- with 256-byte row pieces of 128
// pixels: a stage is 128 folded pixels, its mask slot (NG x 16 x 128 weights, the 128-pixel unit swizzle of
// k_dense_lds) 32 KiB for four groups, the ring three half-stages (one multiplied, two on their way: half the bytes
// per matrix instruction of the float32 case).  Plain loop (no software pipelining: the kernel is bound by its
// matrix instructions): wait, barrier per stage, issue, four blocks with double-buffered fragments.
constexpr int FD16_KB = 128;
constexpr int FD16_RING = 3;
__host__ __device__ constexpr int fold16_slot_bytes(int ng) { return ng * GROUP * FD16_KB * 4; }
__host__ __device__ constexpr int fold16_lds_bytes(int ng, int px_bytes) {
    return FD16_RING * (2 * FD_WAVES * 16 * FD16_KB * px_bytes) + 2 * fold16_slot_bytes(ng);
}
__host__ __device__ static inline int fold16_index(int n, int q) {
    const int blk = q >> 5, kg = (q >> 3) & 3, j = q & 7;
    const int unit = (kg * 8 + blk * 2 + (j >> 2)) ^ (n & 7);
    return n * FD16_KB + unit * 4 + (j & 3);
}

__global__ void k_build_fold_image16(const float *__restrict__ src, float *__restrict__ img, int cpm, int64_t n_px,
                                     int sig_w, int n_fold_rows, const int2 *__restrict__ rows,
                                     const int *__restrict__ colmap, int ng) {
    const int64_t total = (int64_t)ng * GROUP * n_fold_rows * sig_w;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pf = i % ((int64_t)n_fold_rows * sig_w);
        const int v = (int)(i / ((int64_t)n_fold_rows * sig_w));
        const int col = colmap[v];
        const int fy = (int)(pf / sig_w), x = (int)(pf % sig_w);
        float w = 0.f;
        if (col >= 0) {
            const int64_t k = col / cpm, part = col % cpm;
            w = src[(k * n_px + (int64_t)rows[fy].x * sig_w + x) * cpm + part];
        }
        const int64_t slot = pf / FD16_KB;
        const int q = (int)(pf % FD16_KB);
        img[(slot * ng + v / GROUP) * (GROUP * FD16_KB) + fold16_index(v % GROUP, q)] = w;
    }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// the float32 banded image (64-pixel windows) -> 128-pixel windows: window s of the new image starts at pixel
// st[s].z of its row and owns the pixels from st[s].w on; its weights come from the 64-pixel windows src[2 s] (stage
// numbers, -1: none) that start at pixels src[2 s + 1] -- a pixel has a weight in exactly one of them (the others hold 0)
__global__ void k_build_band_image16(const float *__restrict__ img, float *__restrict__ img16,
                                     const int4 *__restrict__ st, const int4 *__restrict__ src, int ng,
                                     int64_t n_stages16) {
    const int64_t total = n_stages16 * ng * GROUP * FD16_KB;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % FD16_KB);
        const int n = (int)((i / FD16_KB) % GROUP);
        const int g = (int)((i / (FD16_KB * GROUP)) % ng);
        const int64_t s = i / ((int64_t)FD16_KB * GROUP * ng);
        const int x = st[s].z + q;
        float w = 0.f;
        if (x >= st[s].w) {
            const int4 ids = src[2 * s], x0s = src[2 * s + 1];
            const int id[4] = {ids.x, ids.y, ids.z, ids.w}, x0[4] = {x0s.x, x0s.y, x0s.z, x0s.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q64 = x - x0[j];
                if (id[j] >= 0 && q64 >= 0 && q64 < FD_KB)
                    w += img[((int64_t)id[j] * ng + g) * (GROUP * FD_KB) + fold_index(n, q64)];
            }
        }
        img16[(s * ng + g) * (GROUP * FD16_KB) + fold16_index(n, q)] = w;
    }
}

// LIST: see k_dense_fold
template <typename T, int NGE, int NGO, bool LIST = false>
__global__ void __launch_bounds__(FD_WAVES * 64)
k_dense_fold16(const T *__restrict__ tile, int64_t ld, int64_t n_frames, int spr /* stages (128 px) per row */,
               const int2 *__restrict__ fold_rows, const float *__restrict__ img, int n_stages,
               float *__restrict__ out, int64_t ld_out, int n_cols, const int *__restrict__ colmap,
               int accumulate, float *__restrict__ partials, int ksplit,
               const unsigned char *__restrict__ zeros, const int32_t *__restrict__ rows,
               const int4 *__restrict__ stage_list, const int *__restrict__ blk_off) {
    static_assert(sizeof(T) <= 2, "1- and 2-byte pixels");
    constexpr int NG = NGE + NGO;
    constexpr int BSLOT = fold16_slot_bytes(NG);
    constexpr int NBI = BSLOT / FD_WAVES / 1024;
    // a stage is 128 pixels of a row: 256 bytes (16 pieces of 16 bytes, 4 rows per copy instruction) for 2-byte
    // pixels, 128 bytes (8 pieces, 8 rows per instruction) for 1-byte pixels
    constexpr int ROWB = FD16_KB * (int)sizeof(T);
    constexpr int PPR = ROWB / 16, RPI = 64 / PPR;
    constexpr int NDH = 16 / RPI, NF = 2 * NDH;
    constexpr int TPART = 16 * ROWB, HALF = 2 * FD_WAVES * TPART;
    constexpr int NB = FD16_KB / 32;                      // 32-pixel blocks per stage
    static_assert(NF + NBI < 64, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int m = lane & 15, kg = lane >> 4;
    // LIST: blockIdx.y = block * ksplit + part of the block's stage list
    const int by = LIST ? blockIdx.y / ksplit : 0;
    const int ks = LIST ? blockIdx.y - by * ksplit : blockIdx.y;
    const int b0 = LIST ? blk_off[by] : 0, b1 = LIST ? blk_off[by + 1] : n_stages;
    const int per = (b1 - b0 + ksplit - 1) / ksplit;
    const int s_begin = b0 + ks * per;
    const int s_end = min(b1, s_begin + per);
    if (LIST) colmap += by * (NG * GROUP);

    const int64_t f_wave = (int64_t)blockIdx.x * FD_WG_ROWS + wave * FD_ROWS;
    auto frame_of = [&](int r) -> int64_t {
        const int64_t f = f_wave + r;
        return f < n_frames ? f : -1;
    };
    auto src_frame_of = [&](int r) -> int64_t {
        int64_t f = frame_of(r);
        if (rows) return f < 0 ? (int64_t)rows[0] : (int64_t)rows[f];
        return f < 0 ? n_frames - 1 : f;
    };
    unsigned char *a_base = lds_raw + wave * (2 * TPART);                   // + q * HALF
    unsigned char *b_base = lds_raw + FD16_RING * HALF;                     // + slot * BSLOT

    f32x4 acc[FD_TILES][NG], acc2[FD_TILES][NG];
#pragma unroll
    for (int tl = 0; tl < FD_TILES; ++tl)
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[tl][g] = acc2[tl][g] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (s_begin < s_end) {
        const unsigned char *src[FD_TILES][NDH];
#pragma unroll
        for (int tl = 0; tl < FD_TILES; ++tl)
#pragma unroll
            for (int t = 0; t < NDH; ++t) {
                const int r = RPI * t + lane / PPR;
                const int piece = (lane & (PPR - 1)) ^ (r & (PPR - 1));
                src[tl][t] = (const unsigned char *)(tile + src_frame_of(tl * 16 + r) * ld) + piece * 16;
            }
        const unsigned char *zsrc = zeros + lane * 16;
        const unsigned char *bsrc = (const unsigned char *)img + wave * (BSLOT / FD_WAVES) + lane * 16;

        int iss = s_begin, iss_fy = LIST ? 0 : s_begin / spr, iss_xs = LIST ? 0 : s_begin % spr;
        int4 iss_st = LIST ? stage_list[s_begin] : int4{0, 0, 0, 0};
        auto issue_half = [&](auto TL, int q) {
            constexpr int tl = decltype(TL)::value;
            const int2 rr = LIST ? int2{iss_st.x, iss_st.y} : fold_rows[iss_fy];
            const int64_t x_px = LIST ? iss_st.z : iss_xs * FD16_KB;
            const int64_t off_a = ((int64_t)rr.x * spr * FD16_KB + x_px) * (int)sizeof(T);
            const int64_t off_c = ((int64_t)rr.y * spr * FD16_KB + x_px) * (int)sizeof(T);
            const bool pair = rr.y >= 0;
            unsigned char *da = a_base + q * HALF, *dc = da + TPART;
#pragma unroll
            for (int t = 0; t < NDH; ++t)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[tl][t] + off_a), (lds_ptr_t)(da + t * 1024), 16, 0,
                                                 2 /*nt*/);
#pragma unroll
            for (int t = 0; t < NDH; ++t) {
                const unsigned char *p = pair ? src[tl][t] + off_c : zsrc;
                __builtin_amdgcn_global_load_lds((glb_ptr_t)p, (lds_ptr_t)(dc + t * 1024), 16, 0, 2 /*nt*/);
            }
            if (tl == FD_TILES - 1 && iss + 1 < s_end) {
                ++iss;
                if (LIST) iss_st = stage_list[iss];
                else if (++iss_xs == spr) { iss_xs = 0; ++iss_fy; }
            }
        };
        auto issue_b = [&](int s, int bslot) {
            unsigned char *db = b_base + bslot * BSLOT + wave * (BSLOT / FD_WAVES);
            const unsigned char *sp = bsrc + (int64_t)min(s, s_end - 1) * BSLOT;
#pragma unroll
            for (int u = 0; u < NBI; ++u)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(sp + u * 1024), (lds_ptr_t)(db + u * 1024), 16, 0, 0);
        };

        const int a_lane = m * ROWB;
        const int b_lane = m * FD16_KB;
        auto b_unit = [&](int blk, int h) { return ((kg * 8 + blk * 2 + h) ^ (m & 7)) << 2; };
        // the lane's 8 pixels of a block: 16 bytes (2-byte pixels) or 8 bytes
        typedef unsigned int raw_t __attribute__((ext_vector_type(sizeof(T) == 2 ? 4 : 2)));
        auto to_float = [&](const raw_t &r, float (&f)[8]) {
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (std::is_signed<T>::value) {
                        f[2 * i] = (float)(int)(short)(r[i] & 0xffffu);
                        f[2 * i + 1] = (float)((int)r[i] >> 16);
                    } else {
                        f[2 * i] = (float)(r[i] & 0xffffu);
                        f[2 * i + 1] = (float)(r[i] >> 16);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (std::is_signed<T>::value) f[4 * i + e] = (float)(int)(signed char)((r[i] >> (8 * e)) & 0xffu);
                        else f[4 * i + e] = (float)((r[i] >> (8 * e)) & 0xffu);
                    }
            }
        };

        // one tile (16 frames) of one stage: NB blocks of 32 folded pixels x NG groups
        auto compute = [&](auto TL, int q, int bslot) {
            constexpr int tl = decltype(TL)::value;
            const unsigned char *as = a_base + q * HALF + a_lane;
            const unsigned char *cs = as + TPART;
            const float *bs = (const float *)(b_base + bslot * BSLOT) + b_lane;
            raw_t ra[2], rc[2];
            f32x4 rb[2][NG][2];
            auto load_block = [&](int blk, int buf) {
                const int u = blk * 4 + kg;                  // 8-pixel unit of this lane inside the part
                if constexpr (sizeof(T) == 2) {
                    ra[buf] = *(const raw_t *)(as + ((u ^ m) << 4));
                    rc[buf] = *(const raw_t *)(cs + ((u ^ m) << 4));
                } else {
                    ra[buf] = *(const raw_t *)(as + (((u >> 1) ^ (m & 7)) << 4) + (u & 1) * 8);
                    rc[buf] = *(const raw_t *)(cs + (((u >> 1) ^ (m & 7)) << 4) + (u & 1) * 8);
                }
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    rb[buf][g][0] = *(const f32x4 *)(bs + g * (GROUP * FD16_KB) + b_unit(blk, 0));
                    rb[buf][g][1] = *(const f32x4 *)(bs + g * (GROUP * FD16_KB) + b_unit(blk, 1));
                }
            };
            load_block(0, 0);
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                const int cur = blk & 1;
                if (blk + 1 < NB) load_block(blk + 1, cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                float fa[8], fc[8], e[8], o[8];
                to_float(ra[cur], fa);
                to_float(rc[cur], fc);
#pragma unroll
                for (int j = 0; j < 8; ++j) { e[j] = fa[j] + fc[j]; o[j] = fa[j] - fc[j]; }
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        acc[tl][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(g < NGE ? e[j] : o[j], rb[cur][g][j >> 2][j & 3],
                                                                          acc[tl][g], 0, 0, 0);
                // conversions and folds as one group ahead of the block's matrix instructions (a VALU operation in
                // front of every MFMA costs 15 % of the pipe, probes/mfma_probe.hip)
                __builtin_amdgcn_sched_group_barrier(0x002, 16 + (NGO > 0 ? 16 : 8), 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8 * NG, 0);
            }
        };

        // Copies are issued in the order  step h, tl = 0: B(s + 1), F(h + 2)   tl = 1: F(h + 2); at the start of step h
        // the copies younger than F(h) (and, tl = 0, than B(s)) are  tl = 0: F(h + 1)   tl = 1: B(s + 1), F(h + 1)
        int since_flush = 0;
        auto step = [&](auto TL, int q, int bslot, int s) {
            constexpr int tl = decltype(TL)::value;
            if (tl == 0) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NF) : "memory");
                __builtin_amdgcn_s_barrier();                // everybody's quarter of B(s) is there, B(s - 1) is free
                issue_b(s + 1, bslot ^ 1);
            } else {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NF + NBI) : "memory");
            }
            issue_half(std::integral_constant<int, tl>{}, q == 0 ? FD16_RING - 1 : q - 1);   // F(h + 2) -> the slot of h - 1
            compute(TL, q, bslot);
            if (tl == 1 && ++since_flush == 4) {             // second accumulation level every 512 folded pixels (the folded
                                                             // values are twice the pixels': a constant full-scale frame on an
                                                             // all-positive column drifted to 1.0e-5 with chains of 1024)
                since_flush = 0;
#pragma unroll
                for (int t2 = 0; t2 < FD_TILES; ++t2)
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        acc2[t2][g] += acc[t2][g];
                        acc[t2][g] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        issue_b(s_begin, 0);
        issue_half(I0{}, 0);
        issue_half(I1{}, 1);
        int q = 0, bslot = 0;
        for (int s = s_begin; s < s_end; ++s) {
            step(I0{}, q, bslot, s);
            q = q + 1 == FD16_RING ? 0 : q + 1;
            step(I1{}, q, bslot, s);
            q = q + 1 == FD16_RING ? 0 : q + 1;
            bslot ^= 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

#pragma unroll
    for (int tl = 0; tl < FD_TILES; ++tl)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int col = colmap[g * GROUP + m];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t f = frame_of(tl * 16 + kg * 4 + r);
                if (f >= 0 && col >= 0) {
                    const float v = acc[tl][g][r] + acc2[tl][g][r];
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

}  // namespace ltmi

using namespace ltmi;


// the folded image of a stack (ltmi_masks::fold)
struct FoldImage {
    int sig_h = 0, sig_w = 0, c2 = 0;            // rows y and c2 - y are mirror partners
    int nge = 0, ngo = 0;                        // groups of even / odd columns
    int n_fold_rows = 0, n_stages = 0;
    float *img = nullptr;
    int2 *rows = nullptr;                        // folded row -> (y, y' or -1)
    int *colmap = nullptr;                       // virtual column -> real column or -1
    unsigned char *zeros = nullptr;              // 1 KiB: the partner of unpaired rows
    size_t img_bytes = 0;
    float *img16 = nullptr;                      // the image in 128-pixel slots (2-byte pixels), built on first use
    int n_stages16 = 0;
    bool img16_failed = false;
};

void ltmi::fold_destroy(ltmi_masks *m) {
    FoldImage *f = (FoldImage *)m->fold;
    if (!f) return;
    if (f->img) (void)hipFree(f->img);
    if (f->rows) (void)hipFree(f->rows);
    if (f->colmap) (void)hipFree(f->colmap);
    if (f->zeros) (void)hipFree(f->zeros);
    if (f->img16) (void)hipFree(f->img16);
    delete f;
    m->fold = nullptr;
}

// Looks for a row mirror under which every column of the stack is even or odd and builds the folded image when
// that saves matrix work.  Not an error when there is none (the handle then works as before).
int ltmi::fold_create(ltmi_masks *m, int sig_h, int sig_w) {
    fold_destroy(m);
    const char *off = getenv("LTMI_DENSE_FOLD");
    if (off && atoi(off) == 0) return LTMI_OK;
    if (m->kind != 0 || !m->gmasks || (int64_t)sig_h * sig_w != m->n_px || sig_h < 4) return LTMI_OK;
    if (sig_w % FD_KB != 0 || m->n_cols > 4 * GROUP) return LTMI_OK;
    const int cpm = m->result_dtype == LTMI_C64 ? 2 : 1;
    const int groups_now = (m->n_cols + GROUP - 1) / GROUP;
    if (groups_now < 2 && !(off && atoi(off) == 2)) return LTMI_OK;     // HBM-bound already (LTMI_DENSE_FOLD=2: tests)
    int *flags = nullptr;
    const size_t nf = (size_t)2 * m->n_cols;
    LTMI_HIP(hipMalloc((void **)&flags, nf * sizeof(int)));
    std::vector<int> ones(nf, 1), got(nf);
    const int64_t total = (int64_t)m->n_masks * cpm * m->n_px;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 65535 * 16);
    int best_c2 = -1;
    std::vector<int> cls;                                               // +1 even, -1 odd per column
    for (int c2 : {sig_h, sig_h - 1, sig_h + 1}) {                      // centres (sig_h - 1) / 2 +- 1 / 2
        hipError_t e = hipMemcpy(flags, ones.data(), nf * sizeof(int), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(ltmi::k_fold_classify, dim3(blocks), dim3(256), 0, 0, (const float *)m->gmasks, cpm,
                               m->n_masks, sig_h, sig_w, c2, flags, flags + m->n_cols);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpy(got.data(), flags, nf * sizeof(int), hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            (void)hipFree(flags);
            LTMI_FAIL((int)e, "classifying the stack's columns failed: %s", hipGetErrorString(e));
        }
        bool all = true;
        std::vector<int> c((size_t)m->n_cols);
        for (int k = 0; k < m->n_cols && all; ++k) {
            if (got[(size_t)k]) c[(size_t)k] = 1;                       // (all-zero columns: even)
            else if (got[(size_t)m->n_cols + k]) c[(size_t)k] = -1;
            else all = false;
        }
        if (all) { best_c2 = c2; cls = c; break; }
    }
    (void)hipFree(flags);
    if (best_c2 < 0) return LTMI_OK;
    int n_even = 0, n_odd = 0;
    for (int k = 0; k < m->n_cols; ++k) (cls[(size_t)k] > 0 ? n_even : n_odd)++;
    const int nge = (n_even + GROUP - 1) / GROUP, ngo = (n_odd + GROUP - 1) / GROUP;
    // kernels are built for 1 - 4 even groups or 1 - 2 + 1 - 2; the fold must save matrix work:
    // (nge + ngo) groups over half of the pixels against groups_now over all of them
    const bool built = (ngo == 0 && nge >= 1 && nge <= 4) || (nge >= 1 && nge <= 2 && ngo >= 1 && ngo <= 2);
    if (!built || nge + ngo >= 2 * groups_now) return LTMI_OK;

    FoldImage *f = new (std::nothrow) FoldImage();
    if (!f) LTMI_FAIL(LTMI_E_NOMEM, "out of host memory");
    m->fold = f;
    f->sig_h = sig_h; f->sig_w = sig_w; f->c2 = best_c2; f->nge = nge; f->ngo = ngo;
    std::vector<int2> rows;
    for (int y = 0; y < sig_h; ++y) {
        const int y2 = best_c2 - y;
        if (y2 >= 0 && y2 < sig_h && y2 < y) continue;                  // the partner of an earlier row
        rows.push_back(int2{y, (y2 > y && y2 < sig_h) ? y2 : -1});
    }
    f->n_fold_rows = (int)rows.size();
    f->n_stages = f->n_fold_rows * (sig_w / FD_KB);
    const int ng = nge + ngo;
    std::vector<int> colmap((size_t)ng * GROUP, -1);
    {
        int ie = 0, io = nge * GROUP;
        for (int k = 0; k < m->n_cols; ++k) colmap[(size_t)(cls[(size_t)k] > 0 ? ie++ : io++)] = k;
    }
    f->img_bytes = (size_t)f->n_stages * ltmi::fold_slot_bytes(ng);
    hipError_t e = hipMalloc((void **)&f->img, f->img_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&f->rows, rows.size() * sizeof(int2));
    if (e == hipSuccess) e = hipMalloc((void **)&f->colmap, colmap.size() * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&f->zeros, 1024);
    if (e == hipSuccess) e = hipMemset(f->zeros, 0, 1024);
    if (e == hipSuccess) e = hipMemcpy(f->rows, rows.data(), rows.size() * sizeof(int2), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(f->colmap, colmap.data(), colmap.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const int64_t tot = (int64_t)ng * GROUP * f->n_fold_rows * sig_w;
        const unsigned bl = (unsigned)std::min<int64_t>((tot + 255) / 256, 65535 * 16);
        hipLaunchKernelGGL(ltmi::k_build_fold_image, dim3(bl), dim3(256), 0, 0, (const float *)m->gmasks, f->img, cpm,
                           m->n_px, sig_w, f->n_fold_rows, (const int2 *)f->rows, (const int *)f->colmap, ng);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    if (e != hipSuccess) {
        // (no room for the image: the handle works without it)
        fold_destroy(m);
        (void)hipGetLastError();
    }
    return LTMI_OK;
}

bool ltmi::fold_takes(const ltmi_masks *m, const float *tile, int64_t ld) {
    const FoldImage *f = (const FoldImage *)m->fold;
    if (!f || m->tune_ksplit_ring == 38) return false;                  // tuning 38: the unfolded kernels (bench, tests)
    return ((uintptr_t)tile % 16 == 0) && (ld * 4) % 16 == 0;
}

template <int NGE, int NGO>
static int launch_fold_t(ltmi_masks *m, const float *tile, int64_t n_frames, int64_t ld, float *out, int64_t ld_out,
                         int accumulate, hipStream_t stream) {
    const FoldImage *f = (const FoldImage *)m->fold;
    const int abl = m->tune_ksplit_ring == 31 ? 2 : (m->tune_ksplit_ring == 32 ? 1 : 0);
    auto kern = abl == 2 ? k_dense_fold<NGE, NGO, 2> : (abl == 1 ? k_dense_fold<NGE, NGO, 1> : k_dense_fold<NGE, NGO, 0>);
    constexpr int LDS = ltmi::fold_lds_bytes(NGE + NGO);
    static bool attr_set[16][3] = {{false}};
    if (!attr_set[m->device & 15][abl]) {
        LTMI_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set[m->device & 15][abl] = true;
    }
    const int64_t gx = (n_frames + FD_WG_ROWS - 1) / FD_WG_ROWS;
    int ksplit = m->tune_ksplit;
    if (ksplit <= 0) ksplit = choose_ksplit(gx, f->n_stages);
    ksplit = std::max(1, std::min(ksplit, f->n_stages));
    {
        const int per = (f->n_stages + ksplit - 1) / ksplit;
        ksplit = (f->n_stages + per - 1) / per;
    }
    if (ksplit > 1) {
        int rc = dense_ensure_partials(m, (size_t)ksplit * n_frames * m->n_cols * sizeof(float), stream);
        if (rc != LTMI_OK) return rc;
    }
    dim3 grid((unsigned)gx, (unsigned)ksplit);
    static const int turn_env = getenv("LTMI_FOLD_TURN") ? atoi(getenv("LTMI_FOLD_TURN")) : -1;
    const bool in_turn = ksplit > 1 && (turn_env >= 0 ? turn_env != 0 : FOLD_TURN_DEFAULT);
    // two waves per SIMD (k_dense_fold8): stacks of 2 or 4 column groups; measured 3 - 5 % SLOWER than the pipelined
    // one-wave-per-SIMD kernel on C5 (profiles/r06_fold.txt), kept as a measurement switch: LTMI_FOLD_WAVES=8
    const char *fw = getenv("LTMI_FOLD_WAVES");               // (read per launch: tests and benches switch it)
    const int want_waves = fw ? atoi(fw) : 0;
    bool eight = false;
    if constexpr ((NGE + NGO) % 2 == 0) eight = abl == 0 && (want_waves == 8 || (want_waves == 0 && FOLD8_DEFAULT));
    if (eight) {
        if constexpr ((NGE + NGO) % 2 == 0) {
            auto k8 = k_dense_fold8<NGE, NGO>;
            static bool attr8[16] = {false};
            if (!attr8[m->device & 15]) {
                LTMI_HIP(hipFuncSetAttribute((const void *)k8, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
                attr8[m->device & 15] = true;
            }
            hipLaunchKernelGGL(k8, grid, dim3(8 * 64), LDS, stream, tile, ld, n_frames, f->sig_w / FD_KB,
                               (const int2 *)f->rows, (const float *)f->img, f->n_stages, out, ld_out, m->n_cols,
                               (const int *)f->colmap, accumulate, dense_partial_sums(m), ksplit,
                               (const unsigned char *)f->zeros, m->roi_rows);
        }
    } else {
        hipLaunchKernelGGL(kern, grid, dim3(FD_WAVES * 64), LDS, stream, tile, ld, n_frames, f->sig_w / FD_KB,
                           (const int2 *)f->rows, (const float *)f->img, f->n_stages, out, ld_out, m->n_cols,
                           (const int *)f->colmap, accumulate, dense_partial_sums(m), in_turn ? -ksplit : ksplit,
                           (const unsigned char *)f->zeros, m->roi_rows, (const int4 *)nullptr, (const int *)nullptr);
    }
    LTMI_HIP(hipGetLastError());
    snprintf(m->last_kernel, sizeof(m->last_kernel), "k_dense_fold%s<f,even=%d,odd=%d,rows %d+%d=%d%s%s> grid=(%u,%u)",
             eight ? "8" : "", NGE, NGO, f->n_fold_rows, f->sig_h - f->n_fold_rows, f->c2, m->roi_rows ? ",rows" : "",
             in_turn && !eight ? ",in turn" : "", grid.x, grid.y);
    if (ksplit > 1) {
        const int rc = dense_reduce_partials(m, ksplit, n_frames, out, ld_out, accumulate, stream);
        if (rc != LTMI_OK) return rc;
    }
    return LTMI_OK;
}

int ltmi::launch_fold(ltmi_masks *m, const float *tile, int64_t n_frames, int64_t ld, float *out, int64_t ld_out,
                       int accumulate, hipStream_t stream) {
    const FoldImage *f = (const FoldImage *)m->fold;
#define LTMI_FOLD_CASE(E_, O_)                                                                                 \
    if (f->nge == E_ && f->ngo == O_)                                                                          \
        return launch_fold_t<E_, O_>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
    LTMI_FOLD_CASE(1, 0) LTMI_FOLD_CASE(2, 0) LTMI_FOLD_CASE(3, 0) LTMI_FOLD_CASE(4, 0)
    LTMI_FOLD_CASE(1, 1) LTMI_FOLD_CASE(2, 1) LTMI_FOLD_CASE(1, 2) LTMI_FOLD_CASE(2, 2)
#undef LTMI_FOLD_CASE
    LTMI_FAIL(LTMI_E_INVALID, "k_dense_fold: no kernel for %d + %d groups", f->nge, f->ngo);
}

// ---- 2-byte pixels ------------------------------------------------------------------------------------------------
bool ltmi::fold_takes16(ltmi_masks *m, const void *tile, int64_t ld, int px_bytes) {
    FoldImage *f = (FoldImage *)m->fold;
    if (!f || m->tune_ksplit_ring == 38 || f->img16_failed) return false;
    if (f->sig_w % FD16_KB != 0 || ((uintptr_t)tile % 16 != 0) || (ld * px_bytes) % 16 != 0) return false;
    if (!f->img16) {
        // built on the first 2-byte tile (a stack that only ever sees float32 frames does not pay for it)
        const int cpm = m->result_dtype == LTMI_C64 ? 2 : 1;
        const int ng = f->nge + f->ngo;
        f->n_stages16 = f->n_fold_rows * (f->sig_w / FD16_KB);
        const size_t bytes = (size_t)f->n_stages16 * ltmi::fold16_slot_bytes(ng);
        hipError_t e = hipMalloc((void **)&f->img16, bytes);
        if (e == hipSuccess) {
            const int64_t tot = (int64_t)ng * GROUP * f->n_fold_rows * f->sig_w;
            const unsigned bl = (unsigned)std::min<int64_t>((tot + 255) / 256, 65535 * 16);
            hipLaunchKernelGGL(ltmi::k_build_fold_image16, dim3(bl), dim3(256), 0, 0, (const float *)m->gmasks, f->img16,
                               cpm, m->n_px, f->sig_w, f->n_fold_rows, (const int2 *)f->rows, (const int *)f->colmap, ng);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipDeviceSynchronize();
        }
        if (e != hipSuccess) {
            if (f->img16) (void)hipFree(f->img16);
            f->img16 = nullptr;
            f->img16_failed = true;
            (void)hipGetLastError();
            return false;
        }
    }
    return true;
}

template <typename T, int NGE, int NGO>
static int launch_fold16_t(ltmi_masks *m, const T *tile, int64_t n_frames, int64_t ld, float *out, int64_t ld_out,
                           int accumulate, hipStream_t stream) {
    const FoldImage *f = (const FoldImage *)m->fold;
    auto kern = k_dense_fold16<T, NGE, NGO, false>;
    constexpr int LDS = ltmi::fold16_lds_bytes(NGE + NGO, (int)sizeof(T));
    static bool attr_set[16] = {false};
    if (!attr_set[m->device & 15]) {
        LTMI_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set[m->device & 15] = true;
    }
    const int64_t gx = (n_frames + FD_WG_ROWS - 1) / FD_WG_ROWS;
    int ksplit = m->tune_ksplit;
    if (ksplit <= 0) ksplit = choose_ksplit(gx, f->n_stages16);
    ksplit = std::max(1, std::min(ksplit, f->n_stages16));
    {
        const int per = (f->n_stages16 + ksplit - 1) / ksplit;
        ksplit = (f->n_stages16 + per - 1) / per;
    }
    if (ksplit > 1) {
        int rc = dense_ensure_partials(m, (size_t)ksplit * n_frames * m->n_cols * sizeof(float), stream);
        if (rc != LTMI_OK) return rc;
    }
    dim3 grid((unsigned)gx, (unsigned)ksplit);
    hipLaunchKernelGGL(kern, grid, dim3(FD_WAVES * 64), LDS, stream, tile, ld, n_frames, f->sig_w / FD16_KB,
                       (const int2 *)f->rows, (const float *)f->img16, f->n_stages16, out, ld_out, m->n_cols,
                       (const int *)f->colmap, accumulate, dense_partial_sums(m), ksplit,
                       (const unsigned char *)f->zeros, m->roi_rows, (const int4 *)nullptr, (const int *)nullptr);
    LTMI_HIP(hipGetLastError());
    snprintf(m->last_kernel, sizeof(m->last_kernel), "k_dense_fold16<%s,even=%d,odd=%d,rows %d+%d=%d%s> grid=(%u,%u)",
             typeid(T).name(), NGE, NGO, f->n_fold_rows, f->sig_h - f->n_fold_rows, f->c2, m->roi_rows ? ",rows" : "",
             grid.x, grid.y);
    if (ksplit > 1) {
        const int rc = dense_reduce_partials(m, ksplit, n_frames, out, ld_out, accumulate, stream);
        if (rc != LTMI_OK) return rc;
    }
    return LTMI_OK;
}

template <typename T>
static int launch_fold16_any(ltmi_masks *m, const T *tile, int64_t n_frames, int64_t ld, float *out, int64_t ld_out,
                             int accumulate, hipStream_t stream) {
    const FoldImage *f = (const FoldImage *)m->fold;
#define LTMI_FOLD_CASE(E_, O_)                                                                                 \
    if (f->nge == E_ && f->ngo == O_)                                                                          \
        return launch_fold16_t<T, E_, O_>(m, tile, n_frames, ld, out, ld_out, accumulate, stream);
    LTMI_FOLD_CASE(1, 0) LTMI_FOLD_CASE(2, 0) LTMI_FOLD_CASE(3, 0) LTMI_FOLD_CASE(4, 0)
    LTMI_FOLD_CASE(1, 1) LTMI_FOLD_CASE(2, 1) LTMI_FOLD_CASE(1, 2) LTMI_FOLD_CASE(2, 2)
#undef LTMI_FOLD_CASE
    LTMI_FAIL(LTMI_E_INVALID, "k_dense_fold16: no kernel for %d + %d groups", f->nge, f->ngo);
}

int ltmi::launch_fold16(ltmi_masks *m, const void *tile, int px_bytes, bool is_signed, int64_t n_frames, int64_t ld,
                        float *out, int64_t ld_out, int accumulate, hipStream_t stream) {
    if (px_bytes == 1)
        return is_signed ? launch_fold16_any<int8_t>(m, (const int8_t *)tile, n_frames, ld, out, ld_out, accumulate, stream)
                         : launch_fold16_any<uint8_t>(m, (const uint8_t *)tile, n_frames, ld, out, ld_out, accumulate, stream);
    return is_signed ? launch_fold16_any<int16_t>(m, (const int16_t *)tile, n_frames, ld, out, ld_out, accumulate, stream)
                     : launch_fold16_any<uint16_t>(m, (const uint16_t *)tile, n_frames, ld, out, ld_out, accumulate, stream);
}

// ---- banded stacks: column blocks with a pixel support each (k_dense_fold<.., LIST>) -----------------------------
//
// A radial-Fourier stack with several bins (analysis/radialfourier.py:106-146 with n_bins > 1, use_sparse=True:
// SURVEY.md 8(d)'s second C5 run, 16 bins x 25 orders) arrives as a CSR matrix, but it is not sparse in the sense
// of the gather / blocked kernels: the 25 complex masks of a bin share ONE support (the bin's ring) and are dense on
// it -- a dense 50-column stack per ring.  The blocked image multiplies it record by record (8 pixels x 16 columns,
// 0.35 - 0.44 of the float32 matrix peak, profiles/r05_sparse.txt 8); here every block of columns with a common
// support becomes a folded dense image over the 64-pixel stages that touch the support, and k_dense_fold walks the
// block's stage list.  Stages at a ring's edge are read by both neighbours.
struct ltmi::KeptCsr {
    int nc = 1;
    int64_t n_px = 0, n_masks = 0;
    std::vector<int64_t> indptr;
    std::vector<int32_t> idx;                   // rows sorted by mask index, no duplicates
    std::vector<float> val;                     // nc floats per entry
};

struct BandImage {
    int sig_h = 0, sig_w = 0, c2 = 0, nge = 0, ngo = 0;
    int n_blocks = 0, n_stages = 0, n_fold_rows = 0;
    float *img = nullptr;
    int4 *stages = nullptr;
    int *blk_off = nullptr, *colmap = nullptr;
    unsigned char *zeros = nullptr;
    double reread = 0.;                         // stages per stage of the whole detector: how often a frame byte is read
    // 1- / 2-byte integer frames: 128-pixel stages (k_dense_fold16); the lists are made with the float32 image, the
    // image itself on the first integer tile (k_build_band_image16)
    std::vector<int4> stages16_host;
    std::vector<int4> src16_host;                // two per stage: the 64-pixel stages it is made of, their first pixels
    std::vector<int> blk_off16_host;
    float *img16 = nullptr;
    int4 *stages16 = nullptr;
    int *blk_off16 = nullptr;
    int n_stages16 = 0;
    bool img16_failed = false;
    double reread16 = 0.;
};

ltmi::KeptCsr *ltmi::band_keep_csr(const int64_t *indptr, const int64_t *indices, const float *vals, int nc,
                                   int64_t n_px, int64_t n_masks) {
    const char *off = getenv("LTMI_SPARSE_BAND");
    if (off && atoi(off) == 0) return nullptr;
    const int64_t nnz = indptr[n_px];
    if (nnz <= 0 || n_masks * nc < 2 * GROUP || n_masks >= (1 << 30)) return nullptr;
    KeptCsr *k = new (std::nothrow) KeptCsr();
    if (!k) return nullptr;
    try {
        k->nc = nc; k->n_px = n_px; k->n_masks = n_masks;
        k->indptr.assign(indptr, indptr + n_px + 1);
        k->idx.resize((size_t)nnz);
        k->val.resize((size_t)nnz * nc);
        std::vector<std::pair<int32_t, int64_t>> row;
        for (int64_t p = 0; p < n_px; ++p) {
            const int64_t e0 = indptr[p], e1 = indptr[p + 1];
            bool sorted = true;
            for (int64_t e = e0 + 1; e < e1 && sorted; ++e) sorted = indices[e - 1] < indices[e];
            if (sorted) {
                for (int64_t e = e0; e < e1; ++e) {
                    k->idx[(size_t)e] = (int32_t)indices[e];
                    for (int c = 0; c < nc; ++c) k->val[(size_t)e * nc + c] = vals[e * nc + c];
                }
                continue;
            }
            row.clear();
            for (int64_t e = e0; e < e1; ++e) row.emplace_back((int32_t)indices[e], e);
            std::sort(row.begin(), row.end());
            for (size_t i = 0; i < row.size(); ++i) {
                if (i && row[i].first == row[i - 1].first) { delete k; return nullptr; }      // duplicates: not here
                k->idx[(size_t)e0 + i] = row[i].first;
                for (int c = 0; c < nc; ++c) k->val[((size_t)e0 + i) * nc + c] = vals[row[i].second * nc + c];
            }
        }
    } catch (const std::exception &) {          // (bad_alloc, length_error: no image)
        delete k;
        return nullptr;
    }
    return k;
}

void ltmi::band_free_csr(KeptCsr *k) { delete k; }

void ltmi::band_destroy(void *band) {
    BandImage *b = (BandImage *)band;
    if (!b) return;
    if (b->img) (void)hipFree(b->img);
    if (b->stages) (void)hipFree(b->stages);
    if (b->blk_off) (void)hipFree(b->blk_off);
    if (b->colmap) (void)hipFree(b->colmap);
    if (b->zeros) (void)hipFree(b->zeros);
    if (b->img16) (void)hipFree(b->img16);
    if (b->stages16) (void)hipFree(b->stages16);
    if (b->blk_off16) (void)hipFree(b->blk_off16);
    delete b;
}

// `other_macs`: padded multiply-adds per frame of the kernel that would run instead (blocked image), 0: unknown.
// Returns the image or nullptr (no mirror, no common supports, not worth it, no memory: the other kernels serve).
static void *band_no(int why) {
    if (getenv("LTMI_BAND_DEBUG")) fprintf(stderr, "band_build: no image (exit %d)\n", why);
    return nullptr;
}

void *ltmi::band_build(const KeptCsr *k, int sig_h, int sig_w, double other_macs) {
    if (!k || (int64_t)sig_h * sig_w != k->n_px || sig_w % FD_KB != 0 || sig_h < 4) return band_no(1);
    const int nc = k->nc;
    const int64_t n_cols = k->n_masks * nc;
    const int64_t nnz = k->indptr[(size_t)k->n_px];
    try {
        // ---- row mirror under which every real column is even or odd (the test of fold_create, on the CSR rows) ----
        int best_c2 = -1;
        std::vector<signed char> cls;
        for (int c2 : {sig_h, sig_h - 1, sig_h + 1}) {
            std::vector<unsigned char> even((size_t)n_cols, 1), odd((size_t)n_cols, 1);
            for (int y = 0; y < sig_h; ++y) {
                const int y2 = c2 - y;
                if (y2 <= y || y2 >= sig_h) continue;
                for (int x = 0; x < sig_w; ++x) {
                    const int64_t p = (int64_t)y * sig_w + x, q = (int64_t)y2 * sig_w + x;
                    int64_t a = k->indptr[(size_t)p], a1 = k->indptr[(size_t)p + 1];
                    int64_t b = k->indptr[(size_t)q], b1 = k->indptr[(size_t)q + 1];
                    while (a < a1 || b < b1) {
                        const int32_t ia = a < a1 ? k->idx[(size_t)a] : INT32_MAX;
                        const int32_t ib = b < b1 ? k->idx[(size_t)b] : INT32_MAX;
                        const int32_t col = std::min(ia, ib);
                        for (int c = 0; c < nc; ++c) {
                            const float va = ia == col ? k->val[(size_t)a * nc + c] : 0.f;
                            const float vb = ib == col ? k->val[(size_t)b * nc + c] : 0.f;
                            if (!(va == vb)) even[(size_t)col * nc + c] = 0;
                            if (!(va == -vb)) odd[(size_t)col * nc + c] = 0;
                        }
                        if (ia == col) ++a;
                        if (ib == col) ++b;
                    }
                }
            }
            bool all = true;
            std::vector<signed char> c((size_t)n_cols);
            for (int64_t j = 0; j < n_cols && all; ++j) {
                if (even[(size_t)j]) c[(size_t)j] = 1;
                else if (odd[(size_t)j]) c[(size_t)j] = -1;
                else all = false;
            }
            if (all) { best_c2 = c2; cls = c; break; }
        }
        if (best_c2 < 0) return band_no(2);

        // ---- masks with the same support form a block ----
        std::vector<uint64_t> hash((size_t)k->n_masks, 1469598103934665603ull);
        std::vector<int64_t> cnt((size_t)k->n_masks, 0), first((size_t)k->n_masks, -1), last((size_t)k->n_masks, -1);
        for (int64_t p = 0; p < k->n_px; ++p)
            for (int64_t e = k->indptr[(size_t)p]; e < k->indptr[(size_t)p + 1]; ++e) {
                const size_t mk = (size_t)k->idx[(size_t)e];
                hash[mk] = (hash[mk] ^ (uint64_t)(p + 1)) * 1099511628211ull;
                if (cnt[mk]++ == 0) first[mk] = p;
                last[mk] = p;
            }
        struct Block { std::vector<int32_t> masks; int n_even = 0, n_odd = 0; };
        std::vector<Block> blocks;
        {
            std::vector<int32_t> order((size_t)k->n_masks);
            for (int64_t i = 0; i < k->n_masks; ++i) order[(size_t)i] = (int32_t)i;
            auto key_less = [&](int32_t a, int32_t b) {
                if (hash[(size_t)a] != hash[(size_t)b]) return hash[(size_t)a] < hash[(size_t)b];
                if (cnt[(size_t)a] != cnt[(size_t)b]) return cnt[(size_t)a] < cnt[(size_t)b];
                if (first[(size_t)a] != first[(size_t)b]) return first[(size_t)a] < first[(size_t)b];
                if (last[(size_t)a] != last[(size_t)b]) return last[(size_t)a] < last[(size_t)b];
                return a < b;
            };
            std::sort(order.begin(), order.end(), key_less);
            auto same = [&](int32_t a, int32_t b) {
                return hash[(size_t)a] == hash[(size_t)b] && cnt[(size_t)a] == cnt[(size_t)b] &&
                       first[(size_t)a] == first[(size_t)b] && last[(size_t)a] == last[(size_t)b];
            };
            for (size_t i = 0; i < order.size(); ++i) {
                const int32_t mk = order[i];
                if (cnt[(size_t)mk] == 0) return band_no(3);          // a mask without entries: its column is nobody's
                int ne = 0, no = 0;
                for (int c = 0; c < nc; ++c) (cls[(size_t)mk * nc + c] > 0 ? ne : no)++;
                const bool fresh = i == 0 || !same(order[i - 1], mk) ||
                                   blocks.back().n_even + ne > 2 * GROUP || blocks.back().n_odd + no > 2 * GROUP;
                if (fresh) blocks.emplace_back();
                blocks.back().masks.push_back(mk);
                blocks.back().n_even += ne;
                blocks.back().n_odd += no;
            }
        }
        if (blocks.size() > 4096) return band_no(4);
        int nge = 1, ngo = 0;
        for (const Block &b : blocks) {
            nge = std::max(nge, (b.n_even + GROUP - 1) / GROUP);
            ngo = std::max(ngo, (b.n_odd + GROUP - 1) / GROUP);
        }
        const int ng = nge + ngo;

        // ---- folded rows; per block and folded row the WINDOWS of 64 (128: integer frames) pixels that cover its support ----
        // A window need not start on a grid of whole windows -- the copies take any 16-byte boundary -- so a run of the
        // support that is shorter than a window can cost one stage wherever it lies; but rows that start inside a
        // 256-byte piece cost the copy path more per stage.  Measured on 1024 x 1024 float32, 25 orders (stages, ms per
        // 8192 frames at 16 bins / per 4096 frames below):
        //     bins   grid of 64 px       starts at 32 px      starts at 16 px
        //      16    15 771  12.1        14 263  11.65        12 656  11.38
        //       8    11 897   4.82       10 835   4.64        10 508   4.86
        //       4     9 969   4.27        9 627   4.31         9 485   4.53
        //       2     8 942   3.59        8 783   3.67         8 783   3.90
        // (uint16, 128-pixel windows, 16 bins: grid 11 631 15.4 ms, 64 px 9 139 12.6 ms, 16 px 8 715 12.3 ms) -- a stage
        // costs 1.065 / 1.17 (uint16: 1.045 / 1.065) of a grid stage; the builder covers the support with all three and
        // keeps the cheapest by that count.  Windows of one row may overlap (the start is rounded down, the last one is
        // pushed back inside the row): a pixel belongs to the FIRST window that covers it (`own` = first pixel a window
        // owns), the later one holds weight 0 there.
        std::vector<int2> rows;
        std::vector<int> fold_row_of((size_t)sig_h, -1);
        for (int y = 0; y < sig_h; ++y) {
            const int y2 = best_c2 - y;
            if (y2 >= 0 && y2 < sig_h && y2 < y) { fold_row_of[(size_t)y] = fold_row_of[(size_t)y2]; continue; }
            fold_row_of[(size_t)y] = (int)rows.size();
            rows.push_back(int2{y, (y2 > y && y2 < sig_h) ? y2 : -1});
        }
        const int spr = sig_w / FD_KB;
        const size_t n_slots = rows.size() * (size_t)spr;
        std::vector<int32_t> block_of((size_t)k->n_masks), vcol((size_t)n_cols);
        for (size_t b = 0; b < blocks.size(); ++b) {
            int ie = 0, io = nge * GROUP;
            for (int32_t mk : blocks[b].masks) {
                block_of[(size_t)mk] = (int32_t)b;
                for (int c = 0; c < nc; ++c)
                    vcol[(size_t)mk * nc + c] = cls[(size_t)mk * nc + c] > 0 ? ie++ : io++;
            }
        }
        struct Win { int fy, x0, own; };
        // candidates: (start alignment in pixels, measured cost of a stage relative to the grid's)
        constexpr int N_ALIGN = 3;
        const int aligns[N_ALIGN] = {FD_KB, 32, 16}, aligns16[N_ALIGN] = {FD16_KB, 64, 16};
        const double stage_cost[N_ALIGN] = {1.0, 1.065, 1.17}, stage_cost16[N_ALIGN] = {1.0, 1.045, 1.065};
        const bool with16 = sig_w % FD16_KB == 0;
        std::vector<std::vector<Win>> wins(blocks.size()), wins16(blocks.size());
        std::vector<std::vector<int>> row_first(blocks.size());      // [fy]: first window of the row in wins[b] (+ end)
        std::vector<std::vector<Win>> cand[N_ALIGN], cand16[N_ALIGN];
        std::vector<std::vector<int>> cand_first[N_ALIGN];
        for (int a = 0; a < N_ALIGN; ++a) {
            cand[a].resize(blocks.size());
            cand16[a].resize(blocks.size());
            cand_first[a].resize(blocks.size());
            for (size_t b = 0; b < blocks.size(); ++b) cand_first[a][b].assign(rows.size() + 1, 0);
        }
        {
            std::vector<std::vector<int>> xs(blocks.size());
            std::vector<int32_t> hit;                                // blocks with pixels in the current folded row
            auto cover = [&](const std::vector<int> &sx, int width, int align, int fy, std::vector<Win> &out) {
                int end = -1;
                for (int x : sx)
                    if (x >= end) {
                        const int x0 = std::min(x / align * align, sig_w - width);
                        out.push_back(Win{fy, x0, std::max(x0, end)});
                        end = x0 + width;
                    }
            };
            for (size_t fy = 0; fy < rows.size(); ++fy) {
                hit.clear();
                for (int which = 0; which < 2; ++which) {
                    const int y = which == 0 ? rows[fy].x : rows[fy].y;
                    if (y < 0) continue;
                    for (int x = 0; x < sig_w; ++x) {
                        const int64_t p = (int64_t)y * sig_w + x;
                        for (int64_t e = k->indptr[(size_t)p]; e < k->indptr[(size_t)p + 1]; ++e) {
                            const int32_t b = block_of[(size_t)k->idx[(size_t)e]];
                            std::vector<int> &v = xs[(size_t)b];
                            if (v.empty()) hit.push_back(b);
                            if (v.empty() || v.back() != x) v.push_back(x);
                        }
                    }
                }
                for (int32_t b : hit) {
                    std::vector<int> &v = xs[(size_t)b];
                    std::sort(v.begin(), v.end());
                    v.erase(std::unique(v.begin(), v.end()), v.end());
                    for (int a = 0; a < N_ALIGN; ++a) {
                        cover(v, FD_KB, aligns[a], (int)fy, cand[a][(size_t)b]);
                        if (with16) cover(v, FD16_KB, aligns16[a], (int)fy, cand16[a][(size_t)b]);
                    }
                    v.clear();
                }
                for (int a = 0; a < N_ALIGN; ++a)
                    for (size_t b = 0; b < blocks.size(); ++b) cand_first[a][b][fy + 1] = (int)cand[a][b].size();
            }
        }
        size_t total = 0, total16 = 0;
        {
            int best = 0, best16 = 0;
            double c_best = 0., c_best16 = 0.;
            for (int a = 0; a < N_ALIGN; ++a) {
                size_t n = 0, n16 = 0;
                for (size_t b = 0; b < blocks.size(); ++b) { n += cand[a][b].size(); n16 += cand16[a][b].size(); }
                if (a == 0 || (double)n * stage_cost[a] < c_best) { best = a; c_best = (double)n * stage_cost[a]; total = n; }
                if (a == 0 || (double)n16 * stage_cost16[a] < c_best16) {
                    best16 = a; c_best16 = (double)n16 * stage_cost16[a]; total16 = n16;
                }
            }
            wins.swap(cand[best]);
            row_first.swap(cand_first[best]);
            wins16.swap(cand16[best16]);
            for (int a = 0; a < N_ALIGN; ++a) {
                std::vector<std::vector<Win>>().swap(cand[a]);
                std::vector<std::vector<Win>>().swap(cand16[a]);
                std::vector<std::vector<int>>().swap(cand_first[a]);
            }
        }
        // worth it?  per frame: matrix work at the fold kernel's rate and the frame bytes it reads (edge stages once per
        // block) against the blocked image's record loop (0.35 - 0.44 of the matrix peak; unknown: the vector ALUs)
        const double band_macs = (double)total * FD_KB * ng * GROUP;
        const double reread = (double)total / (double)n_slots;
        const double t_band = std::max(band_macs * 2. / 130e12, reread * (double)k->n_px * 4. / 6.5e12);
        const double t_other = other_macs > 0. ? other_macs * 2. / 60e12 : (double)nnz * nc / 10e12;
        const char *force = getenv("LTMI_SPARSE_BAND");
        if (!(force && atoi(force) == 1) && !(t_band < 0.8 * t_other)) return band_no(5);
        if (total == 0 || total > (size_t)INT32_MAX / 2) return band_no(6);

        // blocks in the order of their length, longest first (they are dispatched in that order)
        std::vector<int32_t> border(blocks.size());
        for (size_t b = 0; b < blocks.size(); ++b) border[b] = (int32_t)b;
        std::stable_sort(border.begin(), border.end(),
                         [&](int32_t a, int32_t b) { return wins[(size_t)a].size() > wins[(size_t)b].size(); });
        std::vector<int> blk_off(blocks.size() + 1, 0), colmap(blocks.size() * (size_t)ng * GROUP, -1);
        std::vector<int> first_stage(blocks.size(), 0);              // block -> its first stage in the final order
        std::vector<int4> stages(total);
        {
            size_t s = 0;
            for (size_t bi = 0; bi < border.size(); ++bi) {
                const size_t b = (size_t)border[bi];
                blk_off[bi] = (int)s;
                first_stage[b] = (int)s;
                for (const Win &w : wins[b]) stages[s++] = int4{rows[(size_t)w.fy].x, rows[(size_t)w.fy].y, w.x0, w.own};
                for (int32_t mk : blocks[b].masks)
                    for (int c = 0; c < nc; ++c)
                        colmap[bi * (size_t)ng * GROUP + (size_t)vcol[(size_t)mk * nc + c]] = (int)(mk * nc + c);
            }
            blk_off[blocks.size()] = (int)s;
        }
        // 128-pixel windows of the same blocks (integer frames): the image is made on the device from the float32 one,
        // every window from the (at most four) 64-pixel windows of its block and row that reach into it
        std::vector<int4> stages16, src16;
        std::vector<int> blk_off16(blocks.size() + 1, 0);
        bool table_ok = true;
        if (with16) {
            stages16.reserve(total16);
            src16.reserve(2 * total16);
            for (size_t bi = 0; bi < border.size(); ++bi) {
                const size_t b = (size_t)border[bi];
                blk_off16[bi] = (int)stages16.size();
                for (const Win &w : wins16[b]) {
                    stages16.push_back(int4{rows[(size_t)w.fy].x, rows[(size_t)w.fy].y, w.x0, w.own});
                    int ids[4] = {-1, -1, -1, -1}, x0s[4] = {0, 0, 0, 0}, n = 0;
                    for (int j = row_first[b][(size_t)w.fy]; j < row_first[b][(size_t)w.fy + 1]; ++j) {
                        const Win &v = wins[b][(size_t)j];                 // owns [v.own, v.x0 + 64)
                        if (v.own < w.x0 + FD16_KB && v.x0 + FD_KB > w.own) {
                            if (n == 4) { table_ok = false; break; }       // (a support in many short runs)
                            ids[n] = first_stage[b] + j;
                            x0s[n++] = v.x0;
                        }
                    }
                    src16.push_back(int4{ids[0], ids[1], ids[2], ids[3]});
                    src16.push_back(int4{x0s[0], x0s[1], x0s[2], x0s[3]});
                }
            }
            blk_off16[blocks.size()] = (int)stages16.size();
            if (!table_ok) {                                               // integer frames: the other kernels
                stages16.clear();
                src16.clear();
            }
        }
        // image: the ORIGINAL weights of rows y (the first row of a pair), every pixel in the window that owns it
        const size_t slot_floats = (size_t)ng * GROUP * FD_KB;
        std::vector<float> img(total * slot_floats, 0.f);
        for (size_t fy = 0; fy < rows.size(); ++fy) {
            const int y = rows[fy].x;
            for (int x = 0; x < sig_w; ++x) {
                const int64_t p = (int64_t)y * sig_w + x;
                int32_t last_b = -1;
                size_t s = 0;
                int q = 0;
                for (int64_t e = k->indptr[(size_t)p]; e < k->indptr[(size_t)p + 1]; ++e) {
                    const int32_t mk = k->idx[(size_t)e];
                    const int32_t b = block_of[(size_t)mk];
                    if (b != last_b) {
                        // the window of (b, fy) that owns x: the last one whose `own` is <= x
                        int j = row_first[(size_t)b][fy + 1] - 1;
                        while (j > row_first[(size_t)b][fy] && wins[(size_t)b][(size_t)j].own > x) --j;
                        s = (size_t)first_stage[(size_t)b] + (size_t)j;
                        q = x - wins[(size_t)b][(size_t)j].x0;
                        last_b = b;
                    }
                    for (int c = 0; c < nc; ++c) {
                        const int v = vcol[(size_t)mk * nc + c];
                        img[(s * ng + (size_t)(v / GROUP)) * (GROUP * FD_KB) + fold_index(v % GROUP, q)] =
                            k->val[(size_t)e * nc + c];
                    }
                }
            }
        }
        BandImage *bi = new BandImage();
        bi->sig_h = sig_h; bi->sig_w = sig_w; bi->c2 = best_c2; bi->nge = nge; bi->ngo = ngo;
        bi->n_blocks = (int)blocks.size(); bi->n_stages = (int)total; bi->n_fold_rows = (int)rows.size();
        bi->reread = reread;
        bi->stages16_host.swap(stages16);
        bi->src16_host.swap(src16);
        bi->blk_off16_host.swap(blk_off16);
        bi->n_stages16 = (int)bi->stages16_host.size();
        bi->reread16 = with16 ? (double)bi->n_stages16 / (double)(rows.size() * (size_t)(sig_w / FD16_KB)) : 0.;
        hipError_t e = hipMalloc((void **)&bi->img, img.size() * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void **)&bi->stages, stages.size() * sizeof(int4));
        if (e == hipSuccess) e = hipMalloc((void **)&bi->blk_off, blk_off.size() * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **)&bi->colmap, colmap.size() * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **)&bi->zeros, 1024);
        if (e == hipSuccess) e = hipMemset(bi->zeros, 0, 1024);
        if (e == hipSuccess) e = hipMemcpy(bi->img, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(bi->stages, stages.data(), stages.size() * sizeof(int4), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(bi->blk_off, blk_off.data(), blk_off.size() * sizeof(int), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(bi->colmap, colmap.data(), colmap.size() * sizeof(int), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            band_destroy(bi);
            (void)hipGetLastError();
            return band_no(7);
        }
        return bi;
    } catch (const std::exception &) {          // (bad_alloc, length_error: no image)
        return band_no(8);
    }
}

bool ltmi::band_takes(void *band, const ltmi_masks *m, const void *tile, int tile_dtype, int64_t ld) {
    BandImage *b = (BandImage *)band;
    if (!b || m->tune_ksplit_ring == 41 || m->tune_ksplit_ring == 42) return false;
    if ((uintptr_t)tile % 16 != 0) return false;
    if (tile_dtype == LTMI_F32) return (ld * 4) % 16 == 0;
    const int px_bytes = (tile_dtype == LTMI_U8 || tile_dtype == LTMI_I8)
                             ? 1
                             : ((tile_dtype == LTMI_U16 || tile_dtype == LTMI_I16) ? 2 : 0);
    if (!px_bytes || (ld * px_bytes) % 16 != 0 || b->n_stages16 == 0 || b->img16_failed) return false;
    if (!b->img16) {
        // first integer tile: the image in 128-pixel slots from the float32 one
        const int ng = b->nge + b->ngo;
        int4 *src = nullptr;
        hipError_t e = hipMalloc((void **)&b->img16, (size_t)b->n_stages16 * ltmi::fold16_slot_bytes(ng));
        if (e == hipSuccess) e = hipMalloc((void **)&b->stages16, b->stages16_host.size() * sizeof(int4));
        if (e == hipSuccess) e = hipMalloc((void **)&b->blk_off16, b->blk_off16_host.size() * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **)&src, b->src16_host.size() * sizeof(int4));
        if (e == hipSuccess)
            e = hipMemcpy(b->stages16, b->stages16_host.data(), b->stages16_host.size() * sizeof(int4),
                          hipMemcpyHostToDevice);
        if (e == hipSuccess)
            e = hipMemcpy(b->blk_off16, b->blk_off16_host.data(), b->blk_off16_host.size() * sizeof(int),
                          hipMemcpyHostToDevice);
        if (e == hipSuccess)
            e = hipMemcpy(src, b->src16_host.data(), b->src16_host.size() * sizeof(int4), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            const int64_t tot = (int64_t)b->n_stages16 * ng * GROUP * FD16_KB;
            const unsigned bl = (unsigned)std::min<int64_t>((tot + 255) / 256, 65535 * 16);
            hipLaunchKernelGGL(ltmi::k_build_band_image16, dim3(bl), dim3(256), 0, 0, (const float *)b->img, b->img16,
                               (const int4 *)b->stages16, (const int4 *)src, ng, (int64_t)b->n_stages16);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipDeviceSynchronize();
        }
        if (src) (void)hipFree(src);
        if (e != hipSuccess) {
            if (b->img16) (void)hipFree(b->img16);
            if (b->stages16) (void)hipFree(b->stages16);
            if (b->blk_off16) (void)hipFree(b->blk_off16);
            b->img16 = nullptr;
            b->stages16 = nullptr;
            b->blk_off16 = nullptr;
            b->img16_failed = true;
            (void)hipGetLastError();
            return false;
        }
        std::vector<int4>().swap(b->stages16_host);
        std::vector<int4>().swap(b->src16_host);
    }
    return true;
}

// parts per block: enough workgroups for two rounds on the 256 CUs, at least 32 stages each (tuning: ksplit forced)
static int band_ksplit(const ltmi_masks *m, int64_t gx, int n_blocks, int n_stages) {
    if (m->tune_ksplit > 0) return std::min(m->tune_ksplit, 64);
    const int64_t wgs = gx * n_blocks;
    int k = (int)std::min<int64_t>(16, (512 + wgs - 1) / wgs);
    const int avg = std::max(1, n_stages / std::max(1, n_blocks));
    k = std::min(k, std::max(1, avg / 32));
    return std::max(1, k);
}

template <int NGE, int NGO>
static int launch_band_t(ltmi_masks *m, const BandImage *b, const float *tile, int64_t n_frames, int64_t ld, float *out,
                         int64_t ld_out, int n_cols, int accumulate, hipStream_t stream) {
    auto kern = k_dense_fold<NGE, NGO, 0, true>;
    constexpr int LDS = ltmi::fold_lds_bytes(NGE + NGO);
    static bool attr_set[16] = {false};
    if (!attr_set[m->device & 15]) {
        LTMI_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set[m->device & 15] = true;
    }
    const int64_t gx = (n_frames + FD_WG_ROWS - 1) / FD_WG_ROWS;
    const int ksplit = band_ksplit(m, gx, b->n_blocks, b->n_stages);
    if (ksplit > 1) {
        const int rc = dense_ensure_partials(m, (size_t)ksplit * n_frames * n_cols * sizeof(float), stream);
        if (rc != LTMI_OK) return rc;
    }
    dim3 grid((unsigned)gx, (unsigned)(b->n_blocks * ksplit));
    hipLaunchKernelGGL(kern, grid, dim3(FD_WAVES * 64), LDS, stream, tile, ld, n_frames, b->sig_w / FD_KB,
                       (const int2 *)nullptr, (const float *)b->img, b->n_stages, out, ld_out, n_cols,
                       (const int *)b->colmap, accumulate, dense_partial_sums(m), ksplit,
                       (const unsigned char *)b->zeros, m->roi_rows, (const int4 *)b->stages, (const int *)b->blk_off);
    LTMI_HIP(hipGetLastError());
    if (ksplit > 1) {
        const int rc = dense_reduce_partials(m, ksplit, n_frames, out, ld_out, accumulate, stream, n_cols);
        if (rc != LTMI_OK) return rc;
    }
    snprintf(m->last_kernel, sizeof(m->last_kernel),
             "k_dense_fold<f,even=%d,odd=%d,banded: %d blocks, %d stages (x%.2f), rows %d+%d=%d%s> grid=(%u,%u)", NGE, NGO,
             b->n_blocks, b->n_stages, b->reread, b->n_fold_rows, b->sig_h - b->n_fold_rows, b->c2,
             m->roi_rows ? ",rows" : "", grid.x, grid.y);
    return LTMI_OK;
}

template <typename T, int NGE, int NGO>
static int launch_band16_t(ltmi_masks *m, const BandImage *b, const T *tile, int64_t n_frames, int64_t ld, float *out,
                           int64_t ld_out, int n_cols, int accumulate, hipStream_t stream) {
    auto kern = k_dense_fold16<T, NGE, NGO, true>;
    constexpr int LDS = ltmi::fold16_lds_bytes(NGE + NGO, (int)sizeof(T));
    static bool attr_set[16] = {false};
    if (!attr_set[m->device & 15]) {
        LTMI_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set[m->device & 15] = true;
    }
    const int64_t gx = (n_frames + FD_WG_ROWS - 1) / FD_WG_ROWS;
    const int ksplit = band_ksplit(m, gx, b->n_blocks, b->n_stages16);
    if (ksplit > 1) {
        const int rc = dense_ensure_partials(m, (size_t)ksplit * n_frames * n_cols * sizeof(float), stream);
        if (rc != LTMI_OK) return rc;
    }
    dim3 grid((unsigned)gx, (unsigned)(b->n_blocks * ksplit));
    hipLaunchKernelGGL(kern, grid, dim3(FD_WAVES * 64), LDS, stream, tile, ld, n_frames, b->sig_w / FD16_KB,
                       (const int2 *)nullptr, (const float *)b->img16, b->n_stages16, out, ld_out, n_cols,
                       (const int *)b->colmap, accumulate, dense_partial_sums(m), ksplit,
                       (const unsigned char *)b->zeros, m->roi_rows, (const int4 *)b->stages16,
                       (const int *)b->blk_off16);
    LTMI_HIP(hipGetLastError());
    if (ksplit > 1) {
        const int rc = dense_reduce_partials(m, ksplit, n_frames, out, ld_out, accumulate, stream, n_cols);
        if (rc != LTMI_OK) return rc;
    }
    snprintf(m->last_kernel, sizeof(m->last_kernel),
             "k_dense_fold16<%s,even=%d,odd=%d,banded: %d blocks, %d stages (x%.2f), rows %d+%d=%d%s> grid=(%u,%u)",
             typeid(T).name(), NGE, NGO, b->n_blocks, b->n_stages16, b->reread16, b->n_fold_rows,
             b->sig_h - b->n_fold_rows, b->c2, m->roi_rows ? ",rows" : "", grid.x, grid.y);
    return LTMI_OK;
}

template <typename T>
static int band_apply16(ltmi_masks *m, const BandImage *b, const T *tile, int64_t n_frames, int64_t ld, float *out,
                        int64_t ld_out, int n_cols, int accumulate, hipStream_t stream) {
#define LTMI_BAND_CASE(E_, O_)                                                                                 \
    if (b->nge == E_ && b->ngo == O_)                                                                          \
        return launch_band16_t<T, E_, O_>(m, b, tile, n_frames, ld, out, ld_out, n_cols, accumulate, stream);
    LTMI_BAND_CASE(1, 0) LTMI_BAND_CASE(2, 0) LTMI_BAND_CASE(1, 1) LTMI_BAND_CASE(2, 1) LTMI_BAND_CASE(1, 2)
    LTMI_BAND_CASE(2, 2)
#undef LTMI_BAND_CASE
    LTMI_FAIL(LTMI_E_INVALID, "k_dense_fold16 (banded): no kernel for %d + %d groups", b->nge, b->ngo);
}

int ltmi::band_apply(ltmi_masks *m, void *band, const void *tile, int tile_dtype, int64_t n_frames, int64_t ld,
                     float *out, int64_t ld_out, int n_cols, int accumulate, hipStream_t stream) {
    const BandImage *b = (const BandImage *)band;
    switch (tile_dtype) {
        case LTMI_U8:
            return band_apply16<uint8_t>(m, b, (const uint8_t *)tile, n_frames, ld, out, ld_out, n_cols, accumulate, stream);
        case LTMI_I8:
            return band_apply16<int8_t>(m, b, (const int8_t *)tile, n_frames, ld, out, ld_out, n_cols, accumulate, stream);
        case LTMI_U16:
            return band_apply16<uint16_t>(m, b, (const uint16_t *)tile, n_frames, ld, out, ld_out, n_cols, accumulate, stream);
        case LTMI_I16:
            return band_apply16<int16_t>(m, b, (const int16_t *)tile, n_frames, ld, out, ld_out, n_cols, accumulate, stream);
        default: break;
    }
    const float *ft = (const float *)tile;
#define LTMI_BAND_CASE(E_, O_)                                                                                 \
    if (b->nge == E_ && b->ngo == O_)                                                                          \
        return launch_band_t<E_, O_>(m, b, ft, n_frames, ld, out, ld_out, n_cols, accumulate, stream);
    LTMI_BAND_CASE(1, 0) LTMI_BAND_CASE(2, 0) LTMI_BAND_CASE(1, 1) LTMI_BAND_CASE(2, 1) LTMI_BAND_CASE(1, 2)
    LTMI_BAND_CASE(2, 2)
#undef LTMI_BAND_CASE
    LTMI_FAIL(LTMI_E_INVALID, "k_dense_fold (banded): no kernel for %d + %d groups", b->nge, b->ngo);
}
