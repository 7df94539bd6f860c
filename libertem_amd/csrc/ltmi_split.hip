// Dense float32 mask stacks x float32 frames on the bf16 matrix cores, float32-accurate ("split" path).
//
// The f32 matrix instruction (v_mfma_f32_16x16x4_f32: 2048 flop in 32 cycles) makes stacks with several
// 16-column groups matrix-pipe bound on float32 frames -- C5 (RadialFourierAnalysis defaults: 25 complex
// masks = 50 real columns on 1024 x 1024 float32 frames) ran at 0.70 of that pipe's peak and 0.55 of HBM
// (DESIGN.md section 4.1).  v_mfma_f32_16x16x32_bf16 does 16 384 flop in 16 cycles, and a float32 number
// is EXACTLY the sum of three bf16 numbers (8 + 8 + 8 significant bits, same exponent range):
//
//     x = x1 + x2 + x3      x1 = x with the low 16 bits cleared, x2 = (x - x1) likewise, x3 = x - x1 - x2
//     w = w1 + w2 + w3      (round-to-nearest pieces, made once when the image is built)
//
// Every product xi * wj of two bf16 numbers is exact in float32 and the matrix core sums them in
// float32, so   x w  =  x1 w1 + (x1 w2 + x2 w1) + (x2 w2 + x1 w3 + x3 w1)  + O(2^-24 |x w|):
// the three dropped terms (x2 w3, x3 w2, x3 w3) are below the rounding unit of the float32 result.
// Six bf16 instructions (96 cycles) replace eight f32 ones (256 cycles) per 16 frames x 16 columns x
// 32 pixels; splitting the frame fragment costs 5.5 VALU instructions per pixel and lane, beside the
// matrix pipe.  The kernel is the LDS-DMA pipeline of k_dense_lds (ltmi_dense.hip: wave-private frame
// ring filled by global_load_lds, counted vmcnt, one barrier per mask slot) with this inner product.
//
// Result: float32, within float32 rounding of the float64 product (tests/test_kernels_gpu.py), NOT
// bit-identical to k_dense_lds (different summation order, like any other tile shape).  Frames holding
// +-inf produce NaN where the f32 instruction produces +-inf (inf - inf in the split).
//
// STATUS: opt-in (ltmi_masks_set_tuning code 36 / LTMI_SPLIT=1), NOT the default: on C5 it takes 8.2 - 8.9 ms
// against the 7.7 - 7.9 ms of k_dense_lds -- the matrix pipe drops to 34 % busy, but the pre-split image is
// 6 bytes per weight in 64 padded columns and its slots, re-read by every workgroup through the LDS-DMA
// path, cost the frames their bandwidth (profiles/r03_split.txt).  Kept as the measured experiment and as
// a second implementation the float32 kernel is checked against (test_float32_frames_on_bf16_matrix_cores).
//
// Reference interface: the same `tile.reshape((n, -1)) @ masks` of udf/masks.py:79-83 as the other
// dense kernels; reached from ltmi_apply_masks for float32 tiles (ltmi_dense.hip launch_mfma).
#include <type_traits>
#include "ltmi_common.h"

namespace ltmi {

typedef float f32x4s __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void *lds_ptr_s;
typedef const __attribute__((address_space(1))) void *glb_ptr_s;

constexpr int SP_GROUP = 16;
constexpr int SP_ROWS = 16;                 // frames per MFMA tile
constexpr int SP_SUBB = 128;                // bytes of a frame row per sub-chunk = 32 float32 pixels = one block
#ifndef SP_VPM
#define SP_VPM 2                            // VALU instructions scheduled after every MFMA
#endif
constexpr int SP_BSLOT = 12288;             // bytes per mask slot: NG groups x 3 planes x KB px x 16 cols x 2 B
constexpr int SP_WAVES = 4;

// image: [group tile gt][slot][group g of NG][plane 3][unit u = px / 8][column n 16][8 bf16]
//   -> a wave's B fragment read (lane (n, kg) reads unit blk * 4 + kg) is 1 KiB contiguous
struct SplitImage {
    uint16_t *img = nullptr;
    int ng = 0, kb = 0, n_slots = 0, n_gt = 0;
    float *partials = nullptr;
    size_t partials_bytes = 0;
};

__device__ __forceinline__ uint16_t bf16_rn(float v) {           // round to nearest even, finite input
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_f(uint16_t h) { return __uint_as_float((unsigned)h << 16); }

__global__ void k_build_split(const float *__restrict__ src, uint16_t *__restrict__ img,
                              int64_t n_masks, int cpm, int64_t n_px, int n_slots, int ng, int kb) {
    const int64_t total = n_masks * cpm * n_px;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int part = (int)(i % cpm);
        const int64_t kp = i / cpm;
        const int64_t k = kp / n_px, p = kp % n_px;
        const int col = (int)(k * cpm + part);
        const int g = col / SP_GROUP, n = col % SP_GROUP;
        const int gt = g / ng, gl = g % ng;
        const int slot = (int)(p / kb), q = (int)(p % kb);
        const float w = src[i];
        const uint16_t w1 = bf16_rn(w);
        const float r1 = w - bf16_f(w1);
        const uint16_t w2 = bf16_rn(r1);
        const uint16_t w3 = bf16_rn(r1 - bf16_f(w2));
        const size_t slot_base = ((size_t)gt * n_slots + slot) * (SP_BSLOT / 2);
        const size_t plane_elems = (size_t)(kb / 8) * 16 * 8;
        const size_t at = ((size_t)(q >> 3) * 16 + n) * 8 + (q & 7);
        img[slot_base + (size_t)(gl * 3 + 0) * plane_elems + at] = w1;
        img[slot_base + (size_t)(gl * 3 + 1) * plane_elems + at] = w2;
        img[slot_base + (size_t)(gl * 3 + 2) * plane_elems + at] = w3;
    }
}

template <int I, int N, typename F> __device__ __forceinline__ void sp_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sp_static_for<I + 1, N>(f);
    }
}

// TILES 16-frame tiles per wave: 4 waves x 64 frames = 256 frames per workgroup (TILES = 4) is the
// production shape -- the mask slots are copied into LDS once per workgroup and pass, so their DMA
// traffic per frame byte halves against 128 frames (measured: frames + mask slots together move at
// ~7 TB/s whatever the mix, profiles/r03_split.txt); TILES = 2 serves short tiles.
template <int NG, int TILES> struct SplitCfg {
    static constexpr int KB = 128 / NG;                        // pixels per mask slot (12 KiB)
    static constexpr int RING = TILES == 4 ? 3 : 6;            // frame ring depth (sub-chunks of 32 px)
    static constexpr int BR = 4;                               // mask-slot ring depth
    static constexpr int WAVES = SP_WAVES;
    static constexpr int ROWS = SP_ROWS * TILES;
    static constexpr int ASLOT = ROWS * SP_SUBB;               // 4 / 8 KiB per wave and ring slot
    static constexpr int WG_ROWS = WAVES * ROWS;
    static constexpr int A_BYTES = RING * WAVES * ASLOT;       // 96 KiB
    static constexpr int LDS_BYTES = A_BYTES + BR * SP_BSLOT;  // 144 KiB
};

// The kernel.  One "block" = 32 pixels = one sub-chunk of the frame ring (128 B of a row) = one step
// of the software pipeline:
//     block i:   matrix instructions on  P(i) [bf16 pieces of the frame fragment] x B(i) [mask pieces]
//                VALU, in between:       split raw(i+1) -> P(i+1)
//                LDS reads, in between:  raw(i+2), B(i+1)
//                DMA, in between:        frame sub-chunk i + RING + 1 into the ring slot raw(i+1) came from
//                                        (and, where block i+1 opens a mask slot, the slot BR-1 slots
//                                        later, behind the barrier that frees its buffer)
// Bytes in flight per CU decide the speed (the DMA is latency bound): RING-1 frame sub-chunks and
// BR-2..BR-1 mask slots, ~90 KiB of the 144 KiB of LDS.
// so an MFMA never waits for something issued in its own block.  The main loop is unrolled over
// UNROLL = lcm(RING, BR PERB, 2) blocks (ring slot, mask buffer, register buffers: all static); the last
// (blocks % UNROLL) blocks of a workgroup's range run through a plain load-wait-multiply loop.
template <int NG, int TILES>
__global__ void __launch_bounds__(SP_WAVES * 64)
k_dense_split(const float *__restrict__ tile, int64_t ld, int64_t n_frames, int64_t n_px,
              const uint16_t *__restrict__ img, int n_slots, float *__restrict__ out, int64_t ld_out,
              int n_cols, int accumulate, float *__restrict__ partials, int ksplit) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    using CFG = SplitCfg<NG, TILES>;
    constexpr int WAVES = CFG::WAVES, RING = CFG::RING, KB = CFG::KB, ROWS = CFG::ROWS, BR = CFG::BR;
    constexpr int ASLOT = CFG::ASLOT, BSLOT = SP_BSLOT;
    constexpr int PPR = SP_SUBB / 16;                   // 8 pieces (16 B) per row of a sub-chunk
    constexpr int RPI = 64 / PPR;                       // 8 rows per DMA instruction
    constexpr int TILE_BYTES = SP_ROWS * SP_SUBB;       // 2 KiB: one 16-frame tile of a ring slot
    constexpr int ND = ROWS / RPI;                      // DMA instructions per sub-chunk and wave (4)
    constexpr int PERB = KB / 32;                       // blocks per mask slot
    constexpr int BPW = BSLOT / WAVES;
    constexpr int NBI = BPW / 1024;
    static_assert(BPW % 1024 == 0, "whole DMA instructions per wave");
    static_assert(ND * (RING - 2) + RING * NBI < 64 && ND * (BR - 1) * PERB + NBI * (BR - 2) < 64,
                  "vmcnt is a 6-bit counter");
    constexpr int PLANE = (KB / 8) * 16 * 16;           // bytes per (group, plane) of a slot

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int m = lane & 15, kg = lane >> 4;
    const int ks = blockIdx.y;
    const int gt = blockIdx.z;

    const int per = (n_slots + ksplit - 1) / ksplit;
    const int k_begin = ks * per;
    const int k_end = min(n_slots, k_begin + per);
    const uint16_t *img_t = img + (size_t)gt * n_slots * (BSLOT / 2);

    const int64_t f_wave = (int64_t)blockIdx.x * (WAVES * ROWS) + wave * ROWS;
    auto frame_of = [&](int r) -> int64_t {
        const int64_t f = f_wave + r;
        return f < n_frames ? f : -1;
    };
    unsigned char *a_base = lds_raw + wave * ASLOT;
    unsigned char *b_base = lds_raw + CFG::A_BYTES;

    f32x4s acc[TILES][NG], acc2[TILES][NG];
#pragma unroll
    for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[tl][g] = acc2[tl][g] = f32x4s{0.f, 0.f, 0.f, 0.f};

    // lane-constant address parts.  Frame ring: row r of a tile at r * 128 B, its 8 pieces (16 B = 4
    // pixels) stored at piece ^ ((r >> 1) & 7): the 16 lanes of a ds_read_b128 service group (16 rows, same
    // piece) spread over all banks.  The lane's 8 pixels of a block are pieces 2 kg and 2 kg + 1.
    const int a_lane0 = m * SP_SUBB + (((2 * kg) ^ ((m >> 1) & 7)) << 4);
    const int a_lane1 = m * SP_SUBB + (((2 * kg + 1) ^ ((m >> 1) & 7)) << 4);
    const int b_lane = (kg * 16 + m) * 16;              // bytes inside the 4 units of a block

    const int NB = (k_end - k_begin) * PERB;            // blocks of this workgroup
    if (NB > 0) {
        const unsigned char *src[ND];
#pragma unroll
        for (int t = 0; t < ND; ++t) {
            const int r = RPI * t + lane / PPR;
            int64_t f = frame_of(r);
            if (f < 0) f = n_frames - 1;                // (clamped; such results are discarded)
            const int piece = (lane & (PPR - 1)) ^ ((r >> 1) & (PPR - 1));
            src[t] = (const unsigned char *)(tile + f * ld + (int64_t)k_begin * KB) + piece * 16;
        }
        const unsigned char *bsrc = (const unsigned char *)img_t + (int64_t)k_begin * BSLOT +
                                    wave * BPW + lane * 16;
        const int n_bslots = k_end - k_begin;

        auto issue_a = [&](int sub, int slot) {         // frame sub-chunk `sub` (clamped) -> ring slot
            const int sc = min(sub, NB - 1);
            unsigned char *dst = a_base + slot * (WAVES * ASLOT);
#pragma unroll
            for (int t = 0; t < ND; ++t)
                __builtin_amdgcn_global_load_lds((glb_ptr_s)(src[t] + (int64_t)sc * SP_SUBB),
                                                 (lds_ptr_s)(dst + t * 1024), 16, 0, 2 /*nt*/);
        };
        auto issue_b = [&](int bs) {                    // mask slot `bs` (clamped) -> buffer bs % BR
            const int kk = min(bs, n_bslots - 1);
            unsigned char *dst = b_base + (bs % BR) * BSLOT + wave * BPW;
            const unsigned char *sp = bsrc + (int64_t)kk * BSLOT;
#pragma unroll
            for (int u = 0; u < NBI; ++u)
                __builtin_amdgcn_global_load_lds((glb_ptr_s)(sp + u * 1024),
                                                 (lds_ptr_s)(dst + u * 1024), 16, 0, 0);
        };
        auto rd_a = [&](int slot, f32x4s (&raw)[TILES][2]) {
            const unsigned char *at = a_base + slot * (WAVES * ASLOT);
#pragma unroll
            for (int tl = 0; tl < TILES; ++tl) {
                raw[tl][0] = *(const f32x4s *)(at + tl * TILE_BYTES + a_lane0);
                raw[tl][1] = *(const f32x4s *)(at + tl * TILE_BYTES + a_lane1);
            }
        };
        auto rd_b = [&](int buf, int blk_in_slot, u32x4s (&b)[NG][3]) {
            const unsigned char *bs = b_base + buf * BSLOT + b_lane + blk_in_slot * 1024;
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) b[g][pl] = *(const u32x4s *)(bs + (g * 3 + pl) * PLANE);
        };
        // 8 pixels -> three packed bf16 operands (see the header of this file)
        auto split = [&](const f32x4s (&raw)[TILES][2], u32x4s (&p)[TILES][3]) {
#pragma unroll
            for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const float xe = raw[tl][h >> 1][(2 * h) & 3];
                    const float xo = raw[tl][h >> 1][(2 * h + 1) & 3];
                    const unsigned ue = __float_as_uint(xe), uo = __float_as_uint(xo);
                    const float re = xe - __uint_as_float(ue & 0xffff0000u);
                    const float ro = xo - __uint_as_float(uo & 0xffff0000u);
                    const unsigned ure = __float_as_uint(re), uro = __float_as_uint(ro);
                    const float te = re - __uint_as_float(ure & 0xffff0000u);
                    const float to = ro - __uint_as_float(uro & 0xffff0000u);
                    p[tl][0][h] = __builtin_amdgcn_perm(uo, ue, 0x07060302u);
                    p[tl][1][h] = __builtin_amdgcn_perm(uro, ure, 0x07060302u);
                    p[tl][2][h] = __builtin_amdgcn_perm(__float_as_uint(to), __float_as_uint(te),
                                                        0x07060302u);
                }
        };
        auto multiply = [&](const u32x4s (&p)[TILES][3], const u32x4s (&b)[NG][3]) {
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int tl = 0; tl < TILES; ++tl) {
                    const bf16x8s x1 = __builtin_bit_cast(bf16x8s, p[tl][0]);
                    const bf16x8s x2 = __builtin_bit_cast(bf16x8s, p[tl][1]);
                    const bf16x8s x3 = __builtin_bit_cast(bf16x8s, p[tl][2]);
                    const bf16x8s w1 = __builtin_bit_cast(bf16x8s, b[g][0]);
                    const bf16x8s w2 = __builtin_bit_cast(bf16x8s, b[g][1]);
                    const bf16x8s w3 = __builtin_bit_cast(bf16x8s, b[g][2]);
                    f32x4s c = acc[tl][g];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x3, w1, c, 0, 0, 0);   // smallest first
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, w3, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x2, w2, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x2, w1, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, w2, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, w1, c, 0, 0, 0);
                    acc[tl][g] = c;
                }
        };
        auto flush = [&]() {                            // second accumulation level (see k_dense_lds)
#pragma unroll
            for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    acc2[tl][g] += acc[tl][g];
                    acc[tl][g] = f32x4s{0.f, 0.f, 0.f, 0.f};
                }
        };

        constexpr int U2 = BR * PERB;                   // (even)
        constexpr int UNROLL = [] {
            int u = RING;
            while (u % U2 != 0 || u % 2 != 0) u += RING;
            return u;
        }();
        static_assert(UNROLL % RING == 0 && UNROLL % U2 == 0 && UNROLL % 2 == 0, "unroll period");
        const int NBm = NB / UNROLL * UNROLL;           // blocks of the pipelined loop
        int i0 = 0;
        if (NBm > 0) {
            // prologue: the whole frame ring and BR-1 mask slots, all landed; the first fragments in
            // registers; then what "block -1" would have issued
#pragma unroll
            for (int d = 0; d < BR - 1; ++d) issue_b(d);
#pragma unroll
            for (int d = 0; d < RING; ++d) issue_a(d, d);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            u32x4s P[2][TILES][3], Bq[2][NG][3];
            f32x4s RA[2][TILES][2];
            rd_a(0, RA[0]);
            rd_b(0, 0, Bq[0]);
            rd_a(1 % RING, RA[1]);
            split(RA[0], P[0]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue_b(BR - 1);
            issue_a(RING, 0);
            __builtin_amdgcn_sched_barrier(0);

            auto iteration = [&](int i, auto ph) {
                constexpr int PH = decltype(ph)::value;
                constexpr int cur = PH & 1, nxt = cur ^ 1;
                constexpr bool boundary = (PH + 1) % PERB == 0;     // block i+1 opens a mask slot
                if constexpr (boundary) {
                    // that slot was issued BR-1 slots ago; since then: (BR-1) PERB frame sub-chunks
                    // and BR-2 slots.  Then every wave has left the slot before it: barrier.
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND * (BR - 1) * PERB + NBI * (BR - 2))
                                 : "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    issue_b((i + 1) / PERB + BR - 1);
                }
                {
                    // sub-chunk i+2 was issued in block i+1-RING (after that block's slot DMA, if
                    // any): the RING-2 sub-chunks after it and the slots issued in blocks
                    // i+2-RING .. i may stay in flight
                    constexpr int nb = [] {
                        int n = 0;
                        for (int j = PH + 2 - RING; j <= PH; ++j)
                            n += (((j + 1) % PERB + PERB) % PERB == 0) ? 1 : 0;
                        return n;
                    }();
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND * (RING - 2) + nb * NBI) : "memory");
                }
                rd_a((PH + 2) % RING, RA[cur]);                      // raw(i+2); raw(i) was split long ago
                rd_b(((PH + 1) / PERB) % BR, (PH + 1) % PERB, Bq[nxt]);
                // raw(i+1) left its ring slot a block ago (it is in RA[nxt]): the slot takes sub-chunk
                // i + 1 + RING
                issue_a(i + 1 + RING, (PH + 1) % RING);
                split(RA[nxt], P[nxt]);
                multiply(P[cur], Bq[cur]);
                // the block as a pipeline: one MFMA (16 cycles of the matrix pipe), then what fits beside it
                constexpr int NM = 6 * NG * TILES, NR = 2 * TILES + 3 * NG;
                sp_static_for<0, NM>([&](auto K) {
                    constexpr int k = decltype(K)::value;
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if constexpr (k < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, SP_VPM, 0);
                    if constexpr (k % (NM / ND) == 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                });
                __builtin_amdgcn_sched_barrier(0);
            };
            constexpr int FLUSH_BLOCKS = 1024 / (UNROLL * 32) > 1 ? 1024 / (UNROLL * 32) : 1;
            int blocks_done = 0;
            for (; i0 < NBm; i0 += UNROLL) {
                if (blocks_done == FLUSH_BLOCKS) { blocks_done = 0; flush(); }
                ++blocks_done;
                sp_static_for<0, UNROLL>([&](auto I) { iteration(i0 + decltype(I)::value, I); });
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // drain the prefetches
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        // the remaining blocks, one at a time
        if (i0 < NB) flush();
        for (int i = i0; i < NB; ++i) {
            if (i == i0 || i % PERB == 0) {
                __builtin_amdgcn_s_barrier();                         // everyone has left the buffer
                issue_b(i / PERB);
            }
            issue_a(i, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            f32x4s ra[TILES][2];
            u32x4s p[TILES][3], b[NG][3];
            rd_a(0, ra);
            rd_b((i / PERB) % BR, i % PERB, b);
            split(ra, p);
            multiply(p, b);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }

#pragma unroll
    for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t f = frame_of(tl * 16 + kg * 4 + r);
                const int col = (gt * NG + g) * SP_GROUP + m;
                if (f >= 0 && col < n_cols) {
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

__global__ void k_split_reduce(const float *__restrict__ partials, int ksplit, int64_t n_frames,
                               int n_cols, float *__restrict__ out, int64_t ld_out, int accumulate) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_frames * n_cols) return;
    const int64_t f = idx / n_cols;
    const int col = (int)(idx % n_cols);
    float v = 0.f;
    for (int k = 0; k < ksplit; ++k) v += partials[(int64_t)k * n_frames * n_cols + idx];
    float *p = out + f * ld_out + col;
    *p = accumulate ? (*p + v) : v;
}

// ---- host side -------------------------------------------------------------------------------------
static int split_ng(int n_groups) { return n_groups >= 3 ? 4 : 2; }
static int split_kb(int ng) { return 128 / ng; }

// opt-in: LTMI_SPLIT=1 in the environment, or tuning code 36 on the handle
bool split_selected(bool tuned) {
    static const bool on = [] {
        const char *e = getenv("LTMI_SPLIT");
        return e && e[0] == '1';
    }();
    return on || tuned;
}

bool split_wanted(int n_cols, int64_t n_px) {
    // two or more 16-column groups: the f32 matrix pipe is the limit there (a single group streams
    // at 0.88 of HBM on k_dense_lds already); whole mask slots only
    if (n_cols <= SP_GROUP || n_cols > 4 * SP_GROUP) return false;
    const int n_groups = (n_cols + SP_GROUP - 1) / SP_GROUP;
    const int kb = split_kb(split_ng(n_groups));
    return n_px >= 8 * kb && n_px % kb == 0;
}

int split_create(int device, const float *gmasks, int64_t n_masks, int cpm, int64_t n_px, int n_cols,
                 void **image) {
    *image = nullptr;
    SplitImage *im = new SplitImage();
    const int n_groups = (n_cols + SP_GROUP - 1) / SP_GROUP;
    im->ng = split_ng(n_groups);
    im->kb = split_kb(im->ng);
    im->n_slots = (int)(n_px / im->kb);
    im->n_gt = (n_groups + im->ng - 1) / im->ng;
    const size_t bytes = (size_t)im->n_gt * im->n_slots * SP_BSLOT;
    hipError_t e = hipMalloc((void **)&im->img, bytes);
    if (e == hipSuccess) e = hipMemset(im->img, 0, bytes);
    if (e == hipSuccess) {
        const int64_t total = n_masks * cpm * n_px;
        const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 65535 * 16);
        hipLaunchKernelGGL(k_build_split, dim3(blocks), dim3(256), 0, 0, gmasks, im->img, n_masks, cpm,
                           n_px, im->n_slots, im->ng, im->kb);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    if (e != hipSuccess) {
        if (im->img) (void)hipFree(im->img);
        delete im;
        LTMI_FAIL((int)e, "building the split mask image failed: %s", hipGetErrorString(e));
    }
    *image = im;
    return LTMI_OK;
}

void split_destroy(void *image) {
    SplitImage *im = (SplitImage *)image;
    if (!im) return;
    if (im->img) (void)hipFree(im->img);
    if (im->partials) (void)hipFree(im->partials);
    delete im;
}

size_t split_image_bytes(const void *image) {
    const SplitImage *im = (const SplitImage *)image;
    return im ? (size_t)im->n_gt * im->n_slots * SP_BSLOT : 0;
}

template <int NG, int TILES>
static int launch_split(ltmi_masks *m, SplitImage *im, const float *tile, int64_t n_frames, int64_t ld,
                        float *out, int64_t ld_out, int accumulate, hipStream_t stream) {
    using CFG = SplitCfg<NG, TILES>;
    auto kern = k_dense_split<NG, TILES>;
    static bool attr_set[16] = {false};
    if (!attr_set[m->device & 15]) {
        LTMI_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     CFG::LDS_BYTES));
        attr_set[m->device & 15] = true;
    }
    const int64_t gx = (n_frames + CFG::WG_ROWS - 1) / CFG::WG_ROWS;
    const int64_t gz = im->n_gt;
    int ksplit = m->tune_ksplit;
    if (ksplit <= 0) ksplit = choose_ksplit(gx * gz, im->n_slots);
    ksplit = std::max(1, std::min(ksplit, im->n_slots));
    {
        const int per = (im->n_slots + ksplit - 1) / ksplit;
        ksplit = (im->n_slots + per - 1) / per;
    }
    if (ksplit > 1) {
        const size_t need = (size_t)ksplit * n_frames * m->n_cols * sizeof(float);
        if (im->partials_bytes < need) {
            LTMI_HIP(hipStreamSynchronize(stream));
            if (im->partials) LTMI_HIP(hipFree(im->partials));
            im->partials = nullptr;
            im->partials_bytes = 0;
            LTMI_HIP(hipMalloc((void **)&im->partials, need));
            im->partials_bytes = need;
        }
    }
    dim3 grid((unsigned)gx, (unsigned)ksplit, (unsigned)gz);
    hipLaunchKernelGGL(kern, grid, dim3(CFG::WAVES * 64), CFG::LDS_BYTES, stream, tile, ld, n_frames,
                       m->n_px, (const uint16_t *)im->img, im->n_slots, out, ld_out, m->n_cols,
                       accumulate, im->partials, ksplit);
    LTMI_HIP(hipGetLastError());
    snprintf(m->last_kernel, sizeof(m->last_kernel),
             "k_dense_split<f,NG=%d,bf16x3,tiles=%d> grid=(%u,%u,%u)", NG, TILES, grid.x, grid.y,
             grid.z);
    if (ksplit > 1) {
        const int64_t n = n_frames * m->n_cols;
        hipLaunchKernelGGL(k_split_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                           (const float *)im->partials, ksplit, n_frames, m->n_cols, out, ld_out,
                           accumulate);
        LTMI_HIP(hipGetLastError());
    }
    return LTMI_OK;
}

int split_apply(ltmi_masks *m, void *image, const float *tile, int64_t n_frames, int64_t ld, float *out,
                int64_t ld_out, int accumulate, hipStream_t stream) {
    SplitImage *im = (SplitImage *)image;
    // 256-frame workgroups once there are enough frames to fill the chip with them (LTMI_SPLIT_TILES
    // = 2 / 4 in the environment forces one shape: benches)
    static const int forced = [] {
        const char *e = getenv("LTMI_SPLIT_TILES");
        return e ? atoi(e) : 0;
    }();
    const bool big = forced ? forced == 4 : n_frames >= 2048;
#define LTMI_SPLIT_GO(NG_)                                                                         \
    return big ? launch_split<NG_, 4>(m, im, tile, n_frames, ld, out, ld_out, accumulate, stream)  \
               : launch_split<NG_, 2>(m, im, tile, n_frames, ld, out, ld_out, accumulate, stream);
    if (im->ng == 4) { LTMI_SPLIT_GO(4) }
    LTMI_SPLIT_GO(2)
#undef LTMI_SPLIT_GO
}

}  // namespace ltmi
