// ltmi_cryst.hip -- CrystallinityUDF.process_frame for 256 x 256 (k_cryst_fused) and 128 x 128 frames
// (k_cryst_fused128, second part of this file) in ONE kernel, for 512 x 512 and 1024 x 1024 frames in two (k_cryst_rows +
// k_cryst_cols, third part) (reference
// udf/crystallinity.py:73-79):
//
//     intensity[f] = sum( abs(rfft2(frame * real_mask)) * half_fourier_mask )
//
// The hipFFT route (ltmi_fft.hip) writes every frame as float32, transforms it into a 129 x 256
// half spectrum in HBM and reads that back: 128 KiB of pixels become 0.9 MiB of traffic and three
// launches.  Here a workgroup keeps ONE frame in the LDS from the pixels to the single float:
//
//   rows     a wave takes two rows a, b as the complex sequence a + i b through a 256-point radix-4
//            transform, separates the two spectra (Z[k], conj Z[256 - k]: one ds_bpermute pair) and
//            stores the columns kx < K it will need -- K = the ring's outer radius + 1 -- into G[kx][y];
//   columns  a wave transforms column kx of G in place and sums |F[ky][kx]| * mask[ky][kx] into a
//            register; the workgroup writes one float per frame.
//
// The transform (cf_core; lane-level model with every index below: scripts/cryst_fft_model.py).  A wave
// holds the 256 points as 4 registers x 64 lanes; a radix-4 pass works on the index digit that is the
// REGISTER index, so between passes a digit held in two lane bits changes place with the register digit:
//     n = 64 a2 + 16 a1 + 4 a0 + j        lane = (a2 a1 a0), register = j       (8 bytes of a row per lane)
//     swap register <-> lane[5:4]        v_permlane32_swap + v_permlane16_swap: no LDS at all
//     pass over a2 -> c0, twiddle W64^((4 a1 + a0) c0)
//     swap register <-> lane[3:2]        through the LDS: store 64 r + (lane ^ 4 r), load base ^ 4 r
//     pass over a1 -> c1, twiddle W16^(a0 c1)
//     swap register <-> lane[1:0]        through the LDS: store 64 r + (lane ^ r), load base ^ r
//     pass over a0 -> c2
//     swap register <-> lane[5:4]        permlane swaps again
//     twiddle W256^(j k1), pass over j -> k2:   register k2 of lane (c2 c0 c1) holds Z[k1 + 64 k2],
//                                               k1 = c0 + 4 c1 + 16 c2 = sigma(lane)
// Two LDS round trips per transform (a Stockham formulation needs one per pass and one more to bring the
// row in: that version of this kernel ran 16 384 frames in 1.58 ms, LDS and vector ALUs ~55 % busy each).
// The XORs make every 16-lane store group and every 32-lane load group a permutation of the banks (SQ_LDS_BANK_CONFLICT
// = 0 for the transposes: k_cryst_cols; the 8 cycles per row pair the row kernels count belong to the ds_bpermute pairs).  The column stage reads G[kx] directly in the layout after the first swap.
//
// G[kx][y]: 258 float2 per column (516 dwords = 4 mod 32: 8 neighbouring columns, 16 bytes each, hit 32
// banks), y kept at y ^ (2 ((y >> 5) & 1)) ^ (4 ((kx >> 3) & 1)): the first makes the column stage's loads
// (lane 16 j + m reads y = 64 r + 4 m + j) conflict-free, the second the row stage's stores (a group of
// 8 lanes holds the columns sigma(lane) = {0, 4, 8, 12, 1, 5, 9, 13} + 16 i).
//
// LDS: K columns + 2 KiB of row scratch per wave that transforms rows: K = 65 (rad_out 64) leaves room for
// 14 of the 16 waves; rings with K > CF_KMAX columns run the workspace kernels of the third part (M = 1), other frame
// shapes than the four stay on the hipFFT route; float64
// pixels, odd strides and detector corrections reach this kernel as float32 frames written by the conversion
// pass (ltmi_fft.hip).  HBM traffic: the pixels once (+ the two masks from the L2).
#include "ltmi_common.h"
#include <algorithm>
#include <type_traits>

namespace ltmi {

constexpr int CF_N = 256;                       // frame edge
constexpr int CF_COL = CF_N + 2;                // float2 units per column of G
constexpr int CF_SCR = CF_N;                    // float2 units of row scratch per wave
constexpr int CF_LDS_MAX = 160 * 1024 - 256;    // dynamic LDS a workgroup may ask for (static: the partial sums)
constexpr int CF_KMAX = (CF_LDS_MAX - 8 * CF_SCR * 8) / (CF_COL * 8);   // at least 8 row buffers

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct CfLane {                                 // per-lane constants of the transform
    v2f tw[3][3];                               // passes over a2, a1, j: (re, im) of the twiddle of register 1..3
    v2f twr[3][3];                              // (-im, re) of the same: u w = u.xx * tw + u.yy * twr, two packed FMAs
    int wB[4], rB[4];                           // LDS addresses (float2 units) of the swap with lane[3:2]
    int wC[4], rC[4];                           // ... with lane[1:0]
};

__device__ __forceinline__ int cf_sigma(int l) { return (l & 48) | ((l & 3) << 2) | ((l >> 2) & 3); }

__device__ __forceinline__ v2f cf_mul(v2f u, v2f w, v2f wr) {
    return __builtin_elementwise_fma(u.xx, w, u.yy * wr);
}

// radix-4 butterfly, forward: 8 packed adds.  t1 -+ i d has a different sign in each half: the VOP3P
// modifiers express it (op_sel swaps the halves of d, neg_lo / neg_hi negate one of them), the compiler does
// not (it inserts v_xor + v_mov), hence two instructions of inline assembly.  PLAIN = the compiler's version:
// for results that feed a v_permlane*_swap (the hazard recognizer must see their producer).
template <bool PLAIN = false>
__device__ __forceinline__ void cf_bfly(v2f (&u)[4]) {
    const v2f t0 = u[0] + u[2], t1 = u[0] - u[2], t2 = u[1] + u[3], d = u[1] - u[3];
    u[0] = t0 + t2;
    u[2] = t0 - t2;
    if (PLAIN) {
        u[1] = (v2f){t1.x + d.y, t1.y - d.x};
        u[3] = (v2f){t1.x - d.y, t1.y + d.x};
    } else {
        v2f a, b;
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(a) : "v"(t1), "v"(d));
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(b) : "v"(t1), "v"(d));
        u[1] = a;
        u[3] = b;
    }
}

// the wave's own stores become visible to its own loads in program order (DS operations of a wave
// are served in order): only the compiler has to be kept from moving them across each other
__device__ __forceinline__ void cf_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// (__builtin_bit_cast of an ext-vector ELEMENT reads element 0 whatever the element -- clang of ROCm 7.2 --
// so the components go through float temporaries)
// a' = (a.lo32, b.lo32), b' = (a.hi32, b.hi32): register bit <-> lane bit 5
__device__ __forceinline__ void cf_swap32(v2f &a, v2f &b) {
    const float ax = a.x, ay = a.y, bx = b.x, by = b.y;
    const auto x = __builtin_amdgcn_permlane32_swap(__float_as_uint(ax), __float_as_uint(bx), false, false);
    const auto y = __builtin_amdgcn_permlane32_swap(__float_as_uint(ay), __float_as_uint(by), false, false);
    const unsigned x0 = x[0], x1 = x[1], y0 = y[0], y1 = y[1];
    a = (v2f){__uint_as_float(x0), __uint_as_float(y0)};
    b = (v2f){__uint_as_float(x1), __uint_as_float(y1)};
}

// odd rows of 16 lanes of a <-> even rows of b: register bit <-> lane bit 4
__device__ __forceinline__ void cf_swap16(v2f &a, v2f &b) {
    const float ax = a.x, ay = a.y, bx = b.x, by = b.y;
    const auto x = __builtin_amdgcn_permlane16_swap(__float_as_uint(ax), __float_as_uint(bx), false, false);
    const auto y = __builtin_amdgcn_permlane16_swap(__float_as_uint(ay), __float_as_uint(by), false, false);
    const unsigned x0 = x[0], x1 = x[1], y0 = y[0], y1 = y[1];
    a = (v2f){__uint_as_float(x0), __uint_as_float(y0)};
    b = (v2f){__uint_as_float(x1), __uint_as_float(y1)};
}

// register index <-> lane[5:4]
__device__ __forceinline__ void cf_swap_a(v2f (&u)[4]) {
    cf_swap32(u[0], u[2]);
    cf_swap32(u[1], u[3]);
    cf_swap16(u[0], u[1]);
    cf_swap16(u[2], u[3]);
}

// From the layout after the first swap (lane = 16 j + 4 a1 + a0, register = a2) to the spectrum:
// u[k2] = Z[sigma(lane) + 64 k2].  buf: 256 float2 of LDS owned by this wave (destroyed).
template <bool NOLDS = false>
__device__ __forceinline__ void cf_core(v2f *buf, const CfLane &c, v2f (&u)[4]) {
    cf_bfly(u);
#pragma unroll
    for (int r = 1; r < 4; ++r) u[r] = cf_mul(u[r], c.tw[0][r - 1], c.twr[0][r - 1]);
    if (!NOLDS) {
        cf_wave_sync();
#pragma unroll
        for (int r = 0; r < 4; ++r) buf[c.wB[r]] = u[r];
        cf_wave_sync();
#pragma unroll
        for (int r = 0; r < 4; ++r) u[r] = buf[c.rB[r]];
    }
    cf_bfly(u);
#pragma unroll
    for (int r = 1; r < 4; ++r) u[r] = cf_mul(u[r], c.tw[1][r - 1], c.twr[1][r - 1]);
    if (!NOLDS) {
        cf_wave_sync();
#pragma unroll
        for (int r = 0; r < 4; ++r) buf[c.wC[r]] = u[r];
        cf_wave_sync();
#pragma unroll
        for (int r = 0; r < 4; ++r) u[r] = buf[c.rC[r]];
        cf_wave_sync();
    }
    cf_bfly<true>(u);
    cf_swap_a(u);
#pragma unroll
    for (int r = 1; r < 4; ++r) u[r] = cf_mul(u[r], c.tw[2][r - 1], c.twr[2][r - 1]);
    cf_bfly(u);
}

// Detector corrections inside the row stages (round 5): what the kernels get when a call carries them.
struct CfCorr {
    const float *d_px;                  // dark, pixel order (float32)
    float *dmap_p;                      // ... in the lane order of the kernel's real-space mask (packed per call)
    unsigned long long *dummy_flags;    // (the pack kernels write flags)
    const int *pair_ptr, *pcode;        // excluded pixels by item (row pair / group of four rows) of the row stage
    const float *patch;                 // [frame][excluded pixel]: the repaired values
    int n_excl;
};
// pcode: row of the pair (bit 0) | register (2 bits) << 1 | lane << 3 | sub-transform q (2 bits) << 9 | index << 16
template <int NQ>
__device__ __forceinline__ void cf_apply_patches(v2f (&u)[NQ][4], int t, int item, int64_t fr, const int *pair_ptr,
                                                 const int *pcode, const float *patch, int n_excl) {
    const int e1 = pair_ptr[item + 1];
    for (int e = pair_ptr[item]; e < e1; ++e) {                 // (uniform: usually none)
        const int code = pcode[e];
        const float v = patch[fr * n_excl + (code >> 16)];
        if (t == ((code >> 3) & 63)) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (q == ((code >> 9) & 3) && j == ((code >> 1) & 3)) {
                        if (code & 1) u[q][j].y = v;
                        else u[q][j].x = v;
                    }
        }
    }
}

template <typename T>
struct CfRaw {                                  // the pixels one lane holds of a row pair: 4 of row a, 4 of row b
    typedef T __attribute__((ext_vector_type(4))) vec_t;
    vec_t ra, rb;
};
struct CfMask { v4f m01, m23; };                // (ma0, mb0, ma1, mb1), (ma2, mb2, ma3, mb3)

template <typename T, int ABL = 0>
__device__ __forceinline__ void cf_load_raw(CfRaw<T> &p, const T *__restrict__ src, int yp, int t) {
    typedef typename CfRaw<T>::vec_t vec_t;
    if (ABL == 1) {                                 // (timing only: no memory)
        p.ra = (vec_t)(T)yp;
        p.rb = (vec_t)(T)t;
        return;
    }
    const T *row = src + (2 * yp) * CF_N;           // (uniform: scalar base + the lane's offset)
    p.ra = __builtin_nontemporal_load((const vec_t *)(row + 4 * t));
    p.rb = __builtin_nontemporal_load((const vec_t *)(row + CF_N + 4 * t));
}

template <int ABL = 0>
__device__ __forceinline__ void cf_load_mask(CfMask &p, const float *__restrict__ rmask_p, int yp, int t) {
    if (ABL == 1) {
        p.m01 = p.m23 = (v4f)1.f;
        return;
    }
    const float *row = rmask_p + yp * (2 * CF_N);
    p.m01 = *(const v4f *)(row + 8 * t);
    p.m23 = *(const v4f *)(row + 8 * t + 4);
}

// WAVES waves per workgroup (one workgroup per CU); n_scr <= WAVES of them own 2 KiB of row scratch and
// transform row pairs (what the LDS leaves beside the K columns of G), all of them transform columns.
// ABL: timing-only ablations (LTMI_CRYST_ABLATE, uint16 + mask + 16 waves): 1 no global loads, 2 no barriers,
// 3 no LDS transposes, 4 rows only, 5 columns only -- the results are garbage
// CORR (detector corrections inside the row stage, round 5; io/corrections/detector.py:17-101 fused with
// udf/crystallinity.py:73-79): a pixel becomes (x - dark) * (gain * real mask) with both maps in the pairs' lane order
// (dmap_p like rmask_p; rmask_p then holds gain * mask, 0 at the excluded pixels), and the excluded pixels of a row pair
// (pair_ptr / pcode: lane, register, row of the pair, index into the frame's patch values) take their repaired value
// -- the mean of the corrected good neighbours times the mask, computed per frame by k_cryst_patch_values.
template <typename T, bool MASK, int WAVES, int ABL = 0, bool CORR = false>
__global__ void __launch_bounds__(WAVES * 64)
k_cryst_fused(const T *__restrict__ tile, int64_t ld, int64_t n_frames,
              const float *__restrict__ rmask_p, const unsigned long long *__restrict__ rflags,
              const float *__restrict__ mask_p, int K, int n_scr, float *__restrict__ out, int accumulate,
              const float *__restrict__ dmap_p = nullptr, const int *__restrict__ pair_ptr = nullptr,
              const int *__restrict__ pcode = nullptr, const float *__restrict__ patch = nullptr, int n_excl = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cf_smem[];
    __shared__ float part[WAVES];
    const int t = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // (uniform: scalar addresses and branches)
    v2f *G = (v2f *)cf_smem;
    v2f *scr = G + K * CF_COL + w * CF_SCR;
    const int sig = cf_sigma(t);

    CfLane c;
#pragma unroll
    for (int r = 1; r < 4; ++r) {
        double s, co;
        sincospi(-2.0 * (double)((t & 15) * r) / 64.0, &s, &co);
        c.tw[0][r - 1] = (v2f){(float)co, (float)s};
        sincospi(-2.0 * (double)((t & 3) * r) / 16.0, &s, &co);
        c.tw[1][r - 1] = (v2f){(float)co, (float)s};
        sincospi(-2.0 * (double)(sig * r) / 256.0, &s, &co);
        c.tw[2][r - 1] = (v2f){(float)co, (float)s};
#pragma unroll
        for (int pi = 0; pi < 3; ++pi) c.twr[pi][r - 1] = (v2f){-c.tw[pi][r - 1].y, c.tw[pi][r - 1].x};
    }
    {
        const int b2 = (t >> 2) & 3, b0 = t & 3;
        const int base_b = 64 * b2 + ((t & ~12) | (b2 << 2)), base_c = 64 * b0 + ((t & ~3) | b0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            c.wB[r] = 64 * r + (t ^ (r << 2));
            c.rB[r] = base_b ^ (r << 2);
            c.wC[r] = 64 * r + (t ^ r);
            c.rC[r] = base_c ^ r;
        }
    }
    // row stage: the lane's columns kx = sigma(t) (+ 64), the partner lane that holds Z[256 - kx]
    const int back = cf_sigma((64 - sig) & 63) * 4;                 // ds_bpermute address
    const int g_col = sig * CF_COL, g_xor = 4 * ((sig >> 3) & 1);
    // column stage: lane 16 j + m loads y = 64 r + 4 m + j (kept at y ^ 2 (m >> 3), ^ 4 in odd groups of 8 columns)
    const int col_rd = (4 * (t & 15) + (t >> 4)) ^ (2 * ((t >> 3) & 1));
    const bool rows = w < n_scr;

    // Row pairs as one stream of items (frame, y') per wave, two items per turn of the loop: their pixels
    // were loaded during the previous turn (two transforms of lead: with one pair of lead the kernel ran 21 %
    // slower than without memory), are converted, then the loads of the next two items are issued.  The mask
    // values are fetched from the L2 where they are used, and only for pairs whose mask is not all ones (bit
    // y' of rflags: a disk touches ~20 % of the pairs).
    const unsigned long long fl0 = MASK ? rflags[0] : 0, fl1 = MASK ? rflags[1] : 0;
    auto masked = [&](int yp) -> bool { return MASK && (((yp < 64 ? fl0 : fl1) >> (yp & 63)) & 1); };
    int64_t lf = blockIdx.x;                            // load cursor
    int lyp = w;
    auto load_next = [&](CfRaw<T> &dst) {
        if (lf < n_frames) cf_load_raw<T, ABL>(dst, tile + lf * ld, lyp, t);
        lyp += n_scr;
        if (lyp >= CF_N / 2) {
            lyp = w;
            lf += gridDim.x;
        }
    };
    auto convert = [&](const CfRaw<T> &buf, int yp, int64_t fr, v2f (&u)[4]) {
        CfMask bm;
        const bool mk = CORR || masked(yp);
        if (mk) cf_load_mask<ABL>(bm, rmask_p, yp, t);
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = (v2f){(float)buf.ra[j], (float)buf.rb[j]};
        if constexpr (CORR) {
            CfMask dm;
            cf_load_mask<ABL>(dm, dmap_p, yp, t);
            u[0] -= dm.m01.xy; u[1] -= dm.m01.zw;
            u[2] -= dm.m23.xy; u[3] -= dm.m23.zw;
        }
        if (mk) {
            u[0] *= bm.m01.xy; u[1] *= bm.m01.zw;
            u[2] *= bm.m23.xy; u[3] *= bm.m23.zw;
        }
        if constexpr (CORR) {
            if (n_excl > 0 && fr < n_frames) {
                v2f (&uu)[1][4] = reinterpret_cast<v2f (&)[1][4]>(u);
                cf_apply_patches<1>(uu, t, yp, fr, pair_ptr, pcode, patch, n_excl);
            }
        }
    };

    // one row pair: transform, separate, store the K columns
    auto row_pair = [&](v2f (&u)[4], int yp) {
        if (ABL == 5) return;
        cf_swap_a(u);
        cf_core<ABL == 3>(scr, c, u);
        // two real rows out of one complex transform: with Z[k] = (a, b), Z[256 - k] = (c, d)
        //   2 A[k] = (a + c, b - d)     2 B[k] = (b + d, c - a)     (the 1/2 is applied at the end)
        const int pos = (2 * (yp ^ ((yp >> 4) & 1))) ^ g_xor;
        {
            const float gx = t == 0 ? u[0].x : u[3].x, gy = t == 0 ? u[0].y : u[3].y;
            const float cr = __int_as_float(__builtin_amdgcn_ds_bpermute(back, __float_as_int(gx)));
            const float ci = __int_as_float(__builtin_amdgcn_ds_bpermute(back, __float_as_int(gy)));
            if (sig < K)
                *(v4f *)(G + g_col + pos) = (v4f){u[0].x + cr, u[0].y - ci, u[0].y + ci, cr - u[0].x};
        }
        if (K == 65) {
            // the one column kx = 64: Z[64] and Z[192] are registers 1 and 3 of lane 0, no exchange
            if (t == 0)
                *(v4f *)(G + 64 * CF_COL + pos) =
                    (v4f){u[1].x + u[3].x, u[1].y - u[3].y, u[1].y + u[3].y, u[3].x - u[1].x};
        } else if (K > 64) {
            const float gx = t == 0 ? u[3].x : u[2].x, gy = t == 0 ? u[3].y : u[2].y;
            const float cr = __int_as_float(__builtin_amdgcn_ds_bpermute(back, __float_as_int(gx)));
            const float ci = __int_as_float(__builtin_amdgcn_ds_bpermute(back, __float_as_int(gy)));
            if (sig + 64 < K)
                *(v4f *)(G + g_col + 64 * CF_COL + pos) =
                    (v4f){u[1].x + cr, u[1].y - ci, u[1].y + ci, cr - u[1].x};
        }
    };

    // after the last row pair of frame f: columns kx = w + WAVES i transformed, |F| * mask summed per lane, one
    // float per frame.  A wave that owns row scratch runs the exchanges of its column transforms THERE (fixed
    // addresses: 16 address additions per column less), the others in place in the column.
    auto frame_end = [&](int64_t f, auto in_place_tag) {
        constexpr bool IN_PLACE = decltype(in_place_tag)::value;
        if (ABL != 2) __syncthreads();
        float acc = 0.f;
        for (int kx = w; kx < (ABL == 4 ? 0 : K); kx += WAVES) {
            float m[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) m[r] = mask_p[kx * CF_N + 64 * r + t];
            v2f *col = G + kx * CF_COL;
            const int rd = col_rd ^ (4 * ((kx >> 3) & 1));
            v2f u[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) u[r] = col[rd + 64 * r];
            cf_core<ABL == 3>(IN_PLACE ? col : scr, c, u);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (__builtin_amdgcn_ballot_w64(m[r] != 0.f)) {          // (most columns: two of the four)
                    const float a = __builtin_amdgcn_sqrtf(u[r].x * u[r].x + u[r].y * u[r].y);
                    acc += m[r] != 0.f ? a * m[r] : 0.f;
                }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (t == 0) part[w] = acc;
        if (ABL != 2) __syncthreads();
        if (threadIdx.x == 0) {
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < WAVES; ++i) v += part[i];
            v *= 0.5f;
            out[f] = accumulate ? out[f] + v : v;
        }
    };

    if (!rows) {                                        // (no row scratch left for this wave: columns only)
        for (int64_t f = blockIdx.x; f < n_frames; f += gridDim.x) frame_end(f, std::true_type());
        return;
    }
    CfRaw<T> ba, bb;
    load_next(ba);
    load_next(bb);
    int64_t f = blockIdx.x;
    int yp = w;
    while (f < n_frames) {
        v2f ua[4], ub[4];
        const int ypa = yp;
        int ypb = yp + n_scr;                           // (item b: the next pair of this frame or the first of the next)
        const bool a_last = ypb >= CF_N / 2;
        if (a_last) ypb = w;
        const bool have_b = !a_last || f + gridDim.x < n_frames;
        convert(ba, ypa, f, ua);
        if (have_b) convert(bb, ypb, a_last ? f + gridDim.x : f, ub);
        load_next(ba);
        load_next(bb);
        row_pair(ua, ypa);
        if (a_last) {
            frame_end(f, std::false_type());
            f += gridDim.x;
            if (!have_b) break;
        }
        row_pair(ub, ypb);
        yp = ypb + n_scr;
        if (yp >= CF_N / 2) {
            frame_end(f, std::false_type());
            f += gridDim.x;
            yp = w;
        }
    }
}

// The two masks in the order the lanes want them (one launch per call, 0.3 MiB):
//   mask_p[kx][64 r + l]  = half_mask[sigma(l) + 64 r][kx]         (register r of lane l in the column stage)
//   rmask_p[y'][2 x + i]  = real_mask[2 y' + i][x]                 (rows of a pair interleaved like a + i b)
//   rflags bit y'         = the pair y' holds a mask value other than 1 (cleared by the caller)
__global__ void __launch_bounds__(256)
k_cryst_masks(const float *__restrict__ half_mask, int wc, int K, float *__restrict__ mask_p,
              const float *__restrict__ real_mask, float *__restrict__ rmask_p,
              unsigned long long *__restrict__ rflags) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < K * CF_N) {
        const int kx = i / CF_N, q = i - kx * CF_N;
        const int ky = cf_sigma(q & 63) + 64 * (q >> 6);
        mask_p[i] = half_mask[(int64_t)ky * wc + kx];
    }
    if (real_mask && i < CF_N * CF_N) {
        const int yp = i / (2 * CF_N), q = i - yp * (2 * CF_N);
        const float v = real_mask[(2 * yp + (q & 1)) * CF_N + (q >> 1)];
        rmask_p[i] = v;
        // (a wave's 64 elements belong to one pair)
        if (__builtin_amdgcn_ballot_w64(v != 1.f) && (threadIdx.x & 63) == 0)
            atomicOr(&rflags[yp >> 6], 1ull << (yp & 63));
    }
}

// ---- corrections inside the row stages (round 5) -----------------------------------------------------------------
constexpr int CF_MAX_EXCL = 4096;
// gm_px = gain * real mask, d_px = dark, pixel order (float32; the reference corrects in float64 and rounds once:
// (x - dark) * gain, then multiplies by the mask in float32 -- io/corrections/detector.py:17-58, udf/crystallinity.py:73-79)
__global__ void __launch_bounds__(256)
k_cryst_corr_maps(const double *__restrict__ dark, const double *__restrict__ gain,
                  const float *__restrict__ real_mask, int64_t n_px, float *__restrict__ gm_px,
                  float *__restrict__ d_px) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_px) return;
    gm_px[p] = (float)(gain ? gain[p] : 1.0) * (real_mask ? real_mask[p] : 1.f);
    d_px[p] = (float)(dark ? dark[p] : 0.0);
}
// one workgroup: the excluded pixels sorted by item of the row stage (pair_ptr[n_items + 1], pcode: see CfCorr) and
// their entries of gm_px cleared.  mode 0: row pairs, a lane holds 4 M consecutive pixels of both rows (k_cryst_fused:
// M = 1, k_cryst_rows<M>); mode 1: groups of four rows, a lane holds 2 pixels of each (k_cryst_fused128)
__global__ void __launch_bounds__(256)
k_cryst_patch_index(const int32_t *__restrict__ excl, int n_excl, int W, int n_items, int mode, int M,
                    int *__restrict__ pair_ptr, int *__restrict__ pcode, float *__restrict__ gm_px) {
    __shared__ int cnt[513], cur[512];
    for (int i = threadIdx.x; i <= n_items; i += 256) cnt[i] = 0;
    __syncthreads();
    const int rows_per_item = mode == 1 ? 4 : 2;
    for (int e = threadIdx.x; e < n_excl; e += 256) atomicAdd(&cnt[excl[e] / W / rows_per_item], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < n_items; ++i) {
            const int c = cnt[i];
            pair_ptr[i] = run;
            cur[i] = run;
            run += c;
        }
        pair_ptr[n_items] = run;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n_excl; e += 256) {
        const int p = excl[e], y = p / W, x = p - y * W;
        int lane, reg, sub = 0;
        if (mode == 1) {
            lane = x >> 1;
            reg = (x & 1) * 2 + ((y & 3) >> 1);
        } else {
            lane = x / (4 * M);
            const int r = x - lane * 4 * M;
            reg = r / M;
            sub = r - reg * M;
        }
        const int slot = atomicAdd(&cur[y / rows_per_item], 1);
        pcode[slot] = (y & 1) | (reg << 1) | (lane << 3) | (sub << 9) | (e << 16);
        gm_px[p] = 0.f;
    }
}
// patch[f, e] = mean over the good neighbours of excluded pixel e of the CORRECTED pixel (as the float32 value
// the corrected tile would hold), times the real-space mask at e (io/corrections/detector.py:60-101)
template <typename T>
__global__ void __launch_bounds__(256)
k_cryst_patch_values(const T *__restrict__ tile, int64_t ld, int64_t n_frames, const double *__restrict__ dark,
                     const double *__restrict__ gain, const float *__restrict__ real_mask,
                     const int32_t *__restrict__ excl, const int32_t *__restrict__ env,
                     const int32_t *__restrict__ cnt, int n_excl, int max_env, float *__restrict__ patch) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_frames * n_excl) return;
    const int64_t f = i / n_excl;
    const int e = (int)(i % n_excl);
    const int c = cnt[e];
    const T *src = tile + f * ld;
    float v;
    if (c > 0) {
        double acc = 0.0;
        for (int j = 0; j < c; ++j) {
            const int r = env[(int64_t)e * max_env + j];
            acc += (double)(float)(((double)src[r] - (dark ? dark[r] : 0.0)) * (gain ? gain[r] : 1.0));
        }
        v = (float)(acc / (double)c);
    } else {
        // no good neighbour: the pixel keeps its corrected value (nothing to repair with)
        const int p = excl[e];
        v = (float)(((double)src[p] - (dark ? dark[p] : 0.0)) * (gain ? gain[p] : 1.0));
    }
    if (real_mask) v *= real_mask[excl[e]];
    patch[i] = v;
}

bool cryst_fused_takes(int h, int w, int n_cols);
int64_t cryst_corr_workspace_bytes(int h, int w, int64_t n_frames, int n_excl) {
    return (int64_t)3 * h * w * 4 + 64 + 520 * 4 + (int64_t)std::max(n_excl, 1) * 4 +
           (int64_t)n_frames * std::max(n_excl, 1) * 4 + 64;
}
bool cryst_corr_takes(int h, int w, int n_cols, int tile_dtype, int n_excl) {
    return cryst_fused_takes(h, w, n_cols) && n_excl <= CF_MAX_EXCL && dtype_size(tile_dtype) <= 4 &&
           tile_dtype != LTMI_F64;
}

template <typename T>
static int launch_fused_corr(const void *tile, int64_t ld, int64_t n_frames, const float *gm_p,
                             const unsigned long long *rflags, const float *mask_t, int K, const CfCorr *corr,
                             float *out, int accumulate, int n_cu, hipStream_t stream) {
    constexpr int WAVES = 16;
    auto kern = k_cryst_fused<T, true, WAVES, 0, true>;
    const int n_scr = std::min(WAVES, (CF_LDS_MAX - K * CF_COL * 8) / (CF_SCR * 8));
    const int lds = K * CF_COL * 8 + n_scr * CF_SCR * 8;
    int device = 0;
    LTMI_HIP(hipGetDevice(&device));
    static bool attr_set[16] = {false};
    if (!attr_set[device & 15]) {
        LTMI_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CF_LDS_MAX));
        attr_set[device & 15] = true;
    }
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(n_frames, n_cu));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), (size_t)lds, stream, (const T *)tile, ld, n_frames, gm_p,
                       rflags, mask_t, K, n_scr, out, accumulate, (const float *)corr->dmap_p, corr->pair_ptr,
                       corr->pcode, corr->patch, corr->n_excl);
    LTMI_HIP(hipGetLastError());
    return LTMI_OK;
}

int cryst_fused_max_cols() { return CF_KMAX; }
// rings the fused kernels take (given a tile they can read: see cryst_fused)
static bool cf_edge(int e) { return e == 256 || e == 512 || e == 1024; }
bool cryst_fused_takes(int h, int w, int n_cols) {
    if (h == 128 && w == 128) return n_cols >= 1 && n_cols <= 65;
    return cf_edge(h) && cf_edge(w) && n_cols >= 1 && n_cols <= w / 2 + 1;
}
bool cryst_fused_shape(int h, int w) { return (h == 128 && w == 128) || (cf_edge(h) && cf_edge(w)); }
// frames other than 128 x 128 and 256 x 256 (and 256 x 256 frames with rings of more than 71 columns) pass the ring's
// columns of the row transforms through a (frames x n_cols x h) float2 workspace
bool cryst_fused_needs_gbuf(int h, int w, int n_cols) {
    return cf_edge(h) && cf_edge(w) && !(h == 256 && w == 256 && n_cols <= CF_KMAX);
}
// the masks in lane order (+ flags): what the kernels of an h x w plan need
int64_t cryst_fused_workspace_floats(int h, int w) {
    if (cf_edge(h) && cf_edge(w)) return (int64_t)(w / 2 + 1) * h + (int64_t)w * h + 16;   // (covers k_cryst_fused's layout too)
    return (int64_t)CF_KMAX * CF_N + CF_N * CF_N + 8;
}

template <typename T, int WAVES>
static int launch_fused_w(const void *tile, int64_t ld, int64_t n_frames, const float *real_mask,
                          const unsigned long long *rflags, const float *mask_t, int K, float *out,
                          int accumulate, int n_cu, hipStream_t stream) {
    auto kern = real_mask ? k_cryst_fused<T, true, WAVES> : k_cryst_fused<T, false, WAVES>;
    if constexpr (std::is_same<T, uint16_t>::value && WAVES == 16) {
        static const int abl = getenv("LTMI_CRYST_ABLATE") ? atoi(getenv("LTMI_CRYST_ABLATE")) : 0;
        if (real_mask && abl == 1) kern = k_cryst_fused<T, true, WAVES, 1>;
        if (real_mask && abl == 2) kern = k_cryst_fused<T, true, WAVES, 2>;
        if (real_mask && abl == 3) kern = k_cryst_fused<T, true, WAVES, 3>;
        if (real_mask && abl == 4) kern = k_cryst_fused<T, true, WAVES, 4>;
        if (real_mask && abl == 5) kern = k_cryst_fused<T, true, WAVES, 5>;
    }
    const int n_scr = std::min(WAVES, (CF_LDS_MAX - K * CF_COL * 8) / (CF_SCR * 8));
    const int lds = K * CF_COL * 8 + n_scr * CF_SCR * 8;
    int device = 0;
    LTMI_HIP(hipGetDevice(&device));
    static bool attr_set[16][2] = {{false}};          // per device (and per pixel type: one copy per T)
    if (!attr_set[device & 15][real_mask ? 1 : 0] || getenv("LTMI_CRYST_ABLATE")) {
        LTMI_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     CF_LDS_MAX));
        attr_set[device & 15][real_mask ? 1 : 0] = true;
    }
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(n_frames, n_cu));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), (size_t)lds, stream, (const T *)tile, ld,
                       n_frames, real_mask, rflags, mask_t, K, n_scr, out, accumulate, (const float *)nullptr,
                       (const int *)nullptr, (const int *)nullptr, (const float *)nullptr, 0);
    LTMI_HIP(hipGetLastError());
    return LTMI_OK;
}

static int cf_waves() {                               // LTMI_CRYST_WAVES=8 / 16 (measurement switch)
    static int v = 0;
    if (!v) {
        const char *env = getenv("LTMI_CRYST_WAVES");
        v = env && atoi(env) == 8 ? 8 : 16;
    }
    return v;
}

template <typename T>
static int launch_fused(const void *tile, int64_t ld, int64_t n_frames, const float *real_mask,
                        const unsigned long long *rflags, const float *mask_t, int K, float *out,
                        int accumulate, int n_cu, hipStream_t stream) {
    return cf_waves() == 8
        ? launch_fused_w<T, 8>(tile, ld, n_frames, real_mask, rflags, mask_t, K, out, accumulate, n_cu, stream)
        : launch_fused_w<T, 16>(tile, ld, n_frames, real_mask, rflags, mask_t, K, out, accumulate, n_cu, stream);
}

// ---- 128 x 128 frames -------------------------------------------------------------------------------------
// Two 128-point transforms as ONE 256-point transform of the interleaved sequence w[2m] = z1[m], w[2m+1] = z2[m]:
//     W[k] = Z1[k] + w256^k Z2[k],  W[k + 128] = Z1[k] - w256^k Z2[k]      (k < 128; registers k2 and k2 + 2 of a lane)
// so cf_core serves unchanged: a wave transforms FOUR rows 4q .. 4q+3 (z1 = row 4q + i row 4q+1, z2 = the next two;
// lane t holds the pixels 2t, 2t+1 of each), separates Z1 / Z2 with one butterfly and one twiddle, then the two rows of
// each as in the 256 kernel; the column stage transforms the columns c and c + 8 together and needs no twiddle at
// all (|F2| = |W[k] - W[k+128]|).  G[kx][y]: 130 float2 per column, y kept at y ^ ((kx >> 3) & 1): conflict-free
// b64 stores of 16 neighbouring lanes and loads of a column pair (scripts/cryst_fft_model.py, second part).  The
// exchanges of the column stage run in the wave's row scratch (all 16 waves own one: G takes 66 KiB).
constexpr int CG_N = 128;
constexpr int CG_COL = CG_N + 2;                // float2 units per column of G
constexpr int CG_K = CG_N / 2 + 1;              // columns of the half spectrum: G always holds all 65
constexpr int CG_WAVES = 16;
constexpr int CG_PAIRS = 33;                    // column pairs (c, c + 8) + the column 64 alone

template <typename T>
struct CgRaw {                                  // 2 pixels of each of 4 rows
    typedef T __attribute__((ext_vector_type(2))) vec_t;
    vec_t r[4];
};

template <typename T, bool MASK, bool CORR = false>
__global__ void __launch_bounds__(CG_WAVES * 64)
k_cryst_fused128(const T *__restrict__ tile, int64_t ld, int64_t n_frames, const float *__restrict__ rmask_p,
                 const unsigned long long *__restrict__ rflags, const float *__restrict__ mask_p, int K,
                 float *__restrict__ out, int accumulate, const float *__restrict__ dmap_p,
                 const int *__restrict__ pair_ptr, const int *__restrict__ pcode, const float *__restrict__ patch,
                 int n_excl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cf_smem[];
    __shared__ float part[CG_WAVES];
    const int t = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    v2f *G = (v2f *)cf_smem;
    v2f *scr = G + CG_K * CG_COL + w * CF_SCR;
    const int sig = cf_sigma(t);

    CfLane c;
    v2f tz[2], tzr[2];                          // w256^-(sigma + 64 k2): Z2 out of W[k] - W[k + 128]
#pragma unroll
    for (int r = 1; r < 4; ++r) {
        double s, co;
        sincospi(-2.0 * (double)((t & 15) * r) / 64.0, &s, &co);
        c.tw[0][r - 1] = (v2f){(float)co, (float)s};
        sincospi(-2.0 * (double)((t & 3) * r) / 16.0, &s, &co);
        c.tw[1][r - 1] = (v2f){(float)co, (float)s};
        sincospi(-2.0 * (double)(sig * r) / 256.0, &s, &co);
        c.tw[2][r - 1] = (v2f){(float)co, (float)s};
#pragma unroll
        for (int pi = 0; pi < 3; ++pi) c.twr[pi][r - 1] = (v2f){-c.tw[pi][r - 1].y, c.tw[pi][r - 1].x};
    }
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
        double s, co;
        sincospi(2.0 * (double)(sig + 64 * k2) / 256.0, &s, &co);
        tz[k2] = (v2f){(float)co, (float)s};
        tzr[k2] = (v2f){-(float)s, (float)co};
    }
    {
        const int b2 = (t >> 2) & 3, b0 = t & 3;
        const int base_b = 64 * b2 + t, base_c = 64 * b0 + t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            c.wB[r] = 64 * r + (t ^ (r << 2));
            c.rB[r] = base_b ^ (r << 2);
            c.wC[r] = 64 * r + (t ^ r);
            c.rC[r] = base_c ^ r;
        }
    }
    const int back = cf_sigma((64 - sig) & 63) * 4;                 // ds_bpermute address: the lane of 64 - sigma
    const int sw = (sig >> 3) & 1;
    const int g_s = sig * CG_COL + sw, g_d = sig * CG_COL + 1 - sw; // + y (even): where 4 A[kx], 4 B[kx] of a row pair go
    // column stage: lane 16 j + m loads y = 32 r + 2 m + (j >> 1) of the column c (j even) or c + 8 (j odd)
    const int jj = t >> 4, mm = t & 15;
    const int col_rd = (jj & 1) * (8 * CG_COL) + ((2 * mm + (jj >> 1)) ^ (jj & 1));
    const int col_rd64 = 2 * mm + (jj >> 1);                        // (the pair "64, 64")

    // columns the ring does not touch are never stored: zero them once, the transforms read them
    for (int i = K * CG_COL + (int)threadIdx.x; i < CG_K * CG_COL; i += CG_WAVES * 64) G[i] = (v2f){0.f, 0.f};

    const unsigned long long fl = MASK ? rflags[0] : 0;
    auto masked = [&](int q) -> bool { return MASK && ((fl >> q) & 1); };
    typedef typename CgRaw<T>::vec_t vec_t;
    auto load = [&](CgRaw<T> &dst, int64_t f, int q) {
        if (f < n_frames) {
            const T *row = tile + f * ld + (4 * q) * CG_N;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                dst.r[i] = __builtin_nontemporal_load((const vec_t *)(row + i * CG_N + 2 * t));
        }
    };
    auto convert = [&](const CgRaw<T> &b, int q, int64_t fr, v2f (&u)[4]) {
        CfMask bm;
        const bool mk = CORR || masked(q);
        if (mk) {
            const float *row = rmask_p + q * (4 * CG_N);
            bm.m01 = *(const v4f *)(row + 8 * t);
            bm.m23 = *(const v4f *)(row + 8 * t + 4);
        }
        u[0] = (v2f){(float)b.r[0][0], (float)b.r[1][0]};
        u[1] = (v2f){(float)b.r[2][0], (float)b.r[3][0]};
        u[2] = (v2f){(float)b.r[0][1], (float)b.r[1][1]};
        u[3] = (v2f){(float)b.r[2][1], (float)b.r[3][1]};
        if constexpr (CORR) {
            const float *row = dmap_p + q * (4 * CG_N);
            const v4f d01 = *(const v4f *)(row + 8 * t), d23 = *(const v4f *)(row + 8 * t + 4);
            u[0] -= d01.xy; u[1] -= d01.zw;
            u[2] -= d23.xy; u[3] -= d23.zw;
        }
        if (mk) {
            u[0] *= bm.m01.xy; u[1] *= bm.m01.zw;
            u[2] *= bm.m23.xy; u[3] *= bm.m23.zw;
        }
        if constexpr (CORR) {
            if (n_excl > 0 && fr < n_frames) {
                v2f (&uu)[1][4] = reinterpret_cast<v2f (&)[1][4]>(u);
                cf_apply_patches<1>(uu, t, q, fr, pair_ptr, pcode, patch, n_excl);
            }
        }
    };
    // the two rows of one pair out of Z[k] (k2 = 0: za, k2 = 1: zb), both x 2; rows y, y + 1
    auto store_pair = [&](v2f za, v2f zb, int y) {
        const float gx = t == 0 ? za.x : zb.x, gy = t == 0 ? za.y : zb.y;
        const float cr = __int_as_float(__builtin_amdgcn_ds_bpermute(back, __float_as_int(gx)));
        const float ci = __int_as_float(__builtin_amdgcn_ds_bpermute(back, __float_as_int(gy)));
        if (sig < K) {
            G[g_s + y] = (v2f){za.x + cr, za.y - ci};
            G[g_d + y] = (v2f){za.y + ci, cr - za.x};
        }
        if (K == CG_K && t == 0) {                  // kx = 64 is its own partner
            G[64 * CG_COL + y] = (v2f){2.f * zb.x, 0.f};
            G[64 * CG_COL + y + 1] = (v2f){2.f * zb.y, 0.f};
        }
    };
    auto four_rows = [&](v2f (&u)[4], int q) {
        cf_swap_a(u);
        cf_core(scr, c, u);
        const v2f d0 = u[0] - u[2], d1 = u[1] - u[3];
        store_pair(u[0] + u[2], u[1] + u[3], 4 * q);
        store_pair(cf_mul(d0, tz[0], tzr[0]), cf_mul(d1, tz[1], tzr[1]), 4 * q + 2);
    };

    CgRaw<T> ba, bb;
    load(ba, blockIdx.x, w);
    load(bb, blockIdx.x, w + CG_WAVES);
    __syncthreads();                                // (the zeroed columns)
    for (int64_t f = blockIdx.x; f < n_frames; f += gridDim.x) {
        v2f ua[4], ub[4];
        convert(ba, w, f, ua);
        convert(bb, w + CG_WAVES, f, ub);
        load(ba, f + gridDim.x, w);
        load(bb, f + gridDim.x, w + CG_WAVES);
        four_rows(ua, w);
        four_rows(ub, w + CG_WAVES);
        __syncthreads();
        // ---- column pairs p = w + 16 i
        float acc = 0.f;
        for (int p = w; p < CG_PAIRS; p += CG_WAVES) {
            const int c1 = p < 32 ? (p >> 3) * 16 + (p & 7) : 64;
            if (c1 >= K) continue;
            float m[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) m[i] = mask_p[(p * 4 + i) * 64 + t];
            const v2f *col = G + c1 * CG_COL + (p < 32 ? col_rd : col_rd64);
            v2f u[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) u[r] = col[32 * r];
            cf_core(scr, c, u);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const v2f f1 = u[k2] + u[k2 + 2], f2 = u[k2] - u[k2 + 2];
                if (__builtin_amdgcn_ballot_w64(m[k2] != 0.f)) {
                    const float a = __builtin_amdgcn_sqrtf(f1.x * f1.x + f1.y * f1.y);
                    acc += m[k2] != 0.f ? a * m[k2] : 0.f;
                }
                if (__builtin_amdgcn_ballot_w64(m[2 + k2] != 0.f)) {
                    const float a = __builtin_amdgcn_sqrtf(f2.x * f2.x + f2.y * f2.y);
                    acc += m[2 + k2] != 0.f ? a * m[2 + k2] : 0.f;
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (t == 0) part[w] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < CG_WAVES; ++i) v += part[i];
            v *= 0.125f;
            out[f] = accumulate ? out[f] + v : v;
        }
    }
}

// masks of the 128 kernel in lane order:
//   mask_p[p][i][l]  = half_mask[sigma(l) + 64 (i & 1)][c(p) + 8 (i >> 1)]   (pair p: c = 16 (p >> 3) + (p & 7); p = 32: 64, no partner)
//   rmask_p[q][8 t + e] = real_mask[4 q + (e & 3)][2 t + (e >> 2)];   rflags[0] bit q: a value other than 1 in rows 4q .. 4q+3
__global__ void __launch_bounds__(256)
k_cryst_masks128(const float *__restrict__ half_mask, float *__restrict__ mask_p,
                 const float *__restrict__ real_mask, float *__restrict__ rmask_p,
                 unsigned long long *__restrict__ rflags) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < CG_PAIRS * 4 * 64) {
        const int p = i >> 8, e = (i >> 6) & 3, l = i & 63;
        const int c1 = p < 32 ? (p >> 3) * 16 + (p & 7) : 64;
        const int kx = c1 + 8 * (e >> 1), ky = cf_sigma(l) + 64 * (e & 1);
        mask_p[i] = (p == 32 && (e >> 1)) ? 0.f : half_mask[ky * CG_K + kx];
    }
    if (real_mask && i < CG_N * CG_N) {
        const int q = i >> 9, rem = i & 511, tt = rem >> 3, e = rem & 7;
        const float v = real_mask[(4 * q + (e & 3)) * CG_N + 2 * tt + (e >> 2)];
        rmask_p[i] = v;
        if (__builtin_amdgcn_ballot_w64(v != 1.f) && (threadIdx.x & 63) == 0) atomicOr(&rflags[0], 1ull << q);
    }
}

template <typename T>
static int launch_fused128(const void *tile, int64_t ld, int64_t n_frames, const float *real_mask,
                           const unsigned long long *rflags, const float *mask_p, int K, float *out,
                           int accumulate, int n_cu, hipStream_t stream, const CfCorr *corr = nullptr) {
    auto kern = corr ? k_cryst_fused128<T, true, true>
                     : (real_mask ? k_cryst_fused128<T, true> : k_cryst_fused128<T, false>);
    const int lds = CG_K * CG_COL * 8 + CG_WAVES * CF_SCR * 8;
    int device = 0;
    LTMI_HIP(hipGetDevice(&device));
    static bool attr_set[16][3] = {{false}};
    const int variant = corr ? 2 : (real_mask ? 1 : 0);
    if (!attr_set[device & 15][variant]) {
        LTMI_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set[device & 15][variant] = true;
    }
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(n_frames, n_cu));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(CG_WAVES * 64), (size_t)lds, stream, (const T *)tile, ld, n_frames,
                       real_mask, rflags, mask_p, K, out, accumulate, corr ? (const float *)corr->dmap_p : nullptr,
                       corr ? corr->pair_ptr : nullptr, corr ? corr->pcode : nullptr, corr ? corr->patch : nullptr,
                       corr ? corr->n_excl : 0);
    LTMI_HIP(hipGetLastError());
    return LTMI_OK;
}

static int cryst_fused128(const void *tile, int tile_dtype, int64_t n_frames, int64_t ld, const float *real_mask,
                          const float *half_mask, int n_cols, float *work, float *out, int accumulate, int n_cu,
                          hipStream_t stream, bool *handled, const CfCorr *corr = nullptr) {
    if (n_cols < 1 || n_cols > CG_K) return LTMI_OK;
    const size_t esz = (size_t)dtype_size(tile_dtype);
    if (esz > 4 || tile_dtype == LTMI_F64) return LTMI_OK;
    if ((uintptr_t)tile % (2 * esz) != 0 || ld % 2 != 0) return LTMI_OK;
    float *mask_p = work, *rmask_p = work + CG_PAIRS * 4 * 64;
    unsigned long long *rflags = (unsigned long long *)(rmask_p + CG_N * CG_N);
    LTMI_HIP(hipMemsetAsync(rflags, 0, 8, stream));
    hipLaunchKernelGGL(k_cryst_masks128, dim3((unsigned)(CG_N * CG_N / 256)), dim3(256), 0, stream, half_mask,
                       mask_p, real_mask, rmask_p, rflags);
    if (corr)                                       // (the dark map in the same lane order)
        hipLaunchKernelGGL(k_cryst_masks128, dim3((unsigned)(CG_N * CG_N / 256)), dim3(256), 0, stream, half_mask,
                           mask_p, corr->d_px, corr->dmap_p, corr->dummy_flags);
    if (real_mask) real_mask = rmask_p;
    int rc = LTMI_E_DTYPE;
    switch (tile_dtype) {
        case LTMI_BOOL:
        case LTMI_U8: rc = launch_fused128<uint8_t>(tile, ld, n_frames, real_mask, rflags, mask_p, n_cols, out, accumulate, n_cu, stream, corr); break;
        case LTMI_I8: rc = launch_fused128<int8_t>(tile, ld, n_frames, real_mask, rflags, mask_p, n_cols, out, accumulate, n_cu, stream, corr); break;
        case LTMI_U16: rc = launch_fused128<uint16_t>(tile, ld, n_frames, real_mask, rflags, mask_p, n_cols, out, accumulate, n_cu, stream, corr); break;
        case LTMI_I16: rc = launch_fused128<int16_t>(tile, ld, n_frames, real_mask, rflags, mask_p, n_cols, out, accumulate, n_cu, stream, corr); break;
        case LTMI_U32: rc = launch_fused128<uint32_t>(tile, ld, n_frames, real_mask, rflags, mask_p, n_cols, out, accumulate, n_cu, stream, corr); break;
        case LTMI_I32: rc = launch_fused128<int32_t>(tile, ld, n_frames, real_mask, rflags, mask_p, n_cols, out, accumulate, n_cu, stream, corr); break;
        case LTMI_F32: rc = launch_fused128<float>(tile, ld, n_frames, real_mask, rflags, mask_p, n_cols, out, accumulate, n_cu, stream, corr); break;
        default: return LTMI_OK;
    }
    if (rc == LTMI_OK) *handled = true;
    return rc;
}

// ---- frames with edges of 256 / 512 / 1024 pixels (256 x 256: wide rings only): two kernels, the ring's columns of
// the row transforms through HBM ------------------------------------------------------------------------------------
// A frame's K columns of row spectra (K <= N / 2 + 1) are 8 N bytes each: they do not fit the LDS.  k_cryst_rows
// writes them to a workspace G[frame][kx][y] (float2), k_cryst_cols transforms one column per wave and sums the
// ring: 2 + 2 * 8 K / N bytes of traffic per pixel and no spectrum beyond the ring's columns, where the hipFFT route
// moves ~34.  An N = 256 M point transform (M = 2, 4) = the M 256-point transforms of the samples n = q mod M
// (cf_core M times: a lane's 4 M consecutive samples are 4 of each, the layout cf_core starts from), twiddles
// w_N^(q k) and one radix-M butterfly over q:  X[k + 256 m] = sum_q (-i)^(q m) [for M = 4] w_N^(q k) E_q[k].
constexpr int CH_WAVES = 8;
constexpr int CH_KMAX_ANY = 513;                 // columns of the widest half spectrum these kernels take (N = 1024)
constexpr int CH_STAGE = 9;                      // float4 units per column of the staging tile: 8 row pairs + 16 bytes

template <int M> struct ChTw { v2f w[M > 1 ? M - 1 : 1][4], wr[M > 1 ? M - 1 : 1][4]; };   // w_N^(q (sigma + 64 k2)), q = 1 .. M - 1

template <int M>
__device__ __forceinline__ void ch_lane_setup(int t, CfLane &c, ChTw<M> &h) {
    const int sig = cf_sigma(t);
#pragma unroll
    for (int r = 1; r < 4; ++r) {
        double s, co;
        sincospi(-2.0 * (double)((t & 15) * r) / 64.0, &s, &co);
        c.tw[0][r - 1] = (v2f){(float)co, (float)s};
        sincospi(-2.0 * (double)((t & 3) * r) / 16.0, &s, &co);
        c.tw[1][r - 1] = (v2f){(float)co, (float)s};
        sincospi(-2.0 * (double)(sig * r) / 256.0, &s, &co);
        c.tw[2][r - 1] = (v2f){(float)co, (float)s};
#pragma unroll
        for (int pi = 0; pi < 3; ++pi) c.twr[pi][r - 1] = (v2f){-c.tw[pi][r - 1].y, c.tw[pi][r - 1].x};
    }
#pragma unroll
    for (int q = 1; q < M; ++q)
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
            double s, co;
            sincospi(-2.0 * (double)(q * (sig + 64 * k2)) / (double)(256 * M), &s, &co);
            h.w[q - 1][k2] = (v2f){(float)co, (float)s};
            h.wr[q - 1][k2] = (v2f){-(float)s, (float)co};
        }
    const int b2 = (t >> 2) & 3, b0 = t & 3;
    const int base_b = 64 * b2 + t, base_c = 64 * b0 + t;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        c.wB[r] = 64 * r + (t ^ (r << 2));
        c.rB[r] = base_b ^ (r << 2);
        c.wC[r] = 64 * r + (t ^ r);
        c.rC[r] = base_c ^ r;
    }
}

// u[q][j] = z[4 M t + M j + q]  ->  u[m][k2] = X[sigma + 64 k2 + 256 m]
// SWAPPED: the input is in the layout after the first swap already: u[q][r] = z[M (64 r + 4 (t & 15) + (t >> 4)) + q]
template <int M, bool SWAPPED = false>
__device__ __forceinline__ void ch_fft(v2f *scr, const CfLane &c, const ChTw<M> &h, v2f (&u)[M][4]) {
#pragma unroll
    for (int q = 0; q < M; ++q) {
        if (!SWAPPED) cf_swap_a(u[q]);
        cf_core(scr, c, u[q]);
    }
    if (M == 1) return;                          // (256 points: one transform, nothing to combine)
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
        if (M == 2) {
            const v2f w = cf_mul(u[1][k2], h.w[0][k2], h.wr[0][k2]);
            u[1][k2] = u[0][k2] - w;
            u[0][k2] = u[0][k2] + w;
        } else {
            v2f b[4];
            b[0] = u[0][k2];
#pragma unroll
            for (int q = 1; q < M; ++q) b[q] = cf_mul(u[q][k2], h.w[q - 1][k2], h.wr[q - 1][k2]);
            cf_bfly<true>(b);
#pragma unroll
            for (int q = 0; q < M; ++q) u[q][k2] = b[q];
        }
    }
}

// grid: persistent over groups of 8 row pairs (frame f, pairs 8 g .. 8 g + 7: one per wave).  The 2 x K spectra of a
// group are staged in the LDS and leave as 128 contiguous bytes per column: G[(f K + kx) N + 16 g ..].
template <typename T, bool MASK, int M, bool CORR = false>
__global__ void __launch_bounds__(CH_WAVES * 64)
k_cryst_rows(const T *__restrict__ tile, int64_t ld, int64_t n_frames, const float *__restrict__ rmask_p,
             const unsigned long long *__restrict__ rflags, int K, int H, v2f *__restrict__ G,
             const float *__restrict__ dmap_p, const int *__restrict__ pair_ptr, const int *__restrict__ pcode,
             const float *__restrict__ patch, int n_excl) {
    constexpr int N = 256 * M;                                      // row length; H rows per frame
    extern __shared__ __attribute__((aligned(16))) unsigned char cf_smem[];
    const int t = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    v4f *stage = (v4f *)cf_smem;                                    // [K][CH_STAGE]
    v2f *scr = (v2f *)(stage + K * CH_STAGE) + w * CF_SCR;
    const int sig = cf_sigma(t);
    CfLane c;
    ChTw<M> h;
    ch_lane_setup<M>(t, c, h);
    const int back = cf_sigma((64 - sig) & 63) * 4;
    const int st_col = sig * CH_STAGE + (w ^ (2 * ((sig >> 3) & 1)));   // + 64 c CH_STAGE (kx bit 3 = sigma bit 3)
    typedef T __attribute__((ext_vector_type(4 * M))) vec_t;
    const int GROUPS = H / 2 / CH_WAVES;                            // per frame
    const int64_t n_groups = n_frames * GROUPS;
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int64_t f = grp / GROUPS;
        const int g = (int)(grp - f * GROUPS);
        const int yp = CH_WAVES * g + w;
        // (loading a group's pixels one group ahead changed nothing: the kernel is bound by the transforms, at the
        // rate per 256-point transform of k_cryst_fused)
        const T *row = tile + f * ld + (int64_t)(2 * yp) * N + 4 * M * t;
        const vec_t ra = __builtin_nontemporal_load((const vec_t *)row);
        const vec_t rb = __builtin_nontemporal_load((const vec_t *)(row + N));
        v2f u[M][4];
#pragma unroll
        for (int q = 0; q < M; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) u[q][j] = (v2f){(float)ra[M * j + q], (float)rb[M * j + q]};
        if constexpr (CORR) {
            const v4f *dk = (const v4f *)(dmap_p + (int64_t)yp * (2 * N) + 8 * M * t);
#pragma unroll
            for (int q = 0; q < M; ++q) {
                const v4f d0 = dk[2 * q], d1 = dk[2 * q + 1];
                u[q][0] -= d0.xy; u[q][1] -= d0.zw; u[q][2] -= d1.xy; u[q][3] -= d1.zw;
            }
        }
        if (CORR || (MASK && ((rflags[yp >> 6] >> (yp & 63)) & 1))) {
            const v4f *mk = (const v4f *)(rmask_p + (int64_t)yp * (2 * N) + 8 * M * t);
#pragma unroll
            for (int q = 0; q < M; ++q) {
                const v4f m0 = mk[2 * q], m1 = mk[2 * q + 1];
                u[q][0] *= m0.xy; u[q][1] *= m0.zw; u[q][2] *= m1.xy; u[q][3] *= m1.zw;
            }
        }
        if constexpr (CORR) {
            if (n_excl > 0) cf_apply_patches<M>(u, t, yp, f, pair_ptr, pcode, patch, n_excl);
        }
        ch_fft<M>(scr, c, h, u);
        // two real rows out of one complex transform (see k_cryst_fused): the partner X[N - kx] of kx = sigma + 64 c
        // (c = k2 + 4 m) is block 4 M - 1 - c of the lane of 64 - sigma; sigma = 0: X[0] itself / block 4 M - c of lane 0
#pragma unroll
        for (int cb = 0; cb < 2 * M; ++cb) {
            if (64 * cb >= K) break;
            const int k2 = cb & 3, m = cb >> 2, cp = (4 * M - cb) % (4 * M);
            const v2f mine = u[cp >> 2][cp & 3], theirs = u[M - 1 - m][3 - k2], zk = u[m][k2];
            const float gx = t == 0 ? mine.x : theirs.x, gy = t == 0 ? mine.y : theirs.y;
            const float cr = __int_as_float(__builtin_amdgcn_ds_bpermute(back, __float_as_int(gx)));
            const float ci = __int_as_float(__builtin_amdgcn_ds_bpermute(back, __float_as_int(gy)));
            if (sig + 64 * cb < K)
                stage[st_col + 64 * cb * CH_STAGE] = (v4f){zk.x + cr, zk.y - ci, zk.y + ci, cr - zk.x};
        }
        if (K == N / 2 + 1 && t == 0)                                // kx = N / 2 is its own partner
            stage[(N / 2) * CH_STAGE + w] =                        // X[N / 2]: block (N / 2) / 64 of lane 0
                (v4f){2.f * u[M / 2][M == 1 ? 2 : 0].x, 0.f, 2.f * u[M / 2][M == 1 ? 2 : 0].y, 0.f};
        __syncthreads();
        for (int i = threadIdx.x; i < K * 8; i += CH_WAVES * 64) {
            const int kx = i >> 3, pi = i & 7;
            const v4f v = stage[kx * CH_STAGE + (kx < N / 2 ? pi ^ (2 * ((kx >> 3) & 1)) : pi)];
            *(v4f *)(G + ((f * K + kx) * H + 16 * g + 2 * pi)) = v;
        }
        __syncthreads();
    }
}

// one column per wave: N points of G -> sum of |F[ky][kx]| * mask; one workgroup per frame (persistent)
template <int M>
__global__ void __launch_bounds__(CH_WAVES * 64)
k_cryst_cols(const v2f *__restrict__ G, int64_t n_frames, const float *__restrict__ mask_p, int K,
             float *__restrict__ out, int accumulate) {
    constexpr int N = 256 * M;
    __shared__ __attribute__((aligned(16))) v2f scr_all[CH_WAVES * CF_SCR];
    __shared__ float part[CH_WAVES];
    const int t = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    v2f *scr = scr_all + w * CF_SCR;
    CfLane c;
    ChTw<M> h;
    ch_lane_setup<M>(t, c, h);
    for (int64_t f = blockIdx.x; f < n_frames; f += gridDim.x) {
        float acc = 0.f;
        for (int kx = w; kx < K; kx += CH_WAVES) {
            // lane t takes the samples M (64 r + 4 (t & 15) + (t >> 4)) + q -- the layout cf_core starts from, so the
            // first lane exchange is not needed -- and a load instruction covers 64 M contiguous float2 (a lane's 4 M
            // CONSECUTIVE samples made every instruction touch all of the column's lines: 175 us per 1 024 columns
            // of 129 x 512 from HBM, 2.3 x the time of this order)
            const v2f *col = G + (f * K + kx) * N + M * (4 * (t & 15) + (t >> 4));
            v2f u[M][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (M == 1) {
                    u[0][r] = __builtin_nontemporal_load(col + 64 * r);
                } else {
#pragma unroll
                    for (int i = 0; i < M / 2; ++i) {
                        const v4f q = __builtin_nontemporal_load((const v4f *)(col + M * 64 * r) + i);
                        u[2 * i][r] = q.xy;
                        u[(2 * i + 1) % M][r] = q.zw;
                    }
                }
            }
            const float *mrow = mask_p + (int64_t)kx * (4 * M * 64) + t;
            float mk[4 * M];
#pragma unroll
            for (int q = 0; q < 4 * M; ++q) mk[q] = mrow[q * 64];
            ch_fft<M, true>(scr, c, h, u);
#pragma unroll
            for (int q = 0; q < 4 * M; ++q)
                if (__builtin_amdgcn_ballot_w64(mk[q] != 0.f)) {
                    const v2f z = u[q >> 2][q & 3];
                    const float a = __builtin_amdgcn_sqrtf(z.x * z.x + z.y * z.y);
                    acc += mk[q] != 0.f ? a * mk[q] : 0.f;
                }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (t == 0) part[w] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < CH_WAVES; ++i) v += part[i];
            v *= 0.5f;
            out[f] = accumulate ? out[f] + v : v;
        }
        __syncthreads();
    }
}

// masks of these kernels in lane order (N = 256 M):
//   mask_p[kx][q][l]        = half_mask[sigma(l) + 64 (q & 3) + 256 (q >> 2)][kx]          (q < 4 M)
//   rmask_p[y'][8 M t + e]  = real_mask[2 y' + (e & 1)][4 M t + M ((e >> 1) & 3) + (e >> 3)]   (sample class q = e >> 3 first,
//                             rows a / b interleaved);  rflags bit y' (N / 128 words): the pair holds a value other than 1
__global__ void __launch_bounds__(256)
k_cryst_masks_n(const float *__restrict__ half_mask, int W, int H, int K, float *__restrict__ mask_p,
                const float *__restrict__ real_mask, float *__restrict__ rmask_p,
                unsigned long long *__restrict__ rflags) {
    // mask_p: lane order of the COLUMN transforms (H points); rmask_p: of the ROW transforms (W points)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int M = W / 256;
    if (i < (int64_t)K * H) {
        const int kx = (int)(i / H), r = (int)(i - (int64_t)kx * H), q = r >> 6, l = r & 63;
        const int ky = cf_sigma(l) + 64 * (q & 3) + 256 * (q >> 2);
        mask_p[i] = half_mask[(int64_t)ky * (W / 2 + 1) + kx];
    }
    if (real_mask && i < (int64_t)W * H) {
        const int yp = (int)(i / (2 * W)), rem = (int)(i - (int64_t)yp * 2 * W), tt = rem / (8 * M), e = rem - tt * 8 * M;
        const float v = real_mask[(int64_t)(2 * yp + (e & 1)) * W + 4 * M * tt + M * ((e >> 1) & 3) + (e >> 3)];
        rmask_p[i] = v;
        if (__builtin_amdgcn_ballot_w64(v != 1.f) && (threadIdx.x & 63) == 0)
            atomicOr(&rflags[yp >> 6], 1ull << (yp & 63));
    }
}

template <typename T, int M>
static int launch_rows(const void *tile, int64_t ld, int64_t n_frames, const float *real_mask,
                       const unsigned long long *rflags, int K, int H, v2f *G, int n_cu, hipStream_t stream,
                       const CfCorr *corr = nullptr, int64_t patch_frame0 = 0) {
    auto kern = corr ? k_cryst_rows<T, true, M, true>
                     : (real_mask ? k_cryst_rows<T, true, M> : k_cryst_rows<T, false, M>);
    constexpr int N = 256 * M;
    const int lds = K * CH_STAGE * 16 + CH_WAVES * CF_SCR * 8;
    int device = 0;
    LTMI_HIP(hipGetDevice(&device));
    static bool attr_set[16][3] = {{false}};
    const int variant = corr ? 2 : (real_mask ? 1 : 0);
    if (!attr_set[device & 15][variant]) {
        LTMI_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (N / 2 + 1) * CH_STAGE * 16 + CH_WAVES * CF_SCR * 8));
        attr_set[device & 15][variant] = true;
    }
    // persistent workgroups: exactly as many as are resident at once (a workgroup that waits for a CU would start
    // its share of the groups when the others are done with theirs)
    static int resident[16][3][CH_KMAX_ANY + 2] = {{{0}}};          // per device, mask variant and K (the LDS size)
    int &per_cu = resident[device & 15][variant][K];
    if (per_cu == 0) {
        LTMI_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)kern, CH_WAVES * 64, (size_t)lds));
        per_cu = std::max(1, per_cu);
    }
    const int64_t groups = n_frames * (H / 2 / CH_WAVES);
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(groups, (int64_t)n_cu * std::max(1, per_cu)));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(CH_WAVES * 64), (size_t)lds, stream, (const T *)tile, ld, n_frames,
                       real_mask, rflags, K, H, G, corr ? (const float *)corr->dmap_p : nullptr,
                       corr ? corr->pair_ptr : nullptr, corr ? corr->pcode : nullptr,
                       corr ? corr->patch + patch_frame0 * corr->n_excl : nullptr, corr ? corr->n_excl : 0);
    LTMI_HIP(hipGetLastError());
    return LTMI_OK;
}

// frames per pass of the workspace: LTMI_CRYST_PASS_MIB (measurement switch) caps the bytes a pass writes and reads back
static int64_t cryst_pass_frames(int64_t gbuf_frames, int64_t bytes_per_frame, int n_cu) {
    static const long mib = [] { const char *e = getenv("LTMI_CRYST_PASS_MIB"); return e ? atol(e) : 0L; }();
    if (mib <= 0) return gbuf_frames;
    const int64_t n = std::max<int64_t>(1, ((int64_t)mib << 20) / bytes_per_frame);
    return std::min(gbuf_frames, n);
}

// H x W frames, W = 256 M, H = 256 MH (M, MH = 1, 2, 4); gbuf: workspace of gbuf_frames * n_cols * H float2
template <int M, int MH = M>
static int cryst_rows_cols(const void *tile, int tile_dtype, int64_t n_frames, int64_t ld, const float *real_mask,
                           const float *half_mask, int n_cols, float *work, void *gbuf, int64_t gbuf_frames,
                           float *out, int accumulate, int n_cu, hipStream_t stream, bool *handled,
                           const CfCorr *corr = nullptr) {
    constexpr int N = 256 * M, H = 256 * MH;
    if (n_cols < 1 || n_cols > N / 2 + 1 || !gbuf || gbuf_frames < 1) return LTMI_OK;
    const size_t esz = (size_t)dtype_size(tile_dtype);
    if (esz > 4 || tile_dtype == LTMI_F64) return LTMI_OK;
    if ((uintptr_t)tile % std::min<size_t>(16, 4 * M * esz) != 0 || ld % (4 * M) != 0) return LTMI_OK;
    float *mask_p = work, *rmask_p = work + (int64_t)(N / 2 + 1) * H;
    unsigned long long *rflags = (unsigned long long *)(rmask_p + (int64_t)N * H);
    LTMI_HIP(hipMemsetAsync(rflags, 0, 64, stream));
    hipLaunchKernelGGL(k_cryst_masks_n, dim3((unsigned)((int64_t)N * H / 256)), dim3(256), 0, stream, half_mask, N, H,
                       n_cols, mask_p, real_mask, rmask_p, rflags);
    if (corr)                                       // (the dark map in the same lane order)
        hipLaunchKernelGGL(k_cryst_masks_n, dim3((unsigned)((int64_t)N * H / 256)), dim3(256), 0, stream, half_mask, N, H,
                           0, mask_p, corr->d_px, corr->dmap_p, corr->dummy_flags);
    const float *rm = real_mask ? rmask_p : nullptr;
    gbuf_frames = cryst_pass_frames(gbuf_frames, (int64_t)n_cols * H * (int64_t)sizeof(v2f), n_cu);
    for (int64_t f0 = 0; f0 < n_frames; f0 += gbuf_frames) {
        const int64_t n = std::min<int64_t>(gbuf_frames, n_frames - f0);
        const void *src = (const char *)tile + (size_t)f0 * ld * esz;
        int rc = LTMI_E_DTYPE;
        switch (tile_dtype) {
            case LTMI_BOOL:
            case LTMI_U8: rc = launch_rows<uint8_t, M>(src, ld, n, rm, rflags, n_cols, H, (v2f *)gbuf, n_cu, stream, corr, f0); break;
            case LTMI_I8: rc = launch_rows<int8_t, M>(src, ld, n, rm, rflags, n_cols, H, (v2f *)gbuf, n_cu, stream, corr, f0); break;
            case LTMI_U16: rc = launch_rows<uint16_t, M>(src, ld, n, rm, rflags, n_cols, H, (v2f *)gbuf, n_cu, stream, corr, f0); break;
            case LTMI_I16: rc = launch_rows<int16_t, M>(src, ld, n, rm, rflags, n_cols, H, (v2f *)gbuf, n_cu, stream, corr, f0); break;
            case LTMI_U32: rc = launch_rows<uint32_t, M>(src, ld, n, rm, rflags, n_cols, H, (v2f *)gbuf, n_cu, stream, corr, f0); break;
            case LTMI_I32: rc = launch_rows<int32_t, M>(src, ld, n, rm, rflags, n_cols, H, (v2f *)gbuf, n_cu, stream, corr, f0); break;
            case LTMI_F32: rc = launch_rows<float, M>(src, ld, n, rm, rflags, n_cols, H, (v2f *)gbuf, n_cu, stream, corr, f0); break;
            default: return LTMI_OK;
        }
        if (rc != LTMI_OK) return rc;
        static int per_cu = 0;                      // (one value per MH: registers and static LDS only)
        if (per_cu == 0) {
            LTMI_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_cryst_cols<MH>, CH_WAVES * 64, 0));
            per_cu = std::max(1, per_cu);
        }
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(n, (int64_t)n_cu * std::max(1, per_cu)));
        hipLaunchKernelGGL(k_cryst_cols<MH>, dim3(grid), dim3(CH_WAVES * 64), 0, stream, (const v2f *)gbuf, n,
                           (const float *)mask_p, n_cols, out + f0, accumulate);
        LTMI_HIP(hipGetLastError());
    }
    *handled = true;
    return LTMI_OK;
}

// -> LTMI_OK with *handled = true when a fused kernel ran (256 x 256 or 128 x 128 frames)
static int cryst_fused_impl(const void *tile, int tile_dtype, int64_t n_frames, int64_t ld, int sig_h, int sig_w,
                            const float *real_mask, const float *half_mask, int n_cols, float *mask_t, void *gbuf,
                            int64_t gbuf_frames, float *out, int accumulate, int n_cu, hipStream_t stream, bool *handled,
                            const CfCorr *corr) {
    *handled = false;
    if (!mask_t) return LTMI_OK;
    if ((sig_h == 256 || sig_h == 512 || sig_h == 1024) && (sig_w == 256 || sig_w == 512 || sig_w == 1024) &&
        !(sig_h == 256 && sig_w == 256)) {
        // rows of 256 M points, columns of 256 MH points: the ring's columns through the workspace
#define LTMI_CRYST_RC(MW_, MH_)                                                                                   \
        if (sig_w == 256 * MW_ && sig_h == 256 * MH_)                                                            \
            return cryst_rows_cols<MW_, MH_>(tile, tile_dtype, n_frames, ld, real_mask, half_mask, n_cols, mask_t, \
                                             gbuf, gbuf_frames, out, accumulate, n_cu, stream, handled, corr);
        LTMI_CRYST_RC(1, 2) LTMI_CRYST_RC(1, 4) LTMI_CRYST_RC(2, 1) LTMI_CRYST_RC(2, 2) LTMI_CRYST_RC(2, 4)
        LTMI_CRYST_RC(4, 1) LTMI_CRYST_RC(4, 2) LTMI_CRYST_RC(4, 4)
#undef LTMI_CRYST_RC
    }
    if (sig_h == CG_N && sig_w == CG_N)
        return cryst_fused128(tile, tile_dtype, n_frames, ld, real_mask, half_mask, n_cols, mask_t, out, accumulate,
                              n_cu, stream, handled, corr);
    if (sig_h != CF_N || sig_w != CF_N || n_cols < 1) return LTMI_OK;
    if (n_cols > CF_KMAX)                        // a ring too wide for the LDS: the workspace kernels with M = 1
        return cryst_rows_cols<1>(tile, tile_dtype, n_frames, ld, real_mask, half_mask, n_cols, mask_t, gbuf,
                                  gbuf_frames, out, accumulate, n_cu, stream, handled, corr);
    const size_t esz = (size_t)dtype_size(tile_dtype);
    if (esz > 4 || tile_dtype == LTMI_F64) return LTMI_OK;
    if ((uintptr_t)tile % (4 * esz) != 0 || ld % 4 != 0) return LTMI_OK;
    float *rmask_p = mask_t + (int64_t)CF_KMAX * CF_N;
    unsigned long long *rflags = (unsigned long long *)(rmask_p + CF_N * CF_N);
    if (real_mask) LTMI_HIP(hipMemsetAsync(rflags, 0, 16, stream));
    hipLaunchKernelGGL(k_cryst_masks, dim3((unsigned)(CF_N * CF_N / 256)), dim3(256), 0, stream, half_mask,
                       sig_w / 2 + 1, n_cols, mask_t, real_mask, rmask_p, rflags);
    if (real_mask) real_mask = rmask_p;
    int rc = LTMI_E_DTYPE;
    if (corr) {
        // the dark map in the same lane order (K = 0: the half mask is in place), then the CORR instantiation
        hipLaunchKernelGGL(k_cryst_masks, dim3((unsigned)(CF_N * CF_N / 256)), dim3(256), 0, stream, half_mask,
                           sig_w / 2 + 1, 0, mask_t, corr->d_px, corr->dmap_p, corr->dummy_flags);
        switch (tile_dtype) {
            case LTMI_BOOL:
            case LTMI_U8: rc = launch_fused_corr<uint8_t>(tile, ld, n_frames, real_mask, rflags, mask_t, n_cols, corr, out, accumulate, n_cu, stream); break;
            case LTMI_I8: rc = launch_fused_corr<int8_t>(tile, ld, n_frames, real_mask, rflags, mask_t, n_cols, corr, out, accumulate, n_cu, stream); break;
            case LTMI_U16: rc = launch_fused_corr<uint16_t>(tile, ld, n_frames, real_mask, rflags, mask_t, n_cols, corr, out, accumulate, n_cu, stream); break;
            case LTMI_I16: rc = launch_fused_corr<int16_t>(tile, ld, n_frames, real_mask, rflags, mask_t, n_cols, corr, out, accumulate, n_cu, stream); break;
            case LTMI_U32: rc = launch_fused_corr<uint32_t>(tile, ld, n_frames, real_mask, rflags, mask_t, n_cols, corr, out, accumulate, n_cu, stream); break;
            case LTMI_I32: rc = launch_fused_corr<int32_t>(tile, ld, n_frames, real_mask, rflags, mask_t, n_cols, corr, out, accumulate, n_cu, stream); break;
            case LTMI_F32: rc = launch_fused_corr<float>(tile, ld, n_frames, real_mask, rflags, mask_t, n_cols, corr, out, accumulate, n_cu, stream); break;
            default: return LTMI_OK;
        }
        if (rc == LTMI_OK) *handled = true;
        return rc;
    }
    switch (tile_dtype) {
        case LTMI_BOOL:
        case LTMI_U8: rc = launch_fused<uint8_t>(tile, ld, n_frames, real_mask, rflags, mask_t, n_cols, out, accumulate, n_cu, stream); break;
        case LTMI_I8: rc = launch_fused<int8_t>(tile, ld, n_frames, real_mask, rflags, mask_t, n_cols, out, accumulate, n_cu, stream); break;
        case LTMI_U16: rc = launch_fused<uint16_t>(tile, ld, n_frames, real_mask, rflags, mask_t, n_cols, out, accumulate, n_cu, stream); break;
        case LTMI_I16: rc = launch_fused<int16_t>(tile, ld, n_frames, real_mask, rflags, mask_t, n_cols, out, accumulate, n_cu, stream); break;
        case LTMI_U32: rc = launch_fused<uint32_t>(tile, ld, n_frames, real_mask, rflags, mask_t, n_cols, out, accumulate, n_cu, stream); break;
        case LTMI_I32: rc = launch_fused<int32_t>(tile, ld, n_frames, real_mask, rflags, mask_t, n_cols, out, accumulate, n_cu, stream); break;
        case LTMI_F32: rc = launch_fused<float>(tile, ld, n_frames, real_mask, rflags, mask_t, n_cols, out, accumulate, n_cu, stream); break;
        default: return LTMI_OK;
    }
    if (rc == LTMI_OK) *handled = true;
    return rc;
}

// -> LTMI_OK with *handled = true when a fused kernel ran
int cryst_fused(const void *tile, int tile_dtype, int64_t n_frames, int64_t ld, int sig_h, int sig_w,
                const float *real_mask, const float *half_mask, int n_cols, float *mask_t, void *gbuf,
                int64_t gbuf_frames, float *out, int accumulate, int n_cu, hipStream_t stream, bool *handled) {
    return cryst_fused_impl(tile, tile_dtype, n_frames, ld, sig_h, sig_w, real_mask, half_mask, n_cols, mask_t, gbuf,
                            gbuf_frames, out, accumulate, n_cu, stream, handled, nullptr);
}

// RAW frames with detector corrections in ONE pass over the pixels (every shape the fused kernels take).
// `ws`: cryst_corr_workspace_bytes(h, w, n_frames, n_excl).
int cryst_fused_corrected(const void *tile, int tile_dtype, int64_t n_frames, int64_t ld, int sig_h, int sig_w,
                          const double *dark, const double *gain, const int32_t *excl, const int32_t *env,
                          const int32_t *cnt, int n_excl, int max_env, const float *real_mask, const float *half_mask,
                          int n_cols, float *mask_t, void *gbuf, int64_t gbuf_frames, void *ws, float *out,
                          int accumulate, int n_cu, hipStream_t stream, bool *handled) {
    *handled = false;
    if (!mask_t || !ws || !cryst_corr_takes(sig_h, sig_w, n_cols, tile_dtype, n_excl)) return LTMI_OK;
    const int64_t n_px = (int64_t)sig_h * sig_w;
    float *gm_px = (float *)ws, *d_px = gm_px + n_px, *dmap_p = d_px + n_px;
    unsigned long long *dummy = (unsigned long long *)(dmap_p + n_px);
    int *pair_ptr = (int *)(dummy + 8);
    int *pcode = pair_ptr + 520;
    float *patch = (float *)(pcode + std::max(n_excl, 1));
    const bool four_rows = sig_h == CG_N && sig_w == CG_N;
    const bool rows_cols = !four_rows && cryst_fused_needs_gbuf(sig_h, sig_w, n_cols);
    if (rows_cols && (!gbuf || gbuf_frames < 1)) return LTMI_OK;
    const int M = four_rows ? 1 : sig_w / 256;
    hipLaunchKernelGGL(k_cryst_corr_maps, dim3((unsigned)((n_px + 255) / 256)), dim3(256), 0, stream, dark, gain,
                       real_mask, n_px, gm_px, d_px);
    if (n_excl > 0) {
        hipLaunchKernelGGL(k_cryst_patch_index, dim3(1), dim3(256), 0, stream, excl, n_excl, sig_w,
                           four_rows ? sig_h / 4 : sig_h / 2, four_rows ? 1 : 0, M, pair_ptr, pcode, gm_px);
        const int64_t nt = n_frames * n_excl;
        const unsigned gb = (unsigned)((nt + 255) / 256);
#define LTMI_PATCH_CASE(T_)                                                                                       \
        hipLaunchKernelGGL((k_cryst_patch_values<T_>), dim3(gb), dim3(256), 0, stream, (const T_ *)tile, ld, n_frames, \
                           dark, gain, real_mask, excl, env, cnt, n_excl, max_env, patch);
        switch (tile_dtype) {
            case LTMI_BOOL:
            case LTMI_U8: LTMI_PATCH_CASE(uint8_t) break;
            case LTMI_I8: LTMI_PATCH_CASE(int8_t) break;
            case LTMI_U16: LTMI_PATCH_CASE(uint16_t) break;
            case LTMI_I16: LTMI_PATCH_CASE(int16_t) break;
            case LTMI_U32: LTMI_PATCH_CASE(uint32_t) break;
            case LTMI_I32: LTMI_PATCH_CASE(int32_t) break;
            case LTMI_F32: LTMI_PATCH_CASE(float) break;
            default: return LTMI_OK;
        }
#undef LTMI_PATCH_CASE
    }
    LTMI_HIP(hipGetLastError());
    CfCorr corr;
    corr.d_px = d_px;
    corr.dmap_p = dmap_p;
    corr.dummy_flags = dummy;
    corr.pair_ptr = pair_ptr;
    corr.pcode = pcode;
    corr.patch = patch;
    corr.n_excl = n_excl;
    return cryst_fused_impl(tile, tile_dtype, n_frames, ld, sig_h, sig_w, gm_px, half_mask, n_cols, mask_t, gbuf,
                            gbuf_frames, out, accumulate, n_cu, stream, handled, &corr);
}

}  // namespace ltmi
