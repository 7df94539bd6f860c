// Sparse mask stacks on the matrix cores of gfx950 (MI355X): blocked-ELL image + v_mfma_f32_16x16x4_f32
// (k_bell_apply: float32 / int8 / int16 frames) or exact float16 products on v_mfma_f32_16x16x32_f16 (k_bell_flat,
// further down: uint8 / uint16 frames).
//
//   out[f, k] (+)= sum_p tile[f, p] * M[p, k],   M sparse (n_px x n_masks)
//
// Same job as k_sell_apply in ltmi_sparse.hip (the replacement of the numba kernels _rmatmul_csr /
// _rmatmul_csc, src/libertem/common/numba/__init__.py:153-184, reached from
// MaskContainer/ApplyMasksUDF, src/libertem/udf/masks.py:296-338), for stacks whose masks are
// LOCALISED: neighbouring masks share pixels (ring / radial-bin stacks, src/libertem/masks.py
// radial_bins; BASELINE.json config C4).  There a group of 16 consecutive masks touches few
// pixels of a pixel chunk and those pixels are used by several masks of the group, so the
// (16 masks x touched pixels) block is worth multiplying densely:
//
//   * image: a chunk is 512 pixels of a frame -- 8 segments of 64 consecutive pixels that the builder
//     picks so that the work of a chunk is spread evenly over the waves (ChunkPlanner below); for
//     every (chunk, group of 16 masks) the sorted list of touched pixels,
//     padded to a multiple of 8, and the dense 16 x n block of mask values.  Two MFMA steps
//     (2 x 4 pixels) form one 768-byte "block record": lane l holds A[mask l&15][pixel l>>4] of
//     both steps and the two pixel numbers of its l>>4.  C4: 3.65 multiply-adds per stored mask
//     value, but on the matrix pipe and with ONE LDS read per 16 multiply-adds (the SELL kernel
//     needs an LDS gather per multiply-add and is bound by that).
//   * a workgroup = 16 waves (the whole CU, one workgroup at a time) owns 16*TILES frames and 1024
//     masks: wave j owns the groups g with g % 16 == j (4 of them -> 4 x TILES accumulator tiles in
//     registers for the whole sweep; interleaving the groups balances ring stacks, where the groups
//     touching a pixel chunk are consecutive).  TILES = 2 or 4 for 1/2-byte pixels, 1 or 2 for
//     float32, chosen per launch (launch_bell): a record's fixed cost -- ring read, gather
//     addresses, refill -- is paid once for all tiles, so 64 frames per workgroup cost 1.7x of 32,
//     but small launches need the many short workgroups.  (Round 1 / early round 2: 4 waves x 16
//     groups, two workgroups per CU, 32 frames: 0.95 ms per 16 384 frames of C4; now 0.84 ms.)
//     The frames' chunk is copied global -> LDS by the LDS-DMA (global_load_lds_dwordx4), raw pixel
//     type, double buffered; the B operand of a step is one ds_read of the pixel type + conversion.
//   * the block records of a wave are ONE linear stream in execution order (chunk, group, step), so
//     the wave prefetches them through an LDS ring (depth 2 ... 4: what the slabs leave of the
//     160 KiB) that runs ahead across group and chunk boundaries; L2 serves the stream (every
//     workgroup reads the same one).
//
// Complex masks are stored as 2 real masks (re, im interleaved) -- the result row is the
// interleaved complex64 row.  Results are float32 sums in a different order than the SELL kernel;
// both are within the 1e-5 relative tolerance of the float64 oracle.
#include "ltmi_common.h"
#include <vector>
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <type_traits>
#include <typeinfo>

namespace ltmi {

typedef float bf32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 bh16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 bh16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 bh16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int bu32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int bu32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *b_lds_ptr_t;
typedef const __attribute__((address_space(1))) void *b_glb_ptr_t;

#ifndef BE_P_
#define BE_P_ 512
#endif
#ifndef BE_OCC
#define BE_OCC 1
#endif
constexpr int BE_P = BE_P_;          // pixels per chunk
#ifndef BE_SEG_
#define BE_SEG_ 64
#endif
constexpr int BE_SEG = BE_SEG_;      // a chunk = BE_NSEG runs ("segments") of BE_SEG consecutive pixels
constexpr int BE_NSEG = BE_P / BE_SEG;
static_assert(BE_NSEG >= 2 && BE_NSEG <= 64 && (BE_NSEG & (BE_NSEG - 1)) == 0, "segments per chunk");
// the float16 image / k_bell_flat has its own chunk size and more frame buffers (see k_bell_flat)
#ifndef BE_FP_
#define BE_FP_ 512
#endif
#ifndef BE_FNBUF_
#define BE_FNBUF_ 2
#endif
constexpr int BE_FP = BE_FP_;
constexpr int BE_FNSEG = BE_FP / BE_SEG;
constexpr int BE_FNBUF = BE_FNBUF_;
static_assert(BE_FNSEG >= 2 && (BE_FNSEG & (BE_FNSEG - 1)) == 0 && BE_FNBUF >= 2, "float16 chunks");
#ifndef BE_SETS_
#define BE_SETS_ 16
#endif
#ifndef BE_SLOTS_
#define BE_SLOTS_ 4
#endif
constexpr int BE_SETS = BE_SETS_;    // waves per workgroup = interleaved group sets
constexpr int BE_SLOTS = BE_SLOTS_;  // groups per wave
constexpr int BE_PASS = BE_SETS * BE_SLOTS * 16;     // real masks per pass (1024)
#ifndef BE_D_
#define BE_D_ 4
#endif
#ifndef BE_PAD
#define BE_PAD 0
#endif
constexpr int BE_D_MAX = BE_D_;      // block records in flight per wave, at most (LDS permitting)
constexpr int BE_REC = 192;          // dwords per block record (64 lanes x 3)
constexpr int BE_REC16 = 256;        // ... of a float16 record (64 lanes x 4; pixel numbers in the control words)
constexpr int BE_CTL = 4;            // control words per float16 record: flags, pairs a, pairs b, -

struct BellImage {
    uint32_t *stream = nullptr;      // [blocks][64 lanes][A step 0, A step 1, pixels]
    int64_t *stream_off = nullptr;   // [n_pass * 4 + set] first record of the stream
    int *nblk = nullptr;             // [(active index * 4 + set) * 16 + slot] records of the pair
    int *active = nullptr;           // [chunk][BE_NSEG] first pixel of the chunk's segments; the chunks
                                     // of the passes are concatenated
    int *active_off = nullptr;       // [n_pass + 1] first chunk of a pass
    long crit_records = 0;           // sum over the chunks of the busiest wave's records
    bool f16 = false;                // records of 4 pixel pairs with float16 weights (k_bell_apply<.., true>)
    float *inv_scale = nullptr;      // f16: [n_cols] 1 / power-of-two scale of the column's weights
    BellImage *h16 = nullptr;        // the float16 image of the same stack (1- and 2-byte unsigned pixels)
    // f16 images: flat streams of k_bell_flat, one per (pass, wave)
    uint32_t *ctrl = nullptr;        // control word per record (slot | BE_C_SKIP | BE_C_END)
    int64_t *ctrl_off = nullptr;     // [n_pass * BE_SETS] first control word of the stream
    int *n_rec = nullptr;            // [n_pass * BE_SETS] records of the stream (a multiple of BE_FD)
    int n_pass = 0;
    // entries of the last n_px % 16 pixels of a frame (at most 15): when a row is not a multiple of 16
    // bytes its last 16-byte piece is partial and is not fetched by the frame DMA; these few entries are
    // applied by k_bell_tail after the main kernel
    int32_t *tail_px = nullptr, *tail_col = nullptr, *tail_seg = nullptr;   // sorted by column; seg: first entry of a column
    float *tail_val = nullptr;
    int n_tail = 0, n_tail_cols = 0;
    size_t n_blocks = 0;
    double mac_ratio = 0.;           // multiply-adds incl. padding / stored values
    int64_t max_col_entries = 0;     // stored values of the longest real column (length of its float32 chain)
};

template <int I, int N, typename F> __device__ __forceinline__ void bstatic_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        bstatic_for<I + 1, N>(f);
    }
}

template <typename T, int TL, int P = BE_P> struct BeCfg {
    static constexpr int SZ = (int)sizeof(T);
    static constexpr int TILES = TL;                       // 16-frame tiles per workgroup
    static constexpr int FB = 16 * TILES;
    static constexpr int ROW = P * SZ;                     // bytes of a frame's chunk
    static constexpr int RPD = ROW >= 1024 ? 1 : 1024 / ROW;   // frame rows per DMA instruction
    static constexpr int DPR = ROW >= 1024 ? ROW / 1024 : 1;   // DMA instructions per row
    static constexpr int NDMA_WG = FB * ROW / 1024;
    static constexpr int NDMA = NDMA_WG / BE_SETS;         // per wave and chunk
    // a unit = RPD rows (>= 1 KiB) + BE_PAD bytes of padding (0: XOR-swizzled pieces instead)
    static constexpr int UNIT_BYTES = (ROW >= 1024 ? ROW : 1024) + BE_PAD;
    static constexpr int BUF = (FB / RPD) * UNIT_BYTES;
    static constexpr int RING_OFF = 2 * BUF;                // record rings of the waves behind the slabs
    // ring depth: what the 160 KiB leave next to the two slabs (2 with 64-frame slabs of 1 KiB rows)
    static constexpr int D_FIT = (160 * 1024 - 2 * BUF) / (BE_SETS * 1024);
    static constexpr int D = D_FIT < BE_D_MAX ? D_FIT : BE_D_MAX;
    static constexpr int LDS_BYTES = 2 * BUF + BE_SETS * D * 1024;
    __host__ __device__ static constexpr int frame_base(int f) {
        return (f / RPD) * UNIT_BYTES + (f % RPD) * ROW;
    }
    static constexpr int TILE_OFF = (16 / RPD) * UNIT_BYTES;
};

template <typename T> __device__ __forceinline__ float be_lds_value(const unsigned char *p) {
    if constexpr (std::is_same<T, float>::value) return *(const float *)p;
    else return (float)(*(const T *)p);
}

template <typename T, int TL>
__global__ void __launch_bounds__(BE_SETS * 64, BE_OCC)
k_bell_apply(const T *__restrict__ tile, int64_t ld, int64_t n_frames, int64_t n_px,
             const uint32_t *__restrict__ stream, const int64_t *__restrict__ stream_off,
             const int *__restrict__ nblk,
             const int *__restrict__ active, const int *__restrict__ active_off,
             float *__restrict__ out, int64_t ld_out,
             int n_cols, int accumulate, int ablate, unsigned long long *prof,
             const int32_t *__restrict__ rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char be_lds[];
    using C = BeCfg<T, TL>;
    constexpr int TILES = C::TILES, NDMA = C::NDMA, BE_D = C::D;
    static_assert(BE_D >= 2, "two slabs + a record ring of depth 2 must fit the LDS");
    // one tile: the two steps of a record go to two accumulators (no back-to-back dependent MFMAs)
    constexpr int NACC = TILES == 1 ? 2 : 1;
    const int tid = threadIdx.x;
    const int j = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave = group set
    const int lane = tid & 63;
    const int m16 = lane & 15, kg = lane >> 4;
    const int pass = blockIdx.y;
    const int64_t f0 = (int64_t)blockIdx.x * C::FB;
    const int a0 = active_off[pass], a1 = active_off[pass + 1];

    bf32x4 acc[BE_SLOTS][TILES * NACC];
#pragma unroll
    for (int s = 0; s < BE_SLOTS; ++s)
#pragma unroll
        for (int t = 0; t < TILES * NACC; ++t) acc[s][t] = bf32x4{0.f, 0.f, 0.f, 0.f};
    // One or two tiles per wave: the running sums are handed to a second level after every chunk (512 pixels).  One
    // float32 chain per column over a whole frame is biased when the addends repeat -- a constant frame under the unit
    // weights of a ring's interior: once the sum passes 2^25 every addition of 32769 (65535) loses (gains) 1 -- and
    // reached 1.5e-5 (1.1e-5) of the sum for a ring of 2 500 pixels; per chunk the sum stays below 2^24 and the
    // second level adds partial sums of changing size: 8e-7 (scripts/debug_band.py history in
    // profiles/r05_banded.txt 4).  The registers are there (55 -> 89 of 128 for two tiles of 2-byte pixels); four
    // tiles have no room for it -- the launcher does not pick them for stacks with long columns
    // (BellImage::max_col_entries).  Cost: + 2 % on the 16-bin radial Fourier stack.
    constexpr bool TWO_LEVEL = TILES <= 2;
    constexpr int BE_L2 = 1;
    bf32x4 acc2[TWO_LEVEL ? BE_SLOTS : 1][TWO_LEVEL ? TILES : 1];
#pragma unroll
    for (int s = 0; s < (TWO_LEVEL ? BE_SLOTS : 1); ++s)
#pragma unroll
        for (int t = 0; t < (TWO_LEVEL ? TILES : 1); ++t) acc2[s][t] = bf32x4{0.f, 0.f, 0.f, 0.f};

    // ---- frame DMA: instruction q of the workgroup's chunk copy; wave j issues q = j*NDMA + i
    auto issue_dma = [&](int ai, int buf) {
        if (ablate == 1 || ablate == 3) return;  // timing experiments only (LTMI_BELL_ABLATE)
        // the chunk's BE_NSEG segments (runs of BE_SEG pixels anywhere in the frame, chosen by the
        // image builder): lanes 0 .. BE_NSEG-1 hold their first pixel, the others fetch it by bpermute
        const int segv = active[(int64_t)ai * BE_NSEG + (lane & (BE_NSEG - 1))];
        constexpr int PPS = BE_SEG * C::SZ / 16;      // 16-byte pieces per segment
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int q = j * NDMA + i;
            int64_t fr;
            int piece;                            // 16-byte piece of the frame's chunk this lane fetches
            int dst;
            if constexpr (C::RPD > 1) {          // several rows per instruction (1-byte pixels)
                constexpr int LPR = 64 / C::RPD;  // lanes per row
                const int r = q * C::RPD + lane / LPR;             // frame of the workgroup
                fr = f0 + r;
                piece = (lane % LPR) ^ (BE_PAD ? 0 : (r & 7));
                dst = q * C::UNIT_BYTES;
            } else {
                const int r = q / C::DPR;
                fr = f0 + r;
                piece = (q % C::DPR) * 64 + (lane ^ (BE_PAD ? 0 : (r & 7)));
                dst = r * C::UNIT_BYTES + (q % C::DPR) * 1024;
            }
            const int seg_px = __builtin_amdgcn_ds_bpermute((piece / PPS) * 4, segv);
            int64_t byte_in_row = (int64_t)seg_px * C::SZ + (piece % PPS) * 16;
            if (fr > n_frames - 1) fr = n_frames - 1;
            if (rows) fr = rows[fr];                  // a region of interest: result row i = frame rows[i]
            // the last chunk may be partial: pieces past the row are not referenced by any record
            if (byte_in_row + 16 > n_px * C::SZ) byte_in_row = 0;
            const unsigned char *src = (const unsigned char *)(tile + fr * ld) + byte_in_row;
            __builtin_amdgcn_global_load_lds((b_glb_ptr_t)src,
                                             (b_lds_ptr_t)(be_lds + buf * C::BUF + dst), 16, 0, 0);
        }
    };

    // ---- record ring: BE_D slots per wave in LDS, filled by LDS-DMA (dwordx3: 12 bytes per lane),
    // read back with ds_read when the record's turn comes.  All loads of the kernel are
    // DMA and retire in order, so the waits are counted by hand; what lands in REGISTERS is ordinary
    // compiler-tracked LDS data (an earlier version kept the ring in registers behind asm loads: the
    // compiler may copy such registers while they are still in flight).
    // (global_load_lds_dwordx3 puts the 12 bytes of lane l at l * 16: a slot is 1 KiB,
    // probes/dma3_probe.hip)
    constexpr int REC_BYTES = BE_REC * 4, SLOT_BYTES = 1024;
    unsigned char *ring_lds = be_lds + C::RING_OFF + j * (BE_D * SLOT_BYTES);
    const unsigned char *sp = (const unsigned char *)(stream + stream_off[pass * BE_SETS + j] * BE_REC)
                              + lane * 12;
    auto issue_rec = [&](int slot) {
        __builtin_amdgcn_global_load_lds((b_glb_ptr_t)sp, (b_lds_ptr_t)(ring_lds + slot * SLOT_BYTES),
                                         12, 0, 0);
        sp += REC_BYTES;
    };
#pragma unroll
    for (int u = 0; u < BE_D; ++u) issue_rec(u);
    int slot = 0;                       // ring slot holding the next record

    const int lane_base = C::frame_base(m16);
    // BE_PAD == 0: instead of padding the rows, the 16-byte pieces of frame f sit at piece ^ (f & 7)
    // (the DMA lanes fetch the permuted source piece) -- the 16 frames of a tile spread over the
    // banks like with 16 bytes of padding, and the LDS holds a fourth ring slot per wave instead
    const unsigned swz = BE_PAD ? 0u : (unsigned)((m16 & 7) << 4);
    int since_dma = BE_D;               // records consumed since the last frame-DMA issue (saturating)

    if (a0 < a1) issue_dma(a0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // (not __syncthreads: its fence would drain the record ring)
    asm volatile("" ::: "memory");

#ifdef BE_PROF
    // cycle attribution (-DBE_PROF builds only, scripts/bell_variants.sh): s_memtime stamps around
    // the phases of a record; every stamp costs ~60 cycles itself
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, t_prev = __builtin_readcyclecounter();
    const unsigned long long t_begin = t_prev;
#define BE_STAMP(cat)                                                  \
    do {                                                               \
        const unsigned long long t_now = __builtin_readcyclecounter(); \
        pt[cat] += t_now - t_prev;                                     \
        t_prev = t_now;                                                \
    } while (0)
#else
#define BE_STAMP(cat) do { } while (0)
#endif

    for (int ai = a0; ai < a1; ++ai) {
        const int buf = (ai - a0) & 1;
        if (ai + 1 < a1) {
            issue_dma(ai + 1, buf ^ 1);
            if (ablate != 1 && ablate != 3) since_dma = 0;
        }
        const unsigned char *bbase = be_lds + buf * C::BUF + lane_base;
        const int *nb_row = nblk + ((int64_t)ai * BE_SETS + j) * BE_SLOTS;
        int nbs[BE_SLOTS];                            // one 64-byte scalar load
#pragma unroll
        for (int s = 0; s < BE_SLOTS; ++s) nbs[s] = nb_row[s];

        bstatic_for<0, BE_SLOTS>([&](auto S) {
            constexpr int s = decltype(S)::value;
            const int nb = ablate == 2 ? 0 : nbs[s];
            for (int b = 0; b < nb; ++b) {
                BE_STAMP(3);                           // loop control since the last record
                // the oldest record of the ring: BE_D - 1 younger ones stay in flight.  (Right after
                // a frame-DMA issue this also waits for most of that DMA -- record loads retire
                // behind it anyway, the stall would come BE_D records later.)
#ifndef BE_OLD_WAIT
                // ... unless the frame DMA of the next chunk was issued less than BE_D records ago: then
                // the record is OLDER than that DMA (loads retire in order) and the NDMA copies may stay
                // in flight too -- waiting them out here would stall every wave for an HBM round trip
                // per chunk
                if (since_dma < BE_D) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BE_D - 1 + NDMA) : "memory");
                else
#endif
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BE_D - 1) : "memory");
                BE_STAMP(0);
                const unsigned *rp = (const unsigned *)(ring_lds + slot * SLOT_BYTES) + lane * 4;
#ifdef BE_X_NORING      // timing experiments (wrong results): no ring read
                const unsigned x0 = 0x3f800000u + lane + b, x1 = 0x3f000000u + lane, o = ((lane >> 4) * 9 + b) | (((lane >> 4) * 5 + 300) << 16);
#else
                const unsigned x0 = rp[0], x1 = rp[1], o = rp[2];
#endif
                const float a_0 = __uint_as_float(x0);
                const float a_1 = __uint_as_float(x1);
                const unsigned char *p0 = bbase + (((o & 0xffffu) * C::SZ) ^ swz);
                const unsigned char *p1 = bbase + (((o >> 16) * C::SZ) ^ swz);
#ifdef BE_EXP_READY
                // timing experiment (wrong results): the record's words taken as ready-made byte offsets
                p0 = bbase + (o & 0x3feu);
                p1 = bbase + ((o >> 16) & 0x3feu);
#endif
                T b0[TILES], b1[TILES];
#pragma unroll
                for (int t = 0; t < TILES; ++t) {
#ifdef BE_X_GATHER1     // timing experiment (wrong results): two gathers per record instead of 2 * TILES
                    if (t > 0) { b0[t] = b0[0]; b1[t] = b1[0]; continue; }
#endif
                    b0[t] = *(const T *)(p0 + t * C::TILE_OFF);
                    b1[t] = *(const T *)(p1 + t * C::TILE_OFF);
                }
#ifndef BE_NO_GATHER_GROUP
                // all gathers of the record in flight together: left alone the compiler reuses one
                // register and issues the last gather behind the first MFMAs -- a third dependent LDS
                // round trip per record
                {
                    float g0f[TILES], g1f[TILES];       // (the conversions follow the gathers anyway)
#pragma unroll
                    for (int t = 0; t < TILES; ++t) { g0f[t] = (float)b0[t]; g1f[t] = (float)b1[t]; }
                    if constexpr (TILES == 4)
                        asm volatile("" ::"v"(g0f[0]), "v"(g0f[1]), "v"(g0f[2]), "v"(g0f[3]),
                                     "v"(g1f[0]), "v"(g1f[1]), "v"(g1f[2]), "v"(g1f[3]));
                    else if constexpr (TILES == 2)
                        asm volatile("" ::"v"(g0f[0]), "v"(g0f[1]), "v"(g1f[0]), "v"(g1f[1]));
                    else
                        asm volatile("" ::"v"(g0f[0]), "v"(g1f[0]));
                }
#endif
#ifdef BE_PROF
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                BE_STAMP(1);
#endif
#ifndef BE_NOMFMA
#pragma unroll
                for (int t = 0; t < TILES; ++t)
                    acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_0, (float)b0[t], acc[s][t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < TILES; ++t)
                    acc[s][t + (NACC - 1)] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        a_1, (float)b1[t], acc[s][t + (NACC - 1)], 0, 0, 0);
#else               // timing experiment: the same operands through one VALU op each
#pragma unroll
                for (int t = 0; t < TILES; ++t) {
                    acc[s][t][0] += a_0 * (float)b0[t];
                    acc[s][t][1] += a_1 * (float)b1[t];
                }
#endif
                // refill the slot (its three words are in registers: the MFMAs above used them)
                issue_rec(slot);
                slot = slot + 1 == BE_D ? 0 : slot + 1;
                since_dma = since_dma < BE_D ? since_dma + 1 : since_dma;
                BE_STAMP(2);
            }
        });
        if constexpr (TWO_LEVEL) {
            if (((ai - a0) & (BE_L2 - 1)) == BE_L2 - 1) {
#pragma unroll
                for (int s = 0; s < BE_SLOTS; ++s)
#pragma unroll
                    for (int t = 0; t < TILES; ++t) {
                        acc2[s][t] += acc[s][t];
                        acc[s][t] = bf32x4{0.f, 0.f, 0.f, 0.f};
                        if constexpr (NACC == 2) {
                            acc2[s][t] += acc[s][t + 1];
                            acc[s][t + 1] = bf32x4{0.f, 0.f, 0.f, 0.f};
                        }
                    }
            }
        }
        // The next chunk must have landed before anyone reads it.  BE_D records consumed since its
        // issue imply it: their refills were issued after it and at most BE_D loads are in flight.
        BE_STAMP(3);
        if (since_dma < BE_D) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BE_D) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        BE_STAMP(4);                                   // waiting for the next chunk's DMA
        if (ablate != 3) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        BE_STAMP(5);                                   // waiting for the other waves
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the run-ahead record loads
#ifdef BE_PROF
    if (prof && lane == 0) {
#pragma unroll
        for (int c = 0; c < 6; ++c) atomicAdd(prof + c, pt[c]);
        atomicAdd(prof + 6, __builtin_readcyclecounter() - t_begin);
        atomicAdd(prof + 7, 1ull);
    }
#endif

    // ---- results: lane holds columns g*16 + kg*4 .. +3 of frame (tile t, m16)
    if constexpr (NACC == 2) {
#pragma unroll
        for (int s = 0; s < BE_SLOTS; ++s) acc[s][0] += acc[s][1];
    }
    if constexpr (TWO_LEVEL) {
#pragma unroll
        for (int s = 0; s < BE_SLOTS; ++s)
#pragma unroll
            for (int t = 0; t < TILES; ++t) acc[s][t] += acc2[s][t];
    }
#pragma unroll
    for (int s = 0; s < BE_SLOTS; ++s) {
        const int col0 = pass * BE_PASS + (s * BE_SETS + j) * 16 + kg * 4;
        if (col0 >= n_cols) continue;
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            const int64_t f = f0 + t * 16 + m16;
            if (f >= n_frames) continue;
            float *o = out + f * ld_out + col0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (col0 + r < n_cols) o[r] = accumulate ? o[r] + acc[s][t][r] : acc[s][t][r];
        }
    }
}


// ==== float16 path, flat record loop (1- and 2-byte integer pixels): k_bell_flat ===========================
// What the float32 kernel above pays for is the matrix pipe: 3.8 padded multiply-adds per stored value on
// v_mfma_f32_16x16x4_f32 are 0.4 - 0.5 ms per 16 384 frames of C4 at the clock the chip sustains.  For
// unsigned 1- and 2-byte pixels the same products can be formed EXACTLY from float16 operands:
//   * a pixel x = lo + 256 hi, two bytes; 0x6400 | b is the float16 number 1024 + b, so one v_perm_b32 turns two
//     bytes into two float16 values and one v_pk_add_f16 removes the bias (no conversion instruction).  (Bytes as
//     float16 subnormals need no arithmetic at all and the matrix cores accept them, but their accumulation
//     truncates: measured -3e-6 relative bias, not used -- profiles/r03_sparse.txt item 9);
//   * a weight times its column's power-of-two scale S is w1 + w2, two float16 (22 bits); S keeps 256 w S below
//     the float16 maximum and small weights out of its subnormals, the kernel multiplies by 1 / S at the end;
//   * w x S = w1 lo + (256 w1) hi + w2 lo + (256 w2) hi: four exact products per pixel, summed in float32 by
//     v_mfma_f32_16x16x32_f16.  A record = 8 aligned pixel PAIRS x 16 masks; lane group kg holds the pairs qa, qb:
//     K slots [lo(qa) lo(qa+1) hi(qa) hi(qa+1) | the same of qb] against [w(qa) w(qa+1) 256 w(qa) 256 w(qa+1) | ..],
//     one MFMA for w1 and one for w2 per tile -- 2 x 16 pipe cycles for 16 pixels where float32 needs 4 x 32.
// The loop is FLAT: one stream of records per wave for the whole sweep, unrolled BE_FD times, with control words
// per record -- which of the wave's 4 accumulator sets it feeds, whether the wave's work on the chunk ends with
// it (then: wait for the next chunk's frames, barrier, start the copy of the chunk after next), the pairs' pixel
// numbers -- read through the scalar cache.  The record ring (BE_FD records of 4 words per lane) and the 64
// accumulator registers live in AGPRs under fixed names: registers cannot be indexed by a ring position (hence
// the unrolling, ring position = position in the unrolled body), compiler-visible accumulators are copied at
// every merge of the four accumulator-set arms, and a ring behind asm loads must not be moved by the compiler
// while a load is in flight.  What this buys and what still bounds the kernel: profiles/r03_sparse.txt, DESIGN 4.3.
#ifndef BE_DMA_AUX
#define BE_DMA_AUX 2                // cache-policy bits of the frame copies of k_bell_flat: nt (0.677 -> 0.655 ms on C4, copies alone 0.46 -> 0.41)
#endif
#ifndef BE_PHASE_SLEEP
#define BE_PHASE_SLEEP 8            // s_sleep units (64 cycles) per phase step, 16 steps
#endif
#ifndef BE_FD_
#define BE_FD_ 6
#endif
constexpr int BE_FD = BE_FD_;
constexpr unsigned BE_C_SKIP = 4u, BE_C_END = 8u;

#define BE_AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63"
#define BE_ACC_ZERO() asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\tv_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\tv_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\tv_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0\n\tv_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\tv_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\tv_accvgpr_write_b32 a24, 0\n\tv_accvgpr_write_b32 a25, 0\n\tv_accvgpr_write_b32 a26, 0\n\tv_accvgpr_write_b32 a27, 0\n\tv_accvgpr_write_b32 a28, 0\n\tv_accvgpr_write_b32 a29, 0\n\tv_accvgpr_write_b32 a30, 0\n\tv_accvgpr_write_b32 a31, 0\n\tv_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a33, 0\n\tv_accvgpr_write_b32 a34, 0\n\tv_accvgpr_write_b32 a35, 0\n\tv_accvgpr_write_b32 a36, 0\n\tv_accvgpr_write_b32 a37, 0\n\tv_accvgpr_write_b32 a38, 0\n\tv_accvgpr_write_b32 a39, 0\n\tv_accvgpr_write_b32 a40, 0\n\tv_accvgpr_write_b32 a41, 0\n\tv_accvgpr_write_b32 a42, 0\n\tv_accvgpr_write_b32 a43, 0\n\tv_accvgpr_write_b32 a44, 0\n\tv_accvgpr_write_b32 a45, 0\n\tv_accvgpr_write_b32 a46, 0\n\tv_accvgpr_write_b32 a47, 0\n\tv_accvgpr_write_b32 a48, 0\n\tv_accvgpr_write_b32 a49, 0\n\tv_accvgpr_write_b32 a50, 0\n\tv_accvgpr_write_b32 a51, 0\n\tv_accvgpr_write_b32 a52, 0\n\tv_accvgpr_write_b32 a53, 0\n\tv_accvgpr_write_b32 a54, 0\n\tv_accvgpr_write_b32 a55, 0\n\tv_accvgpr_write_b32 a56, 0\n\tv_accvgpr_write_b32 a57, 0\n\tv_accvgpr_write_b32 a58, 0\n\tv_accvgpr_write_b32 a59, 0\n\tv_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0" ::: BE_AGPRS)
#define BE_MFMA_0_0(A_, B_) asm volatile("v_mfma_f32_16x16x32_f16 a[0:3], %0, %1, a[0:3]" ::"v"(A_), "v"(B_) : BE_AGPRS)
#define BE_MFMA_0_1(A_, B_) asm volatile("v_mfma_f32_16x16x32_f16 a[4:7], %0, %1, a[4:7]" ::"v"(A_), "v"(B_) : BE_AGPRS)
#define BE_MFMA_0_2(A_, B_) asm volatile("v_mfma_f32_16x16x32_f16 a[8:11], %0, %1, a[8:11]" ::"v"(A_), "v"(B_) : BE_AGPRS)
#define BE_MFMA_0_3(A_, B_) asm volatile("v_mfma_f32_16x16x32_f16 a[12:15], %0, %1, a[12:15]" ::"v"(A_), "v"(B_) : BE_AGPRS)
#define BE_MFMA_1_0(A_, B_) asm volatile("v_mfma_f32_16x16x32_f16 a[16:19], %0, %1, a[16:19]" ::"v"(A_), "v"(B_) : BE_AGPRS)
#define BE_MFMA_1_1(A_, B_) asm volatile("v_mfma_f32_16x16x32_f16 a[20:23], %0, %1, a[20:23]" ::"v"(A_), "v"(B_) : BE_AGPRS)
#define BE_MFMA_1_2(A_, B_) asm volatile("v_mfma_f32_16x16x32_f16 a[24:27], %0, %1, a[24:27]" ::"v"(A_), "v"(B_) : BE_AGPRS)
#define BE_MFMA_1_3(A_, B_) asm volatile("v_mfma_f32_16x16x32_f16 a[28:31], %0, %1, a[28:31]" ::"v"(A_), "v"(B_) : BE_AGPRS)
#define BE_MFMA_2_0(A_, B_) asm volatile("v_mfma_f32_16x16x32_f16 a[32:35], %0, %1, a[32:35]" ::"v"(A_), "v"(B_) : BE_AGPRS)
#define BE_MFMA_2_1(A_, B_) asm volatile("v_mfma_f32_16x16x32_f16 a[36:39], %0, %1, a[36:39]" ::"v"(A_), "v"(B_) : BE_AGPRS)
#define BE_MFMA_2_2(A_, B_) asm volatile("v_mfma_f32_16x16x32_f16 a[40:43], %0, %1, a[40:43]" ::"v"(A_), "v"(B_) : BE_AGPRS)
#define BE_MFMA_2_3(A_, B_) asm volatile("v_mfma_f32_16x16x32_f16 a[44:47], %0, %1, a[44:47]" ::"v"(A_), "v"(B_) : BE_AGPRS)
#define BE_MFMA_3_0(A_, B_) asm volatile("v_mfma_f32_16x16x32_f16 a[48:51], %0, %1, a[48:51]" ::"v"(A_), "v"(B_) : BE_AGPRS)
#define BE_MFMA_3_1(A_, B_) asm volatile("v_mfma_f32_16x16x32_f16 a[52:55], %0, %1, a[52:55]" ::"v"(A_), "v"(B_) : BE_AGPRS)
#define BE_MFMA_3_2(A_, B_) asm volatile("v_mfma_f32_16x16x32_f16 a[56:59], %0, %1, a[56:59]" ::"v"(A_), "v"(B_) : BE_AGPRS)
#define BE_MFMA_3_3(A_, B_) asm volatile("v_mfma_f32_16x16x32_f16 a[60:63], %0, %1, a[60:63]" ::"v"(A_), "v"(B_) : BE_AGPRS)
#define BE_ARM(S_)                                                         \
    asm volatile("s_nop 1");                                               \
    BE_MFMA_##S_##_0(a1v, bv[0]); BE_MFMA_##S_##_1(a1v, bv[1]);              \
    if constexpr (TILES == 4) { BE_MFMA_##S_##_2(a1v, bv[2]); BE_MFMA_##S_##_3(a1v, bv[3]); } \
    BE_MFMA_##S_##_0(a2v, bv[0]); BE_MFMA_##S_##_1(a2v, bv[1]);              \
    if constexpr (TILES == 4) { BE_MFMA_##S_##_2(a2v, bv[2]); BE_MFMA_##S_##_3(a2v, bv[3]); }

#define BE_ACC_READ_0_0(V0, V1, V2, V3) asm volatile("v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3" : "=v"(V0), "=v"(V1), "=v"(V2), "=v"(V3) :: BE_AGPRS)
#define BE_ACC_READ_0_1(V0, V1, V2, V3) asm volatile("v_accvgpr_read_b32 %0, a4\n\tv_accvgpr_read_b32 %1, a5\n\tv_accvgpr_read_b32 %2, a6\n\tv_accvgpr_read_b32 %3, a7" : "=v"(V0), "=v"(V1), "=v"(V2), "=v"(V3) :: BE_AGPRS)
#define BE_ACC_READ_0_2(V0, V1, V2, V3) asm volatile("v_accvgpr_read_b32 %0, a8\n\tv_accvgpr_read_b32 %1, a9\n\tv_accvgpr_read_b32 %2, a10\n\tv_accvgpr_read_b32 %3, a11" : "=v"(V0), "=v"(V1), "=v"(V2), "=v"(V3) :: BE_AGPRS)
#define BE_ACC_READ_0_3(V0, V1, V2, V3) asm volatile("v_accvgpr_read_b32 %0, a12\n\tv_accvgpr_read_b32 %1, a13\n\tv_accvgpr_read_b32 %2, a14\n\tv_accvgpr_read_b32 %3, a15" : "=v"(V0), "=v"(V1), "=v"(V2), "=v"(V3) :: BE_AGPRS)
#define BE_ACC_READ_1_0(V0, V1, V2, V3) asm volatile("v_accvgpr_read_b32 %0, a16\n\tv_accvgpr_read_b32 %1, a17\n\tv_accvgpr_read_b32 %2, a18\n\tv_accvgpr_read_b32 %3, a19" : "=v"(V0), "=v"(V1), "=v"(V2), "=v"(V3) :: BE_AGPRS)
#define BE_ACC_READ_1_1(V0, V1, V2, V3) asm volatile("v_accvgpr_read_b32 %0, a20\n\tv_accvgpr_read_b32 %1, a21\n\tv_accvgpr_read_b32 %2, a22\n\tv_accvgpr_read_b32 %3, a23" : "=v"(V0), "=v"(V1), "=v"(V2), "=v"(V3) :: BE_AGPRS)
#define BE_ACC_READ_1_2(V0, V1, V2, V3) asm volatile("v_accvgpr_read_b32 %0, a24\n\tv_accvgpr_read_b32 %1, a25\n\tv_accvgpr_read_b32 %2, a26\n\tv_accvgpr_read_b32 %3, a27" : "=v"(V0), "=v"(V1), "=v"(V2), "=v"(V3) :: BE_AGPRS)
#define BE_ACC_READ_1_3(V0, V1, V2, V3) asm volatile("v_accvgpr_read_b32 %0, a28\n\tv_accvgpr_read_b32 %1, a29\n\tv_accvgpr_read_b32 %2, a30\n\tv_accvgpr_read_b32 %3, a31" : "=v"(V0), "=v"(V1), "=v"(V2), "=v"(V3) :: BE_AGPRS)
#define BE_ACC_READ_2_0(V0, V1, V2, V3) asm volatile("v_accvgpr_read_b32 %0, a32\n\tv_accvgpr_read_b32 %1, a33\n\tv_accvgpr_read_b32 %2, a34\n\tv_accvgpr_read_b32 %3, a35" : "=v"(V0), "=v"(V1), "=v"(V2), "=v"(V3) :: BE_AGPRS)
#define BE_ACC_READ_2_1(V0, V1, V2, V3) asm volatile("v_accvgpr_read_b32 %0, a36\n\tv_accvgpr_read_b32 %1, a37\n\tv_accvgpr_read_b32 %2, a38\n\tv_accvgpr_read_b32 %3, a39" : "=v"(V0), "=v"(V1), "=v"(V2), "=v"(V3) :: BE_AGPRS)
#define BE_ACC_READ_2_2(V0, V1, V2, V3) asm volatile("v_accvgpr_read_b32 %0, a40\n\tv_accvgpr_read_b32 %1, a41\n\tv_accvgpr_read_b32 %2, a42\n\tv_accvgpr_read_b32 %3, a43" : "=v"(V0), "=v"(V1), "=v"(V2), "=v"(V3) :: BE_AGPRS)
#define BE_ACC_READ_2_3(V0, V1, V2, V3) asm volatile("v_accvgpr_read_b32 %0, a44\n\tv_accvgpr_read_b32 %1, a45\n\tv_accvgpr_read_b32 %2, a46\n\tv_accvgpr_read_b32 %3, a47" : "=v"(V0), "=v"(V1), "=v"(V2), "=v"(V3) :: BE_AGPRS)
#define BE_ACC_READ_3_0(V0, V1, V2, V3) asm volatile("v_accvgpr_read_b32 %0, a48\n\tv_accvgpr_read_b32 %1, a49\n\tv_accvgpr_read_b32 %2, a50\n\tv_accvgpr_read_b32 %3, a51" : "=v"(V0), "=v"(V1), "=v"(V2), "=v"(V3) :: BE_AGPRS)
#define BE_ACC_READ_3_1(V0, V1, V2, V3) asm volatile("v_accvgpr_read_b32 %0, a52\n\tv_accvgpr_read_b32 %1, a53\n\tv_accvgpr_read_b32 %2, a54\n\tv_accvgpr_read_b32 %3, a55" : "=v"(V0), "=v"(V1), "=v"(V2), "=v"(V3) :: BE_AGPRS)
#define BE_ACC_READ_3_2(V0, V1, V2, V3) asm volatile("v_accvgpr_read_b32 %0, a56\n\tv_accvgpr_read_b32 %1, a57\n\tv_accvgpr_read_b32 %2, a58\n\tv_accvgpr_read_b32 %3, a59" : "=v"(V0), "=v"(V1), "=v"(V2), "=v"(V3) :: BE_AGPRS)
#define BE_ACC_READ_3_3(V0, V1, V2, V3) asm volatile("v_accvgpr_read_b32 %0, a60\n\tv_accvgpr_read_b32 %1, a61\n\tv_accvgpr_read_b32 %2, a62\n\tv_accvgpr_read_b32 %3, a63" : "=v"(V0), "=v"(V1), "=v"(V2), "=v"(V3) :: BE_AGPRS)
#define BE_ACC_STORE(S_)                                                   \
    if (TILES > 0) {                                                    \
        float v0_, v1_, v2_, v3_;                                         \
        BE_ACC_READ_##S_##_0(v0_, v1_, v2_, v3_);                            \
        store_tile(S_, 0, v0_, v1_, v2_, v3_);                           \
    }                                                                  \
    if (TILES > 1) {                                                    \
        float v0_, v1_, v2_, v3_;                                         \
        BE_ACC_READ_##S_##_1(v0_, v1_, v2_, v3_);                            \
        store_tile(S_, 1, v0_, v1_, v2_, v3_);                           \
    }                                                                  \
    if (TILES > 2) {                                                    \
        float v0_, v1_, v2_, v3_;                                         \
        BE_ACC_READ_##S_##_2(v0_, v1_, v2_, v3_);                            \
        store_tile(S_, 2, v0_, v1_, v2_, v3_);                           \
    }                                                                  \
    if (TILES > 3) {                                                    \
        float v0_, v1_, v2_, v3_;                                         \
        BE_ACC_READ_##S_##_3(v0_, v1_, v2_, v3_);                            \
        store_tile(S_, 3, v0_, v1_, v2_, v3_);                           \
    }


// the record ring of k_bell_flat: ring position u = a[64 + 4 u .. + 3], loaded by asm (the compiler neither
// sees the loads nor the registers, so it cannot move a register whose load is still in flight) and
// waited for with hand-counted vmcnt
#define BE_RING_LOAD_0(VOFF, SBASE) asm volatile("global_load_dwordx4 a[64:67], %0, %1" ::"v"(VOFF), "s"(SBASE) : "memory", BE_RING_AGPRS)
#define BE_RING_READ_0(W) asm volatile("v_accvgpr_read_b32 %0, a64\n\tv_accvgpr_read_b32 %1, a65\n\tv_accvgpr_read_b32 %2, a66\n\tv_accvgpr_read_b32 %3, a67" : "=v"(W[0]), "=v"(W[1]), "=v"(W[2]), "=v"(W[3]) :: BE_RING_AGPRS)
#define BE_RING_LOAD_1(VOFF, SBASE) asm volatile("global_load_dwordx4 a[68:71], %0, %1" ::"v"(VOFF), "s"(SBASE) : "memory", BE_RING_AGPRS)
#define BE_RING_READ_1(W) asm volatile("v_accvgpr_read_b32 %0, a68\n\tv_accvgpr_read_b32 %1, a69\n\tv_accvgpr_read_b32 %2, a70\n\tv_accvgpr_read_b32 %3, a71" : "=v"(W[0]), "=v"(W[1]), "=v"(W[2]), "=v"(W[3]) :: BE_RING_AGPRS)
#define BE_RING_LOAD_2(VOFF, SBASE) asm volatile("global_load_dwordx4 a[72:75], %0, %1" ::"v"(VOFF), "s"(SBASE) : "memory", BE_RING_AGPRS)
#define BE_RING_READ_2(W) asm volatile("v_accvgpr_read_b32 %0, a72\n\tv_accvgpr_read_b32 %1, a73\n\tv_accvgpr_read_b32 %2, a74\n\tv_accvgpr_read_b32 %3, a75" : "=v"(W[0]), "=v"(W[1]), "=v"(W[2]), "=v"(W[3]) :: BE_RING_AGPRS)
#define BE_RING_LOAD_3(VOFF, SBASE) asm volatile("global_load_dwordx4 a[76:79], %0, %1" ::"v"(VOFF), "s"(SBASE) : "memory", BE_RING_AGPRS)
#define BE_RING_READ_3(W) asm volatile("v_accvgpr_read_b32 %0, a76\n\tv_accvgpr_read_b32 %1, a77\n\tv_accvgpr_read_b32 %2, a78\n\tv_accvgpr_read_b32 %3, a79" : "=v"(W[0]), "=v"(W[1]), "=v"(W[2]), "=v"(W[3]) :: BE_RING_AGPRS)
#define BE_RING_LOAD_4(VOFF, SBASE) asm volatile("global_load_dwordx4 a[80:83], %0, %1" ::"v"(VOFF), "s"(SBASE) : "memory", BE_RING_AGPRS)
#define BE_RING_READ_4(W) asm volatile("v_accvgpr_read_b32 %0, a80\n\tv_accvgpr_read_b32 %1, a81\n\tv_accvgpr_read_b32 %2, a82\n\tv_accvgpr_read_b32 %3, a83" : "=v"(W[0]), "=v"(W[1]), "=v"(W[2]), "=v"(W[3]) :: BE_RING_AGPRS)
#define BE_RING_LOAD_5(VOFF, SBASE) asm volatile("global_load_dwordx4 a[84:87], %0, %1" ::"v"(VOFF), "s"(SBASE) : "memory", BE_RING_AGPRS)
#define BE_RING_READ_5(W) asm volatile("v_accvgpr_read_b32 %0, a84\n\tv_accvgpr_read_b32 %1, a85\n\tv_accvgpr_read_b32 %2, a86\n\tv_accvgpr_read_b32 %3, a87" : "=v"(W[0]), "=v"(W[1]), "=v"(W[2]), "=v"(W[3]) :: BE_RING_AGPRS)
#define BE_RING_LOAD_6(VOFF, SBASE) asm volatile("global_load_dwordx4 a[88:91], %0, %1" ::"v"(VOFF), "s"(SBASE) : "memory", BE_RING_AGPRS)
#define BE_RING_READ_6(W) asm volatile("v_accvgpr_read_b32 %0, a88\n\tv_accvgpr_read_b32 %1, a89\n\tv_accvgpr_read_b32 %2, a90\n\tv_accvgpr_read_b32 %3, a91" : "=v"(W[0]), "=v"(W[1]), "=v"(W[2]), "=v"(W[3]) :: BE_RING_AGPRS)
#define BE_RING_LOAD_7(VOFF, SBASE) asm volatile("global_load_dwordx4 a[92:95], %0, %1" ::"v"(VOFF), "s"(SBASE) : "memory", BE_RING_AGPRS)
#define BE_RING_READ_7(W) asm volatile("v_accvgpr_read_b32 %0, a92\n\tv_accvgpr_read_b32 %1, a93\n\tv_accvgpr_read_b32 %2, a94\n\tv_accvgpr_read_b32 %3, a95" : "=v"(W[0]), "=v"(W[1]), "=v"(W[2]), "=v"(W[3]) :: BE_RING_AGPRS)
#if BE_FD_ <= 6
#define BE_RING_AGPRS "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87"
#else
#define BE_RING_AGPRS "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95"
#endif

template <typename T, int TL>
// (a[0 .. 63 + 4 BE_FD] are named by hand and not in the compiler's budget)
#define BE_FLAT_NVGPR (128 - 64 - 4 * (BE_FD_ <= 6 ? 6 : 8))
__global__ void __launch_bounds__(BE_SETS * 64, 1) __attribute__((amdgpu_num_vgpr(BE_FLAT_NVGPR)))
k_bell_flat(const T *__restrict__ tile, int64_t ld, int64_t n_frames, int64_t n_px,
            const uint32_t *__restrict__ stream, const int64_t *__restrict__ stream_off,
            const uint32_t *__restrict__ ctrl, const int64_t *__restrict__ ctrl_off,
            const int *__restrict__ n_rec,
            const int *__restrict__ active, const int *__restrict__ active_off,
            float *__restrict__ out, int64_t ld_out, int n_cols, int accumulate, int ablate,
            const int32_t *__restrict__ rows, const float *__restrict__ inv_scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char be_lds[];
    using C = BeCfg<T, TL, BE_FP>;
    constexpr int NBUF = BE_FNBUF, DIST = NBUF >= 3 ? NBUF - 1 : 1;
    static_assert(C::SZ <= 2, "float16 path: 1- and 2-byte pixels");
    constexpr int TILES = C::TILES, NDMA = C::NDMA;
    const int tid = threadIdx.x;
    const int j = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int m16 = lane & 15, kg = lane >> 4;
    const int pass = blockIdx.y;
    const int64_t f0 = (int64_t)blockIdx.x * C::FB;
    const int a0 = active_off[pass], a1 = active_off[pass + 1];

    // The 4 x TILES accumulator tiles live in a[0:63] under fixed names (slot s, tile t: a[16 s + 4 t ..
    // + 3]), outside the compiler's view: which set a record feeds is decided at run time, and through
    // compiler-visible accumulators every merge of the four arms copies whole register tuples (and
    // spills).  Every asm statement that touches them lists all 64 as clobbered, so the compiler
    // keeps nothing of its own there.
    BE_ACC_ZERO();

    auto issue_dma = [&](int ai, int buf, int i_lo = 0, int i_hi = 64) {
        if (ablate == 1 || ablate == 3) return;
        const int segv = active[(int64_t)ai * BE_FNSEG + (lane & (BE_FNSEG - 1))];
        constexpr int PPS = BE_SEG * C::SZ / 16;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            if (i < i_lo || i >= i_hi) continue;
            const int q = j * NDMA + i;
            int64_t fr;
            int piece, dst;
            if constexpr (C::RPD > 1) {
                constexpr int LPR = 64 / C::RPD;
                const int r = q * C::RPD + lane / LPR;
                fr = f0 + r;
                piece = (lane % LPR) ^ (r & 7);
                dst = q * C::UNIT_BYTES;
            } else {
                const int r = q / C::DPR;
                fr = f0 + r;
                piece = (q % C::DPR) * 64 + (lane ^ (r & 7));
                dst = r * C::UNIT_BYTES + (q % C::DPR) * 1024;
            }
            const int seg_px = __builtin_amdgcn_ds_bpermute((piece / PPS) * 4, segv);
            int64_t byte_in_row = (int64_t)seg_px * C::SZ + (piece % PPS) * 16;
            if (fr > n_frames - 1) fr = n_frames - 1;
            if (rows) fr = rows[fr];
            if (byte_in_row + 16 > n_px * C::SZ) byte_in_row = 0;
            const unsigned char *src = (const unsigned char *)(tile + fr * ld) + byte_in_row;
            __builtin_amdgcn_global_load_lds((b_glb_ptr_t)src,
                                             (b_lds_ptr_t)(be_lds + buf * C::BUF + dst), 16, 0, BE_DMA_AUX);
        }
    };

    const int wj = pass * BE_SETS + j;
    const unsigned char *rec_base = (const unsigned char *)(stream + stream_off[wj] * BE_REC16);   // uniform
    const unsigned lane12 = (unsigned)lane * 16u;          // a lane's 16 bytes of a record
    const uint32_t *ctl = ctrl + ctrl_off[wj];
    const int n = n_rec[wj];                       // a multiple of BE_FD; BE_FD more records follow
    constexpr int REC_BYTES = BE_REC16 * 4;

    auto ring_load = [&](auto U, int64_t rec) {
        constexpr int u = decltype(U)::value;
        const uint64_t sb = (uint64_t)(rec_base + rec * REC_BYTES);
        if constexpr (u == 0) BE_RING_LOAD_0(lane12, sb);
        else if constexpr (u == 1) BE_RING_LOAD_1(lane12, sb);
        else if constexpr (u == 2) BE_RING_LOAD_2(lane12, sb);
        else if constexpr (u == 3) BE_RING_LOAD_3(lane12, sb);
        else if constexpr (u == 4) BE_RING_LOAD_4(lane12, sb);
        else if constexpr (u == 5) BE_RING_LOAD_5(lane12, sb);
        else if constexpr (u == 6) BE_RING_LOAD_6(lane12, sb);
        else BE_RING_LOAD_7(lane12, sb);
    };
    auto ring_read = [&](auto U, unsigned (&w)[4]) {
        constexpr int u = decltype(U)::value;
        if constexpr (u == 0) BE_RING_READ_0(w);
        else if constexpr (u == 1) BE_RING_READ_1(w);
        else if constexpr (u == 2) BE_RING_READ_2(w);
        else if constexpr (u == 3) BE_RING_READ_3(w);
        else if constexpr (u == 4) BE_RING_READ_4(w);
        else if constexpr (u == 5) BE_RING_READ_5(w);
        else if constexpr (u == 6) BE_RING_READ_6(w);
        else BE_RING_READ_7(w);
    };
    const unsigned kg8 = (unsigned)kg * 8u;
    static_assert(BE_FD <= 8 && BE_FP <= 512, "ring registers a[64:95]; pair numbers are bytes");

    const int lane_base = C::frame_base(m16);
    const unsigned swz = (unsigned)((m16 & 7) << 4);
    int ai = a0, buf = 0, since = 63;
    // vector-memory operations retire in order: `win` remembers which of the last records' turns ended
    // with the issue of a chunk's frame copy (bit k: k + 1 records ago) -- those NDMA operations are
    // YOUNGER than the load of the record a turn is about to use and may stay in flight with the
    // BE_FD - 1 younger record loads
    unsigned win = 0;
    // Frame copies.  NBUF = 2: the copy of chunk c + 1 starts when everybody has left chunk c - 1, i.e.
    // right after the barrier -- all waves issue at once and the copy has one chunk's time to land.
    // NBUF >= 3: a wave issues its part of chunk c + NBUF - 1 when IT is through with chunk c (that
    // buffer was chunk c - 1's, which everybody left before this wave entered chunk c): the issues
    // are spread over the waves' arrival times -- the early ones do it while they wait at the
    // barrier anyway -- and a copy has NBUF - 2 chunks' time to land.
    int r_cur = 0, r_prev = 63;                    // record loads issued during this / the previous chunk
#ifndef BE_NO_PHASE
    // A launch that fills the chip once starts all workgroups together, and they stay in step: every CU
    // asks for its next chunk at the same moment and computes at the same moment -- the HBM sees
    // bursts and idles in between, and the copies take the longer for it.  Spread the workgroups over one
    // chunk period (~5 us) once, at the start.
    {
        const int steps = (int)((blockIdx.x * 7u + blockIdx.y) & 15u);
        for (int k = 0; k < steps; ++k) __builtin_amdgcn_s_sleep(BE_PHASE_SLEEP);
    }
#endif
    if constexpr (NBUF >= 3) {
#pragma unroll
        for (int d = 0; d < DIST; ++d)
            if (a0 + d < a1) issue_dma(a0 + d, d);
    } else {
        if (a0 < a1) issue_dma(a0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    bstatic_for<0, BE_FD>([&](auto U) { ring_load(U, (int64_t) decltype(U)::value); });
    if constexpr (NBUF < 3) {
        if (a0 + 1 < a1) {
            issue_dma(a0 + 1, 1);
            if (ablate != 1 && ablate != 3) { since = 0; win = 1; }
        }
    }
    const unsigned char *bbase = be_lds + lane_base;

    const bh16x2 k256 = {(_Float16)256.0f, (_Float16)256.0f};
    const bh16x2 kbias = {(_Float16)1024.0f, (_Float16)1024.0f};
    // signed pixels (int8 / int16, round 6): the TOP byte counts from -128: its sign bit is flipped (b ^ 0x80 is b + 128 as
    // an unsigned byte) and the bias taken off is 1024 + 128 -- still exact float16 integers, the same products
    constexpr bool SIGNED = std::is_signed<T>::value;
    const bh16x2 kbias_top = SIGNED ? bh16x2{(_Float16)1152.0f, (_Float16)1152.0f} : kbias;
    constexpr unsigned FLIP = !SIGNED ? 0u : (C::SZ == 2 ? 0x80008000u : 0x00008080u);

    for (int i = 0; i < n; i += BE_FD) {
        unsigned cw[BE_FD], cpa[BE_FD], cpb[BE_FD];
#pragma unroll
        for (int u = 0; u < BE_FD; ++u) {
            cw[u] = ctl[BE_CTL * (i + u)];
            cpa[u] = ctl[BE_CTL * (i + u) + 1];
            cpb[u] = ctl[BE_CTL * (i + u) + 2];
        }
        bstatic_for<0, BE_FD>([&](auto U) {
            constexpr int u = decltype(U)::value;
            const unsigned c = cw[u];
            {
                const int nb = __builtin_popcount(win & ((1u << BE_FD) - 1u));
                if (nb == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BE_FD - 1) : "memory");
                else if (nb == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BE_FD - 1 + NDMA) : "memory");
                else if (nb == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BE_FD - 1 + 2 * NDMA) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BE_FD - 1 + 3 * NDMA) : "memory");
            }
            if (!(c & BE_C_SKIP) && ablate < 4) {   // (ablate 4: frame copies + record stream only, 5: frame copies only)
                // a record = 8 aligned pixel pairs x 16 masks: lane group kg holds the pairs qa, qb; the K
                // slots of v_mfma_f32_16x16x32_f16 are [lo(qa) lo(qa+1) hi(qa) hi(qa+1) | the same of qb]
                // against [w(qa) w(qa+1) 256 w(qa) 256 w(qa+1) | ...], one MFMA for w1, one for w2
                unsigned rw[4];                   // w1(qa) pair, w1(qb) pair, w2(qa) pair, w2(qb) pair
                ring_read(U, rw);
                const bh16x2 w1a = __builtin_bit_cast(bh16x2, rw[0]), w1b = __builtin_bit_cast(bh16x2, rw[1]);
                const bh16x2 w2a = __builtin_bit_cast(bh16x2, rw[2]), w2b = __builtin_bit_cast(bh16x2, rw[3]);
                const bh16x2 w1ah = w1a * k256, w1bh = w1b * k256, w2ah = w2a * k256, w2bh = w2b * k256;
                bh16x8 a1v = {w1a[0], w1a[1], w1ah[0], w1ah[1], w1b[0], w1b[1], w1bh[0], w1bh[1]};
                bh16x8 a2v = {w2a[0], w2a[1], w2ah[0], w2ah[1], w2b[0], w2b[1], w2bh[0], w2bh[1]};
                const unsigned pa = __builtin_amdgcn_ubfe(cpa[u], kg8, 8u), pb = __builtin_amdgcn_ubfe(cpb[u], kg8, 8u);
                const unsigned char *qa = bbase + buf * C::BUF + ((pa * (2 * C::SZ)) ^ swz);
                const unsigned char *qb = bbase + buf * C::BUF + ((pb * (2 * C::SZ)) ^ swz);
                unsigned rawa[TILES], rawb[TILES];
#pragma unroll
                for (int t = 0; t < TILES; ++t) {
                    if constexpr (C::SZ == 2) {
                        rawa[t] = *(const unsigned *)(qa + t * C::TILE_OFF) ^ FLIP;
                        rawb[t] = *(const unsigned *)(qb + t * C::TILE_OFF) ^ FLIP;
                    } else {
                        rawa[t] = *(const unsigned short *)(qa + t * C::TILE_OFF) ^ FLIP;
                        rawb[t] = *(const unsigned short *)(qb + t * C::TILE_OFF) ^ FLIP;
                    }
                }
                bh16x8 bv[TILES];
#pragma unroll
                for (int t = 0; t < TILES; ++t) {
                    // bytes -> float16 without a conversion: 0x6400 | b is 1024 + b exactly
                    bh16x2 la, ha, lb, hb;
                    if constexpr (C::SZ == 2) {
                        la = __builtin_bit_cast(bh16x2, __builtin_amdgcn_perm(0x64646464u, rawa[t], 0x04020400u)) - kbias;
                        ha = __builtin_bit_cast(bh16x2, __builtin_amdgcn_perm(0x64646464u, rawa[t], 0x04030401u)) - kbias_top;
                        lb = __builtin_bit_cast(bh16x2, __builtin_amdgcn_perm(0x64646464u, rawb[t], 0x04020400u)) - kbias;
                        hb = __builtin_bit_cast(bh16x2, __builtin_amdgcn_perm(0x64646464u, rawb[t], 0x04030401u)) - kbias_top;
                    } else {
                        la = __builtin_bit_cast(bh16x2, __builtin_amdgcn_perm(0x64646464u, rawa[t], 0x04010400u)) - kbias_top;
                        lb = __builtin_bit_cast(bh16x2, __builtin_amdgcn_perm(0x64646464u, rawb[t], 0x04010400u)) - kbias_top;
                        ha = hb = bh16x2{(_Float16)0.0f, (_Float16)0.0f};
                    }
                    bv[t] = bh16x8{la[0], la[1], ha[0], ha[1], lb[0], lb[1], hb[0], hb[1]};
                }
                static_assert(BE_SLOTS == 4 && (TILES == 2 || TILES == 4), "accumulator naming");
                const unsigned sl = c & 3u;
                if (sl == 0u) { BE_ARM(0) } else if (sl == 1u) { BE_ARM(1) }
                else if (sl == 2u) { BE_ARM(2) } else { BE_ARM(3) }
            }
            if (ablate == 6) ring_load(U, (int64_t)((i + u + BE_FD) & 15));   // (records from 16 hot KiB: timing only)
            else if (ablate != 5) ring_load(U, (int64_t)(i + u + BE_FD));
            since = since < 63 ? since + 1 : since;
            win <<= 1;
            ++r_cur;
            if ((c & BE_C_END) && NBUF >= 3) {
                // this wave is through with chunk ai: start its part of the copy of chunk ai + DIST, make
                // sure its part of chunk ai + 1 has landed -- younger than that are the record loads of
                // this and the previous chunk (at most BE_FD in flight) and (NBUF - 2) later copies --
                // then everybody meets
                const bool more = ai + DIST < a1;
                if (more) issue_dma(ai + DIST, (buf + DIST) % NBUF);
                const int r2 = r_cur + r_prev;
                constexpr int YOUNGER = (NBUF - 2) * NDMA;
                if (ablate == 1 || ablate == 3) asm volatile("" ::: "memory");
                else if (!more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the last chunks)
                else if (r2 >= BE_FD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BE_FD + YOUNGER) : "memory");
                else if (r2 >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + YOUNGER) : "memory");
                else if (r2 >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 + YOUNGER) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (ablate != 3) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                ++ai;
                buf = buf + 1 == NBUF ? 0 : buf + 1;
                r_prev = r_cur;
                r_cur = 0;
                if (more && ablate != 1 && ablate != 3) win |= 1u;
            } else
            if (c & BE_C_END) {
                // this wave is through with chunk ai: the next chunk's frames must have landed -- their
                // copy is older than the last `since` record loads, and nothing else is younger --
                // then everybody meets
                if (since >= BE_FD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BE_FD) : "memory");
                else if (since >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (since >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (ablate != 3) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                ++ai;
                buf ^= 1;
                if (ai + 1 < a1) {
                    issue_dma(ai + 1, buf ^ 1);
                    if (ablate != 1 && ablate != 3) { since = 0; win |= 1u; }
                }
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (the run-ahead loads)
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");              // the last MFMAs have retired

    auto store_tile = [&](int s, int t, float v0, float v1, float v2, float v3) {
        const int col0 = pass * BE_PASS + (s * BE_SETS + j) * 16 + kg * 4;
        const int64_t f = f0 + t * 16 + m16;
        if (col0 >= n_cols || f >= n_frames) return;
        const float v[4] = {v0, v1, v2, v3};
        float *o = out + f * ld_out + col0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (col0 + r < n_cols) {
                const float x = v[r] * inv_scale[col0 + r];       // undo the column's power-of-two scale
                o[r] = accumulate ? o[r] + x : x;
            }
    };
    BE_ACC_STORE(0) BE_ACC_STORE(1) BE_ACC_STORE(2) BE_ACC_STORE(3)
}

void bell_destroy(void *image) {
    BellImage *b = (BellImage *)image;
    if (!b) return;
    if (b->stream) (void)hipFree(b->stream);
    if (b->stream_off) (void)hipFree(b->stream_off);
    if (b->nblk) (void)hipFree(b->nblk);
    if (b->active) (void)hipFree(b->active);
    if (b->active_off) (void)hipFree(b->active_off);
    if (b->tail_px) (void)hipFree(b->tail_px);
    if (b->tail_col) (void)hipFree(b->tail_col);
    if (b->tail_val) (void)hipFree(b->tail_val);
    if (b->tail_seg) (void)hipFree(b->tail_seg);
    if (b->inv_scale) (void)hipFree(b->inv_scale);
    if (b->ctrl) (void)hipFree(b->ctrl);
    if (b->ctrl_off) (void)hipFree(b->ctrl_off);
    if (b->n_rec) (void)hipFree(b->n_rec);
    if (b->h16) bell_destroy(b->h16);
    delete b;
}

// Multiply-adds of the blocked image per stored value (the padding factor that decides between
// this kernel and the SELL kernel); columns = real columns (2 per complex mask).
double bell_mac_ratio(const int64_t *indptr, const int64_t *indices, int nc, int64_t n_px,
                      int64_t n_masks) {
    const int64_t n_cols = n_masks * nc;
    const int64_t n_groups = (n_cols + 15) / 16;
    const int64_t nnz = indptr[n_px] * nc;
    if (nnz <= 0) return 1e30;
    std::vector<int64_t> last(n_groups, -1);          // last pixel counted for the group
    std::vector<int> cols(n_groups, 0);               // touched pixels of the group in this chunk
    int64_t steps = 0;
    auto flush = [&]() {
        for (int64_t g = 0; g < n_groups; ++g)
            if (cols[g]) { steps += (cols[g] + 7) / 8 * 2; cols[g] = 0; }
    };
    for (int64_t p = 0; p < n_px; ++p) {
        if (p % BE_P == 0 && p) flush();
        for (int64_t e = indptr[p]; e < indptr[p + 1]; ++e)
            for (int c = 0; c < nc; ++c) {
                const int64_t g = (indices[e] * nc + c) / 16;
                if (last[g] != p) { last[g] = p; cols[g]++; }
            }
    }
    flush();
    return (double)steps * 64.0 / (double)nnz;
}

// ---- chunk composition ----------------------------------------------------------------------------
// A workgroup's waves meet at a barrier after every chunk, so a chunk costs what its BUSIEST wave
// does.  With chunks of consecutive pixels that is far from the mean: in a ring stack the ring that is
// tangent to the chunk's detector rows puts ~100 pixels into ONE 16-mask group -- one wave works
// through a dozen records while the others hold two or three (C4: the busiest waves add up to 2.06x the
// mean; the matrix pipes idle at the barriers for the difference).  A chunk is therefore a SET of
// BE_NSEG segments (runs of BE_SEG pixels = one or two cache lines of a frame row) that the builder is
// free to choose: segments whose heavy groups belong to different waves go together.  The choice is a
// small annealing run (deterministic: fixed seed) over swaps of segments between chunks, minimising
// sum over chunks of max over waves of the records (+ a little of the total, which grows when segments
// that share groups are separated); it starts from the better of the natural order and a strided
// interleave.  C4: busiest-wave records 1589 -> ~960 for 4 % more records.
struct ChunkPlanner {
    int n_local;                                    // groups of the pass (<= BE_SETS * BE_SLOTS)
    const std::vector<uint16_t> &cnt;               // [segment][n_local] touched pixels (f16: pixel pairs)
    int upr;                                        // of them per record: 8 pixels / 4 pairs
    int nseg;                                       // segments per chunk
    std::vector<int> tmp;
    ChunkPlanner(int nl, const std::vector<uint16_t> &c, int u, int ns)
        : n_local(nl), cnt(c), upr(u), nseg(ns), tmp((size_t)nl) {}
    // (records of the busiest wave, records of all waves) of a chunk made of `segs`
    void cost(const int *segs, int n, int *crit, int *total) {
        std::fill(tmp.begin(), tmp.end(), 0);
        for (int i = 0; i < n; ++i) {
            if (segs[i] < 0) continue;
            const uint16_t *row = cnt.data() + (size_t)segs[i] * n_local;
            for (int g = 0; g < n_local; ++g) tmp[g] += row[g];
        }
        int load[BE_SETS] = {0}, tot = 0;
        for (int g = 0; g < n_local; ++g) {
            const int rec = (tmp[g] + upr - 1) / upr;
            load[g % BE_SETS] += rec;                 // group g of the pass: wave g % BE_SETS
            tot += rec;
        }
        int mx = 0;
        for (int w = 0; w < BE_SETS; ++w) mx = std::max(mx, load[w]);
        *crit = mx;
        *total = tot;
    }
    // segs: n_chunks * nseg segment numbers (index into cnt), -1 = empty slot; optimised in place
    long plan(std::vector<int> &segs, long *total_out) {
        const int n_chunks = (int)(segs.size() / nseg);
        std::vector<int> cc((size_t)n_chunks), ct((size_t)n_chunks);
        auto eval_all = [&](const std::vector<int> &a, long *tot) {
            long c = 0, t = 0;
            for (int k = 0; k < n_chunks; ++k) {
                int x, y;
                cost(a.data() + (size_t)k * nseg, nseg, &x, &y);
                c += x; t += y;
            }
            *tot = t;
            return c;
        };
        constexpr double W_TOTAL = 0.05;             // weight of a record anywhere vs on the critical wave
        // start: natural order or strided interleave (segment k * n_chunks + c -> chunk c), whichever is
        // better
        std::vector<int> inter(segs.size(), -1);
        {
            std::vector<int> fill((size_t)n_chunks, 0);
            int k = 0;
            for (int v : segs) {
                if (v < 0) continue;
                const int c = k % n_chunks;
                inter[(size_t)c * nseg + fill[c]++] = v;
                ++k;
            }
        }
        long t_nat, t_int;
        const long c_nat = eval_all(segs, &t_nat), c_int = eval_all(inter, &t_int);
        if (c_int + W_TOTAL * t_int < c_nat + W_TOTAL * t_nat) segs = inter;
        for (int k = 0; k < n_chunks; ++k) cost(segs.data() + (size_t)k * nseg, nseg, &cc[k], &ct[k]);
        if (n_chunks >= 2) {
            uint64_t rng = 0x9E3779B97F4A7C15ull;
            auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
            const long iters = std::min<long>(400000, 150L * (long)segs.size() + 2000);
            for (long it = 0; it < iters; ++it) {
                const int A = (int)(next() % n_chunks), B = (int)(next() % n_chunks);
                if (A == B) continue;
                const size_t ia = (size_t)A * nseg + next() % nseg, ib = (size_t)B * nseg + next() % nseg;
                if (segs[ia] < 0 && segs[ib] < 0) continue;
                std::swap(segs[ia], segs[ib]);
                int ca, ta, cb, tb;
                cost(segs.data() + (size_t)A * nseg, nseg, &ca, &ta);
                cost(segs.data() + (size_t)B * nseg, nseg, &cb, &tb);
                const double d = (ca + cb - cc[A] - cc[B]) + W_TOTAL * (ta + tb - ct[A] - ct[B]);
                const double T = std::max(0.02, 1.0 - (double)it / (double)iters);
                const double u = (double)(next() >> 11) * (1.0 / 9007199254740992.0);
                if (d <= 0 || u < std::exp(-d / T)) {
                    cc[A] = ca; ct[A] = ta; cc[B] = cb; ct[B] = tb;
                } else {
                    std::swap(segs[ia], segs[ib]);
                }
            }
        }
        long c = 0, t = 0;
        for (int k = 0; k < n_chunks; ++k) { c += cc[k]; t += ct[k]; }
        *total_out = t;
        return c;
    }
};

static uint16_t half_bits(double x) {
    const _Float16 h = (_Float16)x;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
static double half_value(uint16_t u) {
    _Float16 h;
    memcpy(&h, &u, 2);
    return (double)h;
}

// Build a blocked image from the CSR matrix (n_px x n_masks): f16 = false -> records of 8 pixels with
// float32 weights (any pixel type), f16 = true -> records of 4 aligned pixel pairs with split float16
// weights (k_bell_flat; `col_scale`: the columns' power-of-two scales).
struct ExtraTail {                                   // entries a caller wants applied by k_bell_tail as well
    std::vector<int32_t> px, col;
    std::vector<float> val;
};

static BellImage *build_image(const int64_t *indptr, const int64_t *indices, const float *vals, int nc,
                              int64_t n_px, int64_t n_masks, bool f16,
                              const std::vector<float> &col_scale, int *err,
                              const ExtraTail *extra = nullptr) {
    *err = LTMI_OK;
    BellImage *b = new (std::nothrow) BellImage();
    if (!b) { *err = LTMI_E_NOMEM; return nullptr; }
    b->f16 = f16;
    const int UPR = 8;                                  // units (pixels / f16: pixel pairs) per record
    const int REC = f16 ? BE_REC16 : BE_REC;          // dwords per record
    const int NSEG = f16 ? BE_FNSEG : BE_NSEG;        // segments per chunk
    try {
        const int64_t n_cols = n_masks * nc;
        const int64_t n_groups = (n_cols + 15) / 16;
        constexpr int GPP = BE_SETS * BE_SLOTS;         // groups per pass
        const int n_pass = (int)((n_groups + GPP - 1) / GPP);
        b->n_pass = n_pass;
        struct Ent { uint16_t px; uint16_t m; float v; };
        const int64_t p_tail0 = n_px - n_px % 16;       // pixels from here on: k_bell_tail
        const int64_t n_seg = (p_tail0 + BE_SEG - 1) / BE_SEG;
        std::vector<int32_t> tail_px, tail_col;
        std::vector<float> tail_val;
        for (int64_t p = p_tail0; p < n_px; ++p)
            for (int64_t e = indptr[p]; e < indptr[p + 1]; ++e)
                for (int c = 0; c < nc; ++c) {
                    tail_px.push_back((int32_t)p);
                    tail_col.push_back((int32_t)(indices[e] * nc + c));
                    tail_val.push_back(vals[e * nc + c]);
                }
        if (extra) {
            tail_px.insert(tail_px.end(), extra->px.begin(), extra->px.end());
            tail_col.insert(tail_col.end(), extra->col.begin(), extra->col.end());
            tail_val.insert(tail_val.end(), extra->val.begin(), extra->val.end());
        }
        std::vector<int32_t> tail_seg;
        {
            std::vector<size_t> order(tail_px.size());
            for (size_t i = 0; i < order.size(); ++i) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return tail_col[x] < tail_col[y]; });
            std::vector<int32_t> p2(order.size()), c2(order.size());
            std::vector<float> v2(order.size());
            for (size_t i = 0; i < order.size(); ++i) {
                p2[i] = tail_px[order[i]];
                c2[i] = tail_col[order[i]];
                v2[i] = tail_val[order[i]];
                if (i == 0 || c2[i] != c2[i - 1]) tail_seg.push_back((int32_t)i);
            }
            tail_seg.push_back((int32_t)order.size());
            tail_px.swap(p2);
            tail_col.swap(c2);
            tail_val.swap(v2);
        }
        b->n_tail = (int)tail_px.size();
        b->n_tail_cols = (int)tail_seg.size() - 1;

        // ---- per pass: which segments hold entries, how they are grouped into chunks
        std::vector<int> active, active_off(n_pass + 1, 0);     // active: [chunk][NSEG] segment numbers
        std::vector<int> chunk_of_seg((size_t)n_pass * n_seg, -1), slot_of_seg((size_t)n_pass * n_seg, 0);
        const bool natural = getenv("LTMI_BELL_NATURAL_CHUNKS") != nullptr;     // experiments: no planning
        for (int ps = 0; ps < n_pass; ++ps) {
            const int64_t g_lo = (int64_t)ps * GPP;
            const int n_local = (int)std::min<int64_t>(GPP, n_groups - g_lo);
            // touched pixels per (segment, group of the pass)
            std::vector<int> seg_id((size_t)n_seg, -1);          // segment -> row of `cnt`
            std::vector<int> seg_list;
            std::vector<uint16_t> cnt;
            std::vector<int64_t> last((size_t)n_local, -1);
            for (int64_t p = 0; p < p_tail0; ++p) {
                const int64_t sg = p / BE_SEG;
                const int64_t unit = f16 ? p >> 1 : p;
                for (int64_t e = indptr[p]; e < indptr[p + 1]; ++e)
                    for (int c = 0; c < nc; ++c) {
                        const int64_t g = (indices[e] * nc + c) / 16 - g_lo;
                        if (g < 0 || g >= n_local || last[g] == unit) continue;
                        last[g] = unit;
                        if (seg_id[sg] < 0) {
                            seg_id[sg] = (int)seg_list.size();
                            seg_list.push_back((int)sg);
                            cnt.resize(cnt.size() + (size_t)n_local, 0);
                        }
                        cnt[(size_t)seg_id[sg] * n_local + g]++;
                    }
            }
            const int n_act = (int)seg_list.size();
            const int n_chunks = (n_act + NSEG - 1) / NSEG;
            std::vector<int> segs((size_t)n_chunks * NSEG, -1);
            for (int k = 0; k < n_act; ++k) segs[k] = k;
            ChunkPlanner planner(n_local, cnt, UPR, NSEG);
            long total = 0;
            if (natural) {
                for (int k = 0; k < n_chunks; ++k) {
                    int x, y;
                    planner.cost(segs.data() + (size_t)k * NSEG, NSEG, &x, &y);
                    b->crit_records += x;
                }
            } else {
                b->crit_records += planner.plan(segs, &total);
            }
            const int chunk0 = (int)(active.size() / NSEG);
            for (int k = 0; k < n_chunks; ++k)
                for (int i = 0; i < NSEG; ++i) {
                    const int v = segs[(size_t)k * NSEG + i];
                    // (an empty slot fetches the frame's first pixels: never referenced by a record)
                    active.push_back(v < 0 ? 0 : seg_list[v] * BE_SEG);
                    if (v >= 0) {
                        chunk_of_seg[(size_t)ps * n_seg + seg_list[v]] = chunk0 + k;
                        slot_of_seg[(size_t)ps * n_seg + seg_list[v]] = i;
                    }
                }
            active_off[ps + 1] = (int)(active.size() / NSEG);
        }
        const int n_chunks_all = active_off[n_pass];

        // ---- entries per (chunk, group); pixel numbers are positions in the chunk's LDS image
        // (segment slot * BE_SEG + pixel in the segment), ascending within a (chunk, group) only per
        // segment -- the records do not care
        std::vector<std::vector<Ent>> bucket((size_t)std::max(n_chunks_all, 1) * GPP);
        for (int64_t p = 0; p < p_tail0; ++p) {
            const int64_t sg = p / BE_SEG;
            for (int64_t e = indptr[p]; e < indptr[p + 1]; ++e)
                for (int c = 0; c < nc; ++c) {
                    const int64_t col = indices[e] * nc + c;
                    const int64_t g = col / 16;
                    const int ps = (int)(g / GPP);
                    const int ch = chunk_of_seg[(size_t)ps * n_seg + sg];
                    const int local = slot_of_seg[(size_t)ps * n_seg + sg] * BE_SEG + (int)(p % BE_SEG);
                    bucket[(size_t)ch * GPP + (g - (int64_t)ps * GPP)].push_back(
                        Ent{(uint16_t)local, (uint16_t)(col % 16), vals[e * nc + c]});
                }
        }
        // a (chunk, group) list in ascending position (entries of one pixel stay adjacent)
        for (auto &v : bucket)
            std::stable_sort(v.begin(), v.end(), [](const Ent &a, const Ent &c) { return a.px < c.px; });

        const size_t n_act = (size_t)std::max(n_chunks_all, 1);
        std::vector<int> nblk(n_act * BE_SETS * BE_SLOTS, 0);
        std::vector<int64_t> stream_off((size_t)n_pass * BE_SETS, 0);
        std::vector<uint32_t> stream;
        // f16 (k_bell_flat): a control word per record, one flat stream per (pass, wave)
        std::vector<uint32_t> ctrl;
        std::vector<int64_t> ctrl_off((size_t)n_pass * BE_SETS, 0);
        std::vector<int> n_rec((size_t)n_pass * BE_SETS, 0);
        size_t blocks = 0, pad_blocks = 0;
        std::vector<uint16_t> cols;
        std::vector<uint32_t> pxws;
        for (int ps = 0; ps < n_pass; ++ps)
            for (int j = 0; j < BE_SETS; ++j) {
                stream_off[(size_t)ps * BE_SETS + j] = (int64_t)blocks;
                ctrl.resize((ctrl.size() + 15) / 16 * 16, BE_C_SKIP);         // 64-byte aligned streams
                ctrl_off[(size_t)ps * BE_SETS + j] = (int64_t)ctrl.size();
                const size_t ctrl0 = ctrl.size();
                for (int ai = active_off[ps]; ai < active_off[ps + 1]; ++ai) {
                    const size_t ctrl_chunk0 = ctrl.size();
                    for (int s = 0; s <= BE_SLOTS; ++s) {
                        if (s == BE_SLOTS) {
                            // end of the wave's work on this chunk
                            if (f16) {
                                if (ctrl.size() == ctrl_chunk0) {       // nothing: a record to carry the flag
                                    stream.resize(stream.size() + REC, 0u);
                                    ctrl.push_back(BE_C_SKIP);
                                    ctrl.insert(ctrl.end(), BE_CTL - 1, 0u);
                                    ++blocks;
                                    ++pad_blocks;
                                }
                                ctrl[ctrl.size() - BE_CTL] |= BE_C_END;
                            }
                            break;
                        }
                        const int gl = s * BE_SETS + j;            // group of the pass
                        if ((int64_t)ps * GPP + gl >= n_groups) continue;
                        const std::vector<Ent> &ents = bucket[(size_t)ai * GPP + gl];
                        if (ents.empty()) continue;
                        cols.clear();
                        // units of the pair list: pixels, or (f16) aligned pixel pairs named by their
                        // even pixel
                        for (const Ent &en : ents) {
                            const uint16_t u = f16 ? (uint16_t)(en.px & ~1u) : en.px;
                            if (cols.empty() || cols.back() != u) cols.push_back(u);
                        }
                        const int nb = ((int)cols.size() + UPR - 1) / UPR;
                        nblk[((size_t)ai * BE_SETS + j) * BE_SLOTS + s] = nb;
                        const size_t base = stream.size();
                        stream.resize(base + (size_t)nb * REC, 0u);
                        if (!f16) {
                            // pixel numbers: lane l -> columns 8*blk + (l >> 4) and + 4
                            for (int blk = 0; blk < nb; ++blk)
                                for (int l = 0; l < 64; ++l) {
                                    const int c0 = blk * 8 + (l >> 4), c1 = c0 + 4;
                                    // padding columns (value 0) repeat the block's first pixel rather
                                    // than pixel 0 of the chunk: a non-finite pixel then only reaches
                                    // blocks that really contain it (0 * NaN = NaN)
                                    const uint32_t p0 = c0 < (int)cols.size() ? cols[c0] : cols[blk * 8];
                                    const uint32_t p1 = c1 < (int)cols.size() ? cols[c1] : cols[blk * 8];
                                    stream[base + (size_t)blk * REC + l * 3 + 2] = p0 | (p1 << 16);
                                }
                            size_t ci = 0;
                            for (const Ent &en : ents) {
                                while (cols[ci] != en.px) ++ci;
                                const int blk = (int)(ci / 8), step = (int)((ci % 8) / 4), kk = (int)(ci % 4);
                                const int l = kk * 16 + en.m;
                                uint32_t bits;
                                memcpy(&bits, &en.v, 4);
                                stream[base + (size_t)blk * REC + l * 3 + step] = bits;
                            }
                        } else {
                            // lane l = (mask l & 15, lane group kg = l >> 4): the group holds the pairs
                            // a = 8 blk + kg and b = 8 blk + 4 + kg of the list; words
                            // [w1(a) | w1(a+1) << 16, w1(b) .., w2(a) .., w2(b) ..]; the pairs' numbers q / 2 are
                            // the bytes of the record's control words 1 (a) and 2 (b)
                            for (int blk = 0; blk < nb; ++blk) {
                                uint32_t pa = 0, pb = 0;
                                for (int kk = 0; kk < 4; ++kk) {
                                    const int ca = blk * 8 + kk, cb = ca + 4;
                                    const uint32_t first = cols[blk * 8];
                                    const uint32_t qa = ca < (int)cols.size() ? cols[ca] : first;
                                    const uint32_t qb = cb < (int)cols.size() ? cols[cb] : first;
                                    pa |= (qa >> 1) << (8 * kk);
                                    pb |= (qb >> 1) << (8 * kk);
                                }
                                pxws.push_back(pa);
                                pxws.push_back(pb);
                            }
                            size_t ci = 0;
                            const int64_t col_base = ((int64_t)ps * GPP + gl) * 16;
                            for (const Ent &en : ents) {
                                while (cols[ci] != (uint16_t)(en.px & ~1u)) ++ci;
                                const int blk = (int)(ci / 8), half = (int)((ci % 8) / 4), kk = (int)(ci % 4);
                                const int l = kk * 16 + en.m;
                                const double ws = (double)en.v * (double)col_scale[(size_t)(col_base + en.m)];
                                const uint16_t h1 = half_bits(ws);
                                const uint16_t h2 = half_bits(ws - half_value(h1));
                                const int sh = (en.px & 1) ? 16 : 0;
                                uint32_t *rec = &stream[base + (size_t)blk * REC + l * 4];
                                rec[half] |= (uint32_t)h1 << sh;
                                rec[2 + half] |= (uint32_t)h2 << sh;
                            }
                        }
                        blocks += nb;
                        if (f16)
                            for (int blk = 0; blk < nb; ++blk) {
                                ctrl.push_back((uint32_t)s);
                                ctrl.push_back(pxws[(size_t)blk * 2]);
                                ctrl.push_back(pxws[(size_t)blk * 2 + 1]);
                                ctrl.push_back(0u);
                            }
                        pxws.clear();
                    }
                }
                if (f16) {
                    // whole turns of the unrolled loop, then BE_FD records of slack for its run-ahead loads
                    while (((ctrl.size() - ctrl0) / BE_CTL) % BE_FD) {
                        stream.resize(stream.size() + REC, 0u);
                        ctrl.push_back(BE_C_SKIP);
                        ctrl.insert(ctrl.end(), BE_CTL - 1, 0u);
                        ++blocks;
                        ++pad_blocks;
                    }
                    n_rec[(size_t)ps * BE_SETS + j] = (int)((ctrl.size() - ctrl0) / BE_CTL);
                    stream.resize(stream.size() + (size_t)BE_FD * REC, 0u);
                    for (int k = 0; k < BE_FD; ++k) { ctrl.push_back(BE_C_SKIP); ctrl.insert(ctrl.end(), BE_CTL - 1, 0u); }
                    blocks += BE_FD;
                    pad_blocks += BE_FD;
                } else {
                    // slack for the run-ahead loads of the last records
                    stream.resize(stream.size() + (size_t)BE_D_MAX * REC, 0u);
                    blocks += BE_D_MAX;
                    pad_blocks += BE_D_MAX;
                }
            }
        ctrl.resize(ctrl.size() + 2 * BE_CTL * BE_FD, BE_C_SKIP);          // (the control words are read one turn ahead)
        b->n_blocks = blocks;
        int64_t nnz = indptr[n_px] * nc;
        b->mac_ratio = nnz > 0 ? (double)(blocks - pad_blocks) * (f16 ? 256.0 : 128.0) / (double)nnz : 0.;
        if (active.empty()) active.resize(NSEG, 0);
        hipError_t e = hipMalloc((void **)&b->stream, std::max<size_t>(stream.size(), 1) * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&b->stream_off, stream_off.size() * 8);
        if (e == hipSuccess) e = hipMalloc((void **)&b->nblk, nblk.size() * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&b->active, active.size() * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&b->active_off, active_off.size() * 4);
        if (e == hipSuccess && !stream.empty())
            e = hipMemcpy(b->stream, stream.data(), stream.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(b->stream_off, stream_off.data(), stream_off.size() * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(b->nblk, nblk.data(), nblk.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(b->active, active.data(), active.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(b->active_off, active_off.data(), active_off.size() * 4, hipMemcpyHostToDevice);
        if (b->n_tail > 0) {
            const size_t nt = (size_t)b->n_tail;
            if (e == hipSuccess) e = hipMalloc((void **)&b->tail_px, nt * 4);
            if (e == hipSuccess) e = hipMalloc((void **)&b->tail_col, nt * 4);
            if (e == hipSuccess) e = hipMalloc((void **)&b->tail_val, nt * 4);
            if (e == hipSuccess) e = hipMemcpy(b->tail_px, tail_px.data(), nt * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(b->tail_col, tail_col.data(), nt * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(b->tail_val, tail_val.data(), nt * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMalloc((void **)&b->tail_seg, tail_seg.size() * 4);
            if (e == hipSuccess) e = hipMemcpy(b->tail_seg, tail_seg.data(), tail_seg.size() * 4, hipMemcpyHostToDevice);
        }
        if (f16 && e == hipSuccess) {
            e = hipMalloc((void **)&b->ctrl, ctrl.size() * 4);
            if (e == hipSuccess) e = hipMalloc((void **)&b->ctrl_off, ctrl_off.size() * 8);
            if (e == hipSuccess) e = hipMalloc((void **)&b->n_rec, n_rec.size() * 4);
            if (e == hipSuccess) e = hipMemcpy(b->ctrl, ctrl.data(), ctrl.size() * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(b->ctrl_off, ctrl_off.data(), ctrl_off.size() * 8, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(b->n_rec, n_rec.data(), n_rec.size() * 4, hipMemcpyHostToDevice);
        }
        if (f16 && e == hipSuccess) {
            std::vector<float> inv(col_scale.size());
            for (size_t k = 0; k < inv.size(); ++k) inv[k] = 1.0f / col_scale[k];
            e = hipMalloc((void **)&b->inv_scale, std::max<size_t>(inv.size(), 1) * 4);
            if (e == hipSuccess && !inv.empty())
                e = hipMemcpy(b->inv_scale, inv.data(), inv.size() * 4, hipMemcpyHostToDevice);
        }
        if (e != hipSuccess) {
            set_error("uploading the blocked sparse mask image failed: %s", hipGetErrorString(e));
            *err = (int)e;
            bell_destroy(b);
            return nullptr;
        }
    } catch (const std::bad_alloc &) {
        set_error("out of host memory while packing the blocked sparse mask image");
        *err = LTMI_E_NOMEM;
        bell_destroy(b);
        return nullptr;
    }
    return b;
}

// The image(s) of a stack: the float32 one (any pixel type) and -- when every weight is finite -- the
// float16 one for 1- and 2-byte unsigned pixels.  Column scale of the float16 image: the power of two
// S with max |w| S in (64, 128], so that 256 w S stays below the float16 maximum and small weights
// stay clear of its subnormals; the kernel multiplies the column by 1 / S at the end (exact).
void *bell_build(const int64_t *indptr, const int64_t *indices, const float *vals, int nc,
                 int64_t n_px, int64_t n_masks, int *err) {
    const std::vector<float> none;
    BellImage *b = build_image(indptr, indices, vals, nc, n_px, n_masks, false, none, err);
    if (!b) return nullptr;
    {
        std::vector<int64_t> per_mask((size_t)n_masks, 0);
        for (int64_t e = 0; e < indptr[n_px]; ++e) per_mask[(size_t)indices[e]]++;
        for (int64_t v : per_mask) b->max_col_entries = std::max(b->max_col_entries, v);
    }
    const char *off = getenv("LTMI_BELL_F16");
    if (off && atoi(off) == 0) return b;
    const int64_t n_cols = n_masks * nc;
    std::vector<float> amax((size_t)n_cols, 0.f);
    bool finite = true;
    const int64_t nnz = indptr[n_px];
    for (int64_t e = 0; e < nnz && finite; ++e)
        for (int c = 0; c < nc; ++c) {
            const float v = vals[e * nc + c];
            if (!std::isfinite(v)) { finite = false; break; }
            float &m = amax[(size_t)(indices[e] * nc + c)];
            m = std::max(m, std::fabs(v));
        }
    if (!finite) return b;
    std::vector<float> scale((size_t)n_cols, 1.f);
    for (int64_t k = 0; k < n_cols; ++k)
        if (amax[(size_t)k] > 0.f) {
            int ex;
            (void)std::frexp(amax[(size_t)k], &ex);          // amax = f * 2^ex, f in [0.5, 1)
            // (a column beyond 2^-100 .. 2^100 cannot be scaled into float16 by a float32 power of two
            // with room to spare: the float32 image stays in charge of such stacks)
            if (ex < -100 || ex > 100) return b;
            const int sh = std::max(-120, std::min(120, 7 - ex));   // amax * 2^sh in [64, 128)
            scale[(size_t)k] = std::ldexp(1.0f, sh);
        }
    // Two float16 pieces of w S carry 22 bits while the second piece is a normal float16 number; below
    // ~2^-10 of the column's maximum it falls on the subnormal grid (absolute error 2^-31 max|w|).  The few
    // entries whose pieces miss them by more than 2^-19 relative (C4: 32 of 432 407) are taken out of the
    // float16 image -- they stay stored, with weight 0 -- and are applied by k_bell_tail as ONE float32
    // product each, the reference's arithmetic (common/numba/__init__.py:169-184).  A stack with more than
    // BELL_TAIL_MAX of them keeps the float32 image.
    constexpr size_t BELL_TAIL_MAX = 256;
    ExtraTail extra;
    std::vector<float> vals16(vals, vals + (size_t)nnz * nc);
    for (int64_t p = 0; p < n_px - n_px % 16; ++p)
        for (int64_t e = indptr[p]; e < indptr[p + 1]; ++e)
            for (int c = 0; c < nc; ++c) {
                const int64_t col = indices[e] * nc + c;
                const float v = vals[e * nc + c];
                if (v == 0.f) continue;
                const float ws = v * scale[(size_t)col];
                const _Float16 w1 = (_Float16)ws;
                const float r = ws - (float)w1;
                const _Float16 w2 = (_Float16)r;
                if (std::fabs(r - (float)w2) <= std::ldexp(std::fabs(ws), -19)) continue;
                if (extra.val.size() == BELL_TAIL_MAX) return b;
                extra.px.push_back((int32_t)p);
                extra.col.push_back((int32_t)col);
                extra.val.push_back(v);
                vals16[(size_t)(e * nc + c)] = 0.f;
            }
    int err16 = LTMI_OK;
    b->h16 = build_image(indptr, indices, vals16.data(), nc, n_px, n_masks, true, scale, &err16, &extra);
    // (a failure here -- memory -- leaves the float32 image in charge)
    return b;
}

// the entries of the last n_px % 16 pixels and the entries the float16 image leaves to float32 (see
// BellImage::tail_*): sorted by column; one thread per (frame, column) sums its column's products in float32
// and adds them to the result -- a sum is updated by one thread only: no atomics, launches are reproducible
template <typename T>
__global__ void k_bell_tail(const T *__restrict__ tile, int64_t ld, int64_t n_frames,
                            const int32_t *__restrict__ px, const int32_t *__restrict__ col,
                            const float *__restrict__ val, const int32_t *__restrict__ seg, int n_seg,
                            float *__restrict__ out, int64_t ld_out, const int32_t *__restrict__ rows) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    const T *row = tile + (rows ? (int64_t)rows[f] : f) * ld;
    // (grid.y is capped at 65 535: a stack with more tail columns than that walks them in strides)
    for (int c = blockIdx.y; c < n_seg; c += gridDim.y) {
        const int e0 = seg[c], e1 = seg[c + 1];
        float acc = 0.f;
        for (int e = e0; e < e1; ++e) acc += val[e] * (float)row[px[e]];
        out[f * ld_out + col[e0]] += acc;
    }
}

template <typename T, int TL>
static int launch_bell_t(ltmi_masks *m, BellImage *b, const T *tile, int64_t n_frames, int64_t ld,
                         float *out, int64_t ld_out_f, int n_cols, int accumulate, hipStream_t stream) {
    using C = BeCfg<T, TL>;
    auto kern = k_bell_apply<T, TL>;
    static bool set[16] = {false};
    if (!set[m->device & 15]) {
        LTMI_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     C::LDS_BYTES));
        set[m->device & 15] = true;
    }
    dim3 grid((unsigned)((n_frames + C::FB - 1) / C::FB), (unsigned)b->n_pass);
    const char *abl = getenv("LTMI_BELL_ABLATE");    // 1: no frame DMA, 2: no records (bench only)
    const int ablate = abl ? atoi(abl) : 0;
    unsigned long long *prof = nullptr;
#ifdef BE_PROF
    static unsigned long long *prof_dev = nullptr;
    if (!prof_dev) LTMI_HIP(hipMalloc((void **)&prof_dev, 8 * sizeof(unsigned long long)));
    LTMI_HIP(hipMemsetAsync(prof_dev, 0, 8 * sizeof(unsigned long long), stream));
    prof = prof_dev;
#endif
    hipLaunchKernelGGL(kern, grid, dim3(BE_SETS * 64), C::LDS_BYTES, stream, tile, ld, n_frames,
                       m->n_px, (const uint32_t *)b->stream, (const int64_t *)b->stream_off,
                       (const int *)b->nblk, (const int *)b->active,
                       (const int *)b->active_off, out,
                       ld_out_f, n_cols, accumulate, ablate, prof, m->roi_rows);
    LTMI_HIP(hipGetLastError());
#ifdef BE_PROF
    {
        unsigned long long h[8];
        LTMI_HIP(hipStreamSynchronize(stream));
        LTMI_HIP(hipMemcpy(h, prof_dev, sizeof(h), hipMemcpyDeviceToHost));
        const double w = (double)h[7];
        fprintf(stderr, "BE_PROF cycles per wave (%.0f waves, ~%.0f records each): record-wait %.0f  "
                "lds %.0f  mfma+refill %.0f  loop %.0f  dma-wait %.0f  barrier %.0f  total %.0f\n",
                w, (double)b->n_blocks / BE_SETS, h[0] / w, h[1] / w, h[2] / w, h[3] / w, h[4] / w,
                h[5] / w, h[6] / w);
    }
#endif
    if (b->n_tail > 0) {
        hipLaunchKernelGGL(k_bell_tail<T>, dim3((unsigned)((n_frames + 255) / 256), (unsigned)std::min(b->n_tail_cols, 65535)),
                           dim3(256), 0, stream, tile, ld, n_frames, (const int32_t *)b->tail_px,
                           (const int32_t *)b->tail_col, (const float *)b->tail_val,
                           (const int32_t *)b->tail_seg, (int)b->n_tail_cols, out, ld_out_f, m->roi_rows);
        LTMI_HIP(hipGetLastError());
    }
    snprintf(m->last_kernel, sizeof(m->last_kernel),
             "k_bell_apply<%s,tiles=%d%s> grid=(%u,%u) blocks=%zu x%.2f crit=%ld", typeid(T).name(), TL,
             m->roi_rows ? ",rows" : "", grid.x, grid.y, b->n_blocks, b->mac_ratio, b->crit_records);
    return LTMI_OK;
}

template <typename T, int TL>
static int launch_bell_flat(ltmi_masks *m, BellImage *b, const T *tile, int64_t n_frames, int64_t ld,
                            float *out, int64_t ld_out_f, int n_cols, int accumulate, hipStream_t stream) {
    using C = BeCfg<T, TL, BE_FP>;
    auto kern = k_bell_flat<T, TL>;
    constexpr int LDS = BE_FNBUF * C::BUF;
    static bool set[16] = {false};
    if (!set[m->device & 15]) {
        LTMI_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        set[m->device & 15] = true;
    }
    dim3 grid((unsigned)((n_frames + C::FB - 1) / C::FB), (unsigned)b->n_pass);
    const char *abl = getenv("LTMI_BELL_ABLATE");
    const int ablate = abl ? atoi(abl) : 0;
    hipLaunchKernelGGL(kern, grid, dim3(BE_SETS * 64), LDS, stream, tile, ld, n_frames, m->n_px,
                       (const uint32_t *)b->stream, (const int64_t *)b->stream_off,
                       (const uint32_t *)b->ctrl, (const int64_t *)b->ctrl_off, (const int *)b->n_rec,
                       (const int *)b->active, (const int *)b->active_off, out, ld_out_f, n_cols,
                       accumulate, ablate, m->roi_rows, (const float *)b->inv_scale);
    LTMI_HIP(hipGetLastError());
    if (b->n_tail > 0) {
        hipLaunchKernelGGL(k_bell_tail<T>, dim3((unsigned)((n_frames + 255) / 256), (unsigned)std::min(b->n_tail_cols, 65535)),
                           dim3(256), 0, stream, tile, ld, n_frames, (const int32_t *)b->tail_px,
                           (const int32_t *)b->tail_col, (const float *)b->tail_val,
                           (const int32_t *)b->tail_seg, (int)b->n_tail_cols, out, ld_out_f, m->roi_rows);
        LTMI_HIP(hipGetLastError());
    }
    snprintf(m->last_kernel, sizeof(m->last_kernel),
             "k_bell_flat<%s,tiles=%d,f16%s> grid=(%u,%u) blocks=%zu x%.2f crit=%ld tail=%d", typeid(T).name(), TL,
             m->roi_rows ? ",rows" : "", grid.x, grid.y, b->n_blocks, b->mac_ratio, b->crit_records, b->n_tail);
    return LTMI_OK;
}

// Frames per workgroup: one workgroup (16 waves) per CU at a time.  More 16-frame tiles per wave spread
// the fixed cost of a block record (ring read, gather addresses, refill) over more MFMAs -- 64 frames
// per workgroup take 1.65 - 1.8x the time of 32 (profiles/r02_sparse_experiments.txt) -- but make fewer, longer workgroups: whichever needs the
// shorter sequence of rounds on the 256 CUs (LTMI_BELL_TILES forces one).
template <typename T>
static int launch_bell(ltmi_masks *m, BellImage *b, const T *tile, int64_t n_frames, int64_t ld,
                       float *out, int64_t ld_out_f, int n_cols, int accumulate, hipStream_t stream) {
    constexpr int LO = sizeof(T) == 4 ? 1 : 2, HI = 2 * LO;
    static const int forced = getenv("LTMI_BELL_TILES") ? atoi(getenv("LTMI_BELL_TILES")) : 0;
    auto rounds = [&](int tl) { return (double)(((n_frames + 16 * tl - 1) / (16 * tl) + 255) / 256); };
    const bool hi = forced ? forced == HI : rounds(HI) * 1.7 < rounds(LO);
    // k_bell_apply with four tiles has one accumulation level only: not for columns of more than 2048 stored values
    const bool hi_apply = forced ? hi : (hi && (HI <= 2 || b->max_col_entries <= 2048));
    if constexpr (std::is_integral<T>::value && sizeof(T) <= 2) {
        if (b->h16) {               // 1- / 2-byte integer pixels (signed ones since round 6): the float16 image
            if (hi)
                return launch_bell_flat<T, HI>(m, b->h16, tile, n_frames, ld, out, ld_out_f, n_cols,
                                               accumulate, stream);
            return launch_bell_flat<T, LO>(m, b->h16, tile, n_frames, ld, out, ld_out_f, n_cols,
                                           accumulate, stream);
        }
    }
    if (hi_apply)
        return launch_bell_t<T, HI>(m, b, tile, n_frames, ld, out, ld_out_f, n_cols, accumulate, stream);
    return launch_bell_t<T, LO>(m, b, tile, n_frames, ld, out, ld_out_f, n_cols, accumulate, stream);
}

// handled = false: the tile does not meet the kernel's rules (caller uses the SELL kernel)
int bell_apply(ltmi_masks *m, void *image, int cplx, const void *tile, int tile_dtype,
               int64_t n_frames, int64_t ld_tile, void *out, int64_t ld_out, int accumulate,
               hipStream_t stream, bool *handled) {
    BellImage *b = (BellImage *)image;
    const int sz = dtype_size(tile_dtype);
    *handled = false;
    if (!b || n_frames <= 0) return LTMI_OK;
    // rows of any element alignment (LDS-DMA reads them); a partial last 16-byte piece of a row is not
    // fetched -- its pixels (< 16) are k_bell_tail's
    if (sz <= 0 || !vector_loads_ok(tile, ld_tile, (size_t)sz)) return LTMI_OK;
    const int nc = cplx ? 2 : 1;
    const int n_cols = (int)(m->n_masks * nc);
    float *o = (float *)out;
    const int64_t ldo = ld_out * nc;
    *handled = true;
    switch (tile_dtype) {
        case LTMI_BOOL:
        case LTMI_U8: return launch_bell<uint8_t>(m, b, (const uint8_t *)tile, n_frames, ld_tile, o, ldo, n_cols, accumulate, stream);
        case LTMI_I8: return launch_bell<int8_t>(m, b, (const int8_t *)tile, n_frames, ld_tile, o, ldo, n_cols, accumulate, stream);
        case LTMI_U16: return launch_bell<uint16_t>(m, b, (const uint16_t *)tile, n_frames, ld_tile, o, ldo, n_cols, accumulate, stream);
        case LTMI_I16: return launch_bell<int16_t>(m, b, (const int16_t *)tile, n_frames, ld_tile, o, ldo, n_cols, accumulate, stream);
        case LTMI_F32: return launch_bell<float>(m, b, (const float *)tile, n_frames, ld_tile, o, ldo, n_cols, accumulate, stream);
    }
    *handled = false;
    return LTMI_OK;
}

}  // namespace ltmi
