// Sparse mask stacks on the matrix cores of gfx950 (MI355X): blocked-ELL image + v_mfma_f32_16x16x4_f32.
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
//   * image: for every (pixel chunk of 512 px, group of 16 masks) the sorted list of touched pixels,
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
#include <type_traits>
#include <typeinfo>

namespace ltmi {

typedef float bf32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int bu32x3 __attribute__((ext_vector_type(3)));
typedef __attribute__((address_space(3))) void *b_lds_ptr_t;
typedef const __attribute__((address_space(1))) void *b_glb_ptr_t;

#ifndef BE_P_
#define BE_P_ 512
#endif
#ifndef BE_OCC
#define BE_OCC 1
#endif
constexpr int BE_P = BE_P_;          // pixels per chunk
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

struct BellImage {
    uint32_t *stream = nullptr;      // [blocks][64 lanes][A step 0, A step 1, pixels]
    int64_t *stream_off = nullptr;   // [n_pass * 4 + set] first record of the stream
    int *nblk = nullptr;             // [(active index * 4 + set) * 16 + slot] records of the pair
    int *active = nullptr;           // chunks with entries, concatenated per pass
    int *active_off = nullptr;       // [n_pass + 1]
    int n_pass = 0;
    // entries of the last n_px % 16 pixels of a frame (at most 15): when a row is not a multiple of 16
    // bytes its last 16-byte piece is partial and is not fetched by the frame DMA; these few entries are
    // applied by k_bell_tail after the main kernel
    int32_t *tail_px = nullptr, *tail_col = nullptr;
    float *tail_val = nullptr;
    int n_tail = 0;
    size_t n_blocks = 0;
    double mac_ratio = 0.;           // multiply-adds incl. padding / stored values
};

template <int I, int N, typename F> __device__ __forceinline__ void bstatic_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        bstatic_for<I + 1, N>(f);
    }
}

template <typename T, int TL> struct BeCfg {
    static constexpr int SZ = (int)sizeof(T);
    static constexpr int TILES = TL;                       // 16-frame tiles per workgroup
    static constexpr int FB = 16 * TILES;
    static constexpr int ROW = BE_P * SZ;                  // bytes of a frame's chunk
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
    static_assert(D >= 2, "two slabs + a record ring of depth 2 must fit the LDS");
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
    // One tile per wave (float32 frames, often large detectors): the running sums are handed to a
    // second level every BE_L2 chunks, which keeps the float32 chains short (the registers are there:
    // 16 more tiles).  Two tiles per wave have no room for it.
    constexpr bool TWO_LEVEL = C::SZ == 4;
    constexpr int BE_L2 = 32;
    bf32x4 acc2[TWO_LEVEL ? BE_SLOTS : 1][TWO_LEVEL ? TILES : 1];
#pragma unroll
    for (int s = 0; s < (TWO_LEVEL ? BE_SLOTS : 1); ++s)
#pragma unroll
        for (int t = 0; t < (TWO_LEVEL ? TILES : 1); ++t) acc2[s][t] = bf32x4{0.f, 0.f, 0.f, 0.f};

    // ---- frame DMA: instruction q of the workgroup's chunk copy; wave j issues q = j*NDMA + i
    auto issue_dma = [&](int ai, int buf) {
        if (ablate == 1 || ablate == 3) return;  // timing experiments only (LTMI_BELL_ABLATE)
        const int ch = active[ai];
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int q = j * NDMA + i;
            int64_t fr;
            int64_t byte_in_row;
            int dst;
            if constexpr (C::RPD > 1) {          // several rows per instruction (1-byte pixels)
                constexpr int LPR = 64 / C::RPD;  // lanes per row
                const int r = q * C::RPD + lane / LPR;             // frame of the workgroup
                fr = f0 + r;
                byte_in_row = (int64_t)ch * C::ROW + ((lane % LPR) ^ (BE_PAD ? 0 : (r & 7))) * 16;
                dst = q * C::UNIT_BYTES;
            } else {
                const int r = q / C::DPR;
                fr = f0 + r;
                byte_in_row = (int64_t)ch * C::ROW + (q % C::DPR) * 1024
                              + (lane ^ (BE_PAD ? 0 : (r & 7))) * 16;
                dst = r * C::UNIT_BYTES + (q % C::DPR) * 1024;
            }
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
            since_dma = 0;
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
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BE_D - 1) : "memory");
                BE_STAMP(0);
                const unsigned *rp = (const unsigned *)(ring_lds + slot * SLOT_BYTES) + lane * 4;
                const unsigned x0 = rp[0], x1 = rp[1], o = rp[2];
                const float a_0 = __uint_as_float(x0);
                const float a_1 = __uint_as_float(x1);
                const unsigned char *p0 = bbase + (((o & 0xffffu) * C::SZ) ^ swz);
                const unsigned char *p1 = bbase + (((o >> 16) * C::SZ) ^ swz);
                T b0[TILES], b1[TILES];
#pragma unroll
                for (int t = 0; t < TILES; ++t) {
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

// Build the blocked image from the CSR matrix (n_px x n_masks).  Returns nullptr + error set on
// failure.
void *bell_build(const int64_t *indptr, const int64_t *indices, const float *vals, int nc,
                 int64_t n_px, int64_t n_masks, int *err) {
    *err = LTMI_OK;
    BellImage *b = new (std::nothrow) BellImage();
    if (!b) { *err = LTMI_E_NOMEM; return nullptr; }
    try {
        const int64_t n_cols = n_masks * nc;
        const int64_t n_groups = (n_cols + 15) / 16;
        const int n_pass = (int)((n_groups + BE_SETS * BE_SLOTS - 1) / (BE_SETS * BE_SLOTS));
        const int n_chunks = (int)((n_px + BE_P - 1) / BE_P);
        b->n_pass = n_pass;
        struct Ent { uint16_t px; uint16_t m; float v; };
        // entries per (chunk, group), pixels ascending because the CSR rows are walked in order
        std::vector<std::vector<Ent>> bucket((size_t)n_chunks * n_groups);
        const int64_t p_tail0 = n_px - n_px % 16;       // pixels from here on: k_bell_tail
        std::vector<int32_t> tail_px, tail_col;
        std::vector<float> tail_val;
        for (int64_t p = 0; p < n_px; ++p) {
            const int ch = (int)(p / BE_P);
            for (int64_t e = indptr[p]; e < indptr[p + 1]; ++e)
                for (int c = 0; c < nc; ++c) {
                    const int64_t col = indices[e] * nc + c;
                    if (p >= p_tail0) {
                        tail_px.push_back((int32_t)p);
                        tail_col.push_back((int32_t)col);
                        tail_val.push_back(vals[e * nc + c]);
                        continue;
                    }
                    bucket[(size_t)ch * n_groups + col / 16].push_back(
                        Ent{(uint16_t)(p - (int64_t)ch * BE_P), (uint16_t)(col % 16), vals[e * nc + c]});
                }
        }
        b->n_tail = (int)tail_px.size();
        std::vector<int> active, active_off(n_pass + 1, 0);
        for (int ps = 0; ps < n_pass; ++ps) {
            const int64_t g_lo = (int64_t)ps * BE_SETS * BE_SLOTS;
            const int64_t g_hi = std::min<int64_t>(n_groups, g_lo + BE_SETS * BE_SLOTS);
            for (int ch = 0; ch < n_chunks; ++ch) {
                bool any = false;
                for (int64_t g = g_lo; g < g_hi && !any; ++g)
                    any = !bucket[(size_t)ch * n_groups + g].empty();
                if (any) active.push_back(ch);
            }
            active_off[ps + 1] = (int)active.size();
        }
        const size_t n_act = std::max<size_t>(active.size(), 1);
        std::vector<int> nblk(n_act * BE_SETS * BE_SLOTS, 0);
        std::vector<int64_t> stream_off((size_t)n_pass * BE_SETS, 0);
        std::vector<uint32_t> stream;
        size_t blocks = 0;
        std::vector<uint16_t> cols;
        for (int ps = 0; ps < n_pass; ++ps)
            for (int j = 0; j < BE_SETS; ++j) {
                stream_off[(size_t)ps * BE_SETS + j] = (int64_t)blocks;
                for (int ai = active_off[ps]; ai < active_off[ps + 1]; ++ai) {
                    const int ch = active[ai];
                    for (int s = 0; s < BE_SLOTS; ++s) {
                        const int64_t g = (int64_t)ps * BE_SETS * BE_SLOTS + s * BE_SETS + j;
                        if (g >= n_groups) continue;
                        const std::vector<Ent> &ents = bucket[(size_t)ch * n_groups + g];
                        if (ents.empty()) continue;
                        cols.clear();
                        for (const Ent &en : ents)
                            if (cols.empty() || cols.back() != en.px) cols.push_back(en.px);
                        const int nb = ((int)cols.size() + 7) / 8;
                        nblk[((size_t)ai * BE_SETS + j) * BE_SLOTS + s] = nb;
                        const size_t base = stream.size();
                        stream.resize(base + (size_t)nb * BE_REC, 0u);
                        // pixel numbers: lane l -> columns 8*blk + (l >> 4) and + 4
                        for (int blk = 0; blk < nb; ++blk)
                            for (int l = 0; l < 64; ++l) {
                                const int c0 = blk * 8 + (l >> 4), c1 = c0 + 4;
                                // padding columns (value 0) repeat the block's first pixel rather
                                // than pixel 0 of the chunk: a non-finite pixel then only reaches
                                // blocks that really contain it (0 * NaN = NaN)
                                const uint32_t p0 = c0 < (int)cols.size() ? cols[c0] : cols[blk * 8];
                                const uint32_t p1 = c1 < (int)cols.size() ? cols[c1] : cols[blk * 8];
                                stream[base + (size_t)blk * BE_REC + l * 3 + 2] = p0 | (p1 << 16);
                            }
                        size_t ci = 0;
                        for (const Ent &en : ents) {
                            while (cols[ci] != en.px) ++ci;
                            const int blk = (int)(ci / 8), step = (int)((ci % 8) / 4), kk = (int)(ci % 4);
                            const int l = kk * 16 + en.m;
                            uint32_t bits;
                            memcpy(&bits, &en.v, 4);
                            stream[base + (size_t)blk * BE_REC + l * 3 + step] = bits;
                        }
                        blocks += nb;
                    }
                }
                // slack for the run-ahead loads of the last records
                stream.resize(stream.size() + (size_t)BE_D_MAX * BE_REC, 0u);
                blocks += BE_D_MAX;
            }
        b->n_blocks = blocks;
        int64_t nnz = indptr[n_px] * nc;
        b->mac_ratio = nnz > 0 ? (double)(blocks - (size_t)BE_D_MAX * n_pass * BE_SETS) * 128.0 / (double)nnz : 0.;
        if (active.empty()) active.push_back(0);
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

// the entries of the last n_px % 16 pixels (see BellImage::tail_*): one thread per frame, entries in
// CSR order (a frame's sums are updated by one thread only: no atomics)
template <typename T>
__global__ void k_bell_tail(const T *__restrict__ tile, int64_t ld, int64_t n_frames,
                            const int32_t *__restrict__ px, const int32_t *__restrict__ col,
                            const float *__restrict__ val, int n_tail, float *__restrict__ out,
                            int64_t ld_out, const int32_t *__restrict__ rows) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    const T *row = tile + (rows ? (int64_t)rows[f] : f) * ld;
    float *o = out + f * ld_out;
    for (int e = 0; e < n_tail; ++e) o[col[e]] += val[e] * (float)row[px[e]];
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
        hipLaunchKernelGGL(k_bell_tail<T>, dim3((unsigned)((n_frames + 255) / 256)), dim3(256), 0,
                           stream, tile, ld, n_frames, (const int32_t *)b->tail_px,
                           (const int32_t *)b->tail_col, (const float *)b->tail_val, b->n_tail, out,
                           ld_out_f, m->roi_rows);
        LTMI_HIP(hipGetLastError());
    }
    snprintf(m->last_kernel, sizeof(m->last_kernel),
             "k_bell_apply<%s,tiles=%d%s> grid=(%u,%u) blocks=%zu x%.2f", typeid(T).name(), TL,
             m->roi_rows ? ",rows" : "", grid.x, grid.y, b->n_blocks, b->mac_ratio);
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
    if (hi)
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
