// Sparse mask stacks (CSR) on the vector ALUs of gfx950 (MI355X): k_scatter -- one float32 FMA per stored
// entry, exactly the arithmetic of the reference's numba kernels _rmatmul_csr / _rmatmul_csc
// (src/libertem/common/numba/__init__.py:90-184: `out[row, col] += left * value` per stored entry, one
// float32 chain per (frame, mask) in pixel order), for every pixel type the dense kernels take
// (u8 / i8 / u16 / i16 / f32).  It replaces the float16-piece matrix-core kernel of round 3 (k_bell_flat),
// whose weights carried 22 bits and lost small entries of a column.
//
// Layout of the work.  A workgroup = 16 waves = one CU, 64 frames (lane = frame) x one PASS of up to 1024
// mask columns.  Wave j owns 64 columns of the pass -- two RANGES of 32 consecutive columns (narrow stacks:
// more, shorter ranges), assigned by
// the image builder so that the waves' totals are level -- as 64 accumulator REGISTERS v[56:119] per lane.
// The stored entries of a pixel that fall into one wave are cut into BUNDLES: a pixel, an even accumulator
// slot s0 and 8 weights for the slots s0 .. s0 + 7 (zeros where the mask has no entry; radial-bin stacks
// fill 5.7 of 8).  Processing a bundle is
//      ds_read_u16  x <- LDS[lane's frame row][pixel]        (the frames' chunk lies in LDS, row pitch
//      v_cvt_f32_u32 x                                        1 KiB + 4 B: lanes hit different banks)
//      s_set_gpr_idx_on  s0                                  (VGPR index mode: M0[7:0] = accumulator slot)
//      4 x v_pk_fma_f32  acc[s0 + 2k : +1] += x * (w[2k], w[2k + 1])    destination / src2 RELATIVE to M0
// i.e. 6 vector instructions per 8 slots x 64 frames; the weights are SGPR pairs, fetched with the header
// words through the SCALAR cache (s_load_dwordx16) -- the vector memory path carries nothing but the frame
// copies (global_load_lds_dwordx4, double buffered), which is what k_bell_flat's record stream fought with.
// The accumulator a bundle feeds is data: only the index mode can address it (probes/gpridx_probe.hip: the
// mode applies to the packed VOP3P form, even slots only; costs nothing against static registers).
//
// The loop itself (software pipelined over two register sets, every register named by hand) is generated:
// scripts/gen_scatter_asm.py -> ltmi_scatter_loop.inc.  REGISTER CONTRACT with that file:
//   compiler: v0..v31, s0..s11 (+ vcc); loop: v32..v127, s12..s99, m0, scc.
//   v32 argument lanes | v33 / v38 lane * PITCH + offset of the buffer being read / the other one | v34 lane's
//   byte offset of its 16-byte piece in the frame row (chunk table) | v35 lane * 4 | v36 / v37 row pointers
//   (lanes 0..3) | v39 temp | v40..v55 pixel values + address temps of the two sets | v56..v119 accumulators,
//   v120..v127 padding slots (dummy bundles, windows that reach beyond slot 63).
//   s12..s19 header words of the two sets (bits 0-7 accumulator slot, bit 8 of word 3: end of the wave's work on
//   the chunk, bits 16-31 LDS row offset of the NEXT block's pixel) | s20..s25 stream / table bases |
//   s26..s29 temps | s30 / s31 stream offsets | s32 blocks left | s33 / s34 chunk index, end | s35 buffer |
//   s36..s99 weights of the two sets.
// A chunk = 1 KiB of every frame row = 8 SEGMENTS of 128 bytes which the builder composes (ChunkMixer) so that
// the waves of the workgroup -- they meet at a barrier per chunk -- carry equal work: in a ring stack the ring
// that is tangent to a detector row puts ~70 pixels of that row into ONE 32-column range.
#include "ltmi_common.h"
#include "ltmi_scatter_loop.inc"
#include <cstring>
#include <cmath>
#include <new>
#include <numeric>
#include <type_traits>
#include <typeinfo>

namespace ltmi {

constexpr int SC_WAVES = 16, SC_SLOTS = 64;                         // waves, accumulators per wave
constexpr int SC_PASS = SC_WAVES * SC_SLOTS;                        // columns per pass
constexpr int SC_FB = 64;                                           // frames per workgroup
constexpr int SC_ROW = 1024, SC_SEG = 128, SC_NSEG = SC_ROW / SC_SEG;   // bytes of a row per chunk / per segment
constexpr int SC_PAD_SLOT = 64;                                     // accumulator slot of dummy bundles (v120..)
constexpr unsigned SC_END = 1u << 8;
// columns per RANGE (the unit that is assigned to a wave): 32 for a full pass, fewer for narrow stacks so that
// every wave gets columns (64 columns: 32 ranges of 2)
static inline int range_size(int64_t cols_in_pass) {
    int rs = 2;
    while (rs < 32 && (cols_in_pass + rs - 1) / rs > 2 * SC_WAVES) rs *= 2;
    return rs;
}

constexpr int SC_EPI = SC_WAVES * 64 * 33 * 4;                          // epilogue: 64 x 33 words per wave
constexpr int SC_LDS = 2 * SCAT_BUF > SC_EPI ? 2 * SCAT_BUF : SC_EPI;
static_assert(SCAT_PITCH == SC_ROW + 4 && SCAT_BUF == SC_FB * SCAT_PITCH && SCAT_ACC0 == 56, "generated loop");

struct ScatImage {                       // the image of a stack for ONE pixel size
    int sz = 0, n_pass = 0;
    uint32_t *hdr = nullptr;             // 4 words per block
    float *wts = nullptr;                // 32 weights per block
    int64_t *stream_off = nullptr;       // [n_pass * 16 + 1] first block of a wave's stream
    int32_t *n_blk = nullptr;            // [n_pass * 16] blocks of the stream (incl. the leading dummy)
    int32_t *dma_off = nullptr;          // [chunks of all passes][64] byte offset in the frame row
    int32_t *active_off = nullptr;       // [n_pass + 1]
    int32_t *col_of_slot = nullptr;      // [n_pass * 16 * 64] column of an accumulator slot, -1: none
    int32_t *tail_px = nullptr, *tail_col = nullptr;   // entries of pixels behind the last full 16-byte piece
    float *tail_val = nullptr;
    int n_tail = 0;
    size_t n_blocks = 0, n_bundles = 0;
    long crit_blocks = 0;                // sum over chunks of the busiest wave's blocks
    double fill = 0;                     // stored entries per bundle slot
};

struct ScatSet {                         // host copy of the CSR matrix + the images built so far
    int64_t n_px = 0, n_masks = 0;
    int nc = 1;
    std::vector<int64_t> indptr, indices;
    std::vector<float> vals;
    ScatImage *img[3] = {nullptr, nullptr, nullptr};     // pixel size 1, 2, 4
    bool failed[3] = {false, false, false};
};

static void image_destroy(ScatImage *b) {
    if (!b) return;
    void *p[] = {b->hdr, b->wts, b->stream_off, b->n_blk, b->dma_off, b->active_off, b->col_of_slot,
                 b->tail_px, b->tail_col, b->tail_val};
    for (void *q : p)
        if (q) (void)hipFree(q);
    delete b;
}

void scat_destroy(void *set) {
    ScatSet *s = (ScatSet *)set;
    if (!s) return;
    for (ScatImage *b : s->img) image_destroy(b);
    delete s;
}

// ---- kernel -------------------------------------------------------------------------------------------
#define SC_V32_127 "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"
#define SC_S12_99 "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99"

// 8 accumulators v[A .. A + 7] into compiler-visible values
#define SC_READ8(A0, A1, A2, A3, A4, A5, A6, A7, R)                                                              \
    asm volatile("v_mov_b32 %0, v" #A0 "\n\tv_mov_b32 %1, v" #A1 "\n\tv_mov_b32 %2, v" #A2 "\n\tv_mov_b32 %3, v" #A3     \
                 "\n\tv_mov_b32 %4, v" #A4 "\n\tv_mov_b32 %5, v" #A5 "\n\tv_mov_b32 %6, v" #A6 "\n\tv_mov_b32 %7, v" #A7 \
                 : "=v"(R[0]), "=v"(R[1]), "=v"(R[2]), "=v"(R[3]), "=v"(R[4]), "=v"(R[5]), "=v"(R[6]), "=v"(R[7]))

// ABL > 0: timing-only variants of the uint16 loop (LTMI_SCATTER_ABLATE: 1 weights from one hot line, 2 no
// frame copies, 3 no LDS reads, 4 no FMAs, 5 = 1 + 2, 6 no chunk barriers, 7 = 1 + 6; results are garbage)
template <typename T, int ABL = 0>
__global__ void __launch_bounds__(SC_WAVES * 64, 1) __attribute__((amdgpu_num_vgpr(32)))
k_scatter(const T *__restrict__ tile, int64_t ld, int64_t n_frames, const uint32_t *__restrict__ hdr,
          const float *__restrict__ wts, const int64_t *__restrict__ stream_off,
          const int32_t *__restrict__ n_blk, const int32_t *__restrict__ dma_off,
          const int32_t *__restrict__ active_off, const int32_t *__restrict__ col_of_slot,
          float *__restrict__ out, int64_t ld_out, int n_cols, int accumulate,
          const int32_t *__restrict__ rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sc_lds[];      // at LDS address 0 (the loop assumes it)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int j = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pass = blockIdx.y;
    const int64_t f0 = (int64_t)blockIdx.x * SC_FB;
    const int a0 = active_off[pass], a1 = active_off[pass + 1];
    const int wj = pass * SC_WAVES + j;
    {
        const uint64_t hp = (uint64_t)(hdr + stream_off[wj] * 4);
        const uint64_t wp = (uint64_t)(wts + stream_off[wj] * 32);
        const uint64_t tp = (uint64_t)dma_off;
        const unsigned av[10] = {(unsigned)hp, (unsigned)(hp >> 32), (unsigned)wp, (unsigned)(wp >> 32),
                                 (unsigned)tp, (unsigned)(tp >> 32), (unsigned)n_blk[wj], (unsigned)a0,
                                 (unsigned)a1, (unsigned)(4 * j * SCAT_PITCH)};
        unsigned args = 0;
#pragma unroll
        for (int i = 0; i < 10; ++i) args = lane == i ? av[i] : args;
        // rows 4 j .. 4 j + 3 of the workgroup's frames (beyond the last frame: the last one again)
        int64_t fr = f0 + 4 * j + (lane & 3);
        if (fr > n_frames - 1) fr = n_frames - 1;
        if (rows) fr = rows[fr];
        const uint64_t rp = (uint64_t)(tile + fr * ld);
        const unsigned rowlo = (unsigned)rp, rowhi = (unsigned)(rp >> 32);
        const unsigned lanebase = (unsigned)lane * SCAT_PITCH, lane4 = (unsigned)lane * 4u;
        if (a0 < a1) {
#define SC_RUN(NAME)                                                                                  \
    asm volatile(SCAT_LOOP_##NAME ::"v"(args), "v"(lanebase), "v"(lane4), "v"(rowlo), "v"(rowhi)       \
                 : "memory", "scc", SC_V32_127, SC_S12_99)
            if constexpr (ABL == 1) SC_RUN(u16_a1);
            else if constexpr (ABL == 2) SC_RUN(u16_a2);
            else if constexpr (ABL == 3) SC_RUN(u16_a3);
            else if constexpr (ABL == 4) SC_RUN(u16_a4);
            else if constexpr (ABL == 5) SC_RUN(u16_a5);
            else if constexpr (ABL == 6) SC_RUN(u16_a6);
            else if constexpr (ABL == 7) SC_RUN(u16_a7);
            else if constexpr (std::is_same<T, uint8_t>::value) SC_RUN(u8);
            else if constexpr (std::is_same<T, int8_t>::value) SC_RUN(i8);
            else if constexpr (std::is_same<T, uint16_t>::value) SC_RUN(u16);
            else if constexpr (std::is_same<T, int16_t>::value) SC_RUN(i16);
            else SC_RUN(f32);
#undef SC_RUN
        } else {
            asm volatile("v_mov_b32 v56, 0\n\tv_mov_b32 v57, 0\n\tv_mov_b32 v58, 0\n\tv_mov_b32 v59, 0\n\tv_mov_b32 v60, 0\n\tv_mov_b32 v61, 0\n\tv_mov_b32 v62, 0\n\tv_mov_b32 v63, 0\n\t"
                         "v_mov_b32 v64, 0\n\tv_mov_b32 v65, 0\n\tv_mov_b32 v66, 0\n\tv_mov_b32 v67, 0\n\tv_mov_b32 v68, 0\n\tv_mov_b32 v69, 0\n\tv_mov_b32 v70, 0\n\tv_mov_b32 v71, 0\n\t"
                         "v_mov_b32 v72, 0\n\tv_mov_b32 v73, 0\n\tv_mov_b32 v74, 0\n\tv_mov_b32 v75, 0\n\tv_mov_b32 v76, 0\n\tv_mov_b32 v77, 0\n\tv_mov_b32 v78, 0\n\tv_mov_b32 v79, 0\n\t"
                         "v_mov_b32 v80, 0\n\tv_mov_b32 v81, 0\n\tv_mov_b32 v82, 0\n\tv_mov_b32 v83, 0\n\tv_mov_b32 v84, 0\n\tv_mov_b32 v85, 0\n\tv_mov_b32 v86, 0\n\tv_mov_b32 v87, 0\n\t"
                         "v_mov_b32 v88, 0\n\tv_mov_b32 v89, 0\n\tv_mov_b32 v90, 0\n\tv_mov_b32 v91, 0\n\tv_mov_b32 v92, 0\n\tv_mov_b32 v93, 0\n\tv_mov_b32 v94, 0\n\tv_mov_b32 v95, 0\n\t"
                         "v_mov_b32 v96, 0\n\tv_mov_b32 v97, 0\n\tv_mov_b32 v98, 0\n\tv_mov_b32 v99, 0\n\tv_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"
                         "v_mov_b32 v104, 0\n\tv_mov_b32 v105, 0\n\tv_mov_b32 v106, 0\n\tv_mov_b32 v107, 0\n\tv_mov_b32 v108, 0\n\tv_mov_b32 v109, 0\n\tv_mov_b32 v110, 0\n\tv_mov_b32 v111, 0\n\t"
                         "v_mov_b32 v112, 0\n\tv_mov_b32 v113, 0\n\tv_mov_b32 v114, 0\n\tv_mov_b32 v115, 0\n\tv_mov_b32 v116, 0\n\tv_mov_b32 v117, 0\n\tv_mov_b32 v118, 0\n\tv_mov_b32 v119, 0"
                         ::: SC_V32_127);
        }
    }
    // ---- results: the wave's 64 frames x 64 slots, transposed through LDS (rows of 33 words) so that a
    // store instruction writes two frames' 32 consecutive columns
    __syncthreads();                                     // every wave has left the frame buffers
    float *reg = (float *)(sc_lds + (size_t)j * (64 * 33 * 4));
    const int row_l = lane >> 5, sl = lane & 31;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        float r[8];
#define SC_PUT(S0)                                                                                    \
    _Pragma("unroll") for (int q = 0; q < 8; ++q) reg[lane * 33 + (S0) + q] = r[q];
        if (half == 0) {
            SC_READ8(56, 57, 58, 59, 60, 61, 62, 63, r); SC_PUT(0)
            SC_READ8(64, 65, 66, 67, 68, 69, 70, 71, r); SC_PUT(8)
            SC_READ8(72, 73, 74, 75, 76, 77, 78, 79, r); SC_PUT(16)
            SC_READ8(80, 81, 82, 83, 84, 85, 86, 87, r); SC_PUT(24)
        } else {
            SC_READ8(88, 89, 90, 91, 92, 93, 94, 95, r); SC_PUT(0)
            SC_READ8(96, 97, 98, 99, 100, 101, 102, 103, r); SC_PUT(8)
            SC_READ8(104, 105, 106, 107, 108, 109, 110, 111, r); SC_PUT(16)
            SC_READ8(112, 113, 114, 115, 116, 117, 118, 119, r); SC_PUT(24)
        }
#undef SC_PUT
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int col = col_of_slot[(size_t)wj * SC_SLOTS + half * 32 + sl];
#pragma unroll 2
        for (int it = 0; it < 32; ++it) {
            const int row = it * 2 + row_l;
            const float v = reg[row * 33 + sl];
            const int64_t f = f0 + row;
            if (col >= 0 && col < n_cols && f < n_frames) {
                float *p = out + f * ld_out + col;
                *p = accumulate ? *p + v : v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// the entries of the pixels behind a row's last full 16-byte piece (the frame copies do not fetch it): one
// thread per frame, entries in CSR order
template <typename T>
__global__ void k_scatter_tail(const T *__restrict__ tile, int64_t ld, int64_t n_frames,
                               const int32_t *__restrict__ px, const int32_t *__restrict__ col,
                               const float *__restrict__ val, int n_tail, float *__restrict__ out,
                               int64_t ld_out, const int32_t *__restrict__ rows) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    const T *row = tile + (rows ? (int64_t)rows[f] : f) * ld;
    float *o = out + f * ld_out;
    for (int e = 0; e < n_tail; ++e) o[col[e]] += val[e] * (float)row[px[e]];
}

// ---- chunk composition --------------------------------------------------------------------------------
// cnt[seg][wave] = bundles of segment `seg` for the waves of a pass.  Chunks of 8 segments; a chunk costs
// what its busiest wave needs, in blocks of 4 bundles.  Deterministic annealing over swaps of segments
// between chunks (fixed seed), started from the better of the natural order and a strided interleave.
struct ChunkMixer {
    const std::vector<uint16_t> &cnt;
    int unit;                                      // segments that move together (contiguous bytes of a row: 128 * unit)
    explicit ChunkMixer(const std::vector<uint16_t> &c, int u = 1) : cnt(c), unit(u) {}
    void cost(const int *segs, int *crit, int *total) const {
        int load[SC_WAVES] = {0};
        for (int i = 0; i < SC_NSEG; ++i) {
            if (segs[i] < 0) continue;
            const uint16_t *row = cnt.data() + (size_t)segs[i] * SC_WAVES;
            for (int w = 0; w < SC_WAVES; ++w) load[w] += row[w];
        }
        int mx = 0, tot = 0;
        for (int w = 0; w < SC_WAVES; ++w) {
            const int b = (load[w] + 3) / 4;
            mx = std::max(mx, b);
            tot += std::max(b, 1);
        }
        *crit = mx;
        *total = tot;
    }
    long mix(std::vector<int> &segs) const {
        const int n_chunks = (int)(segs.size() / SC_NSEG);
        auto eval = [&](const std::vector<int> &a) {
            double c = 0;
            for (int k = 0; k < n_chunks; ++k) {
                int x, y;
                cost(a.data() + (size_t)k * SC_NSEG, &x, &y);
                c += x + 0.04 * y;
            }
            return c;
        };
        // (segs comes in natural order: consecutive entries are consecutive segments; a unit = `unit` of them)
        const int upc = SC_NSEG / unit;                       // units per chunk
        std::vector<int> inter(segs.size(), -1);
        {
            std::vector<int> fill((size_t)n_chunks, 0);
            const size_t n_units = segs.size() / unit;
            int k = 0;
            for (size_t u = 0; u < n_units; ++u) {
                bool any = false;
                for (int q = 0; q < unit; ++q) any |= segs[u * unit + q] >= 0;
                if (!any) continue;
                const int c = k % n_chunks;
                for (int q = 0; q < unit; ++q) inter[((size_t)c * upc + fill[c]) * unit + q] = segs[u * unit + q];
                ++fill[c];
                ++k;
            }
        }
        if (eval(inter) < eval(segs)) segs = inter;
        std::vector<int> cc((size_t)n_chunks), ct((size_t)n_chunks);
        for (int k = 0; k < n_chunks; ++k) cost(segs.data() + (size_t)k * SC_NSEG, &cc[k], &ct[k]);
        if (n_chunks >= 2) {
            uint64_t rng = 0x2545F4914F6CDD1Dull;
            auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
            const long iters = std::min<long>(600000, 200L * (long)segs.size() + 2000);
            for (long it = 0; it < iters; ++it) {
                const int A = (int)(next() % n_chunks), B = (int)(next() % n_chunks);
                if (A == B) continue;
                const size_t ia = ((size_t)A * upc + next() % upc) * unit, ib = ((size_t)B * upc + next() % upc) * unit;
                auto swap_units = [&]() { for (int q = 0; q < unit; ++q) std::swap(segs[ia + q], segs[ib + q]); };
                swap_units();
                int ca, ta, cb, tb;
                cost(segs.data() + (size_t)A * SC_NSEG, &ca, &ta);
                cost(segs.data() + (size_t)B * SC_NSEG, &cb, &tb);
                const double d = (ca + cb - cc[A] - cc[B]) + 0.04 * (ta + tb - ct[A] - ct[B]);
                const double temp = std::max(0.02, 0.8 * (1.0 - (double)it / (double)iters));
                const double u = (double)(next() >> 11) * (1.0 / 9007199254740992.0);
                if (d <= 0 || u < std::exp(-d / temp)) {
                    cc[A] = ca; ct[A] = ta; cc[B] = cb; ct[B] = tb;
                } else {
                    swap_units();
                }
            }
        }
        long c = 0;
        for (int k = 0; k < n_chunks; ++k) c += cc[k];
        return c;
    }
};

// ---- image builder ------------------------------------------------------------------------------------
struct Bundle {
    int32_t px;
    uint8_t slot;
    float w[8];
};

template <typename V> static hipError_t upload(V **dst, const std::vector<V> &src) {
    hipError_t e = hipMalloc((void **)dst, std::max<size_t>(src.size(), 1) * sizeof(V));
    if (e == hipSuccess && !src.empty())
        e = hipMemcpy(*dst, src.data(), src.size() * sizeof(V), hipMemcpyHostToDevice);
    return e;
}

static ScatImage *build_image(const ScatSet &s, int sz, int *err) {
    *err = LTMI_OK;
    ScatImage *b = new (std::nothrow) ScatImage();
    if (!b) {
        *err = LTMI_E_NOMEM;
        return nullptr;
    }
    try {
        const int nc = s.nc;
        const int64_t n_px = s.n_px, n_cols = s.n_masks * nc;
        const int px_seg = SC_SEG / sz, px_piece = 16 / sz;
        const int64_t n_px_dma = n_px - n_px % px_piece;          // pixels the 16-byte pieces reach
        const int n_seg = (int)((n_px_dma + px_seg - 1) / px_seg);
        b->sz = sz;
        b->n_pass = (int)((n_cols + SC_PASS - 1) / SC_PASS);
        std::vector<int32_t> tail_px, tail_col;
        std::vector<float> tail_val;
        for (int64_t p = n_px_dma; p < n_px; ++p)
            for (int64_t e = s.indptr[p]; e < s.indptr[p + 1]; ++e)
                for (int c = 0; c < nc; ++c) {
                    tail_px.push_back((int32_t)p);
                    tail_col.push_back((int32_t)(s.indices[e] * nc + c));
                    tail_val.push_back(s.vals[e * nc + c]);
                }
        b->n_tail = (int)tail_val.size();

        std::vector<uint32_t> hdr;
        std::vector<float> wts;
        std::vector<int64_t> stream_off((size_t)b->n_pass * SC_WAVES + 1, 0);
        std::vector<int32_t> n_blk((size_t)b->n_pass * SC_WAVES, 0);
        std::vector<int32_t> dma_off, active_off((size_t)b->n_pass + 1, 0);
        std::vector<int32_t> col_of_slot((size_t)b->n_pass * SC_WAVES * SC_SLOTS, -1);
        size_t stored = 0;

        for (int ps = 0; ps < b->n_pass; ++ps) {
            const int64_t c_lo = (int64_t)ps * SC_PASS, c_hi = std::min<int64_t>(n_cols, c_lo + SC_PASS);
            const int SC_RS = range_size(c_hi - c_lo);
            const int n_rng = (int)((c_hi - c_lo + SC_RS - 1) / SC_RS);
            // a range occupies a CELL of max(RS, 8) accumulator slots; a bundle's window of 8 slots never leaves
            // the cell of its range, so the zero weights of a window only ever meet columns of the same range
            // (a non-finite pixel reaches at most the 7 neighbouring columns of a mask that holds it) or slots
            // no column uses
            const int cell = std::max(SC_RS, 8);
            // (1) weight of a range = its bundles; ranges to (wave, half): heaviest first onto the lightest
            //     wave that has a half left
            auto bundles_of_pixel = [&](int64_t p, auto &&emit) {
                // the pixel's columns inside the pass, in column order (CSR rows hold ascending mask numbers
                // for stacks made by MaskContainer; sort to be safe)
                int cols[SC_PASS];
                float vals[SC_PASS];
                int n = 0;
                for (int64_t e = s.indptr[p]; e < s.indptr[p + 1]; ++e)
                    for (int c = 0; c < nc; ++c) {
                        const int64_t col = s.indices[e] * nc + c;
                        if (col < c_lo || col >= c_hi) continue;
                        if (n == SC_PASS) break;
                        cols[n] = (int)(col - c_lo);
                        vals[n++] = s.vals[e * nc + c];
                    }
                if (n == 0) return;
                for (int i = 1; i < n; ++i)                       // insertion sort (almost always sorted)
                    for (int k = i; k > 0 && cols[k - 1] > cols[k]; --k) {
                        std::swap(cols[k - 1], cols[k]);
                        std::swap(vals[k - 1], vals[k]);
                    }
                emit(cols, vals, n);
            };
            std::vector<long> rng_w((size_t)n_rng, 0);
            for (int64_t p = 0; p < n_px_dma; ++p)
                bundles_of_pixel(p, [&](const int *cols, const float *, int n) {
                    int i = 0;
                    while (i < n) {
                        const int r = cols[i] / SC_RS, s0 = (cols[i] % SC_RS) & ~1;
                        ++rng_w[(size_t)r];
                        while (i < n && cols[i] / SC_RS == r && cols[i] % SC_RS < s0 + 8) ++i;
                    }
                });
            std::vector<int> order((size_t)n_rng);
            std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return rng_w[(size_t)x] > rng_w[(size_t)y]; });
            long wave_w[SC_WAVES] = {0};
            int wave_n[SC_WAVES] = {0};
            std::vector<int> rng_wave((size_t)n_rng, 0), rng_base((size_t)n_rng, 0);
            for (int r : order) {
                int best = -1;
                for (int w = 0; w < SC_WAVES; ++w)
                    if (wave_n[w] < SC_SLOTS / cell && (best < 0 || wave_w[w] < wave_w[best])) best = w;
                rng_wave[(size_t)r] = best;
                rng_base[(size_t)r] = wave_n[best] * cell;
                ++wave_n[best];
                wave_w[best] += rng_w[(size_t)r];
            }
            for (int r = 0; r < n_rng; ++r)
                for (int q = 0; q < SC_RS; ++q) {
                    const int64_t col = c_lo + (int64_t)r * SC_RS + q;
                    if (col < c_hi)
                        col_of_slot[((size_t)ps * SC_WAVES + rng_wave[(size_t)r]) * SC_SLOTS + rng_base[(size_t)r] + q] = (int32_t)col;
                }
            // (2) bundles per (wave, segment)
            std::vector<std::vector<Bundle>> per_seg((size_t)n_seg * SC_WAVES);
            std::vector<uint16_t> cnt((size_t)n_seg * SC_WAVES, 0);
            for (int64_t p = 0; p < n_px_dma; ++p)
                bundles_of_pixel(p, [&](const int *cols, const float *vals, int n) {
                    const int seg = (int)(p / px_seg);
                    // the pixel's entries as (wave, slot), sorted; per range: windows of 8 slots from an even slot,
                    // kept inside the range's cell
                    int key[SC_PASS];
                    for (int i = 0; i < n; ++i) {
                        const int r = cols[i] / SC_RS;
                        key[i] = (rng_wave[(size_t)r] << 20) | ((rng_base[(size_t)r] + cols[i] % SC_RS) << 12) | i;
                    }
                    std::sort(key, key + n);
                    int i = 0;
                    while (i < n) {
                        const int w = key[i] >> 20, slot = (key[i] >> 12) & 0xff;
                        const int cell_lo = slot / cell * cell;
                        const int s0 = std::min(slot & ~1, cell_lo + cell - 8);
                        Bundle bd;
                        bd.px = (int32_t)p;
                        bd.slot = (uint8_t)s0;
                        for (float &x : bd.w) x = 0.f;
                        while (i < n && (key[i] >> 20) == w && ((key[i] >> 12) & 0xff) < s0 + 8 &&
                               ((key[i] >> 12) & 0xff) < cell_lo + cell) {
                            bd.w[((key[i] >> 12) & 0xff) - s0] += vals[key[i] & 0xfff];      // (+=: duplicate entries add up)
                            ++stored;
                            ++i;
                        }
                        per_seg[(size_t)seg * SC_WAVES + w].push_back(bd);
                        if (cnt[(size_t)seg * SC_WAVES + w] < 65535) ++cnt[(size_t)seg * SC_WAVES + w];
                    }
                });
            // (3) chunks: the segments that hold anything, 8 per chunk, composed for level waves
            static const bool natural = getenv("LTMI_SCATTER_NATURAL") != nullptr;      // (bench: no mixing)
            static const int unit = std::min(SC_NSEG, getenv("LTMI_SCATTER_SEG") ? std::max(1, atoi(getenv("LTMI_SCATTER_SEG")) / SC_SEG) : 1);
            std::vector<int> segs;                 // groups of `unit` adjacent segments (contiguous bytes of a row)
            for (int g0 = 0; g0 < n_seg; g0 += unit) {
                bool any = false;
                for (int g = g0; g < std::min(n_seg, g0 + unit); ++g)
                    for (int w = 0; w < SC_WAVES; ++w) any |= cnt[(size_t)g * SC_WAVES + w] != 0;
                if (any)
                    for (int g = g0; g < g0 + unit; ++g) segs.push_back(g < n_seg ? g : -1);
            }
            const int n_chunks = (int)((segs.size() + SC_NSEG - 1) / SC_NSEG);
            segs.resize((size_t)n_chunks * SC_NSEG, -1);
            if (!natural && n_chunks > 0) b->crit_blocks += ChunkMixer(cnt, unit).mix(segs);
            active_off[(size_t)ps + 1] = active_off[(size_t)ps] + n_chunks;
            for (int k = 0; k < n_chunks; ++k)
                for (int l = 0; l < 64; ++l) {
                    const int g = segs[(size_t)k * SC_NSEG + l / 8];
                    int64_t off = g < 0 ? 0 : (int64_t)g * SC_SEG + (l % 8) * 16;
                    if (off + 16 > n_px * sz) off = 0;
                    dma_off.push_back((int32_t)off);
                }
            // (4) the waves' streams
            for (int w = 0; w < SC_WAVES; ++w) {
                const size_t first = hdr.size() / 4;
                stream_off[(size_t)ps * SC_WAVES + w] = (int64_t)first;
                std::vector<uint32_t> own;          // per bundle: slot | flags << 8 | own row offset << 16
                auto push = [&](const Bundle *bd, uint32_t lds_off, uint32_t flags) {
                    own.push_back((bd ? bd->slot : (uint32_t)SC_PAD_SLOT) | flags | (lds_off << 16));
                    for (int q = 0; q < 8; ++q) wts.push_back(bd ? bd->w[q] : 0.f);
                };
                for (int q = 0; q < 4; ++q) push(nullptr, 0, 0);                          // leading dummy block
                for (int k = 0; k < n_chunks; ++k) {
                    size_t in_chunk = 0;
                    for (int q = 0; q < SC_NSEG; ++q) {
                        const int g = segs[(size_t)k * SC_NSEG + q];
                        if (g < 0) continue;
                        for (const Bundle &bd : per_seg[(size_t)g * SC_WAVES + w]) {
                            const uint32_t off = (uint32_t)(q * SC_SEG + (bd.px - (int64_t)g * px_seg) * sz);
                            push(&bd, off, 0);
                            ++in_chunk;
                            ++b->n_bundles;
                        }
                    }
                    // whole blocks; a wave without work on the chunk still gets one (all waves pass the
                    // chunk's barrier)
                    if (in_chunk == 0) { push(nullptr, 0, 0); ++in_chunk; }
                    while (in_chunk % 4 != 0) { push(nullptr, 0, 0); ++in_chunk; }
                    own[own.size() - 1] |= SC_END;                                    // word 3 of the chunk's last block
                }
                for (int q = 0; q < 4; ++q) push(nullptr, 0, 0);                          // read-ahead slack
                const size_t nb = own.size() / 4 - 1;                                     // blocks to process
                n_blk[(size_t)ps * SC_WAVES + w] = (int32_t)nb;
                // header words: own slot + flags, and the row offset of the NEXT block's bundle
                for (size_t i = 0; i < own.size(); ++i) {
                    const uint32_t nxt = i + 4 < own.size() ? own[i + 4] >> 16 : 0u;
                    hdr.push_back((own[i] & 0xffffu) | (nxt << 16));
                }
                b->n_blocks += nb;
            }
            if (natural)
                for (int k = 0; k < n_chunks; ++k) {
                    int x, y;
                    ChunkMixer(cnt).cost(segs.data() + (size_t)k * SC_NSEG, &x, &y);
                    b->crit_blocks += x;
                }
        }
        stream_off.back() = (int64_t)(hdr.size() / 4);
        if (dma_off.empty()) dma_off.assign(64, 0);
        b->fill = b->n_bundles ? (double)stored / (8.0 * (double)b->n_bundles) : 0.0;
        hipError_t e = upload(&b->hdr, hdr);
        if (e == hipSuccess) e = upload(&b->wts, wts);
        if (e == hipSuccess) e = upload(&b->stream_off, stream_off);
        if (e == hipSuccess) e = upload(&b->n_blk, n_blk);
        if (e == hipSuccess) e = upload(&b->dma_off, dma_off);
        if (e == hipSuccess) e = upload(&b->active_off, active_off);
        if (e == hipSuccess) e = upload(&b->col_of_slot, col_of_slot);
        if (e == hipSuccess && b->n_tail) e = upload(&b->tail_px, tail_px);
        if (e == hipSuccess && b->n_tail) e = upload(&b->tail_col, tail_col);
        if (e == hipSuccess && b->n_tail) e = upload(&b->tail_val, tail_val);
        if (e != hipSuccess) {
            set_error("uploading the scatter image failed: %s", hipGetErrorString(e));
            *err = (int)e;
            image_destroy(b);
            return nullptr;
        }
    } catch (const std::bad_alloc &) {
        set_error("out of host memory while packing the scatter image of a sparse stack");
        *err = LTMI_E_NOMEM;
        image_destroy(b);
        return nullptr;
    }
    return b;
}

// Is the stack worth the scatter kernel?  Slots of a bundle that hold a stored entry (8 = every slot):
// below ~1.2 of 8 the gather kernel (k_sell_apply) does the same work with less padding.
double scat_fill(const int64_t *indptr, const int64_t *indices, int nc, int64_t n_px, int64_t n_masks) {
    size_t bundles = 0;
    const int64_t nnz = indptr[n_px] * nc;
    if (nnz <= 0) return 0.0;
    const int SC_RS = range_size(std::min<int64_t>(n_masks * nc, SC_PASS));
    for (int64_t p = 0; p < n_px; ++p) {
        int64_t rng = -1, s0 = 0;                      // (ascending columns assumed for the estimate)
        for (int64_t e = indptr[p]; e < indptr[p + 1]; ++e)
            for (int c = 0; c < nc; ++c) {
                const int64_t col = indices[e] * nc + c;
                if (col / SC_RS != rng || col % SC_RS >= s0 + 8 || col % SC_RS < s0) {
                    ++bundles;
                    rng = col / SC_RS;
                    s0 = (col % SC_RS) & ~(int64_t)1;
                }
            }
    }
    return (double)nnz / (8.0 * (double)std::max<size_t>(bundles, 1));
}

void *scat_build(const int64_t *indptr, const int64_t *indices, const float *vals, int nc, int64_t n_px,
                 int64_t n_masks, int *err) {
    *err = LTMI_OK;
    ScatSet *s = new (std::nothrow) ScatSet();
    if (!s) {
        *err = LTMI_E_NOMEM;
        return nullptr;
    }
    try {
        s->n_px = n_px;
        s->n_masks = n_masks;
        s->nc = nc;
        const int64_t nnz = indptr[n_px];
        s->indptr.assign(indptr, indptr + n_px + 1);
        s->indices.assign(indices, indices + nnz);
        s->vals.assign(vals, vals + nnz * nc);
    } catch (const std::bad_alloc &) {
        delete s;
        *err = LTMI_E_NOMEM;
        set_error("out of host memory while keeping the CSR arrays of a sparse stack");
        return nullptr;
    }
    return s;
}

template <typename T>
static int launch_scatter(ltmi_masks *m, ScatImage *b, const T *tile, int64_t n_frames, int64_t ld, float *out,
                          int64_t ld_out_f, int n_cols, int accumulate, hipStream_t stream) {
    auto kern = k_scatter<T>;
    int abl = 0;
    if constexpr (std::is_same<T, uint16_t>::value) {
        const char *e = getenv("LTMI_SCATTER_ABLATE");
        abl = e ? atoi(e) : 0;
        if (abl == 1) kern = k_scatter<T, 1>;
        else if (abl == 2) kern = k_scatter<T, 2>;
        else if (abl == 3) kern = k_scatter<T, 3>;
        else if (abl == 4) kern = k_scatter<T, 4>;
        else if (abl == 5) kern = k_scatter<T, 5>;
        else if (abl == 6) kern = k_scatter<T, 6>;
        else if (abl == 7) kern = k_scatter<T, 7>;
        else abl = 0;
    }
    static bool set[16][8] = {{false}};
    if (!set[m->device & 15][abl]) {
        LTMI_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SC_LDS));
        set[m->device & 15][abl] = true;
    }
    dim3 grid((unsigned)((n_frames + SC_FB - 1) / SC_FB), (unsigned)b->n_pass);
    hipLaunchKernelGGL(kern, grid, dim3(SC_WAVES * 64), SC_LDS, stream, tile, ld, n_frames,
                       (const uint32_t *)b->hdr, (const float *)b->wts, (const int64_t *)b->stream_off,
                       (const int32_t *)b->n_blk, (const int32_t *)b->dma_off, (const int32_t *)b->active_off,
                       (const int32_t *)b->col_of_slot, out, ld_out_f, n_cols, accumulate, m->roi_rows);
    LTMI_HIP(hipGetLastError());
    if (b->n_tail > 0) {
        hipLaunchKernelGGL(k_scatter_tail<T>, dim3((unsigned)((n_frames + 255) / 256)), dim3(256), 0, stream,
                           tile, ld, n_frames, (const int32_t *)b->tail_px, (const int32_t *)b->tail_col,
                           (const float *)b->tail_val, b->n_tail, out, ld_out_f, m->roi_rows);
        LTMI_HIP(hipGetLastError());
    }
    snprintf(m->last_kernel, sizeof(m->last_kernel),
             "k_scatter<%s%s> grid=(%u,%u) blocks=%zu fill=%.2f crit=%ld", typeid(T).name(),
             m->roi_rows ? ",rows" : "", grid.x, grid.y, b->n_blocks, b->fill, b->crit_blocks);
    return LTMI_OK;
}

// handled = false: the tile does not meet the kernel's rules (the caller goes on to the other sparse kernels)
int scat_apply(ltmi_masks *m, void *set, int cplx, const void *tile, int tile_dtype, int64_t n_frames,
               int64_t ld_tile, void *out, int64_t ld_out, int accumulate, hipStream_t stream, bool *handled) {
    ScatSet *s = (ScatSet *)set;
    *handled = false;
    if (!s || n_frames <= 0) return LTMI_OK;
    const int sz = dtype_size(tile_dtype);
    if (!(sz == 1 || sz == 2 || sz == 4) || tile_dtype == LTMI_U32 || tile_dtype == LTMI_I32) return LTMI_OK;
    if (!vector_loads_ok(tile, ld_tile, (size_t)sz)) return LTMI_OK;
    if (s->n_px * sz > 0x7fffffff) return LTMI_OK;                       // (row offsets are 31-bit)
    const int slot = sz == 1 ? 0 : (sz == 2 ? 1 : 2);
    if (!s->img[slot]) {
        if (s->failed[slot]) return LTMI_OK;
        int err = LTMI_OK;
        s->img[slot] = build_image(*s, sz, &err);
        if (!s->img[slot]) {
            // (no room for the image, say: the blocked image or the gather kernel take the tile -- the scatter
            // image is an optimisation, not a requirement)
            s->failed[slot] = true;
            (void)hipGetLastError();
            return LTMI_OK;
        }
    }
    ScatImage *b = s->img[slot];
    const int nc = cplx ? 2 : 1;
    const int n_cols = (int)(m->n_masks * nc);
    float *o = (float *)out;
    const int64_t ldo = ld_out * nc;
    *handled = true;
    switch (tile_dtype) {
        case LTMI_BOOL:
        case LTMI_U8: return launch_scatter<uint8_t>(m, b, (const uint8_t *)tile, n_frames, ld_tile, o, ldo, n_cols, accumulate, stream);
        case LTMI_I8: return launch_scatter<int8_t>(m, b, (const int8_t *)tile, n_frames, ld_tile, o, ldo, n_cols, accumulate, stream);
        case LTMI_U16: return launch_scatter<uint16_t>(m, b, (const uint16_t *)tile, n_frames, ld_tile, o, ldo, n_cols, accumulate, stream);
        case LTMI_I16: return launch_scatter<int16_t>(m, b, (const int16_t *)tile, n_frames, ld_tile, o, ldo, n_cols, accumulate, stream);
        case LTMI_F32: return launch_scatter<float>(m, b, (const float *)tile, n_frames, ld_tile, o, ldo, n_cols, accumulate, stream);
    }
    *handled = false;
    return LTMI_OK;
}

}  // namespace ltmi
