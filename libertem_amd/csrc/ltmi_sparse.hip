// Sparse mask application for gfx950 (MI355X):  out[f, k] (+)= sum_p tile[f, p] * M[p, k],
// M sparse (n_px x n_masks), given as the CSR matrix the reference builds in _build_sparse
// (src/libertem/common/container.py:53-64).
//
// Replaces the numba kernels _rmatmul_csr / _rmatmul_csc (src/libertem/common/numba/__init__.py:
// 153-184), which walk the nnz of one pixel row at a time with a strided column gather of the
// tile.  On the GPU the data movement is turned around:
//
//   * a workgroup owns F = 16 frames and a block of masks and sweeps the pixel axis in chunks of
//     P = 1024 pixels.  The chunk of the 16 frames is converted to f32 and staged in LDS,
//     pixel-major: slab[p][16 frames] (64 KiB), so one mask entry (pixel, value) needs 4
//     ds_read_b128 to fetch that pixel of all 16 frames.
//   * the masks are re-packed on the host into a sliced-ELL format per pixel chunk: 64 masks
//     (one per lane) share a slice whose length is the longest of them inside the chunk; entry j
//     of the slice is a coalesced 64-lane row of (local pixel, value).  Lanes keep 16 (x masks
//     per thread) accumulators in registers for the whole sweep -- no atomics, deterministic.
//   * 64 consecutive masks share a slice (one wave): neighbouring masks of localised stacks (rings)
//     have similar entry counts inside a pixel chunk, which keeps the ELL padding low
//     (C4: 11 368 rows vs 20 020 when masks are dealt round-robin to the waves).
//   * the 16-B quarter q of a slab row is stored at q ^ ((p >> 2) & 3): without it only p % 4
//     selects the LDS slot of a ds_read_b128 and random gathers are >= 4-way conflicted.
//
// HBM traffic: frames once per pass over MB = 1024 (512 complex) masks; the SELL image comes from L2.
#include "ltmi_common.h"
#include <vector>
#include <algorithm>
#include <cstring>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <typeinfo>

namespace ltmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int SP_F = 16;        // frames per workgroup
constexpr int SP_P = 1024;      // pixels per chunk
constexpr int SP_NT = 256;      // threads per workgroup
constexpr int SP_U = 4;         // SELL rows fetched per group (slices padded to a multiple)

struct CsrImage {
    int cplx = 0;               // 0: f32 values, 1: complex64 values
    int mpt = 4;                // masks per thread per pass
    int mb = 1024;              // masks per pass = 256 * mpt
    int n_pass = 0;
    int n_chunks = 0;
    uint32_t *pix = nullptr;    // [rows][64]
    float *val = nullptr;       // [rows][64] (x2 interleaved re/im for complex)
    double *val64 = nullptr;    // float64 results: the values as doubles instead (f64 != 0)
    int f64 = 0;
    int *row_off = nullptr;     // [(pass * n_chunks + chunk) * mpt * 4 + slot * 4 + wave]
    int *row_len = nullptr;
    size_t n_rows = 0;
    void *scat = nullptr;       // ltmi_scatter.hip: the stack for k_scatter (images built per pixel size on first use)
    bool scat_all = false;      // LTMI_SPARSE_SCATTER=1: k_scatter for every pixel type (default: float32 frames)
    double bell_ratio = 0.;     // padded MACs of the blocked image per stored entry (0: not computed)
    KeptCsr *kept = nullptr;    // host copy until the detector shape is known (ltmi_masks_set_sig_shape) or the first product
    void *band = nullptr;       // ltmi_fold.hip: column blocks with a common support each, folded (float32 frames)
    double nnz_real = 0.;
    int *active = nullptr;      // chunks with entries, concatenated per pass
    int *active_off = nullptr;  // [n_pass + 1]
    void *bell = nullptr;       // blocked image for the matrix-core kernel (ltmi_bell.hip) or null
    // integer result dtypes: the values are held as doubles (f64 = 1) and a product whose every
    // partial sum stays below 2^52 is exact in the float64 gather kernel (see csr_apply)
    int int_result = 0;
    int sum_bits = 0;           // bits of the largest possible |sum of mask values| of one column
};

__device__ __forceinline__ int slab_word(int p, int f) {
    return p * SP_F + ((((f >> 2) ^ ((p >> 2) & 3)) << 2) | (f & 3));
}

// 8 consecutive pixels of one frame -> 8 floats
template <typename T>
__device__ __forceinline__ void load8_guarded(const T *row, int64_t p0, int64_t n_px, bool vec_ok,
                                              float (&f)[8]) {
    if (vec_ok && p0 + 8 <= n_px) {
        if constexpr (sizeof(T) == 2) {
            typedef u32x4 u32x4_u __attribute__((aligned(2)));   // rows at any element alignment
            const u32x4 r = __builtin_nontemporal_load((const u32x4_u *)(row + p0));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (std::is_signed<T>::value) {
                    f[2 * i] = (float)((int)(r[i] << 16) >> 16);
                    f[2 * i + 1] = (float)((int)r[i] >> 16);
                } else {
                    f[2 * i] = (float)(r[i] & 0xffffu);
                    f[2 * i + 1] = (float)(r[i] >> 16);
                }
            }
            return;
        } else if constexpr (sizeof(T) == 1) {
            typedef u32x2 u32x2_u __attribute__((aligned(1)));
            const u32x2 r = __builtin_nontemporal_load((const u32x2_u *)(row + p0));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if constexpr (std::is_signed<T>::value)
                        f[4 * i + b] = (float)((int)(r[i] << (24 - 8 * b)) >> 24);
                    else
                        f[4 * i + b] = (float)((r[i] >> (8 * b)) & 0xffu);
                }
            return;
        } else if constexpr (std::is_same<T, float>::value) {
            typedef f32x4 f32x4_u __attribute__((aligned(4)));
            const f32x4 a = __builtin_nontemporal_load((const f32x4_u *)(row + p0));
            const f32x4 b = __builtin_nontemporal_load((const f32x4_u *)(row + p0) + 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) { f[i] = a[i]; f[4 + i] = b[i]; }
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (p0 + j < n_px) ? (float)row[p0 + j] : 0.f;
}

// the same for an accumulation type A: float (above) or double (float64 results: int32 / uint32 /
// int64 / float64 frames, or float64 mask values -- np.result_type, reference udf/masks.py:362)
template <typename T, typename A>
__device__ __forceinline__ void load8_as(const T *row, int64_t p0, int64_t n_px, bool vec_ok,
                                         A (&f)[8]) {
    if constexpr (std::is_same<A, float>::value) {
        load8_guarded<T>(row, p0, n_px, vec_ok, f);
    } else if constexpr (sizeof(T) <= 2 || std::is_same<T, float>::value) {
        float t[8];
        load8_guarded<T>(row, p0, n_px, vec_ok, t);      // exact in float32, then widened
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (A)t[j];
    } else {
        if (vec_ok && p0 + 8 <= n_px) {
            constexpr int VE = 16 / (int)sizeof(T);      // elements per 16-byte load
            typedef T v_a __attribute__((ext_vector_type(VE)));
            typedef v_a v_t __attribute__((aligned(sizeof(T))));
            const v_t *vp = (const v_t *)(row + p0);
#pragma unroll
            for (int i = 0; i < 8 / VE; ++i) {
                const v_a v = __builtin_nontemporal_load(vp + i);
#pragma unroll
                for (int e = 0; e < VE; ++e) f[i * VE + e] = (A)v[e];
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (p0 + j < n_px) ? (A)row[p0 + j] : (A)0;
    }
}

template <typename T, int MPT, bool CPLX, typename A = float, bool REDO = false>
__global__ void __launch_bounds__(SP_NT)
k_sell_apply(const T *__restrict__ tile, int64_t ld, int64_t n_frames, int64_t n_px,
             const uint32_t *__restrict__ pix, const A *__restrict__ val,
             const int *__restrict__ row_off, const int *__restrict__ row_len,
             const int *__restrict__ active, const int *__restrict__ active_off, int n_chunks,
             A *__restrict__ out, int64_t ld_out, int n_masks, int accumulate, int vec_ok,
             int ablate, const int32_t *__restrict__ rows, const int32_t *__restrict__ sel,
             const int *__restrict__ n_sel, int n_pass, int n_split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char slab_raw[];
    A *slab = (A *)slab_raw;                                          // [SP_P][SP_F]
    typedef A ax4 __attribute__((ext_vector_type(4)));
    typedef A ax2 __attribute__((ext_vector_type(2)));
    constexpr int NC = CPLX ? 2 : 1;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    // REDO: the frames with non-finite results again (ltmi_guard.hip).  `sel` lists their result rows, the length
    // of the list sits in device memory (no host synchronisation): a fixed number of workgroups walks the work items
    // (16 listed frames, pass, a 1 / n_split share of the pass's pixel chunks) and adds its share to the rows, which
    // k_zero_sel_rows cleared -- a handful of listed frames still spreads over the chip; none: every workgroup leaves
    int64_t work = blockIdx.x, n_work = 1;
    if (REDO) {
        n_frames = *n_sel;
        n_work = ((n_frames + SP_F - 1) / SP_F) * n_pass * n_split;
        if (work >= n_work) return;
    }
    // padding entries of the image (value 0) point at pixel row SP_P, which is all zeros: a
    // non-finite pixel of a frame only reaches the masks that really contain it, like in the
    // reference's CSR loop (0 * NaN would be NaN)
    if (tid < SP_F) slab[SP_P * SP_F + tid] = (A)0;
    // loader role: frame lf, pixel group lg (16 groups of 8 px per 128-px sweep step)
    const int lf = tid & 15, lg = tid >> 4;
  for (;;) {
    const int split = REDO ? (int)(work % n_split) : 0;
    const int pass = REDO ? (int)((work / n_split) % n_pass) : (int)blockIdx.y;
    const int64_t f0 = (REDO ? work / ((int64_t)n_split * n_pass) : (int64_t)blockIdx.x) * SP_F;
    int64_t frame = f0 + lf;
    if (frame > n_frames - 1) frame = n_frames - 1;
    if (REDO) frame = sel[frame];
    if (rows) frame = rows[frame];                    // a region of interest: result row i = frame rows[i]
    const T *row = tile + frame * ld;

    A acc[MPT][SP_F][NC];
#pragma unroll
    for (int i = 0; i < MPT; ++i)
#pragma unroll
        for (int f = 0; f < SP_F; ++f)
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[i][f][c] = (A)0;

    int ai0 = active_off[pass], ai1 = active_off[pass + 1];
    if (REDO) {
        const int per = (ai1 - ai0 + n_split - 1) / n_split;
        ai0 += split * per;
        ai1 = ai0 + per < ai1 ? ai0 + per : ai1;
    }
    for (int ai = ai0; ai < ai1; ++ai) {
        const int ch = active[ai];
        const int *lens = row_len + (int64_t)ai * (MPT * 4);
        const int *offs = row_off + (int64_t)ai * (MPT * 4);

        __syncthreads();                             // previous chunk fully consumed
        if (ablate != 2)
#pragma unroll
        for (int it = 0; it < SP_P / 128; ++it) {
            const int pl = it * 128 + lg * 8;
            A x[8];
            load8_as<T, A>(row, (int64_t)ch * SP_P + pl, n_px, vec_ok != 0, x);
#pragma unroll
            for (int j = 0; j < 8; ++j) slab[slab_word(pl + j, lf)] = x[j];
        }
        __syncthreads();

        if (ablate != 1)
#pragma unroll
        for (int i = 0; i < MPT; ++i) {
            const int len = lens[i * 4 + wave];
            const int64_t base = (int64_t)offs[i * 4 + wave] * 64 + lane;
            // slices are padded to a multiple of SP_U rows (zero entries): SP_U rows of
            // (pixel, value) are fetched together, one group ahead of the LDS gathers
            uint32_t pn[SP_U];
            A vrn[SP_U], vin[SP_U];
            auto fetch = [&](int j) {
#pragma unroll
                for (int u = 0; u < SP_U; ++u) {
                    const int64_t e = base + (int64_t)(j + u) * 64;
                    pn[u] = pix[e];
                    if (CPLX) {
                        const ax2 v2 = ((const ax2 *)val)[e];
                        vrn[u] = v2[0];
                        vin[u] = v2[1];
                    } else {
                        vrn[u] = val[e];
                        vin[u] = (A)0;
                    }
                }
            };
            if (len > 0) fetch(0);
            for (int j = 0; j < len; j += SP_U) {
                uint32_t pc[SP_U];
                A vr[SP_U], vi[SP_U];
#pragma unroll
                for (int u = 0; u < SP_U; ++u) { pc[u] = pn[u]; vr[u] = vrn[u]; vi[u] = vin[u]; }
                if (j + SP_U < len) fetch(j + SP_U);
#pragma unroll
                for (int u = 0; u < SP_U; ++u) {
                    const A *rowp = slab + pc[u] * SP_F;
                    const int s = (pc[u] >> 2) & 3;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const ax4 x = *(const ax4 *)(rowp + ((q ^ s) << 2));
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[i][q * 4 + e][0] += x[e] * vr[u];
                            if (CPLX) acc[i][q * 4 + e][1] += x[e] * vi[u];
                        }
                    }
                }
            }
        }
    }

    // mask of (slot i, wave, lane): k = pass*MB + i*256 + lane*4 + wave
#pragma unroll
    for (int i = 0; i < MPT; ++i) {
        const int k = pass * (256 * MPT) + i * 256 + wave * 64 + lane;
        if (k >= n_masks) continue;
#pragma unroll
        for (int f = 0; f < SP_F; ++f) {
            if (f0 + f >= n_frames) break;
            const int64_t orow = REDO ? (int64_t)sel[f0 + f] : f0 + f;
            A *o = out + orow * ld_out + (int64_t)k * NC;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (REDO) unsafeAtomicAdd(&o[c], acc[i][f][c]);
                else o[c] = accumulate ? o[c] + acc[i][f][c] : acc[i][f][c];
            }
        }
    }
    if (!REDO) break;
    work += gridDim.x;
    if (work >= n_work) break;
  }
}

// clears the result rows of the listed frames before k_sell_apply<REDO> adds their sums
template <typename A>
__global__ void __launch_bounds__(256)
k_zero_sel_rows(A *__restrict__ out, int64_t ld_out, int n_cols, const int32_t *__restrict__ sel,
                const int *__restrict__ n_sel) {
    const int n = *n_sel;
    for (int j = blockIdx.x; j < n; j += gridDim.x) {
        A *o = out + (int64_t)sel[j] * ld_out;
        for (int c = threadIdx.x; c < n_cols; c += 256) o[c] = (A)0;
    }
}

// A u16-slab, double-buffered variant with one slice per wave was tried and removed: it did not
// beat this kernel (profiles/r01_sparse_ablation.txt) -- the gather loop is bound by its LDS gathers
// and their latency, see DESIGN.md section 4.3.  Localised stacks go to the blocked image on the
// matrix cores instead (ltmi_bell.hip).

}  // namespace ltmi

using namespace ltmi;

namespace ltmi {

int csr_destroy(ltmi_masks *m) {
    CsrImage *c = (CsrImage *)m->csr;
    if (!c) return LTMI_OK;
    if (c->pix) (void)hipFree(c->pix);
    if (c->val) (void)hipFree(c->val);
    if (c->val64) (void)hipFree(c->val64);
    if (c->row_off) (void)hipFree(c->row_off);
    if (c->row_len) (void)hipFree(c->row_len);
    if (c->active) (void)hipFree(c->active);
    if (c->active_off) (void)hipFree(c->active_off);
    bell_destroy(c->bell);
    scat_destroy(c->scat);
    band_destroy(c->band);
    band_free_csr(c->kept);
    delete c;
    m->csr = nullptr;
    return LTMI_OK;
}

// the redo launch (ltmi_guard.hip): a fixed number of workgroups over the work items of the listed frames
constexpr unsigned SELL_REDO_WGS = 1024;

static int sell_redo_split(const CsrImage *c) {
    // a share of at least 8 pixel chunks per work item, at most 64 shares
    const int s = c->n_chunks / 8;
    return s < 1 ? 1 : (s > 64 ? 64 : s);
}

template <typename KERN>
static int sell_set_lds(KERN kern, int device, size_t lds, bool *set) {
    if (!set[device & 15]) {
        LTMI_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        set[device & 15] = true;
    }
    return LTMI_OK;
}

template <typename T>
static int launch_sell(ltmi_masks *m, CsrImage *c, const T *tile, int64_t n_frames, int64_t ld,
                       float *out, int64_t ld_out_f, int accumulate, hipStream_t stream,
                       const int32_t *sel = nullptr, const int *n_sel = nullptr) {
    const int vec_ok = vector_loads_ok(tile, ld, sizeof(T)) ? 1 : 0;
    const char *abl = getenv("LTMI_SELL_ABLATE");     // 1: loader only, 2: gathers only (bench)
    const int ablate = abl ? atoi(abl) : 0;
    dim3 grid((unsigned)((n_frames + SP_F - 1) / SP_F), (unsigned)c->n_pass);
    // + one all-zero pixel row (index SP_P) for the padding entries of the image
    const size_t lds = (size_t)(SP_P + 1) * SP_F * sizeof(float);
    const int nc = c->cplx ? 2 : 1;
    if (sel) {
        const int split = sell_redo_split(c);
        hipLaunchKernelGGL(k_zero_sel_rows<float>, dim3(256), dim3(256), 0, stream, out, ld_out_f,
                           (int)m->n_masks * nc, sel, n_sel);
        LTMI_HIP(hipGetLastError());
        int rc;
        if (c->cplx) {
            auto kern = k_sell_apply<T, 2, true, float, true>;
            static bool set[16] = {false};
            if ((rc = sell_set_lds(kern, m->device, lds, set)) != LTMI_OK) return rc;
            hipLaunchKernelGGL(kern, dim3(SELL_REDO_WGS), dim3(SP_NT), lds, stream, tile, ld, n_frames, m->n_px,
                               (const uint32_t *)c->pix, (const float *)c->val, (const int *)c->row_off,
                               (const int *)c->row_len, (const int *)c->active, (const int *)c->active_off,
                               c->n_chunks, out, ld_out_f, (int)m->n_masks, 0, vec_ok, 0, m->roi_rows, sel, n_sel,
                               c->n_pass, split);
        } else {
            auto kern = k_sell_apply<T, 4, false, float, true>;
            static bool set[16] = {false};
            if ((rc = sell_set_lds(kern, m->device, lds, set)) != LTMI_OK) return rc;
            hipLaunchKernelGGL(kern, dim3(SELL_REDO_WGS), dim3(SP_NT), lds, stream, tile, ld, n_frames, m->n_px,
                               (const uint32_t *)c->pix, (const float *)c->val, (const int *)c->row_off,
                               (const int *)c->row_len, (const int *)c->active, (const int *)c->active_off,
                               c->n_chunks, out, ld_out_f, (int)m->n_masks, 0, vec_ok, 0, m->roi_rows, sel, n_sel,
                               c->n_pass, split);
        }
        LTMI_HIP(hipGetLastError());
        return LTMI_OK;                                   // (a redo keeps the name of the kernel it follows)
    }
    if (c->cplx) {
        auto kern = k_sell_apply<T, 2, true>;
        static bool set[16] = {false};
        const int rc = sell_set_lds(kern, m->device, lds, set);
        if (rc != LTMI_OK) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(SP_NT), lds, stream, tile, ld, n_frames, m->n_px,
                           (const uint32_t *)c->pix, (const float *)c->val, (const int *)c->row_off,
                           (const int *)c->row_len, (const int *)c->active, (const int *)c->active_off,
                           c->n_chunks, out, ld_out_f, (int)m->n_masks, accumulate, vec_ok, ablate, m->roi_rows,
                           (const int32_t *)nullptr, (const int *)nullptr, c->n_pass, 1);
    } else {
        auto kern = k_sell_apply<T, 4, false>;
        static bool set[16] = {false};
        const int rc = sell_set_lds(kern, m->device, lds, set);
        if (rc != LTMI_OK) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(SP_NT), lds, stream, tile, ld, n_frames, m->n_px,
                           (const uint32_t *)c->pix, (const float *)c->val, (const int *)c->row_off,
                           (const int *)c->row_len, (const int *)c->active, (const int *)c->active_off,
                           c->n_chunks, out, ld_out_f, (int)m->n_masks, accumulate, vec_ok, ablate, m->roi_rows,
                           (const int32_t *)nullptr, (const int *)nullptr, c->n_pass, 1);
    }
    LTMI_HIP(hipGetLastError());
    m->last_exact = true;                                 // only stored entries were multiplied
    snprintf(m->last_kernel, sizeof(m->last_kernel), "k_sell_apply<%s,%s%s> grid=(%u,%u) rows=%zu",
             typeid(T).name(), c->cplx ? "c64" : "f32", m->roi_rows ? ",rows" : "", grid.x, grid.y,
             c->n_rows);
    return LTMI_OK;
}

// float64 results: the same kernel with double slab / accumulators / values (128 KiB of LDS)
template <typename T>
static int launch_sell64(ltmi_masks *m, CsrImage *c, const T *tile, int64_t n_frames, int64_t ld,
                         double *out, int64_t ld_out, int accumulate, hipStream_t stream,
                         const int32_t *sel = nullptr, const int *n_sel = nullptr) {
    const int vec_ok = vector_loads_ok(tile, ld, sizeof(T)) ? 1 : 0;
    dim3 grid((unsigned)((n_frames + SP_F - 1) / SP_F), (unsigned)c->n_pass);
    const size_t lds = (size_t)(SP_P + 1) * SP_F * sizeof(double);
    if (sel) {
        hipLaunchKernelGGL(k_zero_sel_rows<double>, dim3(256), dim3(256), 0, stream, out, ld_out, (int)m->n_masks,
                           sel, n_sel);
        LTMI_HIP(hipGetLastError());
        auto kern = k_sell_apply<T, 4, false, double, true>;
        static bool set[16] = {false};
        const int rc = sell_set_lds(kern, m->device, lds, set);
        if (rc != LTMI_OK) return rc;
        hipLaunchKernelGGL(kern, dim3(SELL_REDO_WGS), dim3(SP_NT), lds, stream, tile, ld, n_frames, m->n_px,
                           (const uint32_t *)c->pix, (const double *)c->val64, (const int *)c->row_off,
                           (const int *)c->row_len, (const int *)c->active, (const int *)c->active_off,
                           c->n_chunks, out, ld_out, (int)m->n_masks, 0, vec_ok, 0, m->roi_rows, sel, n_sel,
                           c->n_pass, sell_redo_split(c));
        LTMI_HIP(hipGetLastError());
        return LTMI_OK;
    }
    auto kern = k_sell_apply<T, 4, false, double>;
    static bool set[16] = {false};
    const int rc = sell_set_lds(kern, m->device, lds, set);
    if (rc != LTMI_OK) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(SP_NT), lds, stream, tile, ld, n_frames, m->n_px,
                       (const uint32_t *)c->pix, (const double *)c->val64, (const int *)c->row_off,
                       (const int *)c->row_len, (const int *)c->active, (const int *)c->active_off,
                       c->n_chunks, out, ld_out, (int)m->n_masks, accumulate, vec_ok, 0, m->roi_rows,
                       (const int32_t *)nullptr, (const int *)nullptr, c->n_pass, 1);
    LTMI_HIP(hipGetLastError());
    m->last_exact = true;
    snprintf(m->last_kernel, sizeof(m->last_kernel), "k_sell_apply<%s,f64%s> grid=(%u,%u) rows=%zu",
             typeid(T).name(), m->roi_rows ? ",rows" : "", grid.x, grid.y, c->n_rows);
    return LTMI_OK;
}

// can this handle read the frames of a region of interest through a row list (ltmi_apply_masks_rows)?
// Both sparse kernels do (the tile dtypes csr_apply dispatches).
bool csr_rows_ok(const ltmi_masks *m, const void *tile, int tile_dtype, int64_t ld_tile) {
    const CsrImage *c = (const CsrImage *)m->csr;
    if (!c || !tile) return false;
    switch (tile_dtype) {
        case LTMI_BOOL: case LTMI_U8: case LTMI_I8: case LTMI_U16: case LTMI_I16: case LTMI_F32:
            return true;
        case LTMI_U32: case LTMI_I32: case LTMI_U64: case LTMI_I64: case LTMI_F64:
            return c->f64 != 0;
    }
    return false;
}

// exact integer sum (held in a double, |v| < 2^53) -> wrap-around integer of the result width: the
// arithmetic of NumPy's / SciPy's integer matmul (the same step as ltmi_dense64.hip k_f64_to_int)
template <typename S>
__global__ void k_sell_f64_to_int(const double *__restrict__ src, int64_t n_frames, int n_masks,
                                  S *__restrict__ out, int64_t ld_out, int accumulate) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_frames * n_masks) return;
    const int64_t f = idx / n_masks;
    const int k = (int)(idx % n_masks);
    const S v = (S)(uint64_t)(int64_t)src[idx];
    S *p = out + f * ld_out + k;
    *p = accumulate ? (S)(*p + v) : v;
}

// integer stack x integer frames: is every possible partial sum below 2^52 (float64 chain exact)?
bool csr_int_exact(const ltmi_masks *m, int tile_dtype) {
    const CsrImage *c = (const CsrImage *)m->csr;
    if (!c || !c->int_result) return false;
    int data_bits;
    switch (tile_dtype) {
        case LTMI_BOOL: data_bits = 1; break;
        case LTMI_U8: case LTMI_I8: data_bits = 8; break;
        case LTMI_U16: case LTMI_I16: data_bits = 16; break;
        case LTMI_U32: case LTMI_I32: data_bits = 32; break;
        default: return false;                            // 64-bit or non-integer tiles
    }
    return data_bits + c->sum_bits <= 52;
}

// the detector shape behind the pixels (ltmi_masks_set_sig_shape): a stack of column blocks with a common support
// each gets its folded dense image (ltmi_fold.hip, band_build); the host copy of the CSR arrays is released
int csr_set_sig_shape(ltmi_masks *m, int sig_h, int sig_w) {
    CsrImage *c = (CsrImage *)m->csr;
    if (!c || !c->kept) return LTMI_OK;
    band_destroy(c->band);
    c->band = band_build(c->kept, sig_h, sig_w, c->bell ? c->bell_ratio * c->nnz_real : 0.);
    band_free_csr(c->kept);
    c->kept = nullptr;
    return LTMI_OK;
}

bool csr_has_band(const ltmi_masks *m) {
    const CsrImage *c = (const CsrImage *)m->csr;
    return c && c->band;
}

// float32 frames: k_scatter only where the blocked image pads at least this much (per stored entry)
static constexpr double SCAT_MIN_BELL_RATIO = 3.0;

int csr_apply(ltmi_masks *m, const void *tile, int tile_dtype, int64_t n_frames, int64_t ld_tile,
              void *out, int64_t ld_out, int accumulate, hipStream_t stream) {
    CsrImage *c = (CsrImage *)m->csr;
    if (m->roi_rows && !csr_rows_ok(m, tile, tile_dtype, ld_tile))
        LTMI_FAIL(LTMI_E_INVALID, "sparse masks: this handle / tile cannot take a row list");
    if (c->int_result) {
        // exact in float64 -> gather into a float64 scratch, then truncate to the result width
        if (!csr_int_exact(m, tile_dtype))
            LTMI_FAIL(LTMI_E_DTYPE, "sparse integer masks: %s tiles against this stack can exceed 2^52 "
                      "(densify the stack: the integer VALU kernel)", dtype_name(tile_dtype));
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
        c->int_result = 0;                                  // (the float64 branch below, once)
        const int rc = csr_apply(m, tile, tile_dtype, n_frames, ld_tile, m->res64, m->n_masks, 0,
                                 stream);
        c->int_result = 1;
        if (rc != LTMI_OK) return rc;
        const int64_t n = n_frames * m->n_masks;
        const dim3 grid((unsigned)((n + 255) / 256));
        const double *src = (const double *)m->res64;
        switch (dtype_size(m->result_dtype)) {
            case 1: hipLaunchKernelGGL(k_sell_f64_to_int<uint8_t>, grid, dim3(256), 0, stream, src, n_frames, (int)m->n_masks, (uint8_t *)out, ld_out, accumulate); break;
            case 2: hipLaunchKernelGGL(k_sell_f64_to_int<uint16_t>, grid, dim3(256), 0, stream, src, n_frames, (int)m->n_masks, (uint16_t *)out, ld_out, accumulate); break;
            case 4: hipLaunchKernelGGL(k_sell_f64_to_int<uint32_t>, grid, dim3(256), 0, stream, src, n_frames, (int)m->n_masks, (uint32_t *)out, ld_out, accumulate); break;
            default: hipLaunchKernelGGL(k_sell_f64_to_int<uint64_t>, grid, dim3(256), 0, stream, src, n_frames, (int)m->n_masks, (uint64_t *)out, ld_out, accumulate); break;
        }
        LTMI_HIP(hipGetLastError());
        const size_t len = strlen(m->last_kernel);
        snprintf(m->last_kernel + len, sizeof(m->last_kernel) - len, " exact-int");
        return LTMI_OK;
    }
    if (c->f64) {
        double *o = (double *)out;
        switch (tile_dtype) {
            case LTMI_BOOL:
            case LTMI_U8: return launch_sell64<uint8_t>(m, c, (const uint8_t *)tile, n_frames, ld_tile, o, ld_out, accumulate, stream);
            case LTMI_I8: return launch_sell64<int8_t>(m, c, (const int8_t *)tile, n_frames, ld_tile, o, ld_out, accumulate, stream);
            case LTMI_U16: return launch_sell64<uint16_t>(m, c, (const uint16_t *)tile, n_frames, ld_tile, o, ld_out, accumulate, stream);
            case LTMI_I16: return launch_sell64<int16_t>(m, c, (const int16_t *)tile, n_frames, ld_tile, o, ld_out, accumulate, stream);
            case LTMI_U32: return launch_sell64<uint32_t>(m, c, (const uint32_t *)tile, n_frames, ld_tile, o, ld_out, accumulate, stream);
            case LTMI_I32: return launch_sell64<int32_t>(m, c, (const int32_t *)tile, n_frames, ld_tile, o, ld_out, accumulate, stream);
            case LTMI_U64: return launch_sell64<uint64_t>(m, c, (const uint64_t *)tile, n_frames, ld_tile, o, ld_out, accumulate, stream);
            case LTMI_I64: return launch_sell64<int64_t>(m, c, (const int64_t *)tile, n_frames, ld_tile, o, ld_out, accumulate, stream);
            case LTMI_F32: return launch_sell64<float>(m, c, (const float *)tile, n_frames, ld_tile, o, ld_out, accumulate, stream);
            case LTMI_F64: return launch_sell64<double>(m, c, (const double *)tile, n_frames, ld_tile, o, ld_out, accumulate, stream);
        }
        LTMI_FAIL(LTMI_E_DTYPE, "sparse masks with float64 results: tile dtype %s is not supported",
                  dtype_name(tile_dtype));
    }
    // one float32 FMA per stored entry on the vector ALUs (k_scatter; tuning 41: SELL kernel, 42: blocked image)
    // (float32 frames by default -- measured 7 % ahead of the float32 blocked image on C4, and a non-finite
    // pixel reaches fewer foreign masks; 1- / 2-byte pixels stay on the blocked images, which are 15 - 30 %
    // faster there: profiles/r04_sparse.txt.  LTMI_SPARSE_SCATTER=1: every pixel type)
    if (c->kept) {                                        // no detector shape came: the host copy is not needed
        band_free_csr(c->kept);
        c->kept = nullptr;
    }
    if (c->band && band_takes(c->band, m, tile, tile_dtype, ld_tile))
        return band_apply(m, c->band, tile, tile_dtype, n_frames, ld_tile, (float *)out,
                          ld_out * (c->cplx ? 2 : 1), (int)(m->n_masks * (c->cplx ? 2 : 1)), accumulate, stream);
    // ... unless the blocked image is well filled: a stack of dense column blocks (radial Fourier with several bins:
    // 1.3 padded MACs per stored entry) runs 3 x faster on the matrix cores (21 ms against 62 ms per 8192 frames
    // of 1024 x 1024, scripts/bench_second_runs.py); C4's rings (4.8) stay here
    const bool bell_better = c->bell && c->bell_ratio > 0. && c->bell_ratio < SCAT_MIN_BELL_RATIO && !c->scat_all;
    if (c->scat && (c->scat_all || tile_dtype == LTMI_F32) && !bell_better && m->tune_ksplit_ring != 41 &&
        m->tune_ksplit_ring != 42) {
        bool handled = false;
        const int rc = scat_apply(m, c->scat, c->cplx, tile, tile_dtype, n_frames, ld_tile, out, ld_out,
                                  accumulate, stream, &handled);
        if (rc != LTMI_OK || handled) return rc;
    }
    // localised stacks: blocked image on the matrix cores (set_tuning 41 forces the SELL kernel)
    if (c->bell && m->tune_ksplit_ring != 41) {
        bool handled = false;
        const int rc = bell_apply(m, c->bell, c->cplx, tile, tile_dtype, n_frames, ld_tile, out,
                                  ld_out, accumulate, stream, &handled);
        if (rc != LTMI_OK || handled) return rc;
    }
    float *o = (float *)out;
    const int64_t ldo = ld_out * (c->cplx ? 2 : 1);
    switch (tile_dtype) {
        case LTMI_BOOL:
        case LTMI_U8: return launch_sell<uint8_t>(m, c, (const uint8_t *)tile, n_frames, ld_tile, o, ldo, accumulate, stream);
        case LTMI_I8: return launch_sell<int8_t>(m, c, (const int8_t *)tile, n_frames, ld_tile, o, ldo, accumulate, stream);
        case LTMI_U16: return launch_sell<uint16_t>(m, c, (const uint16_t *)tile, n_frames, ld_tile, o, ldo, accumulate, stream);
        case LTMI_I16: return launch_sell<int16_t>(m, c, (const int16_t *)tile, n_frames, ld_tile, o, ldo, accumulate, stream);
        case LTMI_F32: return launch_sell<float>(m, c, (const float *)tile, n_frames, ld_tile, o, ldo, accumulate, stream);
    }
    LTMI_FAIL(LTMI_E_DTYPE, "sparse masks: tile dtype %s is not supported with result dtype %s "
              "(supported tiles: uint8 int8 uint16 int16 float32)", dtype_name(tile_dtype),
              dtype_name(m->result_dtype));
}


// The frames `sel[0 .. *n_sel)` of a product again, on the gather kernel -- stored entries only, like the reference's
// CSR loop (common/numba/__init__.py:153-184) -- written over their result rows (ltmi_guard.hip: frames whose
// results came out non-finite on a kernel that also multiplies padding zeros).  The launch covers `max_frames`.
int csr_redo(ltmi_masks *m, const void *tile, int tile_dtype, int64_t max_frames, int64_t ld_tile, void *out,
             int64_t ld_out, const int32_t *sel, const int *n_sel, const int32_t *roi_rows, hipStream_t stream) {
    CsrImage *c = m ? (CsrImage *)m->csr : nullptr;
    if (!c || c->int_result) LTMI_FAIL(LTMI_E_INVALID, "csr_redo: not a float sparse handle");
    const int32_t *keep = m->roi_rows;
    m->roi_rows = roi_rows;
    int rc = LTMI_E_DTYPE;
    if (c->f64) {
        double *o = (double *)out;
        if (tile_dtype == LTMI_F32) rc = launch_sell64<float>(m, c, (const float *)tile, max_frames, ld_tile, o, ld_out, 0, stream, sel, n_sel);
        else if (tile_dtype == LTMI_F64) rc = launch_sell64<double>(m, c, (const double *)tile, max_frames, ld_tile, o, ld_out, 0, stream, sel, n_sel);
    } else if (tile_dtype == LTMI_F32) {
        rc = launch_sell<float>(m, c, (const float *)tile, max_frames, ld_tile, (float *)out,
                                ld_out * (c->cplx ? 2 : 1), 0, stream, sel, n_sel);
    }
    m->roi_rows = keep;
    if (rc == LTMI_E_DTYPE) LTMI_FAIL(LTMI_E_DTYPE, "csr_redo: %s tiles against %s results", dtype_name(tile_dtype),
                                      dtype_name(m->result_dtype));
    return rc;
}

// does the handle hold an image other than the gather kernel's (blocked, scatter, banded)?  Without one every product
// runs on k_sell_apply, which multiplies stored entries only: nothing for ltmi_guard.hip to check
bool csr_has_fast_image(const ltmi_masks *m) {
    const CsrImage *c = m ? (const CsrImage *)m->csr : nullptr;
    return c && (c->bell || c->scat || c->band || c->kept);
}

bool csr_is_f64(const ltmi_masks *m) {
    const CsrImage *c = m ? (const CsrImage *)m->csr : nullptr;
    return c && c->f64;
}

}  // namespace ltmi

static thread_local bool g_gather_only = false;

extern "C" int ltmi_masks_create_csr_gather(int device, const int64_t *indptr, const int64_t *indices,
                                            const void *data, int result_dtype, int64_t n_px,
                                            int64_t n_masks, ltmi_masks **out) {
    g_gather_only = true;
    const int rc = ltmi_masks_create_csr(device, indptr, indices, data, result_dtype, n_px, n_masks, out);
    g_gather_only = false;
    return rc;
}

extern "C" int ltmi_masks_create_csr(int device, const int64_t *indptr, const int64_t *indices,
                                     const void *data, int result_dtype, int64_t n_px,
                                     int64_t n_masks, ltmi_masks **out) {
    if (!indptr || !out || n_px <= 0 || n_masks <= 0)
        LTMI_FAIL(LTMI_E_INVALID, "ltmi_masks_create_csr: bad arguments (n_px=%lld n_masks=%lld)",
                  (long long)n_px, (long long)n_masks);
    const bool int_result = result_dtype >= LTMI_U8 && result_dtype <= LTMI_I64;
    if (result_dtype != LTMI_F32 && result_dtype != LTMI_C64 && result_dtype != LTMI_F64 && !int_result)
        LTMI_FAIL(LTMI_E_DTYPE, "ltmi_masks_create_csr: result dtype %s not supported for sparse "
                  "stacks (float32 / complex64 / float64 / integers; densify for others)",
                  dtype_name(result_dtype));
    const int64_t nnz = indptr[n_px];
    if (nnz < 0 || (nnz > 0 && (!indices || !data)))
        LTMI_FAIL(LTMI_E_INVALID, "ltmi_masks_create_csr: inconsistent CSR arrays");
    for (int64_t e = 0; e < nnz; ++e)
        if (indices[e] < 0 || indices[e] >= n_masks)
            LTMI_FAIL(LTMI_E_SHAPE, "ltmi_masks_create_csr: column index %lld out of range",
                      (long long)indices[e]);
    LTMI_HIP(hipSetDevice(device));
    ltmi_masks *m = new (std::nothrow) ltmi_masks();
    CsrImage *c = new (std::nothrow) CsrImage();
    if (!m || !c) { delete m; delete c; LTMI_FAIL(LTMI_E_NOMEM, "out of host memory"); }
    m->device = device;
    m->kind = 2;
    m->result_dtype = result_dtype;
    m->n_masks = n_masks;
    m->n_px = n_px;
    m->csr = c;
    c->cplx = (result_dtype == LTMI_C64);
    c->mpt = c->cplx ? 2 : 4;
    c->mb = 256 * c->mpt;
    c->n_pass = (int)((n_masks + c->mb - 1) / c->mb);
    c->n_chunks = (int)((n_px + SP_P - 1) / SP_P);
    const int nc = c->cplx ? 2 : 1;
    c->f64 = (result_dtype == LTMI_F64) || int_result;
    c->int_result = int_result ? 1 : 0;
    const float *vals = (const float *)data;              // (f64: `data` holds doubles, see below)
    const double *vals64 = (const double *)data;
    const int64_t *vals_i = (const int64_t *)data;        // (integer results: int64 values)
    if (int_result) {
        // largest possible |column sum|: decides (with the tile dtype) whether the float64 chain is exact
        std::vector<double> col_abs((size_t)n_masks, 0.);
        for (int64_t e = 0; e < nnz; ++e)
            col_abs[(size_t)indices[e]] += std::fabs((double)vals_i[e]);
        double worst = 0.;
        for (double v : col_abs) worst = std::max(worst, v);
        int bits = 0;
        while (bits < 64 && std::ldexp(1.0, bits) <= worst) ++bits;
        c->sum_bits = bits;
    }

    // slice id of mask k: ((pass * n_chunks + chunk) * mpt + slot) * 4 + wave ; lane = (k%256)/4
    const size_t n_slices = (size_t)c->n_pass * c->n_chunks * c->mpt * 4;
    try {
        // count entries per (chunk, mask)
        std::vector<int> cnt((size_t)c->n_chunks * n_masks, 0);
        for (int64_t p = 0; p < n_px; ++p) {
            const int ch = (int)(p / SP_P);
            for (int64_t e = indptr[p]; e < indptr[p + 1]; ++e)
                cnt[(size_t)ch * n_masks + indices[e]]++;
        }
        const int NWS = c->mpt * 4;                       // slices (64 masks each) per pass
        auto slice_of = [&](int64_t k) -> int { return (int)((k % c->mb) / 64); };
        // padded slice lengths per (pass, chunk, slice)
        std::vector<int> len_full(n_slices, 0);
        for (int ch = 0; ch < c->n_chunks; ++ch)
            for (int64_t k = 0; k < n_masks; ++k) {
                const size_t s = ((size_t)(k / c->mb) * c->n_chunks + ch) * NWS + slice_of(k);
                len_full[s] = std::max(len_full[s], cnt[(size_t)ch * n_masks + k]);
            }
        for (size_t s = 0; s < n_slices; ++s) len_full[s] = (len_full[s] + SP_U - 1) / SP_U * SP_U;
        // active chunks per pass
        std::vector<int> active, active_off(c->n_pass + 1, 0);
        for (int ps = 0; ps < c->n_pass; ++ps) {
            for (int ch = 0; ch < c->n_chunks; ++ch) {
                int any = 0;
                for (int q = 0; q < NWS; ++q)
                    any |= len_full[((size_t)ps * c->n_chunks + ch) * NWS + q];
                if (any) active.push_back(ch);
            }
            active_off[ps + 1] = (int)active.size();
        }
        // row tables indexed by (active index, slice); a slice's rows of consecutive active
        // chunks are CONTIGUOUS in the row arrays (one stream per (pass, slice)), so a wave can
        // prefetch entries across chunk boundaries
        const size_t n_tab = std::max<size_t>(active.size(), 1) * NWS;
        std::vector<int> row_len(n_tab, 0), row_off(n_tab, 0);
        size_t rows = 0;
        for (int ps = 0; ps < c->n_pass; ++ps)
            for (int q = 0; q < NWS; ++q)
                for (int ai = active_off[ps]; ai < active_off[ps + 1]; ++ai) {
                    const int ch = active[ai];
                    const int l = len_full[((size_t)ps * c->n_chunks + ch) * NWS + q];
                    row_len[(size_t)ai * NWS + q] = l;
                    row_off[(size_t)ai * NWS + q] = (int)rows;
                    rows += l;
                }
        c->n_rows = rows;
        // chunk -> active index (per pass)
        std::vector<int> ai_of((size_t)c->n_pass * c->n_chunks, -1);
        for (int ps = 0; ps < c->n_pass; ++ps)
            for (int ai = active_off[ps]; ai < active_off[ps + 1]; ++ai)
                ai_of[(size_t)ps * c->n_chunks + active[ai]] = ai;
        if (active.empty()) active.push_back(0);
        // (padding entries and the prefetch slack: the zero pixel row SP_P, value 0)
        std::vector<uint32_t> pix((rows + 4 * SP_U) * 64, (uint32_t)SP_P);
        std::vector<float> val(c->f64 ? 0 : (rows + 4 * SP_U) * 64 * nc, 0.f);
        std::vector<double> val64(c->f64 ? (rows + 4 * SP_U) * 64 : 0, 0.);
        std::vector<int> fill((size_t)c->n_chunks * n_masks, 0);
        for (int64_t p = 0; p < n_px; ++p) {
            const int ch = (int)(p / SP_P);
            for (int64_t e = indptr[p]; e < indptr[p + 1]; ++e) {
                const int64_t k = indices[e];
                const int ps = (int)(k / c->mb);
                const int ai = ai_of[(size_t)ps * c->n_chunks + ch];
                const int lane = (int)(k % 64);
                const int j = fill[(size_t)ch * n_masks + k]++;
                const size_t pos = ((size_t)row_off[(size_t)ai * NWS + slice_of(k)] + j) * 64 + lane;
                pix[pos] = (uint32_t)(p - (int64_t)ch * SP_P);
                if (c->f64) val64[pos] = int_result ? (double)vals_i[e] : vals64[e];
                else
                    for (int q = 0; q < nc; ++q) val[pos * nc + q] = vals[e * nc + q];
            }
        }
        const size_t n_slices_dev = n_tab;
        hipError_t e = hipMalloc((void **)&c->pix, pix.size() * sizeof(uint32_t));
        if (e == hipSuccess && !c->f64) e = hipMalloc((void **)&c->val, val.size() * sizeof(float));
        if (e == hipSuccess && c->f64) e = hipMalloc((void **)&c->val64, val64.size() * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void **)&c->row_off, n_slices_dev * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **)&c->row_len, n_slices_dev * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **)&c->active, active.size() * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **)&c->active_off, active_off.size() * sizeof(int));
        if (e == hipSuccess) e = hipMemcpy(c->active, active.data(), active.size() * sizeof(int), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(c->active_off, active_off.data(), active_off.size() * sizeof(int), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(c->pix, pix.data(), pix.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess && !c->f64) e = hipMemcpy(c->val, val.data(), val.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess && c->f64) e = hipMemcpy(c->val64, val64.data(), val64.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(c->row_off, row_off.data(), n_slices_dev * sizeof(int), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(c->row_len, row_len.data(), n_slices_dev * sizeof(int), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            ltmi::csr_destroy(m);
            delete m;
            LTMI_FAIL((int)e, "uploading the sparse mask image failed: %s", hipGetErrorString(e));
        }
    } catch (const std::bad_alloc &) {
        ltmi::csr_destroy(m);
        delete m;
        LTMI_FAIL(LTMI_E_NOMEM, "out of host memory while packing the sparse mask image");
    }
    // Localised stacks (neighbouring masks share pixels: rings, radial bins) also get the blocked
    // image of ltmi_bell.hip; the padding factor decides.  LTMI_SPARSE_BELL=0 / 1 forces never /
    // always (tests), LTMI_BELL_MAX_RATIO moves the threshold.
    {
        const char *force = getenv("LTMI_SPARSE_BELL");
        const char *thr = getenv("LTMI_BELL_MAX_RATIO");
        const double max_ratio = thr ? atof(thr) : 8.0;
        bool build = nnz > 0 && !c->f64 && !g_gather_only;   // (the blocked image is float32 only)
        if (force && force[0] == '0') build = false;
        else if (build) {
            c->bell_ratio = ltmi::bell_mac_ratio(indptr, indices, nc, n_px, n_masks);
            if (!(force && force[0] == '1')) build = c->bell_ratio <= max_ratio;
        }
        if (build) {
            int err = LTMI_OK;
            c->bell = ltmi::bell_build(indptr, indices, vals, nc, n_px, n_masks, &err);
            if (!c->bell) {
                ltmi::csr_destroy(m);
                delete m;
                return err;
            }
        }
    }
    // Stacks with at least 64 columns whose entries cluster in runs of neighbouring columns (radial bins,
    // rings, any banded stack: >= 1.2 stored entries per 8-column window) also get the scatter image
    // (ltmi_scatter.hip: exact float32 FMAs on the vector ALUs, every 1- / 2- / 4-byte pixel type).
    // LTMI_SPARSE_SCATTER=0 / 1 forces never / always.
    {
        const char *force = getenv("LTMI_SPARSE_SCATTER");
        bool build = nnz > 0 && !c->f64 && n_masks * nc >= 64;
        if (g_gather_only || (force && force[0] == '0')) build = false;
        else if (force && force[0] == '1') { build = nnz > 0 && !c->f64; c->scat_all = true; }
        else if (build) build = ltmi::scat_fill(indptr, indices, nc, n_px, n_masks) >= 0.15;
        if (build) {
            int err = LTMI_OK;
            c->scat = ltmi::scat_build(indptr, indices, vals, nc, n_px, n_masks, &err);
            // (a failed build = no scatter image, like a failed float16 blocked image: the other kernels serve)
            if (!c->scat) (void)hipGetLastError();
        }
    }
    c->nnz_real = (double)nnz * nc;
    if (!c->f64 && !int_result && !g_gather_only) c->kept = ltmi::band_keep_csr(indptr, indices, vals, nc, n_px, n_masks);
    *out = m;
    return LTMI_OK;
}
