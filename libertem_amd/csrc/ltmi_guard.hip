// Non-finite pixels (NaN / Inf in float32 / float64 frames) and stacks whose fast kernels do not multiply exactly
// the entries the stack holds.  gfx950 only.
//
// The reference has two arithmetics, and which zeros meet a non-finite pixel differs between them:
//   * a SPARSE stack (use_sparse = 'scipy.sparse[.csr|.csc]' / 'sparse.pydata'): the loops of
//     src/libertem/common/numba/__init__.py:153-184 (`res_t[col, :] += left[:, row] * val` per STORED entry; pydata
//     path udf/masks.py:71-74) -- a non-finite pixel reaches exactly the masks that store it;
//   * a DENSE stack (use_sparse = False): `flat_tile @ masks` / torch.mm (udf/masks.py:59-66, :76-77) -- every weight
//     is multiplied, 0 * NaN = NaN: a non-finite pixel reaches every mask (NaN wherever the mask's weight is 0).
// The fast kernels of this library blur that line in both directions: blocked / scatter / densified / banded /
// folded images of a sparse stack multiply padding zeros (a NaN pixel poisons masks that do not store it), and the
// banded image of a dense stack skips the zeros between the column blocks (a NaN pixel misses masks it should hit).
// On finite frames all of them give the same sums, so the fast kernels stay as they are and this file adds
// detect-and-redo behind ltmi_apply_masks:
//   1. the product runs on the fast kernel;
//   2. k_flag_rows reads the RESULT rows (n_masks values per frame, ~1 % of the frame's bytes) and lists the frames
//      with a non-finite result; for a dense stack held as CSR also k_scan_unstored: the pixels NO mask stores, which
//      no kernel of that handle reads;
//   3. sparse stack: the listed frames are computed again by the gather kernel k_sell_apply, which only touches stored
//      entries (ltmi_sparse.hip csr_redo) -- launched over all frames, workgroups beyond the list (its length stays
//      on the device: no host synchronisation) leave at once;
//      dense stack as CSR: k_dense_fixup counts, per listed frame, the non-finite pixels and how many of them every
//      mask stores; a mask that does not store all of them is NaN (0 * NaN), the others keep the kernel's sums.
// Clean data pays step 2 only.  Integer frames cannot hold a non-finite pixel and are never guarded.
#include "ltmi_common.h"
#include <string.h>
#include <new>

namespace ltmi {

bool csr_has_fast_image(const ltmi_masks *m);      // ltmi_sparse.hip

struct NfGuard {
    int *ctl = nullptr;          // [0] = number of listed frames, [1 .. 1 + cap) = per-frame "listed" flags
    int32_t *list = nullptr;     // listed frames (result rows)
    int64_t cap = 0;
    void *scratch = nullptr;     // accumulate != 0 / `out` in host memory: the product lands here first
    size_t scratch_bytes = 0;
    const void *last_out = nullptr;   // (one-entry cache of out_is_host)
    bool last_out_host = false;
    bool last_checked = false;        // the last guarded product went through k_flag_rows (ltmi_masks_nonfinite_frames)
};

struct DenseOrigin {
    int32_t *indptr = nullptr;   // (n_px + 1) CSR over pixels
    int32_t *indices = nullptr;  // (nnz) mask index
    int32_t *unstored = nullptr; // pixels no mask stores
    int64_t n_unstored = 0;
};

template <typename A> struct NfBits;
template <> struct NfBits<float> {
    static __device__ __forceinline__ bool bad(float v) {
        return (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u;
    }
};
template <> struct NfBits<double> {
    static __device__ __forceinline__ bool bad(double v) {
        return ((unsigned)__double2hiint(v) & 0x7ff00000u) == 0x7ff00000u;
    }
};

__device__ __forceinline__ void nf_list_frame(int *ctl, int32_t *list, int64_t frame) {
    if (atomicExch(&ctl[1 + frame], 1) == 0) list[atomicAdd(&ctl[0], 1)] = (int32_t)frame;
}

// one wavefront per result row: n_cols real values (complex results: 2 per mask)
template <typename A>
__global__ void __launch_bounds__(256)
k_flag_rows(const A *__restrict__ out, int64_t ld, int64_t n_frames, int n_cols, int *__restrict__ ctl,
            int32_t *__restrict__ list) {
    const int lane = threadIdx.x & 63;
    const int64_t frame = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (frame >= n_frames) return;
    const A *row = out + frame * ld;
    bool bad = false;
    for (int c = lane; c < n_cols; c += 64) bad |= NfBits<A>::bad(row[c]);
    if (__ballot(bad) != 0ull && lane == 0) nf_list_frame(ctl, list, frame);
}

// the pixels no mask stores (dense stack as CSR): one workgroup per frame
template <typename T>
__global__ void __launch_bounds__(256)
k_scan_unstored(const T *__restrict__ tile, int64_t ld, int64_t n_frames, const int32_t *__restrict__ rows,
                const int32_t *__restrict__ px, int64_t n_px_list, int *__restrict__ ctl,
                int32_t *__restrict__ list) {
    __shared__ int any;
    const int64_t frame = blockIdx.x;
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    const T *row = tile + (rows ? (int64_t)rows[frame] : frame) * ld;
    bool bad = false;
    for (int64_t i = threadIdx.x; i < n_px_list; i += 256) bad |= NfBits<T>::bad(row[px[i]]);
    if (bad) any = 1;
    __syncthreads();
    if (threadIdx.x == 0 && any) nf_list_frame(ctl, list, frame);
}

// dense semantics for the listed frames of a CSR-held dense stack: out[f, k] = NaN unless mask k stores EVERY
// non-finite pixel of frame f (then the kernel's sum already is the dense product: nothing was skipped)
template <typename T>
__global__ void __launch_bounds__(256)
k_dense_fixup(const T *__restrict__ tile, int64_t ld, int64_t n_px, const int32_t *__restrict__ rows,
              const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices, int n_masks, int nc,
              float *__restrict__ out, int64_t ld_out, const int *__restrict__ ctl,
              const int32_t *__restrict__ list) {
    extern __shared__ int cnt[];                     // [n_masks] stored non-finite pixels, [n_masks] = all of them
    const int n_listed = ctl[0];
    for (int j = blockIdx.x; j < n_listed; j += gridDim.x) {
        const int64_t frame = list[j];
        const T *row = tile + (rows ? (int64_t)rows[frame] : frame) * ld;
        for (int k = threadIdx.x; k <= n_masks; k += 256) cnt[k] = 0;
        __syncthreads();
        for (int64_t p = threadIdx.x; p < n_px; p += 256) {
            if (!NfBits<T>::bad(row[p])) continue;
            atomicAdd(&cnt[n_masks], 1);
            for (int e = indptr[p]; e < indptr[p + 1]; ++e) atomicAdd(&cnt[indices[e]], 1);
        }
        __syncthreads();
        const int total = cnt[n_masks];
        if (total > 0) {
            float *o = out + frame * ld_out;
            for (int k = threadIdx.x; k < n_masks; k += 256)
                if (cnt[k] < total)
                    for (int c = 0; c < nc; ++c) o[(int64_t)k * nc + c] = __uint_as_float(0x7fc00000u);
        }
        __syncthreads();
    }
}

// result rows from the device scratch to their place (32-bit words: float32 / complex64 / float64 / complex128 rows)
__global__ void __launch_bounds__(256)
k_guard_copy_rows(const uint32_t *__restrict__ src, int64_t ld_src, uint32_t *__restrict__ dst, int64_t ld_dst,
                  int64_t n_rows, int n_words) {
    const int64_t total = n_rows * n_words;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / n_words;
        const int c = (int)(i - r * n_words);
        dst[r * ld_dst + c] = src[r * ld_src + c];
    }
}

// Does `out` point into HOST memory?  Small write-once result rows are written by the kernels straight into the run's
// page-locked host buffer (ltmi_host_device_pointer): the guard's second look at the rows would cross the host link and
// the redo's atomic adds would depend on PCIe atomics -- such products are checked in the device scratch and copied.
static bool out_is_host(NfGuard *g, const void *out) {
    if (g->last_out == out) return g->last_out_host;
    hipPointerAttribute_t at;
    bool host = false;
    if (hipPointerGetAttributes(&at, out) == hipSuccess) host = at.type == hipMemoryTypeHost;
    else (void)hipGetLastError();
    g->last_out = out;
    g->last_out_host = host;
    return host;
}

static bool guard_enabled() {
    static const bool on = [] {
        const char *e = getenv("LTMI_NONFINITE_GUARD");      // 0: the fast kernels' own patterns (timing ablation)
        return !(e && e[0] == '0');
    }();
    return on;
}

bool guard_wanted(const ltmi_masks *m, int tile_dtype) {
    if (tile_dtype != LTMI_F32 && tile_dtype != LTMI_F64) return false;
    if (!guard_enabled()) return false;
    if (m->sparse_origin || m->dense_origin) return true;
    // a CSR handle: the gather kernel of float64 results multiplies stored entries only; the float32 / complex64
    // routes (blocked, scatter, banded images) do not
    return m->kind == 2 && !csr_is_f64(m) && csr_has_fast_image(m);
}

static void dense_origin_free(ltmi_masks *m) {
    if (DenseOrigin *d = (DenseOrigin *)m->dense_origin) {
        if (d->indptr) (void)hipFree(d->indptr);
        if (d->indices) (void)hipFree(d->indices);
        if (d->unstored) (void)hipFree(d->unstored);
        delete d;
        m->dense_origin = nullptr;
    }
}

// a product that bypasses the guard (integer frames, guard switched off): ltmi_masks_nonfinite_frames reports 0 for it
void guard_note_unchecked(ltmi_masks *m) {
    if (NfGuard *g = (NfGuard *)m->guard) g->last_checked = false;
}

void guard_destroy(ltmi_masks *m) {
    if (m->sparse_origin) {
        (void)ltmi_masks_destroy(m->sparse_origin);
        m->sparse_origin = nullptr;
    }
    dense_origin_free(m);
    if (NfGuard *g = (NfGuard *)m->guard) {
        if (g->ctl) (void)hipFree(g->ctl);
        if (g->list) (void)hipFree(g->list);
        if (g->scratch) (void)hipFree(g->scratch);
        delete g;
        m->guard = nullptr;
    }
}

static int guard_ensure(ltmi_masks *m, int64_t n_frames, size_t scratch_bytes, hipStream_t stream, NfGuard **out) {
    NfGuard *g = (NfGuard *)m->guard;
    if (!g) {
        g = new (std::nothrow) NfGuard();
        if (!g) LTMI_FAIL(LTMI_E_NOMEM, "out of host memory");
        m->guard = g;
    }
    if (g->cap < n_frames) {
        if (g->ctl) {
            LTMI_HIP(hipStreamSynchronize(stream));       // (a launch in flight may still read the old lists)
            (void)hipFree(g->ctl);
            (void)hipFree(g->list);
            g->ctl = nullptr;
            g->list = nullptr;
            g->cap = 0;
        }
        LTMI_HIP(hipMalloc((void **)&g->ctl, (size_t)(n_frames + 1) * sizeof(int)));
        LTMI_HIP(hipMalloc((void **)&g->list, (size_t)n_frames * sizeof(int32_t)));
        g->cap = n_frames;
    }
    if (g->scratch_bytes < scratch_bytes) {
        if (g->scratch) {
            LTMI_HIP(hipStreamSynchronize(stream));
            (void)hipFree(g->scratch);
            g->scratch = nullptr;
            g->scratch_bytes = 0;
        }
        LTMI_HIP(hipMalloc(&g->scratch, scratch_bytes));
        g->scratch_bytes = scratch_bytes;
    }
    *out = g;
    return LTMI_OK;
}

int guard_apply(ltmi_masks *m, const void *tile, int tile_dtype, int64_t n_frames, int64_t ld_tile, void *out,
                int64_t ld_out, int accumulate, hipStream_t stream) {
    if (n_frames >= (1ll << 31)) LTMI_FAIL(LTMI_E_SHAPE, "ltmi_apply_masks: too many frames in one call");
    const size_t elem = (size_t)dtype_size(m->result_dtype);
    NfGuard *g = nullptr;
    int rc = guard_ensure(m, n_frames, 0, stream, &g);
    if (rc != LTMI_OK) return rc;
    // `out += product`: a NaN already in `out` is not this product's; the product is checked on its own.  `out` in host
    // memory (directly written result rows): checked in the device scratch, then copied
    const bool via_scratch = accumulate || out_is_host(g, out);
    if (via_scratch) {
        rc = guard_ensure(m, n_frames, (size_t)n_frames * m->n_masks * elem, stream, &g);
        if (rc != LTMI_OK) return rc;
    }
    void *target = via_scratch ? g->scratch : out;
    const int64_t ld_t = via_scratch ? m->n_masks : ld_out;
    rc = apply_masks_unguarded(m, tile, tile_dtype, n_frames, ld_tile, target, ld_t, 0, stream);
    if (rc != LTMI_OK) return rc;
    const bool exact = m->kind == 2 && m->last_exact && !m->dense_origin;   // the gather kernel ran: nothing to check
    g->last_checked = !exact;
    if (!exact) {
        const bool f64 = m->result_dtype == LTMI_F64 || m->result_dtype == LTMI_C128;
        const bool cplx = m->result_dtype == LTMI_C64 || m->result_dtype == LTMI_C128;
        const int n_cols = (int)(m->n_masks * (cplx ? 2 : 1));
        const int64_t ld_real = ld_t * (cplx ? 2 : 1);
        LTMI_HIP(hipMemsetAsync(g->ctl, 0, (size_t)(n_frames + 1) * sizeof(int), stream));
        const dim3 fgrid((unsigned)((n_frames + 3) / 4));
        if (f64)
            hipLaunchKernelGGL(k_flag_rows<double>, fgrid, dim3(256), 0, stream, (const double *)target, ld_real,
                               n_frames, n_cols, g->ctl, g->list);
        else
            hipLaunchKernelGGL(k_flag_rows<float>, fgrid, dim3(256), 0, stream, (const float *)target, ld_real,
                               n_frames, n_cols, g->ctl, g->list);
        LTMI_HIP(hipGetLastError());
        if (DenseOrigin *d = (DenseOrigin *)m->dense_origin) {
            if (tile_dtype != LTMI_F32 || f64)
                LTMI_FAIL(LTMI_E_DTYPE, "a dense stack held as CSR takes float32 frames and float32 / complex64 results");
            const float *t = (const float *)tile;
            if (d->n_unstored > 0) {
                hipLaunchKernelGGL(k_scan_unstored<float>, dim3((unsigned)n_frames), dim3(256), 0, stream, t, ld_tile,
                                   n_frames, m->roi_rows, (const int32_t *)d->unstored, d->n_unstored, g->ctl,
                                   g->list);
                LTMI_HIP(hipGetLastError());
            }
            const size_t lds = (size_t)(m->n_masks + 1) * sizeof(int);
            const unsigned blocks = (unsigned)std::min<int64_t>(n_frames, 2048);
            hipLaunchKernelGGL(k_dense_fixup<float>, dim3(blocks), dim3(256), lds, stream, t, ld_tile, m->n_px,
                               m->roi_rows, (const int32_t *)d->indptr, (const int32_t *)d->indices, (int)m->n_masks,
                               cplx ? 2 : 1, (float *)target, ld_real, (const int *)g->ctl, (const int32_t *)g->list);
            LTMI_HIP(hipGetLastError());
        } else {
            ltmi_masks *redo = m->sparse_origin ? m->sparse_origin : m;
            // (the row in bytes is the same for both handles; a complex128 stack's gather image counts float64 columns)
            const int64_t ld_redo = ld_t * (int64_t)elem / dtype_size(redo->result_dtype);
            rc = csr_redo(redo, tile, tile_dtype, n_frames, ld_tile, target, ld_redo, g->list, g->ctl, m->roi_rows,
                          stream);
            if (rc != LTMI_OK) return rc;
        }
        const size_t len = strlen(m->last_kernel);
        snprintf(m->last_kernel + len, sizeof(m->last_kernel) - len, " +nf");
    }
    if (accumulate)
        return ltmi_add2d(m->device, out, ld_out, g->scratch, m->n_masks, m->result_dtype, n_frames, m->n_masks, 0,
                          (void *)stream);
    if (via_scratch) {
        const int words = (int)(m->n_masks * elem / 4);
        const int64_t total = n_frames * words;
        const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(k_guard_copy_rows, dim3(blocks), dim3(256), 0, stream, (const uint32_t *)g->scratch,
                           (int64_t)words, (uint32_t *)out, ld_out * (int64_t)elem / 4, n_frames, words);
        LTMI_HIP(hipGetLastError());
    }
    return LTMI_OK;
}

}  // namespace ltmi

using namespace ltmi;

extern "C" int ltmi_masks_nonfinite_frames(ltmi_masks *m, void *stream_, int64_t *count) {
    if (!m || !count) LTMI_FAIL(LTMI_E_INVALID, "ltmi_masks_nonfinite_frames: null argument");
    *count = 0;
    const NfGuard *g = (const NfGuard *)m->guard;
    if (!g || !g->ctl || !g->last_checked) return LTMI_OK;        // the last product was not checked: nothing listed
    LTMI_HIP(hipSetDevice(m->device));
    int n = 0;
    LTMI_HIP(hipMemcpyAsync(&n, g->ctl, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream_));
    LTMI_HIP(hipStreamSynchronize((hipStream_t)stream_));
    *count = n;
    return LTMI_OK;
}

extern "C" int ltmi_masks_set_sparse_origin(ltmi_masks *m, ltmi_masks *gather) {
    if (!m || !gather) LTMI_FAIL(LTMI_E_INVALID, "ltmi_masks_set_sparse_origin: null handle");
    if (m->kind == 2 || gather->kind != 2 || m == gather)
        LTMI_FAIL(LTMI_E_INVALID, "ltmi_masks_set_sparse_origin: (dense handle, handle of ltmi_masks_create_csr)");
    if (m->device != gather->device || m->n_px != gather->n_px ||
        m->n_masks * dtype_size(m->result_dtype) != gather->n_masks * dtype_size(gather->result_dtype))
        LTMI_FAIL(LTMI_E_SHAPE, "ltmi_masks_set_sparse_origin: the handles do not describe the same stack "
                  "(%lld x %lld %s on device %d against %lld x %lld %s on device %d)", (long long)m->n_masks,
                  (long long)m->n_px, dtype_name(m->result_dtype), m->device, (long long)gather->n_masks,
                  (long long)gather->n_px, dtype_name(gather->result_dtype), gather->device);
    const bool fl = m->result_dtype == LTMI_F32 || m->result_dtype == LTMI_F64 || m->result_dtype == LTMI_C64 ||
                    m->result_dtype == LTMI_C128;
    if (!fl) LTMI_FAIL(LTMI_E_DTYPE, "ltmi_masks_set_sparse_origin: float / complex results only");
    if (m->sparse_origin) (void)ltmi_masks_destroy(m->sparse_origin);
    m->sparse_origin = gather;
    return LTMI_OK;
}

extern "C" int ltmi_masks_set_dense_origin(ltmi_masks *m, const int64_t *indptr, const int64_t *indices) {
    if (!m || !indptr) LTMI_FAIL(LTMI_E_INVALID, "ltmi_masks_set_dense_origin: null argument");
    if (m->kind != 2 || (m->result_dtype != LTMI_F32 && m->result_dtype != LTMI_C64))
        LTMI_FAIL(LTMI_E_INVALID, "ltmi_masks_set_dense_origin: a float32 / complex64 handle of ltmi_masks_create_csr");
    const int64_t nnz = indptr[m->n_px];
    if (nnz < 0 || nnz >= (1ll << 31) || (nnz > 0 && !indices) || m->n_px >= (1ll << 31))
        LTMI_FAIL(LTMI_E_SHAPE, "ltmi_masks_set_dense_origin: %lld stored entries", (long long)nnz);
    if (m->n_masks + 1 > 15 * 1024)
        LTMI_FAIL(LTMI_E_SHAPE, "ltmi_masks_set_dense_origin: at most %d masks", 15 * 1024 - 1);
    LTMI_HIP(hipSetDevice(m->device));
    DenseOrigin *d = new (std::nothrow) DenseOrigin();
    if (!d) LTMI_FAIL(LTMI_E_NOMEM, "out of host memory");
    std::vector<int32_t> ip((size_t)m->n_px + 1), ix((size_t)std::max<int64_t>(nnz, 1)), un;
    for (int64_t p = 0; p <= m->n_px; ++p) ip[(size_t)p] = (int32_t)indptr[p];
    for (int64_t e = 0; e < nnz; ++e) {
        if (indices[e] < 0 || indices[e] >= m->n_masks) {
            delete d;
            LTMI_FAIL(LTMI_E_SHAPE, "ltmi_masks_set_dense_origin: column index out of range");
        }
        ix[(size_t)e] = (int32_t)indices[e];
    }
    for (int64_t p = 0; p < m->n_px; ++p)
        if (indptr[p + 1] == indptr[p]) un.push_back((int32_t)p);
    hipError_t e = hipMalloc((void **)&d->indptr, ip.size() * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&d->indices, ix.size() * sizeof(int32_t));
    if (e == hipSuccess && !un.empty()) e = hipMalloc((void **)&d->unstored, un.size() * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemcpy(d->indptr, ip.data(), ip.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d->indices, ix.data(), ix.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && !un.empty())
        e = hipMemcpy(d->unstored, un.data(), un.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    d->n_unstored = (int64_t)un.size();
    dense_origin_free(m);
    m->dense_origin = d;
    if (e != hipSuccess) {
        dense_origin_free(m);
        LTMI_FAIL((int)e, "ltmi_masks_set_dense_origin: %s", hipGetErrorString(e));
    }
    {
        const size_t lds = (size_t)(m->n_masks + 1) * sizeof(int);
        if (lds > 48 * 1024)
            LTMI_HIP(hipFuncSetAttribute((const void *)k_dense_fixup<float>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    return LTMI_OK;
}
