// Shared host-side helpers for libltmi (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string>
#include <algorithm>
#include <vector>
#include "../../include/ltmi.h"

namespace ltmi {

void set_error(const char *fmt, ...);

#define LTMI_HIP(expr)                                                                   \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            ltmi::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),       \
                            __FILE__, __LINE__);                                         \
            return (int)_e;                                                              \
        }                                                                                \
    } while (0)

#define LTMI_FAIL(code, ...)                                                             \
    do {                                                                                 \
        ltmi::set_error(__VA_ARGS__);                                                    \
        return (code);                                                                   \
    } while (0)

// gfx950 serves 16-byte vector loads and LDS-DMA loads from any element-aligned address at the speed
// of aligned ones (profiles/r02_unaligned.txt), so tiles whose rows are not 16-B aligned (odd pixel
// counts, e.g. 515 x 515 detectors) take the same vectorised kernels.  LTMI_ALIGNED_DMA_ONLY=1 in the
// environment restores the conservative dispatch (vector paths for 16-B aligned rows only).
static inline bool vector_loads_ok(const void *tile, int64_t ld_elems, size_t elem) {
    static const bool aligned_only = getenv("LTMI_ALIGNED_DMA_ONLY") != nullptr;
    if ((((uintptr_t)tile) % 16 == 0) && ((ld_elems * (int64_t)elem) % 16 == 0)) return true;
    return !aligned_only && ((uintptr_t)tile) % elem == 0;
}

// K split of the LDS-DMA kernels (one workgroup per CU at a time, 256 CUs): `wgs` workgroups at
// ksplit = 1, each walking `n_slots` mask slots.  A split by ks makes ks x wgs workgroups of n_slots / ks
// slots; the launch takes ceil(ks wgs / 256) rounds of (n_slots / ks + start-up) plus, for ks > 1, the
// reduction of the partials.  Picks the ks with the shortest modelled time: fills the chip when there
// are fewer workgroups than CUs AND trims the last, partly filled round of awkward frame counts (313
// workgroups: 2 rounds unsplit = 2.0, ks = 4: 5 rounds of a quarter = 1.25 of one workgroup's time).
static inline int choose_ksplit(int64_t wgs, int n_slots) {
    const double START = 1.5, REDUCE = 6.0;             // in slot times (~3 us / ~14 us on C2)
    int max_ks = n_slots / 8 > 1 ? (n_slots / 8 < 64 ? n_slots / 8 : 64) : 1;
    if (wgs < 256) {
        // fewer workgroups than CUs: up to ~4 rounds' worth of workgroups
        max_ks = (int)std::min<int64_t>(max_ks, (1024 + wgs - 1) / wgs);
    } else {
        // whole rounds, or a last round that is at least 3/4 full: nothing to trim
        const int64_t tail = wgs % 256;
        if (tail == 0 || tail >= 192 || wgs > 4 * 256) return 1;
        max_ks = std::min(max_ks, 8);
    }
    double best = 1e300;
    int best_ks = 1;
    for (int ks = 1; ks <= max_ks; ++ks) {
        if (wgs * ks > 16384 && ks > 1) break;
        const double rounds = (double)((wgs * ks + 255) / 256);
        // (a launch of few rounds runs at the pace of its slowest CUs: ~8 % on a single round)
        const double t = rounds * ((double)n_slots / ks + START) * (1.0 + 0.08 / rounds) +
                         (ks > 1 ? REDUCE : 0.0);
        if (t < best * 0.98) {                          // a larger split has to be worth >= 2 %
            best = t;
            best_ks = ks;
        }
    }
    return best_ks;
}

static inline int dtype_size(int dt) {
    switch (dt) {
        case LTMI_BOOL: case LTMI_U8: case LTMI_I8: return 1;
        case LTMI_U16: case LTMI_I16: return 2;
        case LTMI_U32: case LTMI_I32: case LTMI_F32: return 4;
        case LTMI_U64: case LTMI_I64: case LTMI_F64: case LTMI_C64: return 8;
        case LTMI_C128: return 16;
    }
    return 0;
}

static inline const char *dtype_name(int dt) {
    static const char *n[] = {"bool", "uint8", "int8", "uint16", "int16", "uint32", "int32",
                              "uint64", "int64", "float32", "float64", "complex64", "complex128"};
    return (dt >= 0 && dt <= LTMI_C128) ? n[dt] : "?";
}

struct cfloat { float re, im; };
struct cdouble { double re, im; };

}  // namespace ltmi

struct ltmi_masks;
namespace ltmi {
int dense64_create(ltmi_masks *m);
void dense64_destroy(ltmi_masks *m);
int dense64_apply(ltmi_masks *m, const void *tile, int tile_dtype, int64_t n_frames, int64_t ld,
                  void *out, int64_t ld_out, int accumulate, hipStream_t stream, bool *handled);
bool dense64_rows_ok(const ltmi_masks *m, const void *tile, int tile_dtype, int64_t ld);
size_t dense64_image_bytes(const ltmi_masks *m);
int dense64_build_shifted(ltmi_masks *m, int sig_h, int sig_w, int dy, int dx, double *img,
                          hipStream_t stream);
// blocked-ELL sparse image on the matrix cores (ltmi_bell.hip)
double bell_mac_ratio(const int64_t *indptr, const int64_t *indices, int nc, int64_t n_px,
                      int64_t n_masks);
void *bell_build(const int64_t *indptr, const int64_t *indices, const float *vals, int nc,
                 int64_t n_px, int64_t n_masks, int *err);
void bell_destroy(void *image);
// sparse stacks on the vector ALUs, one float32 FMA per stored entry (ltmi_scatter.hip)
double scat_fill(const int64_t *indptr, const int64_t *indices, int nc, int64_t n_px, int64_t n_masks);
void *scat_build(const int64_t *indptr, const int64_t *indices, const float *vals, int nc, int64_t n_px,
                 int64_t n_masks, int *err);
void scat_destroy(void *set);
int scat_apply(ltmi_masks *m, void *set, int cplx, const void *tile, int tile_dtype, int64_t n_frames,
               int64_t ld_tile, void *out, int64_t ld_out, int accumulate, hipStream_t stream, bool *handled);
// float32 frames x multi-group float32 stacks on the bf16 matrix cores, float32-accurate (ltmi_split.hip)
bool split_selected(bool tuned);
bool split_wanted(int n_cols, int64_t n_px);
int split_create(int device, const float *gmasks, int64_t n_masks, int cpm, int64_t n_px, int n_cols,
                 void **image);
void split_destroy(void *image);
size_t split_image_bytes(const void *image);
int split_apply(ltmi_masks *m, void *image, const float *tile, int64_t n_frames, int64_t ld, float *out,
                int64_t ld_out, int accumulate, hipStream_t stream);
// CrystallinityUDF for 256 x 256 frames in one kernel (ltmi_cryst.hip)
int cryst_fused_max_cols();
bool cryst_fused_shape(int h, int w);
bool cryst_fused_takes(int h, int w, int n_cols);
int64_t cryst_fused_workspace_floats(int h, int w);
bool cryst_fused_needs_gbuf(int h, int w, int n_cols);
int cryst_fused(const void *tile, int tile_dtype, int64_t n_frames, int64_t ld, int sig_h, int sig_w,
                const float *real_mask, const float *half_mask, int n_cols, float *mask_t, void *gbuf,
                int64_t gbuf_frames, float *out, int accumulate, int n_cu, hipStream_t stream, bool *handled);
// dense stacks folded about a mirror of the detector rows (ltmi_fold.hip)
int fold_create(ltmi_masks *m, int sig_h, int sig_w);
void fold_destroy(ltmi_masks *m);
bool fold_takes(const ltmi_masks *m, const float *tile, int64_t ld);
int launch_fold(ltmi_masks *m, const float *tile, int64_t n_frames, int64_t ld, float *out, int64_t ld_out,
                int accumulate, hipStream_t stream);
// banded sparse stacks (column blocks with a common support each: radial Fourier with several bins) on k_dense_fold
struct KeptCsr;                                                             // host copy of a CSR stack, rows sorted
KeptCsr *band_keep_csr(const int64_t *indptr, const int64_t *indices, const float *vals, int nc, int64_t n_px,
                       int64_t n_masks);                                    // nullptr: not a candidate
void band_free_csr(KeptCsr *k);
void *band_build(const KeptCsr *k, int sig_h, int sig_w, double other_macs);    // nullptr: the other kernels serve
void band_destroy(void *band);
bool band_takes(void *band, const ltmi_masks *m, const void *tile, int tile_dtype, int64_t ld);   // float32, 1- / 2-byte integers
int band_apply(ltmi_masks *m, void *band, const void *tile, int tile_dtype, int64_t n_frames, int64_t ld, float *out,
               int64_t ld_out, int n_cols, int accumulate, hipStream_t stream);
bool fold_takes16(ltmi_masks *m, const void *tile, int64_t ld, int px_bytes);   // 1- / 2-byte integer frames (image built on first use)
int launch_fold16(ltmi_masks *m, const void *tile, int px_bytes, bool is_signed, int64_t n_frames, int64_t ld,
                  float *out, int64_t ld_out, int accumulate, hipStream_t stream);
// the K-split workspace of a dense handle (ltmi_dense.hip)
int dense_ensure_partials(ltmi_masks *m, size_t need, hipStream_t stream);
float *dense_partial_sums(const ltmi_masks *m);
int dense_reduce_partials(ltmi_masks *m, int ksplit, int64_t n_frames, float *out, int64_t ld_out, int accumulate,
                          hipStream_t stream, int n_cols = -1);   // n_cols <= 0: the handle's
// ... on RAW frames with the detector corrections applied inside the row stage (256 x 256 frames)
int64_t cryst_corr_workspace_bytes(int h, int w, int64_t n_frames, int n_excl);
bool cryst_corr_takes(int h, int w, int n_cols, int tile_dtype, int n_excl);
int cryst_fused_corrected(const void *tile, int tile_dtype, int64_t n_frames, int64_t ld, int sig_h, int sig_w,
                          const double *dark, const double *gain, const int32_t *excl, const int32_t *env,
                          const int32_t *cnt, int n_excl, int max_env, const float *real_mask, const float *half_mask,
                          int n_cols, float *mask_t, void *gbuf, int64_t gbuf_frames, void *ws, float *out,
                          int accumulate, int n_cu, hipStream_t stream, bool *handled);
// non-finite pixels on sparse stacks (ltmi_guard.hip; the gather-kernel redo: ltmi_sparse.hip)
int csr_redo(ltmi_masks *m, const void *tile, int tile_dtype, int64_t max_frames, int64_t ld_tile, void *out,
             int64_t ld_out, const int32_t *sel, const int *n_sel, const int32_t *roi_rows, hipStream_t stream);
bool csr_is_f64(const ltmi_masks *m);
bool guard_wanted(const ltmi_masks *m, int tile_dtype);
int guard_apply(ltmi_masks *m, const void *tile, int tile_dtype, int64_t n_frames, int64_t ld_tile, void *out,
                int64_t ld_out, int accumulate, hipStream_t stream);
void guard_destroy(ltmi_masks *m);
// ltmi_apply_masks without the guard (ltmi_dense.hip)
int apply_masks_unguarded(ltmi_masks *m, const void *tile, int tile_dtype, int64_t n_frames, int64_t ld_tile,
                          void *out, int64_t ld_out, int accumulate, hipStream_t stream);
int bell_apply(ltmi_masks *m, void *image, int cplx, const void *tile, int tile_dtype,
               int64_t n_frames, int64_t ld_tile, void *out, int64_t ld_out, int accumulate,
               hipStream_t stream, bool *handled);
}  // namespace ltmi

// The opaque handle behind `ltmi_masks*` (shared by ltmi_dense.hip and ltmi_sparse.hip).
struct ltmi_masks {
    int device = 0;
    int kind = 0;            // 0 dense/MFMA-f32, 1 dense/generic, 2 csr (SELL image)
    int result_dtype = 0;
    int64_t n_masks = 0, n_px = 0;
    // kind 0
    int n_cols = 0;          // real f32 columns (2 per mask for complex64)
    int n_groups = 0;        // 16-column groups, padded to a multiple of ng
    int ng = 1;
    int n_chunks = 0;
    float *img = nullptr;
    float *img2 = nullptr;   // slot-major image for k_dense_lds with ng > 1 (KB = 128)
    int n_slots2 = 0;
    float *img3 = nullptr;   // ng3 groups + ne3 VALU columns (16 ng3 + 1..4 columns), slots of 128 px
    int n_slots3 = 0, ne3 = 0, ng3 = 0;   // ng3 full groups + ne3 VALU columns
    void *split = nullptr;   // bf16 x 3 image for float32 frames (ltmi_split.hip), stacks of >= 2 groups
    // float16 image of the standard (ng = 1) layout for unsigned 1- / 2-byte pixels (k_dense_lds X16):
    // w1 / w2 of the scaled weights in the two 16-byte units of a lane's 8 pixels; 1 / scale per column
    float *img_h = nullptr;
    float *img2_h = nullptr;     // ... of image 2 (ng > 1)
    float *img3_h = nullptr;     // ... of image 3 without VALU columns (3 groups)
    float *inv_scale = nullptr;
    // small weights that two float16 pieces do not carry to 2^-19 relative: left out of the float16
    // images; the epilogue of k_dense_lds X16 adds their float32 products
    int32_t *tail_px = nullptr, *tail_col = nullptr;
    float *tail_val = nullptr;
    int tail_n = 0;
    bool x16_used = false;
    // float64 results on the f64 matrix cores (ltmi_dense64.hip)
    double *img64 = nullptr;
    int n_groups64 = 0, n_chunks64 = 0;
    int cpm64 = 1;              // real columns per mask in the f64 image (2: complex128 masks)
    void *ws64 = nullptr;
    size_t ws64_bytes = 0;
    void *res64 = nullptr;   // f64 scratch result of the exact-integer path
    size_t res64_bytes = 0;
    int mask_bits = 64;      // integer stacks: bits needed for max |mask value|
    void *shift_cache = nullptr;   // ltmi_dense.hip: images of the stack shifted by (dy, dx)
    void *fold = nullptr;          // ltmi_fold.hip: image folded about a mirror of the detector rows (ltmi_masks_set_sig_shape)
    float *partials = nullptr;
    size_t partials_bytes = 0;
    int tune_mt = 0, tune_waves = 0, tune_ksplit = 0, tune_ksplit_ring = 0;
    // ltmi_apply_masks_rows: device list of the tile's frames to multiply (result row i = frame rows[i]);
    // set for the duration of that call only
    const int32_t *roi_rows = nullptr;
    // kind 0 and 1
    void *gmasks = nullptr;  // (n_masks, n_px) of the accumulate type
    // kind 2 (ltmi_sparse.hip)
    void *csr = nullptr;
    // kind 0 with more than 64 real columns: the stack again as column blocks of <= 64 columns,
    // each with the tile width that fits it (ltmi_apply_masks walks them; tuning code 33 does not)
    std::vector<ltmi_masks *> blocks;
    std::vector<int64_t> block_first;        // first mask of every block
    // non-finite pixels (ltmi_guard.hip)
    ltmi_masks *sparse_origin = nullptr;   // a DENSE handle that stands for a sparse stack: the stack's gather image (owned)
    void *dense_origin = nullptr;          // a CSR handle that stands for a dense stack: its by-pixel tables
    void *guard = nullptr;                 // flagged-frame list + scratch of the guarded product
    bool last_exact = false;               // the last product ran on a kernel that multiplies stored entries only
    char last_kernel[128] = {0};
};
