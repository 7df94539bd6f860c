/*
 * ltmi.h -- C ABI of libltmi.so: MI355X (gfx950) kernels for LiberTEM's mask-application /
 * virtual-detector hot path.
 *
 * The reference (LiberTEM, pure Python) has no FFI for this path; its operator boundary is the
 * Python UDF interface (`process_tile` / `merge`, src/libertem/udf/base.py:1334-1608).  Each entry
 * point below replaces the array-library call the reference makes *inside* that interface and
 * cites it.  All pointers named `*_dev` / `tile` / `out` are DEVICE pointers owned by the caller
 * (torch-ROCm tensors' data_ptr()); the library never allocates user-visible memory.  It owns only
 * the opaque mask handles (prepared device images of the mask stack + a partial-sum workspace).
 *
 * Conventions
 *   - every function returns 0 (LTMI_OK) on success, a negative LTMI_E_* code for argument errors,
 *     or a positive code: a hipError_t, or LTMI_E_RCCL_BASE + ncclResult_t from the ltmi_comm_*
 *     functions; `ltmi_last_error()` returns a thread-local message.
 *   - nothing throws across the ABI.
 *   - `stream` is a hipStream_t (NULL = the legacy default stream); all work is enqueued on it and
 *     the call returns without synchronising.  A handle must be used on one stream at a time
 *     (its workspace is reused by consecutive calls).
 *   - `device` is the HIP device ordinal; handles remember theirs and every call on a handle
 *     switches to it (hipSetDevice) first.
 *   - tiles are row-major `(n_frames, n_px)` with `ld_tile` ELEMENTS between consecutive frames,
 *     i.e. exactly the flattened C-contiguous tile `tile.reshape((n, -1))` of
 *     src/libertem/udf/masks.py:79-83.
 */
#ifndef LTMI_H
#define LTMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libltmi.so is built with -fvisibility=hidden: the functions declared between this push and the pop
 * at the end of the header are the WHOLE dynamic symbol table of the library (tests/test_runtime_cpu.py
 * compares `nm -D` with this header). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define LTMI_VERSION 1

#define LTMI_OK 0
#define LTMI_E_INVALID (-1)      /* bad argument (null pointer, negative size, ...) */
#define LTMI_E_DTYPE (-2)        /* dtype combination not supported */
#define LTMI_E_SHAPE (-3)        /* shapes do not match the handle */
#define LTMI_E_NOMEM (-4)        /* host allocation failed */
#define LTMI_E_RCCL_BASE 10000   /* a failing RCCL call returns LTMI_E_RCCL_BASE + ncclResult_t */

/* dtype codes (numpy names) */
enum ltmi_dtype {
    LTMI_BOOL = 0, LTMI_U8 = 1, LTMI_I8 = 2, LTMI_U16 = 3, LTMI_I16 = 4, LTMI_U32 = 5,
    LTMI_I32 = 6, LTMI_U64 = 7, LTMI_I64 = 8, LTMI_F32 = 9, LTMI_F64 = 10, LTMI_C64 = 11,
    LTMI_C128 = 12
};

typedef struct ltmi_masks ltmi_masks;      /* opaque */

/* ---- misc ----------------------------------------------------------------------------- */
int ltmi_version(void);
const char *ltmi_last_error(void);
int ltmi_device_count(int *count);
/* name_out: at least 256 bytes. replaces libertem.utils.devices.detect() (utils/devices.py:31-60) */
int ltmi_device_info(int device, char *name_out, int *cu_count, int64_t *hbm_bytes, int *gfx_arch);

/* ---- mask stacks ------------------------------------------------------------------------
 * Dense stack.  Replaces MaskContainer.get_for_sig_slice(sig_slice, transpose=True) +
 * for_backend(m, backend).astype(dtype)  (src/libertem/common/container.py:74-94, :213-217):
 * `masks_host` is the HOST array of the sig-sliced, flattened stack, logical shape
 * (n_masks, n_px), C-contiguous ("mask-major", which is also the reference's physical layout),
 * already cast to `result_dtype` = np.result_type(input_dtype, mask_dtype)
 * (src/libertem/udf/masks.py:362).  The library converts it to its own device image
 * (see DESIGN.md "HBM layout") and keeps no reference to `masks_host`.
 */
int ltmi_masks_create_dense(int device, const void *masks_host, int result_dtype,
                            int64_t n_masks, int64_t n_px, ltmi_masks **out);

/* Sparse stack in CSR over pixels, exactly the matrix the reference builds in
 * _build_sparse (src/libertem/common/container.py:53-64): shape (n_px, n_masks),
 * indptr[n_px + 1], indices[nnz] (mask index), data[nnz] of `result_dtype`: float32, complex64,
 * float64, or -- for the integer result dtypes -- int64 values: ltmi_apply_masks then computes the
 * integer product with wrap-around (SciPy's integer matmul) through the float64 gather kernel and
 * fails with LTMI_E_DTYPE for a tile dtype whose sums could exceed 2^52 (bits of the tile dtype + bits
 * of the largest column sum of |values| > 52: densify and use ltmi_masks_create_dense).
 * complex128: pass (re, im) as two float64 columns (libertem_amd/hip.py MaskHandle.csr_complex128).
 * Host pointers; canonical format (sorted, no duplicates) is not required.
 * Two device images may be built: the sliced-ELL image of the gather kernel (always) and, for float32 /
 * complex64 stacks whose neighbouring masks share pixels (rings, radial bins), the blocked image that
 * runs on the matrix cores; ltmi_apply_masks picks the blocked one when it exists.  Environment (read here): LTMI_SPARSE_BELL=0 / 1 never / always builds the blocked image,
 * LTMI_BELL_MAX_RATIO moves the padding-factor threshold (default 8).
 */
int ltmi_masks_create_csr(int device, const int64_t *indptr, const int64_t *indices,
                          const void *data, int result_dtype, int64_t n_px, int64_t n_masks,
                          ltmi_masks **out);

/* Tells a float32 / complex64 handle the detector shape behind its n_px = sig_h * sig_w pixels (C order),
 * which the reference's MaskContainer knows from the mask arrays it stacks (common/container.py:260-314) and
 * flattens away before the product (udf/masks.py:79-83).  The library then looks for a mirror of the detector
 * rows, y -> c2 - y with c2 in {sig_h - 1, sig_h, sig_h + 1}, under which EVERY real column of the stack is even
 * or odd bit for bit -- radial-Fourier masks ring * exp(i o phi) (analysis/radialfourier.py:106-146), rings,
 * disks, centre-of-mass ramps built around the detector centre are -- and, when that saves matrix work (two or
 * more 16-column groups), keeps a folded image: float32 frames are then multiplied as
 * (x[y] + s x[c2 - y]) * w[y] over half of the rows with the stack's original weights (k_dense_fold).  Results
 * differ from the unfolded product by float32 round-off only; a stack without such a mirror is left as it is.
 * A CSR handle (ltmi_masks_create_csr, float32 / complex64) whose masks fall into blocks with ONE pixel support each
 * -- the orders of a bin in a radial-Fourier stack with several bins (n_bins > 1, use_sparse=True) -- and has such
 * a mirror gets a folded DENSE image per block over the 64-pixel stages that touch the block's support
 * (k_dense_fold / k_dense_fold16 over stage lists; float32 and 1- / 2-byte integer frames) when the cost estimate
 * favours it over the blocked sparse image; ltmi_masks_kind then reports 3.  The handle keeps a host copy of the
 * CSR arrays until this call or its first product.
 * Optional; LTMI_DENSE_FOLD=0 / LTMI_SPARSE_BAND=0 in the environment disable the searches (LTMI_SPARSE_BAND=1:
 * whatever the estimate says). */
int ltmi_masks_set_sig_shape(ltmi_masks *m, int sig_h, int sig_w);

/* Non-finite pixels (NaN / Inf in float32 / float64 frames): which zeros of a stack meet them.
 * The reference has two arithmetics.  A SPARSE stack is multiplied entry by stored entry
 * (`res_t[col, :] += left[:, row] * val`, src/libertem/common/numba/__init__.py:153-184; pydata path
 * src/libertem/udf/masks.py:71-74): a non-finite pixel reaches exactly the masks that store it.  A DENSE stack is
 * multiplied whole (`flat_tile @ masks` / torch.mm, src/libertem/udf/masks.py:59-66, :76-77): 0 * NaN = NaN, a
 * non-finite pixel reaches every mask.  Handles follow the arithmetic of the constructor that made them
 * (ltmi_masks_create_dense: dense, ltmi_masks_create_csr: sparse) on EVERY kernel route: where a fast kernel of a
 * CSR handle also multiplies padding zeros (blocked, scatter, banded images), ltmi_apply_masks[_rows] reads the
 * result rows of float frames, lists the frames with a non-finite result on the device and computes those again
 * on the gather kernel (stored entries only) -- one small extra kernel on clean data, no host synchronisation
 * (csrc/ltmi_guard.hip; LTMI_NONFINITE_GUARD=0 in the environment switches it off for timing comparisons).
 * The two calls below are for a caller that hands a stack to the OTHER constructor because that kernel is faster:
 *
 * ltmi_masks_set_sparse_origin: `m` (from ltmi_masks_create_dense) holds a stack that the reference multiplies
 * SPARSE -- a well-filled CSR stack that was densified; `gather` is the handle of ltmi_masks_create_csr for the same
 * stack (same device, n_px, result row: float32 / complex64 / float64, or float64 column pairs for a complex128
 * stack).  `m` takes ownership of `gather` (destroyed with it) and its products then follow the sparse arithmetic as
 * above, whichever dense kernel runs (LDS-DMA, folded, float64).
 *
 * ltmi_masks_set_dense_origin: `m` (from ltmi_masks_create_csr, float32 / complex64, fewer than 15 360 masks) holds
 * the non-zeros of a stack that the reference multiplies DENSE -- column blocks with a pixel support each, which
 * the banded image serves faster (ltmi_masks_set_sig_shape).  indptr / indices: the CSR arrays given to
 * ltmi_masks_create_csr (host; not kept).  Frames with a non-finite result or a non-finite pixel that no mask
 * stores are listed, and in their rows every mask that does not store ALL non-finite pixels of the frame
 * becomes NaN (a skipped 0 * NaN); masks that store them all keep the kernel's sums, which then equal the dense
 * product.  float32 frames only. */
int ltmi_masks_set_sparse_origin(ltmi_masks *m, ltmi_masks *gather);
/* ltmi_masks_create_csr with the gather kernel's image only (no blocked / scatter / banded images are tried): the
 * handle to pass as `gather` above.  Same arguments, same result dtypes. */
int ltmi_masks_create_csr_gather(int device, const int64_t *indptr, const int64_t *indices,
                                 const void *data, int result_dtype, int64_t n_px, int64_t n_masks,
                                 ltmi_masks **out);
int ltmi_masks_set_dense_origin(ltmi_masks *m, const int64_t *indptr, const int64_t *indices);
/* How many frames of the LAST product on this handle were listed for the redo / fix-up above (frames with a non-finite
 * result, or -- dense stack held as CSR -- a non-finite pixel no mask stores).  Diagnostic; SYNCHRONISES `stream` (the
 * stream of that product).  0 when the product was not checked (integer frames, a handle that is not guarded, the
 * gather kernel).  The reference has no counterpart: its NaNs simply appear in the result. */
int ltmi_masks_nonfinite_frames(ltmi_masks *m, void *stream, int64_t *count);

int ltmi_masks_destroy(ltmi_masks *m);
/* 0 = dense with float32 / complex64 results (f32 matrix cores), 1 = dense with any other result
 * dtype (float64, complex128 on real tiles, exactly representable integer sums: f64 matrix cores;
 * complex tiles and wider integers: VALU kernel), 2 = csr, 3 = csr with a banded dense image (see
 * ltmi_masks_set_sig_shape) */
int ltmi_masks_kind(const ltmi_masks *m, int *kind);

/* ---- the hot call -----------------------------------------------------------------------
 * out[f, k] (+)= sum_p tile[f, p] * masks[k, p]
 * Replaces ApplyMasksEngine.process_flat (torch.mm / `flat_tile @ masks` / rmatmul:
 * src/libertem/udf/masks.py:59-77, src/libertem/common/numba/__init__.py:90-184) fused with the
 * dtype conversion of the tile (io/dataset/memory.py:102-105) and with the accumulation
 * `results.intensity[:] += ...` (src/libertem/udf/masks.py:389-392) when accumulate != 0.
 *   tile       device pointer, (n_frames, n_px) of `tile_dtype` (the dataset's NATIVE dtype --
 *              the astype(input_dtype) copy is fused), ld_tile elements between frames; element
 *              alignment is enough (rows of odd length: the 16-byte loads / LDS-DMA of gfx950 take
 *              any address; LTMI_ALIGNED_DMA_ONLY=1 in the environment keeps the vector paths to
 *              16-byte aligned rows)
 *   out        device pointer, (n_frames, n_masks) of the handle's result dtype, ld_out elements
 *   accumulate 0: out = product, 1: out += product
 */
int ltmi_apply_masks(ltmi_masks *m, const void *tile, int tile_dtype, int64_t n_frames,
                     int64_t ld_tile, void *out, int64_t ld_out, int accumulate, void *stream);

/* The same product for SELECTED frames of a tile, through a device list of frame numbers -- a region
 * of interest without the gathered copy of its frames (reference: the ROI is applied while reading,
 * frame by frame, io/dataset/memory.py:107-131; buffers are ROI-compressed, common/buffers.py:419-505):
 *   out[i, k] (+)= sum_p tile[rows[i], p] * masks[k, p],   0 <= i < n_rows
 * `rows`: DEVICE array of n_rows int32 frame numbers relative to `tile`.  *handled = 0 and nothing is
 * done when the handle / tile combination has no row-list kernel (then gather with ltmi_gather_rows and
 * call ltmi_apply_masks): row lists are served for dense float32 / complex64 stacks on uint8 / int8 /
 * uint16 / int16 / float32 tiles of at least one mask slot per frame, for dense float64 / complex128 /
 * exact-integer results of real tiles (>= 256 pixels per frame), and for sparse (CSR) stacks (both
 * sparse kernels, float32 / complex64 / float64 results).
 */
int ltmi_apply_masks_rows(ltmi_masks *m, const void *tile, int tile_dtype, const int32_t *rows,
                          int64_t n_rows, int64_t ld_tile, void *out, int64_t ld_out, int accumulate,
                          void *stream, int *handled);

/* Shifted masks: out[f, k] (+)= sum over the overlap of frame[f][y, x] * mask_k[y - dy_f, x - dx_f]
 * Replaces ApplyMasksEngine.process_frame_shifted (src/libertem/udf/masks.py:85-124), one call per
 * tile instead of one per frame.  Dense handles only.  `shifts` is a DEVICE array of n_frames
 * (dy, dx) int32 pairs; frames are (sig_h, sig_w) row-major, sig_h * sig_w == the handle's n_px.
 * A positive dy moves the mask down relative to the frame; no overlap contributes 0.
 */
int ltmi_apply_masks_shifted(ltmi_masks *m, const void *tile, int tile_dtype, int64_t n_frames,
                             int64_t ld_tile, int sig_h, int sig_w, const int32_t *shifts,
                             void *out, int64_t ld_out, int accumulate, void *stream);

/* Same operation with the shifts on the HOST (n_frames (dy, dx) int32 pairs): lets the library
 * group the frames by shift and run them through the MFMA kernel against shifted copies of the
 * mask image (built on the device, cached in the handle, <= 4 GiB) -- one launch per tile at
 * close to the unshifted rate (float32 / complex64 results); float64 / complex128 / exact-integer
 * results: the f64 matrix-core kernel per group of frames with the same shift (one call for a tile
 * with a constant shift; up to 256 distinct shifts per tile, up to 4096 while a shift is shared by 8 frames on
 * average and the shifted images fit 4 GiB).  Falls back to the per-frame kernel
 * above (after uploading the shifts) for handles / tiles these paths do not take. */
int ltmi_apply_masks_shifted_host(ltmi_masks *m, const void *tile, int tile_dtype, int64_t n_frames,
                                  int64_t ld_tile, int sig_h, int sig_w, const int32_t *shifts_host,
                                  void *out, int64_t ld_out, int accumulate, void *stream);

/* ---- reductions -------------------------------------------------------------------------
 * SumUDF.process_tile: out[p] (+)= sum_f tile[f, p]        (src/libertem/udf/sum.py:43-48)
 *   out: device (n_px,) of out_dtype; `workspace` device scratch of at least
 *   ltmi_sum_frames_workspace(n_frames, n_px, out_dtype) bytes (may be NULL if that is 0).
 *   out_dtype follows the reference's rule "result dtype = input dtype" (udf/sum.py:38-40):
 *   float32 / float64 for any real tile; complex64 / complex128 for complex (or real) tiles; an
 *   integer dtype for integer tiles (SumUDF(dtype=<integer>)): exact int64 accumulation, stored
 *   truncated = NumPy's wrap-around in the narrower type
 *   (reference tests/analysis/test_analysis_sum.py:157-163 `test_sum_complex`).
 */
int64_t ltmi_sum_frames_workspace(int64_t n_frames, int64_t n_px, int out_dtype);
int ltmi_sum_frames(int device, const void *tile, int tile_dtype, int64_t n_frames, int64_t n_px,
                    int64_t ld_tile, void *out, int out_dtype, int accumulate, void *workspace,
                    void *stream);
/* SumSigUDF.process_tile: out[f] (+)= sum_p tile[f, p]   (src/libertem/udf/sumsigudf.py:30-39);
 * out_dtype float32 / float64 for real tiles, complex64 / complex128 for complex tiles
 * (result_type(input, float32), udf/sumsigudf.py:23) */
int ltmi_sum_sig(int device, const void *tile, int tile_dtype, int64_t n_frames, int64_t n_px,
                 int64_t ld_tile, void *out, int out_dtype, int accumulate, void *stream);

/* merge for sig-kind buffers: dest[i] += src[i]          (src/libertem/udf/sum.py:50-52);
 * every dtype of enum ltmi_dtype, integers wrap around like NumPy's `+=` */
int ltmi_axpy(int device, void *dest, const void *src, int dtype, int64_t n, void *stream);
/* dest[r, c] (+|-)= src[r, c] for r < rows, c < cols; leading dimensions in ELEMENTS; ld_src == 0
 * broadcasts one row.  Replaces the NumPy in-place arithmetic around the kernels:
 *   `results.intensity[sig slice] += partial`  of a partial-width sig slice (src/libertem/udf/sum.py:43-48),
 *   the per-mask dark-frame constant of folded corrections (negate = 1, ld_src = 0;
 *   src/libertem/io/corrections/detector.py:315-338),
 *   `dest.intensity[:] += src.intensity`  (src/libertem/udf/sum.py:50-52). */
int ltmi_add2d(int device, void *dest, int64_t ld_dest, const void *src, int64_t ld_src, int dtype,
               int64_t rows, int64_t cols, int negate, void *stream);
/* dest[i, :] = src[idx[i], :]  -- the frames an ROI selects, gathered inside HBM (the reference
 * reads them frame by frame on the host, src/libertem/io/dataset/memory.py:107-131).
 * idx: DEVICE int64 (n_rows,); rows of row_bytes bytes, ld_src_bytes between source rows. */
int ltmi_gather_rows(int device, const void *src, int64_t ld_src_bytes, const int64_t *idx,
                     int64_t n_rows, int64_t row_bytes, void *dest, void *stream);

/* Host-to-host copy on `threads` threads (<= 0: a default from the core count, at most 16; capped at 64; at
 * least 1 MiB per thread): the staging copy of host-resident frames into page-locked bounce buffers, which has to keep
 * up with the H2D link.  Replaces, on the upload path of host datasets, the per-tile `astype` / slicing copies the
 * reference makes while reading (src/libertem/io/dataset/memory.py:102-131) -- here the bytes travel unconverted and
 * the dtype conversion happens in the kernels.  Plain memcpy semantics (no overlap); calls are serialised. */
int ltmi_host_copy(void *dst, const void *src, int64_t bytes, int threads);

/* Device address of page-locked host memory (hipHostMalloc / hipHostRegister): the place where the
 * kernels above may write small write-once result rows directly (`out` = this address), instead of
 * the reference's per-partition D2H export of device buffers (src/libertem/common/buffers.py:901-907). */
int ltmi_host_device_pointer(int device, void *host, void **dev_out);

/* ---- detector corrections -------------------------------------------------------------------
 * Replaces CorrectionSet.apply -> detector.correct on a tile (src/libertem/io/corrections/
 * corrset.py:140-166, detector.py:17-101, called from io/dataset/base/backend.py:121-124) fused
 * with the astype(read_dtype) copy of io/dataset/memory.py:102-105:
 *   out[f, p] = ((double)tile[f, p] - dark[p]) * gain[p]      (dark / gain may be NULL)
 * `dark`, `gain`: DEVICE float64 arrays of n_px.  out: device (n_frames, n_px) of F32 or F64.
 */
int ltmi_correct(int device, const void *tile, int tile_dtype, int64_t n_frames, int64_t n_px,
                 int64_t ld_tile, const double *dark, const double *gain, void *out, int out_dtype,
                 int64_t ld_out, void *stream);
/* dead-pixel repair, in place: buf[f, excl[e]] = mean(buf[f, env[e][0..cnt[e])]) for all frames.
 * Tables as built by RepairDescriptor (detector.py:278-312): DEVICE int32 arrays excl[n_excl],
 * env[n_excl][max_env] (flat pixel indices of the good neighbours), cnt[n_excl]. */
int ltmi_repair_pixels(int device, void *buf, int dtype, int64_t n_frames, int64_t ld,
                       const int32_t *excl, const int32_t *env, const int32_t *cnt, int n_excl,
                       int max_env, void *stream);

/* Byte-order decode on the device: dst[i] = src[i] with its `itemsize` (1, 2, 4, 8) bytes reversed,
 * n_items items; src == dst (in place) is allowed.  Replaces the byte-swapping decoders of
 * DtypeConversionDecoder (src/libertem/io/dataset/base/decode.py:8-66, 74-100, 123-158), which run
 * on the host for every tile of a non-native-endian dataset; the dtype conversion those decoders
 * fuse (decode_swap_N) happens in the consuming kernels, which take every native dtype. */
int ltmi_byteswap(int device, const void *src, void *dst, int itemsize, int64_t n_items,
                  void *stream);

/* Merlin / Medipix .mib frames decoded on the device.  `src`: DEVICE copy of (part of) a .mib file starting
 * at a frame header; frame f = src + f * frame_stride: `header_bytes` of ASCII header, then the payload.
 * kind 'u' (bits 8 / 16 / 32): big-endian unsigned integers -> native LTMI_U8 / U16 / U32.
 * kind 'r' (bits 1 / 6 / 12 / 24, the detector's raw "R64" words): -> LTMI_U8 / U8 / U16 / U32 (24 bit
 * also LTMI_F32, exact); `quad` != 0: raw rows of a 2x2 detector ([chip 4 | chip 3 | chip 2 | chip 1],
 * chips 3 / 4 rotated by 180 degrees) assembled into height x width frames.  dst: (n_frames, height, width)
 * contiguous, `dst_dtype` must be the dtype named above.  Replaces MIBDecoder and its numba decoders
 * decode_r{1,6,12,24}_swap / decode_r{1,6,12}_swap_2x2 plus the per-row read ranges that feed them
 * (src/libertem/io/dataset/mib.py:224-398, 401-735), which the reference runs on the host per tile. */
int ltmi_mib_decode(int device, const void *src, int64_t frame_stride, int64_t header_bytes, int kind,
                    int bits, int quad, int64_t n_frames, int height, int width, void *dst,
                    int dst_dtype, void *stream);

/* Centre-of-mass post-processing on a 2D scan of ny x nx positions: from the rows (sum, sum*y, sum*x)
 * of the 3-mask product to the shift field and its derived maps, float64.  Replaces the NumPy chain
 * center_shifts -> apply_correction -> magnitude / divergence / curl_2d of src/libertem/udf/com.py:
 * 100-142 (run by COMAnalysis.get_generic_results, src/libertem/analysis/com.py:191-284):
 *   yc = (sum != 0 ? sum*y / sum : ref_y) - ref_y   (float32, like NumPy on float32 arrays), same for x;
 *   (y, x) = transform (2x2 row-major float64: rotation / flip, HOST pointer) . (yc, xc);
 *   magnitude = sqrt(y^2 + x^2); divergence = d y/dy + d x/dx; curl = d y/dx - d x/dy with
 *   np.gradient's stencils (central inside, one-sided at the edges, unit spacing).
 * raw: device float32, row i = scan position i, ld_raw floats between rows (>= 3); outputs: device
 * float64 (ny*nx each); out_mag / out_div / out_curl may be NULL. */
int ltmi_com_fields(int device, const float *raw, int64_t ld_raw, int ny, int nx, double ref_y,
                    double ref_x, const double *transform, double *out_y, double *out_x,
                    double *out_mag, double *out_div, double *out_curl, void *stream);

/* ---- Fourier-space operators (hipFFT) ----------------------------------------------------------
 * A plan owns a batched 2D real-to-complex hipFFT (frames of sig_h x sig_w float32, `max_batch`
 * per execution) and its workspace (f32 input + complex64 half spectra).
 */
typedef struct ltmi_fft_plan ltmi_fft_plan;   /* opaque */
int ltmi_fft_plan_create(int device, int sig_h, int sig_w, int max_batch, ltmi_fft_plan **out);
int ltmi_fft_plan_destroy(ltmi_fft_plan *p);
/* CrystallinityUDF.process_frame (src/libertem/udf/crystallinity.py:73-79) for a whole tile:
 *   out[f] (+)= sum( abs(rfft2(tile[f] * real_mask)) * half_mask )
 * tile: device (n_frames, sig_h*sig_w) of tile_dtype, ld_tile elements apart (native dtype, the
 * astype is fused); real_mask: device float32 (sig_h*sig_w) or NULL; half_mask: device float32
 * (sig_h, sig_w/2+1) = fftshift(ring)[:, :w/2+1] (crystallinity.py:64-65); its non-zeros lie in the
 * rows [0, row_lo) and [row_hi, sig_h) and in the columns [0, n_cols): only that box of the
 * spectrum is read (row_lo = row_hi = sig_h, n_cols = sig_w/2+1 reads everything);
 * out: device float32 (n_frames). */
int ltmi_crystallinity(ltmi_fft_plan *p, const void *tile, int tile_dtype, int64_t n_frames,
                       int64_t ld_tile, const float *real_mask, const float *half_mask, int row_lo,
                       int row_hi, int n_cols, float *out, int accumulate, void *stream);
/* The same on RAW frames with the detector corrections fused into the conversion pass (reference:
 * CorrectionSet.apply inside the tile read, src/libertem/io/corrections/detector.py:17-101, then
 * udf/crystallinity.py:73-79): v = (float)(((double)x - dark) * gain), every excluded pixel = mean
 * of its good neighbours' corrected values; no corrected copy of the tile is written.
 * dark / gain: device float64 (sig_h*sig_w) or NULL; excl (n_excl), env (n_excl, max_env),
 * cnt (n_excl): device int32 repair tables as for ltmi_repair_pixels (n_excl = 0: none).
 * 128 x 128 / 256 x 256 / 512 x 512 / 1024 x 1024 frames: the corrected float32 frames of a batch are written to the plan's workspace and
 * transformed by the fused kernel (see ltmi_fft_plan_last_kernel) instead of hipFFT. */
int ltmi_crystallinity_corrected(ltmi_fft_plan *p, const void *tile, int tile_dtype,
                                 int64_t n_frames, int64_t ld_tile, const double *dark,
                                 const double *gain, const int32_t *excl, const int32_t *env,
                                 const int32_t *cnt, int n_excl, int max_env,
                                 const float *real_mask, const float *half_mask, int row_lo,
                                 int row_hi, int n_cols, float *out, int accumulate, void *stream);
/* Which route the last ltmi_crystallinity* call of this plan took: "k_cryst_fused<...>" (256 x 256
 * frames, rings of up to 71 columns) / "k_cryst_fused128<...>" (128 x 128 frames, any ring): rows, columns
 * and the ring sum of a frame in the LDS of one workgroup; "k_cryst_rows<w><...> + k_cryst_cols<h>" (frames whose edges
 * are 256 / 512 / 1024 pixels in any combination -- 256 x 256 only for rings of more than 71 columns --, any ring: the ring's columns of the row transforms pass through the plan's workspace)
 * (csrc/ltmi_cryst.hip); or "hipfft_r2c<...>".  LTMI_FFT_FUSED=0 at plan creation keeps every
 * frame on hipFFT.  The string lives as long as the plan. */
const char *ltmi_fft_plan_last_kernel(const ltmi_fft_plan *p);

/* ---- multi-GPU: RCCL (over xGMI) behind the C ABI ---------------------------------------------
 * One process per GPU; the path shards over scan positions, the ranks exchange per-partition nav-grid
 * results only.  Replaces, for a reference-side binding, the transport of partition results through
 * the executor (pickled over TCP, src/libertem/executor/dask.py:581-646) and the serial merge on the
 * main process (src/libertem/udf/base.py:2340-2358, udf/sum.py:50-52):
 *   'disjoint' nav buffers (ApplyMasksUDF, SumSigUDF, CoM) -> ltmi_comm_all_gather of equal row blocks,
 *   sig buffers (SumUDF)                                    -> ltmi_comm_all_reduce_sum.
 * The 128-byte id comes from ltmi_comm_unique_id on ONE rank and reaches the others through the
 * launcher's own channel (environment, file, torch.distributed store).  librccl is opened lazily. */
typedef struct ltmi_comm ltmi_comm;   /* opaque */
/* which librccl the ltmi_comm_* functions are bound to: file path (NUL-terminated, at most path_cap bytes),
 * ncclGetVersion() code, and whether that copy was already mapped by the process (a process that runs
 * torch.distributed binds to torch's bundled RCCL -- never a second one; otherwise LTMI_RCCL_LIB, the
 * loader path, /opt/rocm/lib).  Any of the output pointers may be NULL. */
int ltmi_comm_library_info(char *path_out, int64_t path_cap, int *version, int *was_loaded);
int ltmi_comm_unique_id(void *id_out /* 128 bytes */);
int ltmi_comm_create(int device, int rank, int world, const void *id /* 128 bytes */,
                     ltmi_comm **out);
int ltmi_comm_destroy(ltmi_comm *c);
/* recv[r * bytes_per_rank ...] = send of rank r, device pointers, on `stream` */
int ltmi_comm_all_gather(ltmi_comm *c, const void *send, void *recv, int64_t bytes_per_rank,
                         void *stream);
/* buf[i] = sum over ranks, in place; dtype: enum ltmi_dtype without the 16-bit integers */
int ltmi_comm_all_reduce_sum(ltmi_comm *c, void *buf, int dtype, int64_t n, void *stream);

/* ---- tuning / introspection (bench + tests) ------------------------------------------- */
/* force a kernel variant for the dense MFMA path (bench / tests only; (0,0,0) = automatic):
 *   mt in {1,2}, waves in {4,8}: the direct-load kernel k_dense_mfma with that tile shape (8 waves:
 *     uint16 tiles only);
 *   mt = 0, waves = 30: the LDS-DMA kernel k_dense_lds as dispatched; 31 / 32: its timing-only
 *     ablations without DMA / without MFMA (results are garbage; uint16 tiles against one column
 *     group only, like 34); 33: padded column groups instead of VALU columns; 34 / 35: one / two
 *     16-frame tiles per wave; 36: float32 tiles of stacks with
 *     2 - 4 column groups on the bf16 x 3 split kernel (ltmi_split.hip); 37: the float32 matrix
 *     instruction also for 1- / 2-byte integer tiles (by default those take exact float16-piece
 *     products, kernel names ending in ",f16"; LTMI_DENSE_F16=0 in the environment: never);
 *   mt = 0, waves = 40 / 41 (sparse handles): as dispatched / the gather kernel k_sell_apply even
 *     if the blocked image exists;
 *   ksplit 0 = auto.  Returns LTMI_E_INVALID for unsupported values. */
int ltmi_masks_set_tuning(ltmi_masks *m, int mt, int waves, int ksplit);
/* name of the kernel variant the last ltmi_apply_masks on this handle launched */
const char *ltmi_masks_last_kernel(const ltmi_masks *m);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* LTMI_H */
