"""
The hot path restated: what `Context.run_udf(dataset, udf)` computes with the reference's
InlineJobExecutor on a NumPy MemoryDataSet.  (test infrastructure -- see oracle/__init__.py)

Follows udf/base.py:106-123 (_get_dtype), :2100-2335 (UDFPartRunner), :2340-2386 (merge),
udf/masks.py:12-124,353-392, common/container.py:74-94,260-314, common/numba/__init__.py:
90-184, udf/sum.py, udf/sumsigudf.py, udf/com.py:100-142,534-717, analysis/com.py:191-334,
analysis/radialfourier.py:184-195,316-354, corrections/coordinates.py:11-54.
"""
import numpy as np
import scipy.sparse as sp

from . import masks as omasks
from .tiling import partition_boundaries, negotiate_tileshape, iter_tiles, prod


def input_dtype(ds_dtype, preferred=np.float32):
    # udf/base.py:106-123 (no corrections)
    return np.result_type(preferred, ds_dtype)


def _flat_tiles(data, sig_dims, num_partitions, in_dtype, tileshape=None, method='tile'):
    """
    Generator over (row0, row1, sig_origin, sig_shape, tile_array) in reference order.
    tile_array has shape (rows, *sig_shape), dtype in_dtype, C-contiguous
    (io/dataset/memory.py:102-105).
    """
    sig = tuple(data.shape[-sig_dims:])
    n_frames = prod(data.shape[:-sig_dims])
    flat = data.reshape((n_frames,) + sig)
    parts = partition_boundaries(n_frames, num_partitions)
    ts = negotiate_tileshape(data.shape, sig_dims, data.dtype, in_dtype,
                             parts[0][1] - parts[0][0], forced_tileshape=tileshape,
                             method=method)
    for (p0, p1) in parts:
        for (f0, f1, o, s, idx) in iter_tiles(p0, p1, ts, sig):
            sl = (slice(f0, f1),) + tuple(slice(oo, oo + ss) for oo, ss in zip(o, s))
            tile = flat[sl]
            if tile.dtype != in_dtype or not tile.flags.c_contiguous:
                tile = tile.astype(in_dtype)
            yield p0, p1, f0, f1, o, s, tile


def _mask_slice_T(masks_stack, o, s, mask_dtype):
    # common/container.py:81-91: slice sig, flatten, transpose, cast
    sl = (slice(None),) + tuple(slice(oo, oo + ss) for oo, ss in zip(o, s))
    m = masks_stack[sl].reshape((masks_stack.shape[0], -1)).T
    return m.astype(mask_dtype)


def apply_masks(data, masks_stack, sig_dims=2, num_partitions=1, tileshape=None,
                mask_dtype=None, preferred_dtype=None, use_torch=True):
    """
    ApplyMasksUDF with dense masks on the reference's CPU path.
    Returns `intensity.data`: nav + (n_masks,).
    """
    masks_stack = np.asarray(masks_stack)
    if mask_dtype is None:
        mask_dtype = masks_stack.dtype          # udf/masks.py:318-322
    mask_dtype = np.dtype(mask_dtype)
    pref = np.float32 if preferred_dtype is None else preferred_dtype   # udf/masks.py:311-316
    in_dtype = input_dtype(data.dtype, pref)
    res_dtype = np.result_type(in_dtype, mask_dtype)                     # udf/masks.py:362
    nav = data.shape[:-sig_dims]
    n_frames = prod(nav)
    n_masks = masks_stack.shape[0]
    out = np.zeros((n_frames, n_masks), dtype=res_dtype)
    # udf/masks.py:22-33: torch.mm iff float input of the same dtype as the masks
    torch_ok = use_torch and in_dtype.kind == 'f' and in_dtype == mask_dtype
    if torch_ok:
        try:
            import torch
        except ImportError:
            torch_ok = False
    cache = {}
    for (p0, p1, f0, f1, o, s, tile) in _flat_tiles(data, sig_dims, num_partitions, in_dtype,
                                                     tileshape):
        key = (o, s)
        if key not in cache:
            cache[key] = _mask_slice_T(masks_stack, o, s, mask_dtype)
        m = cache[key]
        flat_tile = tile.reshape((tile.shape[0], -1))
        if torch_ok:
            r = torch.mm(torch.from_numpy(flat_tile), torch.from_numpy(m)).numpy()
        else:
            r = flat_tile @ m                                            # udf/masks.py:76-77
        out[f0:f1] += r                                                  # udf/masks.py:389-392
    return out.reshape(nav + (n_masks,))


def apply_masks_shifted(data, masks_stack, shifts, sig_dims=2):
    """
    ApplyMasksUDF(shifts=...) on the CPU path: frame by frame (udf/masks.py:85-124, :394-404).
    `shifts`: (2,) constant (y, x) or nav + (2,) per frame.  A positive y shift moves the mask
    down relative to the frame; non-overlapping parts are discarded.
    """
    masks_stack = np.asarray(masks_stack)
    in_dtype = input_dtype(data.dtype)
    res_dtype = np.result_type(in_dtype, masks_stack.dtype)
    nav = data.shape[:-sig_dims]
    sig = tuple(data.shape[-sig_dims:])
    n_frames = prod(nav)
    n_masks = masks_stack.shape[0]
    flat = data.reshape((n_frames,) + sig)
    shifts = np.asarray(shifts)
    if shifts.ndim == 1:
        shifts = np.broadcast_to(shifts, (n_frames, 2))
    else:
        shifts = shifts.reshape((n_frames, 2))
    out = np.zeros((n_frames, n_masks), dtype=res_dtype)
    H, W = sig
    for f in range(n_frames):
        dy, dx = (int(v) for v in shifts[f].astype(int))
        # left = sig_slice & sig_slice.shift_by(shifts): frame region; right: mask region
        y0, y1 = max(0, dy), min(H, H + dy)
        x0, x1 = max(0, dx), min(W, W + dx)
        if y1 <= y0 or x1 <= x0:
            continue                                  # zero overlap -> contributes 0
        frame = flat[f].astype(in_dtype)[y0:y1, x0:x1].reshape((1, -1))
        m = masks_stack[:, y0 - dy:y1 - dy, x0 - dx:x1 - dx].reshape((n_masks, -1)).T
        out[f] += (frame @ m.astype(masks_stack.dtype)).reshape((n_masks,))
    return out.reshape(nav + (n_masks,))


def rmatmul(left_dense, right_sparse):
    """
    common/numba/__init__.py:90-184, restated with identical loop order
    (CSR: per pixel-row, per nnz, all frames; CSC: per mask column, per nnz, all frames).
    Vectorised over the innermost `left_row` loop, which is elementwise-independent.
    """
    if len(left_dense.shape) != 2:
        raise ValueError(f"Shape of left_dense is not 2D, but {left_dense.shape}.")
    if len(right_sparse.shape) != 2:
        raise ValueError(f"Shape of right_sparse is not 2D, but {right_sparse.shape}.")
    if left_dense.shape[1] != right_sparse.shape[0]:
        raise ValueError("Shape mismatch: left_dense.shape[1] != right_sparse.shape[0]")
    res_t = np.zeros((right_sparse.shape[1], left_dense.shape[0]),
                     dtype=np.result_type(right_sparse, left_dense))
    if isinstance(right_sparse, (sp.csc_matrix, sp.csr_matrix)):
        data, indices, indptr = right_sparse.data, right_sparse.indices, right_sparse.indptr
    if isinstance(right_sparse, sp.csc_matrix):
        for col in range(len(indptr) - 1):
            for index in range(indptr[col], indptr[col + 1]):
                res_t[col, :] += left_dense[:, indices[index]] * data[index]
    elif isinstance(right_sparse, sp.csr_matrix):
        for row in range(len(indptr) - 1):
            if indptr[row + 1] > indptr[row]:
                rowbuf = left_dense[:, row].copy()
                for index in range(indptr[row], indptr[row + 1]):
                    res_t[indices[index], :] += rowbuf * data[index]
    else:
        raise ValueError("Right hand matrix mus be of type scipy.sparse.csc_matrix or "
                         f"scipy.sparse.csr_matrix, got {type(right_sparse)}.")
    return res_t.T.copy()


def apply_masks_sparse(data, masks_csr, sig_dims=2, num_partitions=1, tileshape=None,
                       mask_dtype=None, fmt='csr', product=None):
    """
    ApplyMasksUDF with use_sparse='scipy.sparse[.csr|.csc]' on the CPU path:
    `masks_csr` is the stack as scipy sparse (n_masks, px) (sig flattened C-order).
    Per tile the (px_in_slice, n_masks) CSR/CSC matrix is built as common/container.py:53-64
    does and multiplied with rmatmul (udf/masks.py:34-40, :68-69).
    `product(flat_tile, matrix)`: another implementation of that product for TIMING (bench.py's cpu_baseline: the
    reference's loop is numba-compiled, the restatement here interpreted); parity tests use rmatmul.
    """
    masks_csr = sp.csr_matrix(masks_csr)
    if mask_dtype is None:
        mask_dtype = masks_csr.dtype
    in_dtype = input_dtype(data.dtype)
    res_dtype = np.result_type(in_dtype, mask_dtype)
    nav = data.shape[:-sig_dims]
    sig = tuple(data.shape[-sig_dims:])
    n_frames = prod(nav)
    n_masks = masks_csr.shape[0]
    out = np.zeros((n_frames, n_masks), dtype=res_dtype)
    px_index = np.arange(prod(sig)).reshape(sig)
    cache = {}
    for (p0, p1, f0, f1, o, s, tile) in _flat_tiles(data, sig_dims, num_partitions, in_dtype,
                                                     tileshape):
        key = (o, s)
        if key not in cache:
            sl = tuple(slice(oo, oo + ss) for oo, ss in zip(o, s))
            cols = px_index[sl].reshape(-1)
            sub = masks_csr[:, cols].T.astype(mask_dtype)     # (px_in_slice, n_masks)
            sub = sp.csc_matrix(sub) if fmt == 'csc' else sp.csr_matrix(sub)
            sub.sum_duplicates()
            sub.sort_indices()
            cache[key] = sub
        flat_tile = tile.reshape((tile.shape[0], -1))
        out[f0:f1] += (product or rmatmul)(flat_tile, cache[key])
    return out.reshape(nav + (n_masks,))


def sum_udf(data, sig_dims=2, num_partitions=1, tileshape=None, dtype='float32'):
    """udf/sum.py:6-58 incl. merge order (partition order with the Inline executor)."""
    in_dtype = input_dtype(data.dtype, dtype)
    sig = tuple(data.shape[-sig_dims:])
    total = np.zeros(sig, dtype=in_dtype)
    part_buf = None
    cur = None
    for (p0, p1, f0, f1, o, s, tile) in _flat_tiles(data, sig_dims, num_partitions, in_dtype,
                                                     tileshape):
        if cur != (p0, p1):
            if part_buf is not None:
                total[:] += part_buf                     # udf/sum.py:50-52
            part_buf = np.zeros(sig, dtype=in_dtype)
            cur = (p0, p1)
        sl = tuple(slice(oo, oo + ss) for oo, ss in zip(o, s))
        part_buf[sl] += np.sum(tile, axis=0)             # udf/sum.py:43-48
    if part_buf is not None:
        total[:] += part_buf
    return total


def sumsig_udf(data, sig_dims=2, num_partitions=1, tileshape=None):
    """udf/sumsigudf.py:6-39"""
    in_dtype = input_dtype(data.dtype)
    res_dtype = np.result_type(in_dtype, np.float32)
    nav = data.shape[:-sig_dims]
    out = np.zeros((prod(nav),), dtype=res_dtype)
    for (p0, p1, f0, f1, o, s, tile) in _flat_tiles(data, sig_dims, num_partitions, in_dtype,
                                                     tileshape):
        out[f0:f1] += np.sum(tile.reshape((tile.shape[0], -1)), axis=1)
    return out.reshape(nav)


# --- CoM ---------------------------------------------------------------------------------

def rotate_deg(degrees):
    # corrections/coordinates.py:11-29
    radians = np.pi/180*degrees
    return np.array([
        (np.cos(radians), np.sin(radians)),
        (-np.sin(radians), np.cos(radians))
    ])


def flip_y():
    # corrections/coordinates.py:32-40
    return np.array([(-1, 0), (0, 1)])


def identity():
    return np.eye(2)


def center_shifts(img_sum, img_y, img_x, ref_y, ref_x):
    # udf/com.py:100-107
    x_centers = np.divide(img_x, img_sum, where=img_sum != 0)
    y_centers = np.divide(img_y, img_sum, where=img_sum != 0)
    x_centers[img_sum == 0] = ref_x
    y_centers[img_sum == 0] = ref_y
    x_centers -= ref_x
    y_centers -= ref_y
    return (y_centers, x_centers)


def apply_correction(y_centers, x_centers, scan_rotation, flip_y_, forward=True):
    # udf/com.py:110-127
    shape = y_centers.shape
    transform = flip_y() if flip_y_ else identity()
    transform = rotate_deg(scan_rotation) @ transform
    y_centers = y_centers.reshape(-1)
    x_centers = x_centers.reshape(-1)
    if not forward:
        transform = np.linalg.inv(transform)
    y_t, x_t = transform @ (y_centers, x_centers)
    return (y_t.reshape(shape), x_t.reshape(shape))


def divergence(y_centers, x_centers):
    # udf/com.py:130-131
    return np.gradient(y_centers, axis=0) + np.gradient(x_centers, axis=1)


def curl_2d(y_centers, x_centers):
    # udf/com.py:134-138
    return np.gradient(y_centers, axis=1) - np.gradient(x_centers, axis=0)


def magnitude(y_centers, x_centers):
    # udf/com.py:141-142
    return np.sqrt(y_centers**2 + x_centers**2)


def com_udf(data, num_partitions=1, cy=None, cx=None, r=float('inf'), ri=0.,
            scan_rotation=0., flip_y=False, regression=-1):
    """CoMUDF (udf/com.py:298-717) -> dict of result arrays shaped nav + extra."""
    sig = tuple(data.shape[-2:])
    nav = tuple(data.shape[:-2])
    if cy is None:
        cy = sig[0] // 2                                   # udf/com.py:514-520
    if cx is None:
        cx = sig[1] // 2
    stack = np.stack([np.asarray(m, dtype=np.float32)
                      for m in omasks.com_masks(sig[0], sig[1], cy, cx, r, ri)])
    raw = apply_masks(data, stack, num_partitions=num_partitions, mask_dtype=np.float32)
    res_dtype = np.result_type(input_dtype(data.dtype), np.float32)
    raw = raw.astype(res_dtype)
    raw_shifts = center_shifts(raw[..., 0], raw[..., 1], raw[..., 2], cy, cx)
    raw_com = (raw_shifts[0].copy() + cy, raw_shifts[1].copy() + cx)
    field = apply_correction(raw_shifts[0], raw_shifts[1], scan_rotation, flip_y)
    raw_shifts = np.moveaxis(np.array(raw_shifts), 0, -1)
    raw_com = np.moveaxis(np.array(raw_com), 0, -1)
    field = np.moveaxis(np.array(field), 0, -1)
    valid_mask = np.ones(nav, dtype=bool)
    # udf/com.py:600-648
    reg = np.zeros((3, 2))
    inp = None

    def get_inp():
        inp = np.ones(field.shape[:-1] + (3,))
        y, x = np.ogrid[:field.shape[0], :field.shape[1]]
        inp[..., 1] = y
        inp[..., 2] = x
        return inp

    if isinstance(regression, (int, np.integer)):
        if regression == -1:
            pass
        elif regression == 0:
            reg[0] = np.mean(field[valid_mask], axis=0)
        elif regression == 1:
            inp = get_inp()
            reg[:] = np.linalg.lstsq(inp[valid_mask], field[valid_mask], rcond=None)[0]
        else:
            raise ValueError(f'Unrecognized regression option {regression}')
    else:
        regression = np.array(regression)
        if regression.shape != (3, 2):
            raise ValueError("Regression parameter doesn't have required shape (3, 2).")
        reg[:] = regression
    has_lin = not np.allclose(reg[1:], 0)
    if has_lin and inp is None:
        inp = get_inp()
    if not has_lin:
        inp = None
    if inp is not None:
        field[valid_mask] -= inp[valid_mask] @ reg
    elif not np.allclose(reg[0], 0):
        field[valid_mask] -= reg[0]
    fy, fx = field[..., 0], field[..., 1]
    return {
        'raw_shifts': raw_shifts.astype(res_dtype),
        'raw_com': raw_com.astype(res_dtype),
        'field': field,
        'field_y': fy,
        'field_x': fx,
        'regression': reg.astype(np.float64),
        'magnitude': magnitude(fy, fx),
        'divergence': divergence(fy, fx),
        'curl': curl_2d(fy, fx),
    }


def com_analysis(data, num_partitions=1, cx=None, cy=None, r=float('inf'), ri=0.0,
                 scan_rotation=0., flip_y=False):
    """COMAnalysis: parameters analysis/com.py:313-334, masks :286-311, results :191-284."""
    sig = tuple(data.shape[-2:])
    if cx is None:
        cx = sig[1] / 2
    if cy is None:
        cy = sig[0] / 2
    stack = np.stack([np.asarray(m) for m in omasks.com_masks(sig[0], sig[1], cy, cx, r,
                                                              ri if ri else None)])
    inten = apply_masks(data, stack, num_partitions=num_partitions, mask_dtype=np.float32)
    yc_raw, xc_raw = center_shifts(inten[..., 0], inten[..., 1], inten[..., 2], cy, cx)
    yc, xc = apply_correction(yc_raw, xc_raw, scan_rotation, flip_y)
    out = {'intensity': inten, 'x': xc, 'y': yc, 'magnitude': magnitude(yc, xc)}
    if all(s > 1 for s in yc.shape):
        out['divergence'] = divergence(yc, xc)
        out['curl'] = curl_2d(yc, xc)
    return out


# --- radial Fourier ----------------------------------------------------------------------

def radial_fourier_parameters(sig, cx=None, cy=None, ri=0, ro=None, n_bins=1, max_order=24,
                              use_sparse=None):
    """analysis/radialfourier.py:316-354"""
    detector_y, detector_x = sig
    if cx is None:
        cx = detector_x / 2
    if cy is None:
        cy = detector_y / 2
    if ro is None:
        ro = omasks.bounding_radius(cx, cy, detector_x, detector_y)
    mask_count = n_bins * (max_order + 1)
    bin_width = (ro - ri) / n_bins
    bin_area = np.pi * ro**2 - np.pi * (ro - bin_width)**2
    stack_size = mask_count * detector_y * detector_x * 8
    default = 'scipy.sparse'
    if stack_size < 2**18:
        default = False
    elif bin_area / (detector_x * detector_y) > 0.05 and n_bins < 10:
        default = False
    if use_sparse is None:
        use_sparse = default
    return dict(cx=cx, cy=cy, ri=ri, ro=ro, n_bins=n_bins, max_order=max_order,
                use_sparse=use_sparse, mask_count=mask_count, mask_dtype=np.complex64)


def radial_fourier_analysis(data, num_partitions=1, **params):
    """RadialFourierAnalysis: get_udf + get_udf_results (analysis/radialfourier.py:184-195)."""
    sig = tuple(data.shape[-2:])
    nav = tuple(data.shape[:-2])
    p = radial_fourier_parameters(sig, **params)
    if p['use_sparse'] is False:
        stack = omasks.radial_mask_stack(sig[0], sig[1], p['cx'], p['cy'], p['ri'], p['ro'],
                                         p['n_bins'], p['max_order'])
        inten = apply_masks(data, stack, num_partitions=num_partitions,
                            mask_dtype=np.complex64)
    else:
        csr = omasks.radial_mask_stack_csr(sig[0], sig[1], p['cx'], p['cy'], p['ri'], p['ro'],
                                           p['n_bins'], p['max_order'])
        inten = apply_masks_sparse(data, csr, num_partitions=num_partitions,
                                   mask_dtype=np.complex64)
    raw = inten.reshape((prod(nav), -1)).T
    raw = raw.reshape((p['n_bins'], p['max_order'] + 1) + nav)
    return {'intensity': inten, 'raw_results': raw, 'parameters': p}


# ---------------------------------------------------------------------------------------------------
# CrystallinityUDF  (udf/crystallinity.py:47-79)
# ---------------------------------------------------------------------------------------------------
def crystallinity_udf(data, rad_in, rad_out, real_center=None, real_rad=None):
    """data (*nav, sy, sx).  Per frame, in the reference's arithmetic: the frame arrives as
    input dtype = result_type(float32, ds dtype) (udf/base.py:106-123); with a real-space mask
    (an int array) the product is float64 and so is the transform, without it the float32 frame
    is transformed in single precision; the sum is stored into a float32 nav buffer."""
    sig = data.shape[-2:]
    sy, sx = sig

    def disk(cx, cy, r):                                # masks.py:50-52 (_make_circular_mask)
        x, y = np.ogrid[-cy:sy - cy, -cx:sx - cx]
        return x * x + y * y <= r * r

    real_mask = None
    if not (real_center is None or real_rad is None):
        real_mask = 1 - 1 * disk(real_center[1], real_center[0], real_rad)
    ring = np.fft.fftshift(1 * disk(sx * 0.5, sy * 0.5, rad_out) - 1 * disk(sx * 0.5, sy * 0.5, rad_in))
    half = ring[:, :int(ring.shape[1] * 0.5) + 1]
    frames = data.reshape((-1,) + sig).astype(input_dtype(data.dtype))
    out = np.zeros(len(frames), dtype=np.float32)
    for i, frame in enumerate(frames):
        masked = frame * real_mask if real_mask is not None else frame
        out[i] = np.sum(abs(np.fft.rfft2(masked)) * half)
    return out.reshape(data.shape[:-2])
