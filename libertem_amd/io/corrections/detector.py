"""
Detector corrections on the host: dark frame, gain map, dead-pixel repair from the good pixels of
the surrounding 3^d hypercube.  Same functions and semantics as the reference's
io/corrections/detector.py (`correct` :172-275, `RepairDescriptor` :278-312, `environments`
:104-150, `flatten_filter` :153-169, `correct_dot_masks` :315-338), written with whole-array NumPy
instead of numba loops.  The device path (libltmi `ltmi_correct` / `ltmi_repair_pixels`) uses the
tables built here.
"""
import itertools

import numpy as np

from libertem_amd.common.math import prod


class CorrectError(Exception):
    pass


class RepairValueError(ValueError):
    pass


def neighbour_offsets(ndim):
    """All 3^ndim - 1 non-zero offsets in {-1, 0, 1}^ndim, lexicographic (= the order in which the
    reference enumerates a pixel's environment, detector.py:121-124)."""
    offs = [o for o in itertools.product((-1, 0, 1), repeat=ndim) if any(o)]
    return np.asarray(offs, dtype=np.intp).reshape((-1, ndim))


def environments(excluded_pixels, sigshape):
    """
    excluded_pixels: int array (ndim, k).  Returns (repairs (k, ndim, 3^ndim - 1), repair_counts (k,)):
    the in-bounds neighbours of every excluded pixel, left-packed, in lexicographic offset order.
    """
    excluded_pixels = np.asarray(excluded_pixels, dtype=np.intp)
    sigshape = np.asarray(sigshape, dtype=np.intp)
    ndim = len(sigshape)
    k = excluded_pixels.shape[1] if excluded_pixels.ndim == 2 else 0
    offs = neighbour_offsets(ndim)                                   # (n_off, ndim)
    n_off = len(offs)
    repairs = np.zeros((k, ndim, n_off), dtype=np.intp)
    counts = np.zeros(k, dtype=np.intp)
    if k == 0:
        return repairs, counts
    cand = excluded_pixels.T[:, None, :] + offs[None, :, :]          # (k, n_off, ndim)
    ok = np.all((cand >= 0) & (cand < sigshape[None, None, :]), axis=2)
    for i in range(k):
        sel = cand[i][ok[i]]                                         # (count, ndim), order kept
        counts[i] = len(sel)
        repairs[i, :, :len(sel)] = sel.T
    return repairs, counts


def flatten_filter(excluded_pixels, repairs, repair_counts, sig_shape):
    """Ravel to flat sig indices and drop damaged pixels from every environment."""
    excluded_pixels = np.asarray(excluded_pixels, dtype=np.intp)
    sig_shape = tuple(int(s) for s in sig_shape)
    k = len(repair_counts)
    n_off = repairs.shape[2] if repairs.ndim == 3 else 3 ** len(sig_shape) - 1
    repair_flat = np.zeros((k, n_off), dtype=np.intp)
    new_counts = np.zeros(k, dtype=np.intp)
    if k == 0:
        return np.zeros(0, dtype=np.intp), repair_flat, new_counts
    excluded_flat = np.ravel_multi_index(tuple(excluded_pixels), sig_shape).astype(np.intp)
    bad = set(int(x) for x in excluded_flat)
    for i in range(k):
        c = int(repair_counts[i])
        if c == 0:
            continue
        flat = np.ravel_multi_index(tuple(repairs[i, :, :c]), sig_shape)
        good = [int(x) for x in flat if int(x) not in bad]
        new_counts[i] = len(good)
        repair_flat[i, :len(good)] = good
    return excluded_flat, repair_flat, new_counts


class RepairDescriptor:
    """exclude_flat (k,), repair_flat (k, 3^d - 1), repair_counts (k,) for a signal shape."""

    def __init__(self, sig_shape, excluded_pixels=None, allow_empty=False):
        sig_shape = tuple(int(s) for s in sig_shape)
        if excluded_pixels is None:
            excluded_pixels = np.zeros((len(sig_shape), 0), dtype=np.intp)
        else:
            excluded_pixels = np.array(excluded_pixels)
            if excluded_pixels.ndim != 2:
                excluded_pixels = excluded_pixels.reshape((len(sig_shape), -1))
        repairs, counts = environments(excluded_pixels, np.array(sig_shape))
        self.exclude_flat, self.repair_flat, self.repair_counts = flatten_filter(
            excluded_pixels, repairs, counts, sig_shape)
        self.check_empty_repairs(allow_empty=allow_empty)

    def empty_repairs(self):
        return np.argwhere(self.repair_counts == 0)

    def check_empty_repairs(self, allow_empty):
        if not allow_empty:
            empty = self.empty_repairs()
            if len(empty) > 0:
                raise RepairValueError(
                    f"Empty repair environments for pixel(s) number {empty}.")


def _correct_inplace(flat, dark, gain, desc):
    """flat: (n, m) float array, modified in place.  Arithmetic follows NumPy promotion, i.e. a
    float32 buffer with float64 dark/gain is computed in float64 and rounded once on store --
    what the reference's compiled loop does (detector.py:73)."""
    if dark is not None and gain is not None:
        flat[...] = (flat - dark[None, :]) * gain[None, :]
    elif dark is not None:
        flat[...] = flat - dark[None, :]
    elif gain is not None:
        flat[...] = flat * gain[None, :]
    for p, env, c in zip(desc.exclude_flat, desc.repair_flat, desc.repair_counts):
        if c > 0:
            acc = np.zeros(flat.shape[0], dtype=np.result_type(flat.dtype, np.float64))
            for index in env[:c]:
                acc = acc + flat[:, index]
            flat[:, p] = acc / c
    return flat


def correct(buffer, dark_image=None, gain_map=None, excluded_pixels=None, repair_descriptor=None,
            inplace=False, sig_shape=None, allow_empty=False):
    """
    (buffer - dark_image) * gain_map, then every excluded pixel := mean of its good neighbours.
    buffer: (*nav, *sig).  Returns the corrected array (`buffer` itself if inplace).
    """
    s = buffer.shape
    if dark_image is not None:
        sig_shape = dark_image.shape
        dark_image = np.asarray(dark_image).reshape(-1)
    if gain_map is not None:
        sig_shape = gain_map.shape
        gain_map = np.asarray(gain_map).reshape(-1)
    if sig_shape is None:
        raise ValueError("need either `dark_image`, `gain_map`, or `sig_shape`")
    sig_shape = tuple(sig_shape)
    nav_shape = s[0:len(s) - len(sig_shape)]
    if inplace:
        if buffer.dtype.kind not in ('f', 'c'):
            raise TypeError("In-place correction only supported for floating point data.")
        out = buffer
    else:
        out = buffer.astype(np.result_type(np.float32, buffer))
    if not out.flags['C_CONTIGUOUS'] or np.isfortran(buffer):
        raise CorrectError("For in-place operation, the buffer given must be C-contiguous")
    flat = out.reshape((prod(nav_shape), prod(sig_shape)))
    if not np.shares_memory(flat, out):
        raise CorrectError("cannot view the buffer as (nav, sig) without a copy")
    if repair_descriptor is None:
        repair_descriptor = RepairDescriptor(sig_shape=sig_shape, excluded_pixels=excluded_pixels,
                                             allow_empty=allow_empty)
    else:
        repair_descriptor.check_empty_repairs(allow_empty=allow_empty)
        if excluded_pixels is not None:
            raise ValueError("Invalid arguments: both repair_descriptor and excluded_pixels set")
    _correct_inplace(flat, dark_image, gain_map, repair_descriptor)
    return out


def correct_dot_masks(masks, gain_map, excluded_pixels=None, allow_empty=False):
    """
    Fold gain and dead-pixel repair into masks of a dot product (reference :315-338):
    sum_p m'[p] x[p] == sum_p m[p] corrected(x)[p]  for dark-free data, with
    m'[r] = (m[r] + sum_{e: r in env(e)} m[e] / count_e) * gain[r],  m'[e] = 0.
    masks: (..., *sig) dense array.
    """
    masks = np.asarray(masks)
    mask_shape = masks.shape
    sig_shape = gain_map.shape
    flat = masks.reshape((-1, prod(sig_shape)))
    if excluded_pixels is not None:
        result = flat.copy()
        desc = RepairDescriptor(sig_shape, excluded_pixels=excluded_pixels, allow_empty=allow_empty)
        for e, r, c in zip(desc.exclude_flat, desc.repair_flat, desc.repair_counts):
            result[:, e] = 0
            rep = flat[:, e] / c
            for rr in r[:c]:
                result[:, rr] = result[:, rr] + rep
    else:
        result = flat
    result = result * np.asarray(gain_map).reshape(-1)
    return result.reshape(mask_shape)
