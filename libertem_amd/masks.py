"""
Mask factories: virtual-detector geometries, CoM gradients, anti-aliased radial bins.

Same function names, arguments and bit-for-bit the same values as the reference's
libertem.masks (src/libertem/masks.py) -- pinned in tests/test_masks_golden.py against vectors
produced by the reference itself.  Sparse results are `SparseStack`s (pydata `sparse` is not part
of this build).
"""
import numpy as np

from libertem_amd.common.sparse import SparseStack, is_sparse, to_dense  # noqa: F401


def _pixel_offsets(centerX, centerY, imageSizeX, imageSizeY):
    """Open grids of (row - centerY, col - centerX)."""
    rows = np.arange(imageSizeY)[:, np.newaxis] - centerY
    cols = np.arange(imageSizeX)[np.newaxis, :] - centerX
    return rows, cols


def polar_map(centerX, centerY, imageSizeX, imageSizeY, stretchY=1., angle=0.):
    """(radius, angle) of every pixel; optional elliptical stretch (masks.py:222-263).
    angle = arctan2(dy, dx) as in utils/__init__.py:41-44."""
    yy, xx = np.mgrid[0:imageSizeY, 0:imageSizeX]
    dy = yy - centerY
    dx = xx - centerX
    if stretchY != 1.0 or angle != 0.:
        dy, dx = ((dy*np.cos(angle) - dx*np.sin(angle)) / stretchY,
                  dx*np.cos(angle) + dy*np.sin(angle))
    vec = np.stack((dy.flatten(), dx.flatten())).T
    radius = np.linalg.norm(vec, axis=-1)
    phi = np.arctan2(vec[..., 0], vec[..., 1])
    shape = (imageSizeY, imageSizeX)
    return radius.reshape(shape), phi.reshape(shape)


def bounding_radius(centerX, centerY, imageSizeX, imageSizeY):
    """Radius around the centre that covers the whole frame (masks.py:281-287)."""
    dy = max(centerY, imageSizeY - centerY)
    dx = max(centerX, imageSizeX - centerX)
    return int(np.ceil(np.sqrt(dy**2 + dx**2))) + 1


def _ring_profile(r_flat, r0, width):
    # "0.5": neighbouring bins overlap by one pixel and add up to exactly 1 (masks.py:313-317)
    return np.maximum(0, np.minimum(1, width/2 + 0.5 - np.abs(r_flat - r0)))


def radial_bins(centerX, centerY, imageSizeX, imageSizeY, radius=None, radius_inner=0,
                n_bins=None, normalize=False, use_sparse=None, dtype=None):
    """Anti-aliased concentric rings (masks.py:290-353)."""
    if radius is None:
        radius = bounding_radius(centerX, centerY, imageSizeX, imageSizeY)
    if n_bins is None:
        n_bins = int(np.round(radius - radius_inner))
    r_flat = polar_map(centerX, centerY, imageSizeX, imageSizeY)[0].flatten()
    width = (radius - radius_inner) / n_bins
    bin_area = np.pi * (radius**2 - (radius - width)**2)
    if use_sparse is None:
        use_sparse = bin_area / (imageSizeX * imageSizeY) < 0.1
    centres = np.linspace(radius_inner, radius - width, n_bins) + width/2
    cy_i, cx_i = int(np.round(centerY)), int(np.round(centerX))
    patch_centre = radius_inner < 0.5 and 0 <= cy_i < imageSizeY and 0 <= cx_i < imageSizeX

    def bin_values(r0):
        vals = _ring_profile(r_flat, r0, width)
        return vals

    if not use_sparse:
        layers = []
        for r0 in centres:
            vals = bin_values(r0)
            if normalize:
                s = vals.sum()
                if not np.isclose(s, 0):
                    vals /= s
            layers.append(vals.reshape((imageSizeY, imageSizeX)).astype(dtype))
        if patch_centre:
            layers[0][cy_i, cx_i] = 1 - radius_inner
        return np.stack(layers)

    datas, mask_idx, px_idx = [], [], []
    all_px = np.arange(len(r_flat), dtype=np.int64)
    for b, r0 in enumerate(centres):
        vals = bin_values(r0)
        sel = vals != 0
        vals = vals[sel]
        if normalize:
            s = vals.sum()
            if not np.isclose(s, 0):
                vals /= s
        vals = vals.astype(dtype)
        px = all_px[sel]
        if b == 0 and patch_centre:
            # the reference adds a one-entry COO `np.array([1 - slices[0][index] - radius_inner])`
            # (masks.py:338-349): the scalar read from the slice has the slice's dtype, the patch the
            # dtype NumPy's scalar promotion gives the expression (float32 slices stay float32 for
            # Python numbers under NumPy >= 2), the sum promotes like arrays of the two dtypes
            target = cy_i * imageSizeX + cx_i
            hit = np.flatnonzero(px == target)
            cur = vals[hit[0]] if len(hit) else vals.dtype.type(0)
            patch = np.array([1 - cur - radius_inner])
            vals = vals.astype(np.result_type(vals.dtype, patch.dtype))
            if len(hit):
                vals[hit[0]] = vals[hit[0]] + patch[0]
            else:
                vals = np.concatenate([vals, patch.astype(vals.dtype)])
                px = np.concatenate([px, [target]])
        datas.append(vals)
        mask_idx.append(np.full(len(vals), b, dtype=np.int64))
        px_idx.append(px)
    out_dtype = np.result_type(*[d.dtype for d in datas])
    return SparseStack(np.concatenate(datas).astype(out_dtype), np.concatenate(mask_idx),
                       np.concatenate(px_idx), n_bins, (imageSizeY, imageSizeX))


def _make_circular_mask(centerX, centerY, imageSizeX, imageSizeY, radius, antialiased=False):
    if antialiased:
        return radial_bins(centerX, centerY, imageSizeX, imageSizeY, radius, n_bins=1,
                           use_sparse=False)[0]
    rows, cols = _pixel_offsets(centerX, centerY, imageSizeX, imageSizeY)
    return rows*rows + cols*cols <= radius*radius


def circular(centerX, centerY, imageSizeX, imageSizeY, radius, antialiased=False):
    """Filled disk as bool array, or anti-aliased float disk (masks.py:108-127)."""
    return _make_circular_mask(centerX, centerY, imageSizeX, imageSizeY, radius, antialiased)


def ring(centerX, centerY, imageSizeX, imageSizeY, radius, radius_inner, antialiased=False):
    """Annulus radius_inner < r <= radius (masks.py:130-159)."""
    if antialiased:
        return radial_bins(centerX, centerY, imageSizeX, imageSizeY, radius=radius,
                           radius_inner=radius_inner, n_bins=1, use_sparse=False)[0]
    outer = _make_circular_mask(centerX, centerY, imageSizeX, imageSizeY, radius)
    inner = _make_circular_mask(centerX, centerY, imageSizeX, imageSizeY, radius_inner)
    return outer & ~inner


def radial_gradient_background_subtraction(r, r0, r_outer, delta=1):
    """Linear ramp inside r0, anti-aliased edge, -1 ring out to r_outer (masks.py:176-219)."""
    result = np.zeros_like(r)
    inside = r < r0 - delta/2
    result[inside] = r[inside] / r0
    edge = (r >= r0 - delta/2) * (r < r0 + delta/2)
    result[edge] = (r0 - r[edge]) / (delta/2)
    outside = (r >= r0 + delta/2) * (r <= r_outer)
    result[outside] = -1
    return result


def radial_gradient(centerX, centerY, imageSizeX, imageSizeY, radius, antialiased=False):
    """Linear radial gradient 0..1 within radius (masks.py:162-173)."""
    rows, cols = _pixel_offsets(centerX, centerY, imageSizeX, imageSizeY)
    if antialiased:
        r = np.sqrt(rows**2 + cols**2)
        return radial_gradient_background_subtraction(r=r, r0=radius, r_outer=0)
    return (rows*rows + cols*cols <= radius*radius) * (np.sqrt(rows*rows + cols*cols) / radius)


def balance(template):
    """Scale the negative part so the template sums to zero (masks.py:266-278)."""
    result = template.copy()
    pos = template > 0
    neg = template < 0
    result[neg] *= template[pos].sum() / template[neg].sum() * -1
    return result


def background_subtraction(centerX, centerY, imageSizeX, imageSizeY, radius, radius_inner,
                           antialiased=False):
    """Disk minus area-normalised surrounding ring (masks.py:356-367)."""
    disk = circular(centerX, centerY, imageSizeX, imageSizeY, radius_inner,
                    antialiased=antialiased)
    annulus = ring(centerX, centerY, imageSizeX, imageSizeY, radius, radius_inner,
                   antialiased=antialiased)
    return disk - annulus*np.sum(disk)/np.sum(annulus)


def rectangular(X, Y, Width, Height, imageSizeX, imageSizeY):
    """Bool rectangle from a corner + signed width/height (masks.py:370-411)."""
    mask = np.zeros([imageSizeY, imageSizeX], dtype="bool")
    if Height*Width > 0:
        y0, x0, y1, x1 = min(Y, Y+Height), min(X, X+Width), max(Y, Y+Height), max(X, X+Width)
    elif Height > 0 and Width < 0:
        y0, x0, y1, x1 = Y, X+Width, Y+Height, X
    elif Height < 0 and Width > 0:
        y0, x0, y1, x1 = Y+Height, X, Y, X+Width
    else:
        y0, x0, y1, x1 = 0, 0, -1, -1
    y0, x0, y1, x1 = int(y0), int(x0), int(y1), int(x1)
    mask[max(0, y0):min(y1+1, imageSizeY), max(0, x0):min(x1+1, imageSizeX)] = 1
    return mask


def gradient_x(imageSizeX, imageSizeY, dtype=np.float32):
    """mask[y, x] = x (masks.py:415-418)."""
    return np.tile(np.arange(imageSizeX).astype(dtype), imageSizeY).reshape(
        imageSizeY, imageSizeX)


def gradient_y(imageSizeX, imageSizeY, dtype=np.float32):
    """mask[y, x] = y (masks.py:421-422)."""
    return gradient_x(imageSizeY, imageSizeX, dtype).transpose()


def sparse_template_multi_stack(mask_index, offsetX, offsetY, template, imageSizeX, imageSizeY):
    """Stamp `template` into masks `mask_index` at the given offsets, clipped (masks.py:55-87)."""
    fy, fx = template.shape
    ty, tx = np.mgrid[0:fy, 0:fx]
    datas, mis, pxs = [], [], []
    for m, ox, oy in zip(mask_index, offsetX, offsetY):
        ys = ty.flatten() + oy
        xs = tx.flatten() + ox
        ok = (ys >= 0) * (ys < imageSizeY) * (xs >= 0) * (xs < imageSizeX)
        datas.append(template.flatten()[ok])
        mis.append(np.full(int(np.count_nonzero(ok)), m, dtype=np.int64))
        pxs.append(ys[ok] * imageSizeX + xs[ok])
    return SparseStack(np.concatenate(datas), np.concatenate(mis), np.concatenate(pxs),
                       int(max(mask_index) + 1), (imageSizeY, imageSizeX))


def sparse_circular_multi_stack(mask_index, centerX, centerY, imageSizeX, imageSizeY, radius):
    """Many small disks as one sparse stack (masks.py:90-105)."""
    bbox = int(2*np.ceil(radius) + 1)
    c = int((bbox - 1) // 2)
    template = circular(centerX=c, centerY=c, imageSizeX=bbox, imageSizeY=bbox, radius=radius)
    return sparse_template_multi_stack(
        mask_index=mask_index, offsetX=np.array(centerX, dtype=int) - c,
        offsetY=np.array(centerY, dtype=int) - c, template=template,
        imageSizeX=imageSizeX, imageSizeY=imageSizeY)
