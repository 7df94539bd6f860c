"""
Mask factories restated from src/libertem/masks.py, src/libertem/utils/__init__.py:27-44,
src/libertem/udf/com.py:47-97 and src/libertem/analysis/radialfourier.py:106-161.
(test infrastructure -- see oracle/__init__.py)
"""
import numpy as np
import scipy.sparse as sp


def make_polar(cartesian):
    # utils/__init__.py:27-44
    ds = np.linalg.norm(cartesian, axis=-1)
    alphas = np.arctan2(cartesian[..., 0], cartesian[..., 1])
    return np.array((ds.T, alphas.T)).T


def polar_map(centerX, centerY, imageSizeX, imageSizeY, stretchY=1., angle=0.):
    # masks.py:222-263
    y, x = np.mgrid[0:imageSizeY, 0:imageSizeX]
    dy = y - centerY
    dx = x - centerX
    if stretchY != 1.0 or angle != 0.:
        (dy, dx) = (
            (dy*np.cos(angle) - dx*np.sin(angle)) / stretchY,
            dx*np.cos(angle) + dy*np.sin(angle),
        )
    dy = dy.flatten()
    dx = dx.flatten()
    cartesians = np.stack((dy, dx)).T
    polars = make_polar(cartesians)
    return (
        polars[:, 0].reshape((imageSizeY, imageSizeX)),
        polars[:, 1].reshape((imageSizeY, imageSizeX)),
    )


def bounding_radius(centerX, centerY, imageSizeX, imageSizeY):
    # masks.py:281-287
    dy = max(centerY, imageSizeY - centerY)
    dx = max(centerX, imageSizeX - centerX)
    return int(np.ceil(np.sqrt(dy**2 + dx**2))) + 1


def radial_bins_coo(centerX, centerY, imageSizeX, imageSizeY, radius=None, radius_inner=0,
                    n_bins=None, normalize=False, dtype=None):
    """
    Sparse flavour of radial_bins (masks.py:290-353, `use_sparse=True` branch) returned as
    (data, bin_index, flat_pixel_index) triplets in the reference's per-bin order.
    """
    if radius is None:
        radius = bounding_radius(centerX, centerY, imageSizeX, imageSizeY)
    if n_bins is None:
        n_bins = int(np.round(radius - radius_inner))
    r, phi = polar_map(centerX, centerY, imageSizeX, imageSizeY)
    r = r.flatten()
    width = (radius - radius_inner) / n_bins
    jjs = np.arange(len(r), dtype=np.int64)
    datas, bins, pix = [], [], []
    for b, r0 in enumerate(np.linspace(radius_inner, radius - width, n_bins) + width/2):
        diff = np.abs(r - r0)
        vals = np.maximum(0, np.minimum(1, width/2 + 0.5 - diff))
        select = vals != 0
        vals = vals[select]
        if normalize:
            s = vals.sum()
            if not np.isclose(s, 0):
                vals /= s
        vals = vals.astype(dtype)
        idx = jjs[select]
        if b == 0 and radius_inner < 0.5:
            # centre patch, masks.py:338-349: slices[0] += COO([diff] at index)
            yy = int(np.round(centerY))
            xx = int(np.round(centerX))
            if 0 <= yy < imageSizeY and 0 <= xx < imageSizeX:
                index = yy * imageSizeX + xx
                pos = np.nonzero(idx == index)[0]
                # `slices[0][index]` is a scalar of the slice's dtype (the stored value or the fill value
                # 0); `np.array([diff])` then has whatever dtype NumPy's scalar promotion gives
                # `1 - scalar - radius_inner` (float32 slices stay float32 for Python numbers under
                # NumPy >= 2), and COO + COO promotes like arrays of those dtypes
                cur = vals[pos[0]] if len(pos) else vals.dtype.type(0)
                patch = np.array([1 - cur - radius_inner])
                vals = vals.astype(np.result_type(vals.dtype, patch.dtype))
                if len(pos):
                    vals[pos[0]] = vals[pos[0]] + patch[0]
                else:
                    vals = np.concatenate([vals, patch.astype(vals.dtype)])
                    idx = np.concatenate([idx, [index]])
        datas.append(vals)
        bins.append(np.full(len(vals), b, dtype=np.int64))
        pix.append(idx)
    return datas, bins, pix, n_bins


def radial_bins(centerX, centerY, imageSizeX, imageSizeY, radius=None, radius_inner=0,
                n_bins=None, normalize=False, use_sparse=None, dtype=None):
    # masks.py:290-353
    if radius is None:
        radius = bounding_radius(centerX, centerY, imageSizeX, imageSizeY)
    if n_bins is None:
        n_bins = int(np.round(radius - radius_inner))
    width = (radius - radius_inner) / n_bins
    bin_area = np.pi * (radius**2 - (radius - width)**2)
    if use_sparse is None:
        use_sparse = bin_area / (imageSizeX * imageSizeY) < 0.1
    if use_sparse:
        datas, bins, pix, n_bins = radial_bins_coo(
            centerX, centerY, imageSizeX, imageSizeY, radius, radius_inner, n_bins,
            normalize, dtype)
        out_dtype = np.result_type(*[d.dtype for d in datas])
        m = sp.coo_matrix(
            (np.concatenate(datas).astype(out_dtype),
             (np.concatenate(bins), np.concatenate(pix))),
            shape=(n_bins, imageSizeX * imageSizeY))
        return m.tocsr()  # (n_bins, px) sparse; caller reshapes
    r, phi = polar_map(centerX, centerY, imageSizeX, imageSizeY)
    r = r.flatten()
    slices = []
    for r0 in np.linspace(radius_inner, radius - width, n_bins) + width/2:
        diff = np.abs(r - r0)
        vals = np.maximum(0, np.minimum(1, width/2 + 0.5 - diff))
        if normalize:
            s = vals.sum()
            if not np.isclose(s, 0):
                vals /= s
        slices.append(vals.reshape((imageSizeY, imageSizeX)).astype(dtype))
    if radius_inner < 0.5:
        yy = int(np.round(centerY))
        xx = int(np.round(centerX))
        if yy >= 0 and yy < imageSizeY and xx >= 0 and xx < imageSizeX:
            slices[0][yy, xx] = 1 - radius_inner
    return np.stack(slices)


def _make_circular_mask(centerX, centerY, imageSizeX, imageSizeY, radius, antialiased=False):
    # masks.py:18-52
    if antialiased:
        return radial_bins(centerX, centerY, imageSizeX, imageSizeY, radius, n_bins=1,
                           use_sparse=False)[0]
    x, y = np.ogrid[-centerY:imageSizeY-centerY, -centerX:imageSizeX-centerX]
    return x*x + y*y <= radius*radius


def circular(centerX, centerY, imageSizeX, imageSizeY, radius, antialiased=False):
    # masks.py:108-127
    return _make_circular_mask(centerX, centerY, imageSizeX, imageSizeY, radius, antialiased)


def ring(centerX, centerY, imageSizeX, imageSizeY, radius, radius_inner, antialiased=False):
    # masks.py:130-159
    if antialiased:
        return radial_bins(centerX, centerY, imageSizeX, imageSizeY, radius=radius,
                           radius_inner=radius_inner, n_bins=1, use_sparse=False)[0]
    outer = _make_circular_mask(centerX, centerY, imageSizeX, imageSizeY, radius)
    inner = _make_circular_mask(centerX, centerY, imageSizeX, imageSizeY, radius_inner)
    return outer & ~inner


def radial_gradient_background_subtraction(r, r0, r_outer, delta=1):
    # masks.py:176-219
    result = np.zeros_like(r)
    within = r < r0 - delta/2
    result[within] = r[within] / r0
    transition = (r >= r0 - delta/2) * (r < r0 + delta/2)
    result[transition] = (r0 - r[transition]) / (delta/2)
    without = (r >= r0 + delta/2) * (r <= r_outer)
    result[without] = -1
    return result


def radial_gradient(centerX, centerY, imageSizeX, imageSizeY, radius, antialiased=False):
    # masks.py:162-173
    x, y = np.ogrid[-centerY:imageSizeY-centerY, -centerX:imageSizeX-centerX]
    if antialiased:
        r = np.sqrt(x**2 + y**2)
        return radial_gradient_background_subtraction(r=r, r0=radius, r_outer=0)
    return (x*x + y*y <= radius*radius) * (np.sqrt(x*x + y*y) / radius)


def background_subtraction(centerX, centerY, imageSizeX, imageSizeY, radius, radius_inner,
                           antialiased=False):
    # masks.py:356-367
    mask_1 = circular(centerX, centerY, imageSizeX, imageSizeY, radius_inner,
                      antialiased=antialiased)
    sum_1 = np.sum(mask_1)
    mask_2 = ring(centerX, centerY, imageSizeX, imageSizeY, radius, radius_inner,
                  antialiased=antialiased)
    sum_2 = np.sum(mask_2)
    return mask_1 - mask_2*sum_1/sum_2


def rectangular(X, Y, Width, Height, imageSizeX, imageSizeY):
    # masks.py:370-411
    bool_mask = np.zeros([imageSizeY, imageSizeX], dtype="bool")
    if Height*Width > 0:
        ymin, xmin = min(Y, Y+Height), min(X, X+Width)
        ymax, xmax = max(Y, Y+Height), max(X, X+Width)
    elif Height > 0 and Width < 0:
        ymin, xmin, ymax, xmax = Y, X+Width, Y+Height, X
    elif Height < 0 and Width > 0:
        ymin, xmin, ymax, xmax = Y+Height, X, Y, X+Width
    else:
        ymin, xmin, ymax, xmax = 0, 0, -1, -1
    ymin, xmin, ymax, xmax = int(ymin), int(xmin), int(ymax), int(xmax)
    bool_mask[max(0, ymin):min(ymax+1, imageSizeY), max(0, xmin):min(xmax+1, imageSizeX)] = 1
    return bool_mask


def gradient_x(imageSizeX, imageSizeY, dtype=np.float32):
    # masks.py:415-418
    return np.tile(
        np.ogrid[slice(0, imageSizeX)].astype(dtype), imageSizeY
    ).reshape(imageSizeY, imageSizeX)


def gradient_y(imageSizeX, imageSizeY, dtype=np.float32):
    # masks.py:421-422
    return gradient_x(imageSizeY, imageSizeX, dtype).transpose()


def com_masks(detector_y, detector_x, cy, cx, r, ri=None):
    """udf/com.py:47-97 (+ :547-566 for the annular variant) -> stack (3, H, W)"""
    if ri is None or np.isclose(ri, 0.):
        base = circular(centerX=cx, centerY=cy, imageSizeX=detector_x, imageSizeY=detector_y,
                        radius=r)
    else:
        base = ring(centerX=cx, centerY=cy, imageSizeX=detector_x, imageSizeY=detector_y,
                    radius=r, radius_inner=ri)
    return [
        base,
        gradient_y(imageSizeX=detector_x, imageSizeY=detector_y) * base,
        gradient_x(imageSizeX=detector_x, imageSizeY=detector_y) * base,
    ]


def radial_mask_stack(detector_y, detector_x, cx, cy, ri, ro, n_bins, max_order,
                      dtype=np.complex64):
    """analysis/radialfourier.py:106-146, dense branch -> (n_bins*(max_order+1), H, W)"""
    dtype = np.result_type(dtype, np.complex64)
    rings = radial_bins(centerX=cx, centerY=cy, imageSizeX=detector_x, imageSizeY=detector_y,
                        radius=ro, radius_inner=ri, n_bins=n_bins, use_sparse=False, dtype=dtype)
    orders = np.arange(max_order + 1, dtype=dtype)
    r, phi = polar_map(centerX=cx, centerY=cy, imageSizeX=detector_x, imageSizeY=detector_y)
    modulator = np.exp(phi.astype(dtype) * orders[:, np.newaxis, np.newaxis] * 1j)
    ring_stack = rings[:, np.newaxis, ...] * modulator
    return ring_stack.reshape((-1, detector_y, detector_x))


def radial_mask_stack_csr(detector_y, detector_x, cx, cy, ri, ro, n_bins, max_order,
                          dtype=np.complex64):
    """
    analysis/radialfourier.py:134-145 + :149-161 sparse branch: every ring's nnz repeated for
    each order and multiplied with modulator[order, y, x]; mask index = bin*(max_order+1)+order.
    Returned as CSR (n_masks, px).
    """
    dtype = np.result_type(dtype, np.complex64)
    datas, bins, pix, n_bins = radial_bins_coo(cx, cy, detector_x, detector_y, ro, ri, n_bins,
                                               False, dtype)
    orders = np.arange(max_order + 1, dtype=dtype)
    r, phi = polar_map(centerX=cx, centerY=cy, imageSizeX=detector_x, imageSizeY=detector_y)
    modulator = np.exp(phi.astype(dtype) * orders[:, np.newaxis, np.newaxis] * 1j)
    modulator = modulator.reshape((max_order + 1, -1))
    n_orders = max_order + 1
    dd, rr, cc = [], [], []
    for b in range(n_bins):
        for o in range(n_orders):
            v = datas[b].astype(dtype) * modulator[o, pix[b]]
            dd.append(v.astype(dtype))
            rr.append(np.full(len(v), b * n_orders + o, dtype=np.int64))
            cc.append(pix[b])
    m = sp.coo_matrix((np.concatenate(dd), (np.concatenate(rr), np.concatenate(cc))),
                      shape=(n_bins * n_orders, detector_y * detector_x))
    return m.tocsr()
