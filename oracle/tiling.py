"""
Partitioning and tile-shape negotiation restated for MemoryDataSet-like inputs.
(test infrastructure -- see oracle/__init__.py)

Follows io/dataset/base/partition.py:66-99 (make_slices), io/dataset/base/tiling_scheme.py:
223-526 (Negotiator), io/dataset/base/dataset.py:298-330, io/dataset/base/backend.py:69-119,
io/dataset/memory.py:355-372, common/slice.py:259-312 (subslices).
"""
import math
import warnings
import numpy as np


def prod(t):
    r = 1
    for x in t:
        r *= int(x)
    return r


def partition_boundaries(n_frames, num_partitions):
    # base/partition.py:66-93
    if num_partitions > n_frames:
        warnings.warn("dataset contains fewer frames than specified partitions",
                      RuntimeWarning)
        num_partitions = n_frames
    b = np.linspace(0, n_frames, num=max(2, num_partitions + 1), endpoint=True, dtype=int)
    b = tuple(map(int, b))
    return list(zip(b[:-1], b[1:]))


def _scale(base_shape, factors):
    return tuple(f * bs for f, bs in zip(factors, base_shape))


def _get_scale_factors(shape, containing_shape, size, min_factors=None):
    # tiling_scheme.py:390-426
    if min_factors is None:
        factors = [1] * len(shape)
    else:
        factors = list(min_factors)
    max_factors = tuple(cs // s for s, cs in zip(shape, containing_shape))
    prelim = _scale(shape, factors)
    rest = size / prod(prelim)
    if rest < 1:
        rest = 1
    for idx in range(len(shape)):
        max_factor = max_factors[idx]
        factor = int(math.floor(rest * factors[idx]))
        if factor < factors[idx]:
            factor = factors[idx]
        if factor > max_factor:
            factor = max_factor
        factors[idx] = factor
        prelim = _scale(shape, factors)
        rest = max(1, math.floor(size / prod(prelim)))
    return factors


def negotiate_tileshape(ds_shape, sig_dims, ds_dtype, read_dtype, partition_frames,
                        forced_tileshape=None, depth_pref=32, size_pref=np.inf,
                        method='tile'):
    """
    Returns the tile shape (depth, *sig) the reference negotiates for a MemoryDataSet.

    tiling_scheme.py:223-379 with: no ROI, no corrections, one or more TILE-method UDFs that
    keep the default preferences (udf/base.py:1525-1538).
    """
    ds_sig = tuple(ds_shape[-sig_dims:])
    approx_partition_shape = (partition_frames,) + ds_sig
    itemsize = np.dtype(read_dtype).itemsize
    # base/dataset.py:326-330
    min_sig_size = 4 * 4096 // np.dtype(ds_dtype).itemsize
    # base/backend.py:111-119 via dataset.need_decode (no roi / corrections / decoder)
    need_decode = np.dtype(ds_dtype) != np.dtype(read_dtype)
    # tiling_scheme.py:381-388
    if need_decode:
        io_max_size = 2**20
    else:
        io_max_size = itemsize * prod(approx_partition_shape)
    # tiling_scheme.py:509-525
    if method == 'partition':
        depth = approx_partition_shape[0]
    elif method == 'tile':
        depth = min(depth_pref, approx_partition_shape[0])
    else:
        depth = 1
    # tiling_scheme.py:487-507, memory.py:355-360, base/dataset.py:298-299
    if method in ('frame', 'partition'):
        base_shape = ds_sig
    elif forced_tileshape is not None:
        base_shape = tuple(forced_tileshape[-sig_dims:])
    else:
        base_shape = (1,) * (sig_dims - 1) + (ds_sig[-1],)
    # tiling_scheme.py:456-485
    partition_size = itemsize * prod(approx_partition_shape)
    if method == 'frame':
        size = max(2**20, itemsize * prod(ds_sig))
    elif method == 'partition':
        size = partition_size
    else:
        size = min(size_pref, io_max_size)
        size = max(itemsize * prod(base_shape), size)
    size_px = int(size // itemsize)

    min_factors = _get_scale_factors(base_shape, containing_shape=ds_sig, size=min_sig_size)
    min_base_shape = _scale(base_shape, min_factors)
    max_depth = max(1, size_px // prod(min_base_shape))
    if depth > max_depth:
        depth = max_depth
    full_base_shape = (1,) + tuple(base_shape)
    min_factors = (depth,) + tuple(min_factors)
    factors = _get_scale_factors(full_base_shape, containing_shape=approx_partition_shape,
                                 size=size_px, min_factors=min_factors)
    tileshape = _scale(full_base_shape, factors)
    # memory.py:362-367: the dataset's veto
    if forced_tileshape is not None:
        tileshape = tuple(forced_tileshape)
    return tuple(int(x) for x in tileshape)


def sig_slices(ds_sig, tile_sig):
    """common/slice.py:259-312 on the sig dims: list of (origin, shape), np.ndindex order."""
    ni = tuple(math.ceil(s1 / s) for s1, s in zip(ds_sig, tile_sig))
    out = []
    for indexes in np.ndindex(ni):
        origin = tuple(i * s for i, s in zip(indexes, tile_sig))
        shape = tuple(min(ts, ds - o) for ts, ds, o in zip(tile_sig, ds_sig, origin))
        out.append((origin, shape))
    return out


def iter_tiles(part_start, part_stop, tileshape, ds_sig):
    """
    Tile order inside a partition: frame groups of `depth` outermost, sig slices innermost
    (base/tiling.py:223-239, :86-130). Yields (frame_start, frame_stop, sig_origin, sig_shape,
    scheme_idx).
    """
    depth = tileshape[0]
    slices = sig_slices(ds_sig, tileshape[1:])
    f = part_start
    while f < part_stop:
        f1 = min(f + depth, part_stop)
        for idx, (o, s) in enumerate(slices):
            yield f, f1, o, s, idx
        f = f1
