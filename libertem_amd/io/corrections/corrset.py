"""
`CorrectionSet`: dark frame, gain map and excluded (dead / hot) pixels applied to every tile
before the UDFs see it.  API and semantics of the reference's io/corrections/corrset.py:70-202;
tile-shape adjustment (:13-67, :205-260) restated.

On the MI355X path (`BACKEND_HIP`) tiles are always full frames, so the environment of an
excluded pixel is never cut by a tile boundary; the correction runs on the device
(`device_tables` -> libltmi `ltmi_correct` + `ltmi_repair_pixels`), or is folded into the masks of
linear UDFs (`udf/masks.py`), see DESIGN.md.
"""
import numpy as np

from libertem_amd.common.slice import Slice
from .detector import correct, RepairDescriptor


class ExcludedPixels:
    """Minimal COO container (what the reference keeps as `sparse.COO`): `coords` (ndim, nnz),
    `shape`, `nnz`; sliceable by a tuple of slices like the sig part of a tile slice."""

    def __init__(self, coords, shape):
        self.coords = np.asarray(coords, dtype=np.intp).reshape((len(shape), -1))
        self.shape = tuple(int(s) for s in shape)

    @property
    def nnz(self):
        return self.coords.shape[1]

    def __getitem__(self, slices):
        if not isinstance(slices, tuple):
            slices = (slices,)
        keep = np.ones(self.nnz, dtype=bool)
        starts, new_shape = [], []
        for dim, (sl, size) in enumerate(zip(slices, self.shape)):
            start, stop, step = sl.indices(size)
            if step != 1:
                raise ValueError("only contiguous slices are supported")
            keep &= (self.coords[dim] >= start) & (self.coords[dim] < stop)
            starts.append(start)
            new_shape.append(max(0, stop - start))
        coords = self.coords[:, keep] - np.asarray(starts, dtype=np.intp)[:, None]
        return ExcludedPixels(coords, tuple(new_shape))

    def todense(self):
        out = np.zeros(self.shape, dtype=bool)
        if self.nnz:
            out[tuple(self.coords)] = True
        return out


def _as_excluded(obj):
    """Accept what the reference passes through `sparseconverter.for_backend(..., SPARSE_COO)`
    (corrset.py:106-109): a COO-like object (`.coords`, `.shape`), a scipy.sparse matrix, our
    SparseStack, or a dense "roi-like" array whose non-zero entries are the excluded pixels."""
    if isinstance(obj, ExcludedPixels):
        return obj
    if hasattr(obj, 'coords') and hasattr(obj, 'shape'):
        return ExcludedPixels(np.asarray(obj.coords), obj.shape)
    try:
        import scipy.sparse as sp
        if sp.issparse(obj):
            coo = obj.tocoo()
            nz = coo.data != 0
            return ExcludedPixels(np.stack([coo.row[nz], coo.col[nz]]), coo.shape)
    except ImportError:
        pass
    arr = np.asarray(obj)
    return ExcludedPixels(np.stack(np.nonzero(arr)) if arr.ndim else np.zeros((0, 0)), arr.shape)


def disjunct_multiplier(excluded, sig_shape, base_shape=1, target=1):
    """
    A tile size close to `target` that is a multiple of `base_shape` and none of whose multiples
    (inside the signal extent) is a forbidden boundary position in `excluded`.  Search order as in
    the reference (corrset.py:13-67): start at the multiple nearest to `target`, then alternate
    outwards with growing distance; fall back to one base multiple past the largest forbidden
    position, capped by `sig_shape`.
    """
    excluded = np.asarray(excluded, dtype=np.int64)
    sig_shape, base_shape = int(sig_shape), int(base_shape)
    top = int(excluded.max())
    forbidden = np.zeros(top + 1, dtype=bool)
    forbidden[excluded] = True
    value = base_shape * int(np.round(target / base_shape))
    direction = 1 if value >= target else -1
    for distance in range(top // base_shape + 1):
        value += distance * direction * base_shape
        direction = -direction
        if value <= 0:
            continue
        multiples = np.arange(value, top + 1, value, dtype=np.int64)
        multiples = multiples[multiples < sig_shape]
        if not forbidden[multiples].any():
            return value
    return min((top // base_shape + 1) * base_shape, sig_shape)


def adjust(adjusted_shape_inout, sig_shape, base_shape, excluded_list):
    """In place: per sig dimension, a tile size (multiple of the base shape) such that no
    excluded pixel touches a tile boundary; the full extent if that is hopeless
    (reference corrset.py:205-260)."""
    for dim in range(len(adjusted_shape_inout)):
        extent = int(sig_shape[dim])
        if extent <= 1:
            continue
        positions = np.unique(excluded_list[dim])
        if len(positions) > extent / 3:
            adjusted_shape_inout[dim] = extent
            continue
        # a boundary may sit neither left nor right of a bad pixel
        forbidden = np.concatenate((positions, positions + 1))
        forbidden = forbidden[forbidden <= extent]
        at_zero = bool(np.any(forbidden == 0))
        m = min(extent, disjunct_multiplier(
            excluded=forbidden[forbidden != 0], sig_shape=extent,
            base_shape=base_shape[dim], target=adjusted_shape_inout[dim]))
        min_size = max(m, 2) if at_zero else m
        if adjusted_shape_inout[dim] < min_size or adjusted_shape_inout[dim] % m != 0:
            adjusted_shape_inout[dim] = m


class CorrectionSet:
    """
    Parameters (reference corrset.py:70-119)
    ----------
    dark : array of the dataset's signal shape, subtracted from every frame
    gain : array of the signal shape, multiplied after the subtraction
    excluded_pixels : COO-like / scipy.sparse / roi-like array of the signal shape
    allow_empty : do not raise if an excluded pixel has no good neighbour (it stays unpatched)
    """

    def __init__(self, dark=None, gain=None, excluded_pixels=None, allow_empty=False):
        self._dark = dark
        self._gain = gain
        if excluded_pixels is not None:
            excluded_pixels = _as_excluded(excluded_pixels)
        self._excluded_pixels = excluded_pixels
        self._allow_empty = allow_empty
        self._descriptors = {}
        self._device_tables = {}
        if not allow_empty and excluded_pixels is not None:
            # fail at construction, not on the workers
            RepairDescriptor(sig_shape=excluded_pixels.shape,
                             excluded_pixels=excluded_pixels.coords, allow_empty=False)

    def get_dark_frame(self):
        return self._dark

    def get_gain_map(self):
        return self._gain

    def get_excluded_pixels(self):
        return self._excluded_pixels

    @property
    def allow_empty(self):
        return self._allow_empty

    def have_corrections(self):
        return any(c is not None for c in (self._dark, self._gain, self._excluded_pixels))

    # --- host path ---------------------------------------------------------------------------------
    def apply(self, data, tile_slice):
        """In place on a floating point tile `data` of `tile_slice` (corrset.py:140-166)."""
        if not self.have_corrections():
            return
        sig_slice = tile_slice.get(sig_only=True)
        dark = self._dark[sig_slice] if self._dark is not None else None
        gain = self._gain[sig_slice] if self._gain is not None else None
        correct(buffer=data, dark_image=dark, gain_map=gain,
                repair_descriptor=self.repair_descriptor(tile_slice.discard_nav()), inplace=True,
                sig_shape=tuple(tile_slice.shape.sig), allow_empty=self._allow_empty)

    def repair_descriptor(self, sig_slice):
        key = (tuple(sig_slice.origin), tuple(sig_slice.shape))
        desc = self._descriptors.get(key)
        if desc is None:
            coords = None
            if self._excluded_pixels is not None:
                coords = self._excluded_pixels[sig_slice.get(sig_only=True)].coords
            desc = RepairDescriptor(sig_shape=tuple(sig_slice.shape.sig), excluded_pixels=coords,
                                    allow_empty=self._allow_empty)
            if len(self._descriptors) > 512:
                self._descriptors.clear()
            self._descriptors[key] = desc
        return desc

    def adjust_tileshape(self, tile_shape, sig_shape, base_shape):
        excl = self._excluded_pixels
        if excl is None or excl.nnz == 0:
            return tile_shape
        adjusted = np.array(tile_shape)
        sig = np.array(sig_shape)
        adjust(adjusted_shape_inout=adjusted, sig_shape=sig, base_shape=np.array(base_shape),
               excluded_list=excl.coords)
        invalid = (adjusted <= 0) | (adjusted > sig)
        adjusted[invalid] = sig[invalid]
        return tuple(int(x) for x in adjusted)

    # --- device path ---------------------------------------------------------------------------------
    def full_frame_descriptor(self, sig_shape):
        sig_shape = tuple(int(s) for s in sig_shape)
        full = Slice(origin=(0,) * len(sig_shape),
                     shape=_sig_only_shape(sig_shape))
        return self.repair_descriptor(full)

    def device_tables(self, device, sig_shape):
        """dark / gain as float64 and the repair tables as int32, uploaded once per device:
        dict(dark, gain, excl, env, cnt : torch tensors or None, n_excl, max_env)."""
        key = (int(device), tuple(int(s) for s in sig_shape))
        hit = self._device_tables.get(key)
        if hit is not None:
            return hit
        import torch
        dev = f'cuda:{int(device)}'
        n_px = int(np.prod(sig_shape))

        def up64(a):
            if a is None:
                return None
            a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
            if a.size != n_px:
                raise ValueError(f"correction array has {a.size} entries, frames have {n_px} pixels")
            return torch.from_numpy(a).to(dev)

        desc = self.full_frame_descriptor(sig_shape)
        n_excl = int(len(desc.exclude_flat))
        tables = {'dark': up64(self._dark), 'gain': up64(self._gain), 'excl': None, 'env': None,
                  'cnt': None, 'n_excl': n_excl,
                  'max_env': int(desc.repair_flat.shape[1]) if n_excl else 0}
        if n_excl:
            tables['excl'] = torch.from_numpy(desc.exclude_flat.astype(np.int32)).to(dev)
            tables['env'] = torch.from_numpy(
                np.ascontiguousarray(desc.repair_flat.astype(np.int32))).to(dev)
            tables['cnt'] = torch.from_numpy(desc.repair_counts.astype(np.int32)).to(dev)
        self._device_tables[key] = tables
        return tables


def _sig_only_shape(sig_shape):
    from libertem_amd.common.shape import Shape
    return Shape(tuple(sig_shape), sig_dims=len(sig_shape))
