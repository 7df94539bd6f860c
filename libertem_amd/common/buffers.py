"""
Result / aux buffers of the UDF runtime.

Own implementation of the subset of the reference's libertem.common.buffers.BufferWrapper
(common/buffers.py:326-946) that the mask / sum / CoM path exercises:

* kinds 'nav' | 'sig' | 'single', `extra_shape`, dtype, `where='device'`, `use`
* worker-side shape = (frames in partition after ROI,) + extra for nav buffers (:419-431),
  main-process shape = (prod(nav) or roi count,) + extra (:433-443)
* `.data` reshapes to nav + extra and embeds ROI results into a NaN/0/False filled array
  (:470-505); `.raw_data` is the flat array (:516-522)
* views for dataset / partition / tile / frame (:720-821)
* `allocate(lib)`: host (NumPy) or device (`HipArray`) memory, zero-initialised (:668-686);
  `export()` copies device buffers back to NumPy once per partition (:901-907)
"""
import numpy as np

from .math import prod
from .hiparray import HipArray


def _fill_value(dtype):
    kind = np.dtype(dtype).kind
    if kind in 'fc':
        return np.nan
    if kind == 'b':
        return False
    return 0


def to_numpy(a):
    if isinstance(a, HipArray):
        return a.cpu()
    return np.asarray(a)


class InvalidMaskError(Exception):
    """a valid-mask that cannot belong to its array: not bool, or not broadcastable to the array's shape"""


class ArrayWithMask:
    """A result array together with the mask of its valid entries -- what `UDF.with_mask(data, mask)` hands back from
    `get_results` (reference common/buffers.py:195-232).  `mask`: bool array broadcastable to `arr.shape`, or a bool."""

    def __init__(self, arr, mask):
        if isinstance(mask, (bool, np.bool_)):
            mask = np.array([bool(mask)])
        mask = np.asarray(mask)
        try:
            mask = np.broadcast_to(mask, arr.shape)
        except ValueError:
            raise InvalidMaskError(
                f"`arr` and `mask` must have compatible shapes (arr.shape={arr.shape} vs mask.shape={mask.shape})")
        if mask.dtype != np.dtype(bool):
            raise InvalidMaskError(f"`mask` should have `dtype=bool` (have {mask.dtype})")
        self._arr = arr
        self._mask = mask

    @property
    def mask(self):
        return np.broadcast_to(self._mask, self._arr.shape)

    @property
    def arr(self):
        return self._arr


def get_inner_slice(arr, axis=0):
    """Slice along `axis` over the FIRST run of positions at which every entry of the other axes is non-zero
    (reference common/buffers.py:235-269); an array without such a position gives the empty slice [n:1]."""
    arr = np.asarray(arr)
    full = np.all(arr != 0, axis=tuple(i for i in range(arr.ndim) if i != axis))
    hits = np.flatnonzero(full)
    if hits.size == 0:
        lo, hi = arr.shape[axis], 0
    else:
        lo = hi = int(hits[0])
        while hi + 1 < full.shape[0] and full[hi + 1]:
            hi += 1
    return tuple(slice(lo, hi + 1) if d == axis else slice(None, None, None) for d in range(arr.ndim))


def get_bbox(arr, eps=1e-8):
    """(min_0, max_0, min_1, max_1, ...): per axis the first and last index that holds an entry with |value| >= eps
    (reference common/buffers.py:272-312); an axis without one gives (n, 0) -- an empty slice in get_bbox_slice."""
    arr = np.asarray(arr)
    hit = arr if arr.dtype == bool else np.abs(arr) >= eps
    out = []
    for ax in range(arr.ndim):
        idx = np.flatnonzero(np.any(hit, axis=tuple(i for i in range(arr.ndim) if i != ax)))
        out.extend((int(idx[0]), int(idx[-1])) if idx.size else (arr.shape[ax], 0))
    return tuple(out)


def get_bbox_2d(arr, eps=1e-8):
    return get_bbox(arr, eps)


def get_bbox_slice(arr):
    b = get_bbox(arr)
    return tuple(slice(b[2 * i], b[2 * i + 1] + 1, None) for i in range(len(b) // 2))


def default_mask(kind, shape, n_extra, valid_nav_mask):
    """see BufferWrapper.make_default_mask; a plain function: a mask that is made lazily must not keep its buffer
    (and the delivery slot behind it) alive through a reference cycle"""
    if kind == 'nav':
        mask = np.zeros(shape, dtype=bool)
        v = np.asarray(valid_nav_mask, dtype=bool)
        mask[:] = v.reshape(v.shape + (1,) * n_extra)
        return mask
    return np.ones(shape, dtype=bool)


def reshaped_view(a, shape):
    """`a` with another shape, as a VIEW: AttributeError where NumPy would have to copy (reference :94-119)"""
    res = a.view()
    res.shape = shape
    return res


def disjoint(sl, slices):
    """True iff `sl` overlaps none of `slices` (reference common/buffers.py:122-123)"""
    return all(sl.intersection_with(o).is_null() for o in slices)


class BufferWrapper:
    def __init__(self, kind, extra_shape=(), dtype="float32", where=None, use=None):
        if kind not in ('nav', 'sig', 'single'):
            raise ValueError(f"invalid buffer kind {kind!r}")
        if use not in (None, 'private', 'result_only'):
            raise ValueError(f"invalid use {use!r}")
        self._kind = kind
        self._extra_shape = tuple(int(x) for x in extra_shape)
        self._dtype = np.dtype(dtype)
        self._where = where
        self.use = use
        self._arr = None             # np.ndarray | HipArray (see the `_data` property)
        self._lazy = False           # zeros of `_shape` are materialised on first access
        self.write_once = False      # allocated without zero fill: the UDF writes, never adds
        self._shape = None
        self._ds_shape = None
        self._roi = None
        self._roi_is_zero = None
        self._valid_mask_value = None        # bool array in the raw shape, or a callable that makes it on first use
        self._ds_partitions = None
        self._contiguous_cache = {}

    @property
    def _data(self):
        if self._lazy:
            self._lazy = False
            self._arr = np.zeros(self._shape, dtype=self._dtype)
        return self._arr

    @_data.setter
    def _data(self, value):
        self._lazy = False
        self._arr = value

    # --- declaration properties --------------------------------------------------------------
    @property
    def kind(self):
        return self._kind

    @property
    def extra_shape(self):
        return self._extra_shape

    @property
    def dtype(self):
        return self._dtype

    @property
    def where(self):
        return self._where

    @property
    def shape(self):
        return self._shape

    def result_buffer_type(self):
        return BufferWrapper

    # --- roi / shape -------------------------------------------------------------------------
    def set_roi(self, roi):
        if roi is not None:
            roi = np.asarray(roi, dtype=bool).reshape(-1)
        self._roi = roi
        self._roi_is_zero = None if roi is None else (np.count_nonzero(roi) == 0)

    @property
    def roi_is_zero(self):
        return bool(self._roi_is_zero)

    def _shape_for_kind(self, kind, orig_shape, roi_count=None):
        if kind == 'nav':
            n = prod(orig_shape.nav) if roi_count is None else roi_count
            return (n,) + self._extra_shape
        if kind == 'sig':
            return tuple(orig_shape.sig) + self._extra_shape
        if kind == 'single':
            return self._extra_shape if self._extra_shape else (1,)
        raise ValueError(kind)

    def set_shape_partition(self, partition, roi=None):
        roi_count = None
        if roi is not None:
            roi_part = np.asarray(roi).reshape(-1)[partition.slice.get(nav_only=True)]
            roi_count = int(np.count_nonzero(roi_part))
        assert partition.shape.nav.dims == 1
        self._shape = self._shape_for_kind(self._kind, partition.shape, roi_count)

    def set_shape_ds(self, dataset_shape, roi=None):
        roi_count = None if roi is None else int(np.count_nonzero(roi))
        self._shape = self._shape_for_kind(self._kind, dataset_shape.flatten_nav(), roi_count)
        self._ds_shape = dataset_shape

    # --- allocation ---------------------------------------------------------------------------
    def allocate(self, lib=None, lazy=False, write_once=False, target=None):
        """lib: None/'numpy' -> host zeros; ('hip', device) -> HipArray zeros when this buffer
        was declared where='device', host zeros otherwise (reference :668-686).
        lazy: host zeros are only materialised when somebody looks at them (the main-process
        buffers of a run whose executor replaces them with the merged device result).
        write_once: the UDF promises to WRITE every element exactly once (no `+=`): the device
        buffer is not zero-filled, and `target` -- a callable(shape, dtype) -> HipArray | None of
        the executor -- may place it straight into the run's final host buffer."""
        if self._shape is None:
            raise RuntimeError("shape must be set before allocate()")
        if isinstance(lib, tuple) and lib[0] == 'hip' and self._where == 'device':
            arr = None
            if write_once and target is not None:
                arr = target(self._shape, self._dtype)
            if arr is None:
                arr = (HipArray.empty if write_once else HipArray.zeros)(
                    self._shape, self._dtype, lib[1])
            self._data = arr
            self.write_once = bool(write_once)
        elif lazy:
            self._arr = None
            self._lazy = True
        else:
            self._data = np.zeros(self._shape, dtype=self._dtype)

    def has_data(self):
        return self._lazy or self._arr is not None

    @property
    def on_device(self):
        return isinstance(self._arr, HipArray)

    @property
    def host_mapped(self):
        """True iff the kernels write this buffer straight into page-locked host memory."""
        from .hiparray import HostMappedArray
        return isinstance(self._arr, HostMappedArray)

    def export(self):
        """D2H once per partition (reference :901-907)."""
        if isinstance(self._data, HipArray):
            self._data = self._data.cpu()

    def replace_array(self, data):
        data = np.asarray(data) if not isinstance(data, HipArray) else data
        if tuple(data.shape) != tuple(self._shape) and data.size == prod(self._shape):
            data = data.reshape(self._shape)
        if tuple(data.shape) != tuple(self._shape):
            raise ValueError(f"shape mismatch: buffer {self._shape}, new data {data.shape}")
        self._data = data

    # --- user-facing data ---------------------------------------------------------------------
    @property
    def device_data(self):
        """The HipArray holding the result in HBM (runs with result_where='device'), else None.
        `.data` / `.raw_data` of such a buffer download it on access."""
        return self._data if isinstance(self._data, HipArray) else None

    @property
    def result_array(self):
        """raw_data, except that a result kept in HBM (result_where='device') is NOT downloaded."""
        return self._data if isinstance(self._data, HipArray) else self.raw_data

    @property
    def raw_data(self):
        return None if self._data is None else to_numpy(self._data)

    @property
    def data(self):
        """nav + extra (or sig + extra) shaped NumPy array; ROI results are embedded into a
        fill-valued full array (reference :470-505)."""
        arr = self.raw_data
        if arr is None:
            return None
        if self._kind != 'nav' or self._ds_shape is None:
            return arr
        shape = tuple(self._ds_shape.nav) + self._extra_shape
        if self._roi is None:
            return arr.reshape(shape)
        if self._roi_is_zero:
            return np.full(shape, _fill_value(self._dtype), dtype=self._dtype)
        wrapper = np.full((prod(self._ds_shape.nav),) + self._extra_shape,
                          _fill_value(self._dtype), dtype=self._dtype)
        wrapper[self._roi] = arr
        return wrapper.reshape(shape)

    def __array__(self, dtype=None, copy=None):
        a = self.data
        return a if dtype is None else a.astype(dtype)

    # --- valid masks (reference :524-633) -----------------------------------------------------
    def make_default_mask(self, valid_nav_mask, dataset_shape=None, roi=None):
        """The valid-mask a buffer gets when `get_results` does not say otherwise, in the buffer's RAW shape: nav
        buffers follow `valid_nav_mask` (flat, compressed to `roi`), broadcast over the extra dimensions; sig and
        single buffers are valid everywhere (reference :524-551)."""
        if dataset_shape is None:
            dataset_shape = self._ds_shape
        if roi is None and self._roi is not None:
            roi = self._roi
        roi_count = None if roi is None else int(np.count_nonzero(roi))
        shape = self._shape_for_kind(self._kind, dataset_shape.flatten_nav(), roi_count)
        return default_mask(self._kind, shape, len(self._extra_shape), valid_nav_mask)

    @property
    def _valid_mask(self):
        # (the default mask of a result is made when somebody asks for it: a run that only reads `.data` does not pay
        #  for n_frames x extra bools)
        v = self._valid_mask_value
        if callable(v):
            v = self._valid_mask_value = v()
        return v

    @_valid_mask.setter
    def _valid_mask(self, value):
        self._valid_mask_value = value

    @property
    def valid_mask(self):
        """bool array of the shape of `data`: which entries hold valid results (reference :553-576).  Never set
        (a buffer outside a run's results): everything counts as valid."""
        if self._ds_shape is None:
            raise RuntimeError("`valid_mask` called without setting the dataset shape")
        vm = self._valid_mask
        if vm is None:
            vm = np.ones(self._shape, dtype=bool)
        if self._kind == 'nav':
            full_shape = tuple(self._ds_shape.nav) + self._extra_shape
            if self._roi is not None:
                out = np.zeros(full_shape, dtype=bool)
                out.reshape((prod(self._ds_shape.nav),) + self._extra_shape)[self._roi] = vm
                return out
            return np.asarray(vm).reshape(full_shape)
        return vm

    @valid_mask.setter
    def valid_mask(self, value):
        self._valid_mask = value

    @property
    def valid_slice_bounding(self):
        """slices into `data` around every valid entry -- may include invalid ones (reference :585-595)"""
        return get_bbox_slice(self.valid_mask)

    def get_valid_slice_inner(self, axis=0):
        """slices into `data`, cut along `axis`, that select valid entries only -- may leave valid ones out
        (reference :597-613)"""
        return get_inner_slice(self.valid_mask, axis=axis)

    @property
    def masked_data(self):
        return np.ma.array(self.data, mask=~self.valid_mask)

    @property
    def raw_masked_data(self):
        """`raw_data` (flat nav axis, compressed to the roi) as a masked array (reference :624-633)"""
        vm = self._valid_mask if self._valid_mask is not None else np.ones(self._shape, dtype=bool)
        return np.ma.array(self.raw_data, mask=~np.asarray(vm))

    # --- views --------------------------------------------------------------------------------
    def _slice_for_partition(self, partition):
        """Rows of the dataset-wide (roi-compressed) nav buffer that belong to `partition`."""
        if self._roi is None:
            return partition.slice.origin[0], partition.slice.origin[0] + partition.slice.shape[0]
        s = partition.slice.adjust_for_roi(self._roi)
        return s.origin[0], s.origin[0] + s.shape[0]

    def _rows(self, start, stop):
        if isinstance(self._data, HipArray):
            return self._data.rows(start, stop)
        return self._data[start:stop]

    def get_view_for_dataset(self, dataset):
        if self._kind == 'nav' and isinstance(self._data, np.ndarray) and self._roi is None:
            return self._data.reshape(tuple(dataset.shape.nav) + self._extra_shape)
        return self._data

    def get_view_for_partition(self, partition):
        """Main-process buffer: the part that `partition` fills (reference :746-759)."""
        if self._kind == 'nav':
            start, stop = self._slice_for_partition(partition)
            return self._rows(start, stop)
        return self._data

    def _tile_rows(self, partition, tile):
        """Worker buffer rows of a tile: tile origin relative to the partition origin, both in
        the roi-compressed numbering (reference :792-821)."""
        p0 = partition.slice.adjust_for_roi(self._roi).origin[0] if self._roi is not None \
            else partition.slice.origin[0]
        t0 = tile.tile_slice.origin[0]
        start = t0 - p0
        return start, start + tile.tile_slice.shape[0]

    def get_view_for_tile(self, partition, tile):
        if self._kind == 'nav':
            start, stop = self._tile_rows(partition, tile)
            return self._rows(start, stop)
        if self._kind == 'sig':
            if isinstance(self._data, HipArray):
                return HipSigView(self._data, tile.tile_slice, self._extra_shape)
            sl = tile.tile_slice.get(sig_only=True)
            return self._data[sl]
        return self._data

    get_contiguous_view_for_tile = get_view_for_tile

    def get_view_for_frame(self, partition, tile, frame_idx):
        if self._kind == 'nav':
            start, _ = self._tile_rows(partition, tile)
            if isinstance(self._data, HipArray):
                v = self._data.rows(start + frame_idx, start + frame_idx + 1)
                return v
            if len(self._extra_shape) == 0:
                return self._data[start + frame_idx:start + frame_idx + 1]
            return self._data[start + frame_idx]
        return self.get_view_for_tile(partition, tile)

    def flush(self, debug=False):
        self._contiguous_cache.clear()

    def __repr__(self):
        return (f"<{type(self).__name__} kind={self._kind} dtype={self._dtype} "
                f"extra_shape={self._extra_shape} shape={self._shape}>")

    def __getstate__(self):
        d = dict(self.__dict__)
        if isinstance(d.get('_data'), HipArray):
            d['_data'] = d['_data'].cpu()
        d['_contiguous_cache'] = {}
        return d


class HipSigView:
    """Sig-kind device buffer restricted to a tile's sig slice (whole buffer + the slice)."""
    __slots__ = ('array', 'tile_slice', 'extra_shape')

    def __init__(self, array, tile_slice, extra_shape):
        self.array = array
        self.tile_slice = tile_slice
        self.extra_shape = extra_shape


class PlaceholderBufferWrapper(BufferWrapper):
    """Declared with use='result_only': only filled in get_results (reference :949-986)."""

    def allocate(self, lib=None, lazy=False):
        self._data = None

    def has_data(self):
        return False

    def export(self):
        pass

    def _no_view(self, *a, **k):
        return None

    get_view_for_partition = _no_view
    get_view_for_tile = _no_view
    get_view_for_frame = _no_view
    get_contiguous_view_for_tile = _no_view

    @property
    def data(self):
        raise ValueError("this BufferWrapper doesn't have a value associated with it "
                         "(use='result_only': it only exists in what get_results() returns)")

    @property
    def raw_data(self):
        raise ValueError("this BufferWrapper doesn't have a value associated with it "
                         "(use='result_only': it only exists in what get_results() returns)")

    def result_buffer_type(self):
        return BufferWrapper


class PreallocBufferWrapper(BufferWrapper):
    def __init__(self, data, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._data = data
        self._shape = tuple(data.shape)


class AuxBufferWrapper(BufferWrapper):
    """Read-only per-nav (or sig/single) input data handed to a UDF (reference :995-1048)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._data_coords_global = True

    def set_buffer(self, buf, is_global=True):
        buf = np.asarray(buf)
        self._data = buf.reshape((-1,) + self._extra_shape) if self._kind == 'nav' else buf
        self._shape = self._data.shape
        self._data_coords_global = is_global

    def new_for_partition(self, partition, roi):
        if self._kind != 'nav':
            return self
        assert self._data_coords_global
        new = AuxBufferWrapper(self._kind, self._extra_shape, self._dtype)
        data = self._data
        part = data[partition.slice.origin[0]:partition.slice.origin[0] + partition.slice.shape[0]]
        if roi is not None:
            roi_part = np.asarray(roi).reshape(-1)[partition.slice.get(nav_only=True)]
            part = part[roi_part]
        new.set_buffer(part, is_global=False)
        new.set_roi(roi)
        assert np.allclose(new._data.shape[1:], self._extra_shape) or not self._extra_shape
        return new

    def get_view_for_dataset(self, dataset):
        # the positions the run covers (reference :1025-1026)
        if self._kind == 'nav' and self._roi is not None and self._data_coords_global:
            return self._data[self._roi]
        return self._data

    def get_view_for_partition(self, partition):
        return self._data

    def get_view_for_tile(self, partition, tile):
        if self._kind == 'nav':
            start, stop = self._tile_rows(partition, tile)
            return self._data[start:stop]
        return self._data

    get_contiguous_view_for_tile = get_view_for_tile

    def get_view_for_frame(self, partition, tile, frame_idx):
        if self._kind == 'nav':
            start, _ = self._tile_rows(partition, tile)
            return self._data[start + frame_idx]
        return self._data
