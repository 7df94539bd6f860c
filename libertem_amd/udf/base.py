"""
The UDF runtime: `UDF`, `UDFMeta`, `UDFData`, `UDFTask`, `UDFPartRunner`, `UDFRunner`.

Own implementation of the part of the reference's libertem.udf.base (udf/base.py, 2863 lines) that
`Context.run_udf` needs for the mask / sum / CoM path, with the same method names, argument
meaning and error behaviour:

* user interface: `get_result_buffers`, `get_task_data`, `get_backends`, `get_method`,
  `get_preferred_input_dtype`, `get_tiling_preferences`, `process_tile|frame|partition`,
  `preprocess`, `postprocess`, `merge`, `get_results`, `buffer()`, `aux_data()`, `forbuf()`
  (udf/base.py:1270-1732)
* per-partition driver (udf/base.py:2100-2335): init buffers -> tile loop with views -> export
* main-process driver (udf/base.py:2338-2800): plan, negotiate, one task per partition,
  serial merge in completion order, damage map, lazy `get_results`
* dtype rule `input_dtype = result_type(preferred, dataset dtype)` (udf/base.py:106-123)

New relative to the reference: the execution plan knows BACKEND_HIP.  On a HIP worker a UDF that
lists BACKEND_HIP gets device-resident tiles (`HipArray`, native dtype) and device result buffers;
there is no silent fallback: a HIP-only UDF on a CPU worker raises `HipRequiredError`.
"""
import copy
import itertools
import sys
import uuid
import threading
import weakref
from collections import OrderedDict

import functools
import numpy as np

from libertem_amd.common.math import prod
from libertem_amd.common.slice import Slice
from libertem_amd.common.shape import Shape
from libertem_amd.common.buffers import (
    BufferWrapper, AuxBufferWrapper, PlaceholderBufferWrapper, PreallocBufferWrapper, HipSigView, ArrayWithMask,
    default_mask,
)
from libertem_amd.common.hiparray import HipArray
from libertem_amd.common.fingerprint import fingerprint, is_opaque
from libertem_amd.hip import ReplayMismatch as _ReplayMismatch
from libertem_amd.common.udf import UDFProtocol, UDFMethod, NUMPY, HIP
from libertem_amd.common.exceptions import UDFException, UDFRunCancelled, JobCancelledError, \
    HipRequiredError
from libertem_amd.io.dataset.base import Negotiator


_RUN_IDS = itertools.count()
_PLAN_EPOCH = [0]          # cached run plans of an earlier epoch are never re-used (invalidate_plans)


def invalidate_plans():
    """every cached run plan of this process is stale from now on (Context.invalidate_caches)"""
    _PLAN_EPOCH[0] += 1




def check_cast(fromvar, tovar):
    if not np.can_cast(fromvar.dtype, tovar.dtype, casting='safe'):
        raise TypeError(f"Unsafe automatic casting from {fromvar.dtype} to {tovar.dtype}")


def _get_dtype(udfs, dtype, corrections=None):
    """udf/base.py:106-123"""
    tmp = np.dtype(dtype)
    if corrections is not None and getattr(corrections, 'have_corrections', lambda: False)():
        tmp = np.result_type(np.float32, tmp)
    for udf in udfs:
        tmp = np.result_type(udf.get_preferred_input_dtype(), tmp)
    return np.dtype(tmp)


class UDFMeta:
    """udf/base.py:332-593"""

    def __init__(self, partition_slice, dataset_shape, roi, dataset_dtype, input_dtype,
                 tiling_scheme=None, tiling_index=0, corrections=None, device_class=None,
                 threads_per_worker=None, array_backend=None, valid_nav_mask=None, gpu_id=None,
                 stream_ptr=None):
        self._partition_slice = partition_slice
        #: hipStream_t (int) the worker enqueues on; None = torch's current stream
        self.stream_ptr = stream_ptr
        #: True: the dataset hands out RAW tiles and the UDFs apply `corrections` themselves
        self.corrections_folded = False
        self._dataset_shape = dataset_shape
        self._dataset_dtype = dataset_dtype
        self._input_dtype = input_dtype
        self._tiling_scheme = tiling_scheme
        self._tiling_index = tiling_index
        if device_class is None:
            device_class = 'cpu'
        self._device_class = device_class
        self._gpu_id = gpu_id
        self._roi = roi
        self._slice = None
        self._cached_coordinates = None
        self._corrections = corrections
        self._threads_per_worker = threads_per_worker
        self._array_backend = array_backend
        self._valid_nav_mask = valid_nav_mask

    @property
    def slice(self):
        return self._slice

    @slice.setter
    def slice(self, new_slice):
        self._slice = new_slice

    @property
    def partition_shape(self):
        if self._partition_slice is None:
            raise ValueError("cannot get partition_shape if partition_slice is None")
        return self._partition_slice.shape

    @property
    def dataset_shape(self):
        return self._dataset_shape

    @property
    def tiling_scheme(self):
        return self._tiling_scheme

    @property
    def tiling_scheme_idx(self):
        return self._tiling_index

    @tiling_scheme_idx.setter
    def tiling_scheme_idx(self, new_idx):
        self._tiling_index = new_idx

    @property
    def sig_slice(self):
        """Signal slice of the current tile, taken from the tiling scheme (udf/base.py:438-448)."""
        assert self._tiling_scheme is not None
        return self._tiling_scheme[self._tiling_index]

    @property
    def roi(self):
        return self._roi

    @property
    def dataset_dtype(self):
        return self._dataset_dtype

    @property
    def input_dtype(self):
        return self._input_dtype

    @property
    def corrections(self):
        return self._corrections

    @property
    def device_class(self):
        return self._device_class

    @property
    def gpu_id(self):
        return self._gpu_id

    @property
    def threads_per_worker(self):
        return self._threads_per_worker

    @property
    def array_backend(self):
        return self._array_backend

    @property
    def coordinates(self):
        """nav coordinates of the frames of the current tile (udf/base.py:502-519)."""
        if self._slice is None:
            raise UDFException("coordinates are only available while a tile is processed")
        start = self._slice.origin[0]
        n = self._slice.shape[0]
        nav = tuple(self._dataset_shape.nav)
        if self._roi is not None:
            flat_idx = np.flatnonzero(np.asarray(self._roi).reshape(-1))[start:start + n]
        else:
            flat_idx = np.arange(start, start + n)
        return np.stack(np.unravel_index(flat_idx, nav), axis=1)

    def get_valid_nav_mask(self, full_nav=False):
        m = self._valid_nav_mask
        if m is None:
            return None
        if full_nav and self._roi is not None:
            full = np.zeros(prod(self._dataset_shape.nav), dtype=bool)
            full[np.asarray(self._roi).reshape(-1)] = m
            return full
        return m

    def set_valid_nav_mask(self, new_valid_nav_mask):
        self._valid_nav_mask = new_valid_nav_mask


class MergeAttrMapping:
    """dest/src argument of `merge`: attribute access to raw arrays (udf/base.py:596-625)."""

    def __init__(self, dict_input):
        object.__setattr__(self, '_dict', dict_input)

    def __iter__(self):
        return iter(self._dict)

    def __contains__(self, k):
        return k in self._dict

    def __getattr__(self, k):
        try:
            return self._dict[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        # `dest.x = value` and `dest.x += value` write INTO the buffer (udf/base.py:609-613)
        if k == '_dict':
            object.__setattr__(self, k, v)
            return
        target = self._dict[k]
        if target is not v:                      # (`+=` on an array already updated it in place)
            target[:] = v

    def __getitem__(self, k):
        return self._dict[k]


class UDFData:
    """Container of BufferWrappers with view-aware attribute access (udf/base.py:628-778)."""

    def __init__(self, data):
        self._data = data
        self._views = {}

    def __repr__(self):
        return "<UDFData: %r>" % (self._data,)

    def __getattr__(self, k):
        if k.startswith("_"):
            raise AttributeError(k)
        try:
            return self._get_view_or_data(k)
        except KeyError as e:
            raise AttributeError(str(e))

    def get_buffer(self, name):
        return self._data[name]

    def set_buffer(self, name, buffer):
        self._data[name] = buffer

    def get(self, k, default=None):
        try:
            return self.__getattr__(k)
        except (KeyError, AttributeError):
            return default

    def __setattr__(self, k, v):
        if not k.startswith("_"):
            # `self.results.x = value` and `self.results.x += value` are slice assignments into the current view
            # (udf/base.py:673-678)
            target = getattr(self, k)
            if target is not v:                  # (`+=` on an array already updated it in place)
                target[:] = v
        else:
            super().__setattr__(k, v)

    def _get_view_or_data(self, k):
        if k in self._views:
            return self._views[k]
        res = self._data[k]
        if isinstance(res, BufferWrapper):
            return res.raw_data
        return res

    def __getitem__(self, k):
        return self._data[k]

    def __contains__(self, k):
        return k in self._data

    def items(self):
        return self._data.items()

    def keys(self):
        return self._data.keys()

    def values(self):
        return self._data.values()

    def as_dict(self):
        return dict(self.items())

    def get_proxy(self):
        return MergeAttrMapping({
            k: (self._views[k] if k in self._views else v.raw_data)
            for k, v in self._data.items()
            if isinstance(v, BufferWrapper) and v.has_data() or k in self._views
        })

    def _get_buffers(self, filter_allocated=False):
        for k, buf in self._data.items():
            if isinstance(buf, BufferWrapper):
                if filter_allocated and not buf.has_data():
                    continue
                yield k, buf

    def allocate_for_part(self, partition, roi, lib=None, write_once=(), target=None):
        for k, buf in self._get_buffers():
            buf.set_roi(roi)
            buf.set_shape_partition(partition, roi)
        for k, buf in self._get_buffers():
            if not isinstance(buf, AuxBufferWrapper):
                if k in write_once:
                    buf.allocate(lib=lib, write_once=True,
                                 target=None if target is None else
                                 (lambda shape, dtype, k=k, buf=buf: target(k, buf, shape, dtype)))
                else:
                    buf.allocate(lib=lib)

    def allocate_for_full(self, dataset, roi, lazy=False):
        for k, buf in self._get_buffers():
            buf.set_roi(roi)
            buf.set_shape_ds(dataset.shape, roi)
        for k, buf in self._get_buffers():
            if not isinstance(buf, AuxBufferWrapper):
                buf.allocate(lazy=lazy)

    def set_view_for_dataset(self, dataset):
        for k, buf in self._get_buffers(filter_allocated=True):
            self._views[k] = buf.get_view_for_dataset(dataset)

    def set_view_for_partition(self, partition):
        for k, buf in self._get_buffers(filter_allocated=True):
            self._views[k] = buf.get_view_for_partition(partition)

    def set_view_for_tile(self, partition, tile):
        for k, buf in self._get_buffers(filter_allocated=True):
            self._views[k] = buf.get_view_for_tile(partition, tile)

    set_contiguous_view_for_tile = set_view_for_tile

    def set_view_for_frame(self, partition, tile, frame_idx):
        for k, buf in self._get_buffers(filter_allocated=True):
            self._views[k] = buf.get_view_for_frame(partition, tile, frame_idx)

    def flush(self, debug=False):
        for k, buf in self._get_buffers(filter_allocated=True):
            buf.flush(debug=debug)

    def export(self):
        for k, buf in self._get_buffers(filter_allocated=True):
            buf.export()

    def clear_views(self):
        self._views = {}


class UDFKwargsWrapper(UDFData):
    """UDF parameters; aux buffers are re-sliced per partition (udf/base.py:781-802)."""

    def _get_buffers(self, filter_allocated=False):
        for k, buf in self._data.items():
            if isinstance(buf, AuxBufferWrapper):
                yield k, buf

    def new_for_partition(self, partition, roi):
        for k, buf in self._get_buffers():
            self._data[k] = buf.new_for_partition(partition, roi)


class UDFBase(UDFProtocol):
    def __init__(self, *args, **kwargs):
        self._backend = None
        self.params = None
        self.task_data = None
        self.results = None
        self._requires_custom_merge = None
        self.meta = None
        self._main_process_gpu = None

    def get_task_data(self):
        raise NotImplementedError()

    def get_result_buffers(self):
        raise NotImplementedError()

    def allocate_for_part(self, partition, roi, target=None):
        """target: callable(name, buffer, shape, dtype) -> HipArray | None of the executor: where
        the rows of a write-once nav buffer of THIS partition may be written directly."""
        once = self.get_write_once_buffers() if self._backend == HIP else ()
        for ns in [self.results]:
            ns.allocate_for_part(partition, roi, lib=self.xp, write_once=once, target=target)

    def get_write_once_buffers(self):
        """Names of where='device' result buffers whose every element process_tile WRITES exactly
        once in this run (no accumulation over sig slices, untouched by postprocess): they need
        no zero fill and may live in the run's final host buffer.  Known after set_meta()."""
        return ()

    def allocate_for_full(self, dataset, roi):
        self.params.allocate_for_full(dataset, roi)
        # result buffers: zeros on first touch (a device-merging executor replaces them unseen)
        self.results.allocate_for_full(dataset, roi, lazy=True)

    def set_views_for_dataset(self, dataset):
        for ns in [self.params]:
            ns.set_view_for_dataset(dataset)

    def set_views_for_partition(self, partition):
        for ns in [self.params, self.results]:
            ns.set_view_for_partition(partition)

    def set_views_for_tile(self, partition, tile):
        for ns in [self.params, self.results]:
            ns.set_view_for_tile(partition, tile)

    set_contiguous_views_for_tile = set_views_for_tile

    def flush(self, debug=False):
        for ns in [self.params, self.results]:
            ns.flush(debug=debug)

    def set_views_for_frame(self, partition, tile, frame_idx):
        for ns in [self.params, self.results]:
            ns.set_view_for_frame(partition, tile, frame_idx)

    def clear_views(self):
        for ns in [self.params, self.results]:
            ns.clear_views()

    def init_task_data(self):
        self.task_data = UDFData(self.get_task_data())

    def init_result_buffers(self, executor=None):
        self.results = UDFData(self.get_result_buffers())

    def export_results(self):
        self.results.export()

    def set_meta(self, meta):
        self.meta = meta

    def set_slice(self, slice_):
        self.meta.slice = slice_

    def set_tile_idx(self, idx):
        self.meta.tiling_scheme_idx = idx

    def set_backend(self, backend):
        self._backend = backend

    def get_backends(self):
        raise NotImplementedError()

    @property
    def xp(self):
        """Allocation library for result buffers: None (NumPy) or ('hip', device)."""
        if self._backend == HIP:
            gpu = self.meta.gpu_id if self.meta is not None else 0
            return ('hip', 0 if gpu is None else gpu)
        return None

    def get_method(self):
        """Autodetect from the process_* methods present (udf/base.py:1148-1176)."""
        if hasattr(self, 'process_tile'):
            return UDFMethod.TILE
        if hasattr(self, 'process_frame'):
            return UDFMethod.FRAME
        if hasattr(self, 'process_partition'):
            return UDFMethod.PARTITION
        raise TypeError("UDF should implement one of the `process_*` methods")

    def _check_results(self, decl, arr, name):
        """All returned buffers must be declared, not private, and of the declared dtype KIND
        (udf/base.py:1178-1201)."""
        if name not in decl:
            raise UDFException(
                "buffer '%s' is not declared in `get_result_buffers` "
                "(hint: `self.buffer(..., use='result_only')`" % name)
        buf_decl = decl[name]
        if buf_decl.use == "private":
            raise UDFException("Don't return `use='private'` buffers from `get_results`")
        if np.dtype(arr.dtype).kind != np.dtype(buf_decl.dtype).kind:
            raise UDFException(
                "the returned ndarray '%s' has a different dtype kind (%s) than declared (%s)" % (
                    name, arr.dtype, buf_decl.dtype))

    def get_results(self):
        raise NotImplementedError()

    def _do_get_results(self):
        """Run get_results, wrap arrays into buffers, attach valid masks (udf/base.py:1226-1267)."""
        from libertem_amd.common.buffers import to_numpy
        results_tmp = dict(self.get_results())
        # the declarations of this run (made by init_result_buffers from get_result_buffers())
        decl = self.results.as_dict() if self.results is not None else self.get_result_buffers()
        # buffers with use=None that get_results did not mention are included as they are
        for k, v in decl.items():
            if k not in results_tmp and v.use is None:
                results_tmp[k] = self.results.get_buffer(k).result_array
        results = {}
        for name, arr in results_tmp.items():
            mask = None
            if isinstance(arr, ArrayWithMask):
                arr, mask = arr.arr, arr.mask
            if not isinstance(arr, HipArray):   # (HipArray: a run with result_where='device')
                arr = to_numpy(arr)
            self._check_results(decl, arr, name)
            buf_decl = decl[name]
            buf = PreallocBufferWrapper(
                arr.reshape(self._result_shape(buf_decl, arr)), kind=buf_decl.kind,
                extra_shape=buf_decl.extra_shape, dtype=arr.dtype)
            buf.set_roi(self.meta.roi)
            buf.set_shape_ds(self.meta.dataset_shape, self.meta.roi)
            if mask is None:
                vm = self.meta.get_valid_nav_mask()
                if vm is not None:
                    vm = np.array(vm, copy=True)        # (the damage map goes on changing; made into a mask lazily)
                    # (no reference to `buf` in the callable: a cycle would keep the run's delivery slot reserved
                    #  until the garbage collector runs)
                    mask = functools.partial(default_mask, buf.kind, tuple(buf.shape), len(buf.extra_shape), vm)
            else:
                mask = np.asarray(mask).reshape(buf.shape)
            buf.valid_mask = mask
            results[name] = buf
        return results

    def _result_shape(self, buf_decl, arr):
        shape = getattr(buf_decl, 'shape', None)
        if shape is not None and prod(shape) == arr.size and buf_decl._ds_shape is not None:
            return shape                     # the run's own (dataset-shaped) buffer
        tmp = BufferWrapper(buf_decl.kind, buf_decl.extra_shape, buf_decl.dtype)
        tmp.set_roi(self.meta.roi)
        tmp.set_shape_ds(self.meta.dataset_shape, self.meta.roi)
        if prod(tmp.shape) != arr.size:
            raise UDFException(
                f"result array has {arr.size} elements, the declared buffer needs "
                f"{prod(tmp.shape)} ({tmp.shape})")
        return tmp.shape


def _mixin(method, doc):
    """runtime-checkable protocol `has a method <method>` (udf/base.py:804-960 of the reference): user code may
    inherit from these or ask `isinstance(udf, UDFTileMixin)`; the run itself only looks for the methods"""
    from typing import Protocol, runtime_checkable
    ns = {'__doc__': doc, method: lambda self, *a, **k: (_ for _ in ()).throw(NotImplementedError())}
    ns[method].__name__ = method
    return runtime_checkable(type(Protocol)('UDF' + ''.join(w.title() for w in method.split('_')[1:]) + 'Mixin',
                                            (Protocol,), ns))


UDFFrameMixin = _mixin('process_frame', "Implement `process_frame(frame)` for per-frame processing.")
UDFTileMixin = _mixin('process_tile', "Implement `process_tile(tile)` for per-tile processing.")
UDFPartitionMixin = _mixin('process_partition', "Implement `process_partition(partition)` for whole partitions.")
UDFPreprocessMixin = _mixin('preprocess', "Implement `preprocess()`: runs before the tiles of every task.")
UDFPostprocessMixin = _mixin('postprocess', "Implement `postprocess()`: runs after the tiles of every task.")
UDFMergeAllMixin = _mixin('merge_all', "Implement `merge_all(ordered_results)` instead of `merge`.")


class UDF(UDFBase):
    """The user-facing base class (udf/base.py:1270-1732)."""

    #: True: the per-partition instance of this UDF (params, meta, task data) may be kept and re-used
    #: when the SAME udf object runs again on the same dataset (`run_udf` in a loop); every run
    #: still gets fresh result buffers.  The reference creates a new instance per task
    #: (udf/base.py:1997-2003); user UDFs keep that behaviour, the native operators opt in.
    REUSE_TASK_INSTANCES = False

    #: True: `process_tile` accepts a `HipRowsArray` (the frames of a region of interest as a row list
    #: over the resident array) -- other UDFs are handed the gathered frames.
    ACCEPTS_ROW_VIEWS = False

    def __init__(self, **kwargs):
        super().__init__()
        self._kwargs = kwargs
        self.params = UDFKwargsWrapper(kwargs)
        self.task_data = None
        self.results = None

    def copy(self):
        return self.__class__(**self._kwargs)

    @classmethod
    def new_for_partition(cls, kwargs, partition, roi):
        new_instance = cls(**kwargs)
        new_instance.params.new_for_partition(partition, roi)
        return new_instance

    def copy_for_partition(self, partition, roi):
        new_instance = self.__class__.new_for_partition(dict(self._kwargs), partition, roi)
        return new_instance

    def get_task_data(self):
        return {}

    def get_result_buffers(self):
        raise NotImplementedError()

    @property
    def requires_custom_merge(self):
        if self._requires_custom_merge is None:
            buffers = self.get_result_buffers()
            self._requires_custom_merge = any(
                buffer.kind != 'nav' for buffer in buffers.values() if buffer.use != 'result_only')
        return self._requires_custom_merge

    def merge(self, dest, src):
        """Default merge: slice assignment of nav buffers (udf/base.py:1420-1453)."""
        if self.requires_custom_merge:
            raise NotImplementedError(
                "Default merging only works for kind='nav' buffers. "
                "Please implement a suitable custom merge function.")
        for k in dest:
            check_cast(getattr(src, k), getattr(dest, k))
            getattr(dest, k)[:] = getattr(src, k)

    def get_results(self):
        """Default: all non-private buffers as they are (udf/base.py:1455-1493)."""
        for k, buf in self.results.items():
            if buf.use == 'result_only':
                raise NotImplementedError(
                    "Default get_results doesn't handle use='result_only' buffers")
        return {k: self.results.get_buffer(k).result_array for k, buf in self.results.items()
                if buf.use != 'private'}

    def get_preferred_input_dtype(self):
        return np.float32

    def get_tiling_preferences(self):
        return {"depth": UDF.TILE_DEPTH_DEFAULT, "total_size": UDF.TILE_SIZE_MAX}

    def get_backends(self):
        return (self.BACKEND_NUMPY,)

    def forbuf(self, arr, target):
        """Make `arr` assignable to the result view `target` (udf/base.py:1563-1605)."""
        if isinstance(arr, HipArray) or isinstance(target, (HipArray, HipSigView)):
            return arr
        arr = np.asarray(arr)
        tshape = tuple(target.shape)
        if arr.shape != tshape and arr.size == prod(tshape):
            arr = arr.reshape(tshape)
        return arr

    def cleanup(self):
        pass

    @staticmethod
    def with_mask(data, mask):
        """`data` with its own valid-mask, for `get_results` (udf/base.py:1611-1642): a bool array that broadcasts to
        `data.shape`, or True / False; InvalidMaskError otherwise"""
        return ArrayWithMask(data, mask=mask)

    def buffer(self, kind, extra_shape=(), dtype="float32", where=None, use=None):
        """Declare a result buffer (udf/base.py:1644-1692)."""
        if use is not None and use.lower() == "result_only":
            return PlaceholderBufferWrapper(kind, extra_shape, dtype, use=use)
        return BufferWrapper(kind, extra_shape, dtype, where, use=use)

    @classmethod
    def aux_data(cls, data, kind, extra_shape=(), dtype="float32"):
        """Wrap auxiliary per-nav/sig input data (udf/base.py:1694-1732)."""
        buf = AuxBufferWrapper(kind, extra_shape, dtype)
        buf.set_buffer(data)
        return buf


class NoOpUDF(UDF):
    def __init__(self, preferred_input_dtype=bool):
        super().__init__(preferred_input_dtype=preferred_input_dtype)

    def process_tile(self, tile):
        pass

    def get_result_buffers(self):
        return {}

    def get_preferred_input_dtype(self):
        return self.params.preferred_input_dtype


class UDFParams:
    def __init__(self, kwargs, roi, corrections, tiling_scheme, backends=None):
        self._kwargs = kwargs
        self._roi = roi
        self._corrections = corrections
        self._tiling_scheme = tiling_scheme
        self._backends = backends

    @classmethod
    def from_udfs(cls, udfs, roi, corrections, tiling_scheme, backends=None):
        kwargs = [udf._kwargs for udf in udfs]
        return cls(kwargs=kwargs, roi=roi, corrections=corrections, tiling_scheme=tiling_scheme,
                   backends=backends)

    @property
    def roi(self):
        return self._roi

    @property
    def corrections(self):
        return self._corrections

    @property
    def kwargs(self):
        return self._kwargs

    @property
    def tiling_scheme(self):
        return self._tiling_scheme

    @property
    def backends(self):
        return self._backends


def _canonical_backends(backends):
    if backends is None:
        return None
    if isinstance(backends, str):
        backends = (backends,)
    return tuple(backends)


def _execution_plan(udfs, ds_backends, device_class, restrict=None):
    """
    Pick one array backend for all UDFs of a run (the reference builds a multi-backend plan,
    udf/base.py:162-329; with only NumPy and HIP a single choice suffices):
    HIP iff the worker drives a GPU, every UDF lists BACKEND_HIP and the dataset can deliver it.
    """
    restrict = _canonical_backends(restrict)

    def supported(udf):
        b = _canonical_backends(udf.get_backends())
        b = tuple(x for x in b if x in (NUMPY, HIP))
        if restrict is not None:
            b = tuple(x for x in b if x in restrict)
        return b

    per_udf = [supported(u) for u in udfs]
    for u, b in zip(udfs, per_udf):
        if not b:
            raise ValueError(
                f"UDF {type(u).__name__} has no backend in common with the allowed set "
                f"(udf: {u.get_backends()}, restrict: {restrict}; this build runs {(HIP, NUMPY)})")
    if device_class == 'hip' and HIP in ds_backends and all(HIP in b for b in per_udf):
        return HIP
    if device_class == 'hip' and any(HIP in b for b in per_udf) and all(NUMPY in b for b in per_udf) \
            and NUMPY in ds_backends and not all(HIP in b for b in per_udf):
        # a mix of UDFs with a device path and NumPy-only UDFs (`[SumUDF(), MyNumpyUDF()]`): the reference plans
        # per UDF (udf/base.py:162-329); here one backend serves a run, and since EVERY UDF of this run offers
        # NumPy the run happens on the host -- announced, not silent.  The native operators (ApplyMasksUDF, CoMUDF,
        # CrystallinityUDF) do not list NumPy: a mix with those is refused below.
        import warnings
        hipable = [type(u).__name__ for u, b in zip(udfs, per_udf) if HIP in b]
        others = [type(u).__name__ for u, b in zip(udfs, per_udf) if HIP not in b]
        warnings.warn(
            f"{', '.join(others)} run(s) on NumPy only: this run_udf executes on the host, including the NumPy "
            f"branch of {', '.join(hipable)}; run the device-capable UDFs in a run of their own to keep them on "
            "the MI355X", RuntimeWarning, stacklevel=3)
        return NUMPY
    if device_class == 'hip' and any(HIP in b for b in per_udf):
        # a UDF with a device path never takes its NumPy branch on a GPU worker: no silent CPU fallback
        raise ValueError(
            f"no common array backend on this MI355X worker: dataset offers {ds_backends}, UDFs offer {per_udf} "
            "(UDFs that list BACKEND_HIP run on the device or not at all here)")
    if all(NUMPY in b for b in per_udf) and NUMPY in ds_backends:
        if device_class == 'hip':
            # NumPy UDFs may run on a GPU worker's host side, like the reference runs NumPy UDFs
            # on CUDA workers only if they list BACKEND_CUDA; we allow it for plain NumPy UDFs.
            return NUMPY
        return NUMPY
    hip_only = [type(u).__name__ for u, b in zip(udfs, per_udf) if NUMPY not in b]
    if hip_only and device_class != 'hip':
        raise HipRequiredError(
            f"{', '.join(hip_only)} run(s) only on BACKEND_HIP (MI355X); this worker has device "
            f"class {device_class!r}. Use Context.make_with('hip') / HipJobExecutor. "
            "There is no CPU fallback for the native operators.")
    raise ValueError(
        f"no common array backend: dataset offers {ds_backends}, UDFs offer {per_udf}, "
        f"device class {device_class!r}")


class UDFTask:
    """One task per partition (udf/base.py:1936-2091)."""

    def __init__(self, partition, idx, udf_classes, udf_backends=None, runner_cls=None):
        self.partition = partition
        self.idx = idx
        self._udf_classes = udf_classes
        self._udf_backends = udf_backends
        self._runner_cls = runner_cls or UDFPartRunner
        #: per-task state kept between runs of a cached plan (see UDFRunner._plan_for): the
        #: partition's UDF instances with meta / task data, its device tiles, the sink set-up
        self._keep = None

    def __call__(self, params, env):
        keep = self._keep
        if keep is not None and keep.get('udfs') is not None:
            return self._runner_cls(keep['udfs']).run_for_partition(
                self.partition, params, env, backend_choice=params.backends, keep=keep)
        udfs = [cls.new_for_partition(kwargs, self.partition, params.roi)
                for cls, kwargs in zip(self._udf_classes, params.kwargs)]
        return self._runner_cls(udfs).run_for_partition(
            self.partition, params, env, backend_choice=params.backends, keep=keep)

    def get_partition(self):
        return self.partition

    def get_locations(self):
        return self.partition.get_locations()

    def get_resources(self):
        """Resource tags in the style of udf/base.py:2023-2080: 'HIP' for native UDFs."""
        needs_hip = all(HIP in _canonical_backends(b) for b in (self._udf_backends or [])) \
            and bool(self._udf_backends)
        return {'HIP': 1, 'compute': 1} if needs_hip else {'CPU': 1, 'compute': 1, 'ndarray': 1}

    @property
    def task_frames(self):
        return self.partition.shape[0]

    def __repr__(self):
        return f"<UDFTask {self._udf_classes!r}>"


class UDFPartRunner:
    """Per-partition driver on the worker (udf/base.py:2094-2335)."""

    def __init__(self, udfs, debug=False, progress=False):
        self._udfs = udfs
        self._debug = debug

    def run_for_partition(self, partition, params, env, backend_choice=None, keep=None):
        """keep: None, or the task's dict of state that survives between runs of a cached plan
        (only filled when every UDF allows it, `UDF.REUSE_TASK_INSTANCES`)."""
        if keep is not None and keep.get('udfs') is not None:
            # same udf objects, dataset, executor as the run that filled `keep`: only the result
            # buffers are new
            backend, meta = keep['backend'], keep['meta']
            with env.enter(enable_gpu=(backend == HIP)):
                roi = params.roi
                for i, udf in enumerate(self._udfs):
                    udf.init_result_buffers()
                    udf.allocate_for_part(partition, roi,
                                          target=self._result_target(env, i, partition))
                    if hasattr(udf, 'preprocess'):
                        udf.set_slice(partition.slice)  # (as on the first run: the partition, until the first tile)
                        udf.clear_views()           # (the task's whole buffers: udf/base.py:2250-2252)
                        udf.preprocess()
                # the launches of this task, for the executor's launch-ahead of later runs (hip.LaunchReplay):
                # recorded once, on the first re-run of the kept instances
                from libertem_amd import hip as _hip
                record = backend == HIP and 'replay' not in keep and _hip.LaunchReplay.expected is None \
                    and _hip.LaunchReplay.recording is None and keep.get('tiles') is not None
                if record:
                    _hip.LaunchReplay.recording = rec = []
                try:
                    self._run_udfs(partition, params, env, backend, meta, keep=keep)
                finally:
                    if record:
                        _hip.LaunchReplay.recording = None
                if record:
                    keep['replay'] = None                  # (decided by the executor: merge_results)
                    keep['recorded'] = rec
                self._wrapup_udfs(partition, backend, env)
            return self._hand_over_results(kept=True)
        roi = params.roi
        device_class = env.device_class
        ds_backends = partition._ds.array_backends if hasattr(partition, '_ds') else (NUMPY,)
        backend = _execution_plan(self._udfs, ds_backends, device_class, restrict=backend_choice)
        with env.enter(enable_gpu=(backend == HIP)):
            meta = self._init_udfs(partition, params, env, backend)
            if keep is not None and all(u.REUSE_TASK_INSTANCES for u in self._udfs):
                keep.update(udfs=self._udfs, backend=backend, meta=meta)
            self._run_udfs(partition, params, env, backend, meta, keep=keep)
            self._wrapup_udfs(partition, backend, env)
        return self._hand_over_results(kept=keep is not None and keep.get('udfs') is not None)

    def _hand_over_results(self, kept):
        res = tuple(udf.results for udf in self._udfs)
        if kept:
            # instances that outlive the run must not keep its result memory alive (rows in the
            # executor's page-locked ring / the node-shared segment are reserved by references)
            for udf in self._udfs:
                udf.results = None
        return res

    def _init_udfs(self, partition, params, env, backend):
        roi = params.roi
        dtype = _get_dtype(self._udfs, partition.dtype, params.corrections)
        meta = UDFMeta(
            partition_slice=partition.slice.adjust_for_roi(roi),
            dataset_shape=partition.meta.shape, roi=roi, dataset_dtype=partition.dtype,
            input_dtype=dtype, tiling_scheme=params.tiling_scheme,
            corrections=params.corrections, device_class=env.device_class,
            threads_per_worker=env.threads_per_worker, array_backend=backend,
            gpu_id=env.gpu_id, stream_ptr=getattr(env, 'stream_ptr', None),
        )
        # Linear UDFs can absorb the corrections into their own operands (masks' = R^T m * gain,
        # a per-mask constant for the dark frame) and read the raw frames: no corrected copy of
        # the data is ever written.  Only if EVERY udf of the run does so and tiles are full frames.
        corr = params.corrections
        ts = params.tiling_scheme
        meta.corrections_folded = bool(
            corr is not None and corr.have_corrections() and backend == HIP
            and ts is not None and (ts._debug or {}).get('backend') == HIP and len(ts) == 1
            and all(getattr(u, 'folds_corrections', None) is not None
                    and u.folds_corrections(corr, meta) for u in self._udfs))
        # a tileshape forced on the dataset replaces whatever scheme was negotiated when the tiles
        # are read (MemPartition.get_tiles) or the scheme is re-negotiated for the device: if it cuts
        # the frames, result rows are accumulated over several tiles -- no write-once buffers then
        ds = getattr(partition, '_ds', None)
        forced = ds.get_forced_tileshape() if hasattr(ds, 'get_forced_tileshape') else None
        meta.sig_sliced_tiles = bool(
            forced is not None and tuple(forced)[-len(tuple(partition.meta.shape.sig)):]
            != tuple(partition.meta.shape.sig))
        # until the first tile, `meta.slice` / `meta.coordinates` describe the whole partition (compressed to the roi):
        # get_task_data / preprocess may look at them (udf/base.py:2238-2247)
        pslice = partition.slice if roi is None else partition.slice.adjust_for_roi(roi)
        for i, udf in enumerate(self._udfs):
            udf.set_backend(backend)
            udf.set_meta(meta)
            udf.set_slice(pslice)
            udf.init_result_buffers()
            udf.allocate_for_part(partition, roi, target=self._result_target(env, i, partition))
            udf.init_task_data()
            if hasattr(udf, 'preprocess'):
                udf.clear_views()                   # (the task's whole buffers: udf/base.py:2250-2252)
                udf.preprocess()
        return meta

    @staticmethod
    def _result_target(env, i, partition):
        rt = getattr(env, 'result_target', None)
        if rt is None:
            return None

        def target(name, buf, shape, dtype):
            if buf.kind != 'nav':
                return None
            g0, g1 = buf._slice_for_partition(partition)
            if g1 - g0 != shape[0]:
                return None
            return rt(i, name, g0, tuple(shape), dtype)
        return target

    def _run_udfs(self, partition, params, env, backend, meta, keep=None):
        tiling_scheme = params.tiling_scheme
        methods = keep.get('methods') if keep is not None else None
        if methods is None:
            if backend == HIP and tiling_scheme.intent is not None \
                    and (tiling_scheme._debug or {}).get('backend') != HIP:
                # the scheme was negotiated for NumPy on the main process; re-negotiate for the device
                tiling_scheme = Negotiator().get_scheme(
                    udfs=self._udfs, dataset=partition._ds, read_dtype=meta.input_dtype,
                    approx_partition_shape=partition.shape, roi=params.roi,
                    corrections=params.corrections, backend=HIP)
                meta._tiling_scheme = tiling_scheme
            methods = [udf.get_method() for udf in self._udfs]
            if keep is not None and keep.get('udfs') is not None:
                keep['methods'] = methods
                keep['scheme'] = tiling_scheme
        else:
            tiling_scheme = keep['scheme']
        tiles = keep.get('tiles') if keep is not None else None
        if tiles is None:
            tiles = partition.get_tiles(
                tiling_scheme=tiling_scheme, roi=params.roi, dest_dtype=meta.input_dtype,
                array_backend=backend, env=env,
                corrections=None if getattr(meta, 'corrections_folded', False)
                else params.corrections)
            ds = getattr(partition, '_ds', None)
            if keep is not None and keep.get('udfs') is not None and backend == HIP \
                    and params.roi is None and params.corrections is None \
                    and getattr(ds, 'is_device_resident', False) \
                    and getattr(ds, 'stable_device_tiles', True):
                # device-resident frames: the tiles are zero-copy views of HBM, the same every run
                # (not a streamed .mib series: its partitions share one window of HBM)
                tiles = keep['tiles'] = list(tiles)
        sink = getattr(env, 'row_sink', None)
        sinkable = None
        if sink is not None and backend == HIP and len(tiling_scheme) == 1:
            # tiles are whole frames: the rows of a tile in a 'disjoint' nav buffer are final as
            # soon as its kernels are enqueued -> start their D2H now (UDFs that touch their
            # buffers again in postprocess() are left to the normal export)
            sinkable = []
            for i, (udf, method) in enumerate(zip(self._udfs, methods)):
                decl = getattr(udf, 'get_dist_merge', lambda: None)()
                if decl is None or method != UDFMethod.TILE or hasattr(udf, 'postprocess'):
                    continue
                names = [k for k, how in decl.items() if how == 'disjoint']
                names = [k for k in names if isinstance(udf.results.get_buffer(k), BufferWrapper)
                         and udf.results.get_buffer(k).on_device
                         and not udf.results.get_buffer(k).host_mapped
                         and udf.results.get_buffer(k).kind == 'nav']
                if names:
                    sinkable.append((i, udf, names))
        for tile in tiles:
            for udf, method in zip(self._udfs, methods):
                try:
                    self._run_tile(udf, method, partition, tile)
                except AttributeError as e:
                    # tiles are plain arrays: what DataTile objects once carried is on `self.meta`
                    # (udf/base.py:2196-2206)
                    for old, new in (('tile_slice', 'self.meta.slice'), ('scheme_idx', 'self.meta.tiling_scheme_idx')):
                        if e.args and isinstance(e.args[0], str) and old in e.args[0]:
                            raise AttributeError(
                                f'Attribute {old} for input tiles was removed. Please use {new} instead.') from e
                    raise
            if sinkable:
                for i, udf, names in sinkable:
                    for name in names:
                        buf = udf.results.get_buffer(name)
                        start, stop = buf._tile_rows(partition, tile)
                        g0, _ = buf._slice_for_partition(partition)
                        sink(i, name, buf._rows(start, stop), g0 + start)
        for udf in self._udfs:
            udf.flush(self._debug)

    def _run_tile(self, udf, method, partition, tile):
        data = tile.data
        if hasattr(data, 'materialize') and not (
                method == UDFMethod.TILE and getattr(udf, 'ACCEPTS_ROW_VIEWS', False)):
            # a region of interest as a row list over the resident frames: only the mask operators
            # read through it, everyone else gets the gathered frames (gathered once per tile)
            data = data.materialize()
        if method == UDFMethod.TILE:
            udf.set_contiguous_views_for_tile(partition, tile)
            udf.set_slice(tile.tile_slice)
            udf.set_tile_idx(tile.scheme_idx)
            udf.process_tile(data)
        elif method == UDFMethod.FRAME:
            tile_slice = tile.tile_slice
            for frame_idx in range(data.shape[0]):
                frame_slice = Slice(
                    origin=(tile_slice.origin[0] + frame_idx,) + tile_slice.origin[1:],
                    shape=Shape((1,) + tuple(tile_slice.shape)[1:],
                                sig_dims=tile_slice.shape.sig_dims))
                udf.set_slice(frame_slice)
                udf.set_views_for_frame(partition, tile, frame_idx)
                frame = data.rows(frame_idx, frame_idx + 1).reshape(data.shape[1:]) \
                    if isinstance(data, HipArray) else data[frame_idx]
                udf.process_frame(frame)
        elif method == UDFMethod.PARTITION:
            udf.set_views_for_tile(partition, tile)
            udf.set_slice(tile.tile_slice)
            udf.process_partition(data)

    def _wrapup_udfs(self, partition, backend, env):
        for udf in self._udfs:
            udf.flush(self._debug)
            if hasattr(udf, 'postprocess'):
                udf.clear_views()
                udf.postprocess()
            udf.cleanup()
            udf.clear_views()
            if not env.keep_results_on_device:
                udf.export_results()


class _StalePlan(Exception):
    """a cached plan whose parameter contents no longer match (UDFRunner._prepare_run_for_dataset)"""


class UDFResults:
    """udf/base.py:2806-2831"""

    def __init__(self, buffers, damage):
        self.buffers = buffers
        self.damage = damage


class UDFResultsLazy:
    def __init__(self, udfs, cb, damage):
        self._udfs = udfs
        self._cb = cb
        self.damage = damage
        self._buffers = None

    @property
    def buffers(self):
        if self._buffers is None:
            self._buffers = self._cb()
        return self._buffers


class UDFRunner:
    """Main-process driver (udf/base.py:2338-2800)."""

    _lock = threading.Lock()

    def __init__(self, udfs, debug=False, progress_reporter=None):
        self._udfs = udfs
        self._debug = debug

    @staticmethod
    def _apply_part_result(udfs, damage, part_results, task):
        for results, udf in zip(part_results, udfs):
            udf.meta.set_valid_nav_mask(damage.raw_data)        # (what is merged so far: udf/base.py:2351)
            udf.set_views_for_partition(task.partition)
            udf.merge(dest=udf.results.get_proxy(), src=results.get_proxy())
            udf.clear_views()
        v = damage.get_view_for_partition(task.partition)
        v[:] = True

    @staticmethod
    def _make_udf_result(udfs, damage):
        def _cb():
            with UDFRunner._lock:
                for udf in udfs:
                    udf.clear_views()
                    udf.meta.set_valid_nav_mask(damage.raw_data)
                return tuple(udf._do_get_results() for udf in udfs)
        return UDFResultsLazy(udfs, _cb, damage)

    def _check_preconditions(self, dataset, roi):
        if roi is not None and prod(roi.shape) != prod(dataset.shape.nav):
            raise ValueError("roi: incompatible shapes: %s (roi) vs %s (dataset)" % (
                roi.shape, dataset.shape.nav))

    #: plans kept per dataset (`run_udf` in a loop with the same udf objects)
    PLAN_CACHE_SIZE = 8

    def _plan_key(self, executor, roi, corrections, backends, dry):
        """Identity of a run that may re-use the plan of an earlier one: the SAME udf objects with
        the same parameter objects, on the same executor, no ROI, no corrections.  None: plan afresh."""
        if roi is not None or corrections is not None or dry:
            return None
        parts = []
        for u in self._udfs:
            kw = getattr(u, '_kwargs', None)
            if kw is None or kw.get('cache', True) is False:      # (cache=False: planned afresh, like the reference)
                return None
            parts.append((id(u), tuple((k, id(v)) for k, v in kw.items())))
        from libertem_amd.common import udf as udf_common
        knobs = (udf_common.HIP_DIRECT_ROW_MAX,) + tuple(
            getattr(sys.modules.get(type(u).__module__), 'FOLD_CORRECTIONS', None)
            for u in self._udfs)
        return (id(executor), _canonical_backends(backends), tuple(parts), knobs)

    def _plan_content(self):
        """Content fingerprints of the parameters (common/fingerprint.py): a list that is mutated in
        place between runs (mask factories appended or replaced), an ndarray parameter or an array
        captured by a mask factory that is modified in place, is a different parameter (the reference
        re-instantiates the UDFs and re-evaluates the factories on every run).  Reading the sample
        costs ~80 us for the 4 MiB C2 stack, so a run that hits the cache by identity starts on the
        cached plan and compares the contents while the kernels run (`_prepare_run_for_dataset`)."""
        return tuple(tuple((k, fingerprint(v)) for k, v in u._kwargs.items()) for u in self._udfs)

    def _prepare_run_for_dataset(self, dataset, executor, roi, corrections, backends, dry,
                                 defer_check=False, result_where=None):
        """-> (tasks, params, verify).  `verify` (None, or a callable that raises _StalePlan) is the
        content comparison of a cache hit that the caller asked to run late (`defer_check`): after
        the kernels of the run are enqueued, before anything is delivered."""
        key = self._plan_key(executor, roi, corrections, backends, dry)
        plans = None
        if key is not None:
            try:
                plans = dataset.__dict__.setdefault('_udf_plans', OrderedDict())
            except AttributeError:
                plans = None
        hit = plans.get(key) if plans is not None else None
        verify = None
        if hit is not None and all(a() is b for a, b in zip(hit['udfs'], self._udfs)) \
                and hit['executor'] is executor and hit.get('epoch') == _PLAN_EPOCH[0]:
            def verify(hit=hit, key=key):
                if self._plan_content() != hit['content']:
                    plans.pop(key, None)
                    raise _StalePlan()
            if not defer_check:
                try:
                    verify()
                except _StalePlan:
                    hit = None
                verify = None
        else:
            hit = None
        if hit is not None:
            # same udf objects (and parameter objects) as before: planning, negotiation and task
            # creation are pure functions of them -- only the result buffers are per run
            if defer_check and hasattr(executor, 'launch_ahead'):
                # launch first (the recorded launches of the plan's tasks), book-keep behind the kernel
                executor.launch_ahead(hit['tasks'], result_where)
            plans.move_to_end(key)
            meta = copy.copy(hit['meta'])
            for udf in self._udfs:
                udf.set_meta(meta)
                udf.init_result_buffers()
                udf.allocate_for_full(dataset, roi)
                if hasattr(udf, 'preprocess'):
                    udf.set_views_for_dataset(dataset)
                    udf.preprocess()
            return hit['tasks'], hit['params'], verify
        content = self._plan_content() if plans is not None else None
        if content is not None and is_opaque(content):
            plans = None            # a parameter reaches something a fingerprint cannot look into: never re-used
        tasks, params, meta = self._plan_run(dataset, executor, roi, corrections, backends, dry)
        if plans is not None:
            for t in tasks:
                t._keep = {}
            # The udf objects are only weakly referenced (they carry the results of their last
            # run, which must be free to go when the caller drops them); a dead reference never
            # matches, so a recycled id() cannot produce a false hit.  The parameter objects are
            # pinned: their id()s are part of the key.
            plans[key] = dict(udfs=[weakref.ref(u) for u in self._udfs], executor=executor,
                              tasks=tasks, params=params, meta=meta, content=content, epoch=_PLAN_EPOCH[0],
                              kwargs=[dict(u._kwargs) for u in self._udfs])
            while len(plans) > self.PLAN_CACHE_SIZE:
                plans.popitem(last=False)
        return tasks, params, None

    def _plan_run(self, dataset, executor, roi, corrections, backends, dry):
        self._check_preconditions(dataset, roi)
        backends = _canonical_backends(backends)
        device_class = executor.device_class
        chosen = _execution_plan(self._udfs, dataset.array_backends, device_class,
                                 restrict=backends)
        dtype = _get_dtype(self._udfs, dataset.dtype, corrections)
        meta = UDFMeta(
            partition_slice=None, dataset_shape=dataset.shape, roi=roi,
            dataset_dtype=dataset.dtype, input_dtype=dtype, corrections=corrections,
            array_backend=chosen, device_class=device_class,
            gpu_id=getattr(executor, 'gpu_id', None),
        )
        for udf in self._udfs:
            udf.set_meta(meta)
            method = udf.get_method()
            # a `get_method` of the UDF's own may name what it likes: it has to be a UDFMethod whose process_*
            # exists (udf/base.py:2515-2526)
            if not isinstance(method, UDFMethod):
                raise UDFException('UDF.get_method() returned unrecognized value')
            need = {UDFMethod.TILE: 'process_tile', UDFMethod.FRAME: 'process_frame',
                    UDFMethod.PARTITION: 'process_partition'}[method]
            if not callable(getattr(udf, need, None)):
                raise UDFException(f'UDF declared method {method.value} but does not implement {need}.')
            udf.init_result_buffers()
            udf.allocate_for_full(dataset, roi)
            if hasattr(udf, 'preprocess'):
                udf.set_views_for_dataset(dataset)
                udf.preprocess()
        partition = next(iter(dataset.get_partitions()))
        tiling_scheme = Negotiator().get_scheme(
            udfs=self._udfs, approx_partition_shape=partition.shape, dataset=dataset,
            read_dtype=dtype, roi=roi, corrections=corrections, backend=chosen)
        meta._tiling_scheme = tiling_scheme
        params = UDFParams.from_udfs(udfs=self._udfs, roi=roi, corrections=corrections,
                                     tiling_scheme=tiling_scheme, backends=backends)
        tasks = [] if dry else list(self._make_udf_tasks(dataset, roi, backends))
        return tasks, params, meta

    def _roi_for_partition(self, roi, partition):
        return roi.reshape(-1)[partition.slice.get(nav_only=True)]

    def _make_udf_tasks(self, dataset, roi, backends):
        for idx, partition in enumerate(dataset.get_partitions()):
            if roi is not None:
                roi_for_part = self._roi_for_partition(roi, partition)
                if np.count_nonzero(roi_for_part) == 0:
                    continue                      # udf/base.py:2780-2783
            udf_classes = [udf.__class__ for udf in self._udfs]
            udf_backends = [udf.get_backends() for udf in self._udfs]
            yield UDFTask(partition=partition, idx=idx, udf_classes=udf_classes,
                          udf_backends=udf_backends)

    def run_for_dataset(self, dataset, executor, roi=None, progress=False, corrections=None,
                        backends=None, dry=False, result_where=None):
        for res in self.run_for_dataset_sync(
                dataset=dataset, executor=executor, roi=roi, progress=progress,
                corrections=corrections, backends=backends, dry=dry, iterate=False,
                result_where=result_where):
            pass
        return UDFResults(buffers=res.buffers, damage=res.damage)

    def run_for_dataset_sync(self, dataset, executor, roi=None, progress=False, corrections=None,
                             backends=None, dry=False, iterate=True, result_where=None):
        if roi is not None:
            # a private copy: the result buffers keep the roi, and the caller may reuse its array for the next run
            # (reference tests/udf/test_simple_udf.py test_copy_roi)
            roi = np.array(roi, dtype=bool, copy=True)
        # a cache hit by identity may compare the parameter CONTENTS behind the enqueued kernels
        # (executor hook); a partial-result iteration publishes early, so it compares up front
        defer = (not iterate) and hasattr(executor, 'set_before_wait')
        while True:
            try:
                yield from self._run_attempt(dataset, executor, roi, corrections, backends, dry,
                                             iterate, defer, result_where)
                return
            except _StalePlan:
                # the parameters changed in place since the plan was made: the plan is gone from the
                # cache, nothing of the run was delivered -- plan afresh and run again
                defer = False

    def _run_attempt(self, dataset, executor, roi, corrections, backends, dry, iterate, defer,
                     result_where=None):
        try:
            tasks, params, verify = self._prepare_run_for_dataset(
                dataset, executor, roi, corrections, backends, dry, defer_check=defer,
                result_where=result_where)
        except BaseException:
            # (launch_ahead may have enqueued kernels and set the executor's replay state before the rest of
            # the preparation failed: nothing of it may reach the next run)
            if hasattr(executor, 'drain'):
                executor.drain()
            raise
        cancel_id = f"run-{next(_RUN_IDS)}"
        damage = BufferWrapper(kind='nav', dtype=bool)
        damage.set_roi(roi)
        damage.set_shape_ds(dataset.shape, roi)
        damage.allocate()
        checked = []

        def late_check():
            checked.append(True)
            verify()
        prebuilt = []

        def prebuild():
            # result objects of the run, built while its kernels are still running (the executor calls this
            # once the final arrays are attached): only for UDFs whose get_results() hands the declared
            # buffers on without reading them
            res = self._make_udf_result(self._udfs, damage)
            res.buffers
            prebuilt.append(res)
        try:
            if tasks:
                params_handle = executor.scatter(params)
                try:
                    if not iterate and hasattr(executor, 'set_before_final') and all(
                            type(u).get_results is UDF.get_results for u in self._udfs):
                        executor.set_before_final(prebuild)
                    if verify is not None:
                        executor.set_before_wait(late_check)
                    # hook for executors that merge on the device / across ranks
                    result_iter = executor.run_tasks(tasks, params_handle, cancel_id)
                    if iterate and hasattr(executor, 'merge_results_iter'):
                        for _ in executor.merge_results_iter(self._udfs, damage, result_iter,
                                                             self._apply_part_result):
                            yield self._make_udf_result(self._udfs, damage)
                    elif hasattr(executor, 'merge_results'):
                        if result_where is not None:
                            executor.merge_results(self._udfs, damage, result_iter,
                                                   self._apply_part_result, result_where=result_where)
                        else:
                            executor.merge_results(self._udfs, damage, result_iter,
                                                   self._apply_part_result)
                        if iterate:
                            yield self._make_udf_result(self._udfs, damage)
                    else:
                        for part_results, task in result_iter:
                            with UDFRunner._lock:
                                self._apply_part_result(self._udfs, damage, part_results, task)
                            if iterate:
                                yield self._make_udf_result(self._udfs, damage)
                finally:
                    if verify is not None:
                        executor.set_before_wait(None)
                    executor.scatter_release(params_handle)
            else:
                if iterate:
                    yield self._make_udf_result(self._udfs, damage)
            if verify is not None and not checked:
                late_check()
        except JobCancelledError:
            # (the message of the reference, udf/base.py:2720-2721: partitions whose results were merged)
            done = 0
            try:
                done = sum(1 for t in tasks if np.all(damage.get_view_for_partition(t.partition)))
            except Exception:                               # noqa: BLE001  (cancelled before any buffer existed)
                pass
            raise UDFRunCancelled(f"UDF run cancelled after {done} partitions")
        except _StalePlan:
            raise
        except _ReplayMismatch:
            # the launch that was enqueued ahead is not the one this run makes: wait for it, forget the
            # recorded launches and the plan, run again from scratch (fresh result buffers)
            if hasattr(executor, 'drain'):
                executor.drain()
            for t in tasks:
                if getattr(t, '_keep', None) is not None:
                    t._keep['replay'] = None
                    t._keep.pop('recorded', None)
            plans = getattr(dataset, '__dict__', {}).get('_udf_plans')
            if plans:
                plans.clear()
            raise _StalePlan()
        except BaseException as exc:
            # a stale plan may fail before the comparison is reached (a factory list that grew: the
            # kept task instances and the new result buffers disagree) -- that is a stale plan, not
            # an error of the run.  (BaseException: a KeyboardInterrupt / GeneratorExit must not leave
            # enqueued launches or recording state behind either.)
            from libertem_amd import hip as _hip
            if not isinstance(exc, Exception):
                if hasattr(executor, 'drain'):
                    executor.drain()
                raise
            if _hip.LaunchReplay.expected is not None or _hip.LaunchReplay.recording is not None:
                # a run that enqueued launches ahead (or was recording) ended in an error: nothing of that
                # state may leak into the next run
                if hasattr(executor, 'drain'):
                    executor.drain()
                _hip.LaunchReplay.expected = None
                _hip.LaunchReplay.recording = None
            if verify is not None and not checked:
                if hasattr(executor, 'drain'):
                    executor.drain()
                late_check()
            raise
        if not iterate:
            yield prebuilt[0] if prebuilt else self._make_udf_result(self._udfs, damage)

    @classmethod
    def dry_run(cls, udfs, dataset, roi=None):
        from libertem_amd.executor.inline import InlineJobExecutor
        executor = InlineJobExecutor()
        runner = cls(udfs)
        return runner.run_for_dataset(dataset, executor, roi=roi, dry=True)
