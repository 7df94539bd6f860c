"""
ApplyMasksUDF on MI355X.

Drop-in for the reference's libertem.udf.masks (udf/masks.py:12-404): same constructor, same
result buffer (`intensity`, kind nav, extra_shape (n_masks,), dtype
result_type(input_dtype, mask_dtype)), same `process_tile` / default-`merge` contract.  What
`ApplyMasksEngine.process_flat` does with torch.mm / `flat_tile @ masks` / numba rmatmul in the
reference (:59-77) is ONE libltmi call here: the tile's dtype conversion, the product and the `+=`
into the result view are fused into a HIP kernel (see libertem_amd/csrc/ltmi_dense.hip).

This operator runs on BACKEND_HIP only.  There is no NumPy path in the product.
"""
from collections import OrderedDict

import numpy as np

from libertem_amd.common.math import prod
from libertem_amd.common.udf import UDFMethod
from libertem_amd.common.container import MaskContainer
from libertem_amd.common.buffers import AuxBufferWrapper
from libertem_amd.common.hiparray import HipArray
from libertem_amd.common.exceptions import HipRequiredError
from libertem_amd.udf.base import UDF


# Device mask images survive across tasks and across `run_udf` calls: every task re-instantiates the
# UDF from its kwargs (udf/base.py `new_for_partition`), but the kwargs -- in particular the
# `mask_factories` object -- are shared.  Factories must be pure (the reference re-evaluates them on
# every worker, common/container.py:260-314), so "same factories object + same options" means
# "same stack" and the MaskContainer (host stack + HBM images) can be reused.  The cache pins the
# factories object so its id() stays unique.
_CONTAINER_CACHE = OrderedDict()
_CONTAINER_CACHE_SIZE = 4


def _cached_container(mask_factories, dtype, use_sparse, count, default_sparse):
    key = (id(mask_factories), None if dtype is None else np.dtype(dtype).str, str(use_sparse),
           count, default_sparse)
    hit = _CONTAINER_CACHE.get(key)
    if hit is not None and hit[0] is mask_factories:
        _CONTAINER_CACHE.move_to_end(key)
        return hit[1]
    container = MaskContainer(mask_factories, dtype=dtype, use_sparse=use_sparse, count=count,
                              backend=UDF.BACKEND_HIP, default_sparse=default_sparse)
    _CONTAINER_CACHE[key] = (mask_factories, container)
    while len(_CONTAINER_CACHE) > _CONTAINER_CACHE_SIZE:
        _, (_, old) = _CONTAINER_CACHE.popitem(last=False)
        old.close()
    return container


def clear_mask_cache():
    while _CONTAINER_CACHE:
        _, (_, old) = _CONTAINER_CACHE.popitem(last=False)
        old.close()


class ApplyMasksEngine:
    """Per-task helper shared by ApplyMasksUDF and CoMUDF (reference udf/masks.py:12-124)."""

    def __init__(self, masks, meta, use_torch=True):
        self.masks = masks
        self.meta = meta
        if meta.array_backend != UDF.BACKEND_HIP:
            raise HipRequiredError(
                "ApplyMasksEngine needs BACKEND_HIP (an MI355X worker); got array backend "
                f"{meta.array_backend!r} on device class {meta.device_class!r}")
        self.result_dtype = np.result_type(meta.input_dtype, masks.dtype)
        self.device = meta.gpu_id if meta.gpu_id is not None else 0
        self.stream_ptr = getattr(meta, 'stream_ptr', None)

    def _get_handle(self):
        return self.masks.get_handle_for_sig_slice(self.meta.sig_slice, self.result_dtype,
                                                   self.device)

    def process_tile(self, tile, out=None, accumulate=False):
        """
        tile: HipArray (n, *sig_slice_shape), native dtype.
        out:  HipArray (n, n_masks) of result dtype; allocated if None.
        Returns `out`.
        """
        if not isinstance(tile, HipArray):
            raise HipRequiredError("ApplyMasksEngine.process_tile expects a device tile (HipArray)")
        n = tile.shape[0]
        n_px = prod(tile.shape[1:])
        handle = self._get_handle()
        if handle.n_px != n_px:
            raise ValueError(f"tile has {n_px} px per frame, mask slice has {handle.n_px}")
        if out is None:
            out = HipArray.empty((n, handle.n_masks), self.result_dtype, tile.device)
            accumulate = False
        if out.shape[0] != n or prod(out.shape[1:]) != handle.n_masks:
            raise ValueError(f"result view {out.shape} does not fit {n} frames x "
                             f"{handle.n_masks} masks")
        handle.apply(tile.data_ptr(), tile.dtype, n, tile.ld, out.data_ptr(), out.ld, accumulate,
                     stream=self.stream_ptr)
        return out

    def process_tile_shifted(self, tile, shifts, out, accumulate=True):
        """
        Shifted masks for a whole tile of FULL frames (reference: process_frame_shifted, one call
        per frame, udf/masks.py:85-124).  `shifts`: host int array (n, 2) of (dy, dx).
        """
        import torch
        if not isinstance(tile, HipArray):
            raise HipRequiredError("process_tile_shifted expects a device tile (HipArray)")
        sig = tuple(self.meta.dataset_shape.sig)
        if len(sig) != 2 or tuple(tile.shape[1:]) != sig:
            raise ValueError(
                f"shifted masks need tiles of full 2D frames {sig}, got {tile.shape[1:]} "
                "(do not force a sub-frame tileshape together with shifts=)")
        n = tile.shape[0]
        shifts = np.ascontiguousarray(np.asarray(shifts).reshape((n, 2)).astype(np.int32))
        handle = self._get_handle()
        dev_shifts = torch.from_numpy(shifts).to(f'cuda:{tile.device}', non_blocking=False)
        handle.apply_shifted(tile.data_ptr(), tile.dtype, n, tile.ld, sig[0], sig[1],
                             dev_shifts.data_ptr(), out.data_ptr(), out.ld, accumulate,
                             stream=self.stream_ptr)
        self._keep = dev_shifts          # keep alive until the stream has consumed it
        return out


class ApplyMasksUDF(UDF):
    '''
    Apply masks to signals/frames in the dataset: integrate over regions with binary masks, or
    weighted with float / complex masks.  The result is the buffer `intensity` with shape
    `(*nav_shape, len(masks))`.

    Parameters (identical to the reference, udf/masks.py:127-256)
    ----------
    mask_factories : callable or list of callables returning masks (NumPy arrays, scipy.sparse
        matrices or `libertem_amd.common.sparse.SparseStack`), each of `dataset.shape.sig`.
    use_torch : accepted for compatibility; ignored (the product never uses torch for math).
    use_sparse : None | False | True | 'scipy.sparse' | 'scipy.sparse.csc' | 'sparse.pydata'
        None: sparse iff all factories return sparse masks.  All sparse flavours map to the same
        CSR device kernel.
    mask_count, mask_dtype, preferred_dtype : as in the reference.
    backends : restrict the backends; must contain 'hip' (default).
    shifts : (y, x) tuple for a constant shift of all masks, or
        `ApplyMasksUDF.aux_data(..., kind='nav', extra_shape=(2,))` for per-frame shifts
        (reference udf/masks.py:207-233).  Float values are cast to int.  With shifts the stack is
        always applied densely, one kernel launch per tile of full frames (the reference goes frame
        by frame).
    '''

    def __init__(self, mask_factories, use_torch=True, use_sparse=None, mask_count=None,
                 mask_dtype=None, preferred_dtype=None, backends=None, shifts=None, **kwargs):
        _backends = backends
        supported = (self.BACKEND_HIP,)
        if backends is None:
            backends = supported
        if isinstance(backends, str):
            backends = (backends,)
        backends = tuple(b for b in backends if b in supported)
        if len(backends) == 0:
            raise ValueError(f'No compatible backend found in {_backends}; '
                             f'ApplyMasksUDF runs on {supported} only')
        if shifts is not None:
            if isinstance(use_sparse, str) and use_sparse.startswith('scipy.sparse'):
                raise ValueError(f'Sparse backend {use_sparse} not supported for '
                                 'shifts, use sparse.pydata instead.')
            if not isinstance(shifts, AuxBufferWrapper):
                shifts = np.asarray(shifts)
        self._mask_container = None
        super().__init__(
            mask_factories=mask_factories, use_torch=use_torch, use_sparse=use_sparse,
            mask_count=mask_count, mask_dtype=mask_dtype, preferred_dtype=preferred_dtype,
            backends=backends, shifts=shifts, **kwargs)

    def get_preferred_input_dtype(self):
        if self.params.preferred_dtype is None:
            return super().get_preferred_input_dtype()
        return self.params.preferred_dtype

    def get_mask_dtype(self):
        if self.params.mask_dtype is None:
            return self.masks.dtype
        return self.params.mask_dtype

    def get_mask_count(self):
        if self.params.mask_count is None:
            return len(self.masks)
        return self.params.mask_count

    @property
    def masks(self):
        if self._mask_container is None:
            self._mask_container = self._make_mask_container()
        return self._mask_container

    def _make_mask_container(self):
        p = self.params
        use_sparse = p.use_sparse
        if p.get('shifts') is not None:
            use_sparse = False               # shifted application slices the dense stack
        return _cached_container(p.mask_factories, p.mask_dtype, use_sparse, p.mask_count,
                                 'scipy.sparse')

    def get_task_data(self):
        return {'engine': ApplyMasksEngine(self.masks, self.meta, self.params.use_torch)}

    def get_result_buffers(self):
        dtype = np.result_type(self.meta.input_dtype, self.get_mask_dtype())
        count = self.get_mask_count()
        return {'intensity': self.buffer(kind='nav', extra_shape=(count,), dtype=dtype,
                                         where='device')}

    def get_backends(self):
        return self.params.backends

    def get_method(self):
        return UDFMethod.TILE

    def get_dist_merge(self):
        """nav-kind, default merge: disjoint row ranges per partition."""
        return {'intensity': 'disjoint'}

    def process_tile(self, tile):
        shifts = self.params.get('shifts')
        if shifts is None:
            # fused: results.intensity[:] += tile.reshape(n, -1).astype(input_dtype) @ masks
            self.task_data.engine.process_tile(tile, out=self.results.intensity, accumulate=True)
            return
        n = tile.shape[0]
        sh = np.asarray(shifts)
        if sh.ndim == 1:                       # constant (y, x) shift
            sh = np.broadcast_to(sh.astype(int), (n, 2))
        self.task_data.engine.process_tile_shifted(tile, sh.astype(int),
                                                   out=self.results.intensity, accumulate=True)

