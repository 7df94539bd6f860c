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
from libertem_amd.common.hiparray import HipArray, HipRowsArray
from libertem_amd.common.exceptions import HipRequiredError
from libertem_amd.common.fingerprint import fingerprint, is_opaque
from libertem_amd.udf.base import UDF


# Device mask images survive across tasks and across `run_udf` calls: every task re-instantiates the
# UDF from its kwargs (udf/base.py `new_for_partition`), but the kwargs -- in particular the
# `mask_factories` object -- are shared.  Factories must be pure (the reference re-evaluates them on
# every worker, common/container.py:260-314), so "same factories object + same options" means
# "same stack" and the MaskContainer (host stack + HBM images) can be reused.  The cache pins the
# factories object so its id() stays unique.
_CONTAINER_CACHE = OrderedDict()
_CONTAINER_CACHE_SIZE = 4


def _factories_key(mask_factories):
    """Identity of the factories AND of the list that holds them AND a content fingerprint of every
    array the factories can see (closure cells, defaults, partial arguments, module globals): a list
    that the user mutates in place (append / replace a factory), or a captured array that is modified
    in place between runs, is a different stack (the reference re-evaluates the factories on every
    run, udf/masks.py:331-351, common/container.py:260-314)."""
    return fingerprint(mask_factories)


def _cached_container(mask_factories, dtype, use_sparse, count, default_sparse, cache=True):
    fp = _factories_key(mask_factories) if cache else None
    if fp is None or is_opaque(fp):
        # cache=False, or the factories reach something a fingerprint cannot look into: evaluated afresh,
        # like every run of the reference (common/container.py:260-314)
        return MaskContainer(mask_factories, dtype=dtype, use_sparse=use_sparse, count=count,
                             backend=UDF.BACKEND_HIP, default_sparse=default_sparse)
    key = (fp, None if dtype is None else np.dtype(dtype).str, str(use_sparse), count, default_sparse)
    hit = _CONTAINER_CACHE.get(key)
    if hit is not None and hit[0] is mask_factories:
        _CONTAINER_CACHE.move_to_end(key)
        return hit[1]
    container = MaskContainer(mask_factories, dtype=dtype, use_sparse=use_sparse, count=count,
                              backend=UDF.BACKEND_HIP, default_sparse=default_sparse)
    # pin the factories (and a snapshot of a list's members) so that their id()s stay unique
    pinned = list(mask_factories) if isinstance(mask_factories, (list, tuple)) else None
    _CONTAINER_CACHE[key] = (mask_factories, container, pinned)
    while len(_CONTAINER_CACHE) > _CONTAINER_CACHE_SIZE:
        # evicted containers are NOT closed: a UDF of the current run may still hold them (5+
        # ApplyMasksUDFs in one run_udf); their device images go when the last reference does
        # (MaskHandle.__del__)
        _CONTAINER_CACHE.popitem(last=False)
    return container


#: fold detector corrections into the mask stack of dense ApplyMasksUDF runs (see
#: `fold_corrections_into_masks`); set to False to always correct the frames instead
FOLD_CORRECTIONS = True


def fold_corrections_into_masks(masks, corrections, sig_shape):
    """
    masks: dense (n_masks, *sig) array.  Returns (masks' float64 (n_masks, *sig), const float64
    (n_masks,) or None) such that for every frame x
        sum_p masks[k, p] * corrected(x)[p]  ==  sum_p masks'[k, p] * x[p]  -  const[k]
    where corrected() is CorrectionSet.apply: (x - dark) * gain, then every excluded pixel e := mean
    over its good neighbours env(e).  The repair is linear, so its transpose moves mask weight
    from e to env(e):  masks'[r] = (masks[r] + sum_{e: r in env(e)} masks[e] / |env(e)|) * gain[r],
    masks'[e] = 0  (what the reference offers as detector.correct_dot_masks, :315-338; excluded
    pixels WITHOUT good neighbours stay unpatched here, as in `correct`), and
    const = masks' . dark.
    """
    sig_shape = tuple(int(s) for s in sig_shape)
    n_px = int(np.prod(sig_shape))
    flat = np.asarray(masks, dtype=np.result_type(np.float64, np.asarray(masks).dtype))
    flat = flat.reshape((-1, n_px)).copy()
    desc = corrections.full_frame_descriptor(sig_shape)
    if len(desc.exclude_flat):
        src = flat.copy()
        for e, env, c in zip(desc.exclude_flat, desc.repair_flat, desc.repair_counts):
            if c == 0:
                continue
            flat[:, e] = 0
            share = src[:, e] / c
            for r in env[:c]:
                flat[:, r] += share
    gain = corrections.get_gain_map()
    if gain is not None:
        flat *= np.asarray(gain, dtype=np.float64).reshape(-1)[None, :]
    dark = corrections.get_dark_frame()
    const = None
    if dark is not None:
        const = flat @ np.asarray(dark, dtype=np.float64).reshape(-1)
    return flat.reshape((-1,) + sig_shape), const


def fold_corrections_into_sparse_masks(stack, corrections, sig_shape):
    """
    The same fold for a SPARSE stack (SparseStack / anything `to_sparse_stack` takes):
        masks' = (masks . R) . diag(gain),   const = masks' . dark
    with R the (sparse, n_px x n_px) dead-pixel repair operator of CorrectionSet.apply -- identity
    rows, except row e of an excluded pixel: 1/|env(e)| at its good neighbours.  The folded stack
    stays sparse (an excluded pixel's weight moves to <= 8 neighbours).  Returns (SparseStack of
    float64 / complex128 values, const | None).
    """
    import scipy.sparse as sp
    from libertem_amd.common.sparse import SparseStack, to_sparse_stack
    sig_shape = tuple(int(x) for x in sig_shape)
    n_px = int(np.prod(sig_shape))
    st = to_sparse_stack(stack)
    wide = np.result_type(np.float64, st.data.dtype)
    m = sp.csr_matrix((st.data.astype(wide), (st.mask_idx, st.px_idx)), shape=(st.n_masks, n_px))
    desc = corrections.full_frame_descriptor(sig_shape)
    if len(desc.exclude_flat):
        rows, cols, vals = [], [], []
        keep = np.ones(n_px, dtype=bool)
        for e, env, c in zip(desc.exclude_flat, desc.repair_flat, desc.repair_counts):
            if c == 0:
                continue                     # no good neighbour: stays unpatched, as in `correct`
            keep[e] = False
            rows.extend([int(e)] * int(c))
            cols.extend(int(r) for r in env[:c])
            vals.extend([1.0 / c] * int(c))
        ident = np.flatnonzero(keep)
        r_op = sp.csr_matrix(
            (np.concatenate([np.ones(len(ident)), np.asarray(vals, dtype=np.float64)]),
             (np.concatenate([ident, np.asarray(rows, dtype=np.int64)]),
              np.concatenate([ident, np.asarray(cols, dtype=np.int64)]))), shape=(n_px, n_px))
        m = sp.csr_matrix(m @ r_op)
    gain = corrections.get_gain_map()
    if gain is not None:
        m = sp.csr_matrix(m @ sp.diags(np.asarray(gain, dtype=np.float64).reshape(-1)))
    dark = corrections.get_dark_frame()
    const = None
    if dark is not None:
        const = np.asarray(m @ np.asarray(dark, dtype=np.float64).reshape(-1)).reshape(-1)
    m.eliminate_zeros()
    return SparseStack.from_csr_masks_by_px(m, sig_shape), const


def _folded_plan(corrections, masks_container, mask_factories, sig_shape, count):
    """(container of the folded stack, const | None), cached on the CorrectionSet so that the
    derived factory keeps its identity across tasks and runs (-> `_cached_container` hits)."""
    cache = corrections.__dict__.setdefault('_folded_masks', {})
    key = (id(mask_factories), tuple(sig_shape), np.dtype(masks_container.dtype).str)
    hit = cache.get(key)
    if hit is None or hit[0] is not mask_factories:
        state = {}

        sparse = masks_container.use_sparse is not False

        def folded_factory():
            if 'masks' not in state:
                if sparse:
                    state['masks'], state['const'] = fold_corrections_into_sparse_masks(
                        masks_container.computed_masks, corrections, sig_shape)
                else:
                    state['masks'], state['const'] = fold_corrections_into_masks(
                        np.asarray(masks_container.computed_masks), corrections, sig_shape)
            return state['masks']

        folded_factory()
        if len(cache) > 8:
            cache.clear()
        hit = cache[key] = (mask_factories, folded_factory, state)
    _, factory, state = hit
    # complex stays complex, everything else is carried in the mask dtype of the plain run
    dtype = masks_container.dtype
    if np.dtype(dtype).kind not in 'fc':
        dtype = np.result_type(dtype, np.float32)
    container = _cached_container(
        factory, dtype, masks_container.use_sparse if masks_container.use_sparse is not False
        else False, count, 'scipy.sparse')
    return container, state


def invalidate_cache():
    """forget the evaluated stacks (Context.invalidate_caches): not closed -- a UDF may still hold one"""
    _CONTAINER_CACHE.clear()


def clear_mask_cache():
    while _CONTAINER_CACHE:
        _, entry = _CONTAINER_CACHE.popitem(last=False)
        entry[1].close()


class ApplyMasksEngine:
    """Per-task helper shared by ApplyMasksUDF and CoMUDF (reference udf/masks.py:12-124)."""

    def __init__(self, masks, meta, use_torch=True):
        self.masks = masks
        self.meta = meta
        if meta.array_backend != UDF.BACKEND_HIP:
            raise HipRequiredError(
                "ApplyMasksEngine needs BACKEND_HIP (an MI355X worker); got array backend "
                f"{meta.array_backend!r} on device class {meta.device_class!r}")
        self.result_dtype = np.dtype(np.result_type(meta.input_dtype, masks.dtype))
        self.device = meta.gpu_id if meta.gpu_id is not None else 0
        self.stream_ptr = getattr(meta, 'stream_ptr', None)
        self._const = None           # device tensor (n_masks,): folded dark-frame contribution

    def fold(self, folded_container, plan_state):
        """Switch to the stack with the corrections folded in (the result dtype is unchanged).
        plan_state: dict with 'const' (host float64 | None); the device copy is cached in it."""
        self.masks = folded_container
        const = plan_state['const']
        if const is not None:
            key = ('const_dev', self.device, self.result_dtype.str)
            dev = plan_state.get(key)
            if dev is None:
                import torch
                c = np.ascontiguousarray(np.asarray(const).astype(self.result_dtype))
                if c.dtype.kind == 'c':
                    raise ValueError("folding a dark frame needs a real result dtype")
                dev = plan_state[key] = torch.from_numpy(c).to(f'cuda:{self.device}')
            self._const = dev

    def _get_handle(self, tile_dtype=None, need_dense=False):
        # (tile_dtype: an integer sparse stack stays sparse where the product with tiles of that
        # dtype is exact in float64 -- common/container.py; only asked for integer results)
        ask = tile_dtype is not None and np.dtype(self.result_dtype).kind in 'iu'
        return self.masks.get_handle_for_sig_slice(
            self.meta.sig_slice, self.result_dtype, self.device,
            real_frames=np.dtype(self.meta.input_dtype).kind != 'c',
            tile_dtypes=(tile_dtype,) if ask else (), frame_dtype=self.meta.input_dtype,
            need_dense=need_dense)

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
        if isinstance(tile, HipRowsArray):
            # a region of interest as a row list over the resident frames: the dense float32 kernels
            # read the selected frames in place; anything else gets them gathered
            if out is not None and self._const is None and np.dtype(tile.dtype).kind != 'c':
                handle = self._get_handle(tile.dtype)
                if handle.n_px == n_px and out.shape[0] == n and \
                        prod(out.shape[1:]) == handle.n_masks and \
                        handle.apply_rows(tile.base.data_ptr(), tile.dtype, tile.rows_ptr(), n,
                                          tile.base.ld, out.data_ptr(), out.ld, accumulate,
                                          stream=self.stream_ptr):
                    return out
            tile = tile.materialize(stream=self.stream_ptr)
        if np.dtype(tile.dtype).kind == 'c' and self.result_dtype.kind == 'c' \
                and self._const is None and np.dtype(tile.dtype) == self.result_dtype:
            # complex frames: the frame as 2 n_px real pixels against the real expansion of the stack
            # -- the matrix kernels instead of the generic one (container.get_handle_for_complex_frames).
            # Sparse stacks too: the sparse kernels take real frames only, and the reference multiplies complex
            # frames with them just the same (rmatmul on a complex `left_dense`, common/numba/__init__.py:153-184)
            handle, real = self.masks.get_handle_for_complex_frames(
                self.meta.sig_slice, self.result_dtype, self.device)
            if handle.n_px != 2 * n_px:
                raise ValueError(f"tile has {n_px} px per frame, mask slice has {handle.n_px // 2}")
            n_masks = handle.n_masks // 2
            if out is None:
                out = HipArray.empty((n, n_masks), self.result_dtype, tile.device)
                accumulate = False
            if out.shape[0] != n or prod(out.shape[1:]) != n_masks:
                raise ValueError(f"result view {out.shape} does not fit {n} frames x "
                                 f"{n_masks} masks")
            handle.apply(tile.data_ptr(), real, n, 2 * tile.ld, out.data_ptr(), 2 * out.ld,
                         accumulate, stream=self.stream_ptr)
            return out
        handle = self._get_handle(tile.dtype)
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
        if self._const is not None:
            # dark frame of folded corrections: out[f, k] -= sum_p masks'[k, p] * dark[p]
            from libertem_amd import hip
            hip.add2d(tile.device, out.data_ptr(), out.ld, self._const.data_ptr(), 0,
                      self.result_dtype, n, handle.n_masks, negate=True, stream=self.stream_ptr)
        return out

    def process_tile_shifted(self, tile, shifts, out, accumulate=True):
        """
        Shifted masks for a whole tile of FULL frames (reference: process_frame_shifted, one call
        per frame, udf/masks.py:85-124).  `shifts`: host int array (n, 2) of (dy, dx).
        """
        import torch
        if not isinstance(tile, HipArray):
            raise HipRequiredError("process_tile_shifted expects a device tile (HipArray)")
        if isinstance(tile, HipRowsArray):
            tile = tile.materialize(stream=self.stream_ptr)
        sig = tuple(self.meta.dataset_shape.sig)
        if len(sig) != 2 or tuple(tile.shape[1:]) != sig:
            raise ValueError(
                f"shifted masks need tiles of full 2D frames {sig}, got {tile.shape[1:]} "
                "(do not force a sub-frame tileshape together with shifts=)")
        n = tile.shape[0]
        shifts = np.ascontiguousarray(np.asarray(shifts).reshape((n, 2)).astype(np.int32))
        handle = self._get_handle(need_dense=True)     # (shifted copies are cut out of the dense image)
        handle.apply_shifted_host(tile.data_ptr(), tile.dtype, n, tile.ld, sig[0], sig[1], shifts,
                                  out.data_ptr(), out.ld, accumulate, stream=self.stream_ptr)
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
    cache : (not in the reference) False = evaluate the factories and plan the run afresh every time, like
        the reference does -- for factories whose output depends on something a content fingerprint cannot
        see (a file, random state); default True: stacks, device images and run plans are kept across runs
        and re-used while everything the factories can reach is unchanged (common/fingerprint.py).
    '''

    REUSE_TASK_INSTANCES = True      # (udf/base.py: per-partition instances kept between runs)
    ACCEPTS_ROW_VIEWS = True         # process_tile reads an ROI's frames through a row list (no gather)

    def __init__(self, mask_factories, use_torch=True, use_sparse=None, mask_count=None,
                 mask_dtype=None, preferred_dtype=None, backends=None, shifts=None, cache=True, **kwargs):
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
            backends=backends, shifts=shifts, cache=cache, **kwargs)

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
                                 'scipy.sparse', cache=p.get('cache', True) is not False)

    def folds_corrections(self, corrections, meta):
        """True iff this UDF can take RAW frames and apply `corrections` through its masks."""
        if not FOLD_CORRECTIONS or self.params.get('shifts') is not None:
            return False
        if np.dtype(np.result_type(meta.input_dtype, self.get_mask_dtype())).kind != 'f':
            return False
        return True

    def get_hip_tile_frames(self):
        """Tiling hint for the MI355X policy: sparse stacks with wide result rows (>= 1 KiB per
        frame, e.g. 1024 ring masks = 4 KiB) are run in tiles of 16384 frames so that the D2H of
        finished result rows overlaps the next tile's kernel (io/dataset/base.py)."""
        try:
            count = int(self.get_mask_count())
            sparse = self.masks.use_sparse is not False
        except Exception:
            return None
        return 16384 if sparse and count * 4 >= 1024 else None

    def get_hip_direct_results(self):
        """Tiling hint: result rows small enough to be written straight into the final host buffer
        (common/udf.py HIP_DIRECT_ROW_MAX) -> no pipelined tiles needed (io/dataset/base.py)."""
        from libertem_amd.common import udf as udf_common
        try:
            dtype = np.result_type(self.meta.input_dtype, self.get_mask_dtype())
            return int(self.get_mask_count()) * np.dtype(dtype).itemsize <= \
                udf_common.HIP_DIRECT_ROW_MAX
        except Exception:
            return False

    def get_task_data(self):
        engine = ApplyMasksEngine(self.masks, self.meta, self.params.use_torch)
        if getattr(self.meta, 'corrections_folded', False):
            container, plan_state = _folded_plan(
                self.meta.corrections, self.masks, self.params.mask_factories,
                tuple(self.meta.dataset_shape.sig), self.get_mask_count())
            engine.fold(container, plan_state)
        return {'engine': engine}

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

    def get_write_once_buffers(self):
        """Tiles of whole frames (the MI355X tiling policy): every result row is produced by ONE
        kernel call, `=` instead of `+=` -- no zero fill, and the rows may be written straight
        into the run's final host buffer.  (Not with a folded dark frame: its constant is
        subtracted in a second pass over the rows.)"""
        ts = self.meta.tiling_scheme if self.meta is not None else None
        if ts is None or len(ts) != 1 or getattr(self.meta, 'corrections_folded', False) \
                or getattr(self.meta, 'sig_sliced_tiles', False):
            return ()
        return ('intensity',)

    def process_tile(self, tile):
        shifts = self.params.get('shifts')
        accumulate = not self.results.get_buffer('intensity').write_once
        if shifts is None:
            # fused: results.intensity[:] (+)= tile.reshape(n, -1).astype(input_dtype) @ masks
            self.task_data.engine.process_tile(tile, out=self.results.intensity,
                                               accumulate=accumulate)
            return
        n = tile.shape[0]
        sh = np.asarray(shifts)
        if sh.ndim == 1:                       # constant (y, x) shift
            sh = np.broadcast_to(sh.astype(int), (n, 2))
        self.task_data.engine.process_tile_shifted(tile, sh.astype(int),
                                                   out=self.results.intensity,
                                                   accumulate=accumulate)

