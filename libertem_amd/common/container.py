"""
MaskContainer: lazily evaluated, cached, sig-sliced mask stacks.

Same role and constructor surface as the reference's libertem.common.container.MaskContainer
(common/container.py:97-339): evaluates the factories once per worker, decides dense vs sparse
(:150-177, :245-258, :295-301), and serves the per-sig-slice matrix `(px_in_slice, n_masks)`
cached per slice (:74-94).

What is different here: besides the host matrix (`get_for_sig_slice`) the container serves a
*device handle* (`get_handle_for_sig_slice`) -- the libltmi image of that slice, cast to the result
dtype, resident in HBM for the lifetime of the worker's task.
"""
import logging

import os
import numpy as np
import scipy.sparse as sp
import cloudpickle

from libertem_amd.common.sparse import SparseStack, is_sparse, to_dense, to_sparse_stack
from libertem_amd.common.slice import Slice

log = logging.getLogger(__name__)

_SPARSE_NAMES = ('scipy.sparse', 'scipy.sparse.csc', 'scipy.sparse.csr', 'sparse.pydata',
                 'sparse.pydata.GCXS')


DENSIFY_FILL = float(os.environ.get('LTMI_DENSIFY_FILL', '0.125'))   # 0 disables
DENSIFY_MAX_BYTES = 2 << 30


def _worth_densifying(csr_px_by_masks, result_dtype, frame_dtype=None):
    """HIP backend: multiply a sparse stack as a dense one when that is the faster kernel:
    * its fill, counted in the 16-column groups the dense kernel works in, exceeds DENSIFY_FILL (the
      blocked sparse kernel spends ~3.5 multiply-adds per stored value at less than half the dense
      kernel's matrix-pipe efficiency), or
    * it has few columns and touches most pixels, so that skipping untouched pixel chunks (which only the
      sparse kernels do) would save little -- wide rings: 16 384 frames of 256 x 256 against 24 / 48 / 64 / 96 / 128
      ring bins take 3.0 / 2.3 / 2.1 / 1.9 / 1.7 ms on the gather kernel (the blocked image pads such stacks > 8 x
      and is not built) and 0.33 / 0.40 / 0.55 / 0.95 / 1.15 ms dense on uint16 frames, 0.65 / 0.77 / 0.96 / 1.58 /
      1.76 ms on float32 frames (there k_scatter's 1.12 ms wins from 96 columns on; scripts/bench_wide_rings.py):
      up to 128 real columns for 1- / 2-byte frames, up to 64 otherwise.  The dense kernels also keep two
      accumulation levels (a constant frame under a wide ring: 1e-4 on the gather kernel's single float32 chain)."""
    n_px, n_masks = csr_px_by_masks.shape
    nc = 2 if np.dtype(result_dtype).kind == 'c' else 1
    cols16 = -(-n_masks * nc // 16) * 16
    if DENSIFY_FILL <= 0 or n_px * n_masks * np.dtype(result_dtype).itemsize > DENSIFY_MAX_BYTES:
        return False
    if np.dtype(result_dtype) == np.float64 and cols16 > 16:
        # the float64 matrix kernel reads the frames once per 16-column group
        return False
    if csr_px_by_masks.nnz * nc > DENSIFY_FILL * n_px * cols16:
        return True
    touched = np.count_nonzero(np.diff(csr_px_by_masks.indptr))
    narrow_frames = frame_dtype is not None and np.dtype(frame_dtype).kind in 'iub' and \
        np.dtype(frame_dtype).itemsize <= 2
    return cols16 <= (128 if narrow_frames else 64) and touched > 0.5 * n_px


def _maybe_banded(csr_px_by_masks, result_dtype):
    """Cheap look at a sparse stack that `_worth_densifying` would multiply as a dense one: do its masks fall into
    two or more groups of at least four masks with ONE pixel support each (the orders of a bin in a radial-Fourier
    stack with several bins)?  The library then builds a dense image per group over that group's pixels only
    (ltmi_masks_set_sig_shape, include/ltmi.h), which beats one dense pass per 32 complex masks over all pixels
    (4096 frames of 1024 x 1024 float32, 25 orders: 2 / 3 / 4 / 8 bins 7.4 / 11.1 / 15.2 / 27.5 ms dense,
    3.9 / 4.1 / 4.3 / 4.8 ms banded; C5S_BINS=2|3|4|8 C5S_FRAMES=4096 scripts/bench_second_runs.py c5s)."""
    n_px, n_masks = csr_px_by_masks.shape
    nc = 2 if np.dtype(result_dtype).kind == 'c' else 1
    if n_masks * nc <= 64 or np.dtype(result_dtype) not in (np.dtype(np.float32), np.dtype(np.complex64)):
        return False
    csc = csr_px_by_masks.tocsc()
    csc.sort_indices()
    groups = {}
    for k in range(n_masks):
        idx = csc.indices[csc.indptr[k]:csc.indptr[k + 1]]
        key = (len(idx), hash(idx.tobytes()))
        groups[key] = groups.get(key, 0) + 1
    return len(groups) >= 2 and min(groups.values()) >= 4


def _sparse_int_exact(csr_px_by_masks, tile_dtypes):
    """Integer stack x integer frames through the float64 gather kernel: exact iff every possible
    partial sum stays below 2^52 -- bits of the widest tile dtype + bits of the largest column sum of
    |mask values| (csrc/ltmi_sparse.hip csr_int_exact, the same rule)."""
    bits = 0
    for dt in tile_dtypes:
        dt = np.dtype(dt)
        if dt.kind == 'b':
            b = 1
        elif dt.kind in 'iu' and dt.itemsize <= 4:
            b = 8 * dt.itemsize
        else:
            return False
        bits = max(bits, b)
    if csr_px_by_masks.nnz == 0:
        return True
    from libertem_amd.hip import signed_representative
    mags = sp.csr_matrix((np.abs(signed_representative(csr_px_by_masks.data)), csr_px_by_masks.indices,
                          csr_px_by_masks.indptr), shape=csr_px_by_masks.shape)
    worst = int(np.max(mags.sum(axis=0)))
    return bits + worst.bit_length() <= 52


class MaskContainer:
    def __init__(self, mask_factories, dtype=None, use_sparse=None, count=None, backend=None,
                 default_sparse='scipy.sparse'):
        self.mask_factories = mask_factories
        self._length = count
        self._dtype = dtype
        self._computed_masks = None
        self.backend = backend or 'numpy'
        self._default_sparse = default_sparse
        self._host_cache = {}
        self._handle_cache = {}
        if use_sparse is True:
            self._use_sparse = default_sparse
        elif use_sparse is False:
            self._use_sparse = False
        elif isinstance(use_sparse, str) and (
                use_sparse.lower().startswith('scipy.sparse')
                or use_sparse.lower().startswith('sparse.pydata')):
            self._use_sparse = use_sparse
        elif use_sparse is None:
            self._use_sparse = None          # resolved once the masks exist
        else:
            raise ValueError(f'use_sparse not an allowed value: {use_sparse}')
        self.validate_mask_functions()

    def __getstate__(self):
        state = dict(self.__dict__)
        state['_host_cache'] = {}
        state['_handle_cache'] = {}
        return state

    def validate_mask_functions(self):
        fns = self.mask_factories
        limit = 2**20
        if callable(fns):
            fns = [fns]
        for fn in fns:
            try:
                s = len(cloudpickle.dumps(fn))
            except Exception:
                # tasks never leave the process here (no pickling on the HIP executor): a factory
                # that closes over something unpicklable only loses the size warning
                continue
            if s > limit:
                log.warning('Mask factory size %s larger than warning limit %s, may be inefficient'
                            % (s, limit))

    def __len__(self):
        if self._length is not None:
            return self._length
        if not callable(self.mask_factories):
            return len(self.mask_factories)
        return len(self.computed_masks)

    @property
    def dtype(self):
        if self._dtype is None:
            return self.computed_masks.dtype
        return np.dtype(self._dtype)

    @property
    def use_sparse(self):
        if self._use_sparse is None:
            self._use_sparse = self._default_sparse if is_sparse(self.computed_masks) else False
        return self._use_sparse

    @property
    def computed_masks(self):
        if self._computed_masks is None:
            self._computed_masks = self._compute_masks()
        return self._computed_masks

    def _compute_masks(self):
        """Call the factories and stack the results (common/container.py:260-314)."""
        pieces = []
        if callable(self.mask_factories):
            raw = self.mask_factories()
            if isinstance(raw, (list, tuple)):
                # the reference's sparse.concatenate / np.concatenate accept a list of masks
                raw = [r if is_sparse(r) else np.asarray(r) for r in raw]
                if all(not is_sparse(r) for r in raw):
                    raw = np.stack(raw)
                else:
                    raw = SparseStack.concatenate([
                        SparseStack.from_scipy(r) if sp.issparse(r) else
                        (r if isinstance(r, SparseStack) else to_sparse_stack(r[np.newaxis]))
                        for r in raw])
            pieces.append(raw)
        else:
            for f in self.mask_factories:
                m = f()
                if sp.issparse(m):
                    m = SparseStack.from_scipy(m)            # one mask
                elif isinstance(m, SparseStack):
                    pass
                else:
                    m = np.asarray(m)
                    m = m.reshape((1,) + m.shape)
                pieces.append(m)
        masks_are_sparse = all(is_sparse(m) for m in pieces)
        use_sparse = self._use_sparse
        if use_sparse is None:
            use_sparse = self._default_sparse if masks_are_sparse else False
        if use_sparse is not False:
            return SparseStack.concatenate([to_sparse_stack(m) for m in pieces])
        return np.concatenate([to_dense(m) for m in pieces])

    # --- host matrices ---------------------------------------------------------------------------
    def get_for_sig_slice(self, sig_slice, dtype=None, sparse_backend=None, transpose=True,
                          backend=None):
        """(px_in_slice, n_masks) [transpose=True] host matrix of the slice: dense ndarray or
        scipy CSR/CSC (common/container.py:74-94, :213-217)."""
        if dtype is None:
            dtype = self.dtype
        if sparse_backend is None:
            sparse_backend = self.use_sparse
        key = (sig_slice, np.dtype(dtype).str, sparse_backend, transpose)
        if key not in self._host_cache:
            self._host_cache[key] = self._build_host(sig_slice, dtype, sparse_backend, transpose)
        return self._host_cache[key]

    def get(self, key, dtype=None, sparse_backend=None, transpose=True, backend=None):
        if not isinstance(key, Slice):
            raise TypeError("MaskContainer.get() can only be called with "
                            "DataTile/Slice/Partition instances")
        return self.get_for_sig_slice(key.discard_nav(), dtype, sparse_backend, transpose)

    def get_masks_for_slice(self, slice_, dtype=None, sparse_backend=None, transpose=True, backend='numpy'):
        """host matrix of a sig-only slice (common/container.py:316-333; `backend`: NumPy arrays only here)"""
        return self.get_for_sig_slice(slice_, dtype, sparse_backend, transpose, backend)

    def get_for_idx(self, scheme, idx, *args, **kwargs):
        return self.get_for_sig_slice(scheme[idx], *args, **kwargs)

    def _build_host(self, sig_slice, dtype, sparse_backend, transpose):
        masks = self.computed_masks
        if sparse_backend is False:
            m = sig_slice.get(to_dense(masks), sig_only=True)
            m = m.reshape((m.shape[0], -1))
            if transpose:
                m = m.T
            return m.astype(dtype)
        stack = to_sparse_stack(masks)
        fmt = 'csc' if 'csc' in sparse_backend else 'csr'
        mat = stack.to_px_by_masks(sig_slice=sig_slice, dtype=dtype, fmt=fmt)
        if not transpose:
            mat = mat.T
        return mat

    # --- device handles --------------------------------------------------------------------------
    def get_handle_for_sig_slice(self, sig_slice, result_dtype, device, real_frames=True,
                                 tile_dtypes=(), frame_dtype=None, need_dense=False):
        """libltmi handle of the slice's stack, cast to `result_dtype`, on GPU `device`.
        real_frames: the tiles are real numbers (a complex128 sparse stack may then stay sparse).
        tile_dtypes: the dtypes the tiles can arrive in (an integer sparse stack stays sparse if the
        product with them is exact in float64).
        need_dense: the caller multiplies shifted copies of the stack (ltmi_apply_masks_shifted*), which
        only handles of ltmi_masks_create_dense serve: no banded CSR image for a dense stack.

        Non-finite pixels: whichever kernel the handle ends up on, a stack that the reference multiplies sparse
        (use_sparse is not False: stored entries only, common/numba/__init__.py:153-184) and one it multiplies
        dense (`flat_tile @ masks`, udf/masks.py:76-77: 0 * NaN = NaN) keep their own arithmetic -- a
        densified sparse stack carries its gather image (`set_sparse_origin`), a dense stack held as banded
        CSR is marked (`set_dense_origin`); csrc/ltmi_guard.hip."""
        from libertem_amd import hip
        tile_dtypes = tuple(sorted({np.dtype(d).str for d in tile_dtypes}))
        key = (sig_slice, np.dtype(result_dtype).str, int(device), bool(real_frames), tile_dtypes,
               None if frame_dtype is None else np.dtype(frame_dtype).str, bool(need_dense))
        # integer frames hold no NaN / Inf: nothing to attach for them
        float_frames = frame_dtype is None or np.dtype(frame_dtype).kind in 'fc'
        h = self._handle_cache.get(key)
        if h is None:
            sparse_ok = np.dtype(result_dtype) in (np.dtype(np.float32), np.dtype(np.complex64),
                                                   np.dtype(np.float64))
            if self.use_sparse is not False and np.dtype(result_dtype) == np.complex128 \
                    and real_frames:
                # complex128 stack on real frames: the float64 gather kernel on (re, im) column pairs
                m = sp.csr_matrix(self.get_for_sig_slice(
                    sig_slice, dtype=result_dtype, sparse_backend='scipy.sparse.csr',
                    transpose=True))
                if _worth_densifying(m, result_dtype):
                    dense = np.ascontiguousarray(m.T.toarray().astype(result_dtype, copy=False))
                    h = hip.MaskHandle.dense(device, dense, result_dtype)
                    if float_frames:
                        h.set_sparse_origin(hip.MaskHandle.csr_complex128(device, m, gather_only=True))
                else:
                    h = hip.MaskHandle.csr_complex128(device, m)
            elif self.use_sparse is not False and np.dtype(result_dtype).kind in 'iu' and tile_dtypes:
                # integer stack, integer frames (reference: SciPy integer matmul, wrap-around): the
                # float64 gather kernel + truncation where that is exact, else the dense integer kernels
                m = sp.csr_matrix(self.get_for_sig_slice(
                    sig_slice, dtype=result_dtype, sparse_backend='scipy.sparse.csr',
                    transpose=True))
                if _sparse_int_exact(m, tile_dtypes) and not _worth_densifying(m, result_dtype):
                    h = hip.MaskHandle.csr(device, m, result_dtype)
                else:
                    dense = np.ascontiguousarray(m.T.toarray().astype(result_dtype, copy=False))
                    h = hip.MaskHandle.dense(device, dense, result_dtype)
            elif self.use_sparse is False or not sparse_ok:
                # dense stack; also the route for sparse stacks whose result dtype the sparse kernels
                # do not cover (complex128 on complex frames, integers on wide tiles): densified
                m = self.get_for_sig_slice(sig_slice, dtype=result_dtype, sparse_backend=False,
                                           transpose=False)            # (n_masks, px), C order
                h = None
                if not need_dense:
                    h = self._banded_handle_of_dense(m, sig_slice, result_dtype, device,
                                                     dense_semantics=self.use_sparse is False and float_frames)
                if h is None:
                    h = hip.MaskHandle.dense(device, np.ascontiguousarray(m), result_dtype)
            else:
                m = sp.csr_matrix(self.get_for_sig_slice(
                    sig_slice, dtype=result_dtype, sparse_backend='scipy.sparse.csr',
                    transpose=True))                                     # (px, n_masks)
                sig2 = tuple(int(n) for n in sig_slice.shape.sig)
                if _worth_densifying(m, result_dtype, frame_dtype):
                    # a "sparse" stack that is mostly filled (e.g. the radial Fourier orders of one
                    # wide ring): the dense matrix-core kernel multiplies fewer zeros than the
                    # blocked sparse image pads, and streams the stack instead of gathering --
                    # unless the library finds column blocks with a support each (a few wide bins)
                    h = None
                    if len(sig2) == 2 and m.shape[0] == sig2[0] * sig2[1] and _maybe_banded(m, result_dtype):
                        h = hip.MaskHandle.csr(device, m, result_dtype)
                        h.set_sig_shape(sig2[0], sig2[1])
                        if h.kind() != 3:
                            h.close()
                            h = None
                    if h is None:
                        dense = np.ascontiguousarray(m.T.toarray().astype(result_dtype, copy=False))
                        h = hip.MaskHandle.dense(device, dense, result_dtype)
                        if float_frames:
                            # the reference multiplies this stack entry by stored entry: frames whose results
                            # come out non-finite are computed again on the gather image
                            h.set_sparse_origin(hip.MaskHandle.csr(device, m, result_dtype, gather_only=True))
                else:
                    h = hip.MaskHandle.csr(device, m, result_dtype)
            # the detector shape of the slice: a dense float32 / complex64 stack that is even / odd under a
            # mirror of the detector rows (radial Fourier, rings, centre-of-mass ramps) is then multiplied
            # folded -- half the pixels on the matrix cores (include/ltmi.h: ltmi_masks_set_sig_shape)
            sig = tuple(int(n) for n in sig_slice.shape.sig)
            if len(sig) == 2 and h.n_px == sig[0] * sig[1] and np.dtype(result_dtype) in (
                    np.dtype(np.float32), np.dtype(np.complex64)):
                h.set_sig_shape(sig[0], sig[1])
            self._handle_cache[key] = h
        return h

    def _banded_handle_of_dense(self, m, sig_slice, result_dtype, device, dense_semantics=True):
        """A DENSE stack of more than 64 real columns that is mostly zeros in blocks -- radial Fourier with 2 - 9 wide
        bins, which the reference's heuristic declares dense (analysis/radialfourier.py:334-341) -- costs one pass over
        all pixels per 32 complex masks; as CSR the library gives it one dense image per bin over that bin's pixels
        (ltmi_masks_set_sig_shape, kind 3; 4096 frames of 1024 x 1024 float32, 4 / 8 bins: 15.2 / 27.5 -> 4.3 / 4.6 ms).
        Returns that handle, or None (the dense kernels then).  Finite frames give the same sums; with
        `dense_semantics` the handle is told that the zeros it does not store are weights (ltmi_masks_set_dense_origin):
        a non-finite pixel then reaches every mask, like in `flat_tile @ masks`."""
        from libertem_amd import hip
        rd = np.dtype(result_dtype)
        sig = tuple(int(n) for n in sig_slice.shape.sig)
        nc = 2 if rd.kind == 'c' else 1
        if rd not in (np.dtype(np.float32), np.dtype(np.complex64)) or len(sig) != 2 or \
                m.shape[0] * nc <= 64 or m.shape[1] != sig[0] * sig[1] or \
                os.environ.get('LTMI_SPARSE_BAND', '') == '0':
            return None
        if np.count_nonzero(m) > 0.6 * m.size:
            return None
        csr = sp.csr_matrix(np.ascontiguousarray(m).T)                   # (px, n_masks)
        if not _maybe_banded(csr, rd):
            return None
        if dense_semantics and m.shape[0] >= 15 * 1024:
            return None
        h = hip.MaskHandle.csr(device, csr, rd)
        h.set_sig_shape(sig[0], sig[1])
        if h.kind() != 3:
            h.close()
            return None
        if dense_semantics:
            h.set_dense_origin(csr)
        return h

    def get_handle_for_complex_frames(self, sig_slice, result_dtype, device):
        """Complex frames (complex64 / complex128 datasets) on the real matrix kernels:
        sum_p (xr + i xi)(mr + i mi) = [xr, xi] . [mr, -mi] + i [xr, xi] . [mi, mr], i.e. the frame
        read as 2 n_px real pixels against a REAL stack of 2 n_masks rows over 2 n_px pixels; the
        result row (re_0, im_0, re_1, ...) is the interleaved complex row.  Returns a dense handle
        with n_px' = 2 n_px and n_masks' = 2 n_masks in the real dtype of `result_dtype`."""
        from libertem_amd import hip
        rd = np.dtype(result_dtype)
        if rd.kind != 'c':
            raise ValueError("complex frames have a complex result dtype")
        real = np.dtype(np.float32 if rd == np.complex64 else np.float64)
        key = ('complex-frames', sig_slice, rd.str, int(device))
        h = self._handle_cache.get(key)
        if h is None:
            m = np.asarray(self.get_for_sig_slice(sig_slice, dtype=rd, sparse_backend=False,
                                                  transpose=False))        # (n_masks, px) complex
            n_masks, n_px = m.shape
            stack = np.empty((2 * n_masks, 2 * n_px), dtype=real)
            stack[0::2, 0::2] = m.real
            stack[0::2, 1::2] = -m.imag
            stack[1::2, 0::2] = m.imag
            stack[1::2, 1::2] = m.real
            h = hip.MaskHandle.dense(device, stack, real)
            if self.use_sparse is not False:
                # a SPARSE stack on complex frames: the reference's loops multiply stored entries only, and a complex
                # product lets a NaN in EITHER part of a pixel reach both parts of the sum (common/numba/__init__.py:
                # 153-184 with a complex `left_dense`) -- the gather image of the expansion holds all four real entries
                # of every stored complex entry, zeros included, and serves the frames with non-finite results
                coo = sp.coo_matrix(self.get_for_sig_slice(sig_slice, dtype=rd, sparse_backend='scipy.sparse.csr',
                                                           transpose=True))            # (px, n_masks), stored entries
                p, k, v = coo.row.astype(np.int64), coo.col.astype(np.int64), coo.data.astype(rd)
                rows = np.concatenate([2 * p, 2 * p + 1, 2 * p, 2 * p + 1])
                cols = np.concatenate([2 * k, 2 * k, 2 * k + 1, 2 * k + 1])
                vals = np.concatenate([v.real, -v.imag, v.imag, v.real]).astype(real)
                order = np.lexsort((cols, rows))
                indptr = np.zeros(2 * n_px + 1, dtype=np.int64)
                np.add.at(indptr, rows + 1, 1)
                indptr = np.cumsum(indptr)
                expanded = sp.csr_matrix((vals[order], cols[order], indptr), shape=(2 * n_px, 2 * n_masks))
                h.set_sparse_origin(hip.MaskHandle.csr(device, expanded, real, gather_only=True))
            self._handle_cache[key] = h
        return h, real

    def close(self):
        for h in self._handle_cache.values():
            h.close()
        self._handle_cache = {}
