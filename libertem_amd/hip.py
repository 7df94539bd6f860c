"""
ctypes binding of libltmi.so (include/ltmi.h) -- the only way the product reaches the GPU.

There is deliberately NO CPU fallback anywhere in this module: if the shared library is
missing or a call fails, a `RuntimeError` / `ValueError` is raised.
"""
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
#: LTMI_LIB selects another build of the same library (the AddressSanitizer host build of
#: `python -m libertem_amd.build --asan`); there is no other implementation to select
LIB_PATH = os.environ.get('LTMI_LIB') or os.path.join(_HERE, '_lib', 'libltmi.so')

_lib = None
_lock = threading.Lock()

# numpy dtype -> enum ltmi_dtype (include/ltmi.h)
_DTYPES = {
    np.dtype('bool'): 0, np.dtype('uint8'): 1, np.dtype('int8'): 2, np.dtype('uint16'): 3,
    np.dtype('int16'): 4, np.dtype('uint32'): 5, np.dtype('int32'): 6, np.dtype('uint64'): 7,
    np.dtype('int64'): 8, np.dtype('float32'): 9, np.dtype('float64'): 10,
    np.dtype('complex64'): 11, np.dtype('complex128'): 12,
}

EXPORTS = (
    'ltmi_version', 'ltmi_last_error', 'ltmi_device_count', 'ltmi_device_info',
    'ltmi_masks_create_dense', 'ltmi_masks_create_csr', 'ltmi_masks_destroy', 'ltmi_masks_kind', 'ltmi_masks_set_sig_shape',
    'ltmi_masks_set_sparse_origin', 'ltmi_masks_set_dense_origin', 'ltmi_masks_create_csr_gather',
    'ltmi_masks_nonfinite_frames',
    'ltmi_apply_masks', 'ltmi_apply_masks_rows', 'ltmi_apply_masks_shifted', 'ltmi_apply_masks_shifted_host', 'ltmi_sum_frames_workspace', 'ltmi_sum_frames', 'ltmi_sum_sig',
    'ltmi_axpy', 'ltmi_add2d', 'ltmi_gather_rows', 'ltmi_host_device_pointer', 'ltmi_host_copy', 'ltmi_correct', 'ltmi_repair_pixels', 'ltmi_byteswap', 'ltmi_mib_decode', 'ltmi_com_fields', 'ltmi_fft_plan_create',
    'ltmi_fft_plan_destroy', 'ltmi_crystallinity', 'ltmi_crystallinity_corrected', 'ltmi_fft_plan_last_kernel',
    'ltmi_masks_set_tuning',
    'ltmi_masks_last_kernel', 'ltmi_comm_unique_id', 'ltmi_comm_create', 'ltmi_comm_destroy',
    'ltmi_comm_all_gather', 'ltmi_comm_all_reduce_sum', 'ltmi_comm_library_info',
)


class LtmiError(RuntimeError):
    pass


class ReplayMismatch(RuntimeError):
    """a run did not issue the launch that was enqueued ahead for it (LaunchReplay): the caller drops the
    recorded launch, plans afresh and runs again -- nothing of the abandoned run is delivered"""


class ReplayState:
    """the launch-ahead state of ONE executor (`HipJobExecutor.replay`): `expected` = the launches that were
    enqueued ahead for the run in progress (None: none), `recording` = list that collects the launches of the
    task that is running (None: not recording)"""
    __slots__ = ('expected', 'recording')

    def __init__(self):
        self.expected = None
        self.recording = None


class RunInProgressError(RuntimeError):
    """a run was started on an executor whose `run_udf_iter` is suspended between two partial results"""


class RunGate:
    """Serialises the runs of one executor (its delivery targets, launch-ahead state and streams belong to the
    run in progress): re-entrant for the thread that holds it, and -- unlike threading.RLock -- releasable from
    another thread (a generator of partial results that is closed by the garbage collector).

    A `run_udf_iter` holds the gate from its first step to its end, also while it is suspended at a `yield`
    (`suspended` is set then): the executor's per-run state belongs to it.  Another run on the same executor in
    that window -- from the loop body, from the event loop of an `async for`, from any thread -- can neither wait
    (it would wait for ever: the iteration only resumes when its consumer asks for the next part) nor go ahead
    (it would reuse the suspended run's state), so `acquire` raises RunInProgressError instead."""

    def __init__(self):
        self._lock = threading.Lock()
        self._owner = None
        self._depth = 0
        self.suspended = None        # what is suspended (a string for the message) or None

    def _refuse(self):
        raise RunInProgressError(
            f"{self.suspended} of this context is suspended between two partial results and owns the executor: "
            "consume or close() the iterator before starting another run on the same Context "
            "(or run it on a second Context)")

    def acquire(self):
        me = threading.get_ident()
        if self._owner == me:
            if self.suspended:
                self._refuse()
            self._depth += 1
            return
        while not self._lock.acquire(timeout=0.05):
            if self.suspended:
                self._refuse()
        self._owner = me
        self._depth = 1

    def release(self):
        self._depth -= 1
        if self._depth <= 0:
            self._depth = 0
            self._owner = None
            self.suspended = None
            self._lock.release()

    def __enter__(self):
        self.acquire()
        return self

    def __exit__(self, *exc):
        self.release()


class _ReplayMeta(type):
    # `LaunchReplay.expected` / `.recording` are those of the executor whose run is in progress on THIS thread
    # (`LaunchReplay.bound(state)`, entered by Context.run_udf under the executor's gate); outside a run: a state
    # of the thread's own, so that bare handle calls of two threads never see each other's lists
    def _cur(cls):
        st = getattr(cls._tls, 'state', None)
        if st is None:
            st = cls._tls.state = cls._tls.own = ReplayState()
        return st

    @property
    def expected(cls):
        return cls._cur().expected

    @expected.setter
    def expected(cls, v):
        cls._cur().expected = v

    @property
    def recording(cls):
        return cls._cur().recording

    @recording.setter
    def recording(cls, v):
        cls._cur().recording = v


class LaunchReplay(metaclass=_ReplayMeta):
    """
    Launch first, book-keep behind the kernel.  A run of a cached plan (udf/base.py `_plan_for`) whose
    previous run issued exactly ONE `ltmi_apply_masks` launch per task -- same handle, same resident tile,
    result rows written straight into the run's result buffer -- enqueues that launch as soon as the new
    result buffer exists (executor/hip.py `merge_results`), then runs the normal tile loop: the tile loop's
    own call finds itself in `expected` and returns.  A different call raises ReplayMismatch.
    The state (`expected`, `recording`: ReplayState) belongs to the EXECUTOR; a run binds it to its thread
    for its duration, runs of one executor are serialised by the executor's RunGate.
    """
    _tls = threading.local()
    n_ahead = 0          # launches enqueued ahead so far, all executors (tests, bench)

    class bound:
        """context manager: the calling thread's launches are matched against / recorded into `state`"""

        def __init__(self, state):
            self.state = state

        def __enter__(self):
            tls = LaunchReplay._tls
            self.prev = getattr(tls, 'state', None)
            tls.state = self.state
            return self.state

        def __exit__(self, *exc):
            LaunchReplay._tls.state = self.prev

    @staticmethod
    def signature(handle, tile_ptr, tile_dtype, n_frames, ld_tile, out_ptr, ld_out, accumulate, stream):
        return (id(handle), int(tile_ptr), np.dtype(tile_dtype).str, int(n_frames), int(ld_tile), int(out_ptr),
                int(ld_out), bool(accumulate), stream if isinstance(stream, int) else _stream_ptr(stream))


class KernelTimer:
    """
    Optional HIP-event timing of every `ltmi_apply_masks` launch, on the stream the kernel is
    launched on (bench.py's `roofline.achieved`).  Off by default: zero overhead.
    """
    enabled = False
    events = []

    @classmethod
    def start(cls):
        cls.enabled = True
        cls.events = []

    @classmethod
    def stop(cls):
        """-> list of (milliseconds, n_frames, kernel name); synchronises."""
        import torch
        cls.enabled = False
        torch.cuda.synchronize()
        out = [(a.elapsed_time(b), n, k) for a, b, n, k in cls.events]
        cls.events = []
        return out


def dtype_code(dtype):
    try:
        return _DTYPES[np.dtype(dtype)]
    except KeyError:
        raise ValueError(f"dtype {dtype!r} is not supported by libltmi")


def lib():
    """Load libltmi.so once.  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP library has not been built. "
                "Run `python -m libertem_amd.build` (needs hipcc). There is no CPU fallback."
            )
        # torch bundles its own libamdhip64 (soname libamdhip64.so.7, NEEDED by torch under the
        # unversioned name).  It has to be in the process BEFORE libltmi so that libltmi's
        # DT_NEEDED libamdhip64.so.7 binds to the same runtime instance; two HIP runtimes in one
        # process cannot share device pointers (and the second one sees no devices).
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        c = ctypes
        vp, i64, i32 = c.c_void_p, c.c_int64, c.c_int
        L.ltmi_version.restype = i32
        L.ltmi_last_error.restype = c.c_char_p
        L.ltmi_device_count.argtypes = [c.POINTER(i32)]
        L.ltmi_device_info.argtypes = [i32, c.c_char_p, c.POINTER(i32), c.POINTER(i64),
                                       c.POINTER(i32)]
        L.ltmi_masks_create_dense.argtypes = [i32, vp, i32, i64, i64, c.POINTER(vp)]
        L.ltmi_masks_create_csr.argtypes = [i32, vp, vp, vp, i32, i64, i64, c.POINTER(vp)]
        L.ltmi_masks_create_csr_gather.argtypes = [i32, vp, vp, vp, i32, i64, i64, c.POINTER(vp)]
        L.ltmi_masks_destroy.argtypes = [vp]
        L.ltmi_apply_masks_rows.argtypes = [vp, vp, i32, vp, i64, i64, vp, i64, i32, vp,
                                            c.POINTER(i32)]
        L.ltmi_masks_kind.argtypes = [vp, c.POINTER(i32)]
        L.ltmi_masks_set_sig_shape.argtypes = [vp, i32, i32]
        L.ltmi_masks_set_sparse_origin.argtypes = [vp, vp]
        L.ltmi_masks_set_dense_origin.argtypes = [vp, vp, vp]
        L.ltmi_masks_nonfinite_frames.argtypes = [vp, vp, c.POINTER(i64)]
        L.ltmi_apply_masks.argtypes = [vp, vp, i32, i64, i64, vp, i64, i32, vp]
        L.ltmi_apply_masks_shifted.argtypes = [vp, vp, i32, i64, i64, i32, i32, vp, vp, i64, i32, vp]
        L.ltmi_apply_masks_shifted_host.argtypes = [vp, vp, i32, i64, i64, i32, i32, vp, vp, i64, i32, vp]
        L.ltmi_sum_frames_workspace.argtypes = [i64, i64, i32]
        L.ltmi_sum_frames_workspace.restype = i64
        L.ltmi_sum_frames.argtypes = [i32, vp, i32, i64, i64, i64, vp, i32, i32, vp, vp]
        L.ltmi_sum_sig.argtypes = [i32, vp, i32, i64, i64, i64, vp, i32, i32, vp]
        L.ltmi_axpy.argtypes = [i32, vp, vp, i32, i64, vp]
        L.ltmi_add2d.argtypes = [i32, vp, i64, vp, i64, i32, i64, i64, i32, vp]
        L.ltmi_gather_rows.argtypes = [i32, vp, i64, vp, i64, i64, vp, vp]
        L.ltmi_host_device_pointer.argtypes = [i32, vp, c.POINTER(vp)]
        L.ltmi_host_copy.argtypes = [vp, vp, i64, i32]
        L.ltmi_correct.argtypes = [i32, vp, i32, i64, i64, i64, vp, vp, vp, i32, i64, vp]
        L.ltmi_repair_pixels.argtypes = [i32, vp, i32, i64, i64, vp, vp, vp, i32, i32, vp]
        L.ltmi_byteswap.argtypes = [i32, vp, vp, i32, i64, vp]
        L.ltmi_mib_decode.argtypes = [i32, vp, i64, i64, i32, i32, i32, i64, i32, i32, vp, i32, vp]
        L.ltmi_com_fields.argtypes = [i32, vp, i64, i32, i32, ctypes.c_double, ctypes.c_double, vp, vp,
                                      vp, vp, vp, vp, vp]
        L.ltmi_fft_plan_create.argtypes = [i32, i32, i32, i32, c.POINTER(vp)]
        L.ltmi_fft_plan_destroy.argtypes = [vp]
        L.ltmi_crystallinity.argtypes = [vp, vp, i32, i64, i64, vp, vp, i32, i32, i32, vp, i32, vp]
        L.ltmi_crystallinity_corrected.argtypes = [vp, vp, i32, i64, i64, vp, vp, vp, vp, vp, i32, i32,
                                                   vp, vp, i32, i32, i32, vp, i32, vp]
        L.ltmi_masks_set_tuning.argtypes = [vp, i32, i32, i32]
        L.ltmi_masks_last_kernel.argtypes = [vp]
        L.ltmi_comm_unique_id.argtypes = [vp]
        L.ltmi_comm_create.argtypes = [i32, i32, i32, vp, c.POINTER(vp)]
        L.ltmi_comm_destroy.argtypes = [vp]
        L.ltmi_comm_all_gather.argtypes = [vp, vp, vp, i64, vp]
        L.ltmi_comm_all_reduce_sum.argtypes = [vp, vp, i32, i64, vp]
        L.ltmi_comm_library_info.argtypes = [ctypes.c_char_p, i64, ctypes.POINTER(i32), ctypes.POINTER(i32)]
        L.ltmi_masks_last_kernel.restype = c.c_char_p
        L.ltmi_fft_plan_last_kernel.argtypes = [vp]
        L.ltmi_fft_plan_last_kernel.restype = c.c_char_p
        for name in EXPORTS:
            fn = getattr(L, name)
            if fn.restype is c.c_int and name not in ('ltmi_version',):
                fn.restype = i32
        _lib = L
    return _lib


def check(rc, what):
    if rc == 0:
        return
    msg = lib().ltmi_last_error().decode('utf8', 'replace')
    if rc in (-1, -2, -3):
        raise ValueError(f"{what}: {msg} (code {rc})")
    raise LtmiError(f"{what}: {msg} (code {rc})")


def device_count():
    n = ctypes.c_int(0)
    check(lib().ltmi_device_count(ctypes.byref(n)), 'ltmi_device_count')
    return n.value


def device_info(device):
    name = ctypes.create_string_buffer(256)
    cu = ctypes.c_int(0)
    mem = ctypes.c_int64(0)
    arch = ctypes.c_int(0)
    check(lib().ltmi_device_info(device, name, ctypes.byref(cu), ctypes.byref(mem),
                                 ctypes.byref(arch)), 'ltmi_device_info')
    return {'name': name.value.decode(), 'cu_count': cu.value, 'hbm_bytes': mem.value,
            'gfx_arch': hex(arch.value)}


def _stream_ptr(stream):
    """torch.cuda.Stream | int | None -> hipStream_t value"""
    if stream is None:
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if isinstance(stream, int):
        return ctypes.c_void_p(stream)
    return ctypes.c_void_p(stream.cuda_stream)


def signed_representative(values):
    """Integer mask values as int64 for an integer product that is truncated to the width of their
    dtype afterwards: unsigned values are read as the signed numbers of the same width (65530 as uint16
    is -6 modulo 2^16) -- the same result, the smallest magnitudes (the float64 route of the sparse
    integer product is exact while the column sums of |values| stay small)."""
    values = np.asarray(values)
    if values.dtype.kind == 'u':
        values = values.view(np.dtype(f'i{values.dtype.itemsize}'))
    return values.astype(np.int64)


class MaskHandle:
    """Owns one `ltmi_masks*` (device image of a sig-sliced, flattened mask stack)."""

    def __init__(self, ptr, device, n_masks, n_px, result_dtype, sparse):
        self._ptr = ptr
        self.device = device
        self.n_masks = n_masks
        self.n_px = n_px
        self.result_dtype = np.dtype(result_dtype)
        self.sparse = sparse
        #: result words per mask the library writes (2: a complex stack held as real column pairs)
        self._out_words = 1

    @classmethod
    def dense(cls, device, masks, result_dtype):
        """
        masks: host array (n_masks, n_px); cast to result_dtype here, like
        `for_backend(m, backend).astype(dtype)` (reference common/container.py:90-91).
        """
        result_dtype = np.dtype(result_dtype)
        m = np.ascontiguousarray(np.asarray(masks).astype(result_dtype, copy=False))
        if m.ndim != 2:
            raise ValueError("dense mask stack must be 2D (n_masks, n_px)")
        out = ctypes.c_void_p()
        check(lib().ltmi_masks_create_dense(
            int(device), m.ctypes.data_as(ctypes.c_void_p), dtype_code(result_dtype),
            m.shape[0], m.shape[1], ctypes.byref(out)), 'ltmi_masks_create_dense')
        return cls(out, device, m.shape[0], m.shape[1], result_dtype, False)

    @classmethod
    def csr(cls, device, csr_px_by_masks, result_dtype, gather_only=False):
        """csr_px_by_masks: scipy.sparse CSR (n_px, n_masks) as built by the reference's
        `_build_sparse` (common/container.py:53-64).  gather_only: just the gather kernel's image (the
        handle a densified stack keeps for frames with non-finite pixels, `set_sparse_origin`)."""
        result_dtype = np.dtype(result_dtype)
        indptr = np.ascontiguousarray(csr_px_by_masks.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(csr_px_by_masks.indices, dtype=np.int64)
        if result_dtype.kind in 'iu':
            # integer results: int64 values (the library keeps them as doubles: exact, see
            # csrc/ltmi_sparse.hip csr_apply); uint64 values beyond 2^63 do not fit
            data = np.ascontiguousarray(signed_representative(
                csr_px_by_masks.data.astype(result_dtype, copy=False)))
        else:
            data = np.ascontiguousarray(csr_px_by_masks.data.astype(result_dtype, copy=False))
        n_px, n_masks = csr_px_by_masks.shape
        out = ctypes.c_void_p()
        create = lib().ltmi_masks_create_csr_gather if gather_only else lib().ltmi_masks_create_csr
        check(create(
            int(device), indptr.ctypes.data_as(ctypes.c_void_p),
            indices.ctypes.data_as(ctypes.c_void_p), data.ctypes.data_as(ctypes.c_void_p),
            dtype_code(result_dtype), n_px, n_masks, ctypes.byref(out)), 'ltmi_masks_create_csr')
        return cls(out, device, n_masks, n_px, result_dtype, True)

    @classmethod
    def csr_complex128(cls, device, csr_px_by_masks, gather_only=False):
        """A complex128 sparse stack for REAL frames, without densifying it: (re, im) of mask k are the
        float64 columns 2k, 2k + 1 of the image (the float64 gather kernel), and the result row of a
        frame -- 2 n_masks doubles -- is its complex128 row.  (complex64 stacks do the same inside the
        library.)"""
        import scipy.sparse as sp
        m = sp.csr_matrix(csr_px_by_masks)
        m.sum_duplicates()
        m.sort_indices()
        n_px, n_masks = m.shape
        vals = np.ascontiguousarray(m.data.astype(np.complex128, copy=False))
        counts = np.diff(m.indptr)
        data = np.empty(2 * vals.size, dtype=np.float64)
        indices = np.empty(2 * vals.size, dtype=np.int64)
        data[0::2], data[1::2] = vals.real, vals.imag
        indices[0::2], indices[1::2] = 2 * m.indices.astype(np.int64), 2 * m.indices.astype(np.int64) + 1
        indptr = np.concatenate([[0], np.cumsum(2 * counts)]).astype(np.int64)
        real = sp.csr_matrix((data, indices, indptr), shape=(n_px, 2 * n_masks))
        h = cls.csr(device, real, np.float64, gather_only=gather_only)
        h.n_masks = n_masks
        h.result_dtype = np.dtype(np.complex128)
        h._out_words = 2
        return h

    def kind(self):
        k = ctypes.c_int(-1)
        check(lib().ltmi_masks_kind(self._ptr, ctypes.byref(k)), 'ltmi_masks_kind')
        return k.value

    def set_sig_shape(self, sig_h, sig_w):
        """the detector shape behind the handle's pixels: lets the library fold a stack that is even / odd under a
        mirror of the detector rows (include/ltmi.h)"""
        check(lib().ltmi_masks_set_sig_shape(self._ptr, int(sig_h), int(sig_w)), 'ltmi_masks_set_sig_shape')

    def set_sparse_origin(self, gather):
        """this DENSE handle holds a stack the reference multiplies sparse (a densified CSR stack): `gather`, the
        MaskHandle.csr / csr_complex128 of the same stack, serves the frames whose results come out non-finite --
        stored entries only, like the reference (include/ltmi.h).  Takes ownership of `gather`."""
        check(lib().ltmi_masks_set_sparse_origin(self._ptr, gather._ptr), 'ltmi_masks_set_sparse_origin')
        gather._ptr = None

    def set_dense_origin(self, csr_px_by_masks):
        """this CSR handle holds the non-zeros of a stack the reference multiplies DENSE: a non-finite pixel then
        reaches every mask, also through the zeros the handle does not store (include/ltmi.h)"""
        indptr = np.ascontiguousarray(csr_px_by_masks.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(csr_px_by_masks.indices, dtype=np.int64)
        check(lib().ltmi_masks_set_dense_origin(
            self._ptr, indptr.ctypes.data_as(ctypes.c_void_p), indices.ctypes.data_as(ctypes.c_void_p)),
            'ltmi_masks_set_dense_origin')

    def nonfinite_frames(self, stream=None):
        """frames of the last product that were listed for the non-finite redo / fix-up (synchronises the stream)"""
        n = ctypes.c_int64(0)
        check(lib().ltmi_masks_nonfinite_frames(self._ptr, stream if isinstance(stream, int) else _stream_ptr(stream),
                                                ctypes.byref(n)), 'ltmi_masks_nonfinite_frames')
        return int(n.value)

    def set_tuning(self, mt=0, waves=0, ksplit=0):
        check(lib().ltmi_masks_set_tuning(self._ptr, mt, waves, ksplit), 'ltmi_masks_set_tuning')

    def last_kernel(self):
        return lib().ltmi_masks_last_kernel(self._ptr).decode()

    def apply(self, tile_ptr, tile_dtype, n_frames, ld_tile, out_ptr, ld_out, accumulate,
              stream=None):
        if LaunchReplay.expected is not None or LaunchReplay.recording is not None:
            sig = LaunchReplay.signature(self, tile_ptr, tile_dtype, n_frames, ld_tile, out_ptr, ld_out,
                                         accumulate, stream)
            if LaunchReplay.expected is not None:
                if LaunchReplay.expected and LaunchReplay.expected[0] == sig:
                    LaunchReplay.expected.pop(0)              # enqueued ahead of the book-keeping
                    return
                raise ReplayMismatch(f"launch {sig!r} was not the one enqueued ahead")
            LaunchReplay.recording.append((self, sig))
        if KernelTimer.enabled:
            import torch
            st = torch.cuda.current_stream() if stream is None or isinstance(stream, int) \
                else stream
            if isinstance(stream, int) and st.cuda_stream != stream:
                st = torch.cuda.ExternalStream(stream)
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record(st)
        # argtypes are declared: plain ints marshal as pointers
        rc = _lib.ltmi_apply_masks(
            self._ptr, tile_ptr, _DTYPES.get(tile_dtype) or dtype_code(tile_dtype), n_frames, ld_tile, out_ptr,
            ld_out * self._out_words,
            1 if accumulate else 0, stream if isinstance(stream, int) else _stream_ptr(stream))
        if rc:
            check(rc, 'ltmi_apply_masks')
        if KernelTimer.enabled:
            b.record(st)
            KernelTimer.events.append((a, b, n_frames, self.last_kernel()))

    def apply_rows(self, tile_ptr, tile_dtype, rows_ptr, n_rows, ld_tile, out_ptr, ld_out, accumulate,
                   stream=None):
        """out[i] (+)= product of frame rows[i] of the tile (rows: device int32).  Returns False --
        nothing done -- when this handle / tile has no row-list kernel (gather the frames instead)."""
        handled = ctypes.c_int(0)
        if KernelTimer.enabled:
            import torch
            st = torch.cuda.current_stream() if stream is None or isinstance(stream, int) \
                else stream
            if isinstance(stream, int) and st.cuda_stream != stream:
                st = torch.cuda.ExternalStream(stream)
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record(st)
        check(lib().ltmi_apply_masks_rows(
            self._ptr, ctypes.c_void_p(tile_ptr), dtype_code(tile_dtype), ctypes.c_void_p(rows_ptr),
            int(n_rows), int(ld_tile), ctypes.c_void_p(out_ptr), int(ld_out) * self._out_words,
            1 if accumulate else 0,
            stream if isinstance(stream, int) else _stream_ptr(stream), ctypes.byref(handled)),
            'ltmi_apply_masks_rows')
        if KernelTimer.enabled and handled.value:
            b.record(st)
            KernelTimer.events.append((a, b, n_rows, self.last_kernel()))
        return bool(handled.value)

    def apply_shifted(self, tile_ptr, tile_dtype, n_frames, ld_tile, sig_h, sig_w, shifts_ptr,
                      out_ptr, ld_out, accumulate, stream=None):
        check(lib().ltmi_apply_masks_shifted(
            self._ptr, ctypes.c_void_p(tile_ptr), dtype_code(tile_dtype), n_frames, ld_tile,
            int(sig_h), int(sig_w), ctypes.c_void_p(shifts_ptr), ctypes.c_void_p(out_ptr), ld_out,
            1 if accumulate else 0, _stream_ptr(stream)), 'ltmi_apply_masks_shifted')

    def apply_shifted_host(self, tile_ptr, tile_dtype, n_frames, ld_tile, sig_h, sig_w, shifts,
                           out_ptr, ld_out, accumulate, stream=None):
        """shifts: host int32 array (n_frames, 2) of (dy, dx), C-contiguous."""
        shifts = np.ascontiguousarray(shifts, dtype=np.int32)
        if shifts.shape != (n_frames, 2):
            raise ValueError(f"shifts must have shape ({n_frames}, 2), got {shifts.shape}")
        check(lib().ltmi_apply_masks_shifted_host(
            self._ptr, ctypes.c_void_p(tile_ptr), dtype_code(tile_dtype), n_frames, ld_tile,
            int(sig_h), int(sig_w), shifts.ctypes.data_as(ctypes.c_void_p),
            ctypes.c_void_p(out_ptr), ld_out, 1 if accumulate else 0, _stream_ptr(stream)),
            'ltmi_apply_masks_shifted_host')

    def close(self):
        if self._ptr is not None and self._ptr.value:
            lib().ltmi_masks_destroy(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sum_frames_workspace(n_frames, n_px, out_dtype):
    return int(lib().ltmi_sum_frames_workspace(n_frames, n_px, dtype_code(out_dtype)))


def sum_frames(device, tile_ptr, tile_dtype, n_frames, n_px, ld_tile, out_ptr, out_dtype,
               accumulate, workspace_ptr, stream=None):
    check(lib().ltmi_sum_frames(
        int(device), ctypes.c_void_p(tile_ptr), dtype_code(tile_dtype), n_frames, n_px, ld_tile,
        ctypes.c_void_p(out_ptr), dtype_code(out_dtype), 1 if accumulate else 0,
        ctypes.c_void_p(workspace_ptr), _stream_ptr(stream)), 'ltmi_sum_frames')


def sum_sig(device, tile_ptr, tile_dtype, n_frames, n_px, ld_tile, out_ptr, out_dtype,
            accumulate, stream=None):
    check(lib().ltmi_sum_sig(
        int(device), ctypes.c_void_p(tile_ptr), dtype_code(tile_dtype), n_frames, n_px, ld_tile,
        ctypes.c_void_p(out_ptr), dtype_code(out_dtype), 1 if accumulate else 0,
        _stream_ptr(stream)), 'ltmi_sum_sig')


#: dtypes ltmi_axpy / ltmi_add2d take (every dtype of the enum)
AXPY_DTYPES = frozenset(np.dtype(t) for t in (
    'bool', 'u1', 'i1', 'u2', 'i2', 'u4', 'i4', 'u8', 'i8', 'f4', 'f8', 'c8', 'c16'))


def axpy(device, dest_ptr, src_ptr, dtype, n, stream=None):
    check(lib().ltmi_axpy(int(device), ctypes.c_void_p(dest_ptr), ctypes.c_void_p(src_ptr),
                          dtype_code(dtype), n, _stream_ptr(stream)), 'ltmi_axpy')


def add2d(device, dest_ptr, ld_dest, src_ptr, ld_src, dtype, rows, cols, negate=False, stream=None):
    """dest[r, c] (+|-)= src[r, c]; leading dimensions in elements, ld_src = 0 broadcasts a row."""
    check(lib().ltmi_add2d(int(device), dest_ptr, int(ld_dest), src_ptr, int(ld_src),
                           dtype_code(dtype), int(rows), int(cols), 1 if negate else 0,
                           _stream_ptr(stream)), 'ltmi_add2d')


def host_device_pointer(device, host_ptr):
    """device address of page-locked host memory (raises if it is not device-accessible)"""
    out = ctypes.c_void_p()
    check(lib().ltmi_host_device_pointer(int(device), host_ptr, ctypes.byref(out)),
          'ltmi_host_device_pointer')
    return int(out.value)


def host_copy(dst, src, threads=0):
    """dst[...] = src for two C-contiguous host arrays of the same byte size, on several threads (the GIL is
    released for the call): the staging copy into page-locked bounce buffers at more than the H2D link's rate"""
    if dst.nbytes != src.nbytes or not dst.flags.c_contiguous or not src.flags.c_contiguous:
        raise ValueError("host_copy: C-contiguous arrays of the same size")
    check(lib().ltmi_host_copy(dst.ctypes.data, src.ctypes.data, int(src.nbytes), int(threads)), 'ltmi_host_copy')


def map_or_upload(device, arr):
    """-> (device pointer, keep-alive object) for a C-contiguous host array: zero-copy if the array
    lives in page-locked, device-mapped memory (the result buffers of a run do), else an H2D copy."""
    import torch
    arr = np.ascontiguousarray(arr)
    try:
        return host_device_pointer(device, arr.ctypes.data), arr
    except Exception:
        t = torch.from_numpy(arr).to(f'cuda:{int(device)}')
        return t.data_ptr(), t


def download_pinned(tensor):
    """Device tensor -> NumPy array in page-locked memory (one D2H at link speed; a pageable
    destination is staged by the runtime at a fraction of it).  The array owns its memory."""
    import torch
    host = torch.empty(tensor.shape, dtype=tensor.dtype, pin_memory=True)
    host.copy_(tensor, non_blocking=True)
    torch.cuda.current_stream(tensor.device).synchronize()
    return host.numpy()


def gather_rows(device, src_ptr, ld_src_bytes, idx_ptr, n_rows, row_bytes, dest_ptr, stream=None):
    """dest[i, :] = src[idx[i], :] inside HBM; idx: device int64."""
    check(lib().ltmi_gather_rows(int(device), src_ptr, int(ld_src_bytes), idx_ptr, int(n_rows),
                                 int(row_bytes), dest_ptr, _stream_ptr(stream)), 'ltmi_gather_rows')


def correct(device, tile_ptr, tile_dtype, n_frames, n_px, ld_tile, dark_ptr, gain_ptr, out_ptr,
            out_dtype, ld_out, stream=None):
    """out = ((double)tile - dark) * gain; dark_ptr / gain_ptr: device float64 arrays or None."""
    check(lib().ltmi_correct(
        int(device), tile_ptr, dtype_code(tile_dtype), n_frames, n_px, ld_tile, dark_ptr or None,
        gain_ptr or None, out_ptr, dtype_code(out_dtype), ld_out,
        stream if isinstance(stream, int) else _stream_ptr(stream)), 'ltmi_correct')


def repair_pixels(device, buf_ptr, dtype, n_frames, ld, excl_ptr, env_ptr, cnt_ptr, n_excl, max_env,
                  stream=None):
    check(lib().ltmi_repair_pixels(
        int(device), buf_ptr, dtype_code(dtype), n_frames, ld, excl_ptr, env_ptr, cnt_ptr,
        int(n_excl), int(max_env), stream if isinstance(stream, int) else _stream_ptr(stream)),
        'ltmi_repair_pixels')


def byteswap(device, src_ptr, dst_ptr, itemsize, n_items, stream=None):
    """Reverse the bytes of every `itemsize`-byte item (device pointers; in place if equal)."""
    check(lib().ltmi_byteswap(
        int(device), src_ptr, dst_ptr, int(itemsize), int(n_items),
        stream if isinstance(stream, int) else _stream_ptr(stream)), 'ltmi_byteswap')


def mib_decode(device, src_ptr, frame_stride, header_bytes, kind, bits, quad, n_frames, height, width,
               dst_ptr, dst_dtype, stream=None):
    """Frames of a .mib file (device copy of the file bytes) -> (n_frames, height, width) of `dst_dtype`."""
    check(lib().ltmi_mib_decode(
        int(device), src_ptr, int(frame_stride), int(header_bytes), ord(kind), int(bits), int(bool(quad)),
        int(n_frames), int(height), int(width), dst_ptr, dtype_code(dst_dtype),
        stream if isinstance(stream, int) else _stream_ptr(stream)), 'ltmi_mib_decode')


def com_fields(device, raw_ptr, ld_raw, ny, nx, ref_y, ref_x, transform, out_y, out_x, out_mag=None,
               out_div=None, out_curl=None, stream=None):
    """Shift field + magnitude / divergence / curl of a 2D scan from the raw CoM rows (device
    pointers; `transform`: 2x2 float64, host)."""
    t = np.ascontiguousarray(transform, dtype=np.float64).reshape(4)
    check(lib().ltmi_com_fields(
        int(device), raw_ptr, int(ld_raw), int(ny), int(nx), float(ref_y), float(ref_x),
        t.ctypes.data_as(ctypes.c_void_p), out_y, out_x, out_mag or None, out_div or None,
        out_curl or None, stream if isinstance(stream, int) else _stream_ptr(stream)),
        'ltmi_com_fields')


class FFTPlan:
    """Owns one `ltmi_fft_plan*`: batched 2D R2C hipFFT + workspace for frames of (sig_h, sig_w)."""

    def __init__(self, device, sig_h, sig_w, max_batch):
        out = ctypes.c_void_p()
        check(lib().ltmi_fft_plan_create(int(device), int(sig_h), int(sig_w), int(max_batch),
                                         ctypes.byref(out)), 'ltmi_fft_plan_create')
        self._ptr = out
        self.device, self.sig, self.max_batch = int(device), (int(sig_h), int(sig_w)), int(max_batch)

    def crystallinity(self, tile_ptr, tile_dtype, n_frames, ld_tile, real_mask_ptr, half_mask_ptr,
                      box, out_ptr, accumulate, stream=None):
        """box = (row_lo, row_hi, n_cols): bounding box of the non-zeros of the half mask."""
        check(lib().ltmi_crystallinity(
            self._ptr, tile_ptr, dtype_code(tile_dtype), n_frames, ld_tile, real_mask_ptr or None,
            half_mask_ptr, int(box[0]), int(box[1]), int(box[2]), out_ptr, 1 if accumulate else 0,
            stream if isinstance(stream, int) else _stream_ptr(stream)), 'ltmi_crystallinity')

    def crystallinity_corrected(self, tile_ptr, tile_dtype, n_frames, ld_tile, tables,
                                real_mask_ptr, half_mask_ptr, box, out_ptr, accumulate,
                                stream=None):
        """The same on RAW frames; `tables` = CorrectionSet.device_tables(...): dark / gain (float64
        device tensors or None) and the int32 repair tables."""
        def ptr(t):
            return None if t is None else t.data_ptr()
        check(lib().ltmi_crystallinity_corrected(
            self._ptr, tile_ptr, dtype_code(tile_dtype), n_frames, ld_tile, ptr(tables['dark']),
            ptr(tables['gain']), ptr(tables['excl']), ptr(tables['env']), ptr(tables['cnt']),
            int(tables['n_excl']), int(tables['max_env']), real_mask_ptr or None, half_mask_ptr,
            int(box[0]), int(box[1]), int(box[2]), out_ptr, 1 if accumulate else 0,
            stream if isinstance(stream, int) else _stream_ptr(stream)),
            'ltmi_crystallinity_corrected')

    def last_kernel(self):
        """'k_cryst_fused<...>' or 'hipfft_r2c<...>': the route of the last crystallinity call"""
        return lib().ltmi_fft_plan_last_kernel(self._ptr).decode()

    def close(self):
        if self._ptr is not None and self._ptr.value:
            lib().ltmi_fft_plan_destroy(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def clear_last_runtime_error():
    """forget the HIP runtime's sticky 'last error' of this thread (set by a call that failed outside the
    library, e.g. a host registration that was refused) so that the next launch check does not report it"""
    lib().ltmi_device_count(ctypes.byref(ctypes.c_int(0)))


class Comm:
    """RCCL communicator behind the C ABI (`ltmi_comm*`): what a reference-side binding uses to
    gather nav results / reduce sig results across the GPUs of a node without torch.distributed."""

    ID_BYTES = 128

    @staticmethod
    def library_info():
        """-> dict(path, version, was_loaded): the librccl the communicators are bound to (the copy torch has
        mapped when the process runs torch.distributed -- never a second one)"""
        path = ctypes.create_string_buffer(1024)
        ver, was = ctypes.c_int(0), ctypes.c_int(0)
        check(lib().ltmi_comm_library_info(path, 1024, ctypes.byref(ver), ctypes.byref(was)),
              'ltmi_comm_library_info')
        return dict(path=path.value.decode(), version=int(ver.value), was_loaded=bool(was.value))

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(Comm.ID_BYTES)
        check(lib().ltmi_comm_unique_id(buf), 'ltmi_comm_unique_id')
        return buf.raw

    def __init__(self, device, rank, world, unique_id):
        if len(unique_id) != self.ID_BYTES:
            raise ValueError(f"the RCCL unique id has {self.ID_BYTES} bytes")
        out = ctypes.c_void_p()
        check(lib().ltmi_comm_create(int(device), int(rank), int(world),
                                     ctypes.c_char_p(bytes(unique_id)), ctypes.byref(out)),
              'ltmi_comm_create')
        self._ptr = out
        self.rank, self.world = int(rank), int(world)

    def all_gather(self, send_ptr, recv_ptr, bytes_per_rank, stream=None):
        check(lib().ltmi_comm_all_gather(self._ptr, send_ptr, recv_ptr, int(bytes_per_rank),
                                         _stream_ptr(stream)), 'ltmi_comm_all_gather')

    def all_reduce_sum(self, buf_ptr, dtype, n, stream=None):
        check(lib().ltmi_comm_all_reduce_sum(self._ptr, buf_ptr, dtype_code(dtype), int(n),
                                             _stream_ptr(stream)), 'ltmi_comm_all_reduce_sum')

    def close(self):
        if self._ptr is not None and self._ptr.value:
            lib().ltmi_comm_destroy(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
