"""
MemoryDataSet: frames held in host memory (NumPy) or directly in HBM (torch-ROCm tensor /
HipArray).  Same constructor surface as the reference's MemoryDataSet
(io/dataset/memory.py:202-406): `MemoryDataSet(data=..., tileshape=None, num_partitions=None,
sig_dims=2, ...)`.

Tile delivery:
* BACKEND_NUMPY: host tiles in the negotiated shape, converted with `astype(dest_dtype)` when
  the dtype differs (memory.py:102-105), frame groups outermost, sig slices innermost.
* BACKEND_HIP, device-resident data: zero-copy views of HBM, native dtype (the conversion is fused
  into the kernels).
* BACKEND_HIP, host data: full-frame chunks staged through two pinned host buffers and two device
  buffers; `hipMemcpyAsync` on a copy stream overlaps the upload of chunk i+1 with the kernels of
  chunk i (events order the two streams; no host synchronisation inside the loop).
"""
import os
import time

import weakref
import numpy as np
import psutil

from libertem_amd.common.math import prod
from libertem_amd.common.shape import Shape
from libertem_amd.common.slice import Slice
from libertem_amd.common.udf import NUMPY, HIP
from libertem_amd.common.hiparray import HipArray, torch_dtype_for
from .base import DataSet, DataSetException, DataSetMeta, Partition, DataTile, TilingScheme


def _is_torch_tensor(x):
    return type(x).__module__.startswith('torch') and hasattr(x, 'is_cuda')


class MemoryDataSet(DataSet):
    def __init__(self, tileshape=None, num_partitions=None, data=None, sig_dims=None,
                 check_cast=True, tiledelay=None, datashape=None, base_shape=None,
                 force_need_decode=False, io_backend=None, nav_shape=None, sig_shape=None,
                 sync_offset=0, array_backends=None, dtype=None, shard=None):
        super().__init__()
        if data is None:
            # (reference io/dataset/memory.py:222-226: float32 zeros of `datashape`)
            if datashape is None:
                raise DataSetException('MemoryDataSet can be created from either data [np.ndarray], or datashape '
                                       '[tuple | Shape], both arguments are None')
            data = np.zeros(tuple(datashape), dtype=np.float32)
        if io_backend is not None:
            raise ValueError("MemoryDataSet currently doesn't support alternative I/O backends")
        self._device_array = None
        self._partitions = None
        if isinstance(data, HipArray):
            self._device_array = data
            self._data = None
        elif _is_torch_tensor(data):
            if not data.is_cuda:
                data = data.numpy()
                self._data = data
            else:
                # `dtype` lets the caller declare unsigned data held in a signed tensor
                self._device_array = HipArray.from_torch(data.contiguous(), dtype=dtype)
                self._data = None
        else:
            self._data = np.asarray(data)
        self._swap_itemsize = 0
        if self._data is not None and not self._data.dtype.isnative:
            if self._data.dtype.kind in 'iu':
                # Integers in the other byte order (big-endian detector files) stay as they are: the
                # NumPy path converts per tile with astype() like the reference
                # (io/dataset/memory.py:102-105), the HIP path uploads the raw bytes and swaps them on
                # the device (ltmi_byteswap; reference io/dataset/base/decode.py:123-158 does it on the
                # host for every tile).
                self._swap_itemsize = self._data.dtype.itemsize
            else:
                # floats: one host copy (the reference's decoder has no float byte swapping at all)
                self._data = self._data.astype(self._data.dtype.newbyteorder('='))
        full_shape = tuple(self._device_array.shape if self._device_array is not None
                           else self._data.shape)
        # shard=(rank, world): `data` is this rank's contiguous block of a larger dataset whose
        # first nav axis is `world` times longer (one process per GPU, torch.distributed);
        # partitions of other ranks exist (for planning / result shapes) but hold no data here.
        self._shard = None
        if shard is not None:
            rank, world = int(shard[0]), int(shard[1])
            if not (0 <= rank < world):
                raise DataSetException(f"invalid shard {shard}")
            self._shard = (rank, world)
        if sig_dims is None and sig_shape is None:
            sig_dims = 2
        if sig_shape is not None:
            sig_shape = tuple(sig_shape)
            sig_dims = len(sig_shape)
        if nav_shape is not None or sig_shape is not None:
            sig_s = sig_shape if sig_shape is not None else full_shape[-sig_dims:]
            n_sig = prod(sig_s)
            total = prod(full_shape)
            nav_s = tuple(nav_shape) if nav_shape is not None else (total // n_sig,)
            if prod(nav_s) * n_sig != total:
                raise DataSetException("nav_shape/sig_shape do not match the data size")
            full_shape = tuple(nav_s) + tuple(sig_s)
        if len(full_shape) <= sig_dims:
            raise DataSetException("data must have at least one navigation dimension")
        sync_offset = int(sync_offset or 0)
        if sync_offset != 0:
            # frame g of `data` belongs to scan position g - sync_offset (reference io/dataset/memory.py:352-406 via
            # base/partition.py): positive offsets skip frames, negative ones leave the first positions blank.
            # Positions without a frame hold zero frames in the array (like RawFileDataSet / MIBDataSet of this
            # package): the host tile loop skips them like the reference (`_valid_frames`), the device path multiplies
            # the zeros -- the same sums, masks and CoM results.
            if self._data is None or shard is not None:
                raise DataSetException("sync_offset of a MemoryDataSet needs host data held by one process")
            n_sig = prod(full_shape[-sig_dims:])
            n_nav = prod(full_shape[:-sig_dims])
            if not (-n_nav < sync_offset < n_nav):
                raise DataSetException(
                    f"offset should be in ({-n_nav}, {n_nav}), which is (-image_count, image_count)")
            frames = self._data.reshape((n_nav, n_sig))
            skip, lead = max(0, sync_offset), max(0, -sync_offset)
            avail = max(0, min(n_nav - skip, n_nav - lead))
            moved = np.zeros_like(frames)
            moved[lead:lead + avail] = frames[skip:skip + avail]
            self._data = moved.reshape(full_shape)
            self._valid_frames = (lead, lead + avail)
        self._local_shape = Shape(full_shape, sig_dims=sig_dims)
        if self._shard is not None:
            full_shape = (full_shape[0] * self._shard[1],) + tuple(full_shape[1:])
        self._shape = Shape(full_shape, sig_dims=sig_dims)
        if tileshape is not None:
            tileshape = tuple(tileshape)
            if len(tileshape) != sig_dims + 1:
                raise DataSetException("tileshape must have one nav dim + the sig dims")
        self.tileshape = tileshape
        self._base_shape = base_shape
        self._force_need_decode = force_need_decode
        self._default_partitions = num_partitions is None
        if num_partitions is None:
            if self._device_array is not None:
                num_partitions = 1
            else:
                num_partitions = psutil.cpu_count(logical=False) or 1
        self.num_partitions = int(num_partitions)
        self._tiledelay = tiledelay
        self._check_cast = check_cast
        self._sync_offset = sync_offset
        raw_dtype = self._device_array.dtype if self._device_array is not None \
            else self._data.dtype.newbyteorder('=')
        self._meta = DataSetMeta(shape=self._shape, raw_dtype=raw_dtype,
                                 image_count=prod(self._shape.nav))

    # --- properties -----------------------------------------------------------------------------
    @property
    def data(self):
        return self._data if self._data is not None else self._device_array

    @property
    def dtype(self):
        return self._meta.raw_dtype

    @property
    def shape(self):
        return self._shape

    @property
    def is_device_resident(self):
        return self._device_array is not None

    @property
    def array_backends(self):
        if self._device_array is not None:
            return (HIP,)
        return (NUMPY, HIP)

    #: host data on a GPU executor, `num_partitions` not given: one partition per this many bytes
    HIP_DEFAULT_PARTITION_BYTES = 1 << 30

    def initialize(self, executor):
        # The reference's default -- one partition per CPU core (io/dataset/memory.py:255-256) -- is a
        # worker count.  A GPU executor streams host data through ONE device: 128 partitions of a
        # 2 GiB array cost a third of the host-link rate (36.7 vs 55.4 GB/s) in per-partition
        # set-up, so the default becomes one partition per GiB there.
        if self._default_partitions and self._device_array is None and \
                getattr(executor, 'gpu_id', None) is not None:
            n_local = prod(self._local_shape.nav)
            nbytes = n_local * prod(self.shape.sig) * np.dtype(self.dtype).itemsize
            want = -(-int(nbytes) // self.HIP_DEFAULT_PARTITION_BYTES)
            self.num_partitions = int(max(1, min(self.num_partitions, want, max(1, n_local))))
        return self

    def get_num_partitions(self):
        if self._shard is not None:
            n_local = prod(self._local_shape.nav)
            return max(1, min(self.num_partitions, n_local)) * self._shard[1]
        return self.num_partitions

    @property
    def shard(self):
        return self._shard

    @property
    def local_frame_range(self):
        """[start, stop) of the frames (flattened global nav) whose data this process holds."""
        n_local = prod(self._local_shape.nav)
        if self._shard is None:
            return 0, n_local
        return self._shard[0] * n_local, (self._shard[0] + 1) * n_local

    def get_base_shape(self, roi):
        if self.tileshape is not None:
            return self.tileshape
        if self._base_shape is not None:
            return self._base_shape
        return super().get_base_shape(roi)

    def get_forced_tileshape(self):
        return self.tileshape

    def adjust_tileshape(self, tileshape, roi):
        if self.tileshape is not None:
            return tuple(self.tileshape)
        return tileshape

    def need_decode(self, read_dtype, roi):
        if self._force_need_decode:
            return True
        return super().need_decode(read_dtype, roi)

    def flat_host(self):
        return self._data.reshape((prod(self._local_shape.nav),) + tuple(self._shape.sig))

    def flat_device(self):
        return self._device_array.reshape((prod(self._local_shape.nav),) +
                                          tuple(self._shape.sig))

    def device_frames(self, local0, n):
        """(device array, row of local frame `local0` in it) for the frames [local0, local0 + n) of this
        process's block -- the whole resident array here; a dataset that holds a window of its frames in
        HBM (MIBDataSet in streamed mode) brings them in now."""
        return self.flat_device(), local0

    eager_upload = True     # HIP path: enqueue the upload of chunk i+1 before the kernels of chunk i

    #: SIGNED integers stored in the other byte order, read into a float or a wider integer dtype:
    #: 'reference' -- the reference's decoders compose the UNSIGNED word from the file's bytes and
    #: store that (io/dataset/base/decode.py:15-66; pinned by tests/golden/decode_signed.npz): -1
    #: stored big-endian reads as 65535.0.  Same-width signed reads wrap back to the signed value.
    #: 'signed' -- the arithmetic value (what NumPy's astype gives).  Results are identical for
    #: non-negative data.
    signed_other_order = 'reference'

    def decoded_dtype(self, dest_dtype):
        """dtype the (byte-swapped) pixels are INTERPRETED as before conversion to `dest_dtype`."""
        dt = np.dtype(self.dtype)
        dest = np.dtype(dest_dtype)
        if self._swap_itemsize > 1 and dt.kind == 'i' and self.signed_other_order == 'reference' \
                and not (dest.kind == 'i' and dest.itemsize == dt.itemsize):
            return np.dtype(f'u{dt.itemsize}')
        return dt

    def wait_for_frames(self, upto):
        """Hook for datasets whose frames are still arriving (io/dataset/stream.py): returns once
        the first `upto` frames of the scan are readable.  Everything is there already here."""

    def frames_ready(self, upto):
        """non-blocking twin of `wait_for_frames`: are the first `upto` frames of the scan readable?"""
        return True

    def close_stagers(self):
        """release the upload buffers kept between partitions and runs (host datasets on the HIP backend)"""
        for st in self.__dict__.pop('_hip_stagers', {}).values():
            st.close()

    def __del__(self):
        try:
            self.close_stagers()
        except Exception:
            pass

    def get_slices(self):
        """Sharded data: every rank's block [r*n_local, (r+1)*n_local) is cut into `num_partitions`
        partitions with the reference's np.linspace rule applied INSIDE the block -- integer
        boundaries that never straddle two ranks' data (a global linspace over world*P partitions
        rounds some boundaries to k*n_local - 1)."""
        if self._shard is None:
            yield from super().get_slices()
            return
        n_local = prod(self._local_shape.nav)
        sig = tuple(self._shape.sig)
        P = max(1, min(self.num_partitions, n_local))
        local = tuple(int(b) for b in np.linspace(0, n_local, num=P + 1, endpoint=True, dtype=int))
        for r in range(self._shard[1]):
            for a, b in zip(local[:-1], local[1:]):
                start, stop = r * n_local + a, r * n_local + b
                yield (Slice(origin=(start,) + (0,) * len(sig),
                             shape=Shape((stop - start,) + sig, sig_dims=len(sig))), start, stop)

    def owner_of_frames(self, start, stop):
        """rank that holds frames [start, stop) of the global nav axis, None if replicated."""
        if self._shard is None:
            return None
        n_local = prod(self._local_shape.nav)
        r = start // n_local
        if stop > (r + 1) * n_local:
            raise DataSetException(f"frames {start}..{stop} straddle two shards of {n_local} frames")
        return int(r)

    def get_partitions(self):
        if self._partitions is None:
            self._partitions = [
                MemPartition(dataset=self, meta=self._meta, partition_slice=part_slice, idx=idx,
                             start_frame=start, num_frames=stop - start)
                for idx, (part_slice, start, stop) in enumerate(self.get_slices())]
        yield from self._partitions

    def __repr__(self):
        where = 'HBM' if self.is_device_resident else 'host'
        return f"<MemoryDataSet of {self.dtype} shape={self.shape} ({where})>"

    def __getstate__(self):
        d = super().__getstate__()
        if d.get('_device_array') is not None:
            raise TypeError("a device-resident MemoryDataSet cannot be pickled")
        d.pop('_hip_stagers', None)             # (device buffers of this process)
        d.pop('_udf_plans', None)
        return d


#: host ranges this process has page-locked for uploads: start address -> [bytes, users].  Two datasets over
#: the same array (or a second stager of one dataset) share the registration -- registering a range twice
#: fails, and the runtime's error would surface at the next kernel launch
_REGISTERED = {}


def pin_limit_bytes():
    """the largest host array that is page-locked IN PLACE for uploads (hipHostRegister walks and locks every
    page: O(size) time, counted against RLIMIT_MEMLOCK, and the pages cannot be swapped or migrated until the
    registration goes): LTMI_PIN_MAX_BYTES, default half of the machine's memory.  Larger arrays travel through
    the stager's two page-locked bounce buffers."""
    env = os.environ.get('LTMI_PIN_MAX_BYTES')
    if env:
        return int(float(env))
    try:
        import psutil
        return int(psutil.virtual_memory().total // 2)
    except Exception:
        return 64 << 30


_PAGE = 4096
_STAGE_DEBUG = bool(os.environ.get('LTMI_STAGE_DEBUG'))


def pin_min_bytes():
    """the smallest host array that is page-locked in place: LTMI_PIN_MIN_BYTES, default 32 MiB = the largest
    value glibc's malloc lets its mmap threshold grow to.  Below it an array may sit in the process heap, on pages
    it shares with its neighbours and that malloc trims and hands out again; the runtime pins and maps pages, and
    a GPU copy out of such a range faulted once in ~3 (later ~13) runs of the test suite ("Memory access fault by GPU ...
    on address <heap address>").  Small arrays go through the two bounce buffers -- a host memcpy of a few MiB."""
    env = os.environ.get('LTMI_PIN_MIN_BYTES')
    return int(float(env)) if env else 32 << 20


def pin_user_arrays():
    """LTMI_PIN_USER_ARRAYS=1: also page-lock user ndarrays of `pin_min_bytes()` and more in place wherever they
    live (round 5's rule).  Off by default since round 6: an intermittent GPU memory access fault on copies out of
    page-locked HEAP arrays was never traced to its cause (profiles/r05_host_fault.txt; the stand-alone reproducer
    probes/hostreg_probe.cpp survives every heap / fork / re-registration pattern tried, profiles/r06_host_upload.txt),
    so in-place page-locking is kept to memory whose mapping PROVABLY belongs to one object for its whole life
    (`_own_mapping`).  Everything else is staged through the two page-locked bounce buffers with a multi-threaded
    copy (ltmi_host_copy)."""
    return os.environ.get('LTMI_PIN_USER_ARRAYS', '0') == '1'


def _glibc_mmapped_chunk(root):
    """True iff `root` (an ndarray that owns its data) sits in a glibc malloc chunk that is a mapping of its own:
    malloc serves large requests (>= its mmap threshold: 128 KiB ... 32 MiB, raised as such chunks are freed) with one
    private anonymous mmap() per chunk and marks the chunk IS_MMAPPED (bit 1 of the size word in front of the user
    pointer, which sits 16 bytes into the first page); free() unmaps it.  No other allocation ever shares its pages
    and the addresses are not recycled while the array lives -- unlike chunks of the brk heap / arenas."""
    import ctypes
    import platform
    if platform.libc_ver()[0] != 'glibc' or ctypes.sizeof(ctypes.c_size_t) != 8:
        return False
    if not (isinstance(root, np.ndarray) and root.flags.owndata and root.base is None and root.nbytes > 0):
        return False
    ptr = root.ctypes.data
    if ptr % _PAGE != 16:
        return False
    size_word = ctypes.c_size_t.from_address(ptr - 8).value        # (same page as the data: readable)
    chunk = size_word & ~7
    return bool(size_word & 2) and chunk % _PAGE == 0 and chunk >= root.nbytes + 16


def _own_mapping(arr):
    """True iff the bytes of `arr` lie in a mapping that belongs to ONE object for its whole life and shares no page
    with anything else: an np.memmap / mmap.mmap (shared memory segments included), or an ndarray that owns a glibc
    malloc chunk with a mapping of its own (`_glibc_mmapped_chunk`: what NumPy gets for large arrays)"""
    import mmap
    root = arr
    while True:
        if isinstance(root, np.memmap) or isinstance(root, mmap.mmap):
            return True
        base = getattr(root, 'base', None)
        if base is None:
            return _glibc_mmapped_chunk(root)
        root = base


def _register_host(torch, arr):
    """page-lock `arr` in place (or join an existing registration that covers it) -> key | None.

    Only for memory with a mapping of its own (`_own_mapping`: np.memmap, glibc-mmapped ndarrays), or for any array of
    `pin_min_bytes()` and more with LTMI_PIN_USER_ARRAYS=1.

    The runtime pins and maps PAGES: an array that shares its first or last page with another live registration
    (malloc places arrays of a few MB next to each other) is not registered -- unregistering the neighbour would take
    the shared page away from under this array's copies ("Memory access fault by GPU ... on address <heap address>",
    once in a few runs of the test suite); it is staged through the bounce buffers instead.  The registered range
    itself stays the array's own bytes: a range rounded outwards would make pageable copies of the NEIGHBOURS
    straddle registered and unregistered memory, which the runtime refuses (invalid argument)."""
    ptr, nbytes = arr.ctypes.data, arr.nbytes
    a0 = ptr // _PAGE * _PAGE
    a1 = -(-(ptr + nbytes) // _PAGE) * _PAGE
    for p0, ent in _REGISTERED.items():
        if p0 <= ptr and ptr + nbytes <= p0 + ent[0]:
            ent[1] += 1
            _reg_log('join', ptr, nbytes)
            return p0
        q0 = p0 // _PAGE * _PAGE
        q1 = -(-(p0 + ent[0]) // _PAGE) * _PAGE
        if a0 < q1 and q0 < a1:
            return None
    own = _own_mapping(arr)
    if not own and not pin_user_arrays():
        return None
    if nbytes > pin_limit_bytes() or (nbytes < pin_min_bytes() and not own):
        return None
    try:
        rc = int(torch.cuda.cudart().cudaHostRegister(ptr, nbytes, 0))
    except Exception:
        rc = -1
    if rc != 0:
        _clear_runtime_error(torch)
        return None
    # A registration must never outlive the memory: once the pages are unmapped (free / munmap) the runtime's
    # mapping of them is gone, and an array that malloc later places at the same addresses would "join" the stale
    # entry and be copied through it -- a GPU memory access fault.  The owner of the bytes takes the registration
    # with it whatever happened to the stagers (a close() that raised, a dataset dropped without close).
    root = arr
    while isinstance(getattr(root, 'base', None), np.ndarray):
        root = root.base
    try:
        fin = weakref.finalize(root, _owner_died, torch, ptr)
        fin.atexit = False
    except TypeError:
        fin = None
    _REGISTERED[ptr] = [nbytes, 1, fin]
    _reg_log('register', ptr, nbytes)
    return ptr


def _reg_log(what, ptr, nbytes=0):
    if os.environ.get('LTMI_REG_LOG'):
        import sys
        print(f"[reg] {what} {ptr:#x} +{nbytes:#x} live={len(_REGISTERED)}", file=sys.stderr, flush=True)


def _owner_died(torch, key):
    ent = _REGISTERED.pop(key, None)
    if ent is None:
        return
    _reg_log('owner died: unregister', key, ent[0])
    try:
        torch.cuda.synchronize()
    except Exception:
        pass
    try:
        rc = int(torch.cuda.cudart().cudaHostUnregister(key))
    except Exception:
        rc = -1
    if rc != 0:
        _clear_runtime_error(torch)


def _unregister_host(torch, key):
    ent = _REGISTERED.get(key)
    if ent is None:
        return
    ent[1] -= 1
    if ent[1] <= 0:
        del _REGISTERED[key]
        _reg_log('unregister', key, ent[0])
        if len(ent) > 2 and ent[2] is not None:
            ent[2].detach()
        try:
            rc = int(torch.cuda.cudart().cudaHostUnregister(key))
        except Exception:
            rc = -1
        if rc != 0:
            _clear_runtime_error(torch)


def _clear_runtime_error(torch):
    """a failed registration leaves its error code as the thread's 'last error'; the next
    hipGetLastError() -- libltmi's launch check -- would report it for an innocent kernel"""
    try:
        from libertem_amd import hip
        hip.clear_last_runtime_error()
    except Exception:
        pass


class _HipStager:
    """
    Double-buffered H2D upload of full-frame chunks (see module docstring).

    If the host array can be page-locked in place (hipHostRegister through torch's runtime
    binding) chunks are DMA-ed straight out of the user's memory; otherwise they are staged
    through two pinned bounce buffers.
    """

    #: staging copies of at least this many bytes run on several threads (ltmi_host_copy)
    PARALLEL_COPY_MIN = 4 << 20
    #: ... on this many threads (0: the library's default)
    STAGE_THREADS = int(os.environ.get("LTMI_STAGE_THREADS", "0"))
    #: ... in pieces of this size, each followed by its DMA
    STAGE_PIECE_BYTES = int(os.environ.get("LTMI_STAGE_PIECE_MIB", "64")) << 20

    def __init__(self, device, chunk_frames, sig, dtype, host_array=None, swap_itemsize=0,
                 host_is_pinned=False):
        import torch
        self.torch = torch
        self.device = device
        self.swap_itemsize = int(swap_itemsize)     # > 0: host data is in the other byte order
        self.sig = tuple(sig)
        self.dtype = np.dtype(dtype)
        self.tdt = torch_dtype_for(dtype)
        self.np_storage_dtype = np.dtype(str(self.tdt).replace('torch.', ''))
        shape = (chunk_frames,) + self.sig
        self.dev = [torch.empty(shape, dtype=self.tdt, device=f'cuda:{device}') for _ in range(2)]
        self.copy_stream = torch.cuda.Stream(device=device)
        self.uploaded = [torch.cuda.Event() for _ in range(2)]
        self.consumed = [None, None]
        self.host_done = [None, None]
        self.registered = None
        self.pinned = None
        self.unregister = True
        self._reg_key = None
        if host_is_pinned and host_array is not None and host_array.flags.c_contiguous:
            # (the dataset allocated the array page-locked -- a stream's scan buffer: nothing to register)
            self.registered = (host_array.ctypes.data, host_array.nbytes)
            self.unregister = False
        elif host_array is not None and host_array.flags.c_contiguous and host_array.nbytes > 0:
            self._reg_key = _register_host(torch, host_array)
            if self._reg_key is not None:
                self.registered = (host_array.ctypes.data, host_array.nbytes)
        if self.registered is None:
            self.pinned = [torch.empty(shape, dtype=self.tdt).pin_memory() for _ in range(2)]

    def _in_registered(self, arr):
        if self.registered is None or not arr.flags.c_contiguous:
            return False
        ptr, nbytes = self.registered
        return ptr <= arr.ctypes.data and arr.ctypes.data + arr.nbytes <= ptr + nbytes

    def upload(self, slot, host_chunk):
        """host (numpy, native dtype) -> device, asynchronously on the copy stream."""
        torch = self.torch
        n = host_chunk.shape[0]
        src_np = host_chunk
        if src_np.dtype != self.np_storage_dtype:
            src_np = src_np.view(self.np_storage_dtype)        # bit reinterpretation only
        pieces = None
        if self._in_registered(src_np):
            import warnings
            with warnings.catch_warnings():
                # read-only sources (memory-mapped files) are only ever read through this tensor
                warnings.filterwarnings('ignore', message='The given NumPy array is not writable')
                src = torch.from_numpy(src_np)
        else:
            if self.pinned is None:
                shape = (self.dev[0].shape[0],) + self.sig
                self.pinned = [torch.empty(shape, dtype=self.tdt).pin_memory() for _ in range(2)]
            t_dbg = time.perf_counter() if _STAGE_DEBUG else 0.
            if self.host_done[slot] is not None:
                self.host_done[slot].synchronize()               # bounce buffer free again?
            if _STAGE_DEBUG:
                self._dbg_wait = time.perf_counter() - t_dbg
            dst_np = self.pinned[slot][:n].numpy()
            src = self.pinned[slot][:n]
            if src_np.flags.c_contiguous and src_np.nbytes >= self.PARALLEL_COPY_MIN:
                # several threads (one thread's memcpy would cap the upload below the link), and in pieces:
                # the DMA of piece i runs while piece i + 1 is staged, so only the first piece's copy is exposed
                frame_bytes = max(1, src_np.nbytes // max(1, n))
                # While an earlier chunk's DMA is still running, this chunk is staged whole behind it and travels as
                # ONE copy (every copy costs the engine a start-up gap).  With the engine idle -- the first chunk of
                # a run -- nothing hides the staging: pieces that grow from 1/8 of the full size, each followed by
                # its DMA, so that only the first small copy is exposed.
                busy = any(ev is not None and not ev.query() for ev in self.host_done)
                if busy:
                    pieces = [(0, n)]
                else:
                    pieces, a, size = [], 0, max(1 << 20, self.STAGE_PIECE_BYTES // 8)
                    while a < n:
                        b = min(n, a + max(1, size // frame_bytes))
                        pieces.append((a, b))
                        a, size = b, min(self.STAGE_PIECE_BYTES, size * 2)
            else:
                dst_np[...] = src_np
        with torch.cuda.stream(self.copy_stream):
            if self.consumed[slot] is not None:
                self.copy_stream.wait_event(self.consumed[slot])  # kernels done with the buffer
            if pieces is None:
                self.dev[slot][:n].copy_(src, non_blocking=True)
            else:
                from libertem_amd import hip
                t_dbg = time.perf_counter() if _STAGE_DEBUG else 0.
                for a, b in pieces:
                    t1 = time.perf_counter() if _STAGE_DEBUG else 0.
                    hip.host_copy(dst_np[a:b], src_np[a:b], self.STAGE_THREADS)
                    t2 = time.perf_counter() if _STAGE_DEBUG else 0.
                    self.dev[slot][a:b].copy_(src[a:b], non_blocking=True)
                    if _STAGE_DEBUG:
                        import sys
                        print(f"[stage]    piece {b - a} frames: copy {(t2 - t1) * 1e3:.3f} ms, enqueue "
                              f"{(time.perf_counter() - t2) * 1e3:.3f} ms", file=sys.stderr)
                if _STAGE_DEBUG:
                    import sys
                    now = time.perf_counter()
                    print(f"[stage] t={now * 1e3 % 100000:9.2f} ms slot {slot} {n} frames: waited {self._dbg_wait * 1e3:.2f} ms "
                          f"for the buffer, staged {len(pieces)} pieces in {(now - t_dbg) * 1e3:.2f} ms", file=sys.stderr)
            if self.swap_itemsize > 1:
                # decode on the device, in place, behind the copy on the same stream
                from libertem_amd import hip
                d = self.dev[slot]
                hip.byteswap(self.device, d.data_ptr(), d.data_ptr(), self.swap_itemsize,
                             n * int(np.prod(self.sig, dtype=np.int64)),
                             stream=self.copy_stream.cuda_stream)
            self.uploaded[slot].record(self.copy_stream)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
            self.host_done[slot] = ev
        return n

    def get(self, slot, n):
        """Make the compute stream wait for the upload; return the device chunk."""
        self.torch.cuda.current_stream(self.device).wait_event(self.uploaded[slot])
        return HipArray(self.dev[slot], (n,) + self.sig, self.dtype)

    def release(self, slot):
        ev = self.torch.cuda.Event()
        ev.record(self.torch.cuda.current_stream(self.device))
        self.consumed[slot] = ev

    def close(self):
        try:
            self.torch.cuda.current_stream(self.device).synchronize()
            self.copy_stream.synchronize()
        finally:
            # (whatever the streams say: the registration goes before the memory can)
            if self.registered is not None:
                if self.unregister:
                    _unregister_host(self.torch, self._reg_key)
                self.registered = None


class MemPartition(Partition):
    def __init__(self, dataset, meta, partition_slice, idx, start_frame, num_frames):
        super().__init__(meta=meta, partition_slice=partition_slice, idx=idx)
        self._ds = dataset
        self._start_frame = start_frame
        # index of the partition's first frame inside THIS process's data block
        self._local0 = start_frame - dataset.local_frame_range[0]
        self._num_frames = num_frames

    def _roi_indices(self, roi):
        if roi is None:
            return None
        roi_part = np.asarray(roi).reshape(-1)[self._start_frame:
                                               self._start_frame + self._num_frames]
        return np.flatnonzero(roi_part) + self._local0

    def get_tiles(self, tiling_scheme, dest_dtype="float32", roi=None, array_backend=NUMPY,
                  env=None, corrections=None):
        ds = self._ds
        if corrections is not None and not corrections.have_corrections():
            corrections = None
        lo, hi = ds.local_frame_range
        if not (lo <= self._start_frame and self._start_frame + self._num_frames <= hi):
            raise RuntimeError(
                f"partition {self._idx} (frames {self._start_frame}..+{self._num_frames}) is not "
                f"held by this process (shard {ds.shard} holds frames {lo}..{hi})")
        if ds.tileshape is not None and (array_backend != HIP or tiling_scheme.intent == 'tile'):
            # a forced tileshape always wins (reference memory.py:427-445).  On the device only for
            # runs of process_tile UDFs, like the scheme negotiated for it (Negotiator._make_scheme_hip):
            # next to a process_frame / process_partition UDF the tiles stay whole frames -- the scheme
            # the UDFs' meta holds -- instead of handing frame pieces to process_frame
            tiling_scheme = TilingScheme.make_for_shape(
                tileshape=Shape(ds.tileshape, sig_dims=ds.shape.sig.dims),
                dataset_shape=ds.shape, intent=tiling_scheme.intent)
        tiling_scheme = tiling_scheme.adjust_for_partition(self)
        if array_backend == HIP:
            yield from self._get_tiles_hip(tiling_scheme, roi, env, np.dtype(dest_dtype),
                                           corrections)
        elif array_backend == NUMPY:
            if ds.is_device_resident:
                raise RuntimeError("a device-resident MemoryDataSet only serves BACKEND_HIP")
            ds.wait_for_frames(self._start_frame + self._num_frames)
            yield from self._get_tiles_numpy(tiling_scheme, np.dtype(dest_dtype), roi,
                                             corrections)
        else:
            raise ValueError(f"unsupported array backend {array_backend!r}")

    # --- host tiles ---------------------------------------------------------------------------------
    def _get_tiles_numpy(self, tiling_scheme, dest_dtype, roi, corrections=None):
        flat = self._ds.flat_host()
        sig_dims = self._ds.shape.sig.dims
        idxs = self._roi_indices(roi)
        depth = int(tiling_scheme.depth)
        src_view = None
        if self._ds.decoded_dtype(dest_dtype) != np.dtype(self._ds.dtype):
            src_view = self._ds.decoded_dtype(dest_dtype).newbyteorder(flat.dtype.byteorder)
        if idxs is None:
            n = self._num_frames
            compressed_origin = self._start_frame
        else:
            n = len(idxs)
            compressed_origin = self.slice.adjust_for_roi(roi).origin[0]
        # scan positions a `sync_offset` leaves without a frame -- before the first or after the last one -- are NOT
        # delivered (reference base/partition.py read ranges; tests/udf/test_coords.py): host UDFs see frames that
        # exist, their result rows elsewhere keep their initial value.  (The device path multiplies the zero frames
        # that stand there: the same sums for every native operator, and write-once result rows stay defined.)
        first, last = 0, n
        valid = getattr(self._ds, '_valid_frames', None)
        if valid is not None:
            lo, hi = valid[0] - self._ds.local_frame_range[0], valid[1] - self._ds.local_frame_range[0]
            if idxs is None:
                first, last = max(0, lo - self._local0), min(n, hi - self._local0)
            else:
                first = int(np.searchsorted(idxs, lo, side='left'))
                last = int(np.searchsorted(idxs, hi, side='left'))
        for g0 in range(first, last, depth):
            g1 = min(last, g0 + depth)
            for scheme_idx, sig_slice in tiling_scheme.slices:
                sig_sl = sig_slice.get(sig_only=True)
                if idxs is None:
                    block = flat[(slice(self._local0 + g0, self._local0 + g1),) + sig_sl]
                else:
                    block = flat[idxs[g0:g1]][(slice(None),) + sig_sl]
                if block.dtype != dest_dtype or not block.flags.c_contiguous \
                        or corrections is not None:
                    if src_view is not None:
                        # signed integers in the other byte order, read like the reference does:
                        # the unsigned word (see `signed_other_order`)
                        block = block.view(src_view)
                    block = block.astype(dest_dtype)          # corrections need a private copy
                tile_slice = Slice(
                    origin=(compressed_origin + g0,) + tuple(sig_slice.origin[-sig_dims:]),
                    shape=Shape((g1 - g0,) + tuple(sig_slice.shape.sig), sig_dims=sig_dims))
                if corrections is not None:
                    corrections.apply(block, tile_slice)      # backend.py:121-124
                yield DataTile(block, tile_slice, scheme_idx)

    # --- device tiles -------------------------------------------------------------------------------
    def _sub_tiles(self, chunk, origin_frame, tiling_scheme):
        """Split a full-frame device chunk (n, *sig) into the scheme's sig slices."""
        sig_dims = self._ds.shape.sig.dims
        ds_sig = tuple(self._ds.shape.sig)
        for scheme_idx, sig_slice in tiling_scheme.slices:
            s_shape = tuple(sig_slice.shape.sig)
            s_origin = tuple(sig_slice.origin[-sig_dims:])
            if s_shape != ds_sig and hasattr(chunk, 'materialize'):
                chunk = chunk.materialize()         # (a row list only stands for whole frames)
            if s_shape == ds_sig:
                data = chunk
            elif s_shape[1:] == ds_sig[1:] and all(o == 0 for o in s_origin[1:]):
                # whole sig rows: a row-strided view, no copy
                data = chunk.sig_rows(s_origin[0], s_origin[0] + s_shape[0])
            else:
                import torch
                t = torch.as_strided(
                    chunk.torch.reshape(-1), (chunk.shape[0], prod(ds_sig)), (chunk.ld, 1)
                ).reshape((chunk.shape[0],) + ds_sig)
                sl = (slice(None),) + tuple(slice(o, o + s) for o, s in zip(s_origin, s_shape))
                data = HipArray(t[sl].contiguous(), (chunk.shape[0],) + s_shape, chunk.dtype)
            tile_slice = Slice(origin=(origin_frame,) + s_origin,
                               shape=Shape((chunk.shape[0],) + s_shape, sig_dims=sig_dims))
            yield DataTile(data, tile_slice, scheme_idx)

    def _corrector(self, corrections, device, dest_dtype, depth, env):
        """-> callable(chunk HipArray native) -> corrected HipArray (dest_dtype) in a scratch buffer
        that is re-used for every chunk (all work is ordered on the executor stream)."""
        from libertem_amd import hip
        ds = self._ds
        if dest_dtype not in (np.dtype('float32'), np.dtype('float64')):
            raise ValueError(f"device corrections produce float32 / float64 tiles, not {dest_dtype}")
        sig = tuple(ds.shape.sig)
        n_px = prod(sig)
        tables = corrections.device_tables(device, sig)
        scratch = HipArray.empty((depth,) + sig, dest_dtype, device)
        stream = getattr(env, 'stream_ptr', None)

        def ptr(t):
            return None if t is None else t.data_ptr()

        def run(chunk):
            n = chunk.shape[0]
            out = scratch.rows(0, n)
            hip.correct(device, chunk.data_ptr(), chunk.dtype, n, n_px, chunk.ld,
                        ptr(tables['dark']), ptr(tables['gain']), out.data_ptr(), dest_dtype,
                        out.ld, stream=stream)
            if tables['n_excl']:
                hip.repair_pixels(device, out.data_ptr(), dest_dtype, n, out.ld,
                                  ptr(tables['excl']), ptr(tables['env']), ptr(tables['cnt']),
                                  tables['n_excl'], tables['max_env'], stream=stream)
            return out
        return run

    def _get_tiles_hip(self, tiling_scheme, roi, env, dest_dtype=None, corrections=None):
        ds = self._ds
        device = env.gpu_id if env is not None and env.gpu_id is not None else 0
        idxs = self._roi_indices(roi)
        depth = int(tiling_scheme.depth)
        n = self._num_frames if idxs is None else len(idxs)
        compressed_origin = self._start_frame if idxs is None \
            else self.slice.adjust_for_roi(roi).origin[0]
        if n == 0:
            return
        fix = (lambda chunk: chunk) if corrections is None else \
            self._corrector(corrections, device, np.dtype(dest_dtype), min(depth, n), env)
        if ds.is_device_resident:
            flat, row0 = ds.device_frames(self._local0, self._num_frames)
            if idxs is not None:
                idxs = idxs - self._local0 + row0           # rows of `flat`
            if flat.device != device:
                raise RuntimeError(f"dataset lives on GPU {flat.device}, worker drives GPU {device}")
            if idxs is not None and corrections is None and \
                    int(idxs.max()) < 2 ** 31 - 1 and flat.is_contiguous:
                # the frames the ROI selects as a ROW LIST over the resident array: the mask kernels
                # read them in place (ltmi_apply_masks_rows), other consumers gather on demand
                import torch
                from libertem_amd.common.hiparray import HipRowsArray
                i64 = torch.from_numpy(np.ascontiguousarray(idxs, dtype=np.int64)).to(
                    f'cuda:{device}')
                flat = HipRowsArray(flat, i64, i64.to(torch.int32), tuple(ds.shape.sig))
                base = 0
            elif idxs is not None:
                # the frames the ROI selects, gathered inside HBM (ltmi_gather_rows)
                from libertem_amd import hip
                sel = HipArray.from_numpy(np.ascontiguousarray(idxs, dtype=np.int64), device)
                gathered = HipArray.empty((n,) + tuple(ds.shape.sig), flat.dtype, device)
                isz = np.dtype(flat.dtype).itemsize
                hip.gather_rows(device, flat.data_ptr(), flat.ld * isz, sel.data_ptr(), n,
                                prod(ds.shape.sig) * isz, gathered.data_ptr(),
                                stream=getattr(env, 'stream_ptr', None))
                flat = gathered
                base = 0
            else:
                base = row0
            for g0 in range(0, n, depth):
                g1 = min(n, g0 + depth)
                chunk = fix(flat.rows(base + g0, base + g1))
                yield from self._sub_tiles(chunk, compressed_origin + g0, tiling_scheme)
            return
        # host data: double-buffered upload, chunk i+1 in flight while chunk i is processed
        host = ds.flat_host()
        part_host = host[self._local0:self._local0 + self._num_frames]
        groups = [(g0, min(n, g0 + depth)) for g0 in range(0, n, depth)]
        if idxs is None:
            yield from self._tiles_through_dataset_stager(ds, host, part_host, groups, depth, device,
                                                          dest_dtype, fix, compressed_origin,
                                                          tiling_scheme)
            return
        stager = _HipStager(device, min(depth, n), ds.shape.sig,
                            ds.decoded_dtype(dest_dtype if dest_dtype is not None else ds.dtype),
                            host_array=None, swap_itemsize=ds._swap_itemsize)

        def host_chunk(g0, g1):
            ds.wait_for_frames(self._start_frame + self._num_frames)
            return host[idxs[g0:g1]]

        try:
            stager.upload(0, host_chunk(*groups[0]))
            for i, (g0, g1) in enumerate(groups):
                slot = i & 1
                chunk = stager.get(slot, g1 - g0)
                yield from self._sub_tiles(fix(chunk), compressed_origin + g0, tiling_scheme)
                stager.release(slot)
                if i + 1 < len(groups):
                    # bounce-buffer mode: the host memcpy overlaps the kernels just enqueued
                    stager.upload(slot ^ 1, host_chunk(*groups[i + 1]))
        finally:
            stager.close()

    def _tiles_through_dataset_stager(self, ds, host, part_host, groups, depth, device, dest_dtype, fix,
                                      compressed_origin, tiling_scheme):
        """Whole partitions of host data (no ROI): ONE stager per dataset and device, kept between
        partitions and runs -- the two device buffers, the copy stream and the page-locking of the host
        array are set up once -- and the first chunk of the NEXT partition is uploaded while this
        partition's last kernels run and its partial result is published (a stream: only if its frames
        have arrived).  Buffer reuse is ordered by events (`consumed` / `host_done`), no stream is
        waited for at a partition's end."""
        dt = ds.decoded_dtype(dest_dtype if dest_dtype is not None else ds.dtype)
        key = (device, int(depth), np.dtype(dt).str, int(ds._swap_itemsize))
        stagers = ds.__dict__.setdefault('_hip_stagers', {})
        st = stagers.get(key)
        if st is None:
            for old in stagers.values():
                old.close()
            stagers.clear()
            st = stagers[key] = _HipStager(
                device, min(depth, host.shape[0]), ds.shape.sig, dt, host_array=host,
                swap_itemsize=ds._swap_itemsize,
                host_is_pinned=bool(getattr(ds, 'host_is_pinned', False)))
            st.seq = 0                      # chunks handed out so far: slot = seq & 1
            st.ahead = None                 # (first local frame, frames, slot) of a chunk uploaded ahead
        eager = st.registered is not None and ds.eager_upload

        def host_chunk(g0, g1):
            ds.wait_for_frames(self._start_frame + g1)     # (a stream: frames up to here)
            return part_host[g0:g1]

        def take(g0, g1):
            """slot that holds (or gets) the chunk [g0, g1) of this partition"""
            a = st.ahead
            st.ahead = None
            if a is not None and a[0] == self._local0 + g0 and a[1] >= g1 - g0:
                return a[2]
            slot = st.seq & 1
            st.upload(slot, host_chunk(g0, g1))
            return slot

        slot = take(*groups[0])
        for i, (g0, g1) in enumerate(groups):
            st.seq = slot + 1
            chunk = st.get(slot, g1 - g0)
            nxt = None
            if i + 1 < len(groups) and eager:
                # DMA straight from user memory: enqueue the next upload before the kernels
                nxt = take(*groups[i + 1])
            yield from self._sub_tiles(fix(chunk), compressed_origin + g0, tiling_scheme)
            st.release(slot)
            if i + 1 < len(groups):
                if nxt is None:
                    # bounce-buffer mode / a stream: the host memcpy overlaps the kernels just enqueued
                    # (the frames of the next chunk may not have arrived yet -- wait for them only
                    # after this chunk's kernels are on their way)
                    nxt = take(*groups[i + 1])
                slot = nxt
        # the next partition's first chunk, if it follows in the host array and is there already
        f_next = self._local0 + self._num_frames
        if f_next < host.shape[0]:
            n_next = min(depth, host.shape[0] - f_next)
            if ds.frames_ready(self._start_frame + self._num_frames + n_next):
                s2 = slot ^ 1
                st.upload(s2, host[f_next:f_next + n_next])
                st.ahead = (f_next, n_next, s2)
                st.seq = s2
