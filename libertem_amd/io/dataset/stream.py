"""
StreamDataSet: frames that are still arriving (`ctx.load("stream", frames=..., nav_shape=...,
sig_shape=..., dtype=...)`), the data side of row f4 of SURVEY.md section 8: the reference feeds
partitions of a running acquisition to its workers through queues (executor/pipelined.py:789-1253)
and publishes partial results with `run_udf_iter` (api.py:1053-1152, udf/base.py:2657-2733).

Here one process drives one GPU, so the feed is a thread: it copies the chunks the acquisition
yields into one host buffer and publishes how many frames have landed; a partition's chunks are
uploaded (double-buffered H2D, io/dataset/memory.py) as soon as THEIR frames are there, i.e. the GPU
works on the scan while it is being recorded, and `Context.run_udf_iter` hands out the result after
every partition.  Everything else (tiling, corrections, ROI, sharding) is MemoryDataSet.

In-place feed (`frames=None`): an acquisition that can write into memory it is given -- a detector's DMA
engine, a receiver thread -- fills `ds.scan_buffer` (page-locked: the uploads read it without a bounce copy
and without a feeder memcpy, which bounds the iterator feed at ~40 GB/s) and publishes its progress with
`ds.commit(n_frames_written)`; `ds.finish()` / `ds.fail(exc)` end the acquisition.

Several GPUs: one feeder per rank.  With `shard=(rank, world)` the iterator of a rank delivers ITS
block of the scan (frames [rank * n / world, (rank + 1) * n / world) of the flattened nav axis, the
nav sharding of MemoryDataSet); `run_udf_iter` then advances the ranks in lockstep and every rank
yields the merged partial result after each step (executor/hip.py `_merge_partial_dist`).
"""
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from libertem_amd.common.math import prod
from .base import DataSetException
from .memory import MemoryDataSet


class StreamDataSet(MemoryDataSet):
    """
    Parameters
    ----------
    frames : iterable of array-like, or None
        Each item holds one or more whole frames, in scan order: shape `sig_shape` or
        `(n,) + sig_shape`.  Consumed by a background thread, once.  None: the producer writes the
        frames into `scan_buffer` itself and calls `commit()` (no copy on this side).
    nav_shape, sig_shape : tuple of int
    dtype : numpy dtype of the frames (items are cast to it)
    num_partitions : int, optional
        Partial results are published per partition (default: about 16, at least one frame each).
    timeout : float, optional
        Seconds to wait for missing frames before giving up (default: wait for ever).
    shard : (rank, world), optional
        `frames` delivers only this rank's block of the scan; `nav_shape` is the shape of the WHOLE
        scan, its first axis must be divisible by `world`.
    """

    eager_upload = False    # never wait for the next chunk's frames before launching this chunk

    #: chunks of at least this many bytes are copied into the scan buffer by COPY_THREADS threads (one
    #: thread moves ~20 GB/s, below the 55 GB/s the upload takes them away with)
    PARALLEL_COPY_BYTES = 8 << 20
    COPY_THREADS = 4

    def __init__(self, frames, nav_shape, sig_shape, dtype, num_partitions=None, timeout=None,
                 tileshape=None, shard=None):
        nav_shape = tuple(int(x) for x in nav_shape)
        sig_shape = tuple(int(x) for x in sig_shape)
        if prod(nav_shape) <= 0 or prod(sig_shape) <= 0:
            raise DataSetException(f"empty stream shape {nav_shape} x {sig_shape}")
        local_nav = nav_shape
        self._frame0 = 0                    # global number of this process's first frame
        if shard is not None:
            rank, world = int(shard[0]), int(shard[1])
            if not (0 <= rank < world) or nav_shape[0] % world != 0:
                raise DataSetException(
                    f"cannot shard a scan of {nav_shape} over {world} ranks (shard {shard})")
            local_nav = (nav_shape[0] // world,) + nav_shape[1:]
            self._frame0 = rank * prod(local_nav)
        n_frames = prod(local_nav)
        dt = np.dtype(dtype)
        if not dt.isnative:
            raise DataSetException("a stream delivers frames in the native byte order")
        buf, self.host_is_pinned = self._scan_buffer((n_frames,) + sig_shape, dt, pinned=True)
        if num_partitions is None:
            num_partitions = max(1, min(16, n_frames))
        super().__init__(data=buf.reshape(local_nav + sig_shape), sig_dims=len(sig_shape),
                         num_partitions=num_partitions, tileshape=tileshape, shard=shard)
        self._buf = buf
        self._n_frames = n_frames
        self._timeout = timeout
        self._arrived = 0
        self._finished = False
        self._error = None
        self._cond = threading.Condition()
        self._thread = None
        if frames is not None:
            self._thread = threading.Thread(target=self._pump, args=(iter(frames),), daemon=True,
                                            name='ltmi-stream-feed')
            self._thread.start()

    @staticmethod
    def _scan_buffer(shape, dt, pinned):
        """the host buffer of the scan; page-locked when the producer writes it in place and a GPU is there"""
        if pinned:
            try:
                import torch
                from libertem_amd.io.dataset.memory import pin_limit_bytes
                nbytes = int(np.prod(shape)) * dt.itemsize
                # (page-locked memory is committed when it is allocated: no pass over it to zero it -- a frame is
                #  read only after it has arrived; scans beyond the pin limit use pageable memory + bounce buffers)
                if torch.cuda.is_available() and nbytes <= pin_limit_bytes():
                    from libertem_amd.common.hiparray import torch_dtype_for
                    t = torch.empty(shape, dtype=torch_dtype_for(dt), pin_memory=True)
                    return t.numpy().view(dt), True
            except Exception:                              # pragma: no cover  (no torch / no GPU)
                pass
        # (the same contract in both cases: a frame's bytes mean something once it has ARRIVED -- commit() / the feeder;
        #  readers wait for arrival, `frames_arrived` says how far that is; nothing zeroes what never came)
        return np.empty(shape, dtype=dt), False

    # --- in-place feed ---------------------------------------------------------------------------
    @property
    def scan_buffer(self):
        """(n_frames of this process,) + sig_shape array the producer of an in-place feed writes into.  Frames beyond
        `frames_arrived` hold whatever the memory held (neither the page-locked nor the pageable buffer is zeroed)."""
        return self._buf

    def commit(self, n_frames_written):
        """in-place feed: the first `n_frames_written` frames of this process's block are in `scan_buffer`"""
        n = int(n_frames_written)
        with self._cond:
            if self._thread is not None:
                raise DataSetException("commit() belongs to an in-place feed (frames=None)")
            if not (self._arrived <= n <= self._n_frames):
                raise DataSetException(f"commit({n}): {self._arrived} frames committed, the block has "
                                       f"{self._n_frames}")
            self._arrived = n
            if n == self._n_frames:
                self._finished = True
            self._cond.notify_all()

    def finish(self):
        """in-place feed: no more frames will come (ends a scan that was cut short)"""
        with self._cond:
            self._finished = True
            self._cond.notify_all()

    def fail(self, exc):
        """in-place feed: the acquisition failed; waiting consumers raise"""
        with self._cond:
            self._error = exc
            self._cond.notify_all()

    # --- feed ------------------------------------------------------------------------------------
    def _copy_in(self, start, chunk, pool):
        """self._buf[start:start + n] = chunk (cast + copy), large chunks in parallel slices"""
        n = chunk.shape[0]
        dst = self._buf[start:start + n]
        if chunk.dtype == dst.dtype and chunk.flags.c_contiguous and dst.flags.c_contiguous \
                and chunk.nbytes >= self.PARALLEL_COPY_BYTES:
            # same dtype: the library's multi-threaded memcpy (ltmi_host_copy: ~200 GB/s on 16 threads, GIL released)
            try:
                from libertem_amd import hip
                hip.host_copy(dst, chunk)
                return
            except Exception:                               # noqa: BLE001  (library not built: NumPy copies below)
                pass
        if pool is None or chunk.nbytes < self.PARALLEL_COPY_BYTES or n < 2 * self.COPY_THREADS:
            self._buf[start:start + n] = chunk
            return
        step = -(-n // self.COPY_THREADS)

        def part(a):
            self._buf[start + a:start + min(n, a + step)] = chunk[a:a + step]
        list(pool.map(part, range(0, n, step)))

    def _pump(self, it):
        sig = self._buf.shape[1:]
        pool = ThreadPoolExecutor(self.COPY_THREADS, thread_name_prefix='ltmi-stream-copy') \
            if self.COPY_THREADS > 1 else None
        try:
            self._pump_loop(it, sig, pool)
        finally:
            if pool is not None:
                pool.shutdown(wait=False)

    def _pump_loop(self, it, sig, pool):
        try:
            for item in it:
                chunk = np.asarray(item)
                if chunk.shape == sig:
                    chunk = chunk[None]
                if chunk.shape[1:] != sig:
                    raise DataSetException(
                        f"stream item of shape {chunk.shape} does not hold frames of {sig}")
                n = chunk.shape[0]
                with self._cond:
                    start = self._arrived
                if start + n > self._n_frames:
                    raise DataSetException(
                        f"stream delivered more than the {self._n_frames} frames of the scan")
                self._copy_in(start, chunk, pool)             # cast + copy outside the lock
                with self._cond:
                    self._arrived = start + n
                    self._cond.notify_all()
                if start + n == self._n_frames:
                    break
        except BaseException as e:      # noqa: B036  (handed to the consumer, never swallowed)
            with self._cond:
                self._error = e
                self._cond.notify_all()
            return
        with self._cond:
            self._finished = True
            self._cond.notify_all()

    @property
    def frames_arrived(self):
        with self._cond:
            return self._arrived

    def frames_ready(self, upto):
        upto = max(0, min(int(upto) - self._frame0, self._n_frames))
        with self._cond:
            return self._arrived >= upto

    def wait_for_frames(self, upto):
        """Block until the frames of the scan up to (global) number `upto` that THIS process is
        fed with are in the buffer."""
        upto = max(0, min(int(upto) - self._frame0, self._n_frames))
        with self._cond:
            ok = self._cond.wait_for(
                lambda: self._arrived >= upto or self._error is not None or self._finished,
                timeout=self._timeout)
            if self._error is not None:
                raise DataSetException(f"the frame stream failed: {self._error!r}") from self._error
            if self._arrived >= upto:
                return
            if self._finished:
                raise DataSetException(
                    f"the frame stream ended after {self._arrived} of {self._n_frames} frames")
            if not ok:
                raise DataSetException(
                    f"timed out after {self._timeout} s waiting for frame {upto} "
                    f"({self._arrived} arrived)")

    def __repr__(self):
        return (f"<StreamDataSet of {self.dtype} shape={self.shape} "
                f"({self.frames_arrived}/{self._n_frames} frames arrived)>")

    def __getstate__(self):
        raise TypeError("a StreamDataSet is bound to its feeding thread and cannot be pickled")
