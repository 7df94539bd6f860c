"""
Result delivery between the ranks of ONE node without a data-path collective.

With one process per GPU every rank computes the rows of its own nav shard; the reference's
boundary wants the complete result in the caller's process (udf/base.py:2340-2386).  Gathering the
shards on the devices (all_gather over xGMI) and copying the whole buffer to the host on every
rank moves W times the result over every rank's PCIe link.  Here each rank's D2H engine writes ITS
rows straight into a page-locked host segment that all ranks of the node map (`/dev/shm`), W links
in parallel, each byte once; a flag barrier in the same segment publishes completion.  The buffers
the caller gets are views of that segment.

Segments are a ring of `LTMI_SHM_SLOTS` (default 4) slots per executor, reused run after run (mapping
and page-locking a fresh segment costs milliseconds).  Before a slot is reused, result buffers of
the run that used it last and that are still alive are given a private copy -- arrays obtained
through `BufferWrapper.data` stay valid; a raw ndarray reference taken out of one is only good for
LTMI_SHM_SLOTS - 1 further runs (documented in DESIGN.md section 5).
"""
import atexit
import mmap
import os
import time
import weakref

import numpy as np


class NodeSharedUnavailable(RuntimeError):
    """/dev/shm cannot hold the segment (raised on EVERY rank: the decision is broadcast)."""


def _round_up(n, m):
    return (n + m - 1) // m * m


class NodeShared:
    CTRL_BYTES = 4096

    def __init__(self, dist, torch, gpu_id):
        self.d, self.torch, self.gpu_id = dist, torch, gpu_id
        self.rank, self.W = dist.get_rank(), dist.get_world_size()
        self.key = f"ltmi_{os.getuid()}_{os.environ.get('MASTER_PORT', '0')}"
        self.K = max(2, int(os.environ.get('LTMI_SHM_SLOTS', '4')))
        self.slots = [None] * self.K           # dict(path, mm, np, tensor, cap)
        self.occupants = [[] for _ in range(self.K)]
        self.seq = 0
        self.gen = 0
        self.epoch = 0
        self._closed = False
        c = self._map(f"{self.key}_ctrl", self.CTRL_BYTES, register=False)
        self._ctrl_seg = c
        self.ctrl = c['np'].view(np.int64)
        if 3 * self.W > self.ctrl.size:
            raise RuntimeError(f"too many ranks for the control block: {self.W}")
        atexit.register(self.close)

    # --- segments ----------------------------------------------------------------------------------
    def _map(self, name, nbytes, register=True):
        """Collective: rank 0 creates /dev/shm/<name>, everybody maps it."""
        path = os.path.join('/dev/shm', name)
        ok = [True]
        if self.rank == 0:
            fd = None
            try:
                try:
                    os.unlink(path)
                except FileNotFoundError:
                    pass
                fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
                os.posix_fallocate(fd, 0, nbytes)      # ENOSPC now instead of SIGBUS on first touch
            except OSError as e:
                ok = [f"{e!r}"]
                if fd is not None:
                    os.close(fd)
                    try:
                        os.unlink(path)
                    except OSError:
                        pass
        self.d.broadcast_object_list(ok, src=0)         # (also orders creation before the opens)
        if ok[0] is not True:
            raise NodeSharedUnavailable(f"cannot create {path} ({nbytes} bytes): {ok[0]}")
        if self.rank != 0:
            fd = os.open(path, os.O_RDWR)
        mm = mmap.mmap(fd, nbytes)
        os.close(fd)
        arr = np.frombuffer(mm, dtype=np.uint8)
        if self.rank == 0:
            arr[:] = 0                                     # also faults the pages in
        self.d.barrier()
        seg = dict(path=path, mm=mm, np=arr, cap=nbytes, tensor=None, registered=False)
        if register:
            ptr = arr.ctypes.data
            rc = self.torch.cuda.cudart().cudaHostRegister(ptr, nbytes, 0)
            seg['registered'] = int(rc) == 0
            seg['tensor'] = self.torch.from_numpy(arr)
        return seg

    def _unmap(self, seg):
        if seg is None:
            return
        try:
            if seg['registered']:
                self.torch.cuda.cudart().cudaHostUnregister(seg['np'].ctypes.data)
        except Exception:
            pass
        seg['tensor'] = None
        if self.rank == 0:
            try:
                os.unlink(seg['path'])
            except OSError:
                pass
        # the mapping itself stays alive as long as result views reference it (np.frombuffer)

    # --- per run -----------------------------------------------------------------------------------
    def begin_run(self, nbytes):
        """Collective.  Returns (slot index, uint8 torch tensor over the slot, numpy view)."""
        s = self.seq % self.K
        self.seq += 1
        # results of the run that used this slot last: hand out private copies before overwriting
        for ref in self.occupants[s]:
            bw = ref()
            if bw is not None:
                bw.replace_array(np.array(bw.raw_data, copy=True))
        self.occupants[s] = []
        if self.slots[s] is None or self.slots[s]['cap'] < nbytes:
            # (re)create ALL slots of the ring at the new size now: mapping + page-locking costs
            # milliseconds, better in one (warm-up) run than spread over the next K
            cap = _round_up(max(nbytes, 1 << 21), 1 << 21)
            self.gen += 1
            for k in range(self.K):
                if self.slots[k] is None or self.slots[k]['cap'] < cap:
                    if k != s:
                        for ref in self.occupants[k]:
                            bw = ref()
                            if bw is not None:
                                bw.replace_array(np.array(bw.raw_data, copy=True))
                        self.occupants[k] = []
                    self._unmap(self.slots[k])
                    self.slots[k] = self._map(f"{self.key}_s{k}_g{self.gen}", cap)
        seg = self.slots[s]
        return s, seg['tensor'], seg['np']

    def occupy(self, slot, buffer_wrapper):
        self.occupants[slot].append(weakref.ref(buffer_wrapper))

    def all_ok(self, ok, timeout=120.0):
        """Barrier over the node's ranks that also ANDs a flag.  The flag banks alternate with the
        epoch parity: a rank can be at most one epoch ahead of the slowest one."""
        self.epoch += 1
        W, c = self.W, self.ctrl
        bank = W * (1 + (self.epoch & 1))
        c[bank + self.rank] = 1 if ok else 0
        c[self.rank] = self.epoch
        t0 = time.perf_counter()
        spins = 0
        while True:
            if int(c[:W].min()) >= self.epoch:
                break
            spins += 1
            if spins % 2000 == 0:
                if time.perf_counter() - t0 > timeout:
                    raise RuntimeError(
                        f"rank {self.rank}: timed out waiting for the other ranks of the node "
                        f"(epochs {c[:W].tolist()}, want {self.epoch})")
                time.sleep(0)
        return bool(np.all(c[bank:bank + W] == 1))

    def close(self):
        if self._closed:
            return
        self._closed = True
        for seg in self.slots:
            self._unmap(seg)
        self.slots = [None] * self.K
        seg = self._ctrl_seg
        if seg is not None and self.rank == 0:
            try:
                os.unlink(seg['path'])
            except OSError:
                pass
