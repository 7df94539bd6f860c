"""
Result delivery between the ranks of ONE node without a data-path collective.

With one process per GPU every rank computes the rows of its own nav shard; the reference's
boundary wants the complete result in the caller's process (udf/base.py:2340-2386).  Gathering the
shards on the devices (all_gather over xGMI) and copying the whole buffer to the host on every
rank moves W times the result over every rank's PCIe link.  Here each rank's D2H engine writes ITS
rows straight into a page-locked host segment that all ranks of the node map (`/dev/shm`), W links
in parallel, each byte once; a flag barrier in the same segment publishes completion.  The buffers
the caller gets are views of that segment.

Lifetime: the arrays handed to the caller behave like caller-owned arrays -- they stay valid for as
long as ANY view of them is referenced, on every rank.  Segments form a ring of slots (4 to begin
with, `LTMI_SHM_SLOTS`; mapping and page-locking a fresh one costs milliseconds, so they are reused).
Every run hands out its results as views of a per-run owner object (`_Owner`, an ndarray subclass:
NumPy's base-collapsing stops at it, so every derived view keeps it alive, and it can be weakly
referenced).  A slot is written again only after its owner has died on EVERY rank: each rank
publishes the set of slots whose owner is dead in the end-of-run barrier, the AND of those sets is
the set the next run may choose from -- the same choice on every rank, and no rank can start writing
into memory another rank's caller still looks at.  If all slots are held the ring grows (up to
`LTMI_SHM_MAX_SLOTS`, 16); beyond that a run delivers through the device collectives instead.
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


class NodeSharedBusy(RuntimeError):
    """every slot of the ring is still referenced by earlier results (same answer on every rank):
    this run has to deliver through the device collectives."""


class _Owner(np.ndarray):
    """The array all result views of ONE run derive from.  A subclass so that (a) NumPy does not
    collapse `view.base` past it -- a view of a view of ... keeps it alive -- and (b) it can be
    weakly referenced: `weakref(owner)` dead <=> nobody in this process can see the slot."""


class NodeShared:
    CTRL_BYTES = 4096

    def __init__(self, dist, torch, gpu_id):
        self.d, self.torch, self.gpu_id = dist, torch, gpu_id
        self.rank, self.W = dist.get_rank(), dist.get_world_size()
        self.key = f"ltmi_{os.getuid()}_{os.environ.get('MASTER_PORT', '0')}"
        self.K = max(2, int(os.environ.get('LTMI_SHM_SLOTS', '4')))
        self.K_max = max(self.K, min(60, int(os.environ.get('LTMI_SHM_MAX_SLOTS', '16'))))
        self.slots = [None] * self.K           # dict(path, mm, np, tensor, cap)
        self.owners = [None] * self.K          # weakref to the _Owner of the run that used the slot
        self.common_free = (1 << self.K) - 1   # slots that were free on EVERY rank at the last barrier
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
        seg = dict(path=path, mm=mm, np=arr, cap=nbytes, tensor=None, registered=False, dev=None)
        if register:
            ptr = arr.ctypes.data
            rc = self.torch.cuda.cudart().cudaHostRegister(ptr, nbytes, 0)
            seg['registered'] = int(rc) == 0
            seg['tensor'] = self.torch.from_numpy(arr)
            if seg['registered']:
                try:
                    from libertem_amd import hip
                    seg['dev'] = hip.host_device_pointer(self.gpu_id, ptr)
                except Exception:
                    seg['dev'] = None           # kernels cannot write it: rows are copied out
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
    def _local_free(self):
        m = 0
        for k in range(len(self.slots)):
            ref = self.owners[k]
            if ref is None or ref() is None:
                m |= 1 << k
        return m

    def ready_for(self, nbytes):
        """True if `begin_run(nbytes)` would hand out an existing slot -- no mapping, nothing collective: the
        decision only depends on state that is the same on every rank (launch-ahead, executor/hip.py)."""
        for k in range(len(self.slots)):
            if self.common_free >> k & 1:
                return self.slots[k] is not None and self.slots[k]['cap'] >= nbytes \
                    and self.slots[k]['dev'] is not None
        return False

    def begin_run(self, nbytes):
        """Collective (every rank takes the same decisions: they only depend on `common_free`, the
        slot sizes and `nbytes`, which are the same everywhere).  Returns (slot index, uint8 torch
        tensor over the slot, numpy `_Owner` view of it, device address of the slot | None);
        raises NodeSharedBusy if no slot is free."""
        free = [k for k in range(len(self.slots)) if self.common_free >> k & 1]
        if not free:
            if len(self.slots) >= self.K_max:
                raise NodeSharedBusy(
                    f"all {len(self.slots)} shared result slots are still referenced")
            self.slots.append(None)
            self.owners.append(None)
            free = [len(self.slots) - 1]
        s = free[0]
        if self.slots[s] is None or self.slots[s]['cap'] < nbytes:
            # (re)create all free slots at the new size now: mapping + page-locking costs
            # milliseconds, better in one (warm-up) run than spread over the next ones
            cap = _round_up(max(nbytes, 1 << 21), 1 << 21)
            self.gen += 1
            for k in free:
                if self.slots[k] is None or self.slots[k]['cap'] < cap:
                    self._unmap(self.slots[k])
                    self.slots[k] = self._map(f"{self.key}_s{k}_g{self.gen}", cap)
        seg = self.slots[s]
        owner = seg['np'].view(_Owner)
        self.owners[s] = weakref.ref(owner)
        self.common_free &= ~(1 << s)
        return s, seg['tensor'], owner, seg['dev']

    def all_ok(self, ok, timeout=120.0):
        """Barrier over the node's ranks that ANDs a flag -- and the sets of slots that nobody
        references any more (-> `common_free` for the next run).  The banks alternate with the epoch
        parity: a rank can be at most one epoch ahead of the slowest one."""
        self.epoch += 1
        W, c = self.W, self.ctrl
        bank = W * (1 + (self.epoch & 1))
        c[bank + self.rank] = (self._local_free() << 1) | (1 if ok else 0)
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
        both = int(np.bitwise_and.reduce(c[bank:bank + W]))
        self.common_free = both >> 1
        return bool(both & 1)

    def close(self):
        if self._closed:
            return
        self._closed = True
        for seg in self.slots:
            self._unmap(seg)
        self.slots = [None] * len(self.slots)
        seg = self._ctrl_seg
        if seg is not None and self.rank == 0:
            try:
                os.unlink(seg['path'])
            except OSError:
                pass
