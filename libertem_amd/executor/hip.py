"""
HipJobExecutor: runs UDF tasks on ONE MI355X from the calling process; with torch.distributed
initialised (one process per GPU, backend "nccl" = RCCL over xGMI, or "gloo" in CPU tests) the
partitions are sharded over the ranks and the per-rank results are combined with collectives.

Replaces, for this path, the reference's process/device boundary `executor.run_tasks`
(executor/dask.py:581-646: pickled tasks over TCP, one worker process per GPU with
LIBERTEM_USE_CUDA, results pickled back, serial merge on the main process udf/base.py:2340-2358):

* no pickling: tasks run in-process on a dedicated HIP stream of the worker's GPU;
* device-side merge: UDFs that declare how their buffers combine (`get_dist_merge()`:
  'disjoint' nav rows or 'sum') are merged in HBM -- slice copies / axpy -- and copied to the host
  ONCE per run instead of once per partition (reference: buffers.py:901-907 per partition);
* multi-GPU: rank r owns the contiguous block of partitions  [r*P/W, (r+1)*P/W)  -- the same
  np.linspace nav sharding as the reference's partitioning -- then ONE all_gather (nav-kind,
  disjoint) or all_reduce(sum) (sig-kind) per buffer; every rank ends up with the full result.
  UDFs without a declaration fall back to gathering exported per-partition results and merging
  them in partition order on every rank (reference semantics, slower).
"""
import math
import os
import uuid

import numpy as np

from libertem_amd.common.hiparray import HipArray, HostMappedArray, torch_dtype_for
from libertem_amd.common.buffers import PlaceholderBufferWrapper
from libertem_amd.common.backend import get_use_hip
from .base import JobExecutor, Environment


_DIST_MODULE = False          # False: not imported yet; None: torch.distributed unavailable

from libertem_amd.common import udf as _udf_common     # noqa: E402  (HIP_DIRECT_ROW_MAX, read per run)

# NumPy dtype of the bytes as torch stores them (torch has no unsigned 16 / 32 / 64-bit tensors)
_STORAGE_DTYPE = {np.dtype('uint16'): np.dtype('int16'), np.dtype('uint32'): np.dtype('int32'),
                  np.dtype('uint64'): np.dtype('int64')}


def _dist():
    global _DIST_MODULE
    if _DIST_MODULE is False:
        try:
            import torch.distributed as dist
            _DIST_MODULE = dist if dist.is_available() else None
        except Exception:
            _DIST_MODULE = None
    dist = _DIST_MODULE
    if dist is not None and dist.is_initialized():
        return dist
    return None


#: the executor whose device + stream were last installed as torch's current ones
_CURRENT_OWNER = None


class _LazyTorchDtypes(dict):
    """torch dtype -> NumPy dtype for the dtypes ltmi_comm_all_reduce_sum takes (torch imported late)"""

    def get(self, key, default=None):
        if not self:
            import torch
            self.update({torch.float32: np.float32, torch.float64: np.float64,
                         torch.complex64: np.complex64, torch.complex128: np.complex128,
                         torch.int32: np.int32, torch.int64: np.int64, torch.uint8: np.uint8,
                         torch.int8: np.int8})
        return dict.get(self, key, default)


_NP_OF_TORCH = _LazyTorchDtypes()


class HipJobExecutor(JobExecutor):
    device_class = 'hip'

    def __init__(self, gpu_id=None, distributed=None, require_gpu=True):
        """
        gpu_id : HIP device ordinal; default LIBERTEM_USE_HIP, else LOCAL_RANK, else 0.
        distributed : None = use torch.distributed iff it is initialised; False = never.
        require_gpu : False lets CPU-only tests drive the sharding/merge logic with NumPy UDFs
            (device class stays 'cpu' then; the native UDFs still refuse to run).
        """
        if gpu_id is None:
            gpu_id = get_use_hip()
        if gpu_id is None:
            gpu_id = int(os.environ.get('LOCAL_RANK', 0))
        self._require_gpu = require_gpu
        self._stream = None
        if require_gpu:
            from libertem_amd import hip
            hip.lib()                                   # fail loudly without the HIP library
            import torch
            if not torch.cuda.is_available() or hip.device_count() <= gpu_id:
                raise RuntimeError(
                    f"HipJobExecutor: GPU {gpu_id} not available "
                    f"({hip.device_count()} HIP device(s) visible). There is no CPU fallback.")
            self._stream = torch.cuda.Stream(device=gpu_id)
            self._stream_ptr = int(self._stream.cuda_stream)
            self._torch = torch
            self.gpu_id = gpu_id
            self.device_class = 'hip'
        else:
            self.gpu_id = None
            self.device_class = 'cpu'
        self._distributed = distributed
        self._scattered = {}
        self._pinned = {}

    def sibling(self):
        """an independent executor on the same GPU (a stream, delivery buffers and launch-ahead state of its own): runs
        a UDF while a `run_udf_iter` of this one is suspended between two partial results (Context.run_udf).  With
        several ranks every rank has to start the same nested run -- its collectives use the same process group."""
        return HipJobExecutor(gpu_id=self.gpu_id, distributed=self._distributed, require_gpu=self._require_gpu)

    # --- distributed helpers -----------------------------------------------------------------------
    def _dist(self):
        if self._distributed is False:
            return None
        return _dist()

    @property
    def _collectives_on(self):
        """True iff results have to be combined across ranks.  LTMI_FORCE_COLLECTIVES=1 runs the
        collective code path even with a single rank (used to exercise RCCL on a 1-GPU box)."""
        d = self._dist()
        if d is None:
            return False
        return d.get_world_size() > 1 or os.environ.get('LTMI_FORCE_COLLECTIVES') == '1'

    def _node_shared(self):
        """The shared, page-locked host ring of this node's ranks (executor/nodeshared.py), or None
        when the ranks span several nodes / LTMI_RESULT_VIA=rccl asks for the device collectives."""
        if self.gpu_id is None or not self._collectives_on or getattr(self, '_shared_off', False):
            return None
        via = os.environ.get('LTMI_RESULT_VIA', 'auto')
        if via == 'rccl':
            return None
        if via != 'shm' and int(os.environ.get('LOCAL_WORLD_SIZE', '0')) != self.world_size:
            return None
        if getattr(self, '_shared', None) is None:
            from .nodeshared import NodeShared, NodeSharedUnavailable
            try:
                self._shared = NodeShared(self._dist(), self._torch, self.gpu_id)
            except NodeSharedUnavailable as e:
                self._shared_off = True                  # same on every rank: device collectives
                import logging
                logging.getLogger(__name__).warning("%s -- results go through RCCL", e)
                return None
        return self._shared

    def _make_current(self):
        """Install this executor's GPU and stream as torch's current device / stream (no context
        manager: one process drives one GPU, the executor owns both).  Cheap when nothing changed:
        one raw-stream query guards against user code that switched streams in between."""
        global _CURRENT_OWNER
        torch = self._torch
        if _CURRENT_OWNER is self and \
                torch._C._cuda_getCurrentRawStream(self.gpu_id) == self._stream_ptr:
            return
        torch.cuda.set_device(self.gpu_id)
        torch.cuda.set_stream(self._stream)
        _CURRENT_OWNER = self

    @property
    def rank(self):
        d = self._dist()
        return d.get_rank() if d else 0

    @property
    def world_size(self):
        d = self._dist()
        return d.get_world_size() if d else 1

    def task_owners(self, tasks):
        """rank of every task, the same list on every rank.  Sharded datasets: the rank that HOLDS
        the partition's frames (an ROI may remove whole partitions of one shard, so the position in
        the filtered task list says nothing).  Replicated data: contiguous blocks of the task list,
        the np.linspace nav sharding of the reference's partitioning."""
        W = self.world_size
        if W == 1:
            return [0] * len(tasks)
        owners = []
        for t in tasks:
            ds = getattr(t.partition, '_ds', None)
            own = getattr(ds, 'owner_of_frames', None)
            sl = t.partition.slice
            r = own(sl.origin[0], sl.origin[0] + sl.shape[0]) if own is not None else None
            owners.append(r)
        if all(r is None for r in owners):
            P = len(tasks)
            b = np.linspace(0, P, W + 1, dtype=int)
            owners = [0] * P
            for r in range(W):
                owners[b[r]:b[r + 1]] = [r] * int(b[r + 1] - b[r])
        elif any(r is None for r in owners):
            raise RuntimeError("tasks of sharded and replicated datasets in one run")
        return owners

    def my_tasks(self, tasks):
        """The tasks this rank runs (nav sharding)."""
        if self.world_size == 1:
            return tasks
        r = self.rank
        return [t for t, o in zip(tasks, self.task_owners(tasks)) if o == r]

    # --- executor protocol -----------------------------------------------------------------------------
    def get_local_env(self):
        return Environment(threads_per_worker=None, threaded_executor=False, gpu_id=self.gpu_id,
                           keep_results_on_device=(self.gpu_id is not None), stream=self._stream,
                           ensure_current=(self._make_current if self.gpu_id is not None
                                           else None),
                           row_sink=getattr(self, '_row_sink', None),
                           result_target=getattr(self, '_result_target', None))

    def scatter(self, obj):
        self._scatter_seq = getattr(self, '_scatter_seq', 0) + 1
        handle = f"params-{self._scatter_seq}"
        self._scattered[handle] = obj
        return handle

    def scatter_release(self, handle):
        self._scattered.pop(handle, None)

    def run_tasks(self, tasks, params_handle, cancel_id, task_comm_handler=None):
        params = self._scattered[params_handle]
        self._all_tasks = list(tasks)                  # (known to merge_results before the first task runs)

        def run():
            env = self.get_local_env()                 # (picks up the delivery targets merge_results has set)
            for task in self.my_tasks(self._all_tasks):
                result = task(env=env, params=params)
                yield result, task
        return run()

    def run_function(self, fn, *args, **kwargs):
        return fn(*args, **kwargs)

    def get_available_workers(self):
        from .workers import Worker, WorkerSet
        res = {'HIP': 1, 'compute': 1} if self.gpu_id is not None else \
            {'CPU': 1, 'compute': 1, 'ndarray': 1}
        return WorkerSet([Worker(name=f'hip-{self.gpu_id}', host='localhost', resources=res,
                                 nthreads=1)])

    def close(self):
        self._scattered = {}
        if getattr(self, '_comm_state', None):
            self._comm_state.close()
            self._comm_state = None
        if getattr(self, '_shared', None) is not None:
            self._shared.close()
            self._shared = None

    # --- merging ------------------------------------------------------------------------------------
    _before_wait = None

    def set_before_wait(self, fn):
        """`fn()` is called once in the next `merge_results`, after the kernels of every task of this
        rank are enqueued and before the first wait or delivery -- host work that may hide behind
        the GPU (UDFRunner: the content comparison of a re-used plan).  If it raises, the stream is
        drained, nothing is delivered and the exception propagates."""
        self._before_wait = fn

    def launch_ahead(self, tasks, result_where=None):
        """Launch first, book-keep behind the kernel: every task of a cached plan whose previous run made
        exactly one mask launch into directly written rows of the run's host buffer (recorded by
        merge_results) gets that launch enqueued NOW, into the buffer of the page-locked ring that
        merge_results will adopt for this run; the tile loop's own call is then recognised and skipped
        (hip.LaunchReplay).  Single rank, complete runs (no partial results) only."""
        from libertem_amd import hip as _hip
        if self.gpu_id is None or _hip.LaunchReplay.expected is not None or result_where == 'device':
            return False
        recs = [(getattr(t, '_keep', None) or {}).get('replay') for t in self.my_tasks(list(tasks))]
        if not recs or not all(recs) or len({r[3:] for r in recs}) != 1:
            return False
        total, mode = recs[0][3], recs[0][4]
        if mode == 'ring':
            if self._collectives_on or getattr(self, '_pinned_ring', None) is None:
                return False
            slot = self._pinned_ring.get(total)
        else:
            # several ranks of one node: this rank's rows of the node-shared segment.  Only when the next
            # slot exists already -- handing it out is then a local decision that every rank takes alike,
            # whether it launches ahead or not (executor/nodeshared.py)
            shared = self._node_shared()
            if shared is None or not shared.ready_for(total):
                return False
            slot = shared.begin_run(total)[1:]
        base_dev = slot[2]
        if base_dev is None:
            return False
        self._make_current()
        expected = []
        for handle, sig, off, _, _ in recs:
            sig = sig[:5] + (base_dev + off,) + sig[6:]
            handle.apply(sig[1], sig[2], sig[3], sig[4], sig[5], sig[6], sig[7], stream=sig[8])
            expected.append(sig)
        self._ahead = dict(slot=slot, total=total, mode=mode)
        _hip.LaunchReplay.expected = expected
        _hip.LaunchReplay.n_ahead += len(expected)
        return True

    _before_final = None

    def set_before_final(self, fn):
        """`fn()` is called once in the next `merge_results` when every declared buffer was written
        straight into its final host array: after the arrays are attached to the buffers, before the wait
        for the kernels -- host work on the result OBJECTS (not their contents) that hides behind the GPU."""
        self._before_final = fn

    def drain(self):
        """a run was abandoned half way: wait for what it enqueued, forget its delivery targets"""
        self._ahead = None
        self._before_final = None
        from libertem_amd import hip as _hip
        _hip.LaunchReplay.expected = None
        _hip.LaunchReplay.recording = None
        self._row_sink = None
        self._result_target = None
        for st in (self._stream, getattr(self, '_copy_stream', None)):
            if st is not None:
                st.synchronize()

    def merge_results(self, udfs, damage, result_iter, apply_part_result, result_where=None):
        """
        Consume (part_results, task) pairs of THIS rank, merge, combine across ranks and leave the
        complete result in every udf.results (host buffers; result_where='device': HipArrays in HBM)
        and `damage`.
        """
        for _ in self._merge(udfs, damage, result_iter, partial=False, result_where=result_where):
            pass

    def merge_results_iter(self, udfs, damage, result_iter, apply_part_result):
        """Generator form for `run_udf_iter` (reference udf/base.py:2657-2733): yields after every
        merged partition with the current state published to the host buffers (one D2H per
        buffer and partition -- opt-in cost of watching partial results).  With several ranks the
        ranks advance in lockstep -- one step = the next partition of EVERY rank, then one
        collective per buffer -- and every rank yields the same partial result after each step."""
        if self._collectives_on:
            yield from self._merge_partial_dist(udfs, damage, result_iter)
        else:
            yield from self._merge(udfs, damage, result_iter, partial=True)

    def _merge_plans(self, udfs):
        plans = []
        for udf in udfs:
            decl = getattr(udf, 'get_dist_merge', lambda: None)()
            names = [k for k, b in udf.results.items()
                     if not isinstance(b, PlaceholderBufferWrapper)]
            if decl is not None and set(names) <= set(decl) and self.gpu_id is not None:
                plans.append(('device', decl))
            elif decl is not None and set(names) <= set(decl):
                plans.append(('host-declared', decl))
            else:
                plans.append(('generic', None))
        return plans

    def _merge_partial_dist(self, udfs, damage, result_iter):
        """Partial results across ranks (live acquisitions fed to several GPUs, one feeder per rank:
        io/dataset/stream.py with shard=; reference executor/pipelined.py:789-1253 +
        udf/base.py:2657-2733).  After step k every rank holds the merge of the first k + 1
        partitions of every rank; the damage map says which rows that is."""
        import torch
        d = self._dist()
        plans = self._merge_plans(udfs)
        self._row_sink = None
        self._result_target = None
        self.last_result_via = 'collective'
        it = iter(result_iter)
        first = next(it, None)                  # (runs the executor's task set-up: _all_tasks)
        tasks = self._all_tasks
        owners = self.task_owners(tasks)
        per_rank = [[t for t, o in zip(tasks, owners) if o == r] for r in range(self.world_size)]
        n_steps = max((len(x) for x in per_rank), default=0)
        dev_full = [dict() for _ in udfs]
        local_host = {}                         # (udf idx, name) -> this rank's own merged array
        by_idx = {t.idx: t for t in tasks}

        def publish():
            for i, (udf, (mode, decl)) in enumerate(zip(udfs, plans)):
                if mode == 'generic':
                    continue
                for name, how in decl.items():
                    buf = udf.results.get_buffer(name)
                    if isinstance(buf, PlaceholderBufferWrapper):
                        continue
                    if mode == 'device':
                        self._make_current()
                        full = dev_full[i].get(name)
                        t = torch.zeros(buf.shape, dtype=torch_dtype_for(buf.dtype),
                                        device=f'cuda:{self.gpu_id}') if full is None \
                            else full.clone()
                        host = self._to_host(self._combine(d, t, how))
                    else:
                        mine = local_host.get((i, name))
                        if mine is None:
                            mine = local_host[(i, name)] = np.array(buf.raw_data, copy=True)
                        t = torch.from_numpy(np.array(mine, copy=True))
                        host = self._combine(d, t, how).numpy()
                    if host.dtype != buf.dtype:
                        host = host.view(buf.dtype)
                    buf.replace_array(host)

        pending = first
        for step in range(n_steps):
            new_generic = []
            if pending is not None:
                part_results, task = pending
                entry = {}
                for i, (udf, results, (mode, decl)) in enumerate(zip(udfs, part_results, plans)):
                    if mode == 'device':
                        self._merge_on_device(udf, results, task, decl, dev_full[i],
                                              may_adopt=False)
                    elif mode == 'host-declared':
                        results.export()
                        for name in decl:           # merge into THIS rank's own state
                            buf = udf.results.get_buffer(name)
                            if (i, name) in local_host and \
                                    not isinstance(buf, PlaceholderBufferWrapper):
                                buf.replace_array(local_host[(i, name)])
                        self._apply_one(udf, results, task, damage)
                        for name in decl:
                            buf = udf.results.get_buffer(name)
                            if not isinstance(buf, PlaceholderBufferWrapper):
                                local_host[(i, name)] = buf.raw_data
                    else:
                        results.export()
                        entry[i] = results
                if entry:
                    new_generic.append((task.idx, entry))
                pending = next(it, None)
            publish()
            if any(mode == 'generic' for mode, _ in plans):
                gathered = [None] * self.world_size
                d.all_gather_object(gathered, new_generic)
                for tidx, entry in sorted((p for chunk in gathered for p in chunk),
                                          key=lambda x: x[0]):
                    for i, results in entry.items():
                        self._apply_one(udfs[i], results, by_idx[tidx], damage)
            for r in range(self.world_size):
                for t in per_rank[r][:step + 1]:
                    damage.get_view_for_partition(t.partition)[:] = True
            yield step + 1
        if n_steps == 0:
            yield 0
        if self._stream is not None:
            self._stream.synchronize()

    def _merge(self, udfs, damage, result_iter, partial, result_where=None):
        plans = self._merge_plans(udfs)

        # Delivery of 'disjoint' nav buffers: every row goes to its final place in page-locked host
        # memory while the run is still computing -- small write-once rows are written there by the
        # kernels themselves (zero-copy over the host link: no device buffer, no D2H, no copy
        # stream), wide rows are copied out tile by tile on a copy stream while later tiles compute.
        # Single rank: a buffer of the executor's pinned ring.  Several ranks on one node: a host
        # segment shared by the ranks, every rank delivers ITS rows (no data-path collective,
        # executor/nodeshared.py).
        from libertem_amd import hip as _hip
        streamed = {}                           # (udf index, name) -> [host tensor, rows copied out]
        expected = {}                           # rows this rank has to deliver
        direct = {}                             # (udf index, name) -> rows the kernels wrote directly
        host_np = {}                            # (udf index, name) -> NumPy view of the final buffer
        dev_ptrs = {}
        # result_where='device' (Context.run_udf): the declared buffers stay in HBM as HipArrays
        # (for a follow-up computation on the device); nothing is delivered to the host
        dev_mode = result_where == 'device'
        if dev_mode and (self.gpu_id is None or partial or self._collectives_on):
            raise NotImplementedError(
                "result_where='device' needs a single-rank HIP executor and run_udf (not run_udf_iter)")
        shared = self._node_shared() if not partial and not dev_mode else None
        busy_shared = None
        self._row_sink = None
        self._result_target = None
        sink_on = self.gpu_id is not None and not partial and not dev_mode and \
            (shared is not None or not self._collectives_on)
        layout, total = [], 0
        if sink_on:
            for i, (udf, (mode, decl)) in enumerate(zip(udfs, plans)):
                if mode != 'device':
                    continue
                for name, how in decl.items():
                    buf = udf.results.get_buffer(name)
                    if how != 'disjoint' or isinstance(buf, PlaceholderBufferWrapper):
                        continue
                    dt = np.dtype(buf.dtype)
                    nb = math.prod(buf.shape) * dt.itemsize
                    layout.append((i, name, tuple(buf.shape), dt, total, nb))
                    total += (nb + 4095) // 4096 * 4096
        tens = arr = base_dev = None
        if layout and shared is not None:
            from .nodeshared import NodeSharedUnavailable, NodeSharedBusy
            try:
                ahead = getattr(self, '_ahead', None)
                if ahead is not None:
                    # the run's launches were enqueued ahead into THIS slot (launch_ahead)
                    self._ahead = None
                    if ahead['total'] != total or ahead['mode'] != 'shared' or partial:
                        self.drain()
                        raise _hip.ReplayMismatch("the result layout changed since the launches were recorded")
                    tens, arr, base_dev = ahead['slot']
                else:
                    _, tens, arr, base_dev = shared.begin_run(total)
            except NodeSharedBusy:
                # this run only: device collectives (the end-of-run barrier still runs, it
                # refreshes the set of free slots)
                busy_shared, shared, layout = shared, None, []
            except NodeSharedUnavailable as e:
                self._shared_off = True
                import logging
                logging.getLogger(__name__).warning("%s -- results go through RCCL", e)
                shared, layout = None, []
            if shared is None:
                sink_on = False
        elif layout:
            if getattr(self, '_pinned_ring', None) is None:
                from .pinned import PinnedRing
                from libertem_amd import hip as _hip
                self._pinned_ring = PinnedRing(
                    self._torch, lambda ptr: _hip.host_device_pointer(self.gpu_id, ptr))
            ahead = getattr(self, '_ahead', None)
            if ahead is not None:
                # the run's launches were enqueued ahead into THIS buffer (launch_ahead)
                self._ahead = None
                if ahead['total'] != total or ahead['mode'] != 'ring' or partial:
                    self.drain()
                    raise _hip.ReplayMismatch("the result layout changed since the launches were recorded")
                tens, arr, base_dev = ahead['slot']
            else:
                tens, arr, base_dev = self._pinned_ring.get(total)
        else:
            shared = None
        if getattr(self, '_ahead', None) is not None:
            self._ahead = None
            self.drain()
            raise _hip.ReplayMismatch("launches were enqueued ahead for a run without a host result buffer")
        if layout:
            DIRECT_ROW_MAX = _udf_common.HIP_DIRECT_ROW_MAX
            if DIRECT_ROW_MAX <= 0:
                base_dev = None
            for i, name, shape, dt, off, nb in layout:
                # [torch view of the buffer (made when the first rows are copied out), rows copied]
                streamed[(i, name)] = [None, 0, off, nb, shape, dt]
                # views of the run's owner object: they keep the buffer reserved (on every rank)
                # for as long as the caller references any of them
                host_np[(i, name)] = arr[off:off + nb].view(np.ndarray).view(
                    _STORAGE_DTYPE.get(dt, dt)).reshape(shape)
                expected[(i, name)] = 0
                row_bytes = nb // max(1, shape[0])
                if base_dev is not None and row_bytes <= DIRECT_ROW_MAX:
                    direct[(i, name)] = 0
                    dev_ptrs[(i, name)] = base_dev + off
            del arr
        self.last_result_via = 'shm' if shared is not None else \
            ('collective' if self._collectives_on else 'local')
        keepalive = []                          # device rows with a D2H in flight on the copy stream
        if sink_on and layout:
            import torch as _torch
            if getattr(self, '_copy_stream', None) is None:
                self._copy_stream = _torch.cuda.Stream(device=self.gpu_id)
            copy_stream = self._copy_stream

            def row_sink(i, name, rows, g0):
                key = (i, name)
                if key not in expected or key in direct or rows.shape[0] == 0 \
                        or not rows.is_contiguous:
                    return
                buf = udfs[i].results.get_buffer(name)
                ent = streamed[key]
                if ent[0] is None:
                    ent[0] = tens[ent[2]:ent[2] + ent[3]].view(torch_dtype_for(ent[5])).reshape(ent[4])
                host = ent[0]
                n = rows.shape[0]
                inner = int(np.prod(buf.shape[1:])) if len(buf.shape) > 1 else 1
                src = rows.torch.reshape(-1)[:n * inner].reshape((n,) + tuple(buf.shape[1:]))
                ev = _torch.cuda.Event()
                ev.record(self._stream)
                with _torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(ev)
                    host[g0:g0 + n].copy_(src, non_blocking=True)
                # the source block was allocated on the executor stream: keep it referenced until
                # the copy stream has been synchronised, or the caching allocator may hand it to the
                # next partition while the D2H still reads it
                keepalive.append(src)
                streamed[key][1] += n

            def result_target(i, name, g0, shape, dtype):
                key = (i, name)
                if key not in direct:
                    return None
                full = host_np[key]
                if np.dtype(dtype) != udfs[i].results.get_buffer(name).dtype or \
                        tuple(shape[1:]) != tuple(full.shape[1:]) or g0 + shape[0] > full.shape[0]:
                    return None
                row_bytes = full.strides[0] if full.ndim else full.itemsize
                return HostMappedArray(full[g0:g0 + shape[0]], dev_ptrs[key] + g0 * row_bytes,
                                       self.gpu_id, shape, dtype)
            self._row_sink = row_sink if any(k not in direct for k in expected) else None
            self._result_target = result_target if direct else None

        dev_full = [dict() for _ in udfs]       # per udf: name -> torch tensor (full size)
        if dev_mode:
            import torch as _torch

            def device_target(i, name, g0, shape, dtype):
                # write-once nav rows of a partition go straight into the full-size device buffer
                buf = udfs[i].results.get_buffer(name)
                if np.dtype(dtype) != buf.dtype or tuple(shape[1:]) != tuple(buf.shape[1:]) or \
                        g0 + shape[0] > buf.shape[0]:
                    return None
                t = dev_full[i].get(name)
                if t is None:
                    self._make_current()
                    t = _torch.zeros(tuple(buf.shape), dtype=torch_dtype_for(buf.dtype),
                                     device=f'cuda:{self.gpu_id}')
                    dev_full[i][name] = t
                return HipArray(t[g0:g0 + shape[0]], shape, dtype)
            self._result_target = device_target
        deferred = {}                           # shared mode: (udf idx, name) -> [(start, stop, rows)]
        direct_rows = {}                        # (udf idx, name) -> [(start, stop)] written by kernels
        generic_parts = []                      # (task, {udf idx: exported results})
        torch = None
        if self.gpu_id is not None:
            import torch
        d = self._dist()

        idle = [False]                          # executor stream known to be drained

        def delivered(key):
            return streamed[key][1] + direct.get(key, 0)

        def publish_device(final):
            """declared device buffers -> host arrays of the main-process udfs"""
            shared_ok = False
            if sink_on and layout and final and shared is None and busy_shared is None and direct \
                    and not keepalive and not deferred:
                # Every declared buffer written by the kernels straight into its final place?  Then the
                # arrays can be handed over and the caller's result objects built (`set_before_final`)
                # BEHIND the kernels; the wait for the stream comes last.
                todo = []
                for i, (udf, (mode, decl)) in enumerate(zip(udfs, plans)):
                    if mode != 'device':
                        todo = None
                        break
                    for name in decl:
                        buf = udf.results.get_buffer(name)
                        if isinstance(buf, PlaceholderBufferWrapper):
                            continue
                        key = (i, name)
                        if key in host_np and key in direct and \
                                delivered(key) == buf.shape[0] == expected.get(key):
                            todo.append((buf, key))
                        else:
                            todo = None
                            break
                    if todo is None:
                        break
                if todo:
                    for buf, key in todo:
                        host = host_np[key]
                        buf.replace_array(host if host.dtype == buf.dtype else host.view(buf.dtype))
                    hook2, self._before_final = self._before_final, None
                    try:
                        if hook2 is not None:
                            hook2()
                    finally:
                        self._stream.synchronize()
                        idle[0] = True
                    host_np.clear()
                    streamed.clear()
                    return
            if sink_on and layout and final:
                # my rows are out once the copy stream (copied rows) and the executor stream (rows
                # the kernels wrote into the host buffer themselves) are idle
                if keepalive:
                    self._copy_stream.synchronize()
                    keepalive.clear()
                if direct:
                    self._stream.synchronize()
                    idle[0] = True
            if shared is not None and final:
                # every rank of the node says whether all of ITS rows were delivered
                # (same answer on every rank)
                mine = all(delivered(k) == expected[k] for k in expected)
                shared_ok = shared.all_ok(mine)
            elif busy_shared is not None and final:
                busy_shared.all_ok(True)
            for i, (udf, (mode, decl)) in enumerate(zip(udfs, plans)):
                if mode != 'device':
                    continue
                for name, how in decl.items():
                    buf = udf.results.get_buffer(name)
                    if isinstance(buf, PlaceholderBufferWrapper):
                        continue
                    key = (i, name)
                    if key in host_np and final and (
                            shared_ok if shared is not None else
                            (delivered(key) == buf.shape[0] == expected.get(key))):
                        # every row is in its final place already
                        host = host_np[key]
                        if host.dtype != buf.dtype:
                            host = host.view(buf.dtype)
                        buf.replace_array(host)
                        deferred.pop(key, None)
                        continue
                    idle[0] = False
                    self._flush_deferred(udf, i, name, dev_full[i], deferred, keep=not final)
                    if direct_rows.get(key) and key in host_np:
                        # rows the kernels wrote straight into the host buffer exist nowhere on the
                        # device: bring THIS rank's rows back before the buffers are combined (the
                        # streamed delivery was called off, e.g. another buffer of the run was not
                        # fully delivered on some rank)
                        self._make_current()
                        if name not in dev_full[i]:
                            dev_full[i][name] = torch.zeros(
                                buf.shape, dtype=torch_dtype_for(buf.dtype),
                                device=f'cuda:{self.gpu_id}')
                        src = host_np[key]
                        for start, stop in direct_rows.pop(key):
                            dev_full[i][name][start:stop].copy_(
                                torch.from_numpy(np.array(src[start:stop], copy=True)).view(
                                    dev_full[i][name].dtype).reshape(
                                        dev_full[i][name][start:stop].shape))
                    full = dev_full[i].get(name)
                    self._make_current()
                    if full is None:
                        full = torch.zeros(buf.shape, dtype=torch_dtype_for(buf.dtype),
                                           device=f'cuda:{self.gpu_id}')
                    if dev_mode:
                        buf.replace_array(HipArray(full, tuple(buf.shape), buf.dtype))
                        continue
                    if final and self._collectives_on:
                        # collectives are ordered after the kernels of the executor stream
                        full = self._combine(d, full, how)
                    host = self._to_host(full)
                    if host.dtype != buf.dtype:
                        host = host.view(buf.dtype)
                    buf.replace_array(host)
            if final:
                host_np.clear()
                streamed.clear()

        n_done = 0
        for part_results, task in result_iter:
            gen_entry = {}
            for i, (udf, results, (mode, decl)) in enumerate(zip(udfs, part_results, plans)):
                if mode == 'device':
                    self._merge_on_device(udf, results, task, decl, dev_full[i],
                                          may_adopt=not partial,
                                          defer=(deferred, i, expected, direct, direct_rows)
                                          if sink_on and layout else None)
                else:
                    results.export()
                    gen_entry[i] = results
            damage.get_view_for_partition(task.partition)[:] = True
            n_done += 1
            rec = (getattr(task, '_keep', None) or {}).pop('recorded', None)
            if rec is not None and len(rec) == 1 and not rec[0][1][7] and len(udfs) == 1:
                # one launch, not accumulating, into rows of a directly written buffer: launch-ahead material
                handle, sig = rec[0]
                for key, base in dev_ptrs.items():
                    nb = streamed[key][3]
                    if base <= sig[5] < base + nb:
                        # (offset inside the run's buffer -- page-locked ring / node-shared slot -- and its size)
                        task._keep['replay'] = (handle, sig, streamed[key][2] + sig[5] - base, total,
                                                'ring' if shared is None else 'shared')
            if partial:
                # merge the host-side UDFs right away (tasks arrive in partition order here)
                for i, results in gen_entry.items():
                    self._apply_one(udfs[i], results, task, damage)
                publish_device(final=False)
                yield n_done
            elif gen_entry:
                generic_parts.append((task, gen_entry))
        if partial:
            if n_done == 0:
                publish_device(final=True)
                yield 0
            if self._stream is not None:
                self._stream.synchronize()
            return

        if _hip.LaunchReplay.expected is not None:
            left, _hip.LaunchReplay.expected = _hip.LaunchReplay.expected, None
            if left:
                self.drain()
                raise _hip.ReplayMismatch(f"{len(left)} launch(es) enqueued ahead were not made by the run")
        hook, self._before_wait = self._before_wait, None
        if hook is not None:
            try:
                hook()
            except BaseException:
                self.drain()
                raise
        # ---- declared buffers: collectives on flat tensors ----
        publish_device(final=True)
        for i, (udf, (mode, decl)) in enumerate(zip(udfs, plans)):
            if mode == 'host-declared':
                # CPU rank (gloo tests / NumPy UDFs that declare their merge): merge locally with
                # the UDF's own merge(), then combine the full-size host buffers
                for task, entry in generic_parts:
                    if i in entry:
                        self._apply_one(udf, entry[i], task, damage)
                if self._collectives_on:
                    import torch as _t
                    for name, how in decl.items():
                        buf = udf.results.get_buffer(name)
                        if isinstance(buf, PlaceholderBufferWrapper):
                            continue
                        t = _t.from_numpy(np.ascontiguousarray(buf.raw_data))
                        t = self._combine(d, t, how)
                        buf.replace_array(t.numpy())
        # ---- generic UDFs: ship exported partition results, merge in partition order ----
        gen_idx = [i for i, (mode, _) in enumerate(plans) if mode == 'generic']
        if gen_idx:
            mine = [(task.idx, {i: entry[i] for i in gen_idx if i in entry})
                    for task, entry in generic_parts]
            if self._collectives_on:
                gathered = [None] * self.world_size
                d.all_gather_object(gathered, mine)
                allparts = [p for chunk in gathered for p in chunk]
            else:
                allparts = mine
            by_idx = {t.idx: t for t in self._all_tasks}
            for tidx, entry in sorted(allparts, key=lambda x: x[0]):
                task = by_idx[tidx]
                for i in gen_idx:
                    if i in entry:
                        self._apply_one(udfs[i], entry[i], task, damage)
        # damage: with sharding every partition was processed by some rank
        if self._collectives_on:
            for task in self._all_tasks:
                damage.get_view_for_partition(task.partition)[:] = True
        if self._stream is not None and not idle[0]:
            self._stream.synchronize()
        self._row_sink = None
        self._result_target = None
        self._before_final = None
        yield n_done

    def _to_host(self, t):
        """ONE D2H per buffer and run into page-locked memory, on the executor stream.  A fresh
        pinned tensor per result (torch's caching host allocator recycles the blocks), so the
        returned array is owned by the caller and no host-side copy is needed."""
        import torch
        pinned = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        self._make_current()
        pinned.copy_(t, non_blocking=True)
        self._stream.synchronize()
        return pinned.numpy()

    @staticmethod
    def _apply_one(udf, results, task, damage=None):
        if damage is not None:
            # what is merged SO FAR (udf/base.py:2351) -- some callers mark the task's rows before merging them
            seen = np.array(damage.raw_data, copy=True)
            a, b = damage._slice_for_partition(task.partition)
            seen[a:b] = False
            udf.meta.set_valid_nav_mask(seen)
        udf.set_views_for_partition(task.partition)
        udf.merge(dest=udf.results.get_proxy(), src=results.get_proxy())
        udf.clear_views()

    def _flush_deferred(self, udf, i, name, full, deferred, keep=False):
        """Rows that were (or could have been) streamed to the host were not merged on the device;
        if the streamed delivery did not cover the buffer, build the full-size device buffer from
        the partition results kept aside."""
        import torch
        items = deferred.get((i, name)) if keep else deferred.pop((i, name), None)
        if not items:
            return
        buf_main = udf.results.get_buffer(name)
        self._make_current()
        if name not in full:
            full[name] = torch.zeros(buf_main.shape, dtype=torch_dtype_for(buf_main.dtype),
                                     device=f'cuda:{self.gpu_id}')
        for start, stop, pt in items:
            full[name][start:stop].copy_(pt.reshape(full[name][start:stop].shape))

    def _merge_on_device(self, udf, results, task, decl, full, may_adopt=True, defer=None):
        import torch
        from libertem_amd import hip
        self._make_current()
        for name, how in decl.items():
            buf_main = udf.results.get_buffer(name)
            if isinstance(buf_main, PlaceholderBufferWrapper):
                continue
            part = results.get_buffer(name)._data
            if not isinstance(part, HipArray):
                part = HipArray.from_numpy(np.asarray(part), self.gpu_id)
            if defer is not None and how == 'disjoint' and (defer[1], name) in defer[2]:
                start, stop = buf_main._slice_for_partition(task.partition)
                defer[2][(defer[1], name)] += stop - start
                if isinstance(part, HostMappedArray):
                    # the kernels wrote these rows into the final host buffer themselves
                    defer[3][(defer[1], name)] += stop - start
                    if len(defer) > 4:
                        defer[4].setdefault((defer[1], name), []).append((start, stop))
                    continue
                # rows travel to the host on the copy stream (shared host segment / pinned
                # buffer); keep the partition result only as the fallback source
                defer[0].setdefault((defer[1], name), []).append(
                    (start, stop, part.torch.reshape(part.shape)))
                continue
            if isinstance(part, HostMappedArray):
                raise RuntimeError(f"host-mapped result rows of {name!r} outside a streamed delivery")
            if name not in full:
                if may_adopt and tuple(part.shape) == tuple(buf_main.shape) and \
                        (how == 'disjoint' or how == 'sum'):
                    # first partition covers the whole buffer (always true for 'sum' buffers):
                    # adopt it, no zero-fill, no copy / add
                    full[name] = part.torch.reshape(part.shape)
                    continue
                full[name] = torch.zeros(buf_main.shape, dtype=torch_dtype_for(buf_main.dtype),
                                         device=f'cuda:{self.gpu_id}')
            if how == 'disjoint':
                start, stop = buf_main._slice_for_partition(task.partition)
                if stop > start and part.data_ptr() == full[name][start:stop].data_ptr():
                    continue            # the kernels wrote the rows into the full buffer (device_target)
                full[name][start:stop].copy_(
                    part.torch.reshape(part.shape).reshape(full[name][start:stop].shape))
            elif how == 'sum':
                dst = full[name]
                pt = part.torch.reshape(part.shape)
                if np.dtype(buf_main.dtype) in hip.AXPY_DTYPES and dst.is_contiguous() \
                        and pt.is_contiguous() and np.dtype(buf_main.dtype) != np.bool_:
                    # (bool: `+=` is a logical OR in NumPy and in the reference's merge; adding the
                    # bytes would leave 2 in a bool -- torch's in-place add below keeps the OR)
                    # dest += src in HBM with the library's own kernel (ltmi_axpy)
                    hip.axpy(self.gpu_id, dst.data_ptr(), pt.data_ptr(), buf_main.dtype,
                             dst.numel(), stream=self._stream_ptr)
                else:
                    dst += pt.reshape(dst.shape)
            else:
                raise ValueError(f"unknown dist merge {how!r} for buffer {name!r}")

    def _rccl(self, d):
        """The library's own RCCL communicator (`ltmi_comm_*`, csrc/ltmi_comm.cpp) over the ranks of the
        job, or None: created once -- rank 0 draws the unique id, the control plane (torch.distributed's
        store / object broadcast) carries its 128 bytes to the others, then ncclCommInitRank on every
        rank -- and only for device tensors on the "nccl" backend (gloo test ranks share one GPU, which
        RCCL refuses).  Whether everybody has a communicator is agreed on collectively, so all ranks
        take the same path.  LTMI_COMM=torch keeps the data plane in torch.distributed."""
        if getattr(self, '_comm_state', None) is not None:
            return self._comm_state or None
        self._comm_state = False
        if self.gpu_id is None or os.environ.get('LTMI_COMM', 'ltmi') == 'torch' \
                or d.get_backend() != 'nccl':
            return None
        import torch
        from libertem_amd import hip
        comm, err = None, None
        try:
            box = [hip.Comm.unique_id() if self.rank == 0 else None]
            d.broadcast_object_list(box, src=0)
            self._make_current()
            comm = hip.Comm(self.gpu_id, self.rank, self.world_size, box[0])
        except Exception as e:                       # noqa: BLE001  (decided collectively below)
            err = e
        ok = torch.tensor([0 if comm is None else 1], dtype=torch.int32,
                          device=f'cuda:{self.gpu_id}')
        d.all_reduce(ok, op=d.ReduceOp.MIN)
        if int(ok.item()) != 1:
            import logging
            logging.getLogger(__name__).warning(
                "ltmi_comm could not be set up on every rank (%r here) -- results are combined "
                "through torch.distributed", err)
            if comm is not None:
                comm.close()
            return None
        self._comm_state = comm
        return comm

    def _combine(self, d, full, how):
        """all ranks: disjoint -> every rank's rows are zero outside its own partitions, so a SUM
        all-reduce is an exact concatenation (x + 0 == x); sum -> all-reduce.  Device tensors on the
        nccl backend go through the library's own communicator (`_rccl`): all-gather of equal row
        blocks / all-reduce(sum) on the executor's stream, torch only owns the buffers."""
        import torch
        comm = self._rccl(d) if full.is_cuda else None
        if comm is not None:
            self._make_current()
            self.last_collective = 'ltmi_comm'
            if full.dtype == torch.bool:
                # OR as a sum of 0 / 1 in int32: cannot wrap for any world size (uint8 would at 256 ranks
                # that all hold True); torch's path uses MAX
                t = full.to(torch.int32).contiguous()
                comm.all_reduce_sum(t.data_ptr(), np.int32, t.numel(), stream=self._stream_ptr)
                return t != 0
            if not full.is_contiguous():
                full = full.contiguous()
            if how == 'disjoint' and full.dim() >= 1 and full.shape[0] >= self.world_size \
                    and self._equal_rows(full.shape[0]):
                W, r = self.world_size, self.rank
                rows = full.shape[0] // W
                out = torch.empty_like(full)
                nbytes = rows * (full.numel() // full.shape[0]) * full.element_size()
                comm.all_gather(full[r * rows:(r + 1) * rows].data_ptr(), out.data_ptr(), nbytes,
                                stream=self._stream_ptr)
                return out
            np_dtype = _NP_OF_TORCH.get(full.dtype)
            if np_dtype is not None:
                comm.all_reduce_sum(full.data_ptr(), np_dtype, full.numel(), stream=self._stream_ptr)
                return full
            # (16-bit integers: RCCL has no sum for them) -> torch below
        self.last_collective = 'torch.distributed'
        if full.dtype == torch.bool:
            t = full.to(torch.uint8)
            d.all_reduce(t, op=d.ReduceOp.MAX)
            return t.to(torch.bool)
        if how == 'disjoint' and full.dim() >= 1 and full.shape[0] >= self.world_size \
                and d.get_backend() == 'nccl' and self._equal_rows(full.shape[0]):
            # equal contiguous shards: cheaper all_gather of 1/W of the buffer per rank
            W, r = self.world_size, self.rank
            rows = full.shape[0] // W
            shard = full[r * rows:(r + 1) * rows].contiguous()
            out = torch.empty_like(full)
            d.all_gather_into_tensor(out.reshape(-1), shard.reshape(-1))
            return out
        d.all_reduce(full, op=d.ReduceOp.SUM)
        return full

    def _equal_rows(self, n_rows):
        """True iff every rank owns exactly rows [r*n/W, (r+1)*n/W) of a nav buffer."""
        W = self.world_size
        tasks = getattr(self, '_all_tasks', None)
        if not tasks or n_rows % W != 0:
            return False
        if n_rows != sum(t.partition.slice.shape[0] for t in tasks):
            return False                 # ROI-compressed buffer or skipped partitions
        rows = n_rows // W
        owners = self.task_owners(tasks)
        for r in range(W):
            mine = [t for t, o in zip(tasks, owners) if o == r]
            if not mine:
                return False
            start = mine[0].partition.slice.origin[0]
            stop = mine[-1].partition.slice.origin[0] + mine[-1].partition.slice.shape[0]
            if start != r * rows or stop != (r + 1) * rows:
                return False
        return True
