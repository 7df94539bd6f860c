"""
Executor protocol: `Environment`, `JobExecutor` (subset of the reference's common/executor.py:
52-432 and executor/base.py).
"""
import contextlib


_NULL_CONTEXT = contextlib.nullcontext()
_THREADPOOL_CTL = [False]          # False: not looked up yet; None: threadpoolctl unavailable


def _threadpool_controller():
    """One ThreadpoolController for the process: discovering the loaded BLAS / OpenMP libraries
    costs ~1 ms, limiting them through a known controller a few microseconds."""
    if _THREADPOOL_CTL[0] is False:
        try:
            from threadpoolctl import ThreadpoolController
            _THREADPOOL_CTL[0] = ThreadpoolController()
        except Exception:
            _THREADPOOL_CTL[0] = None
    return _THREADPOOL_CTL[0]


class Environment:
    """
    What a task may assume about its worker (common/executor.py:52-140): thread budget and the
    device it drives.  `device_class` is 'hip' iff `gpu_id` is not None.
    """

    def __init__(self, threads_per_worker=None, threaded_executor=False, gpu_id=None,
                 keep_results_on_device=False, stream=None, ensure_current=None, row_sink=None,
                 result_target=None):
        self._threads_per_worker = threads_per_worker
        self._threaded_executor = threaded_executor
        self._gpu_id = gpu_id
        self.keep_results_on_device = keep_results_on_device
        self.stream = stream
        # hipStream_t value handed to libltmi (None: torch's current stream at call time)
        self.stream_ptr = None if stream is None else int(stream.cuda_stream)
        # callable that makes (gpu_id, stream) torch's current device/stream WITHOUT a context
        # manager (the HIP executor owns the process's device: one process per GPU)
        self._ensure_current = ensure_current
        # callable(udf_index, buffer name, rows: HipArray, global_row_start) or None: finished rows
        # of disjoint nav buffers are copied to the host while later tiles are still computing
        self.row_sink = row_sink
        # callable(udf_index, buffer name, global_row_start, shape, dtype) -> HipArray | None: rows of
        # the run's final host buffer that the kernels of a partition may write directly
        self.result_target = result_target

    @property
    def threads_per_worker(self):
        return self._threads_per_worker

    @property
    def threaded_executor(self):
        return self._threaded_executor

    @property
    def gpu_id(self):
        return self._gpu_id

    @property
    def device_class(self):
        return 'hip' if self._gpu_id is not None else 'cpu'

    def enter(self, enable_gpu=False):
        """Select the device and limit BLAS threads for the duration of a task
        (common/executor.py:111-129)."""
        if self._threads_per_worker is None and (
                not enable_gpu or self._gpu_id is None or self._ensure_current is not None):
            if enable_gpu and self._ensure_current is not None:
                self._ensure_current()
            return _NULL_CONTEXT
        return self._enter_ctx(enable_gpu)

    @contextlib.contextmanager
    def _enter_ctx(self, enable_gpu):
        ctxs = contextlib.ExitStack()
        with ctxs:
            if self._threads_per_worker is not None:
                ctl = _threadpool_controller()
                if ctl is not None:
                    ctxs.enter_context(ctl.limit(limits=self._threads_per_worker))
            if enable_gpu and self._gpu_id is not None:
                import torch
                ctxs.enter_context(torch.cuda.device(self._gpu_id))
                if self.stream is not None:
                    ctxs.enter_context(torch.cuda.stream(self.stream))
            yield


class JobExecutor:
    device_class = 'cpu'
    gpu_id = None

    @property
    def run_gate(self):
        """one run at a time per executor (Context.run_udf / run_udf_iter, sync and async alike): the delivery
        targets, launch-ahead state and streams of an executor belong to the run in progress"""
        gate = self.__dict__.get('_run_gate')
        if gate is None:
            from libertem_amd.hip import RunGate
            gate = self.__dict__.setdefault('_run_gate', RunGate())
        return gate

    @property
    def replay(self):
        """launch-ahead state of this executor (hip.LaunchReplay binds it to the thread of the run)"""
        st = self.__dict__.get('_replay_state')
        if st is None:
            from libertem_amd.hip import ReplayState
            st = self.__dict__.setdefault('_replay_state', ReplayState())
        return st

    def run_tasks(self, tasks, params_handle, cancel_id, task_comm_handler=None):
        raise NotImplementedError()

    def run_function(self, fn, *args, **kwargs):
        raise NotImplementedError()

    def map(self, fn, iterable):
        return [fn(x) for x in iterable]

    def run_each_host(self, fn, *args, **kwargs):
        return {"localhost": fn(*args, **kwargs)}

    def run_each_worker(self, fn, *args, **kwargs):
        return {"inline": fn(*args, **kwargs)}

    def run_process_local(self, task, args=(), kwargs=None):
        return task(*args, **(kwargs or {}))

    def scatter(self, obj):
        raise NotImplementedError()

    def scatter_release(self, handle):
        pass

    def get_available_workers(self):
        raise NotImplementedError()

    def get_local_env(self):
        raise NotImplementedError()

    def close(self):
        pass

    def ensure_sync(self):
        return self

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
