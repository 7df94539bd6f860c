"""
InlineJobExecutor: run tasks one after another in the calling thread on the CPU -- the executor
the reference names as its parity baseline (executor/inline.py:18-143).  `debug=True`
round-trips tasks and results through cloudpickle like the reference does (:106-111).

This executor has device class 'cpu': the native (HIP-only) operators refuse to run on it.
"""
import uuid

import psutil

from .base import JobExecutor, Environment


_CORES = []


def _physical_cores():
    if not _CORES:
        _CORES.append(psutil.cpu_count(logical=False) or 1)
    return _CORES[0]


class InlineJobExecutor(JobExecutor):
    device_class = 'cpu'

    def __init__(self, debug=False, inline_threads=None, *args, **kwargs):
        self._debug = debug
        self._inline_threads = inline_threads
        self._scattered = {}

    def sibling(self):
        """an independent executor of the same kind: runs a UDF while a `run_udf_iter` of this one is suspended between
        two partial results (Context.run_udf)"""
        return InlineJobExecutor(debug=self._debug, inline_threads=self._inline_threads)

    def get_local_env(self):
        threads = self._inline_threads
        if threads is None:
            threads = _physical_cores()
        return Environment(threads_per_worker=threads, threaded_executor=False, gpu_id=None)

    def scatter(self, obj):
        handle = str(uuid.uuid4())
        self._scattered[handle] = obj
        return handle

    def scatter_release(self, handle):
        self._scattered.pop(handle, None)

    def run_tasks(self, tasks, params_handle, cancel_id, task_comm_handler=None):
        params = self._scattered[params_handle]
        env = self.get_local_env()
        for task in tasks:
            if self._debug:
                import cloudpickle
                cloudpickle.loads(cloudpickle.dumps(task))
            result = task(env=env, params=params)
            if self._debug:
                import cloudpickle
                cloudpickle.loads(cloudpickle.dumps(result))
            yield result, task

    def run_function(self, fn, *args, **kwargs):
        return fn(*args, **kwargs)

    def get_available_workers(self):
        from libertem_amd.executor.workers import Worker, WorkerSet
        return WorkerSet([Worker(name='inline', host='localhost',
                                 resources={'CPU': 1, 'compute': 1, 'ndarray': 1}, nthreads=1)])
