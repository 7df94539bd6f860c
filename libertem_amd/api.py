"""
`Context`: the user-facing entry point.

Mirrors the part of the reference's libertem.api.Context (api.py:177-1720) that sits on top of the
mask / sum / CoM / radial-Fourier path: `load`, `run_udf`, `run_udf_iter`, `run`, `map`,
`create_mask_analysis`, `create_com_analysis`, `create_radial_fourier_analysis`,
`create_sum_analysis`, `create_disk_analysis`, `create_ring_analysis`, `create_point_analysis`
with the same argument names and return types (`dict[str, BufferWrapper]` for one UDF, a tuple of
dicts for a list, api.py:1330-1334).

Executors: 'hip' (default when a GPU is visible: one MI355X per process, sharded over ranks if
torch.distributed is initialised) and 'inline' (CPU plumbing for user UDFs written against NumPy).
The reference's dask / pipelined / delayed executors are out of scope.
"""
import numpy as np

from libertem_amd.common.exceptions import ExecutorSpecException
from libertem_amd.io.dataset import load as _load_dataset
from libertem_amd.udf.base import UDFRunner
from libertem_amd.analysis import (
    MasksAnalysis, COMAnalysis, RadialFourierAnalysis, SumAnalysis, DiskMaskAnalysis,
    RingMaskAnalysis, PointMaskAnalysis,
)


def _gpu_available():
    try:
        from libertem_amd import hip
        return hip.device_count() > 0
    except Exception:
        return False


class Context:
    def __init__(self, executor=None, plot_class=None):
        if executor is None:
            executor = self._create_local_executor()
        self.executor = executor

    @classmethod
    def make_with(cls, executor_spec=None, *args, cpus=None, gpus=None, plot_class=None,
                  **kwargs):
        """
        executor_spec : 'hip' | 'inline' | None (= 'hip' when a GPU is visible, else 'inline').
        gpus : for 'hip': int or iterable with ONE device ordinal (one GPU per process; use
            torchrun + torch.distributed for several GPUs).
        """
        if executor_spec is None:
            executor_spec = 'hip' if _gpu_available() else 'inline'
        if executor_spec == 'inline':
            # (the inline executor has neither worker processes nor GPUs to size: api.py make_with of the reference
            #  refuses `cpus` / `gpus` for it)
            if cpus is not None or gpus is not None:
                raise ExecutorSpecException("the 'inline' executor takes neither cpus nor gpus")
            from libertem_amd.executor.inline import InlineJobExecutor
            return cls(executor=InlineJobExecutor(*args, **kwargs))
        if executor_spec == 'hip':
            from libertem_amd.executor.hip import HipJobExecutor
            gpu_id = None
            if gpus is not None:
                ids = [gpus] if isinstance(gpus, int) else list(gpus)
                if len(ids) != 1:
                    raise ExecutorSpecException(
                        "the 'hip' executor drives one GPU per process; launch one process per "
                        "GPU (torchrun) and initialise torch.distributed for multi-GPU runs")
                gpu_id = ids[0]
            return cls(executor=HipJobExecutor(gpu_id=gpu_id, **kwargs))
        raise ExecutorSpecException(
            f"executor_spec {executor_spec!r} is not available in libertem_amd "
            "(available: 'hip', 'inline')")

    def _create_local_executor(self):
        if _gpu_available():
            from libertem_amd.executor.hip import HipJobExecutor
            return HipJobExecutor()
        from libertem_amd.executor.inline import InlineJobExecutor
        return InlineJobExecutor()

    # --- datasets ------------------------------------------------------------------------------
    def load(self, filetype, *args, io_backend=None, **kwargs):
        ds = _load_dataset(filetype, *args, **kwargs)
        ds = ds.initialize(self.executor)
        loaded = self.__dict__.get('_loaded')
        if loaded is None:
            import weakref
            loaded = self.__dict__['_loaded'] = weakref.WeakSet()
        try:
            loaded.add(ds)                  # (close(): the datasets' upload stagers go with the context)
        except TypeError:
            pass
        return ds

    # --- analyses (reference api.py:514-811) -----------------------------------------------------
    def create_mask_analysis(self, factories, dataset, use_sparse=None, mask_count=None,
                             mask_dtype=None, dtype=None):
        return MasksAnalysis(dataset=dataset, parameters={
            "factories": factories, "use_sparse": use_sparse, "mask_count": mask_count,
            "mask_dtype": mask_dtype, "dtype": dtype})

    def create_com_analysis(self, dataset, cx=None, cy=None, mask_radius=None, flip_y=False,
                            mask_radius_inner=None, scan_rotation=0.0):
        if dataset.shape.nav.dims != 2:
            raise ValueError("incompatible dataset: need two navigation dimensions")
        if dataset.shape.sig.dims != 2:
            raise ValueError("incompatible dataset: need two signal dimensions")
        loc = locals()
        parameters = {name: loc[name] for name in ['cx', 'cy', 'flip_y', 'scan_rotation']
                      if loc[name] is not None}
        if mask_radius is not None:
            parameters['r'] = mask_radius
        if mask_radius_inner is not None:
            if mask_radius is None:
                raise ValueError("incompatible parameters: must pass both `mask_radius` and "
                                 "`mask_radius_inner` for annular CoM")
            parameters['ri'] = mask_radius_inner
        return COMAnalysis(dataset=dataset, parameters=parameters)

    def create_radial_fourier_analysis(self, dataset, cx=None, cy=None, ri=None, ro=None,
                                       n_bins=None, max_order=None, use_sparse=None):
        if dataset.shape.sig.dims != 2:
            raise ValueError("incompatible dataset: need two signal dimensions")
        loc = locals()
        parameters = {name: loc[name]
                      for name in ['cx', 'cy', 'ri', 'ro', 'n_bins', 'max_order', 'use_sparse']
                      if loc[name] is not None}
        return RadialFourierAnalysis(dataset=dataset, parameters=parameters)

    def create_disk_analysis(self, dataset, cx=None, cy=None, r=None):
        if dataset.shape.sig.dims != 2:
            raise ValueError("incompatible dataset: need two signal dimensions")
        loc = locals()
        return DiskMaskAnalysis(dataset=dataset, parameters={
            name: loc[name] for name in ['cx', 'cy', 'r'] if loc[name] is not None})

    def create_ring_analysis(self, dataset, cx=None, cy=None, ri=None, ro=None):
        if dataset.shape.sig.dims != 2:
            raise ValueError("incompatible dataset: need two signal dimensions")
        loc = locals()
        return RingMaskAnalysis(dataset=dataset, parameters={
            name: loc[name] for name in ['cx', 'cy', 'ri', 'ro'] if loc[name] is not None})

    def create_point_analysis(self, dataset, x=None, y=None):
        if dataset.shape.nav.dims > 2:
            raise ValueError("incompatible dataset: need at most two navigation dimensions")
        parameters = {k: v for k, v in {'cx': x, 'cy': y}.items() if v is not None}
        return PointMaskAnalysis(dataset=dataset, parameters=parameters)

    def create_pick_analysis(self, dataset, x, y=None, z=None):
        """Pick the frame at (z, y, x) -- as many coordinates as nav dimensions (api.py:813-850)."""
        from libertem_amd.analysis.raw import PickFrameAnalysis
        parameters = {'x': x}
        if y is not None:
            parameters['y'] = y
        if z is not None:
            parameters['z'] = z
        return PickFrameAnalysis(dataset=dataset, parameters=parameters)

    def create_sum_analysis(self, dataset):
        return SumAnalysis(dataset=dataset, parameters={})

    # --- running -------------------------------------------------------------------------------
    def run(self, job, roi=None, progress=False, corrections=None):
        """Run an Analysis and post-process its UDF results (reference api.py:854-912)."""
        analysis = job
        if roi is None:
            roi = analysis.get_roi()
        udf_results = self.run_udf(dataset=analysis.dataset, udf=analysis.get_udf(), roi=roi,
                                   corrections=corrections, progress=progress)
        damage = True if roi is None else np.asarray(roi, dtype=bool)
        return analysis.get_udf_results(udf_results, roi, damage=damage)

    def run_udf(self, dataset, udf, roi=None, corrections=None, progress=False, backends=None,
                plots=None, sync=True, result_where=None):
        """
        Run `udf` (a UDF or a list of UDFs) on `dataset`, restricted to `roi`
        (reference api.py:914-1051).  Returns dict[str, BufferWrapper] (tuple of dicts for a list).

        result_where='device' (not in the reference): the result buffers of device-merged UDFs stay in
        HBM -- `buffer.device_data` is the HipArray, `.data` / `.raw_data` download it when accessed.
        For results that feed a follow-up computation on the GPU and are large next to the host link
        (1024 masks x 65 536 frames of float32 are 256 MiB = 4.9 ms of PCIe for a 3.8 ms job).
        """
        if not sync:
            # (reference api.py:981-1051: a coroutine that yields the same result; here the synchronous run
            #  on the context's single worker thread -- runs of one context are serialised, the event loop
            #  stays free while the GPU works)
            return self._run_udf_async(dataset, udf, roi=roi, corrections=corrections, progress=progress,
                                       backends=backends, plots=plots, result_where=result_where)
        nested = self._nested_context()
        if nested is not None:
            return nested.run_udf(dataset, udf, roi=roi, corrections=corrections, progress=progress, backends=backends,
                                  plots=plots, sync=True, result_where=result_where)
        if result_where not in (None, 'host', 'device'):
            raise ValueError("result_where must be None, 'host' or 'device'")
        if result_where == 'device' and not hasattr(self.executor, '_merge_on_device'):
            raise NotImplementedError("result_where='device' needs the HIP executor")
        if corrections is not None and not corrections.have_corrections():
            corrections = None
        udf_is_list = isinstance(udf, (tuple, list))
        udfs = list(udf) if udf_is_list else [udf]
        if len(udfs) == 0:
            raise ValueError("empty list of UDFs - nothing to do!")
        if roi is not None:
            roi = self._normalize_roi(roi, dataset)
        runner = UDFRunner(udfs)
        # one run at a time per executor -- a sync run issued while an async one is in flight waits for it --
        # with the executor's launch-ahead state bound to this thread for the duration (hip.LaunchReplay)
        with self._run_scope():
            res = runner.run_for_dataset(dataset=dataset, executor=self.executor, roi=roi,
                                         progress=progress, corrections=corrections,
                                         backends=backends,
                                         **({'result_where': 'device'} if result_where == 'device' else {}))
        buffers = res.buffers
        return tuple(buffers) if udf_is_list else buffers[0]

    def _nested_context(self):
        """The reference runs UDFs from inside a `run_udf_iter` loop (tests/test_context.py test_udf_iter: a second run
        per partial result, on the same Context).  Here an executor's delivery buffers and launch-ahead state belong
        to the run in progress, so a run started BY THE THREAD THAT HOLDS A SUSPENDED ITERATION goes to a sibling
        executor (same kind, same GPU, state of its own), made on first use.  Other threads are refused (RunGate)."""
        import threading
        gate = self.executor.run_gate
        if not gate.suspended or gate._owner != threading.get_ident():
            return None
        sib = self.__dict__.get('_sibling')
        if sib is None:
            make = getattr(self.executor, 'sibling', None)
            if make is None:
                return None                                   # (the gate refuses with its message)
            sib = self._sibling = Context(executor=make())
        return sib

    def _run_scope(self):
        import contextlib
        from libertem_amd import hip as _hip
        stack = contextlib.ExitStack()
        stack.enter_context(self.executor.run_gate)
        stack.enter_context(_hip.LaunchReplay.bound(self.executor.replay))
        return stack

    def invalidate_caches(self):
        """Forget every cached mask stack / device handle and run plan of this process: the next run
        re-evaluates the mask factories and plans afresh, like every run of the reference does
        (common/container.py:260-314).  For factories whose output depends on something a content
        fingerprint cannot see (files, random state, objects without a __dict__)."""
        from libertem_amd.udf import masks as _masks
        from libertem_amd.udf import base as _base
        with self.executor.run_gate:
            _masks.invalidate_cache()
            _base.invalidate_plans()

    def _async_pool(self):
        pool = getattr(self, '_async_worker', None)
        if pool is None:
            from concurrent.futures import ThreadPoolExecutor
            pool = self._async_worker = ThreadPoolExecutor(1, thread_name_prefix='ltmi-async-run')
        return pool

    async def _run_udf_async(self, dataset, udf, **kwargs):
        import asyncio
        import functools
        loop = asyncio.get_running_loop()
        return await loop.run_in_executor(
            self._async_pool(), functools.partial(self.run_udf, dataset, udf, sync=True, **kwargs))

    async def _run_udf_iter_async(self, dataset, udf, **kwargs):
        """async generator twin of `run_udf_iter` (reference api.py:1105-1152, `ResultAsyncGenerator`)"""
        import asyncio
        loop = asyncio.get_running_loop()
        pool = self._async_pool()
        gen = self.run_udf_iter(dataset, udf, sync=True, **kwargs)
        done = object()
        try:
            while True:
                part = await loop.run_in_executor(pool, next, gen, done)
                if part is done:
                    return
                yield part
        finally:
            await loop.run_in_executor(pool, gen.close)

    def run_udf_iter(self, dataset, udf, roi=None, corrections=None, progress=False,
                     backends=None, plots=None, sync=True):
        """Generator of partial results after each merged partition (api.py:1053-1152); `sync=False`:
        an async generator of the same partial results."""
        if not sync:
            return self._run_udf_iter_async(dataset, udf, roi=roi, corrections=corrections,
                                            progress=progress, backends=backends, plots=plots)
        return self._run_udf_iter_sync(dataset, udf, roi=roi, corrections=corrections, progress=progress,
                                       backends=backends, plots=plots)

    def _run_udf_iter_sync(self, dataset, udf, roi=None, corrections=None, progress=False,
                           backends=None, plots=None):
        nested = self._nested_context()
        if nested is not None:
            yield from nested._run_udf_iter_sync(dataset, udf, roi=roi, corrections=corrections, progress=progress,
                                                 backends=backends, plots=plots)
            return
        udf_is_list = isinstance(udf, (tuple, list))
        udfs = list(udf) if udf_is_list else [udf]
        if len(udfs) == 0:
            raise ValueError("empty list of UDFs - nothing to do!")
        if roi is not None:
            roi = self._normalize_roi(roi, dataset)
        runner = UDFRunner(udfs)
        if corrections is not None and not corrections.have_corrections():
            corrections = None
        # the executor's gate is held from the first step to the end of the iteration (or close()): partial
        # results are published from the executor's per-run state
        scope = self._run_scope()
        with scope:
            for part in runner.run_for_dataset_sync(dataset=dataset, executor=self.executor, roi=roi,
                                                    progress=progress, corrections=corrections,
                                                    backends=backends, iterate=True):
                if not udf_is_list:
                    part.buffers  # noqa: B018  (materialise lazily built result)
                # (while the consumer holds the part, another run on this executor is refused with a clear
                # error instead of waiting for ever or sharing the suspended run's state: hip.RunGate)
                gate = self.executor.run_gate
                gate.suspended = 'a run_udf_iter'
                try:
                    yield part
                finally:
                    gate.suspended = None

    @staticmethod
    def _normalize_roi(roi, dataset):
        import scipy.sparse as sp
        nav = tuple(dataset.shape.nav)
        if sp.issparse(roi):
            roi = roi.toarray()
        if isinstance(roi, (tuple, list)):
            # coordinate tuple(s): ((y, x), ...) or a single (y, x) (api.py:1280-1288)
            coords = roi
            if all(isinstance(c, (int, np.integer)) for c in coords):
                coords = (coords,)
            # items: ((y, x), value) as in the reference (common/sparse.py:20-32), or (y, x) / (y, x, value)
            items = []
            for c in coords:
                c = tuple(c)
                if len(c) == 2 and isinstance(c[0], (tuple, list, np.ndarray)):
                    items.append((tuple(int(i) for i in c[0]), bool(c[1])))
                elif len(c) == len(nav) + 1:
                    items.append((tuple(int(i) for i in c[:-1]), bool(c[-1])))
                else:
                    items.append((tuple(int(i) for i in c), True))
            values = {v for _, v in items}
            if len(values) > 1:
                raise ValueError(f'Cannot cast iterable roi coords with more than one truth value {values}')
            val = values.pop() if values else True
            arr = np.full(nav, not val, dtype=bool)
            for pos, _ in items:
                arr[pos] = val
            roi = arr
        roi = np.asarray(roi)
        if roi.dtype != np.dtype(bool):
            import warnings
            warnings.warn(f"ROI dtype is {roi.dtype}, expected bool. Attempting cast to bool.")
            roi = roi.astype(bool)
        if roi.shape != nav:
            raise ValueError(f"roi: incompatible shapes: {roi.shape} (roi) vs {nav} (dataset)")
        return roi

    def map(self, dataset, f, roi=None, progress=False, corrections=None, backends=None):
        """Apply `f` to every frame; result is kind='nav' (reference api.py:1617-1670)."""
        from libertem_amd.udf.auto import AutoUDF
        results = self.run_udf(dataset=dataset, udf=AutoUDF(f=f), roi=roi, progress=progress,
                               corrections=corrections, backends=backends)
        return results['result']

    def close(self):
        # the datasets this context loaded give up their upload stagers (device buffers, copy stream and the
        # page-locking of the user's host array -- held between runs, io/dataset/memory.py)
        for ds in list(getattr(self, '_loaded', ()) or ()):
            closer = getattr(ds, 'close_stagers', None)
            if closer is not None:
                try:
                    closer()
                except Exception:                       # noqa: BLE001  (closing must not raise)
                    pass
        sib = self.__dict__.pop('_sibling', None)
        if sib is not None:
            sib.close()
        pool = getattr(self, '_async_worker', None)
        if pool is not None:
            pool.shutdown(wait=True)
            self._async_worker = None
        self.executor.close()

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.close()
