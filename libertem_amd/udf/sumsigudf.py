"""
SumSigUDF on MI355X: per-frame sum over the signal axes.
Drop-in for the reference's libertem.udf.sumsigudf.SumSigUDF (udf/sumsigudf.py:6-45).
"""
import numpy as np

from libertem_amd.common.math import prod
from libertem_amd.common.hiparray import HipArray
from libertem_amd.common.exceptions import HipRequiredError
from libertem_amd.udf.base import UDF


class SumSigUDF(UDF):
    REUSE_TASK_INSTANCES = True      # (udf/base.py: per-partition instances kept between runs)

    def get_backends(self):
        # BACKEND_HIP on an MI355X worker, plain NumPy on a CPU executor (see SumUDF.get_backends)
        return (self.BACKEND_HIP, self.BACKEND_NUMPY)

    def get_result_buffers(self):
        dtype = np.result_type(self.meta.input_dtype, np.float32)
        return {'intensity': self.buffer(kind="nav", dtype=dtype, where='device')}

    def folds_corrections(self, corrections, meta):
        """Per-frame sum of CORRECTED pixels = one weighted sum of the RAW pixels: the corrections
        fold into a single mask w = R^T(1) * gain and a constant w . dark (see udf/masks.py)."""
        import libertem_amd.udf.masks as um
        return bool(um.FOLD_CORRECTIONS) and \
            np.dtype(np.result_type(meta.input_dtype, np.float32)) == np.float32

    def get_task_data(self):
        if self.meta.array_backend == self.BACKEND_NUMPY:
            return {'engine': None}
        if self.meta.array_backend != self.BACKEND_HIP:
            raise HipRequiredError("SumSigUDF needs BACKEND_HIP (an MI355X worker) or BACKEND_NUMPY (a CPU executor)")
        if getattr(self.meta, 'corrections_folded', False):
            from libertem_amd.udf.masks import ApplyMasksEngine, _cached_container, _folded_plan
            sig = tuple(self.meta.dataset_shape.sig)
            ones = _ones_factory(sig)
            plain = _cached_container(ones, np.float32, False, 1, 'scipy.sparse')
            folded, plan_state = _folded_plan(self.meta.corrections, plain, ones, sig, 1)
            engine = ApplyMasksEngine(plain, self.meta, True)
            engine.fold(folded, plan_state)
            return {'engine': engine}
        return {'engine': None}

    def process_tile(self, tile):
        # results.intensity[:] += tile.reshape(n, -1).sum(axis=1)   (udf/sumsigudf.py:30-39)
        if self.meta.array_backend == self.BACKEND_NUMPY:
            # the reference's line, on the host (udf/sumsigudf.py:30-39)
            self.results.intensity[:] += np.sum(tile.reshape((tile.shape[0], -1)), axis=-1)
            return
        from libertem_amd import hip
        out = self.results.intensity
        if not isinstance(tile, HipArray) or not isinstance(out, HipArray):
            raise HipRequiredError("SumSigUDF.process_tile expects device tiles and buffers")
        if out.dtype.kind not in 'fc':
            raise NotImplementedError(f"SumSigUDF: result dtype {out.dtype} not supported")
        n = tile.shape[0]
        accumulate = not self.results.get_buffer('intensity').write_once
        if self.task_data.engine is not None:
            self.task_data.engine.process_tile(tile, out=out.reshape((n, 1)), accumulate=accumulate)
            return
        hip.sum_sig(tile.device, tile.data_ptr(), tile.dtype, n, prod(tile.shape[1:]), tile.ld,
                    out.data_ptr(), out.dtype, accumulate, stream=self.meta.stream_ptr)

    def get_write_once_buffers(self):
        """whole-frame tiles: one kernel call per row (see ApplyMasksUDF.get_write_once_buffers)"""
        ts = self.meta.tiling_scheme if self.meta is not None else None
        if self.meta is not None and self.meta.array_backend == self.BACKEND_NUMPY:
            return ()
        if ts is None or len(ts) != 1 or getattr(self.meta, 'corrections_folded', False) \
                or getattr(self.meta, 'sig_sliced_tiles', False):
            return ()
        return ('intensity',)

    def get_dist_merge(self):
        return {'intensity': 'disjoint'}

    def get_hip_direct_results(self):
        from libertem_amd.common import udf as udf_common
        return udf_common.HIP_DIRECT_ROW_MAX >= 8


_ONES = {}


def _ones_factory(sig):
    """One all-ones mask of the signal shape; the factory object is kept so that the mask-stack
    caches (keyed by factory identity) hit across tasks and runs."""
    f = _ONES.get(sig)
    if f is None:
        def f():
            return np.ones((1,) + tuple(sig), dtype=np.float32)
        _ONES[sig] = f
    return f


def run_sumsig(ctx, dataset):
    return ctx.run_udf(dataset=dataset, udf=SumSigUDF())
