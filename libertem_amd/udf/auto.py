"""AutoUDF: a per-frame function mapped over the dataset on the host (reference udf/auto.py:11-83)."""
import numpy as np

from libertem_amd.udf.base import UDF


class AutoUDF(UDF):
    """`f(frame)` for every frame; the result buffer (kind 'nav') takes shape and dtype from `f` called on a frame
    of ones.  monitor=True adds a result-only buffer 'monitor' holding the last valid result (live plotting)."""

    def __init__(self, f, monitor=False):
        super().__init__(f=f, monitor=monitor)

    def auto_buffer(self, var):
        return self.buffer(kind='nav', extra_shape=var.shape, dtype=var.dtype)

    def auto_monitor_buffer(self, var):
        return self.buffer(kind='single', extra_shape=var.shape, dtype=var.dtype, use='result_only')

    def get_result_buffers(self):
        mock = np.ones(tuple(self.meta.dataset_shape.sig), dtype=self.meta.input_dtype)
        res = np.array(self.params.f(mock))
        buffers = {'result': self.auto_buffer(res)}
        if self.params.monitor:
            buffers['monitor'] = self.auto_monitor_buffer(res)
        return buffers

    def process_frame(self, frame):
        res = self.params.f(frame)
        self.results.result[:] = np.array(res)

    def get_results(self):
        if not self.params.monitor:
            return {}
        valid = np.flatnonzero(self.meta.get_valid_nav_mask())      # (flat, compressed to the roi)
        last = int(valid[-1]) if valid.size else 0
        return {'monitor': self.results.result[last]}
