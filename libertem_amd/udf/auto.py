"""AutoUDF: per-frame function mapped over the dataset on the host (reference udf/auto.py)."""
import numpy as np

from libertem_amd.udf.base import UDF


class AutoUDF(UDF):
    def __init__(self, f, monitor=False):
        super().__init__(f=f, monitor=monitor)

    def auto_buffer(self, var):
        return self.buffer(kind='nav', extra_shape=var.shape, dtype=var.dtype)

    def get_result_buffers(self):
        mock = np.ones(tuple(self.meta.dataset_shape.sig), dtype=self.meta.input_dtype)
        res = np.array(self.params.f(mock))
        return {'result': self.auto_buffer(res)}

    def process_frame(self, frame):
        res = self.params.f(frame)
        self.results.result[:] = np.array(res)
