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
    def get_backends(self):
        return (self.BACKEND_HIP,)

    def get_result_buffers(self):
        dtype = np.result_type(self.meta.input_dtype, np.float32)
        return {'intensity': self.buffer(kind="nav", dtype=dtype, where='device')}

    def get_task_data(self):
        if self.meta.array_backend != self.BACKEND_HIP:
            raise HipRequiredError("SumSigUDF needs BACKEND_HIP (an MI355X worker)")
        return {}

    def process_tile(self, tile):
        # results.intensity[:] += tile.reshape(n, -1).sum(axis=1)   (udf/sumsigudf.py:30-39)
        from libertem_amd import hip
        out = self.results.intensity
        if not isinstance(tile, HipArray) or not isinstance(out, HipArray):
            raise HipRequiredError("SumSigUDF.process_tile expects device tiles and buffers")
        if out.dtype.kind != 'f':
            raise NotImplementedError(f"SumSigUDF: result dtype {out.dtype} not supported yet")
        n = tile.shape[0]
        hip.sum_sig(tile.device, tile.data_ptr(), tile.dtype, n, prod(tile.shape[1:]), tile.ld,
                    out.data_ptr(), out.dtype, True)

    def get_dist_merge(self):
        return {'intensity': 'disjoint'}


def run_sumsig(ctx, dataset):
    return ctx.run_udf(dataset=dataset, udf=SumSigUDF())
