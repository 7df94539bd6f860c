"""
PickUDF: hand the frames selected by a (small) ROI back unchanged, in the dataset's native dtype.
Drop-in for the reference's libertem.udf.raw.PickUDF (udf/raw.py:12-76): a 'single' buffer of shape
(frames in the ROI,) + sig; every partition fills the rows of its own frames, the merge adds the
otherwise-zero buffers.  No arithmetic: on a HIP worker the tile's rows are copied inside HBM into
the result buffer (one D2H per partition with the export), on a NumPy worker it is the reference's
slice assignment.
"""
import logging

import numpy as np

from libertem_amd.common.math import prod
from libertem_amd.common.hiparray import HipArray
from libertem_amd.udf.base import UDF

log = logging.getLogger(__name__)


class PickUDF(UDF):
    def __init__(self):
        super().__init__()

    def get_backends(self):
        return (self.BACKEND_HIP, self.BACKEND_NUMPY)

    def get_preferred_input_dtype(self):
        return self.USE_NATIVE_DTYPE

    def get_result_buffers(self):
        dtype = self.meta.input_dtype
        sigshape = tuple(self.meta.dataset_shape.sig)
        if self.meta.roi is not None:
            navsize = int(np.count_nonzero(self.meta.roi))
        else:
            navsize = prod(self.meta.dataset_shape.nav)
        warn_limit = 2**28
        loaded_size = prod(sigshape) * navsize * np.dtype(dtype).itemsize
        if loaded_size > warn_limit:
            log.warning("PickUDF is loading %s bytes, exceeding warning limit %s. "
                        "Consider using or implementing an UDF to process data on the worker "
                        "nodes instead." % (loaded_size, warn_limit))
        return {'intensity': self.buffer(kind='single', extra_shape=(navsize,) + sigshape,
                                         dtype=dtype, where='device')}

    def process_tile(self, tile):
        # flattened nav space with the ROI applied (udf/raw.py:56-60)
        out = self.results.intensity
        sl = self.meta.slice
        if isinstance(tile, HipArray):
            start = sl.origin[0]
            n = tile.shape[0]
            dst = out.torch.reshape(-1)[:prod(out.shape)].reshape(tuple(out.shape))
            idx = (slice(start, start + n),) + tuple(
                slice(o, o + s) for o, s in zip(sl.origin[1:], tuple(sl.shape)[1:]))
            dst[idx].copy_(tile.torch.reshape(-1)[:prod(tile.shape)].reshape(tuple(tile.shape)))
            return
        out[sl.get()] = tile

    def merge(self, dest, src):
        # full-size buffers from every partition, zero outside its own frames
        dest.intensity[:] += src.intensity

    def merge_all(self, ordered_results):
        chunks = [b.intensity for b in ordered_results.values()]
        return {'intensity': np.stack(chunks, axis=0).sum(axis=0)}
