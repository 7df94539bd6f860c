"""
PickUDF -- frames selected by a (small) ROI come back unchanged, in the dataset's native dtype.

Same contract as the reference's PickUDF (src/libertem/udf/raw.py:12-76): ONE 'single' buffer of shape
(number of selected frames,) + sig_shape; each partition writes the rows of its own frames into an
otherwise zero buffer and the merge adds the buffers up.  Nothing is computed: a HIP worker copies
the tile's rows inside HBM (one D2H per partition at export), a NumPy worker assigns the slice.
"""
import logging

import numpy as np

from libertem_amd.common.math import prod
from libertem_amd.common.hiparray import HipArray
from libertem_amd.udf.base import UDF

log = logging.getLogger(__name__)

#: picking is meant for a handful of frames; beyond this many bytes a warning points to real UDFs
PICK_WARN_BYTES = 1 << 28


class PickUDF(UDF):
    def __init__(self):
        super().__init__()

    def get_backends(self):
        return (self.BACKEND_HIP, self.BACKEND_NUMPY)

    def get_preferred_input_dtype(self):
        return self.USE_NATIVE_DTYPE            # hand the pixels back as they are stored

    def _n_selected(self):
        roi = self.meta.roi
        return prod(self.meta.dataset_shape.nav) if roi is None else int(np.count_nonzero(roi))

    def get_result_buffers(self):
        sig = tuple(self.meta.dataset_shape.sig)
        n_sel = self._n_selected()
        dtype = np.dtype(self.meta.input_dtype)
        n_bytes = n_sel * prod(sig) * dtype.itemsize
        if n_bytes > PICK_WARN_BYTES:
            log.warning("PickUDF is loading %d bytes, exceeding warning limit %d. Consider using or "
                        "implementing an UDF to process data on the worker nodes instead.",
                        n_bytes, PICK_WARN_BYTES)
        # 'single' (not 'nav'): a nav buffer would span the whole scan, NaN-filled outside the ROI
        return {'intensity': self.buffer(kind='single', extra_shape=(n_sel,) + sig, dtype=dtype,
                                         where='device')}

    def process_tile(self, tile):
        target = self.results.intensity
        where = self.meta.slice                 # flat nav (ROI-compressed) + sig coordinates
        if not isinstance(tile, HipArray):
            target[where.get()] = tile
            return
        first = where.origin[0]
        window = (slice(first, first + tile.shape[0]),) + tuple(
            slice(o, o + n) for o, n in zip(where.origin[1:], tuple(where.shape)[1:]))
        dst = target.torch.reshape(-1)[:prod(target.shape)].reshape(tuple(target.shape))
        src = tile.torch.reshape(-1)[:prod(tile.shape)].reshape(tuple(tile.shape))
        dst[window].copy_(src)

    def merge(self, dest, src):
        # every partition delivers a full-size buffer that is zero outside its own frames
        dest.intensity[:] += src.intensity

    def merge_all(self, ordered_results):
        total = None
        for part in ordered_results.values():
            total = np.array(part.intensity) if total is None else total + part.intensity
        return {'intensity': total}
