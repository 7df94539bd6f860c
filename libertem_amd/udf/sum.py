"""
SumUDF on MI355X: sum of all frames, preserving the signal dimension.
Drop-in for the reference's libertem.udf.sum.SumUDF (udf/sum.py:6-58).
"""
import numpy as np

from libertem_amd.common.math import prod
from libertem_amd.common.buffers import HipSigView
from libertem_amd.common.hiparray import HipArray
from libertem_amd.common.exceptions import HipRequiredError
from libertem_amd.udf.base import UDF


class SumUDF(UDF):
    """
    Parameters
    ----------
    dtype : numpy.dtype, optional
        Preferred dtype for computation, default 'float32'.  The actual dtype is
        `numpy.result_type(dtype, dataset dtype)` (udf/sum.py:11-17, :38-40).
    """

    def __init__(self, dtype='float32'):
        super().__init__(dtype=dtype)

    def get_preferred_input_dtype(self):
        return self.params.dtype

    def get_backends(self):
        return (self.BACKEND_HIP,)

    def get_result_buffers(self):
        return {'intensity': self.buffer(kind='sig', dtype=self.meta.input_dtype, where='device')}

    def get_task_data(self):
        if self.meta.array_backend != self.BACKEND_HIP:
            raise HipRequiredError("SumUDF needs BACKEND_HIP (an MI355X worker)")
        if np.dtype(self.meta.input_dtype).kind not in 'f':
            raise NotImplementedError(
                f"SumUDF on MI355X accumulates in float32/float64; input dtype "
                f"{self.meta.input_dtype} is not supported yet")
        return {'workspace': {}}

    def _workspace(self, device, nbytes):
        import torch
        ws = self.task_data.workspace
        if ws.get('bytes', -1) < nbytes:
            ws['t'] = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=f'cuda:{device}')
            ws['bytes'] = nbytes
        return ws['t'].data_ptr()

    def process_tile(self, tile):
        # results.intensity[sig slice] += tile.sum(axis=0)      (udf/sum.py:43-48)
        from libertem_amd import hip
        view = self.results.intensity
        if not isinstance(tile, HipArray) or not isinstance(view, HipSigView):
            raise HipRequiredError("SumUDF.process_tile expects device tiles and buffers")
        out = view.array
        odt = out.dtype
        n = tile.shape[0]
        sig_full = tuple(self.meta.dataset_shape.sig)
        s_origin = tuple(view.tile_slice.origin[-len(sig_full):])
        s_shape = tuple(view.tile_slice.shape.sig)
        n_px = prod(s_shape)
        device = tile.device
        ws = self._workspace(device, hip.sum_frames_workspace(n, n_px, odt))
        whole_rows = s_shape[1:] == sig_full[1:] and all(o == 0 for o in s_origin[1:])
        if whole_rows:
            inner = prod(sig_full[1:])
            out_ptr = out.data_ptr() + s_origin[0] * inner * odt.itemsize
            hip.sum_frames(device, tile.data_ptr(), tile.dtype, n, n_px, tile.ld, out_ptr, odt,
                           True, ws)
        else:
            # partial-width sig slice: reduce into a temporary, add into the strided region
            tmp = HipArray.zeros(s_shape, odt, device)
            hip.sum_frames(device, tile.data_ptr(), tile.dtype, n, n_px, tile.ld, tmp.data_ptr(),
                           odt, False, ws)
            sl = tuple(slice(o, o + s) for o, s in zip(s_origin, s_shape))
            out.torch.reshape(sig_full)[sl] += tmp.torch.reshape(s_shape)

    def merge(self, dest, src):
        dest.intensity[:] += src.intensity                     # udf/sum.py:50-52

    def merge_all(self, ordered_results):
        chunks = [b.intensity for b in ordered_results.values()]
        return {'intensity': np.stack(chunks, axis=0).sum(axis=0)}

    def get_dist_merge(self):
        return {'intensity': 'sum'}
