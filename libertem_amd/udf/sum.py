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

    REUSE_TASK_INSTANCES = True      # (udf/base.py: per-partition instances kept between runs)

    def __init__(self, dtype='float32'):
        super().__init__(dtype=dtype)

    def get_preferred_input_dtype(self):
        return self.params.dtype

    def get_backends(self):
        # BACKEND_HIP on an MI355X worker; plain NumPy on a CPU executor (BASELINE config C1:
        # `Context(InlineJobExecutor()).run_udf(ds, SumUDF())`, the reference's udf/sum.py:43-48 runs anywhere).
        # The executor's device class decides (udf/base.py `_execution_plan`): a GPU worker never takes the
        # NumPy branch, there is no fallback from one to the other.
        return (self.BACKEND_HIP, self.BACKEND_NUMPY)

    def get_result_buffers(self):
        return {
            'intensity': self.buffer(kind='sig', dtype=self.meta.input_dtype, where='device'),
            # number of RAW frames in `intensity` when detector corrections are folded (see
            # folds_corrections); 0 if the tiles arrived corrected
            'n_raw': self.buffer(kind='single', dtype='float64', use='private'),
        }

    def folds_corrections(self, corrections, meta):
        """The correction is linear and identical for every frame, so it commutes with the sum:
        sum_f corrected(x_f) = repair((sum_f x_f - N dark) * gain).  The frames are summed raw (one
        pass over the native data) and the single summed image is corrected in get_results."""
        import libertem_amd.udf.masks as um
        return bool(um.FOLD_CORRECTIONS) and np.dtype(meta.input_dtype).kind == 'f'

    def get_results(self):
        img = self.results.intensity
        n = float(np.asarray(self.results.n_raw).reshape(-1)[0])
        corr = self.meta.corrections if self.meta is not None else None
        if n > 0 and corr is not None and corr.have_corrections():
            sig = tuple(img.shape)
            work = np.asarray(img, dtype=np.float64).reshape(-1).copy()
            dark, gain = corr.get_dark_frame(), corr.get_gain_map()
            if dark is not None:
                work -= n * np.asarray(dark, dtype=np.float64).reshape(-1)
            if gain is not None:
                work *= np.asarray(gain, dtype=np.float64).reshape(-1)
            desc = corr.full_frame_descriptor(sig)
            for e, env, c in zip(desc.exclude_flat, desc.repair_flat, desc.repair_counts):
                if c > 0:
                    work[e] = work[env[:c]].sum() / c
            img = work.reshape(sig).astype(img.dtype)
        return {'intensity': img}

    def get_task_data(self):
        if self.meta.array_backend == self.BACKEND_NUMPY:
            return {'workspace': None}
        if self.meta.array_backend != self.BACKEND_HIP:
            raise HipRequiredError("SumUDF needs BACKEND_HIP (an MI355X worker) or BACKEND_NUMPY (a CPU executor)")
        # result dtype = input dtype (udf/sum.py:38-40): float, complex, or -- SumUDF(dtype=<integer>)
        # on integer frames -- an integer with NumPy's wrap-around
        if np.dtype(self.meta.input_dtype).kind not in 'fciu':
            raise NotImplementedError(
                f"SumUDF on MI355X: input dtype {self.meta.input_dtype} is not supported")
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
        if self.meta.array_backend == self.BACKEND_NUMPY:
            self.results.intensity[:] += np.sum(tile, axis=0)        # the reference's line, on the host
            return
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
        if getattr(self.meta, 'corrections_folded', False) and self.meta.tiling_scheme_idx == 0:
            self.results.n_raw[:] += n                  # once per group of frames (first sig slice)
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
            # out[sig slice] += tmp: rows of the innermost axis, one strided add per block of the
            # outer sig axes (2D detectors: ONE call)
            isz = odt.itemsize
            strides = [prod(sig_full[k + 1:]) for k in range(len(sig_full))]
            if len(sig_full) == 1:
                hip.add2d(device, out.data_ptr() + s_origin[0] * isz, sig_full[0], tmp.data_ptr(),
                          s_shape[0], odt, 1, s_shape[0])
            else:
                rows, cols = s_shape[-2], s_shape[-1]
                for outer in np.ndindex(*s_shape[:-2]):
                    off = sum((o + i) * st for o, i, st in zip(s_origin[:-2], outer, strides[:-2]))
                    off += s_origin[-2] * strides[-2] + s_origin[-1]
                    toff = sum(i * prod(s_shape[k + 1:]) for k, i in enumerate(outer))
                    hip.add2d(device, out.data_ptr() + off * isz, sig_full[-1],
                              tmp.data_ptr() + toff * isz, cols, odt, rows, cols)

    def merge(self, dest, src):
        dest.intensity[:] += src.intensity                     # udf/sum.py:50-52
        dest.n_raw[:] += src.n_raw

    def merge_all(self, ordered_results):
        chunks = [b.intensity for b in ordered_results.values()]
        return {'intensity': np.stack(chunks, axis=0).sum(axis=0),
                'n_raw': np.sum([b.n_raw for b in ordered_results.values()], axis=0)}

    def get_dist_merge(self):
        return {'intensity': 'sum', 'n_raw': 'sum'}
