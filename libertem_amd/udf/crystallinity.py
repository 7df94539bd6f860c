"""
CrystallinityUDF on MI355X: integrate the Fourier spectrum of every frame over a ring.
Drop-in for the reference's libertem.udf.crystallinity (udf/crystallinity.py:7-116):

    intensity[frame] = sum( abs(rfft2(frame * real_mask)) * half_fourier_mask )

The reference runs one `np.fft.rfft2` per frame in `process_frame`; here a whole tile goes through
ONE batched hipFFT plus two small kernels (`ltmi_crystallinity`, csrc/ltmi_fft.hip).
"""
import numpy as np

from libertem_amd.common.hiparray import HipArray
from libertem_amd.common.exceptions import HipRequiredError
from libertem_amd.masks import _make_circular_mask
from libertem_amd.udf.base import UDF, UDFMethod

#: workspace budget of one plan (f32 frames + complex64 half spectra), bytes: created only when a route needs it
#: (hipFFT, corrected frames, the column workspace of 512 / 1024-pixel frames) -- frames of 1024 x 1024 want ~250 per
#: pass of the workspace to fill the chip (one workgroup per frame in the column kernel)
FFT_WORKSPACE_BYTES = 2 * 2**30
_PLANS = {}          # (device, h, w, batch) -> hip.FFTPlan
_MASKS = {}          # (device, sig, params) -> (real_mask tensor | None, half mask tensor)


def crystallinity_masks(sigshape, rad_in, rad_out, real_center, real_rad):
    """(real_mask | None, half_fourier_mask) exactly as the reference builds them
    (udf/crystallinity.py:47-71): integer 0/1 arrays."""
    sy, sx = int(sigshape[0]), int(sigshape[1])
    if not (real_center is None or real_rad is None):
        real_mask = 1 - 1 * _make_circular_mask(real_center[1], real_center[0], sx, sy, real_rad)
    else:
        real_mask = None
    outer = 1 * _make_circular_mask(sx * 0.5, sy * 0.5, sx, sy, rad_out)
    inner = 1 * _make_circular_mask(sx * 0.5, sy * 0.5, sx, sy, rad_in)
    fourier_mask = np.fft.fftshift(outer - inner)
    half = fourier_mask[:, :int(fourier_mask.shape[1] * 0.5) + 1]
    return real_mask, half


def mask_box(half):
    """(row_lo, row_hi, n_cols): the non-zeros of the fft-shifted half mask lie in the rows
    [0, row_lo) and [row_hi, h) and in the columns [0, n_cols)."""
    half = np.asarray(half)
    h = half.shape[0]
    used = np.any(half != 0, axis=1)
    cols = np.flatnonzero(np.any(half != 0, axis=0))
    n_cols = int(cols[-1]) + 1 if len(cols) else 0
    lo = 0
    while lo < h and used[lo]:
        lo += 1
    hi = h
    while hi > lo and used[hi - 1]:
        hi -= 1
    if np.any(used[lo:hi]):                 # not a prefix + suffix: read every row
        lo, hi = h, h
    return lo, hi, n_cols


def _plan(device, sig, batch):
    from libertem_amd import hip
    key = (int(device), int(sig[0]), int(sig[1]), int(batch))
    plan = _PLANS.get(key)
    if plan is None:
        if len(_PLANS) >= 8:
            for old in list(_PLANS.values()):
                old.close()
            _PLANS.clear()
        plan = _PLANS[key] = hip.FFTPlan(device, sig[0], sig[1], batch)
    return plan


class CrystallinityUDF(UDF):
    """
    Parameters (identical to the reference)
    ----------
    rad_in, rad_out : float
        inner / outer radius in pixels of the ring in Fourier space
    real_center : (y, x) or None, real_rad : float or None
        disk in real space that is masked out (zero-order peak) before the transform; if either
        is None no real-space mask is applied.
    """

    REUSE_TASK_INSTANCES = True      # (udf/base.py: per-partition instances kept between runs)

    def __init__(self, rad_in, rad_out, real_center, real_rad, **kwargs):
        super().__init__(rad_in=rad_in, rad_out=rad_out, real_center=real_center,
                         real_rad=real_rad, **kwargs)

    def get_backends(self):
        return (self.BACKEND_HIP,)

    def get_method(self):
        return UDFMethod.TILE                       # whole tiles of full frames, one FFT batch

    def get_result_buffers(self):
        return {'intensity': self.buffer(kind="nav", dtype="float32", where='device')}

    def get_dist_merge(self):
        return {'intensity': 'disjoint'}

    def get_task_data(self):
        if self.meta.array_backend != self.BACKEND_HIP:
            raise HipRequiredError("CrystallinityUDF needs BACKEND_HIP (an MI355X worker)")
        import torch
        sig = tuple(self.meta.partition_shape.sig)
        if len(sig) != 2:
            raise ValueError("CrystallinityUDF needs 2D frames")
        device = self.meta.gpu_id if self.meta.gpu_id is not None else 0
        p = self.params
        rc = None if p.real_center is None else tuple(float(x) for x in p.real_center)
        key = (device, sig, float(p.rad_in), float(p.rad_out), rc,
               None if p.real_rad is None else float(p.real_rad))
        hit = _MASKS.get(key)
        if hit is None:
            real_mask, half = crystallinity_masks(sig, p.rad_in, p.rad_out, p.real_center,
                                                  p.real_rad)
            dev = f'cuda:{device}'
            rm = None if real_mask is None else torch.from_numpy(
                np.ascontiguousarray(real_mask.astype(np.float32))).to(dev)
            hm = torch.from_numpy(np.ascontiguousarray(half.astype(np.float32))).to(dev)
            if len(_MASKS) > 16:
                _MASKS.clear()
            hit = _MASKS[key] = (rm, hm, mask_box(half))
        per_frame = sig[0] * sig[1] * 4 + sig[0] * (sig[1] // 2 + 1) * 8
        depth = int(self.meta.tiling_scheme.depth) if self.meta.tiling_scheme is not None else 1
        batch = max(1, min(FFT_WORKSPACE_BYTES // per_frame, depth,
                           int(self.meta.partition_shape[0])))
        return {'real_mask': hit[0], 'half_mask': hit[1], 'box': hit[2],
                'plan': _plan(device, sig, batch), 'sig': sig}

    def process_tile(self, tile):
        out = self.results.intensity
        if not isinstance(tile, HipArray) or not isinstance(out, HipArray):
            raise HipRequiredError("CrystallinityUDF.process_tile expects device tiles and buffers")
        td = self.task_data
        if tuple(tile.shape[1:]) != tuple(td.sig):
            raise ValueError(
                f"CrystallinityUDF transforms whole frames {td.sig}, got tiles of {tile.shape[1:]} "
                "(do not force a sub-frame tileshape)")
        rm = td.real_mask
        if getattr(self.meta, 'corrections_folded', False):
            # RAW tile: (x - dark) * gain and the dead-pixel patches happen inside the conversion
            # pass of the transform (no corrected copy of the frames)
            tables = self.meta.corrections.device_tables(tile.device, td.sig)
            td.plan.crystallinity_corrected(
                tile.data_ptr(), tile.dtype, tile.shape[0], tile.ld, tables,
                None if rm is None else rm.data_ptr(), td.half_mask.data_ptr(), td.box,
                out.data_ptr(), False, stream=self.meta.stream_ptr)
            return
        td.plan.crystallinity(tile.data_ptr(), tile.dtype, tile.shape[0], tile.ld,
                              None if rm is None else rm.data_ptr(), td.half_mask.data_ptr(),
                              td.box, out.data_ptr(), False, stream=self.meta.stream_ptr)

    def folds_corrections(self, corrections, meta):
        """Detector corrections are applied inside the transform's conversion pass
        (ltmi_crystallinity_corrected): the dataset hands out raw tiles."""
        import libertem_amd.udf.masks as um
        return bool(um.FOLD_CORRECTIONS)


def run_analysis_crystall(ctx, dataset, rad_in, rad_out, real_center=None, real_rad=None, roi=None,
                          progress=False):
    """Reference udf/crystallinity.py:82-116."""
    udf = CrystallinityUDF(rad_in=rad_in, rad_out=rad_out, real_center=real_center,
                           real_rad=real_rad)
    return ctx.run_udf(dataset=dataset, udf=udf, roi=roi, progress=progress)
