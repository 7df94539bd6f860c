"""
Centre-of-mass UDF on MI355X.

Drop-in for the reference's libertem.udf.com (udf/com.py): `CoMUDF.with_params(cy, cx, r, ri,
scan_rotation, flip_y, regression)`, the same nine result buffers, the same host-side
post-processing (`center_shifts`, `apply_correction`, `divergence`, `curl_2d`, `magnitude`,
regression).  The per-frame reduction (3 masks: disk, y*disk, x*disk) runs through the same
fused HIP kernel as ApplyMasksUDF (reference: ApplyMasksEngine reuse, udf/com.py:568-582).
"""
from enum import IntEnum
from typing import NamedTuple, Optional, Union

import numpy as np

from libertem_amd import masks
from libertem_amd.corrections import coordinates
from libertem_amd.common.container import MaskContainer
from libertem_amd.common.math import prod
from libertem_amd.udf.base import UDF
from libertem_amd.udf.masks import ApplyMasksEngine


class RegressionOptions(IntEnum):
    NO_REGRESSION = -1
    SUBTRACT_MEAN = 0
    SUBTRACT_LINEAR = 1


class CoMParams(NamedTuple):
    cy: Optional[float] = None
    cx: Optional[float] = None
    r: float = float('inf')
    ri: Optional[float] = 0.
    scan_rotation: float = 0.
    flip_y: bool = False
    regression: Union[np.ndarray, int] = RegressionOptions.NO_REGRESSION


def com_masks_generic(detector_y, detector_x, base_mask_factory):
    """[selection, y*selection, x*selection] (udf/com.py:69-97)"""
    return [
        base_mask_factory,
        lambda: masks.gradient_y(imageSizeX=detector_x, imageSizeY=detector_y)
        * base_mask_factory(),
        lambda: masks.gradient_x(imageSizeX=detector_x, imageSizeY=detector_y)
        * base_mask_factory(),
    ]


def com_masks_factory(detector_y, detector_x, cy, cx, r):
    """Disk-selected CoM masks (udf/com.py:47-66)"""
    def disk_mask():
        return masks.circular(centerX=cx, centerY=cy, imageSizeX=detector_x,
                              imageSizeY=detector_y, radius=r)
    return com_masks_generic(detector_y, detector_x, disk_mask)


def center_shifts(img_sum, img_y, img_x, ref_y, ref_x):
    """CoM relative to (ref_y, ref_x); frames with zero intensity map to zero shift
    (udf/com.py:100-107)."""
    nonzero = img_sum != 0
    x_centers = np.divide(img_x, img_sum, where=nonzero)
    y_centers = np.divide(img_y, img_sum, where=nonzero)
    x_centers[~nonzero] = ref_x
    y_centers[~nonzero] = ref_y
    x_centers -= ref_x
    y_centers -= ref_y
    return (y_centers, x_centers)


def apply_correction(y_centers, x_centers, scan_rotation, flip_y, forward=True):
    """Rotate / flip the shift vectors (udf/com.py:110-127)."""
    shape = y_centers.shape
    transform = coordinates.flip_y() if flip_y else coordinates.identity()
    transform = coordinates.rotate_deg(scan_rotation) @ transform      # right to left
    if not forward:
        transform = np.linalg.inv(transform)
    y_t, x_t = transform @ (y_centers.reshape(-1), x_centers.reshape(-1))
    return (y_t.reshape(shape), x_t.reshape(shape))


def divergence(y_centers, x_centers):
    return np.gradient(y_centers, axis=0) + np.gradient(x_centers, axis=1)


def curl_2d(y_centers, x_centers):
    # dFy/dx - dFx/dy ; axis 0 is y, axis 1 is x
    return np.gradient(y_centers, axis=1) - np.gradient(x_centers, axis=0)


def magnitude(y_centers, x_centers):
    return np.sqrt(y_centers**2 + x_centers**2)


def coordinate_check(y_centers, x_centers, roi=None):
    """
    RMS curl for scan_rotation = 0..359 degrees, without (straight) and with (flipped) flip_y.
    A purely electrostatic field is curl-free, so the right settings minimise it
    (udf/com.py:145-189).
    """
    straight = np.zeros(360)
    flipped = np.zeros(360)
    if roi is None:
        roi = (slice(0, -1), slice(0, -1))      # last row / column hold artefacts
    for angle in range(360):
        for flip, target in ((False, straight), (True, flipped)):
            yt, xt = apply_correction(y_centers, x_centers, scan_rotation=angle, flip_y=flip)
            curl = curl_2d(yt, xt)
            target[angle] = np.sqrt(np.mean(curl[roi]**2))
    return (straight, flipped)


class GuessResult(NamedTuple):
    """Parameters for CoMUDF.with_params inferred from data (udf/com.py:192-204)."""
    scan_rotation: int
    flip_y: bool
    cy: float
    cx: float


def guess_corrections(y_centers, x_centers, roi=None) -> GuessResult:
    """
    Guess centre offset, scan_rotation and flip_y from CoM data of atomic-resolution 4D STEM:
    minimise RMS curl, zero mean deflection, divergence histogram skewed negative at the atom
    columns (udf/com.py:207-295).
    """
    if roi is None:
        roi = (slice(0, -1), slice(0, -1))
    straight, flipped = coordinate_check(y_centers, x_centers, roi=roi)
    flip_y = bool(np.min(flipped) < np.min(straight))
    angle = np.argmin(flipped) if flip_y else np.argmin(straight)
    cy_, cx_ = apply_correction(y_centers, x_centers, scan_rotation=angle, flip_y=flip_y)
    # a 180 degree rotation only flips the sign of divergence and curl: pick the polarity whose
    # divergence histogram has its heavier tail on the negative side
    div = divergence(cy_, cx_)[roi]
    all_range = np.maximum(-np.min(div), np.max(div))
    hist, bins = np.histogram(div, range=(-all_range, all_range), bins=5)
    if np.sum(hist[:1]) < np.sum(hist[-1:]):
        angle += 180
    if angle > 180:
        angle -= 360
    return GuessResult(scan_rotation=int(angle), flip_y=flip_y, cy=np.mean(y_centers[roi]),
                       cx=np.mean(x_centers[roi]))


_COM_CONTAINERS = {}


class CoMUDF(UDF):
    """
    Centre-of-mass analysis as a UDF.  Result buffers (all kind 'nav' unless noted):
    raw_com (y, x), raw_shifts (dy, dx), field (2,), field_y, field_x, magnitude, divergence,
    curl and regression (kind 'single', (3, 2)).  See udf/com.py:298-376 of the reference.
    """

    REUSE_TASK_INSTANCES = True      # (udf/base.py: per-partition instances kept between runs)
    ACCEPTS_ROW_VIEWS = True         # process_tile reads an ROI's frames through a row list (no gather)

    def __init__(self, com_params: CoMParams = CoMParams()):
        super().__init__(com_params=com_params)

    @classmethod
    def with_params(cls, *, cy=None, cx=None, r=float('inf'), ri=0., scan_rotation=0.,
                    flip_y=False, regression=RegressionOptions.NO_REGRESSION):
        if ri >= r:
            raise ValueError('Inner radius must be less than outer radius for annular CoM')
        return cls(com_params=CoMParams(cy=cy, cx=cx, r=r, ri=ri, scan_rotation=scan_rotation,
                                        flip_y=flip_y, regression=regression))

    def get_backends(self):
        return (self.BACKEND_HIP,)

    def get_result_buffers(self):
        dtype = np.result_type(self.meta.input_dtype, np.float32)
        nav2 = dict(kind='nav', dtype=dtype, extra_shape=(2,), use='result_only')
        nav0 = dict(kind='nav', dtype=dtype, use='result_only')
        return {
            'raw_mask_result': self.buffer(kind='nav', dtype=dtype, extra_shape=(3,),
                                           where='device', use='private'),
            'raw_com': self.buffer(**nav2),
            'raw_shifts': self.buffer(**nav2),
            'field': self.buffer(**nav2),
            'field_y': self.buffer(**nav0),
            'field_x': self.buffer(**nav0),
            'magnitude': self.buffer(**nav0),
            'divergence': self.buffer(**nav0),
            'curl': self.buffer(**nav0),
            'regression': self.buffer(kind='single', extra_shape=(3, 2), dtype=np.float64,
                                      use='result_only'),
        }

    def get_params(self) -> CoMParams:
        sig_shape = tuple(self.meta.dataset_shape.sig)
        p = self.params.com_params
        cy = p.cy if p.cy is not None else sig_shape[0] // 2
        cx = p.cx if p.cx is not None else sig_shape[1] // 2
        return CoMParams(cy=cy, cx=cx, r=p.r, ri=p.ri, scan_rotation=p.scan_rotation,
                         flip_y=p.flip_y, regression=p.regression)

    def get_task_data(self):
        sig_shape = tuple(self.meta.dataset_shape.sig)
        cp = self.get_params()
        if len(sig_shape) != 2:
            raise ValueError('CoMUDF only works with 2D sig shape.')
        if len(self.meta.dataset_shape.nav) != 2:
            raise ValueError('CoMUDF only works with 2D nav shape.')
        if cp.ri is None or np.isclose(cp.ri, 0.):
            mask_factory = com_masks_factory(detector_y=sig_shape[0], detector_x=sig_shape[1],
                                             cx=cp.cx, cy=cp.cy, r=cp.r)
        else:
            mask_factory = com_masks_generic(
                detector_y=sig_shape[0], detector_x=sig_shape[1],
                base_mask_factory=lambda: masks.ring(
                    imageSizeY=sig_shape[0], imageSizeX=sig_shape[1], centerY=cp.cy,
                    centerX=cp.cx, radius=cp.r, radius_inner=cp.ri))
        # the three CoM masks depend only on the geometry: keep their HBM image across tasks / runs
        key = (sig_shape, float(cp.cy), float(cp.cx), float(cp.r),
               None if cp.ri is None else float(cp.ri))
        container = _COM_CONTAINERS.get(key)
        if container is None:
            container = MaskContainer(mask_factories=mask_factory, dtype=np.float32,
                                      use_sparse=False, count=3, backend=self.BACKEND_HIP)
            _COM_CONTAINERS[key] = container
            while len(_COM_CONTAINERS) > 4:
                _COM_CONTAINERS.pop(next(iter(_COM_CONTAINERS))).close()
        engine = ApplyMasksEngine(masks=container, meta=self.meta, use_torch=True)
        if getattr(self.meta, 'corrections_folded', False):
            # detector corrections absorbed by the three masks (udf/masks.py): raw frames are read
            from libertem_amd.udf.masks import _folded_plan
            folded, plan_state = _folded_plan(self.meta.corrections, container,
                                              container.mask_factories, sig_shape, 3)
            engine.fold(folded, plan_state)
        return {'com_params': cp, 'engine': engine}

    def folds_corrections(self, corrections, meta):
        import libertem_amd.udf.masks as um
        return bool(um.FOLD_CORRECTIONS)

    def process_tile(self, tile):
        self.task_data.engine.process_tile(tile, out=self.results.raw_mask_result,
                                           accumulate=True)

    def get_dist_merge(self):
        return {'raw_mask_result': 'disjoint'}

    # --- main-process post-processing (udf/com.py:584-717) ---------------------------------------
    def get_field_results(self, field_y, field_x):
        return {
            'magnitude': magnitude(y_centers=field_y, x_centers=field_x),
            'divergence': divergence(y_centers=field_y, x_centers=field_x),
            'curl': curl_2d(y_centers=field_y, x_centers=field_x),
        }

    def get_regression(self, field, valid_mask):
        inp = None
        result = np.zeros((3, 2))
        cp = self.get_params()

        def get_inp():
            inp = np.ones(field.shape[:-1] + (3,))
            y, x = np.ogrid[:field.shape[0], :field.shape[1]]
            inp[..., 1] = y
            inp[..., 2] = x
            return inp

        if isinstance(cp.regression, (int, np.integer)):
            if cp.regression == -1:
                pass
            elif cp.regression == 0:
                result[0] = np.mean(field[valid_mask], axis=0)
            elif cp.regression == 1:
                inp = get_inp()
                result[:] = np.linalg.lstsq(inp[valid_mask], field[valid_mask], rcond=None)[0]
            else:
                raise ValueError(f'Unrecognized regression option {cp.regression}')
        else:
            regression = np.array(cp.regression)
            if regression.shape != (3, 2):
                raise ValueError(f"Regression parameter {cp.regression} "
                                 "doesn't have required shape (3, 2).")
            result[:] = regression
        has_lin = not np.allclose(result[1:], 0)
        if has_lin and inp is None:
            inp = get_inp()
        if not has_lin:
            inp = None
        return result, inp

    def apply_mean_regression(self, regression, field_inout, valid_mask):
        field_inout[valid_mask] -= regression[0]

    def apply_lin_regression(self, regression, inp, field_inout, valid_mask):
        field_inout[valid_mask] -= inp[valid_mask] @ regression

    def _results_on_device(self, data, cp):
        """Full 2D scan of real float32 sums, no regression: the float64 field and its derived maps
        come from ONE pass on the GPU (ltmi_com_fields) instead of ~10 NumPy passes over the scan;
        the float32 raw_shifts / raw_com are two cheap NumPy expressions as in the reference."""
        if self.meta.roi is not None or data.dtype != np.float32 or data.ndim != 3 \
                or min(data.shape[:2]) < 2 or not isinstance(cp.regression, (int, np.integer)) \
                or cp.regression != -1:
            return None
        try:
            import torch
            from libertem_amd import hip
            if not torch.cuda.is_available():
                return None
        except Exception:
            return None
        transform = coordinates.flip_y() if cp.flip_y else coordinates.identity()
        transform = coordinates.rotate_deg(cp.scan_rotation) @ transform
        ny, nx = data.shape[:2]
        dev = torch.cuda.current_device()
        raw_ptr, keep = hip.map_or_upload(dev, data.reshape(-1, 3))
        out = torch.empty((5, ny * nx), dtype=torch.float64, device=f'cuda:{dev}')
        hip.com_fields(dev, raw_ptr, 3, ny, nx, cp.cy, cp.cx, transform,
                       *[out[i].data_ptr() for i in range(5)],
                       stream=torch.cuda.current_stream(dev))
        f = hip.download_pinned(out)
        del keep
        raw_shifts = center_shifts(img_sum=data[..., 0], img_y=data[..., 1], img_x=data[..., 2],
                                   ref_y=cp.cy, ref_x=cp.cx)
        n = ny * nx
        field = np.stack([f[0], f[1]], axis=-1)
        return {
            'raw_shifts': np.stack([raw_shifts[0].reshape(n), raw_shifts[1].reshape(n)], axis=-1),
            'raw_com': np.stack([raw_shifts[0].reshape(n) + cp.cy, raw_shifts[1].reshape(n) + cp.cx],
                                axis=-1),
            'field': field, 'field_y': f[0].reshape((n, 1)), 'field_x': f[1].reshape((n, 1)),
            'magnitude': f[2].reshape((n, 1)), 'divergence': f[3].reshape((n, 1)),
            'curl': f[4].reshape((n, 1)), 'regression': np.zeros((3, 2)),
        }

    def get_results(self):
        cp = self.get_params()
        raw = self.results.get_buffer('raw_mask_result')
        data = raw.data
        fast = self._results_on_device(data, cp)
        if fast is not None:
            return fast
        raw_shifts = center_shifts(img_sum=data[..., 0], img_y=data[..., 1], img_x=data[..., 2],
                                   ref_y=cp.cy, ref_x=cp.cx)
        raw_com = (raw_shifts[0].copy() + cp.cy, raw_shifts[1].copy() + cp.cx)
        field = apply_correction(y_centers=raw_shifts[0], x_centers=raw_shifts[1],
                                 scan_rotation=cp.scan_rotation, flip_y=cp.flip_y)
        roi = self.meta.roi
        raw_shifts = np.moveaxis(np.array(raw_shifts), 0, -1)
        raw_com = np.moveaxis(np.array(raw_com), 0, -1)
        field = np.moveaxis(np.array(field), 0, -1)
        nav_size = prod(self.meta.dataset_shape.nav)
        valid_mask = self.meta.get_valid_nav_mask(full_nav=True)
        if valid_mask is None:
            valid_mask = np.ones(nav_size, dtype=bool)
        valid_mask = np.asarray(valid_mask).reshape(tuple(self.meta.dataset_shape.nav))
        regression, inp = self.get_regression(field, valid_mask=valid_mask)
        if inp is not None:
            self.apply_lin_regression(regression, inp, field, valid_mask)
        elif not np.allclose(regression[0], 0):
            self.apply_mean_regression(regression, field, valid_mask)
        results = {
            'raw_shifts': raw_shifts, 'raw_com': raw_com, 'field': field,
            'field_y': field[..., 0], 'field_x': field[..., 1],
            'regression': regression.astype(np.float64),
        }
        results.update(self.get_field_results(field_y=field[..., 0], field_x=field[..., 1]))
        buffers = self.get_result_buffers()
        for key, buf in buffers.items():
            if buf.kind == 'nav' and key in results:
                arr = np.asarray(results[key])
                if roi is not None:
                    arr = arr[np.asarray(roi).reshape(tuple(self.meta.dataset_shape.nav))]
                else:
                    arr = arr.reshape((nav_size, -1))
                results[key] = arr
        return results
