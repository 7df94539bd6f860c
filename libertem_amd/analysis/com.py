"""
Centre-of-mass analysis (reference analysis/com.py:191-334): parameters -> 3-mask ApplyMasksUDF
(disk, y*disk, x*disk) -> shifts, magnitude, divergence, curl.  Note the reference's conventions:
`field.raw_data = (x_centers, y_centers)` (x first), defaults cx = W/2, cy = H/2 as floats.
"""
import numpy as np

from libertem_amd import masks
from libertem_amd.udf.com import (
    com_masks_factory, com_masks_generic, center_shifts, apply_correction, divergence, curl_2d,
    magnitude,
)
from .base import AnalysisResult, AnalysisResultSet
from .masks import BaseMasksAnalysis


class COMResultSet(AnalysisResultSet):
    pass


class COMAnalysis(BaseMasksAnalysis, id_="CENTER_OF_MASS"):
    def get_udf_results(self, udf_results, roi, damage):
        data = udf_results['intensity'].data
        return self.get_generic_results(data[..., 0], data[..., 1], data[..., 2], damage=damage)

    def get_generic_results(self, img_sum, img_y, img_x, damage):
        ref_x, ref_y = self.parameters["cx"], self.parameters["cy"]
        y_raw, x_raw = center_shifts(img_sum, img_y, img_x, ref_y, ref_x)
        shape = y_raw.shape
        y_centers, x_centers = apply_correction(
            y_raw, x_raw, scan_rotation=self.parameters["scan_rotation"],
            flip_y=self.parameters["flip_y"])
        if img_sum.dtype.kind == 'c':
            return COMResultSet([
                AnalysisResult(raw_data=np.real(x_centers), key="x_real", title="x [real]"),
                AnalysisResult(raw_data=np.real(y_centers), key="y_real", title="y [real]"),
                AnalysisResult(raw_data=np.imag(x_centers), key="x_imag", title="x [imag]"),
                AnalysisResult(raw_data=np.imag(y_centers), key="y_imag", title="y [imag]"),
            ])
        m = magnitude(y_centers, x_centers)
        results = [
            AnalysisResult(raw_data=(x_centers, y_centers), key="field", title="field",
                           desc="cubehelix colorwheel visualization", include_in_download=False),
            AnalysisResult(raw_data=m, key="magnitude", title="magnitude",
                           desc="magnitude of the vector field"),
            AnalysisResult(raw_data=x_centers, key="x", title="x",
                           desc="x component of the center"),
            AnalysisResult(raw_data=y_centers, key="y", title="y",
                           desc="y component of the center"),
        ]
        if all(s > 1 for s in shape):
            extra = [
                AnalysisResult(raw_data=divergence(y_centers, x_centers), key="divergence",
                               title="divergence", desc="divergence of the vector field"),
                AnalysisResult(raw_data=curl_2d(y_centers, x_centers), key="curl", title="curl",
                               desc="curl of the 2D vector field"),
            ]
            results[2:2] = extra
        return COMResultSet(results)

    def get_mask_factories(self):
        if self.dataset.shape.sig.dims != 2:
            raise ValueError("can only handle 2D signals currently")
        sy, sx = self.dataset.shape.sig
        p = self.parameters
        if p.get('ri'):
            return com_masks_generic(
                detector_y=sy, detector_x=sx,
                base_mask_factory=lambda: masks.ring(
                    imageSizeY=sy, imageSizeX=sx, centerY=p['cy'], centerX=p['cx'],
                    radius=p['r'], radius_inner=p['ri']))
        return com_masks_factory(detector_y=sy, detector_x=sx, cx=p['cx'], cy=p['cy'], r=p['r'])

    def get_parameters(self, parameters):
        detector_y, detector_x = self.dataset.shape.sig
        return {
            'cx': parameters.get('cx', detector_x / 2),
            'cy': parameters.get('cy', detector_y / 2),
            'r': parameters.get('r', float('inf')),
            'ri': parameters.get('ri', 0.0),
            'scan_rotation': parameters.get('scan_rotation', 0.),
            'flip_y': parameters.get('flip_y', False),
            'use_sparse': parameters.get('use_sparse', False),
            'mask_count': 3,
            'mask_dtype': np.float32,
        }
