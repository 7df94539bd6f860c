"""
Centre-of-mass analysis (reference analysis/com.py:191-334): parameters -> 3-mask ApplyMasksUDF
(disk, y*disk, x*disk) -> shifts, magnitude, divergence, curl.  Note the reference's conventions:
`field.raw_data = (x_centers, y_centers)` (x first), defaults cx = W/2, cy = H/2 as floats.
"""
import numpy as np

from libertem_amd import masks
from libertem_amd.udf.com import (                               # noqa: F401  (importable from here too: analysis/com.py:14-19)
    com_masks_factory, com_masks_generic, center_shifts, apply_correction, divergence, curl_2d,
    magnitude, coordinate_check, GuessResult, guess_corrections,
)
from .base import AnalysisResult, AnalysisResultSet
from .masks import BaseMasksAnalysis


class COMResultSet(AnalysisResultSet):
    pass


class COMAnalysis(BaseMasksAnalysis, id_="CENTER_OF_MASS"):
    def get_udf_results(self, udf_results, roi, damage):
        data = udf_results['intensity'].data
        fields = self._fields_on_device(data)
        return self.get_generic_results(data[..., 0], data[..., 1], data[..., 2], damage=damage,
                                        fields=fields)

    def _fields_on_device(self, data):
        """2D scans of real float32 sums: shift field, magnitude, divergence and curl in one pass on
        the GPU (ltmi_com_fields) instead of ~10 NumPy passes over the scan (3 ms of a 26 ms C3
        run).  Other cases (complex data, 1D / 3D scans) keep the NumPy chain of the reference."""
        if data.dtype != np.float32 or data.ndim != 3 or data.shape[-1] != 3 \
                or min(data.shape[:2]) < 2:
            return None
        try:
            import torch
            from libertem_amd import hip
            if not torch.cuda.is_available():
                return None
        except Exception:
            return None
        from libertem_amd.corrections import coordinates
        p = self.parameters
        transform = coordinates.flip_y() if p["flip_y"] else coordinates.identity()
        transform = coordinates.rotate_deg(p["scan_rotation"]) @ transform
        ny, nx = data.shape[:2]
        dev = torch.cuda.current_device()
        # the sums are read where the kernels wrote them (page-locked host memory, zero-copy) and the
        # five maps come back through page-locked memory: 13 MiB over the link at its own speed
        raw_ptr, keep = hip.map_or_upload(dev, data.reshape(-1, 3))
        out = torch.empty((5, ny * nx), dtype=torch.float64, device=f'cuda:{dev}')
        ptr = [out[i].data_ptr() for i in range(5)]
        hip.com_fields(dev, raw_ptr, 3, ny, nx, p["cy"], p["cx"], transform, ptr[0], ptr[1],
                       ptr[2], ptr[3], ptr[4], stream=torch.cuda.current_stream(dev))
        host = hip.download_pinned(out).reshape((5, ny, nx))
        del keep
        return dict(y=host[0], x=host[1], magnitude=host[2], divergence=host[3], curl=host[4])

    def get_generic_results(self, img_sum, img_y, img_x, damage, fields=None):
        ref_x, ref_y = self.parameters["cx"], self.parameters["cy"]
        if fields is not None:
            y_centers, x_centers = fields['y'], fields['x']
            shape = y_centers.shape
        else:
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
        m = fields['magnitude'] if fields is not None else magnitude(y_centers, x_centers)
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
                AnalysisResult(raw_data=fields['divergence'] if fields is not None
                               else divergence(y_centers, x_centers), key="divergence",
                               title="divergence", desc="divergence of the vector field"),
                AnalysisResult(raw_data=fields['curl'] if fields is not None
                               else curl_2d(y_centers, x_centers), key="curl", title="curl",
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
