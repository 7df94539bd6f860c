"""Single-pixel virtual detector as a one-entry sparse mask (reference analysis/point.py:45-78)."""
import numpy as np

from libertem_amd.common.sparse import SparseStack
from .masks import SingleMaskAnalysis


class PointMaskAnalysis(SingleMaskAnalysis, id_="APPLY_POINT_SELECTOR"):
    def get_description(self):
        return "intensity of the integration over the selected point"

    def get_use_sparse(self):
        return True

    def get_mask_factories(self):
        if self.dataset.shape.sig.dims != 2:
            raise ValueError("can only handle 2D signals currently")
        detector_y, detector_x = self.dataset.shape.sig
        cx, cy = self.parameters['cx'], self.parameters['cy']
        sig_shape = tuple(self.dataset.shape.sig)

        def _point():
            return SparseStack(np.array([1]), np.array([0]),
                               np.array([int(cy) * detector_x + int(cx)]), 1, sig_shape)
        return [_point]

    def get_parameters(self, parameters):
        detector_y, detector_x = self.dataset.shape.sig
        return {
            'cx': parameters.get('cx', detector_x / 2),
            'cy': parameters.get('cy', detector_y / 2),
            'mask_count': 1,
            'mask_dtype': np.float32,
        }
