"""Ring virtual detector (reference analysis/ring.py:43-81)."""
import numpy as np

from libertem_amd import masks
from .masks import SingleMaskAnalysis


class RingMaskAnalysis(SingleMaskAnalysis, id_="APPLY_RING_MASK"):
    def get_description(self):
        return "intensity of the integration over the selected ring"

    def get_mask_factories(self):
        if self.dataset.shape.sig.dims != 2:
            raise ValueError("can only handle 2D signals currently")
        detector_y, detector_x = self.dataset.shape.sig
        p = self.parameters
        cx, cy, ri, ro = p['cx'], p['cy'], p['ri'], p['ro']
        return [lambda: masks.ring(centerX=cx, centerY=cy, imageSizeX=detector_x,
                                   imageSizeY=detector_y, radius=ro, radius_inner=ri)]

    def get_parameters(self, parameters):
        detector_y, detector_x = self.dataset.shape.sig
        ro = parameters.get('ro', min(detector_y, detector_x) / 2)
        return {
            'cx': parameters.get('cx', detector_x / 2),
            'cy': parameters.get('cy', detector_y / 2),
            'ri': parameters.get('ri', ro * 0.8),
            'ro': ro,
            'use_sparse': parameters.get('use_sparse', False),
            'mask_count': 1,
            'mask_dtype': np.float32,
        }
