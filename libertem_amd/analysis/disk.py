"""Disk virtual detector (reference analysis/disk.py:43-80)."""
import numpy as np

from libertem_amd import masks
from .masks import SingleMaskAnalysis


class DiskMaskAnalysis(SingleMaskAnalysis, id_="APPLY_DISK_MASK"):
    def get_description(self):
        return "intensity of the integration over the selected disk"

    def get_mask_factories(self):
        if self.dataset.shape.sig.dims != 2:
            raise ValueError("can only handle 2D signals currently")
        detector_y, detector_x = self.dataset.shape.sig
        cx, cy, r = self.parameters['cx'], self.parameters['cy'], self.parameters['r']
        return [lambda: masks.circular(centerX=cx, centerY=cy, imageSizeX=detector_x,
                                       imageSizeY=detector_y, radius=r)]

    def get_parameters(self, parameters):
        detector_y, detector_x = self.dataset.shape.sig
        return {
            'cx': parameters.get('cx', detector_x / 2),
            'cy': parameters.get('cy', detector_y / 2),
            'r': parameters.get('r', min(detector_y, detector_x) / 2 * 0.3),
            'use_sparse': parameters.get('use_sparse', False),
            'mask_count': 1,
            'mask_dtype': np.float32,
        }
