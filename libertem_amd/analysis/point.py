"""Single-pixel virtual detector (reference analysis/point.py:45-78): a sparse stack with one entry."""
import numpy as np

from libertem_amd.common.sparse import SparseStack
from .masks import SingleMaskAnalysis


class PointMaskAnalysis(SingleMaskAnalysis, id_="APPLY_POINT_SELECTOR"):
    WHAT = "point"
    SPARSE = True

    def geometry(self, det_y, det_x, given):
        return dict(cx=given.get('cx', det_x / 2), cy=given.get('cy', det_y / 2))

    def mask(self, p, det_y, det_x):
        pixel = int(p['cy']) * det_x + int(p['cx'])
        return SparseStack(np.array([1]), np.array([0]), np.array([pixel]), 1, (det_y, det_x))
