"""Ring virtual detector (reference analysis/ring.py:43-81): outer radius = half the detector, inner
radius = 80 % of the outer one unless told otherwise."""
from libertem_amd import masks
from .masks import SingleMaskAnalysis


class RingMaskAnalysis(SingleMaskAnalysis, id_="APPLY_RING_MASK"):
    WHAT = "ring"

    def geometry(self, det_y, det_x, given):
        outer = given.get('ro', min(det_y, det_x) / 2)
        return dict(cx=given.get('cx', det_x / 2), cy=given.get('cy', det_y / 2),
                    ri=given.get('ri', 0.8 * outer), ro=outer)

    def mask(self, p, det_y, det_x):
        return masks.ring(centerX=p['cx'], centerY=p['cy'], imageSizeX=det_x, imageSizeY=det_y,
                          radius=p['ro'], radius_inner=p['ri'])
