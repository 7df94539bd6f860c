"""Disk virtual detector (reference analysis/disk.py:43-80): centre of the detector and 30 % of its
half width unless told otherwise."""
from libertem_amd import masks
from .masks import SingleMaskAnalysis


class DiskMaskAnalysis(SingleMaskAnalysis, id_="APPLY_DISK_MASK"):
    WHAT = "disk"

    def geometry(self, det_y, det_x, given):
        return dict(cx=given.get('cx', det_x / 2), cy=given.get('cy', det_y / 2),
                    r=given.get('r', 0.3 * min(det_y, det_x) / 2))

    def mask(self, p, det_y, det_x):
        return masks.circular(centerX=p['cx'], centerY=p['cy'], imageSizeX=det_x, imageSizeY=det_y,
                              radius=p['r'])
