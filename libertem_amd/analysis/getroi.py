"""Region of interest from an analysis' `roi` parameter (reference analysis/getroi.py:4-27): a bool nav mask from
{'shape': 'disk', 'cx', 'cy', 'r'} or {'shape': 'rect', 'x', 'y', 'width', 'height'}; no / an empty spec -> None."""
from libertem_amd import masks


def get_roi(params, shape):
    spec = params.get("roi") if params is not None else None
    if not spec or "shape" not in spec:
        return None
    ny, nx = tuple(shape)
    kind = spec["shape"]
    if kind == "disk":
        return masks.circular(spec["cx"], spec["cy"], nx, ny, spec["r"])
    if kind == "rect":
        return masks.rectangular(spec["x"], spec["y"], spec["width"], spec["height"], nx, ny)
    raise NotImplementedError("unknown shape %s" % kind)
