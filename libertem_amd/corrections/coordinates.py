"""
2x2 coordinate transforms in (y, x) order, as used by CoM's `apply_correction`.
Same functions as the reference's libertem.corrections.coordinates (corrections/coordinates.py).
"""
import numpy as np


def scale(factor):
    return np.eye(2) * factor


def rotate(radians):
    # (y, x) ordering: clockwise for y pointing down
    c, s = np.cos(radians), np.sin(radians)
    return np.array([(c, s), (-s, c)])


def rotate_deg(degrees):
    return rotate(np.pi/180*degrees)


def flip_y():
    return np.array([(-1, 0), (0, 1)])


def flip_x():
    return np.array([(1, 0), (0, -1)])


def identity():
    return np.eye(2)


def scale_rotate_flip_y(mat):
    """(scale, rotation in radians, flip_y) of a matrix made as scale() @ rotate() @ flip_y(); ValueError for
    unequal axis scales or shear (corrections/coordinates.py:57-93)."""
    mat = np.asarray(mat, dtype=float)
    sy, sx = np.linalg.norm(mat[:, 0]), np.linalg.norm(mat[:, 1])
    if not np.allclose(sy, sx):
        raise ValueError(f'y scale {sy} and x scale {sx} are different.')
    unit = mat / sy
    det = unit[0, 0] * unit[1, 1] - unit[0, 1] * unit[1, 0]      # +1: rotation, -1: rotation after a flip
    if not np.allclose(abs(det), 1.):
        raise ValueError(f'Contains shear: flip factor (2D cross product) is {det}.')
    flipped = bool(det < 0)
    rot = unit.copy()
    rot[:, 0] *= det                                             # undo flip_y
    angle = np.arctan2(-rot[1, 0], rot[0, 0])
    other = np.arctan2(rot[0, 1], rot[1, 1])
    if not np.allclose((np.sin(angle), np.cos(angle)), (np.sin(other), np.cos(other))):
        raise ValueError(f'Rotation angle 1 {angle} and rotation angle 2 {other} are inconsistent.')
    return (sy, angle, flipped)
