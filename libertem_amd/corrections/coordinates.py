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
