"""Coordinate helpers of the reference's `libertem.utils` (utils/__init__.py:9-132) that the CoM / radial-Fourier
users of this path reach for: polar <-> cartesian vectors in (y, x) order and rotations of (y, x) vectors.
In pixel coordinates (y down, x right) a positive angle rotates clockwise."""
import numpy as np


def make_cartesian(polar):
    """[(r, phi), ...] -> [(y, x), ...]"""
    polar = np.asarray(polar)
    r, phi = polar[..., 0], polar[..., 1]
    return np.array(((np.sin(phi) * r).T, (np.cos(phi) * r).T)).T


def make_polar(cartesian):
    """[(y, x), ...] -> [(r, phi), ...] with phi = arctan2(y, x)"""
    cartesian = np.asarray(cartesian)
    r = np.linalg.norm(cartesian, axis=-1)
    phi = np.arctan2(cartesian[..., 0], cartesian[..., 1])
    return np.array((r.T, phi.T)).T


def rotate_precalc(y, x, cos_angle, sin_angle):
    """(y, x) rotated by the angle whose cosine and sine are given -> (y', x')"""
    return sin_angle * x + cos_angle * y, cos_angle * x - sin_angle * y


def rotate_rad(y, x, radians):
    return rotate_precalc(y, x, np.cos(radians), np.sin(radians))


def rotate_deg(y, x, degrees):
    return rotate_rad(y, x, np.pi / 180 * degrees)
