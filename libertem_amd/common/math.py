"""Integer helpers of the shape / buffer algebra (reference common/math.py)."""
import math

import numpy as np

_INTEGERS = (int, bool, np.bool_, np.integer)


def prod(iterable):
    """Product of integer sizes in Python integers (no overflow); anything that is not an integer or a bool
    -- floats, complex numbers -- is a ValueError (common/math.py:20-35)."""
    r = 1
    for x in iterable:
        if not isinstance(x, _INTEGERS):
            raise ValueError(f"prod() takes integers, not {type(x).__name__}")
        r *= int(x)
    return r


def count_nonzero(array):
    try:
        return int(np.count_nonzero(array))
    except (TypeError, ValueError):              # (sparse arrays without a count_nonzero of their own)
        return int(array.astype(bool).sum())


def flat_nonzero(array):
    return array.flatten().nonzero()[0]


def make_2D_square(shape):
    """(n,) -> (sqrt(n), sqrt(n)) when n is a square number, everything else unchanged; a size below 1 is a
    ValueError (common/math.py:47-76)"""
    if len(shape) != 1:
        return shape
    size = prod(shape)
    if size < 1:
        raise ValueError('Zero or negative shape.size')
    root = math.isqrt(size)
    return (root, root) if root * root == size else shape
