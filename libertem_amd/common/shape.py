"""
`Shape`: an n-d shape split into navigation and signal parts.
API subset of the reference's libertem.common.shape.Shape (common/shape.py:7-213).
"""
from .math import prod


class Shape:
    __slots__ = ('_t', '_sig_dims')

    def __init__(self, shape, sig_dims):
        self._t = tuple(map(int, shape))
        self._sig_dims = int(sig_dims)
        if self._sig_dims < 0 or self._sig_dims > len(self._t):
            raise ValueError(f"invalid sig_dims {sig_dims} for shape {self._t}")

    @classmethod
    def _trusted(cls, t, sig_dims):
        """internal: `t` is already a tuple of Python ints, `sig_dims` is valid"""
        obj = cls.__new__(cls)
        obj._t = t
        obj._sig_dims = sig_dims
        return obj

    # --- parts -------------------------------------------------------------------------------
    @property
    def nav(self):
        return NavOnlyShape._trusted(self._t[:len(self._t) - self._sig_dims], 0)

    @property
    def sig(self):
        return SigOnlyShape._trusted(self._t[len(self._t) - self._sig_dims:], self._sig_dims)

    @property
    def sig_dims(self):
        return self._sig_dims

    @property
    def nav_dims(self):
        return len(self._t) - self._sig_dims

    @property
    def dims(self):
        return len(self._t)

    @property
    def size(self):
        # (a shape without dimensions covers nothing: common/shape.py:83-99)
        return prod(self._t) if self._t else 0

    def to_tuple(self):
        return self._t

    def flatten_nav(self):
        nd = len(self._t) - self._sig_dims
        return Shape._trusted((prod(self._t[:nd]),) + self._t[nd:], self._sig_dims)

    def flatten_sig(self):
        nd = len(self._t) - self._sig_dims
        return Shape._trusted(self._t[:nd] + (prod(self._t[nd:]),), 1)

    # --- container protocol --------------------------------------------------------------------
    def __iter__(self):
        return iter(self._t)

    def __len__(self):
        return len(self._t)

    def __getitem__(self, k):
        return self._t[k]

    def __eq__(self, other):
        if isinstance(other, Shape):
            return self._t == other._t and self._sig_dims == other._sig_dims
        if isinstance(other, (tuple, list)):
            return self._t == tuple(other)
        return NotImplemented

    def __hash__(self):
        return hash((self._t, self._sig_dims))

    def __add__(self, other):
        """shape + tuple: more SIGNAL dimensions (common/shape.py:176-186)"""
        if not isinstance(other, tuple):
            return NotImplemented
        return Shape(self._t + other, sig_dims=self._sig_dims + len(other))

    def __radd__(self, other):
        """tuple + shape: more NAVIGATION dimensions, behind the ones it has (common/shape.py:188-198)"""
        if not isinstance(other, tuple):
            return NotImplemented
        nd = len(self._t) - self._sig_dims
        return Shape(self._t[:nd] + other + self._t[nd:], sig_dims=self._sig_dims)

    def __repr__(self):
        return repr(self._t)

    def __getstate__(self):
        return {'_t': self._t, '_sig_dims': self._sig_dims}

    def __setstate__(self, state):
        self._t = state['_t']
        self._sig_dims = state['_sig_dims']


class SigOnlyShape(Shape):
    __slots__ = ()

    def __init__(self, shape):
        super().__init__(shape, sig_dims=len(tuple(shape)))

    def flatten_nav(self):
        raise ValueError("a sig-only shape has no nav part")


class NavOnlyShape(Shape):
    __slots__ = ()

    def __init__(self, shape):
        super().__init__(shape, sig_dims=0)

    def flatten_sig(self):
        raise ValueError("a nav-only shape has no sig part")
