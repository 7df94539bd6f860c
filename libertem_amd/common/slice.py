"""
`Slice`: origin + `Shape`, the rectangular region algebra used for partitions and tiles.
API subset of the reference's libertem.common.slice.Slice (common/slice.py:17-409).
"""
import math

import numpy as np

from .shape import Shape


class SliceUsageError(ValueError):
    pass


class Slice:
    __slots__ = ('origin', 'shape')

    def __init__(self, origin, shape):
        if not isinstance(shape, Shape):
            raise SliceUsageError("shape must be a Shape instance")
        self.origin = tuple(int(o) for o in origin)
        self.shape = shape
        if len(self.origin) != len(shape):
            raise SliceUsageError(f"origin {origin} and shape {shape} differ in dimensionality")

    @classmethod
    def from_shape(cls, shape, sig_dims):
        return cls(origin=(0,) * len(tuple(shape)), shape=Shape(tuple(shape), sig_dims=sig_dims))

    def __repr__(self):
        return f"<Slice origin={self.origin} shape={self.shape}>"

    def __hash__(self):
        return hash((self.origin, self.shape))

    def __eq__(self, other):
        return isinstance(other, Slice) and self.origin == other.origin \
            and self.shape == other.shape

    # --- algebra -----------------------------------------------------------------------------
    def clip_to(self, shape):
        """the part of this slice inside an array of `shape` anchored at the origin (common/slice.py:397-399)"""
        return self.intersection_with(Slice((0,) * shape.dims, shape))

    def intersection_with(self, other):
        """Overlap of two slices; an empty overlap yields a slice with zero-sized shape."""
        if len(self.origin) != len(other.origin):
            raise SliceUsageError(
                f"cannot intersect slices of different dimensionality ({len(self.origin)} vs {len(other.origin)})")
        if self.shape.sig_dims != other.shape.sig_dims:
            raise SliceUsageError("cannot intersect slices with different sig dims")
        lo = [max(a, b) for a, b in zip(self.origin, other.origin)]
        hi = [min(a + sa, b + sb) for a, sa, b, sb in
              zip(self.origin, self.shape, other.origin, other.shape)]
        ext = [max(0, h - l) for l, h in zip(lo, hi)]
        return Slice(origin=tuple(lo), shape=Shape(tuple(ext), sig_dims=self.shape.sig_dims))

    def is_null(self):
        return any(s == 0 for s in self.shape)

    def shift(self, other):
        """Express self relative to the origin of `other`."""
        return Slice(origin=tuple(a - b for a, b in zip(self.origin, other.origin)),
                     shape=self.shape)

    def shift_by(self, offset):
        """Move the origin by `offset` (same dimensionality, or sig-only)."""
        offset = tuple(int(o) for o in offset)
        if len(offset) == self.shape.sig_dims and len(offset) != len(self.origin):
            offset = (0,) * self.shape.nav_dims + offset
        if len(offset) != len(self.origin):
            raise SliceUsageError("offset dimensionality mismatch")
        return Slice(origin=tuple(a + b for a, b in zip(self.origin, offset)), shape=self.shape)

    def get(self, arr=None, sig_only=False, nav_only=False):
        """Index `arr` with this slice, or return the tuple of python slices."""
        if sig_only and nav_only:
            raise SliceUsageError("sig_only and nav_only are mutually exclusive")
        sl = tuple(slice(o, o + s) for o, s in zip(self.origin, self.shape))
        if sig_only:
            sl = sl[self.shape.nav_dims:]
            if arr is not None:
                return arr[(Ellipsis,) + sl]
        elif nav_only:
            sl = sl[:self.shape.nav_dims]
        if arr is not None:
            return arr[sl]
        return sl

    def discard_nav(self):
        nd = self.shape.nav_dims
        return Slice(origin=(0,) * nd + self.origin[nd:],
                     shape=Shape((0,) * nd + tuple(self.shape.sig), sig_dims=self.shape.sig_dims))

    @property
    def nav(self):
        nd = self.shape.nav_dims
        return Slice(origin=self.origin[:nd], shape=Shape(tuple(self.shape.nav), sig_dims=0))

    @property
    def sig(self):
        nd = self.shape.nav_dims
        return Slice(origin=self.origin[nd:],
                     shape=Shape(tuple(self.shape.sig), sig_dims=self.shape.sig_dims))

    def subslices(self, shape):
        """All sub-slices of extent `shape`, C order (np.ndindex), border ones clipped."""
        shape = tuple(shape)
        if len(shape) != len(self.origin):
            raise SliceUsageError("cannot create subslices with different dimensionality")
        counts = tuple(math.ceil(s1 / s) for s1, s in zip(self.shape, shape))
        sig_dims = self.shape.sig_dims
        for idx in np.ndindex(counts):
            origin = tuple(o + i * s for o, i, s in zip(self.origin, idx, shape))
            ext = tuple(min(s, so + ss - o) for s, so, ss, o in
                        zip(shape, self.origin, self.shape, origin))
            yield Slice(origin=origin, shape=Shape(ext, sig_dims=sig_dims))

    def flatten_nav(self, containing_shape):
        """Convert an n-d nav slice (whole rows of the containing nav shape) to a flat one."""
        containing_shape = tuple(containing_shape)
        nd = self.shape.nav_dims
        nav_shape = containing_shape[:nd]
        nav_origin = self.origin[:nd]
        flat_origin = int(np.ravel_multi_index(nav_origin, nav_shape)) if nd else 0
        n = 1
        for s in self.shape.nav:
            n *= s
        return Slice(origin=(flat_origin,) + self.origin[nd:],
                     shape=Shape((n,) + tuple(self.shape.sig), sig_dims=self.shape.sig_dims))

    def adjust_for_roi(self, roi):
        """Map a flat-nav slice into the compressed frame numbering of a ROI."""
        if roi is None:
            return self
        if self.shape.nav_dims != 1:
            raise SliceUsageError("adjust_for_roi needs a flat nav axis")
        flat = np.asarray(roi).reshape(-1)
        s_o = self.origin[0]
        s_s = self.shape[0]
        before = int(np.count_nonzero(flat[:s_o]))
        inside = int(np.count_nonzero(flat[s_o:s_o + s_s]))
        return Slice(origin=(before,) + self.origin[1:],
                     shape=Shape((inside,) + tuple(self.shape.sig), sig_dims=self.shape.sig_dims))

    def __getstate__(self):
        return {'origin': self.origin, 'shape': self.shape}

    def __setstate__(self, state):
        self.origin = state['origin']
        self.shape = state['shape']
