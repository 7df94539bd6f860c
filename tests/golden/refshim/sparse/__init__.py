"""Throw-away stand-in for pydata `sparse` (absent): class names only, so that the
reference modules import. Only dense-mask reference paths are exercised through it."""


class SparseArray:
    pass


class COO(SparseArray):
    """Just enough of pydata sparse.COO for the reference's CorrectionSet: coordinate storage,
    construction from coords or from a dense array, slicing with a tuple of contiguous slices."""

    def __init__(self, coords=None, data=None, shape=None, **kw):
        import numpy as np
        if shape is None and coords is not None and not isinstance(coords, COO) \
                and data is None:
            dense = np.asarray(coords)              # COO(dense_array)
            self.shape = dense.shape
            self.coords = np.stack(np.nonzero(dense)).astype(np.intp) if dense.ndim else \
                np.zeros((0, 0), dtype=np.intp)
            self.data = dense[np.nonzero(dense)]
            return
        if shape is None:
            raise NotImplementedError("stand-in COO needs an explicit shape")
        self.shape = tuple(int(s) for s in shape)
        self.coords = np.asarray(coords, dtype=np.intp).reshape((len(self.shape), -1))
        self.data = np.broadcast_to(np.asarray(True if data is None else data),
                                    (self.coords.shape[1],)).copy()

    @property
    def nnz(self):
        return self.coords.shape[1]

    @property
    def ndim(self):
        return len(self.shape)

    def __getitem__(self, key):
        import numpy as np
        if not isinstance(key, tuple):
            key = (key,)
        keep = np.ones(self.nnz, dtype=bool)
        starts, new_shape = [], []
        for dim, size in enumerate(self.shape):
            sl = key[dim] if dim < len(key) else slice(None)
            start, stop, step = sl.indices(size)
            assert step == 1
            keep &= (self.coords[dim] >= start) & (self.coords[dim] < stop)
            starts.append(start)
            new_shape.append(max(0, stop - start))
        coords = self.coords[:, keep] - np.asarray(starts, dtype=np.intp)[:, None]
        return COO(coords=coords, data=self.data[keep], shape=tuple(new_shape))

    def todense(self):
        import numpy as np
        out = np.zeros(self.shape, dtype=self.data.dtype if self.nnz else bool)
        if self.nnz:
            out[tuple(self.coords)] = self.data
        return out

    @classmethod
    def from_scipy_sparse(cls, m):
        raise NotImplementedError("pydata sparse is not available in this image")

    @classmethod
    def from_numpy(cls, m):
        raise NotImplementedError("pydata sparse is not available in this image")


class GCXS(SparseArray):
    pass


class DOK(SparseArray):
    pass


def concatenate(*a, **kw):
    raise NotImplementedError("pydata sparse is not available in this image")


def stack(*a, **kw):
    raise NotImplementedError("pydata sparse is not available in this image")
