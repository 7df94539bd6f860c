"""Throw-away stand-in for pydata `sparse` (absent): class names only, so that the
reference modules import. Only dense-mask reference paths are exercised through it."""


class SparseArray:
    pass


class COO(SparseArray):
    def __init__(self, *a, **kw):
        raise NotImplementedError("pydata sparse is not available in this image")

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
