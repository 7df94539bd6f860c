"""Throw-away stand-in for `numba` (absent): jit decorators are the identity so the
reference's njit kernels run as interpreted Python. Only for golden-vector generation."""
from . import typed, cuda, core  # noqa

__version__ = "0.0.stub"
prange = range


def _deco(*args, **kwargs):
    if len(args) >= 1 and callable(args[0]):
        return args[0]

    def wrap(fn):
        return fn
    return wrap


njit = _deco
jit = _deco
vectorize = _deco
guvectorize = _deco
generated_jit = _deco


def get_num_threads():
    return 1


def set_num_threads(n):
    pass


class _Types:
    def __getattr__(self, k):
        return k


types = _Types()
int64 = 'int64'
float32 = 'float32'
float64 = 'float64'
boolean = 'boolean'
