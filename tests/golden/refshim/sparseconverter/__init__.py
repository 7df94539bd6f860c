"""Throw-away NumPy-only stand-in for `sparseconverter` (absent in this image).
Only for importing the Python reference when generating golden vectors."""
import numpy as np
import scipy.sparse as sp

NUMPY = 'numpy'
NUMPY_MATRIX = 'numpy.matrix'
CUDA = 'cuda'
CUPY = 'cupy'
SPARSE_COO = 'sparse.COO'
SPARSE_GCXS = 'sparse.GCXS'
SPARSE_DOK = 'sparse.DOK'
SCIPY_COO = 'scipy.sparse.coo_matrix'
SCIPY_CSR = 'scipy.sparse.csr_matrix'
SCIPY_CSC = 'scipy.sparse.csc_matrix'
SCIPY_COO_ARRAY = 'scipy.sparse.coo_array'
SCIPY_CSR_ARRAY = 'scipy.sparse.csr_array'
SCIPY_CSC_ARRAY = 'scipy.sparse.csc_array'
CUPY_SCIPY_COO = 'cupyx.scipy.sparse.coo_matrix'
CUPY_SCIPY_CSR = 'cupyx.scipy.sparse.csr_matrix'
CUPY_SCIPY_CSC = 'cupyx.scipy.sparse.csc_matrix'

CPU_BACKENDS = frozenset((
    NUMPY, NUMPY_MATRIX, SPARSE_COO, SPARSE_GCXS, SPARSE_DOK, SCIPY_COO, SCIPY_CSR, SCIPY_CSC,
    SCIPY_COO_ARRAY, SCIPY_CSR_ARRAY, SCIPY_CSC_ARRAY,
))
CUPY_BACKENDS = frozenset((CUPY, CUPY_SCIPY_COO, CUPY_SCIPY_CSR, CUPY_SCIPY_CSC))
CUDA_BACKENDS = CUPY_BACKENDS | {CUDA}
BACKENDS = CPU_BACKENDS | CUDA_BACKENDS
ND_BACKENDS = frozenset((NUMPY, CUDA, CUPY, SPARSE_COO, SPARSE_GCXS, SPARSE_DOK))
D2_BACKENDS = frozenset((
    NUMPY_MATRIX, SCIPY_COO, SCIPY_CSR, SCIPY_CSC, SCIPY_COO_ARRAY, SCIPY_CSR_ARRAY,
    SCIPY_CSC_ARRAY, CUPY_SCIPY_COO, CUPY_SCIPY_CSR, CUPY_SCIPY_CSC,
))
DENSE_BACKENDS = frozenset((NUMPY, NUMPY_MATRIX, CUPY, CUDA))
SPARSE_BACKENDS = BACKENDS - DENSE_BACKENDS

ArrayBackend = str
ArrayT = object


def get_backend(arr):
    if isinstance(arr, np.ndarray):
        return NUMPY
    if sp.issparse(arr):
        fmt = arr.getformat()
        return {'coo': SCIPY_COO, 'csr': SCIPY_CSR, 'csc': SCIPY_CSC}.get(fmt)
    return None


def for_backend(arr, backend, strict=True):
    if backend == SPARSE_COO:
        import sparse
        if isinstance(arr, sparse.COO):
            return arr
        if sp.issparse(arr):
            arr = arr.toarray()
        return sparse.COO(np.asarray(arr))
    if backend in (NUMPY, CUDA):
        import sparse
        if isinstance(arr, sparse.COO):
            return arr.todense()
        if sp.issparse(arr):
            return arr.toarray()
        return np.asarray(arr)
    if backend == SCIPY_CSR:
        return sp.csr_matrix(arr)
    if backend == SCIPY_CSC:
        return sp.csc_matrix(arr)
    if backend == SCIPY_COO:
        return sp.coo_matrix(arr)
    raise NotImplementedError(backend)


def result_type(*args):
    args = [a for a in args if not (isinstance(a, str) and a in BACKENDS)]
    return np.result_type(*args)


def make_like(arr, target, strict=True):
    arr = np.asarray(arr)
    tshape = tuple(target.shape)
    if arr.shape != tshape and arr.size == int(np.prod(tshape, dtype=np.int64)):
        arr = arr.reshape(tshape)
    return arr


def conversion_cost(a, b):
    return 0. if a == b else 1.


def cheapest_pair(source_backends, target_backends):
    for s in source_backends:
        if s in target_backends:
            return (s, s)
    return (tuple(source_backends)[0], tuple(target_backends)[0])


def check_shape(arr, shape):
    pass


def get_device_class(backend):
    if backend in CPU_BACKENDS:
        return 'cpu'
    return 'cuda'
