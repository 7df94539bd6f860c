"""
UDF protocol constants.  Mirrors the reference's libertem.common.udf (common/udf.py:28-67) with one
addition: BACKEND_HIP, the MI355X-native backend.  On BACKEND_HIP tiles and `where='device'` result
buffers are `HipArray`s (device memory); `process_tile` hands their pointers to libltmi.so.

The reference anticipates exactly this slot with BACKEND_CUDA = "NumPy array, but run on CUDA
device class" (common/udf.py:45, docs/source/udf/advanced.rst:679-711); BACKEND_HIP goes one step
further and keeps the tile itself on the device.
"""
from enum import Enum

import numpy as np


class TileDepthEnum(Enum):
    TILE_DEPTH_DEFAULT = object()


class TileSizeEnum(Enum):
    TILE_SIZE_BEST_FIT = object()


class UDFMethod(Enum):
    TILE = 'tile'
    FRAME = 'frame'
    PARTITION = 'partition'


import os

NUMPY = 'numpy'
HIP = 'hip'
#: nav result rows of at most this many bytes are written by the kernels straight into the run's
#: final (page-locked) host buffer; wider rows are copied out on a copy stream (0: always copy)
HIP_DIRECT_ROW_MAX = int(os.environ.get('LTMI_DIRECT_ROW_MAX', '512'))
# names of reference backends that are accepted in `backends=` arguments and ignored
# (there is no CuPy / sparse-tile support in this build)
CUPY = 'cupy'
CUDA = 'cuda'
# the reference's sparse-tile backends (sparseconverter's identifiers, common/udf.py:46-57): a UDF may NAME them in
# `get_backends()` next to BACKEND_NUMPY -- tiles arrive as NumPy arrays / HipArrays here, never as sparse arrays
SPARSE_COO, SPARSE_GCXS, SPARSE_DOK = 'sparse.COO', 'sparse.GCXS', 'sparse.DOK'
SCIPY_COO, SCIPY_CSR, SCIPY_CSC = 'scipy.sparse.coo_matrix', 'scipy.sparse.csr_matrix', 'scipy.sparse.csc_matrix'
SCIPY_COO_ARRAY, SCIPY_CSR_ARRAY, SCIPY_CSC_ARRAY = \
    'scipy.sparse.coo_array', 'scipy.sparse.csr_array', 'scipy.sparse.csc_array'
CUPY_SCIPY_COO, CUPY_SCIPY_CSR, CUPY_SCIPY_CSC = \
    'cupyx.scipy.sparse.coo_matrix', 'cupyx.scipy.sparse.csr_matrix', 'cupyx.scipy.sparse.csc_matrix'


class UDFProtocol:
    USE_NATIVE_DTYPE = bool
    TILE_SIZE_BEST_FIT = TileSizeEnum.TILE_SIZE_BEST_FIT
    TILE_SIZE_MAX = np.inf
    TILE_DEPTH_DEFAULT = TileDepthEnum.TILE_DEPTH_DEFAULT
    TILE_DEPTH_MAX = np.inf
    BACKEND_NUMPY = NUMPY
    BACKEND_HIP = HIP
    BACKEND_CUPY = CUPY
    BACKEND_CUDA = CUDA
    BACKEND_SPARSE_COO, BACKEND_SPARSE_GCXS, BACKEND_SPARSE_DOK = SPARSE_COO, SPARSE_GCXS, SPARSE_DOK
    BACKEND_SCIPY_COO, BACKEND_SCIPY_CSR, BACKEND_SCIPY_CSC = SCIPY_COO, SCIPY_CSR, SCIPY_CSC
    BACKEND_SCIPY_COO_ARRAY, BACKEND_SCIPY_CSR_ARRAY, BACKEND_SCIPY_CSC_ARRAY = \
        SCIPY_COO_ARRAY, SCIPY_CSR_ARRAY, SCIPY_CSC_ARRAY
    BACKEND_CUPY_SCIPY_COO, BACKEND_CUPY_SCIPY_CSR, BACKEND_CUPY_SCIPY_CSC = \
        CUPY_SCIPY_COO, CUPY_SCIPY_CSR, CUPY_SCIPY_CSC
    #: every backend this build can run, in priority order
    BACKEND_ALL = (HIP, NUMPY)
    CPU_BACKENDS = frozenset((NUMPY,))
    HIP_BACKENDS = frozenset((HIP,))
    CUPY_BACKENDS = frozenset((CUPY, CUPY_SCIPY_COO, CUPY_SCIPY_CSR, CUPY_SCIPY_CSC))
    CUDA_BACKENDS = CUPY_BACKENDS | {CUDA}
    SPARSE_BACKENDS = frozenset((SPARSE_COO, SPARSE_GCXS, SPARSE_DOK, SCIPY_COO, SCIPY_CSR, SCIPY_CSC, SCIPY_COO_ARRAY,
                                 SCIPY_CSR_ARRAY, SCIPY_CSC_ARRAY, CUPY_SCIPY_COO, CUPY_SCIPY_CSR, CUPY_SCIPY_CSC))
    DENSE_BACKENDS = frozenset((NUMPY, HIP, CUPY, CUDA))
    ND_BACKENDS = frozenset((NUMPY, HIP, CUDA, CUPY, SPARSE_COO, SPARSE_GCXS, SPARSE_DOK))
    D2_BACKENDS = SPARSE_BACKENDS - ND_BACKENDS
    UDF_METHOD = UDFMethod
