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
    #: every backend this build can run, in priority order
    BACKEND_ALL = (HIP, NUMPY)
    CPU_BACKENDS = frozenset((NUMPY,))
    HIP_BACKENDS = frozenset((HIP,))
    UDF_METHOD = UDFMethod
