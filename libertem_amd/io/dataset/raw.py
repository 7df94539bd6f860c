"""
RawFileDataSet: flat binary files (`ctx.load("raw", path=..., dtype=..., nav_shape=..., sig_shape=...)`,
reference io/dataset/raw.py:62-240).  The file is memory-mapped and handed to the same streaming
machinery as a host-resident MemoryDataSet: frames go to the GPU in their NATIVE dtype through the
double-buffered `hipMemcpyAsync` stager (io/dataset/memory.py), conversion happens in the kernels.
"""
import os
import warnings

import numpy as np

from libertem_amd.common.math import prod
from .base import DataSetException
from .memory import MemoryDataSet


def _reopen(kwargs):
    return RawFileDataSet(**kwargs)


class RawFileDataSet(MemoryDataSet):
    """
    Parameters (reference raw.py:77-104)
    ----------
    path : str
    dtype : numpy dtype of the file (any byte order)
    nav_shape, sig_shape : tuple of int
        (`scan_size` / `detector_size` are accepted as deprecated aliases)
    sync_offset : int
        > 0: that many frames are skipped at the start; < 0: that many blank frames are inserted at
        the start.  Frames missing at the end are blank (zeros), as in the reference.
    num_partitions : int, optional
    """

    def __init__(self, path, dtype, scan_size=None, detector_size=None, enable_direct=False,
                 detector_size_raw=None, crop_detector_to=None, tileshape=None, nav_shape=None,
                 sig_shape=None, sync_offset=0, io_backend=None, num_partitions=None, shard=None):
        # `io_backend` / `enable_direct` choose HOW the reference reads the file (mmap, buffered, O_DIRECT); here the
        # file is always memory-mapped and its frames go to the GPU through the upload stager: accepted, not used
        if enable_direct and io_backend is not None:
            raise ValueError("can't specify io_backend and enable_direct at the same time")
        if enable_direct:
            warnings.warn("enable_direct is deprecated; pass `io_backend=DirectBackend()` instead", FutureWarning)
        if tileshape is not None:
            warnings.warn("tileshape argument is ignored and will be removed after 0.6.0", FutureWarning)
            tileshape = None
        if crop_detector_to is not None:
            warnings.warn("crop_detector_to and detector_size_raw are deprecated, and will be removed after version "
                          "0.6.0. please specify sig_shape instead or use a more specific DataSet like EMPAD",
                          FutureWarning)
            if detector_size is not None:
                raise ValueError("cannot specify both detector_size and crop_detector_to")
            if detector_size_raw != crop_detector_to:
                raise ValueError("RawFileDataSet can't crop detector anymore, please use EMPAD DataSet")
            detector_size = crop_detector_to
        if scan_size is not None:
            warnings.warn("scan_size argument is deprecated. please specify nav_shape instead",
                          FutureWarning)
            if nav_shape is not None:
                raise ValueError("cannot specify both scan_size and nav_shape")
            nav_shape = scan_size
        if detector_size is not None:
            warnings.warn("detector_size argument is deprecated. please specify sig_shape instead",
                          FutureWarning)
            if sig_shape is not None:
                raise ValueError("cannot specify both detector_size and sig_shape")
            sig_shape = detector_size
        if nav_shape is None:
            raise TypeError("missing 1 required argument: 'nav_shape'")
        if sig_shape is None:
            raise TypeError("missing 1 required argument: 'sig_shape'")
        nav_shape = tuple(int(x) for x in nav_shape)
        sig_shape = tuple(int(x) for x in sig_shape)
        dt = np.dtype(dtype)
        self._path = path
        try:
            filesize = os.stat(path).st_size
        except OSError as e:
            raise DataSetException(f"could not open file {path}: {e}")
        frame_bytes = prod(sig_shape) * dt.itemsize
        if prod(sig_shape) > filesize // dt.itemsize:
            raise DataSetException("sig_shape must be less than size: %s" % (filesize // dt.itemsize))
        n_file = filesize // frame_bytes
        n_nav = prod(nav_shape)
        sync_offset = int(sync_offset)
        if not (-n_nav < sync_offset < max(n_file, 1)):
            raise DataSetException(
                f"offset should be in ({-n_nav}, {n_file}), which is (-image_count, image_count)")
        skip = max(0, sync_offset)
        lead_blank = max(0, -sync_offset)
        avail = max(0, min(n_file - skip, n_nav - lead_blank))
        if avail > 0:
            mm = np.memmap(path, dtype=dt, mode='r', offset=skip * frame_bytes,
                           shape=(avail,) + sig_shape)
        else:
            mm = np.zeros((0,) + sig_shape, dtype=dt)
        if lead_blank == 0 and avail == n_nav:
            data = mm                                   # the common case: zero-copy view of the file
        else:
            data = np.zeros((n_nav,) + sig_shape, dtype=dt)
            data[lead_blank:lead_blank + avail] = mm
        self._sync_offset_arg = sync_offset
        self._image_count = int(n_file)
        super().__init__(data=data.reshape(nav_shape + sig_shape), sig_dims=len(sig_shape),
                         num_partitions=num_partitions, shard=shard)
        self._meta.image_count = int(n_file)             # (the frames in the FILE: reference raw.py:185-190)
        if lead_blank or avail < n_nav:
            self._valid_frames = (lead_blank, lead_blank + avail)      # (positions that hold a frame of the file)
        self._ctor = dict(path=path, dtype=dt.str, nav_shape=nav_shape, sig_shape=sig_shape, sync_offset=sync_offset,
                          num_partitions=num_partitions, shard=shard)

    def __reduce__(self):
        # a pickle names the file, it does not carry its frames (reference tests/io/datasets/test_raw.py
        # test_pickle_is_small); plans and uploads are made again where it is loaded
        return (_reopen, (self._ctor,))

    @property
    def path(self):
        return self._path

    def get_diagnostics(self):
        return [{"name": "dtype", "value": str(self._meta.raw_dtype)}]

    def get_cache_key(self):
        # (what identifies the data a result was computed from: reference io/dataset/raw.py:246-254)
        return {"path": self._path, "shape": tuple(self.shape), "dtype": str(self.dtype),
                "sync_offset": self._sync_offset_arg}

    def __repr__(self):
        return f"<RawFileDataSet {self._path} shape={tuple(self.shape)} dtype={self.dtype}>"
