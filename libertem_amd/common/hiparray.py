"""
HipArray: a C-contiguous (row-strided) n-d array in MI355X HBM.

PyTorch-ROCm is used only as the allocator / stream provider (a `torch.Tensor` owns the
memory); the numpy dtype is tracked separately because torch has no arithmetic on
uint16/uint32/uint64 and the kernels do their own conversion anyway.
"""
import numpy as np

from .math import prod

_TORCH_EQUIV = {
    np.dtype('uint16'): 'int16', np.dtype('uint32'): 'int32', np.dtype('uint64'): 'int64',
}


def _torch():
    import torch
    return torch


def torch_dtype_for(dtype):
    torch = _torch()
    dtype = np.dtype(dtype)
    name = _TORCH_EQUIV.get(dtype, dtype.name)
    if name == 'bool':
        return torch.bool
    return getattr(torch, name)


class HipArray:
    """
    shape/dtype follow numpy conventions.  The memory is the first `prod(shape[1:])` elements of
    each of `shape[0]` rows that are `ld` elements apart (ld == prod(shape[1:]) when fully
    contiguous) -- i.e. exactly the (n_frames, n_px, ld) triple libltmi takes.
    """
    __slots__ = ('_t', 'shape', 'dtype', 'ld')
    PINNED_DOWNLOAD_MIN = 256 * 1024        # bytes: larger downloads go through page-locked memory

    def __init__(self, tensor, shape, dtype, ld=None):
        self._t = tensor                    # torch tensor whose data_ptr() is element [0, ...]
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        inner = prod(self.shape[1:]) if len(self.shape) > 0 else 1
        self.ld = inner if ld is None else int(ld)

    # --- construction -----------------------------------------------------------------------
    @classmethod
    def zeros(cls, shape, dtype, device):
        torch = _torch()
        shape = tuple(int(s) for s in shape)
        t = torch.zeros(shape if shape else (1,), dtype=torch_dtype_for(dtype),
                        device=f'cuda:{int(device)}')
        return cls(t, shape, dtype)

    @classmethod
    def empty(cls, shape, dtype, device):
        torch = _torch()
        shape = tuple(int(s) for s in shape)
        t = torch.empty(shape if shape else (1,), dtype=torch_dtype_for(dtype),
                        device=f'cuda:{int(device)}')
        return cls(t, shape, dtype)

    @classmethod
    def from_numpy(cls, arr, device, non_blocking=False):
        torch = _torch()
        arr = np.ascontiguousarray(arr)
        dt = arr.dtype
        if dt in _TORCH_EQUIV:
            arr = arr.view(_TORCH_EQUIV[dt])
        t = torch.from_numpy(arr).to(f'cuda:{int(device)}', non_blocking=non_blocking)
        return cls(t, arr.shape, dt)

    @classmethod
    def from_torch(cls, tensor, dtype=None):
        """Wrap a contiguous CUDA/HIP tensor.  `dtype` overrides the numpy dtype (e.g. uint16 for
        an int16 tensor holding unsigned detector counts)."""
        if not tensor.is_cuda:
            raise ValueError("HipArray.from_torch needs a device tensor")
        if not tensor.is_contiguous():
            raise ValueError("HipArray.from_torch needs a contiguous tensor")
        if dtype is None:
            dtype = np.dtype(str(tensor.dtype).replace('torch.', ''))
        dtype = np.dtype(dtype)
        if dtype.itemsize != tensor.element_size():
            raise ValueError(f"dtype {dtype} does not match tensor element size")
        return cls(tensor, tuple(tensor.shape), dtype)

    # --- properties -------------------------------------------------------------------------
    @property
    def device(self):
        return self._t.device.index

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return prod(self.shape)

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def is_contiguous(self):
        return self.ld == (prod(self.shape[1:]) if self.shape else 1)

    def data_ptr(self):
        return self._t.data_ptr()

    @property
    def torch(self):
        """The backing tensor (storage dtype; unsigned 16/32/64 appear as signed)."""
        return self._t

    # --- views --------------------------------------------------------------------------------
    def rows(self, start, stop):
        """View of rows [start, stop) along axis 0 (always valid, keeps `ld`)."""
        start, stop = int(start), int(stop)
        if not (0 <= start <= stop <= self.shape[0]):
            raise IndexError(f"rows [{start}, {stop}) out of range for {self.shape}")
        flat = self._t.reshape(-1)
        return HipArray(flat[start * self.ld:], (stop - start,) + self.shape[1:], self.dtype,
                        ld=self.ld)

    def reshape(self, shape):
        shape = tuple(int(s) for s in shape)
        if -1 in shape:
            known = prod(s for s in shape if s != -1)
            shape = tuple(self.size // known if s == -1 else s for s in shape)
        if prod(shape) != self.size:
            raise ValueError(f"cannot reshape {self.shape} to {shape}")
        if self.is_contiguous:
            return HipArray(self._t, shape, self.dtype)
        # row-strided: only reshapes that keep axis 0 are views
        if shape and shape[0] == self.shape[0]:
            return HipArray(self._t, shape, self.dtype, ld=self.ld)
        raise ValueError("cannot reshape a row-strided HipArray across its first axis")

    def sig_rows(self, row_start, row_stop):
        """For a (n, H, W...) array: the sub-tile of sig rows [row_start, row_stop) (full width),
        as a row-strided view -- what a `(depth, rows, W)` tile of the reference is."""
        if self.ndim < 2:
            raise ValueError("sig_rows needs at least 2 dims")
        inner = prod(self.shape[2:])
        flat = self._t.reshape(-1)
        return HipArray(flat[row_start * inner:],
                        (self.shape[0], row_stop - row_start) + self.shape[2:], self.dtype,
                        ld=self.ld)

    def contiguous(self):
        if self.is_contiguous:
            return self
        torch = _torch()
        inner = prod(self.shape[1:])
        flat = self._t.reshape(-1)
        v = torch.as_strided(flat, (self.shape[0], inner), (self.ld, 1))
        return HipArray(v.contiguous(), self.shape, self.dtype)

    # --- host transfer -------------------------------------------------------------------------
    def cpu(self):
        """Copy to a numpy array of the logical dtype and shape (synchronises)."""
        torch = _torch()
        inner = prod(self.shape[1:]) if self.shape else 1
        n0 = self.shape[0] if self.shape else 1
        flat = self._t.reshape(-1)
        src = flat[:n0 * inner] if self.is_contiguous else \
            torch.as_strided(flat, (n0, inner), (self.ld, 1))
        if src.numel() * src.element_size() >= self.PINNED_DOWNLOAD_MIN and src.is_contiguous():
            # one D2H at link speed into page-locked memory (the runtime stages a pageable
            # destination at ~1/8 of it; page-locking costs less than that even for a single use)
            pinned = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
            pinned.copy_(src, non_blocking=True)
            torch.cuda.current_stream(src.device).synchronize()
            host = pinned.numpy()
        else:
            host = src.cpu().numpy()
        if host.dtype != self.dtype:
            host = host.view(self.dtype)
        return host.reshape(self.shape)

    def __array__(self, dtype=None, copy=None):
        a = self.cpu()
        return a if dtype is None else a.astype(dtype)

    def fill_(self, value):
        if not self.is_contiguous:
            raise ValueError("fill_ needs a contiguous HipArray")
        self._t.reshape(-1)[:self.size].fill_(value)
        return self

    def __repr__(self):
        return f"<HipArray shape={self.shape} dtype={self.dtype} ld={self.ld} dev={self.device}>"


class HipRowsArray(HipArray):
    """
    Frames of a flat device array selected by a row list -- logical shape (n, *sig), NO copy: what a
    region of interest of a device-resident dataset looks like to the kernels that can read frames
    through a row list (`ltmi_apply_masks_rows`).  Everything else calls `materialize()` (the frames
    gathered into a contiguous HipArray with `ltmi_gather_rows`, cached).
    `base`: HipArray (n_total, ...) whose rows are frames; `idx64` / `rows32`: device tensors with the
    same frame numbers (int64 for the gather kernel, int32 for the row-list kernels).
    """
    __slots__ = ('_base', '_idx64', '_rows32', '_mat')

    def __init__(self, base, idx64, rows32, sig):
        HipArray.__init__(self, base._t, (int(idx64.shape[0]),) + tuple(sig), base.dtype,
                          ld=base.ld)
        self._base, self._idx64, self._rows32, self._mat = base, idx64, rows32, None

    @property
    def base(self):
        return self._base

    def rows_ptr(self):
        return self._rows32.data_ptr()

    def data_ptr(self):
        raise TypeError("HipRowsArray has no contiguous data: use materialize() or a row-list kernel")

    @property
    def torch(self):
        raise TypeError("HipRowsArray has no contiguous data: use materialize()")

    def rows(self, start, stop):
        start, stop = int(start), int(stop)
        if not (0 <= start <= stop <= self.shape[0]):
            raise IndexError(f"rows [{start}, {stop}) out of range for {self.shape}")
        return HipRowsArray(self._base, self._idx64[start:stop], self._rows32[start:stop],
                            self.shape[1:])

    def reshape(self, shape):
        return self.materialize().reshape(shape)

    def cpu(self):
        return self.materialize().cpu()

    def materialize(self, stream=None):
        if self._mat is None:
            from libertem_amd import hip
            n = self.shape[0]
            out = HipArray.empty(self.shape, self.dtype, self._base.device)
            if n:
                isz = self.dtype.itemsize
                hip.gather_rows(self._base.device, self._base.data_ptr(), self._base.ld * isz,
                                self._idx64.data_ptr(), n, prod(self.shape[1:]) * isz,
                                out.data_ptr(), stream=stream)
            self._mat = out
        return self._mat


class HostMappedArray(HipArray):
    """
    Rows of a page-locked HOST buffer that the kernels write directly over the host link
    (zero-copy): the final place of small write-once result rows -- no device buffer, no D2H copy,
    no copy stream; a stream synchronisation makes them readable on the host.
    `host` is the NumPy view of the rows (storage dtype may be the signed twin of an unsigned
    dtype), `dev_ptr` the device address of its first byte.
    """
    __slots__ = ('_host', '_dev_ptr', '_dev')

    def __init__(self, host, dev_ptr, device, shape, dtype, ld=None):
        self._t = None
        self._host = host
        self._dev_ptr = int(dev_ptr)
        self._dev = int(device)
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        inner = prod(self.shape[1:]) if len(self.shape) > 0 else 1
        self.ld = inner if ld is None else int(ld)

    @property
    def device(self):
        return self._dev

    def data_ptr(self):
        return self._dev_ptr

    @property
    def torch(self):
        raise TypeError("a host-mapped result array has no device tensor")

    def rows(self, start, stop):
        start, stop = int(start), int(stop)
        if not (0 <= start <= stop <= self.shape[0]):
            raise IndexError(f"rows [{start}, {stop}) out of range for {self.shape}")
        flat = self._host.reshape(-1)
        return HostMappedArray(flat[start * self.ld:], self._dev_ptr + start * self.ld *
                               self.dtype.itemsize, self._dev, (stop - start,) + self.shape[1:],
                               self.dtype, ld=self.ld)

    def reshape(self, shape):
        shape = tuple(int(s) for s in shape)
        if -1 in shape:
            known = prod(s for s in shape if s != -1)
            shape = tuple(self.size // known if s == -1 else s for s in shape)
        if prod(shape) != self.size:
            raise ValueError(f"cannot reshape {self.shape} to {shape}")
        if self.is_contiguous or (shape and shape[0] == self.shape[0]):
            return HostMappedArray(self._host, self._dev_ptr, self._dev, shape, self.dtype,
                                   ld=None if self.is_contiguous else self.ld)
        raise ValueError("cannot reshape a row-strided array across its first axis")

    def sig_rows(self, row_start, row_stop):
        raise TypeError("host-mapped arrays hold result rows, not frames")

    def contiguous(self):
        if self.is_contiguous:
            return self
        raise TypeError("host-mapped result rows are contiguous by construction")

    def cpu(self):
        """The rows as a NumPy array (a VIEW of the page-locked buffer; the caller must have
        synchronised the stream the kernels ran on)."""
        n = self.size
        host = self._host.reshape(-1)[:n]
        if host.dtype != self.dtype:
            host = host.view(self.dtype)
        return host.reshape(self.shape)

    def fill_(self, value):
        self.cpu()[...] = value
        return self

    def __repr__(self):
        return f"<HostMappedArray shape={self.shape} dtype={self.dtype} ld={self.ld} dev={self._dev}>"
