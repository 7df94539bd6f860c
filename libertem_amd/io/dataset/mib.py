"""
MIBDataSet: Merlin / Medipix `.mib` files (`ctx.load("mib", path=...)`, reference
io/dataset/mib.py:992-1318).  Same parameters, header rules and declared dtypes as the reference; what
differs is where the bytes are decoded: the reference maps the files and runs numba decoders on the host
for every tile of every run (mib.py:401-735); here `initialize()` streams the files ONCE through pinned
bounce buffers into HBM, `ltmi_mib_decode` (csrc/ltmi_mib.hip) strips the per-frame headers and unpacks
the pixels behind each copy, and the dataset is a device-resident array from then on (288 GB of HBM hold
any single acquisition) -- every later `run_udf` starts at the kernels, ROI runs read frames in place.
A block of the scan that does not fit is STREAMED instead: each partition decodes its frames from the files
into one window of HBM when its tiles are asked for (`device_frames`), every run re-reads the files.

Formats: integer files U08 / U16 / U32 (big-endian), raw "R64" files with 1, 6, 12 or 24 bits per pixel,
single chip or 2x2 quad (1 / 6 / 12 bit), `.hdr` side file for the scan shape, files of a series found by
their numeric suffix and ordered by the sequence number of their first frame, `sync_offset`.
"""
import os
import re
import glob
import warnings

import numpy as np

from libertem_amd.common.math import prod
from libertem_amd.common.hiparray import HipArray
from .base import DataSetException, DataSetMeta
from .memory import MemoryDataSet


def read_hdr_file(path):
    """key/value pairs of the acquisition's .hdr file (reference mib.py:77-88)"""
    result = {}
    with open(path, encoding='utf-8', errors='ignore') as f:
        for line in f:
            if line.startswith("HDR") or line.startswith("End\t") or "\t" not in line:
                continue
            k, v = line.split("\t", 1)
            result[k.rstrip(':')] = v.rstrip("\n")
    return result


def is_valid_hdr(path):
    with open(path, encoding='utf-8', errors='ignore') as f:
        return f.readline().startswith("HDR")


def nav_shape_from_hdr(hdr):
    """reference mib.py:98-106"""
    if 'ScanX' in hdr and 'ScanY' in hdr:
        return (int(hdr['ScanY']), int(hdr['ScanX']))
    num_frames = int(hdr['Frames in Acquisition (Number)'])
    scan_x = int(hdr['Frames per Trigger (Number)'])
    return (num_frames // scan_x, scan_x)


def get_filenames(path, disable_glob=False):
    """all files of the series `path` belongs to (reference mib.py:109-127)"""
    if disable_glob:
        return [path]
    stem, ext = os.path.splitext(path)
    ext = ext.lower()
    if ext == '.mib':
        pattern = "%s*.mib" % re.sub(r'[0-9]+$', '', glob.escape(stem))
    elif ext == '.hdr':
        pattern = "%s*.mib" % glob.escape(stem)
    else:
        raise DataSetException("unknown extension")
    return glob.glob(pattern)


def parse_frame_header(first_bytes, filesize):
    """Fields of the first frame header of a file (reference mib.py:802-892): comma separated ASCII;
    [1] sequence number, [2] header size, [3] chips, [4] width, [5] height, [6] U08/U16/U32/R64,
    [7] layout ('   1x1', '   2x2', 'G' suffix ignored), last field: bits per pixel."""
    text = first_bytes.decode('ascii', errors='ignore')
    try:
        header_size = int(text.split(",")[2])
        parts = [p for p in text[:header_size].split(",") if '\x00' not in p]
        mib_dtype = parts[6].lower()
        kind = mib_dtype[0]
        if kind not in ('u', 'r'):
            raise ValueError(f"unknown kind: {kind}")
        height, width = int(parts[5]), int(parts[4])
        bits = int(parts[-1])
        n_chips = int(parts[3])
        lay = parts[7].replace('G', '').split('x')
        layout = (int(lay[0]), int(lay[1]))
    except (IndexError, ValueError) as e:
        raise DataSetException(f"not a .mib frame header: {e}")
    if kind == 'u':
        item = int(mib_dtype[1:]) // 8
        if item not in (1, 2, 4):
            raise DataSetException(f"unknown dtype: {mib_dtype}")
        payload = height * width * item
        declared = np.dtype(f'u{item}')
        storage = declared
    else:
        if bits not in (1, 6, 12, 24):
            raise DataSetException(f"unknown bit depth: {bits}")
        if bits == 24:
            width //= 2                  # two 12-bit images after another
        payload = height * width * {1: 1, 6: 8, 12: 16, 24: 32}[bits] // 8
        # what `dataset.dtype` reports (reference mib.py:771-787: uint64 words for 1 bit, uint16 for 12 AND
        # 24 bit) and what the decoded pixels are stored as in HBM (24 bit: float32 -- exact up to 2**24,
        # the dtype the reference reads them into for every float32-preferring UDF, and a tile dtype
        # of the matrix-core kernels)
        declared = {1: np.dtype('uint64'), 6: np.dtype('uint8'), 12: np.dtype('uint16'),
                    24: np.dtype('uint16')}[bits]
        storage = {1: np.dtype('uint8'), 6: np.dtype('uint8'), 12: np.dtype('uint16'),
                   24: np.dtype('float32')}[bits]
        if n_chips > 1:
            # raw rows of all chips side by side -> shape of the assembled detector
            px = height
            if px * layout[1] * px * layout[0] != height * width:
                raise DataSetException(f"invalid sensor layout {layout} (raw image {height}x{width})")
            height, width = px * layout[1], px * layout[0]
    return {
        'header_size_bytes': header_size, 'dtype': declared, 'storage_dtype': storage,
        'mib_dtype': mib_dtype, 'mib_kind': kind, 'bits_per_pixel': bits,
        'image_size': (height, width), 'image_size_bytes': payload,
        'sequence_first_image': int(parts[1]), 'filesize': filesize,
        'num_images': filesize // (payload + header_size), 'num_chips': n_chips,
        'sensor_layout': layout,
    }


def read_file_header(path):
    with open(path, 'rb') as f:
        size = os.fstat(f.fileno()).st_size
        return parse_frame_header(f.read(1024), size)


def get_image_count_and_sig_shape(path, disable_glob=False):
    fields = [read_file_header(fn) for fn in get_filenames(path, disable_glob)]
    if not fields:
        raise DataSetException("no files found")
    first = min(fields, key=lambda f: f['sequence_first_image'])
    return sum(f['num_images'] for f in fields), first['image_size']


_BOUNCE = {}


def _bounce_buffers(torch, nbytes):
    """two page-locked host buffers of at least `nbytes`, kept for the next load (page-locking 256 MiB
    costs ~16 ms)"""
    have = _BOUNCE.get('bufs')
    if have is None or have[0].numel() < nbytes:
        _BOUNCE['bufs'] = have = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(2)]
    return have


class MIBDataSet(MemoryDataSet):
    """
    Parameters (reference mib.py:1024-1052)
    ----------
    path : str
        the .hdr file or one of the .mib files of the series
    nav_shape : tuple of int, optional
        from the .hdr file if `path` points to it
    sig_shape : tuple of int, optional
        same number of pixels as the frames in the files
    sync_offset : int
        > 0: frames to skip at the start; < 0: blank frames inserted at the start
    disable_glob : bool
        only read `path`, not the other files with the same prefix
    num_partitions : int, optional
    shard : (rank, world), optional
        one process per GPU: decode and hold only this rank's block of the first nav axis
    """
    CHUNK_BYTES = 256 << 20          # file bytes per copy + decode step (two in flight)
    COPY_THREADS = 8
    #: decoded bytes this process may keep in HBM (None: what is free).  A block of the scan that needs more
    #: is STREAMED: no frame is decoded at load time, every partition decodes its frames from the files
    #: into a window of HBM when its tiles are asked for (partitions of at most STREAM_WINDOW_BYTES).
    MAX_RESIDENT_BYTES = None
    STREAM_WINDOW_BYTES = 4 << 30

    def __init__(self, path, tileshape=None, scan_size=None, disable_glob=False, nav_shape=None,
                 sig_shape=None, sync_offset=0, io_backend=None, num_partitions=None, shard=None):
        if io_backend is not None:
            raise ValueError("alternative I/O backends are not part of this build")
        if tileshape is not None:
            warnings.warn("tileshape argument is ignored and will be removed after 0.6.0",
                          FutureWarning)
        self._path = str(path)
        nav_shape = tuple(nav_shape) if nav_shape else None
        if scan_size is not None:
            warnings.warn("scan_size argument is deprecated. please specify nav_shape instead",
                          FutureWarning)
            if nav_shape is not None:
                raise ValueError("cannot specify both scan_size and nav_shape")
            nav_shape = tuple(scan_size)
        if nav_shape is None and not self._path.lower().endswith(".hdr"):
            raise ValueError(
                "either nav_shape needs to be passed, or path needs to point to a .hdr file")
        self._nav_arg = nav_shape
        self._sig_arg = tuple(sig_shape) if sig_shape else None
        self._sync_offset_arg = int(sync_offset)
        self._disable_glob = disable_glob
        self._num_partitions_arg = num_partitions
        self._shard_arg = shard
        self._fields = None
        self._files_sorted = None
        self._image_count = None
        self.decode_seconds = None
        self.decode_bytes = None
        self._streamed = None

    # --- host side: which files, which frames ------------------------------------------------------
    def _scan_files(self):
        filenames = get_filenames(self._path, disable_glob=self._disable_glob)
        if len(filenames) > 16384:
            warnings.warn(
                f"Saving data in many small files (here: {len(filenames)}) is not efficient, please "
                "increase the \"Images Per File\" parameter when acquiring data.", RuntimeWarning)
        files = [(fn, read_file_header(fn)) for fn in filenames]
        files.sort(key=lambda t: t[1]['sequence_first_image'])
        if not files:
            raise DataSetException("no files found")
        first = files[0][1]
        for fn, f in files:
            for key in ('header_size_bytes', 'image_size_bytes', 'mib_dtype', 'bits_per_pixel',
                        'image_size', 'num_chips'):
                if f[key] != first[key]:
                    raise DataSetException(f"{fn}: {key} = {f[key]} differs from the first file's "
                                           f"{first[key]}")
        if first['mib_kind'] == 'r' and first['num_chips'] > 1:
            if first['sensor_layout'] != (2, 2) or first['num_chips'] != 4:
                raise NotImplementedError(
                    f"No support for layout {first['sensor_layout']} yet - please contact us!")
            if first['bits_per_pixel'] == 24:
                raise NotImplementedError(
                    f"bit depth 24 not implemented for layout {first['sensor_layout']}")
        return files, first

    def initialize(self, executor):
        device = getattr(executor, 'gpu_id', None)
        if device is None:
            raise DataSetException(
                "MIBDataSet decodes the files on the GPU (ltmi_mib_decode): the executor drives none")
        files, first = self._scan_files()
        self._files_sorted, self._fields = files, first
        nav_shape = self._nav_arg
        if nav_shape is None:
            nav_shape = nav_shape_from_hdr(read_hdr_file(self._path))
        sig_shape = self._sig_arg
        if sig_shape is None:
            sig_shape = first['image_size']
        elif int(prod(sig_shape)) != int(prod(first['image_size'])):
            raise DataSetException("sig_shape must be of size: %s" % int(prod(first['image_size'])))
        n_nav = int(prod(nav_shape))
        self._image_count = sum(f['num_images'] for _, f in files)
        so = self._sync_offset_arg
        # (reference io/dataset/base/dataset.py:74: the offset lies in (-image_count, image_count); a
        # negative one of n_nav or more frames simply leaves every scan position blank)
        if not (-max(self._image_count, 1) < so < max(self._image_count, 1)):
            raise DataSetException(
                f"offset should be in ({-self._image_count}, {self._image_count}), which is "
                "(-image_count, image_count)")
        # this process's block of scan positions [p0, p1)
        local_nav = tuple(nav_shape)
        p0, p1 = 0, n_nav
        if self._shard_arg is not None:
            rank, world = int(self._shard_arg[0]), int(self._shard_arg[1])
            if nav_shape[0] % world:
                raise DataSetException(f"first nav axis {nav_shape[0]} does not split over {world} ranks")
            local_nav = (nav_shape[0] // world,) + tuple(nav_shape[1:])
            p0 = rank * int(prod(local_nav))
            p1 = p0 + int(prod(local_nav))
        self._streamed = None
        n_local = p1 - p0
        storage = np.dtype(first['storage_dtype'])
        need = n_local * int(prod(first['image_size'])) * storage.itemsize
        stride = first['header_size_bytes'] + first['image_size_bytes']
        if not self._fits_in_hbm(device, executor, need, stride, n_local):
            # a series larger than the HBM it may take: windows of it, decoded per partition
            import torch
            frame_bytes = int(prod(first['image_size'])) * storage.itemsize
            free_bytes, _ = torch.cuda.mem_get_info(device)
            window = int(min(self.STREAM_WINDOW_BYTES, max(frame_bytes, free_bytes // 4)))
            if self.MAX_RESIDENT_BYTES is not None:
                window = int(min(window, max(frame_bytes, self.MAX_RESIDENT_BYTES)))
            want = -(-need // window)
            n_parts = max(int(self._num_partitions_arg or 1), int(want))
            self._streamed = dict(device=device, executor=executor, p0=p0, sync_offset=so, key=None,
                                  frames=None)
            self.decode_seconds, self.decode_bytes = 0.0, 0
            placeholder = torch.empty(1, dtype=torch.uint8, device=f'cuda:{device}')
            frames = HipArray(placeholder, (n_local,) + tuple(first['image_size']), storage)
            MemoryDataSet.__init__(
                self, data=frames.reshape(local_nav + tuple(sig_shape)), sig_dims=len(sig_shape),
                num_partitions=min(n_parts, max(1, n_local)), shard=self._shard_arg)
        else:
            frames = self._decode_to_device(device, executor, p0, p1, so)
            MemoryDataSet.__init__(
                self, data=frames.reshape(local_nav + tuple(sig_shape)), sig_dims=len(sig_shape),
                num_partitions=self._num_partitions_arg, shard=self._shard_arg)
        self._sync_offset = so
        self._meta = DataSetMeta(shape=self._shape, raw_dtype=np.dtype(first['dtype']),
                                 sync_offset=so, image_count=self._image_count)
        return MemoryDataSet.initialize(self, executor)

    def _decode_to_device(self, device, executor, p0, p1, sync_offset):
        """scan positions [p0, p1) -> HipArray (p1 - p0, H, W) of the storage dtype"""
        import time
        import torch
        from libertem_amd import hip
        f = self._fields
        h, w = f['image_size']
        stride = f['header_size_bytes'] + f['image_size_bytes']
        quad = f['mib_kind'] == 'r' and f['num_chips'] > 1
        storage = np.dtype(f['storage_dtype'])
        n = p1 - p0
        # frame g of the series (files in sequence order) sits at scan position g - sync_offset
        g0 = max(p0 + sync_offset, 0)
        g1 = min(p1 + sync_offset, self._image_count)
        n_src = max(0, g1 - g0)
        if getattr(executor, '_make_current', None) is not None:
            executor._make_current()
        need = n * h * w * storage.itemsize
        free_bytes, _ = torch.cuda.mem_get_info(device)
        if need + 2 * min(self.CHUNK_BYTES, max(n_src, 1) * stride) > free_bytes:
            raise DataSetException(
                f"{n} decoded frames of {h}x{w} {storage} need {need / 2**30:.1f} GiB of HBM, "
                f"{free_bytes / 2**30:.1f} GiB are free on GPU {device}: fewer frames per partition "
                "(num_partitions), a part of the scan (nav_shape + sync_offset) or a shard per GPU "
                "(shard=(rank, world))")
        t0 = time.perf_counter()
        out = HipArray.empty((n, h, w), storage, device) if n_src == n else \
            HipArray.zeros((n, h, w), storage, device)          # blank frames stay zero
        if n_src > 0:
            chunk = int(max(1, min(n_src, self.CHUNK_BYTES // stride)))
            pinned = _bounce_buffers(torch, chunk * stride)
            raw = [torch.empty(chunk * stride, dtype=torch.uint8, device=f'cuda:{device}')
                   for _ in range(2)]
            free = [None, None]
            copy_stream = torch.cuda.Stream(device=device)
            copy_stream.wait_stream(torch.cuda.current_stream(device))     # (the zero fill)
            starts = np.cumsum([0] + [fl['num_images'] for _, fl in self._files_sorted])
            maps = {}
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(self.COPY_THREADS)
            for i, c0 in enumerate(range(g0, g1, chunk)):
                c1 = min(g1, c0 + chunk)
                slot = i & 1
                if free[slot] is not None:
                    free[slot].synchronize()
                host = pinned[slot].numpy()
                # file bytes of frames [c0, c1), file by file, packed at the common frame stride
                fi = int(np.searchsorted(starts, c0, side='right') - 1)
                g = c0
                while g < c1:
                    fn, fl = self._files_sorted[fi]
                    a = g - int(starts[fi])
                    b = min(fl['num_images'], a + (c1 - g))
                    if fi not in maps:
                        maps.clear()                                       # one mapping at a time
                        maps[fi] = np.memmap(fn, dtype=np.uint8, mode='r')
                    self._host_copy(pool, host, (g - c0) * stride, maps[fi], a * stride,
                                    (b - a) * stride)
                    g += b - a
                    fi += 1
                nb = (c1 - c0) * stride
                with torch.cuda.stream(copy_stream):
                    raw[slot][:nb].copy_(pinned[slot][:nb], non_blocking=True)
                    dst = out.rows(c0 - sync_offset - p0, c1 - sync_offset - p0)
                    hip.mib_decode(device, raw[slot].data_ptr(), stride, f['header_size_bytes'],
                                   f['mib_kind'], f['bits_per_pixel'], quad, c1 - c0, h, w,
                                   dst.data_ptr(), storage, stream=copy_stream.cuda_stream)
                    ev = torch.cuda.Event()
                    ev.record(copy_stream)
                    free[slot] = ev
            copy_stream.synchronize()
            pool.shutdown()
        torch.cuda.current_stream(device).synchronize()
        if self._streamed is not None:
            self.decode_seconds += time.perf_counter() - t0
            self.decode_bytes += n_src * stride
        else:
            self.decode_seconds = time.perf_counter() - t0
            self.decode_bytes = n_src * stride
        return out

    def _fits_in_hbm(self, device, executor, need, stride, n_local):
        import torch
        if getattr(executor, '_make_current', None) is not None:
            executor._make_current()
        if self.MAX_RESIDENT_BYTES is not None and need > self.MAX_RESIDENT_BYTES:
            return False
        free_bytes, _ = torch.cuda.mem_get_info(device)
        return need + 2 * min(self.CHUNK_BYTES, max(n_local, 1) * stride) <= free_bytes

    @property
    def stable_device_tiles(self):
        return self._streamed is None

    @property
    def is_streamed(self):
        """the decoded frames do not stay in HBM: every partition decodes its own from the files"""
        return self._streamed is not None

    @property
    def data(self):
        if self._streamed is not None:
            raise DataSetException(
                "this .mib series is streamed (larger than the HBM it may take): there is no resident "
                "array of its frames -- run UDFs over it, or load a part (nav_shape + sync_offset)")
        return MemoryDataSet.data.fget(self)

    def device_frames(self, local0, n):
        st = self._streamed
        if st is None:
            return MemoryDataSet.device_frames(self, local0, n)
        if st['key'] != (local0, n):
            st['frames'] = None                     # (one window at a time)
            st['key'] = None
            p = st['p0'] + local0
            st['frames'] = self._decode_to_device(st['device'], st['executor'], p, p + n,
                                                  st['sync_offset'])
            st['key'] = (local0, n)
        return st['frames'], 0

    @staticmethod
    def _host_copy(pool, dst, dst_off, src, src_off, nbytes, piece=16 << 20):
        """file mapping -> pinned buffer on several threads (one memcpy stream reads the page cache at
        ~12 GB/s, a fifth of what the host link takes)"""
        if nbytes <= piece:
            dst[dst_off:dst_off + nbytes] = src[src_off:src_off + nbytes]
            return
        try:
            # the library's copy pool: one thread per L3 domain of the host (csrc/ltmi_capi.cpp)
            from libertem_amd import hip
            hip.host_copy(dst[dst_off:dst_off + nbytes], np.asarray(src[src_off:src_off + nbytes]))
            return
        except Exception:                               # noqa: BLE001  (e.g. a source that is not contiguous)
            pass

        def run(o):
            n = min(piece, nbytes - o)
            dst[dst_off + o:dst_off + o + n] = src[src_off + o:src_off + o + n]
        list(pool.map(run, range(0, nbytes, piece)))

    # --- the reference's descriptive surface --------------------------------------------------------
    @property
    def path(self):
        return self._path

    @property
    def storage_dtype(self):
        """dtype of the decoded pixels in HBM (the declared `dtype` follows the reference)"""
        return np.dtype(self._fields['storage_dtype'])

    def get_diagnostics(self):
        f = self._fields
        return [{"name": "Bits per pixel", "value": str(f['bits_per_pixel'])},
                {"name": "Data kind", "value": str(f['mib_kind'])},
                {"name": "Layout", "value": str(f['sensor_layout'])}]

    @classmethod
    def get_supported_extensions(cls):
        return {"mib", "hdr"}

    @classmethod
    def detect_params(cls, path, executor=None):
        low = path.lower()
        if low.endswith(".mib"):
            image_count, sig_shape = get_image_count_and_sig_shape(path)
            side = int(np.sqrt(image_count))
            nav_shape = (side, side) if side * side == image_count else (image_count,)
        elif low.endswith(".hdr") and is_valid_hdr(path):
            image_count, sig_shape = get_image_count_and_sig_shape(path)
            nav_shape = nav_shape_from_hdr(read_hdr_file(path))
        else:
            return False
        return {"parameters": {"path": path, "nav_shape": nav_shape, "sig_shape": sig_shape},
                "info": {"image_count": image_count, "native_sig_shape": sig_shape}}

    def get_cache_key(self):
        return {"path": self._path, "shape": tuple(self.shape), "sync_offset": self._sync_offset}

    def __repr__(self):
        if self._fields is None:
            return f"<MIBDataSet {self._path} (not initialized)>"
        return f"<MIBDataSet of {self.dtype} shape={self.shape}>"
