"""
DataSet / Partition / DataTile / TilingScheme / Negotiator for the in-memory data path.

Own implementation of the semantics of the reference's io/dataset/base/{dataset,partition,tiling,
tiling_scheme}.py that the hot path depends on:

* partition boundaries: np.linspace over the flattened nav axis (base/partition.py:66-99)
* tile-shape negotiation for the NumPy backend (base/tiling_scheme.py:223-526) -- reproduced so the
  CPU plumbing hands UDFs the same tiles as the reference (golden: tests/golden/tiling.npz)
* tile order: frame groups of `depth` outermost, sig slices innermost (base/tiling.py:86-239)

For BACKEND_HIP the negotiation is replaced by an MI355X policy: full frames, as many frames per
tile as fit the tile budget (the whole partition for device-resident data), because one kernel
launch per ~1 MiB tile would cost more than the kernel itself (SURVEY.md §7 "launch granularity").
"""
import math
import warnings

import os

import numpy as np

from libertem_amd.common.math import prod
from libertem_amd.common.shape import Shape
from libertem_amd.common.slice import Slice
from libertem_amd.common.udf import UDFMethod, UDFProtocol, NUMPY, HIP
from libertem_amd.common.exceptions import UDFException


class DataSetException(Exception):
    pass


class DataSetMeta:
    def __init__(self, shape, array_backends=None, image_count=None, raw_dtype=None, dtype=None,
                 metadata=None, sync_offset=0):
        self.shape = shape
        self.array_backends = array_backends
        self.image_count = image_count if image_count is not None else prod(shape.nav)
        self.raw_dtype = np.dtype(raw_dtype)
        self.dtype = np.dtype(dtype if dtype is not None else raw_dtype)
        self.metadata = metadata
        self.sync_offset = sync_offset

    def __getitem__(self, key):
        return getattr(self, key)


class DataTile:
    """A tile of data plus where it sits in the dataset (reference base/tiling.py:274-323)."""
    __slots__ = ('data', 'tile_slice', 'scheme_idx')

    def __init__(self, data, tile_slice, scheme_idx):
        self.data = data
        self.tile_slice = tile_slice
        self.scheme_idx = scheme_idx
        if tuple(data.shape) != tuple(tile_slice.shape):
            raise ValueError(f"tile data shape {data.shape} != slice shape {tile_slice.shape}")

    @property
    def shape(self):
        return self.data.shape

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def flat_data(self):
        return self.data.reshape((self.data.shape[0], -1))

    def __repr__(self):
        return f"<DataTile {self.tile_slice} scheme_idx={self.scheme_idx}>"


class TilingScheme:
    def __init__(self, slices, tileshape, dataset_shape, intent=None, debug=None):
        self._slices = list(slices)
        self._tileshape = tileshape
        self._dataset_shape = dataset_shape
        self._intent = intent
        self._debug = debug

    @classmethod
    def make_for_shape(cls, tileshape, dataset_shape, intent=None, debug=None):
        if not isinstance(tileshape, Shape):
            tileshape = Shape(tuple(tileshape), sig_dims=dataset_shape.sig.dims)
        sig_slice = Slice(origin=(0,) * dataset_shape.sig.dims,
                          shape=Shape(tuple(dataset_shape.sig), sig_dims=dataset_shape.sig.dims))
        subslices = list(sig_slice.subslices(tuple(tileshape.sig)))
        return cls(subslices, tileshape, dataset_shape, intent=intent, debug=debug)

    def adjust_for_partition(self, partition):
        """process_partition takes WHOLE partitions: with that intent the depth follows the partition's number of
        frames, whatever depth the scheme was made with (tiling_scheme.py:39-70); otherwise self"""
        n = int(partition.slice.shape.nav.size)
        if self._intent != 'partition' or n == self.depth:
            return self
        shape = Shape((n,) + tuple(self._tileshape.sig), sig_dims=self._tileshape.sig.dims)
        return TilingScheme(self._slices, shape, self._dataset_shape, intent=self._intent, debug=self._debug)

    def __getitem__(self, idx):
        return self._slices[idx]

    def __len__(self):
        return len(self._slices)

    @property
    def slices(self):
        return list(enumerate(self._slices))

    @property
    def slices_array(self):
        return np.array([(list(s.origin), list(s.shape)) for s in self._slices])

    @property
    def shape(self):
        return self._tileshape

    @property
    def dataset_shape(self):
        return self._dataset_shape

    @property
    def depth(self):
        return self._tileshape.nav[0]

    @property
    def intent(self):
        return self._intent

    def __repr__(self):
        return f"<TilingScheme (depth={self.depth}) shape={tuple(self._tileshape)} " \
               f"n_slices={len(self)} intent={self._intent}>"


class Negotiator:
    """Tile-shape negotiation (NumPy backend: reference algorithm; HIP backend: MI355X policy)."""

    #: default upper bound for one device-resident tile handed to process_tile (bytes)
    HIP_TILE_BUDGET = 32 * 2**30
    #: chunk size when frames have to be staged from host memory (bytes)
    HIP_STAGING_CHUNK = 256 * 2**20

    def get_scheme(self, udfs, dataset, read_dtype, approx_partition_shape, roi=None,
                   corrections=None, backend=NUMPY):
        if backend == HIP:
            return self._get_scheme_hip(udfs, dataset, approx_partition_shape, read_dtype,
                                        corrections)
        return self._get_scheme_numpy(udfs, dataset, read_dtype, approx_partition_shape, roi,
                                      corrections)

    # --- MI355X policy ---------------------------------------------------------------------------
    #: the HIP policy is a pure function of these few values; a run over the same dataset re-uses
    #: the (immutable) scheme instead of re-building its slices
    _hip_scheme_cache = {}

    #: streamed result export: tiles per device-resident partition and frames a tile must keep
    HIP_PIPELINE_TILES = 2
    HIP_PIPELINE_MIN_FRAMES = 32768
    #: corrected tiles are written to a device scratch buffer of at most this many bytes
    HIP_CORRECTED_CHUNK = 1 * 2**30

    def _get_scheme_hip(self, udfs, dataset, approx_partition_shape, read_dtype=None,
                        corrections=None):
        intent = self._get_intent(udfs)
        corrected = corrections is not None and corrections.have_corrections()
        if corrected and all(getattr(u, 'folds_corrections', None) is not None
                             and getattr(u, 'meta', None) is not None
                             and u.folds_corrections(corrections, u.meta) for u in udfs):
            corrected = False        # linear UDFs read the raw frames (udf/masks.py): no scratch
        forced = dataset.get_forced_tileshape()
        ds_shape = dataset.shape
        key = (intent, None if forced is None else tuple(forced), tuple(ds_shape),
               ds_shape.sig_dims, np.dtype(dataset.dtype).itemsize,
               bool(dataset.is_device_resident), int(approx_partition_shape[0]),
               self.HIP_TILE_BUDGET, self.HIP_STAGING_CHUNK,
               np.dtype(read_dtype).itemsize if corrected else 0, self.HIP_CORRECTED_CHUNK,
               self.HIP_PIPELINE_TILES, self.HIP_PIPELINE_MIN_FRAMES,
               tuple(getattr(u, 'get_hip_tile_frames', lambda: None)() for u in udfs),
               self._results_direct(udfs))
        hit = self._hip_scheme_cache.get(key)
        if hit is None:
            if len(self._hip_scheme_cache) > 64:
                self._hip_scheme_cache.clear()
            hit = self._hip_scheme_cache[key] = self._make_scheme_hip(
                intent, forced, dataset, approx_partition_shape,
                np.dtype(read_dtype).itemsize if corrected else 0, udfs)
        return hit

    @staticmethod
    def _results_direct(udfs):
        """True iff every UDF of the run has its nav rows written by the kernels straight into
        the final host buffer (small write-once rows): there is no D2H to overlap, so the
        partition is NOT split into pipelined tiles -- one launch fills the chip more evenly."""
        if os.environ.get('LTMI_HIP_PIPELINE_ALWAYS') == '1':       # experiments
            return False
        return all(bool(getattr(u, 'get_hip_direct_results', lambda: False)()) for u in udfs)

    def _make_scheme_hip(self, intent, forced, dataset, approx_partition_shape,
                         corrected_itemsize=0, udfs=()):
        ds_sig = tuple(dataset.shape.sig)
        if forced is not None and intent == 'tile':
            tileshape = tuple(forced)
        else:
            frame_bytes = prod(ds_sig) * np.dtype(dataset.dtype).itemsize
            budget = self.HIP_TILE_BUDGET if dataset.is_device_resident \
                else self.HIP_STAGING_CHUNK
            depth = max(1, min(int(approx_partition_shape[0]), budget // max(1, frame_bytes)))
            hints = [h for h in (getattr(u, 'get_hip_tile_frames', lambda: None)() for u in udfs)
                     if h]
            if dataset.is_device_resident and hints and depth > min(hints):
                # a UDF with large result rows asks for smaller tiles (more D2H / compute overlap)
                depth = max(1024, -(-min(hints) // 128) * 128)
            elif dataset.is_device_resident and depth >= 2 * self.HIP_PIPELINE_MIN_FRAMES \
                    and not self._results_direct(udfs):
                # a few tiles per partition so that the D2H of finished result rows overlaps the
                # kernels of the next tile; never fewer frames than fill the chip twice over
                n_tiles = min(self.HIP_PIPELINE_TILES, depth // self.HIP_PIPELINE_MIN_FRAMES)
                depth = -(-depth // n_tiles)
                depth = -(-depth // 128) * 128
            if corrected_itemsize:
                # corrected frames (float) go through a bounded scratch buffer
                depth = max(1, min(depth, self.HIP_CORRECTED_CHUNK //
                                   (prod(ds_sig) * corrected_itemsize)))
            if intent == 'frame':
                depth = 1
            elif intent == 'partition':
                depth = int(approx_partition_shape[0])
            tileshape = (depth,) + ds_sig
        return TilingScheme.make_for_shape(
            tileshape=Shape(tileshape, sig_dims=len(ds_sig)), dataset_shape=dataset.shape,
            intent=intent, debug={'backend': HIP})

    # --- reference algorithm ----------------------------------------------------------------------
    def _get_scheme_numpy(self, udfs, dataset, read_dtype, approx_partition_shape, roi,
                          corrections=None):
        itemsize = np.dtype(read_dtype).itemsize
        min_sig_size = dataset.get_min_sig_size()
        ds_sig_shape = tuple(dataset.shape.sig)
        need_decode = dataset.need_decode(read_dtype=read_dtype, roi=roi)
        if need_decode:
            io_max_size = 2**20
        else:
            io_max_size = itemsize * prod(approx_partition_shape)
        depth = max(self._get_min_depth(udf, approx_partition_shape) for udf in udfs)
        methods = [udf.get_method() for udf in udfs]
        if any(m in (UDFMethod.FRAME, UDFMethod.PARTITION) for m in methods):
            base_shape = ds_sig_shape
        else:
            base_shape = tuple(dataset.get_base_shape(roi))[-len(ds_sig_shape):]
        intent = self._get_intent(udfs)
        sizes = [self._get_size(io_max_size, udf, itemsize, approx_partition_shape, base_shape)
                 for udf in udfs]
        size = max(sizes) if intent == 'partition' else min(sizes)
        size_px = int(size // itemsize)
        if corrections is not None and corrections.have_corrections():
            # no excluded pixel may touch a tile boundary (tiling_scheme.py:294-301)
            base_shape = tuple(corrections.adjust_tileshape(
                tile_shape=base_shape, sig_shape=tuple(ds_sig_shape), base_shape=base_shape))
        min_factors = self._get_scale_factors(base_shape, ds_sig_shape, min_sig_size)
        min_base_shape = self._scale(base_shape, min_factors)
        max_depth = max(1, size_px // prod(min_base_shape))
        depth = min(depth, max_depth)
        full_base_shape = (1,) + tuple(base_shape)
        min_factors = (depth,) + tuple(min_factors)
        factors = self._get_scale_factors(full_base_shape, tuple(approx_partition_shape), size_px,
                                          min_factors=min_factors)
        tileshape = self._scale(full_base_shape, factors)
        tileshape = tuple(dataset.adjust_tileshape(tileshape, roi))
        if any(s > ps for s, ps in zip(tileshape[1:], ds_sig_shape)):
            raise ValueError("generated tileshape does not fit the partition")
        return TilingScheme.make_for_shape(
            tileshape=Shape(tileshape, sig_dims=len(ds_sig_shape)), dataset_shape=dataset.shape,
            intent=intent,
            debug={'size': size, 'size_px': size_px, 'need_decode': need_decode, 'depth': depth})

    @staticmethod
    def _scale(base_shape, factors):
        return tuple(int(f) * int(b) for f, b in zip(factors, base_shape))

    def _get_scale_factors(self, shape, containing_shape, size, min_factors=None):
        factors = [1] * len(shape) if min_factors is None else list(min_factors)
        max_factors = tuple(cs // s for s, cs in zip(shape, containing_shape))
        rest = size / prod(self._scale(shape, factors))
        rest = max(rest, 1)
        for idx in range(len(shape)):
            factor = int(math.floor(rest * factors[idx]))
            factor = max(factor, factors[idx])
            factor = min(factor, max_factors[idx])
            factors[idx] = factor
            rest = max(1, math.floor(size / prod(self._scale(shape, factors))))
        return factors

    def _get_intent(self, udfs):
        methods = tuple(udf.get_method() for udf in udfs)
        if any(m not in tuple(UDFMethod) for m in methods):
            raise UDFException('A UDF declared an invalid processing method')
        if UDFMethod.PARTITION in methods:
            return "partition"
        if UDFMethod.FRAME in methods:
            return "frame"
        if UDFMethod.TILE in methods:
            return "tile"
        raise ValueError('No recognized UDF method, empty udfs arg?')

    def _get_size(self, io_max_size, udf, itemsize, approx_partition_shape, base_shape):
        method = udf.get_method()
        partition_size = itemsize * prod(approx_partition_shape)
        partition_size_sig = itemsize * prod(approx_partition_shape[1:])
        if method == UDFMethod.FRAME:
            return max(2**20, partition_size_sig)
        if method == UDFMethod.PARTITION:
            return partition_size
        prefs = udf.get_tiling_preferences()
        size = prefs.get("total_size", np.inf)
        if size is UDFProtocol.TILE_SIZE_BEST_FIT:
            size = 2**20
        size = min(size, io_max_size)
        return max(itemsize * prod(base_shape), size)

    def _get_min_depth(self, udf, approx_partition_shape):
        method = udf.get_method()
        if method == UDFMethod.PARTITION:
            return approx_partition_shape[0]
        if method == UDFMethod.TILE:
            prefs = udf.get_tiling_preferences()
            depth = prefs.get("depth", UDFProtocol.TILE_DEPTH_DEFAULT)
            if depth is UDFProtocol.TILE_DEPTH_DEFAULT:
                depth = 32
            return int(min(depth, approx_partition_shape[0]))
        return 1


class Partition:
    def __init__(self, meta, partition_slice, idx):
        self.meta = meta
        self.slice = partition_slice
        self._idx = idx
        if partition_slice.shape.nav.dims != 1:
            raise ValueError("nav dims should be flat")

    @classmethod
    def make_slices(cls, shape, num_partitions):
        """(partition slice, start, stop) over the flattened nav axis; np.linspace boundaries."""
        num_frames = prod(shape.nav)
        if num_partitions > num_frames:
            warnings.warn(
                "dataset contains fewer frames than specified partitions, "
                f"setting num_partitions == num_frames == {num_frames} "
                "to avoid creating empty partitions", RuntimeWarning)
            num_partitions = num_frames
        boundaries = np.linspace(0, num_frames, num=max(2, num_partitions + 1), endpoint=True,
                                 dtype=int)
        boundaries = tuple(map(int, boundaries))
        sig = tuple(shape.sig)
        for start, stop in zip(boundaries[:-1], boundaries[1:]):
            yield (Slice(origin=(start,) + (0,) * len(sig),
                         shape=Shape((stop - start,) + sig, sig_dims=len(sig))), start, stop)

    def get_macrotile(self, dest_dtype="float32", roi=None, array_backend=None):
        """The whole partition as ONE tile -- for code that wants process_partition-style input outside a run
        (reference base/partition.py:133-170); an empty tile if the roi leaves nothing of the partition."""
        scheme = TilingScheme.make_for_shape(tileshape=self.shape, dataset_shape=self.meta.shape, intent='partition')
        kw = {} if array_backend is None else {'array_backend': array_backend}
        tiles = list(self.get_tiles(scheme, dest_dtype=dest_dtype, roi=roi, **kw))
        if len(tiles) > 1:
            raise RuntimeError("a macrotile is a single tile")
        if tiles:
            return tiles[0]
        sig = tuple(self.slice.shape.sig)
        tile_slice = Slice(origin=(self.slice.origin[0],) + (0,) * len(sig),
                           shape=Shape((0,) + sig, sig_dims=len(sig)))
        return DataTile(np.zeros(tuple(tile_slice.shape), dtype=dest_dtype), tile_slice=tile_slice, scheme_idx=0)

    @property
    def idx(self):
        return self._idx

    @property
    def dtype(self):
        return self.meta.raw_dtype

    @property
    def shape(self):
        return self.slice.shape.flatten_nav()

    def get_frame_count(self, roi=None):
        if roi is None:
            return self.slice.shape[0]
        return int(np.count_nonzero(np.asarray(roi).reshape(-1)[self.slice.get(nav_only=True)]))

    def get_locations(self):
        return None

    def get_tiles(self, tiling_scheme, dest_dtype="float32", roi=None, array_backend=NUMPY,
                  env=None, corrections=None):
        raise NotImplementedError()

    def __repr__(self):
        return f"<{type(self).__name__} idx={self._idx} slice={self.slice}>"


class RoiHelper:
    """`ds.roi[index]`: a bool nav mask with the indexed positions set (reference base/dataset.py:21-28)"""

    def __init__(self, ds):
        self._ds = ds

    def __getitem__(self, k):
        roi = np.zeros(tuple(self._ds.shape.nav), dtype=bool)
        roi[k] = True
        return roi


class DataSet:
    def __init__(self):
        self._meta = None

    @property
    def roi(self):
        return RoiHelper(self)

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop('_udf_plans', None)       # cached run plans (udf/base.py) stay in this process
        return d

    def initialize(self, executor):
        return self

    @property
    def meta(self):
        return self._meta

    @property
    def dtype(self):
        raise NotImplementedError()

    @property
    def shape(self):
        raise NotImplementedError()

    @property
    def array_backends(self):
        return (NUMPY,)

    @property
    def is_device_resident(self):
        return False

    def get_partitions(self):
        raise NotImplementedError()

    def get_num_partitions(self):
        raise NotImplementedError()

    def get_slices(self):
        yield from Partition.make_slices(shape=self.shape, num_partitions=self.get_num_partitions())

    def get_base_shape(self, roi):
        return (1,) + (1,) * (self.shape.sig.dims - 1) + (self.shape.sig[-1],)

    def get_forced_tileshape(self):
        return None

    def adjust_tileshape(self, tileshape, roi):
        return tileshape

    def need_decode(self, read_dtype, roi):
        # reference base/backend.py:69-119: roi or dtype conversion prevent zero-copy views
        if roi is not None:
            return True
        return np.dtype(self.meta.raw_dtype) != np.dtype(read_dtype)

    def get_min_sig_size(self):
        return 4 * 4096 // np.dtype(self.meta.raw_dtype).itemsize

    def get_correction_data(self):
        return None

    def check_valid(self):
        return True

    def get_diagnostics(self):
        """format-specific [{'name': ..., 'value': ...}, ...] (reference base/dataset.py:198-204)"""
        return []

    @property
    def diagnostics(self):
        """what every dataset can say about itself, after the format's own entries (reference base/dataset.py:177-196)"""
        parts = list(self.get_partitions())
        info = self.get_sync_offset_info()
        return list(self.get_diagnostics()) + [
            {"name": "Partition shape", "value": str(tuple(parts[0].shape)) if parts else "-"},
            {"name": "Number of partitions", "value": str(len(parts))},
            {"name": "Number of frames skipped at the beginning", "value": info["frames_skipped_start"]},
            {"name": "Number of frames ignored at the end", "value": info["frames_ignored_end"]},
            {"name": "Number of blank frames inserted at the beginning", "value": info["frames_inserted_start"]},
            {"name": "Number of blank frames inserted at the end", "value": info["frames_inserted_end"]},
        ]

    def get_sync_offset_info(self):
        """frames skipped / ignored / inserted by `sync_offset` (reference base/dataset.py:70-88)"""
        so = getattr(self, '_sync_offset_arg', None)
        so = int(so if so is not None else (getattr(self, '_sync_offset', 0) or 0))
        n_nav = prod(self.shape.nav)
        n_img = getattr(self, '_image_count', None)
        n_img = int(n_img if n_img is not None else n_nav)
        return {
            "frames_skipped_start": max(0, so),
            "frames_ignored_end": max(0, n_img - n_nav - so),
            "frames_inserted_start": abs(min(0, so)),
            "frames_inserted_end": max(0, n_nav - n_img + so),
        }
