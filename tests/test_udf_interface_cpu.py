"""The UDF interface as user code sees it, cases of the reference's own interface tests re-expressed against this
package (tests/udf/test_aux_data.py, test_coords.py, test_simple_udf.py, test_auto.py, tests/common/test_bufferwrapper.py,
test_math.py, tests/test_masks.py): what `preprocess` / `get_task_data` can see, whole partitions for
`process_partition`, the errors for outdated or impossible UDFs, `ds.roi[...]`, `Context.map` / AutoUDF, sparse
mask stacks as arrays, `sync_offset` of a MemoryDataSet.  NumPy UDFs on the inline executor: no GPU."""
import functools

import numpy as np
import pytest

from libertem_amd.api import Context
from libertem_amd.executor.inline import InlineJobExecutor
from libertem_amd.udf.base import UDF
from libertem_amd.udf.auto import AutoUDF
from libertem_amd.common.udf import UDFMethod
from libertem_amd.common.exceptions import UDFException
from libertem_amd.common import Shape, Slice, SliceUsageError
from libertem_amd.common.math import prod, make_2D_square, count_nonzero
from libertem_amd.common.buffers import PlaceholderBufferWrapper, reshaped_view
from libertem_amd.io.dataset.memory import MemoryDataSet
from libertem_amd import masks as M


@pytest.fixture(params=['inline', pytest.param('hip', marks=pytest.mark.gpu)])
def lt_ctx(request):
    """the inline executor (CPU), and -- under `-m gpu` -- the HIP executor, whose merge / delivery code of its own
    serves NumPy UDFs of users as well"""
    if request.param == 'hip':
        from libertem_amd.executor.hip import HipJobExecutor
        ctx = Context(executor=HipJobExecutor())
        yield ctx
        ctx.close()
    else:
        yield Context(executor=InlineJobExecutor(debug=True))


class EchoUDF(UDF):
    """what the hooks see of a nav-kind aux buffer (test_aux_data.py:10-39)"""

    def get_result_buffers(self):
        return {k: self.buffer(kind="nav", dtype="float32", extra_shape=(2,))
                for k in ('echo', 'echo_preprocess', 'echo_postprocess')} | \
            {'weighted': self.buffer(kind="nav", dtype="float32")}

    def preprocess(self):
        self.results.echo_preprocess[:] = self.params.aux          # the whole task's rows, on both sides

    def process_frame(self, frame):
        assert self.params.aux.shape == (2,)
        self.results.echo[:] = self.params.aux
        self.results.weighted[:] = np.sum(frame) * self.params.aux[0]

    def postprocess(self):
        self.results.echo_postprocess[:] = self.params.aux


@pytest.mark.parametrize('with_roi', [False, True])
def test_aux_data_in_every_hook(lt_ctx, with_roi):
    rng = np.random.default_rng(3)
    data = rng.random((16, 16, 8, 8)).astype(np.float32)
    aux = rng.random((16, 16, 2)).astype(np.float32)
    ds = lt_ctx.load("memory", data=data, tileshape=(7, 8, 8), num_partitions=2, sig_dims=2)
    roi = rng.random((16, 16)) < 0.5 if with_roi else None
    udf = EchoUDF(aux=EchoUDF.aux_data(kind="nav", data=aux, dtype="float32", extra_shape=(2,)), other_stuff=object())
    res = lt_ctx.run_udf(dataset=ds, udf=udf, roi=roi)
    sel = slice(None) if roi is None else roi
    for k in ('echo', 'echo_preprocess', 'echo_postprocess'):
        assert np.array_equal(res[k].raw_data, aux[sel].reshape(-1, 2)), k
    assert np.allclose(res['weighted'].raw_data, (data.sum(axis=(2, 3)) * aux[..., 0])[sel].reshape(-1), rtol=1e-5)


class CoordsInGetTaskData(UDF):
    """`meta.slice` / `meta.coordinates` describe the whole partition until the first tile (test_coords.py:177-213)"""

    def get_result_buffers(self):
        return {'counter': self.buffer(kind='single', dtype=np.int64),
                'first': self.buffer(kind='nav', dtype=np.int64, extra_shape=(2,))}

    def process_tile(self, tile):
        self.results.first[:] = self.meta.coordinates

    def get_task_data(self):
        assert self.meta.slice is not None
        c = self.meta.coordinates
        assert c.shape == (self.meta.slice.shape[0], 2)
        return {'ps': np.zeros((c.shape[0],), dtype=bool), 'coords': c}

    def postprocess(self):
        self.results.counter[0] += self.task_data.ps.shape[0]
        assert np.array_equal(self.results.first, self.task_data.coords)     # the tiles' coordinates add up to it

    def merge(self, dest, src):
        dest.counter += src.counter
        dest.first[:] = src.first


def test_coordinates_available_before_the_first_tile(lt_ctx):
    data = np.zeros((16, 16, 4, 4), dtype=np.float32)
    ds = lt_ctx.load("memory", data=data, tileshape=(7, 4, 4), num_partitions=2, sig_dims=2)
    roi = np.random.default_rng(8).choice([True, False], (16, 16))
    res = lt_ctx.run_udf(dataset=ds, udf=CoordsInGetTaskData(), roi=roi)
    assert res['counter'].data[0] == np.count_nonzero(roi)
    assert np.array_equal(res['first'].raw_data, np.argwhere(roi))


class WholePartition(UDF):
    def get_result_buffers(self):
        return {'sums': self.buffer(kind='nav', dtype=np.float64), 'depths': self.buffer(kind='nav', dtype=np.int64)}

    def process_partition(self, partition):
        assert self.meta.slice.shape[0] == self.meta.partition_shape[0]          # every frame of the partition
        assert partition.shape[1:] == tuple(self.meta.sig_slice.shape)
        self.results.sums[:] += partition.sum(axis=(1, 2))
        self.results.depths[:] = partition.shape[0]


@pytest.mark.parametrize('tileshape', [(3, 3, 7), (15, 3, 7), (4, 1, 7)])
def test_process_partition_takes_whole_partitions_whatever_the_tileshape(lt_ctx, tileshape):
    """test_simple_udf.py:661-745: a depth forced on the dataset does not cut the partitions of process_partition"""
    data = np.random.default_rng(2).random((30, 3, 7)).astype(np.float32)
    ds = MemoryDataSet(data=data, tileshape=tileshape, num_partitions=2, sig_dims=2).initialize(lt_ctx.executor)
    res = lt_ctx.run_udf(dataset=ds, udf=WholePartition())
    assert np.all(res['depths'].data == 15)
    assert np.allclose(res['sums'].data, data.astype(np.float64).sum(axis=(1, 2)))


def test_outdated_tile_attributes_name_their_replacement(lt_ctx):
    class Old1(UDF):
        def get_result_buffers(self):
            return {}

        def process_tile(self, tile):
            tile.scheme_idx

    class Old2(Old1):
        def process_tile(self, tile):
            tile.tile_slice
    ds = lt_ctx.load('memory', data=np.ones((2, 2, 4, 4)))
    with pytest.raises(AttributeError, match='self.meta.tiling_scheme_idx'):
        lt_ctx.run_udf(dataset=ds, udf=Old1())
    with pytest.raises(AttributeError, match='self.meta.slice'):
        lt_ctx.run_udf(dataset=ds, udf=Old2())


@pytest.mark.parametrize('method', [42, UDFMethod.FRAME, UDFMethod.PARTITION, UDFMethod.TILE])
def test_get_method_must_name_an_implemented_method(lt_ctx, method):
    class Bad(UDF):
        def __init__(self, method):
            super().__init__(method=method)

        def get_method(self):
            return self.params.method

        def get_result_buffers(self):
            return {}
    ds = lt_ctx.load('memory', data=np.ones((2, 2, 4, 4)))
    with pytest.raises(UDFException):
        lt_ctx.run_udf(dataset=ds, udf=Bad(method=method))


def test_dataset_roi_helper_and_private_copy(lt_ctx):
    class PerFrame(UDF):
        def get_result_buffers(self):
            return {'intensity': self.buffer(kind='nav', dtype=np.float32)}

        def process_frame(self, frame):
            self.results.intensity[:] = frame.sum()
    data = np.arange(4, dtype=np.float32).reshape(2, 2, 1, 1) * np.ones((2, 2, 4, 4), dtype=np.float32)
    ds = lt_ctx.load('memory', data=data)
    roi = ds.roi[0, 1]
    assert roi.shape == (2, 2) and roi.dtype == bool and roi.sum() == 1 and roi[0, 1]
    assert ds.roi[:, 1].sum() == 2
    res = lt_ctx.run_udf(dataset=ds, udf=PerFrame(), roi=roi)
    before = res['intensity'].data.copy()
    roi[:] = ds.roi[:, :]                                   # the caller reuses its array: the result must not move
    assert np.array_equal(np.isnan(res['intensity'].data), np.isnan(before))
    assert res['intensity'].data[0, 1] == 16. and np.isnan(res['intensity'].data[0, 0])


def test_map_and_auto_udf(lt_ctx):
    rng = np.random.default_rng(6)
    data = rng.random((16, 8, 32, 64)).astype(np.float32)
    ds = MemoryDataSet(data=data, tileshape=(8, 32, 64), num_partitions=2, sig_dims=2).initialize(lt_ctx.executor)
    got = lt_ctx.map(dataset=ds, f=functools.partial(np.sum, axis=-1))       # the BUFFER, not a dict (api.py:1670)
    assert got.data.shape == (16, 8, 32) and np.allclose(got.data, data.sum(axis=-1), rtol=1e-5)

    def weird(frame):
        return ["Shape %s" % str(frame.shape), dict(shape=frame.shape, sum=frame.sum()), lambda x: x, MemoryDataSet]
    item = lt_ctx.map(dataset=ds, f=weird).data[0, 0]
    assert len(item) == 4 and isinstance(item[0], str) and isinstance(item[1], dict) and item[2](1) == 1

    for roi in (None, rng.random((16, 8)) < 0.5):
        udf = AutoUDF(f=functools.partial(np.sum, axis=-1), monitor=True)
        n = 0
        for res in lt_ctx.run_udf_iter(dataset=ds, udf=udf, roi=roi):
            valid = np.flatnonzero(res.damage.raw_data.reshape(-1))
            last = valid[-1] if len(valid) else 0
            assert np.allclose(res.buffers[0]['result'].raw_data[last], res.buffers[0]['monitor'].data)
            n += 1
        assert n == 2


def test_placeholder_buffers_and_reshaped_view():
    buf = PlaceholderBufferWrapper(kind='sig', dtype=np.float32)
    with pytest.raises(ValueError, match="doesn't have a value"):
        np.array(buf)
    with pytest.raises(ValueError):
        buf.raw_data
    assert buf.get_view_for_partition(None) is None and not buf.has_data()
    data = np.zeros((2, 5))
    with pytest.raises(AttributeError):
        reshaped_view(data[:, :3], (-1,))
    v = reshaped_view(data, (-1,))
    v[3] = 7
    assert data[0, 3] == 7 and v.shape == (10,)


def test_math_helpers():
    assert prod([]) == 1 and prod((1, 2, 3)) == 6 and prod((-11, 2, 3)) == -66
    assert prod((2**32, 2**32, 2**32)) == 2**96
    assert prod(np.array((2**62, 2**62, 2**62), dtype=np.int64)) == 2**186
    assert prod(Shape((1, 2, 3), sig_dims=1).nav) == 2 and prod((3, True)) == 3
    for bad in ((1., 2, False), np.array((1., 1 + 2j, 1))):
        with pytest.raises(ValueError):
            prod(bad)
    assert make_2D_square((16,)) == (4, 4) and make_2D_square((15,)) == (15,) and make_2D_square((1,)) == (1, 1)
    assert make_2D_square((4, 4)) == (4, 4) and make_2D_square(()) == ()
    with pytest.raises(ValueError):
        make_2D_square((0,))
    assert count_nonzero(np.eye(3)) == 3 and count_nonzero(np.zeros(4)) == 0


def test_sparse_stacks_as_arrays():
    """tests/test_masks.py:39-66: `stack[i].todense()`, `bins.sum(axis=0).todense()` -- soft radial bins add up to 1"""
    stack = M.sparse_template_multi_stack(mask_index=(0, 1, 2), offsetY=(13, 14, 15), offsetX=(15, 14, 13),
                                          template=np.ones((2, 3)), imageSizeY=32, imageSizeX=32)
    for i, (y, x) in enumerate([(13, 15), (14, 14), (15, 13)]):
        want = np.zeros((32, 32))
        want[y:y + 2, x:x + 3] = 1
        assert np.array_equal(stack[i].todense(), want)
    assert np.array_equal(stack[-1].todense(), stack[2].todense()) and stack[1:].shape == (2, 32, 32)
    with pytest.raises(IndexError):
        stack[3]
    bins = M.radial_bins(35, 37, 80, 80, n_bins=42)
    assert bins.shape == (42, 80, 80)
    total = bins.sum(axis=0).todense()
    inside = np.hypot(*(np.mgrid[0:80, 0:80] - np.array([37, 35]).reshape(2, 1, 1))) < 30
    assert total.shape == (80, 80) and np.allclose(total[inside], 1)
    assert np.allclose(bins.sum(axis=(1, 2)), bins.todense().sum(axis=(1, 2)))
    assert np.isclose(bins.sum(), bins.todense().sum())


def test_memory_dataset_sync_offset(lt_ctx):
    """frame g of the data sits at scan position g - sync_offset; positions a sync_offset leaves without a frame are
    not delivered to host UDFs at all (the reference's tests/udf/test_coords.py test_tiles_positive_offset /
    _negative_offset; io/dataset/base/partition.py read ranges) -- their result rows keep their initial value"""
    class PerTile(UDF):
        def get_result_buffers(self):
            return {'s': self.buffer(kind='nav', dtype=np.float32), 'seen': self.buffer(kind='single', dtype=np.int64)}

        def process_tile(self, tile):
            assert tile.shape[0] == len(self.meta.coordinates)
            self.results.s[:] = tile.sum(axis=(1, 2)) + 1                 # (+ 1: a delivered zero frame would show)
            self.results.seen[0] += tile.shape[0]

        def merge(self, dest, src):
            dest.s[:] = src.s
            dest.seen += src.seen
    data = np.arange(64, dtype=np.float32).reshape(8, 8, 1, 1) * np.ones((8, 8, 2, 2), dtype=np.float32)
    want = {62: [249., 253.] + [0.] * 62, -62: [0.] * 62 + [1., 5.], 3: [4. * g + 1 for g in range(3, 64)] + [0.] * 3}
    for so, w in want.items():
        ds = MemoryDataSet(data=data, tileshape=(4, 2, 2), num_partitions=2, sig_dims=2, sync_offset=so)
        ds = ds.initialize(lt_ctx.executor)
        res = lt_ctx.run_udf(dataset=ds, udf=PerTile())
        assert np.array_equal(res['s'].data.reshape(-1), np.array(w, dtype=np.float32)), so
        assert res['seen'].data[0] == 64 - abs(so)
    roi = np.arange(64).reshape(8, 8) % 3 == 0                            # 22 positions, the last one without a frame
    ds = MemoryDataSet(data=data, num_partitions=2, sig_dims=2, sync_offset=3).initialize(lt_ctx.executor)
    res = lt_ctx.run_udf(dataset=ds, udf=PerTile(), roi=roi)
    assert res['seen'].data[0] == 21 and res['s'].raw_data[-1] == 0 and res['s'].raw_data[0] == 13.
    with pytest.raises(Exception, match='offset should be in'):
        MemoryDataSet(data=data, sig_dims=2, sync_offset=64)

def test_slices_of_different_dimensionality_do_not_intersect():
    s1 = Slice(origin=(1, 1, 1, 1), shape=Shape((2, 2, 2, 2), sig_dims=2))
    s2 = Slice(origin=(1, 1, 1), shape=Shape((2, 2, 2), sig_dims=2))
    s3 = Slice(origin=(1, 1, 1, 1), shape=Shape((2, 2, 2, 2), sig_dims=1))
    for other in (s2, s3):
        with pytest.raises(SliceUsageError):
            s1.intersection_with(other)


def test_partition_macrotile():
    """Partition.get_macrotile (base/partition.py:133-170; tests/io/datasets/test_mem.py:26-43): the whole partition
    as one tile, whatever tileshape the dataset forces; with a roi its selected frames, possibly none"""
    data = np.random.default_rng(0).random((16, 16, 16, 16)).astype(np.float32)
    ds = MemoryDataSet(data=data, tileshape=(16, 16, 16), num_partitions=2).initialize(None)
    p0, p1 = list(ds.get_partitions())
    mt = p0.get_macrotile()
    assert tuple(mt.shape) == (128, 16, 16) and mt.data.dtype == np.float32
    assert np.array_equal(mt.data, data.reshape(256, 16, 16)[:128]) and mt.tile_slice == p0.slice
    roi = np.zeros((16, 16), bool)
    roi[0, 3] = roi[15, 0] = True
    one = p0.get_macrotile(roi=roi, dest_dtype=np.float64)
    assert tuple(one.shape) == (1, 16, 16) and one.data.dtype == np.float64 and np.array_equal(one.data[0], data[0, 3])
    none = p1.get_macrotile(roi=ds.roi[0, 0])
    assert tuple(none.shape) == (0, 16, 16)


def test_dataset_diagnostics_and_cache_key(lt_ctx, tmp_path):
    """`ds.diagnostics` = the format's own entries + what every dataset can say (base/dataset.py:70-88, 177-204);
    `RawFileDataSet.get_cache_key` (io/dataset/raw.py:246-254)"""
    import json
    path = str(tmp_path / 'x.raw')
    np.arange(10 * 4 * 4, dtype=np.float32).tofile(path)
    want = {0: (0, 2, 0, 0), 2: (2, 0, 0, 0), -3: (0, 5, 3, 0)}
    for so, (skipped, ignored, ins_start, ins_end) in want.items():
        ds = lt_ctx.load('raw', path=path, dtype='float32', nav_shape=(2, 4), sig_shape=(4, 4), sync_offset=so)
        info = ds.get_sync_offset_info()
        assert (info['frames_skipped_start'], info['frames_ignored_end'], info['frames_inserted_start'],
                info['frames_inserted_end']) == (skipped, ignored, ins_start, ins_end)
        d = {e['name']: e['value'] for e in ds.diagnostics}
        assert d['dtype'] == 'float32' and d['Number of partitions'] == str(len(list(ds.get_partitions())))
        assert d['Number of frames skipped at the beginning'] == skipped
        assert d['Number of blank frames inserted at the beginning'] == ins_start
        key = ds.get_cache_key()
        assert json.loads(json.dumps(key)) == {"path": path, "shape": [2, 4, 4, 4], "dtype": "float32", "sync_offset": so}
    mem = lt_ctx.load('memory', data=np.zeros((4, 4, 2, 2)), sync_offset=3)
    assert mem.get_sync_offset_info() == {'frames_skipped_start': 3, 'frames_ignored_end': 0,
                                          'frames_inserted_start': 0, 'frames_inserted_end': 3}
    assert mem.get_diagnostics() == []


def test_raw_file_dataset_arguments(lt_ctx, tmp_path):
    """tests/io/datasets/test_raw.py re-expressed: frames in the FILE are `meta.image_count` (extra data at the end is
    cut off, missing frames are blank), a sig_shape larger than the file is refused, the messages of missing
    arguments, deprecated arguments warn, reader back-ends are accepted (the file is always mapped), small pickles"""
    import pickle
    from libertem_amd.io.dataset.base import DataSetException
    path = str(tmp_path / '8x8x8x8')
    data = np.random.default_rng(4).random((8, 8, 8, 8)).astype(np.float32)
    data.tofile(path)
    ds = lt_ctx.load("raw", path=path, nav_shape=(8, 8), sig_shape=(4, 4), dtype="float32")
    assert ds._meta.image_count == 256 and tuple(ds.shape) == (8, 8, 4, 4)
    ds = lt_ctx.load("raw", path=path, nav_shape=(8, 8), sig_shape=(3, 3), dtype="float32", io_backend=object())
    assert ds._meta.image_count == 455
    assert np.array_equal(np.asarray(ds.data).reshape(-1), data.reshape(-1)[:64 * 9])
    many = lt_ctx.load("raw", path=path, nav_shape=(10, 8), sig_shape=(8, 8), dtype="float32")
    assert many._meta.image_count == 64 and many.get_sync_offset_info()['frames_inserted_end'] == 16
    assert not np.asarray(many.data)[8:].any()
    with pytest.raises(DataSetException, match='sig_shape must be less than size'):
        lt_ctx.load("raw", path=path, nav_shape=(8, 8), sig_shape=(65, 65), dtype="float32")
    with pytest.raises(TypeError, match="missing 1 required argument: 'sig_shape'"):
        lt_ctx.load("raw", path=path, nav_shape=(8, 8), dtype="float32")
    with pytest.raises(TypeError, match="missing 1 required argument: 'nav_shape'"):
        lt_ctx.load("raw", path=path, sig_shape=(8, 8), dtype="float32")
    with pytest.warns(FutureWarning, match='scan_size'):
        old = lt_ctx.load("raw", path=path, scan_size=(8, 8), sig_shape=(8, 8), dtype="float32")
    with pytest.warns(FutureWarning, match='enable_direct'):
        lt_ctx.load("raw", path=path, nav_shape=(8, 8), sig_shape=(8, 8), dtype="float32", enable_direct=True)
    with pytest.raises(ValueError, match="can't crop"):
        with pytest.warns(FutureWarning):
            lt_ctx.load("raw", path=path, nav_shape=(8, 8), dtype="float32", detector_size_raw=(8, 8), crop_detector_to=(4, 4))
    blob = pickle.dumps(old)
    assert len(blob) < 2 * 1024
    again = pickle.loads(blob)
    assert tuple(again.shape) == (8, 8, 8, 8) and np.array_equal(np.asarray(again.data), data)


def test_context_argument_handling(lt_ctx):
    """tests/test_context.py re-expressed: an empty list of UDFs, ROIs that are not bool arrays (other dtypes warn
    and are cast; coordinate lists ((y, x), value); a single position; scipy matrices), `make_with('inline')` takes no
    resources, cancellation names the partitions merged so far, the mixin protocols of `libertem.udf`"""
    import scipy.sparse as sp
    from libertem_amd.common.exceptions import ExecutorSpecException
    from libertem_amd.common.executor import JobCancelledError
    from libertem_amd.udf import UDFRunCancelled, UDFFrameMixin, UDFTileMixin, UDFPreprocessMixin

    class PerFrame(UDFFrameMixin, UDF):
        def get_result_buffers(self):
            return {'s': self.buffer(kind='nav', dtype=np.float32)}

        def process_frame(self, frame):
            self.results.s[:] = frame.sum()
    data = np.random.default_rng(1).random((4, 5, 3, 3)).astype(np.float32)
    ds = lt_ctx.load('memory', data=data, num_partitions=2)
    with pytest.raises(ValueError, match="^empty list of UDFs - nothing to do!$"):
        lt_ctx.run_udf(dataset=ds, udf=[])
    with pytest.raises(ValueError, match="^empty list of UDFs - nothing to do!$"):
        list(lt_ctx.run_udf_iter(dataset=ds, udf=[]))
    roi = np.zeros((4, 5), dtype=bool)
    roi[3, 2] = True
    want = lt_ctx.run_udf(dataset=ds, udf=PerFrame(), roi=roi)['s'].raw_data
    assert want.shape == (1,) and np.isclose(want[0], data[3, 2].sum())
    for dtype in (int, float):
        with pytest.warns(UserWarning, match=f"ROI dtype is {np.dtype(dtype)}, expected bool. Attempting cast to bool."):
            got = lt_ctx.run_udf(dataset=ds, udf=PerFrame(), roi=roi.astype(dtype))['s'].raw_data
        assert np.array_equal(got, want)
    for roi_in in (sp.coo_matrix(roi), sp.csr_matrix(roi), (((3, 2), True),), [[[3, 2], True]], (3, 2)):
        assert np.array_equal(lt_ctx.run_udf(dataset=ds, udf=PerFrame(), roi=roi_in)['s'].raw_data, want)
    inverse = lt_ctx.run_udf(dataset=ds, udf=PerFrame(), roi=(((3, 2), False),))['s'].raw_data
    assert inverse.shape == (19,)
    with pytest.raises(ValueError, match='more than one truth value'):
        lt_ctx.run_udf(dataset=ds, udf=PerFrame(), roi=(((3, 2), True), ((0, 0), False)))

    for kw in ({'cpus': 4}, {'gpus': 4}):
        with pytest.raises(ExecutorSpecException):
            Context.make_with('inline', **kw)
    with pytest.raises(ExecutorSpecException):
        Context.make_with('not_an_executor')
    assert isinstance(Context.make_with('inline').executor, InlineJobExecutor)

    class Cancels(UDF):
        def get_result_buffers(self):
            return {'stuff': self.buffer(kind='nav', dtype='float32')}

        def process_frame(self, frame):
            if self.meta.coordinates[0][0] >= 2:
                raise JobCancelledError()
    with pytest.raises(UDFRunCancelled, match=r"^UDF run cancelled after 1 partitions$"):
        for _ in lt_ctx.run_udf_iter(dataset=ds, udf=Cancels()):
            pass
    assert isinstance(PerFrame(), UDFFrameMixin) and not isinstance(PerFrame(), UDFTileMixin)
    assert not isinstance(PerFrame(), UDFPreprocessMixin) and PerFrame().get_method() == UDFMethod.FRAME


def test_backend_names_of_the_reference_can_be_declared(lt_ctx):
    """common/udf.py:43-75: a UDF may name the reference's sparse / CuPy tile back-ends beside NumPy in `get_backends`
    (docs/source/udf/advanced.rst); tiles still arrive as NumPy arrays"""
    class Declares(UDF):
        def get_backends(self):
            return (self.BACKEND_SCIPY_CSR, self.BACKEND_SPARSE_GCXS, self.BACKEND_CUPY, self.BACKEND_NUMPY)

        def get_result_buffers(self):
            return {'s': self.buffer(kind='nav', dtype=np.float32)}

        def process_tile(self, tile):
            assert isinstance(tile, np.ndarray) and self.meta.array_backend == self.BACKEND_NUMPY
            self.results.s[:] = tile.sum(axis=(1, 2))
    data = np.random.default_rng(0).random((3, 4, 5, 5)).astype(np.float32)
    ds = lt_ctx.load('memory', data=data)
    assert np.allclose(lt_ctx.run_udf(dataset=ds, udf=Declares())['s'].data, data.sum(axis=(2, 3)), rtol=1e-5)
    assert UDF.BACKEND_SCIPY_CSR in UDF.SPARSE_BACKENDS and UDF.BACKEND_NUMPY in UDF.DENSE_BACKENDS
    assert UDF.BACKEND_SCIPY_CSC in UDF.D2_BACKENDS and UDF.BACKEND_SPARSE_COO in UDF.ND_BACKENDS
    assert UDF.BACKEND_CUPY_SCIPY_CSR in UDF.CUDA_BACKENDS


def test_libertem_import_alias():
    """libertem_amd.compat.install(): scripts written against the reference's module names import this package's
    modules (the same objects), a module this package does not have is an ImportError; in a subprocess -- the alias is
    per process"""
    import subprocess
    import sys
    code = '''
import numpy as np
import libertem_amd.compat as compat
assert compat.install() and compat.install()
from libertem.api import Context
from libertem.udf import UDF, UDFRunCancelled, UDFTileMixin
from libertem.udf.masks import ApplyMasksUDF
from libertem.udf.sum import SumUDF
from libertem.common.buffers import BufferWrapper, reshaped_view
from libertem.common import Shape, Slice
from libertem.io.dataset.memory import MemoryDataSet
from libertem.executor.inline import InlineJobExecutor
from libertem.analysis.com import guess_corrections
import libertem.masks, libertem_amd.masks, libertem_amd.api
assert Context is libertem_amd.api.Context and libertem.masks is libertem_amd.masks
try:
    import libertem.web
    raise SystemExit("libertem.web should not exist")
except ImportError:
    pass

class PerFrame(UDF):
    def get_result_buffers(self):
        return {"s": self.buffer(kind="nav", dtype=np.float32)}
    def process_frame(self, frame):
        self.results.s += frame.sum()

ctx = Context(executor=InlineJobExecutor())
data = np.arange(2 * 3 * 4 * 4, dtype=np.float32).reshape(2, 3, 4, 4)
ds = ctx.load("memory", data=data, num_partitions=2)
res = ctx.run_udf(dataset=ds, udf=[PerFrame(), SumUDF()])
assert np.array_equal(res[0]["s"].data, data.sum(axis=(2, 3))) and np.array_equal(res[1]["intensity"].data, data.sum(axis=(0, 1)))
compat.uninstall()
print("alias ok")
'''
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', code], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=300)
    assert r.returncode == 0 and 'alias ok' in r.stdout, r.stderr[-2000:]
