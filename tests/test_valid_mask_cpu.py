"""Valid masks of result buffers and the damage map of partial results (`run_udf_iter`), the cases
of the reference's tests/udf/test_valid_mask.py re-expressed against this package: `meta.get_valid_nav_mask()` in
`merge` / `get_results`, `UDF.with_mask`, `BufferWrapper.valid_mask / masked_data / raw_masked_data /
valid_slice_bounding / get_valid_slice_inner / make_default_mask` (common/buffers.py:195-633,
udf/base.py:561-593, :1226-1267, :2340-2380).  NumPy UDFs on the inline executor: no GPU."""
import numpy as np
import pytest

from libertem_amd.api import Context
from libertem_amd.executor.inline import InlineJobExecutor
from libertem_amd.common.shape import Shape
from libertem_amd.common.math import prod
from libertem_amd.udf.base import UDF
from libertem_amd.io.dataset.memory import MemoryDataSet
from libertem_amd.common.buffers import (
    BufferWrapper, InvalidMaskError, ArrayWithMask, get_inner_slice, get_bbox, get_bbox_slice,
)


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


def _ds(ctx, datashape=(16, 16, 32, 32), num_partitions=4):
    ds = MemoryDataSet(datashape=list(datashape), num_partitions=num_partitions)
    return ds.initialize(ctx.executor)


class SeesNavMask(UDF):
    def get_result_buffers(self):
        return {
            'sig_sum': self.buffer(kind='sig', dtype=np.float32),
            'per_frame': self.buffer(kind='nav', dtype=np.float32),
            'scalar': self.buffer(kind='single', dtype=np.float32, extra_shape=(1,)),
        }

    def get_results(self):
        vm = self.meta.get_valid_nav_mask()
        assert vm is not None
        assert vm.sum() > 0, "get_results is not called with an empty valid nav mask"
        assert len(vm.shape) == 1, "valid_nav_mask should be flattened"
        if self.meta.roi is not None:
            assert vm.shape[0] == np.count_nonzero(self.meta.roi), \
                "with a roi, the valid nav mask is compressed to it by default"
            full = self.meta.get_valid_nav_mask(full_nav=True)
            assert full.shape[0] == prod(self.meta.dataset_shape.nav)
            assert np.array_equal(full[np.asarray(self.meta.roi).reshape(-1)], vm)
        return super().get_results()

    def process_frame(self, frame):
        assert self.meta.get_valid_nav_mask() is None
        assert self.meta.get_valid_nav_mask(full_nav=True) is None
        self.results.sig_sum += frame
        self.results.per_frame[:] = frame.sum()
        self.results.scalar[:] = frame.sum()

    def merge(self, dest, src):
        vm = self.meta.get_valid_nav_mask()
        assert vm is not None
        assert not np.all(vm), "the mask during merge holds what is merged ALREADY: never everything"
        dest.sig_sum += src.sig_sum
        dest.scalar += src.scalar
        dest.per_frame[:] = src.per_frame


@pytest.mark.parametrize('roi_kind', [None, 'block', 'random'])
def test_valid_nav_mask_available(lt_ctx, roi_kind):
    ds = _ds(lt_ctx)
    roi = None
    if roi_kind == 'block':
        roi = np.zeros((16, 16), dtype=bool)
        roi[4:-4, 4:-4] = True
    elif roi_kind == 'random':
        roi = np.random.default_rng(5).choice([True, False], size=(16, 16))
    seen = []
    for res in lt_ctx.run_udf_iter(dataset=ds, udf=SeesNavMask(), roi=roi):
        nav = res.buffers[0]['per_frame']
        assert np.array_equal(nav.valid_mask, res.damage.data)       # the default of a nav buffer IS the damage
        assert nav.valid_mask.shape == nav.data.shape == (16, 16)
        seen.append(int(np.count_nonzero(res.damage.data)))
    assert seen == sorted(seen) and seen[-1] == (256 if roi is None else np.count_nonzero(roi))


class MasksOfItsOwn(UDF):
    def get_result_buffers(self):
        return {
            'everything': self.buffer(kind='sig', dtype=np.float32),
            'nothing': self.buffer(kind='sig', dtype=np.float32),
            'default_nav': self.buffer(kind='nav', dtype=np.float32),
            'default_nav_extra': self.buffer(kind='nav', dtype=np.float32, extra_shape=(2,)),
            'half_plane': self.buffer(kind='single', dtype=np.float32, extra_shape=(64, 64)),
        }

    def get_results(self):
        right_half = np.zeros((64, 64), dtype=bool)
        right_half[:, 32:] = True
        return {
            'everything': self.with_mask(self.results.everything, mask=True),
            'nothing': self.with_mask(self.results.nothing, mask=False),
            'default_nav': self.results.default_nav,
            'default_nav_extra': self.results.default_nav_extra,
            'half_plane': self.with_mask(self.results.half_plane, mask=right_half),
        }

    def process_frame(self, frame):
        self.results.everything += frame
        self.results.nothing += frame
        self.results.default_nav[:] = frame.sum()
        self.results.default_nav_extra[:] = frame.sum()
        self.results.half_plane = 42                      # (attribute assignment writes INTO the buffer: udf/base.py:673-678)

    def merge(self, dest, src):
        dest.everything += src.everything
        dest.nothing += src.nothing
        dest.half_plane += src.half_plane
        dest.default_nav[:] = src.default_nav
        dest.default_nav_extra[:] = src.default_nav_extra


@pytest.mark.parametrize('with_roi', [True, False])
def test_adjust_valid_mask(lt_ctx, with_roi):
    """`get_results` sets masks of its own; what it does not mention keeps the default (test_valid_mask.py:144-188)"""
    ds = _ds(lt_ctx)
    roi = np.random.default_rng(1).choice([True, False], size=(16, 16)) if with_roi else None
    want_right_half = np.zeros((64, 64), dtype=bool)
    want_right_half[:, 32:] = True
    n = 0
    for res in lt_ctx.run_udf_iter(dataset=ds, udf=MasksOfItsOwn(), roi=roi):
        b = res.buffers[0]
        assert np.all(b['everything'].valid_mask) and b['everything'].valid_mask.shape == b['everything'].data.shape
        assert not np.any(b['nothing'].valid_mask)
        assert b['nothing'].valid_mask.shape == b['nothing'].data.shape == (32, 32)
        assert np.array_equal(b['default_nav'].valid_mask, res.damage.data)
        assert b['default_nav_extra'].valid_mask.shape == (16, 16, 2)
        assert np.array_equal(b['default_nav_extra'].valid_mask,
                              np.broadcast_to(res.damage.data.reshape((16, 16, 1)), (16, 16, 2)))
        assert np.array_equal(b['half_plane'].valid_mask, want_right_half)
        assert np.all(b['half_plane'].data == 42 * (n + 1))                      # one partition's 42 per merge
        n += 1
    assert n == 4


class MaskGivenAsParameter(UDF):
    def __init__(self, mask):
        super().__init__(mask=mask)

    def get_result_buffers(self):
        return {'custom': self.buffer(kind='single', dtype='float32', extra_shape=(64, 64, 3))}

    def get_results(self):
        return {'custom': self.with_mask(np.zeros((64, 64, 3), dtype="float32"), mask=self.params.mask)}

    def process_frame(self, frame):
        pass

    def merge(self, dest, src):
        pass


@pytest.mark.parametrize("mask_shape", [(32, 32), (1, 32), (64, 64), (64, 64, 4), (1, 1, 4), (1, 1, 1, 1)])
def test_right_half_invalid_shape(mask_shape, lt_ctx):
    """shapes that do not broadcast to (64, 64, 3) (test_valid_mask.py:211-232)"""
    ds = _ds(lt_ctx, (16, 16, 4, 4))
    with pytest.raises(InvalidMaskError):
        for res in lt_ctx.run_udf_iter(dataset=ds, udf=MaskGivenAsParameter(mask=np.zeros(mask_shape, dtype=bool))):
            res.buffers


@pytest.mark.parametrize("mask_dtype", [int, "float32", "complex64"])
def test_right_half_invalid_dtype(mask_dtype, lt_ctx):
    ds = _ds(lt_ctx, (16, 16, 4, 4))
    with pytest.raises(InvalidMaskError):
        for res in lt_ctx.run_udf_iter(dataset=ds, udf=MaskGivenAsParameter(mask=np.zeros((16, 16), dtype=mask_dtype))):
            res.buffers


@pytest.mark.parametrize("mask_shape", [(), (1,), (1, 1), (1, 1, 1), (64, 64, 1), (64, 64, 3)])
def test_right_half_valid(mask_shape, lt_ctx):
    ds = _ds(lt_ctx, (16, 16, 4, 4))
    for res in lt_ctx.run_udf_iter(dataset=ds, udf=MaskGivenAsParameter(mask=np.zeros(mask_shape, dtype=bool))):
        vm = res.buffers[0]['custom'].valid_mask
        assert vm.shape == (64, 64, 3) and not vm.any()


def test_valid_mask_slice_bounding(lt_ctx):
    ds = _ds(lt_ctx)
    for res in lt_ctx.run_udf_iter(dataset=ds, udf=MasksOfItsOwn()):
        b = res.buffers[0]
        buf = b['everything']
        assert buf.data[buf.valid_slice_bounding].shape == buf.data.shape
        buf = b['nothing']
        assert prod(buf.data[buf.valid_slice_bounding].shape) == 0
        buf = b['default_nav']
        assert prod(buf.data[buf.valid_slice_bounding].shape) >= np.count_nonzero(res.damage.data)
        # whole nav rows are valid after every partition of this 16 x 16 scan in 4 partitions
        inner = buf.get_valid_slice_inner(axis=0)
        assert np.all(buf.valid_mask[inner]) and buf.data[inner].size == np.count_nonzero(res.damage.data)
        buf = b['half_plane']
        assert buf.valid_slice_bounding == np.s_[0:64, 32:64]
        assert buf.get_valid_slice_inner(axis=1) == np.s_[:, 32:64]


@pytest.mark.parametrize('with_roi', [False, True])
def test_masked_data_and_raw_masked_data(lt_ctx, with_roi):
    ds = _ds(lt_ctx)
    roi = np.random.default_rng(2).choice(a=[True, False], size=(16, 16)) if with_roi else None
    for res in lt_ctx.run_udf_iter(dataset=ds, udf=MasksOfItsOwn(), roi=roi):
        for k, buf in res.buffers[0].items():
            want = np.sum(buf.data[buf.valid_mask])
            for md in (buf.masked_data, buf.raw_masked_data):
                assert want == md.sum() or np.all(md.mask)          # (all masked: the sum is the `masked` marker)
            if buf.kind == 'nav' and roi is not None:
                matched = np.broadcast_to(roi.reshape(roi.shape + (1,) * len(buf.extra_shape)), buf.valid_mask.shape)
                assert np.count_nonzero(~buf.raw_masked_data.mask) == np.count_nonzero(matched[buf.valid_mask])
                assert buf.raw_masked_data.shape == (np.count_nonzero(roi),) + buf.extra_shape
            assert np.sum(buf.raw_data[buf._valid_mask]) == buf.raw_masked_data.sum() or np.all(buf.raw_masked_data.mask)


def test_get_inner_slice():
    a = np.zeros((16, 16), dtype=bool)
    a[5:7] = 1
    a[8, 8] = 1
    a[-1, -1] = 1
    assert get_inner_slice(a, axis=0) == np.s_[5:7, :]
    b = np.zeros((16, 16, 16), dtype=bool)
    b[5:7] = 1
    b[8, 1] = 1
    b[-1, -1] = 1
    assert get_inner_slice(b, axis=0) == np.s_[5:7, :, :]
    c = np.zeros((16, 16, 16), dtype=bool)
    c[:, 5:7, :] = 1
    assert get_inner_slice(c, axis=1) == np.s_[:, 5:7, :]
    # the FIRST run, not the longest
    d = np.zeros((8, 2), dtype=bool)
    d[1] = d[3:7] = True
    assert get_inner_slice(d, axis=0) == np.s_[1:2, :]
    assert d[get_inner_slice(np.zeros((8, 2), dtype=bool))].size == 0


def test_get_bbox_and_slice():
    a = np.zeros((16, 16), dtype=bool)
    a[6, 6] = 1
    assert get_bbox(a) == (6, 6, 6, 6)
    assert get_bbox_slice(a) == np.s_[6:7, 6:7]
    a = np.zeros((16, 16, 16), dtype=bool)
    a[:, 6, 6] = 1
    assert get_bbox(a) == (0, 15, 6, 6, 6, 6)
    assert get_bbox_slice(a) == np.s_[0:16, 6:7, 6:7]
    a[:, -1, -1] = 1
    assert get_bbox_slice(a) == np.s_[0:16, 6:16, 6:16]
    f = np.zeros((4, 5), dtype=np.float32)
    f[1, 2] = 1e-12                                     # below eps: counts as zero
    f[3, 4] = -2.
    assert get_bbox(f) == (3, 3, 4, 4)
    assert np.zeros((4, 5))[get_bbox_slice(np.zeros((4, 5)))].size == 0


@pytest.mark.parametrize('kind,roi,valid,want_shape', [
    ('nav', None, [1, 1, 0], (3, 1, 2)),
    ('nav', [True, False, True], [1, 0], (2, 1, 2)),
    ('sig', None, [1, 1, 0], (32, 32, 1, 2)),
    ('single', None, [1, 1, 0], (1, 2)),
])
def test_default_mask_extra_shape(kind, roi, valid, want_shape):
    """test_valid_mask.py:386-430"""
    buf = BufferWrapper(kind=kind, extra_shape=(1, 2), dtype="float32")
    valid_nav_mask = np.array(valid, dtype=bool)
    ds_shape = Shape((3, 1, 32, 32), sig_dims=2)
    m = buf.make_default_mask(valid_nav_mask=valid_nav_mask, dataset_shape=ds_shape,
                              roi=None if roi is None else np.array(roi))
    assert m.shape == want_shape and m.dtype == bool
    if kind == 'nav':
        assert np.array_equal(m, np.broadcast_to(valid_nav_mask.reshape((-1, 1, 1)), want_shape))
    else:
        assert m.all()


class SingleBufferFollowsNav(UDF):
    """a kind='single' buffer whose mask follows the nav mask (test_valid_mask.py:433-462)"""

    def get_result_buffers(self):
        nav_shape = tuple(self.meta.dataset_shape.nav)
        if self.meta.roi is not None:
            nav_shape = (int(np.count_nonzero(self.meta.roi)),)
        return {'half_plane': self.buffer(kind='single', dtype=np.float32, extra_shape=nav_shape)}

    def process_frame(self, frame):
        self.results.half_plane[tuple(self.meta.coordinates[0])] = np.sum(frame)

    def get_results(self):
        vm = self.meta.get_valid_nav_mask()
        return {'half_plane': self.with_mask(self.results.half_plane, mask=vm.reshape(self.results.half_plane.shape))}

    def merge(self, dest, src):
        dest.half_plane += src.half_plane


@pytest.mark.parametrize('with_roi', [False])
def test_adjust_valid_mask_extra(lt_ctx, with_roi):
    data = np.random.default_rng(3).random((16, 16, 8, 8)).astype(np.float32)
    ds = MemoryDataSet(data=data, num_partitions=4).initialize(lt_ctx.executor)
    res = lt_ctx.run_udf(dataset=ds, udf=SingleBufferFollowsNav())
    buf = res['half_plane']
    assert buf.valid_mask.shape == (16, 16) and buf.valid_mask.all()
    assert np.allclose(buf.data, data.sum(axis=(2, 3)), rtol=1e-5)


def test_array_with_mask_rules():
    arr = np.zeros((4, 3), dtype=np.float32)
    assert ArrayWithMask(arr, True).mask.shape == (4, 3) and ArrayWithMask(arr, True).mask.all()
    assert not ArrayWithMask(arr, False).mask.any()
    assert ArrayWithMask(arr, np.array([True, False, True])).mask[2].tolist() == [True, False, True]
    with pytest.raises(InvalidMaskError, match='compatible shapes'):
        ArrayWithMask(arr, np.zeros((4,), dtype=bool))
    with pytest.raises(InvalidMaskError, match='dtype=bool'):
        ArrayWithMask(arr, np.zeros((4, 3), dtype=np.uint8))
    assert UDF.with_mask(arr, True).arr is arr
