"""
CPU-side tests of the host runtime: Shape/Slice algebra, buffers, partitioning + tiling
negotiation (against the reference's golden tile shapes), the UDF runner with NumPy UDFs against the
oracle and the golden vectors (config C1 plumbing), error behaviour, and the C-ABI surface.
"""
import ctypes
import os
import re

import numpy as np
import pytest

import recipes
from oracle import path as opath
from libertem_amd.api import Context
from libertem_amd.common import Shape, Slice
from libertem_amd.common.buffers import BufferWrapper
from libertem_amd.common.exceptions import HipRequiredError, UDFException, ExecutorSpecException
from libertem_amd.io.dataset import MemoryDataSet
from libertem_amd.io.dataset.base import Negotiator, Partition
from libertem_amd.udf.base import UDF, UDFRunner, _get_dtype
from libertem_amd.udf.masks import ApplyMasksUDF
from libertem_amd.udf.sum import SumUDF
from libertem_amd.udf.sumsigudf import SumSigUDF
from libertem_amd.udf.com import CoMUDF
from libertem_amd.executor.inline import InlineJobExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# --- NumPy UDFs used to exercise the plumbing (what a user would write) -----------------------
class NumpySumUDF(UDF):
    def __init__(self, dtype='float32'):
        super().__init__(dtype=dtype)

    def get_preferred_input_dtype(self):
        return self.params.dtype

    def get_result_buffers(self):
        return {'intensity': self.buffer(kind='sig', dtype=self.meta.input_dtype)}

    def process_tile(self, tile):
        self.results.intensity[:] += np.sum(tile, axis=0)

    def merge(self, dest, src):
        dest.intensity[:] += src.intensity


class NumpySumSigUDF(UDF):
    def get_result_buffers(self):
        return {'intensity': self.buffer(
            kind='nav', dtype=np.result_type(self.meta.input_dtype, np.float32))}

    def process_tile(self, tile):
        self.results.intensity[:] += np.sum(tile.reshape((tile.shape[0], -1)), axis=1)


class NumpyMasksUDF(UDF):
    def __init__(self, masks):
        super().__init__(masks=masks)

    def get_result_buffers(self):
        dt = np.result_type(self.meta.input_dtype, self.params.masks.dtype)
        return {'intensity': self.buffer(kind='nav', extra_shape=(len(self.params.masks),),
                                         dtype=dt)}

    def process_tile(self, tile):
        m = self.meta.sig_slice.get(self.params.masks, sig_only=True)
        m = m.reshape((len(self.params.masks), -1)).T
        self.results.intensity[:] += tile.reshape((tile.shape[0], -1)) @ m


class FrameUDF(UDF):
    def get_result_buffers(self):
        return {'mx': self.buffer(kind='nav', dtype=np.float32)}

    def process_frame(self, frame):
        self.results.mx[:] = frame.max()


@pytest.fixture(params=['inline', pytest.param('hip', marks=pytest.mark.gpu)])
def ctx(request):
    """the inline executor here; under `-m gpu` the same tests on the HIP executor (NumPy UDFs of users run on the
    host there too, through its own merge and delivery code)"""
    if request.param == 'hip':
        from libertem_amd.executor.hip import HipJobExecutor
        c = Context(executor=HipJobExecutor())
        yield c
        c.close()
    else:
        yield Context(executor=InlineJobExecutor(debug=True, inline_threads=2))


# --- Shape / Slice -------------------------------------------------------------------------------
def test_shape():
    s = Shape((4, 5, 6, 7), sig_dims=2)
    assert tuple(s.nav) == (4, 5) and tuple(s.sig) == (6, 7)
    assert s.size == 840 and s.nav.size == 20
    assert tuple(s.flatten_nav()) == (20, 6, 7)
    assert tuple(s.flatten_sig()) == (4, 5, 42)
    assert s == Shape((4, 5, 6, 7), sig_dims=2) and s != Shape((4, 5, 6, 7), sig_dims=1)
    assert s.nav.dims == 2 and s.sig.dims == 2 and s.dims == 4
    assert s + (1,) == (4, 5, 6, 7, 1)


def test_slice_algebra():
    a = Slice(origin=(0, 0, 0), shape=Shape((4, 8, 8), sig_dims=2))
    b = Slice(origin=(2, 4, 4), shape=Shape((4, 8, 8), sig_dims=2))
    i = a.intersection_with(b)
    assert i.origin == (2, 4, 4) and tuple(i.shape) == (2, 4, 4)
    c = Slice(origin=(10, 0, 0), shape=Shape((1, 8, 8), sig_dims=2))
    assert a.intersection_with(c).is_null()
    assert b.shift(a).origin == (2, 4, 4)
    assert a.shift_by((1, 2)).origin == (0, 1, 2)
    arr = np.arange(6 * 8 * 8).reshape((6, 8, 8))
    assert np.array_equal(b.intersection_with(a).get(arr), arr[2:4, 4:8, 4:8])
    assert np.array_equal(i.get(arr, sig_only=True), arr[:, 4:8, 4:8])
    subs = list(Slice(origin=(0, 0), shape=Shape((5, 7), sig_dims=2)).subslices((2, 4)))
    assert [(s.origin, tuple(s.shape)) for s in subs] == [
        ((0, 0), (2, 4)), ((0, 4), (2, 3)), ((2, 0), (2, 4)), ((2, 4), (2, 3)),
        ((4, 0), (1, 4)), ((4, 4), (1, 3))]
    roi = np.zeros(12, dtype=bool)
    roi[[1, 5, 6, 11]] = True
    p = Slice(origin=(4, 0, 0), shape=Shape((4, 2, 2), sig_dims=2))
    adj = p.adjust_for_roi(roi)
    assert adj.origin[0] == 1 and adj.shape[0] == 2
    n = Slice(origin=(1, 0, 0, 0), shape=Shape((2, 3, 2, 2), sig_dims=2))
    assert n.flatten_nav((4, 3, 2, 2)).origin == (3, 0, 0)


# --- partitioning + tiling negotiation vs the reference ------------------------------------------
@pytest.mark.parametrize('case', recipes.TILING_CASES, ids=lambda c: c['name'])
def test_negotiator_matches_reference(golden_dir, case):
    g = np.load(os.path.join(golden_dir, 'tiling.npz'))
    data = np.zeros(tuple(case['shape']), dtype=case['dtype'])
    ds = MemoryDataSet(data=data, num_partitions=case['num_partitions'], sig_dims=2,
                       tileshape=case.get('tileshape'))
    parts = list(ds.get_partitions())
    ref_parts = g[case['name'] + '__partitions']
    assert [(p.slice.origin[0], p.slice.shape[0]) for p in parts] == \
        [(int(a), int(b)) for a, b in ref_parts]
    udf = NumpySumUDF() if case['udf'] == 'sum' else NumpyMasksUDF(np.zeros((2, 4, 4), np.float32))
    dtype = _get_dtype([udf], ds.dtype)
    scheme = Negotiator().get_scheme(udfs=[udf], dataset=ds, read_dtype=dtype,
                                     approx_partition_shape=parts[0].shape)
    assert tuple(scheme.shape) == tuple(int(x) for x in g[case['name'] + '__tileshape'])
    assert len(scheme) == int(g[case['name'] + '__n_sig_slices'])
    mine = [list(s.origin) + list(s.shape) for _, s in scheme.slices]
    assert mine == [list(map(int, r)) for r in g[case['name'] + '__sig_slices']]


def test_partition_clamp_warning():
    with pytest.warns(RuntimeWarning):
        parts = list(Partition.make_slices(Shape((1, 3, 4, 4), sig_dims=2), 8))
    assert len(parts) == 3


# --- plumbing: config C1 and friends through the runtime, vs golden + oracle -----------------------
@pytest.mark.parametrize('case', recipes.SUM_CASES, ids=lambda c: c['name'])
def test_sum_plumbing(ctx, golden_dir, case):
    g = np.load(os.path.join(golden_dir, 'sums.npz'))
    data = recipes.make_sum_case(case)
    ds = ctx.load('memory', data=data, num_partitions=case['num_partitions'], sig_dims=2,
                  tileshape=case.get('tileshape'))
    kw = case.get('sum_kwargs', {})
    s = ctx.run_udf(dataset=ds, udf=NumpySumUDF(**kw))['intensity']
    ss = ctx.run_udf(dataset=ds, udf=NumpySumSigUDF())['intensity']
    assert s.data.dtype == g[case['name'] + '__sum'].dtype
    assert ss.data.dtype == g[case['name'] + '__sumsig'].dtype
    atol = 1e-6 * np.abs(data).sum(axis=(0, 1)).max() if data.dtype.kind == 'c' else 0
    assert np.allclose(s.data, g[case['name'] + '__sum'], rtol=1e-6, atol=atol)
    assert np.allclose(ss.data, g[case['name'] + '__sumsig'], rtol=1e-6, atol=atol)
    # identical tiles, identical order -> identical bits as the oracle's loop
    assert np.array_equal(s.data, opath.sum_udf(data, num_partitions=case['num_partitions'],
                                                tileshape=case.get('tileshape'), **kw))


def test_c1_config(ctx):
    """BASELINE.json configs[0]: SumUDF-like on 32x32 x 128x128 float32, inline CPU executor."""
    data = np.random.default_rng(0).random((32, 32, 128, 128), dtype=np.float32)
    ds = ctx.load('memory', data=data, num_partitions=4, sig_dims=2)
    from libertem_amd.api import Context
    from libertem_amd.executor.inline import InlineJobExecutor
    from libertem_amd.udf.sum import SumUDF
    from libertem_amd.udf.sumsigudf import SumSigUDF
    # the config as written: Context(InlineJobExecutor()).run_udf(ds, SumUDF()) with the SHIPPED class
    c1 = Context(executor=InlineJobExecutor())
    res = c1.run_udf(dataset=ds, udf=SumUDF())
    assert res['intensity'].data.dtype == np.float32
    assert np.array_equal(res['intensity'].data, opath.sum_udf(data, num_partitions=4))
    assert np.allclose(res['intensity'].data, data.sum(axis=(0, 1)), rtol=1e-5)
    ss = c1.run_udf(dataset=ds, udf=SumSigUDF())
    assert np.allclose(ss['intensity'].data, data.sum(axis=(2, 3)), rtol=1e-5)
    # the test-local NumPy UDF of earlier rounds takes the same tiles in the same order: identical bits
    res2 = ctx.run_udf(dataset=ds, udf=NumpySumUDF())
    assert np.array_equal(res2['intensity'].data, res['intensity'].data)


@pytest.mark.parametrize('name', ['c2_u16_16masks', 'odd_tiles_f32', 'f32_5masks', 'i32_f64',
                                  'u16_c64masks', 'single_frame'])
def test_masks_plumbing(ctx, golden_dir, name):
    g = np.load(os.path.join(golden_dir, 'apply_masks_dense.npz'))
    case = next(c for c in recipes.DENSE_CASES if c['name'] == name)
    data, masks = recipes.make_dense_case(case)
    ds = ctx.load('memory', data=data, num_partitions=case['num_partitions'], sig_dims=2,
                  tileshape=case.get('tileshape'))
    res = ctx.run_udf(dataset=ds, udf=NumpyMasksUDF(masks))['intensity'].data
    ref = g[name]
    assert res.shape == ref.shape and res.dtype == ref.dtype
    assert np.allclose(res, ref, rtol=2e-6, atol=2e-6 * np.abs(ref).max())


def test_roi_iter_multi_udf_frame(ctx):
    data = np.random.default_rng(1).integers(0, 100, (3, 8, 16, 16)).astype(np.uint16)
    masks = np.random.default_rng(2).random((4, 16, 16)).astype(np.float32)
    ds = ctx.load('memory', data=data, num_partitions=3, sig_dims=2)
    roi = np.zeros((3, 8), dtype=bool)
    roi[0, 2:5] = True
    roi[2, 7] = True
    res, res2 = ctx.run_udf(dataset=ds, udf=[NumpyMasksUDF(masks), FrameUDF()], roi=roi)
    ref = opath.apply_masks(data, masks, use_torch=False)
    d = res['intensity'].data
    assert d.shape == (3, 8, 4) and np.isnan(d[1, 0, 0])
    assert np.allclose(d[roi], ref[roi], rtol=1e-6)
    assert res['intensity'].raw_data.shape == (4, 4)
    assert np.allclose(res2['mx'].data[roi], data.max(axis=(2, 3))[roi])
    # coordinate-tuple roi (reference api.py:1280-1288)
    r3 = ctx.run_udf(dataset=ds, udf=NumpySumSigUDF(), roi=(1, 3))
    assert np.count_nonzero(~np.isnan(r3['intensity'].data)) == 1
    # partial results + damage
    seen = [int(p.damage.data.sum()) for p in ctx.run_udf_iter(dataset=ds, udf=NumpySumSigUDF())]
    assert seen == [8, 16, 24]
    # empty roi
    r4 = ctx.run_udf(dataset=ds, udf=NumpySumSigUDF(), roi=np.zeros((3, 8), dtype=bool))
    assert np.all(np.isnan(r4['intensity'].data))
    with pytest.raises(ValueError):
        ctx.run_udf(dataset=ds, udf=NumpySumSigUDF(), roi=np.zeros((2, 2), dtype=bool))


def test_buffer_wrapper_views():
    data = np.zeros((2, 6, 4, 4), dtype=np.float32)
    ds = MemoryDataSet(data=data, num_partitions=2, sig_dims=2)
    parts = list(ds.get_partitions())
    b = BufferWrapper(kind='nav', extra_shape=(3,), dtype=np.float32)
    b.set_shape_ds(ds.shape)
    b.allocate()
    assert b.raw_data.shape == (12, 3) and b.data.shape == (2, 6, 3)
    v = b.get_view_for_partition(parts[1])
    v[:] = 1
    assert b.raw_data[:6].sum() == 0 and b.raw_data[6:].sum() == 18
    s = BufferWrapper(kind='sig', dtype=np.float64)
    s.set_shape_partition(parts[0])
    s.allocate()
    assert s.raw_data.shape == (4, 4)
    one = BufferWrapper(kind='single', extra_shape=(3, 2), dtype=np.float64)
    one.set_shape_ds(ds.shape)
    one.allocate()
    assert one.raw_data.shape == (3, 2)
    with pytest.raises(ValueError):
        BufferWrapper(kind='bogus')


def test_udf_interface_errors(ctx):
    data = np.zeros((2, 2, 4, 4), dtype=np.float32)
    ds = ctx.load('memory', data=data, num_partitions=1, sig_dims=2)

    class NoProcess(UDF):
        def get_result_buffers(self):
            return {}

    with pytest.raises(TypeError):
        ctx.run_udf(dataset=ds, udf=NoProcess())

    class SigNoMerge(UDF):
        def get_result_buffers(self):
            return {'x': self.buffer(kind='sig', dtype=np.float32)}

        def process_tile(self, tile):
            pass

    with pytest.raises(NotImplementedError):
        ctx.run_udf(dataset=ds, udf=SigNoMerge())

    class BadResult(UDF):
        def get_result_buffers(self):
            return {'x': self.buffer(kind='nav', dtype=np.float32)}

        def process_tile(self, tile):
            pass

        def get_results(self):
            return {'x': self.results.x, 'undeclared': np.zeros(4, dtype=np.float32)}

    with pytest.raises(UDFException):
        ctx.run_udf(dataset=ds, udf=BadResult())['x'].data
    with pytest.raises(ExecutorSpecException):
        Context.make_with('dask')


# --- the native operators must refuse to run without the HIP backend -------------------------------
@pytest.mark.parametrize('make_udf', [
    lambda: ApplyMasksUDF(mask_factories=[lambda: np.ones((4, 4))]),
    lambda: CoMUDF.with_params(),
], ids=['masks', 'com'])
def test_native_udfs_fail_loudly_on_cpu(ctx, make_udf):
    if ctx.executor.device_class == 'hip':
        pytest.skip('asserts what a CPU executor does')
    # (SumUDF / SumSigUDF list BACKEND_NUMPY too since round 5: BASELINE config C1 runs them on the inline
    #  executor, test_c1_config; a GPU worker never takes their NumPy branch -- test_gpu_worker_never_takes_numpy_branch)
    data = np.zeros((2, 2, 4, 4), dtype=np.float32)
    ds = ctx.load('memory', data=data, num_partitions=1, sig_dims=2)
    with pytest.raises(HipRequiredError):
        ctx.run_udf(dataset=ds, udf=make_udf())


def test_gpu_worker_never_takes_numpy_branch():
    """_execution_plan: a UDF that lists BACKEND_HIP runs on the device or not at all on a 'hip' worker"""
    from libertem_amd.udf.base import _execution_plan, HIP, NUMPY
    assert _execution_plan([SumUDF()], (NUMPY, HIP), 'hip') == HIP
    assert _execution_plan([SumUDF()], (NUMPY, HIP), 'cpu') == NUMPY
    with pytest.raises(ValueError):
        _execution_plan([SumUDF()], (NUMPY,), 'hip')
    # round-5 advice: `[SumUDF(), MyNumpyOnlyUDF()]` on a GPU context raised; the reference plans per UDF and runs
    # such mixes.  Every UDF of the run offers NumPy -> the run happens on the host, announced by a warning; a mix
    # with a native operator that has no NumPy path (ApplyMasksUDF) is still refused
    with pytest.warns(RuntimeWarning, match='NumpySumUDF'):
        assert _execution_plan([SumUDF(), NumpySumUDF()], (NUMPY, HIP), 'hip') == NUMPY
    with pytest.raises(ValueError):
        _execution_plan([ApplyMasksUDF(mask_factories=[lambda: np.ones((4, 4))]), NumpySumUDF()], (NUMPY, HIP), 'hip')


def test_apply_masks_udf_argument_errors():
    with pytest.raises(ValueError):
        ApplyMasksUDF(mask_factories=[lambda: np.ones((4, 4))], backends=('numpy',))
    with pytest.raises(ValueError):      # reference udf/masks.py:268-277
        ApplyMasksUDF(mask_factories=[lambda: np.ones((4, 4))], shifts=(1, 2),
                      use_sparse='scipy.sparse')
    ApplyMasksUDF(mask_factories=[lambda: np.ones((4, 4))], shifts=(1, 2))
    with pytest.raises(ValueError):
        CoMUDF.with_params(r=3., ri=5.)


def test_hip_executor_requires_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("box has a GPU")
    from libertem_amd.executor.hip import HipJobExecutor
    with pytest.raises(RuntimeError):
        HipJobExecutor()


# --- the C ABI: the library loads and exports every symbol include/ltmi.h declares -----------------
def test_c_abi_exports():
    from libertem_amd import hip
    hdr = open(os.path.join(ROOT, 'include', 'ltmi.h')).read()
    declared = set(re.findall(r'\b(ltmi_[a-z_0-9]+)\s*\(', hdr))
    declared.discard('ltmi_masks')
    assert len(declared) >= 15
    L = hip.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/ltmi.h but not exported"
    assert set(hip.EXPORTS) == declared
    # ... and nothing else: the dynamic symbol table of the library IS the header
    # (-fvisibility=hidden + version script, libertem_amd/build.py)
    import subprocess
    nm = subprocess.run(['nm', '-D', '--defined-only', L._name], stdout=subprocess.PIPE, text=True,
                        check=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if ln.strip()}
    assert exported == declared, sorted(exported ^ declared)
    assert L.ltmi_version() == 1
    n = ctypes.c_int(-1)
    assert L.ltmi_device_count(ctypes.byref(n)) == 0 and n.value >= 0
    # argument errors are reported, not crashed on
    assert L.ltmi_device_count(None) == -1
    assert b'null' in L.ltmi_last_error()


def test_product_does_not_import_oracle():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import libertem_amd.api, libertem_amd.udf.masks, "
            "libertem_amd.udf.com, libertem_amd.analysis, libertem_amd.executor.hip; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), "
            "'product imports the oracle'") % ROOT
    subprocess.run([sys.executable, '-c', code], check=True)
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'libertem_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), f


# --- detector corrections (host path) -------------------------------------------------------------
def _excluded(case, excluded):
    from libertem_amd.io.corrections.corrset import ExcludedPixels
    return None if excluded is None else ExcludedPixels(excluded, tuple(case['sig']))


@pytest.mark.parametrize('case', recipes.CORR_CASES, ids=lambda c: c['name'])
def test_corrections_host_path_vs_reference_golden(ctx, golden_dir, case):
    """CorrectionSet through Context.run_udf with NumPy UDFs on the inline executor, against the
    reference's results (tests/corrections/test_corrset.py drives SumUDF the same way)."""
    from libertem_amd.io.corrections import CorrectionSet
    from libertem_amd.io.corrections import detector
    g = np.load(os.path.join(golden_dir, 'corrections.npz'))
    data, dark, gain, excluded, masks = recipes.make_corr_case(case)
    sig = tuple(case['sig'])
    corr = CorrectionSet(dark=dark, gain=gain, excluded_pixels=_excluded(case, excluded))
    ds = ctx.load('memory', data=data.copy(), num_partitions=case['num_partitions'],
                  sig_dims=len(sig))
    keep = data.copy()
    s = ctx.run_udf(dataset=ds, udf=NumpySumUDF(), corrections=corr)['intensity'].data
    assert np.array_equal(ds.flat_host().reshape(data.shape), keep), "dataset must stay untouched"
    ss = ctx.run_udf(dataset=ds, udf=NumpySumSigUDF(), corrections=corr)['intensity'].data
    mm = ctx.run_udf(dataset=ds, udf=NumpyMasksUDF(masks), corrections=corr)['intensity'].data
    for got, name in ((s, 'sum'), (ss, 'sumsig'), (mm, 'masks')):
        ref = g[f"{case['name']}__{name}"]
        assert got.shape == ref.shape and got.dtype == ref.dtype, name
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())
    out = detector.correct(buffer=data, dark_image=dark, gain_map=gain, excluded_pixels=excluded,
                           sig_shape=sig, inplace=False)
    np.testing.assert_allclose(out, g[f"{case['name']}__corrected"], rtol=2e-6, atol=1e-6)
    if gain is not None:
        dm = detector.correct_dot_masks(masks.astype(np.float64), gain, excluded)
        np.testing.assert_allclose(dm, g[f"{case['name']}__dot_masks"], rtol=1e-12, atol=1e-12)


def test_repair_descriptor_and_tileshape_adjustment_vs_reference(golden_dir):
    from libertem_amd.io.corrections import CorrectionSet
    from libertem_amd.io.corrections.corrset import ExcludedPixels
    from libertem_amd.io.corrections.detector import RepairDescriptor
    g = np.load(os.path.join(golden_dir, 'corrections.npz'))
    for i, (sig, coords) in enumerate(recipes.REPAIR_CASES):
        ex = np.array(coords, dtype=np.int64).T.reshape((len(sig), -1))
        d = RepairDescriptor(sig_shape=sig, excluded_pixels=ex, allow_empty=True)
        assert np.array_equal(d.exclude_flat, g[f"repair{i}__exclude_flat"])
        assert np.array_equal(d.repair_counts, g[f"repair{i}__repair_counts"])
        assert np.array_equal(d.repair_flat, g[f"repair{i}__repair_flat"])
    for i, (tile_shape, sig_shape, base_shape, coords) in enumerate(recipes.ADJUST_CASES):
        corr = CorrectionSet(excluded_pixels=ExcludedPixels(np.array(coords), sig_shape),
                             allow_empty=True)
        got = corr.adjust_tileshape(tile_shape=tile_shape, sig_shape=sig_shape,
                                    base_shape=base_shape)
        assert tuple(got) == tuple(g[f"adjust{i}"]), (i, got)


def test_correction_set_semantics(ctx):
    """Behaviour cases of the reference's tests/corrections/test_corrset.py:27-137 and
    test_detector.py:558-580."""
    from libertem_amd.io.corrections import CorrectionSet
    from libertem_amd.io.corrections.detector import RepairValueError, correct, CorrectError
    rng = np.random.default_rng(5)
    data = rng.random((4, 4, 16, 16)).astype(np.float32)
    ds = ctx.load('memory', data=data, sig_dims=2, num_partitions=2)
    # zero gain -> zero result, with and without dark
    for dark in (np.ones((16, 16)), None):
        r = ctx.run_udf(dataset=ds, udf=NumpySumUDF(),
                        corrections=CorrectionSet(dark=dark, gain=np.zeros((16, 16))))
        assert np.allclose(r['intensity'].data, 0)
    # dark of ones, with and without unit gain
    for gain in (np.ones((16, 16)), None):
        r = ctx.run_udf(dataset=ds, udf=NumpySumUDF(),
                        corrections=CorrectionSet(dark=np.ones((16, 16)), gain=gain))
        assert np.allclose(r['intensity'].data, np.sum(data - 1, axis=(0, 1)), rtol=1e-5)
    # empty excluded-pixel lists are no-ops
    for excl in (np.zeros((2, 0), dtype=np.int64), np.zeros((16, 16))):
        r = ctx.run_udf(dataset=ds, udf=NumpySumUDF(),
                        corrections=CorrectionSet(excluded_pixels=excl, gain=np.ones((16, 16))))
        assert np.allclose(r['intensity'].data, np.sum(data, axis=(0, 1)), rtol=1e-5)
    # no good neighbour: raises at construction unless allow_empty
    from libertem_amd.io.corrections.corrset import ExcludedPixels
    line = ExcludedPixels(np.array([[1, 2, 3]]), (19,))     # pixel 2 has only bad neighbours
    CorrectionSet(excluded_pixels=ExcludedPixels(np.array([[1, 2]]), (19,)))    # fine
    with pytest.raises(RepairValueError):
        CorrectionSet(excluded_pixels=line, gain=np.ones(19), dark=np.ones(19))
    corr = CorrectionSet(excluded_pixels=line, gain=np.ones(19), dark=np.ones(19),
                         allow_empty=True)
    ds1 = ctx.load('memory', data=np.ones((5, 6, 19)), sig_dims=1)
    r = ctx.run_udf(dataset=ds1, udf=NumpySumUDF(), corrections=corr)
    assert np.allclose(r['intensity'].data, 0)              # unpatched, (1 - 1) * 1
    # in-place needs float data and C order
    with pytest.raises(TypeError):
        correct(np.ones((2, 4, 4), dtype=np.uint8), gain_map=np.ones((4, 4)), inplace=True)
    with pytest.raises(CorrectError):
        correct(np.asfortranarray(np.ones((3, 4, 4), dtype=np.float32)),
                gain_map=np.ones((4, 4)), inplace=True)
    # odd 3D signal shape, pixel patched from its 26 neighbours (test_corrset.py:70-91)
    d3 = np.ones((2, 3, 5, 7, 9))
    ex3 = ExcludedPixels(np.array([[2, 4], [2, 5], [2, 5]]), (5, 7, 9))
    ds3 = ctx.load('memory', data=d3, sig_dims=3)
    r = ctx.run_udf(dataset=ds3, udf=NumpySumUDF(dtype='float64'),
                    corrections=CorrectionSet(excluded_pixels=ex3, gain=np.ones((5, 7, 9)),
                                              dark=np.ones((5, 7, 9))))
    assert np.allclose(r['intensity'].data, 0)


# --- raw files -----------------------------------------------------------------------------------------
def test_raw_file_dataset(ctx, tmp_path):
    """ctx.load('raw', ...): memory-mapped flat file, sync_offset semantics of the reference
    (io/dataset/raw.py; tests/io/datasets/test_raw.py: positive offsets skip frames, negative ones
    insert blank frames, frames missing at the end are blank)."""
    rng = np.random.default_rng(0)
    data = rng.integers(0, 1000, (4, 6, 8, 8)).astype(np.uint16)
    path = str(tmp_path / "scan.raw")
    data.tofile(path)
    ds = ctx.load('raw', path=path, dtype='uint16', nav_shape=(4, 6), sig_shape=(8, 8))
    assert tuple(ds.shape) == (4, 6, 8, 8) and ds.dtype == np.uint16
    res = ctx.run_udf(dataset=ds, udf=NumpySumSigUDF())['intensity'].data
    assert np.array_equal(res, data.reshape((4, 6, -1)).sum(axis=-1).astype(np.float32))
    flat = data.reshape((24, 8, 8))
    ds2 = ctx.load('raw', path=path, dtype='uint16', nav_shape=(4, 6), sig_shape=(8, 8),
                   sync_offset=5)
    res2 = ctx.run_udf(dataset=ds2, udf=NumpySumSigUDF())['intensity'].data.reshape(-1)
    assert np.array_equal(res2[:19], flat[5:].reshape((19, -1)).sum(axis=1)) and np.all(res2[19:] == 0)
    ds3 = ctx.load('raw', path=path, dtype='uint16', nav_shape=(4, 6), sig_shape=(8, 8),
                   sync_offset=-3)
    res3 = ctx.run_udf(dataset=ds3, udf=NumpySumSigUDF())['intensity'].data.reshape(-1)
    assert np.all(res3[:3] == 0) and np.array_equal(res3[3:], flat[:21].reshape((21, -1)).sum(axis=1))
    be = str(tmp_path / "be.raw")
    data.astype('>u2').tofile(be)
    ds4 = ctx.load('raw', path=be, dtype='>u2', nav_shape=(24,), sig_shape=(8, 8))
    res4 = ctx.run_udf(dataset=ds4, udf=NumpySumSigUDF())['intensity'].data
    assert np.array_equal(res4, flat.reshape((24, -1)).sum(axis=1))
    from libertem_amd.io.dataset import DataSetException
    with pytest.raises(DataSetException):
        ctx.load('raw', path=path, dtype='uint16', nav_shape=(4, 6), sig_shape=(8, 8),
                 sync_offset=24)
    with pytest.raises(DataSetException):
        ctx.load('raw', path=str(tmp_path / "nope.raw"), dtype='uint16', nav_shape=(4, 6),
                 sig_shape=(8, 8))
    with pytest.raises(DataSetException):
        ctx.load('hdf5', path=path)


# --- crystallinity host helpers ---------------------------------------------------------------------
@pytest.mark.parametrize('dtype', ['>u2', '>i2', '>u4', '<u2', '>f4'])
def test_other_byte_order_datasets(ctx, tmp_path, dtype):
    """Raw files / arrays in the other byte order give the same results as native ones (reference:
    DtypeConversionDecoder, io/dataset/base/decode.py:123-158; tests/io/test_decode_swap.py)."""
    rng = np.random.default_rng(11)
    dt = np.dtype(dtype)
    native = dt.newbyteorder('=')
    if dt.kind == 'f':
        vals = rng.random((3, 5, 16, 16)).astype(native)
    else:
        info = np.iinfo(native)
        vals = rng.integers(max(info.min, -3000), min(info.max, 3000), (3, 5, 16, 16)).astype(native)
    path = str(tmp_path / "scan.raw")
    vals.astype(dt).tofile(path)
    masks = rng.random((2, 16, 16)).astype(np.float32)
    ds = ctx.load('raw', path=path, dtype=dtype, nav_shape=(3, 5), sig_shape=(16, 16),
                  num_partitions=2)
    assert ds.dtype == native and ds.dtype.isnative
    if dt.kind in 'iu' and not dt.isnative:
        assert isinstance(ds.data.base if ds.data.base is not None else ds.data, np.memmap) or \
            not ds.data.flags.owndata                      # still the file mapping: no host copy
    res = ctx.run_udf(dataset=ds, udf=NumpyMasksUDF(masks))
    # signed integers in the other byte order read into float32: the reference's decoders hand out
    # the UNSIGNED word (decode.py:15-66, tests/golden/decode_signed.npz), the default here too
    as_read = vals.view(np.dtype(f'u{dt.itemsize}')) if (dt.kind == 'i' and not dt.isnative) \
        else vals
    ds_n = ctx.load('memory', data=as_read, sig_dims=2, num_partitions=2)
    ref = ctx.run_udf(dataset=ds_n, udf=NumpyMasksUDF(masks))
    assert res['intensity'].data.dtype == ref['intensity'].data.dtype
    assert np.array_equal(res['intensity'].data, ref['intensity'].data)
    # ... and an in-memory array in the other byte order
    ds_m = ctx.load('memory', data=vals.astype(dt), sig_dims=2, num_partitions=2)
    res_m = ctx.run_udf(dataset=ds_m, udf=NumpyMasksUDF(masks))
    assert np.array_equal(res_m['intensity'].data, ref['intensity'].data)
    if dt.kind == 'i' and not dt.isnative:
        # the arithmetic reading on request
        ds_s = ctx.load('memory', data=vals.astype(dt), sig_dims=2, num_partitions=2)
        ds_s.signed_other_order = 'signed'
        ref_s = ctx.run_udf(dataset=ctx.load('memory', data=vals, sig_dims=2, num_partitions=2),
                            udf=NumpyMasksUDF(masks))
        assert np.array_equal(ctx.run_udf(dataset=ds_s, udf=NumpyMasksUDF(masks))['intensity'].data,
                              ref_s['intensity'].data)
        assert not np.array_equal(ref_s['intensity'].data, ref['intensity'].data)


def _pick_check(ctx, golden_dir, case):
    from libertem_amd.udf.raw import PickUDF
    from libertem_amd.analysis.raw import PickFrameAnalysis, PickFFTFrameAnalysis
    g = np.load(os.path.join(golden_dir, 'pick.npz'))
    data = recipes.make_pick_case(case)
    ds = ctx.load('memory', data=data, num_partitions=case['num_partitions'],
                  sig_dims=len(case['sig']))
    roi = np.zeros(case['nav'], dtype=bool)
    for c in case['roi_frames']:
        roi[c] = True
    res = ctx.run_udf(dataset=ds, udf=PickUDF(), roi=roi)['intensity'].data
    ref = g[case['name'] + '__picked']
    assert res.dtype == ref.dtype and np.array_equal(res, ref)
    for cls, tag in ((PickFrameAnalysis, 'frame'), (PickFFTFrameAnalysis, 'fft')):
        params = dict(case['pick'])
        if tag == 'fft' and case['real'] is not None:
            params.update(real_rad=case['real']['rad'], real_centerx=case['real']['cx'],
                          real_centery=case['real']['cy'])
        a = cls(dataset=ds, parameters=params)
        assert np.array_equal(a.get_roi(), g[f"{case['name']}__{tag}__roi"])
        rs = ctx.run(a)
        ref = g[f"{case['name']}__{tag}"]
        got = rs.intensity_complex.raw_data if ref.dtype.kind == 'c' else rs.intensity.raw_data
        assert got.dtype == ref.dtype
        assert np.allclose(got, ref, rtol=1e-6, atol=1e-6 * np.abs(ref).max())
        if ref.dtype.kind != 'c':
            assert np.array_equal(rs.intensity_lin.raw_data, got)


@pytest.mark.parametrize('case', recipes.PICK_CASES, ids=lambda c: c['name'])
def test_pick_udf_and_analyses_vs_reference(ctx, golden_dir, case):
    """PickUDF / PickFrameAnalysis / PickFFTFrameAnalysis on the NumPy backend == the reference
    (udf/raw.py:12-76, analysis/raw.py:83-165, analysis/rawfft.py:38-57)."""
    _pick_check(ctx, golden_dir, case)
    ds = ctx.load('memory', data=np.zeros((2, 3, 4, 4)), sig_dims=2)
    for bad in (dict(x=1), dict(x=1, y=1, z=0)):
        with pytest.raises(ValueError):
            ctx.run(ctx.create_pick_analysis(dataset=ds, **bad))
    assert ctx.run(ctx.create_pick_analysis(dataset=ds, x=2, y=1)).intensity.raw_data.shape == (4, 4)


def _feed(flat, step, delay=0.005, stop_at=None):
    import time
    for i in range(0, len(flat) if stop_at is None else stop_at, step):
        time.sleep(delay)
        yield flat[i:i + step]


@pytest.fixture(params=['inline', pytest.param('hip', marks=pytest.mark.gpu)])
def live_ctx(request):
    # (no debug round trip through pickle: a stream is bound to its feeding thread, like a
    # device-resident dataset is bound to its GPU)
    if request.param == 'hip':
        from libertem_amd.executor.hip import HipJobExecutor
        c = Context(executor=HipJobExecutor())
        yield c
        c.close()
    else:
        yield Context(executor=InlineJobExecutor(debug=False, inline_threads=2))


def test_stream_dataset_partial_results(live_ctx):
    """Row f4: frames of a running acquisition (an iterator) are processed while they arrive and
    `run_udf_iter` publishes the result after every partition (reference api.py:1053-1152)."""
    rng = np.random.default_rng(3)
    data = rng.integers(0, 100, (6, 8, 16, 16)).astype(np.uint16)
    flat = data.reshape((-1, 16, 16))
    masks = rng.random((2, 16, 16)).astype(np.float32)
    ds = live_ctx.load('stream', frames=_feed(flat, 5), nav_shape=(6, 8), sig_shape=(16, 16),
                  dtype=np.uint16, num_partitions=6)
    assert tuple(ds.shape) == (6, 8, 16, 16) and ds.dtype == np.uint16
    done = []
    for part in live_ctx.run_udf_iter(dataset=ds, udf=NumpyMasksUDF(masks)):
        res = part.buffers[0]['intensity'].data.reshape((48, 2))
        done.append(int(np.count_nonzero(res[:, 0])))
        assert ds.frames_arrived >= done[-1]              # only frames that arrived are in
    assert done == [8, 16, 24, 32, 40, 48]
    ref = flat.reshape((48, -1)).astype(np.float32) @ masks.reshape((2, -1)).T
    assert np.allclose(res, ref, rtol=1e-5)
    # single frames as items, ROI, a second UDF on the finished stream object
    ds = live_ctx.load('stream', frames=iter(flat), nav_shape=(6, 8), sig_shape=(16, 16), dtype=np.uint16)
    roi = rng.random((6, 8)) < 0.5
    r = live_ctx.run_udf(dataset=ds, udf=NumpySumUDF(), roi=roi)
    assert np.array_equal(r['intensity'].data, flat[roi.reshape(-1)].astype(np.float32).sum(axis=0))
    r = live_ctx.run_udf(dataset=ds, udf=NumpySumUDF())
    assert np.array_equal(r['intensity'].data, flat.astype(np.float32).sum(axis=0))


def test_run_udf_async(ctx):
    """`sync=False` (reference api.py:981-1051, 1105-1152): run_udf returns a coroutine of the same result,
    run_udf_iter an async generator of the same partial results; the event loop keeps running meanwhile."""
    import asyncio
    rng = np.random.default_rng(9)
    data = rng.integers(0, 100, (4, 6, 8, 8)).astype(np.uint16)
    masks = rng.random((3, 8, 8)).astype(np.float32)
    ds = ctx.load('memory', data=data, num_partitions=4, sig_dims=2)
    want = ctx.run_udf(dataset=ds, udf=NumpyMasksUDF(masks))['intensity'].data

    async def main():
        ticks = []

        async def ticker():
            for _ in range(3):
                ticks.append(1)
                await asyncio.sleep(0)
        coro = ctx.run_udf(dataset=ds, udf=NumpyMasksUDF(masks), sync=False)
        assert asyncio.iscoroutine(coro)
        res, _ = await asyncio.gather(coro, ticker())
        both = await ctx.run_udf(dataset=ds, udf=[NumpyMasksUDF(masks), NumpySumUDF()], sync=False)
        parts = []
        async for part in ctx.run_udf_iter(dataset=ds, udf=NumpyMasksUDF(masks), sync=False):
            parts.append(np.array(part.buffers[0]['intensity'].data))
        return res, both, parts, ticks
    res, both, parts, ticks = asyncio.run(main())
    assert np.array_equal(res['intensity'].data, want) and len(ticks) == 3
    assert isinstance(both, tuple) and np.array_equal(both[0]['intensity'].data, want)
    assert np.array_equal(both[1]['intensity'].data, data.astype(np.float32).sum(axis=(0, 1)))
    assert len(parts) == 4 and np.array_equal(parts[-1], want)
    assert np.count_nonzero(parts[0][..., 0]) == 6


def test_run_inside_suspended_iteration(ctx):
    """A `run_udf_iter` owns its executor until it ends, also between two partial results.  A run started in that
    window BY THE LOOP BODY -- the reference's tests/test_context.py test_udf_iter does exactly that -- goes to a
    sibling executor and the iteration goes on untouched; a run from ANOTHER thread (it used to wait for ever) is
    refused with a clear error."""
    import asyncio
    import threading
    from libertem_amd.hip import RunInProgressError
    rng = np.random.default_rng(19)
    data = rng.integers(0, 100, (4, 6, 8, 8)).astype(np.uint16)
    masks = rng.random((3, 8, 8)).astype(np.float32)
    ds = ctx.load('memory', data=data, num_partitions=4, sig_dims=2)
    want = ctx.run_udf(dataset=ds, udf=NumpyMasksUDF(masks))['intensity'].data
    steps, errors = 0, []
    for part in ctx.run_udf_iter(dataset=ds, udf=NumpyMasksUDF(masks)):
        steps += 1
        # same thread: the result so far, recomputed with the damage as roi, and a whole second iteration
        ref = ctx.run_udf(dataset=ds, udf=NumpyMasksUDF(masks), roi=part.damage.data)['intensity']
        assert np.array_equal(ref.raw_data, part.buffers[0]['intensity'].raw_data[part.damage.raw_data])
        assert sum(1 for _ in ctx.run_udf_iter(dataset=ds, udf=NumpySumUDF())) == 4

        def other():
            try:
                ctx.run_udf(dataset=ds, udf=NumpySumUDF())
            except RunInProgressError as e:
                errors.append(e)
        th = threading.Thread(target=other)
        th.start()
        th.join(timeout=10)
        assert not th.is_alive()
    assert steps == 4 and len(errors) == 4
    assert np.array_equal(part.buffers[0]['intensity'].data, want)
    # the gate is free again
    assert np.array_equal(ctx.run_udf(dataset=ds, udf=NumpyMasksUDF(masks))['intensity'].data, want)
    # a half-consumed iterator that is closed gives the executor back
    it = ctx.run_udf_iter(dataset=ds, udf=NumpyMasksUDF(masks))
    next(it)
    assert np.array_equal(ctx.run_udf(dataset=ds, udf=NumpyMasksUDF(masks))['intensity'].data, want)   # (nested)
    it.close()
    assert np.array_equal(ctx.run_udf(dataset=ds, udf=NumpyMasksUDF(masks))['intensity'].data, want)

    async def main():
        seen = 0
        async for part in ctx.run_udf_iter(dataset=ds, udf=NumpyMasksUDF(masks), sync=False):
            seen += 1
            # (the context's worker thread holds the suspended iteration and runs this one: nested, on the sibling)
            ref = await ctx.run_udf(dataset=ds, udf=NumpyMasksUDF(masks), roi=part.damage.data, sync=False)
            assert np.array_equal(ref['intensity'].raw_data,
                                  part.buffers[0]['intensity'].raw_data[part.damage.raw_data])
        return seen, np.array(part.buffers[0]['intensity'].data)
    seen, last = asyncio.run(main())
    assert seen == 4 and np.array_equal(last, want)


def test_stream_dataset_in_place_feed(live_ctx):
    """`frames=None`: the producer writes into `scan_buffer` itself and commits its progress -- no feeder
    thread, no copy on this side (a detector's DMA target); partial results per partition as with the
    iterator feed; commit() out of order / on an iterator feed and a failed acquisition raise."""
    import threading
    import time
    from libertem_amd.io.dataset import DataSetException
    rng = np.random.default_rng(4)
    flat = rng.integers(0, 100, (48, 16, 16)).astype(np.uint16)
    masks = rng.random((2, 16, 16)).astype(np.float32)
    ds = live_ctx.load('stream', frames=None, nav_shape=(6, 8), sig_shape=(16, 16), dtype=np.uint16,
                       num_partitions=6)
    assert ds.scan_buffer.shape == (48, 16, 16) and ds.frames_arrived == 0

    def produce():
        for i in range(0, 48, 6):
            ds.scan_buffer[i:i + 6] = flat[i:i + 6]
            ds.commit(i + 6)
            time.sleep(0.005)
    th = threading.Thread(target=produce)
    th.start()
    done = []
    for part in live_ctx.run_udf_iter(dataset=ds, udf=NumpyMasksUDF(masks)):
        res = part.buffers[0]['intensity'].data.reshape((48, 2))
        done.append(int(np.count_nonzero(res[:, 0])))
        assert ds.frames_arrived >= done[-1]
    th.join()
    assert done == [8, 16, 24, 32, 40, 48]
    assert np.allclose(res, flat.reshape((48, -1)).astype(np.float32) @ masks.reshape((2, -1)).T, rtol=1e-5)
    with pytest.raises(DataSetException):
        ds.commit(40)                                      # going backwards
    it = live_ctx.load('stream', frames=iter(flat), nav_shape=(6, 8), sig_shape=(16, 16), dtype=np.uint16)
    with pytest.raises(DataSetException, match='in-place'):
        it.commit(3)
    bad = live_ctx.load('stream', frames=None, nav_shape=(6, 8), sig_shape=(16, 16), dtype=np.uint16,
                        num_partitions=6)
    bad.scan_buffer[:8] = flat[:8]
    bad.commit(8)
    bad.fail(RuntimeError("detector lost"))
    with pytest.raises(DataSetException, match='detector lost'):
        live_ctx.run_udf(dataset=bad, udf=NumpySumUDF())
    short = live_ctx.load('stream', frames=None, nav_shape=(6, 8), sig_shape=(16, 16), dtype=np.uint16,
                          num_partitions=6)
    short.commit(16)
    short.finish()
    with pytest.raises(DataSetException, match='ended after 16 of 48'):
        live_ctx.run_udf(dataset=short, udf=NumpySumUDF())


def test_stream_dataset_failures(live_ctx, ctx):
    if ctx.executor.device_class == 'hip':
        pytest.skip('asserts what a CPU executor does')
    from libertem_amd.io.dataset import DataSetException
    ds = ctx.load('stream', frames=[np.ones((4, 4))], nav_shape=(1,), sig_shape=(4, 4),
                  dtype=np.float32)
    with pytest.raises(TypeError, match='cannot be pickled'):       # debug executors pickle tasks
        ctx.run_udf(dataset=ds, udf=NumpySumUDF())
    ctx = live_ctx
    flat = np.ones((12, 4, 4), dtype=np.float32)
    ds = ctx.load('stream', frames=_feed(flat, 4, stop_at=8), nav_shape=(12,), sig_shape=(4, 4),
                  dtype=np.float32, num_partitions=3)
    with pytest.raises(DataSetException, match='ended after 8 of 12 frames'):
        ctx.run_udf(dataset=ds, udf=NumpySumUDF())
    ds = ctx.load('stream', frames=[np.ones((2, 5, 4))], nav_shape=(12,), sig_shape=(4, 4),
                  dtype=np.float32)
    with pytest.raises(DataSetException, match='does not hold frames'):
        ctx.run_udf(dataset=ds, udf=NumpySumUDF())

    def never():
        import time
        yield flat[:4]
        time.sleep(5)
    ds = ctx.load('stream', frames=never(), nav_shape=(12,), sig_shape=(4, 4), dtype=np.float32,
                  num_partitions=3, timeout=0.2)
    with pytest.raises(DataSetException, match='timed out'):
        ctx.run_udf(dataset=ds, udf=NumpySumUDF())
    with pytest.raises(DataSetException):
        ctx.load('stream', frames=[], nav_shape=(0,), sig_shape=(4, 4), dtype=np.float32)


@pytest.mark.parametrize('sig,rad_in,rad_out,center,rad', [
    ((32, 32), 4, 9, (16, 16), 5), ((24, 40), 3, 8, None, None), ((33, 31), 2.5, 11.5, (10.5, 20.25), 4),
])
def test_crystallinity_masks_and_bounding_box(sig, rad_in, rad_out, center, rad):
    """The masks CrystallinityUDF uploads (reference udf/crystallinity.py:47-71) and the bounding
    box the reduction kernel is restricted to."""
    from libertem_amd.udf.crystallinity import crystallinity_masks, mask_box
    real_mask, half = crystallinity_masks(sig, rad_in, rad_out, center, rad)
    sy, sx = sig
    yy, xx = np.ogrid[-sy * 0.5:sy - sy * 0.5, -sx * 0.5:sx - sx * 0.5]
    ring = 1 * (yy * yy + xx * xx <= rad_out ** 2) - 1 * (yy * yy + xx * xx <= rad_in ** 2)
    expect = np.fft.fftshift(ring)[:, :int(sx * 0.5) + 1]
    assert half.shape == (sy, sx // 2 + 1) and np.array_equal(half, expect)
    if center is None:
        assert real_mask is None
    else:
        y, x = np.ogrid[-center[0]:sy - center[0], -center[1]:sx - center[1]]
        assert np.array_equal(real_mask, 1 - 1 * (y * y + x * x <= rad * rad))
    lo, hi, nc = mask_box(half)
    boxed = np.zeros_like(half)
    boxed[:lo, :nc] = half[:lo, :nc]
    boxed[hi:, :nc] = half[hi:, :nc]
    assert np.array_equal(boxed, half)                 # nothing outside the box
    assert 0 <= lo <= hi <= sy and 0 < nc <= sx // 2 + 1
    assert mask_box(np.ones((6, 4))) == (6, 6, 4)      # no structure: every row, every column
    assert mask_box(np.zeros((6, 4))) == (0, 6, 0)


def test_fold_corrections_into_masks_identity():
    """masks' . x - const == masks . corrected(x) for random frames (the linear-algebra identity
    behind the folded corrections of ApplyMasksUDF / CoMUDF / SumSigUDF), incl. dead pixels on the
    border, adjacent dead pixels and one without any good neighbour (allow_empty)."""
    from oracle import corrections as oc
    from libertem_amd.io.corrections import CorrectionSet
    from libertem_amd.udf.masks import fold_corrections_into_masks
    rng = np.random.default_rng(8)
    sig = (9, 11)
    bad = np.zeros(sig, dtype=bool)
    for y, x in [(0, 0), (4, 5), (4, 6), (8, 10), (2, 0)]:
        bad[y, x] = True
    dark = rng.random(sig) * 5
    gain = rng.random(sig) + 0.5
    masks = rng.random((4,) + sig) - 0.3
    frames = rng.random((6,) + sig) * 100
    coords = [tuple(c) for c in np.argwhere(bad)]
    for kw in (dict(dark=dark, gain=gain, excluded_pixels=bad), dict(gain=gain), dict(dark=dark),
               dict(excluded_pixels=bad)):
        corr = CorrectionSet(**kw)
        folded, const = fold_corrections_into_masks(masks, corr, sig)
        corrected = oc.correct(frames, sig, dark=kw.get('dark'), gain=kw.get('gain'),
                               coords=coords if 'excluded_pixels' in kw else None,
                               out_dtype=np.float64)
        want = np.tensordot(corrected, masks, axes=([1, 2], [1, 2]))
        got = np.tensordot(frames, folded, axes=([1, 2], [1, 2]))
        if const is not None:
            got = got - const[None, :]
        else:
            assert 'dark' not in kw
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-9)
    # 1-D signal where pixel 2 has no good neighbour: it stays unpatched (as in `correct`)
    line = np.zeros(7, dtype=bool)
    line[1:4] = True
    corr = CorrectionSet(excluded_pixels=line, gain=np.full(7, 2.0), allow_empty=True)
    m = rng.random((2, 7))
    folded, const = fold_corrections_into_masks(m, corr, (7,))
    x = rng.random((3, 7))
    corrected = oc.correct(x, (7,), gain=np.full(7, 2.0), coords=[(1,), (2,), (3,)],
                           out_dtype=np.float64)
    np.testing.assert_allclose(x @ folded.T, corrected @ m.T, rtol=1e-12)
    assert const is None


def test_fold_corrections_into_sparse_masks_identity():
    """the same identity for sparse stacks: masks' = (masks . R) . diag(gain) stays sparse and
    equals the dense fold"""
    from libertem_amd.io.corrections import CorrectionSet
    from libertem_amd.udf.masks import fold_corrections_into_masks, \
        fold_corrections_into_sparse_masks
    from libertem_amd.common.sparse import SparseStack
    rng = np.random.default_rng(9)
    sig = (9, 11)
    bad = np.zeros(sig, dtype=bool)
    for y, x in [(0, 0), (4, 5), (4, 6), (8, 10), (2, 0)]:
        bad[y, x] = True
    dark = rng.random(sig) * 5
    gain = rng.random(sig) + 0.5
    dense = (rng.random((5,) + sig) - 0.3) * (rng.random((5,) + sig) < 0.2)
    dense[1, 4, 5] = 0.7                                  # weight on a dead pixel
    stack = SparseStack.from_dense(dense)
    for kw in (dict(dark=dark, gain=gain, excluded_pixels=bad), dict(gain=gain), dict(dark=dark),
               dict(excluded_pixels=bad)):
        corr = CorrectionSet(**kw)
        want, want_c = fold_corrections_into_masks(dense, corr, sig)
        got, got_c = fold_corrections_into_sparse_masks(stack, corr, sig)
        assert isinstance(got, SparseStack) and got.nnz <= stack.nnz + 8 * int(bad.sum()) * 5
        np.testing.assert_allclose(got.todense(), want, rtol=1e-12, atol=1e-12)
        if want_c is None:
            assert got_c is None
        else:
            np.testing.assert_allclose(got_c, want_c, rtol=1e-12, atol=1e-12)


_COUNT_UDF_MADE = []


class CountUDF(UDF):
    def __init__(self, scale=1.0, tags=None):
        super().__init__(scale=scale, tags=tags)
        _COUNT_UDF_MADE.append(self)

    def get_result_buffers(self):
        return {'s': self.buffer(kind='nav', dtype=np.float64)}

    def process_tile(self, tile):
        self.results.s[:] += self.params.scale * tile.reshape((tile.shape[0], -1)).sum(axis=1) \
            + (len(self.params.tags) if self.params.tags is not None else 0)


def test_plan_cache_reuses_planning_not_user_udf_instances(ctx):
    """`run_udf` in a loop with the same udf object re-uses the plan (tasks, tiling scheme) kept on
    the dataset; user UDFs still get a NEW instance per task (reference udf/base.py:1997-2003), and
    mutated parameters / a different udf object plan afresh."""
    made = _COUNT_UDF_MADE
    del made[:]
    data = np.random.default_rng(1).random((4, 5, 8, 8)).astype(np.float32)
    ds = ctx.load('memory', data=data, num_partitions=3, sig_dims=2)
    tags = [1, 2]
    udf = CountUDF(scale=2.0, tags=tags)
    expect = 2.0 * data.reshape((4, 5, -1)).sum(axis=-1, dtype=np.float64)
    a = ctx.run_udf(dataset=ds, udf=udf)['s'].data
    n_after_first = len(made)
    plans = ds.__dict__['_udf_plans']
    assert len(plans) == 1
    tasks_first = next(iter(plans.values()))['tasks']
    b = ctx.run_udf(dataset=ds, udf=udf)['s'].data
    assert len(plans) == 1 and next(iter(plans.values()))['tasks'] is tasks_first     # plan hit
    assert len(made) == n_after_first + 3                 # ... but 3 new task instances
    assert np.allclose(a, expect + 2) and np.array_equal(a, b)
    tags.append(3)                                        # list parameter mutated in place
    c = ctx.run_udf(dataset=ds, udf=udf)['s'].data
    # same parameter OBJECTS, different contents: the stale plan is replaced by a fresh one
    assert len(plans) == 1 and next(iter(plans.values()))['tasks'] is not tasks_first
    assert np.allclose(c, expect + 3)
    roi = np.zeros((4, 5), dtype=bool)
    roi[1] = True
    d = ctx.run_udf(dataset=ds, udf=udf, roi=roi)['s']
    assert len(plans) == 1                                # ROI runs are never cached
    assert np.allclose(d.raw_data, (expect + 3)[roi])


def test_result_where_option_and_float64_densify_rule(ctx):
    """`run_udf(result_where=...)`: only None / 'host' / 'device' are accepted and 'device' needs the
    HIP executor; float64 sparse stacks with more than one 16-column group are not densified (the
    float64 matrix kernel would read the frames once per group)."""
    if ctx.executor.device_class == 'hip':
        pytest.skip('asserts what a CPU executor does')
    import scipy.sparse as sp
    from libertem_amd.common.container import _worth_densifying

    class Count(UDF):
        def get_result_buffers(self):
            return {'n': self.buffer(kind='nav', dtype=np.float32)}

        def process_frame(self, frame):
            self.results.n[:] = frame.sum()

    data = np.ones((2, 3, 4, 4), dtype=np.float32)
    ds = ctx.load('memory', data=data, sig_dims=2, num_partitions=2)
    with pytest.raises(ValueError):
        ctx.run_udf(dataset=ds, udf=Count(), result_where='hbm')
    with pytest.raises(NotImplementedError):
        ctx.run_udf(dataset=ds, udf=Count(), result_where='device')
    res = ctx.run_udf(dataset=ds, udf=Count(), result_where='host')
    assert np.all(res['n'].data == 16) and res['n'].device_data is None

    full = sp.csr_matrix(np.ones((64, 40), dtype=np.float64))        # completely filled, 40 columns
    assert _worth_densifying(full.astype(np.float32), np.float32)
    assert not _worth_densifying(full, np.float64)
    assert _worth_densifying(sp.csr_matrix(np.ones((64, 12))), np.float64)     # one column group


# ---- Merlin .mib files: host side (headers, file series, parameters); decoding needs the GPU ------------
@pytest.mark.parametrize('case', recipes.MIB_CASES, ids=lambda c: c['name'])
def test_mib_headers_like_the_oracle(case):
    from oracle import mib as omib
    from libertem_amd.io.dataset.mib import parse_frame_header
    frames, files, hdr = recipes.make_mib_case(case)
    for name, blob in files.items():
        mine = parse_frame_header(blob[:1024], len(blob))
        ref = omib.parse_header(blob[:1024], len(blob))
        for key, val in ref.items():
            assert mine[key] == val, (name, key)
        assert np.dtype(mine['dtype']) == omib.declared_dtype(ref).newbyteorder('=')
        assert mine['num_images'] * (mine['header_size_bytes'] + mine['image_size_bytes']) == len(blob)


def test_mib_file_series_and_parameters(tmp_path):
    from libertem_amd.io.dataset import mib
    from libertem_amd.io.dataset.base import DataSetException
    case = [c for c in recipes.MIB_CASES if c['name'] == 'r6'][0]
    frames, files, hdr = recipes.make_mib_case(case)
    for fn, blob in files.items():
        (tmp_path / fn).write_bytes(blob)
    (tmp_path / 'r6.hdr').write_text(hdr)
    (tmp_path / 'other000001.mib').write_bytes(b'x')
    names = sorted(files)
    # the series from any of its files or from the .hdr file (numeric suffix stripped, mib.py:109-127)
    for p in (names[0], names[1], 'r6.hdr'):
        got = sorted(os.path.basename(f) for f in mib.get_filenames(str(tmp_path / p)))
        assert got == names
    assert mib.get_filenames(str(tmp_path / names[1]), disable_glob=True) == [str(tmp_path / names[1])]
    with pytest.raises(DataSetException, match='unknown extension'):
        mib.get_filenames(str(tmp_path / 'r6.raw'))
    assert mib.is_valid_hdr(str(tmp_path / 'r6.hdr'))
    assert mib.nav_shape_from_hdr(mib.read_hdr_file(str(tmp_path / 'r6.hdr'))) == (2, 3)
    assert mib.nav_shape_from_hdr({'ScanX': '7', 'ScanY': '5'}) == (5, 7)
    assert mib.get_image_count_and_sig_shape(str(tmp_path / 'r6.hdr')) == (6, (32, 64))
    with pytest.raises(ValueError, match='either nav_shape needs to be passed'):
        mib.MIBDataSet(path=str(tmp_path / names[0]))
    with pytest.raises(ValueError, match='cannot specify both'):
        with pytest.warns(FutureWarning):
            mib.MIBDataSet(path=str(tmp_path / names[0]), scan_size=(6,), nav_shape=(6,))
    with pytest.raises(DataSetException, match='not a .mib frame header'):
        mib.read_file_header(str(tmp_path / 'other000001.mib'))

    # no CPU decoder: an executor without a GPU is refused, loudly
    class NoGpu:
        gpu_id = None
    with pytest.raises(DataSetException, match='decodes the files on the GPU'):
        mib.MIBDataSet(path=str(tmp_path / 'r6.hdr')).initialize(NoGpu())


def test_sparse_radial_bins_keep_the_requested_dtype():
    """masks.py:290-353, sparse branch: the slices are `vals.astype(dtype)` and the one-entry centre patch is
    np.array([1 - slices[0][index] - radius_inner]) -- a float32 scalar read from the slice stays float32
    against Python numbers (NumPy >= 2), so the stack is float32 and ApplyMasksUDF's result is too; a
    float64 NumPy scalar for radius_inner promotes, as it would in the reference."""
    from libertem_amd import masks as pm
    from oracle import masks as omasks
    r32 = pm.radial_bins(16, 16, 32, 32, n_bins=10, use_sparse=True, dtype=np.float32)
    assert r32.dtype == np.float32
    o32 = omasks.radial_bins(16, 16, 32, 32, n_bins=10, use_sparse=True, dtype=np.float32)
    assert o32.dtype == np.float32
    assert np.array_equal(np.asarray(r32.todense()).reshape(10, -1), np.asarray(o32.todense()))
    dense = pm.radial_bins(16, 16, 32, 32, n_bins=10, use_sparse=False, dtype=np.float32)
    assert np.allclose(np.asarray(r32.todense()).reshape(dense.shape), dense, rtol=0, atol=1e-7)
    assert pm.radial_bins(16, 16, 32, 32, n_bins=10, use_sparse=True).dtype == np.float64
    assert pm.radial_bins(16, 16, 32, 32, n_bins=10, use_sparse=True, dtype=np.float32,
                          radius_inner=np.float64(0.25)).dtype == np.float64


def test_sparse_integer_exactness_rule():
    """common/container.py::_sparse_int_exact and hip.signed_representative (pure NumPy): an integer
    sparse stack stays sparse iff bits(tile dtype) + bits(largest column sum of |values|) <= 52, with
    unsigned values read as the signed numbers of their width."""
    import scipy.sparse as sp
    from libertem_amd.common.container import _sparse_int_exact
    from libertem_amd.hip import signed_representative
    assert np.array_equal(signed_representative(np.array([65530, 3, 0], np.uint16)), [-6, 3, 0])
    assert np.array_equal(signed_representative(np.array([-6, 3], np.int32)), [-6, 3])
    assert signed_representative(np.array([2**64 - 1], np.uint64))[0] == -1
    rng = np.random.default_rng(0)
    dense = np.where(rng.random((500, 40)) < 0.1, rng.integers(-9, 10, (500, 40)), 0)
    m = sp.csr_matrix(dense.astype(np.int64))
    assert _sparse_int_exact(m, (np.uint16,)) and _sparse_int_exact(m, (np.uint32, np.bool_))
    assert not _sparse_int_exact(m, (np.int64,)) and not _sparse_int_exact(m, (np.float32,))
    assert _sparse_int_exact(sp.csr_matrix(dense.astype(np.uint16)), (np.uint16,))      # 65527 = -9
    big = sp.csr_matrix(dense.astype(np.int64) * (1 << 30))
    assert _sparse_int_exact(big, (np.uint8,)) and not _sparse_int_exact(big, (np.uint16,))
    assert _sparse_int_exact(sp.csr_matrix((500, 40), dtype=np.int64), (np.uint32,))


def test_fingerprint_sees_column_bands_and_single_elements():
    """Round-3 review: an evenly strided sample of a (16, 256, 256) float32 stack (step = one row) hashes
    columns 0..15 of every row only -- a column-band edit or a small block went unseen and the cached
    device image / plan of the old masks was used.  Every byte of every array is hashed, whatever its size
    (round 5: no sampling above 64 MiB any more)."""
    from libertem_amd.common.fingerprint import array_fingerprint, fingerprint
    m = np.random.default_rng(0).random((16, 256, 256)).astype(np.float32)
    f0 = array_fingerprint(m)
    assert array_fingerprint(m.copy()) == f0
    for edit in (lambda a: a.__setitem__((slice(None), slice(None), slice(100, 110)), 0),
                 lambda a: a.__setitem__((3, slice(50, 60), slice(60, 70)), 7),
                 lambda a: a.__setitem__((7, 123, 45), -1.0),
                 lambda a: a.__setitem__((15, 255, 255), 2.0)):
        m2 = m.copy()
        edit(m2)
        assert array_fingerprint(m2) != f0
    # captured by a factory: the factory's fingerprint follows
    fac = (lambda: m)
    g0 = fingerprint(fac)
    m[:, :, 100:110] = 0
    assert fingerprint(fac) != g0
    # a large buffer: ONE changed element is seen
    big = np.zeros((17, 1024, 1024), np.float32)
    fb = array_fingerprint(big)
    big[9, 777, 123] = 1
    assert array_fingerprint(big) != fb
    # non-contiguous views
    v = m[:, ::2, 1::3]
    fv = array_fingerprint(v)
    m[5, 10, 4] += 1
    assert array_fingerprint(v) != fv


def test_fingerprint_follows_objects_and_refuses_what_it_cannot_see():
    """Round-4 review: a bound method's object and captured objects were taken by id() only
    (`mask_factories=holder.make` with `holder.mask[:] = ...` between runs -> old masks).  Their __dict__ is
    followed now; what cannot be looked into yields an OPAQUE fingerprint that never compares equal, so nothing
    that depends on it is cached (the reference re-evaluates every run, common/container.py:260-314)."""
    from libertem_amd.common.fingerprint import fingerprint, is_opaque

    class Holder:
        def __init__(self):
            self.mask = np.ones((4, 4), np.float32)
            self.radius = 3

        def make(self):
            return self.mask * self.radius

    h = Holder()
    f0 = fingerprint(h.make)
    assert not is_opaque(f0) and fingerprint(h.make) == f0
    h.mask[1, 2] = 5                                   # attribute array edited in place
    f1 = fingerprint(h.make)
    assert f1 != f0
    h.radius = 4                                       # plain attribute changed
    assert fingerprint(h.make) != f1
    # closure over an object
    h2 = Holder()
    fac = (lambda: h2.mask)
    g0 = fingerprint(fac)
    h2.mask[:] = 2
    assert fingerprint(fac) != g0
    # a module global that is an object
    global _FP_HOLDER
    _FP_HOLDER = Holder()

    def from_global():
        return _FP_HOLDER.mask
    k0 = fingerprint(from_global)
    _FP_HOLDER.mask[0, 0] = 9
    assert fingerprint(from_global) != k0

    # objects with __slots__ are followed like those with a __dict__
    class Slotted:
        __slots__ = ('a',)

        def __init__(self):
            self.a = np.zeros(3)
    sl = Slotted()
    fac_s = (lambda: sl.a)
    fs = fingerprint(fac_s)
    assert not is_opaque(fs) and fingerprint(fac_s) == fs
    sl.a[1] = 1
    assert fingerprint(fac_s) != fs
    # things that cannot be looked into: never equal, flagged
    gen = np.random.default_rng(0)
    fo = fingerprint(lambda: gen.random(3))
    assert is_opaque(fo) and fingerprint(lambda: gen.random(3)) != fo
    # immutable values a factory captures: by value, not opaque
    dt = np.dtype(np.complex64)
    assert not is_opaque(fingerprint(lambda: np.zeros(3, dtype=dt)[slice(0, 2)]))
    deep = [[[[[[np.zeros(2)]]]]]]
    assert is_opaque(fingerprint(lambda: deep))
    # modules, classes and builtins a factory names are stable, not opaque
    assert not is_opaque(fingerprint(lambda: np.ones((2, 2)) * len(str(Holder))))
    # round-5 advice: objects WITH a __dict__ whose state is not in it must not pass as transparent -- device
    # tensors, file objects, random generators, instances of C-implemented types; sets by content, dicts with keys
    import io
    import random
    import functools
    import torch
    t = torch.zeros(3)
    for hidden in (t, random.Random(1), io.BytesIO(b'abc'), np.random.RandomState(3), memoryview(b'xy')):
        fp = fingerprint(lambda h=hidden: h)
        assert is_opaque(fp), type(hidden)
    st = {1, 2, 3}
    fac_set = (lambda: st)
    s0 = fingerprint(fac_set)
    assert not is_opaque(s0) and fingerprint(fac_set) == s0
    st.add(9)
    st.discard(1)                                      # same id, same length, other members
    assert fingerprint(fac_set) != s0
    d = {'a': 1}
    fac_d = (lambda: d)
    d0 = fingerprint(fac_d)
    d['b'] = d.pop('a')                                # same values, other key
    assert fingerprint(fac_d) != d0
    # a module global that cannot be looked into makes the factory uncacheable (it used to be skipped)
    global _FP_RNG
    _FP_RNG = np.random.default_rng(5)

    def from_global_rng():
        return _FP_RNG.random(3)
    assert is_opaque(fingerprint(from_global_rng))
    # ... a scalar global counts by value
    global _FP_RADIUS
    _FP_RADIUS = 3

    def from_global_scalar():
        return np.ones(4) * _FP_RADIUS
    r0 = fingerprint(from_global_scalar)
    assert not is_opaque(r0)
    _FP_RADIUS = 4
    assert fingerprint(from_global_scalar) != r0
    # the usual factories stay cacheable
    from libertem_amd import masks as M
    assert not is_opaque(fingerprint(functools.partial(M.circular, 3, 3, 8, 8, 2)))
    import scipy.sparse as sp
    m = sp.csr_matrix(np.eye(3))
    assert not is_opaque(fingerprint(lambda: m))


def test_container_looks_for_banded_stacks():
    """MaskContainer._maybe_banded: masks in groups with ONE pixel support each (the orders of a bin of a radial-Fourier
    stack with several bins) are offered to the library as CSR before the stack is multiplied dense; rings with a
    support each, narrow stacks and scattered ones are not."""
    import scipy.sparse as sp
    from libertem_amd.common.container import _maybe_banded, _worth_densifying
    from libertem_amd.analysis.radialfourier import radial_mask_factory
    from libertem_amd import masks as pm
    st = radial_mask_factory(64, 128, 64, 32, 0, pm.bounding_radius(64, 32, 128, 64), 3, 12, True)()
    csr = st.to_px_by_masks(dtype=np.complex64)
    assert csr.shape == (64 * 128, 39)
    assert _worth_densifying(csr, np.complex64) and _maybe_banded(csr, np.complex64)
    assert not _maybe_banded(csr.astype(np.complex128), np.complex128)           # (float32 / complex64 images only)
    rings = sp.csr_matrix(pm.radial_bins(32, 32, 64, 64, n_bins=96, use_sparse=True, dtype=np.float32)
                          .to_px_by_masks(dtype=np.float32))
    assert not _maybe_banded(rings, np.float32)                                   # one mask per support
    narrow = radial_mask_factory(64, 128, 64, 32, 0, pm.bounding_radius(64, 32, 128, 64), 2, 7, True)()
    assert not _maybe_banded(narrow.to_px_by_masks(dtype=np.complex64), np.complex64)     # 32 columns: one dense pass
    scattered = sp.random(4096, 96, density=0.01, format='csr', dtype=np.float32, random_state=np.random.RandomState(3))
    assert not _maybe_banded(scattered, np.float32)


def test_host_copy_on_several_threads():
    """ltmi_host_copy (the staging copy into the upload path's page-locked bounce buffers): plain memcpy semantics for
    every size / thread count, nothing written beyond the range"""
    from libertem_amd import hip
    rng = np.random.default_rng(3)
    src = rng.integers(0, 255, (9 << 20) + 12345, dtype=np.uint8)
    for nbytes in (0, 1, 4095, 1 << 20, (1 << 20) + 1, 5 << 20, src.size):
        for threads in (0, 1, 2, 3, 7, 64, 1000):
            dst = np.full(nbytes + 64, 7, dtype=np.uint8)
            hip.host_copy(dst[:nbytes], src[:nbytes], threads)
            assert np.array_equal(dst[:nbytes], src[:nbytes]) and np.all(dst[nbytes:] == 7), (nbytes, threads)
    with pytest.raises(ValueError):
        hip.host_copy(np.zeros(4, np.uint8), np.zeros(5, np.uint8))
    with pytest.raises(ValueError):
        hip.host_copy(np.zeros((4, 4), np.uint8)[:, ::2], np.zeros((4, 2), np.uint8))


def test_in_place_page_locking_only_for_own_mappings(tmp_path, monkeypatch):
    """Round 6: host arrays are page-locked in place only where their mapping PROVABLY belongs to one object for its whole
    life (the GPU memory access fault on copies out of page-locked heap arrays was never root-caused,
    profiles/r05_host_fault.txt, r06_host_upload.txt): np.memmap / mmap objects, and ndarrays that own a glibc malloc
    chunk with a mapping of its own (IS_MMAPPED).  Anything else -- heap chunks, foreign allocators -- is staged.
    LTMI_PIN_USER_ARRAYS=1 restores round 5's rule (any array of 32 MiB and more)."""
    import torch
    from libertem_amd.io.dataset import memory as M
    mm = np.memmap(tmp_path / 'frames.bin', dtype=np.uint16, mode='w+', shape=(8, 16, 16))
    assert M._own_mapping(mm) and M._own_mapping(mm[2:5]) and M._own_mapping(mm.reshape((8, 256))[1:])
    assert M._own_mapping(np.asarray(mm))                    # an ndarray view whose base chain ends in the map
    big = np.zeros((40 << 20,), dtype=np.uint8)              # above malloc's largest mmap threshold: a chunk of its own
    assert M._own_mapping(big) and M._own_mapping(big[5:]) and M._own_mapping(big.reshape((40, -1))[3:7])
    heap = np.zeros(1000, dtype=np.uint8)                    # a heap chunk
    assert not M._own_mapping(heap)
    foreign = torch.zeros(40 << 20, dtype=torch.uint8).numpy()      # another allocator's memory: nothing to prove
    assert not M._own_mapping(foreign) and not M._own_mapping(foreign[64:])
    monkeypatch.delenv('LTMI_PIN_USER_ARRAYS', raising=False)
    assert not M.pin_user_arrays()
    asked = []

    class FakeTorch:                                         # records registration attempts, refuses them
        class cuda:
            @staticmethod
            def cudart():
                class RT:
                    @staticmethod
                    def cudaHostRegister(ptr, nbytes, flags):
                        asked.append(nbytes)
                        return 1
                return RT
    assert M._register_host(FakeTorch, foreign) is None and asked == []      # not even tried
    assert M._register_host(FakeTorch, heap) is None and asked == []
    assert M._register_host(FakeTorch, big) is None and asked == [big.nbytes]            # a chunk of its own: tried
    assert M._register_host(FakeTorch, np.asarray(mm)) is None and asked[-1] == mm.nbytes    # a map of its own: tried
    monkeypatch.setenv('LTMI_PIN_USER_ARRAYS', '1')
    assert M.pin_user_arrays()
    n = len(asked)
    assert M._register_host(FakeTorch, foreign) is None and asked[n:] == [foreign.nbytes]    # round 5's rule: >= 32 MiB
    assert M._register_host(FakeTorch, foreign[:1 << 20]) is None and len(asked) == n + 1    # ... smaller: never


def test_mask_container_get_masks_for_slice_and_getroi():
    """MaskContainer.get_masks_for_slice (common/container.py:316-333): the host matrix of a sig-only slice, dense
    and scipy CSR; analysis/getroi.py"""
    import scipy.sparse as sp
    from libertem_amd.common.container import MaskContainer
    from libertem_amd.common import Shape, Slice
    from libertem_amd.analysis.getroi import get_roi
    from libertem_amd import masks as M
    rng = np.random.default_rng(4)
    stack = rng.random((3, 8, 8)).astype(np.float32)
    stack[stack < 0.7] = 0
    sl = Slice(origin=(2, 4), shape=Shape((4, 4), sig_dims=2))
    mc = MaskContainer([lambda i=i: stack[i] for i in range(3)], dtype=np.float32, use_sparse=False, count=3)
    dense = mc.get_masks_for_slice(sl)
    assert isinstance(dense, np.ndarray) and dense.shape == (16, 3)
    assert np.array_equal(dense, stack[:, 2:6, 4:8].reshape(3, 16).T)
    assert np.array_equal(mc.get_masks_for_slice(sl, transpose=False), dense.T)
    ms = MaskContainer([lambda i=i: sp.csr_matrix(stack[i]) for i in range(3)], dtype=np.float32,
                       use_sparse='scipy.sparse', count=3)
    sparse = ms.get_masks_for_slice(sl)
    assert sp.issparse(sparse) and sparse.shape == (16, 3)
    assert np.array_equal(sparse.toarray(), dense)
    assert np.array_equal(mc.get(Slice(origin=(0, 2, 4), shape=Shape((5, 4, 4), sig_dims=2))), dense)

    assert get_roi({}, (4, 6)) is None and get_roi({"roi": {}}, (4, 6)) is None
    disk = get_roi({"roi": {"shape": "disk", "cx": 2, "cy": 1, "r": 1.5}}, (4, 6))
    assert disk.shape == (4, 6) and np.array_equal(disk, M.circular(2, 1, 6, 4, 1.5))
    rect = get_roi({"roi": {"shape": "rect", "x": 1, "y": 0, "width": 3, "height": 2}}, (4, 6))
    assert np.array_equal(rect, M.rectangular(1, 0, 3, 2, 6, 4)) and rect.sum() == 12      # (both edges belong: masks.py:370-411)
