"""
GPU parity of the product path: Context('hip').run_udf / run(analysis) with the native UDFs,
against (a) the golden vectors produced by the real reference and (b) the oracle on the same seeded
inputs.  Everything goes through the C ABI (libltmi.so); `-m gpu` only.
"""
import os

import numpy as np
import pytest

import recipes
from oracle import path as opath, masks as omasks

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

F32_TOL = 1e-5      # north_star: within 1e-5 rel for float results


@pytest.fixture(scope='module')
def ctx():
    from libertem_amd.api import Context
    assert torch.cuda.is_available()
    c = Context.make_with('hip', gpus=0)
    yield c
    c.close()


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


def _device_ds(ctx, data, num_partitions, sig_dims=2, **kw):
    from libertem_amd.common.hiparray import HipArray
    arr = HipArray.from_numpy(data, 0)
    return ctx.load('memory', data=arr, num_partitions=num_partitions, sig_dims=sig_dims, **kw)


def _close(a, b, tol):
    scale = max(np.abs(b).max(), 1e-30)
    return np.allclose(a, b, rtol=tol, atol=tol * scale)


def _com_close(got, ref, scale, tol=F32_TOL):
    """CoM tolerance (north star: 1e-5 rel).  `raw_com` = img_y/img_sum is a plain quotient of two
    f32 sums: 1e-5 relative.  Everything downstream (`raw_shifts`, `field`, x / y, magnitude,
    regression coefficients) is `raw_com - centre` or linear in it, and divergence / curl are
    np.gradient differences of neighbouring shifts with weights <= 1: the subtraction cancels the
    leading digits, so the error they inherit is ABSOLUTE, 1e-5 x |raw_com| (`scale`)."""
    return np.allclose(got, ref, rtol=tol, atol=tol * scale)


@pytest.mark.parametrize('resident', ['host', 'device'])
@pytest.mark.parametrize('case', recipes.DENSE_CASES, ids=lambda c: c['name'])
def test_apply_masks_udf_vs_reference_golden(ctx, golden_dir, case, resident):
    from libertem_amd.udf.masks import ApplyMasksUDF
    g = _load(golden_dir, 'apply_masks_dense')
    data, masks = recipes.make_dense_case(case)
    kw = dict(case.get('udf_kwargs', {}))
    udf = ApplyMasksUDF(mask_factories=lambda: masks, **kw)
    if resident == 'device':
        ds = _device_ds(ctx, data, case['num_partitions'], tileshape=case.get('tileshape'))
    else:
        ds = ctx.load('memory', data=data, num_partitions=case['num_partitions'], sig_dims=2,
                      tileshape=case.get('tileshape'))
    res = ctx.run_udf(dataset=ds, udf=udf)['intensity']
    ref = g[case['name']]
    got = res.data
    assert got.shape == ref.shape
    assert got.dtype == ref.dtype
    if ref.dtype.kind in 'iu':
        assert np.array_equal(got, ref)                 # integer path: bit exact
    else:
        tol = F32_TOL if ref.dtype in (np.float32, np.complex64) else 1e-12
        assert _close(got, ref, tol)
    assert res.raw_data.shape == (int(np.prod(case['nav'])), masks.shape[0])


def test_apply_masks_integer_sum_masks_bit_exact(ctx):
    """0/1 masks on detector counts: every partial sum < 2**24 -> float32 result is exact and
    must equal the reference path bit for bit (north_star: 'bit-exact for integer sum masks')."""
    from libertem_amd.udf.masks import ApplyMasksUDF
    rng = np.random.default_rng(77)
    data = rng.integers(0, 200, (5, 13, 128, 128)).astype(np.uint16)
    masks = (rng.random((7, 128, 128)) > 0.4)
    ref = opath.apply_masks(data, masks.astype(np.float32), num_partitions=3)
    assert ref.max() < 2**24
    ds = ctx.load('memory', data=data, num_partitions=3, sig_dims=2)
    got = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(
        mask_factories=[(lambda i=i: masks[i].astype(np.float32)) for i in range(7)]))
    assert np.array_equal(got['intensity'].data, ref)
    # the genuinely integer path: preferred_dtype / mask_dtype int32 (udf/masks.py:311-322)
    got_i = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(
        mask_factories=lambda: masks, preferred_dtype=np.int32, mask_dtype=np.int32))
    assert got_i['intensity'].data.dtype == np.int32
    assert np.array_equal(got_i['intensity'].data, ref.astype(np.int32))


def test_apply_masks_mask_factory_variants(ctx):
    """list of factories vs one stack factory vs scipy.sparse masks
    (tests/analysis/test_analysis_masks.py:215-474 style)."""
    import scipy.sparse as sp
    from libertem_amd.udf.masks import ApplyMasksUDF
    rng = np.random.default_rng(3)
    data = rng.integers(0, 100, (4, 6, 24, 40)).astype(np.uint16)
    m0 = rng.random((24, 40)).astype(np.float32)
    m1 = np.where(rng.random((24, 40)) < 0.1, rng.random((24, 40)), 0).astype(np.float32)
    ref = opath.apply_masks(data, np.stack([m0, m1]), num_partitions=2)
    ds = ctx.load('memory', data=data, num_partitions=2, sig_dims=2)
    a = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=[lambda: m0, lambda: m1]))
    b = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=lambda: np.stack([m0, m1])))
    c = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(
        mask_factories=[lambda: m0, lambda: sp.csr_matrix(m1)]))        # mixed -> dense
    for r in (a, b, c):
        assert _close(r['intensity'].data, ref, F32_TOL)
    assert a['intensity'].data.shape == (4, 6, 2)


@pytest.mark.parametrize('case', recipes.SUM_CASES, ids=lambda c: c['name'])
def test_sum_udfs_vs_reference_golden(ctx, golden_dir, case):
    from libertem_amd.udf.sum import SumUDF
    from libertem_amd.udf.sumsigudf import SumSigUDF
    g = _load(golden_dir, 'sums')
    data = recipes.make_sum_case(case)
    for resident in ('host', 'device'):
        if resident == 'device':
            ds = _device_ds(ctx, data, case['num_partitions'], tileshape=case.get('tileshape'))
        else:
            ds = ctx.load('memory', data=data, num_partitions=case['num_partitions'], sig_dims=2,
                          tileshape=case.get('tileshape'))
        s = ctx.run_udf(dataset=ds, udf=SumUDF(**case.get('sum_kwargs', {})))['intensity'].data
        ss = ctx.run_udf(dataset=ds, udf=SumSigUDF())['intensity'].data
        rs, rss = g[case['name'] + '__sum'], g[case['name'] + '__sumsig']
        assert s.dtype == rs.dtype and s.shape == rs.shape
        assert ss.dtype == rss.dtype and ss.shape == rss.shape
        # complex frames have parts of both signs: the sums cancel, so the float32 round-off is
        # relative to sum |x| (1e-6 of it), not to the sum
        cplx = np.dtype(case['dtype']).kind == 'c'
        a_s = 1e-6 * np.abs(data).sum(axis=(0, 1)).max() if cplx else 0
        a_ss = 1e-6 * np.abs(data).sum(axis=(2, 3)).max() if cplx else 0
        assert np.allclose(s, rs, rtol=1e-6, atol=a_s) and np.allclose(ss, rss, rtol=1e-6, atol=a_ss)
        if np.dtype(case['dtype']).kind in 'iu':
            assert np.array_equal(s, rs) and np.array_equal(ss, rss)    # exact below 2**24


@pytest.mark.parametrize('case', recipes.COM_CASES, ids=lambda c: c['name'])
def test_com_vs_reference_golden(ctx, golden_dir, case):
    from libertem_amd.udf.com import CoMUDF
    g = _load(golden_dir, 'com')
    data = recipes.make_com_case(case)
    ds = ctx.load('memory', data=data, num_partitions=case['num_partitions'], sig_dims=2)
    res = ctx.run_udf(dataset=ds, udf=CoMUDF.with_params(**case['params']))
    assert 'raw_mask_result' not in res                   # use='private'
    scale = np.abs(g[f"{case['name']}__udf__raw_com"]).max()
    for k, v in res.items():
        ref = g[f"{case['name']}__udf__{k}"]
        assert v.data.shape == ref.shape, k
        assert v.data.dtype == ref.dtype, k
        if k == 'raw_com':
            assert np.allclose(v.data, ref, rtol=F32_TOL, atol=0), k
        else:
            assert _com_close(v.data, ref, scale), k
    # the analysis flavour (x first in `field`, float defaults cx = W/2)
    p = case['analysis_params']
    analysis = ctx.create_com_analysis(
        dataset=ds, cx=p.get('cx'), cy=p.get('cy'), mask_radius=p.get('r'),
        mask_radius_inner=p.get('ri'), flip_y=p.get('flip_y', False),
        scan_rotation=p.get('scan_rotation', 0.))
    ares = ctx.run(analysis)
    for k in ('x', 'y', 'magnitude', 'divergence', 'curl'):
        assert _com_close(ares[k].raw_data, g[f"{case['name']}__analysis__{k}"], scale), k
    assert _com_close(ares.field.raw_data[0], g[f"{case['name']}__analysis__x"], scale)


def test_com_vs_scipy_center_of_mass(ctx):
    """reference tests/analysis/test_analysis_com.py:55-89"""
    import scipy.ndimage
    from libertem_amd.udf.com import CoMUDF
    data = recipes.make_com_case(recipes.COM_CASES[0]).astype(np.float32)
    ds = ctx.load('memory', data=data, num_partitions=2, sig_dims=2)
    res = ctx.run_udf(dataset=ds, udf=CoMUDF.with_params())
    for i in range(data.shape[0]):
        for j in range(data.shape[1]):
            cy, cx = scipy.ndimage.center_of_mass(data[i, j].astype(np.float64))
            assert np.allclose(res['raw_com'].data[i, j], (cy, cx), rtol=F32_TOL)


@pytest.mark.parametrize('case', [c for c in recipes.RF_CASES
                                  if not c['name'].startswith('heuristic')],
                         ids=lambda c: c['name'])
def test_radial_fourier_vs_reference_golden(ctx, golden_dir, case):
    g = _load(golden_dir, 'radial_fourier')
    data = recipes.make_rf_case(case)
    ds = ctx.load('memory', data=data, num_partitions=case['num_partitions'], sig_dims=2)
    analysis = ctx.create_radial_fourier_analysis(dataset=ds, **case['params'])
    res = ctx.run(analysis)
    ref = g[f"{case['name']}__raw_results"]
    assert res.raw_results.shape == ref.shape and res.raw_results.dtype == ref.dtype
    assert _close(res.raw_results, ref, F32_TOL)
    assert res.complex_0_1.raw_data.shape == tuple(case['nav'])
    assert np.array_equal(res.dominant_0.raw_data.shape, case['nav'])


@pytest.mark.parametrize('resident', ['host', 'device'])
def test_c5_workload_radial_fourier_1024(ctx, golden_dir, resident):
    """BASELINE.json C5 at its real detector size: create_radial_fourier_analysis DEFAULTS
    (n_bins=1, max_order=24 -> 25 dense complex64 masks = 50 real columns: the 3 MFMA groups + 2
    VALU columns kernel over the 256 MiB image) on 1024x1024 float32 frames, reduced nav; against
    the reference's output (golden) and the oracle, 1e-5 of max|ref|."""
    case = next(c for c in recipes.WORKLOAD_CASES if c['name'] == 'c5_rf_1024')
    g = _load(golden_dir, 'config_workloads')
    data = recipes.make_workload_case(case)
    if resident == 'device':
        ds = _device_ds(ctx, data, case['num_partitions'])
    else:
        ds = ctx.load('memory', data=data, num_partitions=case['num_partitions'], sig_dims=2)
    analysis = ctx.create_radial_fourier_analysis(dataset=ds)
    assert analysis.parameters['use_sparse'] is False
    res = ctx.run(analysis)
    ref = g['c5_rf_1024__raw_results']
    assert res.raw_results.shape == ref.shape == (1, 25, 2, 4)
    assert res.raw_results.dtype == ref.dtype == np.complex64
    assert _close(res.raw_results, ref, F32_TOL)
    ora = opath.radial_fourier_analysis(data, num_partitions=case['num_partitions'])
    assert _close(res.raw_results, ora['raw_results'], F32_TOL)
    udf_res = ctx.run_udf(dataset=ds, udf=analysis.get_udf())['intensity'].data
    assert _close(udf_res, g['c5_rf_1024__intensity'], F32_TOL)


@pytest.mark.parametrize('resident', ['host', 'device'])
def test_c3_workload_com_512(ctx, golden_dir, resident):
    """BASELINE.json C3 at its real detector size: COMAnalysis (default and mask_radius=200) and
    CoMUDF on 512x512 uint16 counts in [0, 4096) -- gradient ramps x 262 144 pixels, sums ~1e11 in
    float32 -- against the reference's output (golden) and the oracle."""
    from libertem_amd.udf.com import CoMUDF
    case = next(c for c in recipes.WORKLOAD_CASES if c['name'] == 'c3_com_512')
    g = _load(golden_dir, 'config_workloads')
    data = recipes.make_workload_case(case)
    if resident == 'device':
        ds = _device_ds(ctx, data, case['num_partitions'])
    else:
        ds = ctx.load('memory', data=data, num_partitions=case['num_partitions'], sig_dims=2)
    scale = 512.
    for i, ap in enumerate(case['analysis_params']):
        analysis = ctx.create_com_analysis(dataset=ds, cx=ap['cx'], cy=ap['cy'],
                                           mask_radius=ap.get('r'))
        inten = ctx.run_udf(dataset=ds, udf=analysis.get_udf())['intensity'].data
        ref_i = g[f'c3_com_512__analysis{i}__intensity']
        assert inten.dtype == ref_i.dtype and inten.shape == ref_i.shape
        for c in range(3):                  # the three raw sums, each relative to its own maximum
            assert _close(inten[..., c], ref_i[..., c], F32_TOL), (i, c)
        # ELEMENT-WISE where the data allow it (round-4 review): non-negative frames give a non-negative total and,
        # with the centre added back, non-negative first moments sum_p y x_p, sum_p x x_p -- no cancellation, so
        # every scan position has to agree to 1e-5 relative with the reference's numbers, no absolute term
        assert np.allclose(inten[..., 0], ref_i[..., 0], rtol=F32_TOL, atol=0), i
        for c, centre in ((1, ap['cy']), (2, ap['cx'])):
            got_m = inten[..., c].astype(np.float64) + centre * inten[..., 0].astype(np.float64)
            ref_m = ref_i[..., c].astype(np.float64) + centre * ref_i[..., 0].astype(np.float64)
            assert (ref_m > 0).all()
            assert np.allclose(got_m, ref_m, rtol=F32_TOL, atol=0), (i, c)
        ares = ctx.run(analysis)
        ora = opath.com_analysis(data, num_partitions=case['num_partitions'], **ap)
        for k in ('x', 'y', 'magnitude', 'divergence', 'curl'):
            assert _com_close(ares[k].raw_data, g[f'c3_com_512__analysis{i}__{k}'], scale), (i, k)
            assert _com_close(ares[k].raw_data, ora[k], scale), (i, k)
    for i, up in enumerate(case['udf_params']):
        res = ctx.run_udf(dataset=ds, udf=CoMUDF.with_params(**up))
        ora = opath.com_udf(data, num_partitions=case['num_partitions'], **up)
        for k, v in res.items():
            ref = g[f'c3_com_512__udf{i}__{k}']
            assert v.data.shape == ref.shape and v.data.dtype == ref.dtype, k
            if k == 'raw_com':
                assert np.allclose(v.data, ref, rtol=F32_TOL, atol=0), k
                assert np.allclose(v.data, ora[k], rtol=F32_TOL, atol=0), k
            else:
                assert _com_close(v.data, ref, scale), (i, k)
                assert _com_close(v.data, ora[k], scale), (i, k)


def test_single_mask_and_sum_analyses(ctx, golden_dir):
    g = _load(golden_dir, 'single_mask_analyses')
    data = recipes.make_com_case(recipes.COM_CASES[0])
    ds = ctx.load('memory', data=data, num_partitions=2, sig_dims=2)
    a = ctx.run(ctx.create_disk_analysis(dataset=ds))
    assert _close(a.intensity.raw_data, g['disk_default'][..., 0], F32_TOL)
    # the result keys of the single-mask analyses (analysis/masks.py:63-76; tests/analysis/test_analysis_shapes.py)
    assert [r.key for r in a.results] == ['intensity', 'intensity_log']
    assert np.array_equal(a.intensity_log.raw_data, a.intensity.raw_data) and a['intensity_log'].title == 'intensity [log]'
    a = ctx.run(ctx.create_disk_analysis(dataset=ds, cx=10, cy=20, r=5))
    assert _close(a.intensity.raw_data, g['disk_params'][..., 0], F32_TOL)
    a = ctx.run(ctx.create_ring_analysis(dataset=ds))
    assert _close(a.intensity.raw_data, g['ring_default'][..., 0], F32_TOL)
    a = ctx.run(ctx.create_ring_analysis(dataset=ds, cx=10, cy=20, ri=3, ro=8))
    assert _close(a.intensity.raw_data, g['ring_params'][..., 0], F32_TOL)
    for name, factories, kw in recipes_single_masks():
        a = ctx.run(ctx.create_mask_analysis(factories=factories, dataset=ds, **kw))
        ref = g[name]
        assert a.mask_0.raw_data.dtype == ref.dtype
        assert _close(a.mask_1.raw_data, ref[..., 1], F32_TOL if ref.dtype == np.float32 else 1e-12)
    s = ctx.run(ctx.create_sum_analysis(dataset=ds))
    assert np.array_equal(s.intensity.raw_data, data.astype(np.float32).sum(axis=(0, 1)))
    # point analysis = one-entry sparse mask -> picks a pixel
    a = ctx.run(ctx.create_point_analysis(dataset=ds, x=7, y=9))
    assert np.array_equal(a.intensity.raw_data, data[..., 9, 7].astype(np.float32))


def recipes_single_masks():
    out = []
    for name, cls, params in recipes.single_mask_analyses():
        if cls == 'masks':
            p = dict(params)
            out.append((name, p.pop('factories'), p))
    return out


def test_roi_and_multiple_udfs(ctx):
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd.udf.sumsigudf import SumSigUDF
    from libertem_amd.udf.sum import SumUDF
    rng = np.random.default_rng(9)
    data = rng.integers(0, 100, (6, 7, 32, 32)).astype(np.uint16)
    masks = rng.random((3, 32, 32)).astype(np.float32)
    roi = np.zeros((6, 7), dtype=bool)
    roi[1:5, 2:6] = True
    roi[5, 6] = True
    ref = opath.apply_masks(data, masks, num_partitions=4)
    for ds in (ctx.load('memory', data=data, num_partitions=4, sig_dims=2),
               _device_ds(ctx, data, 4)):
        r1, r2, r3 = ctx.run_udf(dataset=ds, udf=[
            ApplyMasksUDF(mask_factories=lambda: masks), SumSigUDF(), SumUDF()], roi=roi)
        d = r1['intensity'].data
        assert np.all(np.isnan(d[~roi]))
        assert _close(d[roi], ref[roi], F32_TOL)
        assert np.array_equal(r2['intensity'].data[roi], data.sum(axis=(2, 3))[roi])
        assert np.array_equal(r3['intensity'].data, data[roi].astype(np.float32).sum(axis=0))


def test_full_size_properties(ctx):
    """At the full C2 size (the oracle cannot finish that in seconds): size-independent properties.
    linearity apply(a*m1 + m2) == a*apply(m1) + apply(m2); all-ones mask == SumSigUDF;
    sum over nav of SumSig == sum over sig of SumUDF (checksum of checksums)."""
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd.udf.sumsigudf import SumSigUDF
    from libertem_amd.udf.sum import SumUDF
    n = 256 * 256                       # BASELINE.json C2: 65 536 frames of 256x256 uint16 (8 GiB)
    g = torch.Generator(device='cuda').manual_seed(5)
    frames = torch.empty((n, 256 * 256), dtype=torch.int16, device='cuda')
    for i in range(0, n, 4096):
        frames[i:i + 4096] = torch.randint(0, 64, (4096, 256 * 256), generator=g, device='cuda',
                                           dtype=torch.int32).to(torch.int16)
    frames = frames.reshape((256, 256, 256, 256))
    ds = ctx.load('memory', data=frames, dtype=np.uint16, sig_dims=2, num_partitions=2)
    rng = np.random.default_rng(6)
    m1 = (rng.random((256, 256)) > 0.5).astype(np.float32)
    m2 = (rng.random((256, 256)) > 0.5).astype(np.float32)
    ones = np.ones((256, 256), dtype=np.float32)
    stack = np.stack([m1, m2, 2 * m1 + m2, ones])
    r = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=lambda: stack))['intensity'].data
    # integer-valued, < 2**24: exact
    assert r.max() < 2**24
    assert np.array_equal(r[..., 2], 2 * r[..., 0] + r[..., 1])
    ss = ctx.run_udf(dataset=ds, udf=SumSigUDF())['intensity'].data
    assert np.array_equal(r[..., 3], ss)
    s = ctx.run_udf(dataset=ds, udf=SumUDF())['intensity'].data
    assert np.isclose(s.astype(np.float64).sum(), ss.astype(np.float64).sum(), rtol=1e-6)
    # spot-check 32 random frames against the oracle
    idx = rng.choice(n, 32, replace=False)
    sub = frames.reshape((n, 256, 256))[torch.as_tensor(idx, device='cuda')].cpu().numpy()
    ref = opath.apply_masks(sub.view(np.uint16)[None], stack)
    assert np.array_equal(r.reshape((n, 4))[idx], ref[0])


def test_c2_full_size_16_masks_benchmarked_kernel(ctx):
    """BASELINE.json C2 exactly as bench.py runs it: 65 536 frames of 256x256 uint16 in [0, 4096),
    the 16-mask float32 stack of rng(2), ONE partition -> the matrix-core kernel
    k_dense_lds<NG=1, 2 tiles> on a 512-workgroup grid (not the VALU-column variant a 4-mask stack
    selects).  Checks: the kernel that ran; 32 frames against the oracle ELEMENT-wise (all data
    positive: rtol 1e-5, no norm-wise slack); exact linearity under a power-of-two scaling of the
    stack; linearity in the masks to 1e-5."""
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd import hip
    n = 256 * 256
    g = torch.Generator(device='cuda').manual_seed(11)
    frames = torch.empty((n, 256 * 256), dtype=torch.int16, device='cuda')
    for i in range(0, n, 4096):
        frames[i:i + 4096] = torch.randint(0, 4096, (4096, 256 * 256), generator=g, device='cuda',
                                           dtype=torch.int16)
    ds = ctx.load('memory', data=frames.reshape((256, 256, 256, 256)), dtype=np.uint16, sig_dims=2,
                  num_partitions=1)
    masks = np.random.default_rng(2).random((16, 256, 256)).astype(np.float32)
    udf = ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False, mask_count=16,
                        mask_dtype=np.float32)
    hip.KernelTimer.start()
    r = ctx.run_udf(dataset=ds, udf=udf)['intensity'].data
    kernels = [k for _, _, k in hip.KernelTimer.stop()]
    assert len(kernels) == 1, kernels
    assert 'k_dense_lds' in kernels[0] and 'NG=1' in kernels[0] and 'grid=(512' in kernels[0], kernels
    assert r.shape == (256, 256, 16) and r.dtype == np.float32
    flat = r.reshape((n, 16))
    rng = np.random.default_rng(12)
    idx = np.unique(np.concatenate([[0, n - 1], rng.choice(n, 32, replace=False)]))
    sub = frames[torch.as_tensor(idx, device='cuda')].cpu().numpy().view(np.uint16)
    ref = opath.apply_masks(sub.reshape((1, len(idx), 256, 256)), masks)[0]
    assert np.allclose(flat[idx], ref, rtol=F32_TOL, atol=0)
    ref64 = sub.astype(np.float64) @ masks.reshape((16, -1)).T.astype(np.float64)
    assert np.allclose(flat[idx], ref64, rtol=F32_TOL, atol=0)
    # a power-of-two scaling of the stack is exact in float32
    masks4 = masks * np.float32(4)
    r4 = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=lambda: masks4, use_sparse=False,
                                                   mask_count=16))['intensity'].data
    assert np.array_equal(r4, 4 * r)
    # linearity in the masks: column 15 := 0.5 * m0 + m1 - 0.25 * m2
    mixed = masks.copy()
    mixed[15] = 0.5 * masks[0] + masks[1] - 0.25 * masks[2]
    rm = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=lambda: mixed, use_sparse=False,
                                                   mask_count=16))['intensity'].data
    assert np.array_equal(rm[..., :15], r[..., :15])
    assert np.allclose(rm[..., 15], 0.5 * r[..., 0] + r[..., 1] - 0.25 * r[..., 2], rtol=F32_TOL,
                       atol=0)


@pytest.mark.parametrize('resident', ['host', 'device'])
def test_c4_workload_through_run_udf(ctx, resident):
    """BASELINE.json C4 through the product wiring: radial_bins(n_bins=1024, use_sparse=True) ->
    MaskContainer -> CSR -> blocked device image -> k_bell_apply -> delivery of the wide
    (frames x 1024) result, on 4096 frames of 256x256 uint16, host- and device-resident, and with
    result_where='device'.  Against the oracle's restatement of the reference's CSR loop
    (oracle.path.apply_masks_sparse) on 24 frames and float64 scipy on all of them."""
    import scipy.sparse as sp
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd import masks as M
    from libertem_amd import hip
    from libertem_amd.common.hiparray import HipArray
    rng = np.random.default_rng(44)
    data = rng.integers(0, 4096, (16, 256, 256, 256), dtype=np.uint16)
    if resident == 'device':
        ds = _device_ds(ctx, data, 2)
    else:
        ds = ctx.load('memory', data=data, sig_dims=2, num_partitions=2)

    def rings():
        return M.radial_bins(centerX=128, centerY=128, imageSizeX=256, imageSizeY=256,
                             n_bins=1024, use_sparse=True, dtype=np.float32)
    udf = ApplyMasksUDF(mask_factories=rings, use_sparse='scipy.sparse', mask_count=1024,
                        mask_dtype=np.float32)
    hip.KernelTimer.start()
    res = ctx.run_udf(dataset=ds, udf=udf)
    kernels = {k.split(' ')[0] for _, _, k in hip.KernelTimer.stop()}
    assert kernels and all(k.startswith(('k_bell_apply', 'k_bell_flat', 'k_scatter')) for k in kernels), kernels
    got = res['intensity'].data
    assert got.shape == (16, 256, 1024) and got.dtype == np.float32
    stack = sp.csr_matrix(omasks.radial_bins(128, 128, 256, 256, n_bins=1024, use_sparse=True,
                                             dtype=np.float32))
    assert stack.shape == (1024, 65536) and stack.nnz == 432407
    flat = data.reshape((4096, 65536))
    ref64 = np.asarray(flat.astype(np.float64) @ stack.T.astype(np.float64))
    scale = np.abs(ref64).max()
    g = got.reshape((4096, 1024))
    assert np.allclose(g, ref64, rtol=F32_TOL, atol=F32_TOL * scale)
    # element-wise, every result (all data and weights >= 0: no cancellation), also the innermost rings that
    # hold a handful of pixels
    assert np.allclose(g, ref64, rtol=F32_TOL, atol=0)
    assert np.array_equal(g == 0, ref64 == 0)
    idx = np.concatenate([[0, 4095], rng.choice(4096, 22, replace=False)])
    ref = opath.apply_masks_sparse(flat[idx].reshape((1, 24, 256, 256)), stack)[0]
    assert ref.dtype == np.float32
    assert np.allclose(g[idx], ref, rtol=F32_TOL, atol=F32_TOL * scale)
    # the 16 MiB result kept in HBM
    dev = ctx.run_udf(dataset=ds, udf=udf, result_where='device')
    assert isinstance(dev['intensity'].device_data, HipArray)
    assert np.array_equal(dev['intensity'].data, got)


def test_write_once_rows_not_with_frame_cutting_tileshape(ctx):
    """A tileshape forced on the dataset that cuts the frames replaces the negotiated scheme when the
    tiles are read (intent 'frame': ApplyMasksUDF next to a process_frame UDF): result rows then are
    sums over several tiles -- the write-once shortcut (`=` into an un-zeroed buffer) must be off."""
    from libertem_amd.udf.base import UDF
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd.udf.sumsigudf import SumSigUDF

    class FrameSeen(UDF):
        def get_backends(self):
            return (UDF.BACKEND_HIP,)

        def get_result_buffers(self):
            return {'seen': self.buffer(kind='nav', dtype=np.float32)}

        def process_frame(self, frame):
            self.results.seen[:] += 1

    rng = np.random.default_rng(71)
    data = rng.integers(0, 1000, (3, 7, 32, 64)).astype(np.uint16)
    masks = rng.random((4, 32, 64)).astype(np.float32)
    ref = opath.apply_masks(data, masks, num_partitions=2)
    for tileshape in ((5, 8, 64), (3, 32, 64)):
        ds = ctx.load('memory', data=data, num_partitions=2, sig_dims=2, tileshape=tileshape)
        res = ctx.run_udf(dataset=ds, udf=[ApplyMasksUDF(mask_factories=lambda: masks),
                                           SumSigUDF(), FrameSeen()])
        assert _close(res[0]['intensity'].data, ref, F32_TOL), tileshape
        assert np.array_equal(res[1]['intensity'].data, data.astype(np.float32).sum(axis=(2, 3)))
        assert np.all(res[2]['seen'].data >= 1)
        alone = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=lambda: masks))
        assert _close(alone['intensity'].data, ref, F32_TOL), tileshape


def test_sparse_complex128_stack_stays_sparse(ctx):
    """A sparse complex128 stack on real frames (float64 / uint32 data, or complex128 mask values):
    the float64 gather kernel on (re, im) column pairs instead of a densified stack -- result dtype
    complex128 like the reference's rmatmul (common/numba/__init__.py:126: result_type of both)."""
    import scipy.sparse as sp
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd import hip
    rng = np.random.default_rng(53)
    data = rng.integers(0, 5000, (3, 5, 64, 64)).astype(np.uint32)
    dense = []
    for _ in range(6):
        keep = rng.random((64, 64)) < 0.02
        dense.append(np.where(keep, rng.random((64, 64)) - 0.5 + 1j * (rng.random((64, 64)) - 0.5), 0))
    facs = [(lambda d=d: sp.csr_matrix(d.astype(np.complex128))) for d in dense]
    ref = np.tensordot(data.astype(np.complex128), np.stack(dense), axes=([2, 3], [1, 2]))
    for ds in (_device_ds(ctx, data, 2), ctx.load('memory', data=data, num_partitions=2, sig_dims=2)):
        hip.KernelTimer.start()
        got = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=facs,
                                                        use_sparse='scipy.sparse'))
        kernels = {k.split(' ')[0] for _, _, k in hip.KernelTimer.stop()}
        assert kernels and all('k_sell_apply' in k and 'f64' in k for k in kernels), kernels
        r = got['intensity'].data
        assert r.dtype == np.complex128 and r.shape == (3, 5, 6)
        assert np.allclose(r, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
    # float32 frames with a complex128 mask dtype: same route
    dataf = rng.random((2, 4, 64, 64)).astype(np.float32)
    got = ctx.run_udf(dataset=ctx.load('memory', data=dataf, num_partitions=2, sig_dims=2),
                      udf=ApplyMasksUDF(mask_factories=facs, use_sparse='scipy.sparse',
                                        mask_dtype=np.complex128))['intensity'].data
    reff = np.tensordot(dataf.astype(np.complex128), np.stack(dense), axes=([2, 3], [1, 2]))
    assert got.dtype == np.complex128
    assert np.allclose(got, reff, rtol=1e-12, atol=1e-12 * np.abs(reff).max())


def test_sparse_integer_stack_stays_sparse(ctx):
    """Integer sparse masks on integer frames: integer result dtype with NumPy's / SciPy's wrap-around
    (reference: mask dtype = result_type(mask, frames), common/container.py + rmatmul), bit exact --
    through the float64 gather kernel + truncation while every partial sum stays below 2^52, without
    densifying the stack; a stack whose sums can exceed that goes to the dense integer kernels."""
    import scipy.sparse as sp
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd import hip
    rng = np.random.default_rng(77)
    data = rng.integers(0, 60000, (3, 5, 64, 64)).astype(np.uint16)
    dense = []
    for _ in range(6):
        keep = rng.random((64, 64)) < 0.03
        dense.append(np.where(keep, rng.integers(-9, 10, (64, 64)), 0).astype(np.int64))
    stack = np.stack(dense)
    facs = [(lambda d=d: sp.csr_matrix(d)) for d in dense]
    ref = np.tensordot(data.astype(np.int64), stack, axes=([2, 3], [1, 2]))
    for ds in (_device_ds(ctx, data, 2), ctx.load('memory', data=data, num_partitions=2, sig_dims=2)):
        hip.KernelTimer.start()
        got = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(
            mask_factories=facs, use_sparse='scipy.sparse', preferred_dtype=np.int64,
            mask_dtype=np.int64))
        kernels = {k for _, _, k in hip.KernelTimer.stop()}
        assert kernels and all('k_sell_apply' in k and 'exact-int' in k for k in kernels), kernels
        r = got['intensity'].data
        assert r.dtype == np.int64 and np.array_equal(r, ref)
    # a narrower integer result dtype
    got16 = ctx.run_udf(dataset=ctx.load('memory', data=data, num_partitions=2, sig_dims=2),
                        udf=ApplyMasksUDF(mask_factories=facs, use_sparse='scipy.sparse',
                                          mask_dtype=np.int16, preferred_dtype=np.int16)
                        )['intensity'].data
    # (result_type(int16, uint16) = int32; the wrap-around of narrow result dtypes is checked at the
    # kernel level: test_kernels_gpu.py::test_sparse_integer_results_bit_exact)
    assert got16.dtype == np.int32 and np.array_equal(got16, ref.astype(np.int32))
    # with a region of interest (row list over the resident frames)
    roi = rng.random((3, 5)) < 0.5
    part = ctx.run_udf(dataset=_device_ds(ctx, data, 2), roi=roi,
                       udf=ApplyMasksUDF(mask_factories=facs, use_sparse='scipy.sparse',
                                         preferred_dtype=np.int64, mask_dtype=np.int64))
    assert np.array_equal(part['intensity'].raw_data, ref[roi])
    # sums that can exceed 2^52: still exact, on the dense integer kernels
    huge = [(lambda d=d: sp.csr_matrix(d * (1 << 40))) for d in dense]
    hip.KernelTimer.start()
    got = ctx.run_udf(dataset=ctx.load('memory', data=data, num_partitions=2, sig_dims=2),
                      udf=ApplyMasksUDF(mask_factories=huge, use_sparse='scipy.sparse',
                                        preferred_dtype=np.int64, mask_dtype=np.int64))
    kernels = {k for _, _, k in hip.KernelTimer.stop()}
    assert kernels and not any('k_sell_apply' in k for k in kernels), kernels
    assert np.array_equal(got['intensity'].data, ref * (1 << 40))


def test_masks_modified_in_place_between_runs(ctx):
    """The reference evaluates the mask factories on every run (udf/masks.py:331-351): an array the
    factory closes over may change between two run_udf calls.  The cached device image / the cached
    plan are keyed on a content fingerprint of what the factories can see -- same udf object, same
    factory object, new values -> new result (device- and host-resident frames; dense and sparse)."""
    import scipy.sparse as sp
    from libertem_amd.udf.masks import ApplyMasksUDF
    rng = np.random.default_rng(91)
    data = rng.integers(0, 1000, (4, 6, 64, 64)).astype(np.uint16)
    masks = rng.random((16, 64, 64)).astype(np.float32)
    for ds in (_device_ds(ctx, data, 2), ctx.load('memory', data=data, num_partitions=2, sig_dims=2)):
        udf = ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False, mask_count=16)
        for change in (lambda: None, lambda: masks.__imul__(np.float32(2)),
                       lambda: masks.__setitem__((3, slice(8, 9)), 7.0),
                       lambda: masks.__setitem__((15, 63, 63), -1.0)):
            change()
            for _ in range(2):                      # second run of the same state: cache / plan hit
                got = ctx.run_udf(dataset=ds, udf=udf)['intensity'].data
                assert _close(got, opath.apply_masks(data, masks, num_partitions=2), F32_TOL)
    # (round-3 review) a 4 MiB stack edited in a column band / a small block: an evenly strided sample saw
    # columns 0..15 of every row only
    data2 = rng.integers(0, 1000, (2, 3, 256, 256)).astype(np.uint16)
    big = rng.random((16, 256, 256)).astype(np.float32)
    ds2 = _device_ds(ctx, data2, 1)
    udf2 = ApplyMasksUDF(mask_factories=lambda: big, use_sparse=False, mask_count=16)
    for change in (lambda: None, lambda: big.__setitem__((slice(None), slice(None), slice(100, 110)), 0),
                   lambda: big.__setitem__((3, slice(50, 60), slice(60, 70)), 7.0)):
        change()
        for _ in range(2):
            got = ctx.run_udf(dataset=ds2, udf=udf2)['intensity'].data
            assert _close(got, opath.apply_masks(data2, big, num_partitions=1), F32_TOL)
    # a sparse stack whose value array is scaled in place
    dense = [np.where(rng.random((64, 64)) < 0.05, rng.random((64, 64)), 0).astype(np.float32)
             for _ in range(5)]
    mats = [sp.csr_matrix(d) for d in dense]
    facs = [(lambda m=m: m) for m in mats]
    ds = _device_ds(ctx, data, 2)
    udf = ApplyMasksUDF(mask_factories=facs, use_sparse='scipy.sparse')
    r0 = ctx.run_udf(dataset=ds, udf=udf)['intensity'].data
    assert _close(r0, opath.apply_masks(data, np.stack(dense), num_partitions=2), F32_TOL)
    mats[2].data *= np.float32(4)
    r1 = ctx.run_udf(dataset=ds, udf=udf)['intensity'].data
    assert np.array_equal(r1[..., 2], 4 * r0[..., 2]) and np.array_equal(r1[..., :2], r0[..., :2])


def test_launch_ahead_of_the_bookkeeping(ctx):
    """A plan that is run again and again (same udf object, device-resident frames): from the third run on
    the mask launch of every partition is enqueued as soon as the run's result buffer exists and the tile
    loop's own call is recognised and skipped (hip.LaunchReplay) -- same results, in caller-owned arrays; masks
    edited in place are noticed by the deferred content comparison and the run is repeated with the new
    stack; another udf object does not inherit anything."""
    from libertem_amd import hip
    from libertem_amd.udf.masks import ApplyMasksUDF
    rng = np.random.default_rng(17)
    data = rng.integers(0, 1000, (6, 8, 64, 64)).astype(np.uint16)
    masks = rng.random((16, 64, 64)).astype(np.float32)
    ds = _device_ds(ctx, data, 2)
    udf = ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False, mask_count=16)
    n0 = hip.LaunchReplay.n_ahead
    got = []
    for rep in range(6):
        hip.KernelTimer.start()
        r = ctx.run_udf(dataset=ds, udf=udf)['intensity'].data
        ev = hip.KernelTimer.stop()
        assert len(ev) == 2, ev                           # one launch per partition, never two
        got.append(r)
        assert _close(r, opath.apply_masks(data, masks, num_partitions=2), F32_TOL)
    assert hip.LaunchReplay.n_ahead - n0 == 2 * 4          # runs 3 .. 6, two partitions
    assert all(np.array_equal(g, got[0]) for g in got) and len({g.ctypes.data for g in got}) == 6
    masks *= np.float32(3)
    for rep in range(4):
        r = ctx.run_udf(dataset=ds, udf=udf)['intensity'].data
        assert _close(r, opath.apply_masks(data, masks, num_partitions=2), F32_TOL)
    assert hip.LaunchReplay.expected is None and hip.LaunchReplay.recording is None
    other = ApplyMasksUDF(mask_factories=lambda: masks[:3], use_sparse=False, mask_count=3)
    r = ctx.run_udf(dataset=ds, udf=other)['intensity'].data
    assert _close(r, opath.apply_masks(data, masks[:3], num_partitions=2), F32_TOL)
    # a run with a region of interest plans afresh (no launch-ahead), the next plain run uses it again
    roi = np.zeros((6, 8), bool)
    roi[1::2] = True
    r = ctx.run_udf(dataset=ds, udf=udf, roi=roi)['intensity'].raw_data
    assert _close(r, opath.apply_masks(data, masks, num_partitions=2)[roi], F32_TOL)
    r = ctx.run_udf(dataset=ds, udf=udf)['intensity'].data
    assert _close(r, opath.apply_masks(data, masks, num_partitions=2), F32_TOL)


def test_mask_cache_reuse_and_eviction(ctx):
    from libertem_amd.udf import masks as um
    rng = np.random.default_rng(10)
    data = rng.integers(0, 100, (2, 4, 32, 32)).astype(np.uint16)
    ds = ctx.load('memory', data=data, num_partitions=2, sig_dims=2)
    um.clear_mask_cache()
    m = rng.random((2, 32, 32)).astype(np.float32)

    class Counter:
        # (an attribute of the CLASS: nothing a fingerprint follows -- a counter in a module global, a captured
        #  dict / list or an instance attribute counts as a parameter that changes with every evaluation and makes
        #  the factory uncacheable, common/fingerprint.py)
        calls = 0

    def factory():
        Counter.calls += 1
        return m
    udf = um.ApplyMasksUDF(mask_factories=factory)
    a = ctx.run_udf(dataset=ds, udf=udf)['intensity'].data
    n_first = Counter.calls
    assert n_first >= 1
    b = ctx.run_udf(dataset=ds, udf=udf)['intensity'].data
    assert Counter.calls == n_first        # HBM image reused across tasks and runs
    assert np.array_equal(a, b)
    for i in range(6):                      # evict
        mi = rng.random((1, 32, 32)).astype(np.float32)
        ctx.run_udf(dataset=ds, udf=um.ApplyMasksUDF(mask_factories=lambda mi=mi: mi))
    c = ctx.run_udf(dataset=ds, udf=udf)['intensity'].data
    assert np.array_equal(a, c)


def test_run_udf_loop_reuses_plan_and_follows_mutated_factories(ctx):
    """The same ApplyMasksUDF object run again re-uses plan, per-partition instances and device
    tiles; a factories LIST mutated in place is a new stack (the reference re-evaluates the
    factories every run, common/container.py:260-314); results stay caller-owned arrays."""
    from libertem_amd.udf.masks import ApplyMasksUDF
    rng = np.random.default_rng(31)
    data = rng.integers(0, 500, (6, 7, 32, 32)).astype(np.uint16)
    m = [rng.random((32, 32)).astype(np.float32) for _ in range(3)]
    factories = [lambda: m[0], lambda: m[1]]
    ds = _device_ds(ctx, data, 2)
    udf = ApplyMasksUDF(mask_factories=factories)
    keep = []
    for rep in range(4):
        r = ctx.run_udf(dataset=ds, udf=udf)['intensity'].data
        keep.append(r)
        assert r.shape == (6, 7, 2)
        assert _close(r, opath.apply_masks(data, np.stack(m[:2]), num_partitions=2), F32_TOL)
    assert all(np.array_equal(k, keep[0]) for k in keep)          # earlier results untouched
    assert len({k.ctypes.data for k in keep}) == 4                # ... and in their own memory
    plans = ds.__dict__['_udf_plans']
    assert len(plans) == 1 and next(iter(plans.values()))['tasks'][0]._keep.get('tiles')
    factories.append(lambda: m[2])
    r3 = ctx.run_udf(dataset=ds, udf=udf)['intensity'].data
    assert r3.shape == (6, 7, 3)
    assert _close(r3, opath.apply_masks(data, np.stack(m), num_partitions=2), F32_TOL)


def test_sparse_masks_through_udf(ctx):
    """use_sparse variants (reference tests/analysis/test_analysis_masks.py:278-446)."""
    import scipy.sparse as sp
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd import masks as pm
    rng = np.random.default_rng(40)
    data = rng.integers(0, 1000, (4, 9, 64, 64)).astype(np.uint16)
    dense = [np.where(rng.random((64, 64)) < 0.05, rng.random((64, 64)), 0).astype(np.float32)
             for _ in range(5)]
    ref = opath.apply_masks(data, np.stack(dense), num_partitions=2)
    ref_sp = opath.apply_masks_sparse(
        data, sp.csr_matrix(np.stack(dense).reshape((5, -1))), num_partitions=2)
    assert _close(ref_sp, ref, F32_TOL)
    ds = ctx.load('memory', data=data, num_partitions=2, sig_dims=2)
    facs = [(lambda i=i: sp.csr_matrix(dense[i])) for i in range(5)]
    for kw in (dict(), dict(use_sparse=True), dict(use_sparse='scipy.sparse'),
               dict(use_sparse='scipy.sparse.csc'), dict(use_sparse='sparse.pydata'),
               dict(use_sparse=False)):
        got = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=facs, **kw))
        assert _close(got['intensity'].data, ref, F32_TOL), kw
    # dense factories forced sparse
    got = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(
        mask_factories=lambda: np.stack(dense), use_sparse=True))
    assert _close(got['intensity'].data, ref, F32_TOL)
    # sparse ring stack from the product factory, sub-frame tiles forced
    rings = pm.radial_bins(32, 32, 64, 64, n_bins=40, use_sparse=True, dtype=np.float32)
    ref_r = opath.apply_masks(data, rings.todense(), num_partitions=2)
    ds_t = ctx.load('memory', data=data, num_partitions=2, sig_dims=2, tileshape=(7, 16, 64))
    got = ctx.run_udf(dataset=ds_t, udf=ApplyMasksUDF(mask_factories=lambda: rings))
    assert _close(got['intensity'].data, ref_r, F32_TOL)
    # float64 / uint32 data -> float64 result: the sparse gather kernel in double (not densified)
    from libertem_amd import hip
    ref64 = np.tensordot(data.astype(np.float64), np.stack(dense).astype(np.float64),
                         axes=([2, 3], [1, 2]))
    for wide in (np.float64, np.uint32, np.int32):
        ds_w = ctx.load('memory', data=data.astype(wide), num_partitions=2, sig_dims=2)
        hip.KernelTimer.start()
        got = ctx.run_udf(dataset=ds_w, udf=ApplyMasksUDF(mask_factories=facs))
        kernels = {k.split(' ')[0] for _, _, k in hip.KernelTimer.stop()}
        assert kernels and all('k_sell_apply' in k and 'f64' in k for k in kernels), kernels
        assert got['intensity'].data.dtype == np.float64
        assert _close(got['intensity'].data, ref64, 1e-12)


def _spots_frame(sig, centre, offsets, radius=3.0):
    """disks of equal intensity at centre + offsets"""
    yy, xx = np.mgrid[0:sig[0], 0:sig[1]]
    f = np.zeros(sig, dtype=np.float32)
    for dy, dx in offsets:
        f += ((yy - centre[0] - dy) ** 2 + (xx - centre[1] - dx) ** 2 <= radius ** 2)
    return f


@pytest.mark.parametrize('use_sparse', [False, True])
def test_radial_fourier_symmetry_selection_rules(ctx, use_sparse):
    """The selection rules the reference checks in tests/analysis/test_analysis_radialfourier.py:
    76-188, on own frames: one spot (all orders, phase follows the azimuth), two opposite spots (odd
    orders vanish), four spots at 90 degrees (only multiples of 4 survive); nothing in the inner bin;
    order 0 of the outer bin is the summed intensity."""
    sig, c, d = (64, 64), (32, 32), 12
    frames = np.stack([
        _spots_frame(sig, c, [(0, d)]),                                   # +x
        _spots_frame(sig, c, [(0, -d)]),                                  # -x
        _spots_frame(sig, c, [(0, d), (0, -d)]),                          # 2-fold
        _spots_frame(sig, c, [(0, d), (0, -d), (d, 0), (-d, 0)]),         # 4-fold
    ]).reshape((2, 2) + sig)
    ds = ctx.load('memory', data=frames, sig_dims=2, num_partitions=2)
    res = ctx.run(ctx.create_radial_fourier_analysis(
        dataset=ds, cy=c[0], cx=c[1], ri=0, ro=d + 4, n_bins=2, max_order=8,
        use_sparse=use_sparse))
    total = frames.sum(axis=(2, 3))

    def ch(b, o):
        return getattr(res, f'complex_{b}_{o}').raw_data

    tol = 2e-5 * total.max()
    for o in range(9):
        assert np.all(np.abs(ch(0, o)) <= tol), o                # inner bin is empty
    assert np.allclose(np.abs(ch(1, 0)), total, rtol=1e-5)
    for o in (1, 3, 5, 7):                                        # odd orders: 2-fold kills them
        assert np.all(np.abs(ch(1, o))[1] <= tol), o
        assert np.all(np.abs(ch(1, o))[0] > 0.1 * total[0]), o
    for o in (2, 6):                                              # 4-fold kills 2 and 6
        assert abs(ch(1, o)[1, 1]) <= tol
        assert abs(ch(1, o)[1, 0]) > 0.1 * total[1, 0]
    for o in (4, 8):                                              # everything passes, in phase
        assert np.all(np.abs(ch(1, o)) > 0.1 * total)
        assert np.allclose(np.angle(ch(1, o)), 0, atol=1e-4)
    # a single spot at azimuth 0 / pi: order 1 has phase 0 / pi (the factor is exp(i * o * phi))
    assert abs(np.angle(ch(1, 1)[0, 0])) < 1e-4
    assert abs(abs(np.angle(ch(1, 1)[0, 1])) - np.pi) < 1e-4
    assert res.dominant_0.raw_data.shape == (2, 2)


def test_sparse_stack_densified_when_mostly_filled(ctx):
    """HIP backend: a 'sparse' stack with filled 16-column groups runs on the dense matrix-core kernel,
    a localised ring stack stays sparse (blocked image); results agree with the oracle either way."""
    from libertem_amd.common.container import MaskContainer
    from libertem_amd.common.slice import Slice
    from libertem_amd.common.shape import Shape
    from libertem_amd.udf.base import UDF
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd import masks as pm
    rng = np.random.default_rng(41)
    data = rng.integers(0, 1000, (3, 8, 64, 64)).astype(np.uint16)
    filled = pm.radial_bins(32, 32, 64, 64, n_bins=2, use_sparse=True, dtype=np.float32)     # 2 wide rings
    wide = pm.radial_bins(32, 32, 64, 64, n_bins=64, use_sparse=True, dtype=np.float32)      # <= 64 columns over most pixels
    rings = pm.radial_bins(32, 32, 64, 64, n_bins=192, use_sparse=True, dtype=np.float32)
    kinds = {}
    for name, stack in (('filled', filled), ('wide', wide), ('rings', rings)):
        mc = MaskContainer(lambda stack=stack: stack, dtype=np.float32, use_sparse=True,
                           backend=UDF.BACKEND_HIP)
        full = Slice(origin=(0, 0, 0), shape=Shape((1, 64, 64), sig_dims=2))
        h = mc.get_handle_for_sig_slice(full.discard_nav(), np.float32, 0)
        kinds[name] = h.kind()
        mc.close()
    assert kinds == {'filled': 0, 'wide': 0, 'rings': 2}
    ds = ctx.load('memory', data=data, num_partitions=2, sig_dims=2)
    for stack in (filled, rings):
        got = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=lambda stack=stack: stack))
        ref = opath.apply_masks(data, stack.todense(), num_partitions=2)
        assert _close(got['intensity'].data, ref, F32_TOL)


def test_radial_fourier_sparse(ctx):
    """RadialFourierAnalysis with the sparse complex64 stack == the (pinned) dense one."""
    case = recipes.RF_CASES[0]
    data = recipes.make_rf_case(case)
    ds = ctx.load('memory', data=data, num_partitions=2, sig_dims=2)
    dense = ctx.run(ctx.create_radial_fourier_analysis(dataset=ds, n_bins=2, max_order=4,
                                                       use_sparse=False))
    sparse = ctx.run(ctx.create_radial_fourier_analysis(dataset=ds, n_bins=2, max_order=4,
                                                        use_sparse=True))
    ref = opath.radial_fourier_analysis(data, num_partitions=2, n_bins=2, max_order=4,
                                        use_sparse='scipy.sparse')
    assert _close(sparse.raw_results, dense.raw_results, F32_TOL)
    assert _close(sparse.raw_results, ref['raw_results'], F32_TOL)


# --- mirrors of the reference's own mask tests (tests/analysis/test_analysis_masks.py) -----------
def _naive_mask_apply(masks, data):
    """per-frame dot products in float64/complex128 (idea of tests/utils.py:25-45)"""
    out_dtype = np.result_type(*[m.dtype for m in masks], data.dtype, np.float64)
    res = np.zeros((len(masks),) + tuple(data.shape[:2]), dtype=out_dtype)
    for n, m in enumerate(masks):
        res[n] = np.tensordot(data.astype(out_dtype), m.astype(out_dtype), axes=([2, 3], [0, 1]))
    return res


def _mk_random(size, dtype='float32', seed=0):
    """0/1 valued data with two large outliers (tests/utils.py:48-78)"""
    rng = np.random.default_rng(seed)
    dtype = np.dtype(dtype)
    data = (rng.random(size) > 0.5).astype(dtype)
    flat = data.reshape(-1)
    if dtype.kind in 'iu':
        flat[rng.integers(0, flat.size, 2)] = min(np.iinfo(dtype).max, 2**15)
    else:
        flat[rng.integers(0, flat.size, 2)] = 2.0**20
    return data


@pytest.mark.parametrize('data_dtype,mask_dtype', [
    ('<u2', 'uint16'),        # test_mask_uint  (:151-173)
    ('>u2', 'float32'),       # test_endian     (:176-193)
    ('<i4', 'float32'),       # test_signed     (:196-214)
    ('>f4', 'float32'),
])
def test_reference_dtype_cases(ctx, data_dtype, mask_dtype):
    rng = np.random.default_rng(3)
    if np.dtype(data_dtype).kind == 'f':
        data = rng.random((16, 16, 16, 16)).astype(data_dtype)
    else:
        data = rng.choice(a=0xFFFF, size=(16, 16, 16, 16)).astype(data_dtype)
    mask = _mk_random((16, 16), seed=1).astype(mask_dtype)
    expected = _naive_mask_apply([mask], data)
    ds = ctx.load('memory', data=data, tileshape=(4 * 4, 4, 4), num_partitions=2, sig_dims=2)
    analysis = ctx.create_mask_analysis(dataset=ds, factories=[lambda: mask])
    res = ctx.run(analysis)
    assert res.mask_0.raw_data.shape == (16, 16)
    assert np.allclose(res.mask_0.raw_data, expected[0], rtol=1e-5)
    # sparse flavour of the same program (reference _run_mask_test_program, do_sparse=True)
    import scipy.sparse as sp
    if np.dtype(mask_dtype).kind == 'f':
        analysis = ctx.create_mask_analysis(dataset=ds, factories=[lambda: sp.csr_matrix(mask)],
                                            use_sparse=True)
        res = ctx.run(analysis)
        assert np.allclose(res.mask_0.raw_data, expected[0], rtol=1e-5)


def test_numerics_succeed(ctx):
    """float64 mask_dtype on the highest expected resolution / dynamic range (:948-978)"""
    RESOLUTION, RANGE, VAL = 4096, 1e6, 1.1
    data = np.full((2, 1, RESOLUTION, RESOLUTION), VAL, dtype=np.float32)
    data[0, 0, 0, 0] += VAL * RANGE
    ds = ctx.load('memory', data=data, tileshape=(2, RESOLUTION, RESOLUTION), num_partitions=1,
                  sig_dims=2)
    mask0 = np.ones((RESOLUTION, RESOLUTION), dtype=np.float32)
    analysis = ctx.create_mask_analysis(dataset=ds, factories=[lambda: mask0], mask_count=1,
                                        mask_dtype='float64')
    results = ctx.run(analysis)
    expected = np.array([[[VAL * RESOLUTION**2 + VAL * RANGE], [VAL * RESOLUTION**2]]])
    assert results.mask_0.raw_data.dtype == np.float64
    assert np.allclose(expected[0], results.mask_0.raw_data)


def test_numerics_float32_vs_exact(ctx):
    """float32 masks on a million equal-sign terms per frame: the reference documents that its
    own float32 result is NOT accurate here (test_numerics_fail :909-945: BLAS accumulates
    ~1e-3 relative error).  Parity with the reference to 1e-5 is therefore only meaningful where
    the reference itself is that accurate; on this input the HIP path (short MFMA chains,
    two accumulators per wave, K split) must be at least as close to the exact value as the
    reference's CPU path."""
    RESOLUTION, RANGE, VAL = 1024, 1e6, 1.1
    data = np.full((2, 1, RESOLUTION, RESOLUTION), VAL, dtype=np.float32)
    data[0, 0, 0, 0] += VAL * RANGE
    mask0 = np.ones((RESOLUTION, RESOLUTION), dtype=np.float64)
    ds = ctx.load('memory', data=data, num_partitions=1, sig_dims=2)
    analysis = ctx.create_mask_analysis(dataset=ds, factories=[lambda: mask0], mask_count=1,
                                        mask_dtype='float32')
    got = ctx.run(analysis).mask_0.raw_data
    ref = opath.apply_masks(data, mask0[np.newaxis], mask_dtype=np.float32)[..., 0]
    exact = np.array([[np.float64(np.float32(VAL)) * RESOLUTION**2 + VAL * RANGE],
                      [np.float64(np.float32(VAL)) * RESOLUTION**2]])
    assert got.dtype == np.float32
    err_hip = np.abs(got - exact) / exact
    err_ref = np.abs(ref - exact) / exact
    assert np.all(err_hip <= 1e-4)
    assert np.all(err_hip <= err_ref + 1e-6)


@pytest.mark.parametrize('case', recipes.SHIFT_CASES, ids=lambda c: c['name'])
def test_shifted_masks_vs_reference_golden(ctx, golden_dir, case):
    """ApplyMasksUDF(shifts=...): constant and per-frame (aux data) shifts"""
    from libertem_amd.udf.masks import ApplyMasksUDF
    g = _load(golden_dir, 'shifts')
    data, masks, shifts = recipes.make_shift_case(case)
    if case['shifts'] == 'aux':
        sh = ApplyMasksUDF.aux_data(shifts.reshape((-1, 2)).ravel(), kind='nav',
                                    extra_shape=(2,), dtype=shifts.dtype)
    else:
        sh = tuple(int(x) for x in shifts)
    ref = g[case['name']]
    for ds in (ctx.load('memory', data=data, num_partitions=case['num_partitions'], sig_dims=2),
               _device_ds(ctx, data, case['num_partitions'])):
        got = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=lambda: masks, shifts=sh))
        d = got['intensity'].data
        assert d.shape == ref.shape and d.dtype == ref.dtype
        assert np.allclose(d, ref, rtol=F32_TOL, atol=F32_TOL * max(np.abs(ref).max(), 1e-30))
    assert np.allclose(opath.apply_masks_shifted(data, masks, shifts), ref, rtol=1e-5, atol=1e-3)
    # ROI + per-frame shifts: aux data follows the ROI
    if case['shifts'] == 'aux':
        roi = np.zeros(case['nav'], dtype=bool)
        roi[1, 1:4] = True
        roi[3, 0] = True
        ds = ctx.load('memory', data=data, num_partitions=2, sig_dims=2)
        got = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=lambda: masks, shifts=sh),
                          roi=roi)['intensity'].data
        assert np.allclose(got[roi], ref[roi], rtol=F32_TOL, atol=F32_TOL * np.abs(ref).max())
        assert np.all(np.isnan(got[~roi]))


# --- detector corrections on the device --------------------------------------
@pytest.mark.parametrize('resident', ['host', 'device'])
@pytest.mark.parametrize('case', recipes.CORR_CASES, ids=lambda c: c['name'])
def test_corrections_vs_reference_golden(ctx, golden_dir, case, resident):
    """run_udf(corrections=CorrectionSet(dark, gain, excluded_pixels)) for the native UDFs:
    ltmi_correct + ltmi_repair_pixels feed the same kernels; results vs the reference's."""
    from libertem_amd.io.corrections import CorrectionSet
    from libertem_amd.io.corrections.corrset import ExcludedPixels
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd.udf.sum import SumUDF
    from libertem_amd.udf.sumsigudf import SumSigUDF
    g = _load(golden_dir, 'corrections')
    data, dark, gain, excluded, masks = recipes.make_corr_case(case)
    sig = tuple(case['sig'])
    excl = None if excluded is None else ExcludedPixels(excluded, sig)
    corr = CorrectionSet(dark=dark, gain=gain, excluded_pixels=excl)
    if resident == 'device':
        ds = _device_ds(ctx, data, case['num_partitions'])
    else:
        ds = ctx.load('memory', data=data, num_partitions=case['num_partitions'], sig_dims=2)
    import libertem_amd.udf.masks as um
    from libertem_amd import hip
    udfs = {'sum': SumUDF(), 'sumsig': SumSigUDF(),
            'masks': ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False)}
    for name, udf in udfs.items():
        for fold in (True, False):
            um.FOLD_CORRECTIONS = fold
            hip.KernelTimer.start()
            try:
                got = ctx.run_udf(dataset=ds, udf=udf, corrections=corr)['intensity'].data
            finally:
                um.FOLD_CORRECTIONS = True
                launches = hip.KernelTimer.stop()
            ref = g[f"{case['name']}__{name}"]
            assert got.shape == ref.shape and got.dtype == ref.dtype, (name, got.dtype, ref.dtype)
            assert _close(got, ref, F32_TOL), (name, fold, np.abs(got - ref).max(),
                                               np.abs(ref).max())
            if name == 'masks':
                assert len(launches) >= 1
    # without corrections the same dataset still gives the uncorrected result (nothing was
    # modified in place, no stale scratch)
    plain = ctx.run_udf(dataset=ds, udf=SumSigUDF())['intensity'].data
    assert np.allclose(plain.reshape(-1), data.reshape((-1, prod_sig(sig))).sum(axis=1),
                       rtol=1e-5)


def prod_sig(sig):
    return int(np.prod(sig))


def test_corrections_chunked_scratch(ctx):
    """More frames than fit the correction scratch buffer: several chunks, ragged last one."""
    from libertem_amd.io.corrections import CorrectionSet
    from libertem_amd.io.dataset.base import Negotiator
    from libertem_amd.udf.sumsigudf import SumSigUDF
    from oracle import corrections as oc
    rng = np.random.default_rng(77)
    data = rng.integers(0, 1000, (37, 32, 32)).astype(np.uint16)
    dark = rng.random((32, 32)) * 5
    gain = rng.random((32, 32)) + 0.5
    bad = np.zeros((32, 32), dtype=bool)
    bad[3, 4] = bad[31, 31] = bad[10, 10] = bad[10, 11] = True
    corr = CorrectionSet(dark=dark, gain=gain, excluded_pixels=bad)
    old = Negotiator.HIP_CORRECTED_CHUNK
    Negotiator.HIP_CORRECTED_CHUNK = 10 * 32 * 32 * 4          # 10 frames per chunk
    try:
        ds = _device_ds(ctx, data.reshape((37, 32, 32)), 1)
        got = ctx.run_udf(dataset=ds, udf=SumSigUDF(), corrections=corr)['intensity'].data
    finally:
        Negotiator.HIP_CORRECTED_CHUNK = old
    coords = [tuple(c) for c in np.argwhere(bad)]
    ref = oc.correct(data, (32, 32), dark=dark, gain=gain, coords=coords).astype(np.float64)
    assert np.allclose(got, ref.reshape((37, -1)).sum(axis=1), rtol=1e-5)


# --- Fourier-space operators -------------------------------------------------
@pytest.mark.parametrize('resident', ['host', 'device'])
@pytest.mark.parametrize('case', recipes.CRYST_CASES, ids=lambda c: c['name'])
def test_crystallinity_vs_reference_golden(ctx, golden_dir, case, resident):
    from libertem_amd.udf.crystallinity import CrystallinityUDF, run_analysis_crystall
    g = _load(golden_dir, 'crystallinity')
    data = recipes.make_cryst_case(case)
    if resident == 'device':
        ds = _device_ds(ctx, data, case['num_partitions'])
    else:
        ds = ctx.load('memory', data=data, num_partitions=case['num_partitions'], sig_dims=2)
    udf = CrystallinityUDF(rad_in=case['rad_in'], rad_out=case['rad_out'],
                           real_center=case['real_center'], real_rad=case['real_rad'])
    got = ctx.run_udf(dataset=ds, udf=udf)['intensity'].data
    ref = g[case['name']]
    assert got.shape == ref.shape and got.dtype == ref.dtype
    assert _close(got, ref, F32_TOL), (np.abs(got - ref).max(), np.abs(ref).max())
    if tuple(case['sig']) == (128, 128) or all(e in (256, 512, 1024) for e in case['sig']):
        # these shapes run the hand-written transform kernels (csrc/ltmi_cryst.hip), not hipFFT
        import libertem_amd.udf.crystallinity as cr
        labels = [p.last_kernel() for k, p in cr._PLANS.items() if k[1:3] == tuple(case['sig'])]
        assert labels and all(lb.startswith('k_cryst_') for lb in labels), labels
    again = run_analysis_crystall(ctx, ds, case['rad_in'], case['rad_out'], case['real_center'],
                                  case['real_rad'])['intensity'].data
    assert np.array_equal(again, got)                       # deterministic, plan re-used


def test_crystallinity_with_corrections_fused(ctx):
    """CrystallinityUDF under CorrectionSet: dark / gain / dead-pixel patches fused into the
    transform's conversion pass (raw tiles) == the generic route (corrected copy of the frames) ==
    the oracle on oracle-corrected frames."""
    import libertem_amd.udf.masks as um
    from libertem_amd.io.corrections import CorrectionSet
    from libertem_amd.udf.crystallinity import CrystallinityUDF
    from oracle import corrections as oc
    rng = np.random.default_rng(51)
    data = rng.integers(0, 3000, (4, 6, 32, 48)).astype(np.uint16)
    dark = rng.random((32, 48)) * 6
    gain = rng.random((32, 48)) * 0.6 + 0.7
    bad = np.zeros((32, 48), dtype=bool)
    bad[16, 24] = bad[0, 0] = bad[10, 40] = bad[10, 41] = bad[31, 47] = True
    coords = [tuple(c) for c in np.argwhere(bad)]
    for kw in (dict(dark=dark, gain=gain, excluded_pixels=bad), dict(gain=gain),
               dict(excluded_pixels=bad)):
        corr = CorrectionSet(**kw)
        corrected = oc.correct(data, (32, 48), dark=kw.get('dark'), gain=kw.get('gain'),
                               coords=coords if 'excluded_pixels' in kw else None)
        for real in ((16, 24), None):
            ref = opath.crystallinity_udf(corrected, 3, 9, real, 4 if real else None)
            out = {}
            for fold in (True, False):
                um.FOLD_CORRECTIONS = fold
                try:
                    for resident in ('device', 'host'):
                        ds = _device_ds(ctx, data, 2) if resident == 'device' else \
                            ctx.load('memory', data=data, num_partitions=2, sig_dims=2)
                        udf = CrystallinityUDF(rad_in=3, rad_out=9, real_center=real,
                                               real_rad=4 if real else None)
                        got = ctx.run_udf(dataset=ds, udf=udf, corrections=corr)['intensity'].data
                        assert _close(got, ref, F32_TOL), (sorted(kw), real, fold, resident)
                        out[(fold, resident)] = got
                finally:
                    um.FOLD_CORRECTIONS = True
            assert _close(out[(True, "device")], out[(False, "device")], F32_TOL)


def test_crystallinity_roi_batches_and_analysis(ctx):
    """More frames than one FFT batch, a ragged last batch, an ROI, and the ApplyFFTMask /
    SumfftAnalysis wrappers (reference tests/udf/test_crystallinity.py, analysis/sumfft.py)."""
    import libertem_amd.udf.crystallinity as cr
    from libertem_amd.analysis import ApplyFFTMask, SumfftAnalysis
    rng = np.random.default_rng(9)
    data = rng.integers(0, 500, (5, 7, 32, 32)).astype(np.uint16)
    ref = opath.crystallinity_udf(data, 3, 9, (16, 16), 4)
    old = cr.FFT_WORKSPACE_BYTES
    cr.FFT_WORKSPACE_BYTES = 8 * (32 * 32 * 4 + 32 * 17 * 8)        # 8 frames per batch
    try:
        ds = _device_ds(ctx, data, 1)
        udf = cr.CrystallinityUDF(rad_in=3, rad_out=9, real_center=(16, 16), real_rad=4)
        got = ctx.run_udf(dataset=ds, udf=udf)['intensity'].data
        assert _close(got, ref, F32_TOL)
        roi = np.zeros((5, 7), dtype=bool)
        roi[1, 2:6] = True
        roi[4, 6] = True
        part = ctx.run_udf(dataset=ds, udf=udf, roi=roi)['intensity']
        assert _close(part.raw_data, ref[roi], F32_TOL)
        assert np.all(np.isnan(part.data[~roi]))
    finally:
        cr.FFT_WORKSPACE_BYTES = old
    an = ApplyFFTMask(dataset=ds, parameters=dict(rad_in=3, rad_out=9, real_centery=16,
                                                  real_centerx=16, real_rad=4))
    res = ctx.run(an)
    assert _close(res.intensity.raw_data, ref, F32_TOL)
    sf = ctx.run(SumfftAnalysis(dataset=ds, parameters=dict(real_centery=16, real_centerx=16,
                                                            real_rad=4)))
    total = data.astype(np.float64).sum(axis=(0, 1))
    assert np.allclose(sf.intensity.raw_data, total, rtol=1e-6)
    yy, xx = np.ogrid[-16:16, -16:16]
    mask = 1 - 1 * (yy * yy + xx * xx <= 16)
    expect = np.log(abs(np.fft.fftshift(np.fft.fft2(total * mask))) + 1)
    assert np.allclose(sf.intensity_fft.raw_data, expect, rtol=1e-4, atol=1e-4)


def test_com_with_corrections_folded_and_generic(ctx):
    """CoMUDF under CorrectionSet: the folded path (raw frames, corrected masks) and the generic
    path (corrected frames) agree with the oracle's CoM of oracle-corrected frames."""
    import libertem_amd.udf.masks as um
    from libertem_amd.io.corrections import CorrectionSet
    from libertem_amd.udf.com import CoMUDF
    from oracle import corrections as oc
    rng = np.random.default_rng(31)
    data = rng.integers(0, 2000, (6, 7, 32, 32)).astype(np.uint16)
    yy, xx = np.mgrid[0:32, 0:32]
    data = (data * np.exp(-((yy - 15) ** 2 + (xx - 17) ** 2) / 60.0)).astype(np.uint16)
    dark = rng.random((32, 32)) * 3
    gain = rng.random((32, 32)) * 0.4 + 0.8
    bad = np.zeros((32, 32), dtype=bool)
    bad[14, 16] = bad[3, 3] = bad[20, 21] = True
    corr = CorrectionSet(dark=dark, gain=gain, excluded_pixels=bad)
    coords = [tuple(c) for c in np.argwhere(bad)]
    corrected = oc.correct(data, (32, 32), dark=dark, gain=gain, coords=coords)
    ref = opath.com_udf(corrected, num_partitions=2, cy=15., cx=17., r=12.)
    ds = _device_ds(ctx, data, 2)
    udf = CoMUDF.with_params(cy=15., cx=17., r=12.)
    out = {}
    for fold in (True, False):
        um.FOLD_CORRECTIONS = fold
        try:
            res = ctx.run_udf(dataset=ds, udf=udf, corrections=corr)
        finally:
            um.FOLD_CORRECTIONS = True
        out[fold] = res
        # shifts = com - centre: small differences of numbers ~ the centre coordinates, so the
        # float32 tolerance is relative to |raw_com|, as for the reference's own float32 path
        atol = 2e-5 * np.abs(ref['raw_com']).max()
        for name in ('raw_com', 'raw_shifts', 'field', 'magnitude'):
            got = res[name].data
            assert np.allclose(got, ref[name], rtol=2e-5, atol=atol), \
                (fold, name, np.abs(got - ref[name]).max())
    assert np.allclose(out[True]['field'].data, out[False]['field'].data, rtol=2e-5,
                       atol=2e-5 * np.abs(ref['raw_com']).max())


def test_sparse_masks_with_corrections_folded(ctx):
    """Sparse stacks take the folded route too: RAW frames, (masks . R) . diag(gain) as the sparse
    stack and the dark constant subtracted -- equal to applying the stack to oracle-corrected frames,
    and to the generic route (corrected copy of the frames)."""
    import libertem_amd.udf.masks as um
    from libertem_amd.io.corrections import CorrectionSet
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd import masks as M, hip
    from oracle import corrections as oc
    rng = np.random.default_rng(41)
    data = rng.integers(0, 3000, (5, 8, 64, 64)).astype(np.uint16)
    dark = rng.random((64, 64)) * 4
    gain = rng.random((64, 64)) * 0.5 + 0.75
    bad = np.zeros((64, 64), dtype=bool)
    bad[30, 33] = bad[0, 0] = bad[40, 12] = bad[40, 13] = True
    corr = CorrectionSet(dark=dark, gain=gain, excluded_pixels=bad)
    coords = [tuple(c) for c in np.argwhere(bad)]
    corrected = oc.correct(data, (64, 64), dark=dark, gain=gain, coords=coords)

    def rings():
        return M.radial_bins(centerX=32, centerY=32, imageSizeX=64, imageSizeY=64, n_bins=160,
                             use_sparse=True, dtype=np.float32)
    dense = M.radial_bins(centerX=32, centerY=32, imageSizeX=64, imageSizeY=64, n_bins=160,
                          use_sparse=False, dtype=np.float32)
    ref = np.tensordot(corrected.astype(np.float64), dense.astype(np.float64),
                       axes=([2, 3], [1, 2]))
    ds = _device_ds(ctx, data, 2)
    udf = ApplyMasksUDF(mask_factories=rings, use_sparse='scipy.sparse', mask_count=160,
                        mask_dtype=np.float32)
    out = {}
    for fold in (True, False):
        um.FOLD_CORRECTIONS = fold
        try:
            hip.KernelTimer.start()
            out[fold] = ctx.run_udf(dataset=ds, udf=udf, corrections=corr)['intensity'].data
            kernels = [k for _, _, k in hip.KernelTimer.stop()]
        finally:
            um.FOLD_CORRECTIONS = True
        assert all('k_bell' in k or 'k_sell' in k or 'k_scatter' in k for k in kernels) and kernels, kernels
        assert _close(out[fold], ref, F32_TOL), fold


def test_run_udf_iter_partial_results_on_device(ctx):
    """run_udf_iter on the HIP executor: one partial result per merged partition, damage grows,
    the valid rows of every partial result are already final (reference api.py:1053-1152)."""
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd.udf.sum import SumUDF
    rng = np.random.default_rng(3)
    data = rng.integers(0, 500, (8, 6, 32, 32)).astype(np.uint16)
    masks = rng.random((3, 32, 32)).astype(np.float32)
    ds = _device_ds(ctx, data, 4)
    udfs = [ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False), SumUDF()]
    final = ctx.run_udf(dataset=ds, udf=udfs)
    seen = []
    for part in ctx.run_udf_iter(dataset=ds, udf=udfs):
        dmg = np.array(part.damage.data)
        seen.append(int(dmg.sum()))
        inten = part.buffers[0]['intensity'].data
        assert np.array_equal(inten[dmg], final[0]['intensity'].data[dmg])
        assert np.all(inten[~dmg] == 0)
    assert seen == [12, 24, 36, 48]
    assert np.allclose(part.buffers[1]['intensity'].data, final[1]['intensity'].data, rtol=1e-6)


# --- reference test cases re-expressed (tests/analysis/test_analysis_masks.py:818-907, 1181-1257) ------
def _shift_naive(masks, data, shifts):
    """Known-answer construction of the reference's tests (`naive_shifted_mask_apply`, re-derived):
    intersect frame and shifted mask explicitly; shifts are truncated to int."""
    n, h, w = data.shape
    shifts = np.asarray(shifts)
    if shifts.shape == (2,):
        shifts = np.repeat(shifts[None], n, axis=0)
    out = np.zeros((n, len(masks)))
    for f, (dy, dx) in enumerate(shifts.astype(int)):
        fy0, fy1 = max(0, dy), min(h, h + dy)
        fx0, fx1 = max(0, dx), min(w, w + dx)
        if fy0 >= fy1 or fx0 >= fx1:
            continue
        for k, mk in enumerate(masks):
            out[f, k] = (mk[fy0 - dy:fy1 - dy, fx0 - dx:fx1 - dx] * data[f, fy0:fy1, fx0:fx1]).sum()
    return out


def test_masks_on_1d_3d_signals_and_1d_scans(ctx):
    """time series of 2D frames, spectra (sig_dims=1, line scan and 2D scan), hyperspectral
    (sig_dims=3): result shapes and values."""
    rng = np.random.default_rng(1)
    # (data shape, sig_dims, expected nav shape, forced tileshape of the reference test)
    cases = [((256, 16, 16), 2, (256,), (2, 16, 16)),
             ((256, 256), 1, (256,), (2, 256)),
             ((16, 16, 256), 1, (16, 16), (2, 256)),
             ((6, 5, 8, 16, 16), 3, (6, 5), (1, 8, 16, 16))]
    for shape, sig_dims, nav, tileshape in cases:
        data = rng.integers(0, 1000, shape).astype('<u2')
        mask0 = rng.random(shape[len(shape) - sig_dims:])
        for ts in (None, tileshape):
            ds = ctx.load('memory', data=data, tileshape=ts, num_partitions=2, sig_dims=sig_dims)
            an = ctx.create_mask_analysis(dataset=ds, factories=[lambda: mask0])
            res = ctx.run(an)
            assert res.mask_0.raw_data.shape == nav
            ref = np.tensordot(data.astype(np.float64), mask0,
                               axes=(list(range(len(nav), len(shape))), list(range(sig_dims))))
            assert _close(res.mask_0.raw_data, ref, F32_TOL)


def test_masks_complex_dataset_and_masks(ctx):
    rng = np.random.default_rng(2)
    data = (rng.random((16, 16, 16, 16)) + 1j * rng.random((16, 16, 16, 16))).astype(np.complex64)
    ds = ctx.load('memory', data=data, num_partitions=2, sig_dims=2)
    mask_r = rng.random((16, 16))
    res = ctx.run(ctx.create_mask_analysis(dataset=ds, factories=[lambda: mask_r]))
    assert res.mask_0_complex.raw_data.shape == (16, 16)
    ref = np.tensordot(data.astype(np.complex128), mask_r, axes=([2, 3], [0, 1]))
    assert _close(res.mask_0_complex.raw_data, ref, F32_TOL)
    mask_c = (rng.random((16, 16)) + 1j * rng.random((16, 16))).astype(np.complex64)
    res = ctx.run(ctx.create_mask_analysis(dataset=ds, factories=[lambda: mask_c]))
    ref = np.tensordot(data.astype(np.complex128), mask_c.astype(np.complex128),
                       axes=([2, 3], [0, 1]))
    assert _close(res.mask_0_complex.raw_data, ref, F32_TOL)
    assert _close(res.mask_0.raw_data, np.abs(ref), F32_TOL)


@pytest.mark.parametrize('dtype', ['complex64', 'complex128'])
def test_complex_frames_run_on_the_matrix_kernels(ctx, dtype):
    """complex datasets: the frame is read as 2 n_px real pixels against the real expansion of the
    stack ([mr, -mi] / [mi, mr] rows), so the dense matrix kernels do the work -- several
    partitions, 20 complex masks (40 real columns), accumulate over ROI-free tiles."""
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd import hip
    rng = np.random.default_rng(12)
    data = ((rng.random((6, 9, 24, 24)) - 0.4) + 1j * (rng.random((6, 9, 24, 24)) - 0.6)).astype(dtype)
    masks = ((rng.random((20, 24, 24)) - 0.5) + 1j * (rng.random((20, 24, 24)) - 0.5)).astype(dtype)
    ds = _device_ds(ctx, data, 3)
    hip.KernelTimer.start()
    res = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False,
                                                    mask_count=20, mask_dtype=np.dtype(dtype)))
    kernels = {k.split('<')[0] for _, _, k in hip.KernelTimer.stop()}
    assert kernels and all(k.startswith('k_dense_lds') for k in kernels), kernels
    got = res['intensity'].data
    assert got.dtype == np.dtype(dtype) and got.shape == (6, 9, 20)
    ref = np.tensordot(data.astype(np.complex128), masks.astype(np.complex128), axes=([2, 3], [1, 2]))
    assert _close(got, ref, F32_TOL if dtype == 'complex64' else 1e-12)
    # sparse masks on complex frames (the reference: rmatmul with a complex left operand)
    import scipy.sparse as sp
    sparse_masks = [sp.csr_matrix(np.where(rng.random((24, 24)) < 0.1, rng.random((24, 24)), 0).astype(
        np.float32 if dtype == 'complex64' else np.float64)) for _ in range(5)]
    res = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=[lambda m=m: m for m in sparse_masks],
                                                    use_sparse='scipy.sparse'))
    dense5 = np.stack([m.toarray() for m in sparse_masks]).astype(np.float64)
    ref = np.tensordot(data.astype(np.complex128), dense5, axes=([2, 3], [1, 2]))
    assert res['intensity'].data.dtype == np.dtype(dtype)
    assert _close(res['intensity'].data, ref, F32_TOL if dtype == 'complex64' else 1e-12)
    # real masks on complex frames
    mr = rng.random((3, 24, 24)).astype(np.float32 if dtype == 'complex64' else np.float64)
    res = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=lambda: mr, use_sparse=False))
    ref = np.tensordot(data.astype(np.complex128), mr.astype(np.float64), axes=([2, 3], [1, 2]))
    assert _close(res['intensity'].data, ref, F32_TOL if dtype == 'complex64' else 1e-12)


def test_shifted_masks_reference_cases(ctx):
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd.masks import circular
    rng = np.random.default_rng(5)
    # zero overlap (test_shifted_masks_zero_overlap)
    data = rng.random((2, 18, 12)).astype(np.float32)
    ds = ctx.load('memory', data=data, sig_dims=2)
    m1 = rng.random((18, 12))
    res = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=[lambda: m1], shifts=(-20, 15)))
    assert np.allclose(res['intensity'].data, 0.)
    # stacked masks, float shifts per frame, non-square frames (test_shifted_masks_stacked)
    shifts = rng.uniform(-5., 5., (2, 2))
    masks = rng.random((3, 18, 12))
    udf = ApplyMasksUDF(mask_factories=lambda: masks,
                        shifts=ApplyMasksUDF.aux_data(data=shifts.ravel(), kind='nav',
                                                      extra_shape=(2,), dtype=float))
    res = ctx.run_udf(dataset=ds, udf=udf)
    assert _close(res['intensity'].data, _shift_naive(masks, data.astype(np.float64), shifts),
                  F32_TOL)
    # descan error on a constant frame (test_shifted_masks_descan)
    h = w = 9
    frame = circular(4, 4, w, h, 1)
    frame_sum = frame.sum()
    sh = np.moveaxis(np.mgrid[-2:4:2, -2:4:2], 0, -1)
    frames = np.stack([np.roll(frame, (y, x), axis=(0, 1)) for y, x in sh.reshape(-1, 2)])
    ds9 = ctx.load('memory', data=frames.astype(np.uint8), sig_dims=2)
    mask = circular(4, 4, w, h, 2)
    plain = ctx.run_udf(dataset=ds9, udf=ApplyMasksUDF(mask_factories=[lambda: mask]))
    assert not (plain['intensity'].data == frame_sum).all()
    assert plain['intensity'].data.reshape(3, 3)[1, 1] == frame_sum
    fixed = ctx.run_udf(dataset=ds9, udf=ApplyMasksUDF(
        mask_factories=[lambda: mask],
        shifts=ApplyMasksUDF.aux_data(data=sh.ravel(), kind='nav', extra_shape=(2,), dtype=int)))
    assert (fixed['intensity'].data == frame_sum).all()


def test_raw_file_dataset_on_device(ctx, tmp_path):
    """Frames of a memory-mapped raw file streamed to the GPU in their native dtype."""
    from libertem_amd.udf.masks import ApplyMasksUDF
    rng = np.random.default_rng(0)
    data = rng.integers(0, 4000, (7, 9, 32, 32)).astype(np.uint16)
    path = str(tmp_path / "scan.raw")
    data.tofile(path)
    masks = rng.random((3, 32, 32)).astype(np.float32)
    ds = ctx.load('raw', path=path, dtype='uint16', nav_shape=(7, 9), sig_shape=(32, 32),
                  num_partitions=3)
    res = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=lambda: masks))
    ref = opath.apply_masks(data, masks, num_partitions=3)
    assert _close(res['intensity'].data, ref, F32_TOL)


@pytest.mark.parametrize('dtype', ['>u2', '>i2', '>u4'])
def test_big_endian_raw_file_decoded_on_device(ctx, tmp_path, dtype):
    """A raw file in the other byte order: the mapping is uploaded as it is and decoded on the GPU
    (ltmi_byteswap behind the H2D copy); results equal those of the same values in native order."""
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd.udf.sum import SumUDF
    rng = np.random.default_rng(21)
    dt = np.dtype(dtype)
    native = dt.newbyteorder('=')
    info = np.iinfo(native)
    vals = rng.integers(max(info.min, -3000), min(info.max, 3000), (6, 7, 32, 32)).astype(native)
    path = str(tmp_path / "scan_be.raw")
    vals.astype(dt).tofile(path)
    masks = rng.random((3, 32, 32)).astype(np.float32)
    ds = ctx.load('raw', path=path, dtype=dtype, nav_shape=(6, 7), sig_shape=(32, 32),
                  num_partitions=3)
    assert ds._swap_itemsize == dt.itemsize and ds.dtype == native
    res = ctx.run_udf(dataset=ds, udf=[ApplyMasksUDF(mask_factories=lambda: masks), SumUDF()])
    # signed integers in the other byte order, read into float32: the reference hands out the
    # UNSIGNED word (io/dataset/base/decode.py:15-66, tests/golden/decode_signed.npz) -- so does the
    # device decode by default (the swapped pixels are read as the unsigned twin)
    as_read = vals.view(np.dtype(f'u{dt.itemsize}')) if dt.kind == 'i' else vals
    ds_n = ctx.load('memory', data=as_read, num_partitions=3, sig_dims=2)
    ref = ctx.run_udf(dataset=ds_n, udf=[ApplyMasksUDF(mask_factories=lambda: masks), SumUDF()])
    assert np.array_equal(res[0]['intensity'].data, ref[0]['intensity'].data)
    assert np.array_equal(res[1]['intensity'].data, ref[1]['intensity'].data)
    assert _close(res[0]['intensity'].data, opath.apply_masks(as_read, masks, num_partitions=3),
                  F32_TOL if native.itemsize < 4 else 1e-6)
    # ROI: frames gathered on the host (bounce buffers), still decoded on the device
    roi = rng.random((6, 7)) < 0.5
    part = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=lambda: masks), roi=roi)
    ref_p = ctx.run_udf(dataset=ds_n, udf=ApplyMasksUDF(mask_factories=lambda: masks), roi=roi)
    assert np.array_equal(part['intensity'].raw_data, ref_p['intensity'].raw_data)
    if dt.kind == 'i':
        ds.signed_other_order = 'signed'            # the arithmetic reading on request
        res_s = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=lambda: masks))
        ref_s = ctx.run_udf(dataset=ctx.load('memory', data=vals, num_partitions=3, sig_dims=2),
                            udf=ApplyMasksUDF(mask_factories=lambda: masks))
        assert np.array_equal(res_s['intensity'].data, ref_s['intensity'].data)


@pytest.mark.parametrize('resident', ['host', 'device'])
@pytest.mark.parametrize('case', recipes.PICK_CASES, ids=lambda c: c['name'])
def test_pick_udf_and_analyses_on_device(ctx, golden_dir, case, resident):
    """PickUDF on the HIP worker (rows copied inside HBM) and the pick analyses == the reference."""
    from libertem_amd.udf.raw import PickUDF
    from libertem_amd.analysis.raw import PickFrameAnalysis, PickFFTFrameAnalysis
    g = np.load(os.path.join(golden_dir, 'pick.npz'))
    data = recipes.make_pick_case(case)
    if resident == 'device':
        if data.dtype.kind == 'c':
            pytest.skip("device-resident datasets hold real pixel types")
        ds = _device_ds(ctx, data, case['num_partitions'], sig_dims=len(case['sig']))
    else:
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
        rs = ctx.run(cls(dataset=ds, parameters=params))
        ref = g[f"{case['name']}__{tag}"]
        got = rs.intensity_complex.raw_data if ref.dtype.kind == 'c' else rs.intensity.raw_data
        assert got.dtype == ref.dtype
        assert np.allclose(got, ref, rtol=1e-6, atol=1e-6 * np.abs(ref).max())


def test_stream_dataset_on_device(ctx):
    """Row f4: frames arriving from an iterator are uploaded chunk by chunk as they land and go
    through the HIP kernels; partial results after every partition, final result == MemoryDataSet."""
    import time
    from libertem_amd.udf.masks import ApplyMasksUDF
    rng = np.random.default_rng(8)
    data = rng.integers(0, 4000, (8, 16, 32, 32)).astype(np.uint16)
    flat = data.reshape((-1, 32, 32))
    masks = rng.random((3, 32, 32)).astype(np.float32)

    def feed():
        for i in range(0, len(flat), 16):
            time.sleep(0.005)
            yield flat[i:i + 16]

    ds = ctx.load('stream', frames=feed(), nav_shape=(8, 16), sig_shape=(32, 32), dtype=np.uint16,
                  num_partitions=4)
    counts = []
    for part in ctx.run_udf_iter(dataset=ds, udf=ApplyMasksUDF(mask_factories=lambda: masks)):
        res = part.buffers[0]['intensity'].data.reshape((128, 3))
        counts.append(int(np.count_nonzero(res[:, 0])))
    assert counts == [32, 64, 96, 128]
    ref = ctx.run_udf(dataset=ctx.load('memory', data=data, num_partitions=4, sig_dims=2),
                      udf=ApplyMasksUDF(mask_factories=lambda: masks))
    assert np.array_equal(res, ref['intensity'].data.reshape((128, 3)))
    assert _close(res, opath.apply_masks(data, masks, num_partitions=4).reshape((128, 3)), F32_TOL)


def test_shifted_masks_with_roi_and_partitions(ctx):
    """per-frame shifts (aux data) are re-sliced per partition and compressed by the ROI"""
    from libertem_amd.udf.masks import ApplyMasksUDF
    rng = np.random.default_rng(77)
    nav, sig = (5, 6), (16, 32)
    data = rng.integers(0, 900, nav + sig).astype(np.uint16)
    masks = rng.random((4,) + sig).astype(np.float32)
    shifts = rng.integers(-5, 6, nav + (2,))
    roi = rng.random(nav) < 0.6
    roi[0, 0] = True
    ref_all = _shift_naive(masks.astype(np.float64), data.reshape((-1,) + sig).astype(np.float64),
                           shifts.reshape((-1, 2)))
    for resident in ('host', 'device'):
        ds = _device_ds(ctx, data, 3) if resident == 'device' else \
            ctx.load('memory', data=data, num_partitions=3, sig_dims=2)
        udf = ApplyMasksUDF(mask_factories=lambda: masks,
                            shifts=ApplyMasksUDF.aux_data(shifts.reshape((-1, 2)).ravel(),
                                                          kind='nav', extra_shape=(2,), dtype=int))
        full = ctx.run_udf(dataset=ds, udf=udf)['intensity'].data
        assert _close(full.reshape((-1, 4)), ref_all, F32_TOL)
        part = ctx.run_udf(dataset=ds, udf=udf, roi=roi)['intensity']
        assert _close(part.raw_data, ref_all[roi.reshape(-1)], F32_TOL)
        assert np.all(np.isnan(part.data[~roi]))


@pytest.mark.parametrize('direct_row_max', [0, 512])
def test_streamed_export_with_several_tiles(ctx, direct_row_max, monkeypatch):
    """Device-resident partitions split into tiles (pipelining policy) with the finished rows
    exported through the copy stream (direct_row_max = 0), or -- small write-once rows -- written by
    the kernels straight into the final page-locked host buffer, one launch per partition
    (direct_row_max = 512): dense and per-frame-sum results, ROI, 2 partitions."""
    from libertem_amd.io.dataset.base import Negotiator
    from libertem_amd.common import udf as udf_common
    monkeypatch.setattr(udf_common, 'HIP_DIRECT_ROW_MAX', direct_row_max)
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd.udf.sumsigudf import SumSigUDF
    from libertem_amd import hip
    rng = np.random.default_rng(21)
    data = rng.integers(0, 2000, (30, 25, 16, 16)).astype(np.uint16)        # 750 frames
    masks = rng.random((5, 16, 16)).astype(np.float32)
    old = (Negotiator.HIP_PIPELINE_MIN_FRAMES, Negotiator.HIP_PIPELINE_TILES)
    Negotiator.HIP_PIPELINE_MIN_FRAMES, Negotiator.HIP_PIPELINE_TILES = 64, 3
    Negotiator._hip_scheme_cache.clear()
    try:
        ds = _device_ds(ctx, data, 2)
        hip.KernelTimer.start()
        res = ctx.run_udf(dataset=ds, udf=[ApplyMasksUDF(mask_factories=lambda: masks),
                                           SumSigUDF()])
        launches = hip.KernelTimer.stop()
        if direct_row_max == 0:
            assert len(launches) >= 4, launches        # 2 partitions x >= 2 tiles
        else:
            assert len(launches) == 2, launches        # no D2H to overlap: one launch per partition
        ref = opath.apply_masks(data, masks, num_partitions=2)
        assert _close(res[0]['intensity'].data, ref, F32_TOL)
        assert np.array_equal(res[1]['intensity'].data,
                              data.reshape((30, 25, -1)).sum(axis=-1).astype(np.float32))
        roi = rng.random((30, 25)) < 0.5
        part = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=lambda: masks), roi=roi)
        assert _close(part['intensity'].raw_data, ref[roi], F32_TOL)
    finally:
        Negotiator.HIP_PIPELINE_MIN_FRAMES, Negotiator.HIP_PIPELINE_TILES = old
        Negotiator._hip_scheme_cache.clear()


def test_roi_runs_read_frames_through_a_row_list(ctx):
    """Device-resident data + ROI: the mask operators (ApplyMasksUDF, CoMUDF) read the selected frames
    in place through a row list (`ltmi_apply_masks_rows`, kernel label ',rows'), no gathered copy;
    UDFs that do not take row lists (SumSigUDF) get the gathered frames in the same run."""
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd.udf.sumsigudf import SumSigUDF
    from libertem_amd.udf.com import CoMUDF
    from libertem_amd import hip
    rng = np.random.default_rng(77)
    data = rng.integers(0, 3000, (13, 17, 32, 32)).astype(np.uint16)
    masks = rng.random((5, 32, 32)).astype(np.float32)
    roi = rng.random((13, 17)) < 0.4
    roi[0, :5] = True
    ref = opath.apply_masks(data, masks, num_partitions=3)
    ds = _device_ds(ctx, data, 3)
    hip.KernelTimer.start()
    res = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=lambda: masks), roi=roi)
    kernels = [k for _, _, k in hip.KernelTimer.stop()]
    assert kernels and all(',rows' in k for k in kernels), kernels
    assert _close(res['intensity'].raw_data, ref[roi], F32_TOL)
    assert np.all(np.isnan(res['intensity'].data[~roi]))
    # mixed run: SumSigUDF works on the gathered frames, the masks still through the row list
    hip.KernelTimer.start()
    r1, r2 = ctx.run_udf(dataset=ds, udf=[ApplyMasksUDF(mask_factories=lambda: masks), SumSigUDF()],
                         roi=roi)
    kernels = [k for _, _, k in hip.KernelTimer.stop()]
    assert any(',rows' in k for k in kernels), kernels
    assert _close(r1['intensity'].raw_data, ref[roi], F32_TOL)
    assert np.array_equal(r2['intensity'].raw_data, data.sum(axis=(2, 3))[roi].astype(np.float32))
    # CoM on the ROI (its 3-column stack: the VALU-only kernel, also through the row list)
    com = ctx.run_udf(dataset=ds, udf=CoMUDF.with_params(cy=16, cx=16), roi=roi)
    full = ctx.run_udf(dataset=ds, udf=CoMUDF.with_params(cy=16, cx=16))
    assert np.allclose(com['raw_com'].raw_data, full['raw_com'].data[roi], rtol=1e-6)
    # sparse ring masks: the blocked image's kernel through the row list as well
    from libertem_amd import masks as pm
    rings = pm.radial_bins(16, 16, 32, 32, n_bins=40, use_sparse=True, dtype=np.float32)
    hip.KernelTimer.start()
    rs = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=lambda: rings), roi=roi)
    kernels = [k for _, _, k in hip.KernelTimer.stop()]
    assert kernels
    if any('k_bell_' in k for k in kernels):
        assert all(',rows' in k for k in kernels if 'k_bell_' in k), kernels
    ref_s = data.reshape(13 * 17, -1).astype(np.float64) @ \
        np.asarray(rings.todense()).reshape(40, -1).T.astype(np.float64)
    assert _close(rs['intensity'].raw_data, ref_s.reshape(13, 17, 40)[roi], F32_TOL)
    # float64 results (int32 frames): the f64 LDS-DMA kernel through the row list
    ds32 = _device_ds(ctx, data.astype(np.int32), 3)
    r64 = ctx.run_udf(dataset=ds32, udf=ApplyMasksUDF(mask_factories=lambda: masks), roi=roi)
    assert r64['intensity'].raw_data.dtype == np.float64
    assert _close(r64['intensity'].raw_data, ref[roi], 1e-6)


def test_default_partition_count_on_the_gpu_executor(ctx):
    """`num_partitions` not given: the reference's default is one partition per CPU core (a worker
    count); the GPU executor streams host data through one device, so the default is one partition
    per GiB there (device-resident arrays: one)."""
    from libertem_amd.udf.sumsigudf import SumSigUDF
    data = np.arange(8 * 9 * 16 * 16, dtype=np.uint16).reshape((8, 9, 16, 16))
    ds = ctx.load('memory', data=data, sig_dims=2)
    assert ds.get_num_partitions() == 1
    explicit = ctx.load('memory', data=data, sig_dims=2, num_partitions=5)
    assert explicit.get_num_partitions() == 5
    old = type(ds).HIP_DEFAULT_PARTITION_BYTES
    type(ds).HIP_DEFAULT_PARTITION_BYTES = 8192          # 16 frames per partition
    try:
        small = ctx.load('memory', data=data, sig_dims=2)
        assert small.get_num_partitions() == 5           # 72 frames of 512 B = 36 864 B -> 5
        res = ctx.run_udf(dataset=small, udf=SumSigUDF())
        assert np.array_equal(res['intensity'].data, data.sum(axis=(2, 3)).astype(np.float32))
    finally:
        type(ds).HIP_DEFAULT_PARTITION_BYTES = old


def test_results_kept_on_device(ctx):
    """run_udf(result_where='device'): declared buffers stay in HBM (HipArray behind
    `buffer.device_data`), `.data` downloads on access; several partitions, a 'sum' buffer, a
    sparse stack and an ROI."""
    import scipy.sparse as sp
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd.udf.sumsigudf import SumSigUDF
    from libertem_amd.udf.sum import SumUDF
    from libertem_amd.common.hiparray import HipArray, HostMappedArray

    rng = np.random.default_rng(33)
    data = rng.integers(0, 3000, (12, 11, 32, 32)).astype(np.uint16)
    masks = rng.random((7, 32, 32)).astype(np.float32)
    sparse = [sp.random(32, 32, density=0.05, random_state=k, dtype=np.float32).tocsr()
              for k in range(5)]
    ref = opath.apply_masks(data, masks, num_partitions=3)
    dense_sparse = np.stack([m.toarray() for m in sparse])
    ref_sp = opath.apply_masks(data, dense_sparse, num_partitions=3)
    ds = _device_ds(ctx, data, 3)
    r1, r2, r3, r4 = ctx.run_udf(dataset=ds, udf=[
        ApplyMasksUDF(mask_factories=lambda: masks), SumSigUDF(), SumUDF(),
        ApplyMasksUDF(mask_factories=[(lambda m=m: m) for m in sparse], use_sparse=True)],
        result_where='device')
    for r in (r1, r2, r4):              # (SumUDF post-processes its sig-sized sum on the host)
        dev = r['intensity'].device_data
        assert isinstance(dev, HipArray) and not isinstance(dev, HostMappedArray), type(dev)
        assert dev.torch.is_cuda
    assert tuple(r1['intensity'].device_data.shape) == (12 * 11, 7)
    # usable on the device without a copy ...
    on_dev = r1['intensity'].device_data.torch.reshape(12 * 11, 7).sum(dim=0).cpu().numpy()
    assert np.allclose(on_dev, ref.reshape(-1, 7).sum(axis=0), rtol=1e-4)
    # ... and downloaded on access
    assert _close(r1['intensity'].data, ref, F32_TOL)
    assert np.array_equal(r2['intensity'].data, data.reshape((12, 11, -1)).sum(axis=-1).astype(np.float32))
    assert np.array_equal(r3['intensity'].data, data.astype(np.float32).sum(axis=(0, 1)))
    assert _close(r4['intensity'].data, ref_sp, F32_TOL)
    # the next run without the option delivers host arrays again
    host = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=lambda: masks))
    assert host['intensity'].device_data is None
    assert _close(host['intensity'].data, ref, F32_TOL)
    # ROI
    roi = rng.random((12, 11)) < 0.4
    part = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=lambda: masks), roi=roi,
                       result_where='device')
    assert isinstance(part['intensity'].device_data, HipArray)
    assert _close(part['intensity'].raw_data, ref[roi], F32_TOL)
    with pytest.raises(ValueError):
        ctx.run_udf(dataset=ds, udf=SumSigUDF(), result_where='hbm')


@pytest.mark.parametrize('world', [2, 3])
def test_two_ranks_share_results_through_host_segment(tmp_path, world):
    """The N>1 result path on the GPU (one-GPU box: two gloo ranks driving GPU 0): every rank
    writes the rows of its nav shard into the node-shared page-locked segment, no data-path
    collective; each rank ends up with the complete result == single-process values."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env['PYTHONPATH'] = root + os.pathsep + env.get('PYTHONPATH', '')
    env['LIBERTEM_USE_HIP'] = '0'
    env['OMP_NUM_THREADS'] = '1'
    env['LTMI_SHM_MAX_SLOTS'] = '6'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}',
           '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(root, 'tests', 'dist_worker_gpu.py'), str(tmp_path)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-4000:]
    outs = [np.load(os.path.join(tmp_path, f'rank{k}.npz')) for k in range(world)]
    for o in outs:
        masks, full = o['masks'], o['sh_full']
        exp = opath.apply_masks(full, masks, num_partitions=2 * world)
        assert _close(o['sh_masks'], exp, F32_TOL)
        assert _close(o['coll_masks'], exp, F32_TOL)
        assert np.array_equal(o['sh_sum'], full.astype(np.float32).sum(axis=(0, 1)))
        assert np.array_equal(o['sh_sumsig'], full.astype(np.float32).sum(axis=(2, 3)))
        assert bool(o['sh_first_still_valid']) and bool(o['sh_bare_still_valid'])
        assert bool(o['sh_held_equal']) and str(o['sh_via_when_full']) == 'collective'
        # delivery through the shared segment called off by ONE buffer: the rows the kernels wrote
        # directly into it still reach every rank (through the collectives)
        assert _close(o['mix_masks'], exp, F32_TOL)
        assert np.array_equal(o['mix_late'], o['sh_sumsig'])
        assert np.array_equal(o['mix_sumsig'], o['sh_sumsig'])
        assert 4 <= int(o['sh_slots']) <= 6
        # live feed per rank + run_udf_iter: 2 steps, after step k the first k + 1 partitions of
        # every rank are merged (device merge + collective) and marked in the damage map
        live = o['live']
        exp_live = opath.apply_masks(live, masks).reshape((world, 20, -1))
        assert o['live_step_masks'].shape[0] == 2
        for k, n_done in enumerate((10, 20)):
            got = o['live_step_masks'][k].reshape((world, 20, -1))
            dmg = o['live_step_damage'][k].reshape((world, 20))
            assert np.all(dmg[:, :n_done]) and not np.any(dmg[:, n_done:])
            assert _close(got[:, :n_done], exp_live[:, :n_done], F32_TOL)
            assert np.all(got[:, n_done:] == 0)
        data, roi = o['rep_data'], o['rep_roi']
        exp2 = opath.apply_masks(data, masks, num_partitions=5)
        assert _close(o['rep_masks'], exp2, F32_TOL)
        assert _close(o['rep_roi_raw'], exp2.reshape((45, -1))[roi.reshape(-1)], F32_TOL)
        # the .mib series: each rank decoded only its 3 frames, everybody has the complete result
        fr = o['mib_frames'].reshape(6, -1).astype(np.float64)
        exp_m = (fr @ o['mib_masks'].reshape(4, -1).T.astype(np.float64)).reshape(world, 6 // world, 4)
        assert int(o['mib_local_frames']) == 6 // world
        assert _close(o['mib_full'], exp_m, F32_TOL)
        assert _close(o['mib_roi_raw'], exp_m[o['mib_roi']], F32_TOL)
    for k in ('sh_masks', 'sh_sum', 'sh_sumsig', 'rep_masks', 'rep_roi_raw', 'coll_masks', 'mib_full',
              'mib_roi_raw'):
        for o in outs[1:]:
            assert np.array_equal(outs[0][k], o[k]), k
    assert not [f for f in os.listdir('/dev/shm') if f.startswith(f'ltmi_{os.getuid()}_{port}')]


def test_nccl_backend_collectives_through_ltmi_comm(tmp_path):
    """The multi-GPU result path on the real backend: one "nccl" (RCCL) rank with
    LTMI_FORCE_COLLECTIVES=1 -- RCCL initialises, the executor builds the library's own communicator
    (ltmi_comm_unique_id / _create) and combines device buffers with ltmi_comm_all_gather /
    _all_reduce_sum on its stream (torch.distributed only carries the 128-byte id and barriers).
    With one rank the collectives are identities, so the results must equal the oracle's."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1',
               LOCAL_RANK='0', LOCAL_WORLD_SIZE='1', LTMI_FORCE_COLLECTIVES='1',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('LTMI_COMM', None)
    r = subprocess.run([sys.executable, os.path.join(root, 'tests', 'dist_worker_nccl1.py'),
                        str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:]
    o = np.load(os.path.join(tmp_path, 'nccl1.npz'))
    data, masks = o['data'], o['masks']
    exp = opath.apply_masks(data, masks, num_partitions=3)
    for via in ('rccl', 'auto'):
        assert _close(o[f'{via}_masks'], exp, F32_TOL)
        assert np.array_equal(o[f'{via}_sum'], data.astype(np.float32).sum(axis=(0, 1)))
        assert np.array_equal(o[f'{via}_sumsig'], data.astype(np.float32).sum(axis=(2, 3)))
    # the device collectives went through the library's communicator, not torch.distributed
    assert str(o['rccl_via']) == 'collective' and str(o['rccl_collective']) == 'ltmi_comm'
    assert int(o['iter_steps']) == 3 and _close(o['iter_last'], exp, F32_TOL)
    assert str(o['c_collective']) == 'ltmi_comm'
    assert np.allclose(o['c_sum'], o['c_data'].sum(axis=(0, 1)), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('launcher', ['torchrun', 'self'])
def test_bench_contract_with_two_ranks_on_one_gpu(launcher):
    """bench.py end to end on the N>1 path (sharded dataset, shared-segment delivery, max-over-ranks
    timing, one JSON line from rank 0) -- two gloo ranks on GPU 0 stand in for two GPUs.
    'torchrun': started the way the driver's contract describes; 'self': plain
    `python bench.py --gpus 2`, bench.py spawns its own ranks."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, LTMI_BENCH_DEVICE='0', LTMI_BENCH_BACKEND='gloo', OMP_NUM_THREADS='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR'):
        env.pop(k, None)
    if launcher == 'torchrun':
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
               '--master-addr', '127.0.0.1', '--master-port', str(port),
               os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1']
    else:
        cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3',
               '--warmup', '1']
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['warmup'] == 1 and d['scaling'] == 'weak'
    assert d['value'] > 0 and d['unit'] == 'frames/s' and d['vs_baseline'] is None
    assert d['roofline']['bound'] == 'hbm' and 0 < d['roofline']['frac'] < 1.2
    assert 'cpu_baseline' not in d or d['cpu_baseline'] is None or d['n_gpus'] == 1
    # the N>1 extras: delivery path, per-rank times, the RCCL-style gather of the same steps, and
    # strong scaling on C3 (128 GiB nav-split over the ranks)
    assert d['result_via'] == 'shm' and len(d['per_rank']) == 2
    assert all(p['kernel_ms_per_step'] > 0 for p in d['per_rank'])
    assert 'error' not in d['rccl_path'], d['rccl_path']
    assert d['rccl_path']['result_via'] == 'collective' and d['rccl_path']['value'] > 0
    assert 'error' not in d['strong_c3'], d['strong_c3']
    assert d['strong_c3']['frames_total'] == 512 * 512 and d['strong_c3']['scaling'] == 'strong'


def test_run_udf_async_on_the_gpu(ctx):
    """`run_udf(sync=False)` / `run_udf_iter(sync=False)` on the HIP executor: the run happens on the context's
    worker thread (its own current device / stream), the awaited result equals the synchronous one, also with
    the launch-ahead of a repeated plan."""
    import asyncio
    from libertem_amd.udf.masks import ApplyMasksUDF
    rng = np.random.default_rng(23)
    data = rng.integers(0, 1000, (5, 8, 64, 64)).astype(np.uint16)
    masks = rng.random((16, 64, 64)).astype(np.float32)
    ds = _device_ds(ctx, data, 2)
    udf = ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False, mask_count=16)
    want = ctx.run_udf(dataset=ds, udf=udf)['intensity'].data

    async def main():
        outs = [await ctx.run_udf(dataset=ds, udf=udf, sync=False) for _ in range(4)]
        parts = []
        async for part in ctx.run_udf_iter(dataset=ds, udf=udf, sync=False):
            parts.append(np.array(part.buffers[0]['intensity'].data))
        return outs, parts
    outs, parts = asyncio.run(main())
    assert all(np.array_equal(o['intensity'].data, want) for o in outs)
    assert len(parts) == 2 and np.array_equal(parts[-1], want)
    assert _close(want, opath.apply_masks(data, masks, num_partitions=2), F32_TOL)
    assert np.array_equal(ctx.run_udf(dataset=ds, udf=udf)['intensity'].data, want)      # back on the main thread


def test_bench_contract_with_eight_ranks_on_one_gpu():
    """The shape of the first 8-GPU run, on one GPU: `python bench.py --gpus 8` (bench.py starts its own 8
    ranks; gloo ranks on GPU 0 stand in for 8 GPUs, 8192 frames per rank instead of 65536): 8 nav shards,
    the node-shared result ring with 8 owners, launch-ahead of the recorded launches in the shared segment,
    the RCCL-style gather of the same steps, C3 nav-split 8 ways -- ONE JSON line, every extra present."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LTMI_BENCH_DEVICE='0', LTMI_BENCH_BACKEND='gloo', OMP_NUM_THREADS='1',
               LTMI_BENCH_FRAMES_PER_RANK='8192', LTMI_BENCH_PREHEAT='4')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '6', '--warmup', '2']
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 8 and d['steps'] == 6 and d['scaling'] == 'weak' and d['value'] > 0
    assert d['config']['frames_per_gpu'] == 8192 and d['test_hook_frames_per_rank'] == 8192
    assert d['result_via'] == 'shm' and 'segment' in d['value_path'] and len(d['per_rank']) == 8
    assert sorted(p['rank'] for p in d['per_rank']) == list(range(8))
    assert d['launch_ahead'] > 0                         # (rank 0's count: the steps after the second one)
    assert 'extras_incomplete' not in d, d.get('extras_incomplete')
    assert 'error' not in d['rccl_path'], d['rccl_path']
    assert d['rccl_path']['result_via'] == 'collective' and d['rccl_path']['value'] > 0
    assert 'error' not in d['strong_c3'], d['strong_c3']
    assert d['strong_c3']['scaling'] == 'strong' and d['strong_c3']['frames_total'] == 16384
    assert d['f32_instruction'] and 'error' not in d['f32_instruction']
    # the strict float32-instruction leg as flat scalars inside `roofline` (a record that keeps scalars only carries it)
    assert 0 < d['roofline']['f32_instr_kernel_frac'] < 1 and d['roofline']['f32_instr_ms_per_step'] > 0
    assert d['roofline']['f32_instr_rel_err'] < 1e-5 and len(d['config']['arithmetic_detail']) < 240
    assert not [f for f in os.listdir('/dev/shm') if f.startswith(f'ltmi_{os.getuid()}_')]


# ---- caches and launch-ahead state: staleness and concurrency (round-4 review) ---------------------------------
class _MaskHolder:
    """a factory that is a BOUND METHOD: what it returns depends on attributes of its object"""

    def __init__(self, masks):
        self.mask = masks
        self.gain = np.float32(1)

    def make(self):
        return self.mask * self.gain


def test_cached_stacks_follow_bound_methods_and_captured_objects(ctx):
    """`mask_factories=holder.make` with `holder.mask[:] = ...` (or `holder.gain = 2`) between runs, and a closure
    over an object whose array is edited in place: the reference re-evaluates the factories every run
    (common/container.py:260-314); the cached stack / device image / run plan here must not survive the edit.
    cache=False and Context.invalidate_caches() for what no fingerprint can see."""
    from libertem_amd.udf.masks import ApplyMasksUDF
    rng = np.random.default_rng(77)
    data = rng.integers(0, 1000, (4, 6, 64, 64)).astype(np.uint16)
    ds = _device_ds(ctx, data, 2)
    holder = _MaskHolder(rng.random((5, 64, 64)).astype(np.float32))
    udf = ApplyMasksUDF(mask_factories=holder.make, use_sparse=False, mask_count=5)
    for edit in (lambda: None, lambda: holder.mask.__setitem__((slice(None), slice(10, 20)), 0.25),
                 lambda: setattr(holder, 'gain', np.float32(3)),
                 lambda: holder.mask.__setitem__((2, 5, 7), 100.0)):
        edit()
        for _ in range(3):                                   # (run 3 is launched ahead of the book-keeping)
            got = ctx.run_udf(dataset=ds, udf=udf)['intensity'].data
            assert _close(got, opath.apply_masks(data, holder.make(), num_partitions=2), F32_TOL)
    h2 = _MaskHolder(rng.random((3, 64, 64)).astype(np.float32))
    udf2 = ApplyMasksUDF(mask_factories=lambda: h2.mask, use_sparse=False, mask_count=3)
    for edit in (lambda: None, lambda: h2.mask.__imul__(np.float32(0.5)), lambda: h2.mask.__setitem__((1, 63, 63), -4.0)):
        edit()
        for _ in range(3):
            got = ctx.run_udf(dataset=ds, udf=udf2)['intensity'].data
            assert _close(got, opath.apply_masks(data, h2.mask, num_partitions=2), F32_TOL)
    # a counter the factory keeps in a captured dict changes with every evaluation: such a factory is evaluated
    # afresh every run without being told to (the reference does that for every factory)
    state = {'n': 0}
    base = rng.random((2, 64, 64)).astype(np.float32)

    def counting():
        state['n'] += 1
        return base * np.float32(state['n'])
    udf3 = ApplyMasksUDF(mask_factories=counting, use_sparse=False, mask_count=2)
    r1 = ctx.run_udf(dataset=ds, udf=udf3)['intensity'].data
    n1 = state['n']
    r2 = ctx.run_udf(dataset=ds, udf=udf3)['intensity'].data
    assert state['n'] > n1 and not np.array_equal(r1, r2)
    # what no fingerprint can see: a FILE the factory reads.  cache=False evaluates every run, like the reference;
    # Context.invalidate_caches() drops what a cached udf object holds
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'masks.npy')
        np.save(path, base)
        from_file = (lambda: np.load(path))
        cached = ApplyMasksUDF(mask_factories=from_file, use_sparse=False, mask_count=2)
        fresh = ApplyMasksUDF(mask_factories=from_file, use_sparse=False, mask_count=2, cache=False)
        for u in (cached, fresh):
            assert _close(ctx.run_udf(dataset=ds, udf=u)['intensity'].data,
                          opath.apply_masks(data, base, num_partitions=2), F32_TOL)
        np.save(path, base * 2)
        want = opath.apply_masks(data, base * 2, num_partitions=2)
        assert _close(ctx.run_udf(dataset=ds, udf=fresh)['intensity'].data, want, F32_TOL)
        stale = ctx.run_udf(dataset=ds, udf=cached)['intensity'].data          # (documented: the file is invisible)
        assert not _close(stale, want, F32_TOL)
        ctx.invalidate_caches()
        assert _close(ctx.run_udf(dataset=ds, udf=cached)['intensity'].data, want, F32_TOL)


def test_two_contexts_interleave_cached_plans(ctx):
    """Two Contexts (two executors) in one process taking turns with plans that are launched ahead: the launch-ahead
    state belongs to the executor, not to the process (round-4 review: one run swallowed or rejected the other's
    launch)."""
    from libertem_amd import hip
    from libertem_amd.api import Context
    from libertem_amd.udf.masks import ApplyMasksUDF
    rng = np.random.default_rng(78)
    ctx2 = Context.make_with('hip', gpus=0)
    try:
        d1 = rng.integers(0, 1000, (4, 8, 64, 64)).astype(np.uint16)
        d2 = rng.integers(0, 1000, (6, 4, 64, 64)).astype(np.uint16)
        m1 = rng.random((16, 64, 64)).astype(np.float32)
        m2 = rng.random((7, 64, 64)).astype(np.float32)
        ds1, ds2 = _device_ds(ctx, d1, 2), _device_ds(ctx2, d2, 3)
        u1 = ApplyMasksUDF(mask_factories=lambda: m1, use_sparse=False, mask_count=16)
        u2 = ApplyMasksUDF(mask_factories=lambda: m2, use_sparse=False, mask_count=7)
        ref1, ref2 = opath.apply_masks(d1, m1, num_partitions=2), opath.apply_masks(d2, m2, num_partitions=3)
        n0 = hip.LaunchReplay.n_ahead
        for rep in range(6):
            assert _close(ctx.run_udf(dataset=ds1, udf=u1)['intensity'].data, ref1, F32_TOL)
            assert _close(ctx2.run_udf(dataset=ds2, udf=u2)['intensity'].data, ref2, F32_TOL)
        assert hip.LaunchReplay.n_ahead - n0 == 4 * (2 + 3)           # both kept launching ahead (runs 3 .. 6)
        assert ctx.executor.replay is not ctx2.executor.replay
        assert ctx.executor.replay.expected is None and ctx2.executor.replay.expected is None
        # from two threads at once
        import threading
        errs = []

        def loop(c, ds, u, ref):
            try:
                for _ in range(8):
                    if not _close(c.run_udf(dataset=ds, udf=u)['intensity'].data, ref, F32_TOL):
                        errs.append('wrong result')
            except Exception as e:                                   # noqa: BLE001
                errs.append(repr(e))
        ts = [threading.Thread(target=loop, args=(ctx, ds1, u1, ref1)),
              threading.Thread(target=loop, args=(ctx2, ds2, u2, ref2))]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errs, errs
    finally:
        ctx2.close()


def test_sync_run_while_async_run_is_pending(ctx):
    """`run_udf(sync=False)` runs on the context's worker thread; a synchronous `run_udf` issued from the event-loop
    thread while it is in flight -- same executor, another cached plan -- waits for it (one run at a time per
    executor) instead of sharing its delivery targets and launch-ahead state."""
    import asyncio
    from libertem_amd.udf.masks import ApplyMasksUDF
    rng = np.random.default_rng(79)
    d1 = rng.integers(0, 1000, (16, 16, 64, 64)).astype(np.uint16)
    d2 = rng.integers(0, 1000, (3, 5, 64, 64)).astype(np.uint16)
    m1 = rng.random((16, 64, 64)).astype(np.float32)
    m2 = rng.random((4, 64, 64)).astype(np.float32)
    ds1, ds2 = _device_ds(ctx, d1, 4), _device_ds(ctx, d2, 1)
    u1 = ApplyMasksUDF(mask_factories=lambda: m1, use_sparse=False, mask_count=16)
    u2 = ApplyMasksUDF(mask_factories=lambda: m2, use_sparse=False, mask_count=4)
    ref1, ref2 = opath.apply_masks(d1, m1, num_partitions=4), opath.apply_masks(d2, m2, num_partitions=1)
    for _ in range(3):                                               # both plans cached and launched ahead
        ctx.run_udf(dataset=ds1, udf=u1)
        ctx.run_udf(dataset=ds2, udf=u2)

    async def main():
        for _ in range(5):
            pending = asyncio.ensure_future(ctx.run_udf(dataset=ds1, udf=u1, sync=False))
            await asyncio.sleep(0)                                    # the worker thread picks it up
            got2 = ctx.run_udf(dataset=ds2, udf=u2)['intensity'].data    # sync, from the loop's thread
            got1 = (await pending)['intensity'].data
            assert _close(got1, ref1, F32_TOL) and _close(got2, ref2, F32_TOL)
    asyncio.run(main())
    assert ctx.executor.replay.expected is None and ctx.executor.replay.recording is None


def test_radial_fourier_folded_through_run_udf_with_roi(ctx):
    """RadialFourierAnalysis on float32 frames goes through the row-mirror fold (MaskContainer tells the handle the
    detector shape); with a region of interest the frames are read through a row list.  Against the oracle."""
    from libertem_amd import hip
    rng = np.random.default_rng(91)
    data = rng.random((6, 8, 128, 128)).astype(np.float32)
    ds = _device_ds(ctx, data, 2)
    analysis = ctx.create_radial_fourier_analysis(dataset=ds, n_bins=2, max_order=12)
    hip.KernelTimer.start()
    res = ctx.run_udf(dataset=ds, udf=analysis.get_udf())['intensity'].data
    kernels = {k.split('<')[0] for _, _, k in hip.KernelTimer.stop()}
    assert kernels == {'k_dense_fold'}, kernels
    p = analysis.parameters
    stack = np.asarray(analysis.get_mask_factories()())
    ref = np.tensordot(data.astype(np.float64), stack.astype(np.complex128), axes=([2, 3], [1, 2]))
    assert res.dtype == np.complex64 and _close(res, ref, F32_TOL)
    roi = np.zeros((6, 8), bool)
    roi[::2, 1::3] = True
    hip.KernelTimer.start()
    res_r = ctx.run_udf(dataset=ds, udf=analysis.get_udf(), roi=roi)['intensity'].raw_data
    labels = [k for _, _, k in hip.KernelTimer.stop()]
    assert labels and all('k_dense_fold' in k for k in labels), labels
    assert _close(res_r, ref[roi], F32_TOL)
    assert p['mask_count'] == 26


@pytest.mark.parametrize('use_sparse', [True, None])
@pytest.mark.parametrize('dtype', ['float32', 'uint16'])
def test_radial_fourier_sparse_bins_through_run_udf_with_roi(ctx, monkeypatch, dtype, use_sparse):
    """RadialFourierAnalysis with several bins and use_sparse=True (second C5 run): the CSR stack is a set
    of column blocks with one support each -- folded dense images per bin on k_dense_fold / k_dense_fold16 (the handle
    learns the detector shape from MaskContainer).  Whole scan and a region of interest, against float64."""
    from libertem_amd import hip
    monkeypatch.setenv('LTMI_SPARSE_BAND', '1')             # (a small stack: whatever the cost estimate says)
    rng = np.random.default_rng(92)
    if dtype == 'float32':
        data = rng.random((6, 8, 128, 128)).astype(np.float32)
    else:
        data = rng.integers(0, 4096, (6, 8, 128, 128)).astype(np.uint16)
    ds = _device_ds(ctx, data, 2)
    if use_sparse is None:
        # the reference's heuristic declares a few wide bins DENSE (analysis/radialfourier.py:334-341): MaskContainer
        # finds the blocks in the dense stack and hands it over as CSR all the same
        analysis = ctx.create_radial_fourier_analysis(dataset=ds, n_bins=3, max_order=12)
        assert analysis.parameters['use_sparse'] is False
    else:
        analysis = ctx.create_radial_fourier_analysis(dataset=ds, n_bins=3, max_order=12, use_sparse=True)
        assert analysis.parameters['use_sparse'] is True
    hip.KernelTimer.start()
    res = ctx.run_udf(dataset=ds, udf=analysis.get_udf())['intensity'].data
    labels = [k for _, _, k in hip.KernelTimer.stop()]
    assert labels and all('k_dense_fold' in k and 'banded: 3 blocks' in k for k in labels), labels
    stack = analysis.get_mask_factories()()
    if use_sparse is None:
        dense = np.asarray(stack).reshape((39, -1)).T.astype(np.complex128)
    else:
        dense = np.asarray(stack.to_px_by_masks(dtype=np.complex64).todense()).astype(np.complex128)   # (n_px, 39)
    ref = (data.reshape((48, -1)).astype(np.float64) @ dense).reshape((6, 8, -1))
    assert res.dtype == np.complex64 and _close(res, ref, F32_TOL)
    roi = np.zeros((6, 8), bool)
    roi[1::2, ::3] = True
    hip.KernelTimer.start()
    res_r = ctx.run_udf(dataset=ds, udf=analysis.get_udf(), roi=roi)['intensity'].raw_data
    labels = [k for _, _, k in hip.KernelTimer.stop()]
    assert labels and all('banded' in k for k in labels), labels
    assert _close(res_r, ref[roi], F32_TOL)


# --- non-finite pixels: a sparse stack multiplies stored entries only, a dense one every zero ------------------
def zlib_seed(*what):
    import zlib
    return zlib.crc32(repr(what).encode())


def _nf_scan(rng, nav, sig, stored, unstored, where):
    """float32 scan with NaN / Inf pixels in a few frames: where='unstored': pixels no mask stores;
    'stored': pixels some masks store; a NaN, a +Inf and a +Inf / -Inf pair"""
    n = int(np.prod(nav))
    data = (rng.random((n, int(np.prod(sig)))) + 0.1).astype(np.float32)
    pool = unstored if where == 'unstored' else stored
    assert len(pool) > 8
    data[1, pool[len(pool) // 2]] = np.nan
    data[n // 2, pool[len(pool) // 3]] = np.inf
    data[n - 1, pool[0]] = np.inf
    data[n - 1, pool[-1]] = -np.inf
    data[n - 2, pool[len(pool) // 5]] = np.nan
    return data.reshape(tuple(nav) + tuple(sig))


def _same_nf(res, ref):
    parts = (np.real, np.imag) if np.iscomplexobj(ref) else (np.asarray,)
    return all(np.array_equal(np.isnan(f(res)), np.isnan(f(ref))) and
               np.array_equal(np.isposinf(f(res)), np.isposinf(f(ref))) and
               np.array_equal(np.isneginf(f(res)), np.isneginf(f(ref))) for f in parts)


@pytest.mark.parametrize('where', ['unstored', 'stored'])
@pytest.mark.parametrize('route', ['blocked_rings', 'densified_wide_rings', 'folded_wide_rings', 'banded_radial_fourier'])
def test_sparse_stack_non_finite_pixels_through_run_udf(ctx, monkeypatch, route, where):
    """ApplyMasksUDF / RadialFourierAnalysis with use_sparse='scipy.sparse' on float32 frames that hold NaN / Inf
    pixels, through run_udf, against the oracle's restatement of the reference's CSR loop
    (oracle.path.apply_masks_sparse -> rmatmul, common/numba/__init__.py:153-184; udf/masks.py:68-77): whichever
    kernel MaskContainer picks -- blocked / scatter image, the stack multiplied dense, dense and folded about the
    detector rows, banded images per radial-Fourier bin -- a non-finite pixel reaches exactly the masks that store
    it (a) not at all when no mask stores it, (b) those masks only."""
    import scipy.sparse as sp
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd import masks as M
    from libertem_amd import hip
    rng = np.random.default_rng(zlib_seed(route, where))
    nav = (4, 6)
    if route == 'banded_radial_fourier':
        monkeypatch.setenv('LTMI_SPARSE_BAND', '1')
        sig = (128, 128)
        n_bins, max_order = 3, 12
        csr_masks = sp.csr_matrix(omasks.radial_mask_stack_csr(128, 128, 64., 64., 4., 50., n_bins, max_order))
    else:
        sig = (64, 64)
        n_bins = {'blocked_rings': 192, 'densified_wide_rings': 24, 'folded_wide_rings': 40}[route]
        # (rings about the detector centre are even under a mirror of the rows: multiplied folded; off centre: not)
        cx, cy = (30.3, 33.7) if route == 'densified_wide_rings' else (32, 32)
        csr_masks = sp.csr_matrix(omasks.radial_bins(cx, cy, 64, 64, radius=28, n_bins=n_bins, use_sparse=True,
                                                     dtype=np.float32))             # (n_masks, px)
    counts = np.asarray((csr_masks != 0).sum(axis=0)).reshape(-1)
    stored, unstored = np.flatnonzero(counts > 0), np.flatnonzero(counts == 0)
    data = _nf_scan(rng, nav, sig, stored, unstored, where)
    ds = _device_ds(ctx, data, 2)
    if route == 'banded_radial_fourier':
        analysis = ctx.create_radial_fourier_analysis(dataset=ds, cx=64., cy=64., ri=4., ro=50., n_bins=n_bins,
                                                      max_order=max_order, use_sparse=True)
        udf = analysis.get_udf()
    else:
        def rings():
            return M.radial_bins(cx, cy, 64, 64, radius=28, n_bins=n_bins, use_sparse=True, dtype=np.float32)
        udf = ApplyMasksUDF(mask_factories=rings, use_sparse='scipy.sparse', mask_count=n_bins,
                            mask_dtype=np.float32)
    hip.KernelTimer.start()
    got = ctx.run_udf(dataset=ds, udf=udf)['intensity'].data
    labels = [k for _, _, k in hip.KernelTimer.stop()]
    want = {'blocked_rings': ('k_scatter', 'k_bell'), 'densified_wide_rings': ('k_dense_lds', 'k_dense_mfma'),
            'folded_wide_rings': ('k_dense_fold',), 'banded_radial_fourier': ('k_dense_fold',)}[route]
    assert labels and all(k.startswith(want) or 'column blocks' in k for k in labels), labels
    assert all(k.endswith('+nf') for k in labels), labels                  # the guarded product ran
    if route == 'banded_radial_fourier':
        assert all('banded' in k for k in labels), labels
    ref = opath.apply_masks_sparse(data, csr_masks, num_partitions=2)
    assert got.shape == ref.shape and got.dtype == ref.dtype
    if where == 'unstored':
        assert np.all(np.isfinite(ref))
    else:
        assert not np.all(np.isfinite(ref[0, 1]))
    assert _same_nf(got, ref), labels
    ok = np.isfinite(ref)
    scale = np.abs(ref[ok]).max()
    assert np.allclose(got[ok], ref[ok], rtol=F32_TOL, atol=F32_TOL * scale)
    # a region of interest reads the frames through a row list
    roi = np.zeros(nav, bool)
    roi[0, 1] = roi[3, 5] = roi[2, 0] = roi[1, 3] = True
    got_r = ctx.run_udf(dataset=ds, udf=udf, roi=roi)['intensity'].raw_data
    assert _same_nf(got_r, ref[roi])
    ok = np.isfinite(ref[roi])
    assert np.allclose(got_r[ok], ref[roi][ok], rtol=F32_TOL, atol=F32_TOL * scale)


@pytest.mark.parametrize('where', ['unstored', 'stored'])
def test_dense_banded_radial_fourier_non_finite_pixels_through_run_udf(ctx, monkeypatch, where):
    """The reference's heuristic declares a radial-Fourier stack of a few wide bins DENSE
    (analysis/radialfourier.py:334-341) and multiplies it with `flat_tile @ masks` (udf/masks.py:76-77): 0 * NaN = NaN,
    a non-finite pixel reaches every mask.  MaskContainer hands such a stack over as banded CSR (one dense image per
    bin, which never reads the pixels outside the bins) -- marked as dense (ltmi_masks_set_dense_origin), so the
    results keep the dense arithmetic; against oracle.path.radial_fourier_analysis."""
    from libertem_amd import hip
    monkeypatch.setenv('LTMI_SPARSE_BAND', '1')
    rng = np.random.default_rng(zlib_seed('dense-banded', where))
    nav, sig = (4, 6), (128, 128)
    stack = omasks.radial_mask_stack(128, 128, 64., 64., 4., 50., 3, 12)              # (39, 128, 128) complex64
    counts = (stack != 0).sum(axis=0).reshape(-1)
    stored, unstored = np.flatnonzero(counts > 0), np.flatnonzero(counts == 0)
    data = _nf_scan(rng, nav, sig, stored, unstored, where)
    ds = _device_ds(ctx, data, 2)
    analysis = ctx.create_radial_fourier_analysis(dataset=ds, cx=64., cy=64., ri=4., ro=50., n_bins=3, max_order=12)
    assert analysis.parameters['use_sparse'] is False
    hip.KernelTimer.start()
    got = ctx.run_udf(dataset=ds, udf=analysis.get_udf())['intensity'].data
    labels = [k for _, _, k in hip.KernelTimer.stop()]
    assert labels and all('banded' in k and k.endswith('+nf') for k in labels), labels
    with np.errstate(invalid='ignore'):
        ref = opath.radial_fourier_analysis(data, num_partitions=2, cx=64., cy=64., ri=4., ro=50., n_bins=3,
                                            max_order=12)['intensity']
    assert got.shape == ref.shape and got.dtype == ref.dtype
    assert np.all(np.isnan(ref[0, 1].real)) and np.all(np.isnan(ref[0, 1].imag))      # the NaN frame: every mask
    # NaN pixels: exactly the reference's NaNs; Inf pixels: the same entries are non-finite (whether Inf * (w + 0j)
    # comes out as (Inf, NaN) or (NaN, NaN) is a property of the BLAS kernel behind `flat_tile @ masks`)
    nan_frames = np.isnan(data).any(axis=(2, 3))
    assert _same_nf(got[nan_frames], ref[nan_frames]), labels
    assert np.array_equal(np.isfinite(got.real), np.isfinite(ref.real)) and \
        np.array_equal(np.isfinite(got.imag), np.isfinite(ref.imag)), labels
    ok = np.isfinite(ref)
    assert np.allclose(got[ok], ref[ok], rtol=F32_TOL, atol=F32_TOL * np.abs(ref[ok]).max())
    # shifted masks need the dense image: the banded handle is not taken for them (and results agree)
    from libertem_amd.udf.masks import ApplyMasksUDF
    clean = np.where(np.isfinite(data), data, 1).astype(np.float32)
    ds2 = _device_ds(ctx, clean, 2)
    udf_s = ApplyMasksUDF(mask_factories=lambda: stack, mask_count=39, mask_dtype=np.complex64, shifts=(2, -3))
    got_s = ctx.run_udf(dataset=ds2, udf=udf_s)['intensity'].data
    ref_s = opath.apply_masks_shifted(clean, stack, np.broadcast_to(np.array([2, -3]), (24, 2)))
    assert _close(got_s, ref_s.reshape(got_s.shape), F32_TOL)


@pytest.mark.parametrize('kind', ['ndarray', 'foreign', 'foreign_pinned_by_env', 'memmap'])
def test_host_upload_paths_staged_and_in_place(ctx, tmp_path, monkeypatch, kind):
    """Host-resident frames: memory whose mapping provably belongs to one object -- an np.memmap, a large ndarray
    (a glibc malloc chunk that is an mmap of its own) -- is page-locked in place and DMA-ed from directly; anything
    else (here: an array over another allocator's memory) is staged through the page-locked bounce buffers with the
    multi-threaded copy; LTMI_PIN_USER_ARRAYS=1 restores in-place page-locking of any large array.  Same results
    either way, on a scan of several upload chunks (64 MiB of uint16 frames, forced 8 MiB chunks)."""
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd.io.dataset import base as dsbase
    monkeypatch.setattr(dsbase.Negotiator, 'HIP_STAGING_CHUNK', 8 << 20)
    if kind == 'foreign_pinned_by_env':
        monkeypatch.setenv('LTMI_PIN_USER_ARRAYS', '1')
    else:
        monkeypatch.delenv('LTMI_PIN_USER_ARRAYS', raising=False)
    rng = np.random.default_rng(61)
    shape = (16, 32, 256, 256)                                   # 64 MiB
    if kind == 'memmap':
        data = np.memmap(tmp_path / 'scan.bin', dtype=np.uint16, mode='w+', shape=shape)
        data[...] = rng.integers(0, 4096, shape, dtype=np.uint16)
    elif kind == 'ndarray':
        data = rng.integers(0, 4096, shape, dtype=np.uint16)
    else:
        keep = torch.from_numpy(rng.integers(0, 4096, shape, dtype=np.uint16).view(np.int16)).clone()
        data = keep.numpy().view(np.uint16)                      # torch's CPU allocator, not malloc's mmap chunks
    masks = rng.random((5, 256, 256)).astype(np.float32)
    ds = ctx.load('memory', data=data, sig_dims=2, num_partitions=2)
    udf = ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False, mask_count=5, mask_dtype=np.float32)
    got = ctx.run_udf(dataset=ds, udf=udf)['intensity'].data
    stagers = list(ds.__dict__.get('_hip_stagers', {}).values())
    assert len(stagers) == 1
    in_place = stagers[0].registered is not None
    assert in_place == (kind != 'foreign'), (kind, in_place)
    ref = np.asarray(data).reshape((512, -1)).astype(np.float64) @ masks.reshape((5, -1)).astype(np.float64).T
    assert np.allclose(got.reshape((512, 5)), ref, rtol=F32_TOL, atol=0)
    # the frames change in place between two runs (the reference reads the array afresh every run)
    data[3, 7] = 0
    got2 = ctx.run_udf(dataset=ds, udf=udf)['intensity'].data
    ref[3 * 32 + 7] = 0
    assert np.allclose(got2.reshape((512, 5)), ref, rtol=F32_TOL, atol=0)
    ds.close_stagers()


def test_mixed_numpy_only_and_device_capable_udfs_run_on_the_host(ctx):
    """Round-5 advice: `[SumUDF(), MyNumpyOnlyUDF()]` on a GPU context (the reference plans per UDF, udf/base.py:162-329,
    and runs such mixes): every UDF of the run offers NumPy, so the run happens on the host -- with a RuntimeWarning that
    names the UDFs; the device-capable UDF on its own still runs on the MI355X, and a native operator without a NumPy
    path next to a NumPy-only UDF is refused."""
    from libertem_amd.udf.base import UDF
    from libertem_amd.udf.sum import SumUDF
    from libertem_amd.udf.masks import ApplyMasksUDF

    class NumpyMaxUDF(UDF):
        def get_result_buffers(self):
            return {'peak': self.buffer(kind='nav', dtype='float32')}

        def get_backends(self):
            return (self.BACKEND_NUMPY,)

        def process_frame(self, frame):
            self.results.peak[:] = np.max(frame)

    rng = np.random.default_rng(73)
    data = rng.random((4, 6, 16, 16)).astype(np.float32)
    ds = ctx.load('memory', data=data, num_partitions=2, sig_dims=2)
    with pytest.warns(RuntimeWarning, match='NumpyMaxUDF'):
        s, m = ctx.run_udf(dataset=ds, udf=[SumUDF(), NumpyMaxUDF()])
    assert np.allclose(s['intensity'].data, data.sum(axis=(0, 1)), rtol=1e-6)
    assert np.array_equal(m['peak'].data, data.max(axis=(2, 3)))
    alone = ctx.run_udf(dataset=ds, udf=SumUDF())
    assert np.allclose(alone['intensity'].data, data.sum(axis=(0, 1)), rtol=1e-6)
    with pytest.raises(ValueError, match='no common array backend'):
        ctx.run_udf(dataset=ds, udf=[ApplyMasksUDF(mask_factories=lambda: np.ones((1, 16, 16), np.float32)),
                                     NumpyMaxUDF()])


@pytest.mark.parametrize('route', ['dense_u16_signed', 'dense_f32_40_masks', 'radial_fourier_folded',
                                   'radial_fourier_banded_sparse', 'sparse_rings_u16', 'sparse_rings_f32', 'com_raw_sums'])
def test_run_udf_elementwise_error_bound(ctx, monkeypatch, route):
    """Round-5 review: many run_udf tests compare norm-wise (`_close`: atol = 1e-5 x max |ref|), which hides an error in a
    small entry next to a large one.  Here EVERY entry of the result of every kernel route is held to the bound that is
    right for a float32 sum of signed terms, |got - ref64| <= 1e-5 x sum_p |x_p| |w_p| (the north star's 1e-5 relative to
    what was added up) -- signed masks, signed frames, complex radial-Fourier stacks, sparse stacks -- through run_udf."""
    import scipy.sparse as sp
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd import masks as M
    monkeypatch.setenv('LTMI_SPARSE_BAND', '1')
    rng = np.random.default_rng(zlib_seed('elementwise', route))
    nav, sig = (4, 8), (128, 128)
    n_px = sig[0] * sig[1]
    u16 = rng.integers(0, 4096, nav + sig, dtype=np.uint16)
    f32 = (rng.random(nav + sig) - 0.4).astype(np.float32)
    stack = None
    if route == 'dense_u16_signed':
        data = u16
        w = (rng.random((16,) + sig) - 0.5).astype(np.float32)
        w[3] *= 1e-4                                             # a column five orders below its neighbours
        udf = ApplyMasksUDF(mask_factories=lambda: w, use_sparse=False, mask_count=16, mask_dtype=np.float32)
        stack = w.reshape((16, -1))
    elif route == 'dense_f32_40_masks':
        data = f32
        w = (rng.random((40,) + sig) - 0.5).astype(np.float32)
        udf = ApplyMasksUDF(mask_factories=lambda: w, use_sparse=False, mask_count=40, mask_dtype=np.float32)
        stack = w.reshape((40, -1))
    elif route in ('radial_fourier_folded', 'radial_fourier_banded_sparse'):
        data = f32
        ds0 = _device_ds(ctx, data, 2)
        kw = dict(n_bins=1, max_order=12) if route == 'radial_fourier_folded' else \
            dict(n_bins=3, max_order=12, use_sparse=True)
        an = ctx.create_radial_fourier_analysis(dataset=ds0, **kw)
        udf = an.get_udf()
        st = an.get_mask_factories()()
        stack = np.asarray(st.to_px_by_masks(dtype=np.complex64).todense()).T if hasattr(st, 'to_px_by_masks') \
            else np.asarray(st).reshape((len(st), -1))
    elif route in ('sparse_rings_u16', 'sparse_rings_f32'):
        data = u16 if route.endswith('u16') else f32

        def rings():
            return M.radial_bins(64, 64, 128, 128, n_bins=200, use_sparse=True, dtype=np.float32)
        udf = ApplyMasksUDF(mask_factories=rings, use_sparse='scipy.sparse', mask_count=200, mask_dtype=np.float32)
        stack = np.asarray(sp.csr_matrix(omasks.radial_bins(64, 64, 128, 128, n_bins=200, use_sparse=True,
                                                            dtype=np.float32)).todense())
    else:
        data = u16
        an = ctx.create_com_analysis(dataset=_device_ds(ctx, data, 2), cx=70., cy=60., mask_radius=50.)
        udf = an.get_udf()                                       # the three CoM sums: disk, y * disk, x * disk
        facs = an.get_mask_factories()
        stack = np.stack([np.asarray(f()) for f in facs]).astype(np.float32).reshape((3, -1))
    ds = _device_ds(ctx, data, 2)
    res = ctx.run_udf(dataset=ds, udf=udf)
    got = res['intensity'].data
    flat = data.reshape((-1, n_px)).astype(np.float64)
    wide = stack.astype(np.complex128 if np.iscomplexobj(stack) else np.float64)
    ref = (flat @ wide.T).reshape(got.shape)
    bound = 1e-5 * (np.abs(flat) @ np.abs(wide).T).reshape(got.shape) + 1e-30
    assert got.shape == ref.shape
    err = np.abs(got - ref)
    assert np.all(err <= bound), (route, float((err / bound).max()))


@pytest.mark.parametrize('case', recipes.NONFINITE_CASES, ids=lambda c: c['name'])
def test_non_finite_pixels_vs_reference_golden(ctx, golden_dir, monkeypatch, case):
    """NaN / Inf pixels against outputs of the IMPORTED reference (tests/golden/nonfinite.npz): ApplyMasksUDF with the
    stack given sparse must reproduce the reference's rmatmul (stored entries only) -- on the gather kernel, the
    blocked image, the scatter kernel and the densified route --, given dense its `flat_tile @ masks` / torch.mm
    (a NaN pixel anywhere in a frame makes every mask NaN)."""
    import scipy.sparse as sp
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd.udf import masks as um
    g = _load(golden_dir, 'nonfinite')
    data, stack = recipes.make_nonfinite_case(case)
    n_masks = stack.shape[0]
    ds = _device_ds(ctx, data, 2)
    ref_sparse = g[case['name'] + '__rmatmul_csr'].reshape(tuple(case['nav']) + (n_masks,))
    mats = [sp.csr_matrix(stack[k]) for k in range(n_masks)]
    for env in (dict(LTMI_SPARSE_BELL='0', LTMI_SPARSE_SCATTER='0', LTMI_DENSIFY_FILL='0'),    # gather kernel
                dict(LTMI_SPARSE_BELL='1', LTMI_SPARSE_SCATTER='0', LTMI_DENSIFY_FILL='0'),    # blocked image
                dict(LTMI_SPARSE_BELL='0', LTMI_SPARSE_SCATTER='1', LTMI_DENSIFY_FILL='0'),    # scatter kernel
                dict()):                                                                        # as dispatched
        for k in ('LTMI_SPARSE_BELL', 'LTMI_SPARSE_SCATTER', 'LTMI_DENSIFY_FILL'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        from libertem_amd.common import container as cont
        monkeypatch.setattr(cont, 'DENSIFY_FILL', float(env.get('LTMI_DENSIFY_FILL', '0.125')))
        um.clear_mask_cache()
        got = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=[(lambda m=m: m) for m in mats],
                                                        use_sparse='scipy.sparse', mask_dtype=stack.dtype,
                                                        cache=False))['intensity'].data
        assert got.dtype == ref_sparse.dtype
        assert _same_nf(got, ref_sparse), env
        ok = np.isfinite(ref_sparse)
        assert np.allclose(got[ok], ref_sparse[ok], rtol=F32_TOL, atol=F32_TOL * np.abs(ref_sparse[ok]).max()), env
    um.clear_mask_cache()
    got_d = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=lambda: stack, use_sparse=False,
                                                      mask_count=n_masks, mask_dtype=stack.dtype))['intensity'].data
    for key in ('__udf_dense_torch1', '__udf_dense_torch0'):
        ref = g[case['name'] + key]
        assert got_d.dtype == ref.dtype and got_d.shape == ref.shape
        nan_frames = np.isnan(data).any(axis=(2, 3))
        assert _same_nf(got_d[nan_frames], ref[nan_frames])
        assert np.array_equal(np.isfinite(got_d.real), np.isfinite(ref.real)) and \
            np.array_equal(np.isfinite(got_d.imag), np.isfinite(ref.imag))
        ok = np.isfinite(ref)
        assert np.allclose(got_d[ok], ref[ok], rtol=F32_TOL, atol=F32_TOL * np.abs(ref[ok]).max())


@pytest.mark.parametrize('mask_dtype', ['float32', 'complex64'])
def test_sparse_stack_on_complex_frames_non_finite_pixels(ctx, mask_dtype):
    """Complex frames against a SPARSE stack run on the dense real expansion of the stack; the reference multiplies them
    entry by stored entry (rmatmul with a complex left operand, common/numba/__init__.py:153-184): a NaN in EITHER part of
    a pixel reaches both parts of exactly the masks that store the pixel.  The expansion's gather image (all four real
    entries of a stored complex entry, zeros included) serves the frames with non-finite results."""
    import scipy.sparse as sp
    from libertem_amd.udf.masks import ApplyMasksUDF
    rng = np.random.default_rng(zlib_seed('complex-nf', mask_dtype))
    nav, sig = (3, 4), (16, 24)
    data = ((rng.random(nav + sig) - 0.3) + 1j * (rng.random(nav + sig) - 0.6)).astype(np.complex64)
    md = np.dtype(mask_dtype)
    dense = np.where(rng.random((6,) + sig) < 0.15, rng.random((6,) + sig) + 0.2, 0)
    if md.kind == 'c':
        dense = dense * np.exp(1j * rng.random((6,) + sig) * 5)
    dense = dense.astype(md)
    counts = (dense != 0).sum(axis=0).reshape(-1)
    stored, unstored = np.flatnonzero(counts > 0), np.flatnonzero(counts == 0)
    flat = data.reshape((12, -1))
    flat[1, unstored[3]] = np.nan + 0j                       # no mask stores it: no effect
    flat[2, stored[5]] = complex(np.nan, 0.25)               # NaN in the real part only
    flat[7, stored[11]] = complex(0.5, np.nan)               # ... in the imaginary part only
    flat[9, stored[2]] = complex(np.inf, 0.1)
    ds = _device_ds(ctx, data, 2)
    mats = [sp.csr_matrix(dense[k]) for k in range(6)]
    got = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=[(lambda m=m: m) for m in mats],
                                                    use_sparse='scipy.sparse', mask_dtype=md))['intensity'].data
    assert got.dtype == np.complex64
    with np.errstate(invalid='ignore', over='ignore'):
        ref = opath.apply_masks_sparse(data, sp.csr_matrix(dense.reshape((6, -1))), num_partitions=2, mask_dtype=md)
    assert np.all(np.isfinite(ref[0, 1])) and not np.all(np.isfinite(ref[0, 2]))
    # NaN where the reference is NaN (both parts), finite where it is finite
    assert np.array_equal(np.isfinite(got.real), np.isfinite(ref.real)) and \
        np.array_equal(np.isfinite(got.imag), np.isfinite(ref.imag))
    nan_frames = np.isnan(flat.real).any(axis=1) | np.isnan(flat.imag).any(axis=1)
    g2, r2 = got.reshape((12, 6)), ref.reshape((12, 6))
    assert np.array_equal(np.isnan(g2[nan_frames].real), np.isnan(r2[nan_frames].real))
    assert np.array_equal(np.isnan(g2[nan_frames].imag), np.isnan(r2[nan_frames].imag))
    ok = np.isfinite(ref)
    assert np.allclose(got[ok], ref[ok], rtol=F32_TOL, atol=F32_TOL * np.abs(ref[ok]).max())


@pytest.mark.gpu
@pytest.mark.parametrize('spec', [
    {"shape": "disk", "cx": 5, "cy": 6, "r": 7},
    {"shape": "disk", "cx": -1, "cy": -1, "r": 0},                       # nothing selected
    {"shape": "rect", "x": 3, "y": 2, "width": 6, "height": 9},
])
def test_analysis_roi_parameter(ctx, spec):
    """`parameters={'roi': {...}}` of SumAnalysis / MasksAnalysis (analysis/sum.py:100-101, analysis/masks.py:179-180,
    analysis/getroi.py): the reference's tests/analysis/test_analysis_sum.py:171-240 (`test_sum_with_roi`,
    `test_sum_zero_roi`) re-expressed, plus the rectangle and the masks analysis."""
    from libertem_amd import masks as M
    from libertem_amd.analysis.sum import SumAnalysis
    from libertem_amd.analysis.masks import MasksAnalysis
    rng = np.random.default_rng(11)
    data = rng.integers(0, 4096, (16, 16, 16, 16)).astype('<u2')
    ds = _device_ds(ctx, data, 8)
    if spec["shape"] == "disk":
        mask = M.circular(spec["cx"], spec["cy"], 16, 16, spec["r"])
    else:
        mask = M.rectangular(spec["x"], spec["y"], spec["width"], spec["height"], 16, 16)
    assert mask.shape == (16, 16) and mask.dtype == bool
    analysis = SumAnalysis(dataset=ds, parameters={"roi": spec})
    assert np.array_equal(analysis.get_roi(), mask)
    results = ctx.run(analysis)
    expected = data[mask, ...].astype(np.float64).sum(axis=0)
    assert results['intensity'].raw_data.shape == (16, 16)
    assert not np.allclose(results['intensity'].raw_data, data.astype(np.float64).sum(axis=(0, 1)))
    assert np.allclose(results['intensity'].raw_data, expected)
    assert np.allclose(results['intensity_lin'].raw_data, expected)

    stack = rng.random((3, 16, 16)).astype(np.float32)
    ma = MasksAnalysis(dataset=ds, parameters={"factories": [lambda i=i: stack[i] for i in range(3)], "roi": spec})
    assert np.array_equal(ma.get_roi(), mask)
    res = ctx.run(ma)
    want = np.einsum('nyx,kyx->nk', data[mask].astype(np.float64), stack.astype(np.float64))
    for k in range(3):
        got = res[f'mask_{k}'].raw_data
        assert got.shape == (16, 16)
        assert np.all(np.isnan(got[~mask]))                      # outside the roi: the fill value of float buffers
        if mask.any():
            assert np.allclose(got[mask], want[:, k], rtol=F32_TOL)
    assert SumAnalysis(dataset=ds, parameters={}).get_roi() is None
    with pytest.raises(NotImplementedError, match='unknown shape'):
        SumAnalysis(dataset=ds, parameters={"roi": {"shape": "star"}}).get_roi()


@pytest.mark.gpu
@pytest.mark.parametrize('resident', ['host', 'device'])
def test_runs_inside_a_partial_result_loop(ctx, resident):
    """The reference's tests/test_context.py test_udf_iter: inside `for res in ctx.run_udf_iter(...)` a second run on
    the SAME context with the damage so far as roi gives the partial result.  On the HIP executor the nested run goes
    to a sibling executor on the same GPU (Context._nested_context); host data goes through the dataset's shared
    upload stager in both runs, the prefetched chunk of the suspended run included."""
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd.udf.sumsigudf import SumSigUDF
    rng = np.random.default_rng(23)
    data = rng.integers(0, 4096, (8, 16, 64, 64)).astype(np.uint16)
    masks = rng.random((3, 64, 64)).astype(np.float32)
    if resident == 'device':
        ds = _device_ds(ctx, data, 4)
    else:
        ds = ctx.load('memory', data=data, num_partitions=4, sig_dims=2)

    def udfs():
        return [ApplyMasksUDF(mask_factories=lambda: masks, mask_count=3, mask_dtype=np.float32), SumSigUDF()]
    want = np.einsum('abyx,kyx->abk', data.astype(np.float64), masks.astype(np.float64))
    steps = 0
    for res in ctx.run_udf_iter(dataset=ds, udf=udfs()):
        steps += 1
        dmg = np.array(res.damage.data)
        ref = ctx.run_udf(dataset=ds, udf=udfs(), roi=dmg)
        for i, key in enumerate(('intensity', 'intensity')):
            got, again = res.buffers[i][key].data, ref[i][key].data
            assert np.array_equal(got[dmg], again[dmg])
        assert _close(res.buffers[0]['intensity'].data[dmg], want[dmg], F32_TOL)
        # a whole second iteration in between
        assert sum(1 for _ in ctx.run_udf_iter(dataset=ds, udf=SumSigUDF())) == 4
    assert steps == 4 and dmg.all()
    assert _close(res.buffers[0]['intensity'].data, want, F32_TOL)
    assert np.array_equal(res.buffers[1]['intensity'].data, data.astype(np.float64).sum(axis=(2, 3)).astype(res.buffers[1]['intensity'].data.dtype))
