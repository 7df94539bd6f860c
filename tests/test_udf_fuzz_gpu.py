"""
Randomised end-to-end runs through Context.run_udf on the HIP executor (random nav / sig shapes,
dtypes, partition counts, forced tile shapes, ROIs, host- or device-resident frames, corrections)
against the oracle.  `-m gpu` only.
"""
import numpy as np
import pytest

from oracle import path as opath, corrections as ocorr

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from libertem_amd.api import Context
    c = Context.make_with('hip', gpus=0)
    yield c
    c.close()


def _close(a, b, tol=1e-5):
    scale = max(float(np.abs(b).max()) if b.size else 0., 1e-30)
    return np.allclose(a, b, rtol=tol, atol=tol * scale)


@pytest.mark.parametrize('seed', range(int(__import__('os').environ.get('LTMI_FUZZ_SEEDS', '40'))))
def test_random_run(ctx, seed):
    from libertem_amd.common.hiparray import HipArray
    from libertem_amd.io.corrections import CorrectionSet
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd.udf.sum import SumUDF
    from libertem_amd.udf.sumsigudf import SumSigUDF
    rng = np.random.default_rng(1000 + seed)
    nav = tuple(int(x) for x in rng.integers(1, 7, int(rng.integers(1, 3))))
    sig = (int(rng.choice([8, 16, 17, 32])), int(rng.choice([8, 16, 23, 32])))
    dt = np.dtype(rng.choice(['uint8', 'uint16', 'int16', 'float32', 'int32']))
    n = int(np.prod(nav))
    if dt.kind in 'iu':
        data = rng.integers(0, 200 if dt.itemsize == 1 else 3000, nav + sig).astype(dt)
    else:
        data = (rng.random(nav + sig) * 10).astype(dt)
    num_partitions = int(rng.integers(1, min(n, 4) + 1))
    tileshape = None
    if rng.random() < 0.3:
        tileshape = (int(rng.integers(1, 5)), int(rng.choice([sig[0], max(1, sig[0] // 2)])), sig[1])
    resident = rng.choice(['host', 'device'])
    roi = None
    if rng.random() < 0.4:
        roi = rng.random(nav) < 0.5
    use_corr = rng.random() < 0.3 and tileshape is None
    n_masks = int(rng.choice([1, 3, 17]))
    masks = (rng.random((n_masks,) + sig) - 0.25).astype(np.float32)
    if resident == 'device':
        ds = ctx.load('memory', data=HipArray.from_numpy(data, 0), num_partitions=num_partitions,
                      sig_dims=2, tileshape=tileshape)
    else:
        ds = ctx.load('memory', data=data, num_partitions=num_partitions, sig_dims=2,
                      tileshape=tileshape)
    corr = None
    eff = data
    if use_corr:
        dark = rng.random(sig) * 3
        gain = rng.random(sig) + 0.5
        bad = np.zeros(sig, dtype=bool)
        bad[int(rng.integers(0, sig[0])), int(rng.integers(0, sig[1]))] = True
        corr = CorrectionSet(dark=dark, gain=gain, excluded_pixels=bad)
        eff = ocorr.correct(data, sig, dark=dark, gain=gain,
                            coords=[tuple(c) for c in np.argwhere(bad)])
    sel = eff.reshape((n,) + sig)
    if roi is not None:
        sel = sel[roi.reshape(-1)]
    info = (nav, sig, str(dt), num_partitions, tileshape, resident, roi is not None, use_corr)
    res = ctx.run_udf(dataset=ds, udf=[ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False),
                                       SumUDF(), SumSigUDF()], roi=roi, corrections=corr)
    if len(sel) == 0:
        assert np.all(np.isnan(res[0]['intensity'].data)), info
        assert np.all(res[1]['intensity'].data == 0), info
        return
    ref_m = opath.apply_masks(sel, masks, num_partitions=1)
    ref_s = opath.sum_udf(sel, num_partitions=1, dtype=sel.dtype if sel.dtype.kind == 'f' else 'float32')
    ref_ss = opath.sumsig_udf(sel, num_partitions=1)
    got_m, got_s, got_ss = (res[0]['intensity'], res[1]['intensity'], res[2]['intensity'])
    assert got_m.raw_data.dtype == ref_m.dtype and got_ss.raw_data.dtype == ref_ss.dtype, info
    assert _close(got_m.raw_data.reshape(ref_m.shape), ref_m), info
    assert _close(got_s.data, ref_s), info
    assert _close(got_ss.raw_data.reshape(-1), ref_ss.reshape(-1)), info
    if roi is not None:
        full = got_ss.data
        assert full.shape == nav and np.all(np.isnan(full[~roi])), info


def _same_nf(res, ref):
    parts = (np.real, np.imag) if np.iscomplexobj(ref) else (np.asarray,)
    return all(np.array_equal(np.isnan(f(res)), np.isnan(f(ref))) and
               np.array_equal(np.isposinf(f(res)), np.isposinf(f(ref))) and
               np.array_equal(np.isneginf(f(res)), np.isneginf(f(ref))) for f in parts)


@pytest.mark.parametrize('seed', range(int(__import__('os').environ.get('LTMI_FUZZ_SEEDS', '60'))))
def test_random_mask_kinds(ctx, seed):
    """ApplyMasksUDF over the combinations the reference accepts: dense / sparse stacks (scipy CSR / CSC, 'sparse.pydata'),
    mask dtypes (bool, int32, float32, float64, complex64, complex128), frame dtypes incl. complex64, constant or
    per-frame shifts, regions of interest, NaN pixels in float frames -- against the oracle's restatement of the reference
    path that the combination selects (dense: `flat_tile @ masks`; sparse: rmatmul, stored entries only; shifts: frame by
    frame): same result dtype, same NaN pattern, values within 1e-5 of what was added up."""
    import scipy.sparse as sp
    from libertem_amd.common.hiparray import HipArray
    from libertem_amd.udf.masks import ApplyMasksUDF
    rng = np.random.default_rng(7000 + seed)
    nav = tuple(int(x) for x in rng.integers(2, 7, 2))
    sig = (int(rng.choice([16, 24, 32])), int(rng.choice([16, 32, 64])))
    n = int(np.prod(nav))
    n_px = sig[0] * sig[1]
    dt = np.dtype(rng.choice(['uint8', 'uint16', 'int16', 'float32', 'float64', 'int32', 'complex64']))
    if dt.kind in 'iu':
        data = rng.integers(0, 200 if dt.itemsize == 1 else 3000, nav + sig).astype(dt)
    elif dt.kind == 'c':
        data = ((rng.random(nav + sig) - 0.3) + 1j * (rng.random(nav + sig) - 0.5)).astype(dt)
    else:
        data = ((rng.random(nav + sig) - 0.2) * 10).astype(dt)
    kind = str(rng.choice(['dense', 'scipy.sparse', 'scipy.sparse.csc', 'sparse.pydata']))
    md = np.dtype(rng.choice(['bool', 'int32', 'float32', 'float64', 'complex64', 'complex128']
                             if kind == 'dense' else ['float32', 'float64', 'complex64']))
    n_masks = int(rng.choice([1, 4, 19, 70]))
    fill = float(rng.choice([0.02, 0.3])) if kind != 'dense' else 1.0
    dense = rng.random((n_masks,) + sig) - 0.3
    if md.kind == 'c':
        dense = dense + 1j * (rng.random((n_masks,) + sig) - 0.5)
    if md.kind == 'b':
        dense = dense > 0.2
    elif md.kind == 'i':
        dense = np.round(dense * 8)
    if fill < 1.0:
        dense = np.where(rng.random((n_masks,) + sig) < fill, dense, 0)
    dense = dense.astype(md)
    shifts = None
    if kind == 'dense' and md.kind in 'fc' and dt.kind != 'c' and rng.random() < 0.25:
        shifts = (int(rng.integers(-3, 4)), int(rng.integers(-3, 4))) if rng.random() < 0.5 else \
            rng.integers(-2, 3, nav + (2,))
    nan_frames = dt.kind == 'f' and shifts is None and rng.random() < 0.4
    if nan_frames:
        flat = data.reshape((n, n_px))
        for _ in range(int(rng.integers(1, 4))):
            flat[int(rng.integers(0, n)), int(rng.integers(0, n_px))] = np.nan
    roi = (rng.random(nav) < 0.6) if (shifts is None and rng.random() < 0.3) else None
    num_partitions = int(rng.integers(1, 4))
    resident = str(rng.choice(['host', 'device']))
    src = HipArray.from_numpy(data, 0) if resident == 'device' else data
    ds = ctx.load('memory', data=src, num_partitions=num_partitions, sig_dims=2)
    if kind == 'dense':
        factories = (lambda: dense)
        kw = dict(use_sparse=False, mask_count=n_masks)
    else:
        mats = [sp.csr_matrix(dense[k]) for k in range(n_masks)]
        factories = [(lambda m=m: m) for m in mats]
        kw = dict(use_sparse=kind)
    if shifts is not None:
        kw['shifts'] = shifts if isinstance(shifts, tuple) else \
            ApplyMasksUDF.aux_data(shifts.reshape((-1, 2)), kind='nav', extra_shape=(2,), dtype=int)
        kw.pop('use_sparse', None)
    info = (seed, nav, sig, str(dt), kind, str(md), n_masks, fill, shifts is not None, nan_frames, roi is not None,
            num_partitions, resident)
    got = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=factories, mask_dtype=md, **kw), roi=roi)['intensity']
    sel = data.reshape((n,) + sig)
    if roi is not None:
        sel = sel[roi.reshape(-1)]
    if len(sel) == 0:
        return
    with np.errstate(invalid='ignore'):
        if shifts is not None:
            sh = np.broadcast_to(np.asarray(shifts), (n, 2)) if isinstance(shifts, tuple) else shifts.reshape((n, 2))
            ref = opath.apply_masks_shifted(sel[None], dense, sh.reshape((1, n, 2)))[0]
        elif kind == 'dense':
            ref = opath.apply_masks(sel[None], dense, mask_dtype=md)[0]
        else:
            stack = sp.csr_matrix(dense.reshape((n_masks, -1)))
            ref = opath.apply_masks_sparse(sel[None], stack, mask_dtype=md,
                                           fmt='csc' if kind.endswith('csc') else 'csr')[0]
    res = got.raw_data.reshape(ref.shape)
    assert res.dtype == ref.dtype, info + (res.dtype, ref.dtype)
    if nan_frames:
        assert _same_nf(res, ref), info
    else:
        assert np.all(np.isfinite(res)), info
    wide = np.complex128 if (np.iscomplexobj(sel) or np.iscomplexobj(dense)) else np.float64
    clean = np.where(np.isfinite(sel), sel, 0).reshape((len(sel), -1))
    ok = np.isfinite(ref)
    if ref.dtype.kind in 'iu':
        assert np.array_equal(res, ref), info
    else:
        bound = 1e-5 * (np.abs(clean).astype(np.float64) @ np.abs(dense.reshape((n_masks, -1))).astype(np.float64).T)
        if shifts is not None:
            bound = bound + 1e-5 * np.abs(ref).max()
        tol = 1.0 if ref.dtype in (np.float32, np.complex64) else 1e-6
        assert np.all(np.abs(res[ok] - ref[ok].astype(wide)) <= tol * bound[ok] + 1e-30), \
            info + (float(np.abs(res[ok] - ref[ok]).max()),)


@pytest.mark.parametrize('seed', range(int(__import__('os').environ.get('LTMI_FUZZ_SEEDS', '30'))))
def test_random_analyses(ctx, seed):
    """COMAnalysis / CoMUDF / RadialFourierAnalysis with random geometry (centre off the detector centre, disks, annuli,
    scan rotation, flip, bins, orders, sparse or dense) on random frame dtypes and scan shapes, against the oracle's
    restatement of the reference (analysis/com.py:191-334, udf/com.py:298-717, analysis/radialfourier.py:106-354)."""
    from libertem_amd.common.hiparray import HipArray
    from libertem_amd.udf.com import CoMUDF
    rng = np.random.default_rng(9000 + seed)
    nav = (int(rng.integers(2, 6)), int(rng.integers(2, 6)))
    sig = (int(rng.choice([32, 48, 64])), int(rng.choice([32, 64])))
    dt = np.dtype(rng.choice(['uint8', 'uint16', 'float32']))
    if dt.kind == 'u':
        data = rng.integers(1, 200 if dt.itemsize == 1 else 3000, nav + sig).astype(dt)
    else:
        data = (rng.random(nav + sig) + 0.1).astype(dt)
    parts = int(rng.integers(1, 4))
    src = HipArray.from_numpy(data, 0) if rng.random() < 0.5 else data
    ds = ctx.load('memory', data=src, num_partitions=parts, sig_dims=2)
    which = str(rng.choice(['com_analysis', 'com_udf', 'radial_fourier']))
    info = (seed, which, nav, sig, str(dt), parts)
    cy = float(sig[0] / 2 + rng.integers(-4, 5) + rng.choice([0., 0.5]))
    cx = float(sig[1] / 2 + rng.integers(-4, 5) + rng.choice([0., 0.5]))
    tol = 1e-5
    if which == 'com_analysis':
        kw = dict(cx=cx, cy=cy, scan_rotation=float(rng.choice([0., 33., -90.])), flip_y=bool(rng.random() < 0.3))
        okw = dict(kw)
        if rng.random() < 0.6:
            kw['mask_radius'] = okw['r'] = float(rng.integers(8, 20))
            if rng.random() < 0.4:
                kw['mask_radius_inner'] = okw['ri'] = float(rng.integers(2, 6))
        res = ctx.run(ctx.create_com_analysis(dataset=ds, **kw))
        ref = opath.com_analysis(data, num_partitions=parts, **okw)
        scale = float(max(sig))
        for k in ('x', 'y', 'magnitude', 'divergence', 'curl'):
            if k in ref:
                got = getattr(res, k).raw_data
                assert np.allclose(got, ref[k], rtol=tol, atol=4 * tol * scale), info + (k, kw)
    elif which == 'com_udf':
        kw = dict(cy=cy, cx=cx, r=float(rng.integers(8, 20)), scan_rotation=float(rng.choice([0., 45.])),
                  flip_y=bool(rng.random() < 0.3), regression=int(rng.choice([-1, 0, 1])))
        if rng.random() < 0.3:
            kw['ri'] = float(rng.integers(2, 6))
        res = ctx.run_udf(dataset=ds, udf=CoMUDF.with_params(**kw))
        ref = opath.com_udf(data, num_partitions=parts, **kw)
        scale = float(max(sig))
        for k in ('raw_com', 'raw_shifts', 'field', 'magnitude', 'divergence', 'curl', 'regression'):
            got = res[k].data
            assert got.shape == ref[k].shape, info + (k,)
            assert np.allclose(got, ref[k], rtol=tol, atol=4 * tol * scale), info + (k, kw,
                                                                                   float(np.abs(got - ref[k]).max()))
    else:
        kw = dict(cx=cx, cy=cy, n_bins=int(rng.choice([1, 2, 5])), max_order=int(rng.choice([3, 8, 24])))
        if rng.random() < 0.5:
            kw['ri'], kw['ro'] = float(rng.integers(0, 4)), float(rng.integers(10, 16))
        if rng.random() < 0.5:
            kw['use_sparse'] = bool(rng.random() < 0.5)
        res = ctx.run(ctx.create_radial_fourier_analysis(dataset=ds, **kw))
        ref = opath.radial_fourier_analysis(data, num_partitions=parts, **kw)
        got = res.raw_results
        assert got.shape == ref['raw_results'].shape and got.dtype == ref['raw_results'].dtype, info + (kw,)
        scale = np.abs(ref['raw_results']).max()
        assert np.allclose(got, ref['raw_results'], rtol=tol, atol=tol * scale), info + (kw,)
