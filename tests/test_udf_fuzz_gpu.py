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


@pytest.mark.parametrize('seed', range(40))
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
