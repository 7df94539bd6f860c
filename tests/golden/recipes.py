"""
Seeded input recipes shared by `generate_golden.py` (runs the real reference) and the
parity tests (run the oracle / the HIP path on the same inputs).  Data only, no reference code.

Value ranges are chosen like the reference's `_mk_random` idea (tests/utils.py:48-78):
mostly small values plus a few large outliers, so that errors do not average out.
"""
import numpy as np


def _outliers(rng, arr, n=4, val=None):
    flat = arr.reshape(-1)
    idx = rng.integers(0, flat.size, n)
    if val is None:
        if arr.dtype.kind in 'iu':
            val = min(np.iinfo(arr.dtype).max, 60000)
        else:
            val = 3.0e4
    flat[idx] = val
    return arr


# ---------------------------------------------------------------------------
# dense ApplyMasksUDF cases
# ---------------------------------------------------------------------------
DENSE_CASES = [
    # C2 shape on a reduced nav: real reference tile shape (32, 32, 256), 8 sig slices,
    # ragged last frame group (72 = 2*32 + 8), uneven partitions
    dict(name='c2_u16_16masks', nav=(3, 24), sig=(256, 256), dtype='uint16', n_masks=16,
         mask_dtype='float32', num_partitions=2, seed=101,
         udf_kwargs=dict(use_sparse=False, mask_count=16, mask_dtype=np.float32)),
    # C3 shape (CoM-like 3 masks) reduced nav; tiles (32, 16, 512)
    dict(name='c3_u16_3masks', nav=(2, 20), sig=(512, 512), dtype='uint16', n_masks=3,
         mask_dtype='float32', num_partitions=3, seed=102,
         udf_kwargs=dict(use_sparse=False)),
    # float32 data: whole-partition tiles
    dict(name='f32_5masks', nav=(5, 7), sig=(64, 48), dtype='float32', n_masks=5,
         mask_dtype='float32', num_partitions=4, seed=103, udf_kwargs={}),
    # odd sig shape + forced sub-frame tiling (tests/analysis/test_analysis_masks.py:141-148)
    dict(name='odd_tiles_f32', nav=(4, 9), sig=(17, 23), dtype='float32', n_masks=4,
         mask_dtype='float32', num_partitions=2, seed=104, tileshape=(8, 4, 23), udf_kwargs={}),
    dict(name='u8_2masks', nav=(4, 4), sig=(32, 32), dtype='uint8', n_masks=2,
         mask_dtype='float32', num_partitions=2, seed=105, udf_kwargs={}),
    dict(name='i16_bool_masks', nav=(4, 4), sig=(32, 32), dtype='int16', n_masks=3,
         mask_dtype='bool', num_partitions=2, seed=106, udf_kwargs={}),
    # int16 data x bool masks with dtype=int32 -> exact int32 (test_analysis_masks.py:568-597)
    dict(name='i16_int32_exact', nav=(4, 4), sig=(32, 32), dtype='int16', n_masks=3,
         mask_dtype='bool', num_partitions=2, seed=107,
         udf_kwargs=dict(preferred_dtype=np.int32, mask_dtype=np.int32)),
    # int32 / int64 data -> float64 (test_analysis_masks.py:499-519)
    dict(name='i32_f64', nav=(3, 5), sig=(24, 40), dtype='int32', n_masks=3,
         mask_dtype='float32', num_partitions=2, seed=108, udf_kwargs={}),
    dict(name='i64_f64', nav=(3, 5), sig=(24, 40), dtype='int64', n_masks=3,
         mask_dtype='float32', num_partitions=2, seed=109, udf_kwargs={}),
    dict(name='f64_f64masks', nav=(3, 5), sig=(24, 40), dtype='float64', n_masks=3,
         mask_dtype='float64', num_partitions=2, seed=110, udf_kwargs={}),
    # float64 masks forced down (test_analysis_masks.py:625-646)
    dict(name='u16_f64masks_forced_f32', nav=(3, 5), sig=(24, 40), dtype='uint16', n_masks=3,
         mask_dtype='float64', num_partitions=2, seed=111,
         udf_kwargs=dict(mask_dtype=np.float32)),
    # complex masks on real data (radial-Fourier style) and complex data
    dict(name='u16_c64masks', nav=(3, 5), sig=(24, 40), dtype='uint16', n_masks=4,
         mask_dtype='complex64', num_partitions=2, seed=112, udf_kwargs={}),
    dict(name='c64_data', nav=(3, 5), sig=(24, 40), dtype='complex64', n_masks=3,
         mask_dtype='float32', num_partitions=2, seed=113, udf_kwargs={}),
    # many masks (more than one 16-column group, not a multiple of 16)
    dict(name='u16_37masks', nav=(2, 9), sig=(64, 64), dtype='uint16', n_masks=37,
         mask_dtype='float32', num_partitions=2, seed=114, udf_kwargs={}),
    # single frame / single partition edge
    dict(name='single_frame', nav=(1, 1), sig=(32, 32), dtype='uint16', n_masks=2,
         mask_dtype='float32', num_partitions=1, seed=115, udf_kwargs={}),
    # round 2: the remaining rows of the reference's dtype rule (np.result_type of input and masks)
    # float64 masks -- NumPy's default -- on uint16 frames without mask_dtype: float64 results
    dict(name='u16_f64masks', nav=(3, 5), sig=(24, 40), dtype='uint16', n_masks=3,
         mask_dtype='float64', num_partitions=2, seed=116, udf_kwargs={}),
    # complex64 masks on int32 frames (radial Fourier on a counting detector): complex128
    dict(name='i32_c64masks', nav=(3, 5), sig=(24, 40), dtype='int32', n_masks=4,
         mask_dtype='complex64', num_partitions=2, seed=117, udf_kwargs={}),
    # complex frames x complex masks
    dict(name='c64_data_c64masks', nav=(3, 5), sig=(24, 40), dtype='complex64', n_masks=3,
         mask_dtype='complex64', num_partitions=2, seed=118, udf_kwargs={}),
    dict(name='u32_f32masks', nav=(3, 5), sig=(24, 40), dtype='uint32', n_masks=3,
         mask_dtype='float32', num_partitions=2, seed=119, udf_kwargs={}),
]


def make_dense_case(case):
    rng = np.random.default_rng(case['seed'])
    shape = tuple(case['nav']) + tuple(case['sig'])
    dt = np.dtype(case['dtype'])
    if dt.kind == 'u':
        data = rng.integers(0, 4096 if dt.itemsize > 1 else 200, shape).astype(dt)
        _outliers(rng, data)
    elif dt.kind == 'i':
        data = rng.integers(-2000, 2000, shape).astype(dt)
        _outliers(rng, data, val=30000)
    elif dt.kind == 'f':
        data = rng.random(shape).astype(dt)
        _outliers(rng, data)
    else:
        data = (rng.random(shape) + 1j * rng.random(shape)).astype(dt)
    mdt = np.dtype(case['mask_dtype'])
    mshape = (case['n_masks'],) + tuple(case['sig'])
    if mdt.kind == 'b':
        masks = rng.random(mshape) > 0.5
    elif mdt.kind == 'c':
        masks = (rng.random(mshape) - 0.5 + 1j * (rng.random(mshape) - 0.5)).astype(mdt)
    else:
        masks = (rng.random(mshape) - 0.25).astype(mdt)
    return data, masks


# ---------------------------------------------------------------------------
# Sum / SumSig
# ---------------------------------------------------------------------------
SUM_CASES = [
    # config C1 sig shape, reduced nav
    dict(name='c1_f32', nav=(4, 8), sig=(128, 128), dtype='float32', num_partitions=4, seed=201),
    dict(name='u16', nav=(5, 7), sig=(64, 48), dtype='uint16', num_partitions=3, seed=202),
    # tests/udf/test_sum.py:12-36
    dict(name='odd_tiles', nav=(16, 8), sig=(17, 23), dtype='float32', num_partitions=2,
         seed=203, tileshape=(8, 17, 23)),
    dict(name='u8', nav=(3, 3), sig=(32, 32), dtype='uint8', num_partitions=2, seed=204),
    dict(name='f64', nav=(3, 3), sig=(32, 32), dtype='float64', num_partitions=2, seed=205),
    dict(name='i32', nav=(3, 3), sig=(32, 32), dtype='int32', num_partitions=2, seed=206),
    # complex frames keep their dtype (reference tests/analysis/test_analysis_sum.py:157-163
    # `test_sum_complex`; udf/sum.py:38-40, udf/sumsigudf.py:23)
    dict(name='c64', nav=(4, 5), sig=(32, 24), dtype='complex64', num_partitions=3, seed=207),
    dict(name='c128', nav=(3, 3), sig=(16, 16), dtype='complex128', num_partitions=2, seed=208),
    dict(name='c64_tiles', nav=(6, 4), sig=(16, 32), dtype='complex64', num_partitions=2, seed=209,
         tileshape=(5, 8, 32)),
    dict(name='c64_as_c128', nav=(3, 4), sig=(16, 16), dtype='complex64', num_partitions=2,
         seed=210, sum_kwargs=dict(dtype='float64')),
    # SumUDF(dtype=<integer>) on integer frames: integer result, NumPy wrap-around
    dict(name='i16_as_i32', nav=(4, 4), sig=(32, 32), dtype='int16', num_partitions=2, seed=211,
         sum_kwargs=dict(dtype='int32')),
    dict(name='u8_wraps', nav=(5, 5), sig=(16, 16), dtype='uint8', num_partitions=2, seed=212,
         sum_kwargs=dict(dtype='uint8')),
    dict(name='u16_as_i64', nav=(3, 5), sig=(24, 16), dtype='uint16', num_partitions=3, seed=213,
         sum_kwargs=dict(dtype='int64'), tileshape=(4, 8, 16)),
]


def make_sum_case(case):
    rng = np.random.default_rng(case['seed'])
    shape = tuple(case['nav']) + tuple(case['sig'])
    dt = np.dtype(case['dtype'])
    if dt.kind == 'u':
        data = rng.integers(0, 200, shape).astype(dt)
    elif dt.kind == 'i':
        data = rng.integers(-2000, 2000, shape).astype(dt)
    elif dt.kind == 'c':
        data = (rng.random(shape) - 0.5 + 1j * (rng.random(shape) - 0.25)).astype(dt)
    else:
        data = rng.random(shape).astype(dt)
    return data


# ---------------------------------------------------------------------------
# CoM
# ---------------------------------------------------------------------------
COM_CASES = [
    dict(name='default', nav=(8, 6), sig=(32, 32), dtype='uint16', num_partitions=2, seed=301,
         params=dict(), analysis_params=dict()),
    dict(name='disk', nav=(8, 6), sig=(32, 48), dtype='float32', num_partitions=3, seed=302,
         params=dict(cy=15.2, cx=22.8, r=9.),
         analysis_params=dict(cy=15.2, cx=22.8, r=9.)),
    dict(name='annular_rot_flip', nav=(7, 9), sig=(40, 40), dtype='uint16', num_partitions=2,
         seed=303,
         params=dict(cy=19., cx=21., r=15., ri=4., scan_rotation=33., flip_y=True),
         analysis_params=dict(cy=19., cx=21., r=15., ri=4., scan_rotation=33., flip_y=True)),
    dict(name='regression_lin', nav=(6, 6), sig=(32, 32), dtype='float32', num_partitions=2,
         seed=304, params=dict(regression=1), analysis_params=dict()),
    dict(name='regression_mean', nav=(6, 6), sig=(32, 32), dtype='float32', num_partitions=2,
         seed=305, params=dict(regression=0), analysis_params=dict()),
    # some all-zero frames (tests/analysis/test_analysis_com.py:36-52)
    dict(name='zero_frames', nav=(4, 4), sig=(16, 16), dtype='float32', num_partitions=2,
         seed=306, params=dict(), analysis_params=dict(), zero_frames=[0, 5, 15]),
]


def make_com_case(case):
    rng = np.random.default_rng(case['seed'])
    nav, sig = tuple(case['nav']), tuple(case['sig'])
    dt = np.dtype(case['dtype'])
    # a blob whose position moves with the scan position + noise
    yy, xx = np.mgrid[0:sig[0], 0:sig[1]]
    data = np.zeros(nav + sig, dtype=np.float64)
    for i in range(nav[0]):
        for j in range(nav[1]):
            cy = sig[0] / 2 + 3 * np.sin(i / 2.) + rng.normal() * 0.3
            cx = sig[1] / 2 + 2 * np.cos(j / 3.) + rng.normal() * 0.3
            data[i, j] = 1000 * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / 18.)
    data += rng.random(nav + sig) * 5
    for f in case.get('zero_frames', []):
        data.reshape((-1,) + sig)[f] = 0
    if dt.kind in 'iu':
        return np.round(data).astype(dt)
    return data.astype(dt)


# ---------------------------------------------------------------------------
# Radial Fourier
# ---------------------------------------------------------------------------
RF_CASES = [
    dict(name='dense_2x4', nav=(4, 4), sig=(64, 64), dtype='uint16', num_partitions=2, seed=401,
         params=dict(n_bins=2, max_order=4, use_sparse=False)),
    dict(name='dense_default_small', nav=(3, 3), sig=(48, 40), dtype='float32',
         num_partitions=2, seed=402, params=dict(use_sparse=False)),
    dict(name='offcenter', nav=(3, 3), sig=(48, 40), dtype='float32', num_partitions=2,
         seed=403, params=dict(cx=17.3, cy=25.9, ri=3., ro=14., n_bins=3, max_order=6,
                               use_sparse=False)),
    # parameter heuristics only (C5 / C4' shapes): (analysis/radialfourier.py:316-354)
    dict(name='heuristic_c5', nav=(1, 2), sig=(1024, 1024), dtype='float32', num_partitions=1,
         seed=404, params=dict()),
    dict(name='heuristic_16bins', nav=(1, 2), sig=(256, 256), dtype='float32',
         num_partitions=1, seed=405, params=dict(n_bins=16, max_order=24)),
]


def make_rf_case(case):
    rng = np.random.default_rng(case['seed'])
    shape = tuple(case['nav']) + tuple(case['sig'])
    dt = np.dtype(case['dtype'])
    if case['name'].startswith('heuristic'):
        return np.zeros(shape, dtype=dt)
    if dt.kind == 'u':
        return rng.integers(0, 1000, shape).astype(dt)
    return rng.random(shape).astype(dt)


# ---------------------------------------------------------------------------
# Config workloads at their real detector size (BASELINE.json C3 / C5), reduced nav
# ---------------------------------------------------------------------------
WORKLOAD_CASES = [
    # C5: RadialFourierAnalysis defaults (n_bins=1, max_order=24 -> 25 dense complex64 masks)
    # on 1024x1024 float32 frames, rng(5).random like SURVEY.md 8(d)
    dict(name='c5_rf_1024', kind='rf', nav=(2, 4), sig=(1024, 1024), dtype='float32',
         num_partitions=2, seed=5, params=dict()),
    # C3: COMAnalysis / CoMUDF on 512x512 uint16 counts in [0, 4096): default (r=inf) and
    # mask_radius=200, plus an off-centre CoMUDF
    dict(name='c3_com_512', kind='com', nav=(2, 4), sig=(512, 512), dtype='uint16',
         num_partitions=2, seed=3,
         analysis_params=[dict(cx=256, cy=256), dict(cx=256, cy=256, r=200)],
         udf_params=[dict(), dict(cy=250.5, cx=260.25, r=200.)]),
]


def make_workload_case(case):
    rng = np.random.default_rng(case['seed'])
    shape = tuple(case['nav']) + tuple(case['sig'])
    if case['kind'] == 'rf':
        return rng.random(shape, dtype=np.float32)
    data = rng.integers(0, 4096, shape).astype(case['dtype'])
    # a brighter disk that wanders with the scan position, so that the centre of mass is not
    # just the noise around the detector centre
    yy, xx = np.mgrid[0:shape[2], 0:shape[3]]
    flat = data.reshape((-1,) + shape[2:])
    for f in range(flat.shape[0]):
        cy, cx = shape[2] / 2 + 11.5 * np.sin(f), shape[3] / 2 + 17.25 * np.cos(f)
        disk = (yy - cy) ** 2 + (xx - cx) ** 2 < 60 ** 2
        flat[f][disk] = np.minimum(flat[f][disk].astype(np.int64) + 2000, 4095).astype(data.dtype)
    return data


# ---------------------------------------------------------------------------
# mask factories
# ---------------------------------------------------------------------------
CIRCULAR_CASES = [
    dict(centerX=16, centerY=16, imageSizeX=32, imageSizeY=32, radius=6),
    dict(centerX=10.5, centerY=20.25, imageSizeX=40, imageSizeY=24, radius=7.5),
    dict(centerX=5, centerY=5, imageSizeX=16, imageSizeY=16, radius=float('inf')),
    dict(centerX=16, centerY=16, imageSizeX=32, imageSizeY=32, radius=6, antialiased=True),
]
RING_CASES = [
    dict(centerX=16, centerY=16, imageSizeX=32, imageSizeY=32, radius=10, radius_inner=4),
    dict(centerX=10.5, centerY=20.25, imageSizeX=40, imageSizeY=24, radius=9.5, radius_inner=3.2),
    dict(centerX=16, centerY=16, imageSizeX=32, imageSizeY=32, radius=10, radius_inner=4,
         antialiased=True),
]
RADIAL_BINS_CASES = [
    dict(centerX=16, centerY=16, imageSizeX=32, imageSizeY=32, n_bins=4, dtype=np.float32),
    dict(centerX=10.5, centerY=20.25, imageSizeX=40, imageSizeY=24, radius=12.,
         radius_inner=2., n_bins=5, dtype=np.float32),
    dict(centerX=16, centerY=16, imageSizeX=32, imageSizeY=32, radius=8, n_bins=3,
         normalize=True, dtype=np.float64),
    dict(centerX=16, centerY=16, imageSizeX=32, imageSizeY=32, dtype=np.float32),
    dict(centerX=20, centerY=12, imageSizeX=40, imageSizeY=24, n_bins=64, dtype=np.float32),
]
POLAR_MAP_CASES = [
    dict(centerX=16, centerY=16, imageSizeX=32, imageSizeY=32),
    dict(centerX=10.5, centerY=20.25, imageSizeX=40, imageSizeY=24),
    dict(centerX=10.5, centerY=20.25, imageSizeX=40, imageSizeY=24, stretchY=1.3, angle=0.4),
]
GRADIENT_CASES = [(32, 32), (40, 24), (7, 13)]
BOUNDING_RADIUS_CASES = [(16, 16, 32, 32), (10.5, 20.25, 40, 24), (128, 128, 256, 256),
                         (512, 512, 1024, 1024)]
RADIAL_MASK_FACTORY_CASES = [
    dict(detector_y=32, detector_x=32, cx=16, cy=16, ri=0, ro=12, n_bins=2, max_order=3),
    dict(detector_y=24, detector_x=40, cx=17.3, cy=11.9, ri=2, ro=10, n_bins=3, max_order=5),
]
RECT_CASES = [
    dict(X=3, Y=4, Width=10, Height=6, imageSizeX=32, imageSizeY=24),
    dict(X=20, Y=15, Width=-8, Height=5, imageSizeX=32, imageSizeY=24),
    dict(X=20, Y=15, Width=8, Height=-5, imageSizeX=32, imageSizeY=24),
    dict(X=20, Y=15, Width=-8, Height=-5, imageSizeX=32, imageSizeY=24),
]
BGSUB_CASES = [
    dict(centerX=16, centerY=16, imageSizeX=32, imageSizeY=32, radius=10, radius_inner=5),
    dict(centerX=16, centerY=16, imageSizeX=32, imageSizeY=32, radius=10, radius_inner=5,
         antialiased=True),
]
RADIAL_GRADIENT_CASES = [
    dict(centerX=16, centerY=16, imageSizeX=32, imageSizeY=32, radius=10),
    dict(centerX=16, centerY=16, imageSizeX=32, imageSizeY=32, radius=10, antialiased=True),
]


# ---------------------------------------------------------------------------
# rmatmul
# ---------------------------------------------------------------------------
RMATMUL_CASES = [
    dict(name='f32', rows=13, k=97, cols=11, density=0.1, ldtype='float32', rdtype='float32',
         seed=501),
    dict(name='u16_f32', rows=32, k=256, cols=20, density=0.05, ldtype='uint16',
         rdtype='float32', seed=502),
    dict(name='f32_c64', rows=8, k=64, cols=6, density=0.2, ldtype='float32',
         rdtype='complex64', seed=503),
    dict(name='f64', rows=5, k=33, cols=4, density=0.3, ldtype='float64', rdtype='float64',
         seed=504),
    dict(name='empty_rows', rows=6, k=40, cols=5, density=0.02, ldtype='float32',
         rdtype='float32', seed=505),
]


def make_rmatmul_case(case):
    rng = np.random.default_rng(case['seed'])
    ld = np.dtype(case['ldtype'])
    if ld.kind == 'u':
        left = rng.integers(0, 4096, (case['rows'], case['k'])).astype(ld)
    else:
        left = rng.random((case['rows'], case['k'])).astype(ld)
    rd = np.dtype(case['rdtype'])
    dense = rng.random((case['k'], case['cols']))
    sel = rng.random((case['k'], case['cols'])) < case['density']
    if rd.kind == 'c':
        dense = dense + 1j * rng.random((case['k'], case['cols']))
    right = np.where(sel, dense, 0).astype(rd)
    return left, right


# ---------------------------------------------------------------------------
# non-finite pixels (round 6): which zeros of a stack meet a NaN / Inf pixel -- the reference's sparse loops touch
# stored entries only (common/numba/__init__.py:153-184), its dense product every weight (udf/masks.py:59-77)
# ---------------------------------------------------------------------------
NONFINITE_CASES = [
    dict(name='rings_f32', nav=(3, 5), sig=(32, 32), n_bins=12, mask_dtype='float32', seed=811),
    dict(name='rings_c64', nav=(2, 6), sig=(32, 48), n_bins=7, mask_dtype='complex64', seed=812),
]


def make_nonfinite_case(case):
    """-> (float32 frames with NaN / +Inf / -Inf pixels in a few frames, dense stack (n_masks, *sig) with many exact
    zeros): frame 1: NaN in a pixel no mask stores; frame 2: NaN in a stored pixel; frame 3: +Inf in a stored pixel;
    frame 4: +Inf and -Inf in two stored pixels; the last frame: NaN in both kinds"""
    rng = np.random.default_rng(case['seed'])
    sig = tuple(case['sig'])
    yy, xx = np.mgrid[0:sig[0], 0:sig[1]]
    r = np.hypot(yy - sig[0] / 2 + 0.25, xx - sig[1] / 2 - 0.5)
    edges = np.linspace(2.0, 0.45 * min(sig), case['n_bins'] + 1)
    stack = np.stack([((r >= a) & (r < b)) * (0.25 + rng.random(sig)) for a, b in zip(edges[:-1], edges[1:])])
    mdt = np.dtype(case['mask_dtype'])
    if mdt.kind == 'c':
        stack = stack * np.exp(1j * rng.random(stack.shape) * 6.0)
    stack = stack.astype(mdt)
    n = int(np.prod(case['nav']))
    data = (rng.random((n,) + sig) + 0.1).astype(np.float32)
    counts = (stack != 0).sum(axis=0).reshape(-1)
    stored, unstored = np.flatnonzero(counts > 0), np.flatnonzero(counts == 0)
    assert len(stored) > 20 and len(unstored) > 20 and n >= 6
    flat = data.reshape((n, -1))
    flat[1, unstored[len(unstored) // 2]] = np.nan
    flat[2, stored[len(stored) // 3]] = np.nan
    flat[3, stored[len(stored) // 2]] = np.inf
    flat[4, stored[len(stored) // 5]] = np.inf
    flat[4, stored[(2 * len(stored)) // 3]] = -np.inf
    flat[n - 1, unstored[0]] = np.nan
    flat[n - 1, stored[-1]] = np.nan
    return data.reshape(tuple(case['nav']) + sig), stack


# ---------------------------------------------------------------------------
# partitioning / tiling negotiation
# ---------------------------------------------------------------------------
TILING_CASES = [
    dict(name='c1', shape=(32, 32, 128, 128), dtype='float32', num_partitions=4, udf='sum'),
    dict(name='c2', shape=(16, 256, 256, 256), dtype='uint16', num_partitions=8, udf='masks'),
    dict(name='c2_ragged', shape=(3, 24, 256, 256), dtype='uint16', num_partitions=2,
         udf='masks'),
    dict(name='c3', shape=(4, 128, 512, 512), dtype='uint16', num_partitions=4, udf='masks',
         n_masks=3),
    dict(name='c5', shape=(2, 16, 1024, 1024), dtype='float32', num_partitions=4, udf='masks',
         n_masks=25),
    dict(name='u8_64', shape=(8, 64, 64, 64), dtype='uint8', num_partitions=3, udf='masks'),
    dict(name='odd_sig', shape=(5, 7, 17, 23), dtype='uint16', num_partitions=2, udf='masks',
         n_masks=2),
    dict(name='forced', shape=(4, 9, 17, 23), dtype='float32', num_partitions=2, udf='masks',
         n_masks=2, tileshape=(8, 4, 23)),
    dict(name='more_parts_than_frames', shape=(1, 3, 16, 16), dtype='float32',
         num_partitions=3, udf='sum'),
]


def single_mask_analyses():
    masks2 = [
        lambda: np.ones((32, 32), dtype=np.float32),
        lambda: np.arange(32 * 32, dtype=np.float64).reshape((32, 32)) / 1024.,
    ]
    return [
        ('disk_default', 'disk', {}),
        ('disk_params', 'disk', dict(cx=10, cy=20, r=5)),
        ('ring_default', 'ring', {}),
        ('ring_params', 'ring', dict(cx=10, cy=20, ri=3, ro=8)),
        ('masks_two', 'masks', dict(factories=masks2)),
        ('masks_two_f32', 'masks', dict(factories=masks2, mask_dtype=np.float32)),
    ]


# ---------------------------------------------------------------------------
# shifted masks (ApplyMasksUDF(shifts=...), reference udf/masks.py:85-124)
# ---------------------------------------------------------------------------
SHIFT_CASES = [
    dict(name='const', nav=(4, 5), sig=(24, 32), dtype='uint16', n_masks=3, num_partitions=2,
         seed=601, shifts=(2, -5)),
    dict(name='const_big', nav=(3, 3), sig=(16, 16), dtype='float32', n_masks=2,
         num_partitions=1, seed=602, shifts=(-20, 3)),           # no overlap at all
    dict(name='per_frame', nav=(4, 5), sig=(24, 32), dtype='uint16', n_masks=3,
         num_partitions=3, seed=603, shifts='aux'),
]


def make_shift_case(case):
    rng = np.random.default_rng(case['seed'])
    shape = tuple(case['nav']) + tuple(case['sig'])
    dt = np.dtype(case['dtype'])
    if dt.kind == 'u':
        data = rng.integers(0, 1000, shape).astype(dt)
    else:
        data = rng.random(shape).astype(dt)
    masks = (rng.random((case['n_masks'],) + tuple(case['sig'])) - 0.25).astype(np.float32)
    if case['shifts'] == 'aux':
        shifts = rng.integers(-8, 9, tuple(case['nav']) + (2,))
    else:
        shifts = np.array(case['shifts'])
    return data, masks, shifts


# ---------------------------------------------------------------------------
# detector corrections (reference io/corrections: CorrectionSet, detector.correct,
# RepairDescriptor, correct_dot_masks, tile-shape adjustment)
# ---------------------------------------------------------------------------
CORR_CASES = [
    dict(name='u16_all', nav=(3, 5), sig=(16, 24), dtype='uint16', num_partitions=2, seed=701,
         dark=True, gain=True, excluded=[(0, 0), (5, 7), (5, 8), (15, 23), (9, 0)]),
    dict(name='u16_dark_only', nav=(2, 4), sig=(16, 16), dtype='uint16', num_partitions=1,
         seed=702, dark=True, gain=False, excluded=None),
    dict(name='f32_gain_excl', nav=(7,), sig=(12, 20), dtype='float32', num_partitions=3,
         seed=703, dark=False, gain=True, excluded=[(3, 3), (4, 4), (11, 19)]),
    dict(name='i32_all', nav=(2, 3), sig=(8, 8), dtype='int32', num_partitions=1, seed=704,
         dark=True, gain=True, excluded=[(2, 2)]),
    dict(name='u8_excl_only', nav=(4, 4), sig=(16, 16), dtype='uint8', num_partitions=2,
         seed=705, dark=False, gain=False, excluded=[(7, 7), (7, 8), (8, 7), (8, 8)]),
]


def make_corr_case(case):
    """-> data, dark (float64 | None), gain (float64 | None), excluded coords (ndim, k) | None,
    masks (3, *sig) float32"""
    rng = np.random.default_rng(case['seed'])
    shape = tuple(case['nav']) + tuple(case['sig'])
    dt = np.dtype(case['dtype'])
    if dt.kind in 'ui':
        data = rng.integers(0, 200 if dt.itemsize == 1 else 3000, shape).astype(dt)
    else:
        data = (rng.random(shape) * 100).astype(dt)
    dark = rng.random(case['sig']) * 10 if case['dark'] else None
    gain = rng.random(case['sig']) + 0.5 if case['gain'] else None
    excluded = None
    if case['excluded'] is not None:
        excluded = np.array(case['excluded'], dtype=np.int64).T.reshape((len(case['sig']), -1))
    masks = (rng.random((3,) + tuple(case['sig'])) - 0.25).astype(np.float32)
    return data, dark, gain, excluded, masks


# RepairDescriptor tables: (sig_shape, excluded coords as list of tuples)
REPAIR_CASES = [
    ((16, 24), [(0, 0), (5, 7), (5, 8), (15, 23), (9, 0)]),
    ((19,), [(1,), (2,), (3,), (18,)]),
    ((5, 6, 7), [(2, 3, 4), (0, 0, 0), (4, 5, 6), (2, 3, 5)]),
    ((8, 8), []),
]

# tile-shape adjustment: (tile_shape, sig_shape, base_shape, excluded coords (ndim, k))
# -- the known-answer cases of the reference's tests/corrections/test_corrset.py:140-380
ADJUST_CASES = [
    ((1, 1), (123, 456), (1, 1), [[3], [8]]),
    ((7, 1), (123, 456), (1, 1), [[8], [3]]),
    ((2, 2), (123, 456), (2, 2), [[3, 5], [8, 9]]),
    ((2, 1), (123, 456), (2, 1), [[122], [455]]),
    ((123, 1), (123, 456), (1, 1), [[3], [8]]),
    ((1, 1), (123, 456), (1, 1), [[0, 1, 2], [0, 1, 2]]),
    ((8, 1), (1024, 1024), (8, 1), [[7, 8, 16, 24], [5, 6, 7, 8]]),
    ((8, 8), (64, 64), (8, 8), [list(range(0, 64, 2)), list(range(0, 64, 2))]),
    ((16, 16), (128, 128), (4, 4), [[17, 33, 95], [3, 64, 127]]),
    ((3, 5), (30, 50), (3, 5), [[2, 3, 29], [4, 5, 49]]),
]


# ---------------------------------------------------------------------------
# CrystallinityUDF (reference udf/crystallinity.py)
# ---------------------------------------------------------------------------
CRYST_CASES = [
    dict(name='u16_masked', nav=(4, 5), sig=(32, 32), dtype='uint16', num_partitions=2, seed=801,
         rad_in=4, rad_out=9, real_center=(16, 16), real_rad=5),
    dict(name='f32_plain', nav=(6,), sig=(24, 40), dtype='float32', num_partitions=1, seed=802,
         rad_in=3, rad_out=8, real_center=None, real_rad=None),
    dict(name='u8_odd', nav=(3, 3), sig=(33, 31), dtype='uint8', num_partitions=3, seed=803,
         rad_in=2.5, rad_out=11.5, real_center=(10.5, 20.25), real_rad=4),
    dict(name='u16_64', nav=(2, 8), sig=(64, 64), dtype='uint16', num_partitions=1, seed=804,
         rad_in=6, rad_out=20, real_center=(32, 32), real_rad=8),
    # the frame shapes with hand-written transform kernels (csrc/ltmi_cryst.hip)
    dict(name='u16_128', nav=(5,), sig=(128, 128), dtype='uint16', num_partitions=1, seed=805,
         rad_in=8, rad_out=32, real_center=(64, 64), real_rad=12),
    dict(name='u16_256', nav=(2, 3), sig=(256, 256), dtype='uint16', num_partitions=2, seed=806,
         rad_in=16, rad_out=64, real_center=(128, 128), real_rad=25),
    dict(name='f32_256_plain', nav=(4,), sig=(256, 256), dtype='float32', num_partitions=1, seed=807,
         rad_in=10, rad_out=47.5, real_center=None, real_rad=None),
    dict(name='u16_512', nav=(3,), sig=(512, 512), dtype='uint16', num_partitions=1, seed=808,
         rad_in=32, rad_out=128, real_center=(256, 256), real_rad=50),
    dict(name='u8_1024', nav=(2,), sig=(1024, 1024), dtype='uint8', num_partitions=1, seed=809,
         rad_in=64, rad_out=256, real_center=None, real_rad=None),
    dict(name='u16_256x512', nav=(3,), sig=(256, 512), dtype='uint16', num_partitions=1, seed=810,
         rad_in=16, rad_out=90, real_center=(128, 256), real_rad=30),
    dict(name='f32_512x256_wide', nav=(2,), sig=(512, 256), dtype='float32', num_partitions=1, seed=811,
         rad_in=40, rad_out=300, real_center=None, real_rad=None),
]


def make_cryst_case(case):
    rng = np.random.default_rng(case['seed'])
    shape = tuple(case['nav']) + tuple(case['sig'])
    dt = np.dtype(case['dtype'])
    if dt.kind == 'u':
        base = rng.integers(0, 200 if dt.itemsize == 1 else 2000, shape)
    else:
        base = rng.random(shape) * 50
    # a lattice so that the ring actually catches peaks
    yy, xx = np.mgrid[0:case['sig'][0], 0:case['sig'][1]]
    lattice = 40 * (1 + np.cos(2 * np.pi * xx / 4.0)) * (1 + np.cos(2 * np.pi * yy / 5.0))
    return (base + lattice).astype(dt)


# ---- byte-order decoders (reference io/dataset/base/decode.py; its test tests/io/test_decode_swap.py:166-245)
DECODE_PAIRS = [
    ('uint8', 'uint16'), ('uint8', 'uint32'), ('uint8', 'uint64'), ('uint16', 'uint16'),
    ('uint16', 'uint32'), ('uint16', 'uint64'), ('uint32', 'uint32'), ('uint32', 'uint64'),
    ('uint8', 'float32'), ('uint16', 'float32'), ('uint32', 'float32'),
    ('uint8', 'float64'), ('uint16', 'float64'), ('uint32', 'float64'),
    ('uint64', 'uint64'),
]   # unsigned only, as in the reference's test: its decoders compose UNSIGNED words and rely on numba's
#     wrap-around when storing them into signed outputs, which the identity-njit stand-in cannot show
DECODE_CASES = [dict(name=f"{a}_to_{b}_{'be' if order == '>' else 'le'}", in_dtype=a, out_dtype=b,
                     order=order, seed=900 + i * 2 + (order == '>'), shape=(1, 16, 16))
                for i, (a, b) in enumerate(DECODE_PAIRS) for order in ('<', '>')]


# Signed integers in the other byte order: the reference composes the UNSIGNED word and stores it into
# the read dtype (decode.py:15-66).  Same-width signed outputs wrap back to the signed value, but wider
# integer and float outputs get the unsigned value (-1 stored big-endian reads as 65535.0).  Pinned in
# its own fixture (decode_signed.npz) with the decoders run as plain Python.
# (Same-width signed outputs -- int16 -> int16 ... -- cannot be run without numba: `out[i] = word`
# raises OverflowError in plain NumPy where compiled code wraps around to the signed value.)
DECODE_SIGNED_PAIRS = [
    ('int16', 'int32'), ('int16', 'float32'), ('int16', 'float64'),
    ('int32', 'int64'), ('int32', 'float64'),
]
DECODE_SIGNED_CASES = [
    dict(name=f"{a}_to_{b}_{'be' if order == '>' else 'le'}", in_dtype=a, out_dtype=b, order=order,
         seed=980 + i * 2 + (order == '>'), shape=(1, 16, 16))
    for i, (a, b) in enumerate(DECODE_SIGNED_PAIRS) for order in ('<', '>')]


def make_decode_case(case):
    """values (native order) and the raw bytes as they sit in a file of byte order case['order']"""
    rng = np.random.default_rng(case['seed'])
    dt = np.dtype(case['in_dtype'])
    info = np.iinfo(dt)
    vals = rng.integers(info.min, info.max, size=case['shape'], dtype=dt, endpoint=True)
    stored = vals.astype(dt.newbyteorder(case['order']))
    return vals, stored.reshape(-1).view(np.uint8).copy()


# ---- PickUDF / PickFrameAnalysis / PickFFTFrameAnalysis (udf/raw.py, analysis/raw.py, analysis/rawfft.py)
PICK_CASES = [
    dict(name='u16_2d', nav=(5, 6), sig=(16, 16), dtype='uint16', num_partitions=3, seed=950,
         roi_frames=[(1, 2), (4, 5), (0, 0)], pick=dict(x=2, y=1), real=dict(rad=3, cx=8, cy=8)),
    dict(name='f32_1d', nav=(9,), sig=(12, 20), dtype='float32', num_partitions=2, seed=951,
         roi_frames=[(7,)], pick=dict(x=7), real=None),
    dict(name='c64_3d', nav=(2, 3, 4), sig=(8, 8), dtype='complex64', num_partitions=4, seed=952,
         roi_frames=[(1, 2, 3), (0, 0, 1)], pick=dict(x=3, y=2, z=1), real=None),
]


def make_pick_case(case):
    rng = np.random.default_rng(case['seed'])
    shape = tuple(case['nav']) + tuple(case['sig'])
    dt = np.dtype(case['dtype'])
    if dt.kind == 'c':
        return (rng.random(shape) + 1j * rng.random(shape)).astype(dt)
    if dt.kind == 'f':
        return rng.random(shape).astype(dt)
    return rng.integers(0, 1000, shape).astype(dt)


# ---------------------------------------------------------------------------
# Merlin / Medipix .mib files (reference io/dataset/mib.py): per-frame ASCII header + payload.
# kind 'u': big-endian unsigned integers; kind 'r' ("R64"): 64-bit words whose bytes are stored most
# significant first -- 64 one-bit, 8 six-bit (one byte each) or 4 twelve-bit (two bytes each) pixels
# per word, first pixel in the LEAST significant position; 24 bit = two 12-bit images, high half first.
# Quad (2x2 chips): a raw row is [chip 4 | chip 3 | chip 2 | chip 1], chips 3 and 4 rotated by 180 degrees.
# ---------------------------------------------------------------------------
MIB_CASES = [
    dict(name='u08', kind='u', bits=8, sig=(32, 64), frames=(5,), nav=(1, 5), seed=1301),
    dict(name='u16', kind='u', bits=16, sig=(32, 64), frames=(3, 2), nav=(1, 5), seed=1302),
    dict(name='u32', kind='u', bits=32, sig=(16, 64), frames=(4,), nav=(2, 2), seed=1303),
    # (1-bit files: the reference needs file size % pixels per frame == 0, base/file.py:121-127)
    dict(name='r1', kind='r', bits=1, sig=(32, 128), frames=(32,), nav=(4, 8), seed=1304),
    dict(name='r6', kind='r', bits=6, sig=(32, 64), frames=(3, 3), nav=(2, 3), seed=1305),
    dict(name='r12', kind='r', bits=12, sig=(32, 64), frames=(6,), nav=(2, 3), seed=1306),
    # (24 bit: one frame per file -- the reference's read ranges step over later frames of a file by the
    #  size of ONE 12-bit image, mib.py:224-252, so only single-frame files read back correctly there)
    dict(name='r24', kind='r', bits=24, sig=(32, 64), frames=(1, 1, 1, 1), nav=(2, 2), seed=1307),
    dict(name='r1_quad', kind='r', bits=1, sig=(128, 128), frames=(64,), nav=(8, 8), quad=True,
         seed=1308),
    dict(name='r6_quad', kind='r', bits=6, sig=(128, 128), frames=(2, 2), nav=(2, 2), quad=True,
         seed=1309),
    dict(name='r12_quad', kind='r', bits=12, sig=(128, 128), frames=(4,), nav=(2, 2), quad=True,
         seed=1310),
    # fewer frames in the files than the scan has positions + a sync offset
    dict(name='r12_offset', kind='r', bits=12, sig=(32, 64), frames=(7,), nav=(2, 3), seed=1311,
         sync_offset=2),
    dict(name='u16_neg_offset', kind='u', bits=16, sig=(32, 64), frames=(5,), nav=(2, 3), seed=1312,
         sync_offset=-2),
]


def _mib_words(px, per_word):
    """(rows, W) pixel values -> (rows, W) with every group of `per_word` pixels reversed (the first
    pixel of a word sits in its least significant position, the word is stored big-endian)"""
    rows, w = px.shape
    return px.reshape(rows, w // per_word, per_word)[:, :, ::-1].reshape(rows, w)


def mib_encode_rows(px, bits):
    """(rows, W) pixel values -> (rows, bytes) payload of raw ('R64') data"""
    if bits == 1:
        return np.packbits(_mib_words(px.astype(np.uint8) & 1, 64), axis=1, bitorder='big')
    if bits == 6:
        return _mib_words(px.astype(np.uint8), 8)
    if bits == 12:
        return _mib_words(px.astype(np.uint16), 4).astype('>u2').view(np.uint8)
    raise ValueError(bits)


def mib_frame_payload(frame, case):
    kind, bits = case['kind'], case['bits']
    if kind == 'u':
        return frame.astype(f'>u{bits // 8}').tobytes()
    if bits == 24:
        hi = mib_encode_rows((frame >> 12).astype(np.uint16), 12)
        lo = mib_encode_rows((frame & 0xFFF).astype(np.uint16), 12)
        return hi.tobytes() + lo.tobytes()
    if case.get('quad'):
        h, w = frame.shape
        q1, q2 = frame[:h // 2, :w // 2], frame[:h // 2, w // 2:]
        q3, q4 = frame[h // 2:, :w // 2][::-1, ::-1], frame[h // 2:, w // 2:][::-1, ::-1]
        return np.concatenate([mib_encode_rows(np.ascontiguousarray(q), bits)
                               for q in (q4, q3, q2, q1)], axis=1).tobytes()
    return mib_encode_rows(frame, bits).tobytes()


def mib_header(case, seq):
    """per-frame header as the Merlin software writes it (384 bytes per chip, comma separated)"""
    h, w = case['sig']
    quad = bool(case.get('quad'))
    n_chips = 4 if quad else 1
    size = 384 * n_chips
    if case['kind'] == 'u':
        dt = 'U%02d' % case['bits']
        hw, hh = w, h
    else:
        dt = 'R64'
        if quad:
            hw, hh = 2 * w, h // 2          # raw rows of all four chips side by side
        elif case['bits'] == 24:
            hw, hh = 2 * w, h               # two 12-bit images
        else:
            hw, hh = w, h
    layout = '   2x2' if quad else '   1x1'
    chips = '0F' if quad else '01'
    text = (f"MQ1,{seq:06d},{size:05d},{n_chips:02d},{hw:04d},{hh:04d},{dt},{layout},{chips},"
            f"2020-05-18 16:51:49.971626,0.000555,0,0,0,1.200000E+2,5.110000E+2,0.000000E+0,"
            f"0.000000E+0,3RX,175,511,000,000,125,255,MQ1A,2020-05-18T14:51:49.971626178Z,555000ns,"
            f"{case['bits']}")
    raw = text.encode('ascii') + b','
    assert len(raw) <= size
    return raw + b'\x00' * (size - len(raw))


def make_mib_case(case):
    """-> (frames (n, H, W) in file order, {file name: bytes}, hdr text)"""
    rng = np.random.default_rng(case['seed'])
    n = sum(case['frames'])
    bits = case['bits']
    dt = np.uint8 if bits <= 8 else (np.uint16 if bits <= 16 else np.uint32)
    frames = rng.integers(0, 2 ** bits, size=(n,) + tuple(case['sig']), dtype=np.uint64).astype(dt)
    frames[0].reshape(-1)[:3] = [2 ** bits - 1, 0, 2 ** bits - 1]
    files = {}
    seq = 1
    for i, cnt in enumerate(case['frames']):
        blob = b''.join(mib_header(case, seq + k) + mib_frame_payload(frames[seq - 1 + k], case)
                        for k in range(cnt))
        files[f"{case['name']}{i + 1:06d}.mib"] = blob
        seq += cnt
    nav = case['nav']
    hdr = ("HDR,\t\nTime and Date Stamp (day, mnth, yr, hr, min, s):\t18/05/2020 16:51:48\n"
           f"Frames in Acquisition (Number):\t{int(np.prod(nav))}\n"
           f"Frames per Trigger (Number):\t{nav[-1]}\nEnd\t")
    return frames, files, hdr
