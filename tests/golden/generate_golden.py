#!/usr/bin/env python
"""
Generate the golden vectors under tests/golden/*.npz by running the REAL Python
reference (LiberTEM, /root/reference/src) in this container.

* The reference cannot be imported as-is here because third-party packages are absent
  (opentelemetry, numba, sparse, sparseconverter, jsonschema, autopep8 -- SURVEY.md §8c).
  `tests/golden/refshim/` holds throw-away stand-ins for THOSE THIRD-PARTY packages only
  (no-op tracing, identity `njit`, NumPy-only sparseconverter); every line of LiberTEM
  itself that runs below is the unmodified reference.
* Inputs are produced by the seeded recipes in `tests/golden/recipes.py` (shared with the
  parity tests), so only outputs + an input checksum are stored.
* This script is skipped (exit 0) if /root/reference is absent (e.g. on the GPU box).

Usage:  python tests/golden/generate_golden.py
"""
import os
import sys
import hashlib
import json

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/src'

if not os.path.isdir(REF):
    print("reference not present, nothing to do")
    sys.exit(0)

sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, 'refshim'))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

import recipes  # noqa: E402

from libertem.udf.base import UDFRunner  # noqa: E402
from libertem.udf.masks import ApplyMasksUDF  # noqa: E402
from libertem.udf.sum import SumUDF  # noqa: E402
from libertem.udf.sumsigudf import SumSigUDF  # noqa: E402
from libertem.udf.com import (  # noqa: E402
    CoMUDF, center_shifts, apply_correction, divergence, curl_2d, magnitude,
)
from libertem.io.dataset.memory import MemoryDataSet  # noqa: E402
from libertem.executor.inline import InlineJobExecutor  # noqa: E402
from libertem.analysis.com import COMAnalysis  # noqa: E402
from libertem.analysis.radialfourier import RadialFourierAnalysis, radial_mask_factory  # noqa: E402
from libertem.analysis.sum import SumAnalysis  # noqa: E402
from libertem.analysis.masks import MasksAnalysis  # noqa: E402
from libertem.analysis.disk import DiskMaskAnalysis  # noqa: E402
from libertem.analysis.ring import RingMaskAnalysis  # noqa: E402
from libertem.analysis.point import PointMaskAnalysis  # noqa: E402
from libertem.common.numba import rmatmul  # noqa: E402
from libertem.corrections import coordinates  # noqa: E402
from libertem import masks as ref_masks  # noqa: E402


EX = InlineJobExecutor(inline_threads=1)
MANIFEST = {}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run(ds, udf, roi=None):
    res = UDFRunner([udf]).run_for_dataset(ds, EX, roi=roi)
    return res.buffers[0]


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrays)
    MANIFEST[name] = {k: [list(np.shape(v)), str(np.asarray(v).dtype)] for k, v in arrays.items()}
    print(f"wrote {path} ({os.path.getsize(path)} bytes)")


# ---------------------------------------------------------------------------
# 1. ApplyMasksUDF, dense masks: dtype matrix, tile shapes, partition counts
# ---------------------------------------------------------------------------
def gen_apply_masks_dense():
    out = {}
    for case in recipes.DENSE_CASES:
        data, masks = recipes.make_dense_case(case)
        ds = MemoryDataSet(
            data=data, num_partitions=case['num_partitions'], sig_dims=2,
            tileshape=case.get('tileshape'),
        )
        kwargs = dict(case.get('udf_kwargs', {}))
        udf = ApplyMasksUDF(mask_factories=lambda: masks, **kwargs)
        res = run(ds, udf)['intensity']
        arr = np.array(res.data)
        out[case['name']] = arr
        out[case['name'] + '__sha_data'] = np.frombuffer(
            bytes.fromhex(sha(data)), dtype=np.uint8)
        out[case['name'] + '__sha_masks'] = np.frombuffer(
            bytes.fromhex(sha(masks)), dtype=np.uint8)
        print(case['name'], arr.shape, arr.dtype)
    save('apply_masks_dense', **out)


# ---------------------------------------------------------------------------
# 2. SumUDF / SumSigUDF (config C1 shape on a reduced nav)
# ---------------------------------------------------------------------------
def gen_sums():
    out = {}
    for case in recipes.SUM_CASES:
        data = recipes.make_sum_case(case)
        ds = MemoryDataSet(
            data=data, num_partitions=case['num_partitions'], sig_dims=2,
            tileshape=case.get('tileshape'),
        )
        s = run(ds, SumUDF(**case.get('sum_kwargs', {})))['intensity']
        ss = run(ds, SumSigUDF())['intensity']
        out[case['name'] + '__sum'] = np.array(s.data)
        out[case['name'] + '__sumsig'] = np.array(ss.data)
        out[case['name'] + '__sha_data'] = np.frombuffer(bytes.fromhex(sha(data)), dtype=np.uint8)
        print(case['name'], s.data.dtype, ss.data.dtype)
    # SumAnalysis dtype rule (analysis/sum.py:94-98)
    data = recipes.make_sum_case(recipes.SUM_CASES[1])
    ds = MemoryDataSet(data=data, num_partitions=2, sig_dims=2)
    udf = SumAnalysis(ds, {}).get_udf()
    out['sum_analysis_u16__sum'] = np.array(run(ds, udf)['intensity'].data)
    save('sums', **out)


# ---------------------------------------------------------------------------
# 3. CoM: CoMUDF (all result buffers) and COMAnalysis (udf + post-processing)
# ---------------------------------------------------------------------------
def gen_com():
    out = {}
    for case in recipes.COM_CASES:
        data = recipes.make_com_case(case)
        ds = MemoryDataSet(data=data, num_partitions=case['num_partitions'], sig_dims=2)
        udf = CoMUDF.with_params(**case['params'])
        res = run(ds, udf)
        for k, v in res.items():
            out[f"{case['name']}__udf__{k}"] = np.array(v.data)
        # the analysis flavour
        ap = dict(case['analysis_params'])
        analysis = COMAnalysis(ds, ap)
        ares = run(ds, analysis.get_udf())['intensity']
        inten = np.array(ares.data)
        out[f"{case['name']}__analysis__intensity"] = inten
        p = analysis.parameters
        yc_raw, xc_raw = center_shifts(inten[..., 0], inten[..., 1], inten[..., 2], p['cy'], p['cx'])
        yc, xc = apply_correction(yc_raw, xc_raw, scan_rotation=p['scan_rotation'],
                                  flip_y=p['flip_y'])
        out[f"{case['name']}__analysis__x"] = xc
        out[f"{case['name']}__analysis__y"] = yc
        out[f"{case['name']}__analysis__magnitude"] = magnitude(yc, xc)
        out[f"{case['name']}__analysis__divergence"] = divergence(yc, xc)
        out[f"{case['name']}__analysis__curl"] = curl_2d(yc, xc)
        out[f"{case['name']}__sha_data"] = np.frombuffer(bytes.fromhex(sha(data)), dtype=np.uint8)
        print(case['name'], {k: (v.data.shape, str(v.data.dtype)) for k, v in res.items()})
    # coordinate helpers (corrections/coordinates.py:11-54)
    out['rotate_deg_33'] = coordinates.rotate_deg(33.)
    out['rotate_deg_m90'] = coordinates.rotate_deg(-90.)
    out['flip_y'] = coordinates.flip_y()
    out['identity'] = coordinates.identity()
    save('com', **out)


# ---------------------------------------------------------------------------
# 4. Radial Fourier analysis (dense complex64 masks)
# ---------------------------------------------------------------------------
def gen_radial_fourier():
    out = {}
    for case in recipes.RF_CASES:
        data = recipes.make_rf_case(case)
        ds = MemoryDataSet(data=data, num_partitions=case['num_partitions'], sig_dims=2)
        analysis = RadialFourierAnalysis(ds, dict(case['params']))
        p = analysis.parameters
        out[f"{case['name']}__params"] = np.array(
            [p['cx'], p['cy'], p['ri'], p['ro'], p['n_bins'], p['max_order'], p['mask_count'],
             0 if p['use_sparse'] is False else 1], dtype=np.float64)
        if p['use_sparse'] is not False:
            print(case['name'], 'resolves to sparse -> parameters only')
            continue
        udf_res = run(ds, analysis.get_udf())
        inten = np.array(udf_res['intensity'].data)
        out[f"{case['name']}__intensity"] = inten
        rs = analysis.get_udf_results(udf_res, None, damage=True)
        out[f"{case['name']}__raw_results"] = np.array(rs.raw_results)
        out[f"{case['name']}__sha_data"] = np.frombuffer(bytes.fromhex(sha(data)), dtype=np.uint8)
        print(case['name'], inten.shape, inten.dtype)
    save('radial_fourier', **out)


# ---------------------------------------------------------------------------
# 5. Mask factories, bit for bit
# ---------------------------------------------------------------------------
def gen_mask_factories():
    out = {}
    for i, kw in enumerate(recipes.CIRCULAR_CASES):
        out[f'circular_{i}'] = ref_masks.circular(**kw)
    for i, kw in enumerate(recipes.RING_CASES):
        out[f'ring_{i}'] = ref_masks.ring(**kw)
    for i, kw in enumerate(recipes.RADIAL_BINS_CASES):
        out[f'radial_bins_{i}'] = ref_masks.radial_bins(use_sparse=False, **kw)
    for i, kw in enumerate(recipes.POLAR_MAP_CASES):
        r, phi = ref_masks.polar_map(**kw)
        out[f'polar_map_{i}__r'] = r
        out[f'polar_map_{i}__phi'] = phi
    for i, (x, y) in enumerate(recipes.GRADIENT_CASES):
        out[f'gradient_x_{i}'] = ref_masks.gradient_x(x, y)
        out[f'gradient_y_{i}'] = ref_masks.gradient_y(x, y)
    for i, args in enumerate(recipes.BOUNDING_RADIUS_CASES):
        out[f'bounding_radius_{i}'] = np.array(ref_masks.bounding_radius(*args))
    for i, kw in enumerate(recipes.RADIAL_MASK_FACTORY_CASES):
        out[f'radial_mask_factory_{i}'] = radial_mask_factory(use_sparse=False, **kw)()
    for i, kw in enumerate(recipes.RECT_CASES):
        out[f'rectangular_{i}'] = ref_masks.rectangular(**kw)
    for i, kw in enumerate(recipes.BGSUB_CASES):
        out[f'background_subtraction_{i}'] = ref_masks.background_subtraction(**kw)
    for i, kw in enumerate(recipes.RADIAL_GRADIENT_CASES):
        out[f'radial_gradient_{i}'] = ref_masks.radial_gradient(**kw)
    save('mask_factories', **out)


# ---------------------------------------------------------------------------
# 6. rmatmul (dense x CSR / CSC), the reference's own sparse kernel, run as Python
# ---------------------------------------------------------------------------
def gen_rmatmul():
    out = {}
    for case in recipes.RMATMUL_CASES:
        left, dense_right = recipes.make_rmatmul_case(case)
        csr = sp.csr_matrix(dense_right)
        csc = sp.csc_matrix(dense_right)
        r1 = rmatmul(left, csr)
        r2 = rmatmul(left, csc)
        out[case['name'] + '__csr'] = r1
        out[case['name'] + '__csc'] = r2
        print(case['name'], r1.dtype, r1.shape, np.abs(r1 - r2).max())
    save('rmatmul', **out)


# ---------------------------------------------------------------------------
# 6b. non-finite pixels through the reference: its CSR / CSC loops (stored entries only) and ApplyMasksUDF with a
#     dense stack (torch.mm and `flat_tile @ masks`)
# ---------------------------------------------------------------------------
def gen_nonfinite():
    out = {}
    for case in recipes.NONFINITE_CASES:
        data, stack = recipes.make_nonfinite_case(case)
        n_masks = stack.shape[0]
        flat = data.reshape((-1, stack.shape[1] * stack.shape[2]))
        right = stack.reshape((n_masks, -1)).T                       # (px, n_masks)
        with np.errstate(invalid='ignore', over='ignore'):
            out[case['name'] + '__rmatmul_csr'] = rmatmul(flat, sp.csr_matrix(right))
            out[case['name'] + '__rmatmul_csc'] = rmatmul(flat, sp.csc_matrix(right))
            ds = MemoryDataSet(data=data, num_partitions=2, sig_dims=2)
            # (ApplyMasksUDF(use_sparse='scipy.sparse') itself cannot run here: MaskContainer stacks the masks as pydata
            #  `sparse.COO` first, a third-party package that is absent -- SURVEY.md 8(c); what it then calls per tile is
            #  exactly the rmatmul above, udf/masks.py:34-40, :68-69)
            for use_torch in (True, False):
                udf = ApplyMasksUDF(mask_factories=lambda: stack, use_sparse=False, use_torch=use_torch,
                                    mask_count=n_masks, mask_dtype=stack.dtype)
                out[case['name'] + f'__udf_dense_torch{int(use_torch)}'] = np.array(run(ds, udf)['intensity'].data)
        out[case['name'] + '__sha_data'] = np.frombuffer(bytes.fromhex(sha(data)), dtype=np.uint8)
        out[case['name'] + '__sha_stack'] = np.frombuffer(bytes.fromhex(sha(stack)), dtype=np.uint8)
        a, b = out[case['name'] + '__rmatmul_csr'], out[case['name'] + '__udf_dense_torch0']
        print(case['name'], 'sparse: non-finite entries', int((~np.isfinite(a)).sum()), 'dense:', int((~np.isfinite(b)).sum()))
    save('nonfinite', **out)


# ---------------------------------------------------------------------------
# 7. Partitioning + tiling negotiation for the BASELINE.json config shapes
# ---------------------------------------------------------------------------
def gen_tiling():
    out = {}
    for case in recipes.TILING_CASES:
        shape = tuple(case['shape'])
        # np.zeros is lazily committed; nothing below touches the pages
        data = np.zeros(shape, dtype=case['dtype'])
        ds = MemoryDataSet(
            data=data, num_partitions=case['num_partitions'], sig_dims=2,
            tileshape=case.get('tileshape'),
        )
        if case['udf'] == 'masks':
            nm = case.get('n_masks', 16)
            udf = ApplyMasksUDF(
                mask_factories=lambda: np.zeros((nm,) + shape[2:], dtype=np.float32),
                mask_count=nm, mask_dtype=np.float32, use_sparse=False)
        else:
            udf = SumUDF()
        runner = UDFRunner([udf])
        tasks, params = runner._prepare_run_for_dataset(
            ds, EX, EX._get_local_env(), None, None, None, False)
        ts = params.tiling_scheme
        out[case['name'] + '__tileshape'] = np.array(tuple(ts.shape), dtype=np.int64)
        out[case['name'] + '__n_sig_slices'] = np.array(len(ts), dtype=np.int64)
        out[case['name'] + '__sig_slices'] = np.array(
            [list(s.origin) + list(s.shape) for _, s in ts.slices], dtype=np.int64)
        out[case['name'] + '__partitions'] = np.array(
            [[t.partition.slice.origin[0], t.partition.slice.shape[0]] for t in tasks],
            dtype=np.int64)
        print(case['name'], tuple(ts.shape), len(ts), len(tasks))
    save('tiling', **out)


# ---------------------------------------------------------------------------
# 8. one-mask analyses (disk / ring / point) and MasksAnalysis wiring
# ---------------------------------------------------------------------------
def gen_single_mask_analyses():
    out = {}
    data = recipes.make_com_case(recipes.COM_CASES[0])
    ds = MemoryDataSet(data=data, num_partitions=2, sig_dims=2)
    for name, cls, params in recipes.single_mask_analyses():
        a = {'disk': DiskMaskAnalysis, 'ring': RingMaskAnalysis, 'point': PointMaskAnalysis,
             'masks': MasksAnalysis}[cls](ds, dict(params))
        res = run(ds, a.get_udf())['intensity']
        out[name] = np.array(res.data)
        print(name, res.data.shape, res.data.dtype)
    save('single_mask_analyses', **out)


# ---------------------------------------------------------------------------
# 9. shifted masks
# ---------------------------------------------------------------------------
def gen_shifts():
    out = {}
    for case in recipes.SHIFT_CASES:
        data, masks, shifts = recipes.make_shift_case(case)
        ds = MemoryDataSet(data=data, num_partitions=case['num_partitions'], sig_dims=2)
        if case['shifts'] == 'aux':
            sh = ApplyMasksUDF.aux_data(shifts.reshape((-1, 2)).ravel(), kind='nav',
                                        extra_shape=(2,), dtype=shifts.dtype)
        else:
            sh = tuple(int(x) for x in shifts)
        udf = ApplyMasksUDF(mask_factories=lambda: masks, shifts=sh, use_sparse=False)
        res = run(ds, udf)['intensity']
        out[case['name']] = np.array(res.data)
        out[case['name'] + '__sha_data'] = np.frombuffer(bytes.fromhex(sha(data)), dtype=np.uint8)
        print(case['name'], res.data.shape, res.data.dtype)
    save('shifts', **out)


# ---------------------------------------------------------------------------
# 10. detector corrections
# ---------------------------------------------------------------------------
def gen_corrections():
    import sparse
    from libertem.io.corrections import CorrectionSet
    from libertem.io.corrections import detector
    out = {}
    for case in recipes.CORR_CASES:
        data, dark, gain, excluded, masks = recipes.make_corr_case(case)
        sig = tuple(case['sig'])
        excl = None if excluded is None else sparse.COO(coords=excluded, shape=sig, data=True)
        corr = CorrectionSet(dark=dark, gain=gain, excluded_pixels=excl)
        for uname, udf in (('sum', SumUDF()), ('sumsig', SumSigUDF()),
                           ('masks', ApplyMasksUDF(mask_factories=lambda: masks,
                                                   use_sparse=False))):
            # a fresh copy per run: for float32 data the reference's MemoryDataSet corrects the
            # tiles IN PLACE in the user's array (memory.py:102-106 hands out views when dtype and
            # layout already match), so a second run would see corrected-twice data
            ds = MemoryDataSet(data=data.copy(), num_partitions=case['num_partitions'],
                               sig_dims=len(sig))
            res = UDFRunner([udf]).run_for_dataset(ds, EX, corrections=corr).buffers[0]
            arr = np.array(res['intensity'].data)
            out[f"{case['name']}__{uname}"] = arr
            print(case['name'], uname, arr.shape, arr.dtype)
        # the corrected frames themselves (detector.correct, not in place)
        out[f"{case['name']}__corrected"] = detector.correct(
            buffer=data, dark_image=dark, gain_map=gain, excluded_pixels=excluded,
            sig_shape=sig, inplace=False)
        # masks with gain + repair folded in (dark-free linear equivalence)
        if gain is not None:
            out[f"{case['name']}__dot_masks"] = detector.correct_dot_masks(
                masks.astype(np.float64), gain, excluded)
        out[case['name'] + '__sha_data'] = np.frombuffer(bytes.fromhex(sha(data)), dtype=np.uint8)
    for i, (sig, coords) in enumerate(recipes.REPAIR_CASES):
        ex = np.array(coords, dtype=np.int64).T.reshape((len(sig), -1))
        d = detector.RepairDescriptor(sig_shape=sig, excluded_pixels=ex, allow_empty=True)
        out[f"repair{i}__exclude_flat"] = np.asarray(d.exclude_flat)
        out[f"repair{i}__repair_flat"] = np.asarray(d.repair_flat)
        out[f"repair{i}__repair_counts"] = np.asarray(d.repair_counts)
    for i, (tile_shape, sig_shape, base_shape, coords) in enumerate(recipes.ADJUST_CASES):
        excl = sparse.COO(coords=np.array(coords, dtype=np.int64), shape=sig_shape, data=True)
        corr = CorrectionSet(excluded_pixels=excl, allow_empty=True)
        out[f"adjust{i}"] = np.array(corr.adjust_tileshape(
            tile_shape=tile_shape, sig_shape=sig_shape, base_shape=base_shape))
        print('adjust', i, out[f"adjust{i}"])
    save('corrections', **out)


# ---------------------------------------------------------------------------
# 11. CrystallinityUDF
# ---------------------------------------------------------------------------
def gen_crystallinity():
    from libertem.udf.crystallinity import CrystallinityUDF
    out = {}
    for case in recipes.CRYST_CASES:
        data = recipes.make_cryst_case(case)
        ds = MemoryDataSet(data=data, num_partitions=case['num_partitions'], sig_dims=2)
        udf = CrystallinityUDF(rad_in=case['rad_in'], rad_out=case['rad_out'],
                               real_center=case['real_center'], real_rad=case['real_rad'])
        res = run(ds, udf)['intensity']
        out[case['name']] = np.array(res.data)
        out[case['name'] + '__sha_data'] = np.frombuffer(bytes.fromhex(sha(data)), dtype=np.uint8)
        print(case['name'], res.data.shape, res.data.dtype, float(np.abs(res.data).max()))
    save('crystallinity', **out)


# ---------------------------------------------------------------------------
# 12. byte-order decoders (io/dataset/base/decode.py)
# ---------------------------------------------------------------------------
def gen_decode():
    from libertem.io.dataset.base import decode as ref_decode
    out = {}
    dec = ref_decode.DtypeConversionDecoder()
    for case in recipes.DECODE_CASES:
        vals, raw = recipes.make_decode_case(case)
        in_full = np.dtype(case['in_dtype']).newbyteorder(case['order'])
        read = np.dtype(case['out_dtype'])
        need = bool(dec._need_byteswap(in_full, read))
        native = dec.get_native_dtype(in_full, read)
        fn = dec.get_decode(native_dtype=in_full, read_dtype=read)
        res = np.zeros((1, vals.size), dtype=read)
        n = np.array(case['shape'])
        fn(inp=raw if need else raw.view(np.dtype(case['in_dtype'])), out=res, idx=0,
           native_dtype=np.dtype(case['in_dtype']), rr=np.array([0, 0, raw.nbytes]),
           origin=np.zeros(3, dtype=np.int64), shape=n, ds_shape=n)
        out[case['name']] = res
        out[case['name'] + '__need_swap'] = np.array(need)
        out[case['name'] + '__native'] = np.array(str(np.dtype(native)))
        out[case['name'] + '__sha_raw'] = np.frombuffer(bytes.fromhex(sha(raw)), dtype=np.uint8)
        if need:
            only = {2: ref_decode.decode_swap_only_2, 4: ref_decode.decode_swap_only_4,
                    8: ref_decode.decode_swap_only_8}[in_full.itemsize]
            res2 = np.zeros((1, vals.size), dtype=np.dtype(case['in_dtype']))
            only(inp=raw, out=res2, idx=0, native_dtype=np.dtype(case['in_dtype']),
                 rr=np.array([0, 0, raw.nbytes]), origin=np.zeros(3, dtype=np.int64), shape=n,
                 ds_shape=n)
            out[case['name'] + '__swap_only'] = res2
        print(case['name'], need, native, res.dtype)
    # floats in the other byte order are refused by the reference
    try:
        dec.get_decode(native_dtype=np.dtype('>f4'), read_dtype=np.dtype('float32'))
        out['float_swap_error'] = np.array('')
    except NotImplementedError as e:
        out['float_swap_error'] = np.array(str(e))
    save('decode', **out)


def gen_decode_signed():
    """signed integers through the reference's decoders (run as plain Python: the stand-in njit is the
    identity, NumPy's scalar-into-array assignment wraps like numba's)"""
    from libertem.io.dataset.base import decode as ref_decode
    out = {}
    dec = ref_decode.DtypeConversionDecoder()
    for case in recipes.DECODE_SIGNED_CASES:
        vals, raw = recipes.make_decode_case(case)
        in_full = np.dtype(case['in_dtype']).newbyteorder(case['order'])
        read = np.dtype(case['out_dtype'])
        need = bool(dec._need_byteswap(in_full, read))
        fn = dec.get_decode(native_dtype=in_full, read_dtype=read)
        res = np.zeros((1, vals.size), dtype=read)
        n = np.array(case['shape'])
        with np.errstate(over='ignore'):
            fn(inp=raw if need else raw.view(np.dtype(case['in_dtype'])), out=res, idx=0,
               native_dtype=np.dtype(case['in_dtype']), rr=np.array([0, 0, raw.nbytes]),
               origin=np.zeros(3, dtype=np.int64), shape=n, ds_shape=n)
        out[case['name']] = res
        out[case['name'] + '__need_swap'] = np.array(need)
        out[case['name'] + '__sha_raw'] = np.frombuffer(bytes.fromhex(sha(raw)), dtype=np.uint8)
        print(case['name'], need, res.dtype, int((vals < 0).sum()), 'negative inputs')
    save('decode_signed', **out)


# ---------------------------------------------------------------------------
# 13. PickUDF and the pick analyses
# ---------------------------------------------------------------------------
def gen_pick():
    from libertem.udf.raw import PickUDF
    from libertem.analysis.raw import PickFrameAnalysis
    from libertem.analysis.rawfft import PickFFTFrameAnalysis
    out = {}
    for case in recipes.PICK_CASES:
        data = recipes.make_pick_case(case)
        ds = MemoryDataSet(data=data, num_partitions=case['num_partitions'],
                           sig_dims=len(case['sig']))
        roi = np.zeros(case['nav'], dtype=bool)
        for c in case['roi_frames']:
            roi[c] = True
        res = run(ds, PickUDF(), roi=roi)['intensity']
        out[case['name'] + '__picked'] = np.array(res.data)
        for cls, tag in ((PickFrameAnalysis, 'frame'), (PickFFTFrameAnalysis, 'fft')):
            params = dict(case['pick'])
            if tag == 'fft' and case['real'] is not None:
                params.update(real_rad=case['real']['rad'], real_centerx=case['real']['cx'],
                              real_centery=case['real']['cy'])
            # the result sets of the reference render through matplotlib / colorcet (absent here):
            # capture the array its get_udf_results() hands to get_generic_results() instead
            capture = type('Capture' + tag, (cls,), {
                'get_generic_results': lambda self, data, damage: np.array(data)})
            a = capture(dataset=ds, parameters=params)
            aroi = a.get_roi()
            aroi = np.asarray(aroi.todense()) if hasattr(aroi, 'todense') else np.asarray(aroi)
            out[f"{case['name']}__{tag}__roi"] = aroi
            ures = run(ds, a.get_udf(), roi=aroi)
            out[f"{case['name']}__{tag}"] = a.get_udf_results(ures, aroi, damage=True)
        out[case['name'] + '__sha_data'] = np.frombuffer(bytes.fromhex(sha(data)), dtype=np.uint8)
        print(case['name'], res.data.shape, res.data.dtype)
    save('pick', **out)

# ---------------------------------------------------------------------------
# 14. the BASELINE.json config workloads at their real detector size (reduced nav)
# ---------------------------------------------------------------------------
def gen_config_workloads():
    out = {}
    for case in recipes.WORKLOAD_CASES:
        data = recipes.make_workload_case(case)
        ds = MemoryDataSet(data=data, num_partitions=case['num_partitions'], sig_dims=2)
        name = case['name']
        out[f"{name}__sha_data"] = np.frombuffer(bytes.fromhex(sha(data)), dtype=np.uint8)
        if case['kind'] == 'rf':
            analysis = RadialFourierAnalysis(ds, dict(case['params']))
            p = analysis.parameters
            assert p['use_sparse'] is False, p
            udf_res = run(ds, analysis.get_udf())
            out[f"{name}__intensity"] = np.array(udf_res['intensity'].data)
            rs = analysis.get_udf_results(udf_res, None, damage=True)
            out[f"{name}__raw_results"] = np.array(rs.raw_results)
            print(name, out[f"{name}__intensity"].shape, out[f"{name}__intensity"].dtype)
            continue
        for i, ap in enumerate(case['analysis_params']):
            analysis = COMAnalysis(ds, dict(ap))
            inten = np.array(run(ds, analysis.get_udf())['intensity'].data)
            p = analysis.parameters
            yc_raw, xc_raw = center_shifts(inten[..., 0], inten[..., 1], inten[..., 2],
                                           p['cy'], p['cx'])
            yc, xc = apply_correction(yc_raw, xc_raw, scan_rotation=p['scan_rotation'],
                                      flip_y=p['flip_y'])
            out[f"{name}__analysis{i}__intensity"] = inten
            out[f"{name}__analysis{i}__x"] = xc
            out[f"{name}__analysis{i}__y"] = yc
            out[f"{name}__analysis{i}__magnitude"] = magnitude(yc, xc)
            out[f"{name}__analysis{i}__divergence"] = divergence(yc, xc)
            out[f"{name}__analysis{i}__curl"] = curl_2d(yc, xc)
        for i, up in enumerate(case['udf_params']):
            res = run(ds, CoMUDF.with_params(**up))
            for k, v in res.items():
                out[f"{name}__udf{i}__{k}"] = np.array(v.data)
        print(name, inten.shape, inten.dtype)
    save('config_workloads', **out)


# ---------------------------------------------------------------------------
# 15. Merlin .mib files: the reference's own encoders, decoders and MIBDataSet
# ---------------------------------------------------------------------------
def gen_mib():
    import tempfile
    from libertem.io.dataset import mib as ref_mib
    from libertem.io.dataset.mib import MIBDataSet
    from libertem.udf.raw import PickUDF
    out = {}
    enc = {1: ref_mib.encode_r1, 6: ref_mib.encode_r6, 12: ref_mib.encode_r12}
    for case in recipes.MIB_CASES:
        frames, files, hdr = recipes.make_mib_case(case)
        name = case['name']
        # the recipe's payload bytes against the reference's encoders (single chip, 1 / 6 / 12 bit)
        if case['kind'] == 'r' and case['bits'] in enc and not case.get('quad'):
            h, w = case['sig']
            ref_rows = np.zeros((h, w * {1: 1, 6: 8, 12: 16}[case['bits']] // 8), dtype=np.uint8)
            enc[case['bits']](inp=frames[0], out=ref_rows)
            assert ref_rows.tobytes() == recipes.mib_frame_payload(frames[0], case), name
        with tempfile.TemporaryDirectory() as d:
            for fn, blob in files.items():
                with open(os.path.join(d, fn), 'wb') as f:
                    f.write(blob)
            hdr_path = os.path.join(d, name + '.hdr')
            with open(hdr_path, 'w') as f:
                f.write(hdr)
            ds = MIBDataSet(path=hdr_path, sync_offset=case.get('sync_offset', 0))
            ds = ds.initialize(EX)
            assert tuple(ds.shape.nav) == tuple(case['nav']), (ds.shape, case['nav'])
            roi = np.ones(tuple(ds.shape.nav), dtype=bool)
            picked = run(ds, PickUDF(), roi=roi)['intensity'].data
            so = case.get('sync_offset', 0)
            n_nav = int(np.prod(case['nav']))
            expect = np.zeros((n_nav,) + tuple(case['sig']), dtype=frames.dtype)
            src = frames[max(so, 0):max(so, 0) + n_nav - max(-so, 0)]
            expect[max(-so, 0):max(-so, 0) + len(src)] = src
            # (24 bit: the reference declares uint16 and PickUDF reads into that: values wrap)
            same = np.array_equal(np.asarray(picked).reshape(expect.shape),
                                  expect.astype(picked.dtype))
            print(name, 'reference reads back the recipe frames:', same)
            assert same, name
            sums = run(ds, SumSigUDF())['intensity'].data
            rng = np.random.default_rng(case['seed'] + 5000)
            masks = rng.random((3,) + tuple(ds.shape.sig)).astype(np.float32)
            applied = run(ds, ApplyMasksUDF(mask_factories=lambda: masks))['intensity'].data
            out[name + '__dtype'] = np.array(str(np.dtype(ds.dtype)))
            out[name + '__frames'] = np.asarray(picked)
            out[name + '__sumsig'] = np.asarray(sums)
            out[name + '__masks'] = np.asarray(applied)
            out[name + '__sha_files'] = np.frombuffer(bytes.fromhex(hashlib.sha256(
                b''.join(files[k] for k in sorted(files))).hexdigest()), dtype=np.uint8)
            print(name, ds.dtype, picked.dtype, picked.shape, sums.dtype, applied.dtype)
    save('mib', **out)


GENERATORS = {}

if __name__ == '__main__':
    GENERATORS.update({k[4:]: v for k, v in list(globals().items()) if k.startswith('gen_')})
    if len(sys.argv) > 1:
        # regenerate only the named fixtures, keep the other manifest entries
        with open(os.path.join(HERE, 'MANIFEST.json')) as f:
            MANIFEST.update(json.load(f))
        for name in sys.argv[1:]:
            GENERATORS[name]()
        with open(os.path.join(HERE, 'MANIFEST.json'), 'w') as f:
            json.dump(MANIFEST, f, indent=1, sort_keys=True)
        sys.exit(0)
    gen_decode()
    gen_decode_signed()
    gen_crystallinity()
    gen_corrections()
    gen_shifts()
    gen_apply_masks_dense()
    gen_sums()
    gen_com()
    gen_radial_fourier()
    gen_mask_factories()
    gen_rmatmul()
    gen_nonfinite()
    gen_tiling()
    gen_single_mask_analyses()
    gen_pick()
    gen_config_workloads()
    gen_mib()
    with open(os.path.join(HERE, 'MANIFEST.json'), 'w') as f:
        json.dump(MANIFEST, f, indent=1, sort_keys=True)
