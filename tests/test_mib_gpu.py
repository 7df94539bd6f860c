"""
Merlin .mib files on the GPU (-m gpu): `ltmi_mib_decode` through the C ABI against the oracle
(oracle/mib.py) and the golden vectors the REAL reference's MIBDataSet produced from the same files
(tests/golden/mib.npz), then MIBDataSet end to end (frames, SumSigUDF, ApplyMasksUDF, ROI).
Integer work: bit-exact.
"""
import os

import numpy as np
import pytest

import recipes

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from libertem_amd.api import Context
    assert torch.cuda.is_available()
    c = Context.make_with('hip', gpus=0)
    yield c
    c.close()


def _golden(golden_dir):
    return np.load(os.path.join(golden_dir, 'mib.npz'))


def _write(tmp_path, case):
    frames, files, hdr = recipes.make_mib_case(case)
    d = tmp_path / case['name']
    d.mkdir()
    for fn, blob in files.items():
        (d / fn).write_bytes(blob)
    (d / (case['name'] + '.hdr')).write_text(hdr)
    return frames, files, str(d / (case['name'] + '.hdr'))


@pytest.mark.parametrize('case', recipes.MIB_CASES, ids=lambda c: c['name'])
def test_decode_kernel_vs_oracle(case):
    """one file's bytes on the device -> frames, through the C ABI"""
    from libertem_amd import hip
    from libertem_amd.common.hiparray import HipArray
    from libertem_amd.io.dataset.mib import parse_frame_header
    from oracle import mib as omib
    frames, files, _ = recipes.make_mib_case(case)
    name, blob = sorted(files.items())[0]
    f = parse_frame_header(blob[:1024], len(blob))
    assert f == {**f, **{k: v for k, v in omib.parse_header(blob[:1024], len(blob)).items()}}
    h, w = f['image_size']
    stride = f['header_size_bytes'] + f['image_size_bytes']
    n = f['num_images']
    ref = np.stack([omib.decode_frame(blob[i * stride + f['header_size_bytes']:(i + 1) * stride],
                                      omib.parse_header(blob[:1024], len(blob))) for i in range(n)])
    quad = f['mib_kind'] == 'r' and f['num_chips'] > 1
    storages = [f['storage_dtype']] + ([np.dtype('uint32')] if f['bits_per_pixel'] == 24 else [])
    for storage in storages:
        for shift in (0, 3):                       # file bytes at an odd device address as well
            raw = torch.zeros(len(blob) + 16, dtype=torch.uint8, device='cuda:0')
            raw[shift:shift + len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
            out = HipArray.empty((n, h, w), storage, 0)
            hip.mib_decode(0, raw.data_ptr() + shift, stride, f['header_size_bytes'], f['mib_kind'],
                           f['bits_per_pixel'], quad, n, h, w, out.data_ptr(), storage)
            got = out.cpu()
            assert got.dtype == storage
            assert np.array_equal(got.astype(np.uint32), ref), (case['name'], storage, shift)


def test_decode_kernel_refuses_what_is_not_a_format():
    from libertem_amd import hip
    from libertem_amd.common.hiparray import HipArray
    raw = torch.zeros(4096, dtype=torch.uint8, device='cuda:0')
    out = HipArray.empty((1, 8, 64), np.uint8, 0)
    with pytest.raises(ValueError, match='not a .mib format'):
        hip.mib_decode(0, raw.data_ptr(), 1024, 384, 'r', 7, False, 1, 8, 64, out.data_ptr(), np.uint8)
    with pytest.raises(ValueError, match='decode to uint16'):
        hip.mib_decode(0, raw.data_ptr(), 2048, 384, 'r', 12, False, 1, 8, 64, out.data_ptr(), np.uint8)
    with pytest.raises(ValueError, match='whole 64-bit words'):
        hip.mib_decode(0, raw.data_ptr(), 1024, 384, 'r', 1, False, 1, 8, 32, out.data_ptr(), np.uint8)
    with pytest.raises(ValueError, match='exceed the frame stride'):
        hip.mib_decode(0, raw.data_ptr(), 500, 384, 'r', 6, False, 1, 8, 64, out.data_ptr(), np.uint8)


@pytest.mark.parametrize('case', recipes.MIB_CASES, ids=lambda c: c['name'])
def test_dataset_vs_reference_golden(ctx, golden_dir, tmp_path, case):
    from libertem_amd.udf.masks import ApplyMasksUDF
    from libertem_amd.udf.sumsigudf import SumSigUDF
    from libertem_amd.udf.raw import PickUDF
    g = _golden(golden_dir)
    name = case['name']
    frames, files, hdr_path = _write(tmp_path, case)
    ds = ctx.load('mib', path=hdr_path, sync_offset=case.get('sync_offset', 0))
    assert tuple(ds.shape) == tuple(case['nav']) + tuple(case['sig'])
    assert str(np.dtype(ds.dtype)) == str(np.dtype(str(g[name + '__dtype'])).newbyteorder('='))
    assert ds.is_device_resident
    ref_frames = g[name + '__frames']
    roi = np.ones(tuple(case['nav']), dtype=bool)
    picked = ctx.run_udf(dataset=ds, udf=PickUDF(), roi=roi)['intensity'].data
    if case['bits'] == 24:
        # the reference reads 24-bit pixels into its declared uint16 and wraps; here they are exact
        assert np.array_equal(np.asarray(picked).astype(np.uint32).astype(np.uint16).reshape(
            ref_frames.shape), ref_frames)
    else:
        assert np.array_equal(np.asarray(picked).reshape(ref_frames.shape).astype(ref_frames.dtype),
                              ref_frames)
    sums = ctx.run_udf(dataset=ds, udf=SumSigUDF())['intensity'].data
    ref_s = g[name + '__sumsig']
    assert sums.dtype == ref_s.dtype and np.allclose(sums, ref_s, rtol=1e-6)
    rng = np.random.default_rng(case['seed'] + 5000)
    masks = rng.random((3,) + tuple(case['sig'])).astype(np.float32)
    res = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=lambda: masks))['intensity'].data
    ref_m = g[name + '__masks']
    assert res.dtype == ref_m.dtype, (res.dtype, ref_m.dtype)
    tol = 1e-5 if ref_m.dtype == np.float32 else 1e-12
    assert np.allclose(res, ref_m, rtol=tol, atol=tol * np.abs(ref_m).max())
    # a region of interest reads the resident frames in place
    roi = np.zeros(tuple(case['nav']), dtype=bool)
    roi.reshape(-1)[::2] = True
    res_roi = ctx.run_udf(dataset=ds, udf=ApplyMasksUDF(mask_factories=lambda: masks), roi=roi)
    assert np.allclose(res_roi['intensity'].raw_data, ref_m[roi], rtol=tol,
                       atol=tol * np.abs(ref_m).max())


def test_dataset_parameters_like_the_reference(ctx, tmp_path):
    from libertem_amd.io.dataset.mib import MIBDataSet
    from libertem_amd.io.dataset.base import DataSetException
    case = [c for c in recipes.MIB_CASES if c['name'] == 'r12'][0]
    frames, files, hdr_path = _write(tmp_path, case)
    mib_path = os.path.join(os.path.dirname(hdr_path), sorted(files)[0])
    with pytest.raises(ValueError, match='either nav_shape needs to be passed'):
        MIBDataSet(path=mib_path)
    # a .mib path + nav_shape; other sig_shape with the same number of pixels; 1D / 3D nav
    ds = ctx.load('mib', path=mib_path, nav_shape=(6,), sig_shape=(64, 32))
    assert tuple(ds.shape) == (6, 64, 32)
    with pytest.raises(DataSetException, match='sig_shape must be of size'):
        ctx.load('mib', path=mib_path, nav_shape=(6,), sig_shape=(64, 33))
    with pytest.raises(DataSetException, match='offset should be in'):
        ctx.load('mib', path=mib_path, nav_shape=(2, 3), sync_offset=6)
    d = MIBDataSet.detect_params(hdr_path)
    assert d['parameters']['nav_shape'] == (2, 3) and d['parameters']['sig_shape'] == (32, 64)
    assert d['info']['image_count'] == 6
    assert MIBDataSet.detect_params(mib_path)['parameters']['nav_shape'] == (6,)
    assert MIBDataSet.get_supported_extensions() == {'mib', 'hdr'}
    diag = {x['name']: x['value'] for x in ctx.load('mib', path=hdr_path).get_diagnostics()}
    assert diag == {'Bits per pixel': '12', 'Data kind': 'r', 'Layout': '(1, 1)'}
    # more scan positions than frames: the rest is blank, as in the reference
    ds = ctx.load('mib', path=hdr_path, nav_shape=(2, 4))
    from libertem_amd.udf.sumsigudf import SumSigUDF
    s = ctx.run_udf(dataset=ds, udf=SumSigUDF())['intensity'].data.reshape(-1)
    assert np.array_equal(s[:6], frames.reshape(6, -1).sum(axis=1).astype(np.float32))
    assert np.all(s[6:] == 0)


def test_many_frames_over_several_chunks(ctx, tmp_path):
    """chunked copy + decode (two buffers in flight), files that end inside a chunk"""
    from libertem_amd.io.dataset.mib import MIBDataSet
    from libertem_amd.udf.sumsigudf import SumSigUDF
    case = dict(name='chunks', kind='r', bits=12, sig=(64, 64), frames=(37, 50, 13), nav=(10, 10),
                seed=4242)
    frames, files, hdr_path = _write(tmp_path, case)
    old = MIBDataSet.CHUNK_BYTES
    MIBDataSet.CHUNK_BYTES = 11 * (384 + 64 * 64 * 2)          # 11 frames per step
    try:
        ds = ctx.load('mib', path=hdr_path)
    finally:
        MIBDataSet.CHUNK_BYTES = old
    assert ds.decode_bytes == 100 * (384 + 64 * 64 * 2) and ds.decode_seconds > 0
    s = ctx.run_udf(dataset=ds, udf=SumSigUDF())['intensity'].data.reshape(-1)
    assert np.array_equal(s, frames.reshape(100, -1).sum(axis=1).astype(np.float32))
    got = ds.data.cpu().reshape(frames.shape)
    assert np.array_equal(got, frames)


def test_one_rank_decodes_only_its_block(ctx, tmp_path):
    """shard=(rank, world): one process per GPU holds its block of the first nav axis (sync offset applied
    to the whole series first)"""
    case = [c for c in recipes.MIB_CASES if c['name'] == 'r12_offset'][0]
    frames, files, hdr_path = _write(tmp_path, case)
    for rank in (0, 1):
        ds = ctx.load('mib', path=hdr_path, sync_offset=2, shard=(rank, 2))
        assert tuple(ds.shape) == (2, 3, 32, 64) and ds.shard == (rank, 2)
        local = ds.data.cpu().reshape(3, 32, 64)
        expect = np.zeros((6, 32, 64), dtype=np.uint16)
        expect[:5] = frames[2:7]
        assert np.array_equal(local, expect[rank * 3:rank * 3 + 3])
        assert ds.decode_bytes == (3 if rank == 0 else 2) * (384 + 32 * 64 * 2)


def test_series_larger_than_hbm_is_streamed(ctx, tmp_path, monkeypatch):
    """A block of the scan that does not fit (here: may not take) the HBM is not refused: nothing is decoded at
    load time, every partition decodes its frames from the files into a window of HBM when its tiles are asked
    for -- same results as the resident dataset, run after run, with an ROI, with a sync offset, with a shard."""
    from libertem_amd.io.dataset.base import DataSetException
    from libertem_amd.io.dataset.mib import MIBDataSet
    from libertem_amd.udf.sumsigudf import SumSigUDF
    from libertem_amd.udf.masks import ApplyMasksUDF
    for name, kw in (('r1', {}), ('u16', {}), ('r12_offset', dict(sync_offset=2)), ('u16_neg_offset', dict(sync_offset=-2))):
        case = [c for c in recipes.MIB_CASES if c['name'] == name][0]
        frames, _, hdr_path = _write(tmp_path, case)
        sig = tuple(case['sig'])
        resident = ctx.load('mib', path=hdr_path, **kw)
        assert not resident.is_streamed
        n_nav = int(np.prod(resident.shape.nav))
        frame_bytes = int(np.prod(sig)) * np.dtype(resident.storage_dtype).itemsize
        monkeypatch.setattr(MIBDataSet, 'MAX_RESIDENT_BYTES', 2 * frame_bytes)
        ds = ctx.load('mib', path=hdr_path, **kw)
        monkeypatch.setattr(MIBDataSet, 'MAX_RESIDENT_BYTES', None)
        assert ds.is_streamed and tuple(ds.shape) == tuple(resident.shape)
        assert ds.get_num_partitions() >= -(-n_nav // 2) and ds.decode_bytes == 0
        with pytest.raises(DataSetException, match='streamed'):
            ds.data
        masks = np.random.default_rng(5).random((3,) + sig).astype(np.float32)
        udf = ApplyMasksUDF(mask_factories=lambda: masks, use_sparse=False, mask_count=3, mask_dtype=np.float32)
        want_sum = ctx.run_udf(dataset=resident, udf=SumSigUDF())['intensity'].data
        want = ctx.run_udf(dataset=resident, udf=udf)['intensity'].data
        for _ in range(2):                                   # (the second run re-uses the plan, not the tiles)
            assert np.array_equal(ctx.run_udf(dataset=ds, udf=SumSigUDF())['intensity'].data, want_sum)
            assert np.array_equal(ctx.run_udf(dataset=ds, udf=udf)['intensity'].data, want)
        assert ds.decode_bytes > 0
        roi = np.zeros(resident.shape.nav, dtype=bool)
        roi.reshape(-1)[[0, n_nav - 1, n_nav // 2]] = True
        got = ctx.run_udf(dataset=ds, udf=SumSigUDF(), roi=roi)['intensity'].raw_data
        assert np.array_equal(got, ctx.run_udf(dataset=resident, udf=SumSigUDF(), roi=roi)['intensity'].raw_data)
    # one rank of two, streamed
    case = [c for c in recipes.MIB_CASES if c['name'] == 'r12_offset'][0]
    (tmp_path / 'shard').mkdir()
    frames, _, hdr_path = _write(tmp_path / 'shard', case)
    monkeypatch.setattr(MIBDataSet, 'MAX_RESIDENT_BYTES', 32 * 64 * 2)
    ds = ctx.load('mib', path=hdr_path, sync_offset=2, shard=(1, 2))
    monkeypatch.setattr(MIBDataSet, 'MAX_RESIDENT_BYTES', None)
    assert ds.is_streamed and ds.shard == (1, 2)
    expect = np.zeros((6, 32, 64), dtype=np.uint16)
    expect[:5] = frames[2:7]
    parts = [p for p in ds.get_partitions() if ds.owner_of_frames(p._start_frame, p._start_frame + p._num_frames) == 1]
    assert len(parts) == 3
    for p in parts:
        arr, row0 = ds.device_frames(p._local0, p._num_frames)
        assert np.array_equal(arr.rows(row0, row0 + p._num_frames).cpu().reshape(-1, 32, 64),
                              expect[p._start_frame:p._start_frame + p._num_frames])


def test_partition_that_does_not_fit_is_refused_with_advice(ctx, tmp_path, monkeypatch):
    from libertem_amd.io.dataset.base import DataSetException
    case = [c for c in recipes.MIB_CASES if c['name'] == 'u08'][0]
    _, _, hdr_path = _write(tmp_path, case)
    monkeypatch.setattr(torch.cuda, 'mem_get_info', lambda device=None: (1000, 1 << 38))
    ds = ctx.load('mib', path=hdr_path)                     # (streamed: no frame decoded yet)
    assert ds.is_streamed
    from libertem_amd.udf.sumsigudf import SumSigUDF
    with pytest.raises(DataSetException, match='fewer frames per partition'):
        ctx.run_udf(dataset=ds, udf=SumSigUDF())
