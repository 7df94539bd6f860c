"""Randomised differential test of the C-ABI kernels against NumPy (float64 / integer reference):
random shapes (ragged frame counts, odd pixel counts, 1..70 masks), dtypes, leading dimensions,
accumulate flags, K splits.  `python scripts/fuzz_kernels.py [n_cases] [seed]`"""
import os, sys, time
import numpy as np, torch
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
TILE_DTYPES = ['uint8', 'int8', 'uint16', 'int16', 'int32', 'uint32', 'float32', 'float64', 'int64']
kernels_seen = {}
t0 = time.time()


def dev(a):
    a = np.ascontiguousarray(a)
    v = {np.dtype('uint16'): 'int16', np.dtype('uint32'): 'int32', np.dtype('uint64'): 'int64'}.get(a.dtype)
    if a.dtype.kind == 'c':
        return torch.from_numpy(a).cuda()
    return torch.from_numpy(a.view(v) if v else a).cuda()


def gen(dt, shape):
    dt = np.dtype(dt)
    if dt.kind == 'u':
        return rng.integers(0, min(np.iinfo(dt).max, 5000), shape, endpoint=True).astype(dt)
    if dt.kind == 'i':
        return rng.integers(max(np.iinfo(dt).min, -5000), min(np.iinfo(dt).max, 5000), shape,
                            endpoint=True).astype(dt)
    return (rng.random(shape) - 0.3).astype(dt)


fails = 0
for case in range(n_cases):
    kind = rng.choice(['dense', 'dense', 'dense', 'shift', 'sparse', 'int', 'sparse_int', 'split'])
    n_frames = int(rng.choice([1, 3, 15, 16, 17, 100, 129, 300, 777]))
    tdt = np.dtype(rng.choice(TILE_DTYPES))
    accumulate = bool(rng.integers(0, 2))
    desc = None
    try:
        aligned = bool(rng.integers(0, 2))       # half of the cases on the LDS-DMA (aligned) paths
        if kind == 'shift':
            if aligned:
                h, w = int(rng.choice([16, 24, 32])), int(rng.choice([16, 32, 48]))
            else:
                h, w = int(rng.choice([8, 16, 24, 33])), int(rng.choice([8, 16, 32, 48, 23]))
            n_px = h * w
        elif aligned:
            n_px = int(rng.choice([256, 384, 512, 1024, 2048 + 16, 4096 + 128, 128 * 37 + 48]))
        else:
            n_px = int(rng.choice([1, 7, 64, 128, 255, 256, 257, 384, 1000, 1024, 2048 + 16, 5000]))
        pad = int(rng.choice([0, 0, 16])) if aligned else int(rng.choice([0, 0, 8, 3]))
        ld = n_px + pad
        n_masks = int(rng.choice([1, 2, 5, 16, 17, 25, 33, 49, 50, 52, 64, 70]))
        data = gen(tdt, (n_frames, ld))
        if kind == 'sparse':
            wide_result = False
            if str(tdt) not in ('uint8', 'int8', 'uint16', 'int16', 'float32'):
                if rng.random() < 0.5:
                    wide_result = True          # float64 result: the gather kernel in double
                else:
                    tdt = np.dtype('uint16'); data = gen(tdt, (n_frames, ld))
            elif rng.random() < 0.15:
                wide_result = True              # float64 mask values on narrow pixels
            mdt = np.dtype('float64') if wide_result else np.dtype(rng.choice(['float32', 'complex64']))
            if rng.random() < 0.3:
                n_masks = int(rng.choice([130, 300, 1030, 1100]))      # many groups, two passes
            if rng.random() < 0.5:
                # localised stack: mask k touches the pixels around k * n_px / n_masks (long pairs)
                centre = (np.arange(n_masks) + 0.5) * n_px / n_masks
                width = float(rng.choice([2., 8., 40.])) * max(1., n_px / n_masks)
                dist = np.abs(np.arange(n_px)[:, None] - centre[None, :])
                dense = (dist < width) * (rng.random((n_px, n_masks)) - 0.3)
            else:
                dense = (rng.random((n_px, n_masks)) < 0.1) * (rng.random((n_px, n_masks)) - 0.3)
            if mdt.kind == 'c':
                dense = dense * (1 + 1j * rng.random((n_px, n_masks)))
            dense = dense.astype(mdt)
            rd = np.dtype(np.float64) if wide_result else np.result_type(np.float32, mdt)
            # either sparse kernel: the blocked image (matrix cores) or the SELL gather kernel
            os.environ['LTMI_SPARSE_BELL'] = str(rng.choice(['0', '1']))
            handle = hip.MaskHandle.csr(0, sp.csr_matrix(dense), rd)
            del os.environ['LTMI_SPARSE_BELL']
            ref = data[:, :n_px].astype(np.complex128 if mdt.kind == 'c' else np.float64) @ \
                dense.astype(np.complex128 if mdt.kind == 'c' else np.float64)
            scale = np.abs(data[:, :n_px].astype(np.float64)) @ np.abs(dense).astype(np.float64)
        elif kind == 'sparse_int':
            # integer sparse stack x integer frames: float64 gather + truncation (exact), wrap-around
            if tdt.kind not in 'iu' or tdt.itemsize > 4:
                tdt = np.dtype(rng.choice(['uint8', 'uint16', 'int16', 'int32'])); data = gen(tdt, (n_frames, ld))
            rd = np.dtype(rng.choice(['int32', 'int64', 'uint16', 'uint8']))
            if rng.random() < 0.3:
                n_masks = int(rng.choice([130, 300]))
            dense = ((rng.random((n_px, n_masks)) < 0.08) * rng.integers(-6, 7, (n_px, n_masks))).astype(np.int64)
            handle = hip.MaskHandle.csr(0, sp.csr_matrix(dense), rd)
            ref = data[:, :n_px].astype(np.int64) @ dense
            scale = None
        elif kind == 'split':
            # float32 frames on the bf16 x 3 split kernel (tuning 36) where it applies, else as dispatched
            tdt = np.dtype('float32')
            n_px = int(rng.choice([512, 1024, 2048, 4096, 64 * 37]))
            ld = n_px + pad
            data = gen(tdt, (n_frames, ld))
            data[:, ::5] *= np.float32(1e-4)
            mdt = np.dtype(rng.choice(['float32', 'complex64']))
            n_masks = int(rng.choice([17, 20, 25, 32, 40, 50, 64])) // (2 if mdt.kind == 'c' else 1)
            masks = rng.random((n_masks, n_px)) - 0.25
            if mdt.kind == 'c':
                masks = masks + 1j * (rng.random((n_masks, n_px)) - 0.5)
            masks = masks.astype(mdt)
            rd = mdt
            handle = hip.MaskHandle.dense(0, masks, rd)
            handle.set_tuning(mt=0, waves=36, ksplit=int(rng.choice([0, 0, 3])))
            wide = np.complex128 if rd.kind == 'c' else np.float64
            ref = data[:, :n_px].astype(wide) @ masks.astype(wide).T
            scale = np.abs(data[:, :n_px].astype(np.float64)) @ np.abs(masks).astype(np.float64).T
        elif kind == 'int':
            if tdt.kind not in 'iu':
                tdt = np.dtype('int16'); data = gen(tdt, (n_frames, ld))
            rd = np.dtype(rng.choice(['int32', 'int64', 'uint16']))
            if tdt.itemsize > rd.itemsize:
                rd = np.dtype('int64')
            masks = rng.integers(0, 3, (n_masks, n_px)).astype(rd)
            handle = hip.MaskHandle.dense(0, masks, rd)
            ref = data[:, :n_px].astype(rd) @ masks.T
            scale = None
        else:
            mdt = np.dtype(rng.choice(['float32', 'float32', 'complex64', 'float64']))
            masks = rng.random((n_masks, n_px)) - 0.25
            if mdt.kind == 'c':
                masks = masks + 1j * (rng.random((n_masks, n_px)) - 0.5)
            masks = masks.astype(mdt)
            rd = np.dtype(np.result_type(np.result_type(np.float32, tdt), mdt))
            handle = hip.MaskHandle.dense(0, masks, rd)
            wide = np.complex128 if rd.kind == 'c' else np.float64
            if kind == 'shift':
                shifts = rng.integers(-4, 5, (n_frames, 2)).astype(np.int32)
                ref = np.zeros((n_frames, n_masks), dtype=wide)
                scale = np.zeros((n_frames, n_masks))
                m3 = masks.reshape((n_masks, h, w)).astype(wide)
                d3 = data[:, :n_px].reshape((n_frames, h, w)).astype(np.float64)
                for f in range(n_frames):
                    dy, dx = shifts[f]
                    y0, y1, x0, x1 = max(0, dy), min(h, h + dy), max(0, dx), min(w, w + dx)
                    if y0 < y1 and x0 < x1:
                        mm = m3[:, y0 - dy:y1 - dy, x0 - dx:x1 - dx]
                        ref[f] = (mm * d3[f, y0:y1, x0:x1]).sum(axis=(1, 2))
                        scale[f] = (np.abs(mm) * np.abs(d3[f, y0:y1, x0:x1])).sum(axis=(1, 2))
            else:
                ref = data[:, :n_px].astype(wide) @ masks.astype(wide).T
                scale = np.abs(data[:, :n_px].astype(np.float64)) @ np.abs(masks).astype(np.float64).T
        ksplit = int(rng.choice([0, 0, 0, 2, 5])) if kind in ('dense', 'int') else 0
        if ksplit and kind == 'dense':
            handle.set_tuning(mt=0, waves=0, ksplit=ksplit)
        base = gen(rd if rd.kind in 'iu' else 'float32', (n_frames, n_masks)).astype(rd)
        out = dev(base.copy())
        t = dev(data)
        desc = (kind, n_frames, n_px, ld, n_masks, str(tdt), str(rd), accumulate, ksplit)
        if kind == 'shift':
            handle.apply_shifted_host(t.data_ptr(), tdt, n_frames, ld, h, w, shifts, out.data_ptr(),
                                      n_masks, accumulate)
        else:
            handle.apply(t.data_ptr(), tdt, n_frames, ld, out.data_ptr(), n_masks, accumulate)
        torch.cuda.synchronize()
        res = out.cpu().numpy()
        if res.dtype != rd:
            res = res.view(rd)
        lk = handle.last_kernel()
        kern = lk.split('<')[0] + ('+exact-int' if 'exact-int' in lk else '') + \
            ('+f64' if ',f64>' in lk else '') + \
            ('+shifted' if 'shifted' in lk else '') + ('+valu' if 'VALU' in lk else '') + \
            ('+NG' + lk.split('NG=')[1][0] if 'NG=' in lk else '')
        kernels_seen[kern] = kernels_seen.get(kern, 0) + 1
        expect = ref + base if accumulate else ref
        if rd.kind in 'iu':
            ok = np.array_equal(res, (expect.astype(np.int64) if expect.dtype.kind in 'iu' else expect).astype(rd))
        else:
            tol = (2e-6 if kind == 'split' else 1e-5) if rd in (np.float32, np.complex64) else 1e-12
            ok = bool(np.all(np.abs(res - expect) <= tol * (scale + 1)))
        if not ok:
            fails += 1
            print("MISMATCH", desc, handle.last_kernel(), float(np.abs(res - expect).max()), flush=True)
        handle.close()
    except Exception as e:
        fails += 1
        print("ERROR", desc, repr(e), flush=True)
print(f"{n_cases} cases, {fails} failures, {time.time() - t0:.1f} s; kernels: {kernels_seen}")
sys.exit(1 if fails else 0)
