"""Stress / determinism check of the dense kernels at sizes where the DMA ring runs for thousands of
sub-chunks per workgroup: every launch must reproduce the first one bit for bit, and agree with a
float64 product computed on the GPU by torch.  `python scripts/stress_dense.py [reps]`"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
g = torch.Generator(device='cuda').manual_seed(3)
bad = 0
for (nf, n_px, tdt, n_masks, mdt) in [
        (8192, 65536, 'uint16', 16, 'float32'), (8192, 65536, 'uint16', 3, 'float32'),
        (8192, 65536, 'uint8', 16, 'float32'), (4096, 65536, 'uint16', 52, 'float32'),
        (4096, 65536, 'int32', 16, 'float32'), (4096, 65536, 'float64', 20, 'float64'),
        (1024, 1048576, 'float32', 25, 'complex64'), (4099, 515 * 515, 'uint16', 16, 'float32'),
        (8192, 65536, 'uint16', 64, 'float32'), (300, 262144, 'int16', 33, 'float32')]:
    dt, md = np.dtype(tdt), np.dtype(mdt)
    if dt.kind in 'iu':
        tt = {1: torch.uint8, 2: torch.int16, 4: torch.int32}[dt.itemsize]
        tile = torch.randint(0, 200 if dt.itemsize == 1 else 4096, (nf, n_px), generator=g,
                             device='cuda', dtype=torch.int32).to(tt)
    else:
        tile = torch.rand((nf, n_px), generator=g, device='cuda',
                          dtype=torch.float64 if dt.itemsize == 8 else torch.float32)
    rng = np.random.default_rng(n_masks)
    masks = rng.random((n_masks, n_px)) - 0.2
    if md.kind == 'c':
        masks = masks + 1j * (rng.random((n_masks, n_px)) - 0.5)
    masks = masks.astype(md)
    rd = np.result_type(dt, md)
    h = hip.MaskHandle.dense(0, masks, rd)
    tout = {'float32': torch.float32, 'float64': torch.float64, 'complex64': torch.complex64,
            'complex128': torch.complex128}[rd.name]
    first = None
    for r in range(reps):
        out = torch.full((nf, n_masks), 7, device='cuda', dtype=tout)
        h.apply(tile.data_ptr(), dt, nf, n_px, out.data_ptr(), n_masks, False)
        torch.cuda.synchronize()
        if first is None:
            first = out
        elif not torch.equal(first, out):
            bad += 1
            print("NOT REPRODUCIBLE", tdt, n_masks, r, float((first - out).abs().max()))
    wide = torch.complex128 if rd.kind == 'c' else torch.float64
    mt = torch.from_numpy(masks).cuda().to(wide)
    ref = torch.empty((nf, n_masks), device='cuda', dtype=wide)
    for f0 in range(0, nf, 512):
        ref[f0:f0 + 512] = tile[f0:f0 + 512].to(wide) @ mt.T
    scale = float(ref.abs().max())
    err = float((first.to(wide) - ref).abs().max()) / scale
    tol = 2e-6 if rd in (np.float32, np.complex64) else 1e-13
    ok = err < tol
    bad += 0 if ok else 1
    print(f"{tdt:8s} {nf} x {n_px}, {n_masks} {mdt} masks: {h.last_kernel()[:58]:58s} max err / max |ref| "
          f"{err:.1e} {'ok' if ok else 'TOO LARGE'}; {reps} launches identical", flush=True)
    h.close()
    del tile, ref, mt
print("failures:", bad)
sys.exit(1 if bad else 0)
