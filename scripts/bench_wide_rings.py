"""ring stacks with 24 .. 128 wide bins on 256 x 256 frames: as CSR (what the container passes today above 32 columns) against dense"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip, masks as pm
n = 16384
for dt, tdt in ((np.uint16, torch.int16), (np.float32, torch.float32)):
    tile = torch.randint(0, 4096, (n, 65536), device='cuda', dtype=torch.int16) if dt == np.uint16 else torch.rand((n, 65536), device='cuda')
    for nb in (24, 48, 64, 96, 128):
        rings = pm.radial_bins(128, 128, 256, 256, n_bins=nb, use_sparse=True, dtype=np.float32)
        csr = rings.to_px_by_masks(dtype=np.float32)
        dense = np.ascontiguousarray(np.asarray(csr.todense()).T.astype(np.float32))
        out = torch.zeros((n, nb), device='cuda')
        line = f"{np.dtype(dt).name:8s} {nb:4d} bins:"
        for kind in ('csr', 'dense'):
            h = hip.MaskHandle.csr(0, csr, np.float32) if kind == 'csr' else hip.MaskHandle.dense(0, dense, np.float32)
            h.set_sig_shape(256, 256)
            for _ in range(2):
                h.apply(tile.data_ptr(), np.dtype(dt), n, 65536, out.data_ptr(), nb, False)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                h.apply(tile.data_ptr(), np.dtype(dt), n, 65536, out.data_ptr(), nb, False)
            e1.record(); e1.synchronize()
            line += f"  {kind}: {e0.elapsed_time(e1) / 5:.3f} ms [{h.last_kernel()[:36]}]"
            h.close()
        print(line, flush=True)
