"""Error of the blocked sparse kernel against float64 for several data ranges (which byte path loses precision?)."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip, masks as pm
rings = pm.radial_bins(128, 128, 256, 256, n_bins=1024, use_sparse=True, dtype=np.float32)
csr = rings.to_px_by_masks(dtype=np.float32)
h = hip.MaskHandle.csr(0, csr, np.float32)
dense = csr.toarray().astype(np.float64)
rng = np.random.default_rng(0)
for name, lo, hi_ in (('[0,256)', 0, 256), ('[0,4096)', 0, 4096), ('[0,65536)', 0, 65536), ('multiples of 256', 0, 16)):
    d = rng.integers(lo, hi_, (64, 65536)).astype(np.uint16)
    if name.startswith('mult'):
        d = (d * 256).astype(np.uint16)
    t = torch.from_numpy(d.view(np.int16)).cuda()
    out = torch.zeros((64, 1024), device='cuda')
    h.apply(t.data_ptr(), np.uint16, 64, 65536, out.data_ptr(), 1024, False)
    torch.cuda.synchronize()
    ref = d.astype(np.float64) @ dense
    got = out.cpu().numpy()
    err = np.abs(got - ref)
    big = ref > 1e-3 * ref.max()
    print(f"{name:18s} {h.last_kernel().split(' ')[0]}  max err / max ref {err.max() / ref.max():.2e}   "
          f"max elementwise rel (ref > 1e-3 max) {np.max(err[big] / ref[big]):.2e}   mean signed rel {np.mean((got[big] - ref[big]) / ref[big]):+.2e}")
