import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, scipy.sparse as sp
from libertem_amd import hip, masks as M
rings = M.radial_bins(128, 128, 256, 256, n_bins=1024, use_sparse=True, dtype=np.float32)
csr = rings.to_px_by_masks(dtype=np.float32) if hasattr(rings, 'to_px_by_masks') else sp.csr_matrix(rings.T)
n = 16384
for name, dt, tdt in (('uint16', np.uint16, torch.int16), ('int16', np.int16, torch.int16), ('int8', np.int8, torch.int8), ('uint8', np.uint8, torch.uint8)):
    for f16 in ('1', '0'):
        os.environ['LTMI_BELL_F16'] = f16
        h = hip.MaskHandle.csr(0, sp.csr_matrix(csr), np.float32)
        t = torch.randint(-100 if 'u' not in name else 0, 100, (n, 65536), device='cuda', dtype=tdt)
        out = torch.zeros((n, 1024), device='cuda')
        for _ in range(3): h.apply(t.data_ptr(), dt, n, 65536, out.data_ptr(), 1024, False)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): h.apply(t.data_ptr(), dt, n, 65536, out.data_ptr(), 1024, False)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        byt = n * (65536 * np.dtype(dt).itemsize + 4096)
        print(f"{name:7s} LTMI_BELL_F16={f16}: {ms:.3f} ms = {byt / ms / 1e6 / 8000:.3f} of HBM  {h.last_kernel()[:60]}")
        h.close()
