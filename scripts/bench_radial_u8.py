"""radial Fourier (25 complex masks) on uint8 frames (6-bit / 1-bit Merlin data), folded (k_dense_fold16<h>) and not (NOFOLD=1)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from libertem_amd import hip, masks as pm
from libertem_amd.analysis.radialfourier import radial_mask_factory
for N, frames in ((256, 65536), (1024, 8192)):
    ro = pm.bounding_radius(N / 2, N / 2, N, N)
    flat = np.ascontiguousarray(radial_mask_factory(N, N, N / 2, N / 2, 0, ro, 1, 24, False)().reshape(25, -1))
    h = hip.MaskHandle.dense(0, flat, np.complex64)
    if not os.environ.get('NOFOLD'):
        h.set_sig_shape(N, N)
    tile = torch.randint(0, 64, (frames, N * N), device='cuda', dtype=torch.uint8)
    out = torch.zeros((frames, 25), device='cuda', dtype=torch.complex64)
    for _ in range(2):
        h.apply(tile.data_ptr(), np.uint8, frames, N * N, out.data_ptr(), 25, False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        h.apply(tile.data_ptr(), np.uint8, frames, N * N, out.data_ptr(), 25, False)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(N, h.last_kernel(), f"{ms:.3f} ms  {frames * N * N / ms / 1e6 / 8000:.3f} of HBM")
    h.close()
