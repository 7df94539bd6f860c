"""Dense stacks whose result dtype is not float32/complex64 (float64 for int32 / float64 data,
integer masks with preferred_dtype=int): the correctness kernels, kernel level."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip

frames, n_px, n_masks = 4096, 65536, 16
rng = np.random.default_rng(0)
for tname, tdt, rdt in (('int32', torch.int32, np.float64), ('float64', torch.float64, np.float64),
                        ('uint16 x int32 masks', torch.int16, np.int32)):
    if tdt == torch.float64:
        tile = torch.rand((frames, n_px), device='cuda', dtype=tdt)
        ndt = np.float64
    elif tdt == torch.int32:
        tile = torch.randint(0, 100000, (frames, n_px), device='cuda', dtype=tdt)
        ndt = np.int32
    else:
        tile = torch.randint(0, 4096, (frames, n_px), device='cuda', dtype=torch.int32).to(torch.int16)
        ndt = np.uint16
    masks = rng.integers(0, 2, (n_masks, n_px)).astype(rdt) if np.dtype(rdt).kind == 'i' \
        else rng.random((n_masks, n_px)).astype(rdt)
    h = hip.MaskHandle.dense(0, masks, rdt)
    out = torch.zeros((frames, n_masks), device='cuda',
                      dtype=torch.float64 if rdt == np.float64 else torch.int32)
    for _ in range(2):
        h.apply(tile.data_ptr(), ndt, frames, n_px, out.data_ptr(), n_masks, False)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in evs:
        a.record(); h.apply(tile.data_ptr(), ndt, frames, n_px, out.data_ptr(), n_masks, False); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[2]
    nbytes = frames * n_px * tile.element_size()
    print(f"{tname:22s} {h.last_kernel():50s} {ms:8.3f} ms  {frames / ms / 1e3:7.3f} Mframes/s  "
          f"{nbytes / ms / 1e6:6.0f} GB/s ({nbytes / ms / 1e6 / 8000:.3f} of HBM peak)")
    h.close()
