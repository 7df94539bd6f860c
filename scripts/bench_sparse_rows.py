"""Experiment: C4 ring stack through the row-list entry point with (a) the identity row list and (b) a
list that maps every result row to one of the first 64 frames (all workgroups read the same 8 MiB:
the frame copies are L2 hits).  Separates the cost of HBM latency in the frame copies from the rest of
k_bell_apply."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip, masks as pm

n = 16384
rings = pm.radial_bins(128, 128, 256, 256, n_bins=1024, use_sparse=True, dtype=np.float32)
csr = rings.to_px_by_masks(dtype=np.float32)
h = hip.MaskHandle.csr(0, csr, np.float32)
g = torch.Generator(device='cuda').manual_seed(1)
tile = torch.randint(0, 4096, (n, 65536), generator=g, device='cuda', dtype=torch.int32).to(torch.int16)
out = torch.zeros((n, 1024), device='cuda', dtype=torch.float32)
for name, rows in (('identity', torch.arange(n, dtype=torch.int32, device='cuda')),
                   ('64 hot frames', (torch.arange(n, dtype=torch.int32, device='cuda') % 64))):
    for _ in range(2):
        assert h.apply_rows(tile.data_ptr(), np.uint16, rows.data_ptr(), n, 65536, out.data_ptr(), 1024, False)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in evs:
        a.record()
        h.apply_rows(tile.data_ptr(), np.uint16, rows.data_ptr(), n, 65536, out.data_ptr(), 1024, False)
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    print(f"{name}: {h.last_kernel()}  median {ts[len(ts) // 2]:.3f} ms")
