"""Timing model of a row-mirror fold of C5 (25 complex masks on 1024 x 1024 float32 frames) BEFORE the kernel
exists: the shipped 4-group kernel over 513 x 1024 pixels of every frame (ld = the whole frame) issues the matrix
instructions a fold kernel would (2 even + 2 odd groups over the folded pixels) but copies only half of the frame
bytes -- a lower bound of the fold kernel's matrix-pipe time next to the shipped 3-group + 2 VALU column kernel."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip
frames = int(os.environ.get('FRAMES', 8192))
n_full = 1024 * 1024
tile = torch.rand((frames, n_full), device='cuda', dtype=torch.float32)
rng = np.random.default_rng(2)
for name, nm, n_px in (('shipped C5 (25 complex, all pixels)', 25, n_full),
                       ('model: 32 complex = 4 groups, 513 rows', 32, 513 * 1024),
                       ('model: 16 complex = 2 groups, all pixels (same MFMA count)', 16, n_full)):
    masks = (rng.random((nm, n_px)) + 1j * rng.random((nm, n_px))).astype(np.complex64)
    h = hip.MaskHandle.dense(0, masks, np.complex64)
    out = torch.zeros((frames, nm), device='cuda', dtype=torch.complex64)
    for _ in range(2):
        h.apply(tile.data_ptr(), np.float32, frames, n_full, out.data_ptr(), nm, False)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in evs:
        a.record(); h.apply(tile.data_ptr(), np.float32, frames, n_full, out.data_ptr(), nm, False); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[2]
    print(f"{name}: {h.last_kernel()}  {ms:.3f} ms  bytes touched {frames * n_px * 4 / ms / 1e6:.0f} GB/s")
    h.close()
