import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from libertem_amd import hip
rng = np.random.default_rng(0)
def run(name, dt, tdt, n, npx, ncols, fold_shape=None):
    masks = (rng.random((ncols, npx)) - 0.2).astype(np.float32)
    h = hip.MaskHandle.dense(0, masks, np.float32)
    if dt == np.float32: t = torch.rand((n, npx), device='cuda')
    else: t = torch.randint(0, 4000, (n, npx), device='cuda', dtype=tdt)
    out = torch.zeros((n, ncols), device='cuda')
    for _ in range(3): h.apply(t.data_ptr(), dt, n, npx, out.data_ptr(), ncols, False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): h.apply(t.data_ptr(), dt, n, npx, out.data_ptr(), ncols, False)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"{name:34s} {ms:7.3f} ms  {n * npx * np.dtype(dt).itemsize / ms / 1e6 / 8000:.3f} of HBM  {h.last_kernel()[:70]}")
    h.close(); del t, out
for ncols in (1, 3, 4, 5, 16, 17, 20, 32, 33, 48, 50, 64):
    run(f"f32 512x512 x {ncols} cols, 8192 fr", np.float32, None, 8192, 512 * 512, ncols)
for ncols in (1, 3, 4, 5, 17, 20, 33, 50):
    run(f"u16 256x256 x {ncols} cols, 32768 fr", np.uint16, torch.int16, 32768, 256 * 256, ncols)
