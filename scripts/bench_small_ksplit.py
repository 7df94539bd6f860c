"""k-split sweep of the small C2 tiles (HIP events around 20 back-to-back launches)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip
masks = np.random.default_rng(2).random((16, 65536)).astype(np.float32)
h = hip.MaskHandle.dense(0, masks, np.float32)
tile = torch.randint(0, 4096, (8192, 65536), device='cuda', dtype=torch.int16)
out = torch.zeros((8192, 16), device='cuda')
for n, splits in ((1024, (0, 8, 16, 32, 64)), (2048, (0, 8, 16, 32)), (4096, (0, 4, 8, 12, 16, 24, 32)), (8192, (0, 2, 4, 8, 16))):
    for ks in splits:
        h.set_tuning(0, 30, ks)
        for _ in range(int(os.environ.get('PREHEAT', 3))):
            h.apply(tile.data_ptr(), np.uint16, n, 65536, out.data_ptr(), 16, False)
        if not os.environ.get('NOSYNC'):
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            h.apply(tile.data_ptr(), np.uint16, n, 65536, out.data_ptr(), 16, False)
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{n:5d} frames ksplit {ks:2d}: {h.last_kernel():62s} {us:7.1f} us  {n * 131136 / us / 1e6 / 8:.3f} of HBM")
