"""small C2 tiles: the same tile 20 times (what bench.py's small_tiles does) against 20 different tiles of a 16 GiB buffer"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip
masks = np.random.default_rng(2).random((16, 65536)).astype(np.float32)
h = hip.MaskHandle.dense(0, masks, np.float32)
big = torch.randint(0, 4096, (131072, 65536), device='cuda', dtype=torch.int16)      # 16 GiB
out = torch.zeros((131072, 16), device='cuda')
for n in (1024, 2048, 4096, 8192):
    for mode in ('same tile', 'rotating tiles'):
        offs = [0] * 20 if mode == 'same tile' else [(i * 6151 * 8) % (131072 - n) // 128 * 128 for i in range(20)]
        for o in offs[:3]:
            h.apply(big[o:].data_ptr(), np.uint16, n, 65536, out[o:].data_ptr(), 16, False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for o in offs:
            h.apply(big[o:].data_ptr(), np.uint16, n, 65536, out[o:].data_ptr(), 16, False)
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{n:5d} frames, {mode:15s}: {us:7.1f} us  {n * 131136 / us / 1e6 / 8:.3f} of HBM")
