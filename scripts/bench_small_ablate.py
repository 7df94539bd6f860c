import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip
masks = np.random.default_rng(2).random((16, 65536)).astype(np.float32)
h = hip.MaskHandle.dense(0, masks, np.float32)
tile = torch.randint(0, 4096, (16384, 65536), device='cuda', dtype=torch.int16)
out = torch.zeros((16384, 16), device='cuda')
for n in (1024, 2048, 4096, 8192, 16384):
    for code, name in ((30, 'full'), (31, 'no DMA'), (32, 'no MFMA')):
        h.set_tuning(0, code, 0)
        for _ in range(5):
            h.apply(tile.data_ptr(), np.uint16, n, 65536, out.data_ptr(), 16, False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            h.apply(tile.data_ptr(), np.uint16, n, 65536, out.data_ptr(), 16, False)
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{n:6d} {name:8s} {h.last_kernel():60s} {us:7.1f} us")
