"""small C2 tiles: does the power-of-two frame stride (128 KiB) cost bandwidth?  The same tiles with padded rows."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip
masks = np.random.default_rng(2).random((16, 65536)).astype(np.float32)
h = hip.MaskHandle.dense(0, masks, np.float32)
out = torch.zeros((65536, 16), device='cuda')
for pad in [int(x) for x in os.environ.get('PADS', '0,64,128,256,1024,4096').split(',')]:
    ld = 65536 + pad
    big = torch.randint(0, 4096, (65536, ld), device='cuda', dtype=torch.int16)
    for n in (1024, 2048, 4096, 8192, 65536):
        reps = 20 if n < 65536 else 5
        for _ in range(3):
            h.apply(big.data_ptr(), np.uint16, n, ld, out.data_ptr(), 16, False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            h.apply(big.data_ptr(), np.uint16, n, ld, out.data_ptr(), 16, False)
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        print(f"pad {pad:5d} px  {n:6d} frames  {h.last_kernel()[-22:]:22s}: {us:8.1f} us  {n * 131136 / us / 1e6 / 8:.3f} of HBM", flush=True)
    del big
