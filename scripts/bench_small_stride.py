"""small C2 tiles: does the power-of-two frame stride (128 KiB) cost bandwidth?  The same tiles with padded rows."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip
N = int(os.environ.get('DET', 256))
NM = int(os.environ.get('NMASKS', 16))
DT = os.environ.get('DTYPE', 'uint16')
masks = np.random.default_rng(2).random((NM, N * N)).astype(np.float32)
h = hip.MaskHandle.dense(0, masks, np.float32)
out = torch.zeros((65536, NM), device='cuda')
NF = int(os.environ.get('NFRAMES', 65536))
SIZES = [int(x) for x in os.environ.get('SIZES', '1024,2048,4096,8192,65536').split(',')]
for pad in [int(x) for x in os.environ.get('PADS', '0,64,128,256,1024,4096').split(',')]:
    ld = N * N + pad
    if DT == 'float32':
        big = torch.rand((NF, ld), device='cuda')
    else:
        big = torch.randint(0, 4096, (NF, ld), device='cuda', dtype=torch.int16)
    for n in SIZES:
        reps = 20 if n < 65536 else 5
        for _ in range(3):
            h.apply(big.data_ptr(), np.dtype(DT), n, ld, out.data_ptr(), NM, False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            h.apply(big.data_ptr(), np.dtype(DT), n, ld, out.data_ptr(), NM, False)
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        print(f"pad {pad:5d} px  {n:6d} frames  {h.last_kernel()[-22:]:22s}: {us:8.1f} us  {n * (N * N * np.dtype(DT).itemsize + NM * 4) / us / 1e6 / 8:.3f} of HBM", flush=True)
    del big
