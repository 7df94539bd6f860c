"""one decode launch per format for rocprofv3 (kernel trace / PMC passes): raw 12-bit, 1-bit and quad 6-bit"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from libertem_amd import hip                                   # noqa: E402
from libertem_amd.common.hiparray import HipArray              # noqa: E402

for kind, bits, quad, (h, w), header, storage, n in [
        ('r', 12, False, (256, 256), 384, np.uint16, 16384),
        ('r', 1, False, (256, 256), 384, np.uint8, 57952),
        ('r', 6, True, (512, 512), 768, np.uint8, 8180)]:
    payload = h * w * {1: 1, 6: 8, 12: 16}[bits] // 8
    stride = header + payload
    raw = torch.randint(0, 256, (n * stride,), dtype=torch.uint8, device='cuda')
    out = HipArray.empty((n, h, w), storage, 0)
    for _ in range(5):
        hip.mib_decode(0, raw.data_ptr(), stride, header, kind, bits, quad, n, h, w, out.data_ptr(), storage)
    torch.cuda.synchronize()
    print(kind, bits, quad, n, 'algorithmic bytes per launch', n * (payload + h * w * np.dtype(storage).itemsize))
