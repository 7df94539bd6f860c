"""ltmi_byteswap on 8 GiB in HBM (in place): GB/s read + written."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip
n = 4 << 30
t = torch.zeros(n, dtype=torch.int16, device='cuda')
for item in (2, 4, 8):
    n_items = n * 2 // item
    for _ in range(2):
        hip.byteswap(0, t.data_ptr(), t.data_ptr(), item, n_items)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in evs:
        a.record(); hip.byteswap(0, t.data_ptr(), t.data_ptr(), item, n_items); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[2]
    print(f"ltmi_byteswap itemsize {item}: {ms:.3f} ms for {n * 2 / 2**30:.0f} GiB in place = "
          f"{2 * n * 2 / ms / 1e6:.0f} GB/s read+write ({2 * n * 2 / ms / 1e6 / 80:.1f} % of 8 TB/s)")
