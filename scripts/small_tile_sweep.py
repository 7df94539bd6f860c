"""C2 launches of a few thousand frames (what a live feed produces): the k-split / tile-shape knobs of
ltmi_masks_set_tuning against the default choice.  HIP events around 30 back-to-back launches."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip
masks = np.random.default_rng(2).random((16, 65536)).astype(np.float32)
g = torch.Generator(device='cuda').manual_seed(3)
tile = torch.randint(0, 4096, (8192, 65536), generator=g, device='cuda', dtype=torch.int16)
out_t = torch.zeros((8192, 16), device='cuda', dtype=torch.float32)
for n in (1024, 2048, 4096):
    for mt, waves, ks in ((0, 0, 0), (0, 0, 8), (0, 0, 16), (0, 0, 32), (0, 0, 64), (1, 0, 0), (1, 0, 32), (1, 0, 64),
                          (2, 0, 0), (0, 4, 0), (0, 4, 32), (1, 4, 32), (1, 4, 64)):
        h = hip.MaskHandle.dense(0, masks, np.float32)
        try:
            h.set_tuning(mt=mt, waves=waves, ksplit=ks)
            for _ in range(3):
                h.apply(tile.data_ptr(), np.uint16, n, 65536, out_t.data_ptr(), 16, False)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                h.apply(tile.data_ptr(), np.uint16, n, 65536, out_t.data_ptr(), 16, False)
            e1.record(); e1.synchronize()
            us = e0.elapsed_time(e1) / 30 * 1e3
            print(f"n={n} mt={mt} waves={waves} ksplit={ks}: {us:6.1f} us = {n * 131136 / us / 1e6 / 8:.3f} of HBM  {h.last_kernel()}", flush=True)
        except Exception as e:
            print(f"n={n} mt={mt} waves={waves} ksplit={ks}: {str(e)[:80]}")
        h.close()
