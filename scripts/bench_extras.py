"""Stacks of 16 g + 1..4 columns: g MFMA groups + VALU columns against the padded-group kernel (tuning 33).
    python scripts/bench_extras.py"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip

rng = np.random.default_rng(0)
for dt, n_frames, n_px in ((np.uint16, 65536, 65536), (np.float32, 8192, 1 << 20)):
    tdt = torch.int16 if dt == np.uint16 else torch.float32
    tile = (torch.rand((n_frames, n_px), device='cuda') * 4000).to(tdt)
    for n_masks in (16, 18, 20, 32, 34, 36, 48, 50, 70, 100, 128):
        masks = rng.random((n_masks, n_px), dtype=np.float32)
        h = hip.MaskHandle.dense(0, masks, np.float32)
        out = torch.zeros((n_frames, n_masks), device='cuda')
        line = f"{np.dtype(dt).name} {n_frames}x{n_px} x {n_masks} masks:"
        for code, name in ((30, 'dispatch'), (33, 'padded')):
            h.set_tuning(0, code, 0)
            for _ in range(2):
                h.apply(tile.data_ptr(), np.dtype(dt), n_frames, n_px, out.data_ptr(), n_masks, False)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
            for a, b in evs:
                a.record(); h.apply(tile.data_ptr(), np.dtype(dt), n_frames, n_px, out.data_ptr(), n_masks, False); b.record()
            torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(b) for a, b in evs)[2]
            line += f"  {name} {ms:.3f} ms [{h.last_kernel().split(' grid')[0]}]"
        print(line)
        h.close()
