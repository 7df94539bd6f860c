import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libertem_amd import hip
frames, sig, nm = int(os.environ.get('FRAMES', 8192)), 1024, int(os.environ.get('NM', 25))
n_px = sig * sig
tile = torch.rand((frames, n_px), device='cuda', dtype=torch.float32)
rng = np.random.default_rng(2)
masks = (rng.random((nm, n_px)) + 1j * rng.random((nm, n_px))).astype(np.complex64)
h = hip.MaskHandle.dense(0, masks, np.complex64)
out = torch.zeros((frames, nm), device='cuda', dtype=torch.complex64)
variants = [dict(mt=0, waves=int(os.environ.get('TUNE', 0)), ksplit=0)] if os.environ.get('ONLY_DEFAULT') else [dict(mt=0, waves=0, ksplit=0), dict(mt=1, waves=4, ksplit=4), dict(mt=1, waves=4, ksplit=8),
          dict(mt=2, waves=4, ksplit=4), dict(mt=2, waves=4, ksplit=8)]   # (8-wave k_dense_mfma: uint16 tiles only)
for v in variants:
    h.set_tuning(**v)
    for _ in range(2):
        h.apply(tile.data_ptr(), np.float32, frames, n_px, out.data_ptr(), nm, False)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in evs:
        a.record(); h.apply(tile.data_ptr(), np.float32, frames, n_px, out.data_ptr(), nm, False); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[2]
    print(v, h.last_kernel(), f"{ms:.3f} ms  {frames/ms/1e3:.3f} Mframes/s  {frames*n_px*4/ms/1e6:.0f} GB/s  "
          f"{2*frames*n_px*64/ms/1e9:.1f} TFLOP/s(padded 64 cols)")

# correctness of whatever ran last: 64 frames against a float64 product on the device
idx = torch.arange(0, frames, max(1, frames // 64), device='cuda')[:64]
mt = torch.from_numpy(masks).to('cuda')
ref = tile[idx].to(torch.complex128) @ mt.to(torch.complex128).T
err = ((out[idx].to(torch.complex128) - ref).abs().max() / ref.abs().max()).item()
print(f"max rel err vs float64 on 64 frames: {err:.2e}")
