"""Shifted masks with float64 results (16 float64 masks, 16 384 frames of 256x256 uint16): the f64 kernel with the image of
the shifted stack, per group of frames with the same shift (ltmi_apply_masks_shifted_host), against the per-frame kernel."""
import sys, os, numpy as np, torch, time
sys.path.insert(0, os.getcwd())
from libertem_amd import hip
n, sig, nm = 16384, (256, 256), 16
rng = np.random.default_rng(0)
masks = rng.random((nm, sig[0]*sig[1]))            # float64
h = hip.MaskHandle.dense(0, masks, np.float64)
tile = torch.randint(0, 4096, (n, sig[0]*sig[1]), device='cuda', dtype=torch.int32).to(torch.int16)
out = torch.zeros((n, nm), device='cuda', dtype=torch.float64)
def run(shifts):
    h.apply_shifted_host(tile.data_ptr(), np.uint16, n, sig[0]*sig[1], sig[0], sig[1], shifts, out.data_ptr(), nm, False)
for name, shifts in [('constant shift', np.tile(np.array([[3, -2]], dtype=np.int32), (n, 1))),
                     ('9 shifts', rng.integers(-1, 2, (n, 2)).astype(np.int32)),
                     ('25 shifts', rng.integers(-2, 3, (n, 2)).astype(np.int32)),
                     ('169 shifts (per-frame kernel)', rng.integers(-6, 7, (n, 2)).astype(np.int32))]:
    for _ in range(2): run(shifts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): run(shifts)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name:32s} {dt*1e3:8.2f} ms   {h.last_kernel()}")
h.apply(tile.data_ptr(), np.uint16, n, sig[0]*sig[1], out.data_ptr(), nm, False); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): h.apply(tile.data_ptr(), np.uint16, n, sig[0]*sig[1], out.data_ptr(), nm, False)
torch.cuda.synchronize(); print(f"{'unshifted':32s} {(time.perf_counter()-t0)/5*1e3:8.2f} ms   {h.last_kernel()}")
